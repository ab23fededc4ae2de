"""Time the fp16x3 attention kernel at the bottleneck shapes of configs 2 and 5: staging in every workgroup
(ds_attention_h3) against pre-split K / V images + LDS-DMA (ds_attention_h3_ws); both must agree bit for bit."""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
for (B, E, L) in [(64, 256, 1024), (16, 256, 4096), (64, 128, 1024), (64, 64, 1024), (32, 512, 256)]:
    qkv = torch.randn(B, 3 * E, L, device=dev)
    res = {}
    for name, min_l in (("staged", 1 << 30), ("images", 0)):
        if E > 256 and name == "images":
            continue
        ops.ATTN_IMAGES_MIN_L = ops.ATTN_IMAGES_MIN_L_WIDE = min_l
        out = torch.empty(B, E, L, device=dev)
        nws = ops.attention_workspace_floats(B, E, L)
        ws = torch.empty(nws, device=dev) if nws else None
        rows = ops.amax_new(2 * B, qkv.device)                      # what the in-projection's epilogue leaves (max |q, k|, max |v|)
        ops.absmax_rows(qkv[:, :2 * E], out=rows[:B])
        ops.absmax_rows(qkv[:, 2 * E:], out=rows[B:])
        f = lambda: ops.attention(qkv, E, out=out, precision="fp16x3", workspace=ws, in_amax=rows)   # noqa: E731
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res[name] = out.clone()
        print(f"B{B} E{E} L{L} {name:7s}: {us:8.1f} us  {4.0*L*L*E*B/us/1e6:6.1f} TF-eq", flush=True)
    if len(res) == 2:
        print("   identical:", bool(torch.equal(res["staged"], res["images"])))
