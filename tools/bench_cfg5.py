#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU's share: conditional PUNetG-64 (PorosityEmbedder, dict y),
4x256x256 fields, classifier-free guidance 2.0 (two network evaluations per score), 100-step Heun
(398 network calls), batch 16 per GPU; synthetic random-init weights.  Prints one JSON line.

    python tools/bench_cfg5.py [--batch 16 --size 256 --nsteps 100 --reps 2]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--nsteps", type=int, default=100)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--guidance", type=float, default=2.0)
    ap.add_argument("--precision", default="fp16x3")
    ap.add_argument("--no-batch-cfg", action="store_true", help="two evaluations per guided step instead of one of twice the batch")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fuse-norm", action="store_true")
    a = ap.parse_args()
    import diffsci_amd.models as M
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = M.PUNetG(M.PUNetGConfig(input_channels=4, output_channels=4),
                   conditional_embedding=M.nets.PorosityEmbedder(dembed=64))
    net.conv_precision = a.precision
    net.fuse_norm = not a.no_fuse_norm
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm(), conditional=True).to(dev).eval()
    module.batch_cfg = not a.no_batch_cfg
    module.use_graph = not a.no_graph
    wn = torch.randn(a.batch, 4, a.size, a.size, device=dev)
    y = {"porosity": torch.tensor([0.2], device=dev)}
    t0 = time.time()
    out = module.propagate_white_noise(wn, y=y, guidance=a.guidance, nsteps=a.nsteps)
    torch.cuda.synchronize()
    first = time.time() - t0
    ts = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.time()
        out = module.propagate_white_noise(wn, y=y, guidance=a.guidance, nsteps=a.nsteps)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
        print(f"run {len(ts)}: {ts[-1]:.3f} s", file=sys.stderr, flush=True)
    t = min(ts)
    ncalls = 2 * (2 * a.nsteps - 1)
    print(json.dumps(dict(workload=f"PUNetG-64 cond [{a.batch},4,{a.size},{a.size}] {a.nsteps}-step Heun CFG g={a.guidance}",
                          samples_per_s=round(a.batch / t, 3), s_per_run=round(t, 3), net_calls=ncalls,
                          ms_per_net_call=round(1e3 * t / ncalls, 3), first_call_s=round(first, 2),
                          finite=bool(torch.isfinite(out).all()), precision=a.precision,
                          peak_mem_GiB=round(torch.cuda.max_memory_allocated() / 2**30, 2))))


if __name__ == "__main__":
    main()
