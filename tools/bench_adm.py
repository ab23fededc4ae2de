#!/usr/bin/env python3
"""BASELINE.json configs[2]: ADM-128 (channel_expansion [1,2,4,4], concat skips, attention at 16^2),
3x256x256 fields, batch 32, 50-step sigma-churn (KarrasIntegrator) sampler, 1 GPU, synthetic
random-init weights.  Not the bench.py headline (that is configs[1]); prints one JSON line.

    python tools/bench_adm.py [--batch 32 --size 256 --nsteps 50 --reps 2]
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--nsteps", type=int, default=50)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--skip", default="concat")
    ap.add_argument("--integrator", default="karras")
    ap.add_argument("--precision", default="fp16x3")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fuse-norm", action="store_true")
    ap.add_argument("--fuse-max-cot", type=int, default=None)
    ap.add_argument("--no-up-parity", action="store_true")
    a = ap.parse_args()
    import diffsci_amd.models as M
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    c = a.channels
    net = M.ADM(M.ADMConfig(input_channels=3, output_channels=3, model_channels=c, time_embed_dim=c,
                            output_embed_dim=4 * c, channel_expansion=[1, 2, 4, 4], skip_integration_type=a.skip))
    net.conv_precision = a.precision
    net.fuse_norm = not a.no_fuse_norm
    net.upsample_parity = not a.no_up_parity
    if a.fuse_max_cot is not None:
        net.fuse_max_cot = a.fuse_max_cot
    nparam = sum(p.numel() for p in net.parameters())
    module = M.KarrasModule(net, M.KarrasModuleConfig.from_edm()).to(dev).eval()
    module.use_graph = not a.no_graph
    wn = torch.randn(a.batch, 3, a.size, a.size, device=dev)
    t0 = time.time()
    out = module.propagate_white_noise(wn, nsteps=a.nsteps, integrator=a.integrator)
    torch.cuda.synchronize()
    first = time.time() - t0
    ok = bool(torch.isfinite(out).all())
    ts = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.time()
        out = module.propagate_white_noise(wn, nsteps=a.nsteps, integrator=a.integrator)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
        print(f"run {len(ts)}: {ts[-1]:.3f} s", file=sys.stderr, flush=True)
    t = min(ts)
    nevals = 2 * a.nsteps - 1
    print(json.dumps(dict(workload=f"ADM-{c} {a.skip} [{a.batch},3,{a.size},{a.size}] {a.nsteps}-step {a.integrator}",
                          params_M=round(nparam / 1e6, 1), samples_per_s=round(a.batch / t, 3),
                          s_per_run=round(t, 3), ms_per_eval=round(1e3 * t / nevals, 2), first_call_s=round(first, 2),
                          finite=ok, workspace_GiB=round(net._ws.bytes / 2**30, 2), precision=a.precision,
                          peak_mem_GiB=round(torch.cuda.max_memory_allocated() / 2**30, 2))))


if __name__ == "__main__":
    main()
