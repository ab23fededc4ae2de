"""Diagnostic timings of ds_conv2d on the GPU box (not part of the product or the tests)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def case(B, Cin, Cout, S, ks=3, mode=0, res=False, reps=20):
    for prec in (("fp32", "bf16x6", "fp16x3") if ks == 3 else ("fp32",)):
        _case(B, Cin, Cout, S, ks, mode, res, reps, prec)


def _case(B, Cin, Cout, S, ks, mode, res, reps, prec):
    Sin = S * 2 if mode == 1 else (S // 2 if mode == 2 else S)
    x = torch.randn(B, Cin, Sin, Sin, device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    wp = ops.pack_conv(w, prec)
    out = torch.empty(B, Cout, S, S, device=dev)
    r = torch.randn(B, Cout, S, S, device=dev) if res else None
    ms = timeit(lambda: ops.conv(x, wp, bias=b, res1=r, load_mode=mode, out=out), reps)
    fl = 2.0 * B * Cout * Cin * ks * ks * S * S
    print(f"{prec:7s} B={B} Cin={Cin} Cout={Cout} S={S} ks={ks} mode={mode} res={res}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF/s", flush=True)

if __name__ == "__main__":
    cases = [(64, 64, 64, 128), (64, 64, 64, 128, 3, 0, True), (64, 512, 64, 128), (64, 128, 128, 64), (64, 256, 256, 32),
             (64, 1024, 256, 32), (64, 64, 128, 64, 3, 1), (64, 128, 64, 128, 3, 2), (64, 256, 768, 32, 1), (64, 1, 64, 128), (64, 64, 1, 128),
             (16, 64, 64, 128), (64, 64, 64, 32)]
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        cases = cases[:5]
    for args in cases:
        case(*args)
