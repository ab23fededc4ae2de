"""The 16-byte-load form of the output-layer convolution (ds_conv2d_direct on whole 64-column tiles) against the general kernel
(DS_DIRECT_VEC=0, child process): bit for bit, plus launch times.      python tools/direct_vec_check.py"""
import os
import subprocess
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsci_amd import ops

dev = torch.device("cuda:0")
# (B, Cin, Cout, H, W, circular)
CASES = [(64, 64, 1, 128, 128, 0), (16, 64, 4, 256, 256, 0), (32, 128, 3, 256, 256, 0), (3, 8, 2, 20, 64, 1), (2, 24, 1, 50, 192, 0), (2, 16, 4, 16, 64, 1)]


def run():
    res = []
    for ci, (B, Cin, Cout, H, W, circ) in enumerate(CASES):
        g = torch.Generator().manual_seed(40 + ci)
        x = torch.randn(B, Cin, H, W, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        f = lambda: ops.conv_direct(x, w, b, circular=bool(circ))       # noqa: E731
        y = f()
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        xp = torch.nn.functional.pad(x[:1].double(), (1, 1, 1, 1), mode="circular" if circ else "constant")
        want = torch.nn.functional.conv2d(xp, w.double(), b.double())
        rel = float((y[:1].double() - want).norm() / want.norm())
        res.append((y.cpu(), e0.elapsed_time(e1) / 50 * 1e3, rel))
    return res


if len(sys.argv) > 2 and sys.argv[1] == "--child":
    torch.save(run(), sys.argv[2])
    sys.exit(0)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "r.pt")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", p], env=dict(os.environ, DS_DIRECT_VEC="0"))
    ref = torch.load(p)
os.environ["DS_DIRECT_VEC"] = "1"
bad = 0
for case, (y, us, rel), (yr, usr, _) in zip(CASES, run(), ref):
    same = torch.equal(y, yr)
    bad += 0 if same and rel < 2e-6 else 1
    print(case, "identical" if same else "DIFFERENT", f"rel-L2 vs fp64 {rel:.1e}   general {usr:.1f} us   16-byte loads {us:.1f} us", flush=True)
print("direct_vec_check:", "ALL OK" if not bad else f"{bad} FAILED")
sys.exit(1 if bad else 0)
