"""Debug: 3-D PUNetG eager vs captured forward_with_shifts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import diffsci_amd.models as M
from diffsci_amd import ops
from tests.golden_util import load

dev = torch.device("cuda:0")
v, sd = load("punetg8_3d")
net = M.PUNetG(M.PUNetGConfig(model_channels=8, dimension=3))
net.load_state_dict(sd, strict=True)
net = net.to(dev).eval()
x, t = v["x"].to(dev), v["t"].to(dev)


def variant(name, patch=None):
    undo = patch() if patch else None
    with torch.inference_mode():
        te = net.embed_time(t.reshape(-1).to(x), None)
        shifts = net.time_shifts(te)
        eager = net.forward_with_shifts(x.contiguous(), shifts, row=None).clone()
        out = torch.empty_like(eager)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net.forward_with_shifts(x.contiguous(), shifts, row=None, out=out)
            side.synchronize()
            with ops.Graph() as g:
                net.forward_with_shifts(x.contiguous(), shifts, row=None, out=out)
            out.zero_()
            g.launch()
            side.synchronize()
        print(name, "eager finite", bool(torch.isfinite(eager).all()), "graph finite", bool(torch.isfinite(out).all()),
              "equal", bool(torch.equal(eager, out)), "nodes", g.nodes, flush=True)
    if undo:
        undo()


variant("as-is")
orig = ops.conv3d_mfma


def p1():
    def f(*a, **k):
        k["in_amax"] = ops.NORMALISED
        return orig(*a, **k)
    ops.conv3d_mfma = f
    return lambda: setattr(ops, "conv3d_mfma", orig)


variant("conv3d_mfma NORMALISED", p1)
oa = net._attention


def p2():
    def f(att, xx, pk, ws, **k):
        B, E, Hh, Ww = xx.shape
        m = att.mhattn
        qkv = ops.conv(xx, pk[(id(att), "in")], bias=m.in_proj_bias, in_amax=ops.NORMALISED)
        o = ops.attention(qkv.view(B, 3 * E, Hh * Ww), E, precision="fp16x3", in_amax=ops.NORMALISED)
        return ops.conv(o.view(B, E, Hh, Ww), pk[(id(att), "out")], bias=m.out_proj.bias, res1=xx if net.config.attn_residual else None,
                        res2=k.get("res2"), in_amax=ops.NORMALISED)
    net._attention = f
    return lambda: setattr(net, "_attention", oa)


try:
    variant("attention NORMALISED (allocating)", p2)
except Exception as e:
    print("p2:", type(e).__name__, str(e)[:200])
    net._attention = oa
