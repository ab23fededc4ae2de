"""Per-path dynamic instruction counts of the fused-loader convolution ("precede with the per-path SQ_INSTS_VALU count").

  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU -d DIR -o e --output-format csv -- python3 tools/epilogue_insts.py
  python tools/epilogue_insts.py --reduce DIR/e_counter_collection.csv

Launches ds_conv2d_h3 with the fused norm + SiLU loader at config 2's level-0 / level-1 shapes, each in the VARIANTS below, REPS times
each in a fixed order; the reducer pairs the k_conv3h dispatches of the trace with that order and prints instructions per wave."""
import os
import sys

SHAPES = [(64, 64, 128), (64, 128, 64)]
VARIANTS = ["stats+residual+amax", "residual+amax", "stats+amax", "stats+residual", "bare"]
REPS = 2


def launch():
    sys.path.insert(0, os.getcwd())
    import torch
    from diffsci_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (B, C, S) in SHAPES:
        x = torch.randn(B, C, S, S, device=dev)
        res = torch.randn(B, C, S, S, device=dev)
        pw = ops.pack_conv(torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5), "fp16x3")
        bias, shift = torch.randn(C, device=dev), torch.randn(1, C, device=dev)
        tab = torch.zeros(B, ops.table_channels(C), 4, device=dev)
        tab[:, :C, 1] = 1.0
        ts = torch.zeros(B, C, ops.conv_tile_count(S, S), 4, device=dev)
        am = torch.zeros(B, dtype=torch.int32, device=dev)
        out = torch.empty_like(x)
        for name in VARIANTS:
            kw = {}
            if "stats" in name:
                kw["tile_stats"] = ts
            if "residual" in name:
                kw["res1"] = res
            if "amax" in name:
                kw["out_amax"] = am
            for _ in range(REPS):
                ops.conv(x, pw, bias=bias, shift=shift, prenorm=tab, out=out, **kw)
        torch.cuda.synchronize()


def reduce(path):
    import csv
    per = {}
    order = []
    for r in csv.DictReader(open(path)):
        if "k_conv3h" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        if d not in per:
            per[d] = {"waves": int(r["Grid_Size"]) // 64, "name": r["Kernel_Name"]}
            order.append(d)
        per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    expect = len(SHAPES) * len(VARIANTS) * REPS
    assert len(order) == expect, f"{len(order)} k_conv3h dispatches, expected {expect}"
    i = 0
    for (B, C, S) in SHAPES:
        base = None
        for name in VARIANTS:
            rows = [per[order[i + k]] for k in range(REPS)]
            i += REPS
            w = rows[0]["waves"]
            g = lambda c: sum(r.get(c, 0.0) for r in rows) / REPS / w      # noqa: E731
            valu, mfma, lds, salu = g("SQ_INSTS_VALU"), g("SQ_INSTS_MFMA"), g("SQ_INSTS_LDS"), g("SQ_INSTS_SALU")
            other = valu - mfma
            if base is None:
                base = other
            print(f"B={B} C={C} {S}x{S} {name:20s} waves {w:6d}  per wave: VALU {valu:7.0f} (MFMA {mfma:5.0f}, other {other:6.0f} = "
                  f"{other / max(mfma, 1):.2f} per MFMA, {other - base:+6.0f} vs full)  LDS {lds:5.0f}  SALU {salu:5.0f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
        reduce(sys.argv[2])
    else:
        launch()
