#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py diffsci_amd/csrc/ds_conv3h.hip [--all]     (default: only kernels with scratch or < 2 waves/SIMD)"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import build  # noqa: E402


def table(src):
    name = os.path.basename(src)
    cmd = [build.HIPCC] + build.FLAGS + build.EXTRA_FLAGS.get(name, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", err)[1:]:
        mangled = blk.split("\n")[0].strip()
        dem = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip() or mangled

        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        rows.append((dem, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    return rows


if __name__ == "__main__":
    show_all = "--all" in sys.argv
    for src in [a for a in sys.argv[1:] if not a.startswith("--")]:
        for dem, v, a, sc, occ, lds in table(src):
            if show_all or sc > 0:
                print(f"{os.path.basename(src)}: VGPR {v:3d} AGPR {a:3d} scratch {sc:4d} occ {occ} lds {lds:6d}  {dem[:150]}")
