"""Reduce the MFMA-busy PMC pass (profiles/README.md, round 2) to one row per launch of the dominant kernel.

    python tools/pmc_mfma_busy.py <counter_collection.csv> [--last 29] [--out per_launch.csv]

Pass: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python3 bench.py --roofline-only.  Per launch:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)   (share of elapsed cycles with the matrix pipe busy)
  eff_clock_ghz  = (GRBM_GUI_ACTIVE / 8) / duration
  wait_* / active_inst = SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (wave-cycle shares; SQ_WAVE_CYCLES
  counts in units of 4 cycles)."""
import argparse
import csv
import re
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("counters")
    ap.add_argument("--last", type=int, default=29)
    ap.add_argument("--out")
    a = ap.parse_args()
    by = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(a.counters)):
        if "k_conv3h" not in r["Kernel_Name"] and "k_conv3p" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        by[d][r["Counter_Name"]] = by[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[d] = (re.search(r"k_conv3[hp]<[^>]*>", r["Kernel_Name"]).group(0), int(r["Grid_Size"]),
                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ids = sorted(by)[-a.last:]
    rows = []
    for d in ids:
        c, (name, grid, us) = by[d], meta[d]
        gui = c["GRBM_GUI_ACTIVE"] / 8.0
        wc = c["SQ_WAVE_CYCLES"]
        rows.append(dict(dispatch=d, kernel=name, grid=grid, duration_us=round(us, 1),
                         mfma_busy_frac=round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / gui, 3),
                         eff_clock_ghz=round(gui / us / 1e3, 2),
                         wait_inst_any_frac=round(c["SQ_WAIT_INST_ANY"] / wc, 2), wait_any_frac=round(c["SQ_WAIT_ANY"] / wc, 2),
                         active_inst_frac=round(c["SQ_ACTIVE_INST_ANY"] / wc, 2)))
    keys = list(rows[0])
    if a.out:
        w = csv.DictWriter(open(a.out, "w"), keys)
        w.writeheader()
        w.writerows(rows)
    for r in rows:
        print("  ".join(f"{r[k]}" for k in keys))
    n = len(rows)
    print(f"mean over {n} launches: duration {sum(r['duration_us'] for r in rows) / n:.1f} us, "
          f"MFMA-busy {sum(r['mfma_busy_frac'] for r in rows) / n:.3f}, clock {sum(r['eff_clock_ghz'] for r in rows) / n:.2f} GHz")


if __name__ == "__main__":
    main()
