"""Reduce tools/pmc_stalls.sh: per kernel (k_conv3h / k_conv3p instance), the mean of every counter over its last 10 launches, and the
ratios that answer "what is the kernel waiting for".   python tools/pmc_stalls.py gpurun_out/r04/stalls_TAG"""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [per dispatch]
dur = defaultdict(list)
for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_conv3[hp]<[^>]*>", r["Kernel_Name"])
        if not m:
            continue
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[d] = (m.group(0) + f" grid={r['Grid_Size']}", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    by_k = defaultdict(list)
    for d in sorted(per):
        by_k[meta[d][0]].append(d)
    for k, ds in by_k.items():
        for d in ds[-10:]:
            for c, v in per[d].items():
                vals[k][c].append(v)
            dur[k].append(meta[d][1])
for k in sorted(vals):
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    us = sum(dur[k]) / len(dur[k])
    print(f"\n== {k}   {us:.1f} us under the profiler")
    for n in sorted(c):
        print(f"   {n:36s} {c[n]:16.0f}")
    g = c.get("GRBM_GUI_ACTIVE", 0) / 8.0            # cycles of the launch
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if g and wc:
        print(f"   -> clock {g / us / 1e3:.2f} GHz; wave-cycle shares: waitcnt/barrier {c['SQ_WAIT_ANY'] / wc:.2f}, issue-stalled {c['SQ_WAIT_INST_ANY'] / wc:.2f} "
              f"(of which LDS {c['SQ_WAIT_INST_LDS'] / wc:.2f}); busy issuing: VALU {c['SQ_ACTIVE_INST_VALU'] / wc:.2f} VMEM {c['SQ_ACTIVE_INST_VMEM'] / wc:.2f} LDS {c['SQ_ACTIVE_INST_LDS'] / wc:.2f}")
    if g and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print(f"   -> matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / g:.3f} of the launch, vector + matrix co-executing {c['SQ_VALU_MFMA_COEXEC_CYCLES'] / 1024 / g:.3f}")
    if g and "TA_TA_BUSY" in c:
        print(f"   -> TA busy {c['TA_TA_BUSY'] / 256 / g:.2f} of the launch per CU; address path stalled by TC {c['TA_ADDR_STALLED_BY_TC_CYCLES'] / 256 / g:.2f}, data by TC {c['TA_DATA_STALLED_BY_TC_CYCLES'] / 256 / g:.2f}; "
              f"{c['TA_TOTAL_WAVEFRONTS'] / 256:.0f} wave-instructions per CU")
    if "TCP_TCC_READ_REQ_LATENCY" in c and c.get("TCP_TCC_READ_REQ"):
        print(f"   -> L1 -> L2 read requests {c['TCP_TCC_READ_REQ']:.0f}, mean latency {c['TCP_TCC_READ_REQ_LATENCY'] / c['TCP_TCC_READ_REQ']:.0f} cycles")
    if c.get("TCC_REQ"):
        print(f"   -> L2 hit rate {c['TCC_HIT'] / max(1.0, c['TCC_HIT'] + c['TCC_MISS']):.2f}; tag stall cycles per request {c['TCC_TAG_STALL'] / c['TCC_REQ']:.2f}")
    if c.get("TCC_EA0_RDREQ"):
        print(f"   -> L2 -> fabric reads {c['TCC_EA0_RDREQ']:.0f}, mean latency {c['TCC_EA0_RDREQ_LEVEL'] / c['TCC_EA0_RDREQ']:.0f} L2 cycles")
