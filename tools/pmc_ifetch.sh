#!/bin/bash
# Instruction-fetch latency of the fused-loader convolutions (is the persistent kernel's 86-100 KB of code thrashing the 64 KB
# instruction cache two CUs share?): SQ_IFETCH_LEVEL / SQ_IFETCH = mean fetches in flight per fetch = its latency in SQ cycles.
export TMPDIR=/tmp; O=$PWD/gpurun_out/r04/ifetch_$1; rm -rf $O; mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/p -o c -- python3 tools/conv3p_time.py 10 > $O/p.log 2>&1
echo rc=$?
python3 - $O <<'PY'
import csv, glob, re, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(float)); name = {}
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_conv3[hp]<[^>]*>", r["Kernel_Name"])
        if m:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = m.group(0)
agg = defaultdict(lambda: defaultdict(list))
for d, c in per.items():
    for k, v in c.items(): agg[name[d]][k].append(v)
for k, c in agg.items():
    m = {n: sum(v[-10:]) / len(v[-10:]) for n, v in c.items()}
    print(k, " ifetch", int(m["SQ_IFETCH"]), " level/ifetch = %.1f" % (m["SQ_IFETCH_LEVEL"] / m["SQ_IFETCH"]), " ifetch per 1000 wave-cycles %.1f" % (1000 * m["SQ_IFETCH"] / m["SQ_WAVE_CYCLES"]),
          " VALU insts", int(m["SQ_INSTS_VALU"]), " SALU", int(m["SQ_INSTS_SALU"]))
PY
