#!/bin/bash
# Same-box A/B of bench.py: the tree in ab_r3/ (a `git archive` of the previous round's HEAD, built in place) against the working
# tree, alternating runs.   tools/ab_bench.sh [rounds] [steps]     -> gpurun_out/ab_bench.log
R=${1:-3}; S=${2:-6}
mkdir -p gpurun_out
: > gpurun_out/ab_bench.log
for i in $(seq 1 $R); do
  for t in ${AB_TREES:-ab_r3 .}; do
    X=""; grep -q "no-other-configs" $t/bench.py && X="--no-other-configs"
    ( cd $t && python bench.py --steps $S --warmup 2 --no-cpu-baseline --no-other-precisions $X 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$t', 'samples/s', d['value'], 'launch_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" ) >> gpurun_out/ab_bench.log
  done
done
cat gpurun_out/ab_bench.log
