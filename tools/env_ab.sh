#!/bin/bash
# Same-box A/B of one environment switch of the library:  tools/env_ab.sh VAR "v1 v2 ..." [rounds] [steps]  -> gpurun_out/env_ab_VAR.log
V=$1; VALS=$2; R=${3:-3}; S=${4:-6}
mkdir -p gpurun_out; L=gpurun_out/env_ab_$V.log; : > $L
for i in $(seq 1 $R); do
  for v in $VALS; do
    env $V=$v python bench.py --steps $S --warmup 2 --no-cpu-baseline --no-other-precisions --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$V=$v', 'samples/s', d['value'], 'launch_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])" >> $L
  done
done
cat $L
