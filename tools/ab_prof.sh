#!/bin/bash
# Same-box per-kernel A/B: rocprofv3 kernel statistics of bench.py for ab_r3/ (previous round) and the working tree.
export TMPDIR=/tmp
O=$PWD/gpurun_out/abprof; rm -rf $O; mkdir -p $O
for t in ab_r3 . ab_r3 .; do
  n=$(echo $t | tr -d './'); n=${n:-cur}; i=$(ls $O | grep -c "^$n")
  X=""; grep -q "no-other-configs" $t/bench.py && X="--no-other-configs"
  ( cd $t && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${n}_$i -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions $X > $O/${n}_$i.json 2> $O/${n}_$i.log )
done
find $O -name "*kernel_stats.csv"
