mkdir -p gpurun_out/r04; L=gpurun_out/r04/ab_r3_vs_switches_off.log; : > $L
for i in 1 2 3; do
  for t in ab_r3 off on; do
    if [ $t = ab_r3 ]; then D=ab_r3; E=""; elif [ $t = off ]; then D=.; E="DS_CONV_PC=0 DS_CONV_VEC=0 DS_CONV_TWO_EARLY=0"; else D=.; E=""; fi
    X=""; grep -q "no-other-configs" $D/bench.py && X="--no-other-configs"
    ( cd $D && env $E python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-precisions $X 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$t', 'samples/s', d['value'], 'ms_per_step', d['ms_per_step'], 'launch_ms', d['roofline']['avg_launch_ms'])" ) >> $L
  done
done
cat $L
