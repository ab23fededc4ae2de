export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout=600 > gpurun_out/split3_tests.log 2>&1; echo rc=$? >> gpurun_out/split3_tests.log; tail -4 gpurun_out/split3_tests.log
grep -q "rc=0" gpurun_out/split3_tests.log || exit 1
L=$PWD/diffsci_amd/_lib
for i in 1 2; do
for v in old new; do
if [ $v = new ]; then unset DIFFSCI_HIP_LIB; else export DIFFSCI_HIP_LIB=$L/libdiffsci_hip_$v.so; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-precisions 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['value'], j['roofline']['avg_launch_ms'])"
done; done
for v in old new; do
if [ $v = new ]; then unset DIFFSCI_HIP_LIB; else export DIFFSCI_HIP_LIB=$L/libdiffsci_hip_$v.so; fi
echo cfg5 $v; python tools/bench_cfg5.py --nsteps 6 2>/dev/null | tail -1
echo attn $v; python tools/attn_time.py 2>/dev/null | tail -2
done
