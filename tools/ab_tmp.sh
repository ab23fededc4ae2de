export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adm.py tests/test_gpu_sampler.py -m gpu -x -q --timeout=600 > gpurun_out/convup16_tests.log 2>&1; echo rc=$? >> gpurun_out/convup16_tests.log; tail -5 gpurun_out/convup16_tests.log
grep -q "rc=0" gpurun_out/convup16_tests.log || exit 1
L=$PWD/diffsci_amd/_lib
for i in 1 2; do
for v in old mid pro new; do
if [ $v = new ]; then unset DIFFSCI_HIP_LIB; else export DIFFSCI_HIP_LIB=$L/libdiffsci_hip_$v.so; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-precisions 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['value'], j['roofline']['avg_launch_ms'])"
done; done
for v in old mid pro new; do
if [ $v = new ]; then unset DIFFSCI_HIP_LIB; else export DIFFSCI_HIP_LIB=$L/libdiffsci_hip_$v.so; fi
echo adm $v; python tools/bench_adm.py --nsteps 6 2>/dev/null | tail -1
done
