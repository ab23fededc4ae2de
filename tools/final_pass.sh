# Round-end validation on one MI355X: the whole GPU suite, the driver's default bench line, configs 3 and 5, then the profiler passes.
export TMPDIR=/tmp; O=gpurun_out/r02d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout=600 > $O/gpu_tests.log 2>&1; echo rc=$? >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
grep -q "rc=0" $O/gpu_tests.log || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
python bench.py --steps 3 --warmup 1 > $O/bench_stdout.json 2> $O/bench_stderr.log || exit 1
python tools/bench_adm.py > $O/cfg3_stdout.json 2>/dev/null || exit 1
python tools/bench_cfg5.py > $O/cfg5_stdout.json 2>/dev/null || exit 1
cat $O/bench_stdout.json | cut -c1-400; tail -1 $O/cfg3_stdout.json; tail -1 $O/cfg5_stdout.json
