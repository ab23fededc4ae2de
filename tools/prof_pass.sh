export TMPDIR=/tmp; mkdir -p gpurun_out/r02c && cd /tmp && R=/root/repo && O=$R/gpurun_out/r02c
cd $R
python bench.py --steps 3 --warmup 1 > $O/bench_stdout.json 2> $O/bench_stderr.log && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions > $O/bench_under_rocprof.json 2> $O/rocprof_stats.log && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --roofline-only > $O/fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --roofline-only > $O/write.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/mfma -o m -- python3 bench.py --roofline-only > $O/mfma.log 2>&1
echo rc=$?; cat $O/bench_stdout.json; find $O -name '*.csv' | head -20
