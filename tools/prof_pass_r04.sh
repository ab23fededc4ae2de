# Round-4 profiler passes on one MI355X (same box, one call): the default bench line, kernel statistics of the same command under
# rocprofv3, the two HBM-traffic PMC passes and the MFMA-busy pass over `bench.py --roofline-only`.
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r04p; rm -rf $O; mkdir -p $O
python bench.py --steps 5 --warmup 1 > $O/bench_stdout.json 2> $O/bench_stderr.log && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-precisions --no-other-configs > $O/bench_under_rocprof.json 2> $O/rocprof_stats.log && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 bench.py --roofline-only > $O/fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 bench.py --roofline-only > $O/write.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/mfma -o m -- python3 bench.py --roofline-only > $O/mfma.log 2>&1
echo rc=$?; cut -c1-400 $O/bench_stdout.json; find $O -name '*.csv' | head -20
