set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-eval"
rm -rf gpurun_out/r02_trace gpurun_out/r02_fetch gpurun_out/r02_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_trace -o run -- $CMD > gpurun_out/r02_trace.out 2> gpurun_out/r02_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_fetch -o run -- $CMD > gpurun_out/r02_fetch.out 2> gpurun_out/r02_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_write -o run -- $CMD > gpurun_out/r02_write.out 2> gpurun_out/r02_write.err
python3 profiles/membound_algo.py 5 "$CMD" > gpurun_out/r02_algo.json 2> gpurun_out/r02_algo.err
find gpurun_out/r02_trace gpurun_out/r02_fetch gpurun_out/r02_write -name "*.csv" | head -20
T=$(find gpurun_out/r02_trace -name "*kernel_trace.csv" | head -1); F=$(find gpurun_out/r02_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/r02_write -name "*counter_collection.csv" | head -1)
python3 profiles/pmc_fold.py r02 resnet50_256_b64_bf16 $T $F $W gpurun_out/r02_algo.json | tee gpurun_out/r02_fold.txt
cp $(find gpurun_out/r02_trace -name "*kernel_stats.csv" | head -1) gpurun_out/r02_bench_kernel_stats_eager.csv
# the memory-bound stages in isolation at the model's tensor sizes (incl. the soft-arg-max decode, which the training
# iteration itself does not call): same three passes
export MEMBOUND_EAGER=1
M="python3 profiles/membound_bench.py"
rm -rf gpurun_out/r02m_trace gpurun_out/r02m_fetch gpurun_out/r02m_write
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02m_trace -o run -- $M > /dev/null 2> gpurun_out/r02m_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02m_fetch -o run -- $M > /dev/null 2> gpurun_out/r02m_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02m_write -o run -- $M > /dev/null 2> gpurun_out/r02m_write.err
unset MEMBOUND_EAGER
python3 profiles/pmc_fold.py r02 membound_stages_isolated $(find gpurun_out/r02m_trace -name "*kernel_trace.csv" | head -1) $(find gpurun_out/r02m_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/r02m_write -name "*counter_collection.csv" | head -1) | tee gpurun_out/r02m_fold.txt
python3 profiles/membound_bench.py 2>/dev/null | tee gpurun_out/r02_membound_stages.txt
# kernel stats of the default (graph-replay) run, and the bench line itself
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_graph -o run -- python3 bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-eval > gpurun_out/r02_bench_line_under_rocprof.json 2> gpurun_out/r02_graph.err
cp $(find gpurun_out/r02_graph -name "*kernel_stats.csv" | head -1) gpurun_out/r02_bench_kernel_stats_graph.csv
gzip -c $F > gpurun_out/r02_pmc_fetch_size_counter_collection.csv.gz; gzip -c $W > gpurun_out/r02_pmc_write_size_counter_collection.csv.gz
cp profiles/r02_pmc_traffic.json gpurun_out/r02_pmc_traffic.json
