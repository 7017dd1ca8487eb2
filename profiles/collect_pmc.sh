set -e
R=${R:-r03}                      # round tag of the output files
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-eval"
rm -rf gpurun_out/${R}_trace gpurun_out/${R}_fetch gpurun_out/${R}_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -o run -- $CMD > gpurun_out/${R}_trace.out 2> gpurun_out/${R}_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_fetch -o run -- $CMD > gpurun_out/${R}_fetch.out 2> gpurun_out/${R}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}_write -o run -- $CMD > gpurun_out/${R}_write.out 2> gpurun_out/${R}_write.err
python3 profiles/membound_algo.py 5 "$CMD" > gpurun_out/${R}_algo.json 2> gpurun_out/${R}_algo.err
find gpurun_out/${R}_trace gpurun_out/${R}_fetch gpurun_out/${R}_write -name "*.csv" | head -20
T=$(find gpurun_out/${R}_trace -name "*kernel_trace.csv" | head -1); F=$(find gpurun_out/${R}_fetch -name "*counter_collection.csv" | head -1); W=$(find gpurun_out/${R}_write -name "*counter_collection.csv" | head -1)
python3 profiles/pmc_fold.py ${R} resnet50_256_b64_bf16 $T $F $W gpurun_out/${R}_algo.json | tee gpurun_out/${R}_fold.txt
cp $(find gpurun_out/${R}_trace -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats_eager.csv
# the memory-bound stages in isolation at the model's tensor sizes (incl. the soft-arg-max decode, which the training
# iteration itself does not call): same three passes
export MEMBOUND_EAGER=1
M="python3 profiles/membound_bench.py"
rm -rf gpurun_out/${R}m_trace gpurun_out/${R}m_fetch gpurun_out/${R}m_write
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}m_trace -o run -- $M > /dev/null 2> gpurun_out/${R}m_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}m_fetch -o run -- $M > /dev/null 2> gpurun_out/${R}m_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${R}m_write -o run -- $M > /dev/null 2> gpurun_out/${R}m_write.err
unset MEMBOUND_EAGER
python3 profiles/pmc_fold.py ${R} membound_stages_isolated $(find gpurun_out/${R}m_trace -name "*kernel_trace.csv" | head -1) $(find gpurun_out/${R}m_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/${R}m_write -name "*counter_collection.csv" | head -1) | tee gpurun_out/${R}m_fold.txt
python3 profiles/membound_bench.py 2>/dev/null | tee gpurun_out/${R}_membound_stages.txt
# kernel stats of the default (graph-replay) run, and the bench line itself
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_graph -o run -- python3 bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --no-eval > gpurun_out/${R}_bench_line_under_rocprof.json 2> gpurun_out/${R}_graph.err
cp $(find gpurun_out/${R}_graph -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats_graph.csv
gzip -c $F > gpurun_out/${R}_pmc_fetch_size_counter_collection.csv.gz; gzip -c $W > gpurun_out/${R}_pmc_write_size_counter_collection.csv.gz
cp profiles/${R}_pmc_traffic.json gpurun_out/${R}_pmc_traffic.json
