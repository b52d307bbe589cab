# rocprofv3 evidence for bench.py's default workload: kernel-trace stats, then HBM traffic counters in separate passes
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_stats.log 2>&1
grep '"metric"' $R/gpurun_out/prof_bench_stats.log | cut -c1-300
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_bench_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_bench_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_write.log 2>&1
cd $R
# keep only what is needed from the (large) counter CSVs: rows of the tensor kernel
for d in fetch write; do f=$(find gpurun_out/prof_bench_$d -name "*counter_collection.csv" | head -1); echo $f; head -1 $f > gpurun_out/pmc_${d}_tensor.csv; grep k_hydia_tensor $f >> gpurun_out/pmc_${d}_tensor.csv; wc -l gpurun_out/pmc_${d}_tensor.csv; rm -rf gpurun_out/prof_bench_$d; done
rm -f gpurun_out/prof_bench_stats/*/*kernel_trace.csv
ls -la gpurun_out/prof_bench_stats/*
