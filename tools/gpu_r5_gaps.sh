# round 5: kernel-trace timelines of one-block and eight-block queries: wall against the union of kernel intervals
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out
: > $R/gpurun_out/query_gaps.txt
for L in 14 17; do
  for lanes in 2 1; do
    export HYDIA_LANES=$lanes
    rm -rf $R/gpurun_out/kt_gaps
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_gaps -- python3 $R/tools/prof_query_gaps.py run $L > $R/gpurun_out/kt_gaps.log 2>&1 || { tail -5 $R/gpurun_out/kt_gaps.log; exit 1; }
    echo "== 2^$L vectors, HYDIA_LANES=$lanes" >> $R/gpurun_out/query_gaps.txt
    grep "query wall" $R/gpurun_out/kt_gaps.log | tail -3 >> $R/gpurun_out/query_gaps.txt
    python3 $R/tools/prof_query_gaps.py report $R/gpurun_out/kt_gaps >> $R/gpurun_out/query_gaps.txt
  done
done
rm -rf $R/gpurun_out/kt_gaps; cat $R/gpurun_out/query_gaps.txt
