# round 3: loop A alone under rocprofv3 — per-kernel ms and TB/s (byte ledger), for each env configuration given as arguments
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
export $cfg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rot -- python3 $R/tools/prof_rotate.py 5 > $R/gpurun_out/prof_rot.log 2>&1 || { tail -5 $R/gpurun_out/prof_rot.log; exit 1; }
unset ${cfg%%=*}
tail -1 $R/gpurun_out/prof_rot.log
f=$(find $R/gpurun_out/prof_rot -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_rot_${cfg%%=*}.csv; rm -rf $R/gpurun_out/prof_rot
echo "== $cfg"; python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_rot_${cfg%%=*}.csv $R/gpurun_out/ledger_rot.json | head -14
done
