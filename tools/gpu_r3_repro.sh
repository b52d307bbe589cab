cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 200 python tools/prof_rotate.py 2 > gpurun_out/repro.log 2>&1; echo "exit $?" >> gpurun_out/repro.log
tail -15 gpurun_out/repro.log
