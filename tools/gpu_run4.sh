set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 python tools/diag_decrypt.py > gpurun_out/diag_decrypt.log 2>&1; cat gpurun_out/diag_decrypt.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_client.py -m gpu -q --durations=5 -x --deselect tests/test_gpu_client.py::test_encrypt_decrypt_bit_exact > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu4.log
tail -15 gpurun_out/pytest_gpu4.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu4.log || exit 1
timeout -k 10 300 python tools/prof_similarity.py 17 3 > gpurun_out/prof17_ntt15.log 2>&1; tail -4 gpurun_out/prof17_ntt15.log
HYDIA_NTT_GENERIC=1 timeout -k 10 300 python tools/prof_similarity.py 17 3 > gpurun_out/prof17_nttgen.log 2>&1; tail -4 gpurun_out/prof17_nttgen.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1b_17 -- python3 $GRAFT_REPO_ROOT/tools/prof_similarity.py 17 3 indexScenario > $GRAFT_REPO_ROOT/gpurun_out/rocprof17b.log 2>&1
