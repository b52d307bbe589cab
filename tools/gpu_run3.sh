set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 python tools/diag_decrypt.py > gpurun_out/diag_decrypt.log 2>&1; cat gpurun_out/diag_decrypt.log
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_client.py -m gpu -q --durations=8 > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu3.log
tail -40 gpurun_out/pytest_gpu3.log
for b in 1 2 4; do HYDIA_TENSOR_BPP=$b timeout -k 10 300 python tools/prof_similarity.py 17 3 > gpurun_out/prof17_bpp$b.log 2>&1; tail -4 gpurun_out/prof17_bpp$b.log; done
