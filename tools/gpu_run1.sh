set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch, image_matching_amd as im; print(torch.cuda.is_available()); cc = im.Context(); print('ctx ok with torch loaded', cc.N)" > gpurun_out/torch_coexist.log 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
timeout -k 10 300 python tools/prof_similarity.py 15 3 > gpurun_out/prof15.log 2>&1; tail -8 gpurun_out/prof15.log
timeout -k 10 300 python tools/prof_similarity.py 17 3 > gpurun_out/prof17.log 2>&1; tail -8 gpurun_out/prof17.log
