set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu12.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu12.log; tail -3 gpurun_out/pytest_gpu12.log
grep -q "pytest exit 0" gpurun_out/pytest_gpu12.log || exit 1
for cfg in "4 4 0" "2 4 0" "4 1 0" "2 1 0" "4 4 1" "2 4 1"; do set -- $cfg; unset HYDIA_DB_UNPACKED; [ "$3" = "1" ] && export HYDIA_DB_UNPACKED=1
HYDIA_TENSOR_BPP=$1 HYDIA_TENSOR_NW=$2 timeout -k 10 300 python tools/prof_similarity.py 20 3 computeSimilarity > gpurun_out/t12_$1_$2_$3.log 2>&1; echo "bpp=$1 nw=$2 unpacked=$3: $(grep computeSimilarity gpurun_out/t12_$1_$2_$3.log | cut -c1-170)"; done
