set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "2 8" "2 16" "1 4" "1 8" "1 16" "2 2"; do set -- $cfg
HYDIA_TENSOR_BPP=$1 HYDIA_TENSOR_NW=$2 timeout -k 10 300 python tools/prof_similarity.py 20 3 computeSimilarity > gpurun_out/t13_$1_$2.log 2>&1; echo "bpp=$1 nw=$2: $(grep computeSimilarity gpurun_out/t13_$1_$2.log | cut -c1-170)"; done
