set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1 0 0" "1 0 1" "4 0 1" "2 0 0"; do set -- $cfg; nw=$1; nox=$2; nont=$3; unset HYDIA_TENSOR_NOXCD HYDIA_TENSOR_NONT; [ "$nox" = "1" ] && export HYDIA_TENSOR_NOXCD=1; [ "$nont" = "1" ] && export HYDIA_TENSOR_NONT=1
HYDIA_TENSOR_NW=$nw timeout -k 10 300 python tools/prof_similarity.py 20 3 computeSimilarity > gpurun_out/t10_${nw}_${nox}_${nont}.log 2>&1; echo "nw=$nw noxcd=$nox nont=$nont: $(grep computeSimilarity gpurun_out/t10_${nw}_${nox}_${nont}.log | cut -c1-200)"; done
