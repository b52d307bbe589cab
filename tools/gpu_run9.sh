set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "1 0" "2 0" "4 0" "8 0" "4 1" "1 1"; do set -- $cfg; nw=$1; nox=$2; if [ "$nox" = "1" ]; then export HYDIA_TENSOR_NOXCD=1; else unset HYDIA_TENSOR_NOXCD; fi
HYDIA_TENSOR_NW=$nw timeout -k 10 300 python tools/prof_similarity.py 20 3 computeSimilarity > gpurun_out/t9_nw${nw}_nox${nox}.log 2>&1; echo "nw=$nw noxcd=$nox: $(grep computeSimilarity gpurun_out/t9_nw${nw}_nox${nox}.log | cut -c1-200)"; done
