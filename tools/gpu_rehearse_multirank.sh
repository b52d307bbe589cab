# 2-rank rehearsal of bench.py's multi-rank control flow on ONE GPU (gloo gather through host memory)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HYDIA_BENCH_REHEARSE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --log2n 15 > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err; echo "exit $?"; cat gpurun_out/rehearse2.json | cut -c1-700; tail -5 gpurun_out/rehearse2.err
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --log2n 17 --no-cpu-baseline > gpurun_out/b17.json 2>gpurun_out/b17.err; cat gpurun_out/b17.json | cut -c1-300
# one-rank RCCL group: the real nccl init + gather + device-pointer path of the multi-rank bench
HYDIA_BENCH_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 2 --warmup 1 --log2n 17 --no-cpu-baseline > gpurun_out/force_dist.json 2> gpurun_out/force_dist.err; echo "exit $?"; cat gpurun_out/force_dist.json | cut -c1-300; tail -5 gpurun_out/force_dist.err
