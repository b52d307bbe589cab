# round 3: bench.py's multi-rank control flow on the one-GPU box: two gloo ranks sharing GPU 0 (never a reported number), then a
# one-rank RCCL group (real nccl init / device staging), then the plain one-GPU line at 2^14
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export HYDIA_BENCH_REHEARSE=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --log2n 16 > gpurun_out/bench_rehearse_2ranks.json 2> gpurun_out/bench_rehearse_2ranks.err || { tail -20 gpurun_out/bench_rehearse_2ranks.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_rehearse_2ranks.json')); print(d['config']['workload']); print(d['scaling'], d['ms_per_step'], d['config']['loop_a_across_ranks'], d['config']['secondary'], d['config']['result_correct'])"
unset HYDIA_BENCH_REHEARSE
HYDIA_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 3 --warmup 1 --log2n 17 --no-cpu-baseline > gpurun_out/bench_one_rank_rccl.json 2> gpurun_out/bench_one_rank_rccl.err || { tail -20 gpurun_out/bench_one_rank_rccl.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_one_rank_rccl.json')); print(d['config']['workload']); print(d['scaling'], d['ms_per_step'], d['config']['result_correct'])"
