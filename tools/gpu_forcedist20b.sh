cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for K in 5 20; do
timeout -k 10 600 python bench.py --steps $K --warmup 2 --no-cpu-baseline --random-db > gpurun_out/fd_plain.json 2> gpurun_out/fd_plain.err; python -c "
import json; d=json.load(open('gpurun_out/fd_plain.json')); print('plain K=$K:', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_ms'],2))"
HYDIA_BENCH_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps $K --warmup 2 --no-cpu-baseline --random-db > gpurun_out/fd_dist.json 2> gpurun_out/fd_dist.err; python -c "
import json
for l in open('gpurun_out/fd_dist.json'):
    if l.startswith('{'):
        d=json.loads(l); print('dist K=$K:', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_ms'],2))"
done
