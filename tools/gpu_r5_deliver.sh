# Round-5 deliverables on the GPU box (tests run separately: tools/gpu_r5_tests.sh): smoke, CLI, bench (+ CPU baseline), fixed-cost sizes,
# bench.py --gpus 2 started by bench.py itself (gloo rehearsal on the one GPU) and a one-rank RCCL group, the one-GPU components of the
# multi-GPU step model, rocprofv3 kernel stats + the two PMC passes of loop B on the bench command, per-kernel byte tables
# (2^20 / 2^17 / 2^14 / 2^10, 2^14 hoisted, loop A alone), SQ counters of every kernel of a 2^20 query.
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }; tail -1 gpurun_out/smoke.log
python tools/npz_to_dat.py tests/golden/dataset_2_10.npz /tmp/2_10.dat && (cd gpurun_out && rm -f latency.csv && touch latency.csv && HYDIA_SEED=7 timeout -k 10 300 $R/image_matching_amd/ImageMatching /tmp/2_10.dat 5 > cli_2_10.log 2>&1; echo "cli exit $?" >> cli_2_10.log; HYDIA_SEED=7 HYDIA_DEVICES=0,0,0 timeout -k 10 300 $R/image_matching_amd/ImageMatching /tmp/2_10.dat 5 > cli_2_10_sharded.log 2>&1; echo "cli exit $?" >> cli_2_10_sharded.log; tail -4 cli_2_10.log; cat latency.csv)
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail -5 gpurun_out/bench_final.err; exit 1; }; cut -c1-200 gpurun_out/bench_final.json
for l in 10 14 17; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --log2n $l --no-cpu-baseline > gpurun_out/bench_2p$l.json 2> gpurun_out/bench_2p$l.err || exit 1; python -c "
import json; d=json.load(open('gpurun_out/bench_2p$l.json')); print('2^$l:', round(d['value']), 'vec/s', round(d['ms_per_step'],2), 'ms/step', d['config']['result_correct'], 'step frac', round(d['roofline']['step']['frac'],3))"; done
HYDIA_BENCH_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $(python -c 'import socket; s=socket.socket(); s.bind(("127.0.0.1",0)); print(s.getsockname()[1])') bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_forcedist.json 2> gpurun_out/bench_forcedist.err || exit 1
HYDIA_BENCH_REHEARSE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --total-log2n 17 --log2n 16 --no-cpu-baseline > gpurun_out/bench_rehearse2.json 2> gpurun_out/bench_rehearse2.err || { tail -5 gpurun_out/bench_rehearse2.err; exit 1; }
timeout -k 10 300 python tools/prof_scaling_components.py 20 > gpurun_out/scaling_components.log 2>&1 && timeout -k 10 300 python tools/prof_scaling_components.py 17 >> gpurun_out/scaling_components.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_bench_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_bench_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_write.log 2>&1 || exit 1
for L in 20 17 14 10; do
  HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_led$L -- python3 $R/tools/prof_query_ledger.py $L 3 indexScenario > $R/gpurun_out/prof_led$L.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_led$L -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_q$L.csv; rm -rf $R/gpurun_out/prof_led$L
  python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_q$L.csv $R/gpurun_out/ledger_q$L.json > $R/gpurun_out/kernel_rooflines_q$L.txt
done
export HYDIA_MATVEC=hoisted
HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_led14h -- python3 $R/tools/prof_query_ledger.py 14 3 indexScenario > $R/gpurun_out/prof_led14h.log 2>&1 || exit 1
unset HYDIA_MATVEC
f=$(find $R/gpurun_out/prof_led14h -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_q14_hoisted.csv; rm -rf $R/gpurun_out/prof_led14h
python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_q14_hoisted.csv $R/gpurun_out/ledger_q14.json > $R/gpurun_out/kernel_rooflines_q14_hoisted.txt
HYDIA_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_led14 -- python3 $R/tools/prof_query_ledger.py 14 3 indexScenario > $R/gpurun_out/prof_led14.log 2>&1 || exit 1
rm -rf $R/gpurun_out/prof_led14
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rot -- python3 $R/tools/prof_rotate.py 5 > $R/gpurun_out/prof_rot.log 2>&1 || exit 1
f=$(find $R/gpurun_out/prof_rot -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/kernel_stats_rot.csv; rm -rf $R/gpurun_out/prof_rot
python3 $R/tools/kernel_rooflines.py $R/gpurun_out/kernel_stats_rot.csv $R/gpurun_out/ledger_rot.json > $R/gpurun_out/kernel_rooflines_rot.txt
cd $R
bash tools/gpu_r4_sq_tails.sh 20 > gpurun_out/sq_tails.log 2>&1 || { tail -5 gpurun_out/sq_tails.log; exit 1; }
(cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o stream_rate stream_rate.hip > /dev/null 2>&1; timeout -k 10 200 ./stream_rate > $R/gpurun_out/stream_rate.txt 2>&1 || true)
bash tools/gpu_r5_query_pmc.sh 20 > gpurun_out/query_pmc.log 2>&1 || { tail -5 gpurun_out/query_pmc.log; exit 1; }
bash tools/gpu_r5_loop_a_pmc.sh > gpurun_out/loop_a_pmc.log 2>&1 || { tail -5 gpurun_out/loop_a_pmc.log; exit 1; }
f=$(find gpurun_out/prof_bench_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/bench_kernel_stats.csv; rm -rf gpurun_out/prof_bench_stats
for d in fetch write; do f=$(find gpurun_out/prof_bench_$d -name "*counter_collection.csv" | head -1); head -1 $f > gpurun_out/pmc_${d}_tensor.csv; grep k_hydia_tensor $f >> gpurun_out/pmc_${d}_tensor.csv; rm -rf gpurun_out/prof_bench_$d; done
head -8 gpurun_out/kernel_rooflines_q20.txt; head -6 gpurun_out/kernel_rooflines_q14.txt; head -8 gpurun_out/kernel_rooflines_rot.txt
