# round 5: the bench lines of the FINAL tree (after collect_profiles refreshed tensor_traffic.json / scaling_model.json): headline + two-rank rehearsal
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { tail -5 gpurun_out/bench_final.err; exit 1; }
HYDIA_BENCH_REHEARSE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --total-log2n 17 --log2n 16 --no-cpu-baseline > gpurun_out/bench_rehearse2.json 2> gpurun_out/bench_rehearse2.err || { tail -5 gpurun_out/bench_rehearse2.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_final.json')); r = d['roofline']
print(round(d['value']), round(d['ms_per_step'], 3), 'frac', round(r['frac'], 4), 'traffic', r['traffic'], 'stale', (r['traffic_meta'] or {}).get('stale'), 'ceiling', r['stream_ceiling'], round(r['vs_measured_stream_ceiling'], 3), 'cpu', round(d['cpu_baseline']['value']))
x = json.load(open('gpurun_out/bench_rehearse2.json')); print(x['comm_ms'], x['compute_ms'], x['loop_a_mode'], x['model'])
PY
