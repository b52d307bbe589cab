#!/usr/bin/env python3
"""Copy the outputs of tools/gpu_r4_deliver.sh (gpu_r3_deliver.sh in round 3) from gpurun_out/ into profiles/<round>/ under a version tag and refresh
profiles/tensor_traffic.json from the two PMC passes (tagged with the kernel source hash bench.py checks).
Usage: collect_profiles.py v1 [round directory, default r03]"""
import csv
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r03"
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", ROUND)
os.makedirs(P, exist_ok=True)
sys.path.insert(0, R)
from bench import kernel_sha  # noqa: E402


def cp(src, dst):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    else:
        print("missing", src)


cp("bench_final.json", "bench_2p20_%s.json" % tag)
for l in (10, 14, 17):
    cp("bench_2p%d.json" % l, "bench_2p%d_%s.json" % (l, tag))
cp("bench_kernel_stats.csv", "bench_2p20_kernel_stats_%s.csv" % tag)
cp("kernel_rooflines_q14_hoisted.txt", "kernel_rooflines_q14_hoisted_%s.txt" % tag)
cp("kernel_rooflines_rot.txt", "kernel_rooflines_loop_a_%s.txt" % tag)
cp("pmc_sq_rot_summary.txt", "sq_counters_loop_a_%s.txt" % tag)
cp("scaling_components_20.json", "scaling_components_2p20_%s.json" % tag)
cp("scaling_components_17.json", "scaling_components_2p17_%s.json" % tag)
cp("smoke.log", "smoke_%s.log" % tag)
cp("stream_rate.txt", "stream_rate.txt")
cp("tensor_check.txt", "loop_b_host_check.txt")
cp("sq_counters_tails.txt", "sq_counters_tails_raw_%s.txt" % tag)
cp("kernel_rooflines_q20.txt", "kernel_rooflines_q20_%s.txt" % tag)
for l in (20, 17, 14, 10):
    cp("kernel_rooflines_q%d.txt" % l, "kernel_rooflines_q%d_%s.txt" % (l, tag))
    cp("kernel_stats_q%d.csv" % l, "indexscenario_2p%d_queryonly_kernel_stats_%s.csv" % (l, tag))
    cp("ledger_q%d.json" % l, "byte_ledger_q%d_%s.json" % (l, tag))
cp("pmc_fetch_tensor.csv", "bench_2p20_pmc_fetch_tensor_%s.csv" % tag)
cp("pmc_write_tensor.csv", "bench_2p20_pmc_write_tensor_%s.csv" % tag)
cp("cli_2_10.log", "cli_ImageMatching_2_10.log")
cp("cli_2_10_sharded.log", "cli_ImageMatching_2_10_three_shards.log")
cp("latency.csv", "cli_latency.csv")
cp("pytest_gpu_final.log", "pytest_gpu_%s.log" % tag)
cp("bench_forcedist.json", "bench_one_rank_rccl_%s.json" % tag)
cp("bench_rehearse2.json", "bench_two_ranks_gloo_one_gpu_%s.json" % tag)
cp("query_pmc_q20.txt", "query_pmc_q20_%s.txt" % tag)       # round 5: PMC bytes of every kernel of a 2^20 query / of loop A beside the ledger
cp("loop_a_pmc.txt", "loop_a_pmc_%s.txt" % tag)
cp("kernel_rooflines_q20_two_lanes.txt", "kernel_rooflines_q20_two_lanes_%s.txt" % tag)


def per_pass(path):
    """sum of the counter over the two loop-B launches (limb 0 + limbs 1-11) of one pass, averaged over passes"""
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["Counter_Value"]) for r in rows)
    return tot / (len(rows) / 2), len(rows)


fetch_kb, nf = per_pass(os.path.join(G, "pmc_fetch_tensor.csv"))
write_kb, nw = per_pass(os.path.join(G, "pmc_write_tensor.csv"))
bench = json.load(open(os.path.join(G, "bench_final.json")))
head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=R, capture_output=True, text=True).stdout.strip()
out = {
    "log2n": 20,
    "kernel": "k_hydia_tensor<2,4,nt> (limb 0: 8-byte residues) + k_hydia_tensor24<2,4,B46> (limbs 1-11: 46-bit residues in 736-byte units, group-sequential database)",
    "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb, "launch_pairs_counted": [nf // 2, nw // 2],
    "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide streaming reads; MI355X_MICROARCH.md HBM section)",
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "resident_bytes_per_launch": bench["roofline"].get("bytes_per_launch"),
    "kernel_sha": kernel_sha(), "commit": head,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (%s %s)" % (ROUND, tag),
}
json.dump(out, open(os.path.join(R, "profiles", "tensor_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# round 5: the one-GPU components bench.py --gpus N turns into DESIGN section 7's prediction (profiles/scaling_model.json)
try:
    mp = os.path.join(R, "profiles", "scaling_model.json")
    model = json.load(open(mp))
    for lg in (20, 17):
        model["log2n"][str(lg)] = json.load(open(os.path.join(G, "scaling_components_%d.json" % lg)))["ranks"]
    model["source"] = "profiles/%s/scaling_components_2p20_%s.json, scaling_components_2p17_%s.json" % (ROUND, tag, tag)
    json.dump(model, open(mp, "w"), indent=1)
    print("scaling_model.json refreshed from", model["source"])
except (OSError, KeyError) as e:
    print("scaling_model.json not refreshed:", e)
