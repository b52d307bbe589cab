#!/usr/bin/env python3
"""Copy the outputs of tools/gpu_deliver.sh from gpurun_out/ into profiles/r01/ under a version tag and refresh
profiles/tensor_traffic.json from the two PMC passes.  Usage: collect_profiles.py v5"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles", "r01")


def biggest(pattern, needle):
    """the kernel_stats file of the process that actually ran the workload (the one mentioning the loop-B kernel)"""
    best = None
    for f in glob.glob(pattern):
        txt = open(f).read()
        if needle in txt and (best is None or len(txt) > len(open(best).read())):
            best = f
    return best


shutil.copy(os.path.join(G, "bench_final.json"), os.path.join(P, "bench_2p20_%s.json" % tag))
for l in (10, 14, 17):
    shutil.copy(os.path.join(G, "bench_2p%d.json" % l), os.path.join(P, "bench_2p%d_%s.json" % (l, tag)))
shutil.copy(biggest(os.path.join(G, "prof_bench_stats2", "*", "*kernel_stats.csv"), "k_hydia_tensor"),
            os.path.join(P, "bench_2p20_kernel_stats_%s.csv" % tag))
shutil.copy(biggest(os.path.join(G, "prof_q20c", "*", "*kernel_stats.csv"), "k_hydia_tensor"),
            os.path.join(P, "indexscenario_2p20_queryonly_kernel_stats_%s.csv" % tag))
shutil.copy(os.path.join(G, "pmc2_fetch_tensor.csv"), os.path.join(P, "bench_2p20_pmc_fetch_tensor_%s.csv" % tag))
shutil.copy(os.path.join(G, "pmc2_write_tensor.csv"), os.path.join(P, "bench_2p20_pmc_write_tensor_%s.csv" % tag))
shutil.copy(os.path.join(G, "cli_2_10.log"), os.path.join(P, "cli_ImageMatching_2_10.log"))
shutil.copy(os.path.join(G, "latency.csv"), os.path.join(P, "cli_latency.csv"))


def per_pass(path):
    """sum of the counter over the two loop-B launches (limb 0 + limbs 1-11) of one pass, averaged over passes"""
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["Counter_Value"]) for r in rows)
    launches = len(rows)
    return tot / (launches / 2), launches


fetch_kb, nf = per_pass(os.path.join(G, "pmc2_fetch_tensor.csv"))
write_kb, nw = per_pass(os.path.join(G, "pmc2_write_tensor.csv"))
bench = json.load(open(os.path.join(G, "bench_final.json")))
out = {
    "log2n": 20,
    "kernel": "k_hydia_tensor<2,4,nt> (limb 0: 8-byte residues) + k_hydia_tensor<2,4,nt,packed> (limbs 1-11: 6-byte residues)",
    "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb, "launch_pairs_counted": [nf // 2, nw // 2],
    "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide streaming reads; MI355X_MICROARCH.md HBM section)",
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (%s)" % tag,
}
json.dump(out, open(os.path.join(R, "profiles", "tensor_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print("bench:", bench["value"], bench["ms_per_step"], bench["roofline"]["achieved"], bench["roofline"]["frac"], bench["cpu_baseline"]["value"])
