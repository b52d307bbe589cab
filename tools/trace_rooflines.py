#!/usr/bin/env python3
"""Per-kernel achieved HBM-side bandwidth of ONE 2^20 indexScenario query, from the per-launch trace written by
tools/gpu_trace_q20.sh (gpurun_out/trace_q20.csv: kernel, grid in workgroups, duration).  The bytes of a launch follow from its
grid (limb-polys are 256 KiB at N = 2^15); formulas are spelled out below.  Usage: trace_rooflines.py trace.csv > table.txt"""
import csv
import collections
import re
import sys

LP = 32768 * 8          # one limb-polynomial
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()


def add(name, us, nbytes, note):
    a = agg.setdefault(name, [0, 0.0, 0.0, note])
    a[0] += 1
    a[1] += us
    a[2] += nbytes


for r in rows:
    k = r["kernel"]
    g = [int(v) for v in r["grid_wg"].split("x")]
    us = float(r["us"])
    if k.startswith("k_hydia_tensor"):
        # grid.x = 256 tiles * G/(BPP*NW) block groups, grid.y limbs; bytes from the resident layout: DB + rot + acc
        packed = "true, true" in k
        G = g[0] // 256 * 8
        limbs = g[1]
        db = G * 512 * 2 * limbs * 32768 * (6 if packed else 8)
        rot = 512 * 2 * limbs * LP
        acc = G * 3 * limbs * LP
        add(k, us, db + rot + acc, "DB (6- or 8-byte residues) + rotated queries once + accumulators")
    elif k.startswith("k_ntt15_p1"):
        add(k, us, g[1] * LP * 2, "grid.y limb-polys read + written")
    elif k.startswith("k_ntt15_p2_ip"):
        np_ = int(re.search(r"<(\d+)", k).group(1))
        own = "true" in k
        add(k, us, g[1] * LP * (np_ + (1 if own else 0) + 2), "NP digits (+ own limb) read, 2 accumulator rows written; key tiles from L2")
    elif k.startswith("k_ntt15_p2"):
        m = re.search(r"<(true|false), (\d+), (\d+)>", k)
        np_, st = int(m.group(2)), int(m.group(3))
        extra = {0: 0, 1: 1, 2: 1, 3: 2}[st]     # epilogue operands: acc (1), rescale input (1), acc + addend (2)
        add(k, us, g[1] * np_ * LP * (2 + extra), "limb-polys read + written (+ epilogue operands)")
    elif k.startswith("k_inner_product"):
        nE, X = g[1], g[2]
        if X > 64:   # loop A: one 24 MiB key per rotation (19.9 MiB packed), digits from L2, 2 accumulator rows written
            add("k_inner_product<packed keys> (loop A)", us, X * (3 * 2 * 32768 * (5 * 8 + 11 * 6)) + X * 2 * nE * LP, "packed keys streamed once + acc written")
        else:
            add("k_inner_product (relin, one digit)", us, X * nE * LP * 3, "own limbs read + 2 rows written")
    elif k == "k_base_convert":
        add(k, us, 0, "")
    else:
        add(k, us, 0, "")

tot = sum(a[1] for a in agg.values())
print("one indexScenario query at 2^20 vectors: %d launches, %.2f ms of kernel time" % (len(rows), tot / 1e3))
print("%-44s %5s %9s %8s  %s" % ("kernel", "n", "ms", "TB/s", "bytes counted"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    bw = "%.2f" % (a[2] / a[1] / 1e6) if a[2] else "-"
    print("%-44s %5d %9.3f %8s  %s" % (k[:44], a[0], a[1] / 1e3, bw, a[3]))
