#!/usr/bin/env python3
"""Condense a rocprofv3 kernel_trace.csv into one line per launch of the LAST query of tools/prof_similarity.py:
short kernel name, grid (workgroups), block, duration in microseconds.  Usage: trace_summarise.py <trace.csv> <out.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last query starts at the last launch of the hoisting base conversion that follows the last hydia_tensor-free gap: find the last
# occurrence of the loop-B kernel (limb 0 launch) and walk back to the preceding k_copy / first kernel after the previous query's end
names = [r["Kernel_Name"] for r in rows]
tens = [i for i, n in enumerate(names) if "k_hydia_tensor" in n]
last_t = tens[-2] if len(tens) >= 2 else tens[-1]
prev_t = tens[-3] if len(tens) >= 3 else -1
# previous query's end = its last kernel before this query's first; queries are separated by > 1 ms of host time is not guaranteed, so
# take everything after the previous query's tensor launches and drop the tail of that query by looking for the k_add_scalar (+1) launch
start = prev_t + 1
for i in range(prev_t + 1, last_t):
    if "k_add_scalar" in names[i]:
        start = i + 1
out = csv.writer(open(sys.argv[2], "w"))
out.writerow(["idx", "kernel", "grid_wg", "block", "us"])
for i in range(start, len(rows)):
    r = rows[i]
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0]
    gx, gy, gz = int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    bx, by, bz = int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"])
    wg = (gx // bx) * (gy // by) * (gz // bz)
    out.writerow([i - start, n, "%dx%dx%d" % (gx // bx, gy // by, gz // bz), bx * by * bz,
                  "%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)])
