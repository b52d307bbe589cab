#!/usr/bin/env python3
"""Per-kernel achieved bandwidth of one query: rocprofv3 --stats kernel totals (time) joined with the library's byte ledger
(bytes each launch has to move: tools/prof_query_ledger.py).  The run holds one untimed query before the ledger starts, so
times are scaled by queries / (queries + 1).  Usage: kernel_rooflines.py <kernel_stats.csv> <ledger.json> > table.txt"""
import csv
import json
import re
import sys

stats = list(csv.DictReader(open(sys.argv[1])))
led = json.load(open(sys.argv[2]))
Q = led["queries"]


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


times = {}
for r in stats:
    k = short(r["Name"])
    times[k] = (int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6)
rows, tot_ms, tot_b, unacc = [], 0.0, 0.0, 0.0
for k, v in led["ledger"].items():
    t = times.get(k)
    if k == "k_addsub":  # three template instances share the ledger name
        t = (sum(c for n, (c, m) in times.items() if n.startswith("k_addsub")), sum(m for n, (c, m) in times.items() if n.startswith("k_addsub")))
    if not t or not t[0]:
        continue
    ms_q = t[1] * (v["launches"] / t[0]) / Q     # this query's share of the kernel's profile time
    rows.append((k, v["launches"] // Q, ms_q, v["bytes"] / Q))
    tot_ms += ms_q
    tot_b += v["bytes"] / Q
seen = {r[0] for r in rows}
other = [(k, c, m) for k, (c, m) in times.items() if k not in seen and not k.startswith("k_addsub") and
         not any(s in k for s in ("fill_uniform", "db_repack", "key_pack", "rocclr"))]
print("one %s query at 2^%d vectors (%.1f GiB resident): %.2f ms wall per query, %.2f ms of kernel time in %d launches, %.1f GB moved"
      % (led["scenario"], led["log2n"], led["db_bytes"] / 2 ** 30, led["ms_per_query"], tot_ms, sum(r[1] for r in rows), tot_b / 1e9))
print("%-40s %6s %9s %10s %8s" % ("kernel", "n", "ms", "GB", "TB/s"))
for k, n, ms, b in sorted(rows, key=lambda r: -r[2]):
    print("%-40s %6d %9.3f %10.3f %8.2f" % (k[:40], n, ms, b / 1e9, b / ms / 1e9 if ms else 0))
for k, c, m in other:
    print("%-40s %6d %9.3f %10s %8s   (not in the ledger: %d calls in the whole profile)" % (k[:40], c // (Q + 1), m / (Q + 1), "-", "-", c))
ops = {k[3:]: v["bytes"] / Q for k, v in led["ledger"].items() if k.startswith("op:")}
if ops:
    print("inherent bytes per query by operation (SURVEY 8d pricing: 2 N 8 B per limb-transform, keys once per launch): %.1f GB in all — %s"
          % (sum(ops.values()) / 1e9, ", ".join("%s %.2f" % (k, b / 1e9) for k, b in sorted(ops.items(), key=lambda kv: -kv[1]))))
