# round 5: arbitrary PMC counters of EVERY kernel of one indexScenario query at 2^$1 vectors (single lane), one --pmc pass per quoted group.
# Usage: gpu_r5_query_counters.sh L "CTR_A CTR_B ..." ["CTR_C ..."]   ->  gpurun_out/query_counters_q<L>.txt  (per kernel: launches, ms, ledger GB, counter sums per query)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; L=$1; shift
cd /tmp && export TMPDIR=/tmp
export HYDIA_LANES=1
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_c$i
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_c$i -- python3 $R/tools/prof_query_ledger.py $L 2 indexScenario > $R/gpurun_out/pmc_c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_c.log; exit 1; }
done
rm -rf $R/gpurun_out/kt_c
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_c -- python3 $R/tools/prof_query_ledger.py $L 2 indexScenario > $R/gpurun_out/pmc_c.log 2>&1 || exit 1
cd $R
python3 - $L $i <<'PY'
import csv, glob, collections, json, sys
L, ngrp = sys.argv[1], int(sys.argv[2])
def short(n): return n.replace('void ','').replace('(anonymous namespace)::','').split('(')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); names = []
for g in range(1, ngrp + 1):
    f = glob.glob('gpurun_out/pmc_c%d/*/*counter_collection.csv' % g)[0]
    for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] not in names: names.append(r['Counter_Name'])
led = json.load(open('gpurun_out/ledger_q%s.json' % L))
st = {short(r['Name']): (int(r['Calls']), int(r['TotalDurationNs']) / 1e6) for r in csv.DictReader(open(glob.glob('gpurun_out/kt_c/*/*kernel_stats.csv')[0]))}
calls = 3
out = open('gpurun_out/query_counters_q%s.txt' % L, 'w')
def p(s):
    print(s); out.write(s + "\n")
p("one indexScenario query at 2^%s vectors, single lane: counter sums per query (rocprofv3 --pmc, one pass per group), duration and ledger bytes" % L)
p("%-34s %5s %8s %9s " % ("kernel", "n", "ms", "ledgerGB") + " ".join("%14s" % n[-14:] for n in names))
rows = []
for k, c in agg.items():
    if k not in led['ledger'] or k not in st: continue
    n = led['ledger'][k]['launches'] / led['queries']; ms = st[k][1] / calls; gb = led['ledger'][k]['bytes'] / led['queries'] / 1e9
    rows.append((ms, k, n, gb, [c.get(x, 0) / calls for x in names]))
for ms, k, n, gb, vals in sorted(rows, reverse=True):
    p("%-34s %5d %8.3f %9.3f " % (k[:34], n, ms, gb) + " ".join("%14.4g" % v for v in vals))
PY
rm -rf gpurun_out/pmc_c? gpurun_out/kt_c
