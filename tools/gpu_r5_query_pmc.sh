# round 5: PMC HBM-side bytes (FETCH_SIZE, WRITE_SIZE; separate --pmc passes, counters only) of EVERY kernel of one indexScenario query at
# 2^$1 vectors beside the byte ledger's by-design figure: where does a kernel fetch more than its operands (twiddles, keys, digits
# falling out of L2)?  Usage: gpu_r5_query_pmc.sh L [VAR=value ...]
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; L=$1; shift
for kv in "$@"; do export $kv; done
cd /tmp && export TMPDIR=/tmp
export HYDIA_LANES=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_q_$c -- python3 $R/tools/prof_query_ledger.py $L 2 indexScenario > $R/gpurun_out/pmc_q.log 2>&1 || { tail -5 $R/gpurun_out/pmc_q.log; exit 1; }
done
cd $R
python3 - $L "$@" <<'PY'
import csv, glob, collections, json, sys
L = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_q_%s/*/*counter_collection.csv' % c)[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
led = json.load(open('gpurun_out/ledger_q%s.json' % L))
calls = 3  # prof_query_ledger.py L 2: one warm-up query + two recorded (set-up kernels are divided by the same 3: ignore them)
out = open('gpurun_out/query_pmc_q%s.txt' % L, 'w')
def p(s):
    print(s); out.write(s + "\n")
p("one indexScenario query at 2^%s vectors %s: fabric-side bytes per query by PMC (FETCH_SIZE x2 per the guide's gfx950 correction, WRITE_SIZE) beside the ledger (by design, reads + writes)" % (L, " ".join(sys.argv[2:])))
p("%-36s %8s %12s %10s %12s %10s %8s" % ("kernel", "launches", "fetch x2 GB", "write GB", "fetch+write", "ledger GB", "ratio"))
tf = tw = tl = 0.0
rows = []
for k, c in agg.items():
    if not k.startswith('k_'): continue
    lk = [kk for kk in led['ledger'] if kk == k]
    if not lk: continue
    n = cnt[k]['FETCH_SIZE'] / calls
    f = 2 * c.get('FETCH_SIZE', 0) * 1024 / calls / 1e9; w = c.get('WRITE_SIZE', 0) * 1024 / calls / 1e9
    l = led['ledger'][k]['bytes'] / led['queries'] / 1e9
    rows.append((f + w, k, n, f, w, l))
for fw, k, n, f, w, l in sorted(rows, reverse=True):
    p("%-36s %8.1f %12.3f %10.3f %12.3f %10.3f %8.2f" % (k[:36], n, f, w, fw, l, fw / l if l else 0)); tf += f; tw += w; tl += l
p("%-36s %8s %12.3f %10.3f %12.3f %10.3f %8.2f" % ("total", "", tf, tw, tf + tw, tl, (tf + tw) / tl))
PY
rm -rf gpurun_out/pmc_q_FETCH_SIZE gpurun_out/pmc_q_WRITE_SIZE
