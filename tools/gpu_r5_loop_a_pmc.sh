# round 5: how many bytes does loop A really move through HBM?  PMC FETCH_SIZE / WRITE_SIZE of every kernel of
# rotateQuery (two separate --pmc passes, counters only, the program itself after --) beside the byte ledger's by-design figure.
# FETCH_SIZE is quoted x2 as MI355X_MICROARCH.md's HBM section prescribes for gfx950 wide streaming reads (and raw beside it).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_la_$c -- python3 $R/tools/prof_rotate.py 2 > $R/gpurun_out/pmc_la.log 2>&1 || { tail -5 $R/gpurun_out/pmc_la.log; exit 1; }
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_la_%s/*/*counter_collection.csv' % c)[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
led = json.load(open('gpurun_out/ledger_rot.json'))
calls = 3  # prof_rotate.py 2: one warm-up call + two recorded
out = open('gpurun_out/loop_a_pmc.txt', 'w')
def p(s):
    print(s); out.write(s + "\n")
p("loop A (rotateQuery, 511 hoisted rotations): HBM bytes per call by PMC (FETCH_SIZE, WRITE_SIZE in KiB units; separate passes) beside the ledger")
p("%-34s %9s %12s %12s %12s %12s" % ("kernel", "launches", "fetch GB raw", "fetch GB x2", "write GB", "ledger GB"))
tf = tw = tl = 0.0
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('FETCH_SIZE', 0)):
    n = cnt[k]['FETCH_SIZE'] / calls
    f = c.get('FETCH_SIZE', 0) * 1024 / calls / 1e9; w = c.get('WRITE_SIZE', 0) * 1024 / calls / 1e9
    l = sum(v['bytes'] / led['queries'] for kk, v in led['ledger'].items() if kk.startswith(k[:30])) / 1e9 if k.startswith('k_') else 0.0
    if f + w < 0.01: continue
    p("%-34s %9.1f %12.3f %12.3f %12.3f %12.3f" % (k[:34], n, f, 2 * f, w, l)); tf += f; tw += w; tl += l
p("%-34s %9s %12.3f %12.3f %12.3f %12.3f" % ("total", "", tf, 2 * tf, tw, tl))
PY
rm -rf gpurun_out/pmc_la_FETCH_SIZE gpurun_out/pmc_la_WRITE_SIZE
