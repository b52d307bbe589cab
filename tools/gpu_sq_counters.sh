set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/tools/prof_similarity.py 20 1 indexScenario > $R/gpurun_out/pmc_sq.log 2>&1
tail -3 $R/gpurun_out/pmc_sq.log | cut -c1-200
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:48]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
out = open('gpurun_out/pmc_sq_summary.txt','w')
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES',0)):
    wc = c.get('SQ_WAVE_CYCLES',1) or 1
    line = "%-50s n=%4d wave_cyc=%.3e valu=%.2f lds=%.2f wait_inst=%.2f wait_any=%.2f lds_conf/lds_active=%.3f" % (k, cnt[k], wc, c.get('SQ_ACTIVE_INST_VALU',0)/wc, c.get('SQ_ACTIVE_INST_LDS',0)/wc, c.get('SQ_WAIT_INST_ANY',0)/wc, c.get('SQ_WAIT_ANY',0)/wc, c.get('SQ_LDS_BANK_CONFLICT',0)/max(1,c.get('SQ_LDS_IDX_ACTIVE',1)))
    print(line); out.write(line+"\n")
PY
rm -rf gpurun_out/pmc_sq
