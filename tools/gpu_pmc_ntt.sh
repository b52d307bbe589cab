# SQ / LDS counters of the NTT microbenchmark (tools/bench_ntt.py)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_ntt -- python3 $R/tools/bench_ntt.py > $R/gpurun_out/pmc_ntt.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_ntt2 -- python3 $R/tools/bench_ntt.py > $R/gpurun_out/pmc_ntt2.log 2>&1
cd $R
tail -3 gpurun_out/pmc_ntt2.log
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('pmc_ntt', 'pmc_ntt2'):
    fs = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)
    if not fs: print('no counters in', d); continue
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
with open('gpurun_out/pmc_ntt_summary.txt','w') as out:
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES',0)):
        wc = c.get('SQ_WAVE_CYCLES',1) or 1
        line = "%-40s n=%4d " % (k, cnt[k]) + " ".join("%s=%.3e" % (n, v) for n, v in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmc_ntt gpurun_out/pmc_ntt2
