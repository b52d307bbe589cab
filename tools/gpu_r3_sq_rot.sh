# round 3: SQ counters of loop A's kernels (two --pmc passes, counters only)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_sq1 -- python3 $R/tools/prof_rotate.py 2 > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/tools/prof_rotate.py 2 > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('pmc_sq1', 'pmc_sq2'):
    f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES',): cnt[k] += 1
out = open('gpurun_out/pmc_sq_rot_summary.txt','w')
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES',0)):
    wc = c.get('SQ_WAVE_CYCLES',1) or 1
    n = max(cnt[k], 1)
    line = ("%-28s n=%3d wave_cyc=%.2e busy_cyc=%.2e valu=%.2f lds=%.2f sca=%.2f wait_inst=%.2f wait_any=%.2f wait_lds=%.2f | per launch: waves=%.3g VALU=%.3g SALU=%.3g SMEM=%.3g LDS=%.3g VMEMrd=%.3g VMEMwr=%.3g ldsconf=%.3g"
            % (k[:28], n, wc, c.get('SQ_BUSY_CYCLES',0), c.get('SQ_ACTIVE_INST_VALU',0)/wc, c.get('SQ_ACTIVE_INST_LDS',0)/wc, c.get('SQ_ACTIVE_INST_SCA',0)/wc,
               c.get('SQ_WAIT_INST_ANY',0)/wc, c.get('SQ_WAIT_ANY',0)/wc, c.get('SQ_WAIT_INST_LDS',0)/wc, c.get('SQ_WAVES',0)/n, c.get('SQ_INSTS_VALU',0)/n, c.get('SQ_INSTS_SALU',0)/n,
               c.get('SQ_INSTS_SMEM',0)/n, c.get('SQ_INSTS_LDS',0)/n, c.get('SQ_INSTS_VMEM_RD',0)/n, c.get('SQ_INSTS_VMEM_WR',0)/n, c.get('SQ_LDS_BANK_CONFLICT',0)/n))
    print(line); out.write(line+"\n")
PY
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
