# SQ counters of loop A alone (tools/prof_rotate.py)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_rot -- python3 $R/tools/prof_rotate.py 2 > $R/gpurun_out/pmc_rot.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_rot2 -- python3 $R/tools/prof_rotate.py 2 > $R/gpurun_out/pmc_rot2.log 2>&1
# (a third pass with FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum in ONE --pmc list aborted inside rocprofv3
#  (signal 6) and left the run silent until the watchdog killed it: do not combine those counters in one pass)
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('pmc_rot', 'pmc_rot2'):
    fs = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)
    if not fs: print('no counters in', d); continue
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES',): cnt[k] += 1
with open('gpurun_out/pmc_rot_summary.txt','w') as out:
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES',0))[:8]:
        line = "%-40s n=%4d " % (k, cnt[k]) + " ".join("%s=%.3e" % (n, v) for n, v in sorted(c.items()))
        print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmc_rot gpurun_out/pmc_rot2
