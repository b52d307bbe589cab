# round 4: SQ counters of EVERY kernel of a 2^20 indexScenario (64-block batch): two --pmc passes (counters only), then the
# kernel-trace table of the same command for the durations.  -> gpurun_out/sq_counters_tails.txt
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
L=${1:-20}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_sq1 -- python3 $R/tools/prof_query_ledger.py $L 1 > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq.log; exit 1; }
echo "pass 1 done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/tools/prof_query_ledger.py $L 1 > $R/gpurun_out/pmc_sq.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sq.log; exit 1; }
echo "pass 2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_q$L -- python3 $R/tools/prof_query_ledger.py $L 3 > $R/gpurun_out/kt_q$L.log 2>&1 || { tail -5 $R/gpurun_out/kt_q$L.log; exit 1; }
cd $R
cp $(ls gpurun_out/kt_q$L/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_q${L}_two_lanes.csv
python3 tools/kernel_rooflines.py gpurun_out/kernel_stats_q${L}_two_lanes.csv gpurun_out/ledger_q$L.json > gpurun_out/kernel_rooflines_q${L}_two_lanes.txt 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('pmc_sq1', 'pmc_sq2'):
    f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)[0]
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES',): cnt[k] += 1
        agg[k]['_vgpr'] = float(r.get('VGPR_Count', 0) or 0); agg[k]['_accvgpr'] = float(r.get('Accum_VGPR_Count', 0) or 0)
        agg[k]['_lds'] = float(r.get('LDS_Block_Size', 0) or 0); agg[k]['_wg'] = float(r.get('Workgroup_Size', 0) or 0)
out = open('gpurun_out/sq_counters_tails.txt','w')
hdr = "# per kernel over ALL its launches of one warm-up + one measured 2^20 indexScenario (rocprofv3 --pmc, two passes); fractions are of SQ_WAVE_CYCLES"
print(hdr); out.write(hdr + "\n")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES',0)):
    wc = c.get('SQ_WAVE_CYCLES',1) or 1
    n = max(cnt[k], 1)
    line = ("%-60s n=%4d vgpr=%3d acc=%3d lds=%6d wg=%4d wave_cyc=%.2e busy_cyc=%.2e valu=%.2f lds=%.2f sca=%.2f wait_inst=%.2f wait_any=%.2f wait_lds=%.2f | total: waves=%.3g VALU=%.4g SALU=%.3g SMEM=%.3g LDS=%.3g VMEMrd=%.3g VMEMwr=%.3g ldsconf=%.3g"
            % (k, n, c['_vgpr'], c['_accvgpr'], c['_lds'], c['_wg'], wc, c.get('SQ_BUSY_CYCLES',0), c.get('SQ_ACTIVE_INST_VALU',0)/wc, c.get('SQ_ACTIVE_INST_LDS',0)/wc, c.get('SQ_ACTIVE_INST_SCA',0)/wc,
               c.get('SQ_WAIT_INST_ANY',0)/wc, c.get('SQ_WAIT_ANY',0)/wc, c.get('SQ_WAIT_INST_LDS',0)/wc, c.get('SQ_WAVES',0), c.get('SQ_INSTS_VALU',0), c.get('SQ_INSTS_SALU',0),
               c.get('SQ_INSTS_SMEM',0), c.get('SQ_INSTS_LDS',0), c.get('SQ_INSTS_VMEM_RD',0), c.get('SQ_INSTS_VMEM_WR',0), c.get('SQ_LDS_BANK_CONFLICT',0)))
    print(line[:400]); out.write(line+"\n")
PY
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/kt_q$L
