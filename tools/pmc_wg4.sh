# SQ counters of wino4_wgrad_kernel on one layer shape (three --pmc passes).  usage: pmc_wg4.sh "B Ci Co H 3 reps"
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ARGS=${1:-"32 512 512 32 3 3"}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_VMEM"
rm -rf gpurun_out/pmc_g4; i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); timeout 200 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_g4/p$i -- python tools/bench_one.py wgrad $ARGS > gpurun_out/pmc_g4_$i.log 2>&1; tail -1 gpurun_out/pmc_g4_$i.log | cut -c1-200; done
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_g4/**/*counter_collection.csv', recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'wino4_wgrad_kernel' in r['Kernel_Name']:
            per[(r['Dispatch_Id'], r['Counter_Name'])]+=float(r['Counter_Value'])
    for (d,c),v in per.items(): agg[c].append(v)
for c,v in sorted(agg.items()): print("%-34s n=%d mean=%.5g"%(c,len(v),sum(v)/len(v)))
PY
rm -rf gpurun_out/pmc_g4
