cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
BIN=${1:-tools/_build/wino_b6_o1_w2_r4}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_WAIT_INST_VMEM SQ_INSTS_SMEM"
rm -rf gpurun_out/pmc_b6; i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); timeout 200 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_b6/p$i -- $BIN > gpurun_out/pmc_b6_$i.log 2>&1; tail -1 gpurun_out/pmc_b6_$i.log | cut -c1-200; done
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(dict)
for f in glob.glob('gpurun_out/pmc_b6/**/*counter_collection.csv', recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'conv_wino_b6' in r['Kernel_Name']:
            per[(int(r['Dispatch_Id']), r['Counter_Name'])]+=float(r['Counter_Value'])
    for (d,c),v in per.items(): agg[c][d]=v
ds=sorted(next(iter(agg.values())).keys())
pick=[ds[3], ds[9], ds[15], ds[19]] if len(ds)>=20 else ds
print("dispatches", len(ds), pick)
for c in sorted(agg): print("%-28s"%c, "  ".join("%12.5g"%agg[c].get(d,float('nan')) for d in pick))
PY
