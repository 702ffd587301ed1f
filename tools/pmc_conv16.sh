cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
WHAT=${1:-fwd}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_WAIT_INST_VMEM SQ_INSTS_SMEM"
rm -rf gpurun_out/pmc_c16; i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); timeout 300 rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_c16/p$i -- python tools/bench_conv16.py celeb128 128 $WHAT > gpurun_out/pmc_c16_$i.log 2>&1; done
cat gpurun_out/pmc_c16_1.log | grep -v Warn | tail -16
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_c16/**/*counter_collection.csv', recursive=True):
    per=collections.defaultdict(float); key={}
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'bf16_conv_kernel' in n or 'bf16_wgrad_kernel' in n:
            k=(n.split('(')[0][-60:], r['Grid_Size'])
            per[(int(r['Dispatch_Id']), r['Counter_Name'])]+=float(r['Counter_Value']); key[int(r['Dispatch_Id'])]=k
    for (d,c),v in per.items(): agg[key[d]][c].append(v)
for k in sorted(agg):
    a={c:sum(v)/len(v) for c,v in agg[k].items()}
    wc=a.get('SQ_WAVE_CYCLES',1)
    print(k, "n=%d"%len(next(iter(agg[k].values()))))
    print("   wait_any %.2f  wait_inst %.2f  active %.2f | mfma_busy/(4*wave_cyc/waves_per_simd..) raw: mfma_busy %.3g wave_cyc %.3g gui %.3g waves %.0f" % (a.get('SQ_WAIT_ANY',0)/wc, a.get('SQ_WAIT_INST_ANY',0)/wc, a.get('SQ_ACTIVE_INST_ANY',0)/wc, a.get('SQ_VALU_MFMA_BUSY_CYCLES',0), wc, a.get('GRBM_GUI_ACTIVE',0), a.get('SQ_WAVES',0)))
    print("   mfma_busy_frac %.3f  insts: mfma %.3g valu %.3g salu %.3g lds %.3g vmem %.3g | lds_active %.3g bank_conflict %.3g lds_idx %.3g  wait_inst_lds %.3g" % (a.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*a.get('GRBM_GUI_ACTIVE',1)/8), a.get('SQ_INSTS_MFMA',0), a.get('SQ_INSTS_VALU',0), a.get('SQ_INSTS_SALU',0), a.get('SQ_INSTS_LDS',0), a.get('SQ_INSTS_VMEM',0), a.get('SQ_ACTIVE_INST_LDS',0), a.get('SQ_LDS_BANK_CONFLICT',0), a.get('SQ_LDS_IDX_ACTIVE',0), a.get('SQ_WAIT_INST_LDS',0)))
PY
