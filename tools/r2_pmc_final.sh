# PMC passes of the final round-2 build at the headline workload: HBM traffic (FETCH_SIZE / WRITE_SIZE) and matrix-pipe busy
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2w; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python tools/pmc_step.py > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python tools/pmc_step.py > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic_celeb256_bs128_fp32.json; rm -rf $O/pmc_f $O/pmc_w
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_m -- python tools/pmc_step.py > $O/pmc_m.log 2>&1
python tools/pmc_mfma_busy.py $O/pmc_m > $O/pmc_mfma_busy_celeb256_bs128_fp32.json; rm -rf $O/pmc_m
python -c "
import json
d=json.load(open('$O/pmc_traffic_celeb256_bs128_fp32.json')); print('traffic GB', d['step_total_hbm_bytes']/1e9, d['calibration'].get('read_scale'))
d=json.load(open('$O/pmc_mfma_busy_celeb256_bs128_fp32.json')); print('busy', d['whole_step_mfma_busy_frac'])
for k,v in list(d['kernels'].items())[:4]: print(k, v['mfma_busy_frac'], v['share_of_gpu_active'])
"
