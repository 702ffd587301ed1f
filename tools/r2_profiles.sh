# round-2 profile collection (one gpurun call): kernel stats + PMC passes for the fp32 headline and the bf16 mode,
# bench JSONs of the other configurations.  Outputs under gpurun_out/r2p/ (copied to profiles/ afterwards).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r2p; mkdir -p $O
python bench.py --steps 6 --warmup 2 --cpu-iters 2 > $O/bench_celeb256_bs128_fp32.json 2>/dev/null
python bench.py --config celeb128 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_celeb128_bs128_bf16.json 2>/dev/null
python bench.py --config celeb128 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_celeb128_bs128_fp32.json 2>/dev/null
python bench.py --global-batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_celeb256_bs16_fp32.json 2>/dev/null
python bench.py --config cifar10 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cifar10_bs256_fp32.json 2>/dev/null
python bench.py --bootstrap --global-batch 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_bootstrap256_bs64_fp32.json 2>/dev/null
python bench.py --bootstrap --global-batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_bootstrap256_bs8_fp32.json 2>/dev/null
for f in $O/bench_*.json; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_fp32 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/prof_fp32.log 2>&1
find $O/st_fp32 -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_celeb256_bs128_fp32.csv \; ; rm -rf $O/st_fp32
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_bf16 -- python bench.py --config celeb128 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/prof_bf16.log 2>&1
find $O/st_bf16 -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats_celeb128_bs128_bf16.csv \; ; rm -rf $O/st_bf16
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python tools/pmc_step.py > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python tools/pmc_step.py > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic_celeb256_bs128_fp32.json; rm -rf $O/pmc_f $O/pmc_w
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python tools/pmc_step.py --config celeb128 --dtype bf16 > $O/pmc_f16.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python tools/pmc_step.py --config celeb128 --dtype bf16 > $O/pmc_w16.log 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic_celeb128_bs128_bf16.json; rm -rf $O/pmc_f $O/pmc_w
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_m -- python tools/pmc_step.py > $O/pmc_m.log 2>&1
python tools/pmc_mfma_busy.py $O/pmc_m > $O/pmc_mfma_busy_celeb256_bs128_fp32.json; rm -rf $O/pmc_m
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_m -- python tools/pmc_step.py --config celeb128 --dtype bf16 > $O/pmc_m16.log 2>&1
python tools/pmc_mfma_busy.py $O/pmc_m > $O/pmc_mfma_busy_celeb128_bs128_bf16.json; rm -rf $O/pmc_m
ls -la $O | head -40
python -c "
import json
for f in ('pmc_traffic_celeb256_bs128_fp32','pmc_traffic_celeb128_bs128_bf16'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['step_total_hbm_bytes']/1e9,'GB', d['calibration'])
for f in ('pmc_mfma_busy_celeb256_bs128_fp32','pmc_mfma_busy_celeb128_bs128_bf16'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['whole_step_mfma_busy_frac'], list(d['kernels'].items())[:3])
"
