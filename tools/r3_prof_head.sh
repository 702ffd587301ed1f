#!/bin/bash
# kernel-trace statistics of the headline iteration (256x256, batch 128, fp32)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3p; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_head -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-also > $O/prof_head.log 2>&1
find $O/st_head -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_celeb256_bs128_fp32.csv \; ; rm -rf $O/st_head
tail -1 $O/prof_head.log | cut -c1-160
