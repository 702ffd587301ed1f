#!/bin/bash
# kernel-trace statistics of the small-shard iterations (256x256: 16-image shard, bootstrap 8-image shard)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3p; mkdir -p $O
for spec in "bs16:--global-batch 16" "boot8:--bootstrap --global-batch 8"; do
  tag=${spec%%:*}; fl=${spec#*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$tag -- python bench.py $fl --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/prof_$tag.log 2>&1
  find $O/st_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_${tag}_paired.csv \; ; rm -rf $O/st_$tag
  tail -1 $O/prof_$tag.log | cut -c1-200
done
