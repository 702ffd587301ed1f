#!/bin/bash
# kernel-trace statistics of the small-shard iterations (256x256: 16-image shard, bootstrap 8-image shard)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "bs16:--global-batch 16" "boot8:--bootstrap --global-batch 8"; do
  tag=${spec%%:*}; fl=${spec#*:}
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py $fl --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing > /tmp/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/r3_kernel_stats_${tag}_paired.csv
  tail -1 /tmp/prof_$tag.log | cut -c1-200
done
