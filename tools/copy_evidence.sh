#!/bin/bash
# gpurun_out/<tag>/ (one tools/r6_final_batch.sh call) -> the tracked names under profiles/.   usage: tools/copy_evidence.sh <tag> [round prefix, default r6]
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$1; R=${2:-r6}
cp $O/bench_default_invocation.json profiles/${R}_bench_default_invocation.json
for t in local_bn sync_bn; do cp $O/bench_2rank_gloo_same_device_$t.json profiles/${R}_bench_2rank_self_spawned_gloo_same_device_$t.json; done
for f in $O/kernel_stats_*.csv; do n=$(basename $f); cp $f profiles/${R}_rocprofv3_$n; done
for f in $O/pmc_traffic_*.json $O/pmc_mfma_busy_*.json; do cp $f profiles/${R}_$(basename $f); done
for n in bs16 bs128 boot8 boot64; do cp $O/calls_$n.csv profiles/${R}_kernel_calls_one_iteration_$n.csv; done
for f in $O/layer_times_*.txt; do cp $f profiles/${R}_$(basename $f); done
[ -f $O/tests_full.txt ] && { grep -E "passed|failed" $O/tests_full.txt | tail -2; echo "($(cat $O/lib.txt); pytest tests -m gpu -q on one MI355X, $O)"; } > profiles/${R}_gpu_tests_full_suite.txt
cat $O/lib.txt
