#!/bin/bash
# GPU box: kernel trace of tests/kernel_checks.py::check_bn_fused_squatter (its last leg runs RCCL's reduce kernel on a side
# stream next to the persistent BatchNorm backward) -> gpurun_out/<tag>/rccl_overlap_bn_fused.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
O="gpurun_out/${1:-rccl}"; mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rccl_tr -- python tests/kernel_checks.py bn_fused_squatter > "$O/rccl_overlap_check.log" 2>&1
tail -12 "$O/rccl_overlap_check.log"
python tools/rccl_overlap.py /tmp/rccl_tr | tee "$O/rccl_overlap_bn_fused.txt"
