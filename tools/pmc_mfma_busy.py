"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_F32` pass over tools/pmc_step.py into per-kernel matrix-pipe utilisation (JSON on stdout).

matrix-pipe busy fraction of a dispatch = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip, cycles in which a SIMD's
matrix pipe is busy) / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD).  GRBM_GUI_ACTIVE is reported per XCD instance and summed
by rocprofv3 over the 8 XCDs, hence the /8 (same derivation as profiles/r1_pmc_conv_fwd.txt)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

N_SIMD = 1024
N_XCD = 8


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"((?:conv\w*|wino\w*|bn_\w+|sivae\w*|\w+_kernel)\s*<[^()]*>|\w+)", name)
    return (m.group(1) if m else name[:60]).replace(" ", "")


def _lib_sha():
    """--lib-sha <first 16 hex digits of the sha256 of the libsivae_hip.so the pass ran with> (tools/profile.sh)"""
    return sys.argv[sys.argv.index("--lib-sha") + 1] if "--lib-sha" in sys.argv else None


def main():
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                did = int(r["Dispatch_Id"])
                per[did][r["Counter_Name"]] += float(r["Counter_Value"])
                names[did] = r["Kernel_Name"]
    agg = defaultdict(lambda: defaultdict(float))
    for did, c in per.items():
        a = agg[short(names[did])]
        a["launches"] += 1
        for k, v in c.items():
            a[k] += v
    out = {}
    tot_busy = tot_act = 0.0
    for k, a in agg.items():
        act = a.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot_busy += busy
        tot_act += act
        if busy <= 0:
            continue
        out[k] = {"launches": int(a["launches"]), "gui_active_cycles_per_launch": round(act / a["launches"]),
                  "mfma_busy_frac": round(busy / (N_SIMD * act), 4) if act else None,
                  "mfma_mops_f32_per_launch": round(a.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / a["launches"]),
                  "share_of_gpu_active": None}
    for k in out:
        out[k]["share_of_gpu_active"] = round(agg[k]["GRBM_GUI_ACTIVE"] / N_XCD / tot_act, 4)
    res = {"lib_sha256_16": _lib_sha(),
           "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8)",
           "whole_step_mfma_busy_frac": round(tot_busy / (N_SIMD * tot_act), 4),
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["share_of_gpu_active"]))}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
