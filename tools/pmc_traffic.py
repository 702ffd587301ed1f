"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes made with tools/pmc_step.py into per-kernel HBM
traffic per launch (JSON on stdout).  Counter unit: KiB (cdna_hip_programming.md §7); the gfx950 read-side
under-count is calibrated on the 1 GiB copy_ that pmc_step.py dispatches first, the write side on the fill_."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

GIB = float(1 << 30)


def short(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"((?:conv\w*|bn_\w+|sivae\w*|\w+_kernel)\s*<[^()]*>|\w+)", name)
    key = m.group(1) if m else name[:60]
    return key.replace(" ", "")


def read_pass(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    # one row per (dispatch, XCD/instance) -> sum per dispatch
    per = defaultdict(float)
    names = {}
    for did, name, v in rows:
        per[did] += v
        names[did] = name
    return [(did, names[did], per[did]) for did in sorted(per)]


def main():
    fetch = read_pass(sys.argv[1], "FETCH_SIZE")
    write = read_pass(sys.argv[2], "WRITE_SIZE")
    # calibration dispatches: the first two of each pass (fill_ 1 GiB, copy_ 1 GiB)
    cal_r = GIB / (fetch[1][2] * 1024.0)
    cal_w_fill = GIB / (write[0][2] * 1024.0)
    cal_w_copy = GIB / (write[1][2] * 1024.0)
    out = {"lib_sha256_16": sys.argv[sys.argv.index("--lib-sha") + 1] if "--lib-sha" in sys.argv else None,
           "unit": "bytes per launch", "calibration": {
        "copy_1GiB_FETCH_SIZE_KiB": fetch[1][2], "read_scale": round(cal_r, 4),
        "fill_1GiB_WRITE_SIZE_KiB": write[0][2], "copy_1GiB_WRITE_SIZE_KiB": write[1][2],
        "write_scale": round(0.5 * (cal_w_fill + cal_w_copy), 4),
        "kernels": [short(fetch[0][1]), short(fetch[1][1])]}}
    cal_w = 0.5 * (cal_w_fill + cal_w_copy)
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for did, name, v in fetch[2:]:
        a = agg[short(name)]
        a[0] += 1
        a[1] += v * 1024.0
    for did, name, v in write[2:]:
        agg[short(name)][2] += v * 1024.0
    ks = {}
    for k, (n, f, w) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        if n == 0:
            continue
        ks[k] = {"launches": n, "fetch_raw": round(f / n), "write_raw": round(w / n),
                 "hbm_bytes": round((f * cal_r + w * cal_w) / n)}
    out["kernels"] = ks
    out["step_total_hbm_bytes"] = round(sum(v["hbm_bytes"] * v["launches"] for v in ks.values()))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
