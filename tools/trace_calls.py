"""Per-call kernel durations of the LAST iteration of a `rocprofv3 --kernel-trace` run of bench.py (GPU box).
usage: python tools/trace_calls.py <rocprof out dir> > calls.csv
One iteration = the window between the ends of the third-last and the last adam_kernel call (two optimizer steps per
iteration), so set-up launches and partial iterations stay out; calls are in launch order: the same position in two
traces (two batch sizes) is the same layer and pass."""
import csv, glob, sys, collections
d = sys.argv[1]
path = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
calls = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    calls[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                    int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
                                    int(r["LDS_Block_Size"])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "idx", "us", "blocks_x", "grid_y", "grid_z", "lds"])
adam = sorted(c[0] + int(c[1] * 1e3) for k, v in calls.items() if k.startswith("adam_kernel") for c in v)
t1, t2 = adam[-3], adam[-1]
for k, v in calls.items():
    v.sort()
    for i, c in enumerate([c for c in v if t1 < c[0] <= t2]):
        w.writerow([k[:90], i, "%.1f" % c[1], c[2], c[3], c[4], c[5]])
