"""Which kernels ran AT THE SAME TIME as the persistent BatchNorm backward?  Reads a `rocprofv3 --kernel-trace` directory
(tools/rccl_overlap.sh) and prints, per foreign kernel name (RCCL's kernels, the test's squatter), the number of launches,
their total time and the time they overlapped a bn_bwd_fused_kernel launch; then the first overlapping pairs with their
timestamps and queue ids.   usage: python tools/rccl_overlap.py <rocprof out dir>"""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
bn, other = [], []
for r in rows:
    n, s, e = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    (bn if "bn_bwd_fused_kernel" in n else other).append((s, e, n, q))
bn.sort()
t0 = min(r[0] for r in bn + other)
print("kernel trace: %s  (%d launches, %d of bn_bwd_fused_kernel)" % (path.split("/")[-1], len(rows), len(bn)))
agg = collections.OrderedDict()
pairs = []
for s, e, n, q in sorted(other):
    ov = 0
    for bs, be, bname, bq in bn:
        if be <= s:
            continue
        if bs >= e:
            break
        o = min(e, be) - max(s, bs)
        if o > 0:
            ov += o
            pairs.append((s, e, n, q, bs, be, bname, bq, o))
    a = agg.setdefault(n, [0, 0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += ov
    a[3] += int(ov > 0)
print("%-88s %7s %11s %14s %9s" % ("kernel (not bn_bwd_fused)", "calls", "total ms", "overlap ms", "calls ovl"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    if a[2] == 0 and not any(k in n.lower() for k in ("nccl", "rccl", "onerank", "squatter")):
        continue
    print("%-88s %7d %11.3f %14.3f %9d" % (n[:88], a[0], a[1] / 1e6, a[2] / 1e6, a[3]))
print("\nfirst overlapping pairs (times in us since the first launch of the trace):")
shown = collections.Counter()
for s, e, n, q, bs, be, bname, bq, o in pairs:
    if shown[n] >= 6:
        continue
    shown[n] += 1
    print("  %-50s queue %-3s [%12.1f .. %12.1f]   x   %-40s queue %-3s [%12.1f .. %12.1f]   overlap %9.1f us"
          % (n[:50], q, (s - t0) / 1e3, (e - t0) / 1e3, bname[:40], bq, (bs - t0) / 1e3, (be - t0) / 1e3, o / 1e3))
