"""Compare two rocprofv3 kernel_stats.csv files per kernel: time per iteration of a shard run against 1/ratio of a
large-batch run (where does the small shard lose?), or before/after of one workload.
    python tools/stats_diff.py A.csv itersA B.csv itersB [scaleA]     (scaleA: divide A's times, e.g. 8 for bs128 vs bs16)"""
import csv
import re
import sys


def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "")).replace("void ", "")
        c, t = d.get(n, (0, 0.0))
        d[n] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]) / 1e6)
    return d


def main():
    a, ia, b, ib = load(sys.argv[1]), float(sys.argv[2]), load(sys.argv[3]), float(sys.argv[4])
    sc = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
    ta = sum(v[1] for v in a.values()) / ia / sc
    tb = sum(v[1] for v in b.values()) / ib
    print("A: %.2f ms/iter (scaled), %d launches/iter;  B: %.2f ms/iter, %d launches/iter" % (
        ta, sum(v[0] for v in a.values()) / ia, tb, sum(v[0] for v in b.values()) / ib))
    rows = []
    for k in set(a) | set(b):
        ca, tA = a.get(k, (0, 0.0))
        cb, tB = b.get(k, (0, 0.0))
        rows.append((tB / ib - tA / ia / sc, k, ca / ia, tA / ia / sc, cb / ib, tB / ib))
    rows.sort(reverse=True)
    n = int(sys.argv[6]) if len(sys.argv) > 6 else 40
    for ex, k, ca, tA, cb, tB in rows[:n]:
        print("%-62s B-A=%7.3f   A: %5.0f %7.3f   B: %5.0f %7.3f" % (k[:62], ex, ca, tA, cb, tB))
    print("...")
    for ex, k, ca, tA, cb, tB in rows[-10:]:
        print("%-62s B-A=%7.3f   A: %5.0f %7.3f   B: %5.0f %7.3f" % (k[:62], ex, ca, tA, cb, tB))


if __name__ == "__main__":
    main()
