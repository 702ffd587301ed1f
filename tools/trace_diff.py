"""Per-kernel difference of two tools/trace_calls.py files, the second divided by <ratio> (per-shard-image time).
usage: python tools/trace_diff.py calls_small.csv calls_big.csv 8"""
import csv, collections, sys
def load(p):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        d[r["kernel"]].append(float(r["us"]))
    return d
a, b, q = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
ta = sum(map(sum, a.values())) / 1e3
tb = sum(map(sum, b.values())) / 1e3 / q
print("kernel time per iteration: %.2f ms vs %.2f ms (big / %g); launches %d vs %d" % (ta, tb, q, sum(map(len, a.values())), sum(map(len, b.values()))))
rows = sorted(((sum(a.get(k, [])) / 1e3 - sum(b.get(k, [])) / 1e3 / q, k) for k in set(a) | set(b)), reverse=True)
for d, k in rows[: int(sys.argv[4]) if len(sys.argv) > 4 else 36]:
    x, y = a.get(k, []), b.get(k, [])
    print("%+7.3f ms  %-64s %7.3f vs %7.3f  calls %3d %3d  avg %6.1f us" % (d, k[:64], sum(x) / 1e3, sum(y) / 1e3 / q, len(x), len(y), sum(x) / max(1, len(x))))
