"""Kernel time of the R1 body alone: per-kernel (calls, total) of a run with R1 on every iteration minus a run without.
   python scripts/r1_diff.py <stats_with.csv> <stats_without.csv> <iterations>"""
import csv, sys, collections
def load(p):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(p)):
        d[r["Name"]][0] += int(r["Calls"]); d[r["Name"]][1] += float(r["TotalDurationNs"])
    return d
a, b, n = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows = []
for k in a:
    dc, dt = a[k][0] - b.get(k, [0, 0])[0], a[k][1] - b.get(k, [0, 0.0])[1]
    rows.append((dt / n / 1e3, dc / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"R1 body: {tot / 1e3:.2f} ms of kernel time per iteration")
for t, c, k in rows[:45]:
    print(f"{t:8.1f} us {c:6.1f} launches  {k[:170]}")
