"""Idle time between kernels in a rocprofv3 kernel trace of scripts/prof_r1.py 0 (no R1): busy / span over the last
iterations, gap histogram, the largest gaps with the kernels either side.
   rocprofv3 --kernel-trace --output-format csv -d out -- python3 scripts/prof_r1.py 0 ; python scripts/gap_report.py out"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steady state: the last 40 % of the launches
rows = rows[int(len(rows) * 0.6):]
span = rows[-1][1] - rows[0][0]
busy = 0
gaps = []
end = rows[0][0]
for s, e, n in rows:
    if s > end:
        gaps.append((s - end, n))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print(f"launches {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)  idle {(span - busy) / 1e6:.2f} ms")
h = collections.Counter()
tot = collections.Counter()
for g, _ in gaps:
    b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else "<200us" if g < 200000 else ">=200us"
    h[b] += 1; tot[b] += g
for b in ("<2us", "<5us", "<10us", "<50us", "<200us", ">=200us"):
    print(f"  gaps {b:8s}: {h[b]:6d}  {tot[b] / 1e6:8.3f} ms")
prev = {}
end = rows[0][0]; last = rows[0][2]
big = []
for s, e, n in rows:
    if s - end > 10000:
        big.append((s - end, last[:70], n[:70]))
    if e > end:
        end, last = e, n
big.sort(reverse=True)
agg = collections.Counter(); cnt = collections.Counter()
for g, a, b in big:
    agg[(a, b)] += g; cnt[(a, b)] += 1
for (a, b), g in agg.most_common(25):
    print(f"  {g / 1e3:9.1f} us in {cnt[(a, b)]:3d} gaps  after {a}  before {b}")
