#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python scripts/prof_torch_ops.py > $O/torch_ops.txt 2>&1
grep -n "ATen ops by name" -A32 $O/torch_ops.txt | cut -c1-120
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r0 -- python3 $R/scripts/prof_r1.py 0 > $R/$O/prof_r0.log 2>&1
cd $R
cp $(find $O/prof_r0 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_r0.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5b/kernel_stats_r0.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per iteration:", tot/14/1e6, "launches/it", sum(int(r['Calls']) for r in rows)/14)
at=[(float(r['TotalDurationNs'])/14/1e3,int(r['Calls'])/14,r['Name']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'] or 'Cijk' in r['Name']]
print("ATen+lib us/it", sum(a[0] for a in at), "launches", sum(a[1] for a in at))
for t,c,n in sorted(at,reverse=True)[:16]: print(f"{t:7.1f} us {c:5.1f}x {n[:150]}")
PY
