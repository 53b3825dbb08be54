#!/bin/bash
# round 5, call c: V4 mod_prep kernels (tests + A/B microbench), full GPU suite, bench, kernel trace
O=gpurun_out/r6d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "prep or rng or nsgan or modconv or mod_" > $O/test_prep.txt 2>&1; echo "prep tests rc=$?"; tail -3 $O/test_prep.txt
(python scripts/mb_modprep.py; DGV2_NO_PREP_V4=1 python scripts/mb_modprep.py) 2>&1 | grep -v amdgpu.ids > $O/mb_modprep.txt; cat $O/mb_modprep.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/test_all.txt 2>&1; echo "gpu suite rc=$?"; tail -5 $O/test_all.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6d/bench_default.log') if l.startswith('{')][-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('ms_plain_iteration'), d['extra'].get('ms_r1_iteration'))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_listing.py $f --full > $O/listing.txt; head -34 $O/listing.txt
