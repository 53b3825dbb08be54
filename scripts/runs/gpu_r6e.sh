#!/bin/bash
# round 5, call e: fused head backward + software-pipelined prep backward: tests, bench (new warm-up / probes), trace
O=gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py -x -q -m gpu > $O/test_sel.txt 2>&1; echo "selected tests rc=$?"; tail -3 $O/test_sel.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6e/bench_default.log') if l.startswith('{')][-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('ms_plain_iteration'), d['extra'].get('ms_r1_iteration'))
print('modconv levels', d['roofline_modconv']['levels'])
print('roofline', d['roofline']['selected_by'], d['roofline']['frac'])
PY
tail -3 $O/bench_default.err
python bench.py --workload gfwd --batch-per-gpu 32 --steps 50 --warmup 10 > $O/bench_gfwd.log 2>&1; python -c "
import json; d=json.loads([l for l in open('$O/bench_gfwd.log') if l.startswith('{')][-1]); print('gfwd', round(d['value'],1), round(d['ms_per_step'],4))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_listing.py $f --full --json $O/step_instances.json > $O/listing.txt; head -34 $O/listing.txt
