#!/bin/bash
# round 5, call i: rng kernel without fences, stem + skip-blur gradient fusion, one-rank SUM; A/B of the stem fusion
O=gpurun_out/r6i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_full.py tests/test_gpu_trainer.py -x -q -m gpu > $O/test_all.txt 2>&1; echo "gpu suite rc=$?"; tail -4 $O/test_all.txt
timeout 120 python scripts/mb_rng.py 2>&1 | grep -v amdgpu.ids | tee $O/mb_rng.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_noextra.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_noextra.log') if l.startswith('{')][-1]); print('plain (no extra) line', round(d['value'],1), round(d['ms_per_step'],3))"
DGV2_NO_STEM_SKIP=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_ab_no_stem_skip.log 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_ab_no_stem_skip.log') if l.startswith('{')][-1]); print('A/B without the stem skip fusion', round(d['value'],1), round(d['ms_per_step'],3))"
DGV2_DIST_WORLD1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_one_rank_rccl.log 2> $O/bench_one_rank_rccl.err; python -c "
import json
try:
    d=json.loads([l for l in open('$O/bench_one_rank_rccl.log') if l.startswith('{')][-1]); print('one-rank rccl', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('captured_collectives'))
except Exception as e: print('rccl ERR', e)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r6i/bench_default.log') if l.startswith('{')][-1])
    print('bench', round(d['value'],1), round(d['ms_per_step'],3), d['extra'].get('ms_plain_iteration'), d['extra'].get('ms_r1_iteration'))
except Exception as e: print('bench ERR', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_listing.py $f --full --json $O/step_instances.json > $O/listing.txt; head -34 $O/listing.txt
