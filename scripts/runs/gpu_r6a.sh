#!/bin/bash
# round 5, call a: status-word ABI (45) on the GPU + a baseline bench line and kernel trace of this box
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "x3 or fir or conv_weight_bank" > $O/test_ops.txt 2>&1; echo "ops rc=$?"; tail -2 $O/test_ops.txt
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_full.py -x -q -m gpu > $O/test_tr.txt 2>&1; echo "trainer/full rc=$?"; tail -2 $O/test_tr.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; tail -c 1500 $O/bench_default.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_listing.py $f --full > $O/listing.txt; head -40 $O/listing.txt
