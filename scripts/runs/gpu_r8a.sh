#!/bin/bash
# round 6, call a: baseline of the round -- new tests (probe guards, bank reuse, checkpoint), the trainer / dist suites
# the advisor fixes touch, then the refresh script on this tree (fresh directory, stamped trace)
O=gpurun_out/r8a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_probes.py tests/test_gpu_trainer.py tests/test_gpu_dist.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests.txt
bash scripts/refresh_profiles.sh r8a_refresh
