#!/bin/bash
# round 6, call d: const caches refreshed in place (mixed eager / captured bodies), the 128x1024 reference fixture on the
# HIP path, the trainer / model suites
O=gpurun_out/r8d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_bench_probes.py tests/test_gpu_model.py -x -q -m gpu > $O/tests1.txt 2>&1; echo "tests1 rc=$?"; tail -4 $O/tests1.txt
timeout 900 python -m pytest tests/test_gpu_128x1024.py -x -q -m gpu -k "reference_fixture" > $O/tests2.txt 2>&1; echo "tests2 rc=$?"; tail -12 $O/tests2.txt
