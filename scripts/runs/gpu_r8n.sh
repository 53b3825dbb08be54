#!/bin/bash
# round 6, call n: body + optimizer step as ONE graph for one process / one chunk (no gradient pack copy): trainer, dist,
# 128x1024 and the remaining suites the -x stop of call m did not reach; A/B bench
O=gpurun_out/r8n; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_dist.py tests/test_gpu_fp8.py tests/test_gpu_integration_stubs.py tests/test_gpu_kitti.py tests/test_gpu_pointcloud.py tests/test_gpu_model.py tests/test_gpu_bench_probes.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
timeout 900 python -m pytest tests/test_gpu_128x1024.py -x -q -m gpu -k "not float64_oracle and not fp32_g_and_d" > $O/tests128.txt 2>&1; echo "tests128 rc=$?"; tail -3 $O/tests128.txt
for v in "" "DGV2_NO_FUSED_OPT=1" "" "DGV2_NO_FUSED_OPT=1"; do
  env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench $v', d['value'], d['ms_per_step'])"
done
tail -3 $O/bench.err
