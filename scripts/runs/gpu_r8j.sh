#!/bin/bash
# round 6, call j: conv2 data gradient + conv1 activation backward in one launch (generator levels 4 / 3 / 2), the
# twice-differentiable Linear on gemm_x3, conv8 stride-1 ablation (the MFMA share a Winograd form could cut)
O=gpurun_out/r8j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dgrad_with_upstream or linear_f32 or d_tail" > $O/tests_new.txt 2>&1; echo "new tests rc=$?"; tail -3 $O/tests_new.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py tests/test_gpu_full.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
for v in "" "DGV2_NO_DGRAD_ACTBWD=1"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench$v.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench$v.log') if l.startswith('{')][-1]); print('bench $v', d['value'], d['ms_per_step'])"
done
timeout 600 python scripts/mb_conv8_ablate.py 0 2 46 > $O/conv8_ablation_s1.txt 2>&1; cat $O/conv8_ablation_s1.txt | grep -v amdgpu
