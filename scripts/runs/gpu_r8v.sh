#!/bin/bash
# round 6, call v: Adam without the read of the first moment at beta1 = 0 -- test, trainer fixtures, same-call A/B is not
# possible (one library): bench only
O=gpurun_out/r8v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_trainer.py -x -q -m gpu -k "adam or reference_trainer or run_and_train" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench.log') if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'])"; done
timeout 120 python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "dusty-gan-v2_amd")]
import torch, bench
from gans.models.ops import native
ps = [torch.randn(512, 65536, device="cuda").requires_grad_(True), torch.randn(5_000_000, device="cuda").requires_grad_(True)]
for p in ps: p.grad = torch.randn_like(p)
for b1 in (0.0, 0.9):
    opt = torch.optim.Adam(ps, lr=2e-3, betas=(b1, 0.99), capturable=True)
    native.fused_adam_step(opt)
    t = bench._time_launches(lambda: native.fused_adam_step(opt), 20)
    print(f"adam step over 38.5 M parameters, beta1 = {b1}: {t*1e6:.1f} us")
PY
