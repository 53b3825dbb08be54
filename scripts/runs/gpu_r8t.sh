#!/bin/bash
# round 6, call t: which change of the round stops the discriminator from learning on the synthetic scans (long run: D(real) ~
# D(fake) ~ 0, loss_D = 2 ln 2) -- 400 iterations per switch
O=gpurun_out/r8t; mkdir -p $O
for v in "X=0" "DGV2_TORCH_RNG=1" "DGV2_NO_FUSED_OPT=1" "DGV2_NO_D_TAIL=1" "DGV2_NO_DGRAD_ACTBWD=1" "DGV2_NO_GEMM_X3=1"; do
  echo "== $v" >> $O/switches.txt
  env $v timeout 300 python scripts/long_run.py 400 2>&1 | grep -v amdgpu.ids | grep "^[0-9]" | sed -n '2p;5p;10p' | cut -c1-260 >> $O/switches.txt
done
cat $O/switches.txt
