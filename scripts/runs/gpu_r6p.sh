#!/bin/bash
# round 5, call p: what changed the long run's dynamics (rounds 3 / 4: D flat at 2 ln 2 on the synthetic scans; now D separates)?
O=gpurun_out/r6p; mkdir -p $O
run() { echo "=== $1"; shift; (timeout 600 env "$@" python scripts/long_run.py 1200 2>&1 | grep -v amdgpu | cut -c1-260) ; }
run "shipped" A=1 | tee $O/lr_shipped.txt
run "DGV2_NO_R1_BANK=1" DGV2_NO_R1_BANK=1 | tee $O/lr_no_r1_bank.txt
run "DGV2_TORCH_RNG=1" DGV2_TORCH_RNG=1 | tee $O/lr_torch_rng.txt
run "DGV2_TORCH_RNG=1 DGV2_NO_R1_BANK=1" DGV2_TORCH_RNG=1 DGV2_NO_R1_BANK=1 | tee $O/lr_both.txt
