#!/bin/bash
timeout 300 python scripts/mb_x3_ablate.py 2>&1 | grep -v amdgpu | tee gpurun_out/mb_x3_ablate_r5p.txt
