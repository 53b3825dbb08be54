"""Launches the kernels whose bound DESIGN.md argues about (5 launches each after a warm one), for SQ counter passes:
  conv_pipe_kernel at D block-0 conv1 (128 x 64x512, 32->32)  and at 128 x 8x64, 256->256;  modconv_pe_fwd_kernel level 4.

  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES \
            --kernel-trace --output-format csv -d out_sq1 -- python3 scripts/sq_probe.py
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES \
            --kernel-trace --output-format csv -d out_sq2 -- python3 scripts/sq_probe.py
  python scripts/sq_summary.py out_sq1 out_sq2 > profiles/round2_sq_counters.txt
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import argparse
import torch
import bench
from gans.models.ops import native

args = argparse.Namespace(batch_per_gpu=64, dtype="bf16")
print(bench.roofline_probe(args, reps=5))
print(bench.modconv_probe(args, reps=5))
g = native.ConvGeom(3, 3, 1, 1, True)
x = torch.randn(128, 8, 64, 256, device="cuda", dtype=torch.bfloat16)
w = torch.randn(256, 3, 3, 256, device="cuda", dtype=torch.bfloat16)
bias = torch.randn(256, device="cuda")
for _ in range(8):
    native._conv_fwd_raw(x, w, g, bias, 3, 0.2, 2.0 ** 0.5)
torch.cuda.synchronize()
