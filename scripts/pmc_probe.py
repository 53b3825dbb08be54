"""Launches only the roofline kernels of bench.py (5 launches each) so that a
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass attributes HBM traffic to them.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python3 scripts/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 scripts/pmc_probe.py
    python scripts/pmc_collect.py out_f out_w > profiles/round4_pmc.json
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import argparse
import torch
import bench

args = argparse.Namespace(batch_per_gpu=64, dtype="bf16")
print(bench.x3_probe(args, reps=5))            # conv_x3_kernel<1> (+ the exact-fp32 kernel it is timed beside)
print(bench.s2dgrad_probe(args, reps=5))      # the only conv_pipe_kernel launches of this script
print(bench.roofline_probe(args, reps=5))
print(bench.modconv_probe(args, reps=5))
