"""profiles/round6_bf16_vs_f64_table.txt: per-tensor error of the timed (bf16) mode's gradients against the float64 oracle
through the real objectives (G step, D step, lazy R1) at 64x512, full widths, B = 4 -- the table
tests/test_gpu_full.py::test_bf16_gradients_against_the_float64_oracle_per_tensor bounds."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import conftest  # noqa: E402
import test_gpu_full as T  # noqa: E402
d = conftest.load_golden("model_full.npz")
angle = conftest.load_golden("coords.npz")["angle_64x512"].cuda()
print(T.format_bf16_table(*T.bf16_vs_f64_rows(d, angle, B=int(sys.argv[1]) if len(sys.argv) > 1 else 4)))
