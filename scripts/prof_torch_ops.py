"""torch.profiler view of one eager training iteration: which ATen ops make up the small-launch glue."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
from torch.profiler import profile, ProfilerActivity
args = argparse.Namespace(batch_per_gpu=64, dtype="bf16", ada_p=0.6, no_graph=True, res="64x512", d_epilogue="fp32")
from gans.trainer import Trainer
from gans.utils import init_random_seed
init_random_seed(0, 0)
cfg = bench.make_cfg(args, 0, 1)
tr = Trainer(cfg, sync_scalars=False)
for it in (16, 1, 2):
    tr.step(it)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(3)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total device us {tot:.0f}")
for t, c, k, sh in rows[:90]:
    print(f"{t:9.0f}us x{c:4d} {k[:44]:44s} {sh}")

print("---- every ATen op with device time, by shape ----")
for t, c, k, sh in rows:
    if k.startswith("aten::"):
        print(f"{t:9.0f}us x{c:4d} {k[:28]:28s} {sh}")
print("---- ATen ops by name ----")
agg = {}
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key:
        agg[e.key] = (t, e.count)
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t:9.0f}us x{c:5d} {k}")

print("---- small ATen ops by call site ----")
import collections, re
site = collections.Counter(); sitet = collections.Counter()
for e in prof.events():
    if e.name in ("aten::mul", "aten::fill_", "aten::copy_", "aten::add_", "aten::zero_", "aten::add", "aten::cat", "aten::sum", "aten::div"):
        st = [f for f in (e.stack or []) if "dusty-gan-v2_amd" in f or "bench.py" in f]
        key = (e.name, " <- ".join(re.sub(r".*dusty-gan-v2_amd/", "", f)[:60] for f in st[:3]))
        site[key] += 1
        sitet[key] += getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
for k, c in site.most_common(45):
    print(f"x{c:4d} {sitet[k]:8.0f}us {k[0]:12s} {k[1]}")
