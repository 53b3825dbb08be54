import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
import bench
n = int(sys.argv[1]); b = int(sys.argv[2])
orig = torch.set_num_threads
torch.set_num_threads = lambda k: orig(n)   # force the thread count
t0 = time.time()
print(bench.cpu_baseline(b, steps=1), flush=True)
print("threads", n, "batch", b, "wall", time.time() - t0, flush=True)
