"""Stamp the committed kernel statistics with the hash of the HIP sources they were measured on (bench.stats_current()).
    python scripts/stats_meta.py            # after copying a fresh rocprofv3 kernel_stats.csv to bench.STATS_FILE"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
meta = os.path.join(ROOT, bench.STATS_FILE[:-4] + ".meta.json")
json.dump({"src_sha16": bench.kernel_source_hash(),
           "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 5"},
          open(meta, "w"), indent=1)
print(open(meta).read())
