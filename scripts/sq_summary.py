"""Per-kernel, per-grid averages of the SQ counters of one or more rocprofv3 --pmc passes, with the derived ratios
DESIGN.md quotes.  usage: sq_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_pipe" in k or "modconv_pe" in k or "conv3x3" in k or "modconv_up" in k:
                agg[(k.split("(")[0][:70], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"{k}  grid={g}  launches={len(next(iter(cs.values())))}")
    for c in sorted(m):
        print(f"    {c:28s} {m[c]:16.0f}")
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                print(f"    {c + ' / SQ_WAVE_CYCLES':40s} {m[c] / wc:8.3f}")
    if "SQ_INSTS_MFMA" in m:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in m:
                print(f"    {c + ' per MFMA':40s} {m[c] / m['SQ_INSTS_MFMA']:8.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        print(f"    {'MFMA busy / SQ busy cycles':40s} {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:8.3f}")
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        print(f"    {'LDS bank conflict / LDS active':40s} {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):8.3f}")
