"""Collect per-launch HBM traffic of the roofline kernels from two rocprofv3 PMC passes (see pmc_probe.py).
FETCH_SIZE is doubled: on gfx950 it reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)."""
import csv, glob, json, sys

def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    out = {}
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
# "alias=substring": record under `alias` the kernel whose name contains `substring`
for key in (sys.argv[3:] or ("conv_x3_kernel_dgrad=conv_x3_kernel", "conv_pipe_kernel_s2dgrad=conv_pipe_kernel!float", "conv3x3_strip_kernel", "modconv_pe_fwd_kernel",
                             "modconv_up_kernel", "modconv_up_tl_kernel", "modconv_up_t_kernel", "up2_lag_sumsq_kernel")):
    key, _, sub = key.partition("=")
    sub = sub or key
    sub, _, excl = sub.partition("!")          # "substring!exclude": names containing `exclude` do not count
    if excl:
        fetch_v = {k: v for k, v in fetch.items() if excl not in k}
        write_v = {k: v for k, v in write.items() if excl not in k}
    else:
        fetch_v, write_v = fetch, write
    if sub.endswith("*"):   # every instance whose name contains the substring, each under its own name
        for k in sorted(k for k in fetch if sub[:-1] in k and k in write):
            fv, wv = fetch[k], write[k]
            f_raw, w_raw = sum(fv) / len(fv), sum(wv) / len(wv)
            res[k[:150]] = {"launches": len(fv), "FETCH_SIZE_KB_raw": f_raw, "WRITE_SIZE_KB_raw": w_raw,
                            "fetch_bytes_corrected": 2 * f_raw * 1024, "write_bytes": w_raw * 1024,
                            "traffic_bytes_per_launch": 2 * f_raw * 1024 + w_raw * 1024}
        continue
    fk = [k for k in fetch_v if sub in k and (sub != "modconv_up_kernel" or "modconv_up_t" not in k)]
    wk = [k for k in write_v if sub in k and (sub != "modconv_up_kernel" or "modconv_up_t" not in k)]
    if not fk or not wk:
        continue
    fv, wv = fetch[fk[0]], write[wk[0]]
    f_raw, w_raw = sum(fv) / len(fv), sum(wv) / len(wv)   # rocprofv3 reports KB
    res[key] = {"launches": len(fv), "FETCH_SIZE_KB_raw": f_raw, "WRITE_SIZE_KB_raw": w_raw,
                "fetch_bytes_corrected": 2 * f_raw * 1024, "write_bytes": w_raw * 1024,
                "traffic_bytes_per_launch": 2 * f_raw * 1024 + w_raw * 1024,
                "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)"}
print(json.dumps(res, indent=1))
