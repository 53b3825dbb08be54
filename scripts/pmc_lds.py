"""Launches the LDS-staged MFMA kernels whose bound DESIGN 14.7 / 14.8 argues about (conv_x3 forward with and without the
x_exact promise, its data and weight gradient, conv8 at 8x64 256->256), five launches each, for counter passes:
    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d out_a -- python3 scripts/pmc_lds.py
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d out_b -- python3 scripts/pmc_lds.py
    python scripts/pmc_lds.py collect out_a out_b"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "collect":
    import csv, glob, json, collections
    res = collections.defaultdict(dict)
    for d in sys.argv[2:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not any(s in k for s in ("conv_x3_kernel", "conv_wgrad_x3_kernel", "conv8_kernel")):
                    continue
                res[k[:90]].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for k, c in res.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0            # summed over the 8 XCDs
        o = dict(launches=len(next(iter(c.values()))), **m)
        if cyc:
            o["gpu_cycles_per_launch"] = cyc
            # SQ counters are summed over the 256 CUs; MFMA busy is per SIMD-pipe cycle
            if "SQ_LDS_IDX_ACTIVE" in m: o["lds_array_active_frac"] = m["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc
            if "SQ_LDS_BANK_CONFLICT" in m: o["lds_conflict_frac_of_active"] = m["SQ_LDS_BANK_CONFLICT"] / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m: o["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (256.0 * 4) / cyc
        out[k] = o
    print(json.dumps(out, indent=1))
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
from gans.models.ops import native as nat
B, H, W, C, Cp, O = 128, 4, 32, 513, 528, 512
g = nat.ConvGeom(3, 3, 1, 1, True)
w = torch.randn(O, C, 3, 3, device="cuda") / 64
(wf, wt, w3, w3t), = nat.conv_weight_bank([(w, 1.0, Cp)], torch.float32, image8=[True])
w3t._dgv2_clive = C
x = torch.randn(B, H, W, Cp, device="cuda"); x[..., C:] = 0
xe = x.clone(); xe[..., :512] = xe[..., :512].bfloat16().float()
gy = torch.randn(B, H, W, O, device="cuda")
bias = torch.randn(O, device="cuda")
wr = wf.reshape(O, 3, 3, Cp)
for _ in range(5):
    nat._conv_fwd_raw(x, wr, g, bias, 3, 0.2, 1.4, w8=w3)
    nat._conv_fwd_raw(xe, wr, g, bias, 3, 0.2, 1.4, w8=w3, xexact=512)
    nat._conv_dgrad_raw(gy, None, g, (B, H, W, Cp), wt=wt, w8t=w3t)
    nat._conv_wgrad_raw(gy, x, g, 0.01, x3=C)
    nat._conv_wgrad_raw(gy, xe, g, 0.01, x3=C, xexact=512)
w2 = torch.randn(256, 256, 3, 3, device="cuda") / 48
(f2, t2, i8, i8t), = nat.conv_weight_bank([(w2, 1.0, 256)], torch.bfloat16, image8=[True])
x2 = torch.randn(B, 8, 64, 256, device="cuda").bfloat16()
b2 = torch.randn(256, device="cuda")
for _ in range(5):
    nat._conv_fwd_raw(x2, f2.reshape(256, 3, 3, 256), g, b2, 3, 0.2, 1.4, w8=i8)
torch.cuda.synchronize()
