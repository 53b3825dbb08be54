"""Aggregate a rocprofv3 kernel_stats.csv into per-engine buckets (ms per training iteration).
The number of iterations the profiled process ran is COUNTED from the file: every iteration launches the fused Adam
kernel exactly twice (G and D) plus once more on a lazy-R1 iteration (every 16th), so iterations = Adam launches /
(2 + 1/16).  A second argument overrides the count.
usage: python scripts/prof_buckets.py <kernel_stats.csv> [iterations]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
adam = sum(int(r["Calls"]) for r in rows if "adam_step_kernel" in r["Name"])
if len(sys.argv) > 2:
    steps = float(sys.argv[2])
elif adam:
    steps = adam / (2.0 + 1.0 / 16.0)
else:
    raise SystemExit("no adam_step_kernel launches in the file: pass the iteration count")
print(f"# {adam} Adam launches -> {steps:.1f} training iterations in the profiled process (warm-up, capture, timed and "
      f"extra-measurement iterations alike; probe launches outside iterations are included in the totals)")
B = collections.OrderedDict([
    ("conv_x3 (fp32 epilogue conv)", r"conv_x3_|x3_dgrad_tail|conv_wgrad_x3|x3_image"), ("conv_pipe", r"conv_pipe_kernel"), ("conv8", r"conv8_kernel"), ("conv_strip", r"conv3x3_strip"), ("conv_direct_fallback", r"conv_direct_kernel"),
    ("conv_wgrad_stream", r"wgrad_stream|wgrad_reduce"), ("conv_wgrad_direct", r"wgrad_direct"), ("gemm_tn(im2col wgrad)", r"gemm_tn_kernel.*Im2col"),
    ("gemm_tn", r"gemm_tn_kernel"), ("gemm_nn(conv)", r"gemm_nn_kernel.*Im2col"), ("gemm_nn", r"gemm_nn_kernel"),
    ("fir_mfma", r"fir_same_mfma"), ("resample", r"resample"), ("upfirdn/ada", r"upfirdn|ada_"), ("mod_prep", r"mod_prep"),
    ("bias_act", r"bias_act|bias_grad"), ("sumsq", r"sum_squares"), ("tail/fourier/coords", r"gen_tail|fourier|coords|downsample_angle"),
    ("modconv_pe", r"modconv_pe"), ("modconv_up", r"modconv_up|up2_lag"), ("pe_wgrad", r"pe_wgrad"), ("gemm_x3", r"gemm_x3"), ("stem", r"stem_"), ("adam/lerp", r"adam_|lerp_list"), ("ema/pack/bank", r"ema_scalar|pack2d|weight_bank"),
    ("zero", r"dgv2_zero"), ("blas", r"Cijk|rocblas|hipblas"), ("torch_other", r".*")])
agg = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    name = r["Name"]; t = float(r["TotalDurationNs"]); c = int(r["Calls"])
    for k, pat in B.items():
        if re.search(pat, name):
            agg[k][0] += t; agg[k][1] += c; break
tot = sum(v[0] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:28s} {v[0]/1e6/steps:8.3f} ms/step  {v[1]/steps:8.1f} launches/step")
print(f"{'TOTAL':28s} {tot/1e6/steps:8.3f} ms/step")
others = [(float(r['TotalDurationNs']), r['Name'], int(r['Calls'])) for r in rows if not any(re.search(p, r['Name']) for p in list(B.values())[:-1])]
for t, n, c in sorted(others, reverse=True)[:14]:
    print(f"   other: {t/1e6/steps:7.3f} ms/step x{c/steps:6.1f}  {n[:110]}")
