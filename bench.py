#!/usr/bin/env python3
"""Benchmark of the dusty_v2 G+D training step on MI355X (BASELINE.json metric:
range-images/sec for one G+D step, dusty_v2 64x512).

    python bench.py --gpus 1 --steps 16 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full training iteration of gans/trainer.py: G step, D step, lazy R1 on every
16th iteration, G_ema update, ADA p update every 4th iteration, Adam updates, gradient
all-reduce over RCCL when N > 1.  Inputs are synthetic and resident in HBM (z ~ N(0,1), real
range images from the synthetic depth generator).  Weak scaling: per-GPU batch fixed at 64.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work per image, SURVEY.md section 8(d) (2 FLOP per MAC)
GFLOP_PER_IMG_ITER = 47.7 + 27.0 / 16  # G step 17.6 + D step 29.5 + R1 (~27) amortised over 16
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--dtype", choices=["bf16", "fp32", "fp8"], default="bf16",
                    help="bf16: bf16 trunks (the BASELINE metric); fp32: parity mode; fp8: bf16 trunks with the branch "
                         "operands of the discriminator's ResidualBlocks as OCP e4m3 on v_mfma_f32_16x16x32_fp8_fp8 "
                         "(BASELINE configs[4], quoted at --res 128x1024 --batch-per-gpu 32; DESIGN.md)")
    ap.add_argument("--ada-p", type=float, default=0.6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying hipGraphs")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--workload", choices=["train", "gfwd"], default="train",
                    help="train: the G+D iteration (BASELINE metric, configs[2]); gfwd: generator-only forward "
                         "(configs[1], quoted at --batch-per-gpu 32), eval mode, no graph")
    ap.add_argument("--res", default="64x512", help="HxW of the range image (BASELINE configs[4] runs 128x1024; "
                    "the metric and the roofline probes are quoted on the default 64x512)")
    ap.add_argument("--d-epilogue", choices=["fp32", "bf16"], default="fp32",
                    help="precision of the discriminator's epilogue (mbstd, 3x3 513->512 conv, Linear 65536->512): fp32 "
                         "is what the reference runs (dusty_v2.py:394-395) and what `value` is quoted on; bf16 is an "
                         "opt-in whose throughput is reported next to it as `extra.value_d_epilogue_bf16`")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurement points (ADA p = 0, bf16 "
                    "epilogue, separately timed R1 / plain iterations)")
    ap.add_argument("--stamp", default=os.environ.get("DGV2_BENCH_STAMP"),
                    help="write {src_sha16, lib_sha16, argv} of THIS process to the given file before anything is "
                         "timed: a profiler-wrapped run leaves the identity of the kernels it traced next to its "
                         "trace, and scripts/step_listing.py copies that stamp instead of hashing whatever tree it is "
                         "later run in (scripts/refresh_profiles.sh)")
    return ap.parse_args()


def make_cfg(args, rank, world):
    from gans.config import load_config
    cfg = load_config()
    n16 = -1 if args.dtype in ("bf16", "fp8") else 0
    cfg.model.generator.synthesis_kwargs.num_fp16_layers = n16
    cfg.model.discriminator.layer_kwargs.num_fp16_layers = n16
    cfg.dataset.name = "synthetic"
    cfg.training.rank = rank
    cfg.training.num_gpus = world
    cfg.training.batch_size = args.batch_per_gpu * world
    cfg.training.batch_size_per_gpu = args.batch_per_gpu
    cfg.training.augment.p_init = args.ada_p
    cfg.training.warmup.fade_kimg = 0  # post-fade regime: no warm-up dropout / blur
    cfg.training.resume = None
    cfg.training.hip_graph = not args.no_graph
    res = [int(v) for v in getattr(args, "res", "64x512").lower().split("x")]
    if res != [64, 512]:
        cfg.model.generator.synthesis_kwargs.resolution = res
        cfg.model.discriminator.layer_kwargs.resolution = res
    return cfg


def cpu_baseline(batch, steps=4):
    """The CPU oracle (oracle/step.py, a port of the reference's CPU path pinned by golden vectors)
    timed on this host: full-size dusty_v2, fp32, ADA at p=0.6; one iteration = G step + D step + the lazy-R1 pass at its
    1/16 share + both Adam steps + the EMA update."""
    import numpy as np
    from helpers import ada_from_cfg, build_models, full_cfg
    from oracle import coords as o_coords
    from oracle import step as o_step
    import recipe

    # the oracle's many small ATen ops scale to about a dozen threads; on the 256-core GPU host an unbounded
    # pool makes every op slower than 8 threads do, so the pool is capped and `cores` reports the cap
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = full_cfg()
    G, D = build_models(cfg, "cpu")
    sdG = recipe.fill_state_dict({k: v.clone() for k, v in G.state_dict().items()}, 1234)
    sdD = recipe.fill_state_dict({k: v.clone() for k, v in D.state_dict().items()}, 4321)
    A = ada_from_cfg(cfg, 0.6)
    H, W = 64, 512
    from gans.coords import synthetic_angle_grid
    angle = torch.from_numpy(o_coords.resample_angle_grid(synthetic_angle_grid(64), H, W)).repeat(batch, 1, 1, 1)
    g = torch.Generator().manual_seed(0)
    depth = torch.rand(batch, 1, H, W, generator=g) * (80 - 1.45) + 1.45
    mask = (torch.rand(batch, 1, H, W, generator=g) < 0.85).float()
    x_real = torch.from_numpy(o_coords.fetch_reals(depth.numpy(), mask.numpy(), 1.45, 80.0))

    def one():
        z = torch.randn(batch, 512, generator=g)
        sh = torch.rand(batch, generator=g) * 6.2831853
        u = torch.rand(batch, 1, H, W, generator=g).clamp(1e-6, 1 - 1e-6)
        ada = {"G": A.sample_affine(batch, H, W), "C": A.sample_color(batch)}
        o_step.g_step(sdG, sdD, z, angle, sh, u, ada=ada)
        o_step.d_step(sdG, sdD, z, angle, sh, u, x_real, ada_real=ada, ada_fake=ada)

    one()  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt_gd = (time.perf_counter() - t0) / steps
    # the rest of an iteration: lazy R1 every 16th iteration (double backward through D and ADA),
    # the two Adam steps, the EMA generator -- each timed on the same host and added at its share
    ada1 = {"G": A.sample_affine(batch, H, W), "C": A.sample_color(batch)}
    o_step.r1_step(sdD, x_real, 16.0, ada=ada1)   # warm-up
    t0 = time.perf_counter()
    o_step.r1_step(sdD, x_real, 16.0, ada=ada1)
    dt_r1 = time.perf_counter() - t0

    def params(sd, skip):
        ps = [v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(skip)]
        for q in ps:
            q.grad = torch.randn_like(q)
        return ps
    pG, pD = params(sdG, o_step.G_BUFFER_SUFFIXES), params(sdD, o_step.D_BUFFER_SUFFIXES)
    optG = torch.optim.Adam(pG, lr=0.002, betas=(0.0, 0.99))
    optD = torch.optim.Adam(pD, lr=0.002 * 16 / 17, betas=(0.0, 0.99 ** (16 / 17)))
    ema = [q.detach().clone() for q in pG]
    for _ in range(2):   # the first step allocates the moments
        t0 = time.perf_counter()
        optG.step()
        optD.step()
        with torch.no_grad():
            torch._foreach_lerp_(ema, [q.detach() for q in pG], 1.0 - 0.9995)
        dt_opt = time.perf_counter() - t0
    dt = dt_gd + dt_r1 / 16.0 + dt_opt
    return {"value": batch / dt, "unit": "range-images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} iterations (G step + D step, fp32, B={batch}, ADA p=0.6, full-size dusty_v2 64x512) of "
                      f"oracle/step.py after 1 warm-up: {dt_gd:.2f} s; + one lazy-R1 pass at its 1/16 share ({dt_r1:.2f} s "
                      f"each), + the two Adam steps and the EMA update of torch on the same tensors ({dt_opt:.3f} s); "
                      f"{dt:.2f} s/iteration"}


def _time_launches(fn, reps):
    """Average duration of `reps` back-to-back launches, HIP events on the launching (= torch current) stream."""
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


PMC_FILE = os.path.join("profiles", "round6_pmc.json")


PMC_BATCH_PER_GPU = 64   # scripts/pmc_probe.py launches the probes at this --batch-per-gpu


def _pmc_traffic(kernel_key, batch_per_gpu=PMC_BATCH_PER_GPU):
    """HBM bytes per launch of the probe's launch from the COMMITTED PMC passes (separate `rocprofv3 --pmc FETCH_SIZE`
    / `--pmc WRITE_SIZE` runs of scripts/pmc_probe.py, which issues exactly the launches the probes below time;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  A counter pass cannot run inside this process,
    so the figure is read from the file and labelled with its source; None when no matching record is committed or when
    this run's launch size differs from the one the counters were collected on (--batch-per-gpu != 64)."""
    path = os.path.join(ROOT, PMC_FILE)
    if not os.path.exists(path) or batch_per_gpu != PMC_BATCH_PER_GPU:
        return None
    rec = json.load(open(path)).get(kernel_key)
    return None if rec is None else rec.get("traffic_bytes_per_launch")


MFMA_F32_PEAK_TFLOPS = 157.3    # v_mfma_f32_*_f32, /opt/skills/guides/MI355X_MICROARCH.md
# per-body launch listing of this command's kernel trace (scripts/step_listing.py --json; committed): every (kernel, grid)
# instance with its time per TRAINING iteration = G body + D body + R1 body / 16
STATS_FILE = "profiles/round6_step_instances.json"


def kernel_source_hash():
    """sha256 (16 hex digits) over the HIP sources the library is built from: stamps the committed statistics
    (the profiled bench.py process writes it with --stamp, scripts/step_listing.py copies it) so that a bench line says whether the ranking it selects `roofline` by was measured on THIS tree's kernels."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dusty-gan-v2_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "dusty-gan-v2_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def write_stamp(path):
    """The identity of the kernels this PROCESS runs: hash of the HIP sources beside it and of the library it loaded."""
    import hashlib
    import dgv2_native
    lib = hashlib.sha256(open(dgv2_native.LIB_PATH, "rb").read()).hexdigest()[:16]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    json.dump({"src_sha16": kernel_source_hash(), "lib_sha16": lib, "lib": os.path.relpath(dgv2_native.LIB_PATH, ROOT),
               "argv": sys.argv[1:], "pid": os.getpid()}, open(path, "w"), indent=1)


def stats_current():
    """True / False: the committed listing carries this tree's kernel_source_hash(); None: no file / no stamp."""
    path = os.path.join(ROOT, STATS_FILE)
    if not os.path.exists(path):
        return None
    sha = json.load(open(path)).get("src_sha16")
    return None if sha is None else sha == kernel_source_hash()


def dominant_instance():
    """Which probed kernel instance the committed per-body listing ranks highest: (probe key, kernel name, share of the
    training iteration in %).  An instance = (kernel, grid) with its time per training iteration (G body + D body + R1
    body / 16: the step's own ranking, whatever the warm-up of the profiled process ran); the first instance, largest
    first, that one of the probes covers decides what `roofline` reports."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), STATS_FILE)
    probes = (("conv_x3", lambda n, g: ("conv_x3_kernelILi1E" in n or "conv_x3_kernel<1" in n)),
              # rocprofv3 demangles the bf16 four-class instance badly: <bool _Accum, bLi32ELi1ELi4ELi ...>
              ("s2dgrad", lambda n, g: "conv_pipe_kernel" in n and "Li32ELi1ELi4ELi" in n),
              ("strip", lambda n, g: "conv3x3_strip_kernel" in n))
    if os.path.exists(path):
        for r in json.load(open(path))["instances"]:
            for key, pred in probes:
                if pred(r["name"], r["grid"]):
                    return key, r["name"][:96], round(r["pct"], 2)
    return "conv_x3", None, None


MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_16x16x32_bf16 dense, /opt/skills/guides/MI355X_MICROARCH.md


def x3_probe(args, reps=10):
    """`roofline_conv_x3`: conv_x3_kernel<1> (csrc/conv_x3.hip), the largest kernel instance of the round-4 statistics -- the
    data gradient of the discriminator's fp32 epilogue conv (reference: the cuDNN dgrad autograd calls for
    ops.Conv2d(513, 512, 3, 1, 1, ring), gans/models/dusty_v2.py:377, in fp32 as the reference runs it, :394-395) in the D
    step: gy [2B, 4, 32, 512] fp32 -> gx [2B, 4, 32, 528].  fp32 on the bf16 matrix cores: every operand value as three
    bf16 planes, SIX v_mfma_f32_16x16x32_bf16 products per fp32 multiply (fp32-equivalent: tests/test_gpu_ops.py
    test_conv_x3_is_fp32_equivalent).  MFMA-bound.  `achieved` = the bf16 MFMA FLOP the kernel issues for the algorithmic
    fp32 conv (6 x 2 x pixels x 9 x 512 x 512) / time, against the dense bf16 peak; `fp32_equiv_tflops` = the algorithmic
    fp32 FLOP / time, which `fp32_mfma_peak_frac` prices against the 157.3 TFLOP/s of v_mfma_f32_16x16x4_f32 (what the
    same conv was bound by in round 3: conv_pipe_kernel<float>, timed beside it here)."""
    from gans.models.ops import native
    if args.dtype == "fp32":
        return None
    B, H, W, C, Cp, O = 2 * args.batch_per_gpu, 4, 32, 513, 528, 512
    g = native.ConvGeom(3, 3, 1, 1, True)
    w = torch.randn(O, C, 3, 3, device="cuda") / 64
    (wf, wt, w3, w3t), = native.conv_weight_bank([(w, 1.0, Cp)], torch.float32, image8=[True])
    if w3t is None:
        return None
    w3t._dgv2_clive = C
    gy = torch.randn(B, H, W, O, device="cuda")
    sec = _time_launches(lambda: native._conv_dgrad_raw(gy, None, g, (B, H, W, Cp), wt=wt, w8t=w3t), reps)
    sec32 = _time_launches(lambda: native._conv_dgrad_raw(gy, None, g, (B, H, W, Cp), wt=wt), reps)
    flops = 2.0 * B * H * W * 9 * 512 * O            # the 512 channels on the matrix cores (the 513th: exact-fp32 tail kernel)
    nbytes = (B * H * W * (Cp + O)) * 4 + 3 * 512 * 9 * O * 2
    ach = 6.0 * flops / sec / 1e12
    return {"kernel": "conv_x3_kernel<1> + x3_dgrad_tail_kernel (dgv2_conv3x3_x3_dgrad: D epilogue conv data gradient, "
                      "2B x 4x32, 512 -> 513(528), 3x3 ring, fp32 as 3 bf16 planes x 6 products)",
            "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": _pmc_traffic("conv_x3_kernel_dgrad", args.batch_per_gpu),
            "traffic_source": PMC_FILE + " (rocprofv3 --pmc, committed)", "algorithmic_bytes_per_launch": nbytes,
            "achieved_counts": "ISSUED bf16 MFMA FLOPs: six v_mfma_f32_16x16x32_bf16 products per fp32 multiply (MFMA "
                               "utilisation); the algorithmic figure is fp32_equiv_tflops = achieved / 6",
            "avg_launch_us": sec * 1e6, "fp32_equiv_tflops": flops / sec / 1e12,
            "fp32_mfma_peak_frac": flops / sec / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "exact_fp32_kernel_us": sec32 * 1e6}


def s2dgrad_probe(args, reps=20):
    """`roofline_conv_s2dgrad`: the second largest instance of the committed statistics and the largest of the bf16 conv
    engine (conv_pipe_kernel<bf16, TO = 32, four output classes>, 5.3 %: every stride-2 data gradient of D) at its
    heaviest site, the stride-2 3x3 data gradient of the first ResidualBlock's conv2 in the D step (reference: the cuDNN dgrad of
    ops.Conv2d(32, 64, 3, 2, 1, ring), gans/models/dusty_v2.py:331-333): gy [2B, 32, 256, 64] -> gx [2B, 64, 512, 32],
    the four parity classes of the transposed conv and the replicate-row border terms in ONE launch
    (dgv2_conv_taps_ex).  HBM-bound (22 FLOP per byte): algorithmic bytes = gy read once + gx written once."""
    from gans.models.ops import native
    if args.dtype == "fp32":
        return None
    dt = torch.bfloat16
    B, H, W, C, O = 2 * args.batch_per_gpu, 64, 512, 32, 64
    g = native.ConvGeom(3, 3, 2, 1, True)
    gy = torch.randn(B, H // 2, W // 2, O, device="cuda", dtype=dt)
    wt3 = (torch.randn(C, 9, O, device="cuda") / 24).to(dt)
    sec = _time_launches(lambda: native._conv_dgrad_direct(gy, wt3, g, (B, H, W, C)), reps)
    nbytes = (B * (H // 2) * (W // 2) * O + B * H * W * C + C * 9 * O) * 2
    flops = 2.0 * B * (H // 2) * (W // 2) * 9 * C * O
    ach = nbytes / sec / 1e9
    return {"kernel": "conv_pipe_kernel<bf16, TO=32, 4 classes> (dgv2_conv_taps_ex: D block-0 conv2 data gradient, "
                      "2B x 32x256x64 -> 64x512x32, 3x3 stride 2 ring, one launch)",
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "traffic": _pmc_traffic("conv_pipe_kernel_s2dgrad", args.batch_per_gpu), "traffic_source": PMC_FILE + " (rocprofv3 --pmc, committed)",
            "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": sec * 1e6, "mfma_tflops": flops / sec / 1e12}


def roofline_probe(args, reps=20):
    """`roofline_strip` (round 2's headline probe, kept for continuity): the conv engine at its heaviest HBM-bound
    forward site -- the first ResidualBlock's conv1 in the D step (real + fake = 2 x batch images, 64 x 512,
    32 -> 32 channels, 3x3, ring padding, bias + lrelu fused): dgv2_conv_taps, which runs conv3x3_strip_kernel
    (csrc/conv_strip.hip) for this geometry (conv_pipe_kernel with DGV2_NO_STRIP=1).
    HBM-bound: algorithmic bytes per image = H*W*(C + O)*2 B = 4.19 MB, per launch
    2 x batch images + the 18 KB of weights."""
    from gans.models.ops import native
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    B, H, W, C, O = 2 * args.batch_per_gpu, 64, 512, 32, 32
    g = native.ConvGeom(3, 3, 1, 1, True)
    x = torch.randn(B, H, W, C, device="cuda", dtype=dt)
    w = torch.randn(O, 3, 3, C, device="cuda", dtype=dt)
    bias = torch.randn(O, device="cuda")
    sec = _time_launches(lambda: native._conv_fwd_raw(x, w, g, bias, 3, 0.2, 2.0 ** 0.5), reps)
    nbytes = (B * H * W * (C + O) + O * 9 * C) * x.element_size()
    flops = 2.0 * B * H * W * C * O * 9
    ach = nbytes / sec / 1e9
    strip = dt == torch.bfloat16 and os.environ.get("DGV2_NO_STRIP") is None
    kname = "conv3x3_strip_kernel" if strip else "conv_pipe_kernel"
    return {"kernel": kname + " (dgv2_conv_taps: D block-0 conv1 fwd, 2B x 64x512, 32->32, 3x3 ring, bias+lrelu)",
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "traffic": _pmc_traffic(kname, args.batch_per_gpu), "traffic_source": PMC_FILE + " (rocprofv3 --pmc, committed)",
            "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": sec * 1e6, "mfma_tflops": flops / sec / 1e12}


_GUARDS = []
_GUARD_WORDS = 4096          # int32 canary words behind every guarded buffer (16 KB: more than any kernel's tile row)
_GUARD_PATTERN = 0x5A5AA5A5


def guarded_empty(shape, device="cuda", dtype=torch.float32):
    """torch.empty(shape) with canary words behind it.  The probes below hand raw pointers to C entries that take no
    buffer sizes; round 5 lost a 45-minute profiler-wrapped call to `modconv_probe` allocating the T / W_s images without
    their O // 16 and O // 32 dimensions (fixed in ce3c924 by the shapes of native.mod_up_prepare, now shared as
    native.mod_up_image_shapes): dgv2_modconv_up_t wrote past both.  check_guards() raises when a canary changed."""
    n = 1
    for d in shape:
        n *= int(d)
    esz = torch.empty((), dtype=dtype).element_size()
    nbytes = (n * esz + 15) // 16 * 16
    flat = torch.empty(nbytes + 4 * _GUARD_WORDS, device=device, dtype=torch.uint8)
    flat[nbytes:].view(torch.int32).fill_(_GUARD_PATTERN)
    _GUARDS.append((flat, nbytes, tuple(shape)))
    return flat[:n * esz].view(dtype).view(*shape)


def check_guards(clear=True):
    """Raises if a kernel wrote behind one of the guarded buffers; returns how many were checked."""
    torch.cuda.synchronize()
    bad = [shape for flat, nbytes, shape in _GUARDS
           if not bool((flat[nbytes:].view(torch.int32) == _GUARD_PATTERN).all())]
    n = len(_GUARDS)
    if clear:
        _GUARDS.clear()
    if bad:
        raise RuntimeError(f"a probe's kernel wrote past the end of its buffer(s) of shape {bad}")
    return n


def modconv_levels(args):
    """`roofline_modconv` at the three pyramid levels that run the commuted kernel in the training step: level 4 (the
    headline instance, full detail), level 3 and level 2 (own algorithmic FLOPs 2 B P Ks O over the kernel's launch time)."""
    if args.dtype == "fp32":
        return None
    out = modconv_probe(args)
    out["levels"] = {"4": {k: out[k] for k in ("achieved", "frac", "avg_launch_us", "layer_tflops")}}
    for lvl, (hl, wl, Ka, O) in (("3", (16, 128, 128, 64)), ("2", (8, 64, 256, 128))):
        r = modconv_probe(args, shape=(hl, wl, Ka, O), brief=True)
        out["levels"][lvl] = r
    out["guarded_buffers_checked"] = check_guards()
    return out


def modconv_probe(args, reps=20, shape=(32, 256, 64, 32), brief=False):
    """The MFMA kernel north_star names: the modulated 1x1 conv at its heaviest site, generator level-4 conv1
    (B x 32768 pixels, K = 64 + 512 shared-PE channels, O = 32, bias + lrelu) as the training step runs it:
      dgv2_modconv_up_t_lag  T = W_a . h at 32x256 (the xa columns, commuted past the up-sampling: a quarter of the
                             pixels) AND the layer's input statistic sum up2(h)^2 as a quadratic form of h, one read of h
      dgv2_modconv_up_fwd    y = act(c * (up2(T) + W_s . PE) + bias)   <- the kernel reported (csrc/modconv_up.hip)
    `achieved` = the kernel's own algorithmic FLOPs 2*B*P*Ks*O over its launch time (the four up-sampling K-steps it
    also runs are not counted); `layer_tflops` = the whole layer's 2*B*P*(Ka+Ks)*O (604 MMAC/img) over both
    launches (`layer_us.separate_*`: the two-launch form of the low-resolution part it replaced).
    Compulsory HBM bytes of the kernel = y out + t in (the PE is batch-shared, the weights per-sample 37 KB)."""
    if args.dtype == "fp32":
        return None
    import dgv2_native as N
    from gans.models.ops import native
    from gans.models.ops.common import Resample
    B, Ks = args.batch_per_gpu, 512
    hl, wl, Ka, O = shape
    H, W = 2 * hl, 2 * wl
    P = H * W
    bf = torch.bfloat16
    spec = Resample(up=2, window=[1, 3, 3, 1], ring=True).spec
    h = torch.randn(B, hl, wl, Ka, device="cuda", dtype=bf)
    xs = torch.randn(1, H, W, Ks, device="cuda", dtype=bf)
    w = torch.randn(B, O, Ka + Ks, device="cuda", dtype=bf) / 16
    bias = torch.randn(O, device="cuda")
    cvec = torch.ones(O, device="cuda")
    # this probe calls C entries directly: every buffer a kernel WRITES is a guarded one (canary words behind it,
    # checked by check_guards() below), and the operand images are sized by the product's own shape function
    y = guarded_empty((B, H, W, O), device="cuda", dtype=bf)
    t, wimg = native.mod_up_images(B, hl, wl, Ks, O, "cuda", bf, empty=guarded_empty)

    def lowres():
        N.call("dgv2_modconv_up_t", N.ptr(t), N.ptr(wimg), N.ptr(h), N.ptr(w), N.ptr(cvec), 2.0 ** 0.5 * 0.6, B, hl, wl, Ka, Ks, O, Ka + Ks, Ka, N.BF16, N.stream())
    lowres()
    ih, ch, iw, cw = native._up_tables(spec, hl, wl, h.device)
    xsf = native.pe_frag16(xs)
    sec = _time_launches(lambda: N.call("dgv2_modconv_up_fwd", N.ptr(y), N.ptr(t), N.ptr(xsf), N.ptr(wimg), B, H, W, hl, wl, Ks,
                                        O, N.ptr(ih), N.ptr(ch), N.ptr(iw), N.ptr(cw), N.ptr(bias), N.ptr(cvec), 3, 0.2, 2.0 ** 0.5,
                                        N.BF16, None, 0, None, N.stream()), reps)
    sec_tl = _time_launches(lambda: native.mod_up_prepare(h, xs, w, spec, True, 0.2, 2.0 ** 0.5, want_stat=True), reps)
    if brief:
        fl = 2.0 * B * P * Ks * O
        return {"shape": f"B x {P} px, Ka={Ka} (low res {hl}x{wl}), PE K={Ks}, O={O}", "achieved": fl / sec / 1e12,
                "frac": fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, "avg_launch_us": sec * 1e6,
                "layer_tflops": 2.0 * B * P * (Ka + Ks) * O / (sec + sec_tl) / 1e12}
    sec_lo = _time_launches(lowres, reps)
    sec_sq = _time_launches(lambda: native.up2_lag_sumsq(h, spec), reps)
    # the same layer on the un-commuted kernel (dgv2_modconv_pe_fwd on a materialised up2(h): the full K = Ka + Ks
    # contraction in one launch; levels 3 and 2 run this kernel in the training step)
    hup = native._resample_raw(h, spec, False, (hl, wl))
    sec_pe = _time_launches(lambda: N.call("dgv2_modconv_pe_fwd_sq", N.ptr(y), N.ptr(hup), N.ptr(xs), N.ptr(w), B, P, Ka, Ks, O,
                                           N.ptr(cvec), N.ptr(bias), 3, 0.2, 2.0 ** 0.5, N.BF16, None, 0, None, N.stream()),
                            reps)
    flops = 2.0 * B * P * Ks * O
    nbytes = (B * P * O + B * hl * wl * O + P * Ks + B * O * Ks) * 2
    ach = flops / sec / 1e12
    return {"kernel": "modconv_up_kernel (dgv2_modconv_up_fwd: G level-4 conv1, B x 32768 px, PE K=512, O=32, "
                      "up2 of the low-res xa part as 4 more K-steps)",
            "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": _pmc_traffic("modconv_up_kernel", args.batch_per_gpu),
            "traffic_source": PMC_FILE + " (rocprofv3 --pmc, committed)",
            "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": sec * 1e6,
            "algorithmic_hbm_GBps": nbytes / sec / 1e9,
            "layer_tflops": 2.0 * B * P * (Ka + Ks) * O / (sec + sec_tl) / 1e12,
            "layer_us": {"modconv_up": sec * 1e6, "lowres_t_and_statistic": sec_tl * 1e6,
                         "separate_lowres_t": sec_lo * 1e6, "separate_statistic_lag_sumsq": sec_sq * 1e6},
            "uncommuted_modconv_pe_fwd": {"avg_launch_us": sec_pe * 1e6,
                                          "tflops": 2.0 * B * P * (Ka + Ks) * O / sec_pe / 1e12,
                                          "frac": 2.0 * B * P * (Ka + Ks) * O / sec_pe / 1e12 / MFMA_BF16_PEAK_TFLOPS}}


def build_trainer(args, rank, world, d_epilogue=None):
    from gans.trainer import Trainer
    cfg = make_cfg(args, rank, world)
    trainer = Trainer(cfg, sync_scalars=False)
    trainer.D.epilogue_dtype = d_epilogue or args.d_epilogue
    trainer.D.fp8_branches = args.dtype == "fp8"
    return cfg, trainer


def timed_steps(trainer, first_it, n, barrier):
    """Wall time of `n` consecutive iterations first_it, first_it+1, ... bracketed by barrier + synchronize."""
    barrier()
    t0 = time.perf_counter()
    for k in range(n):
        trainer.step(first_it + k)
    barrier()
    return time.perf_counter() - t0


def warm(trainer, cfg, n):
    # three calls at iteration 16 capture every body, the lazy-R1 one included (with hipGraphs a body needs 2 eager runs
    # + 1 capture before it replays); the requested warm-up iterations then are PLAIN ones, so that a profile of this
    # command carries the regulariser at the share the timed region gives it, not at the warm-up's
    for _ in range(3 if cfg.training.hip_graph else 1):
        trainer.step(16)
    for k in range(max(n, 1)):
        trainer.step(1 + k % 15)


def gfwd_probe_roofline(args, sec_per_batch):
    """configs[1]: the forward is dominated by dgv2_modconv_pe_fwd at level 4 (same kernel as roofline_modconv);
    whole-forward figure: algorithmic FLOPs of the generator forward (2.894 GFLOP + PE projection per
    image) over the measured batch time."""
    flops = args.batch_per_gpu * 2.894e9
    ach = flops / sec_per_batch / 1e12
    return {"kernel": "whole generator forward (modulated convs: 2.894 GFLOP/img algorithmic)", "bound": "mfma",
            "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
            "traffic": None}


def _emit(line, fd):
    """The ONE JSON line of the contract, on the process's real stdout."""
    os.write(fd, (line + "\n").encode())


def main():
    args = parse()
    # stdout carries the JSON line and nothing else: RCCL prints a version banner on stdout when its first communicator
    # is created (seen with 2.26.6), libraries may print more.  File descriptor 1 points at stderr for the duration of
    # the run; the line goes out on a duplicate of the original.
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # DGV2_DIST_WORLD1=1 (see gans/parallel.py): the launcher path and every collective on RCCL with ONE rank -- a
    # functional check of the N > 1 code on a one-GPU box, never a measurement of scaling
    dist_on = world > 1 or bool(os.environ.get("DGV2_DIST_WORLD1"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # DGV2_DIST_SMOKE=1: functional test of the N > 1 path on a ONE-GPU box (every rank on cuda:0, gloo
        # collectives); never a measurement
        if os.environ.get("DGV2_DIST_SMOKE"):
            os.environ["LOCAL_RANK"] = "0"
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local)
            from gans import parallel   # RCCL's kernels on a high-priority stream (see parallel.init_process_group)
            parallel.init_process_group("nccl", device=torch.device("cuda", local))
    from gans.utils import init_random_seed

    init_random_seed(0, rank)
    if args.stamp and rank == 0:
        write_stamp(args.stamp)
    cfg, trainer = build_trainer(args, rank, world)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        """(max over ranks, list of every rank's own time)"""
        if not dist_on:
            return dt, [dt]
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        every = [float(v) for v in every]
        return max(every), every

    if args.workload == "gfwd":   # BASELINE configs[1]: generator-only forward, replayed as ONE hipGraph
        G = trainer.G_ema.eval()
        z = trainer.sample_z(args.batch_per_gpu)
        graph = None
        with torch.no_grad():
            for _ in range(max(args.warmup, 2)):
                G(z, **trainer.auxin)
            if not args.no_graph:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = G(z, **trainer.auxin)
                graph.replay()
            run = graph.replay if graph is not None else (lambda: G(z, **trainer.auxin))
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run()
            barrier()
        dt, every = max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            _emit(json.dumps({
                "metric": f"range-images/sec (generator forward) on dusty_v2 {args.res}",
                "value": args.steps * args.batch_per_gpu * world / dt, "unit": "range-images/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": f"configs[1]: dusty_v2 generator-only forward (eval, EMA weights), {args.res}",
                           "global_batch": args.batch_per_gpu * world, "per_gpu_batch": args.batch_per_gpu,
                           "parallelism": f"dp{world}", "hip_graph": graph is not None},
                "roofline": gfwd_probe_roofline(args, dt / args.steps) if args.res == "64x512" else None,
                "roofline_modconv": modconv_probe(args)}), out_fd)
        if dist_on:
            dist.barrier()   # leave together: rank 0 may still be printing
            dist.destroy_process_group()
        return

    warm(trainer, cfg, args.warmup)
    # Lazy R1 runs on every 16th iteration.  The K timed iterations start where they contain ceil(K / 16) of them (K = 20:
    # two, i.e. 1/10 instead of 1/16), so the timed region never under-counts the regulariser whatever --steps is;
    # `extra` reports the separately timed plain / R1 iterations and the exactly 1/16-weighted step next to it.
    lazy = int(cfg.training.lazy.gp)
    n_r1 = -(-args.steps // lazy)
    first_it = lazy * n_r1 - args.steps + 1 if args.steps % lazy else 1
    first_it = max(first_it, 1)
    dt, every = max_over_ranks(timed_steps(trainer, first_it, args.steps, barrier))
    r1_in_region = sum(1 for k in range(args.steps) if (first_it + k) % lazy == 0)
    imgs = args.steps * args.batch_per_gpu * world
    value = imgs / dt

    extra = {"r1_iterations_in_timed_region": r1_in_region, "first_timed_iteration": first_it,
             "per_rank_images_per_s": [args.steps * args.batch_per_gpu / t for t in every],
             "world_size_seen": dist.get_world_size() if dist_on else 1,
             "backend": dist.get_backend() if dist_on else None}
    if not args.no_extra:
        # plain and R1 iterations timed on their own (every rank takes part: the steps contain collectives)
        t_plain = max_over_ranks(timed_steps(trainer, 1, 6, barrier))[0] / 6
        t_r1 = sum(max_over_ranks(timed_steps(trainer, lazy, 1, barrier))[0] for _ in range(3)) / 3
        w16 = ((lazy - 1) * t_plain + t_r1) / lazy
        extra.update(ms_plain_iteration=1e3 * t_plain, ms_r1_iteration=1e3 * t_r1, ms_per_step_r1_every_16th=1e3 * w16,
                     value_r1_every_16th=args.batch_per_gpu * world / w16)
        # ADA at p = 0 (start of training; the headline point is the controller's target p = 0.6)
        p_keep = trainer.A.p.clone()
        trainer.A.p.zero_()
        extra["value_ada_p0"] = 8 * args.batch_per_gpu * world / max_over_ranks(timed_steps(trainer, 1, 8, barrier))[0]
        trainer.A.p.copy_(p_keep)

    # no step body may have fallen back to eager (Trainer._run warns and continues): a bench line must say so
    graphs_live = trainer.graphs_live()
    extra["graphs_live"] = graphs_live
    extra["overlap_d_reduce"] = bool(getattr(trainer, "split_d", False))   # D's backward in two graphs, head reduced under the tail
    if dist_on:   # the gradient reductions replayed as hipGraphs on a side stream (parallel.FlatGradSync.all_reduce_captured)
        extra["captured_collectives"] = {"G": {str(k): v for k, v in trainer.g_sync.captured().items()},
                                         "D": {str(k): v for k, v in trainer.d_sync.captured().items()}}
    if cfg.training.hip_graph and not (graphs_live and all(graphs_live.values())):
        msg = f"a step body is not replaying as a hipGraph: {graphs_live}"
        if world == 1:
            raise AssertionError(msg)
        print("bench.py: WARNING: " + msg, file=sys.stderr)   # N > 1: report it in extra.graphs_live, keep the line
    import dgv2_native
    extra["status_word"] = dgv2_native.status_read()   # 0: no kernel met a broken promise (x_exact) during the run
    if extra["status_word"]:
        raise AssertionError(f"a kernel reported a broken promise: status word {extra['status_word']} (include/dgv2.h)")
    roof = x3_probe(args) if rank == 0 else None
    roof_s2 = s2dgrad_probe(args) if rank == 0 else None
    roof_strip = roofline_probe(args) if rank == 0 else None
    roof_mod = modconv_levels(args) if rank == 0 else None
    if not args.no_extra and world == 1 and args.dtype == "bf16" and args.res == "64x512":
        # the same step with the discriminator epilogue in bf16 (opt-in; NOT what `value` is quoted on)
        other = "bf16" if args.d_epilogue == "fp32" else "fp32"
        del trainer
        torch.cuda.empty_cache()
        cfg2, tr2 = build_trainer(args, rank, world, d_epilogue=other)
        warm(tr2, cfg2, 1)
        extra[f"value_d_epilogue_{other}"] = args.steps * args.batch_per_gpu / timed_steps(tr2, first_it, args.steps, barrier)
        del tr2

    if rank == 0:
        out = {
            "metric": f"range-images/sec (G+D step) on dusty_v2 {args.res}",
            "value": value, "unit": "range-images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if args.dtype != "fp8" else "fp8 (e4m3 branch operands of D's ResidualBlocks) + bf16",
            "data": "synthetic",
            "config": {"workload": ("configs[2]" if args.res == "64x512" else
                                    ("configs[4] (fp8: e4m3 operands of the discriminator's decimating branch convs)"
                                     if args.dtype == "fp8" else "configs[4] shape in bf16"))
                                   + ": configs/gans/dusty_v2.yaml full G+D train step "
                                   f"(G step + D step + lazy R1 + ADA + EMA + Adam), {args.res} synthetic",
                       "global_batch": args.batch_per_gpu * world, "per_gpu_batch": args.batch_per_gpu,
                       "parallelism": f"dp{world}", "ada_p": args.ada_p, "hip_graph": not args.no_graph,
                       # conv trunks of G and D: bf16 storage, fp32 accumulation.  Heads / skip sums / ADA fp32.
                       # Discriminator epilogue (mbstd, 3x3 513->512 conv, Linear 65536->512 -> 1): see d_epilogue_dtype
                       "d_epilogue_dtype": args.d_epilogue,
                       # the encoding of the constant sensor grid is a table computed once;
                       # DGV2_NO_CONST_CACHE=1 recomputes it every forward (-1 %)
                       "pe_table_precomputed": os.environ.get("DGV2_NO_CONST_CACHE") is None},
            "model_tflops_per_gpu": value / world * GFLOP_PER_IMG_ITER / 1e3 if args.res == "64x512" else None,
            "extra": extra,
            "roofline_conv_x3": roof,
            "roofline_conv_s2dgrad": roof_s2,
            "roofline_strip": roof_strip,
            "roofline_modconv": roof_mod,
        }
        # `roofline` = the probe of the instance the committed statistics rank highest
        key, name, pct = dominant_instance()
        chosen = {"conv_x3": roof, "s2dgrad": roof_s2, "strip": roof_strip}.get(key) or roof
        out["roofline"] = None if chosen is None else dict(
            chosen, selected_by=f"largest share of the training iteration among the probed (kernel, grid) instances in {STATS_FILE}: "
                                 f"{pct} % ({key})",
            stats_measured_on_this_tree=stats_current())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        _emit(json.dumps(out), out_fd)
    if dist_on:
        dist.barrier()   # the other ranks wait for rank 0's roofline probes, then everybody tears down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
