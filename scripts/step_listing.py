"""Per-body launch listing of the training iteration from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`
of bench.py or scripts/prof_plain.py): which kernels one G body / D body / R1 body launches, in order, with grid and the
duration averaged over the replays of that body.  The bodies are cut at the fused Adam launches (the G optimizer's is
the short one, D's the long one; a D body that directly follows a D body is the lazy-R1 pass) and only the bodies with
the modal launch count of their kind are averaged (the graph replays; eager warm-up bodies differ).
usage: python scripts/step_listing.py <kernel_trace.csv> [--full] [--kernel REGEX] [--json OUT --stamp STAMP]
  default: per-body totals, bucket table per plain iteration (G + D body) and launch counts
  --full:  every launch of the three bodies (index, avg us, grid, workgroup, name)
  --kernel REGEX: per-launch durations (every replay) of the launches whose name matches, grouped by grid
  --json OUT: the (kernel, grid) instances of a TRAINING iteration (G body + D body + R1 body / 16) with their time, share
              and launch count, largest first (bench.py ranks `roofline` by it).  The identity of the measured kernels
              is COPIED from STAMP, the file the profiled `bench.py --stamp STAMP` process wrote next to its trace; this
              script never hashes the tree it happens to run in (a listing cut later, on a changed tree, would otherwise
              certify kernels it was not measured on), and refuses --json without a stamp"""
import collections
import csv
import re
import statistics
import sys

BUCKETS = collections.OrderedDict([
    ("conv_x3 (fp32 epilogue conv)", r"conv_x3_|x3_dgrad_tail|conv_wgrad_x3|x3_image"), ("conv_pipe", r"conv_pipe_kernel"),
    ("conv8", r"conv8_kernel|conv8_s2d_kernel"), ("conv_strip", r"conv3x3_strip"), ("conv_deep", r"conv_deep|skip_fused"),
    ("conv_direct_fallback", r"conv_direct_kernel"),
    ("conv_wgrad_stream", r"wgrad_stream|wgrad_reduce"), ("conv_wgrad_direct", r"wgrad_direct"),
    ("gemm_tn", r"gemm_tn_kernel"), ("gemm_nn(conv)", r"gemm_nn_kernel.*Im2col"), ("gemm_nn", r"gemm_nn_kernel"),
    ("fir_mfma", r"fir_same_mfma"), ("resample", r"resample"), ("upfirdn/ada", r"ada_build_kernel|upfirdn|ada_"), ("mod_prep", r"mod_prep"),
    ("bias_act", r"bias_act|bias_grad"), ("sumsq", r"sum_squares"),
    ("tail/fourier/coords", r"gen_tail|fourier|coords|downsample_angle|fetch_reals|synth"),
    ("modconv_pe", r"modconv_pe"), ("modconv_up", r"modconv_up|up2_lag"), ("pe_wgrad", r"pe_wgrad"), ("gemm_x3", r"gemm_x3"),
    ("stem", r"stem_"), ("adam/lerp", r"adam_|lerp_list"), ("ema/pack/bank", r"ema_scalar|pack2d|weight_bank"),
    ("glin/bmm/colsum", r"glin_|bmm_|colsum|transpose_list"), ("mbstd/loss", r"mbstd|loss|nsgan"),
    ("zero", r"dgv2_zero"), ("rng", r"rng_fill|philox"), ("head_bwd", r"head_bwd_kernel"), ("mod_stats", r"mod_stats"),
    ("scale_cast/shift", r"scale_cast|ring_shift"), ("blas", r"Cijk|rocblas|hipblas"),
    ("rccl", r"nccl|rccl|oneRankReduce"), ("memcpy", r"copyBuffer|fillBuffer"), ("aten", r"at::native|at_cuda"),
    ("reducers (own)", r"_reduce_kernel"),
    ("other", r".*")])


def bucket(name):
    for k, pat in BUCKETS.items():
        if re.search(pat, name):
            return k


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Kind"] != "KERNEL_DISPATCH":
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])),
                     (int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"]))))
    rows.sort()
    return rows


def bodies(rows):
    """[(kind, [launch...])] with kind in g / d / r1; the Adam launch closes its body."""
    adam = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[2]]
    if not adam:
        raise SystemExit("no adam_step_kernel launches in the trace")
    # one optimizer step = one or two adjacent Adam launches (G's: the bulk of the tensors + the few large ones)
    groups = []
    for i in adam:
        if groups and groups[-1][-1] == i - 1:
            groups[-1].append(i)
        else:
            groups.append([i])
    # G's parameters are 17.5 MB, D's 154 MB: the two populations of step durations are an order of magnitude apart
    gd = [sum(rows[i][1] - rows[i][0] for i in g) for g in groups]
    cut = (min(gd) * max(gd)) ** 0.5
    out, prev, last_kind = [], -1, None
    for g, dur in zip(groups, gd):
        is_d = dur > cut
        kind = "g" if not is_d else ("r1" if last_kind in ("d", "r1") else "d")
        out.append((kind, rows[prev + 1:g[-1] + 1]))
        prev, last_kind = g[-1], kind
    return out


def main():
    path = sys.argv[1]
    full = "--full" in sys.argv
    pat = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
    rows = load(path)
    if "--stamp" in sys.argv:
        import json
        st = json.load(open(sys.argv[sys.argv.index("--stamp") + 1]))
        print(f"# trace {path.split('/')[-1]} of `bench.py {' '.join(st.get('argv') or [])}` (pid {st.get('pid')}); kernels: "
              f"src_sha16 {st.get('src_sha16')}, {st.get('lib')} sha16 {st.get('lib_sha16')}")
    per = collections.defaultdict(list)
    for kind, b in bodies(rows):
        per[kind].append(b)
    summary = {}
    for kind in ("g", "d", "r1"):
        if not per[kind]:
            continue
        # the most frequent kernel-name sequence of this kind is the replayed graph (eager warm-up bodies, bodies that
        # carry an ADA update or a probe differ)
        seqs = collections.Counter(tuple(l[2] for l in b) for b in per[kind])
        ref = seqs.most_common(1)[0][0]
        bs = [b for b in per[kind] if tuple(l[2] for l in b) == ref]
        n = len(ref)
        avg = [statistics.mean(b[j][1] - b[j][0] for b in bs) / 1e3 for j in range(n)]
        span = statistics.mean(b[-1][1] - b[0][0] for b in bs) / 1e3
        summary[kind] = (bs, avg, span)
        print(f"# body {kind}: {n} launches, {sum(avg) / 1e3:.3f} ms of kernel time, {span / 1e3:.3f} ms first start -> last end, "
              f"averaged over {len(bs)} replays (of {len(per[kind])} bodies in the trace)")
    if "g" in summary and "d" in summary:
        agg = collections.defaultdict(lambda: [0.0, 0])
        for kind in ("g", "d"):
            bs, avg, _ = summary[kind]
            for l, a in zip(bs[0], avg):
                k = bucket(l[2])
                agg[k][0] += a
                agg[k][1] += 1
        tot = sum(v[0] for v in agg.values())
        nl = sum(v[1] for v in agg.values())
        print(f"# plain iteration (G body + D body): {nl} launches, {tot / 1e3:.3f} ms of kernel time")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            print(f"{k:30s} {v[0] / 1e3:8.3f} ms/iteration  {v[1]:5d} launches  {100 * v[0] / tot:5.1f} %")
        # the largest instances (name + grid) of the plain iteration
        inst = collections.defaultdict(lambda: [0.0, 0])
        for kind in ("g", "d"):
            bs, avg, _ = summary[kind]
            for l, a in zip(bs[0], avg):
                inst[(l[2][:100], l[3])][0] += a
                inst[(l[2][:100], l[3])][1] += 1
        print("# largest (kernel, grid) instances of the plain iteration")
        for (name, grid), v in sorted(inst.items(), key=lambda kv: -kv[1][0])[:25]:
            print(f"  {v[0]:8.1f} us x{v[1]:3d}  {100 * v[0] / tot:4.1f} %  grid {grid}  {name}")
    if full:
        for kind, (bs, avg, _) in summary.items():
            print(f"\n## body {kind}")
            for j, (l, a) in enumerate(zip(bs[0], avg)):
                print(f"{j:4d} {a:9.1f} us  grid {str(l[3]):22s} wg {l[4][0]:4d}  {bucket(l[2]):14s} {l[2][:120]}")
    if "--json" in sys.argv:
        import json
        import os
        if "--stamp" not in sys.argv:
            raise SystemExit("--json needs --stamp FILE (written by the profiled `bench.py --stamp FILE` process)")
        stamp = json.load(open(sys.argv[sys.argv.index("--stamp") + 1]))
        if not stamp.get("src_sha16"):
            raise SystemExit("the stamp file carries no src_sha16")
        # rocprofv3 names a trace <pid>_kernel_trace.csv: the stamp must come from the very process that was traced
        m = re.match(r"(\d+)_", os.path.basename(path))
        if m and stamp.get("pid") is not None and int(m.group(1)) != int(stamp["pid"]):
            raise SystemExit(f"trace {os.path.basename(path)} was not written by the stamped process (pid {stamp['pid']})")
        inst = collections.defaultdict(lambda: [0.0, 0.0])
        weight = {"g": 1.0, "d": 1.0, "r1": 1.0 / 16.0}
        for kind, (bs, avg, _) in summary.items():
            for l, a in zip(bs[0], avg):
                inst[(l[2], l[3])][0] += a * weight[kind]
                inst[(l[2], l[3])][1] += weight[kind]
        tot = sum(v[0] for v in inst.values())
        out = {"src_sha16": stamp["src_sha16"], "lib_sha16": stamp.get("lib_sha16"), "bench_argv": stamp.get("argv"),
               "source": os.path.basename(path),
               "us_per_training_iteration": tot, "launches_per_training_iteration": sum(v[1] for v in inst.values()),
               "bodies": {k: {"launches": len(v[1]), "kernel_us": sum(v[1]), "span_us": v[2], "replays": len(v[0])}
                          for k, v in summary.items()},
               "instances": [{"name": n, "grid": list(g), "us": v[0], "launches": v[1], "pct": 100 * v[0] / tot}
                             for (n, g), v in sorted(inst.items(), key=lambda kv: -kv[1][0])]}
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=0)
    if pat:
        print(f"\n## launches matching /{pat}/ in the averaged bodies (every replay)")
        groups = collections.defaultdict(list)
        for kind, (bs, _, _) in summary.items():
            for b in bs:
                for l in b:
                    if re.search(pat, l[2]):
                        groups[(kind, l[2][:90], l[3])].append((l[1] - l[0]) / 1e3)
        for (kind, name, grid), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            print(f"  body {kind:2s} grid {str(grid):22s} n {len(d):4d}  avg {statistics.mean(d):8.1f} us  min {min(d):8.1f}  "
                  f"max {max(d):8.1f}  {name}")


if __name__ == "__main__":
    main()
