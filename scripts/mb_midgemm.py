"""The PE-free modulated 1x1 convs of the middle generator levels (per-sample weights, B = 64): the generic NN engine
(dgv2_bmm_nn_sq) against the sample-walking kernel (dgv2_modconv_pe_fwd_sq) where the latter has the shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")):
    sys.path.insert(0, p)
import torch
import bench
import dgv2_native as N
B = 64
for (I, O, P) in ((128, 128, 2048), (64, 64, 8192), (32, 32, 32768), (256, 256, 512), (128, 64, 8192), (64, 128, 2048),
                  (256, 128, 2048), (128, 256, 512), (128, 64, 2048), (64, 128, 8192), (256, 256, 2048)):
    x = torch.randn(B, P, I, device="cuda").bfloat16()
    w = (torch.randn(B, O, I, device="cuda") / I ** 0.5).bfloat16()
    bias = torch.randn(O, device="cuda"); cvec = torch.ones(O, device="cuda")
    y = torch.empty(B, P, O, device="cuda", dtype=torch.bfloat16)
    y2 = torch.empty_like(y)
    def gen():
        N.call("dgv2_bmm_nn_sq", N.ptr(y), N.ptr(x), N.ptr(w), B, P, I, O, I, O, O * I, N.ptr(cvec), N.ptr(bias), 3, 0.2, 2 ** 0.5,
               None, N.BF16, N.BF16, None, 0, None, N.stream())
    t_gen = bench._time_launches(gen, 20) * 1e6
    t_pe = None
    if N.lib.dgv2_modconv_pe_fwd_sq(N.ptr(y2), N.ptr(x), None, N.ptr(w), B, P, I, 0, O, N.ptr(cvec), N.ptr(bias), 3, 0.2, 2 ** 0.5,
                                     N.BF16, None, 0, None, N.stream()) == 0:
        def pe():
            N.call("dgv2_modconv_pe_fwd_sq", N.ptr(y2), N.ptr(x), None, N.ptr(w), B, P, I, 0, O, N.ptr(cvec), N.ptr(bias), 3, 0.2,
                   2 ** 0.5, N.BF16, None, 0, None, N.stream())
        t_pe = bench._time_launches(pe, 20) * 1e6
        torch.cuda.synchronize()
        assert torch.equal(y, y2) or float((y.float() - y2.float()).abs().max()) <= 2 ** -6 * float(y.float().abs().max())
    gf = 2.0 * B * P * I * O / 1e9
    mb = B * P * (I + O) * 2 / 1e6
    print(f"I={I:4d} O={O:4d} P={P:6d}: {gf:6.2f} GFLOP {mb:7.1f} MB  generic {t_gen:7.1f} us ({gf / t_gen * 1e3:6.1f} TF/s, {mb / t_gen:5.2f} TB/s)"
          + (f"   sample-walking {t_pe:7.1f} us ({gf / t_pe * 1e3:6.1f} TF/s, {mb / t_pe:5.2f} TB/s)" if t_pe else ""))
