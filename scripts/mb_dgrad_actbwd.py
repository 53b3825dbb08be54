"""conv2's data gradient of a generator level + conv1's activation backward: the two launches (dgv2_modconv_pe_fwd with
Ks = 0, dgv2_bias_act_bwd_rs) against the fused dgv2_modconv_pe_dgrad_actbwd; us per launch at B = 64."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dusty-gan-v2_amd")]
import torch
import bench
import dgv2_native as N
from gans.models.ops import native as nat
B = 64
bf = torch.bfloat16
for P, K in ((32768, 32), (8192, 64)):
    gy = torch.randn(B, P, K, device="cuda").to(bf)
    wt = (torch.randn(B, K, K, device="cuda") / K ** 0.5).to(bf)
    yref = torch.randn(B, P, K, device="cuda").to(bf)
    cvec = torch.rand(K, device="cuda") + 0.5
    out = torch.empty_like(gy); gb = torch.empty(K, device="cuda")
    rows = B * P
    scratch = torch.empty(2048 * K, device="cuda") if rows >= 65536 else None
    gd = nat._bmm_nn_raw(gy, wt, bf)
    t1 = bench._time_launches(lambda: nat._bmm_nn_raw(gy, wt, bf), 20)
    t2 = bench._time_launches(lambda: N.call("dgv2_bias_act_bwd_rs", N.ptr(out), N.ptr(gb), N.ptr(gd), N.ptr(yref), rows, K, 0.2,
                                             math.sqrt(2.0), N.ptr(cvec), N.ptr(scratch), 0 if scratch is None else scratch.numel(),
                                             N.BF16, N.stream()), 20)
    t3 = bench._time_launches(lambda: nat._dgrad_actbwd(gy, wt, yref, dict(link={}, alpha=0.2, scale=math.sqrt(2.0), cvec=cvec)), 20)
    mb = B * P * K * 2 / 1e6
    print(f"B={B} P={P} K={K} ({mb:.0f} MB per tensor): data gradient {t1*1e6:6.1f} us + activation backward {t2*1e6:6.1f} us = "
          f"{(t1+t2)*1e6:6.1f} us;  fused {t3*1e6:6.1f} us")
