import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dusty-gan-v2_amd"))
from gans.models.ops import native as nat
torch.manual_seed(0)
B, H, W, C, Cp, O = 4, 4, 32, 513, 528, 512
g = nat.ConvGeom(3, 3, 1, 1, True)
w = torch.randint(-3, 4, (O, C, 3, 3), device="cuda").float()
(wf, wt, w3, w3t), = nat.conv_weight_bank([(w, 1.0, Cp)], torch.float32, image8=[True])
w3t._dgv2_clive = C
gy = torch.randint(-3, 4, (B, H, W, O), device="cuda").float()
ref = nat._conv_dgrad_raw(gy, None, g, (B, H, W, Cp), wt=wt)
outs = [nat._conv_dgrad_raw(gy, None, g, (B, H, W, Cp), wt=wt, w8t=w3t).clone() for _ in range(4)]
for i, o in enumerate(outs):
    d = (o - ref).abs()
    bad = d > 0
    print(f"launch {i}: mismatches {int(bad.sum())} of {bad.numel()}; by channel block of 64: {[int(bad[..., k*64:(k+1)*64].sum()) for k in range(9)]}; by row: {[int(bad[:, r].sum()) for r in range(H)]}; by image: {[int(bad[b].sum()) for b in range(B)]}; vs launch0 equal: {torch.equal(o, outs[0])}")
    if bad.any():
        idx = bad.nonzero()[:5]
        print("   first:", idx.tolist(), [float(o[tuple(i_)]) for i_ in idx], [float(ref[tuple(i_)]) for i_ in idx])
