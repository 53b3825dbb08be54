"""Numerical study (CPU, float64 reference): what a Winograd F(2x2, 3x3) form of D's stride-1 3x3 convs would cost in accuracy
when its transformed operands are rounded to bf16 for the matrix cores (fp32 accumulation), against the direct form with
bf16 operands that the step runs today (conv8.hip / conv_pipe_kernel).  Review item 9 of round 4: "prototype on one layer,
report, do not ship on hope".  Layer: ResidualBlock.conv1 at 16 x 128, 128 -> 128 channels (gans/models/dusty_v2.py:329-333),
ring padding along W, replicate along H; operands drawn as the step sees them (unit-variance activations after the leaky ReLU,
EqualLR weights N(0, 1) * 1 / sqrt(9 C)).
  direct   : y = sum_{c, ky, kx} bf16(x) * bf16(w)                      (fp32 accumulate)
  winograd : U = bf16(G w G^T) [4 x 4 per (o, c)], V = bf16(B^T d B) [4 x 4 per tile, c], M = sum_c U * V (fp32), y = A^T M A
  winograd, V from fp32 x: the input transform applied to the fp32 activation (not available in the step: x is stored as bf16)
Also: the exact-integer test (tests/test_gpu_ops.py: test_conv_bf16_exact_on_integers draws integers in [-4, 4]): is it
still exact?  usage: python scripts/exp/winograd_bf16_error.py"""
import torch

torch.manual_seed(0)
B, C, O, H, W = 4, 128, 128, 16, 128


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def pad(x):   # ring along W, replicate along H (gans/models/ops/common.py:10-24)
    x = torch.cat([x[..., -1:], x, x[..., :1]], dim=-1)
    return torch.cat([x[..., :1, :], x, x[..., -1:, :]], dim=-2)


def direct(x, w):
    return torch.nn.functional.conv2d(pad(x), w)


G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(x, w, round_u=bf, round_v=bf):
    xp = pad(x)                                                   # [B, C, H + 2, W + 2]
    U = round_u(torch.einsum("ij,ocjk,lk->ocil", G, w, G))        # [O, C, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # [B, C, H/2, W/2, 4, 4]
    V = round_v(torch.einsum("ij,bchwjk,lk->bchwil", Bt, d, Bt))
    M = torch.einsum("ocil,bchwil->bohwil", U, V)                 # exact sum (float64) of the rounded products
    Y = torch.einsum("ij,bohwjk,lk->bohwil", At, M, At)           # [B, O, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(x.shape[0], w.shape[0], H, W)


def report(name, y, ref):
    e = (y - ref)
    print(f"  {name:44s} rel L2 {float(e.norm() / ref.norm()):9.2e}   max |err| / rms(ref) {float(e.abs().max() / ref.pow(2).mean().sqrt()):9.2e}")


print(f"layer {H}x{W}, {C} -> {O}, B = {B} (float64 reference)")
x32 = torch.nn.functional.leaky_relu(torch.randn(B, C, H, W, dtype=torch.float64), 0.2) * 2 ** 0.5
w32 = torch.randn(O, C, 3, 3, dtype=torch.float64) / (9 * C) ** 0.5
ref = direct(x32, w32)
xb = bf(x32)                                                       # the activation as the step stores it
ref_b = direct(xb, bf(w32))                                        # what the direct bf16 kernel computes, exactly accumulated
report("direct, bf16 operands (today)", ref_b, ref)
report("winograd, U and V rounded to bf16", winograd(xb, w32), ref)
report("winograd, V from the fp32 activation", winograd(x32, w32), ref)
report("winograd, U bf16, V exact (fp32 planes)", winograd(xb, w32, round_v=lambda t: t), ref)
report("winograd, U exact, V bf16", winograd(xb, w32, round_u=lambda t: t), ref)
y_w = winograd(xb, w32)
print(f"  winograd vs direct-bf16 (the kernel-to-kernel distance)   rel L2 {float((y_w - ref_b).norm() / ref_b.norm()):9.2e}")

# the integer test of the conv engines: integers in [-4, 4] for both operands
xi = torch.randint(-4, 5, (B, C, H, W)).double()
wi = torch.randint(-4, 5, (O, C, 3, 3)).double()
yi = direct(xi, wi)
yw = winograd(xi, wi)
Ui = torch.einsum("ij,ocjk,lk->ocil", G, wi, G)
Vi = torch.einsum("ij,bchwjk,lk->bchwil", Bt, pad(xi).unfold(2, 4, 2).unfold(3, 4, 2), Bt)
print(f"integer operands in [-4, 4]: winograd == direct bit for bit: {bool((yi == yw).all())}; "
      f"U exact in bf16: {bool((bf(Ui) == Ui).all())} (max |U| {float(Ui.abs().max()):.2f}, quarter steps), "
      f"V exact in bf16: {bool((bf(Vi) == Vi).all())} (max |V| {float(Vi.abs().max()):.0f})")
xi = torch.randint(-64, 65, (B, C, H, W)).double()
Vi = torch.einsum("ij,bchwjk,lk->bchwil", Bt, pad(xi).unfold(2, 4, 2).unfold(3, 4, 2), Bt)
print(f"integer activations in [-64, 64] (exact in bf16 as they stand): V exact in bf16: {bool((bf(Vi) == Vi).all())} (max |V| {float(Vi.abs().max()):.0f})")
