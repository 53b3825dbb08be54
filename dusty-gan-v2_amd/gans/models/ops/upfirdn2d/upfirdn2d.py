"""upfirdn2d with the reference's Python interface (gans/models/ops/upfirdn2d/upfirdn2d.py:148-164);
the kernel is dgv2_upfirdn2d.  Backward and double backward are the same op with flipped kernel and
swapped up/down, exactly as in the reference (upfirdn2d.py:20-145)."""
from collections import abc

import torch
from torch.autograd import Function

from .. import native

__all__ = ["upfirdn2d"]


class _UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        g = grad_output.reshape(-1, out_size[0], out_size[1], 1).contiguous()
        gi = native.upfirdn2d_raw(g, grad_kernel, down, up, g_pad)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return gi.view(in_size)

    @staticmethod
    def backward(ctx, gg):
        (kernel,) = ctx.saved_tensors
        up, down, pad, in_size, out_size = ctx.cfg
        gg = gg.reshape(-1, in_size[2], in_size[3], 1).contiguous()
        out = native.upfirdn2d_raw(gg, kernel, up, down, pad)
        return (out.view(in_size[0], in_size[1], out_size[0], out_size[1]),) + (None,) * 8


class _UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, ch, in_h, in_w = input.shape
        ctx.in_size = tuple(input.shape)
        x = input.reshape(-1, in_h, in_w, 1).contiguous()
        kernel = kernel.float().contiguous()
        ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]).contiguous())
        out_h = (in_h * up_y + py0 + py1 - kh + down_y) // down_y
        out_w = (in_w * up_x + px0 + px1 - kw + down_x) // down_x
        ctx.out_size = (out_h, out_w)
        ctx.cfg = (up, down, pad)
        ctx.g_pad = (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                     kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1)
        out = native.upfirdn2d_raw(x, kernel, up, down, pad)
        return out.view(-1, ch, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        up, down, pad = ctx.cfg
        gi = None
        if ctx.needs_input_grad[0]:
            gi = _UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, up, down, pad, ctx.g_pad, ctx.in_size,
                                          ctx.out_size)
        return gi, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if not isinstance(up, abc.Iterable):
        up = (up, up)
    if not isinstance(down, abc.Iterable):
        down = (down, down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    if input.device.type == "cpu":
        raise RuntimeError("upfirdn2d: the MI355X build has no CPU path (use oracle/ for CPU checks)")
    return _UpFirDn2d.apply(input, kernel, tuple(up), tuple(down), tuple(pad))
