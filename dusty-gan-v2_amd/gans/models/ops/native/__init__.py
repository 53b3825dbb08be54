"""Autograd-aware wrappers around the libdgv2 C ABI (include/dgv2.h).

Internal layout is channels-last: activations are contiguous [B, H, W, C] tensors in
float32 (parity mode) or bfloat16 (throughput mode, fp32 accumulate).  Parameters
stay float32 masters; weight gradients are produced in float32.

Every op that sits on the discriminator side of the R1 penalty (gans/trainer.py:419-451
of the reference) is closed under differentiation: linear ops pair a forward Function
with its transpose, convolutions form the {fwd, dgrad, wgrad} triple, bias+lrelu reuses
its masked form -- so double backward never leaves the HIP kernels.
"""
from .act_resample import *  # noqa: F401,F403
from .modgemm import *  # noqa: F401,F403
from .conv import *  # noqa: F401,F403
from .fp8 import *  # noqa: F401,F403
from .stem_tail_ada import *  # noqa: F401,F403
from .modlayer import *  # noqa: F401,F403
from .misc import *  # noqa: F401,F403
from .second_order import *  # noqa: F401,F403
