"""MI355X-native mirror of the reference's `gans` package (hot path only).

Same import paths, constructor kwargs, forward signatures and state-dict layout as
kazuto1011/dusty-gan-v2 for the dusty_v2 G+D training path; the arithmetic runs in the
hand-written HIP kernels of libdgv2.so (see include/dgv2.h).  There is no
CPU fallback: importing the operator layer without the built library raises ImportError.
"""
