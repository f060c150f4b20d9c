"""Op layer: ``torch.autograd.Function`` wrappers over the hand-written gfx950 kernels.

Every op has exactly one GPU implementation (the HIP extension ``_C``; a missing
build raises) and a plain-PyTorch CPU implementation that serves as the numerics
oracle in tests and lets the distributed plumbing run under gloo without a GPU.
"""
from . import sink  # noqa: F401
from ._ext import available as extension_available, load as load_extension  # noqa: F401
from .conv import conv2d_nhwc, linear  # noqa: F401
from .norm import batch_norm_nhwc, layer_norm, layer_norm_fork  # noqa: F401
from .misc import (max_pool_3x3_s2_nhwc, global_avg_pool_nhwc, to_nhwc, add_relu, add, gelu)  # noqa: F401
from .loss import softmax_cross_entropy  # noqa: F401
from .small import (conv5_pool_relu, linear_f32, linear_log_softmax, linear_log_softmax_nll, dropout, channel_dropout_mask, log_softmax, nll_loss)  # noqa: F401
from .tfm import embedding, causal_attention, dropout_residual  # noqa: F401
