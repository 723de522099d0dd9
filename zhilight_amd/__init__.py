"""zhilight_amd -- MI355X (gfx950) native implementation of ZhiLight's quantized-GEMM + fused-attention
decode hot path.  `ops` mirrors the reference's nn::/gptq::/int8_op:: operator functions on top of
the C ABI in include/zhilight_amd.h; there is no CPU or PyTorch fallback."""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
