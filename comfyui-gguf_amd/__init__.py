"""MI355X-native GGUF weight dequantization behind the ComfyUI-GGUF ``dequant`` interface.

Scope: the one hot path of city96/ComfyUI-GGUF -- ``dequant.py`` (packed GGUF blocks -> dense
fp16) -- as hand-written HIP kernels for gfx950 behind a C-ABI shared library
(``include/ggq.h``), with a Python host side that mirrors the reference's
``dequantize_tensor`` / ``dequantize`` / ``dequantize_functions`` surface.
"""
from . import qtypes, synth  # noqa: F401
from .qtypes import GGMLQuantizationType, GGML_QUANT_SIZES  # noqa: F401

__version__ = "0.1.0"
