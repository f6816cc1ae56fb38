"""MI355X-native GGUF weight dequantization behind the ComfyUI-GGUF ``dequant`` interface.

Scope: the one hot path of city96/ComfyUI-GGUF -- ``dequant.py`` (packed GGUF blocks -> dense
fp16) -- as hand-written HIP kernels for gfx950 behind a C-ABI shared library
(``include/ggq.h``), with a Python host side that mirrors the reference's
``dequantize_tensor`` / ``dequantize`` / ``dequantize_functions`` surface.

    dequant     the mirrored reference interface (HIP-backed; raises GGQUnsupported otherwise)
    install     patch an unmodified ComfyUI-GGUF checkout so its nodes run on this path
    autoinstall the literal drop-in: this directory inside ComfyUI/custom_nodes/ finds ComfyUI-GGUF and calls install() on it
    grouped     DequantPlan: a whole weight set in one launch per quant type
    gguf_file   GGUF container reader (native parser) + file -> HBM streaming upload
    gguf_adapter  gguf.GGUFReader as the reference's loader uses it, over gguf_file (install(native_reader=True))
    loader      gguf_sd_loader & co. (reference loader.py:16-141) without the `gguf` package
    resident    opt-in cache that keeps dequantized weights resident in HBM (288 GB make it possible)
    fused       fused dequantize + linear: 1-4 rows (modulation layers) and, on the matrix cores, up to 256 rows; install()'s default
    overlap     opt-in side-stream prefetch: layer i+1's (host->device copy and) unpack under layer i's GEMM
    sharding    tensor-list partitioning for one-process-per-GPU runs (no collectives)
    ops         GGMLTensor / GGMLLinear stand-ins for driving the path without ComfyUI
    manifests   synthetic weight manifests of the BASELINE.json configurations
    synth       seeded synthetic packed blocks
    qtypes      ggml type ids and block geometry
"""
from . import qtypes, synth  # noqa: F401  (torch-free)
from .qtypes import GGMLQuantizationType, GGML_QUANT_SIZES  # noqa: F401

__version__ = "0.1.0"
# A ComfyUI custom-node directory (reference __init__.py:1-9 is activated the same way): ComfyUI requires the mapping to exist; this
# package adds NO nodes -- it accelerates the ones ComfyUI-GGUF registers ("Unet Loader (GGUF)" & co. keep working unchanged).
NODE_CLASS_MAPPINGS = {}
NODE_DISPLAY_NAME_MAPPINGS = {}
_LAZY = ("_native", "autoinstall", "dequant", "install", "grouped", "sharding", "ops", "manifests", "gguf_file", "gguf_adapter", "loader", "resident", "fused", "overlap")


def _running_under_comfyui():
    import os
    import sys
    flag = os.environ.get("GGQ_AUTO_INSTALL", "")
    return flag not in ("0",) and (flag == "1" or "folder_paths" in sys.modules)      # ComfyUI imports folder_paths before any custom node


if _running_under_comfyui():
    from . import autoinstall as _auto
    _auto.arm()


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
