"""ggml quantization type ids and block geometry.

The reference takes both from the third-party ``gguf`` package (``gguf.GGMLQuantizationType``,
``gguf.GGML_QUANT_SIZES``; call sites reference dequant.py:7,34,288-300).  This path must not
depend on ``gguf`` being importable, so the constants it needs are restated here.  The enum is an
``IntEnum`` with ggml's public values, hence a real ``gguf.GGMLQuantizationType`` member (also an
``IntEnum``) hashes and compares equal to the matching member here and can be used as a key into
any table below.

The ``(block_size, type_size)`` pairs are independently forced by the reference's own field
splits (e.g. Q6_K = 128 + 64 + 16 + 2 bytes, dequant.py:144).
"""
from enum import IntEnum


class GGMLQuantizationType(IntEnum):
    F32 = 0
    F16 = 1
    Q4_0 = 2
    Q4_1 = 3
    Q5_0 = 6
    Q5_1 = 7
    Q8_0 = 8
    Q8_1 = 9
    Q2_K = 10
    Q3_K = 11
    Q4_K = 12
    Q5_K = 13
    Q6_K = 14
    Q8_K = 15
    IQ2_XXS = 16
    IQ2_XS = 17
    IQ3_XXS = 18
    IQ1_S = 19
    IQ4_NL = 20
    IQ3_S = 21
    IQ2_S = 22
    IQ4_XS = 23
    I8 = 24
    I16 = 25
    I32 = 26
    I64 = 27
    F64 = 28
    IQ1_M = 29
    BF16 = 30


Q = GGMLQuantizationType

# qtype -> (elements per block, bytes per block)
GGML_QUANT_SIZES = {
    Q.F32: (1, 4),
    Q.F16: (1, 2),
    Q.BF16: (1, 2),
    Q.Q4_0: (32, 18),
    Q.Q4_1: (32, 20),
    Q.Q5_0: (32, 22),
    Q.Q5_1: (32, 24),
    Q.Q8_0: (32, 34),
    Q.Q2_K: (256, 84),
    Q.Q3_K: (256, 110),
    Q.Q4_K: (256, 144),
    Q.Q5_K: (256, 176),
    Q.Q6_K: (256, 210),
    Q.IQ4_NL: (32, 18),
    Q.IQ4_XS: (256, 136),
}

# byte offsets of the fp16 scale fields inside one block (layout table)
SCALE_FIELDS = {
    Q.Q8_0: (0,),
    Q.Q4_0: (0,),
    Q.Q4_1: (0, 2),
    Q.Q5_0: (0,),
    Q.Q5_1: (0, 2),
    Q.Q2_K: (80, 82),
    Q.Q3_K: (108,),
    Q.Q4_K: (0, 2),
    Q.Q5_K: (0, 2),
    Q.Q6_K: (208,),
    Q.IQ4_NL: (0,),
    Q.IQ4_XS: (0,),
}

LEGACY_QTYPES = (Q.Q4_0, Q.Q4_1, Q.Q5_0, Q.Q5_1, Q.Q8_0)
K_QTYPES = (Q.Q2_K, Q.Q3_K, Q.Q4_K, Q.Q5_K, Q.Q6_K)
IQ_QTYPES = (Q.IQ4_NL, Q.IQ4_XS)

# the formats with a hand-written HIP unpacker in csrc/ggq_device.hpp
HIP_QTYPES = LEGACY_QTYPES + K_QTYPES + IQ_QTYPES


def block_geometry(qtype):
    """(block_size, type_size) for a quantized qtype; KeyError if unknown."""
    return GGML_QUANT_SIZES[GGMLQuantizationType(int(qtype))]


def algorithmic_bytes(qtype, n_elements, out_itemsize=2):
    """Bytes one dequant must move: packed read + dense write."""
    bs, ts = block_geometry(qtype)
    return (n_elements // bs) * ts + n_elements * out_itemsize
