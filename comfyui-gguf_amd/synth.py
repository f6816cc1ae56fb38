"""Seeded synthetic packed GGUF blocks (numpy, host side).

No real .gguf file and no quantizer exist in the build environment, so every test, the bench and
the golden-vector generator draw packed blocks from here: uniformly random bytes -- which
exercises every bit pattern of the integer fields, something real weights would not -- with the
fp16 scale fields overwritten:

  mode "nominal"      legacy formats d,m ~ U(1e-3, 2.1e-2); K/IQ formats d,dmin ~ U(1e-4, 2e-3)
  mode "signed"       as nominal with a random sign on every scale field
  mode "adversarial"  scale fields drawn from a pool of hard fp16 values: +-0, subnormals,
                      smallest/largest normals, +-65504, values whose products overflow or land on
                      rounding ties, +-inf, NaN, plus fully random bit patterns
  mode "raw"          nothing overwritten (scale fields are random bits too)
"""
import numpy as np

from .qtypes import GGML_QUANT_SIZES, SCALE_FIELDS, LEGACY_QTYPES, GGMLQuantizationType

_HARD_FP16 = np.array(
    [
        0x0000, 0x8000,              # +-0
        0x0001, 0x8001, 0x03FF, 0x83FF, 0x0200,   # subnormals
        0x0400, 0x8400,              # smallest normals
        0x7BFF, 0xFBFF,              # +-65504
        0x7C00, 0xFC00,              # +-inf
        0x7E00, 0xFE00, 0x7C01,      # NaNs
        0x3C00, 0xBC00,              # +-1
        0x3C01, 0x3BFF, 0x3555, 0x2E66, 0x1400, 0x1001, 0x0801, 0x4200, 0x5640, 0x6400, 0x6C00, 0x7000,
        0xB555, 0xAE66, 0x9400, 0x9001, 0xC200, 0xD640, 0xE400, 0xEC00,
    ],
    dtype=np.uint16,
)


def n_blocks_for(qtype, n_elements):
    bs, _ = GGML_QUANT_SIZES[GGMLQuantizationType(int(qtype))]
    if n_elements % bs:
        raise ValueError(f"{n_elements} elements is not a whole number of {bs}-element blocks")
    return n_elements // bs


def make_blocks(qtype, n_blocks, seed=0, mode="nominal"):
    """uint8 array of shape (n_blocks, type_size)."""
    qtype = GGMLQuantizationType(int(qtype))
    _, ts = GGML_QUANT_SIZES[qtype]
    rng = np.random.default_rng([int(seed), int(qtype)])
    blocks = rng.integers(0, 256, size=(n_blocks, ts), dtype=np.uint8)
    if mode == "raw" or n_blocks == 0:
        return blocks
    lo, hi = (1e-3, 2.1e-2) if qtype in LEGACY_QTYPES else (1e-4, 2e-3)
    for off in SCALE_FIELDS[qtype]:
        if mode in ("nominal", "signed"):
            vals = rng.uniform(lo, hi, size=n_blocks).astype(np.float16)
            if mode == "signed":
                vals = np.where(rng.integers(0, 2, size=n_blocks).astype(bool), -vals, vals).astype(np.float16)
            bits = vals.view(np.uint16)
        elif mode == "adversarial":
            pick = rng.integers(0, len(_HARD_FP16) + 8, size=n_blocks)
            rand_bits = rng.integers(0, 1 << 16, size=n_blocks, dtype=np.uint16)
            bits = np.where(pick < len(_HARD_FP16), _HARD_FP16[np.minimum(pick, len(_HARD_FP16) - 1)], rand_bits)
            bits = bits.astype(np.uint16)
        else:
            raise ValueError(f"unknown mode {mode!r}")
        blocks[:, off] = (bits & 0xFF).astype(np.uint8)
        blocks[:, off + 1] = (bits >> 8).astype(np.uint8)
    return blocks


def make_tensor_bytes(qtype, shape, seed=0, mode="nominal"):
    """Packed bytes (1-D uint8) for a logical tensor of ``shape``."""
    n = int(np.prod(shape))
    return make_blocks(qtype, n_blocks_for(qtype, n), seed=seed, mode=mode).reshape(-1)


def device_blocks(qtype, n_blocks, device, seed, mode="nominal"):
    """Packed bytes (1-D uint8 torch tensor) generated ON ``device`` -- a whole weight set is gigabytes, too slow to draw with
    numpy on the host and push over PCIe: uniformly random bytes, scale fields overwritten with fp16 values of the same
    distributions as :func:`make_blocks` modes "nominal" / "signed" (the values differ from make_blocks for the same seed:
    another generator)."""
    import torch
    qtype = GGMLQuantizationType(int(qtype))
    _, ts = GGML_QUANT_SIZES[qtype]
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    blocks = torch.randint(0, 256, (n_blocks, ts), dtype=torch.uint8, device=device, generator=g)
    if mode == "raw" or n_blocks == 0:
        return blocks.reshape(-1)
    if mode not in ("nominal", "signed"):
        raise ValueError(f"device_blocks: mode {mode!r}")
    lo, hi = (1e-3, 2.1e-2) if qtype in LEGACY_QTYPES else (1e-4, 2e-3)
    for off in SCALE_FIELDS[qtype]:
        vals = (torch.rand(n_blocks, device=device, generator=g) * (hi - lo) + lo).to(torch.float16)
        if mode == "signed":
            vals = torch.where(torch.rand(n_blocks, device=device, generator=g) < 0.5, -vals, vals)
        blocks[:, off:off + 2] = vals.view(torch.uint8).reshape(n_blocks, 2)
    return blocks.reshape(-1)


def k_scmn_exhaustive_blocks(qtype, which="d", seed=0):
    """Q4_K / Q5_K over the whole domain of one of their two products: 65 536 x 8 blocks in which ``which`` ("d" or "dmin") takes every fp16 bit
    pattern, the 6-bit sub-block factor that multiplies it (``sc`` for d, ``m`` for dmin; dequant.py:129-139,159-195) takes every value 0..63 against
    each pattern, and every sub-block holds every quant value (16 for Q4_K, in both nibble positions; 32 for Q5_K).  The other scale field is nominal
    with a random sign, the other 6-bit factors random.  Returns (n_blocks, type_size) uint8."""
    qtype = GGMLQuantizationType(int(qtype))
    ts = {GGMLQuantizationType.Q4_K: 144, GGMLQuantizationType.Q5_K: 176}[qtype]
    rng = np.random.default_rng(seed)
    n = 65536 * 8
    blocks = np.zeros((n, ts), dtype=np.uint8)
    pats = np.repeat(np.arange(65536, dtype=np.uint32), 8)
    other = (rng.uniform(1e-4, 2e-3, size=n) * rng.choice([-1.0, 1.0], size=n)).astype(np.float16).view(np.uint16).astype(np.uint32)
    d_bits, m_bits = (pats, other) if which == "d" else (other, pats)
    blocks[:, 0], blocks[:, 1] = (d_bits & 0xFF).astype(np.uint8), (d_bits >> 8).astype(np.uint8)
    blocks[:, 2], blocks[:, 3] = (m_bits & 0xFF).astype(np.uint8), (m_bits >> 8).astype(np.uint8)
    swept = (8 * (np.arange(n, dtype=np.uint32) % 8))[:, None] + np.arange(8, dtype=np.uint32)[None, :]        # sub-block j of block b: 8 (b % 8) + j
    rand6 = rng.integers(0, 64, size=(n, 8)).astype(np.uint32)
    sc, mn = (swept, rand6) if which == "d" else (rand6, swept)
    for j in range(4):                                                                                         # inverse of get_scale_min
        blocks[:, 4 + j] = ((sc[:, j] & 63) | ((sc[:, j + 4] >> 4) << 6)).astype(np.uint8)
        blocks[:, 8 + j] = ((mn[:, j] & 63) | ((mn[:, j + 4] >> 4) << 6)).astype(np.uint8)
        blocks[:, 12 + j] = ((sc[:, j + 4] & 15) | ((mn[:, j + 4] & 15) << 4)).astype(np.uint8)
    blocks[:, ts - 128:] = ((np.arange(128, dtype=np.uint32) % 16) * 0x11).astype(np.uint8)[None, :]             # byte = 0xkk: nibble k in both positions
    if qtype == GGMLQuantizationType.Q5_K:
        blocks[:, 16:48] = np.where(np.arange(32) >= 16, 0xFF, 0x00).astype(np.uint8)[None, :]                  # qh[l]: high bit of element l of EVERY sub-block = (l >= 16)
    return blocks


def q4_k_exhaustive_blocks(which="d", seed=0):
    """:func:`k_scmn_exhaustive_blocks` for Q4_K, the headline format."""
    return k_scmn_exhaustive_blocks(GGMLQuantizationType.Q4_K, which, seed)


def k_exhaustive_blocks(qtype, which="d", seed=0, d_range=None):
    """The super-block formats over the whole domain of a scale product: every fp16 bit pattern of the scale field ``which`` ("d"; "dmin" where the
    format has one) against EVERY value of the integer sub-block factor that multiplies it, with every quant value present in every sub-block (Q6_K:
    random quants -- 64 values do not fit 16 elements).  Blocks per pattern: Q2_K 1 (16 sub-blocks = the 16 4-bit factors), Q3_K 4 (64 6-bit scales),
    Q4_K / Q5_K / IQ4_XS 8 (64 6-bit factors), Q6_K 16 (256 int8 scales).  ``d_range`` = (lo, hi) restricts the patterns to [lo, hi) so that the big
    ones can be walked in pieces.  Layouts: dequant.py:141-285.  Returns (n_blocks, type_size) uint8."""
    Q = GGMLQuantizationType
    qtype = Q(int(qtype))
    if qtype in (Q.Q4_K, Q.Q5_K):
        b = k_scmn_exhaustive_blocks(qtype, which, seed)
        return b if d_range is None else b[8 * d_range[0]:8 * d_range[1]]
    lo, hi = d_range or (0, 65536)
    reps = {Q.Q2_K: 1, Q.Q3_K: 4, Q.Q6_K: 16, Q.IQ4_XS: 8}[qtype]
    ts = GGML_QUANT_SIZES[qtype][1]
    rng = np.random.default_rng(seed)
    n = (hi - lo) * reps
    blocks = np.zeros((n, ts), dtype=np.uint8)
    pats = np.repeat(np.arange(lo, hi, dtype=np.uint32), reps)
    rep = np.arange(n, dtype=np.uint32) % reps
    other = (rng.uniform(1e-4, 2e-3, size=n) * rng.choice([-1.0, 1.0], size=n)).astype(np.float16).view(np.uint16).astype(np.uint32)

    def put16(off, bits):
        blocks[:, off], blocks[:, off + 1] = (bits & 0xFF).astype(np.uint8), (bits >> 8).astype(np.uint8)

    crumbs = ((np.arange(32, dtype=np.uint32) % 4) * 0x55).astype(np.uint8)                     # byte l: all four 2-bit fields = l % 4
    if qtype == Q.Q2_K:                                                                          # [scales 16][qs 64][d][dmin]; scales = sc | m << 4
        if which not in ("d", "dmin"):
            raise ValueError(which)
        swept, rand4 = np.arange(16, dtype=np.uint32)[None, :].repeat(n, 0), rng.integers(0, 16, size=(n, 16)).astype(np.uint32)
        sc, m = (swept, rand4) if which == "d" else (rand4, swept)
        blocks[:, 0:16] = (sc | (m << 4)).astype(np.uint8)
        blocks[:, 16:80] = np.tile(crumbs, 2)[None, :]
        put16(80, pats if which == "d" else other)
        put16(82, other if which == "d" else pats)
    elif qtype == Q.Q3_K:                                                                        # [hmask 32][qs 64][scales 12][d]; 6-bit scales, see dequant.py:203-210
        v = (16 * rep)[:, None] + np.arange(16, dtype=np.uint32)[None, :]                        # stored scale value of sub-block j (0..63 = -32..31)
        for k in range(8):
            blocks[:, 96 + k] = ((v[:, k] & 15) | ((v[:, k + 8] & 15) << 4)).astype(np.uint8)
        for k in range(4):
            blocks[:, 104 + k] = sum((((v[:, k + 4 * g] >> 4) & 3) << (2 * g)) for g in range(4)).astype(np.uint8)
        blocks[:, 0:32] = np.where((np.arange(32) >> 2) & 1, 0xFF, 0x00).astype(np.uint8)[None, :]   # hmask[l]: the high bit of element l in every 32-run
        blocks[:, 32:96] = np.tile(crumbs, 2)[None, :]
        put16(108, pats)
    elif qtype == Q.Q6_K:                                                                        # [ql 128][qh 64][scales i8 x16][d]
        blocks[:, 0:192] = rng.integers(0, 256, size=(n, 192), dtype=np.uint8)
        blocks[:, 192:208] = ((16 * rep)[:, None] + np.arange(16, dtype=np.uint32)[None, :]).astype(np.uint8)   # all 256 int8 bit patterns
        put16(208, pats)
    else:                                                                                        # IQ4_XS: [d][scales_h u16][scales_l 4][qs 128]
        ls = (8 * rep)[:, None] + np.arange(8, dtype=np.uint32)[None, :]                         # stored 6-bit scale of sub-block ib
        put16(0, pats)
        put16(2, sum(((ls[:, ib] >> 4) & 3) << (2 * ib) for ib in range(8)))
        for k in range(4):
            blocks[:, 4 + k] = ((ls[:, 2 * k] & 15) | ((ls[:, 2 * k + 1] & 15) << 4)).astype(np.uint8)
        blocks[:, 8:136] = np.tile((np.arange(16, dtype=np.uint32) * 0x11).astype(np.uint8), 8)[None, :]        # byte k of a sub-block = 0xkk
    return blocks
