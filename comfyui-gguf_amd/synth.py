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
