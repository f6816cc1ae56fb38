"""Whole-weight-set parity: EVERY output tensor of a DequantPlan against the oracle -- TEST INFRASTRUCTURE (tests/ and the
parity statement of bench.py's cpu_baseline leg; never the thing measured or shipped).

What is compared, per tensor (reference dequant.py:30-44 followed by the ``.to(dtype)`` of dequant.py:23):
  * the whole tensor against the oracle's AVX2+F16C leg (oracle/ggq_oracle_simd.c; OpenMP, ~100x the soft-float checker, so a
    12 G-element weight set takes seconds) -- uploaded and compared on the GPU, bit for bit;
  * that leg itself against the soft-float checker (ggq_oracle.c, the restatement pinned to the reference's golden vectors) on
    three windows of every tensor: first blocks, a middle window, last blocks (on hosts without AVX2/F16C the soft-float
    checker does the whole tensor);
  * bf16 / fp32 results: the expected tensor is the fp16 result put through torch's own ``.to(dtype)`` on the GPU -- literally the
    reference's op (dequant.py:23) -- and, on the same three windows, through the C oracle's cast.
Only the stock fp16 arithmetic (dequant_dtype None) is covered here; the other arithmetic modes have their own tests.
"""
import numpy as np

from . import cast_f16_to_bf16_bits, dequant_f16, geometry, simd_available

_WINDOW_BLOCKS = {32: 4096, 256: 512}       # 131072 elements per window


def check_tensor(packed_dev, qtype, out_dev, windows=True, threads=None):
    """None if ``out_dev`` (fp16 / bf16 / fp32 tensor on the GPU) is what the reference computes from ``packed_dev``, else a
    short description of the first difference.  ``threads``: OpenMP team of the AVX2 leg (several ranks of one node check at once)."""
    import torch
    bs, ts = geometry(qtype)
    host = packed_dev.reshape(-1).cpu().numpy()
    n_blocks = host.size // ts
    simd = simd_available()
    want = dequant_f16(qtype, host, simd=simd, threads=threads).view(np.uint16)
    spots = []
    if windows and n_blocks:
        w = min(_WINDOW_BLOCKS[bs], n_blocks)
        spots = sorted({0, max(0, n_blocks // 2 - w // 2), n_blocks - w})
        if simd:
            for b0 in spots:
                soft = dequant_f16(qtype, host[b0 * ts:(b0 + w) * ts]).view(np.uint16)
                if not np.array_equal(soft, want[b0 * bs:(b0 + w) * bs]):
                    return f"oracle SIMD leg != soft-float checker in blocks [{b0}, {b0 + w})"
    exp = torch.from_numpy(want.view(np.int16)).to(out_dev.device).view(torch.float16)
    got = out_dev.reshape(-1)
    if got.numel() != exp.numel():
        return f"{got.numel()} elements, expected {exp.numel()}"
    if out_dev.dtype is not torch.float16:
        exp = exp.to(out_dev.dtype)                      # dequant.py:23, the reference's own op
    view = torch.int32 if out_dev.dtype is torch.float32 else torch.int16
    if not torch.equal(got.view(view), exp.view(view)):
        idx = int((got.view(view) != exp.view(view)).nonzero()[0])
        return f"element {idx} (block {idx // bs}): got {got[idx].item()!r}, expected {exp[idx].item()!r}"
    if out_dev.dtype is torch.bfloat16:
        for b0 in spots:
            w = min(_WINDOW_BLOCKS[bs], n_blocks)
            c = cast_f16_to_bf16_bits(want[b0 * bs:(b0 + w) * bs])
            g = got[b0 * bs:(b0 + w) * bs].view(torch.int16).cpu().numpy().view(np.uint16)
            if not np.array_equal(c, g):
                return f"bf16 cast differs from the C oracle's in blocks [{b0}, {b0 + w})"
    return None


def check_plan(packed, qtypes, outputs, windows=True, threads=None):
    """(tensors checked, [(index, description) of every tensor that differs])."""
    bad = []
    for i, (p, q, o) in enumerate(zip(packed, qtypes, outputs)):
        msg = check_tensor(p, q, o, windows=windows, threads=threads)
        if msg is not None:
            bad.append((i, msg))
    return len(outputs), bad
