"""Fused dequantize + GEMM on the matrix cores (include/ggq.h ggq_linear_mfma, opt-in) on an MI355X.

A floating-point contraction: parity is a tolerance against an fp64 evaluation of the same op on the ORACLE's weights (the
reference's values, cast to the activation dtype as dequantize_tensor does), with the worst-case fp32-accumulation bound of
tests/test_gpu_linear.py -- plus cases where the fp32 sums are EXACT (integer-valued activations, power-of-two scales) and the
result must therefore equal the correctly rounded fp64 value bit for bit: a dropped or duplicated chunk cannot hide in those.
"""
import numpy as np
import pytest
import torch

import oracle
from test_gpu_linear import _check, _dense_weight, DT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ALL = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "IQ4_NL", "IQ4_XS"]


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_mfma_linear_against_fp64_reference(pkg, name, kind):
    q = pkg.qtypes.Q[name]
    dtype, eps = DT[kind]
    g = torch.Generator(device=DEV)
    g.manual_seed(9)
    # (rows, cols, m, bias, tile): ragged output columns (rows % 32), ragged rows of x (m % 32), one row, several tiles of x,
    # 1 .. 12 spans of K (fewer spans than waves; not a multiple of the 4-way split)
    for rows, cols, m, with_bias, tile in ((203, 3072, 1, True, 0), (17, 256, 40, False, 0), (64, 768, 33, True, 32), (96, 1024, 300, False, 64),
                                           (333, 512, 129, True, 128), (40, 2304, 260, True, 256), (32, 1280, 96, False, 0),
                                           # the shared-tile kernel (256 x 256 output tiles, ggq_gemm.hpp): several tiles both ways with ragged
                                           # edges; exactly one tile and one span; 9 spans (the staging buffer is refilled 8 times) x 3 tiles of x;
                                           # fewer output columns than one MFMA block; shape picked by the library (2 tiles: the K-split kernel)
                                           (520, 1024, 300, True, 256), (256, 256, 256, False, 256), (264, 2304, 513, True, 256), (8, 512, 1000, False, 256),
                                           (304, 768, 200, True, 0)):
        blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=rows + cols, mode="signed")
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(dtype)
        bias = (torch.randn(rows, device=DEV, generator=g) * 0.01).to(dtype) if with_bias else None
        y = pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None)
        assert y.shape == (m, rows) and y.dtype == dtype
        _check(y, x, _dense_weight(q, blocks, kind, rows, cols), bias, eps, cols)
        assert torch.equal(y, pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None))          # deterministic: fixed reduction order


@pytest.mark.parametrize("name", ["Q4_K", "Q8_0", "Q6_K"])
def test_auto_dispatch_reaches_the_shared_tile_kernel(pkg, name):
    """tile_rows=0 with enough 256 x 256 tiles (>= 80: here 9 x 9 = 81) must run the shared-tile kernel -- its bits, not the K-split kernel's
    (no test reached that kernel through the library's own choice); with 4 x 9 = 36 tiles the K-split kernel; and the host-side
    auto policy declines more than 256 rows of x unless told otherwise (dequantize + F.linear is faster there)."""
    q = pkg.qtypes.Q[name]
    g = torch.Generator(device=DEV)
    g.manual_seed(31)
    rows, cols, m = 2304, 256, 2304
    blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=77, mode="signed")
    w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
    x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(torch.bfloat16)
    with pytest.raises(pkg.dequant.GGQUnsupported):
        pkg.fused.linear_mfma(x, w)                                                         # 2304 rows of x: declined by default
    # above 128 rows of x the policy counts multiply-accumulates (fused.AUTO_MAX_MACS = 8e9, 5e9 for the formats with the dearer decode): a 16384 x 2048 weight is served at
    # 200 rows of x (6.7e9) and declined at 256 (8.6e9) in Q4_K, served at 140 (4.7e9) and declined at 200 in Q8_0 / Q6_K
    bs, ts = pkg.qtypes.block_geometry(q)
    big = pkg.ops.GGMLTensor(torch.zeros(16384 * 2048 // bs * ts, dtype=torch.uint8, device=DEV), tensor_type=q, tensor_shape=(16384, 2048))
    xb = torch.zeros(256, 2048, dtype=torch.bfloat16, device=DEV)
    served, declined = (200, 256) if name == "Q4_K" else (140, 200)
    assert pkg.fused.linear_mfma(xb[:served], big).shape == (served, 16384)
    with pytest.raises(pkg.dequant.GGQUnsupported):
        pkg.fused.linear_mfma(xb[:declined], big)
    del big, xb
    auto = pkg.fused.linear_mfma(x, w, auto_max_rows=None)
    tile, ksplit = pkg.fused.linear_mfma(x, w, tile_rows=256), pkg.fused.linear_mfma(x, w, tile_rows=128)
    _check(auto, x, _dense_weight(q, blocks, "bf16", rows, cols), None, DT["bf16"][1], cols)
    assert torch.equal(auto, tile)
    assert not torch.equal(ksplit, tile)                                                    # (the K-split kernel sums in another order: the two are distinguishable)
    small = pkg.fused.linear_mfma(x[:1024], w, auto_max_rows=None)                          # 4 x 9 tiles: the library stays with the K-split kernel
    assert torch.equal(small, pkg.fused.linear_mfma(x[:1024], w, tile_rows=128)) and not torch.equal(small, tile[:1024])
    assert torch.equal(pkg.fused.linear_mfma(x[:200], w), pkg.fused.linear_mfma(x[:200], w, tile_rows=64))       # <= 256 rows on a small weight: served by default, every format
    # an output view that is not 16-byte aligned cannot take the shared-tile epilogue's vector stores: the C entry point says so
    import ctypes
    L = pkg._native.lib()
    y = torch.empty(m * rows + 8, dtype=torch.bfloat16, device=DEV)
    rc = L.ggq_linear_mfma(int(q), w.data_ptr(), rows, cols, x.data_ptr(), m, None, y.data_ptr() + 2, pkg._native.BF16, 256, torch.cuda.current_stream().cuda_stream)
    assert rc == pkg._native.GGQ_ERR_ALIGN


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_mfma16_linear_against_fp64_reference(pkg, name, kind):
    """The 16-row kernel (csrc/ggq_mfma16.hpp, tile_rows=16; round 6): 16 output columns per workgroup, K split over 2..16 waves chosen by the
    library.  One row of x, ragged rows / output columns against 16, one block and two blocks of x per tile, several tiles of x (m > 32), 1..48 spans
    (fewer spans than the smallest workgroup; more than its largest), with and without bias; deterministic."""
    q = pkg.qtypes.Q[name]
    dtype, eps = DT[kind]
    g = torch.Generator(device=DEV)
    g.manual_seed(23)
    for rows, cols, m, with_bias in ((203, 3072, 1, True), (17, 256, 4, False), (64, 768, 16, True), (96, 1024, 17, False), (333, 512, 32, True),
                                     (31, 4096, 13, False), (40, 12288, 9, True), (50, 1280, 70, True), (1, 256, 1, True), (18, 2304, 31, False)):
        blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=rows + cols, mode="signed")
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(dtype)
        bias = (torch.randn(rows, device=DEV, generator=g) * 0.01).to(dtype) if with_bias else None
        y = pkg.fused.linear_mfma(x, w, bias, tile_rows=16, auto_max_rows=None)
        assert y.shape == (m, rows) and y.dtype == dtype
        _check(y, x, _dense_weight(q, blocks, kind, rows, cols), bias, eps, cols)
        assert torch.equal(y, pkg.fused.linear_mfma(x, w, bias, tile_rows=16, auto_max_rows=None))


@pytest.mark.parametrize("name", ["Q4_K", "Q5_K", "Q8_0", "Q6_K"])
@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_k_split_across_workgroups_with_a_workspace(pkg, name, kind):
    """Round 6: a weight with few, long rows gets its K split across workgroups as well (ggq_linear_mfma_ws + splitk_reduce) when the caller provides scratch -- fused.linear_mfma
    asks ggq_linear_mfma_workspace() once per shape and allocates it.  Shapes with 16+ spans and < 256 workgroups: ragged rows of x and of W, one and several tiles of x, a slice count that
    does not divide the spans; tolerance against fp64 on the oracle's weights, deterministic, an exact-arithmetic case (a slice dropped or added twice cannot hide), and the C entry point
    directly: a NULL or too-small workspace is not an error, it keeps K inside the workgroups (bit-equal to ggq_linear_mfma)."""
    q = pkg.qtypes.Q[name]
    dtype, eps = DT[kind]
    L, nat = pkg._native.lib(), pkg._native
    g = torch.Generator(device=DEV)
    g.manual_seed(29)
    for rows, cols, m, with_bias, tile in ((96, 8192, 33, True, 0), (40, 12288, 64, False, 0), (200, 4352, 100, True, 64), (64, 5120, 130, True, 128)):
        ws = int(L.ggq_linear_mfma_workspace(int(q), rows, cols, m, tile))
        assert ws > 0 and ws % (m * rows * 4) == 0, (rows, cols, m, ws)                     # zs slices of m x rows fp32 partials
        blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=rows + cols, mode="signed")
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(dtype)
        bias = (torch.randn(rows, device=DEV, generator=g) * 0.01).to(dtype) if with_bias else None
        y = pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None)
        _check(y, x, _dense_weight(q, blocks, kind, rows, cols), bias, eps, cols)
        assert torch.equal(y, pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None))
        # the C ABI directly: no workspace / one that is too small -> the launch keeps K inside the workgroups and returns what ggq_linear_mfma returns
        code = nat.BF16 if kind == "bf16" else nat.F16
        stream = torch.cuda.current_stream().cuda_stream
        args = (int(q), w.data_ptr(), rows, cols, x.data_ptr(), m, None if bias is None else bias.data_ptr())
        y0, y1, y2, y3 = (torch.empty(m, rows, dtype=dtype, device=DEV) for _ in range(4))
        small = torch.empty(64, dtype=torch.uint8, device=DEV)
        full = torch.empty(ws, dtype=torch.uint8, device=DEV)
        assert L.ggq_linear_mfma(*args, y0.data_ptr(), code, tile, stream) == nat.GGQ_OK
        assert L.ggq_linear_mfma_ws(*args, y1.data_ptr(), code, tile, None, 0, stream) == nat.GGQ_OK
        assert L.ggq_linear_mfma_ws(*args, y2.data_ptr(), code, tile, small.data_ptr(), 64, stream) == nat.GGQ_OK
        assert L.ggq_linear_mfma_ws(*args, y3.data_ptr(), code, tile, full.data_ptr(), ws, stream) == nat.GGQ_OK
        assert L.ggq_linear_mfma_ws(*args, y3.data_ptr(), code, tile, full.data_ptr() + 4, ws - 4, stream) == nat.GGQ_ERR_ALIGN
        assert torch.equal(y0, y1) and torch.equal(y0, y2) and torch.equal(y3, y)
    assert L.ggq_linear_mfma_workspace(int(q), 12288, 3072, 64, 0) == 0                      # enough workgroups: no split
    assert L.ggq_linear_mfma_workspace(int(q), 96, 8192, 4, 0) == 0                          # 4 rows of x: the 16-row kernel, which fills the chip by itself
    # exact arithmetic through the split
    rows, cols, m = 70, 8192, 37
    eb = _exact_blocks(pkg, q, rows, cols, seed=6)
    ew = pkg.ops.GGMLTensor(torch.from_numpy(eb).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
    w64 = _dense_weight(q, eb, kind, rows, cols)
    xi = torch.randint(-2, 3, (m, cols), device=DEV, generator=g).to(dtype)
    x64 = xi.double().cpu().numpy()
    assert (np.abs(x64) @ np.abs(w64).T).max() < 2.0 ** 16 and L.ggq_linear_mfma_workspace(int(q), rows, cols, m, 0) > 0
    assert torch.equal(pkg.fused.linear_mfma(xi, ew).cpu(), torch.from_numpy(x64 @ w64.T).to(dtype)), (name, kind, "exact through split-K")


def _exact_blocks(pkg, q, rows, cols, seed):
    """Packed blocks whose scale fields are powers of two: every weight, every product with a small integer and every partial
    sum is exactly representable in fp32."""
    blocks = pkg.synth.make_blocks(q, rows * cols // pkg.qtypes.block_geometry(q)[0], seed=seed, mode="raw")
    for off in pkg.qtypes.SCALE_FIELDS[q]:
        blocks[:, off], blocks[:, off + 1] = 0x00, 0x1C           # fp16 0x1C00 = 2^-8
    return blocks.reshape(-1)


@pytest.mark.parametrize("name", ["Q8_0", "Q4_0", "Q5_1", "Q4_K", "Q2_K", "Q6_K"])
@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_exact_arithmetic_cases_are_bit_equal(pkg, name, kind):
    """Integer-valued x in [-4, 4], scales 2^-8: all products and all fp32 partial sums are exact whatever their order, so BOTH
    fused kernels must return the correctly rounded exact value -- bit for bit (Weak #3)."""
    q = pkg.qtypes.Q[name]
    dtype, _ = DT[kind]
    rows, cols = 70, 1536
    blocks = _exact_blocks(pkg, q, rows, cols, seed=77)
    w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
    w64 = _dense_weight(q, blocks, kind, rows, cols)
    assert np.all(w64 * 2.0 ** 8 == np.round(w64 * 2.0 ** 8)) and np.abs(w64).max() < 64       # every weight a multiple of 2^-8
    g = torch.Generator(device=DEV).manual_seed(3)
    mfma16 = lambda t: (lambda x, w: pkg.fused.linear_mfma(x, w, tile_rows=t))
    forced = lambda x, w: pkg.fused.linear_mfma(x, w, auto_max_rows=None)        # (whatever the auto policy thinks of 160 rows)
    for m, fn in ((37, pkg.fused.linear_mfma), (3, pkg.fused.linear_small), (160, forced), (1, mfma16(16)), (13, mfma16(16)), (29, mfma16(16)),
                  (5, mfma16(0)), (12, mfma16(0))):
        x = torch.randint(-4, 5, (m, cols), device=DEV, generator=g).to(dtype)
        x64 = x.double().cpu().numpy()
        assert (np.abs(x64) @ np.abs(w64).T).max() < 2.0 ** 16       # every partial sum, in ANY order, is a multiple of 2^-8 below 2^16: exact in fp32
        exact = x64 @ w64.T
        want = torch.from_numpy(exact).to(dtype)                                             # ONE rounding, as the kernel's store
        got = fn(x, w)
        assert torch.equal(got.cpu(), want), (name, kind, m)
    # the shared-tile kernel: 2 x 2 tiles with ragged edges, 6 spans; a weight tile decoded twice, skipped or written to the wrong
    # LDS column shows up as a wrong integer multiple of 2^-8
    rows = 264
    blocks = _exact_blocks(pkg, q, rows, cols, seed=78)
    w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
    w64 = _dense_weight(q, blocks, kind, rows, cols)
    x = torch.randint(-4, 5, (300, cols), device=DEV, generator=g).to(dtype)
    x64 = x.double().cpu().numpy()
    assert (np.abs(x64) @ np.abs(w64).T).max() < 2.0 ** 16
    want = torch.from_numpy(x64 @ w64.T).to(dtype)
    assert torch.equal(pkg.fused.linear_mfma(x, w, tile_rows=256).cpu(), want), (name, kind, "shared-tile kernel")


@pytest.mark.parametrize("name", ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "IQ4_NL"])
@pytest.mark.parametrize("kind", ["f16", "bf16"])
def test_short_last_span_for_32_element_blocks(pkg, name, kind):
    """Round 5: the contraction may end in a span shorter than 256 elements (a multiple of 64) for the 32-element legacy blocks -- SD3.5-large's
    2432-column layers (Q5_0 in the Q4_K_M mix: lcpp.patch falls back when cols % 256 != 0).  One short span only (192), 1.25 spans (320: the short
    one belongs to wave 1), 9.5 spans (2432: wave 1 again, after two full ones), 4.75 spans (1216: wave 0's second); every K-split tile shape;
    tolerance against fp64 on the oracle's weights AND an exact-arithmetic case per shape (a dropped or doubled chunk of the tail cannot hide)."""
    q = pkg.qtypes.Q[name]
    dtype, eps = DT[kind]
    g = torch.Generator(device=DEV)
    g.manual_seed(19)
    for rows, cols, m, with_bias, tile in ((70, 192, 5, True, 0), (33, 320, 40, False, 0), (96, 2432, 31, True, 32), (64, 2432, 100, False, 64),
                                           (40, 1216, 260, True, 128), (7296 // 8, 2432, 256, True, 0), (50, 448, 64, False, 0),
                                           # ... and the 16-row kernel (round 6), whose short span is handled by zeroed bytes instead of skipped k-steps:
                                           # 64 / 128 / 192 elements in the last span, alone and after whole spans, one and two blocks of x, the last row of x
                                           (70, 192, 5, True, 16), (33, 320, 16, False, 16), (96, 2432, 31, True, 16), (23, 2432, 1, False, 16), (40, 1216, 20, True, 16),
                                           (19, 64, 3, False, 16), (50, 448, 32, False, 16), (35, 384, 7, True, 16)):
        blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=rows + cols, mode="signed")
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(dtype)
        bias = (torch.randn(rows, device=DEV, generator=g) * 0.01).to(dtype) if with_bias else None
        y = pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None)
        assert y.shape == (m, rows) and y.dtype == dtype
        _check(y, x, _dense_weight(q, blocks, kind, rows, cols), bias, eps, cols)
        assert torch.equal(y, pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None))
        # exact arithmetic on the same shape: integer x, power-of-two scales -> the correctly rounded exact value, bit for bit
        eb = _exact_blocks(pkg, q, rows, cols, seed=5)
        ew = pkg.ops.GGMLTensor(torch.from_numpy(eb).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        w64 = _dense_weight(q, eb, kind, rows, cols)
        xi = torch.randint(-4, 5, (m, cols), device=DEV, generator=g).to(dtype)
        x64 = xi.double().cpu().numpy()
        if (np.abs(x64) @ np.abs(w64).T).max() < 2.0 ** 16:
            want = torch.from_numpy(x64 @ w64.T).to(dtype)
            assert torch.equal(pkg.fused.linear_mfma(xi, ew, tile_rows=tile, auto_max_rows=None).cpu(), want), (name, kind, rows, cols, m, "exact")


def test_mfma_linear_limits(pkg):
    Q, ops, GGQUnsupported = pkg.qtypes.Q, pkg.ops, pkg.dequant.GGQUnsupported
    mk = lambda q, shape, **kw: ops.GGMLTensor(torch.from_numpy(pkg.synth.make_tensor_bytes(q, shape, seed=1)).to(DEV), tensor_type=q, tensor_shape=shape, **kw)
    x = torch.randn(8, 512, device=DEV, dtype=torch.bfloat16)
    w = mk(Q.Q4_K, (64, 512))
    y = pkg.fused.linear_mfma(x.reshape(2, 4, 512), w)
    assert y.shape == (2, 4, 64)
    assert torch.equal(pkg.fused.linear_mfma(x.t().contiguous().t(), w), y.reshape(8, 64))   # non-contiguous x: handled by a copy
    for bad in (lambda: pkg.fused.linear_mfma(x.float(), w),                                  # fp32 activations
                lambda: pkg.fused.linear_mfma(x, w, dequant_dtype=torch.float32),
                lambda: pkg.fused.linear_mfma(x[:, :96], mk(Q.Q8_0, (64, 96))),               # 32-element blocks: the last span must be a multiple of 64
                lambda: pkg.fused.linear_mfma(x[:, :384], mk(Q.Q8_0, (64, 384)), tile_rows=256),  # ... and the shared-tile kernel takes whole spans only
                lambda: pkg.fused.linear_mfma(x, mk(Q.Q4_K, (64, 512)).as_subclass(torch.Tensor)),  # not a GGMLTensor: no tensor_type
                lambda: pkg.fused.linear_mfma(x, mk(Q.Q4_K, (64, 512), patches=[("p", "k")])),
                lambda: pkg.fused.linear_mfma(x.cpu(), w),
                lambda: pkg.fused.linear_mfma(x, w, tile_rows=48),
                lambda: pkg.fused.linear_mfma(x, mk(Q.Q4_K, (60, 512)), tile_rows=256),        # the shared-tile kernel needs rows % 8 == 0
                lambda: pkg.fused.linear_mfma(x, w, tile_rows=-96)):
        with pytest.raises(GGQUnsupported):
            bad()
    lib, nat = pkg._native.lib(), pkg._native
    assert lib.ggq_linear_mfma(99, 16, 32, 256, 16, 1, None, 16, 1, 0, None) == nat.GGQ_ERR_QTYPE
    assert lib.ggq_linear_mfma(12, 16, 32, 256, 16, 1, None, 16, 2, 0, None) == nat.GGQ_ERR_ARG      # fp32
    assert lib.ggq_linear_mfma(12, 24, 32, 256, 16, 1, None, 16, 1, 0, None) == nat.GGQ_ERR_ALIGN
    assert lib.ggq_linear_mfma(12, None, 0, 256, None, 1, None, None, 1, 0, None) == nat.GGQ_OK       # nothing to do


def test_mfma_randomized_sweep(pkg):
    """80 seeded random cases: format, dtype, rows (ragged against 32), cols (1..14 spans of 256), rows of x (1..700), tile, bias."""
    rng = np.random.default_rng(2026)
    g = torch.Generator(device=DEV).manual_seed(11)
    for case in range(80):
        name = ALL[int(rng.integers(len(ALL)))]
        kind = ("f16", "bf16")[int(rng.integers(2))]
        q = pkg.qtypes.Q[name]
        dtype, eps = DT[kind]
        rows, cols = int(rng.integers(1, 400)), 256 * int(rng.integers(1, 15))
        m = int(rng.choice([1, 2, 7, 31, 32, 33, 64, 65, 100, 128, 129, 257, 700]))
        tile = int(rng.choice([0, 0, 32, 64, 128, 256]))
        if case >= 60:
            tile, m = 16, int(rng.choice([1, 2, 5, 15, 16, 17, 31, 32, 33, 47]))     # the last 20 cases: the 16-row kernel (round 6)
        if tile == 256:
            rows = (rows + 7) // 8 * 8                         # the shared-tile kernel stores 16 bytes (8 columns) per lane
        blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=1000 + case, mode="signed")
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(dtype)
        bias = (torch.randn(rows, device=DEV, generator=g) * 0.01).to(dtype) if rng.integers(2) else None
        y = pkg.fused.linear_mfma(x, w, bias, tile_rows=tile, auto_max_rows=None)
        try:
            _check(y, x, _dense_weight(q, blocks, kind, rows, cols), bias, eps, cols)
        except AssertionError as e:
            raise AssertionError(f"case {case}: {name} {kind} rows={rows} cols={cols} m={m} tile={tile} bias={bias is not None}: {e}") from None
