"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI by the mirrored
Python interface, against (a) the committed golden vectors produced by the reference's own
dequant.py, (b) the CPU oracle on seeded inputs at ragged / adversarial / full BASELINE sizes,
(c) size-independent properties.  Tolerance: NONE -- every comparison is bit-exact (NaN payloads
canonicalised); the 1-ULP allowance of the north star is unused slack."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

ALL = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "IQ4_NL", "IQ4_XS"]
DEV = "cuda:0"


def _bits16(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1)


def _canon_bf16(bits):
    bits = np.asarray(bits, dtype=np.uint16).copy()
    bits[(bits & 0x7FFF) > 0x7F80] = 0x7FC0
    return bits


def _carrier(pkg, blocks, q, shape=None):
    bs, _ = pkg.qtypes.block_geometry(q)
    shape = shape or (blocks.shape[0], bs)
    return pkg.ops.GGMLTensor(torch.from_numpy(np.ascontiguousarray(blocks).reshape(-1)).to(DEV), tensor_type=q, tensor_shape=shape)


@pytest.fixture(scope="module", autouse=True)
def _need_native(pkg):
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    pkg._native.lib()          # raises if the HIP extension is missing: no silent fallback


@pytest.mark.parametrize("name", ALL)
def test_golden_vectors(pkg, golden_dir, name):
    """HIP output == the reference's own dequant.py output, for every committed vector."""
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    q = pkg.qtypes.Q[name]
    blocks = g["blocks"]
    n, bs = blocks.shape[0], pkg.qtypes.block_geometry(q)[0]
    data = torch.from_numpy(blocks.reshape(-1).copy()).to(DEV)
    out = pkg.dequant.dequantize(data, q, (n, bs))                          # dequant.py:30
    assert out.dtype == torch.float16 and tuple(out.shape) == (n, bs) and out.is_cuda
    assert np.array_equal(oracle.canon_nan_f16(_bits16(out)), oracle.canon_nan_f16(g["out_f16"].reshape(-1)))
    half = n * bs // 2                                                       # nominal + signed runs: no NaN, raw bits
    assert np.array_equal(_bits16(out)[:half], g["out_f16"].reshape(-1)[:half])
    # the per-format block function of dequantize_functions (dequant.py:43)
    fn = pkg.dequant.dequantize_functions[q]
    out2 = fn(data.reshape(n, -1), *pkg.qtypes.block_geometry(q))
    assert tuple(out2.shape) == (n, bs) and torch.equal(out2.view(torch.int16), out.view(torch.int16))
    # dequantize_tensor(..., dtype=bf16): fp16 dequant + ONE cast, here fused into the store (dequant.py:23)
    sub = blocks[g["sub"]]
    tb = pkg.dequant.dequantize_tensor(_carrier(pkg, sub, q), torch.bfloat16)
    assert tb.dtype == torch.bfloat16
    assert np.array_equal(_canon_bf16(_bits16(tb)), _canon_bf16(g["tensor_bf16"].reshape(-1)))


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("mode", ["nominal", "adversarial"])
def test_against_oracle_ragged_sizes(pkg, name, mode):
    """Every group-boundary case of both team shapes (one-wave teams own G blocks, workgroup teams 2G): 1 block,
    G-1, G, G+1, 2G-1, 2G, 2G+1, several groups + a ragged tail."""
    q = pkg.qtypes.Q[name]
    bs, ts = pkg.qtypes.block_geometry(q)
    G = 64 if bs == 32 else 8
    for n in (1, G - 1, G, G + 1, 2 * G - 1, 2 * G, 2 * G + 1, 5 * G + 3, 6 * G, 1031 if bs == 256 else 9001):
        blocks = pkg.synth.make_blocks(q, n, seed=n, mode=mode)
        want = oracle.dequant_f16(q, blocks)
        t = _carrier(pkg, blocks, q)
        got = pkg.dequant.dequantize_tensor(t, torch.float16)
        assert np.array_equal(oracle.canon_nan_f16(_bits16(got)), oracle.canon_nan_f16(want)), (name, n)
        got32 = pkg.dequant.dequantize_tensor(t, torch.float32)
        assert got32.dtype == torch.float32
        w32 = want.astype(np.float32)
        g32 = got32.cpu().numpy().reshape(-1)
        both_nan = np.isnan(w32) & np.isnan(g32)
        assert np.array_equal(g32.view(np.uint32)[~both_nan], w32.view(np.uint32)[~both_nan]), (name, n)


_TORCH = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def _canon(a, kind):
    a = np.ascontiguousarray(a).reshape(-1)
    if kind == "f32":
        return oracle.canon_nan(a.view(np.float32))
    return oracle.canon_nan_f16(a) if kind == "f16" else oracle.canon_nan_bf16(a)


def _raw(t):
    return t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int16).cpu().numpy().reshape(-1)


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("compute", ["bf16", "f32"])
def test_dequant_dtype_modes_golden(pkg, golden_dir, name, compute):
    """dequant_dtype = float32 / bfloat16 (Advanced loader, nodes.py:186): the block function's op
    sequence in that dtype, bit-exact to the reference's own output (committed golden vectors)."""
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    q = pkg.qtypes.Q[name]
    blocks = g["blocks"][g["sub"]]
    n, bs = blocks.shape[0], pkg.qtypes.block_geometry(q)[0]
    data = torch.from_numpy(blocks.reshape(-1).copy()).to(DEV)
    out = pkg.dequant.dequantize(data, q, (n, bs), dtype=_TORCH[compute])    # dequant.py:30, dtype = dequant_dtype
    assert out.dtype == _TORCH[compute] and tuple(out.shape) == (n, bs)
    want = g["out_f32"] if compute == "f32" else g["out_bf16"]
    assert np.array_equal(_canon(_raw(out), compute), _canon(want, compute))
    fn = pkg.dequant.dequantize_functions[q]                                 # dequant.py:43 with a dtype argument
    out2 = fn(data.reshape(n, -1), *pkg.qtypes.block_geometry(q), _TORCH[compute])
    assert out2.dtype == _TORCH[compute] and np.array_equal(_raw(out2), _raw(out))


@pytest.mark.parametrize("name", ALL)
def test_every_fp16_bit_pattern_of_every_scale_field(pkg, name):
    """EXHAUSTIVE over the scale operands: all 65 536 bit patterns of each fp16 scale field of the format (both zeros, every subnormal, every
    normal, both infinities, every NaN payload), each against a block of random quants and randomly signed other fields -- the stock fp16
    arithmetic and the two other arithmetic modes, bit-exact against the oracle (NaN payloads canonicalised)."""
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    pats = np.arange(65536, dtype=np.uint32)
    for k, off in enumerate(pkg.qtypes.SCALE_FIELDS[q]):
        blocks = pkg.synth.make_blocks(q, 65536, seed=4242 + k, mode="signed")
        blocks[:, off] = (pats & 0xFF).astype(np.uint8)
        blocks[:, off + 1] = (pats >> 8).astype(np.uint8)
        t = _carrier(pkg, blocks, q)
        for compute in ("f16", "bf16", "f32"):
            want = oracle.dequant_tensor(q, blocks, compute, compute)
            got = pkg.dequant.dequantize_tensor(t, _TORCH[compute], dequant_dtype=None if compute == "f16" else _TORCH[compute])
            assert tuple(got.shape) == (65536, bs)
            g, w = _canon(_raw(got), compute), _canon(want, compute)
            if not np.array_equal(g, w):
                bad = np.flatnonzero(g != w)
                raise AssertionError(f"{name} field@{off} {compute}: {bad.size} elements differ, first at block {bad[0] // bs} (scale bits {bad[0] // bs:#06x})")


@pytest.mark.parametrize("name", ["Q8_0", "Q4_0", "Q5_0", "IQ4_NL"])
def test_every_scale_times_every_quant_value(pkg, name):
    """The formats of the shape `d * value(q)`: EVERY (fp16 scale bit pattern, quant value) pair -- 65 536 x 256 for Q8_0 (the format the north
    star wants bit-exact), x 16 / 32 / 16 for Q4_0 / Q5_0 / IQ4_NL, each value in both nibble positions -- in the three arithmetic modes."""
    q = pkg.qtypes.Q[name]
    bs, ts = pkg.qtypes.block_geometry(q)
    reps = 8 if name == "Q8_0" else 1
    blocks = np.zeros((65536 * reps, ts), dtype=np.uint8)
    pats = np.repeat(np.arange(65536, dtype=np.uint32), reps)
    blocks[:, 0] = (pats & 0xFF).astype(np.uint8)
    blocks[:, 1] = (pats >> 8).astype(np.uint8)
    if name == "Q8_0":
        blocks[:, 2:] = (np.arange(32, dtype=np.uint32)[None, :] + 32 * (np.arange(65536 * reps, dtype=np.uint32) % 8)[:, None]).astype(np.uint8)
    else:
        nib = (np.arange(16, dtype=np.uint32) * 0x11).astype(np.uint8)            # byte k = 0xkk: element k AND element 16 + k take the value k
        blocks[:, ts - 16:] = nib[None, :]
        if name == "Q5_0":
            blocks[:, 2:6] = np.array([0x00, 0x00, 0xFF, 0xFF], dtype=np.uint8)     # qh: bit e set for e >= 16 -> q = 0..15 then 16..31
    t = _carrier(pkg, blocks, q)
    for compute in ("f16", "bf16", "f32"):
        want = oracle.dequant_tensor(q, blocks, compute, compute)
        got = pkg.dequant.dequantize_tensor(t, _TORCH[compute], dequant_dtype=None if compute == "f16" else _TORCH[compute])
        g, w = _canon(_raw(got), compute), _canon(want, compute)
        if not np.array_equal(g, w):
            bad = np.flatnonzero(g != w)
            raise AssertionError(f"{name} {compute}: {bad.size} of {g.size} (scale, quant) pairs differ, first: scale bits {pats[bad[0] // bs]:#06x}, element {bad[0] % bs}")
    if name == "Q5_0":                                                             # the other half of the (nibble position, high bit) table
        blocks[:, 2:6] = np.array([0xFF, 0xFF, 0x00, 0x00], dtype=np.uint8)
        got = pkg.dequant.dequantize_tensor(_carrier(pkg, blocks, q), torch.float16)
        assert np.array_equal(oracle.canon_nan_f16(_bits16(got)), oracle.canon_nan_f16(oracle.dequant_f16(q, blocks)))


@pytest.mark.parametrize("which", ["d", "dmin"])
@pytest.mark.parametrize("name", ["Q4_K", "Q5_K"])
def test_q4_k_every_scale_pattern_times_every_subblock_factor_times_every_quant(pkg, name, which):
    """Q4_K (the headline format) and Q5_K (the other format of FLUX's Q4_K_M mix): every (fp16 scale bit pattern, 6-bit sub-block factor, quant) triple of
    the d*sc*q product -- 65 536 x 64 x 16 (x 32 for Q5_K), 134 M elements -- and every (dmin pattern, m) pair of the subtrahend; fp16 arithmetic -> fp16
    and bf16 results, bit-exact against the oracle."""
    q = pkg.qtypes.Q[name]
    blocks = pkg.synth.k_scmn_exhaustive_blocks(q, which, seed=5)
    t = _carrier(pkg, blocks, q)
    want = oracle.dequant_f16(q, blocks)
    got = pkg.dequant.dequantize_tensor(t, torch.float16)
    g, w = oracle.canon_nan_f16(_bits16(got)), oracle.canon_nan_f16(want)
    if not np.array_equal(g, w):
        bad = np.flatnonzero(g != w)
        raise AssertionError(f"{name} {which}: {bad.size} of {g.size} elements differ, first in block {bad[0] // 256} element {bad[0] % 256}")
    del got, g, w
    got = pkg.dequant.dequantize_tensor(t, torch.bfloat16)
    assert np.array_equal(_canon(_raw(got), "bf16"), _canon(oracle.dequant_tensor(q, blocks, "f16", "bf16"), "bf16"))


@pytest.mark.parametrize("name,which", [("Q2_K", "d"), ("Q2_K", "dmin"), ("Q3_K", "d"), ("IQ4_XS", "d"), ("Q6_K", "d")])
def test_every_scale_pattern_times_every_subblock_factor_other_k_formats(pkg, name, which):
    """The remaining super-block formats over the whole domain of their scale products (synth.k_exhaustive_blocks): every fp16 bit pattern of d (and of
    dmin for Q2_K) x every value of the integer sub-block factor (16 4-bit, 64 6-bit, 256 int8) x every quant value in every sub-block (Q6_K: random
    quants); fp16 arithmetic, bit-exact against the oracle.  Q6_K (268 M elements) is walked in four pieces."""
    q = pkg.qtypes.Q[name]
    pieces = [(k * 16384, (k + 1) * 16384) for k in range(4)] if name == "Q6_K" else [None]
    for rng_ in pieces:
        blocks = pkg.synth.k_exhaustive_blocks(q, which, seed=5, d_range=rng_)
        got = pkg.dequant.dequantize_tensor(_carrier(pkg, blocks, q), torch.float16)
        g, w = oracle.canon_nan_f16(_bits16(got)), oracle.canon_nan_f16(oracle.dequant_f16(q, blocks))
        if not np.array_equal(g, w):
            bad = np.flatnonzero(g != w)
            raise AssertionError(f"{name} {which} {rng_}: {bad.size} of {g.size} elements differ, first in block {bad[0] // 256} element {bad[0] % 256}")
        del got, g, w


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("mode", ["signed", "adversarial"])
def test_dequant_tensor_all_dtype_combinations(pkg, name, mode):
    """dequantize_tensor(tensor, dtype, dequant_dtype) for every (dequant_dtype, dtype) pair the nodes can
    produce, incl. "target": == dequantize(..., dtype=dequant_dtype).to(dtype) of the oracle, bit for bit,
    across group boundaries and a ragged tail."""
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    G = 64 if bs == 32 else 8
    for n in (1, G + 1, 2 * G, 7 * G + 5):
        blocks = pkg.synth.make_blocks(q, n, seed=1000 + n, mode=mode)
        t = _carrier(pkg, blocks, q)
        for compute in ("f16", "bf16", "f32"):
            for out in ("f16", "bf16", "f32"):
                want = oracle.dequant_tensor(q, blocks, compute, out)
                got = pkg.dequant.dequantize_tensor(t, _TORCH[out], dequant_dtype=None if compute == "f16" else _TORCH[compute])
                assert got.dtype == _TORCH[out] and tuple(got.shape) == (n, bs)
                assert np.array_equal(_canon(_raw(got), out), _canon(want, out)), (name, n, compute, out)
                if compute == out:
                    tgt = pkg.dequant.dequantize_tensor(t, _TORCH[out], dequant_dtype="target")
                    assert np.array_equal(_raw(tgt), _raw(got))


@pytest.mark.parametrize("name", ["Q4_K", "Q6_K", "Q8_0"])
@pytest.mark.parametrize("compute", ["bf16", "f32"])
def test_dequant_dtype_modes_full_size(pkg, name, compute):
    """Shape C (3072x12288) at full size in the bf16 / fp32 arithmetic modes, "target" output."""
    q = pkg.qtypes.Q[name]
    packed = pkg.synth.make_tensor_bytes(q, (3072, 12288), seed=12, mode="signed")
    t = pkg.ops.GGMLTensor(torch.from_numpy(packed).to(DEV), tensor_type=q, tensor_shape=(3072, 12288))
    got = pkg.dequant.dequantize_tensor(t, _TORCH[compute], dequant_dtype="target")
    want = oracle.dequant_tensor(q, packed, compute, compute)
    u = np.uint32 if compute == "f32" else np.uint16
    assert np.array_equal(_raw(got).view(u), want.view(u))                   # signed nominal scales: no NaN, raw bits


@pytest.mark.parametrize("name", ALL)
def test_layer_sized_tensors_all_modes_vs_oracle(pkg, name):
    """Single tensors of 8.4 M ... 33.5 M elements take their own launch shape (teams of 4 waves x 8192 elements, ggq_capi.hip
    TuneMid): one tensor of that size class per format -- not a whole number of groups -- in every (arithmetic -> output) mode the
    shape serves, against the oracle; and the neighbouring size classes (just below / above) stay exact too."""
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    n = 9_437_184 // bs + 3                                       # 3072x3072 plus three blocks: a ragged last group
    blocks = pkg.synth.make_blocks(q, n, seed=77, mode="signed")
    t = _carrier(pkg, blocks, q)
    for compute in ("f16", "bf16", "f32"):
        for out in ("f16", "bf16", "f32"):
            if compute != "f16" and out not in (compute, "bf16"):
                continue                                           # the full 3 x 3 table runs at small sizes above
            want = oracle.dequant_tensor(q, blocks, compute, out)
            got = pkg.dequant.dequantize_tensor(t, _TORCH[out], dequant_dtype=None if compute == "f16" else _TORCH[compute])
            u = np.uint32 if out == "f32" else np.uint16
            assert np.array_equal(_raw(got).view(u), want.view(u)), (name, compute, out)
    for n_el in ((1 << 23), (1 << 23) + 256 * 8, (1 << 25) - 256 * 8, (1 << 25)):                   # the class boundaries
        blocks = pkg.synth.make_blocks(q, n_el // bs, seed=78, mode="signed")
        got = pkg.dequant.dequantize_tensor(_carrier(pkg, blocks, q), torch.bfloat16)
        assert np.array_equal(_raw(got).view(np.uint16), oracle.dequant_tensor(q, blocks, "f16", "bf16").view(np.uint16)), (name, n_el)


def test_randomized_sweep(pkg):
    """500 seeded random (format, block count, scale mode, dequant_dtype, dtype) cases against the oracle -- block counts
    drawn around the group boundaries of both team shapes and at random up to 20000."""
    rng = np.random.default_rng(2024)
    formats = pkg.qtypes.HIP_QTYPES
    kinds = ["f16", "bf16", "f32"]
    for _ in range(500):
        q = formats[rng.integers(len(formats))]
        bs, _ = pkg.qtypes.block_geometry(q)
        n = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4097, int(rng.integers(1, 20000))]))
        mode = ["nominal", "signed", "adversarial", "raw"][rng.integers(4)]
        comp, out = kinds[rng.integers(3)], kinds[rng.integers(3)]
        blocks = pkg.synth.make_blocks(q, n, seed=int(rng.integers(1 << 30)), mode=mode)
        got = pkg.dequant.dequantize_tensor(_carrier(pkg, blocks, q), _TORCH[out], dequant_dtype=None if comp == "f16" else _TORCH[comp])
        want = oracle.dequant_tensor(q, blocks, comp, out)
        assert np.array_equal(_canon(_raw(got), out), _canon(want, out)), (q.name, n, mode, comp, out)


def test_trailing_bytes_and_empty(pkg):
    """n_blocks = numel // type_size (dequant.py:41); an empty tensor is legal."""
    q = pkg.qtypes.Q.Q5_K
    blocks = pkg.synth.make_blocks(q, 5, seed=1)
    ragged = np.concatenate([blocks.reshape(-1), np.arange(33, dtype=np.uint8)])
    out = pkg.dequant.dequantize(torch.from_numpy(ragged).to(DEV), q, (5, 256))
    assert np.array_equal(_bits16(out), oracle.dequant_f16(q, blocks).view(np.uint16))
    empty = pkg.dequant.dequantize(torch.zeros(0, dtype=torch.uint8, device=DEV), q, (0, 256))
    assert tuple(empty.shape) == (0, 256) and empty.dtype == torch.float16


def test_unaligned_views_and_2d_byte_rows(pkg):
    """The loader hands (rows, bytes_per_row) uint8 tensors (loader.py:104-106); views into a larger
    buffer may start at any byte.  Results must not depend on either."""
    q = pkg.qtypes.Q.Q6_K
    blocks = pkg.synth.make_blocks(q, 24, seed=5)
    want = oracle.dequant_f16(q, blocks).view(np.uint16)
    rows = torch.from_numpy(blocks.reshape(2, -1).copy()).to(DEV)             # 2 rows x (12 * 210) bytes
    assert np.array_equal(_bits16(pkg.dequant.dequantize(rows, q, (2, 12 * 256))), want)
    for shift in (1, 2, 7, 16, 33):
        big = torch.zeros(blocks.size + 64, dtype=torch.uint8, device=DEV)
        big[shift:shift + blocks.size] = torch.from_numpy(blocks.reshape(-1).copy()).to(DEV)
        view = big[shift:shift + blocks.size]
        assert np.array_equal(_bits16(pkg.dequant.dequantize(view, q, (24 * 256,))), want), shift
    # int8-typed storage of the same bytes (the reference does .view(torch.uint8), dequant.py:39)
    as_i8 = torch.from_numpy(blocks.reshape(-1).copy()).to(DEV).view(torch.int8)
    assert np.array_equal(_bits16(pkg.dequant.dequantize(as_i8, q, (24 * 256,))), want)


def test_config1_q8_0_4096x4096(pkg, golden_dir):
    """BASELINE.json configs[0]: Q8_0 4096x4096, bit-exact vs the reference (sha256 of its output)."""
    with open(os.path.join(golden_dir, "large_hashes.json")) as f:
        h = json.load(f)["Q8_0:4096x4096:seed0:nominal"]
    q = pkg.qtypes.Q.Q8_0
    packed = pkg.synth.make_tensor_bytes(q, (4096, 4096), seed=0)
    assert hashlib.sha256(packed.tobytes()).hexdigest() == h["packed_sha256"]
    out = pkg.dequant.dequantize(torch.from_numpy(packed).to(DEV), q, (4096, 4096))
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == h["out_f16_sha256"]


@pytest.mark.parametrize("name", ALL)
def test_flux_shape_B_hash(pkg, golden_dir, name):
    """configs[1]/[2] shape B (3072x3072), every format: sha256 of the reference's output."""
    with open(os.path.join(golden_dir, "large_hashes.json")) as f:
        hashes = json.load(f)
    q = pkg.qtypes.Q[name]
    seed = 1 if q in pkg.qtypes.LEGACY_QTYPES else 2
    h = hashes[f"{name}:3072x3072:seed{seed}:nominal"]
    packed = pkg.synth.make_tensor_bytes(q, (3072, 3072), seed=seed)
    out = pkg.dequant.dequantize(torch.from_numpy(packed).to(DEV), q, (3072, 3072))
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == h["out_f16_sha256"]


@pytest.mark.parametrize("name", ["Q4_0", "Q4_K", "Q5_K", "Q6_K"])
def test_flux_shape_C_full_size_vs_oracle(pkg, name):
    """configs[1]/[2] shape C (3072x12288 = 37.7 M elements) at full size, against the oracle."""
    q = pkg.qtypes.Q[name]
    packed = pkg.synth.make_tensor_bytes(q, (3072, 12288), seed=11, mode="signed")
    out = pkg.dequant.dequantize(torch.from_numpy(packed).to(DEV), q, (3072, 12288))
    want = oracle.dequant_f16(q, packed)
    assert np.array_equal(_bits16(out), want.view(np.uint16))


@pytest.mark.parametrize("name", ["Q4_0", "Q5_0", "Q8_0", "Q3_K", "Q6_K", "IQ4_NL", "IQ4_XS"])
def test_property_scale_linearity_full_size(pkg, name):
    """Size-independent property at full BASELINE size: for the min-free formats the result is
    d * integer, so doubling every block scale (fp16 exponent + 1) doubles every output exactly."""
    q = pkg.qtypes.Q[name]
    bs, ts = pkg.qtypes.block_geometry(q)
    n = 3072 * 12288 // bs
    blocks = pkg.synth.make_blocks(q, n, seed=3)
    doubled = blocks.copy()
    (off,) = pkg.qtypes.SCALE_FIELDS[q]
    d = doubled[:, off:off + 2].copy().view(np.uint16)
    d += 0x0400                                                              # exponent + 1, nominal scales are normal
    doubled[:, off:off + 2] = d.view(np.uint8)
    a = pkg.dequant.dequantize(torch.from_numpy(blocks.reshape(-1)).to(DEV), q, (n * bs,))
    b = pkg.dequant.dequantize(torch.from_numpy(doubled.reshape(-1)).to(DEV), q, (n * bs,))
    # nominal d >= 1e-4 and |q| >= 1 keep d*q far from the subnormal range, so x2 is exact both ways
    assert torch.equal(b, a * 2)
    assert torch.equal(pkg.dequant.dequantize(torch.from_numpy(blocks.reshape(-1)).to(DEV), q, (n * bs,)), a)   # deterministic


def test_streaming_store_entry_point_gives_the_same_bits(pkg):
    """ggq_dequant stores plain (its caller is a layer whose GEMM reads the weight next), ggq_dequant_stream non-temporal (results
    nobody reads back soon); dequantize_tensor_streaming is the host name of the latter.  Same kernels otherwise: same bits, every
    team shape (a small tensor, a layer-sized one, one beyond the layer-sized range), fp16 and bf16 results."""
    Q = pkg.qtypes.Q
    for q, shape in ((Q.Q4_K, (24, 512)), (Q.Q8_0, (3072, 3072)), (Q.Q5_K, (3072, 12288)), (Q.Q3_K, (4096, 4096))):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        t = pkg.ops.GGMLTensor(pkg.synth.device_blocks(q, n_blocks, DEV, 5, mode="signed"), tensor_type=q, tensor_shape=shape)
        for dt in (torch.float16, torch.bfloat16):
            a = pkg.dequant.dequantize_tensor(t, dt)
            b = pkg.dequant.dequantize_tensor_streaming(t, dt)
            assert a.dtype == b.dtype == dt and torch.equal(a.view(torch.int16), b.view(torch.int16)), (q, shape, dt)
    assert pkg.dequant._ggq_dequant is not pkg._native.lib().ggq_dequant_stream          # the binding was put back


def test_non_default_stream_ordering(pkg):
    """The launch goes to torch's CURRENT stream: producer copy and consumer read on a side stream."""
    q = pkg.qtypes.Q.Q4_K
    blocks = pkg.synth.make_blocks(q, 4096, seed=2)
    want = oracle.dequant_f16(q, blocks).view(np.uint16)
    host = torch.from_numpy(blocks.reshape(-1).copy()).pin_memory()
    side = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(side):
        for _ in range(5):
            dev = host.to(DEV, non_blocking=True)
            out = pkg.dequant.dequantize(dev, q, (4096 * 256,))
            total = out.float().sum()
    side.synchronize()
    assert np.array_equal(_bits16(out), want)
    assert torch.isfinite(total)


def test_plan_mixed_qtypes_matches_per_tensor(pkg):
    """DequantPlan (ggq_plan_*): one launch per quant type over many tensors == per-tensor calls."""
    Q = pkg.qtypes.Q
    spec = [(Q.Q4_K, (96, 512)), (Q.Q6_K, (7, 256)), (Q.Q4_K, (1, 256)), (Q.Q8_0, (33, 96)), (Q.Q5_K, (64, 1024)),
            (Q.Q4_0, (5, 32)), (Q.Q8_0, (1000, 64)), (Q.Q2_K, (3, 768)), (Q.IQ4_XS, (9, 256))]
    items, wants, total = [], [], 0
    for i, (q, shape) in enumerate(spec):
        packed = pkg.synth.make_tensor_bytes(q, shape, seed=100 + i, mode="signed")
        items.append((torch.from_numpy(packed).to(DEV), q, shape))
        wants.append(oracle.dequant_f16(q, packed).view(np.uint16))
        total += pkg.qtypes.algorithmic_bytes(q, shape[0] * shape[1])
    plan = pkg.grouped.DequantPlan(items)
    assert plan.bytes == total and plan.kernels == len({q for q, _ in spec})
    outs = plan.launch()
    torch.cuda.synchronize()
    for (q, shape), out, want in zip(spec, outs, wants):
        assert tuple(out.shape) == shape and out.dtype == torch.float16
        assert np.array_equal(_bits16(out), want), (q, shape)
    # mixed output dtypes in one plan: the bf16 entries equal the fused-cast path
    plan2 = pkg.grouped.DequantPlan([it + (torch.bfloat16 if i % 2 else torch.float16,) for i, it in enumerate(items)])
    outs2 = plan2.launch()
    torch.cuda.synchronize()
    for i, (out, want) in enumerate(zip(outs2, wants)):
        ref = oracle.cast_f16_to_bf16_bits(want) if i % 2 else want
        assert np.array_equal(_bits16(out), ref)
    # mixed arithmetic modes in one plan (dequant_dtype per tensor)
    kinds = ["f16", "bf16", "f32"]
    plan3 = pkg.grouped.DequantPlan([it + (_TORCH[kinds[(i + 1) % 3]], None if i % 3 == 0 else _TORCH[kinds[i % 3]]) for i, it in enumerate(items)])
    assert plan3.kernels == len(items) - 1          # only the two Q8_0 entries share (qtype, compute, out)
    outs3 = plan3.launch()
    torch.cuda.synchronize()
    for i, ((q, shape), out) in enumerate(zip(spec, outs3)):
        packed = items[i][0].cpu().numpy()
        want = oracle.dequant_tensor(q, packed, kinds[i % 3], kinds[(i + 1) % 3])
        assert np.array_equal(_canon(_raw(out), kinds[(i + 1) % 3]), _canon(want, kinds[(i + 1) % 3])), (q, shape, i)
    plan.close(); plan2.close(); plan3.close()


@pytest.mark.parametrize("name", ALL)
def test_fp32_output_through_plans_ragged_and_large(pkg, name):
    """Round 5: fp32 results decode every chunk ONCE and swap halves inside lane pairs (ggq_device.hpp pair_f32; workgroup teams of 2048 or 4096
    elements depending on format and arithmetic, one-wave teams for Q3_K).  Whole plans -- the dequant_many kernels, groups that straddle a tensor's
    end, a partner lane whose chunk lies past it, launches big enough for the XCD run mapping -- against the oracle, fp16 and fp32 arithmetic."""
    q = pkg.qtypes.Q[name]
    bs, ts = pkg.qtypes.block_geometry(q)
    shapes = [(1, 256), (3, 256), (7, 768), (8, 256), (9, 256), (17, 512), (33, 256), (1024, 3072), (5, 1280)]
    items, packed = [], []
    for i, sh in enumerate(shapes):
        p = pkg.synth.make_tensor_bytes(q, sh, seed=7700 + i, mode="signed")
        packed.append(p)
        items.append((torch.from_numpy(p).to(DEV), q, sh))
    for compute in ("f16", "f32"):
        plan = pkg.grouped.DequantPlan(items, out_dtype=torch.float32, dequant_dtype=None if compute == "f16" else torch.float32)
        assert plan.kernels == 1
        outs = plan.launch()
        torch.cuda.synchronize()
        for sh, p, out in zip(shapes, packed, outs):
            assert out.dtype == torch.float32 and tuple(out.shape) == sh
            if compute == "f16":
                want = oracle.dequant_f16(q, p, simd=oracle.simd_available()).astype(np.float32)
            else:
                want = oracle.dequant_f32(q, p)
            assert np.array_equal(_canon(_raw(out), "f32"), _canon(want, "f32")), (name, compute, sh)
        plan.close()
    # the per-tensor entry point on the big one (another team shape may be picked for a single tensor of this size): the same bits
    one = pkg.dequant.dequantize_tensor(_carrier(pkg, packed[7].reshape(-1, ts), q), torch.float32)          # fp16 arithmetic, single-tensor launch
    want = oracle.dequant_f16(q, packed[7], simd=oracle.simd_available()).astype(np.float32)
    assert np.array_equal(_canon(_raw(one), "f32"), _canon(want, "f32")), name


def test_large_plan_uses_the_xcd_run_mapping_and_stays_exact(pkg):
    """Launches of >= 65536 groups switch to the XCD-aware workgroup -> group mapping (a permutation of which
    workgroup does which group, ggq_capi.hip Tune<>): every tensor of a 0.9 G-element two-format plan is
    checked in full against the oracle's SIMD leg and in windows against the soft-float checker."""
    Q = pkg.qtypes.Q
    items, packed = [], []
    for i in range(24):
        q = Q.Q4_K if i % 2 == 0 else Q.Q8_0
        p = pkg.synth.make_tensor_bytes(q, (3072, 12288), seed=300 + i, mode="signed")
        packed.append((q, p))
        items.append((torch.from_numpy(p).to(DEV), q, (3072, 12288)))
    plan = pkg.grouped.DequantPlan(items)
    assert plan.kernels == 2
    outs = plan.launch()
    torch.cuda.synchronize()
    for (q, p), out in zip(packed, outs):
        got = _bits16(out)
        if oracle.simd_available():
            assert np.array_equal(got, oracle.dequant_f16(q, p, simd=True).view(np.uint16)), q
        bs, ts = pkg.qtypes.block_geometry(q)
        n_blocks = p.size // ts
        for b0 in (0, n_blocks // 3, n_blocks - 4096):
            want = oracle.dequant_f16(q, p[b0 * ts:(b0 + 4096) * ts]).view(np.uint16)
            assert np.array_equal(got[b0 * bs:(b0 + 4096) * bs], want), (q, b0)
    plan.close()


def test_ggml_linear_forward_is_the_reference_call_chain(pkg):
    """GGMLOps.Linear's hot loop (ops.py:242-244): dequantize the weight on every forward, then F.linear."""
    Q = pkg.qtypes.Q
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        blocks = pkg.synth.make_blocks(Q.Q5_K, 48 * 3, seed=9)                # weight 48 x 768
        w = _carrier(pkg, blocks, Q.Q5_K, (48, 768))
        bias_blocks = pkg.synth.make_blocks(Q.Q8_0, 2, seed=10)              # quantized bias, 64 elements... 48 used? keep 64
        lin = pkg.ops.GGMLLinear(w)
        x = torch.randn(5, 768, device=DEV, dtype=dtype)
        wref = torch.from_numpy(oracle.dequant_f16(Q.Q5_K, blocks).reshape(48, 768).copy()).to(DEV).to(dtype)
        assert torch.equal(lin(x), torch.nn.functional.linear(x, wref))
        assert torch.equal(lin(x), lin(x))


@pytest.mark.parametrize("name,out", [("Q4_K", "f16"), ("Q8_0", "f32"), ("Q6_K", "bf16")])
def test_outputs_beyond_4_gib(pkg, name, out):
    """Maximum sizes: a dense result larger than 2^32 bytes (and packed offsets beyond 2^31) -- every
    address computation must be 64-bit.  Windows around the 2^31 / 2^32 byte marks of the output, the head,
    the ragged tail and random places are checked against the oracle."""
    q = pkg.qtypes.Q[name]
    bs, ts = pkg.qtypes.block_geometry(q)
    itemsize = 4 if out == "f32" else 2
    n_el = (1 << 32) // itemsize + 2048 * 37 + bs * 3                        # > 4 GiB of output, ragged last group
    n_blocks = n_el // bs
    g = torch.Generator(device=DEV)
    g.manual_seed(123)
    data = torch.randint(0, 256, (n_blocks * ts,), dtype=torch.uint8, device=DEV, generator=g)
    t = pkg.ops.GGMLTensor(data, tensor_type=q, tensor_shape=(n_blocks, bs))
    got = pkg.dequant.dequantize_tensor(t, _TORCH[out])
    assert got.numel() == n_blocks * bs and got.numel() * itemsize > (1 << 32)
    win = 192 if bs == 32 else 24                                            # blocks per window (3 groups)
    marks = [0, n_blocks - win, (1 << 31) // itemsize // bs - win // 2, (1 << 32) // itemsize // bs - win // 2,
             (1 << 30) // ts - win // 2, (1 << 31) // ts - win // 2]
    rng = np.random.default_rng(5)
    marks += [int(x) for x in rng.integers(0, n_blocks - win, size=6)]
    for b0 in marks:
        b0 = max(0, min(int(b0), n_blocks - win))
        packed = data[b0 * ts:(b0 + win) * ts].cpu().numpy()
        want = oracle.dequant_tensor(q, packed, "f16", out)
        piece = got.reshape(-1)[b0 * bs:(b0 + win) * bs]
        assert np.array_equal(_canon(_raw(piece), out), _canon(want, out)), (name, out, b0)
    del got, data
    torch.cuda.empty_cache()


def test_reentrant_from_threads_and_never_synchronises(pkg):
    """SURVEY.md section 8b "Threading": re-entrant (no locks, no mutable globals), enqueues on the calling
    thread's current stream, and returns without waiting for the device."""
    import threading
    Q = pkg.qtypes.Q
    # (1) the call returns while earlier work on the stream is still running
    big = pkg.synth.make_blocks(Q.Q4_K, 3072 * 12288 // 256, seed=70)
    dbig = torch.from_numpy(big.reshape(-1).copy()).to(DEV)
    small = pkg.synth.make_blocks(Q.Q4_K, 16, seed=71)
    dsmall = torch.from_numpy(small.reshape(-1).copy()).to(DEV)
    torch.cuda.synchronize()
    for _ in range(300):                                                     # ~300 x 17 us of queued GPU work, enqueued in ~half that
        pkg.dequant.dequantize(dbig, Q.Q4_K, (3072, 12288))
    ev = torch.cuda.Event()
    ev.record()
    out = pkg.dequant.dequantize(dsmall, Q.Q4_K, (16, 256))
    assert not ev.query(), "dequantize() waited for the device"
    torch.cuda.synchronize()
    assert np.array_equal(_bits16(out), oracle.dequant_f16(Q.Q4_K, small).view(np.uint16))

    # (2) four host threads, each on its own stream, different formats, interleaved calls
    errors = []

    def worker(i, q):
        try:
            bs, _ = pkg.qtypes.block_geometry(q)
            stream = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(stream):
                for rep in range(25):
                    n = 200 + 37 * rep + i
                    blocks = pkg.synth.make_blocks(q, n, seed=1000 * i + rep, mode="signed")
                    got = pkg.dequant.dequantize(torch.from_numpy(blocks.reshape(-1).copy()).to(DEV, non_blocking=True), q, (n, bs))
                    stream.synchronize()
                    if not np.array_equal(_bits16(got), oracle.dequant_f16(q, blocks).view(np.uint16)):
                        errors.append((q.name, rep))
        except Exception as e:                                               # noqa: BLE001 -- surfaced below
            errors.append((q.name, repr(e)))

    threads = [threading.Thread(target=worker, args=(i, q)) for i, q in enumerate((Q.Q4_K, Q.Q8_0, Q.Q6_K, Q.IQ4_XS))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_capturable_in_a_hip_graph(pkg):
    """The launch only enqueues on torch's current stream and never synchronises, so dequant + F.linear can be
    captured once (torch.cuda.graph -> hipGraph) and replayed -- the launch-bound per-layer loop as one graph."""
    Q = pkg.qtypes.Q
    layers = []
    for i, q in enumerate((Q.Q4_K, Q.Q8_0, Q.Q6_K, Q.Q5_0)):
        blocks = pkg.synth.make_blocks(q, 64 * 512 // pkg.qtypes.block_geometry(q)[0], seed=60 + i)
        layers.append((pkg.ops.GGMLLinear(_carrier(pkg, blocks, q, (64, 512))), q, blocks))
    x = torch.randn(8, 512, device=DEV, dtype=torch.float16)
    for lin, _, _ in layers:                                                 # warm-up outside the capture
        lin(x)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ys = [lin(x) for lin, _, _ in layers]
    x.copy_(torch.randn(8, 512, device=DEV, dtype=torch.float16))            # new input, same graph
    graph.replay()
    torch.cuda.synchronize()
    for y, (lin, q, blocks) in zip(ys, layers):
        w = torch.from_numpy(oracle.dequant_f16(q, blocks).reshape(64, 512).copy()).to(DEV)
        assert torch.equal(y, torch.nn.functional.linear(x, w)), q


def test_torch_compile_traces_through_the_custom_op(pkg):
    """Under torch.compile (the reference allows full compile on torch >= 2.8, ops.py:20-42) the launch is
    the opaque custom op ggq::dequantize: one graph, no break, same bits as eager."""
    dq, Q = pkg.dequant, pkg.qtypes.Q
    if dq._dequantize_op is None:
        pytest.skip("torch.library.custom_op not available")
    blocks = pkg.synth.make_blocks(Q.Q4_K, 64 * 2, seed=31)                  # weight 64 x 512
    data = torch.from_numpy(blocks.reshape(-1).copy()).to(DEV)
    x = torch.randn(3, 512, device=DEV, dtype=torch.bfloat16)

    def f(data, x):
        w = dq.dequantize(data, Q.Q4_K, (64, 512)).to(x.dtype)
        return torch.nn.functional.linear(x, w)

    want = f(data, x)
    got = torch.compile(f, backend="aot_eager", fullgraph=True)(data, x)
    assert torch.equal(got, want)
    assert np.array_equal(_bits16(dq.dequantize(data, Q.Q4_K, (64, 512))), oracle.dequant_f16(Q.Q4_K, blocks).view(np.uint16))
    torch.library.opcheck(torch.ops.ggq.dequantize.default, (data, int(Q.Q4_K), 0, 1))


def test_other_layer_types_call_the_same_path(pkg):
    """GGMLOps.Embedding / Conv2d / LayerNorm / GroupNorm (ops.py:246-271): cast_bias_weight -> dequantize_tensor -> the
    functional op, against the same op on the oracle's weights."""
    Q, ops, F = pkg.qtypes.Q, pkg.ops, torch.nn.functional
    # token embedding, Q6_K table 512 x 256 (the T5 / llama loaders keep big tables quantized, loader.py:377-397)
    tb = pkg.synth.make_blocks(Q.Q6_K, 512, seed=81)
    table = torch.from_numpy(oracle.dequant_f16(Q.Q6_K, tb).reshape(512, 256).copy()).to(DEV)
    emb = ops.GGMLEmbedding(_carrier(pkg, tb, Q.Q6_K, (512, 256)))
    idx = torch.randint(0, 512, (3, 17), device=DEV)
    got = emb(idx, out_dtype=torch.float16)
    assert got.dtype == torch.float16 and torch.equal(got, F.embedding(idx, table))
    assert torch.equal(emb(idx, out_dtype=torch.float32), F.embedding(idx, table.float()))
    # conv with a 4-D logical shape stored as Q8_0 blocks, quantized bias
    wb = pkg.synth.make_blocks(Q.Q8_0, 9, seed=82)                           # 8*4*3*3 = 288 elements = 9 blocks
    w = torch.from_numpy(oracle.dequant_f16(Q.Q8_0, wb).reshape(8, 4, 3, 3).copy()).to(DEV)
    conv = ops.GGMLConv2d(_carrier(pkg, wb, Q.Q8_0, (8, 4, 3, 3)), None, padding=1)
    x = torch.randn(2, 4, 9, 9, device=DEV, dtype=torch.float16)
    assert torch.equal(conv(x), F.conv2d(x, w, None, 1, 1))
    # norms keep their (unquantized) F32 weights: the passthrough branch of dequantize_tensor (dequant.py:19-20)
    g = ops.GGMLTensor(torch.linspace(0.5, 1.5, 64, device=DEV), tensor_type=Q.F32, tensor_shape=(64,))
    b = ops.GGMLTensor(torch.linspace(-1, 1, 64, device=DEV), tensor_type=Q.F32, tensor_shape=(64,))
    xs = torch.randn(5, 64, device=DEV, dtype=torch.bfloat16)
    ln = ops.GGMLLayerNorm((64,), g, b)
    assert torch.equal(ln(xs), F.layer_norm(xs, (64,), torch.Tensor(g).to(torch.bfloat16), torch.Tensor(b).to(torch.bfloat16), 1e-5))
    gn = ops.GGMLGroupNorm(8, g, b)
    xg = torch.randn(2, 64, 4, 4, device=DEV, dtype=torch.float16)
    assert torch.equal(gn(xg), F.group_norm(xg, 8, torch.Tensor(g).half(), torch.Tensor(b).half(), 1e-5))


def test_resident_dense_cache_on_the_real_path(pkg):
    """resident.DenseCache around the real dequantize_tensor: the resident tensor IS the kernel's output (bit-exact), is
    reused across forwards of GGMLLinear, and a patched (LoRA) tensor never gets a shared one."""
    Q = pkg.qtypes.Q
    blocks = pkg.synth.make_blocks(Q.Q4_K, 64 * 2, seed=91)
    w = _carrier(pkg, blocks, Q.Q4_K, (64, 512))
    cache = pkg.resident.DenseCache(1e9, pkg.dequant.dequantize_tensor)
    d1 = cache(w, torch.bfloat16)
    assert cache(w, torch.bfloat16) is d1 and cache.stats()["hits"] == 1
    assert np.array_equal(_bits16(d1), oracle.dequant_tensor(Q.Q4_K, blocks, "f16", "bf16"))
    assert np.array_equal(_bits16(cache(w, torch.float16)), oracle.dequant_f16(Q.Q4_K, blocks).view(np.uint16))
    lin = pkg.ops.GGMLLinear(w)
    lin._dequantize = cache                                                   # instance-level: this layer only
    x = torch.randn(4, 512, device=DEV, dtype=torch.bfloat16)
    y = lin(x)
    assert torch.equal(y, torch.nn.functional.linear(x, d1)) and torch.equal(lin(x), y)
    assert cache.stats()["misses"] == 2                                       # bf16 and fp16 results, each computed once
    patched = pkg.ops.GGMLTensor(torch.from_numpy(blocks.reshape(-1).copy()).to(DEV), tensor_type=Q.Q4_K, tensor_shape=(64, 512), patches=[("p", "k")])
    assert cache(patched, torch.bfloat16) is not cache(patched, torch.bfloat16)
    with pytest.raises(pkg.dequant.GGQUnsupported):                           # unsupported requests pass straight through
        cache(pkg.ops.GGMLTensor(torch.zeros(66, dtype=torch.uint8, device=DEV), tensor_type=Q.IQ2_XXS, tensor_shape=(256,)), torch.float16)


def test_unsupported_requests_raise(pkg):
    dq, Q = pkg.dequant, pkg.qtypes.Q
    data = torch.zeros(144 * 4, dtype=torch.uint8, device=DEV)
    with pytest.raises(dq.GGQUnsupported):
        dq.dequantize(data, Q.Q4_K, (4, 256), dtype=torch.float64)           # not an arithmetic mode of the kernels
    with pytest.raises(dq.GGQUnsupported):
        dq.dequantize(data, Q.IQ2_XXS, (4, 256))
    with pytest.raises(dq.GGQUnsupported):
        dq.dequantize(data.cpu(), Q.Q4_K, (4, 256))


def test_c_abi_direct_call(pkg):
    """Straight through ctypes, as INTEGRATION.md's reference-side stub does."""
    nat, Q = pkg._native, pkg.qtypes.Q
    lib = nat.lib()
    blocks = pkg.synth.make_blocks(Q.Q4_1, 777, seed=4)
    data = torch.from_numpy(blocks.reshape(-1).copy()).to(DEV)
    out = torch.empty(777 * 32, dtype=torch.float16, device=DEV)
    rc = lib.ggq_dequant_f16(int(Q.Q4_1), data.data_ptr(), 777, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == nat.GGQ_OK and lib.ggq_last_hip_error() == 0
    torch.cuda.synchronize()
    assert np.array_equal(_bits16(out), oracle.dequant_f16(Q.Q4_1, blocks).view(np.uint16))
