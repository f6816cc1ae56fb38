"""The CPU oracle (oracle/ggq_oracle.c) against the golden vectors produced by the reference's own
dequant.py (oracle/make_golden.py), and -- where /root/reference is present -- against the
reference executed live.  Bit-exact everywhere; NaN payloads are canonicalised (the only freedom
IEEE leaves)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from oracle import reference

ALL = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "IQ4_NL", "IQ4_XS"]


def _canon_bf16(bits):
    bits = np.asarray(bits, dtype=np.uint16).copy()
    bits[(bits & 0x7FFF) > 0x7F80] = 0x7FC0
    return bits


def _canon_f32(bits):
    bits = np.asarray(bits, dtype=np.uint32).copy()
    bits[(bits & 0x7FFFFFFF) > 0x7F800000] = 0x7FC00000
    return bits


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_golden_f16(pkg, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    q = pkg.qtypes.Q[name]
    out = oracle.dequant_f16(q, g["blocks"])
    want = g["out_f16"].reshape(-1)
    assert out.size == want.size
    assert np.array_equal(oracle.canon_nan_f16(out), oracle.canon_nan_f16(want))
    # the nominal and signed runs carry no NaN at all, so there the raw bits must agree
    half = want.size // 2
    assert np.array_equal(out.view(np.uint16)[:half], want[:half])


@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_golden_f32_and_bf16_modes(pkg, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"{name}.npz"))
    q = pkg.qtypes.Q[name]
    sub = g["blocks"][g["sub"]]
    o32 = oracle.dequant_f32(q, sub).view(np.uint32)
    assert np.array_equal(_canon_f32(o32), _canon_f32(g["out_f32"].reshape(-1)))
    obf = oracle.dequant_bf16_bits(q, sub)
    assert np.array_equal(_canon_bf16(obf), _canon_bf16(g["out_bf16"].reshape(-1)))
    # dequantize_tensor(dtype=bf16) = fp16 dequant then one cast (dequant.py:23)
    tb = oracle.cast_f16_to_bf16_bits(oracle.dequant_f16(q, sub))
    assert np.array_equal(_canon_bf16(tb), _canon_bf16(g["tensor_bf16"].reshape(-1)))


def test_oracle_bf16_passthrough(golden_dir):
    g = np.load(os.path.join(golden_dir, "BF16.npz"))
    assert np.array_equal(oracle.bf16_to_f32(g["blocks"]).view(np.uint32), g["out_f32"].reshape(-1))


def test_config1_q8_0_4096x4096_hash(pkg, golden_dir):
    """BASELINE.json configs[0]: Q8_0 single 4096x4096 tensor, bit-exact vs the reference (by hash)."""
    with open(os.path.join(golden_dir, "large_hashes.json")) as f:
        h = json.load(f)["Q8_0:4096x4096:seed0:nominal"]
    packed = pkg.synth.make_tensor_bytes(pkg.qtypes.Q.Q8_0, (4096, 4096), seed=0, mode="nominal")
    assert hashlib.sha256(packed.tobytes()).hexdigest() == h["packed_sha256"]
    out = oracle.dequant_f16(pkg.qtypes.Q.Q8_0, packed)
    assert hashlib.sha256(out.tobytes()).hexdigest() == h["out_f16_sha256"]


@pytest.mark.parametrize("name", ["Q4_0", "Q4_K", "Q6_K"])
def test_flux_shape_hash(pkg, golden_dir, name):
    with open(os.path.join(golden_dir, "large_hashes.json")) as f:
        hashes = json.load(f)
    q = pkg.qtypes.Q[name]
    seed = 1 if q in pkg.qtypes.LEGACY_QTYPES else 2
    h = hashes[f"{name}:3072x3072:seed{seed}:nominal"]
    packed = pkg.synth.make_tensor_bytes(q, (3072, 3072), seed=seed, mode="nominal")
    assert hashlib.sha256(packed.tobytes()).hexdigest() == h["packed_sha256"]
    assert hashlib.sha256(oracle.dequant_f16(q, packed).tobytes()).hexdigest() == h["out_f16_sha256"]


def test_soft_fp16_against_numpy():
    """d2h is a correctly rounded double->fp16; hmul/hadd/hsub are the IEEE fp16 ops."""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    xs = np.concatenate([
        rng.standard_normal(20000) * 10.0 ** rng.integers(-9, 6, 20000),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 3 * 2.0 ** -25,
                  np.inf, -np.inf, 1.0009765625, 1.00048828125, 1.000488281250001]),
    ])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.ggq_oracle_d2h(float(x)) for x in xs], dtype=np.uint16)
    assert np.array_equal(got, want)
    assert L.ggq_oracle_d2h(float("nan")) & 0x7FFF > 0x7C00
    a = rng.integers(0, 1 << 16, 20000, dtype=np.uint16)
    b = rng.integers(0, 1 << 16, 20000, dtype=np.uint16)
    af, bf = a.view(np.float16).astype(np.float64), b.view(np.float16).astype(np.float64)
    with np.errstate(all="ignore"):
        for fn, ref in ((L.ggq_oracle_hmul, af * bf), (L.ggq_oracle_hadd, af + bf), (L.ggq_oracle_hsub, af - bf)):
            want = ref.astype(np.float16)
            got = np.array([fn(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint16)
            assert np.array_equal(oracle.canon_nan_f16(got), oracle.canon_nan_f16(want))


def test_framing_ignores_trailing_bytes(pkg):
    """n_blocks = numel // type_size (dequant.py:41): a ragged tail is dropped, empty input is legal."""
    q = pkg.qtypes.Q.Q4_K
    blocks = pkg.synth.make_blocks(q, 3, seed=9)
    ragged = np.concatenate([blocks.reshape(-1), np.zeros(17, np.uint8)])
    assert np.array_equal(oracle.dequant_f16(q, ragged).view(np.uint16), oracle.dequant_f16(q, blocks).view(np.uint16))
    assert oracle.dequant_f16(q, np.zeros(0, np.uint8)).size == 0


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("mode", ["nominal", "adversarial"])
def test_oracle_matches_live_reference(pkg, name, mode):
    import torch
    ref = reference.load_reference_dequant()
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    blocks = pkg.synth.make_blocks(q, 96, seed=4242, mode=mode)
    want = ref.dequantize(torch.from_numpy(blocks.reshape(-1).copy()), q, (96 * bs,)).numpy()
    assert np.array_equal(oracle.canon_nan_f16(oracle.dequant_f16(q, blocks)), oracle.canon_nan_f16(want))


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("name", ALL)
def test_oracle_matches_live_reference_on_every_scale_bit_pattern(pkg, name):
    """Pins the oracle on the WHOLE domain of the scale operands: all 65 536 bit patterns of every fp16 scale field of every format (random quants,
    randomly signed other fields), the reference's own dequantize() on torch-CPU == the C oracle, in all three arithmetic modes.  (The GPU tests
    hold the HIP path to the oracle on exactly these inputs: tests/test_gpu_parity.py::test_every_fp16_bit_pattern_of_every_scale_field.)"""
    import torch
    ref = reference.load_reference_dequant()
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    pats = np.arange(65536, dtype=np.uint32)
    for k, off in enumerate(pkg.qtypes.SCALE_FIELDS[q]):
        blocks = pkg.synth.make_blocks(q, 65536, seed=4242 + k, mode="signed")       # the very blocks of the GPU test
        blocks[:, off] = (pats & 0xFF).astype(np.uint8)
        blocks[:, off + 1] = (pats >> 8).astype(np.uint8)
        data = torch.from_numpy(blocks.reshape(-1).copy())
        want = ref.dequantize(data, q, (65536 * bs,)).numpy()
        assert np.array_equal(oracle.canon_nan_f16(oracle.dequant_f16(q, blocks)), oracle.canon_nan_f16(want)), (name, off)
        want32 = ref.dequantize(data, q, (65536 * bs,), dtype=torch.float32).numpy()
        assert np.array_equal(_canon_f32(oracle.dequant_f32(q, blocks).view(np.uint32)), _canon_f32(want32.view(np.uint32))), (name, off, "f32")
        wantbf = ref.dequantize(data, q, (65536 * bs,), dtype=torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(_canon_bf16(oracle.dequant_bf16_bits(q, blocks)), _canon_bf16(wantbf)), (name, off, "bf16")


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
def test_oracle_matches_live_reference_on_every_q8_0_scale_quant_pair(pkg):
    """Q8_0, the format the north star wants bit-exact: every one of the 65 536 x 256 (scale bits, quant) pairs, reference torch-CPU == oracle."""
    import torch
    ref = reference.load_reference_dequant()
    q = pkg.qtypes.Q.Q8_0
    blocks = np.zeros((65536 * 8, 34), dtype=np.uint8)
    pats = np.repeat(np.arange(65536, dtype=np.uint32), 8)
    blocks[:, 0] = (pats & 0xFF).astype(np.uint8)
    blocks[:, 1] = (pats >> 8).astype(np.uint8)
    blocks[:, 2:] = (np.arange(32, dtype=np.uint32)[None, :] + 32 * (np.arange(65536 * 8, dtype=np.uint32) % 8)[:, None]).astype(np.uint8)
    want = ref.dequantize(torch.from_numpy(blocks.reshape(-1).copy()), q, (65536 * 8 * 32,)).numpy()
    assert np.array_equal(oracle.canon_nan_f16(oracle.dequant_f16(q, blocks)), oracle.canon_nan_f16(want))


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("which", ["d", "dmin"])
@pytest.mark.parametrize("name", ["Q4_K", "Q5_K"])
def test_oracle_matches_live_reference_on_every_q4_k_product(pkg, name, which):
    """Q4_K / Q5_K over the whole domain of each of their two products (synth.k_scmn_exhaustive_blocks: 65 536 scale patterns x 64 sub-block factors x 16 quants,
    134 M elements): the reference's dequantize() on torch-CPU == the C oracle.  The GPU test holds the HIP path to the oracle on the same blocks."""
    import torch
    ref = reference.load_reference_dequant()
    q = pkg.qtypes.Q[name]
    blocks = pkg.synth.k_scmn_exhaustive_blocks(q, which, seed=5)
    want = ref.dequantize(torch.from_numpy(blocks.reshape(-1)), q, (blocks.shape[0] * 256,)).numpy()
    assert np.array_equal(oracle.canon_nan_f16(oracle.dequant_f16(q, blocks)), oracle.canon_nan_f16(want))


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("name,which", [("Q2_K", "d"), ("Q2_K", "dmin"), ("Q3_K", "d"), ("IQ4_XS", "d"), ("Q6_K", "d")])
def test_oracle_matches_live_reference_on_every_scale_product_other_k_formats(pkg, name, which):
    """synth.k_exhaustive_blocks (every scale bit pattern x every integer sub-block factor x the quants): the reference's dequantize() on torch-CPU == the
    C oracle for the remaining super-block formats.  Q6_K (256 int8 scales per pattern: 268 M elements) is walked in eight pieces to bound memory."""
    import torch
    ref = reference.load_reference_dequant()
    q = pkg.qtypes.Q[name]
    pieces = [(k * 8192, (k + 1) * 8192) for k in range(8)] if name == "Q6_K" else [None]
    for rng_ in pieces:
        blocks = pkg.synth.k_exhaustive_blocks(q, which, seed=5, d_range=rng_)
        want = ref.dequantize(torch.from_numpy(blocks.reshape(-1)), q, (blocks.shape[0] * 256,)).numpy()
        assert np.array_equal(oracle.canon_nan_f16(oracle.dequant_f16(q, blocks)), oracle.canon_nan_f16(want)), (name, which, rng_)
        del want


@pytest.mark.parametrize("mode", ["nominal", "signed", "adversarial", "raw"])
def test_simd_throughput_leg_equals_the_soft_float_checker(pkg, golden_dir, mode):
    """oracle/ggq_oracle_simd.c (bench.py's cpu_baseline leg) == oracle/ggq_oracle.c, bit for bit, and hence
    == the reference's golden vectors."""
    if not oracle.simd_available():
        pytest.skip("host has no AVX2+F16C")
    for q in pkg.qtypes.HIP_QTYPES:
        bs, _ = pkg.qtypes.block_geometry(q)
        blocks = pkg.synth.make_blocks(q, 4099 if bs == 32 else 515, seed=77, mode=mode)
        soft = oracle.dequant_f16(q, blocks)
        for threads in (1, 3):
            fast = oracle.dequant_f16(q, blocks, threads=threads, simd=True)
            assert np.array_equal(oracle.canon_nan_f16(fast), oracle.canon_nan_f16(soft)), (q.name, mode, threads)
        g = np.load(os.path.join(golden_dir, f"{q.name}.npz"))
        fast = oracle.dequant_f16(q, g["blocks"], simd=True)
        assert np.array_equal(oracle.canon_nan_f16(fast), oracle.canon_nan_f16(g["out_f16"].reshape(-1))), q.name
    assert oracle.dequant_f16(pkg.qtypes.Q.Q4_K, np.zeros(0, np.uint8), simd=True).size == 0
    with pytest.raises(ValueError):
        oracle.dequant_f16(99, np.zeros(144, np.uint8), simd=True)


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
def test_committed_fixtures_are_what_the_generator_writes(golden_dir, tmp_path, capsys):
    """tests/golden/*.npz regenerated by oracle/make_golden.py (the reference's own dequant.py, verbatim) into a scratch
    directory: every array of every fixture identical to the committed one -- the fixtures are the script's output, nothing else."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ggq_make_golden", os.path.join(os.path.dirname(os.path.abspath(oracle.__file__)), "make_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.main(out_dir=str(tmp_path), large=False)
    capsys.readouterr()
    names = sorted(f for f in os.listdir(golden_dir) if f.endswith(".npz"))
    assert names == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) and len(names) == 13
    for f in names:
        a, b = np.load(os.path.join(golden_dir, f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), (f, k)
