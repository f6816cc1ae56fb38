"""Host-side logic that needs no GPU: tables, synthetic blocks, the C-ABI library's exports and
argument checking, the mirrored dequant interface's dispatch / error behaviour, sharding,
manifests, and install() in front of the real reference modules (when /root/reference exists)."""
import ctypes
import importlib.util
import os
import re
import sys
import types

import numpy as np
import pytest
import torch

import oracle
from oracle import reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_qtype_tables_agree_with_oracle(pkg):
    qt = pkg.qtypes
    for q in qt.HIP_QTYPES + (qt.Q.BF16,):
        assert oracle.geometry(q) == qt.block_geometry(q), q
    # algorithmic bytes of BASELINE.md section 3
    assert qt.algorithmic_bytes(qt.Q.Q4_K, 3072 * 3072) == 24_182_784
    assert qt.algorithmic_bytes(qt.Q.Q6_K, 3072 * 12288) == 106_463_232
    assert qt.algorithmic_bytes(qt.Q.Q8_0, 4096 * 4096) == 51_380_224


def test_synth_is_deterministic_and_sanitised(pkg):
    q = pkg.qtypes.Q.Q4_K
    a = pkg.synth.make_blocks(q, 100, seed=3)
    b = pkg.synth.make_blocks(q, 100, seed=3)
    assert np.array_equal(a, b) and a.shape == (100, 144)
    d = a[:, 0:2].copy().view(np.float16).reshape(-1)
    assert np.all((d >= 1e-4 * 0.99) & (d <= 2e-3 * 1.01))
    assert not np.array_equal(a, pkg.synth.make_blocks(q, 100, seed=4))
    assert pkg.synth.make_blocks(q, 0).shape == (0, 144)
    with pytest.raises(ValueError):
        pkg.synth.n_blocks_for(q, 100)


def _declared_symbols():
    names = set()
    for header in ("ggq.h", "ggq_gguf.h"):
        with open(os.path.join(ROOT, "include", header)) as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        names |= set(re.findall(r"\b(ggq_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol(pkg):
    """The C-ABI library loads and exports everything include/ggq.h declares (no compute calls)."""
    nat = pkg._native
    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(nat.LIB_PATH)
    declared = _declared_symbols()
    assert set(declared) == set(nat.SYMBOLS), (declared, sorted(nat.SYMBOLS))
    for name in declared:
        getattr(L, name)
    lib = nat.lib()
    assert lib.ggq_abi_version() == nat.ABI_VERSION


def test_c_abi_queries_and_argument_checks(pkg):
    """Pure host-side entry points and the argument validation that runs before any launch."""
    nat, qt = pkg._native, pkg.qtypes
    lib = nat.lib()
    for q in qt.HIP_QTYPES:
        assert lib.ggq_supported(int(q)) == 1
        assert (lib.ggq_block_size(int(q)), lib.ggq_type_size(int(q))) == qt.block_geometry(q)
    for bad in (0, 1, 9, 15, 16, 30, 99, -1):      # F32, F16, Q8_1, Q8_K, IQ2_XXS, BF16, junk
        assert lib.ggq_supported(bad) == 0 and lib.ggq_block_size(bad) == 0 and lib.ggq_type_size(bad) == 0
    assert lib.ggq_strerror(0) == b"ok" and b"aligned" in lib.ggq_strerror(nat.GGQ_ERR_ALIGN)
    q4k = int(qt.Q.Q4_K)
    assert lib.ggq_dequant(99, 16, 1, 16, 0, 0, None) == nat.GGQ_ERR_QTYPE
    assert lib.ggq_dequant(q4k, 16, 1, 16, 0, 7, None) == nat.GGQ_ERR_ARG       # bad out dtype
    assert lib.ggq_dequant(q4k, 16, 1, 16, 3, 0, None) == nat.GGQ_ERR_ARG       # bad compute dtype
    assert lib.ggq_dequant(q4k, None, 1, 16, 0, 0, None) == nat.GGQ_ERR_ARG      # NULL packed
    assert lib.ggq_dequant(q4k, 24, 1, 32, 0, 0, None) == nat.GGQ_ERR_ALIGN      # packed % 16 != 0
    assert lib.ggq_dequant(q4k, 32, 1, 8, 2, 1, None) == nat.GGQ_ERR_ALIGN       # out % 16 != 0
    assert lib.ggq_dequant(q4k, None, 0, None, 1, 2, None) == nat.GGQ_OK         # empty tensor: no-op
    assert lib.ggq_dequant_f16(99, 16, 1, 16, None) == nat.GGQ_ERR_QTYPE
    plan = ctypes.c_void_p()
    bad = (nat.ggq_desc * 1)(nat.ggq_desc(99, 0, 16, 16, 1, 0, 0))
    assert lib.ggq_plan_create(bad, 1, ctypes.byref(plan)) == nat.GGQ_ERR_QTYPE and not plan.value
    bad = (nat.ggq_desc * 1)(nat.ggq_desc(q4k, 0, 16, 16, 1, 5, 0))
    assert lib.ggq_plan_create(bad, 1, ctypes.byref(plan)) == nat.GGQ_ERR_ARG and not plan.value
    assert lib.ggq_plan_launch(None, None) == nat.GGQ_ERR_ARG
    assert lib.ggq_plan_bytes(None) == 0
    with pytest.raises(nat.GGQNativeError, match="aligned"):
        nat.check(nat.GGQ_ERR_ALIGN, "x")


def test_missing_extension_fails_loudly(pkg, monkeypatch):
    nat = pkg._native
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", os.path.join(ROOT, "does", "not", "exist.so"))
    with pytest.raises(nat.GGQNativeError, match="no CPU or torch fallback"):
        nat.lib()


def test_fma_guard_regex(pkg):
    nat = pkg._native
    nat.check_no_fma("v_pk_mul_f16 v1, v2, v3\nv_pk_add_f16 v1, v1, v4 neg_lo:[0,1]\nv_mad_u32_u24 v0, v1, v2, v3\n")
    for bad in ("v_pk_fma_f16 v1, v2, v3, v4", "v_fma_f16 v1, v2, v3, v4", "v_fma_mix_f32 v1, v2, v3, v4", "v_fmac_f16_e32 v1, v2, v3"):
        with pytest.raises(nat.GGQNativeError):
            nat.check_no_fma(bad)


def test_spill_guard(pkg):
    """The build refuses kernels that keep more than a few bytes per lane in scratch memory (round 6: the 16-row fused kernel spilled 72-108 bytes for Q8_0 and ran at half speed)."""
    nat = pkg._native
    meta = "amdhsa.kernels:\n  - .args: []\n    .name:           {name}\n    .private_segment_fixed_size: {n}\n    .sgpr_count:     40\n    .vgpr_count:     96\n"
    nat.check_no_spills(meta.format(name="_Zfine", n=0) + meta.format(name="_Zalmost", n=nat.MAX_SCRATCH_BYTES))
    with pytest.raises(nat.GGQNativeError, match="_Zspills"):
        nat.check_no_spills(meta.format(name="_Zfine", n=0) + meta.format(name="_Zspills", n=72))


class _Carrier:
    """Anything with .data/.tensor_type/.tensor_shape is what dequantize_tensor reads (dequant.py:16-17)."""

    def __init__(self, data, tensor_type, tensor_shape):
        self.data, self.tensor_type, self.tensor_shape, self.shape = data, tensor_type, tensor_shape, tensor_shape

    def to(self, *a, **k):
        return self.data.to(*a, **k)


def test_dispatch_and_error_behaviour_on_cpu(pkg, golden_dir):
    dq, Q = pkg.dequant, pkg.qtypes.Q
    assert dq.is_torch_compatible(None) and not dq.is_quantized(None)
    w = torch.randn(4, 4)
    assert dq.is_torch_compatible(w) and not dq.is_quantized(w)          # no tensor_type attr
    f16 = _Carrier(torch.randn(4, 4).half(), Q.F16, torch.Size((4, 4)))
    assert not dq.is_quantized(f16)
    assert dq.dequantize_tensor(f16, torch.float32).dtype == torch.float32   # passthrough .to(dtype), dequant.py:19-20
    q = _Carrier(torch.zeros(144, dtype=torch.uint8), Q.Q4_K, torch.Size((1, 256)))
    assert dq.is_quantized(q)
    with pytest.raises(dq.GGQUnsupported, match="cpu"):                  # no CPU fallback in this package
        dq.dequantize_tensor(q, torch.float16)
    with pytest.raises(dq.GGQUnsupported):
        dq.dequantize(q.data, Q.Q4_K, (1, 256))
    with pytest.raises(dq.GGQUnsupported, match="cpu"):                  # any arithmetic mode: GPU-resident bytes only
        dq.dequantize(q.data, Q.Q4_K, (1, 256), dtype=torch.float32)
    with pytest.raises(dq.GGQUnsupported, match="float64"):              # dequant_dtype the kernels do not compute in
        dq.dequantize(q.data, Q.Q4_K, (1, 256), dtype=torch.float64)
    with pytest.raises(dq.GGQUnsupported, match="float64"):
        dq.dequantize_tensor(q, torch.float64, dequant_dtype="target")
    odd = _Carrier(torch.zeros(66, dtype=torch.uint8), Q.IQ2_XXS, torch.Size((256,)))
    with pytest.raises(dq.GGQUnsupported, match="IQ2_XXS"):
        dq.dequantize_tensor(odd, torch.float16)
    # the row lookup (embedding caller) has no CPU path either, and says so before touching the library
    ids = torch.tensor([0, 0])
    with pytest.raises(dq.GGQUnsupported, match="GPU"):
        dq.dequantize_rows(q, ids, torch.float16)
    with pytest.raises(dq.GGQUnsupported, match="2-D"):
        dq.dequantize_rows(_Carrier(torch.zeros(144, dtype=torch.uint8), Q.Q4_K, torch.Size((256,))), ids, torch.float16)
    with pytest.raises(dq.GGQUnsupported, match="whole blocks"):
        dq.dequantize_rows(_Carrier(torch.zeros(144, dtype=torch.uint8), Q.Q4_K, torch.Size((2, 128))), ids, torch.float16)
    with pytest.raises(dq.GGQUnsupported, match="IQ2_XXS"):
        dq.dequantize_rows(odd, ids, torch.float16)
    emb = pkg.ops.GGMLEmbedding(pkg.ops.GGMLTensor(torch.zeros(2, 4), tensor_type=Q.F32, tensor_shape=(2, 4)))
    assert emb(ids, out_dtype=torch.float32).shape == (2, 4)             # CPU / dense table: plain F.embedding
    assert set(dq.dequantize_functions) == set(pkg.qtypes.HIP_QTYPES) | {Q.BF16}
    assert dq.dequantize_functions[Q.Q6_K].__name__ == "dequantize_blocks_Q6_K"
    with pytest.raises(ValueError):
        dq.dequantize_functions[Q.Q6_K](torch.zeros(1, 210, dtype=torch.uint8), 256, 144)
    # BF16 "blocks" are a bit reinterpretation: served by one torch op wherever the data lives
    g = np.load(os.path.join(golden_dir, "BF16.npz"))
    out = dq.dequantize(torch.from_numpy(g["blocks"].copy()), Q.BF16, (4096,))
    assert out.dtype == torch.float32
    assert np.array_equal(out.numpy().view(np.uint32), g["out_f32"].reshape(-1))


def test_custom_op_fake_impl_shapes(pkg):
    """`ggq::dequantize` (the torch.compile entry): the fake implementation predicts shape and dtype
    from the packed byte count alone -- checked on meta tensors, no GPU."""
    dq, Q = pkg.dequant, pkg.qtypes.Q
    if dq._dequantize_op is None:
        pytest.skip("torch.library.custom_op not available")
    for q, nbytes, n_el in ((Q.Q4_K, 144 * 4, 1024), (Q.Q8_0, 34 * 3 + 5, 96), (Q.Q6_K, 0, 0)):
        for code, dt in ((0, torch.float16), (1, torch.bfloat16), (2, torch.float32)):
            out = dq._dequantize_op_fake(torch.empty(nbytes, dtype=torch.uint8, device="meta"), int(q), 0, code)
            assert out.shape == (n_el,) and out.dtype == dt and out.device.type == "meta"
    assert hasattr(torch.ops.ggq, "dequantize")


def test_ggml_tensor_carries_attrs(pkg):
    T, Q = pkg.ops.GGMLTensor, pkg.qtypes.Q
    t = T(torch.zeros(288, dtype=torch.uint8), tensor_type=Q.Q4_K, tensor_shape=(2, 256))
    assert t.shape == torch.Size((2, 256)) and t.size() == torch.Size((288,))
    u = t.to(torch.device("cpu"))
    assert isinstance(u, T) and u.tensor_type == Q.Q4_K and u.shape == torch.Size((2, 256))
    assert t.clone() is t and t.detach() is t
    assert pkg.dequant.is_quantized(t)
    e = t.new_empty((7,))
    assert isinstance(e, T) and e.tensor_type == Q.Q4_K and e.shape == torch.Size((7,)) and e.size() == torch.Size((7,))
    assert t.copy_(torch.zeros(288, dtype=torch.uint8)) is not None
    assert t.copy_(torch.zeros(5, dtype=torch.uint8)) is None          # shape mismatch: logged and ignored (ops.py:70-75)


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
def test_stand_in_layers_equal_the_reference_classes_live(pkg, monkeypatch):
    """ops.py's GGMLTensor and the Linear / Embedding call chains against the reference's own classes (ops.py executed
    verbatim over a fake `comfy`), same packed bytes, on CPU where both sides end in the reference's torch dequantizer."""
    reference.ensure_gguf()
    for k, v in _fake_comfy().items():
        monkeypatch.setitem(sys.modules, k, v)
    root = types.ModuleType("refops")
    root.__path__ = [reference.REFERENCE_DIR]
    monkeypatch.setitem(sys.modules, "refops", root)
    mods = {}
    for name in ("dequant", "ops"):
        spec = importlib.util.spec_from_file_location(f"refops.{name}", os.path.join(reference.REFERENCE_DIR, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        monkeypatch.setitem(sys.modules, f"refops.{name}", m)
        spec.loader.exec_module(m)
        mods[name] = m
    ro, Q = mods["ops"], pkg.qtypes.Q
    packed = torch.from_numpy(pkg.synth.make_tensor_bytes(Q.Q4_K, (2, 256), seed=3).copy())
    ours = pkg.ops.GGMLTensor(packed.clone(), tensor_type=Q.Q4_K, tensor_shape=torch.Size((2, 256)), patches=[("p", "k")])
    theirs = ro.GGMLTensor(packed.clone(), tensor_type=Q.Q4_K, tensor_shape=torch.Size((2, 256)), patches=[("p", "k")])

    def view(t):
        return (type(t).__name__, getattr(t, "tensor_type", None), tuple(t.shape), tuple(t.size()), t.dtype, list(getattr(t, "patches", [])))

    for op in (lambda t: t, lambda t: t.to(torch.device("cpu")), lambda t: t.to(torch.uint8), lambda t: t.clone(), lambda t: t.detach(),
               lambda t: t.new_empty((7,)), lambda t: t.new_empty((3, 2))):
        assert view(op(ours)) == view(op(theirs))
    assert (ours.clone() is ours) and (theirs.clone() is theirs) and (ours.detach() is ours)
    assert (ours.copy_(torch.zeros(5, dtype=torch.uint8)) is None) and (theirs.copy_(torch.zeros(5, dtype=torch.uint8)) is None)
    for t in (ours, theirs):                                             # a real .to() hands out a COPY of the patch list (ops.py:61)
        moved = t.to(torch.int16)
        moved.patches.append("x")
        assert moved is not t and len(t.patches) == 1 and len(moved.patches) == 2
    assert pkg.dequant.is_quantized(ours) and mods["dequant"].is_quantized(theirs)
    # the call chains: with install() in front, our stand-in layers and the reference's layers run the same dequantizer on CPU
    pkg.install.install(mods["dequant"], ro)
    try:
        monkeypatch.setattr(pkg.ops.GGMLLayer, "_dequantize", staticmethod(mods["dequant"].dequantize_tensor))
        w = pkg.synth.make_tensor_bytes(Q.Q6_K, (8, 256), seed=4)
        x = torch.randn(3, 256)
        mine = pkg.ops.GGMLLinear(pkg.ops.GGMLTensor(torch.from_numpy(w.copy()), tensor_type=Q.Q6_K, tensor_shape=(8, 256)))
        lin = ro.GGMLOps.Linear(256, 8)
        lin.weight, lin.bias = torch.nn.Parameter(ro.GGMLTensor(torch.from_numpy(w.copy()), tensor_type=Q.Q6_K, tensor_shape=torch.Size((8, 256))), requires_grad=False), None
        assert torch.equal(mine(x), lin(x))
        ids = torch.tensor([[1, 7, 7, 0]])
        emb_mine = pkg.ops.GGMLEmbedding(pkg.ops.GGMLTensor(torch.from_numpy(w.copy()), tensor_type=Q.Q6_K, tensor_shape=(8, 256)))
        emb = ro.GGMLOps.Embedding(8, 256)
        emb.weight = torch.nn.Parameter(ro.GGMLTensor(torch.from_numpy(w.copy()), tensor_type=Q.Q6_K, tensor_shape=torch.Size((8, 256))), requires_grad=False)
        for out_dtype in (None, torch.float32, torch.bfloat16):
            a, b = emb_mine(ids, out_dtype=out_dtype), emb(ids, out_dtype=out_dtype)
            assert a.dtype == b.dtype and torch.equal(a, b), out_dtype
    finally:
        pkg.install.uninstall(mods["dequant"])


def test_dense_cache_bookkeeping(pkg):
    """resident.DenseCache on CPU with a counting stand-in for the kernel call: hits, per-mode keys, invalidation by
    in-place writes and by the packed tensor's death, LRU eviction under the byte budget, the LoRA bypass."""
    import gc
    T, Q = pkg.ops.GGMLTensor, pkg.qtypes.Q
    calls = []

    def fake(tensor, dtype=None, dequant_dtype=None):
        calls.append((id(tensor), dtype, dequant_dtype))
        return torch.zeros(getattr(tensor, "tensor_shape", tensor.shape), dtype=dtype or torch.float16)

    cache = pkg.resident.DenseCache(3 * 256 * 2 * 2, fake, require_gpu=False)          # room for three (2, 256) fp16 results
    mk = lambda: T(torch.zeros(288, dtype=torch.uint8), tensor_type=Q.Q4_K, tensor_shape=(2, 256))
    a, b = mk(), mk()
    da = cache(a, torch.float16)
    assert cache(a, torch.float16) is da and len(calls) == 1 and cache.stats()["hits"] == 1
    assert cache(a, torch.bfloat16) is not da and len(calls) == 2                      # another output dtype: another entry
    assert cache(a, torch.float16, torch.float32) is not da and len(calls) == 3        # another arithmetic mode too
    assert cache.stats()["entries"] == 3 and cache.bytes == 3 * 1024
    db = cache(b, torch.float16)                                                       # over budget: the least recently used goes
    assert cache.stats()["entries"] == 3 and cache(b, torch.float16) is db
    assert cache(a, torch.float16) is not da and len(calls) == 5                       # ... which was a's fp16 entry
    a.add_(1)                                                                          # in-place write into the packed bytes
    n = len(calls)
    cache(a, torch.float16)
    assert len(calls) == n + 1                                                         # version changed: recomputed
    before = cache.stats()["entries"]
    del b, db
    gc.collect()
    assert cache.stats()["entries"] == before - 1                                      # the entry died with its tensor
    p = T(torch.zeros(288, dtype=torch.uint8), tensor_type=Q.Q4_K, tensor_shape=(2, 256), patches=[("lora", "key")])
    assert cache(p, torch.float16) is not cache(p, torch.float16) and cache.stats()["bypassed"] == 2   # patched: always fresh
    plain = torch.zeros(4)
    assert cache(plain, torch.float32) is not None and cache.stats()["bypassed"] == 3                  # not a GGML tensor
    big = T(torch.zeros(288 * 64, dtype=torch.uint8), tensor_type=Q.Q4_K, tensor_shape=(128, 256))
    cache(big, torch.float16)
    assert cache.bytes <= cache.budget                                                 # larger than the whole budget: not kept
    cache.clear()
    assert cache.stats()["entries"] == 0 and cache.bytes == 0
    # low-VRAM mode: every call brings a NEW tensor object (GGMLTensor.to per forward, ops.py:57-62,209) -- nothing can ever hit.
    # After EPHEMERAL_STREAK entries that died unused the cache stands aside and only probes now and then.
    R = pkg.resident
    for i in range(R.EPHEMERAL_STREAK + 40):
        t = mk()
        cache(t, torch.float16)
        del t
    st = cache.stats()
    assert st["entries"] == 0 and st["ephemeral_bypassed"] == 40 - 40 // R.PROBE_EVERY
    keep = mk()                                                                        # a resident tensor shows up: the next probe caches it,
    for i in range(2 * R.PROBE_EVERY):
        cache(keep, torch.float16)
    assert cache.stats()["entries"] == 1 and cache.stats()["hits"] > st["hits"]        # ... its hits end the streak
    n = cache.stats()["ephemeral_bypassed"]
    other = mk()
    cache(other, torch.float16)
    assert cache.stats()["entries"] == 2 and cache.stats()["ephemeral_bypassed"] == n and not cache.stats()["standing_aside"]
    # a RESIDENT model that runs ONE forward and is then freed (a text encoder encoded once, then unloaded): all of its entries die
    # without a hit too -- but long after they were made, not within the call that made them.  That must not make the cache stand aside
    # for the next model.
    cache2 = R.DenseCache(1 << 30, fake, require_gpu=False)
    model = [mk() for _ in range(R.EPHEMERAL_STREAK + 8)]
    for t in model:
        cache2(t, torch.float16)
    del model, t
    gc.collect()
    assert cache2.stats()["entries"] == 0 and not cache2.stats()["standing_aside"]
    nxt = mk()
    cache2(nxt, torch.float16)
    assert cache2(nxt, torch.float16) is cache2(nxt, torch.float16) and cache2.stats()["ephemeral_bypassed"] == 0


@pytest.mark.skipif(not reference.available(), reason="reference sources neither live nor staged")
def test_reference_harness_dry_run_on_cpu(pkg, monkeypatch):
    """tests/ref_harness.py builds the reference's real layers for tests/test_gpu_reference.py; here the same builders run with
    device = cpu (install() falls through to the reference, so equality is trivial) -- a dry run of the harness, and a check of
    its oracle bridge against the reference's CPU results."""
    import ref_harness as H
    mods = reference.load_reference_package("ggq_refdry", setitem=monkeypatch.setitem)
    rd, ro, Q = mods["dequant"], mods["ops"], pkg.qtypes.Q
    assert reference.source() in ("live", "staged")
    packed = pkg.synth.make_tensor_bytes(Q.Q4_K, (6, 768), seed=11, mode="adversarial")
    w = H.ggml(ro, packed, Q.Q4_K, (6, 768), "cpu", rows=6)
    for dd in H.DEQUANT_DTYPES:
        for dtype in H.DTYPES:
            assert H.same_bits(rd.dequantize_tensor(w, dtype, dd), H.oracle_tensor(Q.Q4_K, packed, dtype, dd, (6, 768))), (dd, dtype)
    lin, packed = H.make_linear(ro, pkg, Q.Q6_K, 24, 512, "cpu", seed=21, patches=H.lora_patch((24, 512), seed=5), dequant_dtype="target")
    plain, _ = H.make_linear(ro, pkg, Q.Q6_K, 24, 512, "cpu", seed=21, dequant_dtype="target")
    x = torch.randn(3, 512)
    want = lin(x)
    assert not torch.equal(want, plain(x))                              # the fake calculate_weight really patches
    wref = H.oracle_tensor(Q.Q6_K, packed, torch.float32, "target", (24, 512))
    assert torch.equal(plain(x), torch.nn.functional.linear(x, wref, torch.Tensor(plain.bias)))
    emb, _ = H.make_embedding(ro, pkg, Q.Q8_0, 40, 512, "cpu", seed=13)
    conv, _ = H.make_conv2d(ro, pkg, Q.Q5_0, 16, 8, 4, 4, "cpu", seed=17)
    ids, img = torch.tensor([[0, 39, 7]]), torch.randn(2, 8, 12, 12)
    e0, c0 = emb(ids, out_dtype=torch.float32), conv(img)
    for options in ({}, {"dense_cache_gb": 1}, {"fused_small_m": True}, {"gather_embedding": True}, {"overlap": True}, {"fused_mfma": True}, {"cpu_route_mb": 1}):
        with H.Installed(pkg, mods, **options):
            assert torch.equal(lin(x), want) and torch.equal(emb(ids, out_dtype=torch.float32), e0) and torch.equal(conv(img), c0)
    moved = plain.to("cpu")
    assert type(moved.weight) is ro.GGMLTensor and tuple(moved.weight.shape) == (24, 512)


def test_partition_covers_and_balances(pkg):
    sh, man = pkg.sharding, pkg.manifests
    for manifest in (man.flux_dev(), man.sd35_t5(), man.flux_linear_pool(12, 3)):
        for world in (1, 2, 3, 4, 8):
            bins = sh.partition(manifest, world)
            flat = sorted(i for b in bins for i in b)
            assert flat == list(range(len(manifest)))
            if len(manifest) >= 100:                                   # whole tensors only: small lists cannot balance
                assert sh.imbalance(manifest, world) < 1.02
            assert bins == sh.partition(manifest, world)            # deterministic: ranks agree without talking
            assert [manifest[i] for i in bins[world - 1]] == sh.shard(manifest, world - 1, world)
    assert sh.partition([], 2) == [[], []]
    with pytest.raises(ValueError):
        sh.partition(man.flux_dev(), 0)


def test_manifests_are_well_formed(pkg):
    qt, man = pkg.qtypes, pkg.manifests
    for manifest in (man.flux_dev(), man.flux_dev("Q8_0"), man.sd35_t5(), man.flux_linear_pool(qt.Q.Q6_K)):
        names = [e[0] for e in manifest]
        assert len(set(names)) == len(names)
        for _, q, shape in manifest:
            bs, _ = qt.block_geometry(q)
            assert q in qt.HIP_QTYPES and len(shape) == 2
            assert shape[1] % bs == 0, (q, shape)          # rows are whole blocks (lcpp.patch:227-253)
    flux = man.flux_dev()
    n_el = sum(a * b for _, _, (a, b) in flux)
    assert 11.5e9 < n_el < 12.1e9                          # FLUX.1-dev: ~11.9 B parameters in the blocks


# ---------------------------------------------------------------- install() in front of the reference

def _fake_comfy():
    """The comfy symbols reference ops.py touches: oracle/fake_comfy.py, shared with the GPU tests."""
    from oracle import fake_comfy
    return fake_comfy.build()


def _reference_modules(monkeypatch):
    """The reference's real dequant.py / ops.py imported as ``refpkg.*`` over the fake comfy."""
    reference.ensure_gguf()
    for k, v in _fake_comfy().items():
        monkeypatch.setitem(sys.modules, k, v)
    pk = types.ModuleType("refpkg")
    pk.__path__ = [reference.REFERENCE_DIR]
    monkeypatch.setitem(sys.modules, "refpkg", pk)
    mods = {}
    for name in ("dequant", "ops"):
        spec = importlib.util.spec_from_file_location(f"refpkg.{name}", os.path.join(reference.REFERENCE_DIR, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        monkeypatch.setitem(sys.modules, f"refpkg.{name}", m)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["dequant"], mods["ops"]


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
def test_install_default_policy_and_vram_accounting(pkg, monkeypatch):
    """Round 5: (1) the default install turns the no-VRAM opt-ins on, ``exact`` / GGQ_EXACT turns them off, contradictions and ``fast`` without
    ``ref_ops`` raise; (2) options that HOLD device memory make the reference's fake state dict (GGMLLayer.ggml_save_to_state_dict, ops.py:145-160:
    what ComfyUI sizes a GGUF model by) grow by exactly ``scratch_reservation(...)["total"]`` on the largest layer -- and a default install by 0."""
    rd, ro = _reference_modules(monkeypatch)
    inst, Q = pkg.install, pkg.qtypes.Q
    for v in ("GGQ_FAST", "GGQ_EXACT", "GGQ_FUSED_SMALL_M", "GGQ_FUSED_MFMA", "GGQ_GATHER_EMBEDDING", "GGQ_DENSE_CACHE_GB", "GGQ_OVERLAP"):
        monkeypatch.delenv(v, raising=False)
    rows, cols = 8, 512
    blocks = pkg.synth.make_blocks(Q.Q4_K, rows * cols // 256, seed=3)
    lin = ro.GGMLOps.Linear(cols, rows)
    lin.weight = torch.nn.Parameter(ro.GGMLTensor(torch.from_numpy(blocks.reshape(-1).copy()), tensor_type=Q.Q4_K, tensor_shape=torch.Size((rows, cols))), requires_grad=False)
    lin.bias = None
    lin.largest_layer = True                                             # what ggml_load_from_state_dict sets for loader.py:134-137's tensor
    small = ro.GGMLOps.Linear(cols, rows)
    small.weight, small.bias = lin.weight, None
    reference_save = ro.GGMLLayer.ggml_save_to_state_dict

    def fake_bytes(layer):
        sd = layer.state_dict()
        assert all(t.device.type == "meta" for t in sd.values())
        return sum(t.numel() * t.element_size() for t in sd.values()), set(sd)

    packed_bytes, dense_bytes = blocks.size, rows * cols * 2
    base, base_keys = fake_bytes(lin)
    assert base == packed_bytes + dense_bytes and base_keys == {"weight", "temp.weight"}      # the reference's own estimate
    # (1) policy
    assert inst.DEFAULT_FAST is True
    inst.install(rd, ro)
    try:
        assert inst._installed[id(rd)]["options"] == {"fused_small_m": True, "fused_mfma": 256, "gather_embedding": True}
        assert hasattr(ro.GGMLOps.Linear.forward_ggml_cast_weights, "__wrapped__") and hasattr(ro.GGMLOps.Embedding.forward_ggml_cast_weights, "__wrapped__")
        assert ro.GGMLLayer.ggml_save_to_state_dict is reference_save and fake_bytes(lin) == (base, base_keys)      # holds nothing: reserves nothing
        assert inst.scratch_reservation(rd, dense_bytes, packed_bytes) == {"dense_cache": 0, "overlap": 0, "total": 0}
        assert "GGQ_EXACT=1" in inst.describe(rd)
    finally:
        inst.uninstall(rd)
    assert not hasattr(ro.GGMLOps.Linear.forward_ggml_cast_weights, "__wrapped__")
    inst.install(rd)                                                      # functions only: nothing above them to fuse
    assert inst._installed[id(rd)]["options"] == {}
    inst.uninstall(rd)
    for kw, env in (({"exact": True}, {}), ({}, {"GGQ_EXACT": "1"}), ({}, {"GGQ_FAST": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        inst.install(rd, ro, **kw)
        opts = inst._installed[id(rd)]["options"]
        assert not hasattr(ro.GGMLOps.Linear.forward_ggml_cast_weights, "__wrapped__") and "fused_small_m" not in opts
        assert ("exact" in opts) == ("GGQ_FAST" not in env)
        assert ("exact=True" if "exact" in opts else "bit-exact unpack + F.linear") in inst.describe(rd)
        inst.uninstall(rd)
        for k in env:
            monkeypatch.delenv(k)
    inst.install(rd, ro, exact=True, fused_small_m=True)                  # an option asked for by name wins over the switch
    assert inst._installed[id(rd)]["options"] == {"fused_small_m": True, "exact": True}
    inst.uninstall(rd)
    with pytest.raises(ValueError):
        inst.install(rd, ro, fast=True, exact=True)
    with pytest.raises(ValueError):
        inst.install(rd, fast=True)                                       # used to be ignored silently
    monkeypatch.setenv("GGQ_FAST", "1")
    with pytest.raises(ValueError):
        inst.install(rd)
    monkeypatch.delenv("GGQ_FAST")
    assert id(rd) not in inst._installed and rd.dequantize_tensor.__module__ == rd.__name__
    # (2) accounting
    for options, want in (({"dense_cache_gb": 2}, 2 * 10 ** 9), ({"overlap": True}, 2 * dense_bytes + 3 * packed_bytes),
                          ({"dense_cache_gb": 0.5, "overlap": True}, 5 * 10 ** 8 + 2 * dense_bytes + 3 * packed_bytes)):
        inst.install(rd, ro, exact=True, **options)
        try:
            assert ro.GGMLLayer.ggml_save_to_state_dict.__wrapped__ is reference_save
            assert inst.scratch_reservation(rd, dense_bytes, packed_bytes)["total"] == want
            got, keys = fake_bytes(lin)
            assert got == base + want and keys == base_keys | {"temp.ggq_scratch"}, options
            assert fake_bytes(small) == (packed_bytes, {"weight"})        # only the largest layer carries the reservation (like temp.weight)
            assert inst.scratch_bytes(rd)["total"] == 0                   # nothing is HELD yet: the reservation is the worst case, up front
        finally:
            inst.uninstall(rd)
        assert ro.GGMLLayer.ggml_save_to_state_dict is reference_save and fake_bytes(lin) == (base, base_keys)
    # dequant_dtype float32: the reference's temp.weight is fp32 (ops.py:156) and so are overlap's dense slots
    lin.dequant_dtype = torch.float32
    inst.install(rd, ro, exact=True, overlap=True)
    try:
        assert fake_bytes(lin)[0] == packed_bytes + 2 * dense_bytes + (2 * 2 * dense_bytes + 3 * packed_bytes)
    finally:
        inst.uninstall(rd)


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
def test_install_keeps_reference_behaviour_on_cpu(pkg, monkeypatch):
    """With install() applied, the reference's own GGMLOps.Linear still runs end to end; CPU-resident
    weights keep flowing through the reference's original functions (the HIP path only takes GPU data)."""
    rd, ro = _reference_modules(monkeypatch)
    Q = pkg.qtypes.Q
    blocks = pkg.synth.make_blocks(Q.Q4_K, 8 * 2, seed=21)                 # weight 8 x 512
    weight = ro.GGMLTensor(torch.from_numpy(blocks.reshape(-1).copy()), tensor_type=Q.Q4_K, tensor_shape=torch.Size((8, 512)))
    lin = ro.GGMLOps.Linear(512, 8)
    lin.weight, lin.bias = torch.nn.Parameter(weight, requires_grad=False), None
    x = torch.randn(3, 512)
    before, before_row = lin(x), lin(x[:1])
    orig = pkg.install.install(rd, ro)
    try:
        assert rd.dequantize is not orig["dequantize"] and ro.dequantize_tensor is rd.dequantize_tensor
        after = lin(x)                                                        # CPU weight -> reference path
        assert torch.equal(before, after)
        want = torch.from_numpy(oracle.dequant_f16(Q.Q4_K, blocks).reshape(8, 512).copy()).float()
        assert torch.equal(after, torch.nn.functional.linear(x, want))
        # CPU-resident bytes in another arithmetic mode: the reference's own code, not an error
        w32 = rd.dequantize_tensor(weight, torch.float32, dequant_dtype=torch.float32)
        assert w32.dtype == torch.float32
        assert np.array_equal(w32.numpy().reshape(-1), oracle.dequant_f32(Q.Q4_K, blocks))
    finally:
        pkg.install.uninstall(rd)
    assert rd.dequantize is orig["dequantize"] and ro.dequantize_tensor is orig["dequantize_tensor"]
    # with the opt-in resident cache in front: CPU tensors bypass it and still reach the reference's own code
    pkg.install.install(rd, ro, dense_cache_gb=1)
    try:
        assert torch.equal(lin(x), before)
        st = pkg.install.dense_cache(rd).stats()
        assert st["bypassed"] >= 1 and st["entries"] == 0
    finally:
        pkg.install.uninstall(rd)
    assert pkg.install.dense_cache(rd) is None
    # with the opt-in fused small-m linear: a CPU input is not the kernel's business -> the reference's method, same result
    ref_forward = ro.GGMLOps.Linear.forward_ggml_cast_weights
    pkg.install.install(rd, ro, fused_small_m=True)
    try:
        assert ro.GGMLOps.Linear.forward_ggml_cast_weights is not ref_forward
        assert torch.equal(lin(x), before)
        assert torch.equal(lin(x[:1]), before_row)
    finally:
        pkg.install.uninstall(rd)
    assert ro.GGMLOps.Linear.forward_ggml_cast_weights is ref_forward
    with pytest.raises(ValueError):
        pkg.install.install(rd, fused_small_m=True)
    pkg.install.uninstall(rd)
    # with the opt-in row lookup: a CPU table is not the kernel's business -> the reference's method, same result
    table = ro.GGMLTensor(torch.from_numpy(pkg.synth.make_tensor_bytes(Q.Q8_0, (16, 256), seed=5).copy()), tensor_type=Q.Q8_0, tensor_shape=torch.Size((16, 256)))
    emb = ro.GGMLOps.Embedding(16, 256)
    emb.weight = torch.nn.Parameter(table, requires_grad=False)
    ids = torch.tensor([[0, 3, 15, 3]])
    ref_rows = emb(ids, out_dtype=torch.float32)
    ref_emb_forward = ro.GGMLOps.Embedding.forward_ggml_cast_weights
    pkg.install.install(rd, ro, gather_embedding=True)
    try:
        assert ro.GGMLOps.Embedding.forward_ggml_cast_weights is not ref_emb_forward
        assert torch.equal(emb(ids, out_dtype=torch.float32), ref_rows)
    finally:
        pkg.install.uninstall(rd)
    assert ro.GGMLOps.Embedding.forward_ggml_cast_weights is ref_emb_forward


def test_tools_and_entry_points_at_least_parse():
    """Every script under tools/ (they need a GPU to RUN) byte-compiles, and the argparse-driven ones answer --help."""
    import py_compile
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(f for f in os.listdir(os.path.join(root, "tools")) if f.endswith(".py"))
    assert len(scripts) >= 8
    for f in scripts + ["../bench.py", "../__graft_entry__.py"]:
        py_compile.compile(os.path.join(root, "tools", f), doraise=True)
    for f in ("flux_forward_emulation.py", "mode_table.py", "build_variant.py", "upload_sweep.py"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", f), "--help"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "usage" in r.stdout.lower(), (f, r.stderr[-500:])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout


@pytest.mark.skipif(not reference.available(), reason="/root/reference not present (GPU box)")
def test_dispatch_tables_equal_the_reference_live(pkg):
    """The format table and the predicates around it against the reference's dequant.py executed verbatim: the HIP path serves
    exactly the formats the reference has block functions for (nothing falls back to gguf's numpy path that did not before)."""
    ref = reference.load_reference_dequant()
    dq, Q = pkg.dequant, pkg.qtypes.Q
    assert {int(k) for k in dq.dequantize_functions} == {int(k) for k in ref.dequantize_functions}
    assert {int(k) for k in dq.TORCH_COMPATIBLE_QTYPES if k is not None} == {int(k) for k in ref.TORCH_COMPATIBLE_QTYPES if k is not None}
    assert (None in dq.TORCH_COMPATIBLE_QTYPES) == (None in ref.TORCH_COMPATIBLE_QTYPES)
    probes = [None, torch.zeros(3), _Carrier(torch.zeros(4, 4).half(), Q.F16, torch.Size((4, 4))), _Carrier(torch.zeros(4), Q.F32, torch.Size((4,))),
              _Carrier(torch.zeros(8, dtype=torch.uint8), Q.BF16, torch.Size((4,))), _Carrier(torch.zeros(144, dtype=torch.uint8), Q.Q4_K, torch.Size((256,))),
              _Carrier(torch.zeros(66, dtype=torch.uint8), Q.IQ2_XXS, torch.Size((256,)))]
    for t in probes:
        assert dq.is_quantized(t) == ref.is_quantized(t) and dq.is_torch_compatible(t) == ref.is_torch_compatible(t), getattr(t, "tensor_type", t)


def test_bench_traffic_is_only_reported_for_the_build_it_was_measured_on(pkg, monkeypatch, tmp_path):
    """bench.load_traffic: the PMC figure of profiles/pmc_traffic.json is reported only when the loaded library's ggq_build_id() equals the
    id recorded with the counters; otherwise traffic is None and traffic_source says why (Weak #8)."""
    import json
    import bench
    build_id = pkg._native.lib().ggq_build_id().decode()
    assert build_id == pkg._native.source_id() and len(build_id) == 16            # the in-tree library is stamped with the source identity
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.load_traffic(pkg, "Q4_K:pairs64")[0] is None                       # no file
    (prof / "pmc_traffic.json").write_text(json.dumps({"_build_id": "0123456789abcdef", "Q4_K:pairs64": 123}))
    t, src = bench.load_traffic(pkg, "Q4_K:pairs64")
    assert t is None and "stale" in src and "0123456789abcdef" in src and build_id in src
    (prof / "pmc_traffic.json").write_text(json.dumps({"_build_id": build_id, "_provenance": "how", "Q4_K:pairs64": 123}))
    assert bench.load_traffic(pkg, "Q4_K:pairs64") == (123, f"how; library build {build_id}")
    t, src = bench.load_traffic(pkg, "Q9_Z:pairs64")
    assert t is None and "no PMC entry" in src
    # the committed table: either it belongs to the committed kernels (then the figure is sane), or bench.py will refuse to report it
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    monkeypatch.setattr(bench, "ROOT", ROOT)
    t, src = bench.load_traffic(pkg, "Q4_K:pairs64")
    if committed["_build_id"] == build_id:
        assert t == committed["Q4_K:pairs64"] and abs(t / 7738490880 - 1.0) < 0.01
    else:
        assert t is None and "stale" in src


def test_bench_median_and_min_helper():
    import time
    import bench
    calls = []
    med, tmin, reps = bench._median_min(lambda: (calls.append(1), time.sleep(0.001)), budget_s=0.0, min_reps=10, warmup=3)
    assert reps == 10 and len(calls) == 13 and 0.001 <= tmin <= med < 0.05          # >= 3 warm-up, >= 10 timed passes, median and min
    med, tmin, reps = bench._median_min(lambda: time.sleep(0.001), budget_s=0.05, min_reps=3, warmup=0)
    assert 20 <= reps <= 200                                                         # ... and as many as the time budget allows


def test_q4_k_exhaustive_generator_covers_what_it_says(pkg):
    """synth.q4_k_exhaustive_blocks: every fp16 pattern of the swept scale field x every 6-bit factor that multiplies it x every quant in both nibbles
    (decoded here with the reference's get_scale_min layout, dequant.py:129-139) -- the exhaustive parity tests stand on this."""
    import numpy as np
    for which, lo in (("d", 0), ("dmin", 2)):
        b = pkg.synth.q4_k_exhaustive_blocks(which, seed=5)
        assert b.shape == (65536 * 8, 144) and b.dtype == np.uint8
        s = b[:, 4:16].astype(np.uint32)
        sc, mn = np.zeros((b.shape[0], 8), np.uint32), np.zeros((b.shape[0], 8), np.uint32)
        for j in range(4):
            sc[:, j], mn[:, j] = s[:, j] & 63, s[:, j + 4] & 63
            sc[:, j + 4] = (s[:, j + 8] & 15) | ((s[:, j] >> 6) << 4)
            mn[:, j + 4] = (s[:, j + 8] >> 4) | ((s[:, j + 4] >> 6) << 4)
        swept = sc if which == "d" else mn
        assert np.array_equal(swept, (8 * (np.arange(b.shape[0]) % 8))[:, None] + np.arange(8)[None, :])          # 64 factors per pattern
        field = b[:, lo].astype(np.uint32) | (b[:, lo + 1].astype(np.uint32) << 8)
        assert np.array_equal(field, np.repeat(np.arange(65536), 8))                                               # every bit pattern, 8 blocks each
        qs = b[0, 16:]
        for p in range(4):                                                                                         # each 32-byte run feeds two sub-blocks
            assert sorted(set(qs[32 * p:32 * p + 32] & 15)) == list(range(16)) and sorted(set(qs[32 * p:32 * p + 32] >> 4)) == list(range(16))
    b5 = pkg.synth.k_scmn_exhaustive_blocks(pkg.qtypes.Q.Q5_K, "d", seed=5)
    assert b5.shape == (65536 * 8, 176)
    qh, qs = b5[0, 16:48].astype(np.uint32), b5[0, 48:].astype(np.uint32)
    for sb in range(8):                                                                                            # dequant.py:159-178: q = nibble | bit sb of qh[l] << 4
        nib = (qs[32 * (sb // 2):32 * (sb // 2) + 32] >> (4 * (sb & 1))) & 15
        assert sorted(set((nib | (((qh >> sb) & 1) << 4)).tolist())) == list(range(32))


def test_k_exhaustive_generators_cover_what_they_say(pkg):
    """synth.k_exhaustive_blocks for Q2_K / Q3_K / Q6_K / IQ4_XS, decoded here with the layouts of dequant.py:141-285: every scale bit pattern, every
    integer sub-block factor against each, every quant value in every sub-block."""
    import numpy as np
    Q, gen = pkg.qtypes.Q, pkg.synth.k_exhaustive_blocks
    for which, off in (("d", 80), ("dmin", 82)):
        b = gen(Q.Q2_K, which, seed=1)
        assert b.shape == (65536, 84)
        swept = (b[:, 0:16] & 15) if which == "d" else (b[:, 0:16] >> 4)
        assert (swept == np.arange(16)[None, :]).all()
        assert ((b[:, off].astype(np.uint32) | (b[:, off + 1].astype(np.uint32) << 8)) == np.arange(65536)).all()
        qs = b[0, 16:80].astype(int)
        el = np.array([(qs[32 * (e // 128) + e % 32] >> (2 * ((e % 128) // 32))) & 3 for e in range(256)])
        assert all(sorted(set(el[16 * j:16 * j + 16].tolist())) == [0, 1, 2, 3] for j in range(16))
    b = gen(Q.Q3_K, "d", seed=1)
    assert b.shape == (65536 * 4, 110)
    scb, v = b[:, 96:108].astype(np.uint32), np.zeros((b.shape[0], 16), np.uint32)
    for j in range(16):
        v[:, j] = ((scb[:, j] & 15) if j < 8 else (scb[:, j - 8] >> 4)) | (((scb[:, 8 + j % 4] >> (2 * (j // 4))) & 3) << 4)
    assert (v == (16 * (np.arange(b.shape[0]) % 4))[:, None] + np.arange(16)[None, :]).all()
    assert ((b[:, 108].astype(np.uint32) | (b[:, 109].astype(np.uint32) << 8)) == np.repeat(np.arange(65536), 4)).all()
    hm, qs = b[0, 0:32].astype(int), b[0, 32:96].astype(int)
    q3 = np.array([((qs[32 * (e // 128) + e % 32] >> (2 * ((e % 128) // 32))) & 3) - (0 if (hm[e % 32] >> (e // 32)) & 1 else 4) for e in range(256)])
    assert all(sorted(set(q3[16 * j:16 * j + 16].tolist())) == list(range(-4, 4)) for j in range(16))
    b = gen(Q.Q6_K, "d", seed=1, d_range=(100, 104))
    assert b.shape == (64, 210) and (b[:, 192:208].astype(np.uint32) == (16 * (np.arange(64) % 16))[:, None] + np.arange(16)[None, :]).all()
    assert ((b[:, 208].astype(np.uint32) | (b[:, 209].astype(np.uint32) << 8)) == np.repeat(np.arange(100, 104), 16)).all()
    b = gen(Q.IQ4_XS, "d", seed=1)
    assert b.shape == (65536 * 8, 136)
    sh, sl = b[:, 2].astype(np.uint32) | (b[:, 3].astype(np.uint32) << 8), b[:, 4:8].astype(np.uint32)
    ls = np.stack([((sl[:, ib // 2] >> (4 * (ib % 2))) & 15) | (((sh >> (2 * ib)) & 3) << 4) for ib in range(8)], 1)
    assert (ls == (8 * (np.arange(b.shape[0]) % 8))[:, None] + np.arange(8)[None, :]).all()
    assert ((b[:, 0].astype(np.uint32) | (b[:, 1].astype(np.uint32) << 8)) == np.repeat(np.arange(65536), 8)).all()
    for ib in range(8):
        run = b[0, 8 + 16 * ib:8 + 16 * ib + 16]
        assert sorted(set((run & 15).tolist())) == list(range(16)) and sorted(set((run >> 4).tolist())) == list(range(16))


def test_shipped_library_reads_no_environment(pkg):
    """Measurement knobs (GGQ_XRUN_LOG2, GGQ_LDS_PAD, GGQ_TILE_MIN_M, ...) exist only in lab builds (-DGGQ_LAB, tools/build_variant.py): the
    shipped library holds no GGQ_* string at all (`strings libggq_hip.so | grep GGQ_` is empty) and does not import getenv."""
    path = pkg._native.LIB_PATH
    if os.environ.get("GGQ_HIP_LIB"):
        pytest.skip("a variant library is loaded")
    with open(path, "rb") as f:
        blob = f.read()
    assert b"GGQ_" not in blob
    import subprocess
    nm = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout


def test_in_process_multi_device_placement(pkg):
    """grouped.ShardedPlan.assignment: the split of ONE process over several GPUs is the very partition the one-process-per-GPU runs
    use (sharding.partition), shard r on devices[r] -- disjoint, complete, balanced; devices may repeat; empty shards drop out."""
    SP = pkg.grouped.ShardedPlan
    manifest = pkg.manifests.sd35_t5("Q4_K_M")
    devices = [f"cuda:{i}" for i in range(8)]
    rows = SP.assignment(manifest, devices)
    assert [str(d) for d, _ in rows] == devices
    assert [ix for _, ix in rows] == pkg.sharding.partition(manifest, 8)
    assert sorted(i for _, ix in rows for i in ix) == list(range(len(manifest)))
    loads = [sum(pkg.sharding.tensor_cost(manifest[i]) for i in ix) for _, ix in rows]
    assert max(loads) / (sum(loads) / 8) < 1.01
    two = SP.assignment(manifest, ["cuda:0", "cuda:0"])                      # a one-GPU test box: two shards on one device
    assert [str(d) for d, _ in two] == ["cuda:0", "cuda:0"] and [ix for _, ix in two] == pkg.sharding.partition(manifest, 2)
    few = SP.assignment(manifest[:3], devices)                              # 3 tensors, 8 devices: 3 shards
    assert len(few) == 3 and sorted(i for _, ix in few for i in ix) == [0, 1, 2]
    with pytest.raises(ValueError):
        SP.assignment(manifest, [])
    with pytest.raises(ValueError):
        pkg.loader.gguf_sd_loader("nope.gguf", devices=["cuda:0"], device="cuda:0")
    with pytest.raises(ValueError):
        pkg.loader.gguf_sd_loader("nope.gguf", devices=[])


def test_bench_parity_statement_is_the_min_over_ranks():
    """bench.py: `cpu_baseline.parity_vs_gpu` of an N-rank line = every rank's verdict; one differing tensor on one rank -> MISMATCH."""
    import bench
    ok = [{"rank": r, "tensors": 128, "differ": 0, "first": [], "seconds": 1.0} for r in range(8)]
    s = bench.parity_statement(ok, "bit-exact", with_reference=True)
    assert s.startswith("bit-exact (8 x 128 = 1024 tensors on 8 ranks: every output of every rank's timed launch vs the oracle") and "rank 0's first 2 also vs the reference" in s
    assert bench.parity_statement(ok[:1], "bit-exact", with_reference=False) == "bit-exact (128 tensors: every output of the timed launch vs the oracle)"
    ragged = [dict(r, tensors=t) for r, t in zip(ok, (66, 69, 69, 69, 70, 70, 68, 68))]
    assert "66 + 69 + 69 + 69 + 70 + 70 + 68 + 68 = 549 tensors on 8 ranks" in bench.parity_statement(ragged, "bit-exact")
    bad = [dict(r) for r in ok]
    bad[5].update(differ=1, first=[[17, "element 3 (block 0): got 1.0, expected 2.0"]])
    s = bench.parity_statement(bad, "bit-exact")
    assert s.startswith("MISMATCH (1 of 1024 tensors differ from the oracle on ranks [5]")
    assert bench.parity_statement(ok, "MISMATCH").startswith("MISMATCH")                 # rank 0's check against the reference itself counts too
