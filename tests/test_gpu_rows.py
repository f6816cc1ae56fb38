"""Row lookup straight from a packed table (include/ggq.h ggq_dequant_rows; the embedding caller, reference ops.py:251-260) on an
MI355X: ``dequantize_rows(table, indices, dtype, dequant_dtype)`` must equal ``F.embedding(indices, dequantize_tensor(table, dtype,
dequant_dtype))`` bit for bit -- checked against the CPU oracle's dequantization of the whole table, in every arithmetic / output
mode, and against the two-step path of the layer."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ALL = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "IQ4_NL", "IQ4_XS"]
KINDS = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def _canon(a, kind):
    a = np.ascontiguousarray(a)
    if kind == "f32":
        return oracle.canon_nan(a.view(np.float32))
    return oracle.canon_nan_f16(a) if kind == "f16" else oracle.canon_nan_bf16(a)


def _bits(t, kind):
    t = t.contiguous().cpu()
    return t.numpy() if kind == "f32" else t.view(torch.int16).numpy().view(np.uint16)


def _table(pkg, q, n_rows, cols, seed, mode="nominal"):
    blocks = pkg.synth.make_tensor_bytes(q, (n_rows, cols), seed=seed, mode=mode)
    return blocks, pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(n_rows, cols))


@pytest.mark.parametrize("name", ALL)
def test_rows_equal_the_oracle_rows_in_every_mode(pkg, name):
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    g = torch.Generator().manual_seed(int(q))
    # 8 / 24: whole groups; 40 x 32: a partial group inside every row; 1, 3, 14 (3584-wide K-quant rows), 67: rows that start
    # at 2-, 4- or 8-byte aligned addresses only and end inside a group
    for blocks_per_row in (8, 24, 1, 3, 14, 67) + ((40,) if bs == 32 else ()):
        cols, n_rows = blocks_per_row * bs, 37
        blocks, table = _table(pkg, q, n_rows, cols, seed=100 + blocks_per_row, mode="signed" if blocks_per_row == 24 else "nominal")
        idx = torch.randint(0, n_rows, (3, 19), generator=g)
        idx[0, :4] = torch.tensor([0, n_rows - 1, 5, 5])                       # first, last, a repeated row
        for compute, out in (("f16", "f16"), ("f16", "bf16"), ("f16", "f32"), ("bf16", "bf16"), ("f32", "f32"), ("f32", "f16"), ("bf16", "f32")):
            got = pkg.dequant.dequantize_rows(table, idx.to(DEV), KINDS[out], KINDS[compute])
            assert got.shape == (3, 19, cols) and got.dtype == KINDS[out]
            want = oracle.dequant_tensor(q, blocks, compute, out).reshape(n_rows, cols)[idx.numpy().reshape(-1)]
            assert np.array_equal(_canon(_bits(got, out).reshape(-1, cols), out), _canon(want, out)), (name, blocks_per_row, compute, out)
        # dequant_dtype="target" and the default (None): the reference's argument meaning
        a = pkg.dequant.dequantize_rows(table, idx.to(DEV), torch.bfloat16, "target")
        b = pkg.dequant.dequantize_rows(table, idx.to(DEV), torch.bfloat16, torch.bfloat16)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        assert pkg.dequant.dequantize_rows(table, idx.to(DEV)).dtype == torch.float16


def test_rows_match_the_two_step_path_of_the_layer(pkg):
    Q, ops = pkg.qtypes.Q, pkg.ops
    _, table = _table(pkg, Q.Q4_K, 4099, 1024, seed=7)                         # a small "vocabulary"
    emb = ops.GGMLEmbedding(table)
    ids = torch.randint(0, 4099, (2, 77), device=DEV)
    for out_dtype in (None, torch.bfloat16, torch.float16, torch.float32):
        emb.gather_rows = True
        fast = emb(ids, out_dtype=out_dtype)
        emb.gather_rows = False
        slow = emb(ids, out_dtype=out_dtype)                                   # dequantize the whole table, then F.embedding
        assert fast.dtype == slow.dtype and fast.shape == slow.shape == (2, 77, 1024)
        assert torch.equal(fast, slow)
    emb.gather_rows = True
    assert torch.equal(emb(ids.to(torch.int32), out_dtype=torch.float16), emb(ids, out_dtype=torch.float16))
    assert emb(ids[:0], out_dtype=torch.float16).shape == (0, 77, 1024)
    # other arithmetic modes of the loader node reach the kernel too
    emb.dequant_dtype = torch.float32
    fast = emb(ids, out_dtype=torch.bfloat16)
    emb.gather_rows = False
    assert torch.equal(fast, emb(ids, out_dtype=torch.bfloat16))


def test_rows_limits(pkg):
    Q = pkg.qtypes.Q
    U = pkg.dequant.GGQUnsupported
    ids = torch.zeros(4, dtype=torch.int64, device=DEV)
    ragged = pkg.ops.GGMLTensor(torch.zeros(16 * 84 + 84, dtype=torch.uint8, device=DEV), tensor_type=Q.Q2_K, tensor_shape=(16, 256))
    with pytest.raises(U):
        pkg.dequant.dequantize_rows(ragged, ids, torch.float16)                # packed bytes do not match the logical shape
    f16_table = pkg.ops.GGMLTensor(torch.zeros(16, 256, dtype=torch.float16, device=DEV), tensor_type=Q.F16, tensor_shape=(16, 256))
    with pytest.raises(U):
        pkg.dequant.dequantize_rows(f16_table, ids, torch.float16)             # stored dense: nothing to unpack
    emb = pkg.ops.GGMLEmbedding(f16_table)                                     # ... so the layer takes the two-step path by itself
    assert torch.equal(emb(ids, out_dtype=torch.float16), f16_table.as_subclass(torch.Tensor)[ids])
    _, table = _table(pkg, Q.Q8_0, 8, 256, seed=2)
    with pytest.raises(U):
        pkg.dequant.dequantize_rows(table, ids.cpu(), torch.float16)           # indices on the CPU
    with pytest.raises(U):
        pkg.dequant.dequantize_rows(table, ids.to(torch.float32), torch.float16)
    with pytest.raises(U):
        pkg.dequant.dequantize_rows(table, ids, torch.float64)
    # out-of-range ids are clamped, not faulted on (F.embedding would assert)
    wild = torch.tensor([-3, 7, 8, 10 ** 9], device=DEV)
    got = pkg.dequant.dequantize_rows(table, wild, torch.float16)
    full = pkg.dequant.dequantize_tensor(table, torch.float16)
    assert torch.equal(got, full[torch.tensor([0, 7, 7, 7], device=DEV)])
    # direct C call: argument validation
    nat = pkg._native
    L = nat.lib()
    assert L.ggq_dequant_rows(int(Q.Q8_0), None, 8, 8, None, 0, None, 0, 0, None) == nat.GGQ_OK           # nothing to do
    assert L.ggq_dequant_rows(999, table.data_ptr(), 8, 8, ids.data_ptr(), 4, got.data_ptr(), 0, 0, None) == nat.GGQ_ERR_QTYPE
    assert L.ggq_dequant_rows(int(Q.Q8_0), table.data_ptr(), 8, 8, ids.data_ptr(), 4, got.data_ptr(), 0, 5, None) == nat.GGQ_ERR_ARG
    assert L.ggq_dequant_rows(int(Q.Q8_0), table.data_ptr() + 2, 8, 8, ids.data_ptr(), 4, got.data_ptr(), 0, 0, None) == nat.GGQ_ERR_ALIGN


def test_install_gather_embedding_wraps_the_embedding_forward(pkg, monkeypatch):
    """install(..., gather_embedding=True) wraps ``GGMLOps.Embedding.forward_ggml_cast_weights`` (reference ops.py:251-260).  The
    reference is not on the GPU box, so the wrapper is driven on a class with that method's shape.  Under install() the lookup keeps F.embedding's
    error behaviour: an asynchronous device-side assert on ids outside the table (round 5: the option is part of the default install), unless GGQ_CHECK_INDICES=0."""
    Q, ops = pkg.qtypes.Q, pkg.ops
    monkeypatch.delenv("GGQ_CHECK_INDICES", raising=False)
    asserted = []
    real_assert = torch._assert_async
    monkeypatch.setattr(torch, "_assert_async", lambda cond, msg="": (asserted.append(msg), real_assert(cond, msg))[1])

    class Embedding(ops.GGMLLayer):
        calls = 0
        max_norm = None
        padding_idx = None

        def forward_ggml_cast_weights(self, input, out_dtype=None):
            type(self).calls += 1
            weight, _ = self.cast_bias_weight(self, device=input.device, dtype=out_dtype)
            return torch.nn.functional.embedding(input, weight, self.padding_idx).to(dtype=out_dtype)

        def forward(self, input, out_dtype=None):
            return self.forward_ggml_cast_weights(input, out_dtype)

    original = Embedding.forward_ggml_cast_weights
    _, table = _table(pkg, Q.Q6_K, 300, 2048, seed=9)
    emb = Embedding(table)
    ids = torch.randint(0, 300, (4, 33), device=DEV)
    want = {dt: emb(ids, out_dtype=dt) for dt in (None, torch.bfloat16, torch.float16)}
    record = pkg.install._gather_embedding(Embedding, pkg.dequant.GGQUnsupported)
    try:
        before = Embedding.calls
        for dt, ref in want.items():
            got = emb(ids, out_dtype=dt)
            assert got.dtype == ref.dtype and torch.equal(got, ref)
        assert Embedding.calls == before                                         # the row kernel served all of them
        assert len(asserted) == 3 and all("outside the embedding table" in m for m in asserted)      # ... each with the id check F.embedding implies
        pkg.dequant.dequantize_rows(table, ids, torch.float16)                   # a direct call of the library function: the kernel clamps, no check
        assert len(asserted) == 3
        emb.max_norm = 1.0                                                       # renormalising lookups need the dense table
        emb(ids, out_dtype=torch.float16)
        assert Embedding.calls == before + 1
        emb.max_norm = None
        table.patches = [("lora", "key")]
        emb(ids, out_dtype=torch.float16)
        assert Embedding.calls == before + 2
    finally:
        owner, name, fn = record
        setattr(owner, name, fn)
    assert Embedding.forward_ggml_cast_weights is original
