"""torch.compile through the reference's layers with install() underneath, WITHOUT a GPU: the custom ops get stub CPU kernels (zeros of the right shape), so
Dynamo can trace and run the graph here and the properties that do not depend on the kernels' arithmetic are checked on every CPU run:
  * graph-break count 0 -- rounds 2-5 broke the graph once per layer at ``as_subclass(torch.Tensor)`` (Dynamo cannot trace it); VERDICT round 5, Next #3;
  * the DEFAULT install puts ``ggq::linear_small`` / ``ggq::linear_mfma`` / ``ggq::dequantize_rows`` into the graph (it used to stand aside and leave the
    exact path's ``ggq::dequantize`` + F.linear), ``exact`` keeps ``ggq::dequantize``;
  * layers the fused kernels decline while tracing (LoRA patches, too many rows) trace the reference's method.
The bit-for-bit comparison compiled == eager runs on the GPU (tests/test_gpu_reference.py)."""
import pytest
import torch

from oracle import reference
import ref_harness as H

pytestmark = pytest.mark.skipif(not reference.available(), reason="reference sources not present")


@pytest.fixture(scope="module")
def rig(pkg):
    import sys
    added, missing = [], object()

    def setitem(mapping, key, value):                     # remember what sys.modules held: everything is put back at teardown (the drop-in tests scan sys.modules)
        added.append((mapping, key, mapping.get(key, missing)))
        mapping[key] = value
    mods = reference.load_reference_package(name="ggq_refpkg_trace", setitem=setitem)
    D, F = pkg.dequant, pkg.fused
    if D._dequantize_op is None or F._linear_small_op is None:
        pytest.skip("torch without torch.library.custom_op")
    try:                                                   # stub CPU kernels: shape/dtype of the fake implementation, zeros
        D._dequantize_op.register_kernel("cpu")(lambda data, qtype, compute, out: D._dequantize_op_fake(data, qtype, compute, out).zero_())
        D._dequantize_rows_op.register_kernel("cpu")(lambda p, i, q, n, c, cd, o, ck: D._dequantize_rows_op_fake(p, i, q, n, c, cd, o, ck).zero_())
        F._linear_small_op.register_kernel("cpu")(lambda x, p, b, q, r, c: F._linear_fake(x, p, b, q, r, c).zero_())
        F._linear_mfma_op.register_kernel("cpu")(lambda x, p, b, q, r, c, t: F._linear_fake(x, p, b, q, r, c).zero_())
    except Exception as e:                                 # noqa: BLE001
        pytest.skip(f"cannot register stub kernels: {e}")
    old = (F._TRACE_ANY_DEVICE, D._TRACE_ANY_DEVICE)
    F._TRACE_ANY_DEVICE = D._TRACE_ANY_DEVICE = True
    yield mods
    F._TRACE_ANY_DEVICE, D._TRACE_ANY_DEVICE = old
    for mapping, key, prev in reversed(added):
        if prev is missing:
            mapping.pop(key, None)
        else:
            mapping[key] = prev


def _explain(fn, *args):
    torch._dynamo.reset()
    try:
        ex = torch._dynamo.explain(fn)(*args)
    except Exception as e:                                 # noqa: BLE001 -- Dynamo's support for the reference's subclass is the reference's business
        pytest.skip(f"torch.compile cannot trace the reference's GGMLTensor on this torch: {type(e).__name__}: {str(e)[:200]}")
    finally:
        torch._dynamo.reset()
    ops = [str(n.target) for g in ex.graphs for n in g.graph.nodes if n.op == "call_function"]
    return ex, ops


@pytest.mark.parametrize("m", [1, 2, 64])
@pytest.mark.parametrize("options,expect", [({}, "ggq.dequantize.default"), ({"fast": True}, None)], ids=["exact", "default"])
def test_compiled_linear_has_no_graph_break_and_the_right_op(pkg, rig, m, options, expect):
    ro, Q = rig["ops"], pkg.qtypes.Q
    lin, _ = H.make_linear(ro, pkg, Q.Q4_K, 32, 512, "cpu", seed=41)
    x = torch.randn(m, 512, dtype=torch.float16)
    with H.Installed(pkg, rig, **options):
        ex, ops = _explain(lambda t: lin(t), x)
    assert ex.graph_break_count == 0, [str(r)[:300] for r in ex.break_reasons]
    want = expect or ("ggq.linear_small.default" if m == 1 else "ggq.linear_mfma.default")
    assert want in ops, ops
    if expect is None:
        assert "ggq.dequantize.default" not in ops, ops        # the fused op replaced unpack + F.linear, it did not join them


def test_compiled_embedding_and_declined_layers(pkg, rig):
    ro, Q = rig["ops"], pkg.qtypes.Q
    emb, _ = H.make_embedding(ro, pkg, Q.Q6_K, 64, 512, "cpu", seed=5)
    patched, _ = H.make_linear(ro, pkg, Q.Q4_K, 32, 512, "cpu", seed=42, patches=H.lora_patch((32, 512), 7))
    big, _ = H.make_linear(ro, pkg, Q.Q4_K, 32, 512, "cpu", seed=43)
    ids = torch.tensor([[0, 63, 7, 7, 12]])
    x, x300 = torch.randn(4, 512, dtype=torch.float16), torch.randn(300, 512, dtype=torch.float16)
    with H.Installed(pkg, rig, fast=True):
        ex, ops = _explain(lambda i: emb(i, out_dtype=torch.float16), ids)
        assert ex.graph_break_count == 0 and "ggq.dequantize_rows.default" in ops, (ops, ex.break_reasons)
        ex, ops = _explain(lambda a: big(a), x300)                                       # 300 rows: above what the default fuses
        assert ex.graph_break_count == 0 and "ggq.dequantize.default" in ops and not any("ggq.linear" in o for o in ops), ops
        ex, ops = _explain(lambda a: patched(a), x)                                      # LoRA-patched weight: the reference's get_weight
        assert "ggq.dequantize.default" in ops and not any("ggq.linear" in o for o in ops), ops
