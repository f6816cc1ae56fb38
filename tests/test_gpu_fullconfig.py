"""BASELINE.json configs[3] and configs[4] at FULL size, every tensor checked (Next #1 / Weak #1).

``manifests.flux_dev("Q4_K_M")`` (304 tensors, 11.8 G elements, Q4_K + Q5_K) and ``manifests.sd35_t5("Q4_K_M")`` (549 tensors,
Q5_0 + Q4_K + Q6_K) are each built as ONE DequantPlan -- the launch bench.py times -- with fp16 and with bf16 output, and EVERY
output tensor is compared bit for bit with the oracle (oracle/plan_check.py: the whole tensor against the AVX2 leg, that leg
against the soft-float checker on three windows per tensor; bf16 through the reference's own ``.to(dtype)`` and the C cast).
The same tensors then go through the per-layer entry point the node calls (``dequantize_tensor``, reference ops.py:177), one
launch per tensor, and must equal the plan's output: the layer-sized launch shapes (TuneMid, XCD run mapping, > 65 536 groups)
see 21504x3072, 18432x3072, 3072x15360, 12288x3072, 9216x3072 (Q5_K), 7296x2432 / 9728x2432 (Q5_0) and 32128x4096 (Q6_K) here.
"""
import pytest
import torch

from oracle import plan_check

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(pkg, manifest, out_dtype, seed0):
    items = []
    for i, (_, q, shape) in enumerate(manifest):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        items.append((pkg.synth.device_blocks(q, n_blocks, DEV, seed0 + i, mode="signed"), q, shape))
    return pkg.grouped.DequantPlan(items, out_dtype=out_dtype)


@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("config", ["flux_dev", "sd35_t5"])
def test_every_tensor_of_the_full_weight_set(pkg, config, out_dtype):
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    manifest = getattr(pkg.manifests, config)("Q4_K_M")
    assert len(manifest) == {"flux_dev": 304, "sd35_t5": 549}[config]
    plan = _build(pkg, manifest, out_dtype, seed0=40_000)
    outs = plan.launch()
    torch.cuda.synchronize()
    qtypes = [q for _, q, _ in manifest]
    n, bad = plan_check.check_plan(plan._keep, qtypes, outs)
    assert n == len(manifest) and not bad, bad[:5]
    # the per-layer call (one launch per tensor, the shape picked per tensor size) returns the same bits: one tensor of every
    # distinct (format, shape) of the set
    seen = set()
    for (name, q, shape), packed, out in zip(manifest, plan._keep, outs):
        if (q, shape) in seen:
            continue
        seen.add((q, shape))
        t = pkg.ops.GGMLTensor(packed, tensor_type=q, tensor_shape=shape)
        single = pkg.dequant.dequantize_tensor(t, out_dtype)
        assert single.dtype == out_dtype and tuple(single.shape) == tuple(shape)
        assert torch.equal(single.view(torch.int16), out.view(torch.int16)), (name, q.name, shape)
    assert len(seen) >= 8
    plan.close()
