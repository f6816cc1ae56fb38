"""Fused dequantize + linear for 1..4 input rows (include/ggq.h ggq_linear_small, opt-in) on an MI355X.

This is a floating-point CONTRACTION, so -- unlike the dequant kernels -- parity cannot be bit-exact: any two GEMV kernels
differ by the order of their fp32 additions.  What is checked:
  * against an fp64 evaluation of the same op on the ORACLE's weights (the reference's values, cast to the activation dtype as
    dequantize_tensor does): |y - y_ref| <= cols * 2^-24 * sum|w x|  +  eps(dtype) * |y_ref|  -- a worst-case fp32 accumulation
    bound plus the one final rounding;
  * that torch's own dequantize-then-F.linear (the drop-in default) satisfies the same bound, i.e. both are the same op.
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ALL = ["Q4_0", "Q4_1", "Q5_0", "Q5_1", "Q8_0", "Q2_K", "Q3_K", "Q4_K", "Q5_K", "Q6_K", "IQ4_NL", "IQ4_XS"]
DT = {"f16": (torch.float16, 2.0 ** -11), "bf16": (torch.bfloat16, 2.0 ** -8), "f32": (torch.float32, 2.0 ** -24)}


def _dense_weight(q, blocks, kind, rows, cols):
    w = oracle.dequant_tensor(q, blocks, "f16", kind)
    if kind == "f32":
        return w.astype(np.float64).reshape(rows, cols)
    if kind == "f16":
        return w.view(np.float16).astype(np.float64).reshape(rows, cols)
    return (w.astype(np.uint32) << 16).view(np.float32).astype(np.float64).reshape(rows, cols)


def _check(y, x, w64, bias, eps, cols):
    x64 = x.detach().to(torch.float64).cpu().numpy()
    ref = x64 @ w64.T
    bound = (np.abs(x64) @ np.abs(w64).T) * cols * 2.0 ** -24
    if bias is not None:
        b64 = bias.detach().to(torch.float64).cpu().numpy()
        ref = ref + b64
        bound = bound + np.abs(b64) * 2.0 ** -23
    err = np.abs(y.detach().to(torch.float64).cpu().numpy() - ref)
    tol = bound + eps * np.abs(ref) + 1e-30
    assert np.all(err <= tol), float((err / tol).max())


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("kind", ["f16", "bf16", "f32"])
def test_fused_linear_against_fp64_reference(pkg, name, kind):
    q = pkg.qtypes.Q[name]
    bs, _ = pkg.qtypes.block_geometry(q)
    dtype, eps = DT[kind]
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    for rows, cols, m, with_bias in ((203, 3072, 1, True), (17, 256, 4, False), (1, 1024, 2, True), (1031, 512, 3, False)):
        blocks = pkg.synth.make_tensor_bytes(q, (rows, cols), seed=rows + cols, mode="signed")
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        x = (torch.randn(m, cols, device=DEV, generator=g) * 0.5).to(dtype)
        bias = (torch.randn(rows, device=DEV, generator=g) * 0.01).to(dtype) if with_bias else None
        y = pkg.fused.linear_small(x, w, bias)
        assert y.shape == (m, rows) and y.dtype == dtype
        w64 = _dense_weight(q, blocks, kind, rows, cols)
        _check(y, x, w64, bias, eps, cols)
        # the drop-in default (dequantize, then torch's GEMM / GEMV) is the same op within the same bound
        y2 = torch.nn.functional.linear(x, pkg.dequant.dequantize_tensor(w, dtype), bias)
        _check(y2, x, w64, bias, eps, cols)


def test_fused_linear_in_the_layer_and_its_limits(pkg):
    Q, ops = pkg.qtypes.Q, pkg.ops
    blocks = pkg.synth.make_tensor_bytes(Q.Q4_K, (96, 3072), seed=3)
    w = ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=Q.Q4_K, tensor_shape=(96, 3072))
    bias_blocks = pkg.synth.make_tensor_bytes(Q.Q8_0, (96,), seed=4)
    qbias = ops.GGMLTensor(torch.from_numpy(bias_blocks).to(DEV), tensor_type=Q.Q8_0, tensor_shape=(96,))
    lin = ops.GGMLLinear(w, qbias)
    x = torch.randn(2, 1, 3072, device=DEV, dtype=torch.bfloat16)               # (batch, 1, hidden): the modulation input
    ref = lin(x)                                                                # default: dequantize + F.linear
    lin.fuse_small_m = True
    y = lin(x)
    assert y.shape == ref.shape == (2, 1, 96) and y.dtype == torch.bfloat16
    assert torch.allclose(y.float(), ref.float(), rtol=2 ** -6, atol=1e-3)
    xs = x.transpose(0, 1)                                                      # non-contiguous view: handled by a copy
    assert torch.equal(lin(xs).transpose(0, 1), y)
    many = torch.randn(5, 3072, device=DEV, dtype=torch.bfloat16)               # five rows: the layer falls back by itself
    assert torch.equal(lin(many), torch.nn.functional.linear(many, pkg.dequant.dequantize_tensor(w, torch.bfloat16), pkg.dequant.dequantize_tensor(qbias, torch.bfloat16)))
    with pytest.raises(pkg.dequant.GGQUnsupported):
        pkg.fused.linear_small(many, w)
    with pytest.raises(pkg.dequant.GGQUnsupported):                             # dequant_dtype modes keep the two-step path
        pkg.fused.linear_small(x, w, None, torch.float32)
    wide = ops.GGMLTensor(torch.zeros(2 * 65536 // 32 * 34, dtype=torch.uint8, device=DEV), tensor_type=Q.Q8_0, tensor_shape=(2, 65536))
    with pytest.raises(pkg.dequant.GGQUnsupported):                             # a row of 69632 packed bytes does not fit the LDS staging
        pkg.fused.linear_small(torch.zeros(1, 65536, device=DEV, dtype=torch.float16), wide)


def test_install_fused_small_m_wraps_the_linear_forward(pkg):
    """install(..., fused_small_m=True) wraps ``GGMLOps.Linear.forward_ggml_cast_weights`` (reference ops.py:242-244).  The
    reference is not on the GPU box, so the wrapper is driven on a class with that method's shape: small inputs take the fused
    kernel, larger ones and LoRA-patched weights the wrapped method, and uninstalling restores it."""
    Q, ops = pkg.qtypes.Q, pkg.ops

    class Linear(ops.GGMLLinear):
        calls = 0

        def forward_ggml_cast_weights(self, input):
            type(self).calls += 1
            weight, bias = self.cast_bias_weight(input)
            return torch.nn.functional.linear(input, weight, bias)

        def forward(self, input):
            return self.forward_ggml_cast_weights(input)

    original = Linear.forward_ggml_cast_weights
    blocks = pkg.synth.make_tensor_bytes(Q.Q6_K, (64, 1024), seed=8)
    w = ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=Q.Q6_K, tensor_shape=(64, 1024))
    lin = Linear(w)
    x = torch.randn(1, 1024, device=DEV, dtype=torch.float16)
    ref = lin(x)
    record = pkg.install._fuse_linear(Linear, pkg.dequant.GGQUnsupported, True, 0)
    try:
        before = Linear.calls
        y = lin(x)
        assert Linear.calls == before                                            # the fused kernel served it
        assert torch.allclose(y.float(), ref.float(), rtol=2 ** -9, atol=1e-3)
        big = torch.randn(8, 1024, device=DEV, dtype=torch.float16)
        assert torch.equal(lin(big), torch.nn.functional.linear(big, pkg.dequant.dequantize_tensor(w, torch.float16)))
        assert Linear.calls == before + 1
        w.patches = [("lora", "key")]                                            # patched weight: needs the dense tensor
        lin(x)
        assert Linear.calls == before + 2
    finally:
        owner, name, fn = record
        setattr(owner, name, fn)
    assert Linear.forward_ggml_cast_weights is original


@pytest.mark.parametrize("name", ["Q4_K", "Q5_K", "Q8_0", "Q3_K"])
@pytest.mark.parametrize("kind", ["f16", "bf16", "f32"])
def test_fused_linear_exact_arithmetic_is_bit_equal(pkg, name, kind):
    """Scales 2^-8 and integer activations: every weight, product and fp32 partial sum is exact in any order, so the kernel must return THE
    correctly rounded value -- any weight that is not the reference's (a scale pair taken from the wrong sub-block by the Q4_K / Q5_K
    lane exchange, say) shows as a wrong multiple of 2^-8.  Row lengths cover a single partial pass of the exchange (256 columns = 32
    chunks), whole passes (8192 = 4 passes) and ragged ones (2304 = 1 pass + half a trip; 3072 = 1.5 passes)."""
    q = pkg.qtypes.Q[name]
    dtype, _ = DT[kind]
    g = torch.Generator(device=DEV).manual_seed(11)
    bs, ts = pkg.qtypes.block_geometry(q)
    long_row = 8192 if 8192 // bs * ts <= 6000 else 4096                        # a row's packed bytes must fit a wave's LDS slice (Q8_0: 4096 columns)
    for rows, cols, m in ((33, 256, 1), (50, 2304, 2), (41, 3072, 4), (9, long_row, 3)):
        blocks = pkg.synth.make_blocks(q, rows * cols // bs, seed=cols + m, mode="raw")
        for off in pkg.qtypes.SCALE_FIELDS[q]:
            blocks[:, off], blocks[:, off + 1] = 0x00, 0x1C                       # fp16 0x1C00 = 2^-8
        blocks = blocks.reshape(-1)
        w = pkg.ops.GGMLTensor(torch.from_numpy(blocks).to(DEV), tensor_type=q, tensor_shape=(rows, cols))
        w64 = _dense_weight(q, blocks, kind, rows, cols)
        assert np.all(w64 * 2.0 ** 8 == np.round(w64 * 2.0 ** 8))
        x = torch.randint(-2, 3, (m, cols), device=DEV, generator=g).to(dtype)
        x64 = x.double().cpu().numpy()
        assert (np.abs(x64) @ np.abs(w64).T).max() < 2.0 ** 16                   # every partial sum a multiple of 2^-8 below 2^16: exact in fp32
        want = torch.from_numpy(x64 @ w64.T).to(dtype)
        got = pkg.fused.linear_small(x, w)
        assert torch.equal(got.cpu(), want), (name, kind, rows, cols, m)
