"""lookahead.DequantAhead on an MI355X: the unpack of the next layers rides in the launch of the one that was asked for
(include/ggq.h ggq_dequant_batch).  Same kernels, so every result must equal the per-layer call bit for bit -- against the oracle
too -- whatever the prediction did: hit, changed order, another dtype, a weight written to in between."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mk(pkg, q, shape, seed):
    packed = pkg.synth.make_tensor_bytes(q, shape, seed=seed, mode="signed")
    return pkg.ops.GGMLTensor(torch.from_numpy(packed).to(DEV), tensor_type=q, tensor_shape=shape), packed


def _bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16).reshape(-1)


def test_batch_entry_point_equals_single_calls(pkg):
    """ggq_dequant_batch through raw ctypes: 11 tensors of 4 (format, mode) groups, more than 8 of one group (two launches of it), an empty one."""
    import ctypes
    Q, nat = pkg.qtypes.Q, pkg._native
    spec = [(Q.Q4_K, (40, 512), torch.float16, None)] * 9 + [(Q.Q5_K, (3, 256), torch.bfloat16, None), (Q.Q8_0, (33, 96), torch.float32, torch.float32),
                                                              (Q.Q4_K, (8, 256), torch.bfloat16, torch.bfloat16)]
    kinds = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32", None: "f16"}
    descs = (nat.ggq_desc * (len(spec) + 1))()
    keep, outs, wants = [], [], []
    for i, (q, shape, dt, cd) in enumerate(spec):
        t, packed = _mk(pkg, q, shape, 50 + i)
        out = torch.empty(shape, dtype=dt, device=DEV)
        bs, ts = pkg.qtypes.block_geometry(q)
        descs[i] = nat.ggq_desc(int(q), pkg.dequant._OUT_CODE[dt], t.data_ptr(), out.data_ptr(), t.numel() // ts, pkg.dequant._COMPUTE_CODE[cd], 0)
        keep.append(t); outs.append(out)
        wants.append(oracle.dequant_tensor(q, packed, kinds[cd], kinds[dt]))
    descs[len(spec)] = nat.ggq_desc(int(Q.Q4_K), 0, None, None, 0, 0, 0)                       # nothing to do: allowed
    rc = nat.lib().ggq_dequant_batch(descs, len(spec) + 1, torch.cuda.current_stream().cuda_stream)
    assert rc == nat.GGQ_OK
    torch.cuda.synchronize()
    for (q, shape, dt, cd), out, want in zip(spec, outs, wants):
        got = out.view(torch.int32 if dt is torch.float32 else torch.int16).cpu().numpy().reshape(-1)
        assert np.array_equal(got.view(want.dtype if want.dtype != np.float32 else np.uint32), want.view(np.uint32) if want.dtype == np.float32 else want), (q, dt, cd)
    bad = (nat.ggq_desc * 1)(nat.ggq_desc(99, 0, keep[0].data_ptr(), outs[0].data_ptr(), 1, 0, 0))
    assert nat.lib().ggq_dequant_batch(bad, 1, None) == nat.GGQ_ERR_QTYPE
    assert nat.lib().ggq_dequant_batch(descs, nat.BATCH_MAX + 1, None) == nat.GGQ_ERR_ARG


def test_lookahead_chain_is_bit_identical(pkg):
    Q = pkg.qtypes.Q
    spec = [(Q.Q4_K, (96, 512)), (Q.Q5_K, (64, 1024)), (Q.Q4_K, (3072, 3072)), (Q.Q8_0, (33, 96)), (Q.Q4_K, (40, 256)), (Q.Q6_K, (7, 256)),
            (Q.Q4_K, (512, 768)), (Q.Q4_0, (5, 32)), (Q.Q4_K, (1, 256)), (Q.IQ4_XS, (9, 256)), (Q.Q4_K, (24, 512))]
    ws, wants = [], []
    for i, (q, shape) in enumerate(spec):
        t, packed = _mk(pkg, q, shape, 200 + i)
        ws.append(t)
        wants.append(oracle.cast_f16_to_bf16_bits(oracle.dequant_f16(q, packed)))
    ahead = pkg.lookahead.DequantAhead(4, pkg.dequant.dequantize_tensor)
    for p in range(4):
        for w, want in zip(ws, wants):
            got = ahead(w, torch.bfloat16)
            assert got.dtype == torch.bfloat16 and tuple(got.shape) == tuple(w.tensor_shape)
            assert np.array_equal(_bits(got), want), (p, w.tensor_type)
    st = ahead.stats()
    assert st["hits"] > 2 * len(ws) and st["launches"] < 11 + 3 * 4 + 3 and st["stale_dropped"] == 0
    # another order: whatever was unpacked ahead for the old order is dropped or handed out correctly -- values never change
    for w, want in list(zip(ws, wants))[::-1] + list(zip(ws, wants))[::2]:
        assert np.array_equal(_bits(ahead(w, torch.bfloat16)), want)
    # another dtype / arithmetic than predicted, and a weight that is written to between the prediction and its call
    for w, (q, shape) in zip(ws, spec):
        ahead(w, torch.bfloat16)
    want16 = oracle.dequant_f16(spec[1][0], ws[1].cpu().numpy()).view(np.uint16)
    ahead(ws[0], torch.bfloat16)                                                     # predicts ws[1] in bf16
    assert np.array_equal(_bits(ahead(ws[1], torch.float16)), want16)               # asked for in fp16: recomputed
    ahead(ws[2], torch.bfloat16)                                                     # predicts ws[3], ws[4], ws[5]
    ws[3].view(torch.uint8)[3:40] ^= 0x5A                                            # in-place write into the packed bytes (version bump)
    torch.cuda.synchronize()
    new_want = oracle.cast_f16_to_bf16_bits(oracle.dequant_f16(spec[3][0], ws[3].cpu().numpy()))
    assert np.array_equal(_bits(ahead(ws[3], torch.bfloat16)), new_want) and ahead.stats()["stale_dropped"] >= 1
    # results are FRESH tensors nobody else holds: writing into one (the LoRA branch patches in place) changes nothing later
    a = ahead(ws[6], torch.bfloat16)
    a.zero_()
    for w in ws[7:] + ws[:6]:
        ahead(w, torch.bfloat16)
    assert np.array_equal(_bits(ahead(ws[6], torch.bfloat16)), wants[6])
    # a side stream: predictions made on one stream are not handed to a call on another
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for w, want in zip(ws, wants):
            if w is not ws[3]:
                assert np.array_equal(_bits(ahead(w, torch.bfloat16)), want)
    side.synchronize()


def test_lookahead_under_the_reference_layers(pkg, monkeypatch):
    """install(lookahead=4) over the reference's own GGMLOps.Linear (reference ops.py executed verbatim): a chain of its layers, one of
    them LoRA-patched, gives the same outputs as without the option, pass after pass."""
    from oracle import reference
    if not reference.available():
        pytest.skip("reference sources neither live nor staged")
    import ref_harness as H
    mods = reference.load_reference_package("ggq_reflook", setitem=monkeypatch.setitem)
    ro, Q = mods["ops"], pkg.qtypes.Q
    dev = torch.device(DEV)
    chain = [H.make_linear(ro, pkg, Q.Q4_K, 96, 512, dev, seed=1)[0], H.make_linear(ro, pkg, Q.Q6_K, 40, 768, dev, seed=2, bias=False)[0],
             H.make_linear(ro, pkg, Q.Q5_0, 64, 512, dev, seed=3, patches=H.lora_patch((64, 512), seed=4))[0],
             H.make_linear(ro, pkg, Q.Q8_0, 48, 1024, dev, seed=5, dequant_dtype="target")[0], H.make_linear(ro, pkg, Q.IQ4_NL, 32, 512, dev, seed=6)[0]]
    xs = [torch.randn(17, lin.in_features, device=dev, dtype=torch.bfloat16, generator=torch.Generator(device=dev).manual_seed(i)) for i, lin in enumerate(chain)]
    want = [lin(x) for lin, x in zip(chain, xs)]
    with H.Installed(pkg, mods, lookahead=4):
        for _ in range(4):
            for lin, x, w in zip(chain, xs, want):
                got = lin(x)
                assert type(got) is torch.Tensor and torch.equal(got, w)
        st = pkg.install.lookahead_stats(mods["dequant"])
        sb = pkg.install.scratch_bytes(mods["dequant"])
    assert st["hits"] >= 6 and st["stale_dropped"] == 0 and sb["total"] == sb["lookahead"] >= 0
