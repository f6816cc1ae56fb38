"""INTEGRATION.md section 3 shows the ctypes patch a maintainer would paste into the reference's dequant.py (dequant.py:30-44).  This test EXECUTES
that very text: the fenced block is cut out of the document, exec'ed into a fresh verbatim load of the reference's ``dequant`` module with
GGQ_HIP_LIB pointing at the in-tree library, and the patched module is then held against the oracle for all 12 formats x the three arithmetic
modes, plus the fall-through cases (CPU tensor, a qtype without a kernel) -- so the snippet cannot rot unnoticed (Missing #5)."""
import importlib.util
import os
import re

import numpy as np
import pytest
import torch

import oracle
from oracle import reference
import ref_harness as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def integration_snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. The reference-side binding"):]
    sec = sec[:sec.index("\n## 4.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1, "INTEGRATION.md section 3 holds exactly one python block"
    return blocks[0]


def test_the_documented_ctypes_patch_runs_and_matches_the_oracle(pkg, monkeypatch):
    if not reference.available():
        pytest.skip("reference sources not present (neither /root/reference nor oracle/_ref)")
    monkeypatch.setenv("GGQ_HIP_LIB", pkg._native.LIB_PATH)
    reference.ensure_gguf()
    spec = importlib.util.spec_from_file_location("ggq_reference_dequant_doc", os.path.join(reference.REFERENCE_DIR, "dequant.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                    # the reference, verbatim
    torch_path = mod.dequantize
    exec(compile(integration_snippet(), "INTEGRATION.md#3", "exec"), mod.__dict__)   # ... and the documented patch on top of it
    assert mod._ggq is not None and mod.dequantize is not torch_path and mod._dequantize_torch is torch_path
    calls = []
    real = mod._ggq.ggq_dequant

    class Counted:                                                  # ctypes function pointers take no attributes: wrap the library object instead
        def __getattr__(self, name):
            return getattr(mod_lib, name)

        def ggq_dequant(self, *a):
            calls.append(a[0])
            return real(*a)
    mod_lib = mod._ggq
    mod._ggq = Counted()
    dev = torch.device("cuda:0")
    Q = pkg.qtypes.Q
    kinds = {None: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}
    for q in pkg.qtypes.HIP_QTYPES:
        bs, ts = pkg.qtypes.block_geometry(q)
        n = 1003 if bs == 32 else 131
        blocks = pkg.synth.make_blocks(q, n, seed=int(q) + 50, mode="signed")
        data = torch.from_numpy(blocks.reshape(-1).copy()).to(dev)
        for dtype, kind in kinds.items():
            before = len(calls)
            got = mod.dequantize(data, q, (n, bs), dtype=dtype)
            assert len(calls) == before + 1 and calls[-1] == int(q), (q.name, kind, "the patch did not take the HIP path")
            want = oracle.dequant_tensor(q, blocks, kind, kind)
            u = np.uint32 if kind == "f32" else np.uint16
            bits = got.cpu().view(torch.int32 if kind == "f32" else torch.int16).numpy().reshape(-1).view(u)
            assert got.shape == (n, bs) and np.array_equal(bits, want.view(u)), (q.name, kind)
        # ... and the reference's own caller above it: dequantize_tensor -> (patched) dequantize -> .to(dtype)   (dequant.py:15-23)
        t = pkg.ops.GGMLTensor(data, tensor_type=q, tensor_shape=(n, bs))
        got = mod.dequantize_tensor(t, torch.bfloat16, None)
        assert H.same_bits(got, H.oracle_tensor(q, blocks, torch.bfloat16, None, (n, bs))), q.name
    # fall-through: CPU-resident bytes (load time), and a qtype the library has no kernel for, keep the reference's eager path
    n_before = len(calls)
    blocks = pkg.synth.make_blocks(Q.Q4_K, 9, seed=3, mode="signed")
    cpu = torch.from_numpy(blocks.reshape(-1).copy())
    got = mod.dequantize(cpu, Q.Q4_K, (9, 256))
    assert got.device.type == "cpu" and np.array_equal(got.view(torch.int16).numpy().reshape(-1).view(np.uint16), oracle.dequant_f16(Q.Q4_K, blocks).view(np.uint16))
    bf = torch.randn(64).to(torch.bfloat16)
    got = mod.dequantize(bf.view(torch.uint8).to(dev), Q.BF16, (64,))              # BF16 "blocks": dequant.py:61-62, not a ggq kernel
    assert torch.equal(got.cpu(), bf.float())
    assert len(calls) == n_before
