#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own dequant.py (verbatim, from
/root/reference, under the gguf stub of oracle/reference.py) on seeded packed blocks.

TEST INFRASTRUCTURE.  Run in the build container (the only place /root/reference exists):

    python oracle/make_golden.py

Outputs (committed):
  tests/golden/<QTYPE>.npz        packed input blocks + the reference's outputs
      blocks        uint8  (n_blocks, type_size)   4 equal runs: nominal | signed | adversarial | raw
      out_f16       uint16 fp16 bits of dequantize(data, qtype, oshape, dtype=None)      dequant.py:30
      sub           int64  indices of the blocks that also carry the other modes below
      out_f32       float32 bits (uint32) of dequantize(..., dtype=torch.float32) on blocks[sub]
      out_bf16      bf16 bits (uint16)    of dequantize(..., dtype=torch.bfloat16) on blocks[sub]
      tensor_bf16   bf16 bits of dequantize_tensor(GGMLTensor-like, dtype=bfloat16)  on blocks[sub]   dequant.py:15-23
  tests/golden/BF16.npz           dequantize_blocks_BF16 (dequant.py:61-62)
  tests/golden/large_hashes.json  sha256 of the reference fp16 output for large seeded inputs that are
                                  too big to commit (config 1: Q8_0 4096x4096; every format at 3072x3072);
                                  the inputs are regenerated from comfyui-gguf_amd/synth.py by seed.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import reference  # noqa: E402
from ggq_pkg import load_package  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PER_MODE = {32: 128, 256: 32}          # blocks per mode, by block size
MODES = ("nominal", "signed", "adversarial", "raw")
LARGE = [("Q8_0", (4096, 4096), 0)]    # (qtype, shape, seed) -- config 1 of BASELINE.json


class _FakeGGML:
    """What dequantize_tensor reads off a GGMLTensor (dequant.py:16-17,23): attrs + .data."""

    def __init__(self, data, tensor_type, tensor_shape):
        self.data, self.tensor_type, self.tensor_shape = data, tensor_type, tensor_shape
        self.shape = tensor_shape
        self.device = data.device


def main(out_dir=GOLDEN, large=True):
    """``large=False`` skips the hashes of the big inputs (tests regenerate the small fixtures to check the committed ones)."""
    pkg = load_package()
    qt, synth = pkg.qtypes, pkg.synth
    ref = reference.load_reference_dequant()
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(0)

    for q in qt.HIP_QTYPES:
        bs, ts = qt.block_geometry(q)
        per = PER_MODE[bs]
        blocks = np.concatenate([synth.make_blocks(q, per, seed=1000 + i, mode=m) for i, m in enumerate(MODES)])
        n = blocks.shape[0]
        data = torch.from_numpy(blocks.reshape(-1).copy())
        out = ref.dequantize(data, q, (n, bs))
        assert out.dtype == torch.float16 and tuple(out.shape) == (n, bs)
        sub = np.concatenate([np.arange(i * per, i * per + per // 4) for i in range(len(MODES))])
        sdata = torch.from_numpy(blocks[sub].reshape(-1).copy())
        o32 = ref.dequantize(sdata, q, (len(sub), bs), dtype=torch.float32)
        obf = ref.dequantize(sdata, q, (len(sub), bs), dtype=torch.bfloat16)
        tbf = ref.dequantize_tensor(_FakeGGML(sdata, q, torch.Size((len(sub), bs))), dtype=torch.bfloat16)
        assert o32.dtype == torch.float32 and obf.dtype == torch.bfloat16 and tbf.dtype == torch.bfloat16
        np.savez_compressed(
            os.path.join(out_dir, f"{q.name}.npz"),
            blocks=blocks,
            out_f16=out.view(torch.int16).numpy().view(np.uint16),
            sub=sub,
            out_f32=o32.view(torch.int32).numpy().view(np.uint32),
            out_bf16=obf.view(torch.int16).numpy().view(np.uint16),
            tensor_bf16=tbf.view(torch.int16).numpy().view(np.uint16),
        )
        print(f"{q.name}: {n} blocks, {n * bs} elements")

    raw = np.random.default_rng(77).integers(0, 256, size=2 * 4096, dtype=np.uint8)
    o = ref.dequantize(torch.from_numpy(raw.copy()), qt.Q.BF16, (4096,))
    assert o.dtype == torch.float32
    np.savez_compressed(os.path.join(out_dir, "BF16.npz"), blocks=raw, out_f32=o.view(torch.int32).numpy().view(np.uint32))

    if not large:
        return
    hashes = {}
    large = LARGE + [(q.name, (3072, 3072), 1 if q in qt.LEGACY_QTYPES else 2) for q in qt.HIP_QTYPES]
    for name, shape, seed in large:
        q = qt.Q[name]
        packed = synth.make_tensor_bytes(q, shape, seed=seed, mode="nominal")
        out = ref.dequantize(torch.from_numpy(packed), q, shape)
        assert out.dtype == torch.float16 and not torch.isnan(out).any()
        key = f"{name}:{shape[0]}x{shape[1]}:seed{seed}:nominal"
        hashes[key] = {
            "packed_sha256": hashlib.sha256(packed.tobytes()).hexdigest(),
            "out_f16_sha256": hashlib.sha256(out.numpy().tobytes()).hexdigest(),
            "n_elements": int(np.prod(shape)),
        }
        print(key, hashes[key]["out_f16_sha256"][:16])
    with open(os.path.join(out_dir, "large_hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
