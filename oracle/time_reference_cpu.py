#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/).  Time the REFERENCE's own torch-CPU dequant (dequant.py, imported verbatim from /root/reference) and
the C oracle on the same packed bytes, on THIS machine's host cores.  Only runs where /root/reference
exists (the build container, not the GPU box); the figure goes into DESIGN.md as context for bench.py's
`cpu_baseline` (kind "port"), which is the only CPU leg that can run beside the GPU."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # oracle/ lives at the repo root
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import reference  # noqa: E402
from ggq_pkg import load_package  # noqa: E402


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    pkg = load_package()
    ref = reference.load_reference_dequant()
    out = {"host": {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads()}}
    for name in (sys.argv[1:] or ["Q4_K", "Q8_0", "Q4_0", "Q6_K"]):
        q = pkg.qtypes.Q[name]
        shape = (3072, 3072)
        packed = pkg.synth.make_tensor_bytes(q, shape, seed=2)
        data = torch.from_numpy(packed)
        nbytes = pkg.qtypes.algorithmic_bytes(q, shape[0] * shape[1])
        want = ref.dequantize(data, q, shape)                       # warm-up + parity of the two CPU legs
        assert np.array_equal(want.numpy().view(np.uint16).reshape(-1), oracle.dequant_f16(q, packed).view(np.uint16))
        m_ref, b_ref = med(lambda: ref.dequantize(data, q, shape), 12)
        m_c, b_c = med(lambda: oracle.dequant_f16(q, packed), 12)
        out[name] = {"reference_torch_cpu_GBps": round(nbytes / m_ref / 1e9, 3), "reference_best_GBps": round(nbytes / b_ref / 1e9, 3),
                     "oracle_c_openmp_GBps": round(nbytes / m_c / 1e9, 3), "oracle_best_GBps": round(nbytes / b_c / 1e9, 3),
                     "reference_ms": round(m_ref * 1e3, 1), "oracle_ms": round(m_c * 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
