#!/usr/bin/env python3
"""Load-time dequantization of a big CPU-resident table (reference loader.py:253-254,270,386,397: token_embd / mmproj): the
reference's own torch-CPU path against the GPU route of install(cpu_route_mb=...) -- upload the packed bytes, unpack on the
MI355X, copy the dense result back.  T5-xxl's 32128 x 4096 embedding table.  Prints one JSON line.  (The reference is imported
through oracle/reference.py -- measurement only; without it only the GPU route is timed.)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 2)


def main():
    pkg = load_package()
    out = {"table": "T5-xxl token_embd 32128 x 4096", "host": {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads()}, "ms": {}}
    from oracle import reference
    ref = reference.load_reference_dequant() if reference.available() else None
    for qname in ("Q6_K", "Q8_0", "Q4_K"):
        q = pkg.qtypes.Q[qname]
        packed = torch.from_numpy(pkg.synth.make_tensor_bytes(q, (32128, 4096), seed=1))
        t = pkg.ops.GGMLTensor(packed, tensor_type=q, tensor_shape=(32128, 4096))
        row = {"packed_MB": round(packed.numel() / 1e6, 1)}
        for name, dtype in (("fp16", torch.float16), ("fp32", torch.float32)):
            got = pkg.dequant.dequantize_tensor_via_gpu(t, dtype)                          # warm-up: allocator, pinned staging
            row[f"gpu_route_{name}"] = med(lambda: pkg.dequant.dequantize_tensor_via_gpu(t, dtype), 5)
            if ref is not None:
                want = ref.dequantize_tensor(t, dtype)
                row[f"reference_cpu_{name}"] = med(lambda: ref.dequantize_tensor(t, dtype), 3)
                row[f"identical_{name}"] = bool(torch.equal(got.view(torch.int16 if dtype is torch.float16 else torch.int32), want.view(torch.int16 if dtype is torch.float16 else torch.int32)))
        out["ms"][qname] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
