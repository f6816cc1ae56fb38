#!/usr/bin/env python3
"""Where the host-side microseconds of one dequantize_tensor() call go (the per-layer hot loop, reference ops.py:177): the
whole call, and its parts timed in isolation -- the output allocation, the C-ABI call through ctypes / through the C-API
module, the HIP launch itself (a kernel over 0 groups cannot be launched, so: the smallest tensor), attribute reads on the
GGMLTensor subclass, the torch-function guard.  Prints one JSON line (us per call, median of 7 runs of 2000 calls)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402


def per_call(fn, n=2000, runs=7):
    ts = []
    for _ in range(runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ts.append((time.perf_counter() - t0) / n * 1e6)
        torch.cuda.synchronize()
    ts.sort()
    return round(ts[len(ts) // 2], 3)


def main():
    pkg = load_package()
    dq, nat, Q = pkg.dequant, pkg._native, pkg.qtypes.Q
    dev = torch.device("cuda:0")
    q = Q.Q4_K
    small = pkg.ops.GGMLTensor(torch.randint(0, 256, (144 * 16,), dtype=torch.uint8, device=dev), tensor_type=q, tensor_shape=(16, 256))
    layer = pkg.ops.GGMLTensor(torch.randint(0, 256, (144 * 36864,), dtype=torch.uint8, device=dev), tensor_type=q, tensor_shape=(3072, 3072))
    out_small = torch.empty((16, 256), dtype=torch.bfloat16, device=dev)
    lib = nat.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    res["dequantize_tensor(small tensor, bf16)"] = per_call(lambda: dq.dequantize_tensor(small, torch.bfloat16))
    res["dequantize_tensor(3072x3072, bf16) host side"] = per_call(lambda: dq.dequantize_tensor(layer, torch.bfloat16), n=500)
    res["torch.empty((3072,3072), bf16, device)"] = per_call(lambda: torch.empty((3072, 3072), dtype=torch.bfloat16, device=dev))
    p, o = small.data_ptr(), out_small.data_ptr()
    res["ctypes ggq_dequant (launch included)"] = per_call(lambda: lib.ggq_dequant(12, p, 16, o, 0, 1, stream))
    res["ctypes ggq_supported (no launch)"] = per_call(lambda: lib.ggq_supported(12))
    fast = getattr(dq, "_fast", None)
    if fast is not None:
        res["C-API module dequant (launch included)"] = per_call(lambda: fast.dequant(12, p, 16, o, 0, 1, stream))
    res["_raw_stream + _cur_device"] = per_call(lambda: (dq._raw_stream(0), dq._cur_device()))
    res["with DisableTorchFunctionSubclass (enter/exit)"] = per_call(lambda: dq._NoTorchFunction().__enter__())
    res["subclass attribute reads (tensor_type, tensor_shape)"] = per_call(lambda: (small.tensor_type, small.tensor_shape))
    def probes():
        with dq._NoTorchFunction():
            return small.is_cuda, small.dtype, small.is_contiguous(), small.data_ptr(), small.numel()
    res["5 tensor probes under the guard"] = per_call(probes)
    res["torch.compiler.is_compiling()"] = per_call(dq._is_compiling)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
