#!/usr/bin/env python3
"""Per-layer call latency of the hot path as ComfyUI drives it: one dequantize_tensor() per quantized
layer per forward (reference ops.py:177), tensors of FLUX.1-dev shape B (3072x3072), rotating over a
pool larger than the Infinity Cache.  Prints host-side us/call (enqueue cost), end-to-end us/call of the eager loop (host-bound
for the small shape: the GPU idles between kernels, each one starts cold) and `gpu_bound_us_per_call`: the same launches replayed
from a captured HIP graph, back to back with no host in between -- what a layer's unpack costs inside a GPU-bound model step,
where its ramp overlaps the previous kernel's drain."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402


def main():
    pkg = load_package()
    qt = pkg.qtypes
    dev = torch.device("cuda:0")
    out = {}
    cols = int(os.environ.get("GGQ_LAYER_COLS", "3072"))       # 3072 = FLUX shape B, 12288 = shape C
    for qname in (sys.argv[1:] or ["Q4_K", "Q8_0", "Q6_K"]):
        q = qt.Q[qname]
        bs, ts = qt.block_geometry(q)
        n_blocks = 3072 * cols // bs
        pool = [pkg.ops.GGMLTensor(torch.randint(0, 256, (n_blocks * ts,), dtype=torch.uint8, device=dev), tensor_type=q, tensor_shape=(3072, cols))
                for _ in range(64 if cols <= 3072 else 24)]
        for dtype in (torch.float16, torch.bfloat16):
            for _ in range(2):
                for t in pool:
                    pkg.dequant.dequantize_tensor(t, dtype)
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                for t in pool:
                    pkg.dequant.dequantize_tensor(t, dtype)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            n = reps * len(pool)
            nbytes = qt.algorithmic_bytes(q, 3072 * cols)
            # GPU-bound: capture one pass over the pool into a graph (outputs are the graph's own buffers), replay it
            side = torch.cuda.Stream()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                keep = [pkg.dequant.dequantize_tensor(t, dtype) for t in pool]          # warm the allocator on the capture stream
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, stream=side):
                    keep = [pkg.dequant.dequantize_tensor(t, dtype) for t in pool]
            graph.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            t_graph = a.elapsed_time(b) * 1e-3
            del graph, keep
            out[f"{qname}->{str(dtype).split('.')[-1]}"] = {"host_us_per_call": round(t_host / n * 1e6, 2), "e2e_us_per_call": round(t_all / n * 1e6, 2),
                                                            "e2e_GBps": round(nbytes * n / t_all / 1e9, 1),
                                                            "gpu_bound_us_per_call": round(t_graph / n * 1e6, 2), "gpu_bound_GBps": round(nbytes * n / t_graph / 1e9, 1)}
    import json
    print(json.dumps(out))


if __name__ == "__main__":
    main()
