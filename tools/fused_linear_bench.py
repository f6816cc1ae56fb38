#!/usr/bin/env python3
"""Fused dequantize + linear (m <= 4) against dequantize-then-F.linear on FLUX's modulation shapes: GPU time per call
(HIP events over a rotating pool of layers, packed bytes >> Infinity Cache), and the packed-read rate of the fused kernel."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402


def main():
    pkg = load_package()
    dev = torch.device("cuda:0")
    out = {}
    for qname in (sys.argv[1:] or ["Q4_K", "Q8_0", "Q6_K"]):
        q = pkg.qtypes.Q[qname]
        bs, ts = pkg.qtypes.block_geometry(q)
        for rows, cols in ((18432, 3072), (9216, 3072)):
            n_layers = 40 if rows > 10000 else 80                      # ~1.3 GB packed for Q4_K
            g = torch.Generator(device=dev)
            g.manual_seed(1)
            layers = []
            for _ in range(n_layers):
                nb = rows * cols // bs
                data = torch.randint(0, 256, (nb, ts), dtype=torch.uint8, device=dev, generator=g)
                for off in pkg.qtypes.SCALE_FIELDS[q]:
                    vals = (torch.rand(nb, device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
                    data[:, off:off + 2] = vals.view(torch.uint8).reshape(nb, 2)
                layers.append(pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols)))
            for m in (1, 4):
                x = torch.randn(m, cols, device=dev, dtype=torch.bfloat16) * 0.05

                def timed(fn):
                    for w in layers[:4]:
                        fn(w)
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for w in layers:
                        fn(w)
                    b.record()
                    torch.cuda.synchronize()
                    return a.elapsed_time(b) * 1e3 / len(layers)
                def graphed(fn):
                    # GPU time without the host's issue rate: the same calls replayed from a captured HIP graph
                    side = torch.cuda.Stream()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(side):
                        keep = [fn(w) for w in layers]
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g, stream=side):
                            keep = [fn(w) for w in layers]
                    g.replay()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(5):
                        g.replay()
                    b.record()
                    torch.cuda.synchronize()
                    del keep
                    return a.elapsed_time(b) * 1e3 / (5 * len(layers))
                fused = timed(lambda w: pkg.fused.linear_small(x, w))
                fused_gpu = graphed(lambda w: pkg.fused.linear_small(x, w))
                dense = [pkg.dequant.dequantize_tensor(w, torch.bfloat16) for w in layers[:12]]
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    for wd in dense:
                        torch.nn.functional.linear(x, wd)
                b.record()
                torch.cuda.synchronize()
                dense_us = a.elapsed_time(b) * 1e3 / (3 * len(dense))
                del dense
                two = timed(lambda w: torch.nn.functional.linear(x, pkg.dequant.dequantize_tensor(w, torch.bfloat16)))
                packed = rows * cols // bs * ts
                out[f"{qname} {rows}x{cols} m={m}"] = {"fused_us": round(fused, 2), "dequant_plus_linear_us": round(two, 2), "speedup": round(two / fused, 2),
                                                        "fused_packed_read_GBps": round(packed / fused / 1e3, 1),
                                                        "fused_graph_replay_us": round(fused_gpu, 2), "fused_graph_replay_packed_GBps": round(packed / fused_gpu / 1e3, 1),
                                                        "linear_on_dense_resident_us": round(dense_us, 2)}
            del layers
            torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
