#!/usr/bin/env python3
"""The hot path in its calling context: every quantized linear of FLUX.1-dev (304 layers, Q4_K_M mix, synthetic weights)
run the way GGMLOps.Linear.forward does (reference ops.py:242-244) -- dequantize the weight, F.linear, drop it -- for one
denoising step's worth of tokens, against the same F.linear calls on weights dequantized once up front.

    python tools/flux_forward_emulation.py [--tokens 4608] [--dtype bfloat16] [--reps 5] [--dense-cache-gb N] [--fused-small-m] [--fused-mfma MAX_M] [--overlap] [--lowvram]

Prints one JSON line: ms per emulated step with on-the-fly dequant, with resident dense weights, and the difference
(the cost of the dequant path per step).  Layers run back to back on one stream; img/txt token counts are not modelled
separately (every layer sees --tokens rows), modulation layers see 1 row (they act on the conditioning vector).
--dense-cache-gb / --fused-small-m / --overlap switch on the opt-ins of install() for the quantized pass (resident.DenseCache,
fused.linear_small, overlap.LayerPrefetcher); --lowvram keeps the packed weights on the CPU (the reference's low-VRAM mode:
``s.weight.to(device)`` per layer per forward, ops.py:209)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=4608)        # 4096 image + 512 text tokens at 1024x1024
    ap.add_argument("--model", default="flux", choices=["flux", "sd35", "t5"], help="which weight set's linears: FLUX.1-dev (configs[3]); SD3.5-large MMDiT or the "
                    "T5-xxl encoder (the two halves of configs[4]; T5's token_embd table is an embedding lookup, not a linear, and is left out)")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--mix", default="Q4_K_M")
    ap.add_argument("--dense-cache-gb", type=float, default=0.0, help="opt-in resident.DenseCache budget (0 = off, the reference's behaviour)")
    ap.add_argument("--fused-small-m", action="store_true", help="opt-in fused dequantize + linear for the 1-row (modulation) layers")
    ap.add_argument("--fused-mfma", type=int, default=0, metavar="MAX_M", help="opt-in fused dequantize + GEMM on the matrix cores for inputs of up to MAX_M rows")
    ap.add_argument("--overlap", action="store_true", help="opt-in side-stream prefetch of the next layer's weight (overlap.LayerPrefetcher)")
    ap.add_argument("--graph", action="store_true", help="also time both steps replayed from a captured HIP graph: GPU time without the host's issue rate")
    ap.add_argument("--lowvram", action="store_true", help="packed weights live on the CPU and are copied per layer per forward (ops.py:209)")
    args = ap.parse_args()
    pkg = load_package()
    pkg.ops.GGMLLinear.fuse_small_m = args.fused_small_m
    pkg.ops.GGMLLinear.fuse_mfma_max_m = args.fused_mfma
    dev = torch.device("cuda:0")
    dtype = getattr(torch, args.dtype)
    manifest = {"flux": pkg.manifests.flux_dev, "sd35": pkg.manifests.sd35_large, "t5": pkg.manifests.t5_xxl_encoder}[args.model](args.mix)
    manifest = [e for e in manifest if e[0] != "token_embd.weight"]
    label = {"flux": "FLUX.1-dev", "sd35": "SD3.5-large", "t5": "T5-xxl encoder"}[args.model]
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    layers, inputs = [], {}
    for name, q, (rows, cols) in manifest:
        bs, ts = pkg.qtypes.block_geometry(q)
        n_blocks = rows * cols // bs
        data = torch.randint(0, 256, (n_blocks, ts), dtype=torch.uint8, device=dev, generator=g)
        for off in pkg.qtypes.SCALE_FIELDS[q]:                     # small scales: keep activations finite
            vals = (torch.rand(n_blocks, device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
            data[:, off:off + 2] = vals.view(torch.uint8).reshape(n_blocks, 2)
        w = pkg.ops.GGMLTensor(data.reshape(-1).cpu() if args.lowvram else data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols))
        m = 1 if ("mod" in name or "adaLN" in name) else args.tokens      # modulation layers act on the conditioning vector: one row
        if (m, cols) not in inputs:
            inputs[(m, cols)] = torch.randn(m, cols, device=dev, dtype=dtype) * 0.05
        layers.append((pkg.ops.GGMLLinear(w), inputs[(m, cols)]))

    cache = None
    if args.dense_cache_gb:
        cache = pkg.resident.DenseCache(args.dense_cache_gb * 1e9, pkg.dequant.dequantize_tensor)
        pkg.ops.GGMLLayer._dequantize = staticmethod(cache)


    prefetcher = None
    if args.overlap:
        _, prefetcher = pkg.overlap.attach(pkg.ops.GGMLLayer, resident=True)

    def step_quantized():
        for lin, x in layers:
            lin(x)

    dense = [pkg.dequant.dequantize_tensor(lin.weight.to(dev), dtype) for lin, _ in layers]

    def step_dense():
        for (lin, x), w in zip(layers, dense):
            torch.nn.functional.linear(x, w)

    def timed(fn):
        fn()
        fn()                                                   # the prefetcher learns the layer order in the first pass
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    q_med, q_min = timed(step_quantized)
    d_med, d_min = timed(step_dense)
    graph_ms = None
    if args.graph and not args.lowvram and not args.overlap:
        graph_ms = {}
        for name, fn in (("on_the_fly", step_quantized), ("dense", step_dense)):
            side = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                fn()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=side):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            graph_ms[name] = round(a.elapsed_time(b) / args.reps, 2)
            del g
    flops = sum(2.0 * x.shape[0] * lin.weight.shape[0] * lin.weight.shape[1] for lin, x in layers)
    n_el = sum(lin.weight.shape[0] * lin.weight.shape[1] for lin, _ in layers)
    print(json.dumps({
        "workload": f"{label} linears ({len(layers)} layers, {args.mix}, {n_el / 1e9:.2f} G weights), {args.tokens} tokens, {args.dtype}",
        "ms_per_step_dequant_on_the_fly": round(q_med, 2), "ms_per_step_dense_resident": round(d_med, 2),
        "dequant_cost_ms_per_step": round(q_med - d_med, 2), "dequant_share_of_step_pct": round(100 * (q_med - d_med) / q_med, 1),
        "best_ms": {"on_the_fly": round(q_min, 2), "dense": round(d_min, 2)},
        "graph_replay_ms_per_step": graph_ms,
        "gemm_TFLOPs_dense": round(flops / d_med / 1e9, 1),
        "dense_cache": cache.stats() if cache is not None else None,
        "overlap": prefetcher.stats() if prefetcher is not None else None, "lowvram": args.lowvram,
        "dense_weight_GB": round(n_el * 2 / 1e9, 1), "packed_weight_GB": round(sum(lin.weight.numel() for lin, _ in layers) / 1e9, 2)}))


if __name__ == "__main__":
    main()
