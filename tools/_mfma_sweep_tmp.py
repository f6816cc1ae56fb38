import json, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggq_pkg import load_package
pkg = load_package(); dev = torch.device("cuda:0"); q = pkg.qtypes.Q.Q4_K; bs, ts = pkg.qtypes.block_geometry(q)
g = torch.Generator(device=dev).manual_seed(0)
def pool_for(rows, cols, n):
    out = []
    for i in range(n):
        data = torch.randint(0, 256, (rows * cols // bs, ts), dtype=torch.uint8, device=dev, generator=g)
        for off in pkg.qtypes.SCALE_FIELDS[q]:
            vals = (torch.rand(data.shape[0], device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
            data[:, off:off + 2] = vals.view(torch.uint8).reshape(-1, 2)
        out.append(pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols)))
    return out
# GPU time per call: many calls queued back to back, events around; the queue is deep so host cost hides when the kernel is long
def gpu_us(fn, pool, reps=60):
    for w in pool[:2]: fn(w)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(pool[i % len(pool)])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
rows_out = []
for rows, cols in ((3072, 3072), (12288, 3072), (4096, 4096), (10240, 4096), (3072, 12288)):
    pool = pool_for(rows, cols, 12)
    for m in (128, 256, 512, 1024):
        x = torch.randn(m, cols, device=dev, dtype=torch.bfloat16) * 0.05
        r = {"weight": f"{rows}x{cols}", "m": m}
        r["default"] = round(gpu_us(lambda w: torch.nn.functional.linear(x, pkg.dequant.dequantize_tensor(w, torch.bfloat16)), pool), 1)
        dense = [pkg.dequant.dequantize_tensor(w, torch.bfloat16) for w in pool]
        r["dense"] = round(gpu_us(lambda w: torch.nn.functional.linear(x, w), dense), 1)
        del dense
        for t in (64, 128, -64, -128, -256):
            r[f"t{t}"] = round(gpu_us(lambda w: pkg.fused.linear_mfma(x, w, tile_rows=t), pool), 1)
        rows_out.append(r); print(json.dumps(r), flush=True)
