import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggq_pkg import load_package
pkg = load_package(); dev = torch.device("cuda:0"); q = pkg.qtypes.Q.Q4_K; bs, ts = pkg.qtypes.block_geometry(q)
g = torch.Generator(device=dev).manual_seed(0)
rows, cols = 12288, 3072
pool = []
for i in range(6):
    data = torch.randint(0, 256, (rows * cols // bs, ts), dtype=torch.uint8, device=dev, generator=g)
    for off in pkg.qtypes.SCALE_FIELDS[q]:
        vals = (torch.rand(data.shape[0], device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
        data[:, off:off + 2] = vals.view(torch.uint8).reshape(-1, 2)
    pool.append(pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols)))
for m, t in ((32, 32), (128, 64), (512, 128), (512, 64)):
    x = torch.randn(m, cols, device=dev, dtype=torch.bfloat16) * 0.05
    for w in pool:
        pkg.fused.linear_mfma(x, w, tile_rows=t)
    torch.cuda.synchronize()
