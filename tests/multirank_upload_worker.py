"""Worker of tests/test_gpu_multirank.py: one of N real processes (torch.distributed.run, gloo) that loads ITS shard of a .gguf
file onto the GPU (gguf_sd_loader(..., shard=(rank, world))), dequantizes it there and reports, per tensor, the sha256 of the
packed bytes it holds and of the dense fp16 result.  No tensor data crosses between the ranks: the partition is derived from the
file's tensor table alone."""
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402


def main():
    path, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_package()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    sd = pkg.loader.gguf_sd_loader(path, device=dev, shard=(rank, world))
    report = {}
    for k, v in sd.items():
        packed = torch.Tensor(v).cpu().contiguous().view(torch.uint8).numpy().tobytes()
        ent = {"packed": hashlib.sha256(packed).hexdigest(), "qtype": int(getattr(v, "tensor_type", -1) or 0)}
        if pkg.dequant.is_quantized(v):
            dense = pkg.dequant.dequantize_tensor(v, torch.float16)
            ent["dense"] = hashlib.sha256(dense.cpu().contiguous().view(torch.int16).numpy().tobytes()).hexdigest()
        report[k] = ent
    torch.cuda.synchronize()
    dist.barrier()
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(report, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
