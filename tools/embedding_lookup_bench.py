#!/usr/bin/env python3
"""The embedding caller (reference ops.py:251-260): dequantize the whole table + F.embedding, against the row lookup that unpacks
only the rows the ids name (dequant.dequantize_rows).  GPU time per forward (HIP events) and the transient memory of each way."""
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
    cases = [("T5-xxl token table", "Q8_0", 32128, 4096, 512), ("T5-xxl token table", "Q4_K", 32128, 4096, 512),
             ("152k x 3584 LLM vocabulary", "Q4_K", 151936, 3584, 512), ("152k x 3584 LLM vocabulary", "Q6_K", 151936, 3584, 77)]
    for label, qname, n_rows, cols, n_ids in cases:
        q = pkg.qtypes.Q[qname]
        bs, ts = pkg.qtypes.block_geometry(q)
        nb = n_rows * cols // bs
        data = torch.randint(0, 256, (nb, ts), dtype=torch.uint8, device=dev)
        for off in pkg.qtypes.SCALE_FIELDS[q]:
            vals = (torch.rand(nb, device=dev) * 1e-3 + 1e-4).to(torch.float16)
            data[:, off:off + 2] = vals.view(torch.uint8).reshape(nb, 2)
        table = pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(n_rows, cols))
        emb = pkg.ops.GGMLEmbedding(table)
        ids = torch.randint(0, n_rows, (1, n_ids), device=dev)

        def timed(gather):
            emb.gather_rows = gather
            for _ in range(3):
                emb(ids, out_dtype=torch.bfloat16)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                y = emb(ids, out_dtype=torch.bfloat16)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / 20, (torch.cuda.max_memory_allocated() - base) / 1e6, y
        t_rows, m_rows, y1 = timed(True)
        t_full, m_full, y2 = timed(False)
        out[f"{label}, {qname}, {n_ids} ids"] = {"row_lookup_us": round(t_rows, 1), "whole_table_then_embedding_us": round(t_full, 1), "speedup": round(t_full / t_rows, 1),
                                                  "transient_MB": {"row_lookup": round(m_rows, 1), "whole_table": round(m_full, 1)}, "bit_identical": bool(torch.equal(y1, y2))}
        del table, emb, data
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
