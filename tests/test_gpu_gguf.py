"""GGUF file -> HBM streaming and the device-resident state dict, on an MI355X: the uploaded arena is
the file's data section byte for byte, and every tensor dequantized from it equals the oracle."""
import numpy as np
import pytest
import torch

import oracle
from gguf_writer import GGUFWriter

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mixed_file(pkg, tmp_path, n_rows=48):
    Q, synth = pkg.qtypes.Q, pkg.synth
    w = GGUFWriter(arch="flux")
    spec, packed = [], {}
    for i, q in enumerate(pkg.qtypes.HIP_QTYPES * 2):
        bs, _ = pkg.qtypes.block_geometry(q)
        shape = (n_rows + i, bs * (1 + i % 3))
        name = f"model.diffusion_model.blocks.{i}.weight"
        data = synth.make_tensor_bytes(q, shape, seed=500 + i, mode="signed")
        spec.append((name, q, shape))
        packed[name] = data
        w.add_tensor(name, q, tuple(reversed(shape)), data)
    w.add_tensor("model.diffusion_model.bias", Q.F32, (64,), np.linspace(-1, 1, 64, dtype=np.float32))
    return w.write(str(tmp_path / "mixed.gguf")), spec, packed


@pytest.mark.parametrize("threads,chunk", [(0, 0), (1, 4096), (3, 8192), (8, 1 << 20)])
def test_upload_is_the_data_section_byte_for_byte(pkg, tmp_path, threads, chunk):
    path, spec, packed = _mixed_file(pkg, tmp_path)
    with pkg.gguf_file.GGUFFile(path) as f:
        arena = f.upload(DEV, threads=threads, chunk_bytes=chunk)
        torch.cuda.synchronize()
        whole = np.fromfile(path, dtype=np.uint8)
        assert arena.dtype == torch.uint8 and arena.numel() == f.data_bytes
        assert np.array_equal(arena.cpu().numpy(), whole[f.data_offset:])
        for t in f.tensors:
            v = f.device_bytes(arena, t)
            assert v.data_ptr() % 16 == 0 and v.numel() == t.nbytes
    pkg._native.lib().ggq_gguf_upload_release()


def test_device_state_dict_dequantizes_to_the_oracle(pkg, tmp_path):
    path, spec, packed = _mixed_file(pkg, tmp_path)
    sd = pkg.loader.gguf_sd_loader(path, device=DEV)
    pre = "model.diffusion_model."
    assert sd["bias"].is_cuda and sd["bias"].dtype == torch.float32 and np.array_equal(torch.Tensor(sd["bias"]).cpu().numpy(), np.linspace(-1, 1, 64, dtype=np.float32))
    for name, q, shape in spec:
        t = sd[name[len(pre):]]
        assert t.is_cuda and t.tensor_type == q and t.shape == torch.Size(shape)
        got = pkg.dequant.dequantize_tensor(t, torch.float16)
        assert tuple(got.shape) == shape
        assert np.array_equal(got.view(torch.int16).cpu().numpy().reshape(-1).view(np.uint16), oracle.dequant_f16(q, packed[name]).view(np.uint16)), name
    # the whole weight set in one plan, bf16 out (the production dtype)
    plan, keys = pkg.loader.state_dict_plan(sd, dtype=torch.bfloat16)
    assert len(keys) == len(spec) and plan.kernels == len(pkg.qtypes.HIP_QTYPES)
    outs = plan.launch()
    torch.cuda.synchronize()
    for k, out in zip(keys, outs):
        name = pre + k
        q = sd[k].tensor_type
        want = oracle.dequant_tensor(q, packed[name], "f16", "bf16")
        assert np.array_equal(out.view(torch.int16).cpu().numpy().reshape(-1).view(np.uint16), want), name
    plan.close()
    # the same file loaded the reference's way (CPU mmap views) holds the same bytes
    sd_cpu = pkg.loader.gguf_sd_loader(path)
    for k in keys:
        assert torch.equal(torch.Tensor(sd_cpu[k]), torch.Tensor(sd[k]).cpu())


def test_sharded_upload_one_process_per_gpu(pkg, tmp_path):
    """shard=(rank, world): every rank computes the same tensor-list partition from the file's table alone and uploads
    only its own tensors (coalesced runs, one arena); the shards are disjoint, cover the file, and hold the file's bytes."""
    path, spec, packed = _mixed_file(pkg, tmp_path)
    pre = "model.diffusion_model."
    world, seen, total_arena = 3, {}, 0
    for rank in range(world):
        sd = pkg.loader.gguf_sd_loader(path, device=DEV, shard=(rank, world))
        assert sd, rank
        for k, v in sd.items():
            assert k not in seen
            seen[k] = rank
            assert v.is_cuda and v.data_ptr() % 16 == 0
            name = pre + k
            if name in packed:
                assert np.array_equal(torch.Tensor(v).cpu().numpy().reshape(-1), packed[name]), k
                got = pkg.dequant.dequantize_tensor(v, torch.float16)
                assert np.array_equal(got.view(torch.int16).cpu().numpy().reshape(-1).view(np.uint16), oracle.dequant_f16(v.tensor_type, packed[name]).view(np.uint16))
    assert set(seen) == {n[len(pre):] for n, _, _ in spec} | {"bias"}
    assert len(set(seen.values())) == world                                   # every rank got work
    with pytest.raises(ValueError, match="device"):
        pkg.loader.gguf_sd_loader(path, shard=(0, 2))
