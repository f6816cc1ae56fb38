"""The N-DEVICE code, for the day a box with several GPUs runs ``pytest -m gpu`` (Next #3).  On the one-GPU test boxes every
test here is collected and SKIPPED ("needs >= 2 GPUs"); on a multi-GPU box they are the first thing that ever executes:

  * cross-device ``ShardedPlan.place`` copies, ``ggq_plan_create`` / ``ggq_plan_launch`` under a non-zero / non-current device;
  * every per-layer entry point (``dequantize_tensor``, ``dequantize_rows``, ``ggq_linear_small``, ``ggq_linear_mfma``, the side-stream
    prefetcher) on ``cuda:1`` while ``cuda:0`` is the current device -- the per-device caches of csrc/ggq_linear.hip and dequant._DEVICE_OK;
  * ``gguf_sd_loader(devices=[...])`` over DISTINCT devices and ``state_dict_sharded_plan``;
  * ``bench.py --gpus 2`` through REAL RCCL fences (``world.backend == "nccl"``, two distinct devices) and ``--inproc-gpus 2`` on two devices;
  * the positive side of the two one-GPU-only tests of test_gpu_multirank.py (RCCL simply works; two ranks on two devices are accepted).

Everything is compared with the oracle, like the single-device tests."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import plan_check

N_GPUS = torch.cuda.device_count() if torch.cuda.is_available() else 0
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(N_GPUS < 2, reason=f"needs >= 2 GPUs (this box has {N_GPUS})")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _items(pkg, manifest, seed0, device="cuda:0"):
    out = []
    for i, (_, q, shape) in enumerate(manifest):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        out.append((pkg.synth.device_blocks(q, n_blocks, torch.device(device), seed0 + i), q, shape))
    return out


@pytest.mark.parametrize("current", [0, 1])
def test_sharded_plan_over_two_devices_vs_oracle(pkg, current):
    """place() moves every shard's packed bytes to its device; the plans are created and launched whatever device is CURRENT."""
    manifest = pkg.manifests.sd35_t5("Q4_K_M")[:30] + pkg.manifests.flux_dev("Q4_K_M")[:10]
    items = _items(pkg, manifest, 9100)                                       # everything starts on cuda:0
    with torch.cuda.device(current):
        plan = pkg.grouped.ShardedPlan.place(items, ["cuda:0", "cuda:1"], out_dtype=torch.bfloat16)
        assert [str(d) for d in plan.devices] == ["cuda:0", "cuda:1"] and plan.streams is None
        assert plan.indices == pkg.sharding.partition(manifest, 2)
        assert all(p.device == d and all(t.device == d for t in p._keep) and all(o.device == d for o in p.outputs) for p, d in zip(plan.plans, plan.devices))
        for _ in range(2):
            plan.launch()
        assert torch.cuda.current_device() == current                         # launch() restores the caller's device
        plan.synchronize()
    outs = plan.outputs_in_order()
    n, bad = plan_check.check_plan([it[0] for it in items], [q for _, q, _ in manifest], outs)
    assert n == len(manifest) and not bad, bad[:3]
    plan.close()


def test_plan_launch_refuses_the_wrong_current_device_at_the_c_abi(pkg):
    """ggq_plan_launch (include/ggq.h): the tables live on the plan's device -- the C entry answers GGQ_ERR_ARG under another current device;
    DequantPlan.launch switches for the call, so the Python side never sees it."""
    nat = pkg._native
    items = _items(pkg, pkg.manifests.flux_linear_pool(pkg.qtypes.Q.Q4_K, 1), 9200, "cuda:1")
    plan = pkg.grouped.DequantPlan(items)
    assert plan.device == torch.device("cuda:1")
    with torch.cuda.device(0):
        assert nat.lib().ggq_plan_launch(plan._plan, torch.cuda.current_stream(torch.device("cuda:1")).cuda_stream) == nat.GGQ_ERR_ARG
        plan.launch()
    torch.cuda.synchronize(1)
    n, bad = plan_check.check_plan([it[0] for it in items], [it[1] for it in items], plan.outputs)
    assert not bad
    plan.close()


def test_per_layer_entry_points_on_the_second_device_while_the_first_is_current(pkg):
    """dequantize_tensor / dequantize_rows / linear_small / linear_mfma with tensors on cuda:1 and cuda:0 current: the kernels run on cuda:1's
    current stream, the per-device property caches (csrc/ggq_linear.hip device_props, dequant._DEVICE_OK) get their SECOND entry."""
    Q, T = pkg.qtypes.Q, pkg.ops.GGMLTensor
    dev = torch.device("cuda:1")
    assert torch.cuda.current_device() == 0
    for q in (Q.Q4_K, Q.Q8_0, Q.Q6_K):
        bs, ts = pkg.qtypes.block_geometry(q)
        rows, cols = 96, 1024
        blocks = pkg.synth.make_blocks(q, rows * cols // bs, seed=int(q))
        w = T(torch.from_numpy(blocks.reshape(-1).copy()).to(dev), tensor_type=q, tensor_shape=(rows, cols))
        for kind, dt in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
            got = pkg.dequant.dequantize_tensor(w, dt)
            assert got.device == dev
            want = oracle.dequant_tensor(q, blocks, "f16", kind)
            bits = got.cpu().view(torch.int32 if kind == "f32" else torch.int16).numpy().reshape(-1)
            assert np.array_equal(bits.view(np.uint32 if kind == "f32" else np.uint16), want.view(np.uint32 if kind == "f32" else np.uint16)), (q.name, kind)
        w16 = torch.from_numpy(oracle.dequant_f16(q, blocks).reshape(rows, cols).copy()).to(dev)
        ids = torch.tensor([[0, rows - 1, 5, 5]], device=dev)
        assert torch.equal(pkg.dequant.dequantize_rows(w, ids, torch.float16), w16[ids])
        for m, fn in ((1, pkg.fused.linear_small), (4, pkg.fused.linear_small), (48, pkg.fused.linear_mfma)):
            x = torch.randn(m, cols, device=dev, dtype=torch.bfloat16)
            y = fn(x, w)
            assert y.device == dev
            wd = w16.to(torch.bfloat16).double()
            ref = x.double() @ wd.T
            tol = cols * 2.0 ** -24 * (x.double().abs() @ wd.abs().T) + 2.0 ** -8 * ref.abs() + 1e-30
            assert bool(((y.double() - ref).abs() <= tol).all()), (q.name, m)
    assert torch.cuda.current_device() == 0
    assert pkg.dequant._DEVICE_OK.get(1)


def test_low_vram_prefetcher_on_the_second_device(pkg):
    """overlap.LayerPrefetcher (install(overlap=True)) with CPU-resident packed weights and inputs on cuda:1, cuda:0 current: its streams, events
    and scratch slots are per device (ggq_overlap_create on the input's device)."""
    Q = pkg.qtypes.Q
    dev = torch.device("cuda:1")
    layers, refs = [], []
    for i in range(4):
        rows, cols = 256, 1024
        blocks = pkg.synth.make_blocks(Q.Q4_K, rows * cols // 256, seed=70 + i)
        layers.append(pkg.ops.GGMLLinear(pkg.ops.GGMLTensor(torch.from_numpy(blocks.reshape(-1).copy()), tensor_type=Q.Q4_K, tensor_shape=(rows, cols))))
        refs.append(torch.from_numpy(oracle.dequant_f16(Q.Q4_K, blocks).reshape(rows, cols).copy()).to(dev))
    record, pf = pkg.overlap.attach(pkg.ops.GGMLLayer)
    try:
        x = torch.randn(8, 1024, device=dev, dtype=torch.float16)
        for _ in range(3):                                                       # pass 1 learns the order, passes 2-3 run prefetched
            for lin, w in zip(layers, refs):
                assert torch.equal(lin(x), torch.nn.functional.linear(x, w))
        assert pf.stats()["hits"] > 0
    finally:
        owner, name, fn = record
        setattr(owner, name, fn)
        pf.close()
    assert torch.cuda.current_device() == 0


def test_loader_places_shards_on_two_devices(pkg, tmp_path):
    """gguf_sd_loader(devices=["cuda:0", "cuda:1"]): ONE parse, one partition, every shard uploaded to its device, ONE state dict in the
    file's order; state_dict_sharded_plan dequantizes it on both devices; state_dict_plan refuses (its return shape is for one device)."""
    from test_gpu_gguf import _mixed_file
    path, spec, packed = _mixed_file(pkg, tmp_path)
    sd = pkg.loader.gguf_sd_loader(path, devices=["cuda:0", "cuda:1"])
    whole = pkg.loader.gguf_sd_loader(path)                                     # the reference's behaviour: CPU views, file order
    assert list(sd) == list(whole)
    on = {str(v.device) for v in sd.values()}
    assert on == {"cuda:0", "cuda:1"}
    for k in sd:
        assert torch.equal(torch.Tensor(sd[k]).cpu(), torch.Tensor(whole[k])), k
    marks = [k for k, v in sd.items() if getattr(v, "is_largest_weight", False)]
    assert marks == [k for k, v in whole.items() if getattr(v, "is_largest_weight", False)] and len(marks) == 1
    with pytest.raises(ValueError):
        pkg.loader.state_dict_plan(sd)
    plan, keys = pkg.loader.state_dict_sharded_plan(sd, dtype=torch.float16)
    assert len(plan.plans) == 2 and {str(d) for d in plan.devices} == on
    plan.launch()
    plan.synchronize()
    pre = "model.diffusion_model."
    for k, o in zip(keys, plan.outputs_in_order()):
        assert o.device == sd[k].device
        want = oracle.dequant_f16(sd[k].tensor_type, packed[pre + k])
        assert np.array_equal(o.cpu().numpy().reshape(-1).view(np.uint16), want.view(np.uint16)), k
    plan.close()
    # one process per GPU, rank 1 of 2 on cuda:1: the same partition
    part = pkg.loader.gguf_sd_loader(path, device="cuda:1", shard=(1, 2))
    assert part and all(str(v.device) == "cuda:1" and str(sd[k].device) == "cuda:1" for k, v in part.items())


def _torchrun(nproc, args, env_extra, timeout=1500):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("GGQ_BENCH_BACKEND",):
        if k not in env_extra:
            env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("backend", [None, "try-nccl"])
def test_bench_two_ranks_on_two_devices_through_rccl(backend):
    """`bench.py --gpus 2` exactly as the driver launches it, NO rig (and once with the rig's RCCL attempt, which here simply succeeds): the
    fences go through RCCL, the two ranks sit on two distinct devices, every rank's outputs are checked, and the aggregate is about twice one
    device's rate (weak scaling, nothing shared but the host)."""
    args = ["bench.py", "--gpus", "2", "--steps", "8", "--warmup", "2", "--regions", "3", "--no-per-qtype", "--no-per-mode", "--cpu-seconds", "2", "--no-workloads", "--pairs", "16"]
    proc = _torchrun(2, args, {"GGQ_BENCH_BACKEND": backend} if backend else {})
    assert proc.returncode == 0, proc.stderr[-3000:]
    (line,) = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    w = line["world"]
    assert w["backend"] == "nccl" and w["size"] == 2 and w["distinct_devices"] == 2, w
    assert len({r["uuid"] for r in w["ranks"]}) == 2 and len({r["pid"] for r in w["ranks"]}) == 2
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["shard_cover"].startswith("disjoint, complete")
    assert line["cpu_baseline"]["parity_vs_gpu"].startswith("bit-exact (2 x 32 = 64 tensors on 2 ranks")
    one = subprocess.run([sys.executable, "bench.py", "--steps", "8", "--warmup", "2", "--regions", "3", "--no-per-qtype", "--no-per-mode", "--cpu-seconds", "0",
                          "--no-workloads", "--pairs", "16", "--no-ceiling"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    (base,) = [json.loads(ln) for ln in one.stdout.splitlines() if ln.startswith("{")]
    assert 1.7 <= line["value"] / base["value"] <= 2.2, (base["value"], line["value"])


@pytest.mark.timeout(1200)
def test_bench_sharded_weight_set_two_ranks_on_two_devices():
    """configs[4] sharded over two real devices: strong scaling, disjoint cover, per-rank parity."""
    args = ["bench.py", "--gpus", "2", "--workload", "sd35-t5", "--steps", "6", "--warmup", "2", "--regions", "3", "--cpu-seconds", "2"]
    proc = _torchrun(2, args, {})
    assert proc.returncode == 0, proc.stderr[-3000:]
    (line,) = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert line["world"]["backend"] == "nccl" and line["world"]["distinct_devices"] == 2
    assert line["scaling"] == "strong" and sum(s["tensors"] for s in line["config"]["shards"]) == 549
    assert line["cpu_baseline"]["parity_vs_gpu"].startswith("bit-exact") and "on 2 ranks" in line["cpu_baseline"]["parity_vs_gpu"]


@pytest.mark.timeout(900)
def test_bench_inproc_two_devices_line():
    """`bench.py --inproc-gpus 2`: ONE process, two DISTINCT devices (no rig): the shards' launches overlap, the step takes about one shard's time."""
    env = dict(os.environ)
    env.pop("GGQ_BENCH_BACKEND", None)
    proc = subprocess.run([sys.executable, "bench.py", "--inproc-gpus", "2", "--workload", "sd35-t5", "--steps", "4", "--warmup", "1"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=800)
    assert proc.returncode == 0, proc.stderr[-2000:]
    (line,) = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    shards = line["config"]["shards"]
    assert line["n_gpus"] == 2 and len(shards) == 2 and len({s["device"] for s in shards}) == 2
    assert sum(s["tensors"] for s in shards) == 549
    assert line["cpu_baseline"]["parity_vs_gpu"].startswith("bit-exact (549 tensors on 2 shards")
    assert line["ms_per_step"] <= 1.35 * max(s["gpu_ms_per_step"] for s in shards)          # overlapped, not serialised
