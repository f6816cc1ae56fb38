"""The N > 1 path with REAL processes on the one GPU the test box has: `bench.py --gpus 2` launched the
way the driver launches it (python -m torch.distributed.run, one process per rank), the ranks sharing device 0 and fencing
through gloo (GGQ_BENCH_BACKEND=gloo -- RCCL needs one device per rank); and the sharded file -> HBM upload from two processes.
What it proves: the launch contract, the sharding, the fences and the MAX-over-ranks reduction run end to end; with two ranks on
one device the aggregate must come out at about the single-rank rate (the ranks share the HBM), which is the only scaling
statement one GPU can make."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, script_args, timeout=1500, backend="gloo"):
    env = dict(os.environ, GGQ_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", GGQ_BENCH_RCCL_TIMEOUT_S="120")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return proc


def _bench(nproc, extra, cpu_seconds=0, workloads=False):
    args = ["bench.py", "--gpus", str(nproc), "--steps", "6", "--warmup", "2", "--regions", "3", "--no-per-qtype", "--no-per-mode", "--cpu-seconds", str(cpu_seconds)]
    args += ([] if workloads else ["--no-workloads"]) + extra
    if nproc == 1:
        proc = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stderr[-3000:]
    else:
        proc = _run(nproc, args)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]                     # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_pool_two_ranks_on_one_gpu():
    one = _bench(1, ["--pairs", "16"])
    two = _bench(2, ["--pairs", "16"])
    for line in (one, two):
        assert all(k in line for k in CONTRACT)
        assert line["unit"] == "GB/s" and line["scaling"] == "weak" and line["higher_is_better"] is True and line["dtype"] == "f16"
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["bytes_per_step_per_gpu"] == one["config"]["bytes_per_step_per_gpu"]        # weak scaling: same pool per rank
    assert len(two["config"]["timed_regions_ms_per_step"]) == 3
    # two ranks time-share one device: each needs about twice as long per step, the aggregate stays at the device's rate
    # (measured: aggregate x1.07 at 2 ranks, x1.04 at 4; step time x1.87 / x3.83 -- the bounds leave room for a noisy box, not for a broken reduction)
    ratio = two["value"] / one["value"]
    assert 0.75 <= ratio <= 1.15, (one["value"], two["value"])
    assert 1.5 <= two["ms_per_step"] / one["ms_per_step"] <= 2.7


@pytest.mark.timeout(900)
def test_bench_sharded_weight_set_two_ranks_on_one_gpu(pkg):
    one = _bench(1, ["--workload", "sd35-t5"])
    two = _bench(2, ["--workload", "sd35-t5"])
    assert one["scaling"] == two["scaling"] == "strong" and two["n_gpus"] == 2
    assert two["config"]["bytes_per_step"] == one["config"]["bytes_per_step"]                         # the SAME tensor list, sharded
    assert 1.0 <= two["config"]["imbalance"] < 1.01
    # disjoint cover, computed the way every rank computes it (no exchange)
    manifest = pkg.manifests.sd35_t5("Q4_K_M")
    parts = pkg.sharding.partition(manifest, 2)
    assert sorted(i for p in parts for i in p) == list(range(len(manifest)))
    assert 0.75 <= two["value"] / one["value"] <= 1.15, (one["value"], two["value"])


def _self_proving(line, ranks, tensors_per_rank=None, total=None):
    """What an N-rank line must carry to count as measured (Next #1): N ranks observed, what each ran, parity from
    EVERY rank, a non-null CPU baseline."""
    w = line["world"]
    assert w["size"] == ranks and len(w["ranks"]) == ranks and sorted(r["rank"] for r in w["ranks"]) == list(range(ranks))
    assert w["backend"].startswith("gloo") and "test rig" in w["backend"]          # this box has one GPU: the rig, and the line says so
    assert len({r["pid"] for r in w["ranks"]}) == ranks                              # real processes
    shards = line["config"]["shards"]
    assert [s["rank"] for s in shards] == list(range(ranks)) and all(s["tensors"] > 0 and s["gpu_ms_per_step"] > 0 for s in shards)
    assert line["config"]["shard_cover"].startswith("disjoint, complete")
    if total is not None:
        assert sum(s["tensors"] for s in shards) == total
    # ms_per_step is the MAX over ranks of the reported region
    assert line["ms_per_step"] >= max(s["gpu_ms_per_step"] for s in shards) * 0.999
    cb = line["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["kind"] in ("reference", "port") and cb["cores"] >= 1
    assert cb["parity_vs_gpu"].startswith("bit-exact") and f"on {ranks} ranks" in cb["parity_vs_gpu"], cb["parity_vs_gpu"]
    by = cb["parity_by_rank"]
    assert [r["rank"] for r in by] == list(range(ranks)) and all(r["differ"] == 0 and r["tensors"] > 0 for r in by)
    assert [r["tensors"] for r in by] == [s["tensors"] for s in shards]            # every rank checked exactly what it ran
    if tensors_per_rank is not None:
        assert all(r["tensors"] == tensors_per_rank for r in by)


@pytest.mark.timeout(2400)
def test_bench_eight_ranks_default_line_is_self_proving(pkg):
    """`bench.py --gpus 8` the way the driver launches it -- 8 processes, here time-sharing the box's one GPU under the gloo rig: the
    weak-scaling headline AND the configs[3] / configs[4] strong-scaling sub-lines, every one with per-rank parity and a CPU baseline."""
    line = _bench(8, ["--pairs", "4"], cpu_seconds=4, workloads=True)
    assert all(k in line for k in CONTRACT) and line["n_gpus"] == 8 and line["scaling"] == "weak"
    _self_proving(line, 8, tensors_per_rank=8, total=64)
    for name, n_tensors in (("flux", 304), ("sd35-t5", 549)):
        sub = line["workloads"][name]
        assert sub["scaling"] == "strong" and sub["n_gpus"] == 8
        sub["world"] = line["world"]
        _self_proving(sub, 8, total=n_tensors)
        assert 1.0 <= sub["config"]["imbalance"] < 1.01
        manifest = getattr(pkg.manifests, "flux_dev" if name == "flux" else "sd35_t5")("Q4_K_M")
        assert sub["config"]["bytes_per_step"] == sum(pkg.sharding.tensor_cost(e) for e in manifest)
        assert sum(s["bytes"] for s in sub["config"]["shards"]) == sub["config"]["bytes_per_step"]
    assert "skipped" in line["workloads"]["per_layer"]


@pytest.mark.timeout(1800)
def test_bench_sd35_t5_eight_ranks_on_one_gpu():
    """`--workload sd35-t5 --gpus 8` = BASELINE configs[4] as a line of its own."""
    line = _bench(8, ["--workload", "sd35-t5"], cpu_seconds=4)
    assert all(k in line for k in CONTRACT) and line["n_gpus"] == 8 and line["scaling"] == "strong"
    _self_proving(line, 8, total=549)


@pytest.mark.timeout(900)
def test_rccl_failure_falls_back_to_gloo_fences_and_says_so():
    """GGQ_BENCH_BACKEND=try-nccl: the rig WITH the RCCL attempt.  Two ranks on this box's one GPU: RCCL refuses (duplicate device), every rank learns
    so over the gloo control group, the fences fall back to gloo and the line says why in world.backend -- the run does not die (Next #1e)."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs: RCCL simply works -- the positive case is tests/test_gpu_multidevice.py::test_bench_two_ranks_on_two_devices_through_rccl[try-nccl]")
    args = ["bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--regions", "3", "--no-per-qtype", "--no-per-mode", "--cpu-seconds", "2", "--no-workloads", "--pairs", "4"]
    proc = _run(2, args, timeout=800, backend="try-nccl")
    (line,) = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert line["world"]["backend"].startswith("gloo (RCCL group failed on"), line["world"]["backend"]
    assert line["n_gpus"] == 2 and line["cpu_baseline"]["parity_vs_gpu"].startswith("bit-exact (2 x 8 = 16 tensors on 2 ranks")


@pytest.mark.timeout(600)
def test_bench_refuses_n_ranks_on_fewer_devices():
    """Without the rig an N-GPU line must come from N devices: 2 ranks on this 1-GPU box are refused before anything is timed."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs: two ranks on two devices are accepted -- tests/test_gpu_multidevice.py::test_bench_two_ranks_on_two_devices_through_rccl[None]")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GGQ_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "2"]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=500)
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]          # no rate printed
    assert "one rank per GPU" in proc.stderr or "distinct device" in proc.stderr


@pytest.mark.timeout(600)
def test_sharded_upload_from_two_processes(pkg, tmp_path):
    from test_gpu_gguf import _mixed_file
    path, spec, packed = _mixed_file(pkg, tmp_path)
    out = tmp_path / "reports"
    out.mkdir()
    _run(2, [os.path.join("tests", "multirank_upload_worker.py"), path, str(out)])
    reports = [json.load(open(out / f"rank{r}.json")) for r in range(2)]
    assert reports[0] and reports[1] and not set(reports[0]) & set(reports[1])                       # every rank got work; disjoint
    pre = "model.diffusion_model."
    assert set(reports[0]) | set(reports[1]) == {n[len(pre):] for n, _, _ in spec} | {"bias"}         # together: the whole file
    for rep in reports:
        for k, ent in rep.items():
            if pre + k not in packed:
                continue
            data = packed[pre + k]
            assert ent["packed"] == hashlib.sha256(np.ascontiguousarray(data).tobytes()).hexdigest(), k
            q = pkg.qtypes.Q(ent["qtype"])
            assert ent["dense"] == hashlib.sha256(oracle.dequant_f16(q, data).view(np.uint16).tobytes()).hexdigest(), k
