"""ONE process, several shards (Next #4; SURVEY.md section 7 step 6 / 8e: "one host thread + one stream per device").
The test box has one GPU, so the shards sit on the same device with a stream each -- the code path a multi-GPU ComfyUI process takes,
minus the second device: placement, one plan per shard, launches from the calling thread, outputs against the oracle."""
import numpy as np
import pytest
import torch

from oracle import plan_check

pytestmark = pytest.mark.gpu
# two DISTINCT devices whenever the box has them (tests/test_gpu_multidevice.py holds the tests that need them); on a one-GPU box the two shards
# share cuda:0 with a stream each
DEVICES = ["cuda:0", "cuda:1"] if torch.cuda.is_available() and torch.cuda.device_count() >= 2 else ["cuda:0", "cuda:0"]
SHARED = DEVICES[0] == DEVICES[1]


def _items(pkg, manifest, seed0):
    out = []
    for i, (_, q, shape) in enumerate(manifest):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        out.append((pkg.synth.device_blocks(q, n_blocks, torch.device("cuda:0"), seed0 + i), q, shape))
    return out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sharded_plan_two_shards_two_streams_vs_oracle(pkg, dtype):
    manifest = pkg.manifests.sd35_t5("Q4_K_M")[:40] + pkg.manifests.flux_dev("Q4_K_M")[:12]
    items = _items(pkg, manifest, 4100)
    plan = pkg.grouped.ShardedPlan.place(items, DEVICES, out_dtype=dtype)
    assert len(plan.plans) == 2 and (plan.streams is not None) == SHARED and (not SHARED or plan.streams[0] != plan.streams[1])
    assert plan.indices == pkg.sharding.partition(manifest, 2)
    assert plan.bytes == sum(pkg.sharding.tensor_cost(e) for e in manifest)
    for _ in range(3):
        plan.launch()
    # consumed on the CURRENT stream without any host-side wait: launch() made it depend on the shard streams
    sums = [o.view(torch.int16).to(torch.int64).sum() for o in plan.outputs_in_order()]
    plan.synchronize()
    outs = plan.outputs_in_order()
    assert [int(s) for s in sums] == [int(o.view(torch.int16).to(torch.int64).sum()) for o in outs]
    assert [tuple(o.shape) for o in outs] == [tuple(s) for _, _, s in manifest]
    n, bad = plan_check.check_plan([it[0] for it in items], [q for _, q, _ in manifest], outs)
    assert n == len(manifest) and not bad, bad[:3]
    # the same tensors through ONE plan: the same bits
    single = pkg.grouped.DequantPlan(items, out_dtype=dtype)
    single.launch()
    torch.cuda.synchronize()
    assert all(torch.equal(a.view(torch.int16), b.view(torch.int16).to(a.device)) for a, b in zip(single.outputs, outs))
    plan.close()
    single.close()


def test_sharded_plan_on_current_streams_and_errors(pkg):
    manifest = pkg.manifests.flux_linear_pool(pkg.qtypes.Q.Q5_K, 2)
    items = _items(pkg, manifest, 4300)
    parts = pkg.sharding.partition(manifest, 2)
    plan = pkg.grouped.ShardedPlan([("cuda:0", [items[i] for i in p]) for p in parts])       # shards handed over already placed
    assert plan.streams is None
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                                                              # "current stream" = whatever is current at launch
        plan.launch()
    side.synchronize()
    for p, shard in zip(parts, plan.outputs):
        n, bad = plan_check.check_plan([items[i][0] for i in p], [manifest[i][1] for i in p], shard)
        assert not bad
    with pytest.raises(ValueError):
        plan.outputs_in_order()
    with pytest.raises(ValueError):
        pkg.grouped.ShardedPlan([("cuda:0", [])])
    plan.close()


def test_loader_places_shards_on_a_device_list(pkg, tmp_path):
    """gguf_sd_loader(devices=[...]): one process, the file's tensors partitioned over the list, one state dict; state_dict_plan
    over tensors of ONE device stays a DequantPlan (here both shards sit on cuda:0)."""
    from test_gpu_gguf import _mixed_file
    import oracle
    path, spec, packed = _mixed_file(pkg, tmp_path)
    sd = pkg.loader.gguf_sd_loader(path, devices=["cuda:0", "cuda:0"])
    whole = pkg.loader.gguf_sd_loader(path, device="cuda:0")
    assert list(sd) == list(whole)                                          # the file's order, as the reference loader yields it
    marks = [k for k, v in sd.items() if getattr(v, "is_largest_weight", False)]
    assert marks == [k for k, v in whole.items() if getattr(v, "is_largest_weight", False)] and len(marks) == 1
    for k in sd:
        assert sd[k].is_cuda and torch.equal(torch.Tensor(sd[k]).cpu(), torch.Tensor(whole[k]).cpu()), k
    plan, keys = pkg.loader.state_dict_plan(sd, dtype=torch.float16)
    plan.launch()
    torch.cuda.synchronize()
    pre = "model.diffusion_model."
    for k, o in zip(keys, plan.outputs):
        want = oracle.dequant_f16(sd[k].tensor_type, packed[pre + k])
        assert np.array_equal(o.cpu().numpy().reshape(-1).view(np.uint16), want.view(np.uint16)), k
    plan.close()


@pytest.mark.timeout(900)
def test_bench_inproc_gpus_line():
    """`bench.py --inproc-gpus 2 --workload sd35-t5`: ONE process, two shards (on this one-GPU box: the rig lets them share the device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GGQ_BENCH_BACKEND="gloo")
    proc = subprocess.run([sys.executable, "bench.py", "--inproc-gpus", "2", "--workload", "sd35-t5", "--steps", "4", "--warmup", "1"], cwd=root, env=env,
                          capture_output=True, text=True, timeout=800)
    assert proc.returncode == 0, proc.stderr[-2000:]
    (line,) = [json.loads(ln) for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and len(line["config"]["shards"]) == 2
    assert sum(s["tensors"] for s in line["config"]["shards"]) == 549 and sum(s["bytes"] for s in line["config"]["shards"]) == line["config"]["bytes_per_step"]
    assert line["cpu_baseline"]["parity_vs_gpu"].startswith("bit-exact (549 tensors on 2 shards")
    assert line["ms_per_step"] >= max(s["gpu_ms_per_step"] for s in line["config"]["shards"]) * 0.999
