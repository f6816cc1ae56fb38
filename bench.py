#!/usr/bin/env python3
"""bench.py -- dequant GB/s (packed in -> fp16 out) and % of HBM3E peak, per quant type.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--qtype Q4_K] [--pairs 64]

A "step" is ONE pass of the hot path over the whole synthetic pool: a DequantPlan launch
(include/ggq.h ggq_plan_launch) over `pairs` x (3072x3072 + 3072x12288) FLUX.1-dev-shaped
linears of one quant type, every tensor with its own packed and output buffers.  The default pool
(64 pairs = 3.0 G elements) is sized so that the PACKED bytes alone (0.99 GB for Q2_K ... 3.2 GB
for Q8_0; 1.7 GB for Q4_K) are several times the 256 MiB Infinity Cache: with a pool whose packed
bytes fit in it (8 pairs of Q4_K = 212 MB) the cache keeps the inputs resident across steps and the
same kernel reads 5-15 % "faster" than HBM can deliver (profiles/r01_microbench_j_bigpool.txt).
Packed inputs are resident in HBM before the timed region starts; 7.7 GB of traffic per Q4_K step.

Headline workload: Q4_K (the north-star target format, BASELINE.json configs[2]); configs[1]
(Q4_0) and every other format are measured the same way and reported under "per_qtype".

Multi-GPU (driver: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...):
one process per GPU, WEAK scaling -- the global tensor list is N pools, partitioned by
comfyui-gguf_amd/sharding.py; no collective on the data path.  torch.distributed (RCCL) only
fences the timed region (barrier) and takes the MAX time over ranks.  The N > 1 line is self-proving:
`world` lists N distinct devices (the run refuses fewer), `config.shards` what every rank ran and its own
time, `cpu_baseline` the reference's CPU leg from rank 0 with `parity_vs_gpu` = EVERY rank's outputs
against the oracle, and `workloads` carries configs[3] / configs[4] STRONG-scaling (the same 304 / 549
tensors sharded over the N ranks).

Other workloads (not the driver's default; same JSON contract, one line):
  --workload flux        BASELINE configs[3]: the full FLUX.1-dev weight set (304 quantized tensors, Q4_K_M mix:
                         Q4_K + Q5_K), HBM-resident, one mixed-format plan launch per step; with --gpus N the
                         tensor list is SHARDED (strong scaling), no collectives.
  --workload sd35-t5     BASELINE configs[4]: SD3.5-large + T5-xxl weight tensors (549 tensors), same treatment.
  --workload flux-gguf   the FLUX weight set written to a synthetic .gguf file, then parsed by the native
                         reader, streamed file -> pinned -> HBM and dequantized: the PCIe-inclusive rate.
  --workload fused       the fused dequantize + linear kernels of install()'s default on the FLUX set at 1 / 4 / 32 / 64 / 256 rows of x, graph-replayed, with their
                         packed-read roofline (also sub-lines `fused_small_m` / `fused_mfma` of the default run).
  --workload fused-error the numerics table behind install()'s default (tools/fused_error.py): fused linears vs unpack + F.linear against fp64; not a throughput line.
  --workload per-layer   the FLUX weight set the way the node drives it (ops.py:177): one dequantize_tensor() launch per tensor, bf16
                         result -- standalone (graph-replayed, both store policies), eager, in context (unpack + F.linear per layer vs
                         dense-resident weights), and the reference's own eager torch ops on the same device tensors beside it.

The default run reports all of these as `workloads` sub-lines of the one JSON line, and `reference_on_this_gpu` for the headline pool.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import sys
import time

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL's intra-node transport); the launcher normally exports it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ggq_pkg import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md:35


def global_manifest(pkg, qtype, pairs, world):
    """Weak scaling: the job's tensor list is `world` copies of the per-GPU pool."""
    m = []
    for r in range(world):
        m += [(f"gpu{r}.{name}", q, shape) for name, q, shape in pkg.manifests.flux_linear_pool(qtype, pairs)]
    return m


def max_over_ranks(value, device, group=None):
    """MAX of a python float over all ranks (identity when not distributed).  ``group``: the RCCL fence group when there is one
    (device tensor), else the default group (gloo: host tensor)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class World:
    """The process group as bench.py uses it: fences around the timed regions, MAX of the step time, and object gathers of what
    every rank did (shard, per-rank time, parity verdict).  NOTHING of the data path goes through it.

    Two groups over the same ranks: a gloo CONTROL group (always: it cannot fail for GPU reasons, so every rank can take the same
    decision about the other one) and the RCCL group the timed regions are fenced through (backend "nccl" IS RCCL on ROCm).  If the
    RCCL group cannot be created or its first all-reduce fails on ANY rank, every rank learns so over the control group and the
    fences fall back to gloo -- the line then says so in ``world.backend`` instead of the run dying (Next #1e).
    GGQ_BENCH_BACKEND=gloo is the TEST RIG: no RCCL attempt, and ranks may share a device (N ranks on the one GPU of a test box);
    GGQ_BENCH_BACKEND=try-nccl is the rig WITH the RCCL attempt -- on a one-GPU box RCCL refuses two ranks on one device, which is how the
    fallback itself gets exercised (tests/test_gpu_multirank.py)."""

    def __init__(self, rank, local_rank, size, device, launched):
        self.rank, self.local_rank, self.size, self.device = rank, local_rank, size, device
        self.active = launched
        self.rig = os.environ.get("GGQ_BENCH_BACKEND", "nccl") in ("gloo", "try-nccl")
        self.fence_group, self.backend = None, None
        if not launched:
            return
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=size)
        if os.environ.get("GGQ_BENCH_BACKEND") == "gloo":
            self.backend = "gloo (GGQ_BENCH_BACKEND=gloo test rig: no RCCL attempt, ranks may share a device)"
            return
        err, pg = None, None
        try:
            pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=float(os.environ.get("GGQ_BENCH_RCCL_TIMEOUT_S", "600"))), device_id=device)
            t = torch.ones(1, device=device)
            dist.all_reduce(t, group=pg)
            torch.cuda.synchronize(device)
            if int(t.item()) != size:
                err = f"RCCL all-reduce over {size} ranks returned {t.item()}"
        except Exception as e:                                      # RuntimeError / DistBackendError: report, do not die
            err = f"{type(e).__name__}: {e}".replace("\n", " ")[:240]
        errs = self.gather(err)
        bad = [e for e in errs if e]
        if bad:
            self.backend = f"gloo (RCCL group failed on {len(bad)} of {size} ranks, fences fell back to gloo: {bad[0]})" + (" [test rig: ranks share a device]" if self.rig else "")
        else:
            self.fence_group, self.backend = pg, "nccl"

    def fence(self):
        """barrier + device synchronize on both sides (the bracket of every timed region)."""
        torch.cuda.synchronize(self.device)
        if self.active:
            import torch.distributed as dist
            dist.barrier(group=self.fence_group)
        torch.cuda.synchronize(self.device)

    def max(self, value):
        return max_over_ranks(value, self.device, self.fence_group) if self.active else float(value)

    def gather(self, obj):
        """[obj of rank 0, ..., obj of rank N-1] on every rank (control group)."""
        if not self.active:
            return [obj]
        import torch.distributed as dist
        rows = [None] * self.size
        dist.all_gather_object(rows, obj)
        return rows

    def wait_host(self):
        """Host-side barrier on the control group (the waiting ranks sleep in a socket read instead of spinning a GPU kernel in an
        RCCL barrier while rank 0 runs its CPU legs)."""
        if self.active:
            import torch.distributed as dist
            dist.barrier()

    def cpu_threads(self):
        """Host threads one rank's CHECKER may use: the ranks of a node check their outputs at the same time."""
        return max(1, (os.cpu_count() or 1) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", self.size))))

    def close(self):
        if self.active:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()


def device_blocks(pkg, qtype, n_blocks, device, seed):
    """Random packed blocks generated ON the device, nominal fp16 scale fields (comfyui-gguf_amd/synth.py)."""
    return pkg.synth.device_blocks(qtype, n_blocks, device, seed)


def build_pool(pkg, entries, device, seed0):
    items = []
    for i, (_, q, shape) in enumerate(entries):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        items.append((device_blocks(pkg, q, n_blocks, device, seed0 + i), q, shape))
    return pkg.grouped.DequantPlan(items, out_dtype=torch.float16)


def timed_steps(plan, steps, warmup, device, fence):
    """W untimed + exactly K timed launches, HIP events on the stream the kernels run on."""
    stream = torch.cuda.current_stream(device)
    for _ in range(warmup):
        plan.launch(stream)
    fence()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record(stream)
    for _ in range(steps):
        plan.launch(stream)
    stop.record(stream)
    torch.cuda.synchronize(device)
    wall_ms = (time.perf_counter() - t0) * 1e3
    fence()
    return start.elapsed_time(stop), wall_ms


_CEILING = {}


def measured_ceiling(pkg, device, gib=2, reps=7):
    """What THIS box's memory system gives streams with no arithmetic, measured in this process by the product library's own probes
    (include/ggq.h ggq_calibrate: 16 B per lane, 1 KiB per wave instruction like the dequant kernels): a fill, a copy and a read over
    ``gib`` GiB (8x the 256 MiB Infinity Cache), plain and non-temporal, HIP events on the launch stream, median of ``reps`` launches after
    one warm-up; the better of plain / non-temporal counts.  Cached per device: the headline and the sub-lines quote the same figures."""
    key = str(device)
    if key in _CEILING:
        return _CEILING[key]
    nat = pkg._native
    L = nat.lib()
    n = gib << 30
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    src.view(torch.int64).random_()
    stream = torch.cuda.current_stream(device)
    out = {}
    with torch.cuda.device(device):
        for name, kind, moved in (("fill", nat.CAL_FILL, n), ("fill_nt", nat.CAL_FILL_NT, n), ("copy", nat.CAL_COPY, 2 * n), ("copy_nt", nat.CAL_COPY_NT, 2 * n),
                                  ("read", nat.CAL_READ, n)):
            ts = []
            for i in range(reps + 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                nat.check(L.ggq_calibrate(kind, src.data_ptr(), dst.data_ptr(), n, stream.cuda_stream), "ggq_calibrate")
                b.record(stream)
                torch.cuda.synchronize(device)
                if i:
                    ts.append(a.elapsed_time(b))
            ts.sort()
            out[name] = round(moved / (ts[len(ts) // 2] * 1e-3) / 1e9, 1)
    del src, dst
    torch.cuda.empty_cache()
    res = {"measured_fill_GBps": max(out["fill"], out["fill_nt"]), "measured_copy_GBps": max(out["copy"], out["copy_nt"]), "measured_read_GBps": out["read"],
           "measured_all_GBps": out,
           "measured_how": f"ggq_calibrate over {gib} GiB in this process, median of {reps} launches (HIP events); copy counts read + written bytes"}
    _CEILING[key] = res
    return res


def with_ceiling(roofline, ceiling, read_bytes, write_bytes):
    """Add the measured figures and the ceiling of a stream with THIS kernel's read : write mix to a roofline object.  The mix is split into a
    copy part (every read byte paired with a written byte, at the measured copy rate) and a fill part (the remaining written bytes, at the
    measured fill rate): blend = (r + w) / (2r / copy + (w - r) / fill).  frac_of_blend = achieved / blend: how much of what this memory system
    gives a stream of that mix the kernel reaches -- beside frac, which stays against the 8 TB/s spec peak."""
    if not ceiling:
        return roofline
    r, w = float(read_bytes), float(write_bytes)
    copy, fill, read = ceiling["measured_copy_GBps"], ceiling["measured_fill_GBps"], ceiling["measured_read_GBps"]
    t = (2 * min(r, w) / copy + (w - r) / fill) if w >= r else (2 * w / copy + (r - w) / read)
    blend = (r + w) / t
    roofline.update({k: ceiling[k] for k in ("measured_fill_GBps", "measured_copy_GBps", "measured_read_GBps")})
    roofline.update({"blend_ceiling_GBps": round(blend, 1), "frac_of_blend": round(roofline["achieved"] / blend, 4),
                     "blend_mix": {"read_share": round(r / (r + w), 4), "write_share": round(w / (r + w), 4)},
                     "blend_how": "copy part (2 x read bytes at measured copy rate) + fill part (write - read bytes at measured fill rate); " + ceiling["measured_how"]})
    return roofline


class GpuTelemetry:
    """Clocks, power and temperature of the GPU while a leg runs: `rocm-smi --json` sampled once a second from a helper thread (the launching thread is
    never interrupted).  What it is for: the in-context cost is a difference of two long steps and moved 0.65 -> 2.4 ms from box to box in round 4; the
    line now says what the silicon was doing meanwhile.  ``summary()`` is None when rocm-smi is not there or prints nothing usable."""

    def __init__(self, device, period_s=1.0):
        self.index = torch.device(device).index or 0
        self.period, self.samples, self._stop, self._thread = period_s, [], None, None

    def _sample(self):
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(out).values()))
        except Exception:                                              # noqa: BLE001 -- telemetry must never fail the bench
            return None
        row = {}
        for k, v in card.items():
            kl = k.lower()
            m = re.search(r"-?\d+(?:\.\d+)?", str(v).split("(")[-1] if "clock" in kl else str(v))
            if not m:
                continue
            x = float(m.group(0))
            if "sclk" in kl and "clock speed" in kl:
                row["sclk_MHz"] = x
            elif "mclk" in kl and "clock speed" in kl:
                row["mclk_MHz"] = x
            elif "power" in kl and "socket" in kl or "average graphics package power" in kl:
                row["power_W"] = x
            elif "temperature" in kl and "junction" in kl:
                row["junction_C"] = x
            elif "temperature" in kl and ("hbm" in kl or "mem" in kl) and "hbm_C" not in row:
                row["hbm_C"] = x
        return row or None

    def __enter__(self):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                r = self._sample()
                if r:
                    self.samples.append(r)
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)
        return False

    def summary(self):
        if not self.samples:
            return None
        out = {"samples": len(self.samples), "source": "rocm-smi --showclocks --showpower --showtemp --json, 1 Hz while the legs ran"}
        for k in ("sclk_MHz", "mclk_MHz", "power_W", "junction_C", "hbm_C"):
            v = [s[k] for s in self.samples if k in s]
            if v:
                out[k] = {"min": min(v), "mean": round(sum(v) / len(v), 1), "max": max(v)}
        return out


def per_launch_stats(plan, reps, device):
    """Median / min duration of single launches (one HIP-event pair per launch, same stream) -- the
    distribution behind the average the timed region reports (median and min)."""
    stream = torch.cuda.current_stream(device)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record(stream)
    for i in range(reps):
        plan.launch(stream)
        evs[i + 1].record(stream)
    torch.cuda.synchronize(device)
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(reps))
    return ms[len(ms) // 2], ms[0]


def _median_min(fn, budget_s, min_reps, warmup):
    for _ in range(warmup):
        fn()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < min_reps or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], times[0], len(times)


def cpu_baseline_reference(pkg, plan, qtypes, budget_s):
    """kind 'reference': the reference's OWN torch-CPU path -- ``dequantize(data, qtype, oshape)`` of its dequant.py
    (dequant.py:30-44), executed verbatim (oracle/reference.py: /root/reference, or the copy oracle/stage_reference.py staged
    into oracle/_ref for the GPU box) -- timed on THIS box's host cores, in this process, on the same packed bytes the GPU just
    dequantized: the first (3072x3072, 3072x12288) pair of the pool (or the first tensors of a mixed set), copied back from
    HBM.  >= 3 warm-up and >= 10 timed passes, median and min; torch's intra-op thread count and the visible CPU count are
    stated.  Also the strongest parity statement the run can make: the reference's output == the GPU's output, bit for bit."""
    import numpy as np
    from oracle import reference
    if not reference.available():
        return None
    ref = reference.load_reference_dequant()
    n_sample = min(2, len(plan.outputs))
    packed = [plan._keep[i].cpu() for i in range(n_sample)]
    shapes = [tuple(plan.outputs[i].shape) for i in range(n_sample)]
    qs = [qtypes[i] for i in range(n_sample)]
    nbytes = sum(pkg.qtypes.algorithmic_bytes(q, int(np.prod(sh))) for q, sh in zip(qs, shapes))
    parity = True
    for i in range(n_sample):
        want = ref.dequantize(packed[i], qs[i], shapes[i])
        parity = parity and want.dtype == torch.float16 and bool(torch.equal(want.view(torch.int16), plan.outputs[i].cpu().view(torch.int16)))

    def one_pass():
        for p, q, sh in zip(packed, qs, shapes):
            ref.dequantize(p, q, sh)

    # torch's default intra-op thread count on a many-core host is not its fastest setting for these memory-bound elementwise ops
    # (128 threads: 0.6 GB/s on the GPU box's 256-CPU host): time the default AND a few smaller teams, report the best median
    # with ITS thread count, list them all.
    default_threads = torch.get_num_threads()
    counts = [default_threads] + [c for c in (32, 16, 8) if c < default_threads]
    by_threads, best = {}, None
    try:
        for c in counts:
            torch.set_num_threads(c)
            med, tmin, reps = _median_min(one_pass, budget_s / len(counts), 10, 3)
            by_threads[str(c)] = {"median_GBps": round(nbytes / med / 1e9, 3), "min_GBps": round(nbytes / tmin / 1e9, 3), "passes": reps}
            if best is None or med < best[0]:
                best = (med, tmin, reps, c)
    finally:
        torch.set_num_threads(default_threads)
    med, tmin, reps, threads = best
    medians = [v["median_GBps"] for v in by_threads.values()]
    return {"value": round(nbytes / med / 1e9, 3), "unit": "GB/s", "cores": threads, "threads": threads, "host_cpus": os.cpu_count(), "kind": "reference",
            # the reference's CPU path allocates every intermediate of its op chain afresh: it is bound by page faults, not by arithmetic or DRAM,
            # and the same function on the same host swings by up to 10x between tensor sizes, thread counts and passes
            "range_GBps": {"lowest_median_over_thread_counts": min(medians), "highest_median_over_thread_counts": max(medians),
                           "best_single_pass": max(v["min_GBps"] for v in by_threads.values())},
            "note": "a range, not a point: the reference's torch-CPU path is page-fault-bound on its freshly allocated intermediates; value = the best median; "
                    "cores = threads = torch intra-op threads of that median, host_cpus = CPUs the host shows",
            "sample": (f"reference dequant.py:30 dequantize() verbatim ({reference.source()} copy) on torch-CPU, best of {counts} intra-op threads "
                       f"(torch default {default_threads}; {os.cpu_count()} host CPUs visible); {' + '.join(f'{q.name} {sh[0]}x{sh[1]}' for q, sh in zip(qs, shapes))} "
                       f"(same packed bytes the GPU read), {reps} passes after 3 warm-up, median; (in+out) bytes / time"),
            "by_threads": by_threads, "min_GBps": round(nbytes / tmin / 1e9, 3), "median_ms": round(med * 1e3, 2), "min_ms": round(tmin * 1e3, 2),
            "parity_vs_gpu": "bit-exact" if parity else "MISMATCH"}


def reference_gpu_leg(pkg, plan, qtypes, device):
    """The reference's own eager torch ops (dequant.py:30-44, verbatim) on THIS GPU, over the same device-resident packed bytes the HIP
    path just read: the first (3072x3072, 3072x12288) pair of the pool.  What the reference delivers on an MI355X without this
    library -- a baseline beside `cpu_baseline`, and one more checker (outputs compared on the device, bit for bit)."""
    from oracle import reference
    if not reference.available():
        return None
    ref = reference.load_reference_dequant()
    n_sample = min(2, len(plan.outputs))
    packed = [plan._keep[i] for i in range(n_sample)]
    shapes = [tuple(plan.outputs[i].shape) for i in range(n_sample)]
    qs = [qtypes[i] for i in range(n_sample)]
    import numpy as np
    nbytes = sum(pkg.qtypes.algorithmic_bytes(q, int(np.prod(sh))) for q, sh in zip(qs, shapes))
    parity = all(torch.equal(ref.dequantize(p, q, sh).view(torch.int16), plan.outputs[i].view(torch.int16)) for i, (p, q, sh) in enumerate(zip(packed, qs, shapes)))

    def one_pass():
        for p, q, sh in zip(packed, qs, shapes):
            ref.dequantize(p, q, sh)

    for _ in range(3):
        one_pass()
    torch.cuda.synchronize(device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps):
        one_pass()
    b.record()
    torch.cuda.synchronize(device)
    ms = a.elapsed_time(b) / reps
    torch.cuda.empty_cache()
    return {"value": round(nbytes / (ms * 1e-3) / 1e9, 1), "unit": "GB/s", "ms_per_pass": round(ms, 4),
            "sample": (f"reference dequant.py:30 dequantize() verbatim ({reference.source()} copy), eager torch-ROCm ops on cuda: "
                       f"{' + '.join(f'{q.name} {sh[0]}x{sh[1]}' for q, sh in zip(qs, shapes))} (the pool's first pair, device-resident), {reps} passes after 3 warm-up, HIP events"),
            "parity_vs_hip_path": "bit-exact" if parity else "MISMATCH"}


def cpu_baseline_port(pkg, plan, qtype, budget_s):
    """kind 'port': the oracle's throughput leg (oracle/ggq_oracle_simd.c: the same op sequence with AVX2+F16C and OpenMP) on the
    first tensors of the pool -- the fastest CPU implementation of this path the repo has, reported BESIDE the reference's own
    torch-CPU figure so that the GPU/CPU ratio is not flattered by a slow baseline; hosts without AVX2/F16C time the soft-float
    checker instead.  Also re-checks parity: the GPU output of the sample must equal the soft-float oracle's bit for bit."""
    import numpy as np
    import oracle
    n_sample = min(8, len(plan.outputs))                       # 8 tensors = 4 (B + C) pairs, 189 M elements: out of any CPU cache
    packed = [plan._keep[i].cpu().numpy() for i in range(n_sample)]
    outs = [np.empty(plan.outputs[i].numel(), dtype=np.uint16) for i in range(n_sample)]
    max_threads = int(oracle.lib().ggq_oracle_max_threads())
    simd = oracle.simd_available()
    want = oracle.dequant_f16(qtype, packed[0])                # soft-float checker: parity of the GPU and of the timed leg
    got = plan.outputs[0].cpu().numpy().reshape(-1)
    parity = bool(np.array_equal(got.view(np.uint16), want.view(np.uint16)))
    if simd:
        parity = parity and bool(np.array_equal(oracle.dequant_f16(qtype, packed[0], simd=True).view(np.uint16), want.view(np.uint16)))
    nbytes = sum(pkg.qtypes.algorithmic_bytes(qtype, o.size) for o in outs)

    def one_pass(c):
        for p, o in zip(packed, outs):
            oracle.dequant_f16(qtype, p, threads=c, simd=simd, out=o)

    # OpenMP on every hardware thread of a shared box is not always the fastest setting: try a few team
    # sizes, report the best median and ITS thread count.
    counts = sorted({c for c in (1, 16, 64, 128, max_threads) if c <= max_threads})
    best = None
    for c in counts:
        med, tmin, reps = _median_min(lambda: one_pass(c), budget_s / len(counts), 3, 1)
        if best is None or med < best[0]:
            best = (med, c, reps, tmin)
    med, threads, reps, tmin = best
    leg = "oracle/ggq_oracle_simd.c (AVX2+F16C)" if simd else "oracle/ggq_oracle.c (soft-float; host lacks AVX2/F16C)"
    return {"value": round(nbytes / med / 1e9, 3), "unit": "GB/s", "cores": threads, "threads": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "sample": (f"{qtype.name}: first {n_sample} pool tensors ({sum(o.size for o in outs)} elements, same packed bytes), "
                       f"{reps} passes of {leg} with OpenMP on {threads} threads (best of team sizes {counts}; "
                       f"{os.cpu_count()} host CPUs visible), median; fastest pass {nbytes / tmin / 1e9:.3f} GB/s"),
            "parity_vs_gpu": "bit-exact" if parity else "MISMATCH"}


def parity_all_ranks(plan, qtypes, W):
    """EVERY rank checks EVERY output tensor of its OWN timed launch against the oracle, outside the timed regions
    (oracle/plan_check.py: whole tensors vs the AVX2 leg, that leg vs the soft-float checker on three windows per tensor), with its
    share of the host's threads; the verdicts are gathered, so the line of an N-GPU run states N ranks' parity, not rank 0's."""
    from oracle import plan_check
    t0 = time.perf_counter()
    n, bad = plan_check.check_plan(plan._keep, qtypes, plan.outputs, threads=W.cpu_threads() if W.size > 1 else None)
    return W.gather({"rank": W.rank, "tensors": n, "differ": len(bad), "first": [list(x) for x in bad[:3]], "seconds": round(time.perf_counter() - t0, 2)})


def parity_statement(rows, first=None, with_reference=False):
    """One string for `cpu_baseline.parity_vs_gpu`: the MIN over ranks of the per-rank verdicts (any differing tensor on any rank
    -> MISMATCH) and of rank 0's first-tensor checks against the reference itself."""
    n = sum(r["tensors"] for r in rows)
    bad = [r for r in rows if r["differ"]]
    if bad or (first is not None and first != "bit-exact"):
        return (f"MISMATCH ({sum(r['differ'] for r in bad)} of {n} tensors differ from the oracle on ranks {[r['rank'] for r in bad]}: "
                f"{bad[0]['first'] if bad else []}; rank 0 first-tensor check {first})")
    per = [r["tensors"] for r in rows]
    if len(rows) == 1:
        where, whose, who = f"{n} tensors", "the", "the"
    else:
        where = (f"{len(rows)} x {per[0]} = {n}" if len(set(per)) == 1 else f"{' + '.join(map(str, per))} = {n}") + f" tensors on {len(rows)} ranks"
        whose, who = "every rank's", "rank 0's"
    return (f"bit-exact ({where}: every output of {whose} timed launch vs the oracle"
            + (f", {who} first 2 also vs the reference's dequantize() on torch-CPU" if with_reference else "") + ")")


def cpu_baselines(pkg, plan, qtypes, budget_s, W):
    """(cpu_baseline, cpu_baseline_port) on rank 0, None elsewhere; EVERY rank takes part (its own parity check is gathered).
    The reference's own torch-CPU path when its sources are present (north_star: "the reference's CPU torch path timed on the same
    box's host cores in the same run"), the AVX2 port beside it.  Without the reference (neither /root/reference nor a staged
    oracle/_ref) the port IS the baseline, labelled as such.  For N > 1 the CPU legs still run, on rank 0, on rank 0's first tensors,
    while the other ranks wait at the next fence: the line of a multi-GPU run carries the same baseline and a parity verdict from
    every rank."""
    rows = parity_all_ranks(plan, qtypes, W)                    # collective: before the rank-0-only legs
    if W.rank != 0:
        W.wait_host()                                           # sleep on the control group while rank 0 times the CPU legs
        return None, None
    ref = cpu_baseline_reference(pkg, plan, qtypes, budget_s * 0.5)
    port = cpu_baseline_port(pkg, plan, qtypes[0], budget_s * 0.5) if len(set(qtypes[:8])) == 1 else None
    head = ref if ref is not None else port
    if head is None:                                            # mixed-format set without the reference's sources: parity only
        head = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "no CPU leg (mixed formats, reference sources absent): parity only"}
        ref = head
    head["parity_vs_gpu"] = parity_statement(rows, head.get("parity_vs_gpu"), with_reference=head.get("kind") == "reference")
    head["parity_check_s"] = max(r["seconds"] for r in rows)
    head["parity_by_rank"] = [{k: r[k] for k in ("rank", "tensors", "differ", "seconds")} for r in rows]
    W.wait_host()
    return (ref, port) if ref is not None else (port, None)


def shard_report(pkg, manifest, indices, plan, gpu_ms_step, W):
    """What every rank ACTUALLY ran (gathered, not recomputed on rank 0): its tensor indices, bytes, kernels and its own GPU time per
    step in the reported region -- plus the statement that the shards are disjoint and cover the tensor list."""
    rows = W.gather({"rank": W.rank, "device": str(W.device), "indices": list(indices), "bytes": plan.bytes, "kernels": plan.kernels,
                     "gpu_ms_per_step": round(gpu_ms_step, 5)})
    flat = sorted(i for r in rows for i in r["indices"])
    cover = "disjoint, complete" if flat == list(range(len(manifest))) else f"BROKEN ({len(flat)} indices for {len(manifest)} tensors)"
    shards = [{"rank": r["rank"], "device": r["device"], "tensors": len(r["indices"]), "bytes": r["bytes"], "kernels_per_step": r["kernels"],
               "gpu_ms_per_step": r["gpu_ms_per_step"], "GBps": round(r["bytes"] / (r["gpu_ms_per_step"] * 1e-3) / 1e9, 1)} for r in rows]
    return shards, f"{cover} ({len(flat)} of {len(manifest)} tensors over {len(rows)} ranks)"


def run_flux(pkg, args, W, workload=None, cpu_seconds=None):
    """configs[3] / configs[4]: a full weight set, mixed quant types, resident in HBM, the TENSOR LIST sharded over the ranks
    (strong scaling: the same 304 / 549 tensors whatever N is; no collective on the data path).  Collective: every rank calls it."""
    workload = workload or args.workload
    cpu_seconds = args.cpu_seconds if cpu_seconds is None else cpu_seconds
    rank, world, device = W.rank, W.size, W.device
    if workload == "sd35-t5":
        manifest, label = pkg.manifests.sd35_t5(args.mix), "BASELINE configs[4]: SD3.5-large + T5-xxl weight tensors"
    else:
        manifest, label = pkg.manifests.flux_dev(args.mix), "BASELINE configs[3]: full FLUX.1-dev weight set"
    indices = pkg.sharding.partition(manifest, world)[rank]
    mine = [manifest[i] for i in indices]
    plan = build_pool(pkg, mine, device, seed0=7000 + 1000 * rank)
    ms_per_step, gpu_ms_step, wall_ms_step, regions = median_region(pkg, plan, args, W, args.regions)
    total_bytes = sum(pkg.sharding.tensor_cost(e) for e in manifest)
    value = total_bytes / (ms_per_step * 1e-3) / 1e9
    shards, cover = shard_report(pkg, manifest, indices, plan, gpu_ms_step, W)
    result = None
    if rank == 0:
        achieved = plan.bytes / (gpu_ms_step * 1e-3) / 1e9
        qcount = {}
        for _, q, _ in manifest:
            qcount[q.name] = qcount.get(q.name, 0) + 1
        traffic, traffic_source = load_traffic(pkg, f"{workload}:{args.mix}") if world == 1 else (None, "PMC figures are per single-GPU launch")
        result = {
            "metric": "dequant GB/s (packed in -> fp16 out), (in+out) bytes / time",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{label} ({len(manifest)} tensors, {args.mix} mix {qcount}), "
                                   f"HBM-resident, tensor list sharded over {world} GPU(s)",
                       "elements": sum(s[0] * s[1] for _, _, s in manifest), "bytes_per_step": total_bytes,
                       "kernels_per_step_rank0": plan.kernels, "imbalance": round(pkg.sharding.imbalance(manifest, world), 4),
                       "parallelism": f"tensor-list sharding x{world}, no collectives",
                       "shards": shards, "shard_cover": cover,
                       "timed_regions_ms_per_step": regions, "reported": "median region; ms_per_step = MAX over ranks"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "ggq::dequant_many<Fmt*, ...> (one launch per format present)",
                         "algorithmic_bytes_per_launch": plan.bytes, "avg_launch_ms": round(gpu_ms_step, 5),
                         "host_wall_ms_per_step": round(wall_ms_step, 5), "of": "rank 0's shard"},
            "cpu_baseline": None,
        }
        if not args.no_ceiling:
            in_b = sum(t.numel() for t in plan._keep)
            with_ceiling(result["roofline"], measured_ceiling(pkg, W.device), in_b, plan.bytes - in_b)
    if cpu_seconds > 0:
        base, port = cpu_baselines(pkg, plan, [q for _, q, _ in mine], cpu_seconds, W)
        if rank == 0:
            result["cpu_baseline"] = base
            if port is not None:
                result["cpu_baseline_port"] = port
    plan.close()
    return result


def run_inproc(pkg, args, device):
    """--inproc-gpus N: ONE process drives N GPUs (ComfyUI is single-process) -- the tensor list of the workload partitioned with the same
    sharding.partition, one DequantPlan per device (grouped.ShardedPlan), every launch enqueued by this one host thread, no collective.
    A step = one launch on every device; timed with a HIP event pair per device stream AND the host clock around the K steps
    (`ms_per_step` = the slowest device's; `host_wall_ms_per_step` shows whether one thread keeps N devices fed).  Under the test rig
    (GGQ_BENCH_BACKEND=gloo) the N shards may share devices -- then they time-share and the line says so."""
    n = args.inproc_gpus
    n_dev = torch.cuda.device_count()
    rig = os.environ.get("GGQ_BENCH_BACKEND", "nccl") == "gloo"
    if n > n_dev and not rig:
        sys.exit(f"--inproc-gpus {n} but only {n_dev} GPU(s) visible")
    devices = [torch.device("cuda", i % n_dev) for i in range(n)]
    if args.workload == "sd35-t5":
        manifest, label = pkg.manifests.sd35_t5(args.mix), "BASELINE configs[4]: SD3.5-large + T5-xxl weight tensors"
    elif args.workload == "flux":
        manifest, label = pkg.manifests.flux_dev(args.mix), "BASELINE configs[3]: full FLUX.1-dev weight set"
    else:
        manifest, label = global_manifest(pkg, pkg.qtypes.Q[args.qtype], args.pairs, n), f"BASELINE configs[2] {args.qtype} pool x {n}"
    rows = pkg.grouped.ShardedPlan.assignment(manifest, devices)
    shards = []
    for r, (d, ix) in enumerate(rows):
        shards.append((d, [(device_blocks(pkg, manifest[i][1], pkg.synth.n_blocks_for(manifest[i][1], manifest[i][2][0] * manifest[i][2][1]), d, 7000 + 1000 * r + k),
                            manifest[i][1], manifest[i][2]) for k, i in enumerate(ix)]))
    plan = pkg.grouped.ShardedPlan(shards, own_streams=True, indices=[ix for _, ix in rows])

    def sync_all():
        for d in set(devices):
            torch.cuda.synchronize(d)

    regions = []
    for r in range(args.regions):
        for _ in range(args.warmup if r == 0 else 0):
            plan.launch()
        sync_all()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan.plans]
        t0 = time.perf_counter()
        for (a, _), st in zip(evs, plan.streams):
            a.record(st)
        for _ in range(args.steps):
            plan.launch()
        for (_, b), st in zip(evs, plan.streams):
            b.record(st)
        t_enq = time.perf_counter() - t0
        sync_all()
        wall = time.perf_counter() - t0
        per = [a.elapsed_time(b) / args.steps for a, b in evs]
        regions.append((max(per), per, wall * 1e3 / args.steps, t_enq * 1e3 / args.steps))
    regions.sort(key=lambda x: x[0])
    ms, per, wall_ms, enq_ms = regions[len(regions) // 2]
    from oracle import plan_check
    bad, checked = [], 0
    for (d, items), p in zip(shards, plan.plans):
        k, b = plan_check.check_plan([it[0] for it in items], [it[1] for it in items], p.outputs)
        checked, bad = checked + k, bad + b
    total = sum(pkg.sharding.tensor_cost(e) for e in manifest)
    result = {
        "metric": "dequant GB/s (packed in -> fp16 out), (in+out) bytes / time", "value": round(total / (ms * 1e-3) / 1e9, 1), "unit": "GB/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True,
        "scaling": "strong" if args.workload != "pool" else "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{label} ({len(manifest)} tensors), HBM-resident, sharded over {n} device(s) driven by ONE process / ONE host thread",
                   "parallelism": f"in-process tensor-list sharding x{n} (grouped.ShardedPlan), no collectives, no peer access",
                   "devices": [str(d) for d in devices], "distinct_devices": len(set(devices)), "bytes_per_step": total,
                   "shards": [{"device": str(d), "tensors": len(items), "bytes": p.bytes, "gpu_ms_per_step": round(t, 5)} for (d, items), p, t in zip(shards, plan.plans, per)],
                   "imbalance": round(pkg.sharding.imbalance(manifest, n), 4), "host_wall_ms_per_step": round(wall_ms, 5),
                   "host_enqueue_ms_per_step": round(enq_ms, 5), "timed_regions_ms_per_step": [round(x[0], 5) for x in regions]},
        "roofline": {"bound": "hbm", "achieved": round(plan.plans[0].bytes / (per[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(plan.plans[0].bytes / (per[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "of": f"shard 0 on {devices[0]}"},
        "cpu_baseline": {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "no CPU timing leg on this line (see the default line); parity only",
                         "parity_vs_gpu": f"bit-exact ({checked} tensors on {len(plan.plans)} shards: every output vs the oracle)" if not bad else f"MISMATCH {bad[:3]}"},
    }
    plan.close()
    return result


def run_flux_gguf(pkg, args, device):
    """File -> HBM -> dense: write the FLUX.1-dev weight set as a synthetic .gguf, then time the native
    parse, the streaming upload (pread -> pinned -> H2D) and the dequant of everything that landed."""
    import tempfile
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gguf_writer import GGUFWriter                     # test infrastructure: the product only READS GGUF
    manifest = pkg.manifests.flux_dev(args.mix)
    if args.limit_tensors:
        manifest = manifest[:args.limit_tensors]
    w = GGUFWriter(arch="flux")
    t0 = time.perf_counter()
    for i, (name, q, shape) in enumerate(manifest):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        w.add_tensor("model.diffusion_model." + name, q, (shape[1], shape[0]), device_blocks(pkg, q, n_blocks, device, 9000 + i).cpu().numpy())
    tmpdir = tempfile.mkdtemp(prefix="ggq_bench_", dir=os.environ.get("TMPDIR") or None)
    path = os.path.join(tmpdir, "flux1-dev-synthetic.gguf")
    w.write(path)
    del w
    t_write = time.perf_counter() - t0
    file_bytes = os.path.getsize(path)
    try:
        t0 = time.perf_counter()
        f = pkg.gguf_file.GGUFFile(path)
        t_parse = time.perf_counter() - t0
        f.close()
        ups = []
        for _ in range(max(2, args.steps)):                # first pass also pins the staging buffers
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            sd = pkg.loader.gguf_sd_loader(path, device=device, upload_threads=args.upload_threads)
            torch.cuda.synchronize(device)
            ups.append(time.perf_counter() - t0)
            if len(ups) < max(2, args.steps):
                del sd
        t_load = min(ups[1:])
        plan, keys = pkg.loader.state_dict_plan(sd, dtype=torch.float16)
        gpu_ms, _ = timed_steps(plan, 10, 2, device, lambda: torch.cuda.synchronize(device))
        t_deq = gpu_ms / 10 * 1e-3
        packed_bytes = sum(sd[k].numel() for k in keys)
        # consistency spot check (GPU only; the parity tests proper are tests/test_gpu_gguf.py): the first tensor through
        # the whole-file plan equals the same tensor through the per-tensor entry point, and its bytes equal the file's
        k0 = keys[0]
        single = pkg.dequant.dequantize_tensor(sd[k0], torch.float16)
        same = bool(torch.equal(single.view(torch.int16), plan.outputs[0].view(torch.int16)))
        with pkg.gguf_file.GGUFFile(path) as f0:
            t0_info = next(t for t in f0.tensors if t.name.endswith(k0))
            same = same and bool(torch.equal(t0_info.data, torch.Tensor(sd[k0]).cpu()))
        e2e = t_load + t_deq
        result = {
            "metric": "GGUF file -> HBM -> dense fp16: (packed in + dense out) bytes / (load + dequant) time",
            "value": round(plan.bytes / e2e / 1e9, 1), "unit": "GB/s", "n_gpus": 1, "steps": len(ups) - 1, "warmup": 1,
            "ms_per_step": round(e2e * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] streamed: synthetic FLUX.1-dev .gguf ({len(manifest)} tensors, {args.mix}), "
                                   "native parse + threaded pread->pinned->H2D upload + one mixed-format dequant pass",
                       "file_bytes": file_bytes, "packed_bytes": packed_bytes, "dense_bytes": plan.bytes - packed_bytes,
                       "write_file_s": round(t_write, 2), "parse_ms": round(t_parse * 1e3, 3),
                       "load_ms_best": round(t_load * 1e3, 2), "load_ms_all": [round(u * 1e3, 1) for u in ups],
                       "upload_GBps_packed": round(file_bytes / t_load / 1e9, 2), "dequant_ms": round(t_deq * 1e3, 3),
                       "dequant_GBps": round(plan.bytes / t_deq / 1e9, 1), "upload_threads": args.upload_threads or 8,
                       "first_tensor_plan_vs_single_and_file_bytes": "identical" if same else "MISMATCH"},
            "roofline": {"bound": "pcie", "achieved": round(file_bytes / t_load / 1e9, 2), "peak": 63.0, "unit": "GB/s",
                         "frac": round(file_bytes / t_load / 1e9 / 63.0, 4), "traffic": None,
                         "note": "host->device link bound (PCIe Gen5 x16 ~63 GB/s one way); the dequant kernels are ~100x faster"},
            "cpu_baseline": None,
        }
        plan.close()
    finally:
        try:
            os.remove(path)
            os.rmdir(tmpdir)
        except OSError:
            pass
        pkg._native.lib().ggq_gguf_upload_release()
    return result


def load_traffic(pkg, workload_key):
    """(HBM bytes per launch, source) from the committed PMC summary -- collected in separate rocprofv3 --pmc passes and corrected
    per MI355X_MICROARCH.md (profiles/README.md).  The figure is only reported when the library THIS run loaded was compiled from
    the very sources the counters were collected on (``ggq_build_id()`` == the id recorded in profiles/pmc_traffic.json); after
    any kernel change it is null until the counters are re-collected, and ``traffic_source`` says why."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    build_id = pkg._native.lib().ggq_build_id().decode()
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, f"no profiles/pmc_traffic.json (library build {build_id})"
    pmc_id = table.get("_build_id")
    if pmc_id != build_id:
        return None, f"stale: PMC counters were collected on library build {pmc_id}, this run loaded build {build_id} -- re-collect (tools/pmc_summarize.py)"
    if workload_key not in table:
        return None, f"no PMC entry for {workload_key!r} (library build {build_id})"
    return table[workload_key], f"{table.get('_provenance', 'profiles/pmc_traffic.json')}; library build {build_id}"


def observed_world(W):
    """What the process group really looks like, gathered from every rank -- so that an N-GPU line proves N ranks on N DISTINCT
    devices took part: group size, the backend the timed regions were fenced through, and per rank the device index, name, uuid and
    PCI bus id.  Collective (every rank calls it)."""
    props = torch.cuda.get_device_properties(W.device)
    mine = {"rank": W.rank, "device": str(W.device), "name": props.name, "arch": getattr(props, "gcnArchName", None),
            "uuid": str(getattr(props, "uuid", "")) or None, "pci_bus_id": getattr(props, "pci_bus_id", None),
            "total_memory_GB": round(props.total_memory / 1e9, 1), "pid": os.getpid()}
    rows = W.gather(mine)
    # a physical device = its (uuid, PCI bus id) pair: two ranks share a device only if BOTH coincide (a driver that reports one of them as a
    # constant must not make a legitimate N-GPU run look like one device); without either, the per-process device index is all there is
    ids = {(r["uuid"], r["pci_bus_id"]) if (r["uuid"] or r["pci_bus_id"] is not None) else (r["device"],) for r in rows}
    return {"size": W.size, "backend": W.backend, "ranks": rows, "distinct_devices": len(ids)}


def run_per_layer(pkg, args, device, fence):
    """The call the node really makes (reference ops.py:177): ONE dequantize_tensor() per quantized layer per forward, here over
    the 304 tensors of the FLUX.1-dev set in model order, bf16 result (FLUX computes in bf16) -- 304 launches per pass instead of
    the 2 of the whole-set plan.  Three views, because they answer different questions:
      * `value` = STANDALONE, GPU-bound: the 304 launches replayed back to back from a captured HIP graph, nothing reading the results --
        with the shipped store policy of the per-layer entry point (write-through, sc1) and, beside it, with non-temporal stores
        (ggq_dequant_stream), which are faster here precisely because nobody reads the weight back;
      * `eager` = the python loop as ComfyUI runs it (host enqueue cost included; outputs go back to torch's allocator after every call);
      * `in_context` = what the path costs where it actually runs: every layer's unpack followed by its F.linear on 4608 tokens (the
        reference's forward, ops.py:242-244), against the same GEMMs on dense weights kept resident -- there the shipped write-through
        stores win by a wide margin over non-temporal ones, because the GEMM finds the weight in the Infinity Cache; that is why they ship.
    Three timed regions each, median reported, HIP events on the launch stream."""
    manifest = pkg.manifests.flux_dev(args.mix)
    if args.limit_tensors:
        manifest = manifest[:args.limit_tensors]               # smoke runs / tests
    tensors = []
    for i, (_, q, shape) in enumerate(manifest):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        tensors.append(pkg.ops.GGMLTensor(device_blocks(pkg, q, n_blocks, device, 7000 + i), tensor_type=q, tensor_shape=shape))
    nbytes = sum(pkg.sharding.tensor_cost(e) for e in manifest)
    dq, dq_stream = pkg.dequant.dequantize_tensor, pkg.dequant.dequantize_tensor_streaming
    dtype = torch.bfloat16
    passes = max(2, args.steps // 10)
    stream = torch.cuda.current_stream(device)

    def region(fn, n):
        fence()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(stream)
        for _ in range(n):
            fn()
        b.record(stream)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(device)
        fence()
        return a.elapsed_time(b) / n, t_host * 1e3 / n

    def median(fn, n, warm=2):
        for _ in range(warm):
            fn()
        rows = sorted(region(fn, n) for _ in range(args.regions))
        return rows[len(rows) // 2], [round(r[0], 5) for r in rows]

    def eager_pass():
        for t in tensors:
            dq(t, dtype)

    (e_ms, e_host), eager_regions = median(eager_pass, passes)
    from oracle import plan_check
    standalone, parity = {}, None
    for policy, fn in (("shipped_sc1", dq), ("streaming_nt", dq_stream)):
        side = torch.cuda.Stream(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            keep = [fn(t, dtype) for t in tensors]                     # warm the allocator on the capture stream
            torch.cuda.synchronize(device)
            with torch.cuda.graph(graph, stream=side):
                keep = [fn(t, dtype) for t in tensors]
        graph.replay()
        torch.cuda.synchronize(device)
        if parity is None:
            # parity of what the per-layer launches wrote (outside the timed regions): every tensor vs the oracle
            n, bad = plan_check.check_plan([t.as_subclass(torch.Tensor) for t in tensors], [q for _, q, _ in manifest], keep, windows=False)
            parity = f"bit-exact ({n} tensors)" if not bad else f"MISMATCH {bad[:3]}"
        (g_ms, _), regs = median(graph.replay, passes, warm=1)
        standalone[policy] = {"ms_per_pass": round(g_ms, 5), "GBps": round(nbytes / (g_ms * 1e-3) / 1e9, 1), "regions_ms": regs,
                              "us_per_launch": round(g_ms * 1e3 / len(manifest), 3)}
        del graph, keep
        torch.cuda.empty_cache()

    # in context: unpack + F.linear per layer (4608 tokens; the modulation layers see one row), vs the same GEMMs on resident dense weights
    tokens = 4608
    xs = {}
    for name, _, (rows, cols) in manifest:
        m = 1 if "mod" in name else tokens
        if (m, cols) not in xs:
            xs[(m, cols)] = torch.randn(m, cols, device=device, dtype=dtype) * 0.05
    layer_x = [xs[(1 if "mod" in name else tokens, shape[1])] for name, _, shape in manifest]
    lin = torch.nn.functional.linear
    dense = [dq(t, dtype) for t in tensors]

    def step_with(fn):
        def step():
            for t, x in zip(tensors, layer_x):
                lin(x, fn(t, dtype))
        return step

    def step_dense():
        for w, x in zip(dense, layer_x):
            lin(x, w)

    # The cost is a DIFFERENCE of two ~70 ms steps, so a 1 % drift of the GPU's clock between the two legs reads as 0.7 ms (rounds 3-4 measured the dense leg
    # first, then the quantized ones: 0.65-2.4 ms across boxes).  Since round 5 the three legs are INTERLEAVED region by region (dense, sc1, nt, dense, ...), each
    # reported as the median of its regions, and the GPU's clocks / power are sampled beside them (rocm-smi, once a second, off the launching thread).
    legs = {"dense": step_dense, "shipped_sc1": step_with(dq), "streaming_nt": step_with(dq_stream)}
    for fn in legs.values():
        fn()
        fn()
    seen = {k: [] for k in legs}
    with GpuTelemetry(device) as tele:
        for _ in range(max(3, args.regions)):
            for k, fn in legs.items():
                seen[k].append(region(fn, 3)[0])
    med = {k: sorted(v)[len(v) // 2] for k, v in seen.items()}
    d_ms = med["dense"]
    ctx = {"tokens": tokens, "ms_per_step_dense_resident": round(d_ms, 3), "dense_regions_ms": [round(v, 5) for v in seen["dense"]],
           "how": "dense / sc1 / nt regions interleaved (3 steps each), medians; cost = median(leg) - median(dense)", "gpu_telemetry": tele.summary()}
    for policy in ("shipped_sc1", "streaming_nt"):
        ctx[policy] = {"ms_per_step": round(med[policy], 3), "dequant_cost_ms_per_step": round(med[policy] - d_ms, 3), "regions_ms": [round(v, 5) for v in seen[policy]],
                       "cost_by_region_ms": [round(q - d, 3) for q, d in zip(seen[policy], seen["dense"])]}
    del dense
    torch.cuda.empty_cache()

    # the reference's OWN path on this GPU: its eager torch ops (dequant.py:15-44, executed verbatim from /root/reference or the staged
    # oracle/_ref copy) over the same 304 device tensors -- what a user of the reference gets on an MI355X today.  A baseline and a
    # checker (every tensor compared with the HIP path's result, bit for bit), never part of a timed region of the product.
    ref_gpu = None
    from oracle import reference
    if reference.available():
        ref = reference.load_reference_dequant()
        plain = []
        for t, (_, q, shape) in zip(tensors, manifest):
            u = t.as_subclass(torch.Tensor)
            u = u.view(u.shape)                        # a fresh tensor object to hang the reference's attributes on (ops.py:44-60)
            u.tensor_type, u.tensor_shape = q, torch.Size(shape)
            plain.append(u)
        bad = [name for (name, _, _), u, t in zip(manifest, plain, tensors)
               if not torch.equal(ref.dequantize_tensor(u, dtype).view(torch.int16), dq(t, dtype).view(torch.int16))]

        def ref_pass():
            for u in plain:
                ref.dequantize_tensor(u, dtype)

        def ref_step():
            for u, x in zip(plain, layer_x):
                lin(x, ref.dequantize_tensor(u, dtype))

        (r_ms, _), r_regs = median(ref_pass, 1, warm=1)
        (rs_ms, _), rs_regs = median(ref_step, 1, warm=1)
        ref_gpu = {"what": f"reference dequantize_tensor() verbatim ({reference.source()} copy), eager torch-ROCm ops on the same device tensors, bf16 result",
                   "standalone_ms_per_pass": round(r_ms, 3), "standalone_GBps": round(nbytes / (r_ms * 1e-3) / 1e9, 1), "standalone_regions_ms": r_regs,
                   "in_context_ms_per_step": round(rs_ms, 3), "in_context_dequant_cost_ms_per_step": round(rs_ms - d_ms, 3), "in_context_regions_ms": rs_regs,
                   "hip_path_speedup_standalone_eager": round(r_ms / e_ms, 1),
                   "hip_path_step_speedup_in_context": round(rs_ms / ctx["shipped_sc1"]["ms_per_step"], 2),
                   "parity_vs_hip_path": f"bit-exact ({len(plain)} tensors)" if not bad else f"MISMATCH {bad[:3]}"}
        del plain
        torch.cuda.empty_cache()

    g = standalone["shipped_sc1"]
    line = {
        "metric": "dequant GB/s, one dequantize_tensor() launch per layer (packed in -> bf16 out), (in+out) bytes / time",
        "value": g["GBps"], "unit": "GB/s", "ms_per_step": g["ms_per_pass"],
        "config": {"workload": f"FLUX.1-dev weight set ({len(manifest)} tensors, {args.mix}) through the per-layer entry point, one launch per tensor in model "
                               "order, bf16 result; value = standalone GPU-bound (the launches replayed from a captured HIP graph, nothing reads the results) with the "
                               "shipped store policy (write-through sc1 stores: chosen for the in-context cost, see in_context)",
                   "launches_per_pass": len(manifest), "passes_per_region": passes, "bytes_per_pass": nbytes,
                   "standalone_gpu_bound": standalone,
                   "eager_regions_ms": eager_regions, "eager_ms_per_pass": round(e_ms, 5), "eager_GBps": round(nbytes / (e_ms * 1e-3) / 1e9, 1),
                   "eager_host_enqueue_us_per_call": round(e_host * 1e3 / len(manifest), 2),
                   "in_context": ctx,
                   "reference_on_this_gpu": ref_gpu,
                   "parity_vs_oracle": parity},
        "roofline": {"bound": "hbm", "achieved": g["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(g["GBps"] / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "ggq::dequant_one<Fmt*, ...> (one launch per tensor; team shape picked per tensor size; sc1 stores)",
                     "algorithmic_bytes_per_launch": nbytes // len(manifest), "avg_launch_ms": round(g["ms_per_pass"] / len(manifest), 6),
                     "with_streaming_stores": {"achieved": standalone["streaming_nt"]["GBps"], "frac": round(standalone["streaming_nt"]["GBps"] / HBM_PEAK_GBS, 4)}},
        "cpu_baseline": None,
    }
    if not args.no_ceiling:
        out_b = 2 * sum(sh[0] * sh[1] for _, _, sh in manifest)          # bf16 results
        with_ceiling(line["roofline"], measured_ceiling(pkg, device), nbytes - out_b, out_b)
    return line


def run_fused(pkg, args, device):
    """The fused dequantize + linear kernels that ARE install()'s default for small inputs (reference ops.py:242-244, GGMLOps.Linear.forward_ggml_cast_weights),
    under the same clock as everything else: every quantized linear of the FLUX.1-dev set (304 layers, Q4_K_M mix, 6.79 GB packed -- 26x the Infinity
    Cache, so a pass streams every weight from HBM) called through the default policy (fused.linear_auto) with m rows of bf16 input, one pass =
    304 launches replayed from a captured HIP graph, HIP events around `passes` replays, median of `regions`.
      fused_small_m: m = 1, 4     fused_mfma: m = 32, 64, 256
    roofline: bound "hbm", achieved = PACKED bytes of the layers the fused path took / time -- these kernels read the packed weight once and write m x rows
    outputs, so the packed read is the algorithmic traffic; TFLOP/s beside it.  Layers the policy declines at that m (the auto
    rule hands tall weights above 128 rows to unpack + hipBLASLt) are listed and left out of both the bytes and the time.  Parity (outside the timed
    regions): three layers per m against an fp64 product on the ORACLE's weights, fp32-accumulation bound."""
    import numpy as np
    import oracle
    manifest = pkg.manifests.flux_dev(args.mix)
    if args.limit_tensors:
        manifest = manifest[:args.limit_tensors]
    dtype = torch.bfloat16
    tensors = []
    for i, (_, q, shape) in enumerate(manifest):
        n_blocks = pkg.synth.n_blocks_for(q, shape[0] * shape[1])
        tensors.append(pkg.ops.GGMLTensor(device_blocks(pkg, q, n_blocks, device, 7000 + i), tensor_type=q, tensor_shape=shape))
    packed_of = [t.as_subclass(torch.Tensor).numel() for t in tensors]
    passes = max(3, args.steps // 10)
    stream = torch.cuda.current_stream(device)
    unsupported = pkg.dequant.GGQUnsupported
    sample = sorted({(q, shape): i for i, (_, q, shape) in reversed(list(enumerate(manifest)))}.items(), key=lambda kv: kv[0][1][0] * kv[0][1][1])[:3]
    sample = [i for _, i in sample]                                       # the three smallest distinct (format, shape) layers: the CPU side stays in seconds
    dense64 = {}

    def check(i, x, y):
        _, q, (rows, cols) = manifest[i]
        if i not in dense64:
            raw = tensors[i].as_subclass(torch.Tensor).cpu().numpy()
            w = oracle.dequant_tensor(q, raw.reshape(-1, pkg.qtypes.block_geometry(q)[1]), "f16", "bf16")
            dense64[i] = torch.from_numpy(np.ascontiguousarray(w).view(np.int16).copy()).view(torch.bfloat16).reshape(rows, cols).double()
        w64, x64 = dense64[i], x.double().cpu()
        ref = x64 @ w64.T
        tol = cols * 2.0 ** -24 * (x64.abs() @ w64.abs().T) + 2.0 ** -8 * ref.abs() + 1e-30
        return bool(((y.double().cpu() - ref).abs() <= tol).all())

    out = {"fused_small_m": {}, "fused_mfma": {}}
    for group, ms in (("fused_small_m", (1, 4)), ("fused_mfma", (32, 64, 256))):
        for m in ms:
            xs = {}
            for _, _, (rows, cols) in manifest:
                if cols not in xs:
                    xs[cols] = torch.randn(m, cols, device=device, dtype=dtype) * 0.05
            taken, declined = [], []
            for i, t in enumerate(tensors):
                try:
                    pkg.fused.linear_auto(xs[t.shape[1]], t)
                    taken.append(i)
                except unsupported:
                    declined.append(i)
            torch.cuda.synchronize(device)

            def one_pass():
                return [pkg.fused.linear_auto(xs[tensors[i].shape[1]], tensors[i]) for i in taken]
            side = torch.cuda.Stream(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                keep = one_pass()
                torch.cuda.synchronize(device)
                with torch.cuda.graph(graph, stream=side):
                    keep = one_pass()
            graph.replay()
            torch.cuda.synchronize(device)
            ok = all(check(i, xs[tensors[i].shape[1]], keep[taken.index(i)]) for i in sample if i in taken)
            regs = []
            for _ in range(args.regions):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(passes):
                    graph.replay()
                b.record(stream)
                torch.cuda.synchronize(device)
                regs.append(a.elapsed_time(b) / passes)
            regs.sort()
            ms_pass = regs[len(regs) // 2]
            nbytes = sum(packed_of[i] for i in taken)
            flops = sum(2.0 * m * manifest[i][2][0] * manifest[i][2][1] for i in taken)
            gbs = nbytes / (ms_pass * 1e-3) / 1e9
            def kernel_of(i):                                    # which kernel fused.linear_auto + the library's own choice end up in ("Policy")
                rows_w, cols_w = manifest[i][2]
                bs, ts = pkg.qtypes.block_geometry(manifest[i][1])
                if m == 1 and rows_w < pkg.fused.SMALL_M_TALL_ROWS and cols_w // bs * ts + 15 <= 6 * 64 * 16:
                    return "ggq::linear_small"
                return "ggq::linear_mfma16" if m <= 8 or (m <= 16 and rows_w <= 4096) else "ggq::linear_mfma"
            kernels = sorted({kernel_of(i) for i in taken})
            out[group][f"m={m}"] = {
                "layers_fused": len(taken), "layers_declined": len(declined), "declined_shapes": sorted({"x".join(map(str, manifest[i][2])) for i in declined}),
                "ms_per_pass": round(ms_pass, 4), "us_per_layer": round(ms_pass * 1e3 / max(1, len(taken)), 2), "regions_ms": [round(r, 4) for r in regs],
                "TFLOPs": round(flops / (ms_pass * 1e-3) / 1e12, 1),
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                             "kernel": " + ".join(kernels), "algorithmic_bytes_per_pass": nbytes,
                             "algorithmic_bytes": "packed bytes of every fused layer, read once (outputs: m x rows x 2 B, < 1 % of that)"},
                "parity_vs_fp64_on_oracle_weights": f"within the fp32-accumulation bound ({sum(1 for i in sample if i in taken)} layers)" if ok else "OUTSIDE THE BOUND",
            }
            del graph, keep
            torch.cuda.empty_cache()
    out["how"] = (f"FLUX.1-dev {args.mix} linears ({len(manifest)} layers, {sum(packed_of) / 1e9:.2f} GB packed), bf16 input of m rows, the default policy fused.linear_auto per layer; "
                  f"one pass captured as a HIP graph, median of {args.regions} regions of {passes} replays, HIP events on the replay stream")
    return out


def run_fused_error(pkg, device):
    """--workload fused-error: NOT a throughput line -- the numerics table install()'s default stands on (tools/fused_error.py measure(): fused linears vs
    unpack + F.linear, both against an fp64 product on the oracle's weights, every linear shape of FLUX.1-dev / SD3.5-large / T5-xxl x {1, 4, 64, 256} rows x
    {bf16, fp16}).  `value` = the worst RMS-error ratio fused / default over all cases (1.0 = equal, lower is better)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fused_error", os.path.join(ROOT, "tools", "fused_error.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out = tool.measure(pkg, device)
    s = out["summary"]
    return {"metric": "worst RMS-error ratio, fused dequantize+linear / (unpack + F.linear), both vs an fp64 product on the oracle's weights", "value": s["worst_rms_ratio_fused_over_default"],
            "unit": "ratio", "n_gpus": 1, "steps": s["cases"], "warmup": 0, "ms_per_step": None, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "bf16+f16",
            "data": "synthetic", "config": {"workload": "fused-error: every distinct linear shape of FLUX.1-dev / SD3.5-large / T5-xxl x {1,4,64,256} rows x {bf16,f16}", "summary": s},
            "cases": out["cases"], "roofline": None, "cpu_baseline": None}


def median_region(pkg, plan, args, W, regions, steps=None, warmup=None):
    """`regions` timed regions of exactly K launches each (the first after W warm-up launches), every one bracketed by the
    fence on both sides and reduced with MAX over ranks; the reported step time is the MEDIAN region's.  Returns
    (ms_per_step of the median region [MAX over ranks], this rank's gpu ms per step in that region, wall ms per step, all regions)."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    rows = []
    for r in range(regions):
        gpu_ms, wall_ms = timed_steps(plan, steps, warmup if r == 0 else 0, W.device, W.fence)
        rows.append((W.max(gpu_ms / steps), gpu_ms / steps, wall_ms / steps))
    order = sorted(range(regions), key=lambda k: rows[k][0])
    med = rows[order[regions // 2]]
    return med[0], med[1], med[2], [round(r[0], 5) for r in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--qtype", default="Q4_K", help="headline quant type (default: the north-star target Q4_K)")
    ap.add_argument("--pairs", type=int, default=64, help="(3072x3072 + 3072x12288) pairs in the per-GPU pool")
    ap.add_argument("--no-per-qtype", action="store_true", help="skip the per-format table")
    ap.add_argument("--no-per-mode", action="store_true", help="skip the (dequant_dtype, dtype) table of the headline format")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the cpu_baseline legs (0 = skip)")
    ap.add_argument("--regions", type=int, default=3, help="timed regions of K steps each; the median region is reported")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the measured fill / copy / read ceilings (ggq_calibrate) beside the spec peak")
    ap.add_argument("--no-workloads", action="store_true", help="skip the configs[3] / configs[4] sub-lines of the default run")
    ap.add_argument("--workload", default="pool", choices=["pool", "flux", "sd35-t5", "flux-gguf", "per-layer", "fused-error", "fused"], help="see the module docstring")
    ap.add_argument("--mix", default="Q4_K_M", help="quant mix of the flux workloads (manifests.flux_dev)")
    ap.add_argument("--upload-threads", type=int, default=0, help="flux-gguf: reader threads of the streaming upload (0 = default 8)")
    ap.add_argument("--limit-tensors", type=int, default=0, help="flux-gguf / per-layer: only the first N tensors (smoke runs, tests)")
    ap.add_argument("--inproc-gpus", type=int, default=0, help="ONE process driving N GPUs (grouped.ShardedPlan) instead of one process per GPU; with --workload pool|flux|sd35-t5")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.inproc_gpus:
        if world != 1 or args.gpus != 1 or args.workload not in ("pool", "flux", "sd35-t5"):
            sys.exit("--inproc-gpus N is a single-process run (no torch.distributed launch, --gpus 1) of --workload pool, flux or sd35-t5")
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        pkg = load_package()
        pkg._native.lib()
        print(json.dumps(run_inproc(pkg, args, torch.device("cuda", 0))), flush=True)
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with "
                     f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # GGQ_BENCH_BACKEND=gloo (test rig): exercise the N > 1 code path on a box with fewer GPUs than ranks -- ranks share
    # devices round-robin and fence through gloo.  The driver's runs use the default: RCCL fences, one rank per GPU.
    rig = os.environ.get("GGQ_BENCH_BACKEND", "nccl") in ("gloo", "try-nccl")
    n_dev = torch.cuda.device_count()
    if not rig and local_rank >= n_dev:
        sys.exit(f"LOCAL_RANK={local_rank} but only {n_dev} GPU(s) visible: one rank per GPU")
    device = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(device)

    # launched by torch.distributed.run (RANK set) -> join the process group even for N=1, so the
    # 1-GPU run exercises the very same fence / MAX code the 2/4/8-GPU runs use
    W = World(rank, local_rank, world, device, launched=world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ))

    pkg = load_package()
    pkg._native.lib()                       # fail loudly if the HIP extension is missing
    qt = pkg.qtypes
    head_q = qt.Q[args.qtype]
    world_info = observed_world(W)          # every rank takes part; rank 0 reports it
    if world > 1 and not rig and world_info["distinct_devices"] < world:
        # an N-GPU line must come from N devices: refuse to print a rate that N ranks time-sharing fewer GPUs produced
        if rank == 0:
            print(f"bench.py: --gpus {world} but the ranks sit on {world_info['distinct_devices']} distinct device(s): "
                  f"{[(r['rank'], r['device'], r['uuid'] or r['pci_bus_id']) for r in world_info['ranks']]}", file=sys.stderr, flush=True)
        W.close()
        sys.exit(3)

    if args.workload != "pool":
        if args.workload in ("flux", "sd35-t5"):
            result = run_flux(pkg, args, W)
            if result is not None:
                result["world"] = world_info
        elif args.workload == "fused-error":
            if world != 1:
                sys.exit("--workload fused-error is a single-GPU measurement")
            result = run_fused_error(pkg, device)
        elif args.workload == "fused":
            if world != 1:
                sys.exit("--workload fused is a single-GPU measurement")
            fused = run_fused(pkg, args, device)
            head = fused["fused_mfma"]["m=32"]
            result = {"metric": "fused dequantize + linear, packed GB/s read (FLUX.1-dev set, 32 rows of x)", "value": head["roofline"]["achieved"], "unit": "GB/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_pass"], "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "config": {"workload": fused["how"]}, "roofline": head["roofline"],
                      "workloads": {"fused_small_m": fused["fused_small_m"], "fused_mfma": fused["fused_mfma"]}}
        elif args.workload == "per-layer":
            if world != 1:
                sys.exit("--workload per-layer is a single-GPU measurement")
            result = run_per_layer(pkg, args, device, W.fence)
            result.update({"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                           "dtype": "f16", "data": "synthetic"})
        else:
            if world != 1:
                sys.exit("--workload flux-gguf is a single-GPU measurement")
            result = run_flux_gguf(pkg, args, device)
        if rank == 0:
            print(json.dumps(result), flush=True)
        W.close()
        return

    manifest = global_manifest(pkg, head_q, args.pairs, world)
    indices = pkg.sharding.partition(manifest, world)[rank]
    mine = [manifest[i] for i in indices]
    plan = build_pool(pkg, mine, device, seed0=1000 * rank)
    bytes_rank = plan.bytes
    assert plan.kernels == 1

    ms_per_step, gpu_ms_step, wall_ms_step, regions = median_region(pkg, plan, args, W, args.regions)
    shards, cover = shard_report(pkg, manifest, indices, plan, gpu_ms_step, W)
    total_bytes = sum(sh["bytes"] for sh in shards)                        # what the ranks really moved (identical pools: N x rank 0's)
    value = total_bytes / (ms_per_step * 1e-3) / 1e9

    result = None
    if rank == 0:
        achieved = bytes_rank / (gpu_ms_step * 1e-3) / 1e9                # rank 0's own kernel, median region
        n_el = sum(s[0] * s[1] for _, _, s in mine)
        wl = f"BASELINE configs[2] {head_q.name}: FLUX.1-dev linear shapes, {args.pairs} x (3072x3072 + 3072x12288) per GPU"
        traffic, traffic_source = load_traffic(pkg, f"{head_q.name}:pairs{args.pairs}")
        med_ms, min_ms = per_launch_stats(plan, max(20, args.steps), device)
        in_bytes = sum(t.numel() for t in plan._keep)
        result = {
            "metric": "dequant GB/s (packed in -> fp16 out), (in+out) bytes / time",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": wl, "qtype": head_q.name, "elements_per_gpu": n_el, "bytes_per_step_per_gpu": bytes_rank,
                       "tensors_per_gpu": len(mine), "parallelism": f"tensor-list sharding x{world}, no collectives",
                       "pct_hbm_peak_per_gpu": round(100.0 * value / world / HBM_PEAK_GBS, 2),
                       "timed_regions_ms_per_step": regions, "reported": f"median of {len(regions)} timed regions of {args.steps} steps; ms_per_step = MAX over ranks",
                       "library_build": pkg._native.lib().ggq_build_id().decode(),
                       # the three rates of SURVEY.md section 8d, per GPU (rank 0): packed in / dense out / both, over the same time
                       "rates_GBps": {"in": round(in_bytes / (gpu_ms_step * 1e-3) / 1e9, 1),
                                      "out": round((bytes_rank - in_bytes) / (gpu_ms_step * 1e-3) / 1e9, 1),
                                      "in_plus_out": round(achieved, 1)}},
            "world": world_info,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": f"ggq::dequant_many<Fmt{head_q.name}, ...>", "algorithmic_bytes_per_launch": bytes_rank,
                         "avg_launch_ms": round(gpu_ms_step, 5), "median_launch_ms": round(med_ms, 5), "min_launch_ms": round(min_ms, 5),
                         "host_wall_ms_per_step": round(wall_ms_step, 5)},
        }
        if not args.no_ceiling:
            with_ceiling(result["roofline"], measured_ceiling(pkg, device), in_bytes, bytes_rank - in_bytes)
        if world > 1:
            result["config"]["shards"], result["config"]["shard_cover"] = shards, cover
            result["roofline"]["of"] = "rank 0's GPU"
    plan_head = plan

    per_qtype = {}
    if not args.no_per_qtype and rank == 0 and world == 1:
        steps_q = max(10, args.steps // 3)
        for q in qt.HIP_QTYPES:
            if q == head_q:
                per_qtype[q.name] = {"GB/s": result["roofline"]["achieved"], "pct_hbm_peak": round(100 * result["roofline"]["frac"], 2),
                                     "regions_GBps": [round(bytes_rank / (r * 1e-3) / 1e9, 1) for r in regions]}
                continue
            p = build_pool(pkg, pkg.manifests.flux_linear_pool(q, args.pairs), device, seed0=50_000 + 100 * int(q))
            _, ms, _, regs = median_region(pkg, p, args, W, args.regions, steps=steps_q, warmup=args.warmup)   # same fences, same median-of-regions as the headline
            gbs = p.bytes / (ms * 1e-3) / 1e9
            per_qtype[q.name] = {"GB/s": round(gbs, 1), "pct_hbm_peak": round(100 * gbs / HBM_PEAK_GBS, 2),
                                 "regions_GBps": [round(p.bytes / (r * 1e-3) / 1e9, 1) for r in regs]}
            p.close()
            del p
            torch.cuda.empty_cache()
        result["per_qtype"] = per_qtype
        result["per_qtype_how"] = f"every format: median of {args.regions} timed regions of {steps_q} launches (all listed), {args.warmup} warm-up launches, fences as the headline"

    if not args.no_per_mode and rank == 0 and world == 1:
        # the other (dequant_dtype -> dtype) combinations dequantize_tensor can be asked for (dequant.py:15-23,
        # nodes.py:186), same packed pool, fresh outputs; bytes = packed + dense bytes of THAT output dtype
        per_mode = {}
        steps_m = max(10, args.steps // 3)
        names = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}
        shapes = [tuple(o.shape) for o in plan_head.outputs]
        for cd in (torch.float16, torch.bfloat16, torch.float32):
            for od in (torch.float16, torch.bfloat16, torch.float32):
                if cd == torch.float16 and od == torch.float16:
                    continue
                p = pkg.grouped.DequantPlan([(d, head_q, sh) for d, sh in zip(plan_head._keep, shapes)], out_dtype=od, dequant_dtype=cd)
                _, ms, _, regs = median_region(pkg, p, args, W, args.regions, steps=steps_m, warmup=args.warmup)
                gbs = p.bytes / (ms * 1e-3) / 1e9
                per_mode[f"{names[cd]}->{names[od]}"] = {"GB/s": round(gbs, 1), "pct_hbm_peak": round(100 * gbs / HBM_PEAK_GBS, 2),
                                                         "regions_GBps": [round(p.bytes / (r * 1e-3) / 1e9, 1) for r in regs]}
                p.close()
                del p
                torch.cuda.empty_cache()
        result["per_mode"] = per_mode

    # parity of EVERY rank's timed launch + the CPU legs on rank 0 (outside every timed region; the other ranks wait at the next fence)
    if rank == 0:
        result["cpu_baseline"] = None
    if args.cpu_seconds > 0:
        base, port = cpu_baselines(pkg, plan_head, [head_q] * len(plan_head.outputs), args.cpu_seconds, W)
        if rank == 0:
            result["cpu_baseline"] = base
            if port is not None:
                result["cpu_baseline_port"] = port
            if world == 1:
                rg = reference_gpu_leg(pkg, plan_head, [head_q] * len(plan_head.outputs), device)
                if rg is not None:
                    rg["hip_path_speedup"] = round(result["roofline"]["achieved"] / rg["value"], 1)
                    result["reference_on_this_gpu"] = rg
    plan_head.close()
    del plan_head, plan
    torch.cuda.empty_cache()
    if not args.no_workloads:
        # BASELINE configs[3] and configs[4] measured in the SAME run (each is also a workload of its own: --workload flux /
        # sd35-t5): the full weight sets, one mixed-format plan launch per step, with their own roofline, parity and CPU legs.
        # For N > 1 these are the STRONG-scaling lines -- the same 304 / 549 tensors sharded over the N ranks (configs[4] is exactly
        # "SD3.5-large + T5-xxl sharded across N GPUs") -- and every rank takes part.
        subs = {}
        for wl_name in ("flux", "sd35-t5"):
            sub = run_flux(pkg, args, W, workload=wl_name, cpu_seconds=min(args.cpu_seconds, 6.0))
            if rank == 0:
                subs[wl_name] = {k: sub[k] for k in ("value", "unit", "n_gpus", "scaling", "ms_per_step", "config", "roofline", "cpu_baseline")}
            torch.cuda.empty_cache()
        if rank == 0 and world == 1:
            # ... the same FLUX set the way the node drives it: one dequantize_tensor() launch per layer, bf16 out (Next #2)
            subs["per_layer"] = run_per_layer(pkg, args, device, lambda: torch.cuda.synchronize(device))
            torch.cuda.empty_cache()
            # ... and configs[3] STREAMED: the same FLUX set as a synthetic .gguf file -> native parse -> threaded pread -> pinned -> HBM -> dense
            # (the PCIe-inclusive rate of the boundary; never `value`).  Needs ~7 GB of scratch disk: skipped with the reason if that fails.
            try:
                steps = args.steps
                args.steps = 2
                sub = run_flux_gguf(pkg, args, device)
                args.steps = steps
                subs["flux-gguf"] = {k: sub[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline")}
            except (OSError, RuntimeError, MemoryError) as e:
                args.steps = steps
                subs["flux-gguf"] = {"skipped": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
            # ... and the kernels that are install()'s default for SMALL inputs: the fused dequantize + linear launches, FLUX set, m = 1 / 4 and 32 / 64 / 256
            # rows of x, each with its own packed-read roofline (Next #1: half of the default path had no driver-run measurement)
            fused = run_fused(pkg, args, device)
            subs["fused_small_m"], subs["fused_mfma"] = dict(fused["fused_small_m"], how=fused["how"]), dict(fused["fused_mfma"], how=fused["how"])
            torch.cuda.empty_cache()
        elif rank == 0:
            subs["per_layer"] = subs["flux-gguf"] = subs["fused_small_m"] = subs["fused_mfma"] = {"skipped": "single-GPU measurements: see the --gpus 1 line"}
        if rank == 0:
            result["workloads"] = subs
    if rank == 0:
        print(json.dumps(result), flush=True)
    W.close()


if __name__ == "__main__":
    main()
