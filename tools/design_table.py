"""ONE source for the figures the documents quote (Next #7).  Every current-state number in DESIGN.md, INTEGRATION.md and
README.md lives inside a generated block

    <!-- numbers:begin NAME (tools/design_table.py) -->  ...  <!-- numbers:end NAME -->

whose text this script derives from the committed measurement files and nothing else:

    profiles/rNN_bench_n1.json                              one `python bench.py` line (the newest profiles/r*_bench_n1.json unless given)
    profiles/pmc_traffic.json                               PMC bytes per launch
    profiles/rNN_flux_forward_emulation_token_sweep.json    tools/token_sweep.py (and rNN_forward_emulation_token_sweep_sd35.json / _t5.json), newest round
    profiles/rNN_fused_error.json                           tools/fused_error.py, newest round

    python tools/design_table.py [bench.json]            # prints every block
    python tools/design_table.py --write                 # rewrites the blocks in the three documents
    python tools/design_table.py --check                 # exit 1 if a document's block differs from what the files say (tests/test_docs.py)

Numbers OUTSIDE the blocks are history: each carries the profiles/ file it came from and the round it was measured in.
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "INTEGRATION.md", "README.md")
BEGIN = "<!-- numbers:begin {name} (tools/design_table.py) -->"
END = "<!-- numbers:end {name} -->"


def newest_bench():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_n1.json")))
    return files[-1]


def load(path):
    with open(path) as f:
        text = f.read().strip()
    return json.loads(text.splitlines()[-1]) if text.count("\n") and not text.startswith("{\n") and not text.startswith("[") else json.loads(text)


def rel(path):
    return os.path.relpath(path, ROOT)


BYTES_PER_ELEMENT_IN = {"Q4_0": 18 / 32, "Q4_1": 20 / 32, "Q5_0": 22 / 32, "Q5_1": 24 / 32, "Q8_0": 34 / 32, "Q2_K": 84 / 256, "Q3_K": 110 / 256, "Q4_K": 144 / 256,
                        "Q5_K": 176 / 256, "Q6_K": 210 / 256, "IQ4_NL": 18 / 32, "IQ4_XS": 136 / 256}
POOL_ELEMENTS = 64 * (3072 * 3072 + 3072 * 12288)


def traffic_span(tr, modes):
    """PMC bytes / algorithmic bytes over the pool entries of pmc_traffic.json: the formats (fp16 result) or Q4_K's other modes; outliers above 1 % are named."""
    ratios = {}
    for key, val in tr.items():
        parts = key.split(":")
        if len(parts) < 2 or parts[1] != "pairs64" or parts[0] not in BYTES_PER_ELEMENT_IN:
            continue
        if modes != (len(parts) == 3):
            continue
        out_bytes = 4 if (len(parts) == 3 and parts[2].endswith("->f32")) else 2
        ratios[parts[0] if not modes else parts[2]] = val / (POOL_ELEMENTS * (BYTES_PER_ELEMENT_IN[parts[0]] + out_bytes))
    if not ratios:
        return "—"
    usual = [r for r in ratios.values() if r < 1.01]
    odd = ", ".join(f"{k} {r:.3f}" for k, r in ratios.items() if r >= 1.01)
    span = f"{min(usual):.4f}–{max(usual):.4f}" if usual else ""
    return f"{span} ({len(ratios)} measured" + (f"; {odd}: two XCDs fetch the line two groups share" if odd else "") + ")"


def headline_table(d, tr, src):
    pq, pm, w = d["per_qtype"], d["per_mode"], d["workloads"]
    rf = d["roofline"]
    rows = [f"Source: `{src}` (one `python bench.py` run on one MI355X box), `profiles/pmc_traffic.json`.", "",
            "| what | GB/s (in+out) | fraction of 8 TB/s · of the measured blend ceiling | PMC traffic ÷ algorithmic |", "|---|---|---|---|"]
    blend = f" · **{rf['frac_of_blend']:.3f}** of {rf['blend_ceiling_GBps']:.0f}" if "frac_of_blend" in rf else ""
    tq = tr.get("Q4_K:pairs64")
    rows.append(f"| **headline: Q4_K pool → fp16** | **{d['value']}** ({d['ms_per_step']} ms; regions {d['config']['timed_regions_ms_per_step']}) | **{rf['frac']:.3f}**{blend} | "
                f"{(tq / rf['algorithmic_bytes_per_launch']) if tq else float('nan'):.4f} |")
    if "measured_fill_GBps" in rf:
        rows.append(f"| measured in the same process (`ggq_calibrate`, 2 GiB): fill {rf['measured_fill_GBps']:.0f} · copy {rf['measured_copy_GBps']:.0f} (read + write) · read {rf['measured_read_GBps']:.0f} | "
                    f"blend for the headline's {100 * rf['blend_mix']['read_share']:.0f} % read / {100 * rf['blend_mix']['write_share']:.0f} % write mix: {rf['blend_ceiling_GBps']:.0f} | — | — |")
    rows.append("| per format → fp16: " + ", ".join(f"{k} {v['GB/s']:.0f}" for k, v in pq.items())
                + f" | {min(v['GB/s'] for v in pq.values()):.0f}–{max(v['GB/s'] for v in pq.values()):.0f} | "
                  f"{min(v['pct_hbm_peak'] for v in pq.values()) / 100:.2f}–{max(v['pct_hbm_peak'] for v in pq.values()) / 100:.2f} | {traffic_span(tr, modes=False)} |")
    rows.append("| Q4_K, other (arithmetic→out) modes: " + ", ".join(f"{k} {v['GB/s']:.0f}" for k, v in pm.items())
                + f" | {min(v['GB/s'] for v in pm.values()):.0f}–{max(v['GB/s'] for v in pm.values()):.0f} | "
                  f"{min(v['pct_hbm_peak'] for v in pm.values()) / 100:.2f}–{max(v['pct_hbm_peak'] for v in pm.values()) / 100:.2f} | {traffic_span(tr, modes=True)} |")
    for k, lab in (("flux", "FLUX.1-dev Q4_K_M weight set (304 tensors, 2 launches)"), ("sd35-t5", "SD3.5-large + T5-XXL Q4_K_M (549 tensors, 3 launches)")):
        r = w[k]["roofline"]
        t = tr.get(k + ":Q4_K_M")
        fb = f" · {r['frac_of_blend']:.3f}" if "frac_of_blend" in r else ""
        rows.append(f"| {lab} → fp16 | {w[k]['value']} ({w[k]['ms_per_step']} ms) | {r['frac']:.3f}{fb} | {(t / r['algorithmic_bytes_per_launch']) if t else float('nan'):.4f} |")
    pl = w["per_layer"]
    c = pl["config"]
    sa = c["standalone_gpu_bound"]
    fb = f" · {pl['roofline']['frac_of_blend']:.3f} of the blend" if "frac_of_blend" in pl["roofline"] else ""
    rows.append(f"| **per layer**: the same FLUX set, one `dequantize_tensor` launch per tensor (304), bf16, graph-replayed, nothing reading the results | **{pl['value']}** sc1 stores "
                f"(shipped; {sa['shipped_sc1']['us_per_launch']} µs/launch) · {sa['streaming_nt']['GBps']} non-temporal · eager {c['eager_GBps']} "
                f"(host enqueue {c['eager_host_enqueue_us_per_call']} µs/call) | **{pl['roofline']['frac']:.3f}**{fb} · {pl['roofline']['with_streaming_stores']['frac']:.3f} non-temporal | — |")
    ic = c["in_context"]
    rows.append(f"| **in context**: emulated FLUX step, 4608 tokens, dense-resident {ic['ms_per_step_dense_resident']} ms | dequant cost per step: "
                f"**{ic['shipped_sc1']['dequant_cost_ms_per_step']} ms** sc1 (shipped) vs {ic['streaming_nt']['dequant_cost_ms_per_step']} ms non-temporal (rounds 1–2) | — | — |")
    rg, rl = d.get("reference_on_this_gpu"), c.get("reference_on_this_gpu")
    if rg and rl:
        rows.append(f"| **the reference's own eager torch ops on this GPU** (verbatim `dequant.py`, same device tensors, every result bit-equal to the HIP path's) | pool pair {rg['value']} "
                    f"(HIP path {rg['hip_path_speedup']}×) · FLUX set per layer {rl['standalone_GBps']} ({rl['standalone_ms_per_pass']} ms per pass; HIP path {rl['hip_path_speedup_standalone_eager']}×) · "
                    f"in context {rl['in_context_ms_per_step']} ms per step = {rl['in_context_dequant_cost_ms_per_step']} ms of dequant (whole step {rl['hip_path_step_speedup_in_context']}× faster with the HIP path) "
                    f"| {rg['value'] / 8000:.3f} | — |")
    fg = w["flux-gguf"]
    if "config" in fg:
        rows.append(f"| `flux-gguf`: native parse + threaded pread→pinned→H2D + one dequant pass over a synthetic 6.8 GB FLUX .gguf | upload {fg['config']['upload_GBps_packed']} GB/s packed "
                    f"({fg['config']['load_ms_best']} ms), dequant {fg['config']['dequant_GBps']} ({fg['config']['dequant_ms']} ms) | {fg['roofline']['frac']:.3f} of PCIe Gen5 x16 (63 GB/s) — link-bound, never `value` | — |")
    cb = d.get("cpu_baseline") or {}
    if cb:
        rng = cb.get("range_GBps")
        span = (f"; range over thread counts and passes {rng['lowest_median_over_thread_counts']}–{rng['best_single_pass']} (page-fault-bound, not a point)" if rng else "")
        rows.append(f"| `cpu_baseline` ({cb['kind']}): the reference's torch-CPU `dequantize()` on the box's host ({cb.get('host_cpus', '?')} CPUs), best median at {cb['cores']} threads | "
                    f"{cb['value']}{span} | — | — |")
    return "\n".join(rows)


def summary(d, src):
    """The five figures INTEGRATION.md and README.md lead with."""
    rf = d["roofline"]
    pl = d["workloads"]["per_layer"]
    ic = pl["config"]["in_context"]
    rl = pl["config"].get("reference_on_this_gpu") or {}
    rows = [f"From `{src}` (one box, one run):",
            f"- whole-weight-set launch, Q4_K → fp16: **{d['value']} GB/s** = **{rf['frac']:.3f}** of the 8 TB/s HBM3E peak"
            + (f", **{rf['frac_of_blend']:.3f}** of what this box's memory system gives a stream of the same read/write mix ({rf['blend_ceiling_GBps']:.0f} GB/s, measured in the same process)" if "frac_of_blend" in rf else "") + ";",
            f"- one launch per layer (how ComfyUI drives it), FLUX.1-dev set → bf16: {pl['value']} GB/s = {pl['roofline']['frac']:.3f} of peak"
            + (f", {pl['roofline']['frac_of_blend']:.3f} of the blend ceiling" if "frac_of_blend" in pl["roofline"] else "") + ";",
            f"- in context (emulated FLUX.1-dev step, 4608 tokens, bf16, `exact` install): {ic['shipped_sc1']['ms_per_step']} ms against {ic['ms_per_step_dense_resident']} ms with fully dense-resident weights "
            f"= **{ic['shipped_sc1']['dequant_cost_ms_per_step']} ms** for re-dequantizing all 304 linears every step;"]
    if rl:
        rows.append(f"- the reference's own eager torch path on the same GPU, same tensors, bit-equal results: {rl['in_context_ms_per_step']} ms per step "
                    f"({rl['in_context_dequant_cost_ms_per_step']} ms of it dequant) — the whole step is {rl['hip_path_step_speedup_in_context']}× faster on the HIP path.")
    return "\n".join(rows)


def token_table(ts, src):
    rows = [f"Source: `{src}` (`tools/token_sweep.py`, {ts['workload']}, {ts['device']}).", "",
            "| tokens | `exact` (unpack + F.linear everywhere) | **default** (fused ≤ 4 rows + fused MFMA ≤ 256 rows) | dense-resident | default − dense |", "|---|---|---|---|---|"]
    for t, r in ts["by_tokens"].items():
        rows.append(f"| {t} | {r['exact_ms']} | **{r['default_ms']}** | {r['dense_resident_ms']} | {r['default_minus_dense_ms']:+.2f} |")
    return "\n".join(rows)


def fused_error_block(fe, src):
    s = fe.get("summary") or fe["config"]["summary"]          # tools/fused_error.py's own output, or the bench line of `bench.py --workload fused-error`
    kernels = {}
    for c in fe["cases"]:
        if c.get("fused"):
            kernels[c["fused"]] = kernels.get(c["fused"], 0) + 1
    by_kernel = ", ".join(f"`{k}` {v}" for k, v in sorted(kernels.items()))
    dec = {}
    for c in fe["cases"]:
        if not c.get("fused"):
            dec.setdefault(f"{c['qtype']} {c['rows']}×{c['cols']}", set()).add(c["m"])
    declined = "; ".join(f"{k} at {'/'.join(str(m) for m in sorted(v))} rows" for k, v in sorted(dec.items())) or "none"
    return "\n".join([
        f"Source: `{src}` (`bench.py --workload fused-error` = `tools/fused_error.py`, MI355X): {s['cases']} cases = every distinct linear shape of FLUX.1-dev / SD3.5-large / T5-xxl × "
        f"{{1, 4, 64, 256}} rows × {{bf16, fp16}}, error against an fp64 product on the ORACLE's weights, relative to RMS(exact).",
        f"- fused kernel ran in {s['fused_ran']} cases ({by_kernel}); the other {s['declined']} are declined by the auto policy and keep unpack + `F.linear`, which is faster there: {declined};",
        f"- worst RMS-error ratio fused ÷ default: **{s['worst_rms_ratio_fused_over_default']:.7f}**; worst max-error ratio: **{s['worst_max_ratio_fused_over_default']:.4f}**; "
        f"max error never above the default path's by more than {max(0.0, s['worst_max_excess_in_output_ulps']):.2f} output rounding steps;",
        f"- outputs bit-identical to the default path's: ≥ {100 * s['min_same_bits_share']:.2f} % in every case; run-to-run: fused non-deterministic in {s['fused_nondeterministic']} cases, default in {s['default_nondeterministic']};",
        f"- verdict of the rule in `tools/fused_error.py` (RMS within 2 %, max within one output rounding step, deterministic): **fused_no_worse = {s['fused_no_worse']}**."])


def fused_roofline(d, src):
    """The fused dequantize + linear kernels under the bench's own clock (workloads.fused_small_m / fused_mfma of the default run): packed-read roofline per m."""
    w = d["workloads"]
    rows = [f"Source: `{src}` `workloads.fused_small_m` / `workloads.fused_mfma` ({w['fused_mfma']['how']}).", "",
            "| rows of x | kernel(s) | layers fused (declined) | ms per pass · µs per layer | packed GB/s read | fraction of 8 TB/s | TFLOP/s |", "|---|---|---|---|---|---|---|"]
    for group in ("fused_small_m", "fused_mfma"):
        for k, v in w[group].items():
            if k == "how":
                continue
            r = v["roofline"]
            rows.append(f"| {k[2:]} | `{r['kernel']}` | {v['layers_fused']} ({v['layers_declined']}) | {v['ms_per_pass']} · {v['us_per_layer']} | **{r['achieved']:.0f}** | **{r['frac']:.3f}** | {v['TFLOPs']} |")
    return "\n".join(rows)


def newest_profile(suffix):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return files[-1] if files else None


def blocks(bench_path=None):
    bench_path = bench_path or newest_bench()
    d = load(bench_path)
    tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    out = {"headline-table": headline_table(d, tr, rel(bench_path)), "summary": summary(d, rel(bench_path))}
    if "fused_mfma" in d.get("workloads", {}) and "how" in d["workloads"]["fused_mfma"]:
        out["fused-roofline"] = fused_roofline(d, rel(bench_path))
    for name, fname in (("token-sweep", "flux_forward_emulation_token_sweep.json"), ("token-sweep-sd35", "forward_emulation_token_sweep_sd35.json"),
                        ("token-sweep-t5", "forward_emulation_token_sweep_t5.json")):
        p = newest_profile(fname)
        if p:
            out[name] = token_table(load(p), rel(p))
    p = newest_profile("fused_error.json")
    if p:
        out["fused-error"] = fused_error_block(load(p), rel(p))
    return out


def apply(text, blk):
    """(new text, names found) -- every numbers block of ``text`` replaced by the generated one."""
    found = []

    def sub(m):
        name = m.group(1)
        found.append(name)
        if name not in blk:
            raise SystemExit(f"unknown numbers block {name!r}")
        return BEGIN.format(name=name) + "\n" + blk[name] + "\n" + END.format(name=name)
    pat = re.compile(r"<!-- numbers:begin (\S+) \(tools/design_table\.py\) -->.*?<!-- numbers:end \1 -->", re.S)
    return pat.sub(sub, text), found


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    blk = blocks(args[0] if args else None)
    stale = []
    for doc in DOCS:
        p = os.path.join(ROOT, doc)
        old = open(p).read()
        new, found = apply(old, blk)
        if new != old:
            stale.append(doc)
            if "--write" in sys.argv:
                open(p, "w").write(new)
    if "--check" in sys.argv:
        if stale:
            sys.exit(f"numbers blocks out of date in {stale}: run `python tools/design_table.py --write`")
        return
    if "--write" not in sys.argv:
        for name, text in blk.items():
            print(f"==== {name}\n{text}\n")


if __name__ == "__main__":
    main()
