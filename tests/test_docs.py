"""The documents quote measured figures from ONE place (Next #7): every current-state number sits in a
``<!-- numbers:begin NAME (tools/design_table.py) -->`` block that tools/design_table.py derives from the committed measurement files
(profiles/r05_bench_n1.json, pmc_traffic.json, the token sweep, the fused-error table).  A block that says something those files do not
say fails here -- `python tools/design_table.py --write` regenerates them."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("design_table", os.path.join(ROOT, "tools", "design_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_numbers_block_is_what_the_measurement_files_say():
    T = _tool()
    blk = T.blocks()
    want = {"DESIGN.md": {"headline-table", "fused-error"}, "INTEGRATION.md": {"summary", "token-sweep", "token-sweep-sd35", "token-sweep-t5", "fused-error"}, "README.md": {"summary"}}
    for doc in T.DOCS:
        old = open(os.path.join(ROOT, doc)).read()
        new, found = T.apply(old, blk)
        assert want[doc] <= set(found), f"{doc}: missing generated blocks {want[doc] - set(found)}"
        if new != old:
            a, b = old.splitlines(), new.splitlines()
            first = next(i for i, (x, y) in enumerate(zip(a, b)) if x != y) if len(a) == len(b) else min(len(a), len(b))
            raise AssertionError(f"{doc}: a numbers block is out of date (first differing line {first + 1}); run `python tools/design_table.py --write`")


def test_the_bench_file_the_blocks_quote_is_a_contract_line_of_this_tree():
    """The quoted bench line carries the round-5 fields (measured ceiling beside the spec peak, cpu_baseline as a range with host_cpus / threads)."""
    T = _tool()
    d = T.load(T.newest_bench())
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "measured_fill_GBps", "measured_copy_GBps", "measured_read_GBps", "blend_ceiling_GBps", "frac_of_blend"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and abs(rf["frac_of_blend"] - rf["achieved"] / rf["blend_ceiling_GBps"]) < 1e-3
    pl = d["workloads"]["per_layer"]["roofline"]
    assert "frac_of_blend" in pl and "blend_ceiling_GBps" in pl
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and isinstance(cb["host_cpus"], int) and cb["threads"] == cb["cores"] and "range_GBps" in cb


def test_every_c_identifier_the_documents_name_is_declared_in_the_headers():
    """DESIGN.md / INTEGRATION.md / README.md talk about the C ABI by name; a ``ggq_*`` identifier in them that include/*.h does not declare (an entry point
    deleted rounds ago, a typo) fails here (Weak #8: DESIGN.md listed ggq_dequant_batch after ABI 10 had removed it)."""
    import re
    declared = set()
    for h in ("ggq.h", "ggq_gguf.h"):
        declared |= set(re.findall(r"\bggq_[a-z0-9_]+\b", open(os.path.join(ROOT, "include", h)).read()))
    # names that are not C identifiers of the ABI: the python package's own module / file / op names, build artefacts, lab tools
    known_other = {"ggq_pkg", "ggq_hip", "ggq_fast", "ggq_capi", "ggq_gguf", "ggq_linear", "ggq_overlap", "ggq_pyfast", "ggq_device", "ggq_mfma", "ggq_mfma16", "ggq_gemm",
                   "ggq_host", "ggq_oracle", "ggq_oracle_simd", "ggq_microbench", "ggq_lab_engine", "ggq_stream", "ggq_scratch", "ggq_build_", "ggq_refpkg",
                   "ggq_gemm64", "ggq_reference_dequant"}
    bad = {}
    for doc in ("DESIGN.md", "INTEGRATION.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for name in set(re.findall(r"\bggq_[a-z0-9_]+\b", text)):
            if name in declared or name in known_other or name.rstrip("_") in known_other:
                continue
            if re.search(r"lib" + name + r"\b|" + name + r"\.(?:hip|hpp|h|c|py|so)\b|_" + name + r"\b", text):     # a file name (libggq_hip.so, ggq_capi.hip, _ggq_fast)
                continue
            bad.setdefault(doc, []).append(name)
    assert not bad, f"identifiers the headers do not declare: {bad}"
