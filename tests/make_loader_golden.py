#!/usr/bin/env python3
"""Generate tests/golden/loader_cases.json by running the REFERENCE's own loader.py (verbatim, from /root/reference) on the case
files of tests/test_gguf.py -- TEST INFRASTRUCTURE, run in the build container (the only place /root/reference exists):

    python tests/make_loader_golden.py

The reference reads the files through a gguf-py-shaped adapter over this package's native parser (gguf-py itself is absent), so the
fixture pins everything the loader does WITH a parsed file, not the container parsing (tests/test_gguf.py header)."""
import os
import pathlib
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

from ggq_pkg import load_package  # noqa: E402
import test_gguf  # noqa: E402


def main():
    out = os.path.join(HERE, "golden", "loader_cases.json")
    with tempfile.TemporaryDirectory(prefix="ggq_loader_golden_") as tmp:
        cases = test_gguf.write_loader_golden(load_package(), out, pathlib.Path(tmp))
    print(out, {k: sum("error" in c for c in v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
