"""Reach the product package (for its qtype constants and synthetic-block generator only)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from ggq_pkg import load_package  # noqa: E402,F401
