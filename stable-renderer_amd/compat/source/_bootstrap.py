"""makes ``stable_renderer_amd`` importable from the shim packages (repo root = three levels up)"""
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import stable_renderer_amd  # noqa: E402,F401
