"""reference import path ``common_utils`` (only what the example scripts use: path_utils, stable_render_utils)"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import _bootstrap  # noqa: E402,F401
