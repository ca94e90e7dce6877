"""reference import path ``engine`` -> stable_renderer_amd.engine (see stable-renderer_amd/compat/__init__.py)"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import _bootstrap  # noqa: E402,F401
