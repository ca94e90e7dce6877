"""reference: source/common_utils/path_utils.py:1-70 -- the directory constants the example scripts build paths from.
PROJECT_DIR comes from $SR_PROJECT_DIR (default: this repository); RESOURCES_DIR from $SR_RESOURCES_DIR (default PROJECT_DIR /
resources); when there is no example-workflows directory there, the shipped graphs kept as test data are used."""
import os
from pathlib import Path as _Path

_Concrete = type(_Path())          # PosixPath here; subclassing the concrete flavour keeps `/` returning this class


def _text_forms(p):
    """the spellings a path may be compared by: as written, and resolved against the working directory"""
    return {os.fspath(p), os.path.abspath(os.fspath(p))}


class Path(_Concrete):
    """pathlib path that also compares equal to a plain string naming the same location (the example scripts compare
    these constants with str arguments); two paths are equal when their absolute spellings agree"""

    def __eq__(self, other):
        if isinstance(other, (str, os.PathLike)):
            return bool(_text_forms(self) & _text_forms(other)) if isinstance(other, str) else \
                os.path.abspath(os.fspath(self)) == os.path.abspath(os.fspath(other))
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = _Concrete.__hash__


_REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", ".."))
PROJECT_DIR = Path(os.environ.get("SR_PROJECT_DIR", _REPO))
RESOURCES_DIR = Path(os.environ.get("SR_RESOURCES_DIR", os.path.join(str(PROJECT_DIR), "resources")))
# constant name -> (parent constant, leaf): the layout the reference's scripts expect under the project root
_LAYOUT = (("SOURCE_DIR", "PROJECT_DIR", "source"), ("ENGINE_DIR", "SOURCE_DIR", "engine"), ("SHADER_DIR", "ENGINE_DIR", "shaders"),
           ("BUILTIN_WORKFLOW_DIR", "ENGINE_DIR", "workflows"), ("COMFYUI_DIR", "SOURCE_DIR", "comfyUI"), ("UI_DIR", "SOURCE_DIR", "ui"),
           ("EXAMPLE_3D_MODEL_DIR", "RESOURCES_DIR", "example-3d-models"), ("EXAMPLE_MAP_OUTPUT_DIR", "RESOURCES_DIR", "example-map-outputs"),
           ("EXAMPLE_WORKFLOWS_DIR", "RESOURCES_DIR", "example-workflows"), ("TEMP_DIR", "PROJECT_DIR", "tmp"),
           ("COMFYUI_TEMP_DIR", "TEMP_DIR", "comfyui"), ("OUTPUT_DIR", "PROJECT_DIR", "output"), ("CACHE_DIR", "OUTPUT_DIR", ".cache"),
           ("MAP_OUTPUT_DIR", "OUTPUT_DIR", "runtime_map"))
for _name, _parent, _leaf in _LAYOUT:
    globals()[_name] = globals()[_parent] / _leaf
if not EXAMPLE_WORKFLOWS_DIR.exists():                           # noqa: F821 (defined by the loop above)
    EXAMPLE_WORKFLOWS_DIR = Path(_REPO) / 'tests' / 'golden' / 'workflows'

__all__ = ['Path', 'PROJECT_DIR', 'SOURCE_DIR', 'ENGINE_DIR', 'SHADER_DIR', 'BUILTIN_WORKFLOW_DIR', 'COMFYUI_DIR', 'UI_DIR',
           'RESOURCES_DIR', 'EXAMPLE_3D_MODEL_DIR', 'EXAMPLE_MAP_OUTPUT_DIR', 'EXAMPLE_WORKFLOWS_DIR', 'TEMP_DIR', 'COMFYUI_TEMP_DIR',
           'OUTPUT_DIR', 'CACHE_DIR', 'MAP_OUTPUT_DIR']
