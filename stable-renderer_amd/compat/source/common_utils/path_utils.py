"""reference: source/common_utils/path_utils.py:1-70 -- the directory constants the example scripts build paths from.
PROJECT_DIR comes from $SR_PROJECT_DIR (default: this repository); RESOURCES_DIR from $SR_RESOURCES_DIR (default PROJECT_DIR /
resources); when there is no example-workflows directory there, the shipped graphs kept as test data are used."""
import os
from pathlib import Path as _Path

_PathType = type(_Path())


class Path(_PathType):
    '''String comparable Path object.'''

    def __eq__(self, other):
        if isinstance(other, str):
            return str(self.absolute()) == other or str(self) == other
        elif isinstance(other, _PathType):
            return str(self.absolute()) == str(other.absolute())
        return super().__eq__(other)

    __hash__ = _PathType.__hash__


_REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", ".."))
PROJECT_DIR = Path(os.environ.get("SR_PROJECT_DIR", _REPO))
SOURCE_DIR = PROJECT_DIR / 'source'
ENGINE_DIR = SOURCE_DIR / 'engine'
SHADER_DIR = ENGINE_DIR / 'shaders'
BUILTIN_WORKFLOW_DIR = ENGINE_DIR / 'workflows'
COMFYUI_DIR = SOURCE_DIR / 'comfyUI'
UI_DIR = SOURCE_DIR / 'ui'
RESOURCES_DIR = Path(os.environ.get("SR_RESOURCES_DIR", str(PROJECT_DIR / 'resources')))
EXAMPLE_3D_MODEL_DIR = RESOURCES_DIR / 'example-3d-models'
EXAMPLE_MAP_OUTPUT_DIR = RESOURCES_DIR / 'example-map-outputs'
EXAMPLE_WORKFLOWS_DIR = RESOURCES_DIR / 'example-workflows'
if not EXAMPLE_WORKFLOWS_DIR.exists():
    EXAMPLE_WORKFLOWS_DIR = Path(_REPO) / 'tests' / 'golden' / 'workflows'
TEMP_DIR = PROJECT_DIR / 'tmp'
COMFYUI_TEMP_DIR = TEMP_DIR / 'comfyui'
OUTPUT_DIR = PROJECT_DIR / 'output'
CACHE_DIR = OUTPUT_DIR / '.cache'
MAP_OUTPUT_DIR = OUTPUT_DIR / 'runtime_map'

__all__ = ['Path', 'PROJECT_DIR', 'SOURCE_DIR', 'ENGINE_DIR', 'SHADER_DIR', 'BUILTIN_WORKFLOW_DIR', 'COMFYUI_DIR', 'UI_DIR',
           'RESOURCES_DIR', 'EXAMPLE_3D_MODEL_DIR', 'EXAMPLE_MAP_OUTPUT_DIR', 'EXAMPLE_WORKFLOWS_DIR', 'TEMP_DIR', 'COMFYUI_TEMP_DIR',
           'OUTPUT_DIR', 'CACHE_DIR', 'MAP_OUTPUT_DIR']
