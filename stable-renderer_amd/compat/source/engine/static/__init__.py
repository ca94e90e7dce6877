"""reference: source/engine/static/__init__.py (``from .enums import *`` + the resource classes)"""
from stable_renderer_amd.corrmap import CorrespondMap, IDMap  # noqa: F401
from stable_renderer_amd.engine import Material, Material_MTL, Mesh, Texture  # noqa: F401
from .enums import *  # noqa: F401,F403
from . import enums as _e

__all__ = ["CorrespondMap", "IDMap", "Material", "Material_MTL", "Mesh", "Texture"] + list(_e.__all__)
