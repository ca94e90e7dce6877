"""reference: source/engine/static/enums.py (the enums the example scripts import)"""
from stable_renderer_amd.engine import (DefaultTextureType, EngineMode, GLFW_Key, ProjectionType, RenderMode, RenderOrder,  # noqa: F401
                                        TextureDataType, TextureFilter, TextureFormat, TextureInternalFormat, TextureWrap)

__all__ = ["DefaultTextureType", "EngineMode", "GLFW_Key", "ProjectionType", "RenderMode", "RenderOrder", "TextureDataType",
           "TextureFilter", "TextureFormat", "TextureInternalFormat", "TextureWrap"]
