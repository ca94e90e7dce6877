"""reference: source/engine/runtime/components/__init__.py"""
from stable_renderer_amd.engine import (AutoRotation, Camera, CameraController, CorrMapRenderer, EqualIntervalRotation,  # noqa: F401
                                        MeshRenderer, SpriteInfo, Transform)

__all__ = ["AutoRotation", "Camera", "CameraController", "CorrMapRenderer", "EqualIntervalRotation", "MeshRenderer", "SpriteInfo",
           "Transform"]
