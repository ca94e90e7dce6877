from stable_renderer_amd.engine import GameObject  # noqa: F401  (reference: source/engine/runtime/gameObj.py)

__all__ = ["GameObject"]
