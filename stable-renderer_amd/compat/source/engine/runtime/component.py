from stable_renderer_amd.engine import Component  # noqa: F401  (reference: source/engine/runtime/component.py)

__all__ = ["Component"]
