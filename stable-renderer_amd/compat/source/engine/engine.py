from stable_renderer_amd.engine import Engine  # noqa: F401  (reference: source/engine/engine.py)

__all__ = ["Engine"]
