"""reference: source/common_utils/stable_render_utils/__init__.py (data classes + corresponders)"""
from stable_renderer_amd.corresponder import Corresponder, DefaultCorresponder, OverlapCorresponder  # noqa: F401
from stable_renderer_amd.types import EnvPrompt, Sprite, SpriteInfos  # noqa: F401

__all__ = ["Corresponder", "DefaultCorresponder", "OverlapCorresponder", "EnvPrompt", "Sprite", "SpriteInfos"]
