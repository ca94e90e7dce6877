"""Data types that cross the hot path (comfyUI/types/hidden.py:249-351 EngineData; comfyUI/types/basic.py LATENT).
Containers only: every tensor is an ordinary torch tensor living in HBM."""
import itertools
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch

from .corrmap import CorrespondMap, IDMap

_SERIAL = itertools.count(1)


@dataclass
class Sprite:
    """common_utils/stable_render_utils/sprite.py:15-32"""
    spriteID: int
    prompt: str = ""
    prompt_weight: float = 1.0
    neg_prompt: str = ""
    neg_prompt_weight: float = 1.0


class SpriteInfos(dict):
    """{spriteID: Sprite} (sprite.py:35-39)"""


@dataclass
class EnvPrompt:
    """common_utils/stable_render_utils/prompts.py:4-19"""
    prompt: str = ""
    negative_prompt: str = ""
    weight: float = 1.0
    negative_weight: float = 1.0


class LATENT(dict):
    """{'samples': (N,4,h,w), 'noise': (N,4,h,w), 'noise_mask'?, 'batch_index'?}"""


@dataclass
class EngineData:
    frame_indices: List[int] = field(default_factory=list)
    color_maps: Optional[torch.Tensor] = None      # (N,H,W,3)
    id_maps: Optional[IDMap] = None                # (N,H,W,4) int32
    pos_maps: Optional[torch.Tensor] = None        # (N,H,W,3)
    normal_maps: Optional[torch.Tensor] = None     # (N,H,W,3)
    depth_maps: Optional[torch.Tensor] = None      # (N,H,W,3) (depth repeated)
    canny_maps: Optional[torch.Tensor] = None      # (N,H,W,3)
    noise_maps: Optional[LATENT] = None            # LATENT(samples=zeros, noise=pooled engine noise)
    masks: Optional[torch.Tensor] = None           # (N,H,W) = 1 - alpha
    correspond_maps: Optional[Dict[Tuple[int, int], CorrespondMap]] = None
    sprite_infos: Any = None
    env_prompts: Any = None
    serial: int = field(default_factory=lambda: next(_SERIAL))     # a fresh value per EngineData: EngineDataNode.IsChanged

    @property
    def frame_count(self):
        return len(self.frame_indices)


@dataclass
class InferenceOutput:
    frame_color: Optional[torch.Tensor] = None     # (N,H,W,3|4) in [0,1]
    extra: Dict[str, Any] = field(default_factory=dict)
