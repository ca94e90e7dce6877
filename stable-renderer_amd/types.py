"""Data types that cross the hot path (comfyUI/types/hidden.py:249-351 EngineData; comfyUI/types/basic.py LATENT).
Containers only: every tensor is an ordinary torch tensor living in HBM."""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import torch

from .corrmap import CorrespondMap, IDMap


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

    @property
    def frame_count(self):
        return len(self.frame_indices)


@dataclass
class InferenceOutput:
    frame_color: Optional[torch.Tensor] = None     # (N,H,W,3|4) in [0,1]
    extra: Dict[str, Any] = field(default_factory=dict)
