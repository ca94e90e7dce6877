"""Corresponder protocol and the two shipped implementations, same names / attributes / call contract as the
reference (common_utils/stable_render_utils/corresponder.py:29-376).  The cross-frame work runs as HIP kernels:

* ``OverlapCorresponder.step_finished`` -> ``sr_overlap_step`` (gather, per-vertexID mean, blend, last-writer-wins
  scatter, AdaIN) on the device-resident latent, using the per-call ``OverlapIndex`` built from the id maps;
* ``OverlapCorresponder.pre_atten_inject`` is realised at plan-build time: the UNet plan projects K/V only from the
  randomly chosen batch entries and every batch entry attends to them (``DiffusionRunner.sample(inject_n_rand=...)``);
* ``DefaultCorresponder.finished`` -> ``CorrespondMap.update`` (``sr_corrmap_update``).
"""
from typing import Any, Protocol

import torch


class Corresponder(Protocol):
    def prepare(self, engine_data: Any): ...
    def pre_atten_inject(self, block, engine_data, q_context, k_context, v_context, layer): ...
    def post_atten_inject(self, block, engine_data, origin_values, layer): ...
    def step_finished(self, engine_data, sampling_context): ...
    def finished(self, engine_data, images): ...


def _update_corrmaps(self, engine_data, images):
    if not self.update_corrmap or images is None or engine_data.id_maps is None:
        return
    id_maps = engine_data.id_maps.tensor
    masks = engine_data.id_maps.masks            # id-map masks (1 = no id), not EngineData.masks
    if masks is None:
        masks = torch.ones(id_maps.shape[:-1], dtype=torch.bool, device=id_maps.device)
    corrmaps = engine_data.correspond_maps
    if corrmaps:
        for (spriteID, materialID), corrmap in corrmaps.items():
            corrmap.update(color_frames=images, id_maps=id_maps, mode=self.update_corrmap_mode, masks=masks,
                           spriteID=spriteID, materialID=materialID,
                           ignore_obj_mat_id=getattr(self, "ignore_obj_mat_id_when_update", False), inverse_masks=True)


class DefaultCorresponder:
    """corresponder.py:100-155: no cross-frame work while sampling; ``finished`` bakes the decoded frames."""

    def __init__(self, layer_range=(6,), update_corrmap=True, update_corrmap_mode='first_avg', post_attn_inject_ratio=0.6,
                 ignore_obj_mat_id_when_update=False):
        self.layer_range = tuple(layer_range)
        self.update_corrmap = update_corrmap
        self.update_corrmap_mode = update_corrmap_mode
        self.post_attn_inject_ratio = post_attn_inject_ratio
        self.ignore_obj_mat_id_when_update = ignore_obj_mat_id_when_update

    def post_atten_inject(self, block, engine_data, origin_values, layer):
        return origin_values                     # the reference returns early (corresponder.py:124)

    finished = _update_corrmaps


class OverlapCorresponder:
    """corresponder.py:157-376."""

    def __init__(self, layer_range=(6,), update_corrmap=True, update_corrmap_mode='first',
                 pre_attn_inject_num_random_frames=1, post_attn_inject_ratio=0.6, step_finished_inject_ratio=0.1,
                 step_finished_stop_inject_timestep=500):
        self.layer_range = tuple(layer_range)
        self.update_corrmap = update_corrmap
        self.update_corrmap_mode = update_corrmap_mode
        self.pre_attn_inject_num_random_frames = pre_attn_inject_num_random_frames
        self._random_frame_indices = None
        self.post_attn_inject_ratio = post_attn_inject_ratio
        self.step_finished_inject_ratio = step_finished_inject_ratio
        self.step_finished_stop_inject_timestep = step_finished_stop_inject_timestep

    def prepare(self, engine_data):
        pass

    def pre_atten_inject(self, block, engine_data, q_context, k_context, v_context, layer):
        """Tensor-level form kept for API compatibility (zero-copy views; corresponder.py:188-220).  The sampling path
        does not call it per block: the same selection is compiled into the UNet plan."""
        if self.pre_attn_inject_num_random_frames < 0:
            return q_context, k_context, v_context
        if self._random_frame_indices is None:
            self._random_frame_indices = torch.randint(1, k_context.shape[0], (self.pre_attn_inject_num_random_frames,))
        idx = [int(i) for i in self._random_frame_indices]
        k = torch.cat([k_context[i] for i in idx], dim=0).unsqueeze(0).expand(k_context.shape[0], -1, -1)
        v = torch.cat([v_context[i] for i in idx], dim=0).unsqueeze(0).expand(v_context.shape[0], -1, -1)
        return q_context, k, v

    def post_atten_inject(self, block, engine_data, origin_values, layer):
        return origin_values                     # dead after the early return in the reference (corresponder.py:228)

    def step_finished(self, engine_data, sampling_context):
        timestep = sampling_context.timestep
        if timestep < self.step_finished_stop_inject_timestep:
            return
        x = sampling_context.noise               # (N,4,h,w) fp32, device resident; mutated in place (contract)
        idx = engine_data.id_maps.overlap_index(x.shape[2], x.shape[3])
        idx.step(x, self.step_finished_inject_ratio)

    # NB the reference's OverlapCorresponder defines no ``finished`` (corresponder.py:157-376), so its node returns a
    # do-nothing VAE callback (_nodes/samplers.py:113-125): corr-maps are only baked by DefaultCorresponder.finished.


__all__ = ['Corresponder', 'DefaultCorresponder', 'OverlapCorresponder']
