"""Stable-rendering node surface (the drop-in boundary of SURVEY.md §8b): classes with the reference's names and
``__call__`` signatures (comfyUI/stable_rendering/_nodes/{data,samplers}.py, comfyUI/nodes.py VAEDecode/custom_ksampler),
executing on the HIP path.  The node *registration* machinery (AdvancedNodeBase -> ComfyUI web UI, node_base.py:179-686) is
UI plumbing and out of scope; a graph executor only needs these callables."""
import copy
import os
from functools import partial
from typing import Callable, Optional, Tuple

import numpy as np
import torch

from .corrmap import CorrespondMap, IDMap
from .corresponder import DefaultCorresponder as _DefaultCorresponder
from .corresponder import OverlapCorresponder as _OverlapCorresponder
from .sampling import DiffusionRunner
from .types import EngineData, InferenceOutput, LATENT


class StableRenderingNode:
    NAMESPACE = "StableRendering"
    Category = "stable-rendering"


class MODEL:
    """What CheckpointLoaderSimple hands to samplers here: a HIP UNet (+ lazily built runners keyed by batch/shape/ControlNets).
    ``state_dict`` (host tensors) is kept when a later LoraLoaderModelOnly must re-pack patched weights."""

    def __init__(self, unet, state_dict=None, cfg=None, dtype=None):
        self.unet, self.state_dict = unet, state_dict
        self.cfg = cfg if cfg is not None else unet.cfg
        self.dtype = dtype if dtype is not None else unet.dtype
        self._runners = {}

    def runner(self, N, h, w, cfg, use_graph=True, controlnets=()):
        key = (N, h, w, float(cfg), use_graph, tuple((id(c), c.strength) for c in controlnets))
        if key not in self._runners:
            self._runners[key] = DiffusionRunner(self.unet, N, h, w, cfg, use_graph=use_graph, controlnets=list(controlnets))
        return self._runners[key]


def pooled_of(c):
    """the 'pooled_output' a CLIPTextEncode-style node left in the conditioning, or None"""
    if isinstance(c, torch.Tensor) or not c:
        return None
    return c[0][1].get("pooled_output") if isinstance(c[0], (list, tuple)) else None


def unwrap_conditioning(c):
    """CONDITIONING ([[cond, {..., 'control': AppliedControl}], ...], comfyUI/nodes.py:53-65, 806-848) or a bare
    (1|N, 77, ctx) tensor -> (entries for DiffusionRunner.set_cond_entries, [AppliedControl, ...] newest first).  Several
    entries / masks / strengths / areas are composed by the sampler as calc_cond_uncond_batch does (conditioning.py)."""
    from .conditioning import entries_of
    entries = entries_of(c)
    if not entries:
        raise ValueError("empty conditioning")
    ctl = None
    for e in entries:                                  # the batch's control is the last member's (samplers.py:258); the shipped
        if e.get("control") is not None:               # graphs apply the same ControlNet chain to every entry of a list
            ctl = e["control"]
    return entries, (ctl.chain() if ctl is not None else [])


class EngineDataNode(StableRenderingNode):
    """_nodes/data.py:36-63: unpack the hidden EngineData into its 11 outputs."""

    N_OUTPUTS = 11

    def IsChanged(self, engine_data: EngineData):
        return None if engine_data is None else engine_data.serial      # a new EngineData per engine frame (data.py:65-69)

    def __call__(self, engine_data: EngineData):
        if engine_data is None:
            return (None, None, None, None, None, None, None, None, None, {}, "")
        return (engine_data.color_maps, engine_data.id_maps, engine_data.pos_maps, engine_data.normal_maps,
                engine_data.depth_maps, engine_data.canny_maps, engine_data.noise_maps, engine_data.masks,
                engine_data.correspond_maps, engine_data.sprite_infos, engine_data.env_prompts)


class FrameDataNode(EngineDataNode):
    """``FrameData``: the 7-output node an earlier revision of the reference had where ``EngineData`` is today; two shipped example
    graphs (miku-img2img-example-unix.json, miku-controlnet-no-lora-workflow.json) still name it, and the reference itself no
    longer loads them (Workflow.Load -> ValueError "Cannot find the type FrameData").  NOT registered by default -- same error
    here; ``graph_nodes.register_legacy_aliases()`` opts in."""
    N_OUTPUTS = 7

    def __call__(self, engine_data: EngineData):
        if engine_data is None:
            return (None,) * 7
        return (engine_data.color_maps, engine_data.id_maps, engine_data.pos_maps, engine_data.normal_maps,
                engine_data.depth_maps, engine_data.noise_maps, engine_data.masks)


class VirtualEngineDataNode(StableRenderingNode):
    """_nodes/data.py:71-105: build EngineData when running a graph without the engine."""
    PriorNode = True

    def __call__(self, color_maps=None, id_maps=None, pos_maps=None, normal_maps=None, depth_maps=None, canny_maps=None,
                 noise_maps=None, masks=None, correspond_maps=None, sprites=None, env_prompt=None, context=None) -> EngineData:
        ed = self._make(color_maps, id_maps, pos_maps, normal_maps, depth_maps, canny_maps, noise_maps, masks, correspond_maps,
                        sprites, env_prompt)
        if context is not None:
            context.engine_data = ed          # later nodes' hidden EngineData input (data.py:88-105)
        return ed

    @staticmethod
    def _make(color_maps, id_maps, pos_maps, normal_maps, depth_maps, canny_maps, noise_maps, masks, correspond_maps, sprites,
              env_prompt):
        n = len(id_maps) if id_maps is not None else (0 if color_maps is None else len(color_maps))
        return EngineData(frame_indices=list(range(n)), color_maps=color_maps, id_maps=id_maps, pos_maps=pos_maps,
                          normal_maps=normal_maps, depth_maps=depth_maps, canny_maps=canny_maps, noise_maps=noise_maps,
                          masks=masks, correspond_maps=correspond_maps, sprite_infos=sprites, env_prompts=env_prompt)


def _extract_index(name, default):
    """common_utils/path_utils.py extract_index: the last number in the file name"""
    import re
    m = re.findall(r"\d+", os.path.splitext(name)[0])
    return int(m[-1]) if m else default


def _sorted_files(directory, exts):
    names = [f for f in os.listdir(directory) if f.endswith(exts) and os.path.exists(os.path.join(directory, f))]
    return sorted(names, key=lambda n: _extract_index(n, names.index(n)))


def _check_loader_args(directory, frame_start, num_frames, sd_version):
    if not os.path.exists(directory):
        raise FileNotFoundError(f"Directory {directory} not found")
    if frame_start < 0:
        raise ValueError("frame_start takes value larger than or equal to 0, got ", frame_start)
    if num_frames <= 0:
        raise ValueError("num_frames takes value larger 0, got ", frame_start)
    if sd_version not in ["SD15", "SDXL"]:
        raise ValueError("sd_version should be either SD15 or SDXL")


class IDSequenceLoader(StableRenderingNode):
    """_nodes/loaders.py:312-326"""
    Category = "loader"

    def __call__(self, directory, frame_start: int = 0, num_frames: int = 16, device="cuda") -> IDMap:
        return IDMap.from_directory(directory=directory, frame_start=frame_start, num_frames=num_frames,
                                    use_frame_indices_from_filename=False, device=device)


class ImageSequenceLoader(StableRenderingNode):
    """_nodes/loaders.py:19-76: RGB images of a dump directory, nearest-resized to the SD size, (N,H,W,3) in [0,1]"""
    Category = "loader"

    def __call__(self, directory, frame_start: int = 0, num_frames: int = 16, sd_version="SD15", device="cuda"):
        from PIL import Image
        _check_loader_args(directory, frame_start, num_frames, sd_version)
        size = (512, 512) if sd_version == "SD15" else (1024, 1024)
        out = []
        for fn in _sorted_files(directory, (".jpeg", ".png", ".bmp", ".jpg"))[frame_start: frame_start + num_frames]:
            a = torch.from_numpy(np.array(Image.open(os.path.join(directory, fn)).convert("RGB"))).permute(2, 0, 1)[None]
            a = torch.nn.functional.interpolate(a, size=size)                   # nearest on uint8, as the reference
            out.append(a.permute(0, 2, 3, 1) / 255.0)
        return torch.cat(out, 0).to(device) if out else None


class NoiseSequenceLoader(StableRenderingNode):
    """_nodes/loaders.py:79-152: dumped engine noise (N,H,W,4) -> LATENT(noise = AdaIN(strip-pooled noise, full noise)).
    The pooling is the same 64-consecutive-pixel row strip as RenderManager._save_frame_data and runs in ``sr_noise_pool``
    (mask 0); the strip mean is kept in fp32 where the reference rounds it to the dump's fp16 before AdaIN."""
    Category = "loader"

    def __call__(self, directory, frame_start: int = 0, num_frames: int = 16, sd_version="SD15", device="cuda") -> LATENT:
        from . import ops as O
        _check_loader_args(directory, frame_start, num_frames, sd_version)
        ts = []
        for fn in _sorted_files(directory, (".jpeg", ".png", ".bmp", ".jpg", ".npy"))[frame_start: frame_start + num_frames]:
            path = os.path.join(directory, fn)
            if path.endswith(".npy"):
                t = torch.from_numpy(np.load(path)).squeeze()
                if t.dim() != 3:
                    raise ValueError(f"Invalid shape of noise tensor: {t.shape}.")
                if not (t.shape[-1] == 4 or t.shape[1] == 4):
                    raise ValueError(f"Invalid noise tensor shape: {t.shape}.")
            else:
                from PIL import Image
                t = torch.from_numpy(np.array(Image.open(path).convert("RGBA"))).permute(2, 0, 1) / 255.0
            ts.append(t)
        if not ts:
            return None
        if any(t.shape != ts[0].shape for t in ts):
            raise ValueError("Tensor data has inconsistent shapes.")
        noise = torch.stack(ts, 0)
        _, height, width, channel = noise.shape
        assert channel == 4, "Noise shape should be in BHW4"
        unit = 64 if sd_version == "SD15" else 128
        if height % unit != 0 or width % unit != 0:
            raise ValueError(f"Noise shape for {sd_version} should be divisible by {unit}")
        # reshape_magnitude = height // 64 (SD15) / height // 128 (SDXL): view(-1, m, m, 4).mean((1, 2)) averages m*m CONSECUTIVE
        # pixels and the result is viewed (height / m, width / m) -- a 256^2 SD1.5 dump becomes a 64 x 64 latent (loaders.py:131-146)
        m = height // unit
        if (height * width) % (m * m) or width % m:
            raise RuntimeError(f"shape '[-1, {m}, {m}, 4]' is invalid for input of size {noise.numel()}")      # what torch's view raises
        nz = noise.to(device=device, dtype=torch.float16).contiguous()
        zeros = torch.zeros(1, height, width, dtype=torch.float16, device=device)          # mask 0: the noise itself
        bg = torch.zeros(1, height, width, 4, dtype=torch.float32, device=device)
        outs = [O.noise_pool(nz[i:i + 1], 1.0 - zeros, bg, magnitude=m)[1] for i in range(nz.shape[0])]
        lat = torch.cat(outs, 0)
        return LATENT(samples=torch.zeros_like(lat), noise=lat)


class EmptyCorrMaps(StableRenderingNode):
    """_nodes/data.py:10-25"""

    def __call__(self, k: int = 3, width: int = 512, height: int = 512, create_count: int = 1):
        return {(i + 1, i + 1): CorrespondMap(k=k, width=width, height=height) for i in range(create_count)}


class InferenceOutputNode(StableRenderingNode):
    """_nodes/data.py:107-139"""
    IsOutputNode = True
    Unique = True

    @classmethod
    def INPUT_TYPES(cls):
        # `save` belongs to __server_call__ (data.py:126-139); graphs saved from the web UI carry it
        return {"required": {"colorImg": ("IMAGE", {})}, "optional": {"save": ("BOOLEAN", {"default": False})},
                "hidden": {"context": "INFERENCE_CONTEXT"}}

    def __call__(self, colorImg, context=None) -> InferenceOutput:
        out = InferenceOutput(colorImg)
        if context is not None:
            context.final_output = out
        return out


class DefaultCorresponder(StableRenderingNode):
    """_nodes/samplers.py:20-68 -> (Corresponder, VAEDecodeCallback)"""
    Category = "sampling"
    N_OUTPUTS = 2

    def __call__(self, engine_data: EngineData, update_corrmap: bool = True, update_mode='first_avg',
                 post_attn_inject_ratio: float = 0.6) -> Tuple[object, Callable]:
        c = _DefaultCorresponder(update_corrmap=update_corrmap, update_corrmap_mode=update_mode,
                                 post_attn_inject_ratio=post_attn_inject_ratio)
        return c, partial(c.finished, engine_data)


class OverlapCorresponder(StableRenderingNode):
    """_nodes/samplers.py:71-125; the reference's OverlapCorresponder has no ``finished`` -> do-nothing VAE callback"""
    Category = "sampling"
    N_OUTPUTS = 2

    def __call__(self, engine_data: EngineData, update_corrmap: bool = True, update_mode='first_avg',
                 pre_attn_inject_num_of_random_frames: int = 1, post_attn_inject_ratio: float = 0.6,
                 step_finished_inject_ratio: float = 0.5, step_finished_stop_inject_timestep: int = 500):
        c = _OverlapCorresponder(update_corrmap=update_corrmap, update_corrmap_mode=update_mode,
                                 pre_attn_inject_num_random_frames=pre_attn_inject_num_of_random_frames,
                                 post_attn_inject_ratio=post_attn_inject_ratio,
                                 step_finished_inject_ratio=step_finished_inject_ratio,
                                 step_finished_stop_inject_timestep=step_finished_stop_inject_timestep)
        if hasattr(c, "finished"):
            return c, partial(c.finished, engine_data)
        return c, (lambda *a, **k: None)


def custom_ksampler(model: MODEL, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent, denoise=1.0,
                    noise_option='random', callbacks=None, engine_data=None, corresponder=None, **kwargs):
    """comfyUI/nodes.py:1438-1495.  positive / negative: CONDITIONING lists or bare (1|N, 77, ctx) embeddings; ControlNets
    applied to the positive conditioning serve both halves of the batch (control_apply_to_uncond, samplers.py:520-548).
    ``engine_data`` / ``corresponder`` reach the attention blocks through the plan (K/V injection)."""
    latent_image = latent["samples"]
    pooled = (pooled_of(positive), pooled_of(negative))
    positive, controls = unwrap_conditioning(positive)
    negative, _ = unwrap_conditioning(negative)
    N, _, h, w = latent_image.shape
    if noise_option == 'disable':
        noise = torch.zeros_like(latent_image)
    elif noise_option == 'incoming' and "noise" in latent:
        noise = latent["noise"]
    elif noise_option in ('random', 'incoming'):
        if seed is None:
            seed = int(torch.randint(0, 2 ** 32, (1,)).item())
        g = torch.manual_seed(seed)                                   # comfy/sample.py:19-28 prepare_noise
        noise = torch.randn(latent_image.size(), dtype=latent_image.dtype, generator=g, device="cpu")
    else:
        raise ValueError(f"Invalid noise option: {noise_option}")
    nets = []
    for c in controls:
        net = c.net
        rng = getattr(c, "timestep_percent_range", (0.0, 1.0))
        if net.strength != c.strength or tuple(getattr(net, "timestep_percent_range", (0.0, 1.0))) != tuple(rng):
            # same packed weights, its own strength / schedule window (set_cond_hint(hint, strength, timestep_percent_range))
            cache = net.__dict__.setdefault("_by_strength", {})
            key = (c.strength, tuple(rng))
            if key not in cache:
                cache[key] = copy.copy(net)
                cache[key].strength = c.strength
                cache[key].timestep_percent_range = tuple(rng)
            net = cache[key]
        nets.append(net)
    run = model.runner(N, h, w, cfg, controlnets=nets)
    run.set_cond_entries(positive, negative)
    adm = model.cfg.get("adm_in_channels")
    if adm:                                            # SDXL family: y from the pooled text embedding + size embeddings
        from .sampling import encode_adm_sdxl
        if pooled[0] is None:
            raise ValueError("this model takes vector conditioning: the positive conditioning carries no 'pooled_output'")
        ys = [encode_adm_sdxl(pl if pl is not None else pooled[0], width=w * 8, height=h * 8) for pl in pooled]
        if ys[0].shape[1] != adm:
            raise ValueError(f"pooled_output width {ys[0].shape[1] - 1536} does not match adm_in_channels {adm} - 1536")
        run.set_vector_conditioning(ys[0], ys[1])
    if controls:
        run.set_control_hints([c.hint for c in controls])
    n_rand = None
    if corresponder is not None and engine_data is not None and isinstance(corresponder, _OverlapCorresponder):
        n_rand = corresponder.pre_attn_inject_num_random_frames
    cb = None
    if callbacks:
        def cb(ctx):
            for c in callbacks:
                c(ctx)
    samples, inj = run.sample(noise, steps, sampler_name, scheduler, denoise=denoise, latent_image=latent_image, seed=seed,
                              inject_n_rand=n_rand, step_callback=cb)
    if inj is not None and corresponder is not None:
        corresponder._random_frame_indices = torch.tensor(inj)
    out = LATENT(latent)
    out["samples"] = samples
    return (out,)


class CorrespondSampler(StableRenderingNode):
    """_nodes/samplers.py:128-201"""
    Category = "sampling"

    def __call__(self, model: MODEL, positive, negative, corresponder, engine_data: EngineData, latent: Optional[LATENT] = None,
                 steps: int = 20, cfg: float = 8.0, sampler_name="euler", scheduler="normal", denoise: float = 1.0) -> LATENT:
        if isinstance(corresponder, _OverlapCorresponder) and sampler_name not in ['ddim', 'ddpm']:
            raise ValueError("OverlapCorresponder only works with ddim or ddpm sampler_name.")
        if hasattr(corresponder, 'prepare'):
            corresponder.prepare(engine_data)
        callbacks = []
        if hasattr(corresponder, "step_finished"):
            callbacks = [partial(corresponder.step_finished, engine_data)]
        if latent is None:
            if engine_data is None:
                raise ValueError("Input latent is None and engine_data is also None.")
            latent = engine_data.noise_maps
        return custom_ksampler(model=model, seed=None, steps=steps, cfg=cfg, sampler_name=sampler_name, scheduler=scheduler,
                               positive=positive, negative=negative, latent=latent, denoise=denoise, noise_option='incoming',
                               engine_data=engine_data, corresponder=corresponder, callbacks=callbacks)[0]


class VAEDecode:
    """comfyUI/nodes.py:287-303 (with the reference's ``callback`` hook)"""

    def __init__(self):
        self._plans = {}

    def decode(self, vae, samples, callback=None):
        z = samples["samples"]
        key = tuple(z.shape)
        if key not in self._plans:
            self._plans[key] = vae.build(z.shape[0], z.shape[2], z.shape[3])
        p = self._plans[key]
        p["z"].copy_(z)
        p["plan"].run()
        images = p["img"]
        if callback is not None:
            callback(images)
        return (images,)
