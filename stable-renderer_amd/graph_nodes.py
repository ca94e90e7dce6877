"""The ComfyUI built-in nodes the shipped workflows use (resources/example-workflows/*.json), with the reference's class names,
FUNCTION / RETURN_TYPES / INPUT_TYPES surface (comfyUI/nodes.py), executing on the HIP path, plus the registration of the
stable-rendering nodes of nodes.py under the names the graphs use.  Imported (once) by workflow._ensure_default_nodes()."""
import os
import zlib

import torch

from . import nodes as N
from . import weights as WT
from .workflow import Lazy, register_node


def _dtype():
    return {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[os.environ.get("SR_DTYPE", "fp16")]


# ---- text conditioning -------------------------------------------------------------------------------------------------
class SyntheticCLIP:
    """Stand-in text encoder with the CLIP object's surface (``tokenize`` / ``encode_from_tokens``, comfy/sd.py:95-140): a
    deterministic unit-variance (1, 77, ctx_dim) embedding seeded by the text.  Text encoding runs once per prompt and is cached
    across frames by the executor — it is off the hot path, and no text-encoder weights exist offline; a
    checkpoint provider may hand over any object with this surface instead (e.g. a transformers CLIPTextModel wrapper)."""

    def __init__(self, ctx_dim=768, n_ctx=77, seed=0, pooled_dim=None):
        self.ctx_dim, self.n_ctx, self.seed, self.pooled_dim = ctx_dim, n_ctx, seed, pooled_dim

    def tokenize(self, text):
        return str(text)

    def encode_from_tokens(self, tokens, return_pooled=False):
        g = torch.Generator().manual_seed((zlib.crc32(tokens.encode("utf-8")) + 7919 * self.seed) & 0x7FFFFFFF)
        cond = torch.randn(1, self.n_ctx, self.ctx_dim, generator=g)
        pooled = cond[:, -1].clone() if self.pooled_dim is None else torch.randn(1, self.pooled_dim, generator=g)
        return (cond, pooled) if return_pooled else cond


class _NoCLIP:
    def tokenize(self, text):
        raise RuntimeError("this checkpoint came without a text encoder: register a checkpoint provider with `clip=` "
                           "(stable_renderer_amd.weights.register_checkpoint) or feed CONDITIONING tensors directly")

    encode_from_tokens = tokenize


def _text_encode(clip, text, weight=1.0):
    cond, pooled = clip.encode_from_tokens(clip.tokenize(text), return_pooled=True)
    d = {"pooled_output": pooled}
    if weight != 1.0:
        d["strength"] = weight                    # conditions.py:11-12 _set_cond_strength
    return [cond, d]


def _mask_text_encode(clip, text="", mask=None, inverse_mask=False, strength=1.0, mode="default"):
    """_nodes/conditions.py:23-50: the text conditioning with a 'mask' / 'mask_strength' / 'set_area_to_bounds' record (what
    ConditioningSetMask adds, comfyUI/nodes.py:236-262); the sampler resizes the mask to the latent and weights this entry's
    prediction with it (comfy/samplers.py:76-91)"""
    if mask is None:
        return _text_encode(clip, text)
    cond, pooled = clip.encode_from_tokens(clip.tokenize(text), return_pooled=True)
    if inverse_mask:
        mask = 1 - mask
    if mask.dim() < 3:
        mask = mask.unsqueeze(0)
    return [cond, {"pooled_output": pooled, "mask": mask, "set_area_to_bounds": mode == "set_cond_area", "mask_strength": strength}]


class MaskedTextEncode(N.StableRenderingNode):
    """_nodes/conditions.py:52-76: CLIPTextEncode + ConditioningSetMask in one node"""
    Category = "conditioning"

    def __call__(self, clip, text: str = "", mask=None, inverse_mask: bool = False, strength: float = 1.0, mode='default'):
        return [_mask_text_encode(clip, text, mask, inverse_mask, strength, mode)]


class CLIPTextEncode:
    """comfyUI/nodes.py:53-65"""
    RETURN_TYPES = ("CONDITIONING",)
    FUNCTION = "encode"
    CATEGORY = "conditioning"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"text": ("STRING", {"multiline": True}), "clip": ("CLIP",)}}

    def encode(self, clip, text):
        return ([_text_encode(clip, text)],)


class SceneTextEncode(N.StableRenderingNode):
    """_nodes/conditions.py:78-160.  ``merge=True`` (what every shipped workflow uses): one positive and one negative prompt
    concatenated from the sprites' and the environment's prompts.  ``merge=False``: one conditioning per prompt; with an IDMap
    every sprite's prompts act only where the id map shows that sprite (masked conditioning, composed by the sampler)."""
    Category = "conditioning"
    N_OUTPUTS = 2

    def __call__(self, clip, sprite_infos, env_prompts=None, merge: bool = True, idmap=None):
        sprites = list(sprite_infos.values()) if hasattr(sprite_infos, "values") else list(sprite_infos or [])
        envs = list(env_prompts or [])
        conds, neg_conds = [], []
        if merge or merge is None:
            pos, neg = "", ""
            for s in sprites:
                if getattr(s, "prompt", None) and getattr(s, "prompt_weight", 1.0) != 0:
                    pos += s.prompt + ", "
                if getattr(s, "neg_prompt", None) and getattr(s, "neg_prompt_weight", 1.0) != 0:
                    neg += s.neg_prompt + ", "
            for e in envs:
                if getattr(e, "prompt", None) and getattr(e, "weight", 1.0) != 0:
                    pos += e.prompt + ", "
                if getattr(e, "negative_prompt", None) and getattr(e, "negative_weight", 1.0) != 0:
                    neg += e.negative_prompt + ", "
            conds.append(_text_encode(clip, pos))
            neg_conds.append(_text_encode(clip, neg))
            return conds, neg_conds
        for s in sprites:
            for text, w in ((getattr(s, "prompt", None), getattr(s, "prompt_weight", 1.0)),
                            (getattr(s, "neg_prompt", None), getattr(s, "neg_prompt_weight", 1.0))):
                if not text or w == 0:
                    continue
                if idmap is None:
                    # (the reference extends `conds` with the two halves of the pair here, conditions.py:124, and returns None
                    #  for a weight != 1, :19-20: its sampler then fails; this keeps the evident intent -- one entry per prompt)
                    conds.append(_text_encode(clip, text, w))
                else:
                    # per-sprite area: mask = pixels whose id map carries this sprite (first frame only: "not for baking",
                    # conditions.py:80-91), prompt weight as the mask strength; BOTH prompts go to `conds` as in the reference
                    t = idmap.tensor
                    t = t[0] if t.dim() == 4 else t
                    mask = (t[..., 0] == int(s.spriteID)).to(torch.float32).unsqueeze(0)
                    conds.append(_mask_text_encode(clip, text, mask, strength=w))
        for e in envs:
            if getattr(e, "prompt", None) and getattr(e, "weight", 1.0) != 0:
                conds.append(_text_encode(clip, e.prompt, e.weight))
            if getattr(e, "negative_prompt", None) and getattr(e, "negative_weight", 1.0) != 0:
                neg_conds.append(_text_encode(clip, e.negative_prompt, e.negative_weight))
        if not neg_conds:
            neg_conds.append(_text_encode(clip, ""))
        return conds, neg_conds


# ---- loaders -------------------------------------------------------------------------------------------------------------
class CheckpointLoaderSimple:
    """comfyUI/nodes.py:554-573 -> (MODEL, CLIP, VAE)"""
    RETURN_TYPES = ("MODEL", "CLIP", "VAE")
    FUNCTION = "load_checkpoint"
    CATEGORY = "loaders"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"ckpt_name": ("STRING", {})}}

    def load_checkpoint(self, ckpt_name, output_vae=True, output_clip=True):
        from .unet import UNet, SD15_CFG
        from .vae import VAEDecoder
        r = WT.resolve("checkpoints", ckpt_name)
        dt = _dtype()
        if "unet" in r and isinstance(r["unet"], dict):          # a registered provider
            cfg = dict(r.get("unet_cfg") or SD15_CFG)
            unet_sd, vae_sd, clip = r["unet"], r.get("vae"), r.get("clip")
            vae = VAEDecoder(vae_sd, dtype=dt) if (output_vae and vae_sd is not None) else None
            if vae is not None and r.get("vae_encoder") is not None:
                from .vae import VAEEncoder
                vae = VAE(vae, VAEEncoder(r["vae_encoder"], dtype=dt))
        else:                                                    # a full SD1.x checkpoint file
            unet_sd, vae_sd, _ = WT.split_checkpoint(r)
            from .unet import SDXL_CFG
            cfg = dict(SDXL_CFG if "label_emb.0.0.weight" in unet_sd else SD15_CFG)   # (model_detection.py: SDXL carries label_emb)
            clip = None
            vae = VAEDecoder(vae_sd, dtype=dt, prefix="decoder.") if output_vae else None
            if vae is not None and "encoder.conv_in.weight" in vae_sd:
                from .vae import VAEEncoder
                vae = VAE(vae, VAEEncoder(vae_sd, dtype=dt, prefix="encoder."))
        if clip is None:
            clip = _NoCLIP()
        model = N.MODEL(UNet(unet_sd, cfg, dtype=dt), state_dict=unet_sd, cfg=cfg, dtype=dt)
        return (model, clip if output_clip else None, vae)


class LoraLoaderModelOnly:
    """comfyUI/nodes.py:689-700: LoRA merged into the UNet weights (model_patcher.calculate_weight) before they are packed"""
    RETURN_TYPES = ("MODEL",)
    FUNCTION = "load_lora_model_only"
    CATEGORY = "loaders"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"model": ("MODEL",), "lora_name": ("STRING", {}),
                             "strength_model": ("FLOAT", {"default": 1.0, "min": -20.0, "max": 20.0, "step": 0.01})}}

    def load_lora_model_only(self, model, lora_name, strength_model):
        from .unet import UNet
        if strength_model == 0:
            return (model,)
        if model.state_dict is None:
            raise ValueError("LoraLoaderModelOnly needs a MODEL that kept its host state dict")
        lora = WT.resolve("loras", lora_name)
        km = WT.unet_lora_key_map(model.cfg, model.state_dict.keys())
        sd, unused = WT.apply_lora(model.state_dict, lora, float(strength_model), km)
        m = N.MODEL(UNet(sd, model.cfg, dtype=model.dtype), state_dict=sd, cfg=model.cfg, dtype=model.dtype)
        m.lora_unused_keys = unused
        return (m,)


class ControlNetLoader:
    """comfyUI/nodes.py:770-783"""
    RETURN_TYPES = ("CONTROL_NET",)
    FUNCTION = "load_controlnet"
    CATEGORY = "loaders"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"control_net_name": ("STRING", {})}}

    def load_controlnet(self, control_net_name):
        from .controlnet import ControlNet
        r = WT.resolve("controlnet", control_net_name)
        if "state_dict" in r and isinstance(r["state_dict"], dict):
            return (ControlNet(r["state_dict"], r.get("cfg"), dtype=_dtype()),)
        sd = {(k[len("control_model."):] if k.startswith("control_model.") else k): v for k, v in r.items()}
        return (ControlNet(sd, None, dtype=_dtype()),)


class AppliedControl:
    """What ControlNetApply leaves in the conditioning: ``control_net.copy().set_cond_hint(hint, strength)`` chained through
    ``previous_controlnet`` (comfy/controlnet.py:37-93)"""

    def __init__(self, net, hint, strength, previous=None, timestep_percent_range=(0.0, 1.0)):
        self.net, self.hint, self.strength, self.previous = net, hint, float(strength), previous
        self.timestep_percent_range = (float(timestep_percent_range[0]), float(timestep_percent_range[1]))

    def chain(self):
        c, out = self, []
        while c is not None:
            out.append(c)
            c = c.previous
        return out


class ControlNetApply:
    """comfyUI/nodes.py:806-848"""
    RETURN_TYPES = ("CONDITIONING",)
    FUNCTION = "apply_controlnet"
    CATEGORY = "conditioning"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"conditioning": ("CONDITIONING",), "control_net": ("CONTROL_NET",), "image": ("IMAGE",),
                             "strength": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 10.0, "step": 0.01})}}

    def apply_controlnet(self, conditioning, control_net, image, strength):
        if strength == 0:
            return (conditioning,)
        if image.dim() == 2:
            image = torch.stack([image, image, image], dim=0).unsqueeze(0)
        elif image.dim() == 3:
            image = image.unsqueeze(0)
        hint = image.movedim(-1, 1) if (image.shape[-1] in (3, 4) and image.shape[1] >= 256) else image
        out = []
        for t in conditioning:
            d = dict(t[1])
            d["control"] = AppliedControl(control_net, hint, strength, d.get("control"))
            d["control_apply_to_uncond"] = True
            out.append([t[0], d])
        return (out,)


class ControlNetApplyAdvanced:
    """comfyUI/nodes.py:850-896: the net goes on BOTH conditionings, with a start / end percent of the schedule outside which it is
    not run (set_cond_hint(hint, strength, (start_percent, end_percent)), comfy/controlnet.py:53-62, 184-189)"""
    RETURN_TYPES = ("CONDITIONING", "CONDITIONING")
    RETURN_NAMES = ("positive", "negative")
    FUNCTION = "apply_controlnet"
    CATEGORY = "conditioning"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"positive": ("CONDITIONING",), "negative": ("CONDITIONING",), "control_net": ("CONTROL_NET",),
                             "image": ("IMAGE",), "strength": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 10.0, "step": 0.01}),
                             "start_percent": ("FLOAT", {"default": 0.0, "min": 0.0, "max": 1.0, "step": 0.001}),
                             "end_percent": ("FLOAT", {"default": 1.0, "min": 0.0, "max": 1.0, "step": 0.001})}}

    def apply_controlnet(self, positive, negative, control_net, image, strength, start_percent, end_percent):
        if strength == 0:
            return (positive, negative)
        hint = image.movedim(-1, 1)
        made, out = {}, []
        for conditioning in (positive, negative):
            c = []
            for t in conditioning:
                d = dict(t[1])
                prev = d.get("control")
                if id(prev) not in made:
                    made[id(prev)] = AppliedControl(control_net, hint, strength, prev, (start_percent, end_percent))
                d["control"] = made[id(prev)]
                d["control_apply_to_uncond"] = False
                c.append([t[0], d])
            out.append(c)
        return (out[0], out[1])


# ---- sampling / decode ---------------------------------------------------------------------------------------------------
class KSampler:
    """comfyUI/nodes.py:1497-1520 -> common_ksampler (custom_ksampler with noise_option='random')"""
    RETURN_TYPES = ("LATENT",)
    FUNCTION = "sample"
    CATEGORY = "sampling"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"model": ("MODEL",), "seed": ("INT", {"default": 0}), "steps": ("INT", {"default": 20}),
                             "cfg": ("FLOAT", {"default": 8.0}), "sampler_name": ("STRING", {}), "scheduler": ("STRING", {}),
                             "positive": ("CONDITIONING",), "negative": ("CONDITIONING",), "latent_image": ("LATENT",),
                             "denoise": ("FLOAT", {"default": 1.0})}}

    def sample(self, model, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=1.0):
        return N.custom_ksampler(model, seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, denoise=denoise)


class VAE:
    """What CheckpointLoaderSimple hands out as VAE: the decoder plan object, plus the encoder when the checkpoint has one
    (comfy/sd.py:196-371 VAE.decode / VAE.encode).  Attribute access falls through to the decoder, so code written against the
    decoder alone keeps working."""

    def __init__(self, decoder, encoder=None):
        self.decoder, self.encoder = decoder, encoder
        self._enc_plans = {}

    def __getattr__(self, name):
        return getattr(self.__dict__["decoder"], name)

    def encode(self, pixels):
        """pixels (N,H,W,C) in [0,1] -> (N,4,H/8,W/8) fp32; crops to a multiple of 8 as vae_encode_crop_pixels (sd.py:292-299)"""
        if self.encoder is None:
            raise ValueError("this VAE came without encoder weights (encoder.* / quant_conv keys)")
        x = (pixels.shape[1] // 8) * 8
        y = (pixels.shape[2] // 8) * 8
        if pixels.shape[1] != x or pixels.shape[2] != y:
            xo, yo = (pixels.shape[1] % 8) // 2, (pixels.shape[2] % 8) // 2
            pixels = pixels[:, xo:x + xo, yo:y + yo, :]
        key = (pixels.shape[0], pixels.shape[1], pixels.shape[2])
        if key not in self._enc_plans:
            self._enc_plans[key] = self.encoder.build(*key)
        return self.encoder.encode(self._enc_plans[key], pixels.to(self.encoder.device))


class VAEEncode:
    """comfyUI/nodes.py:319-332"""
    RETURN_TYPES = ("LATENT",)
    FUNCTION = "encode"
    CATEGORY = "latent"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"pixels": ("IMAGE",), "vae": ("VAE",)}}

    def encode(self, vae, pixels):
        if pixels.dim() == 3:
            pixels = pixels.unsqueeze(0)
        return (N.LATENT(samples=vae.encode(pixels[:, :, :, :3])),)


class LoadImage:
    """comfyUI/nodes.py:1623-1665 -> (IMAGE (1,H,W,3) in [0,1], MASK = 1 - alpha or 64x64 zeros).  ``image`` is a path, or a name
    under $SR_INPUT_DIR (the reference's input directory)"""
    RETURN_TYPES = ("IMAGE", "MASK")
    FUNCTION = "load_image"
    CATEGORY = "image"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"image": ("STRING", {})}}

    def load_image(self, image):
        import numpy as np
        from PIL import Image, ImageOps, ImageSequence
        path = image if os.path.isabs(image) or os.path.exists(image) else os.path.join(os.environ.get("SR_INPUT_DIR", "input"), image)
        img = Image.open(path)
        images, masks = [], []
        for i in ImageSequence.Iterator(img):
            i = ImageOps.exif_transpose(i)
            if i.mode == "I":
                i = i.point(lambda v: v * (1 / 255))
            images.append(torch.from_numpy(np.array(i.convert("RGB")).astype(np.float32) / 255.0)[None,])
            if "A" in i.getbands():
                masks.append((1.0 - torch.from_numpy(np.array(i.getchannel("A")).astype(np.float32) / 255.0)).unsqueeze(0))
            else:
                masks.append(torch.zeros((64, 64), dtype=torch.float32).unsqueeze(0))
        if len(images) > 1:
            return (torch.cat(images, dim=0), torch.cat(masks, dim=0))
        return (images[0], masks[0])


class VAEDecode(N.VAEDecode):
    """comfyUI/nodes.py:287-303"""
    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "decode"
    CATEGORY = "latent"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"samples": ("LATENT",), "vae": ("VAE",)}, "optional": {"callback": ("VAEDECODECALLBACK",)}}


# ---- logic (_nodes/logic.py) -----------------------------------------------------------------------------------------------
class IsNotNone(N.StableRenderingNode):
    Category = "Logic"

    def __call__(self, value, mode='strict') -> bool:
        if mode == 'strict':
            return value is not None
        try:
            return bool(value)
        except Exception:
            return value is not None


class If(N.StableRenderingNode):
    Category = "Logic"
    LAZY_INPUTS = ("true_value", "false_value")

    def __call__(self, condition: bool, true_value: Lazy, false_value: Lazy):
        return true_value.value if condition else false_value.value


class IfValTypeEqual(N.StableRenderingNode):
    Category = "Logic"

    def __call__(self, val, type_name: str) -> bool:
        return type(val).__name__.upper() == type_name.upper()


for _name, _cls in (("CheckpointLoaderSimple", CheckpointLoaderSimple), ("LoraLoaderModelOnly", LoraLoaderModelOnly),
                    ("ControlNetLoader", ControlNetLoader), ("ControlNetApply", ControlNetApply), ("ControlNetApplyAdvanced", ControlNetApplyAdvanced), ("CLIPTextEncode", CLIPTextEncode),
                    ("SceneTextEncode", SceneTextEncode), ("MaskedTextEncode", MaskedTextEncode), ("KSampler", KSampler), ("VAEDecode", VAEDecode), ("VAEEncode", VAEEncode), ("LoadImage", LoadImage),
                    ("IsNotNone", IsNotNone), ("If", If), ("IfValTypeEqual", IfValTypeEqual),
                    ("EngineData", N.EngineDataNode), ("VirtualEngineData", N.VirtualEngineDataNode),
                    ("InferenceOutput", N.InferenceOutputNode), ("EmptyCorrMaps", N.EmptyCorrMaps),
                    ("DefaultCorresponder", N.DefaultCorresponder), ("OverlapCorresponder", N.OverlapCorresponder),
                    ("CorrespondSampler", N.CorrespondSampler), ("IDSequenceLoader", N.IDSequenceLoader),
                    ("ImageSequenceLoader", N.ImageSequenceLoader), ("NoiseSequenceLoader", N.NoiseSequenceLoader)):
    register_node(_name, _cls)

from . import extra_nodes as _X  # noqa: E402

for _name, _cls in _X.ALL.items():
    register_node(_name, _cls)


def register_legacy_aliases():
    """opt-in: node names of earlier reference revisions that shipped example graphs still use (see nodes.FrameDataNode)"""
    register_node("FrameData", N.FrameDataNode)
