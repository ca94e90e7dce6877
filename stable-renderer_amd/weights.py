"""Where model weights come from for the graph loaders (CheckpointLoaderSimple / ControlNetLoader / LoraLoaderModelOnly,
comfyUI/nodes.py:554-573, 689-700, 770-783) and the LoRA merge they need.

There is no network and no checkpoint in this image, so names resolve in this order: (1) a provider registered in-process
(``register_checkpoint`` ...: tests and bench register seeded synthetic weights of the exact SD1.5 shapes), (2) a
``.safetensors`` file under ``$SR_MODELS_DIR/{checkpoints,controlnet,loras}/<name>`` (the reference's folder_paths layout),
else FileNotFoundError.  Nothing here touches the GPU; tensors stay on the host until UNet / VAEDecoder pack them."""
import os

import torch

_REG = {"checkpoints": {}, "controlnet": {}, "loras": {}}


def _norm(name):
    return str(name).replace("\\", "/")         # workflows saved on Windows carry backslashes (bake.json: "lcm\\SD1.5\\...")


def register_checkpoint(name, provider):
    """provider() -> dict(unet=state_dict, vae=state_dict, clip=TextEncoder | None, unet_cfg=dict | None, vae_ch=int)"""
    _REG["checkpoints"][_norm(name)] = provider


def register_controlnet(name, provider):
    """provider() -> dict(state_dict=..., cfg=dict | None)"""
    _REG["controlnet"][_norm(name)] = provider


def register_lora(name, provider):
    """provider() -> {lora key: tensor}"""
    _REG["loras"][_norm(name)] = provider


def clear_registry():
    for d in _REG.values():
        d.clear()


def _file(kind, name):
    root = os.environ.get("SR_MODELS_DIR")
    if root:
        p = os.path.join(root, kind, _norm(name))
        if os.path.isfile(p):
            return p
    return None


def _load_file(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd)


def resolve(kind, name):
    """-> what the registered provider returns, or the raw state dict of the file"""
    prov = _REG[kind].get(_norm(name))
    if prov is not None:
        return prov()
    path = _file(kind, name)
    if path is None:
        raise FileNotFoundError(f"{kind[:-1] if kind.endswith('s') else kind} '{name}' is neither registered "
                                f"(stable_renderer_amd.weights.register_*) nor a file under $SR_MODELS_DIR/{kind}/")
    return _load_file(path)


def split_checkpoint(sd):
    """A full SD checkpoint -> (unet, vae, clip) state dicts with the prefixes of comfy/supported_models_base.py stripped
    (``model.diffusion_model.``, ``first_stage_model.``, ``cond_stage_model.``)."""
    out = {"unet": {}, "vae": {}, "clip": {}}
    for k, v in sd.items():
        for pre, dst in (("model.diffusion_model.", "unet"), ("first_stage_model.", "vae"), ("cond_stage_model.", "clip")):
            if k.startswith(pre):
                out[dst][k[len(pre):]] = v
                break
    return out["unet"], out["vae"], out["clip"]


# ---- LoRA ---------------------------------------------------------------------------------------------------------------
_RESNET = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.0": "norm2",
           "out_layers.3": "conv2", "skip_connection": "conv_shortcut"}
_BASIC = {"input_blocks.0.0": "conv_in", "out.0": "conv_norm_out", "out.2": "conv_out", "time_embed.0": "time_embedding.linear_1",
          "time_embed.2": "time_embedding.linear_2"}


def _diffusers_module_names(cfg):
    """{diffusers module prefix -> ldm module prefix} for the UNet layout of ``cfg`` (the block arithmetic of
    comfy/utils.py:203-267 unet_to_diffusers, restated per block instead of per tensor)."""
    nrb, depth = list(cfg["num_res_blocks"]), list(cfg["transformer_depth"])
    depth_out = list(cfg["transformer_depth_output"])
    m = {}
    for x in range(len(cfg["channel_mult"])):
        n = 1 + (nrb[x] + 1) * x
        for i in range(nrb[x]):
            m[f"down_blocks.{x}.resnets.{i}"] = f"input_blocks.{n}.0"
            if depth.pop(0) > 0:
                m[f"down_blocks.{x}.attentions.{i}"] = f"input_blocks.{n}.1"
            n += 1
        m[f"down_blocks.{x}.downsamplers.0.conv"] = f"input_blocks.{n}.0.op"
    m["mid_block.attentions.0"] = "middle_block.1"
    m["mid_block.resnets.0"], m["mid_block.resnets.1"] = "middle_block.0", "middle_block.2"
    rr = list(reversed(nrb))
    for x in range(len(cfg["channel_mult"])):
        n = (rr[x] + 1) * x
        for i in range(rr[x] + 1):
            m[f"up_blocks.{x}.resnets.{i}"] = f"output_blocks.{n}.0"
            c = 1
            if depth_out.pop() > 0:
                m[f"up_blocks.{x}.attentions.{i}"] = f"output_blocks.{n}.1"
                c = 2
            if i == rr[x]:
                m[f"up_blocks.{x}.upsamplers.0.conv"] = f"output_blocks.{n}.{c}.conv"
            n += 1
    return m


def unet_lora_key_map(cfg, weight_names):
    """-> {lora module key: UNet weight name} covering the two spellings comfy accepts (comfy/lora.py:212-234):
    ``lora_unet_<ldm module path with _>`` and ``lora_unet_<diffusers module path with _>``."""
    names = [n for n in weight_names if n.endswith(".weight")]
    have = set(names)
    km = {}
    for n in names:
        km["lora_unet_" + n[:-7].replace(".", "_")] = n
    mods = _diffusers_module_names(cfg)
    inv = {v: k for k, v in mods.items()}
    for n in names:
        base = n[:-7]
        dname = None
        if base in _BASIC:
            dname = _BASIC[base]
        else:
            # longest ldm module prefix that is a mapped block
            parts = base.split(".")
            for cut in range(len(parts), 0, -1):
                pre = ".".join(parts[:cut])
                if pre in inv:
                    rest = ".".join(parts[cut:])
                    if inv[pre].split(".")[-2] == "resnets" or inv[pre].startswith("mid_block.resnets"):
                        if rest not in _RESNET:
                            break
                        rest = _RESNET[rest]
                    dname = inv[pre] + ("." + rest if rest else "")
                    break
        if dname is not None and n in have:
            km["lora_unet_" + dname.replace(".", "_")] = n
    return km


def apply_lora(state_dict, lora, strength, key_map):
    """-> new state dict with W += strength * alpha/rank * (up @ down) for every mapped LoRA pair (comfy/lora.py:13-48
    load_lora "regular" spelling + comfy/model_patcher.py calculate_weight "lora" branch, fp32 accumulation, result cast back).
    LoRA keys that map to nothing are returned so the caller can report them."""
    out = dict(state_dict)
    used = set()
    for mod, wname in key_map.items():
        up, down = lora.get(mod + ".lora_up.weight"), lora.get(mod + ".lora_down.weight")
        if up is None or down is None or wname not in state_dict:
            continue
        used.update((mod + ".lora_up.weight", mod + ".lora_down.weight"))
        alpha = lora.get(mod + ".alpha")
        scale = 1.0
        if alpha is not None:
            used.add(mod + ".alpha")
            scale = float(alpha) / down.shape[0]
        w = state_dict[wname]
        if mod + ".lora_mid.weight" in lora:
            raise NotImplementedError("LoCon mid (tucker) LoRA weights")
        delta = torch.mm(up.flatten(start_dim=1).float(), down.flatten(start_dim=1).float()).reshape(w.shape)
        out[wname] = (w.float() + (strength * scale) * delta).to(w.dtype)
    unused = [k for k in lora if k not in used]
    return out, unused
