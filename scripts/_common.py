"""Shared by the example scripts of this directory: put the reference's import paths on sys.path (stable-renderer_amd/compat) and
register synthetic checkpoints under the names the shipped workflow graphs load (no real checkpoint can travel: seeded
random weights of the real shapes, or a 1/5-width model with ``tiny``)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import stable_renderer_amd.compat as compat  # noqa: E402

compat.install()
GOLDEN = os.path.join(ROOT, "tests", "golden")


def register_synthetic_models(tiny=False, dtype="fp16"):
    from stable_renderer_amd import synth, weights as WT
    from stable_renderer_amd.graph_nodes import SyntheticCLIP
    from stable_renderer_amd.model_shapes import (controlnet_names_shapes, unet_names_shapes, vae_decoder_names_shapes,
                                                  vae_encoder_names_shapes)
    from stable_renderer_amd.unet import SD15_CFG
    os.environ.setdefault("SR_DTYPE", dtype)
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64) if tiny else dict(SD15_CFG)
    ns, norms = unet_names_shapes(cfg)
    vns, vnorms = vae_decoder_names_shapes(ch=32 if tiny else 128)
    ens, enorms = vae_encoder_names_shapes(ch=32 if tiny else 128)
    cns, cnorms = controlnet_names_shapes(cfg)
    WT.clear_registry()
    WT.register_checkpoint("dreamshaper_8.safetensors", lambda: dict(
        unet=synth.synth_state_dict(ns, seed=1, norm_names=norms), vae=synth.synth_state_dict(vns, seed=3, norm_names=vnorms),
        vae_encoder=synth.synth_state_dict(ens, seed=4, norm_names=enorms), clip=SyntheticCLIP(ctx_dim=cfg["context_dim"]), unet_cfg=cfg))
    for i, name in enumerate(("control_v11f1p_sd15_depth_fp16.safetensors", "control_v11p_sd15_normalbae_fp16.safetensors")):
        WT.register_controlnet(name, lambda i=i: dict(state_dict=synth.synth_state_dict(cns, seed=20 + i, norm_names=cnorms), cfg=cfg))
    WT.register_lora("lcm/SD1.5/pytorch_lora_weights.safetensors", lambda: {})
    return cfg


def boat_mesh_path():
    """the reference's boat.obj when a resources directory is configured ($SR_RESOURCES_DIR), else the boat-shaped test mesh"""
    res = os.environ.get("SR_RESOURCES_DIR")
    p = os.path.join(res, "example-3d-models", "boat", "boat.obj") if res else None
    return p if p and os.path.exists(p) else os.path.join(GOLDEN, "boatlike.obj")
