"""The shipped workflow graphs (tests/golden/workflows/*.json = resources/example-workflows of the reference) executed by
workflow.PromptExecutor on the HIP path: graph result == the same nodes called by hand; loaders are cached across frames while
everything fed by the frame's EngineData re-runs."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
WF = os.path.join(os.path.dirname(__file__), "golden", "workflows")
H = W_ = 256                                                    # ControlNetApply only moves channels for images >= 256 px


def _register(monkeypatch):
    from stable_renderer_amd import synth, weights as WT
    from stable_renderer_amd.graph_nodes import SyntheticCLIP
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes, controlnet_names_shapes, vae_encoder_names_shapes
    from stable_renderer_amd.unet import SD15_CFG
    monkeypatch.setenv("SR_DTYPE", "fp32")
    monkeypatch.setenv("SR_AUTOTUNE", "0")                     # same tiles in both plan builds -> bit-identical results
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    ns, norms = unet_names_shapes(cfg)
    vns, vnorms = vae_decoder_names_shapes(ch=32)
    cns, cnorms = controlnet_names_shapes(cfg)
    ens, enorms = vae_encoder_names_shapes(ch=32)
    WT.clear_registry()
    WT.register_checkpoint("dreamshaper_8.safetensors", lambda: dict(
        unet=synth.synth_state_dict(ns, seed=1, norm_names=norms), vae=synth.synth_state_dict(vns, seed=3, norm_names=vnorms),
        vae_encoder=synth.synth_state_dict(ens, seed=4, norm_names=enorms), clip=SyntheticCLIP(ctx_dim=64), unet_cfg=cfg))
    for i, name in enumerate(("control_v11f1p_sd15_depth_fp16.safetensors", "control_v11p_sd15_normalbae_fp16.safetensors")):
        WT.register_controlnet(name, lambda i=i: dict(state_dict=synth.synth_state_dict(cns, seed=20 + i, norm_names=cnorms), cfg=cfg))
    g = torch.Generator().manual_seed(9)
    lora = {}
    for mod, (o, i_) in (("lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q", (64, 64)),
                         ("lora_unet_mid_block_attentions_0_proj_in", (256, 256)),
                         ("lora_unet_up_blocks_3_resnets_2_conv1", (64, 128))):
        r = 4
        lora[mod + ".lora_up.weight"] = 0.05 * torch.randn((o, r) + ((1, 1) if mod.endswith("conv1") else ()), generator=g)
        lora[mod + ".lora_down.weight"] = 0.05 * torch.randn((r, i_) + ((3, 3) if mod.endswith("conv1") else ()), generator=g)
        lora[mod + ".alpha"] = torch.tensor(2.0)
    WT.register_lora("lcm/SD1.5/pytorch_lora_weights.safetensors", lambda: dict(lora))
    return cfg


def _engine_data(seed, N=2, k=3):
    from stable_renderer_amd.corrmap import CorrespondMap, IDMap
    from stable_renderer_amd.types import EngineData, EnvPrompt, LATENT, Sprite, SpriteInfos
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(N, H, W_, 4, dtype=torch.int32)
    ids[..., 0], ids[..., 1] = 1, 1
    ids[..., 2] = torch.randint(0, k * k, (N, H, W_), generator=g, dtype=torch.int32)
    ids[..., 3] = torch.randint(0, H * W_, (N, H, W_), generator=g, dtype=torch.int32)          # fully covered frames
    dev = "cuda"
    return EngineData(frame_indices=list(range(N)), id_maps=IDMap(ids.to(dev)),
                      color_maps=torch.rand(N, H, W_, 3, generator=g).to(dev), normal_maps=torch.rand(N, H, W_, 3, generator=g).to(dev),
                      depth_maps=torch.rand(N, H, W_, 1, generator=g).expand(-1, -1, -1, 3).contiguous().to(dev),
                      noise_maps=LATENT(samples=torch.zeros(N, 4, H // 8, W_ // 8).to(dev), noise=torch.randn(N, 4, H // 8, W_ // 8, generator=g).to(dev)),
                      correspond_maps={(1, 1): CorrespondMap(k=k, width=W_, height=H)},
                      sprite_infos=SpriteInfos({1: Sprite(1, prompt="a red ball", neg_prompt="blurry")}),
                      env_prompts=[EnvPrompt(prompt="forest", negative_prompt="watermark")])


def _by_hand(ed, controls):
    """bake.json spelled out as direct node calls"""
    from stable_renderer_amd import graph_nodes as G, nodes as Nn
    model, clip, vae = G.CheckpointLoaderSimple().load_checkpoint("dreamshaper_8.safetensors")
    (model,) = G.LoraLoaderModelOnly().load_lora_model_only(model, "lcm\\SD1.5\\pytorch_lora_weights.safetensors", 1)
    assert model.lora_unused_keys == []
    pos, neg = G.SceneTextEncode()(clip, ed.sprite_infos, ed.env_prompts, True, ed.id_maps)
    for name, img in controls:
        (cn,) = G.ControlNetLoader().load_controlnet(name)
        (pos,) = G.ControlNetApply().apply_controlnet(pos, cn, img, 1)
    corr, cb = Nn.DefaultCorresponder()(ed, True, "first_avg", 0.6)
    lat = Nn.CorrespondSampler()(model, pos, neg, corr, ed, latent=ed.noise_maps, steps=4, cfg=2, sampler_name="euler",
                                 scheduler="sgm_uniform", denoise=1)
    (img,) = G.VAEDecode().decode(vae, lat, cb)
    return img.clone()


@pytest.mark.parametrize("graph", ["bake", "no-control-bake"])
def test_shipped_graph_equals_direct_node_calls(graph, monkeypatch):
    from stable_renderer_amd import workflow as W
    _register(monkeypatch)
    ed = _engine_data(5)
    ex = W.PromptExecutor(dev_mode=True)
    ctx = W.run_workflow(os.path.join(WF, graph + ".json"), engine_data=ed, executor=ex)
    assert ctx.success and ctx.final_output is not None
    img = ctx.final_output.frame_color.clone()
    assert tuple(img.shape) == (2, H, W_, 3) and float(img.min()) >= 0 and float(img.max()) <= 1 and float(img.std()) > 0
    cm = ed.correspond_maps[(1, 1)]
    written = int(cm._writtens.sum())
    assert written > 0                                                   # VAEDecode.callback -> DefaultCorresponder.finished
    vals = cm._values.clone()

    ed2 = _engine_data(5)                                                # same frame again, fresh corr-map, by hand
    controls = [("control_v11p_sd15_normalbae_fp16.safetensors", ed2.normal_maps),
                ("control_v11f1p_sd15_depth_fp16.safetensors", ed2.depth_maps)] if graph == "bake" else []
    ref = _by_hand(ed2, controls)
    torch.cuda.synchronize()
    assert torch.equal(img, ref)
    assert torch.equal(vals, ed2.correspond_maps[(1, 1)]._values) and written == int(ed2.correspond_maps[(1, 1)]._writtens.sum())


@pytest.mark.parametrize("graph", ["miku-control", "no-normal-bake", "no-mask-prompt-bake"])
def test_other_shipped_graphs_run(graph, monkeypatch):
    """the remaining loadable example graphs: KSampler + lcm sampler with a seed and two ControlNets (miku-control), one ControlNet
    (no-normal-bake), plain CLIPTextEncode prompts (no-mask-prompt-bake); twice with the same inputs -> the same frames"""
    from stable_renderer_amd import workflow as W
    _register(monkeypatch)
    outs = []
    for _ in range(2):
        ed = _engine_data(7)
        ex = W.PromptExecutor(dev_mode=True)
        torch.manual_seed(5)
        ctx = W.run_workflow(os.path.join(WF, graph + ".json"), engine_data=ed, executor=ex)
        assert ctx.success
        img = ctx.final_output.frame_color.clone()
        assert tuple(img.shape) == (2, H, W_, 3) and bool(torch.isfinite(img).all()) and float(img.std()) > 0
        assert float(img.min()) >= 0 and float(img.max()) <= 1
        written = int(ed.correspond_maps[(1, 1)]._writtens.sum())
        assert (written > 0) == (graph != "miku-control")          # only the CorrespondSampler graphs carry the bake callback
        outs.append(img)
    assert torch.equal(outs[0], outs[1])


def test_img2img_graph_runs_end_to_end(monkeypatch, tmp_path):
    """miku-img2img-example-unix.json (the 'img2img' of the headline metric): FrameData colour -> VAEEncode -> KSampler (lcm, 4 steps,
    cfg 2, denoise 0.55) -> VAEDecode -> InferenceOutput, and its LoadImage -> VAEEncode branch when the engine sends no colour.
    The reference's HEAD cannot load this graph (it names the legacy ``FrameData`` node): the alias is opted into here."""
    import numpy as np
    from PIL import Image
    from stable_renderer_amd import graph_nodes as G, workflow as W
    _register(monkeypatch)
    monkeypatch.setitem(W.NODE_CLASS_MAPPINGS, "FrameData", None)
    W.NODE_CLASS_MAPPINGS.pop("FrameData")
    path = os.path.join(WF, "miku-img2img-example-unix.json")
    with pytest.raises(ValueError, match="Cannot find the type"):
        W.Workflow.Load(path)                                           # as the reference, without the alias
    G.register_legacy_aliases()
    try:
        ed = _engine_data(3, N=1)
        torch.manual_seed(11)
        ctx = W.run_workflow(path, engine_data=ed, executor=W.PromptExecutor(dev_mode=True))
        assert ctx.success, getattr(ctx, "error", None)
        img = ctx.final_output.frame_color.clone()
        assert tuple(img.shape) == (1, H, W_, 3) and bool(torch.isfinite(img).all()) and 0 <= float(img.min()) and float(img.max()) <= 1
        # denoise 0.55 keeps the encoded frame's structure: the result moves when the input colour moves
        ed2 = _engine_data(3, N=1)
        ed2.color_maps = 1.0 - ed2.color_maps
        torch.manual_seed(11)
        ctx2 = W.run_workflow(path, engine_data=ed2, executor=W.PromptExecutor(dev_mode=True))
        assert (ctx2.final_output.frame_color - img).abs().max().item() > 1e-3
        # no colour from the engine: If(IsNotNone(color)) takes the LoadImage branch
        rgb = (np.random.RandomState(0).rand(H, W_, 3) * 255).astype(np.uint8)
        Image.fromarray(rgb).save(tmp_path / "newplot-2.png")
        monkeypatch.setenv("SR_INPUT_DIR", str(tmp_path))
        ed3 = _engine_data(3, N=1)
        ed3.color_maps = None
        torch.manual_seed(11)
        ctx3 = W.run_workflow(path, engine_data=ed3, executor=W.PromptExecutor(dev_mode=True))
        assert ctx3.success and tuple(ctx3.final_output.frame_color.shape) == (1, H, W_, 3)
        assert "28" in ctx3.executed_node_ids and "27" in ctx3.executed_node_ids        # LoadImage + its VAEEncode ran
    finally:
        W.NODE_CLASS_MAPPINGS.pop("FrameData", None)


def test_loaders_are_cached_across_frames(monkeypatch):
    from stable_renderer_amd import workflow as W
    _register(monkeypatch)
    ex = W.PromptExecutor(dev_mode=True)
    wf = W.Workflow.Load(os.path.join(WF, "bake.json"))
    first = W.run_workflow(wf, engine_data=_engine_data(5), executor=ex)
    loaders = {"4", "11", "29", "31"}                                    # checkpoint, LoRA, two ControlNets
    assert loaders <= first.executed_node_ids
    img1 = first.final_output.frame_color.clone()
    second = W.run_workflow(wf, engine_data=_engine_data(6), executor=ex)
    assert second.success and not (loaders & second.executed_node_ids)
    assert {"38", "39", "37", "47", "8", "23", "30", "32"} <= second.executed_node_ids      # everything fed by the frame re-ran
    assert not torch.equal(img1, second.final_output.frame_color)
    same = W.run_workflow(wf, engine_data=second.engine_data, executor=ex)                   # identical frame: all cached
    assert same.success and same.executed_node_ids == set()


def test_missing_weights_fail_loudly(monkeypatch):
    from stable_renderer_amd import weights as WT, workflow as W
    monkeypatch.delenv("SR_MODELS_DIR", raising=False)
    WT.clear_registry()
    ctx = W.run_workflow(os.path.join(WF, "no-control-bake.json"), engine_data=_engine_data(5), executor=W.PromptExecutor(dev_mode=False))
    assert not ctx.success and ctx.final_output is None
    ev, mes = ctx.status_messages[-1]
    assert ev == "execution_error" and mes["node_type"] == "CheckpointLoaderSimple" and "FileNotFoundError" in mes["exception_type"]


def test_sdxl_family_checkpoint_through_the_nodes(monkeypatch):
    """an SDXL-topology checkpoint provider: KSampler builds y = [pooled | size embeddings] (SDXL.encode_adm) from the
    conditioning's pooled_output; a conditioning without it is refused"""
    from stable_renderer_amd import graph_nodes as G, synth, weights as WT
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.types import LATENT
    monkeypatch.setenv("SR_DTYPE", "fp32")
    monkeypatch.setenv("SR_AUTOTUNE", "0")
    cfg = dict(in_channels=4, out_channels=4, model_channels=64, num_res_blocks=[2, 2, 2], channel_mult=[1, 2, 4],
               transformer_depth=[0, 0, 1, 1, 2, 2], transformer_depth_middle=2, transformer_depth_output=[0, 0, 0, 1, 1, 1, 2, 2, 2],
               context_dim=128, num_heads=-1, num_head_channels=32, use_linear_in_transformer=True, adm_in_channels=1536 + 64)
    ns, norms = unet_names_shapes(cfg)
    vns, vnorms = vae_decoder_names_shapes(ch=32)
    WT.clear_registry()
    WT.register_checkpoint("sd_xl_tiny.safetensors", lambda: dict(
        unet=synth.synth_state_dict(ns, seed=1, norm_names=norms), vae=synth.synth_state_dict(vns, seed=3, norm_names=vnorms),
        clip=G.SyntheticCLIP(ctx_dim=128, pooled_dim=64), unet_cfg=cfg))
    model, clip, vae = G.CheckpointLoaderSimple().load_checkpoint("sd_xl_tiny.safetensors")
    (pos,), (neg,) = G.CLIPTextEncode().encode(clip, "a castle"), G.CLIPTextEncode().encode(clip, "blurry")
    lat = LATENT(samples=torch.zeros(2, 4, 16, 16, device="cuda"))
    (out,) = G.KSampler().sample(model, 11, 2, 4.0, "euler", "normal", pos, neg, lat)
    (img,) = G.VAEDecode().decode(vae, out)
    assert tuple(img.shape) == (2, 128, 128, 3) and bool(torch.isfinite(img).all()) and float(img.std()) > 0
    with pytest.raises(ValueError, match="pooled_output"):
        G.KSampler().sample(model, 11, 2, 4.0, "euler", "normal", torch.zeros(1, 77, 128), torch.zeros(1, 77, 128), lat)
    WT.clear_registry()
