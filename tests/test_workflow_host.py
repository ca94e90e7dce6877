"""Workflow.Load / build_prompt / PromptExecutor and the LoRA merge, on CPU.

Goldens (oracle/gen_golden.py workflow): tests/golden/workflow_prompts.json = the reference's own Workflow.build_prompt on the
shipped example graphs (tests/golden/workflows/*.json are those data files), or the exception the reference raises on them;
lora_key_map_sd15.json / lora_merge.npz = comfy's key map and calculate_weight."""
import json
import os

import numpy as np
import pytest
import torch

from stable_renderer_amd import weights as WT
from stable_renderer_amd import workflow as W
from stable_renderer_amd.types import EngineData

GOLD = os.path.join(os.path.dirname(__file__), "golden")
WF = os.path.join(GOLD, "workflows")
with open(os.path.join(GOLD, "workflow_prompts.json")) as f:
    REF = json.load(f)


def _plain(prompt):
    return {k: {"inputs": {a: (list(b) if isinstance(b, list) else b) for a, b in v["inputs"].items()}, "class_type": v["class_type"]}
            for k, v in prompt.items()}


def test_build_prompt_matches_the_reference_where_the_reference_loads():
    ok = [n for n, r in REF.items() if "prompt" in r]
    assert ok == ["no-mask-prompt-bake"]
    for name in ok:
        wf = W.Workflow.Load(os.path.join(WF, name + ".json"))
        prompt, ids, extra = wf.build_prompt()
        assert _plain(prompt) == REF[name]["prompt"]
        assert ids == REF[name]["node_ids_to_be_ran"]
        assert wf.has_output_node == REF[name]["has_output_node"]
        assert wf.name == name and extra["extra_pnginfo"]["workflow"]["nodes"]


def test_unknown_node_type_is_the_reference_error():
    for name, r in REF.items():
        if r.get("error") == "ValueError":
            with pytest.raises(ValueError, match="Cannot find the type"):
                W.Workflow.Load(os.path.join(WF, name + ".json"))


def test_graphs_the_reference_cannot_load_take_the_optional_default():
    """bake / miku-control / no-control-bake / no-normal-bake: the reference dies with KeyError('merge') (workflow.py:190);
    here SceneTextEncode.merge takes its default and everything else follows the same translation rules"""
    names = [n for n, r in REF.items() if r.get("error") == "KeyError"]
    assert sorted(names) == ["bake", "miku-control", "no-control-bake", "no-normal-bake"]
    for name in names:
        prompt, ids, _ = W.Workflow.Load(os.path.join(WF, name + ".json")).build_prompt()
        assert ids == ["23"]
        scene = [v for v in prompt.values() if v["class_type"] == "SceneTextEncode"][0]
        assert scene["inputs"]["merge"] is True and scene["inputs"]["clip"][1] == 1
        samp = [v for v in prompt.values() if v["class_type"] in ("CorrespondSampler", "KSampler")][0]["inputs"]
        assert samp["steps"] == 4 and samp["cfg"] == 2 and samp["scheduler"] == "sgm_uniform"
    p = W.Workflow.Load(os.path.join(WF, "bake.json")).build_prompt()[0]
    assert p["30"]["inputs"]["conditioning"] == ["32", 0] and p["30"]["inputs"]["image"] == ["38", 4]        # depth after normal
    assert p["32"]["inputs"]["image"] == ["38", 3] and p["8"]["inputs"]["callback"] == ["37", 1]


def test_load_errors():
    with pytest.raises(FileNotFoundError):
        W.Workflow.Load("/nonexistent/x.json")
    with pytest.raises(ValueError):
        W.Workflow.Load(WF)
    with pytest.raises(ValueError, match="cannot find `nodes`"):
        W.Workflow({"links": []})


# ---- executor semantics on toy nodes -----------------------------------------------------------------------------------
CALLS = []


class _Src:
    RETURN_TYPES = ("INT",)
    FUNCTION = "go"

    @classmethod
    def INPUT_TYPES(s):
        return {"required": {"v": ("INT", {"default": 0})}}

    def go(self, v):
        CALLS.append(("src", v))
        return (v,)


class _Add:
    def __call__(self, a, b=10):
        CALLS.append(("add", a, b))
        return a + b


class _Boom:
    def __call__(self, a):
        CALLS.append(("boom",))
        raise RuntimeError("boom")


class _Frame:
    """like EngineDataNode: hidden EngineData in, IsChanged = its serial"""
    N_OUTPUTS = 2

    def IsChanged(self, engine_data: EngineData):
        return engine_data.serial

    def __call__(self, engine_data: EngineData):
        CALLS.append(("frame", engine_data.serial))
        return engine_data.serial, len(engine_data.frame_indices)


class _Prior:
    PriorNode = True

    def __call__(self, context=None):
        CALLS.append(("prior",))
        context.engine_data = EngineData(frame_indices=[0, 1, 2])
        return context.engine_data


class _Out:
    IsOutputNode = True

    def __call__(self, x, context=None):
        CALLS.append(("out", x))
        context.final_output = x
        return x


for n, c in (("T_Src", _Src), ("T_Add", _Add), ("T_Boom", _Boom), ("T_Frame", _Frame), ("T_Prior", _Prior), ("T_Out", _Out)):
    W.register_node(n, c)


def _run(ex, prompt, **kw):
    CALLS.clear()
    return ex.execute(prompt, node_ids_to_be_ran=["9"], **kw)


def test_executor_caches_and_invalidates():
    ex = W.PromptExecutor(dev_mode=False)
    prompt = {"1": {"class_type": "T_Src", "inputs": {"v": 3}},
              "2": {"class_type": "T_Add", "inputs": {"a": ["1", 0], "b": 4}},
              "9": {"class_type": "T_Out", "inputs": {"x": ["2", 0]}}}
    ctx = _run(ex, prompt)
    assert ctx.success and ctx.final_output == 7 and CALLS == [("src", 3), ("add", 3, 4), ("out", 7)]
    ctx = _run(ex, prompt)                                   # nothing changed: every output is reused
    assert ctx.success and CALLS == [] and ctx.outputs["9"] == [7]
    prompt["2"]["inputs"]["b"] = 5                           # a changed widget re-runs the node and what depends on it
    ctx = _run(ex, prompt)
    assert CALLS == [("add", 3, 5), ("out", 8)] and ctx.outputs["9"] == [8]
    prompt["1"]["inputs"]["v"] = 1                           # upstream change propagates
    ctx = _run(ex, prompt)
    assert CALLS == [("src", 1), ("add", 1, 5), ("out", 6)]


def test_hidden_engine_data_and_ischanged():
    ex = W.PromptExecutor(dev_mode=False)
    prompt = {"1": {"class_type": "T_Frame", "inputs": {}},
              "5": {"class_type": "T_Src", "inputs": {"v": 2}},
              "2": {"class_type": "T_Add", "inputs": {"a": ["1", 1], "b": ["5", 0]}},
              "9": {"class_type": "T_Out", "inputs": {"x": ["2", 0]}}}
    ed = EngineData(frame_indices=[0, 1])
    ctx = _run(ex, prompt, frame_data=ed)
    assert ctx.final_output == 4 and ("frame", ed.serial) in CALLS and ("src", 2) in CALLS
    _run(ex, prompt, frame_data=ed)                          # same frame: cached
    assert CALLS == []
    ed2 = EngineData(frame_indices=[0, 1, 2, 3])
    ctx = _run(ex, prompt, frame_data=ed2)                   # new frame: the frame node and its consumers re-run, loaders do not
    assert [c[0] for c in CALLS] == ["frame", "add", "out"] and ctx.final_output == 6


def test_prior_nodes_run_first_and_set_engine_data():
    ex = W.PromptExecutor(dev_mode=False)
    prompt = {"9": {"class_type": "T_Out", "inputs": {"x": ["1", 1]}},
              "1": {"class_type": "T_Frame", "inputs": {}},
              "7": {"class_type": "T_Prior", "inputs": {}}}
    ctx = _run(ex, prompt)
    assert ctx.success and CALLS[0] == ("prior",) and ctx.final_output == 3


def test_lazy_if_runs_one_branch_only():
    ex = W.PromptExecutor(dev_mode=False)
    prompt = {"1": {"class_type": "T_Src", "inputs": {"v": 1}},
              "2": {"class_type": "T_Boom", "inputs": {"a": ["1", 0]}},
              "3": {"class_type": "If", "inputs": {"condition": True, "true_value": ["1", 0], "false_value": ["2", 0]}},
              "4": {"class_type": "IsNotNone", "inputs": {"value": ["3", 0], "mode": "strict"}},
              "9": {"class_type": "T_Out", "inputs": {"x": ["4", 0]}}}
    ctx = _run(ex, prompt)
    assert ctx.success and ctx.final_output is True and ("boom",) not in CALLS


def test_errors_become_status_messages():
    ex = W.PromptExecutor(dev_mode=False)
    prompt = {"1": {"class_type": "T_Src", "inputs": {"v": 1}},
              "2": {"class_type": "T_Boom", "inputs": {"a": ["1", 0]}},
              "9": {"class_type": "T_Out", "inputs": {"x": ["2", 0]}}}
    ctx = _run(ex, prompt)
    assert not ctx.success and ctx.final_output is None
    ev, mes = ctx.status_messages[-1]
    assert ev == "execution_error" and mes["node_id"] == "2" and mes["node_type"] == "T_Boom"
    assert mes["exception_type"].endswith("RuntimeError") and mes["exception_message"] == "boom" and "1" in mes["executed"]
    with pytest.raises(RuntimeError):                         # dev mode re-raises (execution.py:1094-1097)
        W.PromptExecutor(dev_mode=True).execute(prompt, node_ids_to_be_ran=["9"])
    bad = {"9": {"class_type": "NoSuchNode", "inputs": {}}}
    ctx = W.PromptExecutor(dev_mode=False).execute(bad, node_ids_to_be_ran=["9"])
    assert not ctx.success


def test_node_input_types_from_signatures():
    t = W.node_input_types(W.get_node_cls_by_name("CorrespondSampler"))
    assert list(t["required"]) == ["model", "positive", "negative", "corresponder"] and t["hidden"] == {"engine_data": "ENGINE_DATA"}
    assert t["optional"]["steps"][1]["default"] == 20 and t["optional"]["latent"][1]["default"] is None
    t = W.node_input_types(W.get_node_cls_by_name("InferenceOutput"))
    assert "save" in t["optional"] and t["hidden"] == {"context": "INFERENCE_CONTEXT"}
    assert W.node_lazy_inputs(W.get_node_cls_by_name("If")) == ("true_value", "false_value")


# ---- LoRA ------------------------------------------------------------------------------------------------------------------
def test_lora_key_map_matches_comfy():
    from stable_renderer_amd.model_shapes import unet_names_shapes
    from stable_renderer_amd.unet import SD15_CFG
    with open(os.path.join(GOLD, "lora_key_map_sd15.json")) as f:
        ref = json.load(f)
    names, _ = unet_names_shapes(SD15_CFG)
    km = WT.unet_lora_key_map(SD15_CFG, [n for n, _ in names])
    assert km == ref
    assert km["lora_unet_down_blocks_0_attentions_0_transformer_blocks_0_attn1_to_q"] == "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight"
    assert km["lora_unet_up_blocks_1_upsamplers_0_conv"] == "output_blocks.5.2.conv.weight"


def test_lora_merge_matches_calculate_weight():
    d = np.load(os.path.join(GOLD, "lora_merge.npz"))
    for name in ("linear", "conv3", "conv1"):
        w, up, down = (torch.from_numpy(d[f"{name}_{k}"]) for k in ("w", "up", "down"))
        for tag, alpha, strength in (("a", None, 1.0), ("b", 2.0, 0.75)):
            lora = {"lora_unet_m.lora_up.weight": up, "lora_unet_m.lora_down.weight": down, "stray.key": torch.zeros(1)}
            if alpha is not None:
                lora["lora_unet_m.alpha"] = torch.tensor(alpha)
            out, unused = WT.apply_lora({"m.weight": w, "other.weight": w}, lora, strength, {"lora_unet_m": "m.weight"})
            assert torch.equal(out["m.weight"], torch.from_numpy(d[f"{name}_{tag}_out"])), (name, tag)
            assert out["other.weight"] is w and unused == ["stray.key"]


def test_weight_registry_and_files(tmp_path, monkeypatch):
    WT.register_lora("a\\b.safetensors", lambda: {"k": torch.ones(1)})
    assert WT.resolve("loras", "a/b.safetensors")["k"].item() == 1.0
    with pytest.raises(FileNotFoundError, match="register_"):
        WT.resolve("checkpoints", "dreamshaper_8.safetensors")
    from safetensors.torch import save_file
    os.makedirs(tmp_path / "checkpoints")
    save_file({"model.diffusion_model.out.2.bias": torch.zeros(4), "first_stage_model.decoder.conv_in.bias": torch.ones(2),
               "cond_stage_model.x": torch.ones(1)}, str(tmp_path / "checkpoints" / "m.safetensors"))
    monkeypatch.setenv("SR_MODELS_DIR", str(tmp_path))
    unet, vae, clip = WT.split_checkpoint(WT.resolve("checkpoints", "m.safetensors"))
    assert list(unet) == ["out.2.bias"] and list(vae) == ["decoder.conv_in.bias"] and list(clip) == ["x"]
    WT._REG["loras"].pop("a/b.safetensors")
