"""Workflow graphs without the ComfyUI server: ``Workflow.Load`` / ``build_prompt``
(engine/static/workflow.py:36-594) and a minimal ``PromptExecutor`` (comfyUI/execution.py:540-1165) over this package's
node classes, so the shipped ``resources/example-workflows/*.json`` graphs run unchanged on the HIP path.

What is kept from the reference: the UI-JSON -> prompt translation (links -> ``[from_node, slot]`` bindings,
``widget_kw_values`` / defaults, invalid-node pruning), PRIOR nodes first then output nodes, recursive dependency execution,
hidden inputs (``EngineData``, ``InferenceContext``, ``PROMPT``, ``UNIQUE_ID``) injected by the executor, lazy inputs
(``If``), per-node output caching across ``execute`` calls with ``IsChanged`` + changed-inputs invalidation, and errors
reported as ``context.success = False`` with an ``execution_error`` status message instead of an exception.
What is not: the web server / UI messages, ``__server_call__``, type adapters, list-mapped execution (``INPUT_IS_LIST``).

One deliberate deviation: an optional input that has a default but no widget value (SceneTextEncode.merge in bake.json)
makes the reference raise KeyError at workflow.py:190 (it indexes the *required* table); here the default is taken."""
import inspect
import json
import os
import sys
import traceback
import typing
import uuid
from typing import Dict, List, Optional

from .types import EngineData, InferenceOutput


# ---- prompt-side types --------------------------------------------------------------------------------------------------
class NodeBindingParam(list):
    """[from_node_id, from_output_slot] (comfyUI/types/runtime.py NodeBindingParam)"""

    @property
    def from_node_id(self):
        return self[0]

    @property
    def from_output_slot(self):
        return self[1]


class PROMPT(dict):
    def __init__(self, data=None, id=None):
        super().__init__()
        self.id = id or uuid.uuid4().hex
        for k, v in (data or {}).items():
            node = dict(v)
            node["inputs"] = {n: (NodeBindingParam(x) if _is_binding(x) else x) for n, x in v.get("inputs", {}).items()}
            self[str(k)] = node

    def get_node_type_name(self, node_id):
        return self[str(node_id)]["class_type"]


def _is_binding(x):
    return isinstance(x, (list, tuple)) and len(x) == 2 and isinstance(x[0], str) and isinstance(x[1], int) \
        and not isinstance(x[1], bool)


class InferenceContext:
    """comfyUI/types/hidden.py InferenceContext: what one ``execute`` carries"""

    def __init__(self, prompt, extra_data=None, outputs=None, engine_data=None):
        self.prompt, self.extra_data = prompt, extra_data or {}
        self.outputs: Dict[str, list] = outputs if outputs is not None else {}
        self.engine_data: Optional[EngineData] = engine_data
        self.current_node_id: Optional[str] = None
        self.status_messages: List[tuple] = []
        self.executed_node_ids = set()
        self.to_be_executed: List[tuple] = []
        self.success = False
        self.final_output: Optional[InferenceOutput] = None

    def remove_from_execute_waitlist(self, node_id):
        self.to_be_executed = [t for t in self.to_be_executed if t[-1] != node_id]


class Lazy:
    """An input that is computed only when ``.value`` is read (comfyUI/types/runtime.py Lazy; IfNode, logic.py:38-75)"""

    def __init__(self, executor, context, from_node_id, from_output_slot):
        self._ex, self._ctx, self.from_node_id, self.from_output_slot = executor, context, from_node_id, from_output_slot

    @property
    def value(self):
        if self.from_node_id not in self._ctx.outputs:
            ok, err, ex = self._ex._recursive_execute(self._ctx, self.from_node_id)
            if not ok:
                raise ex if ex is not None else RuntimeError(str(err))
        return self._ctx.outputs[self.from_node_id][self.from_output_slot]


# ---- node registry ------------------------------------------------------------------------------------------------------
NODE_CLASS_MAPPINGS: Dict[str, type] = {}
_HIDDEN_BY_TYPE = {EngineData: "ENGINE_DATA", InferenceContext: "INFERENCE_CONTEXT", PROMPT: "PROMPT"}
_HIDDEN_BY_NAME = {"engine_data": "ENGINE_DATA", "context": "INFERENCE_CONTEXT", "prompt": "PROMPT", "unique_id": "UNIQUE_ID",
                   "png_info": "EXTRA_PNGINFO"}


def register_node(name, cls):
    NODE_CLASS_MAPPINGS[name] = cls
    return cls


def get_node_cls_by_name(name):
    _ensure_default_nodes()
    return NODE_CLASS_MAPPINGS.get(name)


def node_function(cls):
    """the callable a node executes: FUNCTION (comfy style) or ``__call__`` (StableRenderingNode style)"""
    fn = getattr(cls, "FUNCTION", None)
    return getattr(cls, fn) if fn else cls.__call__


def _is_lazy_annotation(a):
    return a is Lazy or typing.get_origin(a) is Lazy or (isinstance(a, str) and a.startswith("Lazy"))


def node_input_types(cls):
    """-> {'required': {name: (type_name, {default})}, 'optional': {...}, 'hidden': {name: type_name}} derived from the
    node function's signature the way AdvancedNodeBase does (node_base.py:330-418): parameters without a default are
    required, with one optional; ``EngineData`` / ``InferenceContext`` / ``PROMPT`` parameters are hidden."""
    if "INPUT_TYPES" in cls.__dict__:
        return cls.INPUT_TYPES()
    sig = inspect.signature(node_function(cls))
    out = {"required": {}, "optional": {}, "hidden": {}}
    for i, (pname, p) in enumerate(sig.parameters.items()):
        if i == 0 and pname in ("self", "cls", "s"):
            continue
        if p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
            continue
        ann = p.annotation
        hidden = _HIDDEN_BY_TYPE.get(ann) if isinstance(ann, type) else None
        if hidden is None and pname in _HIDDEN_BY_NAME and (ann is inspect.Parameter.empty or isinstance(ann, type) and ann in _HIDDEN_BY_TYPE):
            hidden = _HIDDEN_BY_NAME[pname]
        if hidden is None and pname in getattr(cls, "HIDDEN_INPUTS", ()):
            hidden = _HIDDEN_BY_NAME.get(pname, pname.upper())
        if hidden is not None:
            out["hidden"][pname] = hidden
            continue
        tname = getattr(ann, "__name__", str(ann)).upper() if ann is not inspect.Parameter.empty else "*"
        if p.default is inspect.Parameter.empty:
            out["required"][pname] = (tname, {})
        else:
            out["optional"][pname] = (tname, {"default": p.default})
    return out


def node_lazy_inputs(cls):
    if hasattr(cls, "LAZY_INPUTS"):
        return tuple(cls.LAZY_INPUTS)
    try:
        sig = inspect.signature(node_function(cls))
    except (TypeError, ValueError):
        return ()
    return tuple(n for n, p in sig.parameters.items() if _is_lazy_annotation(p.annotation))


_defaults_done = False


def _ensure_default_nodes():
    global _defaults_done
    if _defaults_done:
        return
    _defaults_done = True
    from . import graph_nodes                    # noqa: F401  (registers on import)


# ---- UI JSON -> prompt ---------------------------------------------------------------------------------------------------
class InvalidNodeError(Exception):
    pass


class WorkflowNodeLink(tuple):
    """(id, from_node, from_slot, to_node, to_slot, type) — workflow.py:36-66"""
    id = property(lambda s: s[0])
    from_node_id = property(lambda s: str(s[1]))
    from_output_slot = property(lambda s: s[2])
    to_node_id = property(lambda s: str(s[3]))
    to_input_slot = property(lambda s: s[4])
    val_type = property(lambda s: s[5])

    def to_node_binding_param(self):
        return NodeBindingParam([self.from_node_id, self.from_output_slot])


class WorkflowNodeInfo(dict):
    def __init__(self, origin, workflow):
        super().__init__(origin)
        self.workflow = workflow
        self["id"] = str(self["id"])
        cls = get_node_cls_by_name(self["type"])
        if cls is None:
            raise ValueError(f"Cannot find the type {self['type']}.")          # workflow.py:33-35
        self.cls_type = cls
        spec = node_input_types(cls)
        req, opt = spec.get("required", {}), spec.get("optional", {})
        links = workflow["links"]
        linked = {i["name"]: i for i in (self.get("inputs") or [])}
        kw = dict(self.get("widget_kw_values") or {})
        wlist = self.get("widgets_values") or []
        if not workflow.is_stable_renderer_workflow and isinstance(wlist, dict):
            kw = dict(wlist)
        positional = list(wlist) if (not workflow.is_stable_renderer_workflow and isinstance(wlist, list)) else None

        def default_of(name):
            info = req.get(name) or opt.get(name)
            return info[1].get("default") if info is not None and len(info) >= 2 else None

        inputs = {}
        for name in list(req) + list(opt):
            if name in linked:
                link_id = linked[name].get("link")
                if link_id:
                    inputs[name] = links[link_id].to_node_binding_param()
                elif default_of(name) is not None:
                    inputs[name] = default_of(name)
                elif name in opt:
                    inputs[name] = None
                else:
                    raise InvalidNodeError(f"Cannot find the link id for input {name}.")
                continue
            if positional is not None:                        # plain ComfyUI export: widget values in declaration order
                if positional:
                    inputs[name] = positional.pop(0)
                elif default_of(name) is not None:
                    inputs[name] = default_of(name)
                elif name in opt:
                    inputs[name] = None
                else:
                    raise ValueError(f"Cannot find the widget value for input {name}.")
                continue
            val = kw.pop(name, None)
            if val:                                           # truthiness, as workflow.py:215 (False / 0 / "" fall through)
                inputs[name] = val
            elif name in req:
                if default_of(name) is not None:
                    inputs[name] = default_of(name)
                elif not kw and not self.get("widget_kw_values"):
                    raise ValueError(f"Cannot find the widget value for input {name}.")
                # else: skipped, __call__ and __server_call__ may differ (workflow.py:222-225)
            elif not self.get("widget_kw_values") and default_of(name) is not None:
                inputs[name] = default_of(name)               # the reference raises KeyError here (see module docstring)
            else:
                inputs[name] = None
        self.inputs = inputs
        self.outputs = [dict(name=o["name"], type_name=o["type"], slot=i, to_nodes=[str(x) for x in (o.get("links") or [])])
                        for i, o in enumerate(self.get("outputs") or []) if o.get("links")]

    cls_type_name = property(lambda s: s["type"])
    id = property(lambda s: s["id"])


class Workflow(dict):
    """engine/static/workflow.py:381-575"""

    def __init__(self, *args, **kwargs):
        if len(args) == 1 and not kwargs and isinstance(args[0], str):
            super().__init__(json.loads(args[0]))
        else:
            super().__init__(*args, **kwargs)
        self.original_data = json.loads(json.dumps(dict(self)))
        self["links"] = {l[0]: WorkflowNodeLink(l) for l in self.get("links", [])}
        if "nodes" not in self:
            raise ValueError("Invalid workflow file, cannot find `nodes`.")
        self["nodes"] = self._parse_nodes(self["nodes"])

    @property
    def is_stable_renderer_workflow(self):
        return self.get("stable_renderer_version") is not None

    name = property(lambda s: s.get("name"))
    nodes = property(lambda s: s["nodes"])
    node_links = property(lambda s: s["links"])
    version = property(lambda s: None if s.get("version") is None else str(s.get("version")))

    @property
    def has_output_node(self):
        return any(_is_output_node(n.cls_type) for n in self.nodes.values())

    def _parse_nodes(self, infos):
        datas, invalid = {}, set()
        for info in infos:
            info = dict(info)
            info["id"] = str(info["id"])
            try:
                datas[info["id"]] = WorkflowNodeInfo(info, self)
            except InvalidNodeError:
                invalid.add(info["id"])                       # e.g. a forgotten node with a dangling required input
        changed = True
        while changed:                                         # nodes fed by an invalid node are invalid too
            changed = False
            for nid, d in list(datas.items()):
                if any(isinstance(v, NodeBindingParam) and v[0] in invalid for v in d.inputs.values()):
                    del datas[nid]
                    invalid.add(nid)
                    changed = True
        for d in datas.values():
            for o in d.outputs:
                o["to_nodes"] = [t for t in o["to_nodes"] if t not in invalid]
        return datas

    def build_prompt(self):
        """-> (PROMPT, ids of the OUTPUT nodes to run, extra_data) — workflow.py:488-519"""
        ids = sorted(self.nodes, key=int)
        prompt, to_run = {}, []
        for nid in ids:
            n = self.nodes[nid]
            prompt[nid] = {"inputs": dict(n.inputs), "class_type": n.cls_type_name}
            if _is_output_node(n.cls_type):
                to_run.append(nid)
        return PROMPT(prompt), to_run, {"extra_pnginfo": {"workflow": self.original_data}}

    @classmethod
    def Load(cls, path):
        path = str(path)
        if not os.path.exists(path) and not path.endswith(".json"):
            alt = os.path.join(os.environ.get("SR_WORKFLOW_DIR", ""), path + ".json")
            path = alt if os.path.exists(alt) else path
        if not os.path.exists(path):
            raise FileNotFoundError(f"Cannot find the workflow file {path}.")
        if not os.path.isfile(path):
            raise ValueError(f"The path {path} is not a file.")
        with open(path) as f:
            wf = cls(json.load(f))
        wf["name"] = os.path.basename(path).split(".")[0]
        return wf


def _is_output_node(cls):
    return bool(getattr(cls, "OUTPUT_NODE", False) or getattr(cls, "IsOutputNode", False))


def _is_prior_node(cls):
    return bool(getattr(cls, "PRIOR_NODE", False) or getattr(cls, "PriorNode", False))


# ---- executor -------------------------------------------------------------------------------------------------------------
class PromptExecutor:
    """Runs a PROMPT on this package's nodes.  One instance is kept across frames: node objects (and what they cache: loaded
    models, launch plans, captured hipGraphs) live in ``node_pool``; outputs of nodes whose inputs and ``IsChanged`` value did
    not change are reused (execution.py:995-1165)."""
    instance: Optional["PromptExecutor"] = None

    def __init__(self, dev_mode=None):
        self.node_pool: Dict[tuple, object] = {}
        self.outputs: Dict[tuple, list] = {}
        self.all_prompts: Dict[tuple, dict] = {}
        self.latest_context: Optional[InferenceContext] = None
        self.dev_mode = (os.environ.get("DEV_MODE", "0") == "1") if dev_mode is None else dev_mode
        PromptExecutor.instance = self

    def reset(self):
        self.node_pool.clear()
        self.outputs.clear()
        self.all_prompts.clear()

    # -- helpers
    def _node(self, node_id, cls_name):
        key = (node_id, cls_name)
        if key not in self.node_pool:
            cls = get_node_cls_by_name(cls_name)
            if cls is None:
                raise ValueError(f"Node class `{cls_name}` not found.")
            obj = cls()
            obj.ID = node_id
            self.node_pool[key] = obj
        return self.node_pool[key]

    def _get_input_data(self, inputs, node_id, context, lazy=()):
        cls_name = context.prompt[node_id]["class_type"]
        cls = get_node_cls_by_name(cls_name)
        spec = node_input_types(cls)
        known = set(spec.get("required", {})) | set(spec.get("optional", {}))
        data = {}
        for name, val in inputs.items():
            if name not in known:
                continue                                        # e.g. `save` of InferenceOutput.__server_call__
            if isinstance(val, NodeBindingParam):
                src, slot = val
                if name in lazy and src not in context.outputs:
                    data[name] = Lazy(self, context, src, slot)
                    continue
                if src not in context.outputs:
                    continue
                data[name] = context.outputs[src][slot]
                if name in lazy:
                    data[name] = _Ready(data[name])
            else:
                data[name] = val
        for name, hidden in spec.get("hidden", {}).items():
            hidden = hidden[0] if isinstance(hidden, tuple) else hidden
            if hidden == "PROMPT":
                data[name] = context.prompt
            elif hidden == "UNIQUE_ID":
                data[name] = node_id
            elif hidden == "EXTRA_PNGINFO":
                data[name] = context.extra_data.get("extra_pnginfo")
            elif hidden == "ENGINE_DATA":
                data[name] = context.engine_data
            elif hidden == "INFERENCE_CONTEXT":
                data[name] = context
            elif name in context.extra_data:
                data[name] = context.extra_data[name]
        return data

    def _is_changed(self, node, cls, data):
        fn = getattr(cls, "IsChanged", None) or getattr(cls, "IS_CHANGED", None)
        if fn is None:
            return ""
        names = set(inspect.signature(fn).parameters)
        kw = {k: v for k, v in data.items() if k in names}
        return getattr(node, fn.__name__)(**kw)

    def _delete_if_changed(self, context, node_id, memo):
        """-> True when the cached output of ``node_id`` had to be dropped (execution.py:839-930)"""
        if node_id in memo:
            return memo[node_id]
        prompt = context.prompt
        cls_name = prompt[node_id]["class_type"]
        cls = get_node_cls_by_name(cls_name)
        if cls is None:
            raise ValueError(f"Node class `{cls_name}` not found.")
        inputs = prompt[node_id]["inputs"]
        key = (node_id, cls_name)
        old = self.all_prompts.get(key)
        to_delete = False
        changed = ""
        if getattr(cls, "IsChanged", None) or getattr(cls, "IS_CHANGED", None):
            try:
                changed = self._is_changed(self._node(node_id, cls_name), cls, self._get_input_data(inputs, node_id, context))
            except Exception:
                if self.dev_mode:
                    raise
                to_delete = True
            prompt[node_id]["is_changed"] = changed
        elif "ENGINE_DATA" in [h[0] if isinstance(h, tuple) else h for h in node_input_types(cls).get("hidden", {}).values()]:
            # a node fed the hidden EngineData (DefaultCorresponder binds it into its VAE-decode callback, CorrespondSampler
            # reads its id maps) must not be served from the previous frame's cache: the reference compares only the visible
            # inputs here (execution.py:908-921) and would keep the first frame's EngineData in that callback
            changed = None if context.engine_data is None else ("engine_data", context.engine_data.serial)
            prompt[node_id]["is_changed"] = changed
        if node_id not in context.outputs:
            memo[node_id] = True
            return True
        if not to_delete:
            if old is None or changed != old.get("is_changed", "") or _plain(inputs) != old["inputs"]:
                to_delete = True
            else:
                for v in inputs.values():
                    if isinstance(v, NodeBindingParam):
                        if v[0] not in context.outputs or self._delete_if_changed(context, v[0], memo):
                            to_delete = True
                            break
        if to_delete:
            context.outputs.pop(node_id, None)
        memo[node_id] = to_delete
        return to_delete

    def _will_execute(self, context, node_id, memo):
        if node_id in memo:
            return memo[node_id]
        if node_id in context.outputs:
            return []
        todo = []
        for v in context.prompt[node_id]["inputs"].values():
            if isinstance(v, NodeBindingParam) and v[0] not in context.outputs:
                todo += self._will_execute(context, v[0], memo)
        memo[node_id] = todo + [node_id]
        return memo[node_id]

    def _recursive_execute(self, context, node_id):
        """-> (success, error_details, exception)"""
        prompt = context.prompt
        context.current_node_id = node_id
        if node_id in context.outputs:
            return True, None, None
        cls_name = prompt[node_id]["class_type"]
        cls = get_node_cls_by_name(cls_name)
        if cls is None:
            return False, {"node_id": node_id}, ValueError(f"Node class `{cls_name}` not found.")
        inputs = prompt[node_id]["inputs"]
        lazy = node_lazy_inputs(cls)
        for name, v in inputs.items():
            if isinstance(v, NodeBindingParam) and v[0] not in context.outputs and name not in lazy:
                res = self._recursive_execute(context, v[0])
                if not res[0]:
                    return res
        context.current_node_id = node_id
        data = None
        try:
            data = self._get_input_data(inputs, node_id, context, lazy)
            node = self._node(node_id, cls_name)
            fn = getattr(cls, "FUNCTION", None)
            call = getattr(node, fn) if fn else node
            params = inspect.signature(call).parameters
            if not any(p.kind == p.VAR_KEYWORD for p in params.values()):
                data = {k: v for k, v in data.items() if k in params}      # e.g. InferenceOutput's server-only `save`
            out = call(**data)
            n_out = _n_outputs(cls)
            if isinstance(out, dict) and "result" in out:          # comfy UI nodes: {"ui": ..., "result": (...)}
                out = out["result"]
            if not isinstance(out, tuple) or (n_out == 1 and fn is None):
                out = (out,)
            context.outputs[node_id] = list(out)
        except Exception as ex:
            if self.dev_mode:
                raise
            typ, _, tb = sys.exc_info()
            err = {"node_id": node_id, "exception_message": str(ex), "exception_type": f"{typ.__module__}.{typ.__name__}",
                   "traceback": traceback.format_tb(tb), "current_inputs": {k: _fmt(v) for k, v in (data or {}).items()},
                   "current_outputs": {k: [_fmt(x) for x in v] for k, v in context.outputs.items()}}
            return False, err, ex
        context.executed_node_ids.add(node_id)
        return True, None, None

    # -- entry
    def execute(self, prompt, prompt_id=None, extra_data=None, node_ids_to_be_ran=None, frame_data=None):
        """-> InferenceContext (``success``, ``final_output``, ``outputs``, ``status_messages``)"""
        import torch
        _ensure_default_nodes()
        if not isinstance(prompt, PROMPT):
            prompt = PROMPT(prompt, id=prompt_id)
        to_run = [str(x) for x in (node_ids_to_be_ran or [])]
        outputs = {nid: self.outputs[(nid, prompt[nid]["class_type"])] for nid in prompt
                   if (nid, prompt[nid]["class_type"]) in self.outputs}
        for key in [k for k in self.outputs if k[0] not in prompt or prompt[k[0]]["class_type"] != k[1]]:
            del self.outputs[key]
        context = InferenceContext(prompt, extra_data, outputs, frame_data)
        self.latest_context = context
        context.status_messages.append(("execution_start", {"prompt_id": prompt.id}))
        with torch.inference_mode():
            for key in [k for k in self.node_pool if k[0] not in prompt or prompt[k[0]]["class_type"] != k[1]]:
                del self.node_pool[key]
            memo = {}
            for nid in prompt:
                try:
                    self._delete_if_changed(context, nid, memo)
                except Exception as ex:
                    if self.dev_mode:
                        raise
                    self._handle_error(context, {"node_id": nid, "exception_message": str(ex), "exception_type": type(ex).__name__,
                                                 "traceback": [], "current_inputs": {}, "current_outputs": {}}, ex)
                    return context
            context.status_messages.append(("execution_cached", {"nodes": list(context.outputs), "prompt_id": prompt.id}))
            for nid in prompt:
                cls = get_node_cls_by_name(prompt[nid]["class_type"])
                if cls is not None and _is_prior_node(cls):
                    context.to_be_executed.append((0, nid))
            for nid in to_run:
                context.to_be_executed.append((1, nid))
            context.success = True
            while context.to_be_executed:
                # PRIOR nodes first; among equals the one depending on the fewest unexecuted nodes
                wm = {}
                context.to_be_executed = sorted(((t[0], len(self._will_execute(context, t[-1], wm)), t[-1])
                                                 for t in context.to_be_executed))
                nid = context.to_be_executed.pop(0)[-1]
                ok, err, ex = self._recursive_execute(context, nid)
                context.success = ok
                if not ok:
                    self._handle_error(context, err, ex)
                    break
        for nid in prompt:
            key = (nid, prompt[nid]["class_type"])
            if nid in context.outputs:
                self.outputs[key] = context.outputs[nid]
            self.all_prompts[key] = {"inputs": _plain(prompt[nid]["inputs"]), "is_changed": prompt[nid].get("is_changed", "")}
        return context

    def _handle_error(self, context, err, ex):
        context.success = False
        nid = err.get("node_id")
        mes = dict(err, prompt_id=context.prompt.id, node_type=context.prompt[nid]["class_type"] if nid in context.prompt else None,
                   executed=list(context.executed_node_ids))
        context.status_messages.append(("execution_error", mes))
        for key in [k for k in self.outputs if k[0] not in context.outputs and k[0] not in context.executed_node_ids]:
            del self.outputs[key]
            self.all_prompts.pop(key, None)


class _Ready:
    """a lazy input whose producer already ran"""

    def __init__(self, v):
        self.value = v


def _n_outputs(cls):
    rt = getattr(cls, "RETURN_TYPES", None)
    if rt is not None:
        return len(rt)
    return getattr(cls, "N_OUTPUTS", 1)


def _plain(inputs):
    return {k: (list(v) if isinstance(v, NodeBindingParam) else (v if isinstance(v, (int, float, str, bool, type(None))) else id(v)))
            for k, v in inputs.items()}


def _fmt(v):
    if v is None or isinstance(v, (int, float, bool, str)):
        return v
    s = str(type(v).__name__)
    shape = getattr(v, "shape", None)
    return f"{s}{tuple(shape)}" if shape is not None else s


def run_workflow(path_or_workflow, engine_data=None, executor=None):
    """Load (if a path) + build_prompt + execute: the body of workflow.py:580-594 / DiffusionManager.SubmitPrompt
    (diffusionManager.py:262-330).  -> InferenceContext"""
    wf = path_or_workflow if isinstance(path_or_workflow, Workflow) else Workflow.Load(path_or_workflow)
    prompt, to_run, extra = wf.build_prompt()
    ex = executor or PromptExecutor.instance or PromptExecutor()
    return ex.execute(prompt, node_ids_to_be_ran=to_run, extra_data=extra, frame_data=engine_data)


__all__ = ["Workflow", "PromptExecutor", "PROMPT", "NodeBindingParam", "InferenceContext", "Lazy", "register_node",
           "get_node_cls_by_name", "NODE_CLASS_MAPPINGS", "run_workflow", "node_input_types"]
