"""Host side of the sampling stack (schedules + the per-step driver).  Mirrors, with the same names and argument
meaning: ``ModelSamplingDiscrete`` (comfyUI/comfy/model_sampling.py:75-150), ``calculate_sigmas_scheduler`` /
``KSampler`` (comfy/samplers.py:415-451, 937-1078), ``calc_cond_uncond_batch`` + ``sampling_function`` (batch =
[uncond | cond], CFG; samplers.py:176-358), ``BaseModel.apply_model`` with EPS scaling (comfy/model_base.py:93-127),
the k-diffusion loops ``sample_euler`` / ``sample_ddpm`` / ``sample_lcm`` (comfy/k_diffusion/sampling.py:129-149,
749-793) and ``custom_ksampler`` (comfyUI/nodes.py:1438-1495).  The sigma schedule is a few hundred scalars computed
once on the host; everything per step (UNet plan, CFG, sampler update, latent overlap) runs as HIP kernels.
"""
import math
import os

import torch

from . import ops as O

LATENT_SCALE = 0.18215            # comfy/latent_formats.py SD15.scale_factor
LATENT_SCALE_SDXL = 0.13025       # comfy/latent_formats.py SDXL.scale_factor


def latent_scale_of(unet_cfg):
    """latent_formats scale of a model family: checkpoints with vector conditioning (adm_in_channels) are the SDXL family
    (comfy/supported_models.py:153-175 SDXL / SDXLRefiner -> latent_formats.SDXL), everything else here is SD1.x / SD2.x"""
    return LATENT_SCALE_SDXL if unet_cfg.get("adm_in_channels") else LATENT_SCALE
SCHEDULER_NAMES = ["normal", "karras", "exponential", "sgm_uniform", "simple", "ddim_uniform"]
SAMPLER_NAMES = ["euler", "ddim", "ddpm", "lcm"]


# view shard: replay the cut plan segments as hipGraphs?  "1" / "0" decide; unset = eagerly for one call at a time, as graphs once
# calls are in flight (pipeline.InflightCalls turns it on): see DiffusionRunner._sharded_eval
_GRAPH_SEGMENTS = {"1": True, "0": False}.get(os.environ.get("SR_SHARD_GRAPH_SEGMENTS", ""))

class ModelSamplingDiscrete:
    def __init__(self, linear_start=0.00085, linear_end=0.012, timesteps=1000):
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        self.sigmas = (((1 - ac) / ac) ** 0.5).float()
        self.log_sigmas = self.sigmas.log()

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def timestep(self, sigma):
        sigma = torch.as_tensor(sigma, dtype=torch.float32)
        d = sigma.log().reshape(1, -1) - self.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape)

    def sigma(self, timestep):
        t = torch.clamp(torch.as_tensor(timestep).float(), min=0, max=len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()


def calculate_sigmas_scheduler(ms, scheduler_name, steps):
    if scheduler_name in ("normal", "sgm_uniform"):
        start, end = ms.timestep(ms.sigma_max), ms.timestep(ms.sigma_min)
        ts = torch.linspace(start, end, steps + 1)[:-1] if scheduler_name == "sgm_uniform" else torch.linspace(start, end, steps)
        return torch.FloatTensor([float(ms.sigma(t)) for t in ts] + [0.0])
    if scheduler_name == "simple":
        ss = len(ms.sigmas) / steps
        return torch.FloatTensor([float(ms.sigmas[-(1 + int(i * ss))]) for i in range(steps)] + [0.0])
    if scheduler_name == "ddim_uniform":
        ss = max(len(ms.sigmas) // steps, 1)
        sig, i = [], 1
        while i < len(ms.sigmas):
            sig.append(float(ms.sigmas[i]))
            i += ss
        return torch.FloatTensor(sig[::-1] + [0.0])
    if scheduler_name == "karras":
        rho = 7.0
        ramp = torch.linspace(0, 1, steps)
        a, b = float(ms.sigma_min) ** (1 / rho), float(ms.sigma_max) ** (1 / rho)
        s = (b + ramp * (a - b)) ** rho
        return torch.cat([s, s.new_zeros([1])])
    if scheduler_name == "exponential":
        s = torch.linspace(math.log(float(ms.sigma_max)), math.log(float(ms.sigma_min)), steps).exp()
        return torch.cat([s, s.new_zeros([1])])
    raise ValueError("error invalid scheduler " + str(scheduler_name))


def encode_adm_sdxl(pooled_output, width, height, crop_w=0, crop_h=0, target_width=None, target_height=None):
    """SDXL.encode_adm (comfy/model_base.py:352-369): y = [pooled text embedding | 256-wide sinusoidal embeddings of height,
    width, crop_h, crop_w, target_height, target_width]  -> (n, pooled + 1536)"""
    target_width = width if target_width is None else target_width
    target_height = height if target_height is None else target_height
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    out = []
    for v in (height, width, crop_h, crop_w, target_height, target_width):
        args = torch.Tensor([v])[:, None].float() * freqs[None]
        out.append(torch.cat([torch.cos(args), torch.sin(args)], dim=-1))
    flat = torch.flatten(torch.cat(out)).unsqueeze(dim=0).repeat(pooled_output.shape[0], 1)
    return torch.cat((pooled_output.to(flat.device).float(), flat), dim=1)


def _common_upscale_center(x, width, height):
    """comfy.utils.common_upscale(samples, width, height, 'nearest-exact', 'center') (comfy/utils.py:418-443): centre crop to the
    target aspect ratio, then nearest-exact resize.  x (N,C,H,W)"""
    ow, oh = x.shape[3], x.shape[2]
    old_aspect, new_aspect = ow / oh, width / height
    cx = cy = 0
    if old_aspect > new_aspect:
        cx = round((ow - ow * (new_aspect / old_aspect)) / 2)
    elif old_aspect < new_aspect:
        cy = round((oh - oh * (old_aspect / new_aspect)) / 2)
    return torch.nn.functional.interpolate(x[:, :, cy:oh - cy, cx:ow - cx], size=(height, width), mode="nearest-exact")


class SamplingCallbackContext:
    """comfyUI/types/runtime.py:543-593 (fields a corresponder reads)."""

    def __init__(self, noise, step_index, denoised, total_steps, timesteps, sigmas):
        self.noise, self.step_index, self.denoised = noise, step_index, denoised
        self.total_steps, self.timesteps, self.sigmas = total_steps, timesteps, sigmas

    @property
    def timestep(self):
        return self.timesteps[self.step_index]

    @property
    def sigma(self):
        return self.sigmas[self.step_index]


class KSampler:
    """Schedule holder (comfy/samplers.py:954-1003).  ``timesteps`` is NOT re-sliced when denoise < 1, exactly as in
    the reference (samplers.py:1000-1003 vs types/runtime.py:585-588)."""

    def __init__(self, steps, sampler="euler", scheduler="normal", denoise=None, model_sampling=None):
        self.ms = model_sampling or ModelSamplingDiscrete()
        self.sampler_name = sampler if sampler in SAMPLER_NAMES else SAMPLER_NAMES[0]
        self.scheduler = scheduler if scheduler in SCHEDULER_NAMES else SCHEDULER_NAMES[0]
        self.steps = steps
        if denoise is None or denoise > 0.9999:
            self.sigmas = calculate_sigmas_scheduler(self.ms, self.scheduler, steps)
            self.timesteps = [int(self.ms.timestep(s)) for s in self.sigmas]
        else:
            new_steps = int(steps / denoise)
            sig = calculate_sigmas_scheduler(self.ms, self.scheduler, new_steps)
            self.timesteps = [int(self.ms.timestep(s)) for s in sig]
            self.sigmas = sig[-(steps + 1):]


class DiffusionRunner:
    """One UNet (HIP plan) + sampler state for a fixed batch of N frames at latent size (h, w).

    ``sample()`` is the body of custom_ksampler -> comfy.sample.sample -> KSampler.sample -> sampler loop with the
    model call replaced by the native plan; the latent stays resident in HBM for the whole run."""

    def __init__(self, unet, N, h, w, cfg_scale, n_ctx=77, use_graph=True, shard=None, controlnets=None):
        """shard: optional parallel.ViewShard — this process then holds ``shard.n_local`` of the ``shard.n_views`` views of ONE
        overlapped group; N must equal shard.n_local."""
        self.shard = shard
        if shard is not None and shard.active:
            assert N == shard.n_local
        self.unet, self.N, self.h, self.w = unet, N, h, w
        # ControlNets (comfy/controlnet.py:180-214 get_control, chained through previous_controlnet: residuals are summed,
        # control_merge :60-100); their plans read the UNet plan's own x / t / ctx buffers
        self.controlnets = list(controlnets or [])
        self._hints = None
        self.cfg_scale = float(cfg_scale)
        self.copies = 1 if math.isclose(self.cfg_scale, 1.0) else 2     # samplers.py:335 (skip uncond at cfg 1)
        self.n_ctx = n_ctx
        self.use_graph = use_graph
        self.graph_segments = _GRAPH_SEGMENTS               # view shard: replay the cut plan segments as hipGraphs (see _sharded_eval)
        self.ms = ModelSamplingDiscrete()
        self.latent_scale = latent_scale_of(unet.cfg)   # process_latent_in / process_latent_out (samplers.py:905, :933)
        self._plan = None
        self._inject = "unset"
        dev = unet.device
        self.x = torch.zeros(N, 4, h, w, dtype=torch.float32, device=dev)
        self.den = torch.empty_like(self.x)
        self.d = torch.empty_like(self.x)
        self._stream = torch.cuda.Stream(device=dev) if use_graph else None
        self.time_comm, self._comm_events = False, []
        self._entries, self._general = None, None

    def _ensure_plan(self, inject_idx):
        """the plan depends only on HOW MANY frames are injected; which ones is a device tensor rewritten per run"""
        key = None if inject_idx is None else len(inject_idx)
        sharded = self.shard is not None and self.shard.active
        if self._plan is None or self._inject != key:
            control, inputs, cn = None, None, None
            if self.controlnets:
                control, inputs, cn = self._build_controls(self.N * self.copies, self.h, self.w, self.n_ctx)
            self._plan = self.unet.build(self.N * self.copies, self.h, self.w, inject_idx=inject_idx, n_ctx=self.n_ctx,
                                         inject_external=sharded and inject_idx is not None, control=control, inputs=inputs)
            self._plan["cn"] = cn
            self._hints_loaded = False
            self._inject = key
            self._captured = False
        if inject_idx is not None and not sharded:
            if any(int(i) < 0 or int(i) >= self.N * self.copies for i in inject_idx):
                raise IndexError(f"injected frame index out of range: {list(inject_idx)}")
            self._plan["inject"].copy_(torch.tensor([int(i) for i in inject_idx], dtype=torch.int32))
        self._inject_global = inject_idx
        return self._plan

    def _build_controls(self, B, h, w, n_ctx):
        """ControlNet plans on shared (x, t, ctx) input buffers + the plan that sums their residuals (control_merge)"""
        from .plan import PlanBuilder
        cfgu = self.unet.cfg
        pb = PlanBuilder(self.unet.device, self.unet.dtype)
        inputs = (pb.buf(B, cfgu["in_channels"], h, w, dtype=torch.float32, zero=True),
                  pb.buf(B, dtype=torch.float32, zero=True), pb.buf(B, n_ctx, cfgu["context_dim"], zero=True))
        cps = [c.build(B, h, w, *inputs, n_ctx=n_ctx) for c in self.controlnets]
        outs, mid = list(cps[0]["output"]), cps[0]["middle"]
        for cp in cps[1:]:                          # control_merge: element-wise sums of the residual lists
            merged = []
            for a, b in zip(outs, cp["output"]):
                if a is None or b is None:
                    merged.append(a if b is None else b)
                else:
                    y = pb.buf(*a.shape)
                    pb.add(a, b, y)
                    merged.append(y)
            outs = merged
            if mid is not None and cp["middle"] is not None:
                m2 = pb.buf(*mid.shape)
                pb.add(mid, cp["middle"], m2)
                mid = m2
            elif mid is None:
                mid = cp["middle"]
        return dict(output=outs, middle=mid), inputs, dict(plans=cps, merge=pb.take())

    def _run_controls(self, p, sigma):
        """the ControlNet encoders of one model call, then the plan that sums their residuals.  A net outside its sigma window
        (ControlNetApplyAdvanced's start / end percent -> ControlBase.timestep_range, comfy/controlnet.py:53-62) is not run:
        get_control then returns the previous nets' residuals alone (:184-189) -- here its residual buffers are zeroed once on
        leaving the window, so the merge adds nothing for it"""
        from . import conditioning as CD
        for net, cp in zip(self.controlnets, p["cn"]["plans"]):
            lo, hi = getattr(net, "timestep_percent_range", (0.0, 1.0))
            active = True
            if (lo, hi) != (0.0, 1.0):
                active = not (sigma > CD.percent_to_sigma(self.ms, lo) or sigma < CD.percent_to_sigma(self.ms, hi))
            if active:
                cp["step"].run()
                cp["_zeroed"] = False
            elif not cp.get("_zeroed"):
                for t in list(cp["output"]) + [cp["middle"]]:
                    if t is not None:
                        t.zero_()
                cp["_zeroed"] = True
        p["cn"]["merge"].run()

    def set_control_hints(self, hints):
        """hints: one (N,3,8h,8w) tensor in [0,1] per ControlNet (ControlNetApply's image.movedim(-1,1), nodes.py:745-760);
        the same hint serves the cond and uncond halves of the batch"""
        if len(hints) != len(self.controlnets):
            raise ValueError(f"{len(self.controlnets)} ControlNets need as many hints, got {len(hints)}")
        self._hints = list(hints)
        self._hints_loaded = False

    def set_vector_conditioning(self, y_positive, y_negative=None):
        """SDXL-family ``y`` (pooled text + size / crop embeddings, (1 | N, adm_in_channels)); the negative half defaults to the
        positive one (comfy/model_base.py SDXL.encode_adm builds both from their own pooled outputs)"""
        self._y = (y_positive, y_positive if y_negative is None else y_negative)

    def set_conditioning(self, positive, negative):
        """positive / negative: (1 | N, n_ctx, ctx_dim) text embeddings (CONDRegular.process_cond repeats a single
        embedding to the batch, comfy/conds.py:23-26).  Batch order = [uncond frames..., cond frames...]."""
        self._pos, self._neg = positive, negative
        self._entries = None

    def set_cond_entries(self, positive, negative):
        """positive / negative: lists of conditioning entries (conditioning.entries_of): several prompts with masks, strengths
        and areas, composed as calc_cond_uncond_batch does (comfy/samplers.py:176-320).  One plain entry each is the ordinary
        [uncond | cond] batch."""
        from . import conditioning as CD
        CD.check_supported(positive)
        CD.check_supported(negative)
        if CD.is_plain(positive) and CD.is_plain(negative):
            self._entries = None
            return self.set_conditioning(positive[0]["cond"], negative[0]["cond"])
        self._entries = (list(positive), list(negative))
        self._general = None

    def _build_general(self, n_rand):
        """group the entries into model calls and build one UNet plan per call shape (conditioning.groups_of)"""
        from . import conditioning as CD
        N, h, w = self.N, self.h, self.w
        C = self.unet.cfg["in_channels"]
        pos, neg = CD.prepare(self._entries[0], self._entries[1], h, w, self.ms)
        sharded = self.shard is not None and self.shard.active
        if sharded:
            # a mask batch that follows the views of the WHOLE group (one mask per view) is cut to this rank's views; a single
            # mask serves every view as it does unsharded (get_area_and_mult repeats the mask batch over the latent batch)
            def own_views(e):
                m = e.get("mask")
                if m is not None and m.shape[0] == self.shard.n_views and self.shard.n_views != N:
                    e = dict(e)
                    e["mask"] = m[self.shard.slice]
                elif m is not None and m.shape[0] not in (1, N):
                    raise ValueError(f"a mask batch of {m.shape[0]} fits neither the group ({self.shard.n_views} views) nor this rank ({N})")
                return e
            pos, neg = [own_views(e) for e in pos], [own_views(e) for e in neg]
        # entries with a sigma window (ConditioningSetTimestepRange) are skipped outside it (get_area_and_mult returns None,
        # samplers.py:60-67), so the list of model calls depends on the step: one VARIANT (groups + plans) per distinct set of
        # active entries, all of them built before the first step of a run (_general_plans)
        self._general = dict(pos=pos, neg=neg, variants={}, groups=None, built_for=None,
                             out_c=torch.empty_like(self.x), cnt_c=torch.empty_like(self.x),
                             out_u=torch.empty_like(self.x), cnt_u=torch.empty_like(self.x))
        self._general_select(float(self.ms.sigma_max))        # (G["groups"]: the model calls at the top of the schedule)
        return self._general

    def _general_key(self, sigma):
        from . import conditioning as CD
        G = self._general
        return (tuple(CD.entry_active(e, sigma) for e in G["pos"]), tuple(CD.entry_active(e, sigma) for e in G["neg"]))

    def _general_select(self, sigma):
        """-> the groups (model calls) of the entries active at this sigma; G["groups"] points at them"""
        from . import conditioning as CD
        G = self._general
        key = self._general_key(sigma)
        if key not in G["variants"]:
            N, h, w = self.N, self.h, self.w
            C = self.unet.cfg["in_channels"]
            pos = [e for e, on in zip(G["pos"], key[0]) if on]
            neg = [e for e, on in zip(G["neg"], key[1]) if on]
            groups = CD.groups_of(pos, neg, N, C, h, w, use_uncond=self.copies == 2)
            dev = self.x.device
            for g in groups:
                g["chunks"] = len(g["members"])
                g["mult"] = torch.stack([m for _, _, m in g["members"]]).to(dev).contiguous()        # (chunks,N,C,ah,aw)
                g["kinds"] = torch.tensor([k for k, _, _ in g["members"]], dtype=torch.int32, device=dev)
                g["n_ctx"] = int(g["members"][0][1]["cond"].shape[1])
            G["variants"][key] = groups
        G["groups"] = G["variants"][key]
        return G["groups"]

    def _general_plans(self, inject, sigmas=None):
        """build (once per injected-frame COUNT) and load the plans of the general path: of every variant the run's sigmas select"""
        G = self._general
        sharded = self.shard is not None and self.shard.active
        key = None if inject is None else len(inject)
        first = G["groups"]
        for sg in (sigmas if sigmas is not None else []):
            self._general_select(float(sg))
        if first is not None:
            G["groups"] = first
        all_groups = [g for v in G["variants"].values() for g in v]
        if G["built_for"] != ("built", key) or any("plan" not in g for g in all_groups):
            for g in all_groups:
                if "plan" in g and G["built_for"] == ("built", key):
                    continue
                ah, aw, _, _ = g["area"]
                B = self.N * g["chunks"]
                control, inputs, cn = None, None, None
                if self.controlnets:
                    control, inputs, cn = self._build_controls(B, ah, aw, g["n_ctx"])
                g["plan"] = self.unet.build(B, ah, aw, inject_idx=inject, n_ctx=g["n_ctx"], control=control, inputs=inputs,
                                            inject_external=sharded and inject is not None)
                g["plan"]["cn"] = cn
            G["built_for"] = ("built", key)
        self._inject_global = inject
        for g in all_groups:
            p, N = g["plan"], self.N
            if inject is not None:
                B = (self.shard.n_views if sharded else N) * g["chunks"]      # the indices address the WHOLE group's batch of this call
                if any(int(i) < 0 or int(i) >= B for i in inject):
                    raise IndexError(f"injected frame index out of range: {list(inject)}")
                if not sharded:
                    p["inject"].copy_(torch.tensor([int(i) for i in inject], dtype=torch.int32))
            dt = p["ctx"].dtype
            for j, (_, e, _) in enumerate(g["members"]):
                p["ctx"][j * N:(j + 1) * N].copy_(e["cond"].to(p["ctx"].device).to(dt).expand(N, -1, -1))
            if p.get("y") is not None:
                from .sampling import encode_adm_sdxl as _adm
                adm = self.unet.cfg["adm_in_channels"]
                for j, (_, e, _) in enumerate(g["members"]):
                    if e.get("pooled_output") is None:
                        raise ValueError("this model takes vector conditioning: a conditioning entry carries no 'pooled_output'")
                    y = _adm(e["pooled_output"], width=self.w * 8, height=self.h * 8)
                    p["y"][j * N:(j + 1) * N, :adm].copy_(y.to(p["y"].device).to(p["y"].dtype).expand(N, -1))
            p["prologue"].run()
            if p.get("cn") is not None:
                if self._hints is None:
                    raise ValueError("ControlNets are attached: call set_control_hints() before sampling")
                for cp, hint in zip(p["cn"]["plans"], self._hints):
                    hv = hint.to(cp["hint"].device, torch.float32)
                    th, tw = cp["hint"].shape[2], cp["hint"].shape[3]
                    if (hv.shape[2], hv.shape[3]) != (th, tw):
                        # a model call on a conditioning AREA: the control net gets the cropped latent, and ControlNet.get_control
                        # resizes the WHOLE hint to 8x that crop (common_upscale(cond_hint_original, w*8, h*8, 'nearest-exact',
                        # 'center'), comfy/controlnet.py:193-201, comfy/utils.py:418-443) -- it does not cut the window out of it.
                        # Done once per sampling run, as the reference caches it per size.
                        hv = _common_upscale_center(hv, tw, th)
                    for j in range(g["chunks"]):
                        cp["hint"][j * N:(j + 1) * N].copy_(hv.expand(N, -1, -1, -1) if hv.shape[0] == 1 else hv)
                    cp["prologue"].run()
        return G

    def _general_denoise(self, sigma, timestep_index, want_d):
        """one sampling_function call (samplers.py:323-358) over the prepared groups -> self.den (and self.d)"""
        G = self._general
        self._general_select(sigma)
        G["out_c"].zero_()
        G["out_u"].zero_()
        G["cnt_c"].fill_(1e-37)
        G["cnt_u"].fill_(1e-37)
        for g in G["groups"]:
            p = g["plan"]
            O.cond_crop_scale(self.x, p["x"], g["area"], g["chunks"], sigma)
            p["t"].fill_(float(timestep_index))
            if p.get("cn") is not None:
                self._run_controls(p, sigma)
            if p["schedule"]:                                # view-sharded group: segments + K/V-source broadcasts
                self._sharded_eval(p, chunks=g["chunks"])
            else:
                p["step"].run()
            O.cond_accumulate(self.x, p["out"], g["mult"], g["kinds"], G["out_c"], G["cnt_c"], G["out_u"], G["cnt_u"], g["area"],
                              g["chunks"], sigma)
        O.cfg_combine(self.x, G["out_c"], G["cnt_c"], G["out_u"], G["cnt_u"], self.den, self.d if want_d else None, sigma,
                      self.cfg_scale)

    def _load_ctx(self, p):
        N = self.N
        dt = p["ctx"].dtype
        pos = self._pos.to(p["ctx"].device).to(dt).expand(N, -1, -1)
        if self.copies == 2:
            neg = self._neg.to(p["ctx"].device).to(dt).expand(N, -1, -1)
            p["ctx"][:N].copy_(neg)
            p["ctx"][N:].copy_(pos)
        else:
            p["ctx"].copy_(pos)
        if p.get("y") is not None:
            if getattr(self, "_y", None) is None:
                raise ValueError("this UNet takes vector conditioning (adm_in_channels): call set_vector_conditioning() before sampling")
            yb, adm = p["y"], self.unet.cfg["adm_in_channels"]
            yp, yn = (t.to(yb.device).to(yb.dtype).expand(N, -1) for t in self._y)
            if self.copies == 2:
                yb[:N, :adm].copy_(yn)
                yb[N:, :adm].copy_(yp)
            else:
                yb[:, :adm].copy_(yp)
        p["prologue"].run()
        if p.get("cn") is not None:
            if self._hints is None:
                raise ValueError("ControlNets are attached: call set_control_hints() before sampling")
            for cp, hint in zip(p["cn"]["plans"], self._hints):
                hb = cp["hint"]
                hv = hint.to(hb.device, torch.float32)
                for c in range(self.copies):
                    hb[c * N:(c + 1) * N].copy_(hv.expand(N, -1, -1, -1) if hv.shape[0] == 1 else hv)
                cp["prologue"].run()                        # hint encoder: once per run

    def model_eps(self, p, sigma, timestep_index):
        """calc_cond_uncond_batch + apply_model: xin = x/sqrt(sigma^2+1) for both chunks, t = argmin|log sigma|"""
        n = self.x.numel()
        O.eps_scale_input(self.x, p["x"], self.copies, sigma)
        p["t"].fill_(float(timestep_index))
        if p.get("cn") is not None:                         # ControlNet encoders on the same (x, t, ctx), then their merge; no
            self._run_controls(p, sigma)                    # cross-view work inside them (controlnet.py:205-213 passes no corresponder)
        if p["schedule"]:
            return self._sharded_eval(p)
        if self.use_graph:
            if not self._captured:
                torch.cuda.current_stream().synchronize()
                p["step"].capture(self._stream)
                self._captured = True
                self._stream.synchronize()
            p["step"].launch()
        else:
            p["step"].run()
        return p["out"]

    def _sharded_eval(self, p, chunks=None):
        """view-sharded group: the injected frame's post-LayerNorm tokens live on ONE rank.  The step plan is cut twice per
        transformer block (BlockLowering.schedule): after norm1 the owner's rows go out over xGMI as an asynchronous RCCL
        broadcast while every rank computes its own Q projection; the compute stream only waits for the rows before the K / V
        projections.  Segments are replayed as hipGraphs when the runner uses graphs.  ``comm_ms`` (when
        ``time_comm``) accumulates the time the compute stream spent stalled in those waits = the EXPOSED communication."""
        from . import parallel as PAR
        sched = p["schedule"]
        # One call at a time, the 33 segments of an evaluation are launched eagerly (sr_plan_run): a hipGraphLaunch per segment costs
        # more start-up latency than the ~12 kernels of a segment save in launch gaps -- measured as one rank of a shard (world-1
        # RCCL group, same box): 145.0 -> 138.6 ms per call at one view per rank, 187.3 -> 179.7 at two.  With calls in flight it
        # is the other way round: three host threads launching 6 800 kernels per call each are what bounds the rank (the B = 2
        # evaluations of different calls do overlap on the GPU: 5.96 -> 3.99 ms per evaluation side by side), and a graph launch
        # per segment takes the host out of the way: 110.5 -> 97.7 ms per call at one view per rank with three in flight, 153.2 ->
        # 138.7 at two.  graph_segments: None = this rule (InflightCalls sets True), True / False = SR_SHARD_GRAPH_SEGMENTS.
        use_graph = self.use_graph and bool(self.graph_segments)
        if use_graph and not p.get("_segments_captured"):            # (per plan: conditioning lists run several plans per step)
            torch.cuda.current_stream().synchronize()
            for kind, *rest in sched:                       # all captures up front: none while a collective is in flight
                if kind == "run" and rest[0].n > 0:
                    rest[0].capture(self._stream)
            self._stream.synchronize()
            p["_segments_captured"] = True
            self._captured = True
        pending = []
        for kind, *rest in sched:
            if kind == "run":
                rest[0].launch() if use_graph else rest[0].run()
            elif kind == "bcast":
                for ln, src in rest[0]:                      # the rows and, with the folded LayerNorm, their statistics
                    for j, g in enumerate(self._inject_global):
                        owner, li = self.shard.owner_of(g, chunks)
                        if owner == self.shard.rank:
                            src[j].copy_(ln[li])
                        pending.append(PAR.broadcast_start(src[j], owner, self.shard.group))
            else:                                            # "wait"
                # only a transfer that really runs beside the compute stream (RCCL) has an exposed part to time; the staged
                # gloo form completed inside broadcast_start, two events around a no-op would measure nothing
                timed = self.time_comm and any(h.is_async for h in pending)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                for h in pending:
                    h.wait()
                if timed:
                    e1.record()
                    self._comm_events.append((e0, e1))
                pending = []
        return p["out"]

    def exposed_comm_ms(self):
        """sum of the compute-stream stalls in the K/V-source waits recorded since the last call (needs time_comm = True);
        None when no asynchronous transfer was waited for (gloo staging, one rank); synchronises"""
        torch.cuda.synchronize()
        if not self._comm_events:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self._comm_events)
        self._comm_events = []
        return ms

    def sample(self, noise, steps, sampler_name, scheduler, denoise=1.0, latent_image=None, seed=None,
               inject_n_rand=None, step_callback=None, noise_fn=None, rng_turn=None, pre_step_callback=None):
        """-> samples (N,4,h,w) fp32 on device (already divided by the latent scale, samplers.py:933).

        RNG draw order on the *global CPU generator* replicates the reference: custom_ksampler's seed draw
        (nodes.py:1455), SAMPLER_METHOD's reseed+draw for "ddim" (samplers.py:766-768), pre_atten_inject's randint
        on the first attention block (corresponder.py:204-205), then the sampler's per-step randn_like.

        pre_step_callback(x, i, timestep): called before the model evaluation of step i with the step's input latent (the
        evaluation only reads it) -- the view-sharded pipeline starts the latent all-gather of the step's overlap here."""
        ks = KSampler(steps, sampler_name, scheduler, denoise, self.ms)
        sig = ks.sigmas
        sampler = ks.sampler_name
        dev = self.x.device
        # rng_turn: context manager that admits concurrent callers in call order (pipeline.CallOrder) around the draws on the
        # process-wide CPU generator, so calls in flight on several streams draw exactly what the sequential loop draws
        import contextlib
        with (rng_turn if rng_turn is not None else contextlib.nullcontext()):
            if seed is None:
                seed = int(torch.randint(0, 2 ** 32, (1,)).item())
            if sampler == "ddim":
                g = torch.manual_seed(seed + 1)
                n_grp = noise.shape[0] if self.shard is None else self.shard.n_views     # the draw is over the WHOLE group
                torch.randn((n_grp,) + tuple(noise.shape[1:]), generator=g, device="cpu")
            general = getattr(self, "_entries", None) is not None
            G = None
            if general:
                n_rand_ = inject_n_rand if (inject_n_rand is not None and inject_n_rand >= 0) else None
                G = self._general if self._general is not None else self._build_general(n_rand_)
                # the model calls of the first step that HAS active entries: pre_atten_inject draws its indices on the run's first
                # UNet call, from that call's batch (corresponder.py:204-205)
                first_groups = next((gr for gr in (self._general_select(float(v)) for v in sig[:-1]) if gr), None)
                if first_groups is None:
                    raise ValueError("no conditioning entry is active at any step of the schedule")
                G["groups"] = first_groups
            inject = None
            if inject_n_rand is not None and inject_n_rand >= 0:
                n_all = self.N if self.shard is None else self.shard.n_views
                B = n_all * (G["groups"][0]["chunks"] if general else self.copies)
                inject = torch.randint(1, B, (inject_n_rand,)).tolist()       # global RNG; B counts cond + uncond entries
                if self.shard is not None and self.shard.active:              # every rank must use rank 0's draw
                    from . import parallel as PAR
                    t_inj = torch.tensor(inject, dtype=torch.int64)
                    PAR.broadcast(t_inj, 0, self.shard.group)
                    inject = t_inj.tolist()
            # ddpm / lcm draw one noise tensor per step from the global generator (default_noise_sampler).  With calls in flight
            # (rng_turn) those draws would interleave with other calls' draws: take them all here, inside this call's turn, in
            # the order the loop below would have made them -- the generator then sees exactly the sequential loop's sequence
            predrawn = None
            if rng_turn is not None and noise_fn is None and sampler in ("ddpm", "lcm"):
                predrawn = [torch.randn(tuple(self.x.shape), dtype=torch.float32) for i in range(len(sig) - 1) if float(sig[i + 1]) > 0]
        latent = torch.zeros_like(noise) if latent_image is None else latent_image * self.latent_scale
        max_denoise = math.isclose(float(self.ms.sigma_max), float(sig[0]), rel_tol=1e-05) or float(sig[0]) > float(self.ms.sigma_max)
        s0 = float(torch.sqrt(1.0 + sig[0] ** 2.0)) if max_denoise else float(sig[0])
        self.x.copy_(noise.to(dev, torch.float32))
        O.axpby(self.x, latent.to(dev, torch.float32).contiguous(), 1.0, s0)        # x = noise*s0 + latent
        if general:
            self._general_plans(inject, [float(v) for v in sig[:-1]])
            p = G["groups"][0]["plan"]
        else:
            p = self._ensure_plan(inject)
            self._load_ctx(p)
        if noise_fn is None and predrawn is not None:
            def noise_fn():
                return predrawn.pop(0).to(dev)
        elif noise_fn is None:
            def noise_fn():
                return torch.randn(tuple(self.x.shape), dtype=torch.float32).to(dev)    # default_noise_sampler (CPU x)
        t_index = [int(t) for t in self.ms.timestep(sig[:-1])]          # ModelSamplingDiscrete.timestep, once per run
        for i in range(len(sig) - 1):
            s, sn = float(sig[i]), float(sig[i + 1])
            if pre_step_callback is not None:
                pre_step_callback(self.x, i, ks.timesteps[i])
            if general:
                self._general_denoise(s, t_index[i], sampler in ("euler", "ddim"))
            else:
                eps = self.model_eps(p, s, t_index[i])
                O.cfg_denoise(self.x, eps, self.den, self.d if sampler in ("euler", "ddim") else None, self.copies, s, self.cfg_scale)
            if step_callback is not None:
                step_callback(SamplingCallbackContext(self.x, i, self.den, len(sig) - 1, ks.timesteps, sig.tolist()))
            if sampler in ("euler", "ddim"):
                O.euler_step(self.x, self.d, sn - s)
            elif sampler == "ddpm":
                O.ddpm_step(self.x, self.den, noise_fn() if sn > 0 else None, s, sn)
            elif sampler == "lcm":
                O.lcm_step(self.x, self.den, noise_fn() if sn > 0 else None, sn)
            else:
                raise ValueError(sampler)
        out = torch.empty_like(self.x)
        out.zero_()
        O.axpby(out, self.x, 1.0 / self.latent_scale, 0.0)
        run_plans = [g["plan"] for v in G["variants"].values() for g in v] if general else [p]
        flags = [q["inject_err"] for q in run_plans if q.get("inject_err") is not None]
        if flags and int(torch.stack([f.reshape(()) for f in flags]).max().item()) != 0:      # the run's one host sync (results are due anyway)
            for f in flags:
                f.zero_()
            raise IndexError("injected frame index outside the batch reached sr_gather_rows (k_context[idx], corresponder.py:207-214)")
        return out, inject
