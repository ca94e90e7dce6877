"""The render-then-diffuse frame loop, end to end on one GPU:

  rasterise N views (sr_raster_draw)                       RenderManager.on_frame_run           renderManager.py:950-1043
  -> EngineData (ids, masks, pooled + AdaIN'd noise)       RenderManager._save_frame_data       renderManager.py:877-948
  -> CorrespondSampler (KSampler loop, UNet plan, CFG,     _nodes/samplers.py:128-201, nodes.py:1438-1495,
     per-step latent overlap, K/V injection)               comfy/samplers.py:176-358, corresponder.py:188-376
  -> VAE decode                                            comfy/sd.py:329-346
  -> corresponder.finished -> CorrespondMap.update         corresponder.py:130-155, corrmap.py:578-736

This is the body of the reference's bake call (DiffusionManager.SubmitPrompt with a bake workflow,
diffusionManager.py:289-352) for a scene of the ``scripts/bake_ball.py`` kind; all stages stay in HBM."""
import contextlib
import copy
import threading

import torch

from . import ops as O
from . import scene as S
from .corrmap import CorrespondMap, IDMap
from .corresponder import DefaultCorresponder, OverlapCorresponder
from .sampling import DiffusionRunner
from .types import EngineData, LATENT


class BakeBallScene:
    """scripts/bake_ball.py:19-62: camera (0,0.68,2.3) -> (0,0.68,0), sphere scale 0.70 with a diffuse texture, corr-map
    proxy sphere scale 0.85 (k=6, texcoord ids, baking mode, TRANSPARENT queue), 1 degree of Y rotation per frame."""

    def __init__(self, W=512, H=512, k=6, seed=0, device="cuda"):
        self.W, self.H, self.k = W, H, k
        self.camera = S.Camera((0, 0.68, 2.3), (0, 0.68, 0), fov=45.0, near=0.1, far=100.0)
        self.sphere = S.Mesh.Sphere(32)
        g = torch.Generator().manual_seed(seed)
        self.noise_tex = torch.randn(512, 512, 4, generator=g).half().to(device)        # Texture.CreateNoiseTex (texture.py:507-568)
        self.diffuse = torch.rand(64, 64, 4, generator=g).to(device)
        self.diffuse[..., 3] = 1.0
        self.corrmap = CorrespondMap(k=k, height=H, width=W, device=device)
        self.sprite, self.material = 2, 2

    def tasks(self, frame):
        rot = S.rotate_y(float(frame))
        at = S.translate((0, 0.68, 0))
        m1 = S.matmul(at, S.matmul(rot, S.scale(0.70)))
        m2 = S.matmul(at, S.matmul(rot, S.scale(0.85)))
        return [
            S.DrawTask(self.sphere, m1, sprite_id=1, material_id=1, render_mode=0, diffuse_tex=self.diffuse, order=999.7),
            S.DrawTask(self.sphere, m2, sprite_id=self.sprite, material_id=self.material, render_mode=2, corrmap_k=self.k,
                       use_texcoord_id=True, id_size=(self.W, self.H), noise_tex=self.noise_tex, order=2000.3),
        ]


class BoatScene:
    """BASELINE config 3 (scripts/boat_example.py:84-103): a mesh loaded with ``Mesh.Load`` at the origin turning about Y
    (AutoRotation; the headless driver fixes the pose per frame: ``deg_per_frame``), camera (0,3,-3) looking at the origin,
    diffuse texture on the DefaultOpaqueMaterial.  The views are tied together the way the reference's bake scripts do it
    (scripts/bake_ball.py:36-55, bake_example.py:52-70): a corr-map proxy sphere around the object -- k x k maps, texcoord
    ids, BAKING mode, TRANSPARENT queue -- turning with it, so that the id / noise planes are anchored to the surface."""

    def __init__(self, obj_path, W=512, H=512, k=6, seed=0, device="cuda", deg_per_frame=2.5, proxy_scale=1.5):
        self.W, self.H, self.k = W, H, k
        self.camera = S.Camera((0, 3, -3), (0, 0, 0), fov=45.0, near=0.1, far=100.0)
        self.mesh = S.Mesh.Load(obj_path)
        self.sphere = S.Mesh.Sphere(32)
        g = torch.Generator().manual_seed(seed)
        self.noise_tex = torch.randn(512, 512, 4, generator=g).half().to(device)
        self.diffuse = torch.rand(64, 64, 4, generator=g).to(device)
        self.diffuse[..., 3] = 1.0
        self.corrmap = CorrespondMap(k=k, height=H, width=W, device=device)
        self.sprite, self.material = 2, 2
        self.deg_per_frame, self.proxy_scale = deg_per_frame, proxy_scale

    def tasks(self, frame):
        rot = S.rotate_y(self.deg_per_frame * float(frame))
        return [
            S.DrawTask(self.mesh, rot, sprite_id=1, material_id=1, render_mode=0, diffuse_tex=self.diffuse, order=999.8),
            S.DrawTask(self.sphere, S.matmul(rot, S.scale(self.proxy_scale)), sprite_id=self.sprite, material_id=self.material,
                       render_mode=2, corrmap_k=self.k, use_texcoord_id=True, id_size=(self.W, self.H), noise_tex=self.noise_tex,
                       order=2000.2),
        ]


class MultiObjScene:
    """BASELINE config 5's scene (scripts/multi_obj_example.py:24-52): a loaded mesh at the origin turning about Y
    (AutoRotation 4 deg x DeltaTime; the headless driver fixes the pose per frame), a textured sphere at (-1.5, 0.5, 1.0)
    scale 0.5 and a textured plane of scale 5, CameraController start pose (4, 3.5, 4) -> (0, 0.4, 0).  The shipped script only
    rasterises (``disableComfyUI=True``); the bake composition around it is the one of the reference's bake scripts
    (scripts/bake_ball.py:36-55): a corr-map proxy sphere about the turning object -- k x k maps, texcoord ids, BAKING mode,
    TRANSPARENT queue -- which is what ties the views of one call together."""

    def __init__(self, obj_path, W=512, H=512, k=6, seed=0, device="cuda", deg_per_frame=4.0, proxy_scale=1.6):
        self.W, self.H, self.k = W, H, k
        self.camera = S.Camera((4.0, 3.5, 4.0), (0, 0.4, 0), fov=45.0, near=0.1, far=100.0)
        self.mesh = S.Mesh.Load(obj_path)
        self.sphere, self.plane = S.Mesh.Sphere(32), S.Mesh.Plane()
        g = torch.Generator().manual_seed(seed)
        self.noise_tex = torch.randn(512, 512, 4, generator=g).half().to(device)
        self.diffuse = torch.rand(64, 64, 4, generator=g).to(device)
        self.diffuse[..., 3] = 1.0
        checker = ((torch.arange(64)[:, None] // 8 + torch.arange(64)[None, :] // 8) % 2).float()
        self.debug_tex = torch.stack([checker, 1 - checker, checker * 0 + 0.5, checker * 0 + 1.0], -1).contiguous().to(device)
        self.corrmap = CorrespondMap(k=k, height=H, width=W, device=device)
        self.sprite, self.material = 4, 4
        self.deg_per_frame, self.proxy_scale = deg_per_frame, proxy_scale

    def tasks(self, frame):
        rot = S.rotate_y(self.deg_per_frame * float(frame))
        ball = S.matmul(S.translate((-1.5, 0.5, 1.0)), S.scale(0.5))
        return [
            S.DrawTask(self.mesh, rot, sprite_id=1, material_id=1, render_mode=0, diffuse_tex=self.diffuse, order=999.80),
            S.DrawTask(self.sphere, ball, sprite_id=2, material_id=2, render_mode=0, diffuse_tex=self.debug_tex, order=999.85),
            S.DrawTask(self.plane, S.scale(5.0), sprite_id=3, material_id=3, render_mode=0, diffuse_tex=self.debug_tex, order=999.83),
            S.DrawTask(self.sphere, S.matmul(rot, S.scale(self.proxy_scale)), sprite_id=self.sprite, material_id=self.material,
                       render_mode=2, corrmap_k=self.k, use_texcoord_id=True, id_size=(self.W, self.H), noise_tex=self.noise_tex,
                       order=2000.2),
        ]


class CallOrder:
    """Tickets for the two sections of a call that touch process-wide state — the draws on the global CPU generator at the
    start of sampling and the frame-ordered ('first' priority) corr-map update — so that calls in flight on several streams
    execute them in call order and produce exactly what the sequential loop produces."""

    def __init__(self, first=0):
        self._cv = threading.Condition()
        self._next = {"rng": first, "bake": first}
        self._failed = False

    @contextlib.contextmanager
    def turn(self, kind, index):
        with self._cv:
            self._cv.wait_for(lambda: self._next[kind] == index or self._failed)
            if self._failed:
                raise RuntimeError("another in-flight call failed")
        try:
            yield
        except BaseException:
            with self._cv:
                self._failed = True
                self._cv.notify_all()
            raise
        with self._cv:
            self._next[kind] = index + 1
            self._cv.notify_all()

    def fail(self):
        with self._cv:
            self._failed = True
            self._cv.notify_all()


class InflightCalls:
    """K bake calls in flight on one GPU: one host thread + HIP stream + FramePipeline (own launch plans, hipGraph, G-buffer
    and split-K scratch; shared weights, scene and corr-map) per slot.  A single call leaves CUs idle in every kernel's tail
    wave and in the launch gaps of ~600 kernels per UNet evaluation; a second call's kernels fill them.  Sampling never reads
    the corr-map (BAKING mode rasterises ids, not colours), so only the RNG draws and the corr-map updates are ordered
    (CallOrder); frames are numbered as the sequential loop numbers them."""

    def __init__(self, pipe, inflight=2):
        self.pipes = [pipe] + [pipe.spawn(i) for i in range(1, inflight)]
        self.streams = [torch.cuda.Stream(device=pipe.unet.device) for _ in self.pipes]
        if len(self.pipes) > 1:
            for p in self.pipes:                             # view shard: with calls in flight the host threads are the bound --
                if p.shard is not None and p.runner.graph_segments is None:     # replay the plan segments as hipGraphs
                    p.runner.graph_segments = True           # (DiffusionRunner._sharded_eval; SR_SHARD_GRAPH_SEGMENTS overrides)

    def warm(self, calls=1):
        """build plans / tune tiles / capture graphs one pipeline at a time (timing-based tuning and graph capture want the GPU
        to themselves); every pipeline ends on its own stream"""
        for i, (p, st) in enumerate(zip(self.pipes, self.streams)):
            with torch.cuda.stream(st), O.workspace_slot(i):
                for _ in range(calls):
                    p.call()
            st.synchronize()
        self._sync_frames()

    def _sync_frames(self):
        f = max(p.frame0 for p in self.pipes)
        for p in self.pipes:
            p.frame0 = f

    def run(self, n_calls):
        """n_calls calls, call c handled by slot c % K -> list of per-call outputs is not kept (the corr-map is the product);
        returns after every stream has drained"""
        base = self.pipes[0].frame0
        order = CallOrder()
        errs = []
        torch.cuda.synchronize()

        def work(i):
            p, st = self.pipes[i], self.streams[i]
            try:
                torch.cuda.set_device(st.device)            # the current device is per thread (rank r works on GPU r)
                with torch.cuda.stream(st), O.workspace_slot(i):
                    for c in range(i, n_calls, len(self.pipes)):
                        p.frame0 = base + c * p.N_all
                        p.call(order=(order, c))
                    st.synchronize()
            except BaseException as e:                      # noqa: BLE001 - re-raised on the caller's thread
                errs.append(e)
                order.fail()
        ts = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(self.pipes))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        for p in self.pipes:
            p.frame0 = base + n_calls * p.N_all


class FramePipeline:
    def __init__(self, unet, vae, scene, n_views=8, steps=20, cfg=8.0, sampler="ddim", scheduler="normal",
                 corresponder=None, use_graph=True, bg_seed=1, shard=None, controls=None, keep_planes=False):
        """shard: optional parallel.ViewShard over the ``n_views`` of ONE overlapped group (one process per GPU): this process
        then rasterises / diffuses / decodes only its own views; id maps are all-gathered once per call, latents once per
        overlap step, the injected frame's tokens are broadcast per transformer block and decoded frames go to rank 0 for
        the ordered corr-map update.  Without it the pipeline is a self-contained replica."""
        self.unet, self.vae, self.scene = unet, vae, scene
        # controls: [("depth" | "normal" | "color" | "canny", ControlNet)] -- ControlNets driven by the G-buffer planes of the
        # same views, as the reference's miku-control workflow (depth + normalbae; EngineData.depth_maps / normal_maps ->
        # ControlNetApply)
        self.controls = list(controls or [])
        self.shard = shard if (shard is not None and shard.active) else None
        # ControlNets inside a view-sharded group (BASELINE config 4): every rank runs the encoders on its OWN views' hints -- the
        # reference calls the control model without a corresponder (controlnet.py:205-213), so there is no cross-view exchange in it
        self.N_all = n_views
        n_views = n_views if self.shard is None else self.shard.n_local
        self.N, self.steps, self.cfg, self.sampler, self.scheduler = n_views, steps, cfg, sampler, scheduler
        dev = unet.device
        self.W, self.H = scene.W, scene.H
        self.h, self.w = self.H // 8, self.W // 8
        self.gbuf = S.GBuffer(self.W, self.H, device=dev)
        self.corresponder = corresponder if corresponder is not None else OverlapCorresponder(
            step_finished_inject_ratio=0.5, step_finished_stop_inject_timestep=500, update_corrmap_mode="first")
        # the reference's OverlapCorresponder has no finished(); the bake (corr-map update) is DefaultCorresponder's.
        # ignore_obj_mat_id_when_update=True is the reference option that avoids _update's double-gather IndexError on
        # partially covered frames (corrmap.py:703/710; reproduced in corrmap.py of this package)
        self.baker = DefaultCorresponder(update_corrmap_mode="first", ignore_obj_mat_id_when_update=True)
        self.runner = DiffusionRunner(unet, n_views, self.h, self.w, cfg, use_graph=use_graph, shard=self.shard,
                                      controlnets=[c for _, c in self.controls])
        self.vplan = vae.build(n_views, self.h, self.w)
        g = torch.Generator().manual_seed(bg_seed)
        self.bg_noise = torch.randn(1, self.H, self.W, 4, generator=g).to(dev)                # RenderManager.GlobalBGNoise
        # per-call EngineData accumulators (N,H,W,*) resident in HBM
        self.ids = torch.zeros(n_views, self.H, self.W, 4, dtype=torch.int32, device=dev)
        self.colors = torch.zeros(n_views, self.H, self.W, 3, dtype=torch.float16, device=dev)
        self.masks = torch.zeros(n_views, self.H, self.W, dtype=torch.float16, device=dev)
        self.noise = torch.zeros(n_views, 4, self.h, self.w, dtype=torch.float32, device=dev)
        # keep_planes: EngineData also carries normal / depth / canny maps (what a workflow graph's ControlNetApply nodes read)
        self.keep_planes = bool(keep_planes or self.controls)
        self.normal_depth = torch.zeros(n_views, self.H, self.W, 4, dtype=torch.float16, device=dev) if self.keep_planes else None
        self.canny = torch.zeros(n_views, self.H, self.W, 3, dtype=torch.float32, device=dev) if self.keep_planes else None
        self.frame0 = 0

    def set_prompt(self, positive, negative):
        self._prompt = (positive, negative)
        self.runner.set_conditioning(positive, negative)

    def spawn(self, slot):
        """a second pipeline over the SAME weights, scene and corr-map with its own plans / buffers (InflightCalls); built with
        split-K scratch number ``slot``"""
        shard = None
        if self.shard is not None:
            # calls in flight inside a view-sharded group: every slot gets its OWN process group (communicator), so the collectives
            # of different calls never share an ordering domain; every rank must spawn its slots in the same order (new_group is
            # collective).  Rehearsed with gloo (2 ranks) and in a one-rank RCCL group; not the default anywhere (bench.py
            # --shard-inflight): with several communicators in flight RCCL relies on their kernels being co-schedulable
            import torch.distributed as dist
            from .parallel import ViewShard
            shard = ViewShard(self.N_all, group=dist.new_group())
        with O.workspace_slot(slot):
            p = FramePipeline(self.unet, self.vae, self.scene, n_views=self.N_all, steps=self.steps, cfg=self.cfg,
                              sampler=self.sampler, scheduler=self.scheduler, corresponder=copy.copy(self.corresponder),
                              use_graph=self.runner.use_graph, controls=self.controls, keep_planes=self.keep_planes, shard=shard)
        p.bg_noise = self.bg_noise
        p.frame0 = self.frame0
        if getattr(self, "_prompt", None) is not None:
            p.set_prompt(*self._prompt)
        return p

    def render_views(self):
        """N consecutive frames -> EngineData (the per-frame part of _save_frame_data)."""
        first = self.frame0 + (0 if self.shard is None else self.shard.rank * self.N)
        for i in range(self.N):
            self.gbuf.render(self.scene.tasks(first + i), self.scene.camera)
            self.ids[i].copy_(self.gbuf.id)
            self.colors[i].copy_(self.gbuf.color[..., :3])
            alpha = self.gbuf.color[..., 3].contiguous()
            self.masks[i].copy_(1.0 - alpha)
            if self.keep_planes:
                self.normal_depth[i].copy_(self.gbuf.normal_depth)
                self.canny[i].copy_(self.gbuf.canny)
            _, nz = O.noise_pool(self.gbuf.noise.unsqueeze(0), alpha.unsqueeze(0), self.bg_noise)
            self.noise[i].copy_(nz[0])
        self.frame0 += self.N_all
        idm = IDMap(self.ids)
        self._ids_all = None
        if self.shard is not None:
            self._ids_all = IDMap(self.shard.gather_latents(self.ids))        # every rank needs every view's ids
        planes = {}
        if self.keep_planes:                                    # _save_frame_data: normal = rgb, depth = a repeated to 3 channels
            planes = dict(normal_maps=self.normal_depth[..., :3], depth_maps=self.normal_depth[..., 3:4].expand(-1, -1, -1, 3),
                          canny_maps=self.canny)
        return EngineData(frame_indices=list(range(self.N)), color_maps=self.colors, id_maps=idm, masks=self.masks, **planes,
                          noise_maps=LATENT(samples=torch.zeros_like(self.noise), noise=self.noise),
                          correspond_maps={(self.scene.sprite, self.scene.material): self.scene.corrmap})

    def diffuse(self, ed, rng_turn=None):
        corr = self.corresponder
        cb, pre, n_rand = None, None, None
        if isinstance(corr, OverlapCorresponder):
            if self.sampler not in ("ddim", "ddpm"):
                raise ValueError("OverlapCorresponder only works with ddim or ddpm sampler_name.")   # _nodes/samplers.py:163-164
            n_rand = corr.pre_attn_inject_num_random_frames

            if self.shard is None:
                def cb(ctx):
                    corr.step_finished(ed, ctx)
            else:
                idx_all = self._ids_all.overlap_index(self.h, self.w)
                pending = {}

                # the step's input latent is final before its UNet evaluation starts (the evaluation only reads it), so the
                # all-gather the overlap needs is started HERE and runs on RCCL's stream underneath the evaluation; the
                # callback after the evaluation only waits for it
                def pre(x, i, timestep):
                    if float(timestep) >= corr.step_finished_stop_inject_timestep:
                        pending[i] = self.shard.gather_latents_start(x)

                def cb(ctx):
                    if ctx.timestep < corr.step_finished_stop_inject_timestep:
                        return
                    self.shard.overlap_step(ctx.noise, lambda full: idx_all.step(full, corr.step_finished_inject_ratio),
                                            handle=pending.pop(ctx.step_index, None))
        if self.controls:
            self.runner.set_control_hints(self.control_hints())
        samples, inj = self.runner.sample(ed.noise_maps["noise"], self.steps, self.sampler, self.scheduler,
                                          latent_image=ed.noise_maps["samples"], inject_n_rand=n_rand, step_callback=cb,
                                          rng_turn=rng_turn, pre_step_callback=pre)
        if isinstance(corr, OverlapCorresponder) and inj is not None:
            corr._random_frame_indices = torch.tensor(inj)
        return samples

    def control_hints(self):
        """the G-buffer planes of the rendered views as ControlNet hints, one (N,3,H,W) tensor per attached net"""
        planes = {"depth": lambda: self.normal_depth[..., 3:4].expand(-1, -1, -1, 3), "normal": lambda: self.normal_depth[..., :3],
                  "color": lambda: self.colors, "canny": lambda: self.canny}
        return [planes[k]().permute(0, 3, 1, 2).float().contiguous() for k, _ in self.controls]

    def decode(self, samples):
        self.vplan["z"].copy_(samples)
        self.vplan["plan"].run()
        return self.vplan["img"]                       # (N, H, W, 3) fp32 in [0,1]

    def call(self, timings=None, order=None):
        """one bake call = N frames; returns the decoded frames.  timings: optional dict filled with per-stage wall ms
        (forces a device sync after every stage: diagnostics only).  order: (CallOrder, call index) when several calls are in
        flight (InflightCalls)."""
        import time

        def mark(name, t0):
            if timings is not None:
                torch.cuda.synchronize()
                timings[name] = timings.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
            return time.perf_counter()
        t = time.perf_counter()
        ed = self.render_views()
        t = mark("raster+engine_data", t)
        samples = self.diffuse(ed, rng_turn=None if order is None else order[0].turn("rng", order[1]))
        t = mark("sampling", t)
        images = self.decode(samples)
        t = mark("vae_decode", t)
        if self.shard is None:
            with (contextlib.nullcontext() if order is None else order[0].turn("bake", order[1])):
                self.baker.finished(ed, images)
                if order is not None:
                    torch.cuda.current_stream().synchronize()       # the next call's update runs on another stream
        else:
            with (contextlib.nullcontext() if order is None else order[0].turn("bake", order[1])):   # (call order on every rank alike)
                frames = self.shard.gather_frames_to_rank0(images)            # 'first' priority = frame order
                if self.shard.rank == 0:
                    ed_all = EngineData(frame_indices=list(range(self.N_all)), id_maps=self._ids_all, correspond_maps=ed.correspond_maps)
                    self.baker.finished(ed_all, frames)
                if order is not None:
                    torch.cuda.current_stream().synchronize()
        mark("corrmap_update", t)
        return images


def build_sd15_pipeline(dtype=torch.float16, n_views=8, steps=20, cfg=8.0, W=512, H=512, seed=0, unet_cfg=None,
                        use_graph=True, vae_ch=128, device="cuda", shard=None, controls=None):
    """Random-init SD1.5-shaped UNet + VAE decoder (no checkpoints offline; synth.py) on the bake_ball scene."""
    from . import synth
    from .unet import UNet, SD15_CFG
    from .vae import VAEDecoder
    from .model_shapes import unet_names_shapes, vae_decoder_names_shapes
    cfgu = dict(SD15_CFG if unet_cfg is None else unet_cfg)
    ns, norms = unet_names_shapes(cfgu)
    unet = UNet(synth.synth_state_dict(ns, seed=seed, norm_names=norms), cfgu, dtype=dtype, device=device)
    vns, vnorms = vae_decoder_names_shapes(ch=vae_ch)
    vae = VAEDecoder(synth.synth_state_dict(vns, seed=seed + 2, norm_names=vnorms), dtype=dtype, device=device)
    scene = BakeBallScene(W, H, device=device)
    cns = []
    for i, (kind, strength) in enumerate(controls or []):       # e.g. [("depth", 1.0), ("normal", 1.0)]: miku-control.json's pair
        from .controlnet import ControlNet
        from .model_shapes import controlnet_names_shapes
        cns_, cnorms = controlnet_names_shapes(cfgu)
        cns.append((kind, ControlNet(synth.synth_state_dict(cns_, seed=seed + 20 + i, norm_names=cnorms), cfgu, dtype=dtype,
                                     device=device, strength=strength)))
    pipe = FramePipeline(unet, vae, scene, n_views=n_views, steps=steps, cfg=cfg, use_graph=use_graph, shard=shard, controls=cns)
    g = torch.Generator().manual_seed(seed + 11)
    cd = cfgu["context_dim"]
    pipe.set_prompt(torch.randn(1, 77, cd, generator=g), torch.randn(1, 77, cd, generator=g))
    return pipe
