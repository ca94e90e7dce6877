"""Headless ``Engine`` with the reference's script API (engine/engine.py:44-368): subclass, override ``beforePrepare`` to
build the scene out of ``GameObject`` + components, call ``Sample.Run(winSize=..., mode=..., baking_interval=..., ...)``.
No window, no GLFW, no input: the stage loop is rasterise -> (every ``baking_interval`` frames) diffuse -> write back.
Frame -> pose is deterministic (``EqualIntervalRotation``: 360/interval degrees per frame) instead of wall-clock driven
(SURVEY.md App. A).  Only what the hot path needs is modelled; kwargs of the reference that concern the window / UI are
accepted and ignored."""
from enum import Enum
from typing import List, Optional

import numpy as np
import torch

from . import scene as S
from .corrmap import CorrespondMap, IDMap
from .types import EngineData, LATENT
from . import ops as O


class EngineMode(Enum):
    GAME = 0
    BAKE = 1


class RenderMode:
    NORMAL, BAKED, BAKING = 0, 1, 2


class Transform:
    def __init__(self, position, scale):
        self.position = np.asarray(position, np.float32)
        self.scale = np.ones(3, np.float32) * np.asarray(scale, np.float32)
        self.yaw_deg = 0.0
        self._look_at = None

    def lookAt(self, target):
        self._look_at = np.asarray(target, np.float32)

    def rotateLocalY(self, deg):
        self.yaw_deg += deg

    @property
    def matrix(self):
        return S.matmul(S.translate(self.position), S.matmul(S.rotate_y(self.yaw_deg), S.scale(self.scale)))


class Component:
    def __init__(self, gameObj, **kw):
        self.gameObj = gameObj

    def fixedUpdate(self):
        pass


class GameObject:
    _all: List["GameObject"] = []

    def __init__(self, name, position=(0, 0, 0), scale=1.0):
        self.name = name
        self.transform = Transform(position, scale)
        self.components = []
        GameObject._all.append(self)

    def addComponent(self, cls, **kw):
        c = cls(self, **kw)
        self.components.append(c)
        return c

    def getComponent(self, cls):
        for c in self.components:
            if isinstance(c, cls):
                return c
        return None


class Camera(Component):
    def __init__(self, gameObj, fov=45.0, near_plane=0.1, far_plane=100.0, bgPrompt=None, **kw):
        super().__init__(gameObj)
        self.fov, self.near_plane, self.far_plane, self.bgPrompt = fov, near_plane, far_plane, bgPrompt

    def to_scene_camera(self):
        t = self.gameObj.transform
        target = t._look_at if t._look_at is not None else t.position + np.array([0, 0, 1], np.float32)   # forward = +Z
        return S.Camera(t.position, target, fov=self.fov, near=self.near_plane, far=self.far_plane)


class Material:
    _next_id = 1

    def __init__(self, order):
        self.materialID = Material._next_id          # process-global counter starting at 1 (material.py:22-29)
        Material._next_id += 1
        self.render_order = order
        self.textures = {}

    @staticmethod
    def DefaultOpaqueMaterial():
        return Material(S.RenderOrder.OPAQUE)

    @staticmethod
    def DefaultTransparentMaterial():
        return Material(S.RenderOrder.TRANSPARENT)

    def addDefaultTexture(self, tex, kind):
        self.textures[kind] = tex


class DefaultTextureType:
    DiffuseTex, NoiseTex, NormalTex = "diffuse", "noise", "normal"


class Texture:
    @staticmethod
    def CreateNoiseTex(width=512, height=512, device="cuda", seed=None):
        g = None if seed is None else torch.Generator().manual_seed(seed)
        return torch.randn(height, width, 4, generator=g).half().to(device)      # texture.py:507-568 (RGBA16F, NEAREST)


class SpriteInfo(Component):
    _next_id = 1

    def __init__(self, gameObj, auto_spriteID=True, spriteID=None, prompt=''):
        super().__init__(gameObj)
        if spriteID is None:
            spriteID = SpriteInfo._next_id          # stable_render_utils/sprite.py:5-12
            SpriteInfo._next_id += 1
        self.spriteID, self.prompt = spriteID, prompt


class EqualIntervalRotation(Component):
    def __init__(self, gameObj, interval=360):
        super().__init__(gameObj)
        self.interval = interval

    def fixedUpdate(self):
        self.gameObj.transform.rotateLocalY(360.0 / self.interval)


class MeshRenderer(Component):
    def __init__(self, gameObj, mesh=None, materials=None):
        super().__init__(gameObj)
        self.mesh, self.materials = mesh, list(materials or [])

    def addMaterial(self, m):
        self.materials.append(m)

    def tasks(self, cam_view):
        out = []
        sp = self.gameObj.getComponent(SpriteInfo)
        for m in self.materials:
            z = self._cam_z(cam_view)
            if z <= 0:
                continue                              # mesh_renderer.py:90-117: objects behind the camera are skipped
            order = (m.render_order - 1.0 / (z + 1.0)) if m.render_order < S.RenderOrder.TRANSPARENT else (m.render_order + 1.0 / (z + 1.0))
            out.append(S.DrawTask(self.mesh, self.gameObj.transform.matrix, sprite_id=sp.spriteID if sp else 0,
                                  material_id=m.materialID, render_mode=RenderMode.NORMAL,
                                  diffuse_tex=m.textures.get(DefaultTextureType.DiffuseTex),
                                  noise_tex=m.textures.get(DefaultTextureType.NoiseTex), order=order))
        return out

    def _cam_z(self, view):
        p = np.append(self.gameObj.transform.position, 1.0).astype(np.float32)
        return float(-(view.T @ p)[2])


class CorrMapRenderer(MeshRenderer):
    """corrmap_renderer.py:122-190: sphere proxy drawn with renderMode BAKING (bake mode) or BAKED, corr-map k, texcoord ids"""

    def __init__(self, gameObj, corrmaps=None, materials=None, use_texcoord_id=True, mesh=None):
        super().__init__(gameObj, mesh=mesh or S.Mesh.Sphere(32), materials=materials)
        self.corrmap: CorrespondMap = corrmaps
        self.use_texcoord_id = use_texcoord_id

    def tasks(self, cam_view, mode=EngineMode.BAKE):
        ts = super().tasks(cam_view)
        for t in ts:
            t.render_mode = RenderMode.BAKING if mode == EngineMode.BAKE else RenderMode.BAKED
            t.corrmap_k = self.corrmap.k
            t.use_texcoord_id = self.use_texcoord_id
            t.id_size = (self.corrmap.width, self.corrmap.height)
            t.corrmap = self.corrmap if t.render_mode == RenderMode.BAKED else None
        return ts


class _Managers:
    pass


class Engine:
    """Subclass and override the hooks; ``Run(**kwargs)`` / ``Bake(**kwargs)`` as in the reference (engine.py:343-368)."""
    _instance: Optional["Engine"] = None

    def __init__(self, winSize=(512, 512), mode=EngineMode.GAME, baking_interval=8, target_device=0, pipeline=None,
                 max_frames=None, diffuse_workflow=None, disableComfyUI=False, **ignored):
        """diffuse_workflow: a ``workflow.Workflow`` or the path of a workflow JSON (engine.py:97, diffusionManager.py:36-77): the
        graph every submitted EngineData runs through (``workflow.PromptExecutor``, kept across frames so loaders are cached);
        ``disableComfyUI=True`` rasterises only, as in the reference.  ``pipeline`` (a callable EngineData -> images) takes
        precedence when given."""
        Engine._instance = self
        GameObject._all = []
        self.mode, self.baking_interval = mode, baking_interval
        self.device = torch.device("cuda", target_device if isinstance(target_device, int) else 0)
        W, H = winSize
        self.WindowManager = _Managers(); self.WindowManager.WindowSize = (W, H)
        self.RuntimeManager = _Managers(); self.RuntimeManager.FrameCount = 0
        self.RenderManager = _Managers()
        self.RenderManager.GlobalBGNoise = torch.randn(1, H, W, 4, dtype=torch.float32).to(self.device)   # renderManager.py:869-875
        self.gbuf = S.GBuffer(W, H, device=self.device)
        self.pipeline = pipeline                          # callable(EngineData) -> images (N,H,W,3) or None (raster only)
        self.DiffusionManager = _Managers()
        self.DiffusionManager.Workflow, self.DiffusionManager.Executor = None, None
        if pipeline is None and diffuse_workflow is not None and not disableComfyUI:
            from . import workflow as WF
            wf = diffuse_workflow if isinstance(diffuse_workflow, WF.Workflow) else WF.Workflow.Load(diffuse_workflow)
            ex = WF.PromptExecutor()
            self.DiffusionManager.Workflow, self.DiffusionManager.Executor = wf, ex

            def _submit(engine_data):
                ctx = WF.run_workflow(wf, engine_data=engine_data, executor=ex)
                if not ctx.success or ctx.final_output is None:          # renderManager.py:1014-1015
                    mes = ctx.status_messages[-1][1] if ctx.status_messages else {}
                    raise ValueError(f"Prompt execution failed: {mes.get('exception_type')}: {mes.get('exception_message')}")
                return ctx.final_output.frame_color
            self.pipeline = _submit
        self.max_frames = max_frames
        self._exit = False
        self._acc = {}
        self.outputs = []

    # hooks (engine.py:266-278)
    def beforePrepare(self): ...
    def afterPrepare(self): ...
    def beforeFrameBegin(self): ...
    def beforeFrameRun(self): ...
    def beforeFrameEnd(self): ...
    def beforeRelease(self): ...
    def afterRelease(self): ...

    def Exit(self):
        self._exit = True

    @classmethod
    def Run(cls, **kwargs):
        e = cls(**kwargs)
        e.run()
        return e

    @classmethod
    def Bake(cls, **kwargs):
        kwargs["mode"] = EngineMode.BAKE
        return cls.Run(**kwargs)

    def _render_frame(self):
        cam_c = next(c for o in GameObject._all for c in o.components if isinstance(c, Camera))
        cam = cam_c.to_scene_camera()
        view = cam.view()
        tasks = []
        for o in GameObject._all:
            for c in o.components:
                if isinstance(c, CorrMapRenderer):
                    tasks += c.tasks(view, self.mode)
                elif isinstance(c, MeshRenderer):
                    tasks += c.tasks(view)
        self.gbuf.render(tasks, cam)

    def _save_frame_data(self):
        """RenderManager._save_frame_data (renderManager.py:877-948), tensors stay in HBM"""
        g = self.gbuf
        a = self._acc
        alpha = g.color[..., 3].contiguous()
        _, nz = O.noise_pool(g.noise.unsqueeze(0), alpha.unsqueeze(0), self.RenderManager.GlobalBGNoise)
        for k, v in (("color_maps", g.color[..., :3].unsqueeze(0)), ("masks", (1.0 - alpha).unsqueeze(0)),
                     ("id_maps", g.id.unsqueeze(0)), ("pos_maps", g.pos.unsqueeze(0)),
                     ("normal_maps", g.normal_depth[..., :3].unsqueeze(0)),
                     ("depth_maps", g.normal_depth[..., 3:4].expand(-1, -1, 3).unsqueeze(0)),
                     ("canny_maps", g.canny.unsqueeze(0)), ("noise_maps", nz)):
            a.setdefault(k, []).append(v.clone())
        a.setdefault("frame_indices", []).append(self.RuntimeManager.FrameCount)

    def _engine_data(self):
        a = self._acc
        cat = lambda k: torch.cat(a[k], 0)
        corr = {}
        for o in GameObject._all:
            r = o.getComponent(CorrMapRenderer)
            if r is not None:
                sp = o.getComponent(SpriteInfo)
                for m in r.materials:
                    corr[(sp.spriteID if sp else 0, m.materialID)] = r.corrmap
        noise = cat("noise_maps")
        # sprites and environment prompts submitted by the components (renderManager.py:678-703, ai/sprite.py:44, camera bgPrompt)
        from .types import EnvPrompt, Sprite, SpriteInfos
        sprites = SpriteInfos()
        for o in GameObject._all:
            sp = o.getComponent(SpriteInfo)
            if sp is not None:
                sprites[sp.spriteID] = Sprite(sp.spriteID, prompt=sp.prompt or "")
        cams = [c for o in GameObject._all for c in o.components if isinstance(c, Camera)]
        bg = cams[0].bgPrompt if cams else None
        envs = [EnvPrompt(prompt=bg or "") for _ in a["frame_indices"]]
        return EngineData(sprite_infos=sprites, env_prompts=envs, frame_indices=list(range(len(a["frame_indices"]))), color_maps=cat("color_maps"),
                          id_maps=IDMap(cat("id_maps").contiguous()), pos_maps=cat("pos_maps"), normal_maps=cat("normal_maps"),
                          depth_maps=cat("depth_maps"), canny_maps=cat("canny_maps"), masks=cat("masks"),
                          noise_maps=LATENT(samples=torch.zeros_like(noise), noise=noise), correspond_maps=corr)

    def run(self):
        self.beforePrepare()
        self.afterPrepare()
        while not self._exit:
            self.beforeFrameBegin()
            if self._exit:
                break
            fc = self.RuntimeManager.FrameCount
            for o in GameObject._all:                   # first frame included, as the reference's fixedUpdate rule
                for c in o.components:
                    c.fixedUpdate()
            self.beforeFrameRun()
            self._render_frame()
            self._save_frame_data()
            submit = (self.mode == EngineMode.GAME) or (fc % self.baking_interval == 0 and fc != 0)   # diffusionManager.py:96-102
            if submit:
                ed = self._engine_data()
                if self.pipeline is not None:
                    self.outputs.append(self.pipeline(ed))
                else:
                    self.outputs.append(ed)
                self._acc = {}
            self.beforeFrameEnd()
            self.RuntimeManager.FrameCount += 1
            if self.max_frames is not None and self.RuntimeManager.FrameCount >= self.max_frames:
                break
        self.beforeRelease()
        self.afterRelease()
