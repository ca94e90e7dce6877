"""Headless ``Engine`` with the reference's script API (engine/engine.py:44-368): subclass, override ``beforePrepare`` to
build the scene out of ``GameObject`` + components, call ``Sample.Run(winSize=..., mode=..., baking_interval=..., ...)``.
No window, no GLFW, no input: the stage loop is update components -> rasterise -> (every ``baking_interval`` frames, or every
frame in GAME mode) diffuse -> write back.  Frame -> pose is deterministic: ``RuntimeManager.DeltaTime`` is the fixed 1/60 s
of the reference's fixedUpdate clock (runtimeManager.py:308-317) instead of wall-clock time, so
``EqualIntervalRotation`` turns 360/interval degrees per frame and ``AutoRotation`` ``angular_spd * DeltaTime``.

The classes mirror what the reference's ``scripts/*.py`` touch, with the same names, constructor keywords and meaning:
``GameObject`` / ``Component`` / ``Transform`` (runtime/gameObj.py, component.py, components/transform.py:20-395), ``Camera`` /
``CameraController`` (components/camera), ``MeshRenderer`` / ``CorrMapRenderer`` (components/renderer), ``SpriteInfo``,
``EqualIntervalRotation`` / ``AutoRotation`` (components/control/rotations.py), ``Material`` / ``Material_MTL`` (static/material),
``Texture`` (static/texture/texture.py: ``Load``, ``CreateNoiseTex``, the raw-bytes constructor), the enums the scripts import
(static/enums.py).  ``compat/source`` re-exports them under the reference's import paths.  Kwargs of the reference that
concern the window / UI / logging are accepted and ignored."""
import math
import os
from enum import Enum
from typing import List, Optional

import numpy as np
import torch

from . import scene as S
from .corrmap import CorrespondMap, IDMap
from .types import EngineData, EnvPrompt, LATENT
from . import ops as O


# ---- enums (engine/static/enums.py) ------------------------------------------------------------------------------------------
class EngineMode(Enum):
    GAME = 0
    BAKE = 1


class RenderMode:
    NORMAL, BAKED, BAKING = 0, 1, 2


RenderOrder = S.RenderOrder


class DefaultTextureType(Enum):
    """enums.py:95-130 (shader sampler names as values)"""
    DiffuseTex = "diffuseTex"
    NormalTex = "normalTex"
    SpecularTex = "specularTex"
    EmissionTex = "emissionTex"
    OcclusionTex = "occlusionTex"
    MetallicTex = "metallicTex"
    RoughnessTex = "roughnessTex"
    DisplacementTex = "displacementTex"
    AlphaTex = "alphaTex"
    NoiseTex = "noiseTex"
    CorrespondMap = "correspond_map"


class TextureWrap(Enum):
    REPEAT, MIRRORED_REPEAT, CLAMP_TO_EDGE, CLAMP_TO_BORDER = range(4)


class TextureFilter(Enum):
    NEAREST, LINEAR, NEAREST_MIPMAP_NEAREST, LINEAR_MIPMAP_NEAREST, NEAREST_MIPMAP_LINEAR, LINEAR_MIPMAP_LINEAR = range(6)


class TextureFormat(Enum):
    RED, RG, RGB, RGBA, BGR, BGRA = "L", "LA", "RGB", "RGBA", "BGR", "BGRA"

    @property
    def channels(self):
        return {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4, "BGR": 3, "BGRA": 4}[self.value]


class TextureDataType(Enum):
    UNSIGNED_BYTE, BYTE, UNSIGNED_SHORT, SHORT, UNSIGNED_INT, INT, HALF, FLOAT = (
        np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.float16, np.float32)


class TextureInternalFormat(Enum):
    RGB, RGBA, RGB8, RGBA8, RGB16F, RGBA16F, RGB32F, RGBA32F, RGBA32I, RED = range(10)


class ProjectionType(Enum):
    PERSPECTIVE, ORTHOGRAPHIC = 0, 1


class GLFW_Key(Enum):
    """the keys the example scripts poll; the headless InputManager never reports one as pressed"""
    W, A, S, D, Q, E, SPACE, ESCAPE, LEFT_SHIFT, UP, DOWN, LEFT, RIGHT = range(13)


_DEFAULT_DEVICE = "cuda"          # where textures are created; "cpu" while a dry-run Engine is alive


# ---- transform (runtime/components/transform.py) --------------------------------------------------------------------------------
def _rot(axis, deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    x, y, z = axis
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]], np.float64)


class Transform:
    """position / rotation / scale of one GameObject (no parenting: the example scenes are flat).  The rotation is kept as the
    3x3 matrix whose columns are the local right / up / forward axes in world space; forward = local +Z (transform.py:169-172),
    ``lookAt`` builds the left-handed look rotation of ``glm.quatLookAtLH`` (:200-210, :236-250), ``rotateLocalX/Y/Z`` post-multiply
    (``glm.rotate(q, angle, axis)``, :132-155), the model matrix is T * R * S (:339-352)."""

    def __init__(self, position=(0, 0, 0), scale=1.0):
        self._pos = np.zeros(3, np.float64)
        self.position = position
        self._scale = np.ones(3, np.float64)
        self.scale = scale
        self.R = np.eye(3)
        self.yaw_deg = 0.0                               # (kept for callers of the round-1 API: total rotateLocalY)

    @property
    def position(self):
        return self._pos

    @position.setter
    def position(self, v):
        self._pos = np.asarray(v, np.float64).reshape(3).copy()

    localPosition = position

    @property
    def scale(self):
        return self._scale

    @scale.setter
    def scale(self, v):
        self._scale = np.ones(3, np.float64) * np.asarray(v, np.float64)

    localScale = scale

    @property
    def forward(self):
        return self.R[:, 2].copy()

    @forward.setter
    def forward(self, f):
        f = np.asarray(f, np.float64)
        f = f / np.linalg.norm(f)
        up = np.array([0.0, 1.0, 0.0])
        if float(self.up @ up) < 0:
            up = -up
        r = np.cross(up, f)
        n = np.linalg.norm(r)
        if n < 1e-12:                                     # looking straight up / down: any right axis orthogonal to f
            r = np.cross(np.array([0.0, 0.0, 1.0]), f)
            n = np.linalg.norm(r)
        r = r / n
        self.R = np.stack([r, np.cross(f, r), f], axis=1)

    @property
    def up(self):
        return self.R[:, 1].copy()

    @property
    def right(self):
        return self.R[:, 0].copy()

    def lookAt(self, target):
        if isinstance(target, GameObject):
            target = target.transform.position
        elif isinstance(target, Transform):
            target = target.position
        self.forward = np.asarray(target, np.float64) - self._pos

    def rotate(self, axis, angle, radian=False):
        self.R = self.R @ _rot(np.asarray(axis, np.float64), math.degrees(angle) if radian else angle)

    def rotateLocalX(self, angle, radian=False):
        self.rotate((1, 0, 0), angle, radian)

    def rotateLocalY(self, angle, radian=False):
        self.rotate((0, 1, 0), angle, radian)
        self.yaw_deg += math.degrees(angle) if radian else angle

    def rotateLocalZ(self, angle, radian=False):
        self.rotate((0, 0, 1), angle, radian)

    @property
    def matrix(self):
        """model matrix in the [col][row] float32 storage of scene.py (glm order T * R * S)"""
        M = np.eye(4)
        M[:3, :3] = self.R * self._scale[None, :]
        M[:3, 3] = self._pos
        return M.T.astype(np.float32).copy()

    globalTransformMatrix = matrix
    transformMatrix = matrix

    def transformPoint(self, p):
        return self.R @ (np.asarray(p, np.float64) * self._scale) + self._pos

    def inverseTransformPoint(self, p):
        return (self.R.T @ (np.asarray(p, np.float64) - self._pos)) / self._scale

    def transformDirection(self, d):
        v = self.R @ (np.asarray(d, np.float64) * self._scale)
        return v / np.linalg.norm(v)


# ---- game objects and components ---------------------------------------------------------------------------------------------
class Component:
    """runtime/component.py: hooks called once per frame in this order: fixedUpdate, update, lateUpdate"""

    def __init__(self, gameObj, enable=True, **kw):
        self.gameObj = gameObj
        self.enable = enable

    @property
    def transform(self):
        return self.gameObj.transform

    @property
    def engine(self):
        return Engine._instance

    def awake(self): ...
    def start(self): ...
    def fixedUpdate(self): ...
    def update(self): ...
    def lateUpdate(self): ...
    def onDestroy(self): ...


class GameObject:
    _all: List["GameObject"] = []

    def __init__(self, name, position=(0, 0, 0), scale=1.0, rotation=None, active=True, **kw):
        self.name = name
        self.transform = Transform(position, scale)
        if rotation is not None:                          # euler degrees (x, y, z), applied as the reference's rotate(x, y, z)
            rx, ry, rz = rotation
            self.transform.R = _rot((0, 0, 1), rz) @ _rot((0, 1, 0), ry) @ _rot((1, 0, 0), rx)
        self.components = []
        self.active = active
        GameObject._all.append(self)

    def addComponent(self, cls, *a, **kw):
        c = cls(self, *a, **kw)
        self.components.append(c)
        c.awake()
        return c

    def getComponent(self, cls):
        for c in self.components:
            if isinstance(c, cls):
                return c
        return None

    def getComponents(self, cls):
        return [c for c in self.components if isinstance(c, cls)]


class Camera(Component):
    """components/camera/camera.py:20-146: fov 45, near 0.1, far 100; view = glm.lookAt(pos, pos + forward, up) (:94-99)"""
    _Main_Camera = None

    def __init__(self, gameObj, enable=True, fov=45.0, near_plane=0.1, far_plane=100.0, bgPrompt=None, **kw):
        super().__init__(gameObj, enable)
        self.fov, self.near_plane, self.far_plane, self.bgPrompt = fov, near_plane, far_plane, bgPrompt
        if Camera._Main_Camera is None or Camera._Main_Camera.gameObj not in GameObject._all:
            Camera._Main_Camera = self

    @staticmethod
    def MainCamera():
        return Camera._Main_Camera

    def to_scene_camera(self):
        t = self.gameObj.transform
        return S.Camera(t.position.astype(np.float32), (t.position + t.forward).astype(np.float32), up=t.up.astype(np.float32),
                        fov=self.fov, near=self.near_plane, far=self.far_plane)


class CameraController(Component):
    """components/control/camera_controller: keyboard / mouse fly camera.  Headless: only its start pose acts
    (``defaultPos`` / ``defaultLookAt``)."""

    def __init__(self, gameObj, enable=True, defaultPos=None, defaultLookAt=None, **kw):
        super().__init__(gameObj, enable)
        if defaultPos is not None:
            self.transform.position = defaultPos
        if defaultLookAt is not None:
            self.transform.lookAt(defaultLookAt)


class EqualIntervalRotation(Component):
    """control/rotations.py:4-33"""

    def __init__(self, gameObj, enable=True, interval: int = 18, axis='y', update_mode='fixed_update'):
        super().__init__(gameObj, enable)
        self.rotation_interval, self.rotation_axis, self.update_mode = 360 / interval, axis, update_mode
        self.interval = interval

    def _update(self):
        if self.rotation_axis not in "xyz":
            raise ValueError(f"Invalid axis: {self.rotation_axis}")
        getattr(self.transform, "rotateLocal" + self.rotation_axis.upper())(self.rotation_interval)

    def update(self):
        if self.update_mode == 'update':
            self._update()

    def fixedUpdate(self):
        if self.update_mode == 'fixed_update':
            self._update()


class AutoRotation(Component):
    """control/rotations.py:35-53: angular_spd (degrees / second) * DeltaTime per frame"""

    def __init__(self, gameObj, enable=True, angular_spd: float = 4, axis='y'):
        super().__init__(gameObj, enable)
        self.angular_spd, self.rotation_axis = angular_spd, axis

    def update(self):
        if self.rotation_axis not in "xyz":
            raise ValueError(f"Invalid axis: {self.rotation_axis}")
        getattr(self.transform, "rotateLocal" + self.rotation_axis.upper())(self.angular_spd * self.engine.RuntimeManager.DeltaTime)


class SpriteInfo(Component):
    """components/ai/sprite.py + common_utils/stable_render_utils/sprite.py:5-32: process-global sprite ids from 1"""
    _next_id = 1

    def __init__(self, gameObj, enable=True, auto_spriteID=True, spriteID=None, prompt='', prompt_weight=1.0, neg_prompt='',
                 neg_prompt_weight=1.0):
        super().__init__(gameObj, enable)
        if spriteID is None:
            spriteID = SpriteInfo._next_id
            SpriteInfo._next_id += 1
        self.spriteID, self.prompt, self.prompt_weight = spriteID, prompt, prompt_weight
        self.neg_prompt, self.neg_prompt_weight = neg_prompt, neg_prompt_weight


# ---- static resources: textures, materials ----------------------------------------------------------------------------------------
class Texture:
    """static/texture/texture.py.  A texture here is an HBM tensor ``data`` (H, W, 4): row 0 = v in [0, 1/H) (GL's bottom row --
    ``Load`` flips the image as the reference does, :421-431).  Wrap is REPEAT.  A diffuse texture whose min filter is a LINEAR
    mip-map mode (the default, as in the reference) is sampled TRILINEAR by the rasterizer (scene.build_mip_chain,
    sr_draw.diffuse_levels; anisotropy not restated); noise, id and corr-map lookups are NEAREST in the reference and here."""

    def __init__(self, name=None, width=None, height=None, format=TextureFormat.RGB, data_type=None, data=None,
                 min_filter=TextureFilter.LINEAR_MIPMAP_LINEAR, mag_filter=TextureFilter.LINEAR, s_wrap=TextureWrap.REPEAT,
                 t_wrap=TextureWrap.REPEAT, internal_format=None, share_to_torch=False, device=None, **kw):
        self.name, self.format, self.data_type, self.internal_format = name, format, data_type, internal_format
        self.min_filter, self.mag_filter, self.s_wrap, self.t_wrap = min_filter, mag_filter, s_wrap, t_wrap
        self.device = device or _DEFAULT_DEVICE
        self.data = None
        if isinstance(data, torch.Tensor):
            self.data = self._to_rgba(data)
        elif data is not None:                                        # raw bytes, row-major, `format` channels of `data_type`
            if width is None or height is None:
                raise ValueError("Texture(data=bytes) needs width and height")
            dt = (data_type or TextureDataType.UNSIGNED_BYTE).value
            a = np.frombuffer(data, dtype=dt).reshape(height, width, format.channels).copy()
            t = torch.from_numpy(a.astype(np.float32) / (255.0 if dt == np.uint8 else 1.0))
            self.data = self._to_rgba(t)
        self.width = width if self.data is None else self.data.shape[1]
        self.height = height if self.data is None else self.data.shape[0]

    def _to_rgba(self, t):
        t = t.float()
        if t.dim() == 2:
            t = t.unsqueeze(-1)
        c = t.shape[-1]
        if c == 1:
            t = torch.cat([t, t, t, torch.ones_like(t)], -1)
        elif c == 3:                                                  # Texture.Load defaults to RGB: sampled alpha = 1
            t = torch.cat([t, torch.ones_like(t[..., :1])], -1)
        elif c != 4:
            raise ValueError(f"texture with {c} channels")
        return t.contiguous().to(self.device)

    def load(self):                                                   # the GL upload of the reference: data already lives in HBM
        return self

    sendToGPU = load

    def tensor(self, update=True, flip=False):
        return self.data.flip(0) if flip else self.data

    @classmethod
    def Load(cls, path, name=None, format=TextureFormat.RGB, device=None, **kw):
        """texture.py:409-447: PIL open -> convert -> FLIP_TOP_BOTTOM -> bytes"""
        from PIL import Image
        mode = "RGBA" if format in (TextureFormat.RGBA, TextureFormat.BGRA) else ("L" if format == TextureFormat.RED else "RGB")
        img = Image.open(path).convert(mode).transpose(Image.FLIP_TOP_BOTTOM)
        a = np.asarray(img, np.float32) / 255.0
        return cls(name=name or os.path.basename(str(path)), format=format, data=torch.from_numpy(a), device=device, **kw)

    @staticmethod
    def CreateNoiseTex(*args, device=None, seed=None, **kw):
        """texture.py:507-568 ``CreateNoiseTex(name=None, width=512, height=512, ...)``: randn RGBA16F, NEAREST.  Also accepts the
        (width, height) positional form.  Returns the fp16 (H, W, 4) tensor the rasterizer samples (a Material takes either a
        tensor or a Texture)."""
        args = list(args)
        if args and (args[0] is None or isinstance(args[0], str)):
            args.pop(0)
        width = kw.get("width", args[0] if len(args) > 0 else 512)
        height = kw.get("height", args[1] if len(args) > 1 else 512)
        g = None if seed is None else torch.Generator().manual_seed(seed)
        return torch.randn(height, width, 4, generator=g).half().to(device or _DEFAULT_DEVICE)


class Material:
    """static/material/material.py: materialIDs are a process-global counter from 1 (:22-29)"""
    _next_id = 1

    def __init__(self, order=RenderOrder.OPAQUE, real_name=None, **kw):
        self.materialID = Material._next_id
        Material._next_id += 1
        self.render_order = order
        self.textures = {}
        self.real_name = real_name
        self.params = kw

    @classmethod
    def DefaultOpaqueMaterial(cls, **kw):
        return cls(RenderOrder.OPAQUE, **kw)

    @classmethod
    def DefaultTransparentMaterial(cls, **kw):
        return cls(RenderOrder.TRANSPARENT, **kw)

    @classmethod
    def DefaultDebugMaterial(cls, **kw):
        m = cls(RenderOrder.OPAQUE, **kw)
        m.debug = True
        return m

    def addDefaultTexture(self, tex, kind):
        self.textures[kind] = tex

    def hasDefaultTexture(self, kind):
        return kind in self.textures

    def tensor_of(self, kind):
        """-> the (H, W, 4) tensor the rasterizer samples for this slot, or None"""
        t = self.textures.get(kind)
        if isinstance(t, Texture):
            return t.data
        return t if isinstance(t, torch.Tensor) else None

    def filter_of(self, kind):
        """-> 'trilinear' for a Texture whose min filter is a LINEAR mip-map mode (the reference's default, texture.py:57-60: file
        textures are sampled GL_LINEAR_MIPMAP_LINEAR / GL_LINEAR), 'nearest' for NEAREST textures and bare tensors"""
        t = self.textures.get(kind)
        if isinstance(t, Texture) and t.min_filter in (TextureFilter.LINEAR, TextureFilter.LINEAR_MIPMAP_NEAREST,
                                                       TextureFilter.NEAREST_MIPMAP_LINEAR, TextureFilter.LINEAR_MIPMAP_LINEAR):
            return "trilinear"
        return "nearest"


class Material_MTL(Material):
    """static/material/material_MTL.py:13-110: one material per ``newmtl`` block; ``map_Kd`` / ``map_bump`` textures are looked up
    beside the .mtl file"""

    @classmethod
    def Load(cls, path, add_textures=True, device=None):
        mats, cur, lines = [], None, {}
        base = os.path.dirname(str(path))
        with open(path) as f:
            for line in f:
                p = line.split()
                if not p or p[0].startswith("#"):
                    continue
                if p[0] == "newmtl":
                    cur = cls.DefaultOpaqueMaterial(real_name=" ".join(p[1:]))
                    mats.append(cur)
                elif cur is not None:
                    if p[0] in ("Ka", "Kd", "Ks"):
                        cur.params[p[0].lower()] = tuple(map(float, p[1:4]))
                    elif p[0] in ("Ns", "Ni", "d"):
                        cur.params[p[0].lower()] = float(p[1])
                    elif p[0] == "illum":
                        cur.params["illum"] = int(p[1])
                    elif add_textures and p[0] in ("map_Kd", "map_bump", "map_Ks", "map_d"):
                        kind = {"map_Kd": DefaultTextureType.DiffuseTex, "map_bump": DefaultTextureType.NormalTex,
                                "map_Ks": DefaultTextureType.SpecularTex, "map_d": DefaultTextureType.AlphaTex}[p[0]]
                        fp = os.path.join(base, p[-1])
                        if os.path.exists(fp):
                            cur.addDefaultTexture(Texture.Load(fp, device=device), kind)
        return tuple(mats)


# ---- renderers -------------------------------------------------------------------------------------------------------------------
class MeshRenderer(Component):
    """components/renderer/mesh_renderer.py: one G-buffer task per material (mesh group i drawn with material i)"""

    def __init__(self, gameObj, enable=True, mesh=None, materials=None, use_texcoord_id=False, **kw):
        super().__init__(gameObj, enable)
        self.mesh = mesh
        self.materials = [materials] if isinstance(materials, Material) else list(materials or [])
        self.use_texcoord_id = use_texcoord_id

    def addMaterial(self, m, duplicateCheck=True):
        self.materials.append(m)

    def load_MTL_Materials(self, mats):
        """mesh_renderer.py:55-74: order the MTL materials as the mesh's ``usemtl`` groups name them; unnamed parts turn pink"""
        if self.mesh is None:
            raise ValueError('No mesh loaded. Cannot load MTL materials.')
        mats = [mats] if isinstance(mats, Material) else list(mats)
        by_name = {m.real_name: m for m in mats}
        for g in getattr(self.mesh, "materials", []) or []:
            name = g.get("NAME")
            if name is not None:
                self.addMaterial(by_name[name] if name in by_name else Material.DefaultDebugMaterial(), duplicateCheck=False)

    @property
    def spriteID(self):
        sp = self.gameObj.getComponent(SpriteInfo)
        return sp.spriteID if sp else None

    def _order(self, m, cam_view):
        z = self._cam_z(cam_view)                                      # camera-local z of the object origin (mesh_renderer.py:90-117)
        if z <= 0 and m.render_order < RenderOrder.OVERLAY:
            return None
        if m.render_order < RenderOrder.TRANSPARENT:
            return m.render_order - 1.0 / (z + 1.0)
        if m.render_order < RenderOrder.OVERLAY:
            return m.render_order + 1.0 / (z + 1.0)
        return m.render_order

    def _sub_mesh(self, i):
        groups = getattr(self.mesh, "groups", None)
        if not groups or len(self.materials) <= 1:
            return self.mesh
        return self.mesh.group_mesh(i)

    def tasks(self, cam_view):
        out = []
        sid = self.spriteID
        for i, m in enumerate(self.materials):
            order = self._order(m, cam_view)
            if order is None:
                continue
            mesh = self._sub_mesh(i)
            if mesh is None:
                continue
            out.append(S.DrawTask(mesh, self.gameObj.transform.matrix, sprite_id=sid or 0, material_id=m.materialID,
                                  render_mode=RenderMode.NORMAL, diffuse_tex=m.tensor_of(DefaultTextureType.DiffuseTex),
                                  diffuse_filter=m.filter_of(DefaultTextureType.DiffuseTex),
                                  noise_tex=m.tensor_of(DefaultTextureType.NoiseTex),
                                  normal_tex=m.tensor_of(DefaultTextureType.NormalTex),
                                  use_texcoord_id=bool(self.use_texcoord_id and mesh.has_uvs),
                                  has_vertex_color=mesh.colors is not None, order=order))
        return out

    def _cam_z(self, view):
        p = np.append(self.gameObj.transform.position, 1.0).astype(np.float32)
        return float(-(view.T @ p)[2])


class CorrMapRenderer(MeshRenderer):
    """corrmap_renderer.py:122-190: proxy mesh (a sphere by default) drawn with renderMode BAKING in bake mode, BAKED otherwise;
    corr-map k, texcoord ids; needs a sprite id (``_drawAvailable``, :117-120)"""

    def __init__(self, gameObj, enable=True, corrmaps=None, materials=None, use_texcoord_id=True, mesh=None,
                 auto_noise_map_if_not_exist=True, **kw):
        super().__init__(gameObj, enable, mesh=mesh or S.Mesh.Sphere(32), materials=materials, use_texcoord_id=use_texcoord_id)
        self.corrmaps = [corrmaps] if isinstance(corrmaps, CorrespondMap) else list(corrmaps or [])
        self.auto_noise_map_if_not_exist = auto_noise_map_if_not_exist

    @property
    def corrmap(self):
        return self.corrmaps[0] if self.corrmaps else None

    def start(self):
        for i, mat in enumerate(self.materials[:len(self.corrmaps)]):
            if not mat.hasDefaultTexture(DefaultTextureType.NoiseTex) and self.auto_noise_map_if_not_exist:
                mat.addDefaultTexture(Texture.CreateNoiseTex(), DefaultTextureType.NoiseTex)

    def tasks(self, cam_view, mode=EngineMode.BAKE):
        if self.spriteID is None or len(self.corrmaps) != len(self.materials) or self.mesh is None:
            return []
        ts = super().tasks(cam_view)
        for t, cm in zip(ts, self.corrmaps):
            t.render_mode = RenderMode.BAKING if mode == EngineMode.BAKE else RenderMode.BAKED
            t.corrmap_k = cm.k
            t.use_texcoord_id = bool(self.use_texcoord_id and t.mesh.has_uvs)
            t.id_size = (cm.width, cm.height)
            t.corrmap = cm if t.render_mode == RenderMode.BAKED else None
        return ts


class _Managers:
    pass


class _InputManager:
    def GetKey(self, key):
        return False

    GetKeyDown = GetKeyUp = GetMouseBtn = GetKey


# ---- the engine ------------------------------------------------------------------------------------------------------------------
class Engine:
    """Subclass and override the hooks; ``Run(**kwargs)`` / ``Bake(**kwargs)`` as in the reference (engine.py:343-368)."""
    _instance: Optional["Engine"] = None

    def __init__(self, winSize=(512, 512), mode=EngineMode.GAME, baking_interval=8, target_device=0, pipeline=None,
                 max_frames=None, diffuse_workflow=None, disableComfyUI=False, fixed_delta_time=1.0 / 60.0, dry_run=None,
                 **ignored):
        """diffuse_workflow: a ``workflow.Workflow`` or the path of a workflow JSON (engine.py:97, diffusionManager.py:36-77): the
        graph every submitted EngineData runs through (``workflow.PromptExecutor``, kept across frames so loaders are cached);
        ``disableComfyUI=True`` rasterises only, as in the reference.  ``pipeline`` (a callable EngineData -> images) takes
        precedence when given.  ``fixed_delta_time``: the headless clock (seconds per frame).  ``dry_run`` (or
        $SR_ENGINE_DRY_RUN=1): build and animate the scene and record every frame's (camera, draw tasks) in ``self.frames``
        without touching a GPU -- what the scene-compatibility tests inspect; $SR_ENGINE_MAX_FRAMES bounds the loop."""
        global _DEFAULT_DEVICE
        self.dry_run = (os.environ.get("SR_ENGINE_DRY_RUN") == "1") if dry_run is None else bool(dry_run)
        if max_frames is None and os.environ.get("SR_ENGINE_MAX_FRAMES"):
            max_frames = int(os.environ["SR_ENGINE_MAX_FRAMES"])
        _DEFAULT_DEVICE = "cpu" if self.dry_run else "cuda"
        from . import corrmap as _cm
        _cm.DEFAULT_DEVICE = _DEFAULT_DEVICE
        self.frames = []
        Engine._instance = self
        GameObject._all = []
        Camera._Main_Camera = None
        self.mode, self.baking_interval = mode, baking_interval
        self.Mode = mode
        self.device = torch.device("cpu") if self.dry_run else torch.device("cuda", target_device if isinstance(target_device, int) else 0)
        W, H = winSize
        self.WindowManager = _Managers()
        self.WindowManager.WindowSize, self.WindowManager.AspectRatio = (W, H), W / H
        self.RuntimeManager = _Managers()
        self.RuntimeManager.FrameCount, self.RuntimeManager.DeltaTime = 0, float(fixed_delta_time)
        self.InputManager = _InputManager()
        self.RenderManager = _Managers()
        self.RenderManager.GlobalBGNoise = torch.randn(1, H, W, 4, dtype=torch.float32).to(self.device)   # renderManager.py:869-875
        self.gbuf = None if self.dry_run else S.GBuffer(W, H, device=self.device)
        self.pipeline = pipeline                          # callable(EngineData) -> images (N,H,W,3) or None (raster only)
        self.DiffusionManager = _Managers()
        self.DiffusionManager.Workflow, self.DiffusionManager.Executor = None, None
        self.diffuse_workflow = diffuse_workflow
        if pipeline is None and diffuse_workflow is not None and not disableComfyUI and not self.dry_run:
            from . import workflow as WF
            wf = diffuse_workflow if isinstance(diffuse_workflow, WF.Workflow) else WF.Workflow.Load(str(diffuse_workflow))
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
        kwargs["disableComfyUI"] = False                  # engine.py:344-351
        return cls.Run(**kwargs)

    def scene_tasks(self):
        """-> (scene camera, G-buffer tasks of the current scene state)"""
        cam_c = Camera.MainCamera() or next(c for o in GameObject._all for c in o.components if isinstance(c, Camera))
        cam = cam_c.to_scene_camera()
        view = cam.view()
        tasks = []
        for o in GameObject._all:
            if not o.active:
                continue
            for c in o.components:
                if not c.enable:
                    continue
                if isinstance(c, CorrMapRenderer):
                    tasks += c.tasks(view, self.mode)
                elif isinstance(c, MeshRenderer):
                    tasks += c.tasks(view)
        return cam, tasks

    def _render_frame(self):
        cam, tasks = self.scene_tasks()
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
            if r is not None and r.spriteID is not None:
                for m, cm in zip(r.materials, r.corrmaps):          # RenderManager.SubmitCorrmap(spriteID, materialID, corrmap)
                    corr[(r.spriteID, m.materialID)] = cm
        noise = cat("noise_maps")
        # sprites and environment prompts submitted by the components (renderManager.py:678-703, ai/sprite.py:44, camera bgPrompt)
        from .types import Sprite, SpriteInfos
        sprites = SpriteInfos()
        for o in GameObject._all:
            sp = o.getComponent(SpriteInfo)
            if sp is not None:
                sprites[sp.spriteID] = Sprite(sp.spriteID, prompt=sp.prompt or "", prompt_weight=sp.prompt_weight,
                                              neg_prompt=sp.neg_prompt or "", neg_prompt_weight=sp.neg_prompt_weight)
        cam = Camera.MainCamera()
        bg = cam.bgPrompt if cam is not None else None
        env = bg if isinstance(bg, EnvPrompt) else EnvPrompt(prompt=bg or "")
        envs = [env for _ in a["frame_indices"]]
        return EngineData(sprite_infos=sprites, env_prompts=envs, frame_indices=list(range(len(a["frame_indices"]))), color_maps=cat("color_maps"),
                          id_maps=IDMap(cat("id_maps").contiguous()), pos_maps=cat("pos_maps"), normal_maps=cat("normal_maps"),
                          depth_maps=cat("depth_maps"), canny_maps=cat("canny_maps"), masks=cat("masks"),
                          noise_maps=LATENT(samples=torch.zeros_like(noise), noise=noise), correspond_maps=corr)

    def build_scene(self):
        """beforePrepare + component start(): the scene as the first frame will see it (what a dry run inspects)"""
        self.beforePrepare()
        for o in GameObject._all:
            for c in o.components:
                c.start()
        self.afterPrepare()

    def run(self):
        self.build_scene()
        while not self._exit:
            self.beforeFrameBegin()
            if self._exit:
                break
            fc = self.RuntimeManager.FrameCount
            for hook in ("fixedUpdate", "update", "lateUpdate"):     # first frame included, as the reference's fixedUpdate rule
                for o in GameObject._all:
                    if o.active:
                        for c in o.components:
                            if c.enable:
                                getattr(c, hook)()
            self.beforeFrameRun()
            if self.dry_run:
                self.frames.append(self.scene_tasks())
                self.beforeFrameEnd()
                self.RuntimeManager.FrameCount += 1
                if self.max_frames is not None and self.RuntimeManager.FrameCount >= self.max_frames:
                    break
                continue
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


Mesh = S.Mesh
