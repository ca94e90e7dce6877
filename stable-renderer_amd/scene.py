"""Headless scene description + G-buffer rasterisation front-end.

Host-side mirror of the pieces of the reference engine that feed the G-buffer pass: ``Mesh.Sphere / Mesh.Plane /
Mesh.Load`` (engine/static/mesh/mesh.py:321-569), ``Camera`` matrices (engine/runtime/components/camera/camera.py:
94-120: glm.lookAt / glm.perspective, RH, -1..1 depth), ``Transform`` TRS (engine/runtime/components/transform.py:
339-352), the MV / MV_IT uniforms (engine/managers/runtimeManager.py:165-181), ``MeshRenderer`` / ``CorrMapRenderer``
draw parameters (runtime/components/renderer/mesh_renderer.py:76-123, corrmap_renderer.py:122-149) and the task order
+ depth-test rule of ``RenderManager`` (engine/managers/renderManager.py:499-522).  Matrices are a few dozen scalars
computed on the host in fp32 with GLM's formulas (PyGLM itself is absent: parity unpinned); all per-vertex
and per-pixel work is in ``sr_raster_draw``.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import ops as O

F = np.float32


# ---- GLM restatement (column-major 4x4, numpy float32; M[col][row] like glm) --------------------------------------
def perspective(fovy_rad, aspect, near, far):
    t = F(math.tan(fovy_rad / 2.0))
    m = np.zeros((4, 4), F)
    m[0][0] = F(1.0) / (F(aspect) * t)
    m[1][1] = F(1.0) / t
    m[2][2] = -F(far + near) / F(far - near)
    m[2][3] = -F(1.0)
    m[3][2] = -(F(2.0) * F(far) * F(near)) / F(far - near)
    return m


def _norm(v):
    v = np.asarray(v, F)
    return v / F(np.sqrt(np.dot(v, v)))


def look_at(eye, center, up):
    eye, center, up = np.asarray(eye, F), np.asarray(center, F), np.asarray(up, F)
    f = _norm(center - eye)
    s = _norm(np.cross(f, up))
    u = np.cross(s, f)
    m = np.eye(4, dtype=F)
    m[0][0], m[1][0], m[2][0] = s
    m[0][1], m[1][1], m[2][1] = u
    m[0][2], m[1][2], m[2][2] = -f
    m[3][0], m[3][1], m[3][2] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    return m


def translate(v):
    m = np.eye(4, dtype=F)
    m[3][:3] = np.asarray(v, F)
    return m


def scale(v):
    m = np.eye(4, dtype=F)
    v = np.asarray(v, F) * np.ones(3, F)
    m[0][0], m[1][1], m[2][2] = v
    return m


def rotate_y(deg):
    a = math.radians(deg)
    c, s = F(math.cos(a)), F(math.sin(a))
    m = np.eye(4, dtype=F)
    m[0][0], m[0][2], m[2][0], m[2][2] = c, -s, s, c
    return m


def matmul(a, b):
    """glm a*b for [col][row] storage"""
    return (b.astype(F) @ a.astype(F)).astype(F)


def inverse_transpose(m):
    mm = m.astype(np.float64).T                      # to row-major math matrix
    it = np.linalg.inv(mm).T
    return it.T.astype(F)                            # back to [col][row]


# ---- meshes ------------------------------------------------------------------------------------------------------
def strip_to_triangles(idx):
    """GL_TRIANGLE_STRIP -> triangle list keeping orientation; provoking (flat) vertex stays the last one."""
    tris = []
    for k in range(len(idx) - 2):
        a, b, c = idx[k], idx[k + 1], idx[k + 2]
        tris.append((a, b, c) if k % 2 == 0 else (b, a, c))
    return np.asarray(tris, np.int32)


class Mesh:
    def __init__(self, positions, normals, uvs, tris, colors=None, vertex_ids=None, cullback=True, name=None):
        self.positions = np.ascontiguousarray(positions, F)
        self.normals = np.ascontiguousarray(normals, F)
        self.uvs = np.ascontiguousarray(uvs, F)
        self.tris = np.ascontiguousarray(tris, np.int32)
        self.colors = None if colors is None else np.ascontiguousarray(colors, F)
        self.vertex_ids = None if vertex_ids is None else np.ascontiguousarray(vertex_ids, np.int32)
        self.cullback = cullback
        self.name = name
        self._dev = None
        self.has_uvs = bool(np.any(self.uvs != 0))
        self.groups = None                 # [(material name, first triangle, triangle count)] of an OBJ with `usemtl` parts
        self.materials = []                # [{"NAME": ...}] in group order (what assimp reports, mesh.py:360-372)
        self.tangents = None               # per-vertex tangent / bitangent (only needed when a normal map is bound)
        self.bitangents = None
        self._group_meshes = {}

    def group_mesh(self, i):
        """the sub-mesh drawn with material slot i (mesh.draw(slot), mesh.py:300-318); shares the vertex arrays"""
        if not self.groups or i >= len(self.groups):
            return None
        if i not in self._group_meshes:
            _, t0, n = self.groups[i]
            m = Mesh(self.positions, self.normals, self.uvs, self.tris[t0:t0 + n], self.colors, self.vertex_ids, self.cullback,
                     name=f"{self.name}#{i}")
            m.tangents, m.bitangents = self.tangents, self.bitangents
            self._group_meshes[i] = m
        return self._group_meshes[i]

    def compute_tangents(self):
        """per-vertex tangent space from the uv gradients, accumulated over the triangles and normalised: the construction of
        assimp's aiProcess_CalcTangentSpace (mesh.py:337-338 loads with it), without its smoothing-angle vertex splits --
        parity unpinned (assimp is absent), it only matters when a normal map is bound"""
        if self.tangents is not None:
            return
        P, U = self.positions.astype(np.float64), self.uvs.astype(np.float64)
        T, B = np.zeros_like(P), np.zeros_like(P)
        for a, b, c in self.tris:
            e1, e2 = P[b] - P[a], P[c] - P[a]
            d1, d2 = U[b] - U[a], U[c] - U[a]
            det = d1[0] * d2[1] - d2[0] * d1[1]
            if abs(det) < 1e-20:
                continue
            t = (e1 * d2[1] - e2 * d1[1]) / det
            bt = (e2 * d1[0] - e1 * d2[0]) / det
            for v in (a, b, c):
                T[v] += t
                B[v] += bt
        nz = lambda v: v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-20)
        self.tangents, self.bitangents = np.ascontiguousarray(nz(T), F), np.ascontiguousarray(nz(B), F)
        self._dev = None

    def device(self, dev):
        if self._dev is None:
            t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
            self._dev = dict(pos=t(self.positions), normal=t(self.normals), uv=t(self.uvs), tris=t(self.tris),
                             color=t(self.colors), vid=t(self.vertex_ids), tangent=t(self.tangents), bitangent=t(self.bitangents))
        return self._dev

    @staticmethod
    def Sphere(segment=32):
        """mesh.py:518-569: one indexed TRIANGLE_STRIP; the seam column j == segment is not referenced and the id VBO
        is filled from the index buffer (mesh.py:279-281)."""
        pos, nrm, uv = [], [], []
        for i in range(segment + 1):
            for j in range(segment + 1):
                xs, ys = j / segment, i / segment
                x = math.cos(xs * 2 * math.pi) * math.sin(ys * math.pi)
                y = math.cos(ys * math.pi)
                z = math.sin(xs * 2 * math.pi) * math.sin(ys * math.pi)
                pos.append((x, y, z)); nrm.append((x, y, z)); uv.append((xs, ys))
        idx = []
        for i in range(segment):
            for j in range(segment):
                idx.append((i + 1) * (segment + 1) + j)
                idx.append(i * (segment + 1) + j)
        return Mesh(pos, nrm, uv, strip_to_triangles(idx), name="sphere")

    @staticmethod
    def Plane(edge=1):
        """mesh.py:472-515"""
        xs = np.linspace(-0.5, 0.5, edge + 1)
        pos, nrm, uv = [], [], []
        for z in xs:
            for x in xs:
                pos.append((x, 0, z)); nrm.append((0, 1, 0)); uv.append((x + 0.5, z + 0.5))
        tris = []
        for i in range(edge):
            for j in range(edge):
                a, b, c, d = i * (edge + 1) + j, (i + 1) * (edge + 1) + j, (i + 1) * (edge + 1) + j + 1, i * (edge + 1) + j + 1
                tris += [(a, b, c), (a, c, d)]
        return Mesh(pos, nrm, uv, tris, name="plane")

    @staticmethod
    def Load(path, alias=None, cullback=True, **kw):
        """Wavefront OBJ (positions / uvs / normals, polygons fan-triangulated, `usemtl` parts kept as material groups).  The
        reference loads through assimp (Triangulate | CalcTangentSpace | JoinIdenticalVertices, mesh.py:321-408), whose vertex
        order is not reproducible without assimp: vertices here are unique (v, vt, vn) triples in first-use order, triangles are
        regrouped by material in first-use order (assimp splits a mesh per material the same way)."""
        vs, vts, vns, verts, index, tris = [], [], [], [], {}, []
        tri_mat, mat_names, cur_mat = [], [], None
        with open(path) as f:
            for line in f:
                p = line.split()
                if not p:
                    continue
                if p[0] == "v":
                    vs.append(tuple(map(float, p[1:4])))
                elif p[0] == "vt":
                    vts.append((float(p[1]), float(p[2])))
                elif p[0] == "vn":
                    vns.append(tuple(map(float, p[1:4])))
                elif p[0] == "usemtl":
                    cur_mat = " ".join(p[1:])
                    if cur_mat not in mat_names:
                        mat_names.append(cur_mat)
                elif p[0] == "f":
                    ids = []
                    for tok in p[1:]:
                        q = (tok.split("/") + ["", ""])[:3]
                        key = (int(q[0]), int(q[1]) if q[1] else 0, int(q[2]) if q[2] else 0)
                        if key not in index:
                            index[key] = len(verts)
                            verts.append(key)
                        ids.append(index[key])
                    for k in range(1, len(ids) - 1):
                        tris.append((ids[0], ids[k], ids[k + 1]))
                        tri_mat.append(cur_mat)
        fix = lambda i, n: i - 1 if i > 0 else n + i
        pos = [vs[fix(a, len(vs))] for a, _, _ in verts]
        uv = [vts[fix(b, len(vts))] if b else (0.0, 0.0) for _, b, _ in verts]
        if all(c for _, _, c in verts):
            nrm = [vns[fix(c, len(vns))] for _, _, c in verts]
        else:
            nrm = np.zeros((len(pos), 3), F)
            P = np.asarray(pos, F)
            for a, b, c in tris:
                n = np.cross(P[b] - P[a], P[c] - P[a])
                nrm[a] += n; nrm[b] += n; nrm[c] += n
            nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
        groups = None
        if len(mat_names) > 1 or (mat_names and None not in tri_mat):
            order = [n for n in ([None] if None in tri_mat else []) + mat_names]
            new_tris, groups = [], []
            for n in order:
                part = [t for t, m_ in zip(tris, tri_mat) if m_ == n]
                if part:
                    groups.append((n, len(new_tris), len(part)))
                    new_tris += part
            tris = new_tris
        m = Mesh(pos, nrm, uv, tris, cullback=cullback, name=alias or str(path))
        if groups:
            m.groups = groups
            m.materials = [{"NAME": g[0]} for g in groups if g[0] is not None]
        return m


class Camera:
    def __init__(self, position, target, up=(0, 1, 0), fov=45.0, near=0.1, far=100.0):
        self.position, self.target, self.up = position, target, up
        self.fov, self.near, self.far = fov, near, far

    def view(self):
        return look_at(self.position, self.target, self.up)

    def projection(self, aspect):
        return perspective(math.radians(self.fov), aspect, self.near, self.far)


class RenderOrder:
    OPAQUE, TRANSPARENT, OVERLAY = 1000, 2000, 3000


def build_mip_chain(tex):
    """(H, W, 4) float32 image -> (flat float32 array: level 0 followed by its mip levels, number of levels).  Level k + 1 is the
    2x2 box filter of level k, size max(1, w >> 1) x max(1, h >> 1) (an odd last row / column is dropped), down to 1 x 1 -- what
    glGenerateMipmap produces for the reference's file textures (engine/static/texture/texture.py:276-289).  Summation order fixed:
    ((a + b) + (c + d)) * 0.25 in fp32."""
    lvl = np.ascontiguousarray(np.asarray(tex, dtype=F))
    if lvl.ndim != 3 or lvl.shape[-1] != 4:
        raise ValueError("texture must be (H, W, 4)")
    out = [lvl.reshape(-1)]
    while lvl.shape[0] > 1 or lvl.shape[1] > 1:
        h, w = lvl.shape[:2]
        h2, w2 = max(h >> 1, 1), max(w >> 1, 1)
        y0 = np.arange(h2) * 2 if h > 1 else np.zeros(1, int)     # (a dimension that is already 1 texel is averaged with itself)
        x0 = np.arange(w2) * 2 if w > 1 else np.zeros(1, int)
        y1, x1 = (y0 + 1 if h > 1 else y0), (x0 + 1 if w > 1 else x0)
        a, b = lvl[y0][:, x0], lvl[y0][:, x1]
        c, d = lvl[y1][:, x0], lvl[y1][:, x1]
        lvl = (((a + b) + (c + d)) * F(0.25)).astype(F)
        out.append(lvl.reshape(-1))
    return np.concatenate(out), len(out)


class DrawTask:
    """One (mesh x material) G-buffer task.  render_mode: 0 NORMAL, 1 BAKED, 2 BAKING (engine/static/enums.py:174-238).
    diffuse_filter: "nearest" (one level) or "trilinear" (mip chain built on first use; the reference's default for file textures,
    engine/static/texture/texture.py:57-60) -- colour plane only, ids / noise / corr-map lookups are NEAREST in the reference too."""

    def __init__(self, mesh, model, sprite_id=1, material_id=1, render_mode=0, corrmap_k=3, use_texcoord_id=False,
                 id_size=(512, 512), noise_tex=None, diffuse_tex=None, corrmap=None, order=RenderOrder.OPAQUE,
                 has_vertex_color=False, normal_tex=None, diffuse_filter="nearest"):
        if diffuse_filter not in ("nearest", "trilinear"):
            raise ValueError("diffuse_filter must be 'nearest' or 'trilinear'")
        self.diffuse_filter = diffuse_filter
        self.mesh, self.model = mesh, np.asarray(model, F)
        self.sprite_id, self.material_id, self.render_mode, self.corrmap_k = sprite_id, material_id, render_mode, corrmap_k
        self.use_texcoord_id, self.id_size = use_texcoord_id, id_size
        self.noise_tex, self.diffuse_tex, self.corrmap = noise_tex, diffuse_tex, corrmap
        self.normal_tex = normal_tex                      # tangent-space normal map (frag.glsl:114-123); needs mesh tangents
        self.order, self.has_vertex_color = order, has_vertex_color


def draw_params(task, view, proj):
    """-> dict of the uniforms of one task (shared by the HIP path and the oracle harness)"""
    MV = matmul(view, task.model)
    return dict(MV=MV.reshape(-1), MV_IT=inverse_transpose(MV).reshape(-1), P=np.asarray(proj, F).reshape(-1),
                depth_test=0 if RenderOrder.TRANSPARENT <= task.order < RenderOrder.OVERLAY else 1)


class GBuffer:
    """The six MRT planes + depth as HBM tensors (formats of renderManager.py:206-391)."""

    def __init__(self, W, H, device="cuda"):
        self.W, self.H = W, H
        dev = torch.device(device)
        self.color = torch.zeros(H, W, 4, dtype=torch.float16, device=dev)
        self.id = torch.zeros(H, W, 4, dtype=torch.int32, device=dev)
        self.pos = torch.zeros(H, W, 3, dtype=torch.float32, device=dev)
        self.normal_depth = torch.zeros(H, W, 4, dtype=torch.float16, device=dev)
        self.noise = torch.zeros(H, W, 4, dtype=torch.float16, device=dev)
        self.canny = torch.zeros(H, W, 3, dtype=torch.float32, device=dev)
        self.zbuf = torch.ones(H, W, dtype=torch.float32, device=dev)
        g = L.GBuffer()
        g.color, g.id, g.pos, g.normal_depth = O._p(self.color), O._p(self.id), O._p(self.pos), O._p(self.normal_depth)
        g.noise, g.canny, g.zbuf, g.W, g.H = O._p(self.noise), O._p(self.canny), O._p(self.zbuf), W, H
        self.c = g
        self._scratch = None

    def clear(self):
        L.check(L.lib().sr_gbuffer_clear(C.byref(self.c), O.stream_ptr()))

    def draw(self, task, view, proj):
        dev = self.color.device
        if task.normal_tex is not None and task.mesh.tangents is None:       # TBN branch: tangents made on first use (mesh.py:337)
            task.mesh.compute_tangents()
        md = task.mesh.device(dev)
        up = draw_params(task, view, proj)
        d = L.Draw()
        d.pos, d.normal, d.uv, d.color, d.vertex_id, d.tris = (O._p(md["pos"]), O._p(md["normal"]), O._p(md["uv"]),
                                                               O._p(md["color"]), O._p(md["vid"]), O._p(md["tris"]))
        d.nv, d.nt = task.mesh.positions.shape[0], task.mesh.tris.shape[0]
        d.MV[:] = up["MV"].tolist(); d.MV_IT[:] = up["MV_IT"].tolist(); d.P[:] = up["P"].tolist()
        d.sprite_id, d.material_id, d.corrmap_k = task.sprite_id, task.material_id, task.corrmap_k
        d.use_texcoord_id, d.render_mode = int(task.use_texcoord_id), task.render_mode
        d.has_vertex_color, d.depth_test, d.cull_back = int(task.has_vertex_color), up["depth_test"], int(task.mesh.cullback)
        d.id_w, d.id_h = task.id_size
        def dev_tex(name, dtype):
            t = getattr(task, name)
            if t is None:
                return None
            if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):      # textures must be HBM resident
                cache = task.__dict__.setdefault("_dev_tex", {})
                if name not in cache or cache[name][0] is not t:
                    cache[name] = (t, t.to(dev, dtype).contiguous())
                t = cache[name][1]
            if t.dim() != 3 or t.shape[-1] != 4:
                raise ValueError(f"{name} must be (H, W, 4)")
            return t
        ntex, dtex = dev_tex("noise_tex", torch.float16), dev_tex("diffuse_tex", torch.float32)
        nmtex = dev_tex("normal_tex", torch.float32)
        if nmtex is not None:
            d.tangent, d.bitangent = O._p(md["tangent"]), O._p(md["bitangent"])
            d.normal_tex, d.normal_h, d.normal_w = O._p(nmtex), nmtex.shape[0], nmtex.shape[1]
        if ntex is not None:
            d.noise_tex, d.noise_h, d.noise_w = O._p(ntex), ntex.shape[0], ntex.shape[1]
        if dtex is not None:
            d.diffuse_tex, d.diffuse_h, d.diffuse_w = O._p(dtex), dtex.shape[0], dtex.shape[1]
            if task.diffuse_filter == "trilinear":
                cache = task.__dict__.setdefault("_dev_tex", {})
                if "diffuse_mips" not in cache or cache["diffuse_mips"][0] is not task.diffuse_tex:
                    chain, levels = build_mip_chain(task.diffuse_tex.detach().float().cpu().numpy())
                    cache["diffuse_mips"] = (task.diffuse_tex, torch.from_numpy(chain).to(dev), levels)
                d.diffuse_tex, d.diffuse_levels = O._p(cache["diffuse_mips"][1]), cache["diffuse_mips"][2]
        if task.corrmap is not None:
            d.corrmap_tex, d.corr_h, d.corr_w = O._p(task.corrmap._values), task.corrmap.height, task.corrmap.width
        need = L.lib().sr_raster_scratch_bytes(d.nt, self.W, self.H)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=dev)
        L.check(L.lib().sr_raster_draw(C.byref(d), C.byref(self.c), O._p(self._scratch), self._scratch.numel(), O.stream_ptr()))

    def render_identical(self, tasks, camera, scratch=None):
        """identical-G-buffer tasks (renderManager.py:95-133, 954-959): every task is drawn ALONE into a cleared scratch G-buffer
        and merged into this one by depth, so each object's planes come out as if nothing occluded it except closer objects'
        whole-pixel wins; ``self`` must have been cleared (or hold earlier merges)"""
        view, proj = camera.view(), camera.projection(self.W / self.H)
        tmp = scratch if scratch is not None else GBuffer(self.W, self.H, device=self.color.device)
        for t in sorted(tasks, key=lambda t: t.order):
            tmp.clear()
            tmp.draw(t, view, proj)
            L.check(L.lib().sr_gbuffer_depth_merge(C.byref(self.c), C.byref(tmp.c), O.stream_ptr()))
        return tmp

    def display(self, is_baking=False, enableGammaCorrection=False, enableHDR=False, gamma=1.0, exposure=1.0, saturation=1.0,
                brightness=1.0, contrast=1.0):
        """defer render + post process (default_defer_render / default_post_process shaders; Engine kwargs enableHDR,
        enableGammaCorrection, gamma, exposure, saturation, brightness, contrast, engine.py:76-92) -> (H, W, 4) fp32 RGBA"""
        out = torch.empty(self.H, self.W, 4, dtype=torch.float32, device=self.color.device)
        L.check(L.lib().sr_defer_post(O._p(self.color), O._p(self.id), O._p(out), self.W, self.H, int(is_baking),
                                      int(enableGammaCorrection), int(enableHDR), float(gamma), float(exposure), float(saturation),
                                      float(brightness), float(contrast), O.stream_ptr()))
        return out

    def render(self, tasks, camera):
        """RenderManager.on_frame_run's G-buffer part: clear, tasks sorted by order (stable), depth test off for the
        TRANSPARENT queue (renderManager.py:508-513)."""
        view, proj = camera.view(), camera.projection(self.W / self.H)
        self.clear()
        for t in sorted(tasks, key=lambda t: t.order):
            self.draw(t, view, proj)
