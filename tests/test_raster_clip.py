"""Near-plane handling of the G-buffer rasteriser (oracle/raster_ref.c header): a triangle with vertices behind the eye is
rasterised in homogeneous coordinates.  No GL output exists for such triangles in the reference, so the check is geometric: the
same surface, clipped by hand in model space against a plane just in front of the near plane and drawn through the ordinary
(all w > 0) path, must cover the same pixels and carry the same view-space positions, uvs and depth."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def ground_scene(W, H):
    """a 100 x 100 ground quad under a camera that stands on it: it runs through the near plane and far behind the eye"""
    from stable_renderer_amd import scene as S
    cam = S.Camera((0.0, 1.0, 0.0), (0.0, 0.6, -3.0), fov=60.0, near=0.1, far=100.0)
    quad = S.Mesh.Plane(2)
    quad.cullback = False
    model = S.scale(100.0)
    return cam, quad, model


def clip_mesh_in_front(mesh, MV, zmax):
    """Sutherland-Hodgman of every triangle against view z <= zmax (in front of the eye), attributes interpolated linearly in
    model space; -> (positions, normals, uvs, tris) of a mesh whose triangles all lie in front"""
    P, N, U = mesh.positions.astype(np.float64), mesh.normals.astype(np.float64), mesh.uvs.astype(np.float64)
    M = np.array(MV, np.float64).reshape(4, 4).T                   # column-major storage -> row-major matrix
    vz = lambda p: (M @ np.array([p[0], p[1], p[2], 1.0]))[2]
    pos, nrm, uv, tris = [], [], [], []
    for tri in mesh.tris:
        poly = [(P[i], N[i], U[i]) for i in tri]
        out = []
        for a in range(len(poly)):
            cur, nxt = poly[a], poly[(a + 1) % len(poly)]
            zc, zn = vz(cur[0]), vz(nxt[0])
            if zc <= zmax:
                out.append(cur)
            if (zc <= zmax) != (zn <= zmax):
                t = (zmax - zc) / (zn - zc)
                out.append(tuple(c + (n - c) * t for c, n in zip(cur, nxt)))
        for k in range(1, len(out) - 1):
            base = len(pos)
            for v in (out[0], out[k], out[k + 1]):
                pos.append(v[0]); nrm.append(v[1]); uv.append(v[2])
            tris.append((base, base + 1, base + 2))
    return np.array(pos), np.array(nrm), np.array(uv), np.array(tris)


def oracle_render(W, H, cam, mesh, model):
    from stable_renderer_amd import scene as S
    import raster_ref as R
    ref = R.GBufferRef(W, H)
    ref.clear()
    task = S.DrawTask(mesh, model, sprite_id=3, material_id=4, render_mode=0, order=999.5)
    ref.draw(task, S.draw_params(task, cam.view(), cam.projection(W / H)))
    return ref


def test_homogeneous_path_equals_a_hand_clipped_mesh():
    from stable_renderer_amd import scene as S
    W, H = 160, 120
    cam, quad, model = ground_scene(W, H)
    a = oracle_render(W, H, cam, quad, model)                       # crosses the near plane: homogeneous path
    MV = S.matmul(cam.view(), model)
    pos, nrm, uv, tris = clip_mesh_in_front(quad, MV, -0.1 * 1.0001)   # view z <= -near (a hair inside, as GL's clip leaves it)
    clipped = S.Mesh(pos, nrm, uv, tris, cullback=False)
    b = oracle_render(W, H, cam, clipped, model)                    # every triangle in front: the pinned fixed-point path
    ca, cb = a.id[..., 0] != 0, b.id[..., 0] != 0
    assert 0.3 < ca.mean() < 0.7                                    # the ground fills the lower part of the view
    assert (ca != cb).sum() <= 4, (ca != cb).sum()                  # inclusive edges vs the top-left rule on the shared diagonal only
    both = ca & cb
    assert np.allclose(a.pos[both], b.pos[both], rtol=2e-4, atol=2e-4)
    assert np.allclose(a.zbuf[both], b.zbuf[both], atol=2e-6)
    # rows at the bottom of the image look at ground closer than the near plane could ever show if w <= 0 vertices were dropped
    assert ca[-1].all() and not ca[0].any()
    assert (a.zbuf[ca] >= 0).all() and (a.zbuf[ca] <= 1).all()


def test_triangle_entirely_behind_the_eye_draws_nothing_and_near_clip_discards_fragments():
    from stable_renderer_amd import scene as S
    W = H = 64
    cam = S.Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), fov=60.0, near=0.5, far=10.0)
    tri = lambda z: S.Mesh([(-1, -1, z), (1, -1, z), (0, 1, z)], [(0, 0, 1)] * 3, [(0, 0), (1, 0), (0, 1)], [(0, 1, 2)], cullback=False)
    behind = oracle_render(W, H, cam, tri(2.0), S.scale(1.0))
    assert not (behind.id[..., 0] != 0).any()
    before_near = oracle_render(W, H, cam, tri(-0.25), S.scale(1.0))    # in front of the eye (w > 0) but nearer than the near plane
    assert not (before_near.id[..., 0] != 0).any()
    visible = oracle_render(W, H, cam, tri(-2.0), S.scale(1.0))
    assert (visible.id[..., 0] != 0).any()
