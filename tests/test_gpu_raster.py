"""HIP tiled rasterizer vs the C oracle (oracle/raster_ref.c): every G-buffer plane BIT EXACT (ids, fp16 colour /
normal+depth / noise bits, fp32 pos / canny / depth)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(W, H, frame, k=6):
    from stable_renderer_amd import scene as S
    cam = S.Camera((0, 0.68, 2.3), (0, 0.68, 0))
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(64, 64, 4, generator=g).half()
    diffuse = torch.rand(32, 32, 4, generator=g)
    diffuse[..., 3] = (diffuse[..., 3] > 0.3).float() * 0.5 + 0.5          # some alpha < 1 -> blend branch
    rot = S.rotate_y(frame * 1.0)
    sphere = S.Mesh.Sphere(32)
    m1 = S.matmul(S.translate((0, 0.68, 0)), S.matmul(rot, S.scale(0.70)))
    m2 = S.matmul(S.translate((0, 0.68, 0)), S.matmul(rot, S.scale(0.85)))
    plane = S.Mesh.Plane(4)
    plane.colors = np.random.RandomState(1).rand(plane.positions.shape[0], 3).astype(np.float32)
    plane._dev = None
    mp = S.matmul(S.translate((0, 0.0, 0)), S.scale(3.0))
    tasks = [
        S.DrawTask(plane, mp, sprite_id=3, material_id=4, render_mode=0, has_vertex_color=True, order=999.5),
        S.DrawTask(sphere, m1, sprite_id=1, material_id=1, render_mode=0, diffuse_tex=diffuse, noise_tex=noise, order=999.7),
        S.DrawTask(sphere, m2, sprite_id=2, material_id=2, render_mode=2, corrmap_k=k, use_texcoord_id=True,
                   id_size=(W, H), noise_tex=noise, order=2000.3),
    ]
    return cam, tasks


@pytest.mark.parametrize("size,frame", [((512, 512), 0), ((512, 512), 37), ((200, 136), 5)])
def test_raster_bit_exact(size, frame):
    from stable_renderer_amd import scene as S
    import raster_ref as R
    W, H = size
    cam, tasks = _scene(W, H, frame)
    gb = S.GBuffer(W, H)
    gb.render(tasks, cam)
    torch.cuda.synchronize()
    ref = R.GBufferRef(W, H)
    ref.clear()
    view, proj = cam.view(), cam.projection(W / H)
    for t in sorted(tasks, key=lambda t: t.order):
        ref.draw(t, S.draw_params(t, view, proj),
                 noise_tex=None if t.noise_tex is None else t.noise_tex.numpy().view(np.uint16),
                 diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.numpy())
    cov = (ref.id[..., 0] != 0).mean()
    assert 0.15 < cov < 0.95, cov                                            # the scene really covers pixels
    assert (ref.id[..., 2] == 2048).any() and (ref.id[..., 2] < 36).any()
    assert np.array_equal(gb.id.cpu().numpy(), ref.id)
    assert np.array_equal(gb.zbuf.cpu().numpy().view(np.uint32), ref.zbuf.view(np.uint32))
    assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), ref.color)
    assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), ref.normal_depth)
    assert np.array_equal(gb.noise.cpu().numpy().view(np.uint16), ref.noise)
    assert np.array_equal(gb.pos.cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    assert np.array_equal(gb.canny.cpu().numpy(), ref.canny)


def test_raster_baked_mode_reads_corrmap():
    """render_mode 1: colour comes from the corr-map 2D array at (uv.y, uv.x, map_index); alpha-0 texels keep the
    previous pixel including its ids (frag.glsl:185-236)."""
    from stable_renderer_amd import scene as S
    from stable_renderer_amd.corrmap import CorrespondMap
    import raster_ref as R
    W = H = 128
    cam = S.Camera((0, 0, 2.5), (0, 0, 0))
    cm = CorrespondMap(k=3, height=32, width=32)
    g = torch.Generator().manual_seed(5)
    vals = torch.rand(9, 32 * 32, 4, generator=g)
    vals[..., 3] = (vals[..., 3] > 0.4).float()
    cm._values.copy_(vals.half())
    sphere = S.Mesh.Sphere(16)
    bg = S.DrawTask(S.Mesh.Plane(1), S.matmul(S.translate((0, 0, -1)), S.matmul(S.scale(6.0), np.eye(4, dtype=np.float32))),
                    sprite_id=9, material_id=9, order=999.0)
    bg.mesh.normals[:] = (0, 0, 1)
    bg.mesh.positions[:, [1, 2]] = bg.mesh.positions[:, [2, 1]]           # stand the plane up facing +z
    bg.mesh.cullback = False
    baked = S.DrawTask(sphere, np.eye(4, dtype=np.float32), sprite_id=2, material_id=2, render_mode=1, corrmap_k=3,
                       use_texcoord_id=True, id_size=(32, 32), corrmap=cm, order=2000.5)
    gb = S.GBuffer(W, H)
    gb.render([bg, baked], cam)
    torch.cuda.synchronize()
    ref = R.GBufferRef(W, H)
    ref.clear()
    view, proj = cam.view(), cam.projection(1.0)
    ref.draw(bg, S.draw_params(bg, view, proj))
    ref.draw(baked, S.draw_params(baked, view, proj), corrmap_tex=cm._values.cpu().numpy().view(np.uint16), corr_hw=(32, 32))
    assert (ref.id[..., 0] == 2).any() and (ref.id[..., 0] == 9).any()
    assert np.array_equal(gb.id.cpu().numpy(), ref.id)
    assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), ref.color)
    assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), ref.normal_depth)


@pytest.mark.parametrize("fr", (0, 11, 25, 41))
def test_hip_rasterizer_matches_reference_dump(gold, fr):
    """the HIP rasterizer against the G-buffers the reference's OpenGL pass dumped (fixture raster_pin.npz, see
    tests/test_raster_pin.py): coverage IoU, view-space pos, texcoord ids within one texel"""
    from stable_renderer_amd import scene as S
    from test_raster_pin import check_against_reference_dump
    d = gold("raster_pin")
    m = S.Mesh.Sphere(32)
    m_vu = S.Mesh(m.positions, m.normals, m.uvs[:, ::-1].copy(), m.tris)
    MV, P = d[f"MV_{fr}"], d["P"]
    out = []
    for mesh in (m, m_vu):
        gb = S.GBuffer(512, 512)
        gb.clear()
        t = S.DrawTask(mesh, MV, use_texcoord_id=True, id_size=(int(d["tex"]), 0))
        gb.draw(t, np.eye(4, dtype=np.float32), P)               # draw_params: MV = view * model with view = I
        torch.cuda.synchronize()
        out.append(gb)
    cov = (out[0].id[..., 0] != 0).cpu().numpy()
    check_against_reference_dump(d, fr, cov, out[0].pos.cpu().numpy(), out[0].id[..., 3].cpu().numpy(), out[1].id[..., 3].cpu().numpy())


@pytest.mark.parametrize("fr", (8, 31, 50))
def test_hip_rasterizer_normal_and_depth_planes_match_reference_dump(gold, fr):
    """the HIP rasterizer's normal + depth plane against the reference's own normal/ and depth/ dumps of the same sphere scene
    (fixture raster_pin_planes.npz): packed normals within 1/255, reversed min-max-normalised depth, empty noise / canny planes"""
    from stable_renderer_amd import scene as S
    from test_raster_pin import check_planes_against_reference_dump
    d = gold("raster_pin_planes")
    gb = S.GBuffer(512, 512)
    gb.clear()
    t = S.DrawTask(S.Mesh.Sphere(32), d[f"MV_{fr}"], use_texcoord_id=True, id_size=(1024, 0))
    gb.draw(t, np.eye(4, dtype=np.float32), d["P"])
    torch.cuda.synchronize()
    cov = (gb.id[..., 0] != 0).cpu().numpy()
    check_planes_against_reference_dump(d, fr, cov, gb.normal_depth.float().cpu().numpy())
    assert not bool(gb.noise.any()) and not bool(gb.canny.any())


@pytest.mark.parametrize("frame", (0, 5))
def test_boatlike_obj_through_mesh_load_bit_exact(frame):
    """BASELINE config 3's geometry path: a Blender-style OBJ (v/vt/vn corners, quads + n-gons to fan-triangulate, relative
    indices; tests/golden/boatlike.obj, same attribute set as the reference's boat.obj) through Mesh.Load, drawn with the boat
    scene's camera + corr-map proxy sphere at 512^2: every plane bit exact vs oracle/raster_ref.c"""
    import os
    import raster_ref as R
    from stable_renderer_amd import scene as S
    from stable_renderer_amd.pipeline import BoatScene
    sc = BoatScene(os.path.join(os.path.dirname(__file__), "golden", "boatlike.obj"), 512, 512)
    assert sc.mesh.tris.shape == (1450, 3) and sc.mesh.positions.shape == (916, 3)
    gb = S.GBuffer(512, 512)
    tasks = sc.tasks(frame)
    gb.render(tasks, sc.camera)
    torch.cuda.synchronize()
    ref = R.GBufferRef(512, 512)
    ref.clear()
    view, proj = sc.camera.view(), sc.camera.projection(1.0)
    for t in sorted(tasks, key=lambda t: t.order):
        ref.draw(t, S.draw_params(t, view, proj),
                 noise_tex=None if t.noise_tex is None else t.noise_tex.cpu().numpy().view(np.uint16),
                 diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.cpu().numpy())
    boat_px = int((ref.color[..., 3] != 0).sum())
    assert boat_px > 10000 and (ref.id[..., 0] == 2).sum() > 100000            # the boat and its proxy are on screen
    assert np.array_equal(gb.id.cpu().numpy(), ref.id)
    assert np.array_equal(gb.zbuf.cpu().numpy().view(np.uint32), ref.zbuf.view(np.uint32))
    assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), ref.color)
    assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), ref.normal_depth)
    assert np.array_equal(gb.noise.cpu().numpy().view(np.uint16), ref.noise)
    assert np.array_equal(gb.pos.cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    assert np.array_equal(gb.canny.cpu().numpy(), ref.canny)


def test_normal_map_tbn_branch_bit_exact():
    """hasNormalTex branch of the fragment shader (frag.glsl:114-123): per-vertex tangent space (Mesh.compute_tangents), a tangent-
    space normal texture, view normal = normalize(MV_IT * normalize(TBN * normalize(tex*2-1))): normal+depth and canny planes (and
    the corr-map index they drive) bit exact vs oracle/raster_ref.c, and different from the mesh-normal result"""
    import os
    import raster_ref as R
    from stable_renderer_amd import scene as S
    W = H = 256
    cam = S.Camera((0, 3, -3), (0, 0, 0))
    mesh = S.Mesh.Load(os.path.join(os.path.dirname(__file__), "golden", "boatlike.obj"))
    g = torch.Generator().manual_seed(8)
    nmap = torch.rand(32, 32, 4, generator=g)
    nmap[..., 2] = 0.6 + 0.4 * nmap[..., 2]                              # mostly +z, as tangent-space normal maps are
    diffuse = torch.rand(16, 16, 4, generator=g)
    diffuse[..., 3] = 1.0
    model = S.rotate_y(20.0)

    def render(with_map, mode):
        t = S.DrawTask(mesh, model, sprite_id=3, material_id=2, render_mode=mode, corrmap_k=6, use_texcoord_id=True, id_size=(W, H),
                       diffuse_tex=diffuse, normal_tex=nmap if with_map else None, order=999.0)
        gb = S.GBuffer(W, H)
        gb.render([t], cam)
        torch.cuda.synchronize()
        ref = R.GBufferRef(W, H)
        ref.clear()
        if with_map:
            mesh.compute_tangents()
        ref.draw(t, S.draw_params(t, cam.view(), cam.projection(1.0)), diffuse_tex=diffuse.numpy(),
                 normal_tex=nmap.numpy() if with_map else None)
        assert (ref.id[..., 0] == 3).sum() > 5000
        assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), ref.normal_depth)
        assert np.array_equal(gb.id.cpu().numpy(), ref.id)
        assert np.array_equal(gb.canny.cpu().numpy(), ref.canny)
        assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), ref.color)
        return ref
    # NORMAL mode: the mapped normal lands in the normal+depth plane
    assert not np.array_equal(render(True, 0).normal_depth, render(False, 0).normal_depth)
    # BAKING mode keeps normal+depth of the snapshot but the corr-map index (id plane) follows the mapped normal
    assert not np.array_equal(render(True, 2).id[..., 2], render(False, 2).id[..., 2])


def test_identical_gbuffer_merge_and_display_pass():
    """AddIdenticalGBufferTask semantics (each object alone, merged by depth, renderManager.py:95-133) and the defer + post-process
    display pass (default_defer_render / default_post_process shaders) against their numpy restatements"""
    import raster_ref as R
    from stable_renderer_amd import scene as S
    W = H = 160
    cam, tasks = _scene(W, H, 3)
    tasks = tasks[:2]                                           # the vertex-coloured plane and the textured sphere
    gb = S.GBuffer(W, H)
    gb.clear()
    gb.render_identical(tasks, cam)
    torch.cuda.synchronize()
    view, proj = cam.view(), cam.projection(1.0)
    acc = R.GBufferRef(W, H)
    acc.clear()
    acc.zbuf[:] = 1.0
    for t in sorted(tasks, key=lambda t: t.order):
        one = R.GBufferRef(W, H)
        one.clear()
        one.draw(t, S.draw_params(t, view, proj),
                 noise_tex=None if t.noise_tex is None else t.noise_tex.numpy().view(np.uint16),
                 diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.numpy())
        won = R.depth_merge(acc, one)
        assert won.any()
    assert np.array_equal(gb.id.cpu().numpy(), acc.id)
    assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), acc.color)
    assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), acc.normal_depth)
    assert np.array_equal(gb.pos.cpu().numpy().view(np.uint32), acc.pos.view(np.uint32))
    assert (acc.id[..., 0] == 1).any() and (acc.id[..., 0] == 3).any()
    # display pass on a baking frame: rainbow tint of AI ids + post process
    cam, tasks = _scene(W, H, 3)
    gb.render(tasks, cam)
    ref = R.GBufferRef(W, H)
    ref.clear()
    for t in sorted(tasks, key=lambda t: t.order):
        ref.draw(t, S.draw_params(t, view, proj),
                 noise_tex=None if t.noise_tex is None else t.noise_tex.numpy().view(np.uint16),
                 diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.numpy())
    for kw in (dict(is_baking=True), dict(is_baking=False, enableGammaCorrection=True, gamma=2.2, exposure=1.3, saturation=0.8,
                                           brightness=1.1, contrast=1.2, enableHDR=True)):
        img = gb.display(**kw).cpu().numpy()
        want = R.defer_post(ref.color, ref.id, is_baking=kw.get("is_baking", False), gamma_on=kw.get("enableGammaCorrection", False),
                            hdr_on=kw.get("enableHDR", False), gamma=kw.get("gamma", 1.0), exposure=kw.get("exposure", 1.0),
                            saturation=kw.get("saturation", 1.0), brightness=kw.get("brightness", 1.0), contrast=kw.get("contrast", 1.0))
        assert np.allclose(img, want, atol=2e-6, rtol=1e-5), np.abs(img - want).max()
    tinted = gb.display(is_baking=True).cpu().numpy()
    plain = gb.display(is_baking=False).cpu().numpy()
    assert (np.abs(tinted - plain).max(-1) > 0.01).sum() > 1000


def test_near_plane_crossing_triangles_bit_exact():
    """a ground quad that runs through the near plane and behind the eye (homogeneous rasterisation path) + an ordinary sphere
    standing on it: all seven planes equal to the C oracle's"""
    from stable_renderer_amd import scene as S
    import raster_ref as R
    from test_raster_clip import ground_scene
    W, H = 320, 200
    cam, quad, model = ground_scene(W, H)
    g = torch.Generator().manual_seed(3)
    diffuse = torch.rand(32, 32, 4, generator=g)
    tasks = [S.DrawTask(quad, model, sprite_id=3, material_id=4, render_mode=0, diffuse_tex=diffuse, order=999.5),
             S.DrawTask(S.Mesh.Sphere(12), S.matmul(S.translate((0.2, 0.5, -2.5)), S.scale(0.5)), sprite_id=1, material_id=1,
                        render_mode=0, order=999.7)]
    gb = S.GBuffer(W, H)
    gb.render(tasks, cam)
    torch.cuda.synchronize()
    ref = R.GBufferRef(W, H)
    ref.clear()
    view, proj = cam.view(), cam.projection(W / H)
    for t in tasks:
        ref.draw(t, S.draw_params(t, view, proj), diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.numpy())
    assert (ref.id[-1, :, 0] == 3).all() and (ref.id[..., 0] == 1).any()
    assert np.array_equal(gb.id.cpu().numpy(), ref.id)
    assert np.array_equal(gb.zbuf.cpu().numpy().view(np.uint32), ref.zbuf.view(np.uint32))
    assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), ref.color)
    assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), ref.normal_depth)
    assert np.array_equal(gb.pos.cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    assert np.array_equal(gb.canny.cpu().numpy(), ref.canny)


def _checker(n, cell=1):
    yy, xx = np.mgrid[0:n, 0:n]
    c = (((yy // cell) + (xx // cell)) & 1).astype(np.float32)
    t = np.stack([c, 1.0 - c, 0.25 + 0.5 * c, np.ones_like(c)], -1)
    return torch.from_numpy(np.ascontiguousarray(t))


@pytest.mark.parametrize("case", ["minified", "magnified", "ground_through_near_plane", "non_power_of_two"])
def test_trilinear_diffuse_texture_bit_exact_and_filtered(case):
    """diffuse_filter='trilinear' (the reference's default for file textures: GL_LINEAR_MIPMAP_LINEAR, texture.py:57-60) --
    the mip chain of scene.build_mip_chain sampled by raster_tiles<true> equals oracle/raster_ref.c tex_trilinear bit for bit
    on every plane, for minification (levels > 0), magnification (bilinear on level 0), the homogeneous (near-plane) path and a
    texture whose sizes are not powers of two; a one-texel checkerboard seen from afar comes out GREY (NEAREST gives 0 / 1)"""
    from stable_renderer_amd import scene as S
    import raster_ref as R
    W, H = 256, 192
    cam = S.Camera((0, 0.3, 3.2), (0, 0, 0))
    sphere = S.Mesh.Sphere(24)
    model = S.matmul(S.rotate_y(20.0), S.scale(1.1))
    if case == "minified":
        tex = _checker(256)
    elif case == "magnified":
        tex = torch.rand(6, 5, 4, generator=torch.Generator().manual_seed(2))
    elif case == "non_power_of_two":
        tex = torch.rand(37, 50, 4, generator=torch.Generator().manual_seed(4))
    else:
        from test_raster_clip import ground_scene
        W, H = 320, 200
        cam, sphere, model = ground_scene(W, H)                      # (a quad through the near plane and behind the eye)
        tex = _checker(128, 2)
    task = S.DrawTask(sphere, model, sprite_id=1, material_id=1, render_mode=0, diffuse_tex=tex, diffuse_filter="trilinear", order=999.5)
    near = S.DrawTask(sphere, model, sprite_id=1, material_id=1, render_mode=0, diffuse_tex=tex, order=999.5)
    gb, gn = S.GBuffer(W, H), S.GBuffer(W, H)
    gb.render([task], cam)
    gn.render([near], cam)
    torch.cuda.synchronize()
    ref = R.GBufferRef(W, H)
    ref.clear()
    ref.draw(task, S.draw_params(task, cam.view(), cam.projection(W / H)), diffuse_tex=tex.numpy(), diffuse_mips=S.build_mip_chain(tex.numpy()))
    cov = ref.id[..., 0] != 0
    assert cov.mean() > 0.1
    assert np.array_equal(gb.id.cpu().numpy(), ref.id)
    assert np.array_equal(gb.color.cpu().numpy().view(np.uint16), ref.color)
    assert np.array_equal(gb.normal_depth.cpu().numpy().view(np.uint16), ref.normal_depth)
    assert np.array_equal(gb.pos.cpu().numpy().view(np.uint32), ref.pos.view(np.uint32))
    # only the colour plane depends on the filter
    assert np.array_equal(gb.id.cpu().numpy(), gn.id.cpu().numpy()) and np.array_equal(gb.pos.cpu().numpy(), gn.pos.cpu().numpy())
    col, coln = gb.color.float().cpu().numpy(), gn.color.float().cpu().numpy()
    assert not np.array_equal(col, coln)
    if case == "minified":                                          # ~1.5 texels per pixel and more: the checker averages out
        red, redn = col[..., 0][cov], coln[..., 0][cov]
        assert set(np.unique(redn)) <= {0.0, 1.0}
        assert abs(red.mean() - 0.5) < 0.05 and red.std() < 0.5 * redn.std()
    if case == "magnified":                                         # bilinear: values strictly between texel values appear
        assert len(np.unique(col[..., 0][cov])) > 10 * len(np.unique(coln[..., 0][cov]))
