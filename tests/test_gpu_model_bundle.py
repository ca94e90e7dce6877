"""Operator-level C ABI (include/sr_hip.h: sr_model_load / sr_unet_forward / sr_vae_decode): a UNet and a VAE decoder lowered and
tuned by the Python host are exported as model bundles (bundle.py) and then run by a CHILD PROCESS THAT IMPORTS NEITHER torch NOR
this package -- ctypes on libsr_hip.so and numpy only, the stand-in for a C++ / Go / Rust host -- which must reproduce the Python
path's outputs bit for bit (same kernels, same pinned tiles; UNetModel.forward openaimodel.py:841-946, VAE.decode sd.py:329-346)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

CHILD = r'''
import ctypes as C, sys, numpy as np
assert "torch" not in sys.modules
lib = C.CDLL(sys.argv[1])
vp = C.c_void_p
lib.sr_last_error.restype = C.c_char_p
lib.sr_model_load.argtypes = [C.c_char_p, C.POINTER(vp)]
lib.sr_unet_forward.argtypes = [vp, vp, vp, vp, vp, vp]
lib.sr_vae_decode.argtypes = [vp, vp, vp, vp]
lib.sr_model_write.argtypes = [vp, C.c_char_p, vp, vp]
lib.sr_model_io.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int64)]
lib.sr_model_free.argtypes = [vp]
def ck(rc):
    assert rc == 0, lib.sr_last_error().decode()
d = np.load(sys.argv[2])
m = vp()
ck(lib.sr_model_load(sys.argv[3].encode(), C.byref(m)))
x, t, ctx = np.ascontiguousarray(d["x"]), np.ascontiguousarray(d["t"]), np.ascontiguousarray(d["ctx"])
out = np.empty_like(x)
nb = C.c_int64()
ck(lib.sr_model_io(m, b"ctx", None, C.byref(nb)))
assert nb.value == ctx.nbytes, (nb.value, ctx.nbytes)
if "inject" in d.files:
    inj = np.ascontiguousarray(d["inject"].astype(np.int32))
    ck(lib.sr_model_write(m, b"inject", inj.ctypes.data_as(vp), None))
ck(lib.sr_unet_forward(m, x.ctypes.data_as(vp), t.ctypes.data_as(vp), ctx.ctypes.data_as(vp), out.ctypes.data_as(vp), None))
ck(lib.sr_device_sync())
out2 = np.empty_like(x)                      # second evaluation, same prompt: ctx = NULL keeps the projected K / V
ck(lib.sr_unet_forward(m, x.ctypes.data_as(vp), t.ctypes.data_as(vp), None, out2.ctypes.data_as(vp), None))
ck(lib.sr_device_sync())
ck(lib.sr_model_free(m))
v = vp()
ck(lib.sr_model_load(sys.argv[4].encode(), C.byref(v)))
z = np.ascontiguousarray(d["z"])
img = np.empty(tuple(d["img_shape"]), np.float32)
ck(lib.sr_vae_decode(v, z.ctypes.data_as(vp), img.ctypes.data_as(vp), None))
ck(lib.sr_device_sync())
ck(lib.sr_model_free(v))
bad = vp()
assert lib.sr_model_load(b"/nonexistent.srm", C.byref(bad)) != 0 and b"cannot open" in lib.sr_last_error()
np.savez(sys.argv[5], out=out, out2=out2, img=img)
'''


def _sd(name, seed):
    from stable_renderer_amd import synth
    with open(os.path.join(GOLD, name)) as f:
        k = json.load(f)
    return synth.synth_state_dict([(n, tuple(s)) for n, s in k["names_shapes"]], seed=seed, norm_names=k["norm_names"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_unet_and_vae_bundles_run_from_a_process_without_torch(tmp_path, dtype):
    from stable_renderer_amd import _lib, bundle
    from stable_renderer_amd.unet import SD15_CFG, UNet
    from stable_renderer_amd.vae import VAEDecoder
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    net = UNet(_sd("unet_tiny_keys.json", 1), cfg, dtype=dtype)
    g = torch.Generator().manual_seed(5)
    B, h, w = 4, 16, 16
    x, t, ctx = torch.randn(B, 4, h, w, generator=g), torch.tensor([981.0] * B), torch.randn(B, 77, 64, generator=g)
    p = net.build(B, h, w, inject_idx=[2], n_ctx=77)
    p["x"].copy_(x)
    p["t"].copy_(t)
    p["ctx"].copy_(ctx.to(dtype))
    p["prologue"].run()
    p["step"].run()
    torch.cuda.synchronize()
    ref = p["out"].cpu().numpy().copy()
    info = bundle.export_unet(str(tmp_path / "unet.srm"), net, p)
    assert info["saved"] > 100                                 # the packed weights travelled
    dec = VAEDecoder(_sd("vae_dec_keys.json", 2), dtype=dtype)
    z = torch.randn(2, 4, 8, 8, generator=g)
    vp_ = dec.build(2, 8, 8)
    vp_["z"].copy_(z)
    vp_["plan"].run()
    torch.cuda.synchronize()
    ref_img = vp_["img"].cpu().numpy().copy()
    bundle.export_vae(str(tmp_path / "vae.srm"), dec, vp_)
    np.savez(tmp_path / "in.npz", x=x.numpy(), t=t.numpy(), ctx=ctx.to(dtype).numpy(), inject=np.array([2]), z=z.numpy(),
             img_shape=np.array(ref_img.shape))
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, str(child), _lib.LIB_PATH, str(tmp_path / "in.npz"), str(tmp_path / "unet.srm"),
                        str(tmp_path / "vae.srm"), str(tmp_path / "out.npz")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    o = np.load(tmp_path / "out.npz")
    assert np.array_equal(o["out"], ref) and np.array_equal(o["out2"], ref)     # same kernels, same tiles: bit for bit
    assert np.array_equal(o["img"], ref_img)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.1
