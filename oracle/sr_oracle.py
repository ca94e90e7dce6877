"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy for index/integer work, torch-CPU fp32 for the floating-point network) of the
reference algorithms on the render-then-diffuse hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package (``stable-renderer_amd/``) never does and fails loudly when the HIP library is missing.

Parity pin: every function below is checked in ``tests/test_oracle_golden.py`` against golden vectors
produced by running the reference itself on CPU (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
The rasterizer restatement lives in ``oracle/raster_ref.c`` (parity unpinned vs a GL driver, see header there).

Each function cites the reference file:line it follows (paths relative to /root/reference/source).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NON_AI_MAP_INDEX = 2048


# ------------------------------------------------------------------------------------------------------
# common_utils/math_utils.py:27-80
def calc_map_mean_std(feat: torch.Tensor, eps=1e-5):
    n, c = feat.shape[:2]
    var = feat.reshape(n, c, -1).var(dim=2) + eps          # unbiased
    half = var.dtype == torch.float16
    if half:
        var = var.float()
    std = var.sqrt().view(n, c, 1, 1)
    mean = feat.reshape(n, c, -1).mean(dim=2).view(n, c, 1, 1)
    if half:
        std = std.half()
    return mean, std


def adain(content: torch.Tensor, style: torch.Tensor, eps=1e-5, mode="NCHW"):
    if mode == "NHWC":
        content = content.permute(0, 3, 1, 2)
        style = style.permute(0, 3, 1, 2)
    sm, ss = calc_map_mean_std(style, eps)
    cm, cs = calc_map_mean_std(content, eps)
    return (content - cm) / cs * ss + sm


# common_utils/math_utils.py:86-161
def group_by_average(keys: np.ndarray, values: np.ndarray):
    """keys (M,), values (M,C) -> (avg expanded back to rows (M,C) float32, unique keys sorted)."""
    uniq, inv = np.unique(keys, return_inverse=True)
    vals = values.astype(np.float32)
    sums = np.zeros((len(uniq), vals.shape[1]), dtype=np.float32)
    np.add.at(sums, inv, vals)                              # sequential fp32 accumulation, row order
    cnt = np.zeros((len(uniq), 1), dtype=np.float32)
    np.add.at(cnt, inv, np.float32(1.0))
    return (sums / cnt)[inv], uniq


# ------------------------------------------------------------------------------------------------------
# engine/static/corrmap.py:119-126
def idmap_masks(ids: np.ndarray) -> np.ndarray:
    no_id = (ids[..., 2] == NON_AI_MAP_INDEX) | np.all(ids == 0, axis=-1)
    return no_id.astype(np.float32)


# engine/static/corrmap.py:220-280
def vertex_screen_info(ids: np.ndarray, frame_indices=None) -> np.ndarray:
    """(N,H,W,4) int -> (M,7) float32 rows (obj, mat, map_idx, vid, x/H, y/W, frame), (frame,y,x) order.
    NB the reference divides x by *height* and y by *width* (corrmap.py:243, :252)."""
    n, h, w, _ = ids.shape
    if frame_indices is None:
        frame_indices = list(range(n))
    xs = (np.arange(w, dtype=np.int64).astype(np.float32) / np.float32(h)).astype(np.float32)
    ys = (np.arange(h, dtype=np.int64).astype(np.float32) / np.float32(w)).astype(np.float32)
    out = np.empty((n, h, w, 7), dtype=np.float32)
    out[..., :4] = ids.astype(np.float32)
    out[..., 4] = xs[None, None, :]
    out[..., 5] = ys[None, :, None]
    out[..., 6] = np.asarray(frame_indices, dtype=np.int32).astype(np.float32)[:, None, None]
    flat = out.reshape(-1, 7)
    flat = flat[flat[:, 2] != NON_AI_MAP_INDEX]
    flat = flat[(flat[:, 0] != 0) | (flat[:, 1] != 0) | (flat[:, 2] != 0) | (flat[:, 3] != 0)]
    return flat


# common_utils/stable_render_utils/corresponder.py:298-376
def overlap_step(x: torch.Tensor, ids: np.ndarray, ratio: float) -> torch.Tensor:
    """One OverlapCorresponder.step_finished on latents x (N,4,h,w) fp32 (caller checks the timestep gate).
    Gather latent cell of every id pixel, mean over rows sharing vertexID, blend, scatter back (duplicate
    targets: last row wins, as index_put on CPU), then AdaIN(content=x, style=blended)."""
    vsi = vertex_screen_info(ids)
    n, c, h, w = x.shape
    sx = (vsi[:, 4] * np.float32(w)).astype(np.int32)       # fp32 multiply, truncate
    sy = (vsi[:, 5] * np.float32(h)).astype(np.int32)
    fr = vsi[:, 6].astype(np.int32)
    xc = x.float().numpy().copy()
    v = xc[fr, :, sy, sx]                                   # (M, C)
    avg, _ = group_by_average(vsi[:, 3], v)
    blended = (np.float32(1.0 - ratio) * v + np.float32(ratio) * avg).astype(np.float32)
    # sequential scatter == last writer wins
    xc[fr, :, sy, sx] = blended
    return adain(x.clone().float(), torch.from_numpy(xc))


# common_utils/stable_render_utils/corresponder.py:188-220
def pre_atten_inject(n_ctx: torch.Tensor, idx):
    kv = torch.cat([n_ctx[int(i)] for i in idx], dim=0).unsqueeze(0).expand(n_ctx.shape[0], -1, -1)
    return n_ctx, kv, kv


# engine/managers/renderManager.py:926-936
def noise_pool(noise_f16: torch.Tensor, alpha_f16: torch.Tensor, bg_f32: torch.Tensor):
    """noise (1,H,W,4) fp16, alpha (1,H,W) fp16 (colour alpha), bg (1,H,W,4) fp32 -> pooled (H/8,W/8,4) fp32,
    latent noise (1,4,H/8,W/8).  'view(-1,8,8,4).mean((1,2))' = mean of 64 consecutive pixels of a row."""
    H, W = noise_f16.shape[1:3]
    mask = (1.0 - alpha_f16).unsqueeze(-1).expand_as(noise_f16)      # fp16
    n = noise_f16 * (1.0 - mask) + bg_f32 * mask                       # fp16*fp16 -> fp16, + fp32 -> fp32
    pooled = n.reshape(-1, 64, 4).mean(dim=1).view(H // 8, W // 8, 4)
    out = adain(pooled.unsqueeze(0), noise_f16, mode="NHWC").contiguous()
    return pooled, out


# ------------------------------------------------------------------------------------------------------
# engine/static/corrmap.py:578-736
def corrmap_update(values: np.ndarray, writtens: np.ndarray, frames: np.ndarray, ids: np.ndarray,
                   sprite=None, material=None, mode="first_avg", masks=None, inverse_masks=False,
                   ignore_obj_mat_id=False, channel_count=4):
    """In-place on values (k*k, V, C) fp16 / writtens (k*k, V) bool.  frames (N,H,W,3|4) float, ids (N,H,W,4).
    Reproduces the reference including its double gather of the colour rows (corrmap.py:703,710): after the
    mask compaction the colour rows are re-indexed with ORIGINAL pixel indices -> IndexError / wrong rows
    when the mask is partial and the sprite/material filter runs."""
    if ids.ndim != 4:
        raise ValueError("id maps must be (N,H,W,4): the reference never returns for a 3-D id map "
                         "(corrmap.py:629 appends to the list it iterates)")
    if frames.ndim == 3:
        frames = frames[None]
    if len(frames) != len(ids):
        raise ValueError("The length of color_frames and id_maps should be the same")
    for f in range(len(frames)):
        col = frames[f].astype(np.float32)
        if channel_count < col.shape[-1]:
            col = col[..., :channel_count]
        elif channel_count == 4 and col.shape[-1] == 3:
            col = np.concatenate([col, np.ones_like(col[..., :1])], -1)
        idm = ids[f].reshape(-1, 4).astype(np.int64)
        npx = idm.shape[0]
        row = np.arange(npx, dtype=np.int64)                # original pixel index of each surviving id row
        col = col.reshape(npx, -1)
        if masks is not None:
            m = masks[f].reshape(-1).astype(np.float32)
            if inverse_masks:
                m = 1 - m
            keep = m > 0
            row, idm = row[keep], idm[keep]
            col = col[row]
        if not ignore_obj_mat_id:
            if sprite is not None:
                k2 = idm[:, 0] == sprite
                row, idm = row[k2], idm[k2]
            if material is not None:
                k3 = idm[:, 1] == material
                row, idm = row[k3], idm[k3]
            col = col[row]                                  # <- quirk: indexes the (possibly compacted) rows
        mi, vid = idm[:, 2], idm[:, 3]
        if mode in ("first", "first_avg"):
            w = writtens[mi, vid]
            mi, vid, col = mi[~w], vid[~w], col[~w]
        values[mi, vid] = col.astype(np.float16)            # duplicates: last row wins
        writtens[mi, vid] = True


# ------------------------------------------------------------------------------------------------------
# comfy/model_sampling.py:75-150 (ModelSamplingDiscrete, linear-in-sqrt beta schedule)
def sigma_table(linear_start=0.00085, linear_end=0.012, timesteps=1000):
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    return (((1 - ac) / ac) ** 0.5).float()


class ModelSampling:
    def __init__(self):
        self.sigmas = sigma_table()
        self.log_sigmas = self.sigmas.log()
        self.sigma_min = self.sigmas[0]
        self.sigma_max = self.sigmas[-1]

    def timestep(self, sigma):
        sigma = torch.as_tensor(sigma, dtype=torch.float32)
        d = sigma.log().reshape(1, -1) - self.log_sigmas[:, None]
        return d.abs().argmin(dim=0).view(sigma.shape)

    def sigma(self, t):
        t = torch.clamp(torch.as_tensor(t).float(), min=0, max=len(self.sigmas) - 1)
        lo, hi, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[lo] + w * self.log_sigmas[hi]).exp()


# comfy/samplers.py:415-451, 937-951 ; comfy/k_diffusion/sampling.py:17-36
def scheduler_sigmas(ms: ModelSampling, name: str, steps: int) -> torch.Tensor:
    if name in ("normal", "sgm_uniform"):
        start, end = ms.timestep(ms.sigma_max), ms.timestep(ms.sigma_min)
        ts = torch.linspace(start, end, steps + 1)[:-1] if name == "sgm_uniform" else torch.linspace(start, end, steps)
        return torch.FloatTensor([float(ms.sigma(t)) for t in ts] + [0.0])
    if name == "simple":
        ss = len(ms.sigmas) / steps
        return torch.FloatTensor([float(ms.sigmas[-(1 + int(i * ss))]) for i in range(steps)] + [0.0])
    if name == "ddim_uniform":
        ss = max(len(ms.sigmas) // steps, 1)
        sig = []
        i = 1
        while i < len(ms.sigmas):
            sig.append(float(ms.sigmas[i]))
            i += ss
        return torch.FloatTensor(sig[::-1] + [0.0])
    if name == "karras":
        rho = 7.0
        ramp = torch.linspace(0, 1, steps)
        a, b = float(ms.sigma_min) ** (1 / rho), float(ms.sigma_max) ** (1 / rho)
        s = (b + ramp * (a - b)) ** rho
        return torch.cat([s, s.new_zeros([1])])
    if name == "exponential":
        s = torch.linspace(math.log(float(ms.sigma_max)), math.log(float(ms.sigma_min)), steps).exp()
        return torch.cat([s, s.new_zeros([1])])
    raise ValueError(name)


# comfy/samplers.py:979-1003 (KSampler.calculate_sigmas / set_steps; discard-penultimate samplers omitted)
def ksampler_sigmas(ms: ModelSampling, scheduler: str, steps: int, denoise=None):
    if denoise is None or denoise > 0.9999:
        sig = scheduler_sigmas(ms, scheduler, steps)
        ts = [int(ms.timestep(s)) for s in sig]
        return sig, ts
    new_steps = int(steps / denoise)
    sig = scheduler_sigmas(ms, scheduler, new_steps)
    ts = [int(ms.timestep(s)) for s in sig]           # NOT re-sliced (samplers.py:1000-1003 quirk)
    return sig[-(steps + 1):], ts


# comfy/model_sampling.py:7-29
def eps_input(x, sigma):
    return x / (sigma.view(-1, 1, 1, 1) ** 2 + 1.0) ** 0.5


def eps_denoised(x, out, sigma):
    return x - out * sigma.view(-1, 1, 1, 1)


# comfy/k_diffusion/sampling.py:129-149 / 749-793
def sample_loop(denoise_fn, x, sigmas, sampler="euler", callback=None):
    """denoise_fn(x, sigma_vec) -> denoised.  sampler in euler|ddim|ddpm|lcm (ddim == euler w/o mask).
    callback(i, x, denoised) fires after the model call, before the update (may mutate x in place)."""
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        den = denoise_fn(x, sigmas[i] * s_in)
        if sampler in ("euler", "ddim"):
            d = (x - den) / sigmas[i]           # sample_euler computes d BEFORE the callbacks mutate x (:140-144)
        if callback is not None:
            callback(i, x, den)
        if sampler in ("euler", "ddim"):
            x = x + d * (sigmas[i + 1] - sigmas[i])
        elif sampler == "ddpm":
            sg, sp = sigmas[i], sigmas[i + 1]
            xs = x / torch.sqrt(1.0 + sg ** 2.0)
            noise = (x - den) / sg
            ac, acp = 1 / (sg * sg + 1), 1 / (sp * sp + 1)
            al = ac / acp
            mu = (1.0 / al).sqrt() * (xs - (1 - al) * noise / (1 - ac).sqrt())
            if sp > 0:
                mu = mu + ((1 - al) * (1. - acp) / (1. - ac)).sqrt() * torch.randn_like(x)
            x = mu
            if sp != 0:
                x = x * torch.sqrt(1.0 + sp ** 2.0)
        elif sampler == "lcm":
            x = den
            if sigmas[i + 1] > 0:
                x = x + sigmas[i + 1] * torch.randn_like(x)
        else:
            raise ValueError(sampler)
    return x


# ------------------------------------------------------------------------------------------------------
# UNet (comfy/ldm/modules/diffusionmodules/openaimodel.py, comfy/ldm/modules/attention.py), functional,
# driven by the checkpoint-format state dict.
SD15_CFG = dict(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=[2, 2, 2, 2],
                channel_mult=[1, 2, 4, 4], transformer_depth=[1, 1, 1, 1, 1, 1, 0, 0], transformer_depth_middle=1,
                transformer_depth_output=[1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0], context_dim=768, num_heads=8)


def timestep_embedding(t, dim, max_period=10000):          # util.py:241-261
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(x, sd, p, eps, groups=32):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _attention(q, k, v, heads):                             # attention.py:92-140 (attention_basic, fp32)
    b, tq, c = q.shape
    d = c // heads
    q = q.view(b, tq, heads, d).transpose(1, 2)
    k = k.view(k.shape[0], -1, heads, d).transpose(1, 2)
    v = v.view(v.shape[0], -1, heads, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    o = torch.matmul(s.softmax(dim=-1), v)
    return o.transpose(1, 2).reshape(b, tq, c)


def _resblock(sd, p, x, emb):                               # openaimodel.py:253-281
    h = F.conv2d(F.silu(_gn(x, sd, p + ".in_layers.0", 1e-5)), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.conv2d(F.silu(_gn(h, sd, p + ".out_layers.0", 1e-5)), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def _tblock(sd, p, x, ctx, heads, inject_idx):              # attention.py:495-654
    n = F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    kv = n
    if inject_idx is not None:
        _, kv, _ = pre_atten_inject(n, inject_idx)
    q = F.linear(n, sd[p + ".attn1.to_q.weight"])
    k = F.linear(kv, sd[p + ".attn1.to_k.weight"])
    v = F.linear(kv, sd[p + ".attn1.to_v.weight"])
    a = _attention(q, k, v, heads)
    x = x + F.linear(a, sd[p + ".attn1.to_out.0.weight"], sd[p + ".attn1.to_out.0.bias"])
    n = F.layer_norm(x, x.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
    q = F.linear(n, sd[p + ".attn2.to_q.weight"])
    k = F.linear(ctx, sd[p + ".attn2.to_k.weight"])
    v = F.linear(ctx, sd[p + ".attn2.to_v.weight"])
    a = _attention(q, k, v, heads)
    x = x + F.linear(a, sd[p + ".attn2.to_out.0.weight"], sd[p + ".attn2.to_out.0.bias"])
    n = F.layer_norm(x, x.shape[-1:], sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    g = F.linear(n, sd[p + ".ff.net.0.proj.weight"], sd[p + ".ff.net.0.proj.bias"])
    a, gate = g.chunk(2, dim=-1)
    return x + F.linear(a * F.gelu(gate), sd[p + ".ff.net.2.weight"], sd[p + ".ff.net.2.bias"])


def _stransformer(sd, p, x, ctx, heads, depth, inject_idx):  # attention.py:699-726 (conv proj, SD1.x)
    b, c, h, w = x.shape
    x_in = x
    x = _gn(x, sd, p + ".norm", 1e-6)
    x = F.conv2d(x, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    x = x.flatten(2).transpose(1, 2)
    for i in range(depth):
        x = _tblock(sd, f"{p}.transformer_blocks.{i}", x, ctx, heads, inject_idx)
    x = x.transpose(1, 2).reshape(b, c, h, w)
    x = F.conv2d(x, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return x + x_in


def unet_forward(sd, cfg, x, t, ctx, inject_idx=None, control=None):
    """openaimodel.py:841-946.  sd: state dict with 'input_blocks.*' style keys (no prefix), fp32."""
    mc, heads = cfg["model_channels"], cfg["num_heads"]
    emb = timestep_embedding(t, mc)
    emb = F.linear(F.silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                   sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    hs = []
    h = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    hs.append(h)
    td = list(cfg["transformer_depth"])
    bi = 1
    nlev = len(cfg["channel_mult"])
    for lev in range(nlev):
        for _ in range(cfg["num_res_blocks"][lev]):
            h = _resblock(sd, f"input_blocks.{bi}.0", h, emb)
            depth = td.pop(0)
            if depth > 0:
                h = _stransformer(sd, f"input_blocks.{bi}.1", h, ctx, heads, depth, inject_idx)
            hs.append(h)
            bi += 1
        if lev != nlev - 1:
            h = F.conv2d(h, sd[f"input_blocks.{bi}.0.op.weight"], sd[f"input_blocks.{bi}.0.op.bias"], stride=2, padding=1)
            hs.append(h)
            bi += 1
    h = _resblock(sd, "middle_block.0", h, emb)
    h = _stransformer(sd, "middle_block.1", h, ctx, heads, cfg["transformer_depth_middle"], inject_idx)
    h = _resblock(sd, "middle_block.2", h, emb)
    if control is not None and control.get("middle"):
        h = h + control["middle"].pop()
    tdo = list(cfg["transformer_depth_output"])
    bi = 0
    for lev in reversed(range(nlev)):
        for i in range(cfg["num_res_blocks"][lev] + 1):
            hsp = hs.pop()
            if control is not None and control.get("output"):
                c_ = control["output"].pop()
                if c_ is not None:
                    hsp = hsp + c_
            h = torch.cat([h, hsp], dim=1)
            h = _resblock(sd, f"output_blocks.{bi}.0", h, emb)
            depth = tdo.pop()                               # openaimodel.py pops from the END
            j = 1
            if depth > 0:
                h = _stransformer(sd, f"output_blocks.{bi}.1", h, ctx, heads, depth, inject_idx)
                j = 2
            if lev > 0 and i == cfg["num_res_blocks"][lev]:
                # Upsample.forward(x, output_shape = hs[-1].shape) (openaimodel.py:100-117, :59-60): x2 unless a size was odd
                size = (hs[-1].shape[2], hs[-1].shape[3]) if hs else (h.shape[2] * 2, h.shape[3] * 2)
                h = F.interpolate(h, size=size, mode="nearest")
                h = F.conv2d(h, sd[f"output_blocks.{bi}.{j}.conv.weight"], sd[f"output_blocks.{bi}.{j}.conv.bias"], padding=1)
            bi += 1
    h = F.silu(_gn(h, sd, "out.0", 1e-5))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


# ------------------------------------------------------------------------------------------------------
# VAE decoder (comfy/ldm/modules/diffusionmodules/model.py:541-650; ddconfig comfy/sd.py:259)
def _vae_res(sd, p, x):
    h = F.conv2d(F.silu(_gn(x, sd, p + ".norm1", 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(h, sd, p + ".norm2", 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def _vae_attn(sd, p, x):
    b, c, h, w = x.shape
    n = _gn(x, sd, p + ".norm", 1e-6)
    q = F.conv2d(n, sd[p + ".q.weight"], sd[p + ".q.bias"]).flatten(2).transpose(1, 2)
    k = F.conv2d(n, sd[p + ".k.weight"], sd[p + ".k.bias"]).flatten(2).transpose(1, 2)
    v = F.conv2d(n, sd[p + ".v.weight"], sd[p + ".v.bias"]).flatten(2).transpose(1, 2)
    o = _attention(q, k, v, 1).transpose(1, 2).reshape(b, c, h, w)
    return x + F.conv2d(o, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def vae_decoder(sd, z, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    h = F.conv2d(z, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = _vae_res(sd, "mid.block_1", h)
    h = _vae_attn(sd, "mid.attn_1", h)
    h = _vae_res(sd, "mid.block_2", h)
    for lev in reversed(range(len(ch_mult))):
        for i in range(num_res_blocks + 1):
            h = _vae_res(sd, f"up.{lev}.block.{i}", h)
        if lev != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"up.{lev}.upsample.conv.weight"], sd[f"up.{lev}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(h, sd, "norm_out", 1e-6))
    return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def vae_encoder_moments(sd, x, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """Encoder.forward (model.py:500-520) + quant_conv (autoencoder.py:175-181): x (B,3,H,W) in [-1,1] -> (B,2z,h,w)"""
    h = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    for lev in range(len(ch_mult)):
        for i in range(num_res_blocks):
            h = _vae_res(sd, f"down.{lev}.block.{i}", h)
        if lev != len(ch_mult) - 1:                                    # Downsample (:88-92): pad bottom/right, stride-2 conv
            h = F.conv2d(F.pad(h, (0, 1, 0, 1), mode="constant", value=0), sd[f"down.{lev}.downsample.conv.weight"],
                         sd[f"down.{lev}.downsample.conv.bias"], stride=2)
    h = _vae_res(sd, "mid.block_1", h)
    h = _vae_attn(sd, "mid.attn_1", h)
    h = _vae_res(sd, "mid.block_2", h)
    h = F.conv2d(F.silu(_gn(h, sd, "norm_out", 1e-6)), sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def vae_encode(sd, pixels, noise=None, **kw):
    """VAE.encode (sd.py:353-371): pixels (B,H,W,3) in [0,1]; posterior sample mean + std*randn (distributions.py:24-37),
    noise drawn from the global generator when not given"""
    mom = vae_encoder_moments(sd, pixels[..., :3].movedim(-1, 1) * 2.0 - 1.0, **kw)
    mean, logvar = torch.chunk(mom, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    if noise is None:
        noise = torch.randn(mean.shape)
    return mean + std * noise


def vae_decode_image(sd, z, **kw):                          # comfy/sd.py:329-346 (Decoder part)
    return torch.clamp((vae_decoder(sd, z, **kw) + 1.0) / 2.0, min=0.0, max=1.0).movedim(1, -1)


# ------------------------------------------------------------------------------------------------------
# Whole sampling stack: custom_ksampler -> comfy.sample.sample -> KSampler -> sampling_function ->
# calc_cond_uncond_batch -> BaseModel.apply_model (nodes.py:1438, samplers.py:176-358, model_base.py:93-127)
LATENT_SCALE = 0.18215                                     # comfy/latent_formats.py SD15


def sample_frames(sd, cfg, noise, pos, neg, ids, steps, cfg_scale, sampler, scheduler, denoise=1.0,
                  overlap=None, latent=None, seed=None, cond_entries=None):
    """noise (N,4,h,w); pos/neg (1,77,C); ids (N,H,W,4) or None; overlap = dict(ratio, stop, n_rand) or None.
    RNG draw order replicates the reference (global torch RNG): custom_ksampler's seed randint, then
    pre_atten_inject's randint on the first attention block, then the sampler's per-step randn_like."""
    ms = ModelSampling()
    sig, timesteps = ksampler_sigmas(ms, scheduler, steps, denoise)
    n = noise.shape[0]
    if seed is None:
        seed = int(torch.randint(0, 2 ** 32, (1,)).item())  # custom_ksampler(seed=None) (nodes.py:1455)
    if latent is None:
        latent = torch.zeros_like(noise)
    latent = latent * LATENT_SCALE                          # process_latent_in (samplers.py:888, latent_formats.py)
    if sampler == "ddim":
        # "ddim" = euler + inpaint_options{"random"} (samplers.py:821-822): SAMPLER_METHOD.sample RESEEDS the
        # global RNG with seed+1 and draws one noise tensor from it (samplers.py:766-768) even with no mask.
        g = torch.manual_seed(seed + 1)
        torch.randn(noise.shape, generator=g, device="cpu")
    # noise_scaling (model_sampling.py:21-29); max_denoise when sigma[0] ~ sigma_max (samplers.py:745-751)
    max_denoise = math.isclose(float(ms.sigma_max), float(sig[0]), rel_tol=1e-05) or float(sig[0]) > float(ms.sigma_max)
    x = noise * (torch.sqrt(1.0 + sig[0] ** 2.0) if max_denoise else sig[0]) + latent
    state = {"idx": None}
    use_cfg = not math.isclose(cfg_scale, 1.0)

    entries = None
    if cond_entries is not None:                            # (pos entries, neg entries): masks / strengths / areas
        entries = prepare_cond_entries(cond_entries[0], cond_entries[1], noise.shape[2], noise.shape[3], ms)

    def denoise_fn(xx, sigma):
        if entries is not None:
            def model_fn(xin, s2, c):
                inj = None
                if overlap is not None and overlap.get("n_rand", 1) >= 0:
                    if state["idx"] is None:
                        state["idx"] = torch.randint(1, xin.shape[0], (overlap.get("n_rand", 1),))
                    inj = state["idx"]
                out = unet_forward(sd, cfg, eps_input(xin, s2), ms.timestep(s2).float(), c, inject_idx=inj)
                return eps_denoised(xin, out, s2)
            return sampling_function(model_fn, xx, sigma, entries[1], entries[0], cfg_scale)
        # batch order: uncond chunk first, then cond (samplers.py:230-262)
        if use_cfg:
            xin = torch.cat([xx, xx])
            s2 = torch.cat([sigma, sigma])
            c = torch.cat([neg.expand(n, -1, -1), pos.expand(n, -1, -1)])
        else:
            xin, s2, c = xx, sigma, pos.expand(n, -1, -1)
        inj = None
        if overlap is not None and overlap.get("n_rand", 1) >= 0:
            if state["idx"] is None:
                state["idx"] = torch.randint(1, xin.shape[0], (overlap.get("n_rand", 1),))
            inj = state["idx"]
        t = ms.timestep(s2).float()
        out = unet_forward(sd, cfg, eps_input(xin, s2), t, c, inject_idx=inj)
        den = eps_denoised(xin, out, s2)
        if use_cfg:
            u, cnd = den[:n], den[n:]
            return u + (cnd - u) * cfg_scale
        return den

    def cb(i, xx, den):
        if overlap is None or ids is None:
            return
        if timesteps[i] < overlap["stop"]:
            return
        xx.copy_(overlap_step(xx, ids, overlap["ratio"]))

    out = sample_loop(denoise_fn, x, sig, sampler, cb)
    return out / LATENT_SCALE, state["idx"]                 # process_latent_out (samplers.py:933)


# ------------------------------------------------------------------------------------------------------
# Conditioning composition: several positive / negative conditionings with masks, strengths and areas
# (comfy/samplers.py: get_area_and_mult :50-127, cond_cat :159-174, calc_cond_uncond_batch :176-320, sampling_function :323-358,
#  get_mask_aabb / resolve_areas_and_cond_masks :452-540, create_cond_with_same_area_if_none :542-575; sample() :887-912).
# An entry is a dict: cond (1|N,T,C) tensor, optional mask (Nm,H,W) / mask_strength / set_area_to_bounds / strength /
# area = (h, w, y, x) in latent cells or ("percentage", h, w, y, x).
def resolve_entries(entries, h, w):
    """resolve_areas_and_cond_masks: percentage areas -> cells; masks bilinearly resized to the latent; set_area_to_bounds ->
    area = bounding box of max|mask| over the batch, at least 8x8"""
    out = []
    for e in entries:
        e = dict(e)
        a = e.get("area")
        if a is not None and a[0] == "percentage":
            e["area"] = (max(1, round(a[1] * h)), max(1, round(a[2] * w)), round(a[3] * h), round(a[4] * w))
        if e.get("mask") is not None:
            m = e["mask"].float()
            if m.dim() == 2:
                m = m.unsqueeze(0)
            if m.shape[1] != h or m.shape[2] != w:
                m = F.interpolate(m.unsqueeze(1), size=(h, w), mode="bilinear", align_corners=False).squeeze(1)
            if e.get("set_area_to_bounds", False):
                b = m.abs().max(dim=0).values
                if not bool((b != 0).any()):
                    e["area"] = (8, 8, 0, 0)
                else:
                    ys, xs = torch.where(b != 0)                       # (torch.where(mask) on a float mask = non-zero)
                    y0, y1, x0, x1 = int(ys.min()), int(ys.max()), int(xs.min()), int(xs.max())
                    e["area"] = (max(8, y1 - y0 + 1), max(8, x1 - x0 + 1), y0, x0)
            e["mask"] = m
        out.append(e)
    return out


def add_opposite_areas(conds, c):
    """create_cond_with_same_area_if_none(conds, c): if c has an area and conds holds none with exactly that area, append a
    copy of c carrying the conditioning tensor of the smallest entry of conds that encloses it (or an area-less one)"""
    if c.get("area") is None:
        return
    ca = c["area"]
    smallest = None
    for x in conds:
        if x.get("area") is not None:
            a = x["area"]
            if ca[2] >= a[2] and ca[3] >= a[3] and a[0] + a[2] >= ca[0] + ca[2] and a[1] + a[3] >= ca[1] + ca[3]:
                if smallest is None or smallest.get("area") is None:
                    smallest = x
                elif smallest["area"][0] * smallest["area"][1] > a[0] * a[1]:
                    smallest = x
        elif smallest is None:
            smallest = x
    if smallest is None:
        return
    if smallest.get("area") is not None and tuple(smallest["area"]) == tuple(ca):
        return
    o = dict(c)
    o["cond"] = smallest["cond"]
    conds.append(o)


def area_and_mult(e, x):
    """get_area_and_mult -> (input_x, mult, area); mask-less areas are feathered over 8 cells towards inner borders"""
    area = (x.shape[2], x.shape[3], 0, 0) if e.get("area") is None else tuple(int(v) for v in e["area"])
    strength = float(e.get("strength", 1.0))
    ix = x[:, :, area[2]:area[0] + area[2], area[3]:area[1] + area[3]]
    if e.get("mask") is not None:
        m = e["mask"]
        assert m.shape[1] == x.shape[2] and m.shape[2] == x.shape[3]
        m = m[:, area[2]:area[0] + area[2], area[3]:area[1] + area[3]] * float(e.get("mask_strength", 1.0))
        m = m.unsqueeze(1).repeat(ix.shape[0] // m.shape[0], ix.shape[1], 1, 1)
    else:
        m = torch.ones_like(ix)
    mult = m * strength
    if e.get("mask") is None:
        rr = 8
        if area[2] != 0:
            for t in range(rr):
                mult[:, :, t:1 + t, :] *= ((1.0 / rr) * (t + 1))
        if (area[0] + area[2]) < x.shape[2]:
            for t in range(rr):
                mult[:, :, area[0] - 1 - t:area[0] - t, :] *= ((1.0 / rr) * (t + 1))
        if area[3] != 0:
            for t in range(rr):
                mult[:, :, :, t:1 + t] *= ((1.0 / rr) * (t + 1))
        if (area[1] + area[3]) < x.shape[3]:
            for t in range(rr):
                mult[:, :, :, area[1] - 1 - t:area[1] - t] *= ((1.0 / rr) * (t + 1))
    return ix, mult, area


def percent_to_sigma(ms, percent):
    """ModelSamplingDiscrete.percent_to_sigma (comfy/model_sampling.py:138-144)"""
    if percent <= 0.0:
        return 999999999.9
    if percent >= 1.0:
        return 0.0
    return float(ms.sigma(torch.tensor((1.0 - percent) * 999.0)))


def with_timestep_ranges(entries, ms):
    """calculate_start_end_timesteps (samplers.py:578-602): start_percent / end_percent -> timestep_start / timestep_end (sigmas)"""
    out = []
    for e in entries:
        if "start_percent" in e or "end_percent" in e:
            e = dict(e)
            if "start_percent" in e:
                e["timestep_start"] = percent_to_sigma(ms, e["start_percent"])
            if "end_percent" in e:
                e["timestep_end"] = percent_to_sigma(ms, e["end_percent"])
        out.append(e)
    return out


def entry_active(e, sigma0):
    """get_area_and_mult's first test (samplers.py:60-67): an entry outside its sigma window returns None -- it is not run at
    this step and contributes nothing to out / count"""
    if "timestep_start" in e and sigma0 > e["timestep_start"]:
        return False
    if "timestep_end" in e and sigma0 < e["timestep_end"]:
        return False
    return True


def calc_cond_uncond_batch(model_fn, cond, uncond, x, sigma):
    """model_fn(input_x (B,4,ah,aw), sigma (B,), ctx (B,T,C)) -> DENOISED prediction (apply_model's return).  Entries whose
    cropped input has the same shape and token count run as ONE batch, in the reference's order: the batchable entries of
    [cond..., uncond...] reversed (free memory is never the limit here).  -> (cond_pred, uncond_pred, batches) where batches
    lists, per model call, [(kind, entry index), ...] in batch order (test hook)."""
    out_c, cnt_c = torch.zeros_like(x), torch.ones_like(x) * 1e-37
    out_u, cnt_u = torch.zeros_like(x), torch.ones_like(x) * 1e-37
    n = x.shape[0]
    s0 = float(sigma.reshape(-1)[0])
    to_run = [(area_and_mult(e, x), 0, ("pos", i), e) for i, e in enumerate(cond or []) if entry_active(e, s0)]
    to_run += [(area_and_mult(e, x), 1, ("neg", i), e) for i, e in enumerate(uncond or []) if entry_active(e, s0)]
    batches = []
    while to_run:
        first = to_run[0]
        idxs = [i for i in range(len(to_run)) if to_run[i][0][0].shape == first[0][0].shape
                and to_run[i][3]["cond"].shape[1:] == first[3]["cond"].shape[1:]]
        idxs.reverse()
        picked = [to_run.pop(i) for i in idxs]
        xin = torch.cat([p[0][0] for p in picked])
        ctx = torch.cat([p[3]["cond"].expand(n, -1, -1) for p in picked])
        sg = torch.cat([sigma] * len(picked))[:len(xin)]
        out = model_fn(xin, sg, ctx).chunk(len(picked))
        batches.append([p[2] for p in picked])
        for o, p in zip(out, picked):
            (_, mult, a), kind = p[0], p[1]
            tgt, cnt = (out_c, cnt_c) if kind == 0 else (out_u, cnt_u)
            tgt[:, :, a[2]:a[0] + a[2], a[3]:a[1] + a[3]] += o * mult
            cnt[:, :, a[2]:a[0] + a[2], a[3]:a[1] + a[3]] += mult
    return out_c / cnt_c, out_u / cnt_u, batches


def sampling_function(model_fn, x, sigma, uncond, cond, cond_scale):
    """:323-358: uncond skipped at cfg 1; result = uncond_pred + (cond_pred - uncond_pred) * scale"""
    c, u, _ = calc_cond_uncond_batch(model_fn, cond, None if math.isclose(cond_scale, 1.0) else uncond, x, sigma)
    return u + (c - u) * cond_scale


def prepare_cond_entries(pos, neg, h, w, ms=None):
    """samplers.sample() :887-912 for mask / area entries: resolve both lists, percent ranges -> sigma windows (with a model
    sampling object), then give every area an opposite entry"""
    pos, neg = resolve_entries(pos, h, w), resolve_entries(neg, h, w)
    if ms is not None:
        neg, pos = with_timestep_ranges(neg, ms), with_timestep_ranges(pos, ms)
    for c in list(pos):
        add_opposite_areas(neg, c)
    for c in list(neg):
        add_opposite_areas(pos, c)
    return pos, neg


# ------------------------------------------------------------------------------------------------------
# ControlNet (comfy/cldm/cldm.py:284-311) + control_merge (comfy/controlnet.py:95-141)
def common_upscale_center(x, width, height):
    """comfy.utils.common_upscale(x, width, height, 'nearest-exact', 'center') (comfy/utils.py:418-443): what ControlNet.get_control
    does to the WHOLE hint image when the model call runs on a cropped conditioning area (comfy/controlnet.py:193-201)"""
    ow, oh = x.shape[3], x.shape[2]
    old_aspect, new_aspect = ow / oh, width / height
    cx = cy = 0
    if old_aspect > new_aspect:
        cx = round((ow - ow * (new_aspect / old_aspect)) / 2)
    elif old_aspect < new_aspect:
        cy = round((oh - oh * (old_aspect / new_aspect)) / 2)
    return F.interpolate(x[:, :, cy:oh - cy, cx:ow - cx], size=(height, width), mode="nearest-exact")


def controlnet_forward(sd, cfg, x, hint, t, ctx, strength=1.0):
    """-> dict(output=[12 tensors], middle=[1 tensor]) as consumed by unet_forward(control=...)"""
    mc, heads = cfg["model_channels"], cfg["num_heads"]
    emb = timestep_embedding(t, mc)
    emb = F.linear(F.silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                   sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    g = hint
    for i, stride in ((0, 1), (2, 1), (4, 2), (6, 1), (8, 2), (10, 1), (12, 2), (14, 1)):
        g = F.conv2d(g, sd[f"input_hint_block.{i}.weight"], sd[f"input_hint_block.{i}.bias"], stride=stride, padding=1)
        if i != 14:
            g = F.silu(g)
    outs = []

    def zc(i, h):
        return F.conv2d(h, sd[f"zero_convs.{i}.0.weight"], sd[f"zero_convs.{i}.0.bias"])
    h = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1) + g
    outs.append(zc(0, h))
    td = list(cfg["transformer_depth"])
    bi, nlev = 1, len(cfg["channel_mult"])
    for lev in range(nlev):
        for _ in range(cfg["num_res_blocks"][lev]):
            h = _resblock(sd, f"input_blocks.{bi}.0", h, emb)
            depth = td.pop(0)
            if depth > 0:
                h = _stransformer(sd, f"input_blocks.{bi}.1", h, ctx, heads, depth, None)
            outs.append(zc(bi, h))
            bi += 1
        if lev != nlev - 1:
            h = F.conv2d(h, sd[f"input_blocks.{bi}.0.op.weight"], sd[f"input_blocks.{bi}.0.op.bias"], stride=2, padding=1)
            outs.append(zc(bi, h))
            bi += 1
    h = _resblock(sd, "middle_block.0", h, emb)
    h = _stransformer(sd, "middle_block.1", h, ctx, heads, cfg["transformer_depth_middle"], None)
    h = _resblock(sd, "middle_block.2", h, emb)
    mid = F.conv2d(h, sd["middle_block_out.0.weight"], sd["middle_block_out.0.bias"])
    return {"output": [o * strength for o in outs], "middle": [mid * strength]}


# ------------------------------------------------------------------------------------------------------
# Legacy overlap (legacy_codes/stable_rendering_algo/overlap/overlap.py:83-152,180-222; algorithms.py:34-118)
def legacy_corr_map(ids: np.ndarray):
    """(T,H,W,4) int ids -> dict {id_tuple: [((y,x), frame), ...]} in (frame,y,x) order; all-zero ids skipped
    (data_classes/correspondence_map.py:150-166)."""
    m = {}
    T, H, W, _ = ids.shape
    for f in range(T):
        for i in range(H):
            for j in range(W):
                k = tuple(int(v) for v in ids[f, i, j])
                if k == (0, 0, 0, 0):
                    continue
                m.setdefault(k, []).append(((i, j), f))
    return m


def _legacy_weights(algo, fs, ys, xs, vn):
    fs, ys, xs = np.asarray(fs, np.float32), np.asarray(ys, np.float32), np.asarray(xs, np.float32)
    n = len(fs)
    if algo == "average":
        return np.ones((n, n), np.float32)
    if algo == "frame":
        return (1.0 / (np.abs(fs[:, None] - fs[None, :]) + 1.0)).astype(np.float32)
    if algo == "pixel":
        return (1.0 / (np.abs(xs[:, None] - xs[None, :]) + np.abs(ys[:, None] - ys[None, :]) + 1.0)).astype(np.float32)
    if algo == "view_normal":                                     # |1 - vn_j| broadcast over rows (algorithms.py:112-114)
        d = np.abs(np.ones_like(vn)[:, None] - vn[None, :])
        return (1.0 / (d + 1.0)).astype(np.float32)
    raise ValueError(algo)


def legacy_overlap(frames: np.ndarray, ids: np.ndarray, alpha, radius=0, algo="average", view_normal=None, sequential=True):
    """frames (T,1,C,H,W) fp32 at corr-map resolution.  sequential=True reproduces the reference exactly: the 'copy' it
    writes to shares storage with the source (``.detach()``, overlap.py:103), so later vertices see earlier updates
    (only observable for radius > 0).  sequential=False is the order-independent form the HIP kernel computes."""
    src = frames.astype(np.float32).copy()
    out = src if sequential else src.copy()
    T, _, C, H, W = src.shape
    for key, info in legacy_corr_map(ids).items():
        if len(info) == 1:
            continue
        pos, fs = zip(*info)
        ys, xs = zip(*pos)
        fs, ys, xs = list(fs), list(ys), list(xs)
        lat = src[fs, 0][:, :, ys, xs] if False else np.stack([src[f, 0, :, y, x] for f, y, x in zip(fs, ys, xs)])   # (n,C)
        pooled = np.zeros_like(lat)
        for t, (f, y, x) in enumerate(zip(fs, ys, xs)):
            acc = np.zeros(C, np.float32)
            for k in range(-radius, radius + 1):                 # a diagonal of 2r+1 pixels, clamped (overlap.py:69-76)
                acc += src[f, 0, :, min(max(y + k, 0), H - 1), min(max(x + k, 0), W - 1)]
            pooled[t] = acc / np.float32(2 * radius + 1)
        vn = None if view_normal is None else np.asarray([view_normal[f, y, x] for f, y, x in zip(fs, ys, xs)], np.float32)
        w = _legacy_weights(algo, fs, ys, xs, vn)
        ov = (w @ pooled) / w.sum(axis=0).reshape(-1, 1)          # column sums applied per ROW, as the reference does
        new = np.float32(alpha) * ov + np.float32(1 - alpha) * lat
        for t, (f, y, x) in enumerate(zip(fs, ys, xs)):
            out[f, 0, :, y, x] = new[t]
    return out


def legacy_resize_overlap(latents: np.ndarray, ids: np.ndarray, alpha, radius=0, algo="average", view_normal=None, sequential=True):
    """ResizeOverlap.__call__ with nearest interpolation: latents (T,1,C,h,w), ids (T,H,W,4)"""
    if alpha == 0:
        return latents
    T, _, C, h, w = latents.shape
    H, W = ids.shape[1:3]
    t = torch.from_numpy(latents.astype(np.float32))
    up = torch.stack([F.interpolate(t[i], size=(H, W), mode="nearest") for i in range(T)]).numpy()
    ov = legacy_overlap(up, ids, alpha, radius, algo, view_normal, sequential)
    dn = torch.stack([F.interpolate(torch.from_numpy(ov[i]), size=(h, w), mode="nearest") for i in range(T)]).numpy()
    return np.where(dn != 0, dn, latents)
