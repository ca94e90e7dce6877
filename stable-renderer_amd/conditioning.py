"""Conditioning lists with masks, strengths and areas: the host side of ``calc_cond_uncond_batch``.

Mirrors, with the same meaning: ``convert_cond`` (comfyUI/comfy/sample.py:71-112), ``resolve_areas_and_cond_masks`` /
``get_mask_aabb`` (comfy/samplers.py:452-540), ``create_cond_with_same_area_if_none`` (:542-575), ``get_area_and_mult``
(:50-127) and the batching rule of ``calc_cond_uncond_batch`` (:207-262: entries whose cropped input and token count agree run
as one model call, in reversed list order; free memory never limits a batch on a 288 GB part).  Everything here is list / mask
preparation done ONCE per sampling run on tensors of latent size; the per-step arithmetic (crop + EPS scale, weighted
accumulation, CFG) runs in ``sr_cond_crop_scale`` / ``sr_cond_accumulate`` / ``sr_cfg_combine``.

An entry is a dict: ``cond`` (1|N, T, C) tensor, optional ``mask`` (Nm, H, W), ``mask_strength``, ``set_area_to_bounds``,
``strength``, ``area`` = (h, w, y, x) in latent cells or ("percentage", h, w, y, x), ``pooled_output``, ``control``.
"""
import torch
import torch.nn.functional as F


def entries_of(conditioning):
    """CONDITIONING ([[cond, {...}], ...], comfyUI/nodes.py:53-65) or a bare (1|N, T, C) tensor -> list of entries"""
    if isinstance(conditioning, torch.Tensor):
        return [dict(cond=conditioning)]
    out = []
    for item in conditioning:
        if not isinstance(item, (list, tuple)) or len(item) != 2 or not isinstance(item[1], dict):
            # SceneTextEncode(merge=False, idmap=None) flattens its pairs with `conds += [cond, dict]` in the reference
            # (_nodes/conditions.py:124): convert_cond then fails on the bare tensor; same outcome here
            raise TypeError("conditioning must be a list of [tensor, dict] pairs")
        e = dict(item[1])
        e["cond"] = item[0]
        out.append(e)
    return out


# a key calc_cond_uncond_batch acts on that this package does not implement: an entry carrying it is refused instead of being
# run as if the key were absent (the reference feeds gligen boxes to the transformer blocks, samplers.py:103-115)
_UNSUPPORTED_KEYS = ("gligen",)


def check_supported(entries):
    for e in entries:
        bad = [k for k in _UNSUPPORTED_KEYS if e.get(k) is not None]
        if bad:
            raise NotImplementedError("conditioning entry carries %s (GLIGEN): not implemented, refusing to run it without"
                                      % ", ".join(bad))


def percent_to_sigma(ms, percent):
    """ModelSamplingDiscrete.percent_to_sigma (comfy/model_sampling.py:138-144)"""
    if percent <= 0.0:
        return 999999999.9
    if percent >= 1.0:
        return 0.0
    return float(ms.sigma(torch.tensor((1.0 - percent) * 999.0)))


def with_timestep_ranges(entries, ms):
    """calculate_start_end_timesteps (comfy/samplers.py:578-602): ConditioningSetTimestepRange's start_percent / end_percent
    (comfyUI/nodes.py:270-285) become the sigma window [timestep_end, timestep_start] of the entry"""
    out = []
    for e in entries:
        if "start_percent" in e or "end_percent" in e:
            e = dict(e)
            if "start_percent" in e:
                e["timestep_start"] = percent_to_sigma(ms, float(e["start_percent"]))
            if "end_percent" in e:
                e["timestep_end"] = percent_to_sigma(ms, float(e["end_percent"]))
        out.append(e)
    return out


def entry_active(e, sigma):
    """get_area_and_mult's first test (comfy/samplers.py:60-67): outside its sigma window an entry is not run at this step and
    adds nothing to out / count"""
    if e.get("timestep_start") is not None and sigma > e["timestep_start"]:
        return False
    if e.get("timestep_end") is not None and sigma < e["timestep_end"]:
        return False
    return True


def has_window(e):
    """the entry can be inactive at some sigma"""
    return (e.get("timestep_start") is not None or e.get("timestep_end") is not None
            or float(e.get("start_percent", 0.0)) > 0.0 or float(e.get("end_percent", 1.0)) < 1.0)


def is_plain(entries):
    """one full-area entry with unit strength: the [uncond | cond] fast path applies"""
    check_supported(entries)
    if len(entries) != 1:
        return False
    e = entries[0]
    return e.get("mask") is None and e.get("area") is None and float(e.get("strength", 1.0)) == 1.0 and not has_window(e)


def resolve_entries(entries, h, w):
    out = []
    for e in entries:
        e = dict(e)
        a = e.get("area")
        if a is not None and a[0] == "percentage":
            e["area"] = (max(1, round(a[1] * h)), max(1, round(a[2] * w)), round(a[3] * h), round(a[4] * w))
        if e.get("mask") is not None:
            m = e["mask"].detach().to("cpu", torch.float32)
            if m.dim() == 2:
                m = m.unsqueeze(0)
            if m.shape[1] != h or m.shape[2] != w:
                m = F.interpolate(m.unsqueeze(1), size=(h, w), mode="bilinear", align_corners=False).squeeze(1)
            if e.get("set_area_to_bounds", False):
                bounds = m.abs().max(dim=0).values
                nz = torch.nonzero(bounds)
                if nz.numel() == 0:
                    e["area"] = (8, 8, 0, 0)                     # all-zero mask: smallest legal area, a no-op anyway
                else:
                    y0, x0 = int(nz[:, 0].min()), int(nz[:, 1].min())
                    y1, x1 = int(nz[:, 0].max()), int(nz[:, 1].max())
                    e["area"] = (max(8, y1 - y0 + 1), max(8, x1 - x0 + 1), y0, x0)
            e["mask"] = m
        out.append(e)
    return out


def add_opposite_area(conds, c):
    """every area needs an entry of the opposite sign with the same area (samplers.py:542-575)"""
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
    if "pooled_output" in smallest:
        o["pooled_output"] = smallest["pooled_output"]
    conds.append(o)


def prepare(pos, neg, h, w, ms=None):
    """samplers.sample() :887-912: resolve, sigma windows (ms: the model sampling object), opposite-area entries"""
    pos, neg = resolve_entries(pos, h, w), resolve_entries(neg, h, w)
    if ms is not None:
        neg, pos = with_timestep_ranges(neg, ms), with_timestep_ranges(pos, ms)
    for c in list(pos):
        add_opposite_area(neg, c)
    for c in list(neg):
        add_opposite_area(pos, c)
    return pos, neg


def mult_of(e, N, C, h, w):
    """-> (mult (N,C,ah,aw) fp32 host tensor, area (ah, aw, y0, x0)): mask * mask_strength * strength, or the feathered
    all-ones window of a mask-less area (get_area_and_mult)"""
    area = (h, w, 0, 0) if e.get("area") is None else tuple(int(v) for v in e["area"])
    ah, aw, y0, x0 = area
    if y0 < 0 or x0 < 0 or y0 + ah > h or x0 + aw > w or ah < 1 or aw < 1:
        raise ValueError(f"conditioning area {area} does not fit the {h}x{w} latent")
    strength = float(e.get("strength", 1.0))
    if e.get("mask") is not None:
        m = e["mask"]
        if m.shape[1] != h or m.shape[2] != w:
            raise AssertionError("mask was not resized to the latent")
        m = m[:, y0:y0 + ah, x0:x0 + aw] * float(e.get("mask_strength", 1.0))
        if N % m.shape[0]:
            raise ValueError(f"a mask batch of {m.shape[0]} does not divide the latent batch {N}")
        mult = m.unsqueeze(1).repeat(N // m.shape[0], C, 1, 1) * strength
    else:
        mult = torch.ones(N, C, ah, aw, dtype=torch.float32) * strength
        rr = 8
        if y0 != 0:
            for t in range(rr):
                mult[:, :, t:1 + t, :] *= ((1.0 / rr) * (t + 1))
        if (ah + y0) < h:
            for t in range(rr):
                mult[:, :, ah - 1 - t:ah - t, :] *= ((1.0 / rr) * (t + 1))
        if x0 != 0:
            for t in range(rr):
                mult[:, :, :, t:1 + t] *= ((1.0 / rr) * (t + 1))
        if (aw + x0) < w:
            for t in range(rr):
                mult[:, :, :, aw - 1 - t:aw - t] *= ((1.0 / rr) * (t + 1))
    return mult.contiguous(), area


def groups_of(pos, neg, N, C, h, w, use_uncond=True):
    """-> list of model calls in execution order; each = dict(area, members=[(kind 0|1, entry, mult)], ...) with members in BATCH
    order.  kind 0 = cond, 1 = uncond (skipped altogether at cfg 1, samplers.py:335)."""
    to_run = [(0, e) + mult_of(e, N, C, h, w) for e in pos]
    if use_uncond:
        to_run += [(1, e) + mult_of(e, N, C, h, w) for e in neg]
    groups = []
    while to_run:
        first = to_run[0]
        fa, ft = first[3][:2], tuple(first[1]["cond"].shape[1:])
        idxs = [i for i in range(len(to_run)) if to_run[i][3][:2] == fa and tuple(to_run[i][1]["cond"].shape[1:]) == ft]
        # (can_concat_cond compares the SHAPE of the cropped inputs, not their position: windows of equal size batch together.
        #  The accumulation kernel takes one window per call, so equal-size windows at different places run as separate calls:
        #  same sums, a different batch composition -- visible only to K/V injection, which such lists do not combine with.)
        idxs = [i for i in idxs if to_run[i][3] == first[3]]
        idxs.reverse()
        members = [to_run.pop(i) for i in idxs]
        groups.append(dict(area=first[3], members=[(k, e, m) for k, e, m, _ in members]))
    return groups
