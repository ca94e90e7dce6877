"""The remaining nodes of the stable-rendering surface: noise-sequence generators, legacy loaders, the small
image / text / video utilities and the legacy overlap sampler.  They sit beside the hot path (graph inputs, post-processing): tensor
bookkeeping on the device, no model.  Same names, argument meaning and error behaviour as the reference classes cited per node;
registered by graph_nodes."""
import os

import numpy as np
import torch

from . import nodes as N
from .corrmap import IDMap
from .types import LATENT


def _sd_size(sd_version):
    if sd_version not in ["SD15", "SDXL"]:
        raise ValueError("sd_version should be either SD15 or SDXL")
    return 512 if sd_version == "SD15" else 1024


class CreateIdenticalNoiseSequence(N.StableRenderingNode):
    """_nodes/loaders.py:274-309: ONE latent and ONE noise image repeated for every frame.  Both generators of the reference are the
    global default generator (``torch.manual_seed`` returns it), so after ``manual_seed(seed); manual_seed(seed + 1)`` the latent is
    the first draw and the noise the second draw of the stream seeded with ``seed + 1``."""
    Category = "loader"

    def __call__(self, seed: int, num_frames: int, sd_version='SD15', device="cuda") -> LATENT:
        side = _sd_size(sd_version) // 8
        if num_frames <= 0:
            raise ValueError("num_frames should be larger than 0.")
        torch.manual_seed(seed)
        g = torch.manual_seed(seed + 1)
        latent = torch.randn([1, 4, side, side], device="cpu", generator=g).repeat(num_frames, 1, 1, 1).to(device)
        noise = torch.randn([1, 4, side, side], device="cpu", generator=g).repeat(num_frames, 1, 1, 1).to(device)
        return LATENT(samples=latent, noise=noise)


class CreateNoiseSequenceFromIdMap(N.StableRenderingNode):
    """_nodes/loaders.py:154-271: a full-resolution latent and noise image shared by all frames, in which every pixel that shows a
    vertex gets that VERTEX's random value (one draw per unique vertexID, the same in every frame: tensor_group_by_then_randn_init,
    math_utils.py:164-229), then reduced 8x.  The per-vertex draws of the reference come from the CUDA generator (``randn_like`` of a
    cuda tensor) and cannot be reproduced bit for bit on another device; here they are drawn on the host from the global generator,
    in the sorted order of the unique ids.  The 'mean' / 'max' / 'min' options keep the reference's ``view(-1, 4, 8, 8)`` regrouping
    (256 consecutive values of a row-major plane, not 8x8 blocks: the output batch doubles, :262-271)."""
    Category = "loader"

    def __call__(self, id_map: IDMap, seed: int, sd_version='SD15', downsample_option='nearest') -> LATENT:
        side = _sd_size(sd_version)
        if downsample_option not in ["mean", "max", "min", "nearest"]:
            raise ValueError("downsample_option should be either mean, max, min, or nearest")
        if id_map is None or id_map.tensor.numel() == 0:
            raise ValueError("ID map is empty.")
        ids = id_map.tensor
        dev = ids.device
        n = ids.shape[0]
        torch.manual_seed(seed)
        g = torch.manual_seed(seed + 1)
        latent = torch.randn([1, 4, side, side], device="cpu", generator=g).repeat(n, 1, 1, 1).to(dev)
        noise = torch.randn([1, 4, side, side], device="cpu", generator=g).repeat(n, 1, 1, 1).to(dev)
        # rows of create_vertex_screen_info (corrmap.py:220-280): pixels with an id that is not the non-AI index; x/H, y/W ratios
        valid = (ids[..., 2] != 2048) & (ids != 0).any(-1)
        f, y, x = torch.nonzero(valid, as_tuple=True)
        H, W = ids.shape[1], ids.shape[2]
        sx = ((x.float() / H) * side).long()
        sy = ((y.float() / W) * side).long()
        if bool((sx >= side).any()) or bool((sy >= side).any()):
            raise IndexError("index out of bounds: id-map pixel maps outside the noise image (non-square id map)")
        uniq, inv = torch.unique(ids[f, y, x, 3], return_inverse=True)
        for t in (latent, noise):                                       # latent first, then noise, as the reference draws them
            rv = torch.randn(uniq.shape[0], 4).to(dev)
            t[f, :, sy, sx] = rv[inv]
        if downsample_option == "nearest":
            st = 8
            return LATENT(samples=latent[:, :, ::st, ::st].contiguous(), noise=noise[:, :, ::st, ::st].contiguous())
        red = {"mean": lambda t: t.mean(dim=(1, 2)), "max": lambda t: t.amax(dim=(1, 2)), "min": lambda t: t.amin(dim=(1, 2))}[downsample_option]
        noise = red(noise.contiguous().view(-1, 4, 8, 8)).view(-1, 4, side // 8, side // 8)
        return LATENT(samples=torch.zeros_like(noise), noise=noise)


def _legacy_index(path, i):
    """_nodes/legacy/loaders.py:33-39: `name_<n>.ext` or `<n>_name.ext`, else the list position"""
    name = os.path.basename(str(path))
    if name.split('.')[0].split('_')[-1].isdigit():
        return int(name.split('.')[0].split('_')[-1])
    if name.split('_')[0].isdigit():
        return int(name.split('_')[0])
    return i


def _read_rgba(path):
    from PIL import Image
    return torch.from_numpy(np.array(Image.open(path).convert("RGBA"))).permute(2, 0, 1)       # CHW uint8 (read_image RGB_ALPHA)


class LegacyImageSequenceLoader(N.StableRenderingNode):
    """_nodes/legacy/loaders.py:13-58 -> (IMAGE (N,H,W,3) in [0,1], MASK (N,H,W) = 1 - alpha)"""
    Category = "legacy_loader"
    N_OUTPUTS = 2

    def __call__(self, imgs):
        imgs = list(imgs)
        ordered = sorted(imgs, key=lambda p: _legacy_index(p, imgs.index(p)))
        images, masks = [], []
        for p in ordered:
            if not os.path.exists(p):
                continue
            t = _read_rgba(p).permute(1, 2, 0).unsqueeze(0) / 255.0
            images.append(t[..., :3])
            m = (1 - t[..., -1]).squeeze()
            masks.append(m.unsqueeze(0) if m.dim() == 2 else m)
        return torch.cat(images, dim=0), torch.cat(masks, dim=0)


def _load_planes(paths, what):
    paths = [str(p) for p in paths]
    ordered = sorted(paths, key=lambda p: _legacy_index(p, paths.index(p)))
    out = []
    for p in ordered:
        if not os.path.exists(p):
            continue
        if p.endswith('.npy'):
            t = torch.from_numpy(np.load(p)).squeeze()
            if t.dim() != 3:
                raise ValueError(f"Invalid shape of {what} tensor: {t.shape}.")
            if not (t.shape[-1] == 4 or t.shape[1] == 4):
                raise ValueError(f"Invalid {what} tensor shape: {t.shape}.")
            if t.shape[-1] == 4:
                t = t.permute(2, 0, 1)
        else:
            t = _read_rgba(p) / 255.0
        out.append(t)
    for t in out:
        if t.shape != out[0].shape:
            raise ValueError(f"Tensor data has inconsistent shapes: {t.shape} and {out[0].shape}.")
    return ordered, out


class LegacyNoiseSequenceLoader(N.StableRenderingNode):
    """_nodes/legacy/loaders.py:61-103: planes concatenated along dim 0 exactly as the reference does (``torch.cat`` of CHW tensors)"""
    Category = "legacy_loader"

    def __call__(self, data_paths) -> LATENT:
        _, ts = _load_planes(data_paths, "noise")
        t = torch.cat(ts, dim=0)
        return LATENT(samples=torch.zeros_like(t), noise=t)


class LegacyIDSequenceLoader(N.StableRenderingNode):
    """_nodes/legacy/loaders.py:106-147"""
    Category = "legacy_loader"

    def __call__(self, data_paths, device="cuda") -> IDMap:
        ordered, ts = _load_planes(data_paths, "id")
        frame_indices = [_legacy_index(p, i) for i, p in enumerate(ordered)]
        t = torch.stack([x.permute(1, 2, 0) if x.shape[0] == 4 else x for x in ts], 0)        # IDMap wants (N,H,W,4)
        return IDMap(frame_indices=frame_indices, tensor=t.to(device))


# ---- processing (_nodes/processing) -----------------------------------------------------------------------------------------------
class RemoveBGNode(N.StableRenderingNode):
    """_nodes/processing/img.py:63-82.  The reference downloads ``skytnt/anime-seg`` (isnetis.onnx) from the Hugging Face hub and
    runs it with onnxruntime; neither the network nor onnxruntime is available here, so the node needs a segmentation callable
    (``set_segmenter(fn)``: image (H,W,3) in [0,1] -> mask (H,W) in [0,1]) and fails loudly without one."""
    Category = "processing"
    _segmenter = None

    @classmethod
    def set_segmenter(cls, fn):
        cls._segmenter = staticmethod(fn) if fn is not None else None

    def __call__(self, image):
        if RemoveBGNode._segmenter is None:
            raise RuntimeError("RemoveBGNode: no segmentation model available offline (the reference fetches skytnt/anime-seg "
                               "isnetis.onnx at run time); provide one with RemoveBGNode.set_segmenter(fn)")
        imgs = image if image.dim() == 4 else image.unsqueeze(0)
        out = []
        for im in imgs:
            mask = RemoveBGNode._segmenter(im[..., :3]).to(im.dtype).to(im.device).clamp(0, 1)[..., None]
            # img.py:47-53 works on uint8: white where the mask is 0, the mask as alpha
            rgb = ((mask * (im[..., :3] * 255).floor().clamp(0, 255) + 255 * (1 - mask)).floor().clamp(0, 255)) / 255.0
            out.append(torch.cat([rgb, ((mask * 255).floor() / 255.0)], -1).unsqueeze(0))
        return torch.cat(out, dim=0)


class RGBAToRGB(N.StableRenderingNode):
    """_nodes/processing/img.py:85-110"""
    Category = "processing"

    def __call__(self, image, color: str = "ffffff"):
        assert image.dim() >= 3
        assert image.shape[-1] == 4, "Input image must be in RGBA format"
        assert len(color) == 6, "Color must be a hex string"
        try:
            rgb = tuple(int(color[i:i + 2], 16) for i in [0, 2, 4])
        except ValueError:
            raise ValueError(f"Invalid color format {color}, color must be a hex string")
        background = torch.tensor(rgb).to(image.device)                 # (0..255 ints, as the reference: not normalised)
        c, alpha = image[..., :3], image[..., 3]
        return (1 - alpha[..., None]) * background + alpha[..., None] * c


class RGBAThreshold(N.StableRenderingNode):
    """_nodes/processing/img.py:113-131"""
    Category = "processing"

    def __call__(self, image, threshold: float = 0.5):
        assert image.dim() >= 3
        assert image.shape[-1] == 4, "Input image must be in RGBA format"
        mask = image[..., 3] > threshold
        return torch.cat([image[..., :3], mask[..., None]], dim=-1)


class TextConcat(N.StableRenderingNode):
    Category = "processing"

    def __call__(self, text_a: str, text_b: str) -> str:
        return text_a + text_b


class TextReplace(N.StableRenderingNode):
    Category = "processing"

    def __call__(self, text: str, pattern: str, replace: str) -> str:
        return text.replace(pattern, replace)


class SimpleVideoCombine(N.StableRenderingNode):
    """_nodes/processing/video.py:26-76: frames -> an animated GIF (the reference hands a ``UIImage`` to the web UI; headless the
    file is written under ``$SR_OUTPUT_DIR`` (default ./output) and its path returned).  alpha below the threshold is cut to 0."""
    Category = "video"
    IsOutputNode = True

    def __call__(self, images, alpha_threshold: float = 0.5, enable_alpha_threshold: bool = True, frame_rate: int = 8,
                 loop_count: int = 0, filename_prefix: str = "", pingpong: bool = False, save_output: bool = True, prompt=None,
                 extra_pnginfo=None):
        from PIL import Image
        frames = []
        for im in images:
            if enable_alpha_threshold:
                if im.shape[-1] == 4:
                    im = torch.cat([im[..., :3] * (im[..., 3:] > alpha_threshold), (im[..., 3:] > alpha_threshold).to(im.dtype)], -1)
                else:
                    im = torch.cat([im, torch.ones_like(im[..., :1])], dim=-1)
            a = np.clip(im.detach().float().cpu().numpy() * 255.0, 0, 255).astype(np.uint8)          # _tensor_to_bytes
            frames.append(Image.fromarray(a, mode="RGBA" if a.shape[-1] == 4 else "RGB"))
        if pingpong:
            frames = frames + frames[-2:0:-1]
        out_dir = os.path.join(os.environ.get("SR_OUTPUT_DIR", "output"), "" if save_output else "temp")
        os.makedirs(out_dir, exist_ok=True)
        i = 0
        while os.path.exists(os.path.join(out_dir, f"{filename_prefix}{i:05d}.gif")):
            i += 1
        path = os.path.join(out_dir, f"{filename_prefix}{i:05d}.gif")
        frames[0].save(path, save_all=True, append_images=frames[1:], duration=round(1000 / frame_rate), loop=loop_count, disposal=2)
        return path


# ---- legacy nodes (legacy_codes/nodes) -----------------------------------------------------------------------------------------------
class OverlapScheduler(N.StableRenderingNode):
    """legacy_codes/nodes/schedulers.py:7-39 (note its defaults: start_step 1, end_step 1000)"""
    Category = "scheduler"

    def __call__(self, every_step: int = 1, start_step: int = 1, end_step: int = 1000, start_timestep: int = 0, end_timestep: int = 1000,
                 interpolate_begin: float = 0.0, interpolate_end: float = 1.0, power: float = 1.0, interpolate_type='constant',
                 no_interpolate_return: float = 0.0):
        from .legacy_overlap import Scheduler
        return Scheduler(every_step=every_step, start_step=start_step, end_step=end_step, start_timestep=start_timestep,
                         end_timestep=end_timestep, interpolate_begin=interpolate_begin, interpolate_end=interpolate_end, power=power,
                         interpolate_type=interpolate_type, no_interpolate_return=no_interpolate_return)


class CorrespondenceMapLoader(N.StableRenderingNode):
    """the legacy graph (source/comfyUI/workflows/stable_renderer_ultimate.json) names this node; its source is not in the reference
    repository.  Here: the dict-based CorrespondenceMap (legacy_codes/.../correspondence_map.py:25-173) built from a directory of
    dumped id maps, as ``CorrespondenceMap.from_existing_directory_numpy`` does."""
    Category = "legacy_loader"

    def __call__(self, directory, num_frames=None, device="cuda"):
        from .legacy_overlap import CorrespondenceMap
        names = N._sorted_files(directory, (".npy",))
        if num_frames is not None:
            names = names[:num_frames]
        ids = torch.from_numpy(np.stack([np.load(os.path.join(directory, n)) for n in names]).astype(np.int32)).to(device)
        return CorrespondenceMap(ids)


class CorrMapLatentNoiseInitializer(N.StableRenderingNode):
    """legacy_codes/nodes/latent.py:7-40: latent and noise shared by all frames at corr-map resolution; every vertex seen more than
    once gets ONE fresh draw of 4 values (global generator, dict order, latent then noise per vertex) at all its pixels; nearest
    reduction to (height/8, width/8)"""
    Category = "Latent"

    def __call__(self, width: int, height: int, batch_size: int, seed: int, correspondence_map):
        cm = correspondence_map
        torch.manual_seed(seed)
        g = torch.manual_seed(seed + 1)
        W, H = cm.size
        dev = cm.pix_vert.device
        latent = torch.randn([1, 4, H, W], device="cpu", generator=g).repeat(batch_size, 1, 1, 1)
        noise = torch.randn([1, 4, H, W], device="cpu", generator=g).repeat(batch_size, 1, 1, 1)
        off = cm.offsets.cpu().numpy()
        multi = [int(v) for v in cm.order.cpu().numpy() if off[v + 1] - off[v] > 1]
        draws = torch.randn(len(multi), 2, 4)                           # per vertex: randn(4) for the latent, then randn(4) for the noise
        f, y, x = cm.tr_f.cpu().long(), cm.tr_y.cpu().long(), cm.tr_x.cpu().long()
        for j, v in enumerate(multi):
            s = slice(off[v], off[v + 1])
            latent[f[s], :, y[s], x[s]] = draws[j, 0]
            noise[f[s], :, y[s], x[s]] = draws[j, 1]
        h, w = height // 8, width // 8
        iy = torch.clamp((torch.arange(h).float() * (H / h)).floor().long(), max=H - 1)
        ix = torch.clamp((torch.arange(w).float() * (W / w)).floor().long(), max=W - 1)
        pick = lambda t: t[:, :, iy][:, :, :, ix].contiguous().to(dev)
        return LATENT(samples=pick(latent), noise=pick(noise))


class StableRenderSampler(N.StableRenderingNode):
    """legacy_codes/nodes/samplers.py:16-144: KSampler whose step callback runs the legacy ResizeOverlap on the noisy latent (and /
    or, with ddpm, on the denoised prediction); samplers other than ddim / ddpm fall back to ddpm as in the reference"""
    Category = "sampling"

    def __call__(self, model, positive, negative, latent_image, correspondence_map, alpha_scheduler, kernel_radius_scheduler,
                 overlap_algorithm="average", noise_option='default', apply_overlap_option='noise', noise_seed: int = 0,
                 steps: int = 20, cfg: float = 8.0, sampler_name="euler", scheduler="normal", denoise: float = 1.0):
        from . import legacy_overlap as LO
        if sampler_name not in ["ddim", "ddpm"]:
            sampler_name = "ddpm"
        algos = {"average": LO.AverageDistance, "frame_distance": LO.FrameDistance, "pixel_distance": LO.PixelDistance,
                 "perpendicular_view_normal": LO.PerpendicularViewNormal}
        if overlap_algorithm not in algos:
            raise ValueError(f"Unknown overlap algorithm: {overlap_algorithm}")
        overlap = LO.ResizeOverlap(alpha_scheduler=alpha_scheduler, kernel_radius_scheduler=kernel_radius_scheduler,
                                   algorithm=algos[overlap_algorithm](), verbose=False)

        def run_on(t, ctx):
            est = 1000 - int(((ctx.step_index + 1) / ctx.total_steps) * 1000)
            seq = overlap([fr.unsqueeze(0) for fr in t], corr_map=correspondence_map, step=ctx.step_index, timestep=est)
            for i, fr in enumerate(seq):
                t[i].copy_(fr.reshape(t[i].shape))

        def execute_overlap(ctx):
            if apply_overlap_option not in ('noise', 'denoised', 'both'):
                raise ValueError(f"Unknown apply_overlap_option: {apply_overlap_option}")
            if apply_overlap_option == 'noise' or sampler_name != "ddpm":
                run_on(ctx.noise, ctx)
            elif apply_overlap_option == 'denoised':
                run_on(ctx.denoised, ctx)
            else:
                run_on(ctx.noise, ctx)
                run_on(ctx.denoised, ctx)
        opt = {'default': 'random'}.get(noise_option, noise_option)
        return N.custom_ksampler(model, noise_seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image,
                                 denoise=denoise, noise_option=opt, callbacks=[execute_overlap])[0]


ALL = dict(CreateIdenticalNoiseSequence=CreateIdenticalNoiseSequence, CreateNoiseSequenceFromIdMap=CreateNoiseSequenceFromIdMap,
           LegacyImageSequenceLoader=LegacyImageSequenceLoader, LegacyNoiseSequenceLoader=LegacyNoiseSequenceLoader,
           LegacyIDSequenceLoader=LegacyIDSequenceLoader, RemoveBGNode=RemoveBGNode, RGBAToRGB=RGBAToRGB, RGBAThreshold=RGBAThreshold,
           TextConcat=TextConcat, TextReplace=TextReplace, SimpleVideoCombine=SimpleVideoCombine, OverlapScheduler=OverlapScheduler,
           CorrespondenceMapLoader=CorrespondenceMapLoader, CorrMapLatentNoiseInitializer=CorrMapLatentNoiseInitializer,
           StableRenderSampler=StableRenderSampler)
