"""IDMap / CorrespondMap with the reference's names and semantics (engine/static/corrmap.py:48-886), backed by HIP
kernels.  Tensors live in HBM; the only host work is argument normalisation and the error behaviour the reference
exhibits (IndexError on out-of-range ids, the double-gather quirk of ``_update``)."""
import json
import os
import re
import zipfile
from typing import Literal, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops as O

UpdateMode = Literal['replace', 'replace_avg', 'first', 'first_avg']


DEFAULT_DEVICE = "cuda"          # a dry-run Engine (no GPU: scene inspection only) switches this to "cpu"


class IDMap:
    """(N, H, W, 4) int32 ``(spriteID, materialID, map_index, vertexID)``; mask = map_index==2048 or all-zero
    (corrmap.py:119-126).  NB the reference's ``height``/``width`` properties return ``shape[-2]``/``shape[-1]``
    (= W and 4 for NHWC, corrmap.py:85-93); kept for drop-in compatibility."""

    def __init__(self, tensor: torch.Tensor, frame_indices=None, masks: Optional[torch.Tensor] = None):
        if tensor.dim() == 3:
            tensor = tensor.unsqueeze(0)
        if tensor.dim() != 4:
            raise ValueError("Invalid shape of real ID tensor.")
        self.tensor = tensor.to(torch.int32).contiguous()
        if frame_indices is None:
            frame_indices = list(range(self.tensor.shape[0]))
        elif isinstance(frame_indices, int):
            frame_indices = [frame_indices]
        self.frame_indices = list(frame_indices)
        if masks is None:
            masks = O.idmap_masks(self.tensor) if self.tensor.is_cuda else None
            if masks is None:
                raise L.SrHipError("IDMap: id tensor must live on the GPU (no CPU fallback in the product path)")
        elif masks.dim() == 2:
            masks = torch.stack([masks] * self.frame_count, dim=0)
        self.masks = masks
        self._overlap_index = {}

    @property
    def frame_count(self):
        return len(self.frame_indices)

    def __len__(self):
        return self.frame_count

    def __getitem__(self, i):
        return self.tensor[i]

    @property
    def height(self):
        return self.tensor.shape[-2]

    @property
    def width(self):
        return self.tensor.shape[-1]

    @classmethod
    def from_directory(cls, directory, frame_start=None, num_frames=None, use_frame_indices_from_filename=True, device="cuda"):
        """corrmap.py:138-198: stack ``*.npy`` id maps of a dump directory (sorted by the number in the file name)."""
        assert os.path.exists(directory)
        frame_start = frame_start or 0

        def idx_of(name, default):
            m = re.findall(r"\d+", os.path.splitext(name)[0])
            return int(m[-1]) if m else default
        names = [f for f in os.listdir(directory) if f.endswith(".npy")]
        names = sorted(names, key=lambda n: idx_of(n, names.index(n)))
        frame_indices = [idx_of(n, -1) for n in names] if use_frame_indices_from_filename else list(range(len(names)))
        num_frames = num_frames or len(frame_indices)
        frame_indices = frame_indices[frame_start: frame_start + num_frames]
        assert all(i != -1 for i in frame_indices), "Illegal filename(s) found."
        ts = []
        for n in names[frame_start: frame_start + num_frames]:
            t = torch.from_numpy(np.load(os.path.join(directory, n))).squeeze()
            if t.dim() != 3 or not (t.shape[-1] == 4 or t.shape[1] == 4):
                raise ValueError(f"Invalid id tensor shape: {t.shape}.")
            ts.append(t)
        if not ts:
            raise ValueError("No valid id data found.")
        if any(t.shape != ts[0].shape for t in ts):
            raise ValueError("Tensor data has inconsistent shapes.")
        return cls(torch.stack(ts, 0).to(device), frame_indices=frame_indices)

    def overlap_index(self, lh, lw):
        """cached device structure replacing ``create_vertex_screen_info`` (corrmap.py:220-280) + per-step unique"""
        key = (lh, lw)
        if key not in self._overlap_index:
            if self.frame_indices != list(range(self.frame_count)):
                # the reference uses frame_indices as *batch indices* of the latent (corresponder.py:314,323)
                raise IndexError("frame_indices must be 0..N-1 to index the latent batch")
            self._overlap_index[key] = O.OverlapIndex(self.tensor, lh, lw)
        return self._overlap_index[key]


class CorrespondMap:
    """values (k*k, H*W, C) fp16 + written flags (corrmap.py:372-412), update() = corrmap.py:578-736."""

    def __init__(self, k=3, height=512, width=512, channel_count=4, name=None, device=None):
        if channel_count != 4:
            raise ValueError("only RGBA corr-maps are supported by the HIP path")
        self.k, self.height, self.width, self.channel_count, self.name = k, height, width, channel_count, name
        self.device = torch.device(device or DEFAULT_DEVICE)
        self._values = torch.zeros(k * k, height * width, channel_count, dtype=torch.float16, device=self.device)
        self._writtens = torch.zeros(k * k, height * width, dtype=torch.uint8, device=self.device)
        self._winner = torch.empty(k * k * height * width, dtype=torch.int32, device=self.device)
        self._err = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __getitem__(self, i):
        return self._values[i]

    def clear(self):
        self._values.zero_()
        self._writtens.zero_()

    @property
    def writtens(self):
        return self._writtens.bool()

    # ---- on-disk format of the reference (corrmap.py:738-872): k*k PNGs + "<i>_written.png" + meta.json ----------------
    def get_map(self, i):
        return self._values[i].view(self.height, self.width, self.channel_count)

    def get_written_flag_map(self, i):
        return self._writtens[i].view(self.height, self.width)

    def dump(self, path, name=None, zip=False, force=False):
        from PIL import Image
        name = name or self.name or "corrmap"
        real_name, suffix = name, ('.zip' if zip else '')
        if not force:
            count = 1
            while os.path.exists(os.path.join(path, real_name + suffix)):
                real_name = f"{name}_{count}"
                count += 1
        work = os.path.join(path, real_name + ("_tmp" if zip else ""))
        os.makedirs(work, exist_ok=True)
        files = []
        for i in range(self.k * self.k):
            img = 255. * self.get_map(i).cpu().numpy()            # fp16 numpy product, as the reference (rounds in fp16)
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), mode='RGBA').save(os.path.join(work, f"{i}.png"))
            wr = 255.0 * self.get_written_flag_map(i).float().cpu().numpy()
            Image.fromarray(np.clip(wr, 0, 255).astype(np.uint8), mode='L').save(os.path.join(work, f"{i}_written.png"))
            files += [f"{i}.png", f"{i}_written.png"]
        with open(os.path.join(work, 'meta.json'), 'w') as f:
            json.dump({"k": self.k, "height": self.height, "width": self.width, "channel_count": self.channel_count, "name": name}, f)
        files.append('meta.json')
        if not zip:
            return work
        real_path = os.path.join(path, real_name + suffix)
        with zipfile.ZipFile(real_path, 'w') as z:
            for fn in files:
                z.write(os.path.join(work, fn), fn)
                os.remove(os.path.join(work, fn))
        os.rmdir(work)
        return real_path

    @classmethod
    def Load(cls, path, name=None, device=None):
        import io
        from PIL import Image
        if os.path.isfile(path):
            z = zipfile.ZipFile(path, 'r')
            rd = lambda fn: io.BytesIO(z.read(fn))
        else:
            rd = lambda fn: open(os.path.join(path, fn), 'rb')
        meta = json.load(rd('meta.json'))
        m = cls(name=name or meta['name'], k=meta['k'], height=meta['height'], width=meta['width'],
                channel_count=meta['channel_count'], device=device)
        for i in range(m.k * m.k):
            img = torch.tensor(np.array(Image.open(rd(f"{i}.png"))), dtype=torch.float32) / 255.
            m._values[i] = img.view(-1, m.channel_count).to(m._values.dtype).to(m.device)
            wr = torch.tensor(np.array(Image.open(rd(f"{i}_written.png"))), dtype=torch.float32) / 255.
            m._writtens[i] = wr.bool().view(-1).to(torch.uint8).to(m.device)
        return m

    def update(self, color_frames, id_maps, spriteID=None, materialID=None, mode: UpdateMode = 'first_avg', masks=None,
               inverse_masks=False, ignore_obj_mat_id=False):
        if isinstance(id_maps, IDMap):
            id_maps = id_maps.tensor
        if isinstance(color_frames, (list, tuple)):
            color_frames = torch.stack(list(color_frames), 0)
        if isinstance(id_maps, (list, tuple)):
            id_maps = torch.stack([i.tensor if isinstance(i, IDMap) else i for i in id_maps], 0)
        if color_frames.dim() == 3:
            color_frames = color_frames.unsqueeze(0)
        if color_frames.dim() != 4:
            raise ValueError("The shape of color_frames is invalid. Got: ", color_frames.shape)
        if id_maps.dim() != 4:
            # a 3-D id map never returns in the reference (corrmap.py:629 appends to the list it iterates)
            raise ValueError("The shape of id_maps is invalid. Got: ", id_maps.shape)
        if masks is not None:
            if not isinstance(masks, torch.Tensor):
                raise ValueError("Invalid type of masks. Got: ", type(masks))
            if masks.dim() == 4 and masks.shape[-1] == 1:
                masks = masks.squeeze(-1)
            if masks.dim() == 2:
                masks = masks.unsqueeze(0)
            if masks.dim() != 3:
                raise ValueError("The shape of masks is invalid. Got: ", masks.shape)
            masks = masks.to(self.device, torch.float32)
            if inverse_masks:
                masks = 1 - masks
        if len(color_frames) != len(id_maps):
            raise ValueError(f"The length of color_frames and id_maps should be the same, but got: {len(color_frames)} and {len(id_maps)}")
        if masks is not None and len(masks) != len(color_frames):
            raise ValueError(f"The length of masks should be the same as color_frames, but got: {len(masks)} and {len(color_frames)}")
        for f in range(len(color_frames)):
            self._update(color_frames[f], id_maps[f], spriteID, materialID, mode, None if masks is None else masks[f],
                         ignore_obj_mat_id)

    def _update(self, color_frame, id_map, spriteID, materialID, mode, mask, ignore_obj_mat_id):
        dev = self.device
        col = color_frame.to(dev, torch.float32)
        if col.shape[-1] > 4:
            col = col[..., :4]
        col = col.reshape(-1, col.shape[-1]).contiguous()
        ids = id_map.to(dev, torch.int32).reshape(-1, 4).contiguous()
        n = ids.shape[0]
        src_index = None
        chk_s = int((not ignore_obj_mat_id) and spriteID is not None)
        chk_m = int((not ignore_obj_mat_id) and materialID is not None)
        if mask is not None:
            mask = mask.reshape(-1).contiguous()
            if not ignore_obj_mat_id:
                # reference quirk (corrmap.py:703 then :710): the colour rows are first compacted by the mask and then
                # re-indexed with ORIGINAL pixel indices.  Row i therefore reads colour[R[i]] with R = the compacted
                # pixel list, and the reference raises IndexError as soon as a surviving row has i >= len(R).
                R = torch.nonzero(mask > 0).reshape(-1).to(torch.int32)
                M = R.numel()
                keep = mask > 0
                if chk_s:
                    keep = keep & (ids[:, 0] == spriteID)
                if chk_m:
                    keep = keep & (ids[:, 1] == materialID)
                surv = torch.nonzero(keep).reshape(-1)
                if surv.numel() and int(surv.max()) >= M:
                    raise IndexError(f"index {int(surv.max())} is out of bounds for dimension 0 with size {M}")
                src_index = torch.zeros(n, dtype=torch.int32, device=dev)
                src_index[:M] = R
        self._err.zero_()
        L.check(L.lib().sr_corrmap_update(O._p(col), col.shape[-1], O._p(ids), O._p(mask), O._p(src_index), n,
                                          int(spriteID or 0), int(materialID or 0), chk_s, chk_m,
                                          int(mode in ("first", "first_avg")), O._p(self._values), O._p(self._writtens),
                                          self.k * self.k, self.height * self.width, O._p(self._winner), O._p(self._err),
                                          O.stream_ptr()))
        if int(self._err.item()):
            raise IndexError("map_index / vertexID out of range for this CorrespondMap")
