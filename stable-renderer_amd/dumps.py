"""G-buffer dump layout of the reference (DiffusionManager._outputMap / _outputNumpyData / _outputDepthMap,
engine/managers/diffusionManager.py:160-259, call site renderManager.py:966-988): ``<root>/<name>/<name>_<frame>.{png,npy}``;
colour / normal / canny as RGBA8 PNG (value*255 truncated, alpha 255 added to 3-channel maps, 1- or 2-D maps repeated to
grey), id / pos / noise as raw ``.npy``, depth min-max normalised over its positive values with alpha = depth > 0.
Dumps written here load with the reference's ``IDSequenceLoader`` / ``NoiseSequenceLoader`` / ``ImageSequenceLoader`` and vice
versa (``nodes.py`` has the loader mirrors).  Host-side file I/O only (numpy + PIL): the planes come off the device once."""
import os

import numpy as np
from PIL import Image


class GBufferDump:
    def __init__(self, output_path):
        self.output_path = output_path

    def _dir(self, name):
        d = os.path.join(self.output_path, name)
        os.makedirs(d, exist_ok=True)
        return d

    def output_numpy(self, name, data, frame_num=None):
        fn = f"{name}_{frame_num}.npy" if frame_num is not None else f"{name}.npy"
        np.save(os.path.join(self._dir(name), fn), np.asarray(data))

    def output_map(self, name, map_data, multi255=True, data_type=np.uint8, frame_num=None):
        m = np.asarray(map_data)
        if multi255:
            m = m * 255
        if m.ndim == 2:
            m = np.repeat(m[:, :, None], 3, axis=2)
        elif m.shape[2] == 1:
            m = np.repeat(m, 3, axis=2)
        if m.shape[2] == 3:
            m = np.concatenate([m, np.ones((m.shape[0], m.shape[1], 1), dtype=data_type) * 255], axis=2)
        name = name.lower()
        fn = f"{name}_{frame_num}.png" if frame_num is not None else f"{name}.png"
        Image.fromarray(m.astype(data_type), "RGBA").save(os.path.join(self._dir(name), fn))

    def output_depth(self, depth, frame_num=None):
        d = np.asarray(depth)
        dmax, dmin = np.max(d), np.min(d[d > 0])
        diff = dmax - dmin
        dn = (d - dmin) / diff if diff != 0 else d
        gray = (np.clip(dn, 0, 1) * 255).astype(np.uint8)
        alpha = (dn > 0).astype(np.uint8) * 255
        fn = f"depth_{frame_num}.png" if frame_num is not None else "depth.png"
        Image.fromarray(np.stack([gray, gray, gray, alpha], axis=-1), "RGBA").save(os.path.join(self._dir("depth"), fn))

    def dump_gbuffer(self, gbuf, frame_num):
        """one frame of ``scene.GBuffer`` planes, as RenderManager does when ShouldOutputFrame (renderManager.py:966-988)"""
        nd = gbuf.normal_depth.float().cpu().numpy()
        self.output_map("color", gbuf.color.float().cpu().numpy(), frame_num=frame_num)
        self.output_map("normal", nd[:, :, :3], frame_num=frame_num)
        self.output_map("canny", gbuf.canny.float().cpu().numpy(), frame_num=frame_num)
        self.output_numpy("id", gbuf.id.cpu().numpy(), frame_num)
        self.output_numpy("pos", gbuf.pos.cpu().numpy(), frame_num)
        self.output_numpy("noise", gbuf.noise.cpu().numpy(), frame_num)
        self.output_depth(nd[:, :, 3], frame_num)
