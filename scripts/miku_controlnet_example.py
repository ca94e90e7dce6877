"""Port of the reference's scripts/miku_controlnet_example.py (BASELINE config 4's scene): a loaded mesh with a noise texture on
every material, a sprite prompt, an ``EnvPrompt`` on the camera, ``CameraController`` start pose, and the ``miku-control.json`` graph
(depth + normal ControlNets driven by the G-buffers, KSampler lcm).  ``miku.obj`` is not in the reference repository: the
boat-shaped mesh stands in."""
import argparse

import _common as C  # noqa: F401
import torch

from engine.runtime.gameObj import GameObject
from engine.runtime.component import Component
from engine.runtime.components import Camera, MeshRenderer, CameraController, SpriteInfo
from engine.engine import Engine
from engine.static import Mesh, Material, Texture, DefaultTextureType
from common_utils.path_utils import EXAMPLE_WORKFLOWS_DIR
from common_utils.stable_render_utils import EnvPrompt


class AutoRotation(Component):
    def update(self):
        self.transform.rotateLocalY(2.5 * self.engine.RuntimeManager.DeltaTime)


class Sample(Engine):
    def beforePrepare(self):
        mesh = Mesh.Load(C.boat_mesh_path(), alias='miku', cullback=False)
        mats = [Material.DefaultOpaqueMaterial(real_name=m["NAME"]) for m in mesh.materials] or [Material.DefaultOpaqueMaterial()]
        noise_map = Texture.CreateNoiseTex('miku noise map', 512, 512)
        for mat in mats:
            mat.addDefaultTexture(noise_map, DefaultTextureType.NoiseTex)

        env_prompt = EnvPrompt('no background', negative_prompt="watermark")
        camera = GameObject('Camera')
        camera.addComponent(Camera, bgPrompt=env_prompt)
        camera.addComponent(CameraController, defaultPos=[2.6, 2.2, 2.6], defaultLookAt=[0, 0.3, 0])

        miku = GameObject('miku', position=[0, 0, 0], scale=[1.0, 1.0, 1.0])
        meshRenderer = miku.addComponent(MeshRenderer, mesh=mesh)
        if mesh.materials:
            meshRenderer.load_MTL_Materials(mats)
        else:
            meshRenderer.addMaterial(mats[0])
        miku.addComponent(AutoRotation)
        miku.addComponent(SpriteInfo, auto_spriteID=True, prompt='miku, 1 girl, anime, waifu, long blue hair')


def main(frames=3, tiny=False, size=512):
    C.register_synthetic_models(tiny=tiny)
    return Sample.Run(winSize=(size, size), needOutputMaps=False, saveSDColorOutput=False, disableComfyUI=False, verbose=False,
                      max_frames=frames, diffuse_workflow=EXAMPLE_WORKFLOWS_DIR / 'miku-control.json')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--tiny", action="store_true")
    a = ap.parse_args()
    e = main(a.frames, a.tiny)
    torch.cuda.synchronize()
    print("frames:", e.RuntimeManager.FrameCount, "diffusion calls:", len(e.outputs), tuple(e.outputs[-1].shape))
