"""Port of the reference's scripts/boat_example.py (BASELINE config 3's scene) through the same script API and import paths:
a mesh from ``Mesh.Load`` with diffuse + normal + noise textures on a DefaultOpaqueMaterial, turning under ``AutoRotation``, camera
(0,3,-3) -> origin; GAME mode = one diffusion call per frame through a shipped workflow graph.  Assets that cannot travel are
synthetic (random textures; the reference's boat.obj is used when $SR_RESOURCES_DIR points at its resources)."""
import argparse
import os

import _common as C  # noqa: F401  (installs the import-path shim)
import torch

from engine.runtime.components import Camera, MeshRenderer
from engine.runtime.gameObj import GameObject
from engine.runtime.component import Component
from engine.engine import Engine
from engine.static import Material, DefaultTextureType, Mesh, Texture
from common_utils.path_utils import EXAMPLE_WORKFLOWS_DIR


class AutoRotation(Component):
    def update(self):
        self.transform.rotateLocalY(2.5 * self.engine.RuntimeManager.DeltaTime)


class Sample(Engine):
    def beforePrepare(self):
        boatMesh = Mesh.Load(C.boat_mesh_path())
        boatMaterial = Material.DefaultOpaqueMaterial()
        g = torch.Generator().manual_seed(4)
        diffuse = torch.rand(64, 64, 3, generator=g)
        nmap = torch.rand(64, 64, 3, generator=g)
        nmap[..., 2] = 0.7 + 0.3 * nmap[..., 2]
        boatMaterial.addDefaultTexture(Texture(data=diffuse), DefaultTextureType.DiffuseTex)
        boatMaterial.addDefaultTexture(Texture(data=nmap), DefaultTextureType.NormalTex)
        boatMaterial.addDefaultTexture(Texture.CreateNoiseTex(), DefaultTextureType.NoiseTex)

        boat = GameObject('Boat', position=[0, 0, 0])
        boat.addComponent(MeshRenderer, mesh=boatMesh, materials=boatMaterial)
        boat.addComponent(AutoRotation)

        camera = GameObject('Camera', position=[0, 3, -3])
        camera.addComponent(Camera)
        camera.transform.lookAt([0, 0, 0])


def main(frames=4, tiny=False, size=512, diffuse=True):
    if diffuse:
        C.register_synthetic_models(tiny=tiny)
    return Sample.Run(winSize=(size, size), mapSavingInterval=1, needOutputMaps=False, outputAICannyMap=False, saveSDColorOutput=False,
                      disableComfyUI=not diffuse, max_frames=frames, diffuse_workflow=EXAMPLE_WORKFLOWS_DIR / 'no-control-bake.json')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--no-diffuse", action="store_true")
    a = ap.parse_args()
    e = main(a.frames, a.tiny, diffuse=not a.no_diffuse)
    torch.cuda.synchronize()
    print("frames:", e.RuntimeManager.FrameCount, "diffusion calls:", len(e.outputs))
