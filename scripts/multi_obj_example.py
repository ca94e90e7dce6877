"""Port of the reference's scripts/multi_obj_example.py (BASELINE config 5's scene, rasterisation only as shipped:
``disableComfyUI=True``): a loaded mesh, a textured sphere and a textured plane in one scene, ``CameraController`` start pose."""
import argparse

import _common as C  # noqa: F401
import torch

from engine.runtime.components import Camera, MeshRenderer
from engine.runtime.gameObj import GameObject
from engine.runtime.component import Component
from engine.engine import Engine
from engine.runtime.components import CameraController
from engine.static import Mesh, Texture, Material
from engine.static.enums import DefaultTextureType


class AutoRotation(Component):
    def update(self):
        self.transform.rotateLocalY(4 * self.engine.RuntimeManager.DeltaTime)


class Sample(Engine):
    def beforePrepare(self):
        mesh = Mesh.Load(C.boat_mesh_path(), alias='miku')

        camera = GameObject('Camera')
        camera.addComponent(Camera)
        camera.addComponent(CameraController, defaultPos=[4.0, 3.5, 4.0], defaultLookAt=[0, 0.4, 0])

        obj = GameObject('miku', position=[0, 0, 0], scale=[1.0, 1.0, 1.0])
        meshRenderer = obj.addComponent(MeshRenderer, mesh=mesh)
        meshRenderer.addMaterial(Material.DefaultOpaqueMaterial())
        obj.addComponent(AutoRotation)

        debug_mat = Material.DefaultOpaqueMaterial()
        checker = (torch.arange(64)[:, None] // 8 + torch.arange(64)[None, :] // 8) % 2
        debug_mat.addDefaultTexture(Texture(data=torch.stack([checker, 1 - checker, checker * 0 + 0.5], -1).float()),
                                    DefaultTextureType.DiffuseTex)

        ball = GameObject('ball', position=[-1.5, 0.5, 1.0], scale=[0.5, 0.5, 0.5])
        ball_meshRenderer = ball.addComponent(MeshRenderer, mesh=Mesh.Sphere())
        ball_meshRenderer.addMaterial(debug_mat)

        plane = GameObject('plane', position=[0, 0, 0], scale=[5, 5, 5])
        plane_meshRenderer = plane.addComponent(MeshRenderer, mesh=Mesh.Plane())
        plane_meshRenderer.addMaterial(debug_mat)


def main(frames=3, size=512):
    return Sample.Run(debug=False, winSize=(size, size), mapSavingInterval=4, disableComfyUI=True, needOutputMaps=False,
                      disable_cuda_gl_share=True, max_frames=frames)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    a = ap.parse_args()
    e = main(a.frames)
    torch.cuda.synchronize()
    ed = e.outputs[-1]
    print("frames:", e.RuntimeManager.FrameCount, "sprites/materials on screen:", sorted(set(map(tuple, ed.id_maps.tensor[0, ..., :2].reshape(-1, 2).tolist()))))
