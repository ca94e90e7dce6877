"""TEST INFRASTRUCTURE ONLY (container-side): import harness for the *reference* Python sources.

Used only by ``oracle/gen_golden.py`` to generate the golden vectors committed under ``tests/golden``.
It never runs on the GPU box (``/root/reference`` does not exist there) and nothing in the product
package imports it.  Recipe follows SURVEY.md Appendix B: third-party *non-arithmetic* dependencies that
are absent from this image (GL, GLFW, Qt, taichi, ...) are served as inert stub modules so that the
reference's pure-torch arithmetic (math_utils, corrmap, corresponder, comfy samplers, UNet, VAE) can be
imported and executed on CPU.  torch / einops / scipy / numpy are the real packages, so every number that
comes out of the reference through this harness is authoritative.
"""
import importlib.abc
import importlib.machinery
import logging
import os
import sys
import types

REF = os.environ.get("SR_REFERENCE_ROOT", "/root/reference")

_STUB_ROOTS = {
    "dotenv", "colorama", "typeguard", "taichi", "glm", "glfw", "OpenGL", "pycuda", "cuda", "assimp_py",
    "torchvision", "deprecated", "torchsde", "cv2", "onnxruntime", "numba", "xformers", "kornia",
    "omegaconf", "trampoline", "concurrent_log_handler", "PySide6", "diffusers", "imageio", "pynput",
}


class _Anything:
    """Attribute sink: every attribute / call / index yields another sink; usable as decorator."""

    def __init__(self, name="stub"):
        self.__dict__["_n"] = name

    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        return _Anything(self._n + "." + k)

    def __call__(self, *a, **k):
        # pass-through decorator behaviour: f = deco(f) and f = deco(...)(f)
        if len(a) == 1 and not k and (isinstance(a[0], (types.FunctionType, type))):
            return a[0]
        return _Anything(self._n + "()")

    def __getitem__(self, k):
        return _Anything(self._n + "[]")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)

    def __or__(self, o):
        return self

    def __ror__(self, o):
        return self

    def __bool__(self):
        return False

    def __str__(self):
        return ""

    def __repr__(self):
        return "<stub %s>" % self._n


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__") and k.endswith("__"):
            raise AttributeError(k)
        v = _Anything(self.__name__ + "." + k)
        setattr(self, k, v)
        return v


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        name = module.__name__
        if name == "concurrent_log_handler":
            class ConcurrentTimedRotatingFileHandler(logging.Handler):
                def __init__(self, *a, **k):
                    logging.Handler.__init__(self)

                def emit(self, record):
                    pass
            module.ConcurrentTimedRotatingFileHandler = ConcurrentTimedRotatingFileHandler
        elif name == "PySide6.QtCore":
            class Signal:
                def __init__(self, *a, **k):
                    self._s = []

                def connect(self, f):
                    self._s.append(f)

                def disconnect(self, f=None):
                    self._s = [] if f is None else [g for g in self._s if g is not f]

                def emit(self, *a, **k):
                    for f in list(self._s):
                        f(*a, **k)
            module.QObject = object
            module.Signal = Signal
        elif name == "colorama":
            class _E:
                def __getattr__(self, k):
                    return ""
            module.Fore = module.Style = module.Back = _E()
            module.init = lambda *a, **k: None
        elif name == "taichi":
            module.kernel = lambda f: f
            module.func = lambda f: f
        elif name == "dotenv":
            module.load_dotenv = lambda *a, **k: False


_installed = False


def install():
    """Make ``import common_utils...`` / ``import comfy...`` resolve to the reference sources (CPU only)."""
    global _installed
    if _installed:
        return
    _installed = True
    sys.dont_write_bytecode = True          # never write into /root/reference
    os.environ.setdefault("DEV_MODE", "1")
    os.environ.setdefault("HOME", "/tmp/sr_oracle_home")
    os.makedirs(os.environ["HOME"], exist_ok=True)
    sys.argv = ["oracle", "--cpu"]
    sys.meta_path.insert(0, _StubFinder())
    sys.path[:0] = [os.path.join(REF, "source"), os.path.join(REF, "source", "comfyUI")]
    import comfy.options
    comfy.options.enable_args_parsing()


def import_corrmap():
    """engine.static.corrmap without engine/static/__init__.py (which pulls in OpenGL/GLM)."""
    install()
    if "engine.static.corrmap" in sys.modules:
        return sys.modules["engine.static.corrmap"]
    import engine  # noqa: F401  (namespace/package root)
    pkg = types.ModuleType("engine.static")
    pkg.__path__ = [os.path.join(REF, "source", "engine", "static")]

    class Texture:  # placeholder type, only used in isinstance checks
        pass
    pkg.Texture = Texture
    sys.modules["engine.static"] = pkg
    import engine.static.enums  # noqa: F401
    import engine.static.resources_obj  # noqa: F401
    import engine.static.corrmap as cm
    return cm
