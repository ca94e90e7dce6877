"""Import-path shim: the reference's scripts do ``sys.path.append(<repo>/source)`` and then ``from engine.engine import Engine``,
``from engine.runtime.components import Camera, ...``, ``from engine.static import Mesh, Texture, ...``,
``from common_utils.path_utils import *`` (scripts/*.py:1-13).  ``compat/source`` holds packages with those names that re-export
this package's headless implementations, so a reference script runs unchanged once this directory is on ``sys.path`` in place of
the reference's ``source`` (``install()`` puts it first)."""
import os
import sys

SOURCE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "source")
FALLBACK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fallback")     # stand-ins used only when the real package is absent


def install():
    if SOURCE not in sys.path:
        sys.path.insert(0, SOURCE)
    if FALLBACK not in sys.path:
        sys.path.append(FALLBACK)
    for name in ("engine", "common_utils"):                      # a reference `source` imported earlier would shadow the shim
        m = sys.modules.get(name)
        if m is not None and not getattr(m, "__file__", "").startswith(SOURCE):
            for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
                del sys.modules[k]
    return SOURCE
