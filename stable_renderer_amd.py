"""Import alias: the package directory is ``stable-renderer_amd/`` (repo layout contract); a hyphen is not
a legal Python identifier, so this module turns itself into that package (``__path__`` + exec of its
``__init__``).  ``import stable_renderer_amd.raster`` etc. then resolve inside ``stable-renderer_amd/``."""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg = _os.path.join(_here, "stable-renderer_amd")
__path__ = [_pkg]
__file__ = _os.path.join(_pkg, "__init__.py")
with open(__file__, "r") as _f:
    exec(compile(_f.read(), __file__, "exec"), globals())
