"""Import alias: the package directory is `pano-nerf_amd/` (not a valid Python identifier), so this
thin module re-points `pano_nerf_amd` at it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pano-nerf_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
