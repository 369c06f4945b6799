"""Import alias: the package directory is `pano-nerf_amd/` (the name the layout prescribes; not a valid Python
identifier), so `import pano_nerf_amd` loads that directory as a regular package under this name — with a proper
`__spec__`, `__file__` and `__path__` pointing at the real directory (nothing is exec'd into this stub)."""
import importlib.util as _u
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pano-nerf_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod  # `import pano_nerf_amd` returns the real package from here on
_spec.loader.exec_module(_mod)
