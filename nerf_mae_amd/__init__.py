"""Import shim: the product package lives in the directory `nerf-mae_amd/` (not a valid Python
identifier); this module makes it importable as `nerf_mae_amd`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nerf-mae_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
