"""Import alias: the product package lives in ``ltx-2-mlx_amd/`` (hyphenated, as the repo layout
requires), which Python cannot import by name.  This stub makes ``import ltx_2_mlx_amd`` resolve
to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ltx-2-mlx_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
