"""Import shim: the package lives in ``driving-dirty_amd/`` (a name Python cannot import
directly because of the hyphen).  ``import driving_dirty_amd`` resolves to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "driving-dirty_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
