"""Import shim: the product package lives in the directory `self-driving-model_amd/` (the name the
project layout prescribes), which is not a valid Python identifier.  This module makes it importable
as `self_driving_model_amd` (and its sub-packages as `self_driving_model_amd.models...`)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "self-driving-model_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
