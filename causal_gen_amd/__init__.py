"""Import shim: the package lives in ``causal-gen_amd/`` (the name the build contract fixes), which is not a
valid Python identifier.  ``import causal_gen_amd.vae`` resolves into that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "causal-gen_amd")]
exec(open(_os.path.join(__path__[0], "__init__.py")).read())
