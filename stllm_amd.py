"""Import shim: the package directory is ``st-llm_amd/`` (layout contract), which is not a
valid Python identifier — this module makes it importable as ``stllm_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "st-llm_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
