"""Import shim: the package directory is ``enf-pde_amd/`` (not a Python identifier).

``import enf_pde_amd`` resolves to that directory as a regular package
(``enf_pde_amd.enf.models``, ``enf_pde_amd.fitting`` ...).
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "enf-pde_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
