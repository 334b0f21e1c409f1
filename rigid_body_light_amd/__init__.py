"""rigid_body_light_amd -- MI355X-native blob-mobility hot path behind the
Rigid_Body_Light operator surface.

    from rigid_body_light_amd import RigidBody        # same class surface as
    from Rigid import RigidBody                       # the reference's package

The compute lives in librbl.so (hand-written HIP for gfx950, C ABI in
include/rbl.h); `c_rigid` is the pybind11 shim exposing `CManyBodies` with the
reference's method names.  There is NO CPU fallback: if the native libraries
are missing the import fails, and compute calls raise RuntimeError when no HIP
device is present.
"""
import os as _os

# PyTorch wheels bundle their own HIP/HSA runtime (torch/lib/libamdhip64.so, soname
# libamdhip64.so.7).  A process must hold exactly ONE HIP runtime: if librbl.so pulled in
# /opt/rocm's copy first, torch could no longer see the GPU.  Importing torch first makes
# its runtime the one librbl's NEEDED libamdhip64.so.7 resolves to.  (A pure C/C++ user of
# librbl.so without torch simply gets the system runtime.)
try:
    import torch as _torch  # noqa: F401
except ImportError:  # torch is plumbing (device memory / streams / distributed), not required
    _torch = None

_HERE = _os.path.dirname(_os.path.abspath(__file__))

try:
    from . import c_rigid  # noqa: F401  (in-tree extension, built by rigid_body_light_amd/build.py)
except ImportError as _e:  # fail loudly: the product has no Python/CPU path
    raise ImportError(
        "rigid_body_light_amd: native extension c_rigid/librbl.so not built "
        "(run `python rigid_body_light_amd/build.py`): %s" % (_e,)
    ) from _e

from .rigid_body import RigidBody  # noqa: E402,F401
from .synth import load_structure, make_config, STRUCT_DIR  # noqa: E402,F401

__all__ = ["RigidBody", "c_rigid", "load_structure", "make_config", "STRUCT_DIR"]
