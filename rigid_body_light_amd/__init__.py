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

_HERE = _os.path.dirname(_os.path.abspath(__file__))

try:
    from . import c_rigid  # noqa: F401  (in-tree extension, built by rigid_body_light_amd/build.py)
except ImportError as _e:  # fail loudly: the product has no Python/CPU path
    raise ImportError(
        "rigid_body_light_amd: native extension c_rigid/librbl.so not built "
        "(run `python -m rigid_body_light_amd.build`): %s" % (_e,)
    ) from _e

from .Rigid import RigidBody  # noqa: E402,F401
from .synth import load_structure, make_config, STRUCT_DIR  # noqa: E402,F401

__all__ = ["RigidBody", "c_rigid", "load_structure", "make_config", "STRUCT_DIR"]
