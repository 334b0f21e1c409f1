"""Drop-in alias: `from Rigid import RigidBody` / `from Rigid import c_rigid`
resolve to the MI355X-native implementation (reference package layout:
src/__init__.py:1, CMakeLists.txt:24-27)."""
from rigid_body_light_amd import RigidBody, c_rigid  # noqa: F401
