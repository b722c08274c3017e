"""vc3d: MI355X (gfx950) kernels and executor for the 3D-ResNet hot path.

Importing this package loads libvc3d.so; it raises if the library is missing
(there is no CPU or eager-PyTorch fallback for the compute path).
"""
from . import capi  # noqa: F401  (loads the shared library, fails loudly)
from . import ops  # noqa: F401
from .engine import R3DEngine  # noqa: F401
from .autograd import backbone_apply, linear_apply  # noqa: F401
