"""mfvit: MI355X-native (gfx950) engine of the Multi-Feature-ViT hot path.

ctypes binding of libmfvit_hip.so (``_lib``), tensor-level op wrappers (``ops``) and the module classes the
drop-in files (``vits``, ``vits_returnftrs``, ``model.*``, ``moco.*``) are built from.
"""
from ._lib import MfvitError, lib  # noqa: F401
