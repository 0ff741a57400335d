"""Drop-in for ``vits_returnftrs`` (imported as ``vits`` at MAIN_CA:44; absent from the reference tree): the same
constructors as ``vits`` whose modules additionally expose ``features3D(img) -> (B, 197, 384)`` (FUS:80,83,128,133).
Here every backbone has ``features3D``, so this module re-exports ``vits``.
"""
from vits import *  # noqa: F401,F403
from vits import __all__  # noqa: F401
