"""Stand-alone entry points of the exchange modules (PreNorm(CrossAttention)(x), MultiScaleTransformerEncoder(xs, xl)).

The live path of the reference reaches these only through Fus_CrossViT, which runs them fused (mfvit.fusion).  The
stand-alone module calls are served by the same kernels; until that wiring lands they fail loudly rather than fall
back to eager PyTorch.
"""
from . import _lib


def prenorm_cross_attention(prenorm, x):
    raise _lib.MfvitError("stand-alone PreNorm(CrossAttention)(x) is not wired to the HIP kernels yet; use Fus_CrossViT "
                          "(the only live caller in the reference, FUS:25,30)")


def exchange(encoder, xs, xl):
    raise _lib.MfvitError("stand-alone MultiScaleTransformerEncoder(xs, xl) is not wired to the HIP kernels yet; use "
                          "Fus_CrossViT (the only live caller in the reference, FUS:88-99,137-138)")
