"""CPU oracle for the MF-ViT hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain fp32/fp64 PyTorch-CPU restatement of the reference algorithm
(endiqq/Multi-Feature-ViT) for the one hot path this repository accelerates.  It exists so
that the HIP kernels can be checked against something that follows the reference line by line.

Rules (enforced by tests/test_boundary_cpu.py::test_product_never_imports_oracle):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
    may import it, and there only as the checker / reported CPU baseline;
  * nothing under ``multi-feature-vit_amd/`` imports it, and the product path raises when the
    HIP library is missing instead of falling back to this code.

Pinning status
  * in-tree reference code (CrossAttention, PreNorm, MultiScaleTransformerEncoder, Fus_CrossViT,
    MoCo builder pieces, LARS, LR / momentum schedules): PINNED against outputs of the reference
    itself, generated in the build container by ``oracle/make_golden.py`` and committed under
    ``tests/golden/`` (the reference ships no tests or golden vectors of its own).
  * ViT-S/16 backbone (``vits.py`` / ``vits_returnftrs.py`` are ABSENT from the reference tree; they
    are facebookresearch/moco-v3 ``vits.py`` on top of timm ``VisionTransformer``, unpinned by the
    reference): PARITY UNPINNED.  The restatement follows the public moco-v3 / timm-0.4.9
    definition and is cross-checked structurally against an independently
    written ``transformers.ViTModel`` of the same configuration built from a local config object.
"""
