"""Drop-in for the reference's ``moco/optimizer.py`` (moco_pretraining/moco/moco/optimizer.py:10-43): ``LARS`` with the same
constructor ``(params, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001)``; ``step()`` is two kernel launches
over all tensors (norms, then update) instead of a Python loop with two ``torch.norm`` calls per tensor."""
from mfvit.optim import LARS  # noqa: F401
