"""Drop-in for the ``_noprediction_q`` builder variant (identical to the main builder except that the key branch skips
the predictor: its line 175).  MAIN_MOCO:34 keeps this import commented out."""
from moco.builder_vit_mocov3structure_mocov2loss import MoCo as _MoCo
from moco.builder_vit_mocov3structure_mocov2loss import MoCo_ViT as _MoCo_ViT
from moco.builder_vit_mocov3structure_mocov2loss import concat_all_gather  # noqa: F401


class MoCo(_MoCo):
    def __init__(self, *a, **k):
        k.setdefault("predict_keys", False)
        super().__init__(*a, **k)


class MoCo_ViT(_MoCo_ViT):
    def __init__(self, *a, **k):
        k.setdefault("predict_keys", False)
        super().__init__(*a, **k)
