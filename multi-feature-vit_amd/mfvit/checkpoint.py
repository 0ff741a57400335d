"""Checkpoint / state-dict compatibility with the reference's drivers.

The modules of this package keep the reference's parameter names, so the reference's own code works on them unchanged; these
helpers restate the dictionary layouts and key surgery so that a caller (or a test) does not need the drivers:

  MAIN_MOCO = main_covid_mocov3based_..._vitsmall.py, MAIN_SS = main_vit_covid_..._v3structure_vitsmall.py,
  MAIN_CA   = main_vit_covid_..._crossvit_2vits_2additionaloutputs_trainval_sum.py
"""
import os

import torch


def pretrain_checkpoint(model, optimizer, epoch, arch, scaler=None):
    """The dict MoCo pretraining saves (MAIN_MOCO:461-467 / 473-479); `scaler` is added when resuming needs it (MAIN_MOCO:368)."""
    state = {"epoch": epoch + 1, "arch": arch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}
    if scaler is not None:
        state["scaler"] = scaler.state_dict()
    return state


def finetune_checkpoint(model, optimizer, epoch, arch, best_metric_val, best_metric_val_test=None, best_metric_test=None):
    """The dict the finetune drivers save (MAIN_SS:567-575; MAIN_CA:712-720 leaves the two test metrics out)."""
    state = {"epoch": epoch + 1, "arch": arch, "state_dict": model.state_dict()}
    if best_metric_val_test is not None:
        state["best_metric_val_test"] = best_metric_val_test
    state["best_metric_val"] = best_metric_val
    if best_metric_test is not None:
        state["best_metric_test"] = best_metric_test
    state["optimizer"] = optimizer.state_dict()
    return state


def save_checkpoint(checkpoint_folder, state, is_best, filename="last_checkpoint.pth.tar"):
    """MAIN_CA:1002-1011: the best model goes to model_best.pth.tar, anything else to `filename`."""
    if is_best:
        filename = "model_best.pth.tar"
    path = os.path.join(checkpoint_folder, filename)
    torch.save(state, path)
    return path


def strip_moco_prefix(state_dict, linear_keyword="head"):
    """In-place key surgery of MAIN_SS:326-333: keep `module.base_encoder.*` except the projector (`.head`), drop the prefix,
    delete everything else (momentum encoder, predictor, queue).  The reference always saves from a DistributedDataParallel wrapper
    (keys start with 'module.'); a model trained with this package's GradSync is not wrapped, so the bare `base_encoder.*` layout
    is accepted as well (a leading 'module.' is optional)."""
    wrapped = any(k.startswith("module.") for k in state_dict)
    prefix = "module.base_encoder." if wrapped else "base_encoder."
    for k in list(state_dict.keys()):
        if k.startswith(prefix) and not k.startswith(prefix + linear_keyword):
            state_dict[k[len(prefix):]] = state_dict[k]
        del state_dict[k]
    return state_dict


def load_pretrained_backbone(model, checkpoint, linear_keyword="head"):
    """MAIN_SS:323-337: load a MoCo pretraining checkpoint (path or dict) into a finetune backbone; everything but the
    classifier must be found.  Returns the load_state_dict result."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location="cpu")
    state_dict = strip_moco_prefix(dict(checkpoint["state_dict"]), linear_keyword)
    msg = model.load_state_dict(state_dict, strict=False)
    assert set(msg.missing_keys) == {"%s.weight" % linear_keyword, "%s.bias" % linear_keyword}, msg.missing_keys
    return msg


def sanity_check(state_dict, pretrained, semi_supervised=False, linear_keyword="head", prefix=""):
    """MAIN_SS / MAIN_CA:1013-1040: after linear probing nothing but the classifier may differ from the pretrained weights.
    `prefix` is what the finetune state dict puts in front of the backbone keys ('' for a bare backbone, 'module.' under DDP)."""
    if semi_supervised:
        return True
    if isinstance(pretrained, (str, os.PathLike)):
        pretrained = torch.load(pretrained, map_location="cpu")
    pre = pretrained["state_dict"]
    pre_prefix = "module.base_encoder." if any(k.startswith("module.") for k in pre) else "base_encoder."
    for k in list(state_dict.keys()):
        if "%s.weight" % linear_keyword in k or "%s.bias" % linear_keyword in k:
            continue
        k_pre = pre_prefix + k[len(prefix):]
        assert (state_dict[k].cpu() == pre[k_pre].cpu()).all(), "{} is changed in linear classifier training.".format(k)
    return True
