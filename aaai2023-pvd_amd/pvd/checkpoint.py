"""Checkpoints in the reference's file format and with the reference's loading rules
(distill_mutual/utils.py:1405-1559): a ``torch.save``d dict

    {"epoch", "global_step", "stats", ["resolution" (VM)], ["mean_count", "mean_density" (cuda_ray)], "model": state_dict}

whose ``model`` entry is ``NeRFNetwork.state_dict()`` -- same keys, logical shapes and dtypes as the reference's
(tests/test_golden.py::test_network_state_dict_layout).  The VM factors and the Plenoxel volume of this code base are
stored channels-last; a state-dict only fixes the LOGICAL layout, ``load_state_dict`` copies element by element, so files
move both ways: what is written here is made contiguous in the reference's (channel-major) order first, and a
reference file loads into the channels-last parameters unchanged.

Loading rules kept from the reference:
  * teacher: ``load_state_dict(strict=False)`` + ``mean_count`` / ``mean_density`` (utils.py:1477-1494);
  * student: from ``ckpt_student`` if given, else FROM THE TEACHER's file (utils.py:1529-1537) -- every same-named tensor
    carries over: ``density_grid``, ``density_bitfield``, ``step_counter``, ``aabb_*``, ``color_net.*`` and, like -> like,
    the encoder and ``sigma_net``; a VM student is first resampled to the file's ``resolution`` (:1539-1540).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _reference_layout(t):
    """A tensor as the reference would hold it: plain contiguous (its conv-style tables are channel-major)."""
    return t.detach().contiguous() if t.dim() >= 4 else t.detach().clone()


def checkpoint_dict(model, epoch=0, global_step=0, stats=None, extra=None):
    """The dict ``Trainer.save_checkpoint`` writes (utils.py:1405-1447, `full` is forced off there)."""
    state = {"epoch": int(epoch), "global_step": int(global_step),
             "stats": stats if stats is not None else {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}}
    if model.model_type == "vm":
        state["resolution"] = list(model.resolution)
    if model.cuda_ray:
        state["mean_count"] = int(model.mean_count)
        state["mean_density"] = float(model.mean_density)
    if extra:
        state.update(extra)
    state["model"] = {k: _reference_layout(v) for k, v in model.state_dict().items()}
    return state


def save_checkpoint(path, model, **kw):
    torch.save(checkpoint_dict(model, **kw), path)
    return path


@torch.no_grad()
def upsample_vm(model, resolution):
    """VM factors resampled to `resolution` (bilinear, align_corners) -- reference: upsample_params / upsample_model,
    network.py:559-587.  Keeps the channels-last storage."""
    from .network import _channels_last
    getattr(model, "_pvd_flush_params", lambda: None)()  # (a flat optimizer's deferred decays, before whole tables are read)
    res = [int(r) for r in resolution]
    for mats, vecs in ((model.sigma_mat, model.sigma_vec), (model.color_mat, model.color_vec)):
        for i in range(3):
            m0, m1 = model.mat_ids[i]
            m = F.interpolate(mats[i].data, size=(res[m1], res[m0]), mode="bilinear", align_corners=True)
            v = F.interpolate(vecs[i].data, size=(res[model.vec_ids[i]], 1), mode="bilinear", align_corners=True)
            mats[i] = nn.Parameter(_channels_last(m))
            vecs[i] = nn.Parameter(_channels_last(v))
    model.resolution = res
    if hasattr(model, "_aabb_host"):
        del model._aabb_host


def _load_model(model, ckpt):
    """load_state_dict(strict=False), as both loaders do; returns (missing, unexpected)."""
    # a flat optimizer's deferred decays are applied BEFORE the parameters are overwritten: a later flush would otherwise
    # decay the freshly loaded rows by steps they never took
    getattr(model, "_pvd_flush_params", lambda: None)()
    missing, unexpected = model.load_state_dict(ckpt["model"], strict=False)
    import sys
    hip = sys.modules.get("pvd_hip")  # derived caches (packed head weights, f16 table shadows) key on this
    if hip is not None:
        hip.note_weights_changed(list(model.parameters()))
    if model.cuda_ray:
        if "mean_count" in ckpt:
            model.mean_count = ckpt["mean_count"]
        if "mean_density" in ckpt:
            model.mean_density = ckpt["mean_density"]
    return list(missing), list(unexpected)


def load_teacher_checkpoint(model_tea, path, map_location=None):
    """reference: load_teacher_checkpoint, utils.py:1477-1494."""
    ckpt = torch.load(path, map_location=map_location or next(model_tea.parameters()).device, weights_only=False)
    return _load_model(model_tea, ckpt)


def load_student_checkpoint(model_stu, ckpt_teacher, ckpt_student=None, map_location=None):
    """reference: load_student_checkpoint, utils.py:1529-1556: the student's own file if there is one, otherwise the teacher's
    file (so the student marches on the teacher's occupancy grid and starts from its colour head)."""
    path = ckpt_student if ckpt_student else ckpt_teacher
    ckpt = torch.load(path, map_location=map_location or next(model_stu.parameters()).device, weights_only=False)
    if model_stu.model_type == "vm" and "resolution" in ckpt:
        upsample_vm(model_stu, ckpt["resolution"])
    return _load_model(model_stu, ckpt)
