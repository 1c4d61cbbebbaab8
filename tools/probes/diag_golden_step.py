"""Diagnostic: which tensor of the fp32 HIP distillation step departs from the CPU oracle-ops step (reference_step fixture inputs)."""
import os, sys, types
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from test_golden_step import G, config, load
from oracle_ops import oracle_ops
from pvd.ops import hip_ops
from pvd.trainer import DistillTrainer
from pvd.workload import make_model

def run(case, stage, device, ops, order_note=""):
    opt = config(case)
    dev = torch.device(device)
    torch.manual_seed(0)
    tea = make_model(ops, opt, opt.teacher_type, True, dev); stu = make_model(ops, opt, opt.model_type, False, dev)
    load(tea, case, "tea"); load(stu, case, "stu")
    tr = DistillTrainer(opt, tea, stu, dev, fp16=False)
    pre = "%s__s%d__" % (case, stage)
    tr.global_step = tr.opt.global_step = int(G[pre + "global_step"]); tr.loss_rate_fea_sc = float(G[pre + "fea_rate_before"])
    ro, rd = torch.from_numpy(G["rays_o"]).to(dev), torch.from_numpy(G["rays_d"]).to(dev)
    stu.train(); tea.train(); tr._zero_grads()
    torch.manual_seed(int(G[pre + "seed"])); bg = torch.rand([1, ro.shape[1], 3]).to(dev)
    loss, info, ps, pt = tr.compute_loss(ro, rd, bg)
    f = lambda t: None if t is None else t.detach().float().cpu()
    return dict(loss=float(loss.detach()), stu_color=f(stu.color_l), tea_color=f(tea.color_l), stu_sigma=f(stu.sigma_l), tea_sigma=f(tea.sigma_l),
                stu_fea=f(stu.feature_sigma_color), tea_fea=f(tea.feature_sigma_color), ps=f(ps), pt=f(pt))

hip = types.SimpleNamespace(**{**vars(hip_ops()), "flat_adamw": None})
for case, stage in (("hash_vm", 2), ("hash_hash", 3), ("hash_vm_teafirst", 3)):
    a = run(case, stage, "cpu", oracle_ops()); b = run(case, stage, "cuda:0", hip)
    print("== %s stage %d: loss cpu %.6f hip %.6f ref %.6f" % (case, stage, a["loss"], b["loss"], float(G["%s__s%d__loss" % (case, stage)])))
    for k in a:
        if k == "loss" or a[k] is None or b[k] is None: continue
        d = (a[k] - b[k]).abs()
        n = 1607
        rows = d.reshape(d.shape[0], -1).max(1).values if d.dim() > 1 else d
        print("   %-10s max|d| %.3e (used rows %.3e, padding rows %.3e)  max|ref| %.3e  shape %s" % (
            k, d.max().item(), rows[:n].max().item() if rows.shape[0] > n else float('nan'), rows[n:].max().item() if rows.shape[0] > n else float('nan'),
            a[k].abs().max().item(), tuple(a[k].shape)))
