"""AdamW on flat buffers: every parameter (and its gradient and both moments) is a strided view into one fp32
buffer per kind, so the update is one streaming HIP kernel (pvd_adamw_step) instead of a multi-tensor launch
per parameter group, the gradient exchange of ray-DP is one all-reduce, and zeroing the gradients one memset.
Drop-in for torch.optim.AdamW(betas, eps, weight_decay) as the reference constructs it
(main_distill_mutual.py:334-339), including GradScaler's unscale / skip-on-inf protocol and tensor learning
rates (LR schedulers fill them in place), so a captured HIP graph sees the schedule."""
import torch

import pvd_hip


def _view(flat, p, off):
    return torch.as_strided(flat, p.size(), p.stride(), storage_offset=off)


class FlatAdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True  # GradScaler hands us grad_scale / found_inf instead of syncing

    def __init__(self, param_groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        groups = []
        for g in param_groups:
            ps = [p for p in g["params"] if p.requires_grad]
            if ps:
                groups.append({"params": ps, "lr": float(g["lr"])})
        dev = groups[0]["params"][0].device
        super().__init__(groups, dict(betas=betas, eps=eps, weight_decay=weight_decay))
        # layout: group after group, each parameter padded to a multiple of 4 elements
        offs, ends, off = [], [], 0
        for g in self.param_groups:
            for p in g["params"]:
                assert p.dtype == torch.float32 and p.is_cuda and p._is_non_overlapping_and_dense() if hasattr(p, "_is_non_overlapping_and_dense") else True
                offs.append(off)
                off += (p.numel() + 3) // 4 * 4
            ends.append(off)
        self.n, self.segment_ends = off, ends
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr_dev = torch.tensor([g["lr"] for g in self.param_groups], dtype=torch.float32, device=dev)
        self.params, self.offsets = [p for g in self.param_groups for p in g["params"]], offs
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                v = _view(self.flat_p, p, o)
                v.copy_(p.data)
                p.data = v  # same shape / strides / values, storage inside the flat buffer
                p.grad = _view(self.flat_g, p, o)
        for k, g in enumerate(self.param_groups):
            g["lr"] = self.lr_dev[k:k + 1]  # LRScheduler.step() fills tensor lrs in place

    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        self.reattach()

    def reattach(self):
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = _view(self.flat_g, p, o)

    @torch.no_grad()
    def step(self, closure=None):
        d = self.defaults
        pvd_hip.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.segment_ends, self.lr_dev, d["betas"][0], d["betas"][1],
                           d["eps"], d["weight_decay"], self.step_count, getattr(self, "grad_scale", None), getattr(self, "found_inf", None))
        # (GradScaler sets grad_scale / found_inf right before step() and deletes them afterwards)
