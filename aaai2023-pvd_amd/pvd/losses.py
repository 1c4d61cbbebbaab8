"""Stage-3 distillation objective (normL2) as three launches: sums of squares, (optional all-reduce of 4
scalars under ray-DP), finalise; one launch backward.  Reference formulation: Trainer.get_loss and the
loss assembly in train_step, distill_mutual/utils.py:941-952, 1109-1189 -- kept in trainer.py as the generic path
(other loss types, CPU oracle)."""
import torch
from torch.autograd import Function

import pvd_hip


class _DistillNormL2(Function):
    """(img_stu, img_tea [.,N,3], fea_stu, fea_tea [M,16], col_stu, col_tea [M,3], rates[4] device, dp)
    -> (loss scalar, norms[4] detached: rgb, fea, sigma, colour)"""

    @staticmethod
    def forward(ctx, img_s, img_t, fea_s, fea_t, col_s, col_t, rates, dp, fea_decay, extra, defer):
        dev = img_s.device
        args = [t.detach().float().contiguous() for t in (img_s, img_t, fea_s, fea_t, col_s, col_t)]
        S = torch.empty(4 + 4 * 1024, dtype=torch.float32, device=dev)  # 4 sums + per-workgroup partials (scratch)
        exchange = dp is not None and dp.enabled
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        coef = torch.empty(4, dtype=torch.float32, device=dev)
        norms = torch.empty(4, dtype=torch.float32, device=dev)
        # defer: the value of the objective is not needed before the backward pass (a training step): finishing it (one
        # single-workgroup launch between the two passes) moves into the backward launch, every workgroup of which redoes the
        # 256-term reduction for itself; loss / norms are then FILLED IN BY THE BACKWARD PASS
        defer = bool(defer) and torch.is_grad_enabled() and any(t.requires_grad for t in (img_s, fea_s, col_s))
        pvd_hip.distill_sumsq(*args, S, reduce=exchange, rates_decay=rates if defer else None, fea_decay=fea_decay)
        if exchange:
            dp.all_reduce_sum_(S[:4])  # global norms: sum of squares over all shards
        if not defer:
            pvd_hip.distill_loss_final(S, rates, loss, coef, norms, n_img=args[0].numel(), M=args[2].shape[0], reduce=not exchange,
                                       fea_decay=fea_decay, extra=extra)
        ctx.save_for_backward(*args, coef)
        ctx.deferred, ctx.reduce = defer, not exchange
        if defer:  # buffers the backward launch reads / fills (not autograd state: the kernels write them behind its back)
            ctx.late = (S, rates, loss, norms, extra)
        ctx.shapes = (img_s.shape, fea_s.shape, col_s.shape)
        ctx.mark_non_differentiable(norms)
        ctx.set_materialize_grads(False)
        return loss[0], norms

    @staticmethod
    def backward(ctx, g_loss, _g_norms):
        img_s, img_t, fea_s, fea_t, col_s, col_t, coef = ctx.saved_tensors
        g_img, g_fea, g_col = torch.empty_like(img_s), torch.empty_like(fea_s), torch.empty_like(col_s)
        up = g_loss.detach().float().reshape(1).contiguous()
        if ctx.deferred:
            S, rates, loss, norms, extra = ctx.late
            pvd_hip.distill_loss_backward(img_s, img_t, fea_s, fea_t, col_s, col_t, S, rates, up, loss, coef, norms, g_img, g_fea, g_col,
                                          reduce=ctx.reduce, extra=extra)
        else:
            pvd_hip.distill_sumsq_backward(img_s, img_t, fea_s, fea_t, col_s, col_t, coef, up, g_img, g_fea, g_col)
        s_img, s_fea, s_col = ctx.shapes
        return g_img.view(s_img), None, g_fea.view(s_fea), None, g_col.view(s_col), None, None, None, None, None, None


def distill_loss_normL2(img_s, img_t, fea_s, fea_t, col_s, col_t, rates, dp=None, fea_decay=1.0, extra=None, defer=False):
    """fea_decay: multiply rates[1] in place before use (the per-step decay of the feature rate); extra: partial sums of a
    parameter-only term to add to the loss value (no gradient: e.g. the L1 regulariser applied inside the optimizer).
    defer=True: the returned loss / norms tensors are filled in by the BACKWARD pass (one launch fewer per training step);
    only for callers that look at them after loss.backward()."""
    return _DistillNormL2.apply(img_s, img_t, fea_s, fea_t, col_s, col_t, rates, dp, float(fea_decay), extra, bool(defer))
