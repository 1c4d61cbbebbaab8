"""Stage-3 distillation objective (normL2) as three launches: sums of squares, (optional all-reduce of 4
scalars under ray-DP), finalise; one launch backward.  Reference formulation: Trainer.get_loss and the
loss assembly in train_step, distill_mutual/utils.py:941-952, 1109-1189 -- kept in trainer.py as the generic path
(other loss types, CPU oracle)."""
import torch
from torch.autograd import Function

import pvd_hip


class ObjectiveRide:
    """The stage-3 objective riding on the student's compositing launches (pvd_composite_objective_*, csrc/raymarching.hip): the
    trainer creates one per step from the TEACHER's outputs, the renderer adds the student's feature / colour rows and hands it
    to composite_rays_train_bg, whose forward launch also leaves the partial sums of the four squared norms (`S`, `nparts`);
    _DistillNormL2 then skips its own sums launch, and in the backward pass leaves coefficients and gradient buffers here for
    the compositing backward launch instead of launching k_sumsq4_bwd (`armed`).  Two launches fewer on the step's chain."""

    def __init__(self, img_t, fea_t, col_t, rates_decay=None, fea_decay=1.0, fixed_parts=False):
        """rates_decay (the device-side rates [4]) + fea_decay: the objective is FINISHED by the backward launch as well (no
        k_loss_final between the passes; loss / norms are filled in by the backward pass): the feature rate's per-step decay is
        then applied by the forward launch.
        fixed_parts (ray-DP): the forward launch leaves a number of partial sums that depends on the ray count alone, the same on every
        rank; _DistillNormL2 all-reduces the PARTIALS over the ranks and the backward launch finishes the objective from the summed
        partials like from its own -- one small collective between the two compositing launches and nothing else."""
        self.fixed_parts = bool(fixed_parts)
        f = lambda t: t.detach().float().contiguous()
        self.img_t, self.fea_t, self.col_t = f(img_t).reshape(-1, 3), f(fea_t), f(col_t)
        self.fea_s = self.col_s = self.S = None
        self.nparts, self.armed = 0, False
        self.coef = self.upstream = self.g_fea = self.g_col = None
        self.rates_decay, self.fea_decay, self.decayed, self.finish = rates_decay, float(fea_decay), False, None

    def with_student(self, fea_s, col_s):
        """The student's rows; False (and the ride is off) when the shapes / dtypes are not what the fused launches read."""
        if (fea_s is None or col_s is None or fea_s.dim() != 2 or fea_s.shape[1] != 16 or fea_s.dtype != torch.float32
                or fea_s.shape != self.fea_t.shape or tuple(col_s.shape) != (fea_s.shape[0], 3) or self.col_t.shape != col_s.shape):
            return False
        self.fea_s, self.col_s = fea_s.detach().contiguous(), col_s.detach().float().contiguous()
        return True


class _DistillNormL2(Function):
    """(img_stu, img_tea [.,N,3], fea_stu, fea_tea [M,16], col_stu, col_tea [M,3], rates[4] device, dp)
    -> (loss scalar, norms[4] detached: rgb, fea, sigma, colour)"""

    @staticmethod
    def forward(ctx, img_s, img_t, fea_s, fea_t, col_s, col_t, rates, dp, fea_decay, extra, defer, ride=None):
        dev = img_s.device
        args = [t.detach().float().contiguous() for t in (img_s, img_t, fea_s, fea_t, col_s, col_t)]
        exchange = dp is not None and dp.enabled
        ride = ride if (ride is not None and ride.S is not None and ride.nparts >= 2 and not defer
                        and (not exchange or (ride.fixed_parts and ride.decayed))) else None
        ctx.ride = ride
        if ride is not None:  # the compositing launch already formed the partial sums (ObjectiveRide)
            loss = torch.empty(1, dtype=torch.float32, device=dev)
            coef = torch.empty(4, dtype=torch.float32, device=dev)
            norms = torch.empty(4, dtype=torch.float32, device=dev)
            if exchange:  # under ray-DP the norms are global: the partial sums of all shards, summed (the same buffer size on every rank)
                assert ride.rates_decay is rates and any(t.requires_grad for t in (img_s, fea_s, col_s))
                dp.all_reduce_sum_(ride.S[4:4 + 4 * ride.nparts])
            if ride.decayed and ride.rates_decay is rates and any(t.requires_grad for t in (img_s, fea_s, col_s)):
                # ... and applied the rate decay: the backward launch finishes the objective (loss / norms / coef filled in THERE)
                ride.finish = (rates, extra, ride.S, loss, norms)
            else:
                assert not ride.decayed, "the forward launch decayed the feature rate: the objective must be finished by the backward launch"
                pvd_hip.distill_loss_final(ride.S, rates, loss, coef, norms, reduce=int(ride.nparts), fea_decay=fea_decay, extra=extra)
            ctx.save_for_backward(*args, coef)
            ctx.deferred, ctx.reduce = False, True
            ctx.shapes = (img_s.shape, fea_s.shape, col_s.shape)
            ctx.mark_non_differentiable(norms)
            ctx.set_materialize_grads(False)
            return loss[0], norms
        S = torch.empty(4 + 4 * 1024, dtype=torch.float32, device=dev)  # 4 sums + per-workgroup partials (scratch)
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
        s_img, s_fea, s_col = ctx.shapes
        if ctx.ride is not None:
            # nothing is launched here: the student's compositing backward (next in autograd's order, same stream) forms the image
            # gradient from the coefficients and fills g_fea / g_col; g_img is a placeholder it ignores
            r = ctx.ride
            r.coef, r.upstream, r.g_fea, r.g_col, r.armed = coef, up, g_fea, g_col, True
            return g_img.view(s_img), None, g_fea.view(s_fea), None, g_col.view(s_col), None, None, None, None, None, None, None
        if ctx.deferred:
            S, rates, loss, norms, extra = ctx.late
            pvd_hip.distill_loss_backward(img_s, img_t, fea_s, fea_t, col_s, col_t, S, rates, up, loss, coef, norms, g_img, g_fea, g_col,
                                          reduce=ctx.reduce, extra=extra)
        else:
            pvd_hip.distill_sumsq_backward(img_s, img_t, fea_s, fea_t, col_s, col_t, coef, up, g_img, g_fea, g_col)
        return g_img.view(s_img), None, g_fea.view(s_fea), None, g_col.view(s_col), None, None, None, None, None, None, None


def distill_loss_normL2(img_s, img_t, fea_s, fea_t, col_s, col_t, rates, dp=None, fea_decay=1.0, extra=None, defer=False, ride=None):
    """fea_decay: multiply rates[1] in place before use (the per-step decay of the feature rate); extra: partial sums of a
    parameter-only term to add to the loss value (no gradient: e.g. the L1 regulariser applied inside the optimizer).
    defer=True: the returned loss / norms tensors are filled in by the BACKWARD pass (one launch fewer per training step);
    only for callers that look at them after loss.backward()."""
    return _DistillNormL2.apply(img_s, img_t, fea_s, fea_t, col_s, col_t, rates, dp, float(fea_decay), extra, bool(defer), ride)


class _FusedMSE(torch.autograd.Function):
    """mean((pred - target)^2) with its gradient formed in the forward's one launch (pvd_mse_forward); the backward is one multiply by
    the upstream gradient.  The teacher's objective (just_train_tea/utils.py:573-581)."""

    @staticmethod
    def forward(ctx, pred, target):
        import pvd_hip
        p, t = pred.float().contiguous(), target.float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        d = torch.empty_like(p)
        pvd_hip.mse_forward(p, t, loss.view(1), d)
        ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None


def mse_fused(pred, target):
    return _FusedMSE.apply(pred, target)
