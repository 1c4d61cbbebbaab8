"""Ray data parallelism (new work, the reference has none -- tools/details.md:24): every rank renders its own rays against
replicated models; one flat-bucket all-reduce (SUM) of the student gradient per step over RCCL/xGMI; norm-type losses are made
global by all-reducing the sum of squares first.  Also the flat gradient bucket the exchange moves."""
import os

import torch
import torch.distributed as dist


class RayDP:
    """Ray-level data parallel context.  world_size == 1 -> every collective is a no-op."""

    def __init__(self, group=None):
        # PVD_DP_FORCE=1 keeps every collective live in a world of one rank (test mode: the communication library's
        # streams / watchdog next to graph capture, on a single-GPU box)
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("PVD_DP_FORCE") == "1")
        self.group = group
        self.world_size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.capture = None  # a SegmentedCapture while the trainer records a step
        # RCCL collectives can be recorded INTO the step's HIP graph (probed on MI355X / ROCm 7: tools/probe_rccl_capture.py):
        # the whole step is then ONE graph launch instead of three graphs with two eager collectives between them (~60 us of
        # fixed overhead per step).  gloo cannot be captured; PVD_DP_INGRAPH=0 keeps the segmented form; a capture that
        # fails falls back to it (DistillTrainer.capture_step).
        self.ingraph = (self.enabled and dist.get_backend(group) == "nccl" and os.environ.get("PVD_DP_INGRAPH", "1") != "0")


    def _issue(self, run, replay=None):
        """A collective at this point of the step: recorded as a node of the graph being captured (RCCL, `ingraph`), or run eagerly
        between two graphs of a segmented capture, or simply run."""
        if self.capture is not None and self.capture.active and self.ingraph:
            run()
        elif self.capture is not None and self.capture.active:
            self.capture.break_for(run, replay)  # collectives stay out of the graphs: eager, between two replays
        else:
            run()

    def _staged(self, *tensors):
        """gloo (the test backend: several ranks sharing one GPU) moves host memory: stage device tensors through the host."""
        return self.enabled and dist.get_backend(self.group) == "gloo" and any(t.is_cuda for t in tensors)

    def reduce_scatter_sum_(self, out, inp):
        """out = this rank's chunk of SUM over ranks of inp (inp.numel() == world_size * out.numel(); `out` may be the rank's own
        chunk of `inp`, the in-place form).  Every element is summed by exactly one rank."""
        assert inp.is_contiguous() and out.is_contiguous() and inp.numel() == self.world_size * out.numel()
        if self._staged(out, inp):
            def run():
                h_in, h_out = inp.detach().cpu(), torch.empty(out.shape, dtype=out.dtype)
                dist.reduce_scatter_tensor(h_out, h_in, op=dist.ReduceOp.SUM, group=self.group)
                out.copy_(h_out)
        else:
            def run():
                dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)
        self._issue(run)
        return out

    def all_gather_(self, out, inp):
        """out = concatenation over ranks of inp (`inp` may be the rank's own chunk of `out`)."""
        assert inp.is_contiguous() and out.is_contiguous() and out.numel() == self.world_size * inp.numel()
        if self._staged(out, inp):
            def run():
                h_in, h_out = inp.detach().cpu(), torch.empty(out.shape, dtype=out.dtype)
                dist.all_gather_into_tensor(h_out, h_in, group=self.group)
                out.copy_(h_out)
        else:
            def run():
                dist.all_gather_into_tensor(out, inp, group=self.group)
        self._issue(run)
        return out

    def sharded_sum_(self, t):
        """SUM over ranks as the standard pair reduce-scatter + all-gather (PVD_DP_EXCHANGE=sharded on a path without the flat
        optimizer's sharded update -- CPU / gloo tests, the generic optimizers): the bytes of the all-reduce, every element summed by
        exactly ONE rank, so the replicas receive identical bits by construction."""
        n = self.world_size
        assert t.is_contiguous()
        flat = t.reshape(-1)
        chunk = (flat.numel() + n - 1) // n
        buf = flat
        if chunk * n != flat.numel():
            buf = torch.zeros(chunk * n, dtype=flat.dtype, device=flat.device)
            buf[:flat.numel()].copy_(flat)
        mine = buf[self.rank * chunk:(self.rank + 1) * chunk]
        red = torch.empty_like(mine)
        self.reduce_scatter_sum_(red, buf)
        out = torch.empty_like(buf)
        self.all_gather_(out, red)
        flat.copy_(out[:flat.numel()])
        return t

    def all_reduce_sum_(self, t, overlap=None):
        """overlap: a callable launching device work that does not depend on the result (e.g. replaying the graph of the
        next step's parameter-independent prefix); in a captured step it is issued while the collective is in flight."""
        if self.enabled:
            if (os.environ.get("PVD_DP_EXCHANGE", "allreduce") == "sharded" and self.world_size > 1 and overlap is None
                    and t.numel() >= int(os.environ.get("PVD_DP_SHARDED_MIN", "65536"))):  # (scalars and short buffers: one latency-bound all-reduce)
                return self.sharded_sum_(t)
            run = lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)  # noqa: E731
            replay = None
            if overlap is not None and self.capture is not None and self.capture.active and not self.ingraph:
                def replay():
                    work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    overlap()
                    work.wait()
            self._issue(run, replay)
        return t

    def broadcast_(self, t, src=0):
        """t = rank `src`'s t on every rank (never part of a recorded step: occupancy upkeep and evaluation run between the graphs)."""
        if not self.enabled or self.world_size == 1:
            return t
        assert t.is_contiguous()
        if self._staged(t):
            h = t.detach().cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=self.group)
        return t

    def sync_occupancy(self, model, src=0):
        """SURVEY 8(e), occupancy state: the replicas march on ONE occupancy grid.  `update_extra_state` draws random cells and
        jitter (renderer.py:691-693,708-722 of the reference), so after an update every rank takes rank `src`'s density grid,
        bitfield, running mean and sweep count (8.4 MB + 256 KiB every 16 steps); `mean_count` -- the size of a rank's OWN sample
        buffer -- stays per rank.  The device-side update needs it as much as the torch one: its list of occupied cells is
        compacted with atomics, so the list's order -- and with it which occupied cells a draw lands on -- differs from run to
        run even on bit-identical replicas, and a cell drawn twice keeps whichever density was written last, as with the
        reference's `tmp_grid[cas, indices] = sigmas` on a GPU (tests/test_hip_dp_eval_occupancy.py).  Returns True if this rank's grid was
        already rank src's (the occupancy epoch, which invalidates recorded steps and the exchange's row set, moves only if not)."""
        if not self.enabled or not getattr(model, "cuda_ray", False):
            return True
        grid, bits = model.density_grid, model.density_bitfield
        before = (grid.clone(), bits.clone())
        self.broadcast_(grid, src)
        self.broadcast_(bits, src)
        md = model.mean_density
        scal = torch.tensor([float(md), float(model.iter_density)], dtype=torch.float64, device=grid.device)
        self.broadcast_(scal, src)
        if self.rank != src:
            if torch.is_tensor(md):
                md.fill_(float(scal[0]))
            else:
                model.mean_density = float(scal[0])
            model.iter_density = int(scal[1])
        same = torch.equal(before[0], grid) and torch.equal(before[1], bits)
        if not same:
            model.note_occupancy_changed()
        return same

    def render_sharded(self, model, rays_o, rays_d, **render_kw):
        """SURVEY 8(e), evaluation: the N rays of one image [1, N, 3] (row-major pixels, so a slice is a band of image rows) are cut
        into world_size contiguous slices of ceil(N / G) rays -- the last one padded with copies of the image's last ray, so that
        every rank renders and gathers the same shape --, rank r renders slice r through `model.render(staged=True)` and ONE
        all-gather of [ceil(N / G), 4] (rgb, depth) leaves the whole image on every rank.  The reference gathers whole per-rank
        predictions the same way (distill_mutual/utils.py:1243-1258: all_gather of preds / preds_depth).  Rays are independent in
        the inference path, so the result has the bits of a one-rank render (tests/test_dist_gloo.py, tests/test_hip_dp_exchange.py).
        A per-ray `bg_color` [1, N, 3] is sliced with the rays."""
        if not self.enabled or self.world_size == 1:
            out = model.render(rays_o, rays_d, staged=True, **render_kw)
            return {"image": out["image"], "depth": out["depth"]}
        assert rays_o.dim() == 3 and rays_o.shape[0] == 1 and rays_o.shape == rays_d.shape, "one image per call: [1, N, 3]"
        N, G = rays_o.shape[1], self.world_size
        per = (N + G - 1) // G
        lo = min(self.rank * per, N)
        hi = min(lo + per, N)

        def piece(t):  # this rank's band, padded to `per` rays with the image's last ray
            part = t[:, lo:hi]
            if hi - lo < per:
                part = torch.cat([part, t[:, N - 1:N].expand(-1, per - (hi - lo), -1)], dim=1)
            return part.contiguous()
        kw = dict(render_kw)
        bg = kw.get("bg_color")
        if torch.is_tensor(bg) and bg.dim() == 3 and bg.shape[1] == N:
            kw["bg_color"] = piece(bg)
        out = model.render(piece(rays_o), piece(rays_d), staged=True, **kw)
        mine = torch.cat([out["image"].reshape(per, 3).float(), out["depth"].reshape(per, 1).float()], dim=1).contiguous()
        everyone = torch.empty(G * per, 4, dtype=torch.float32, device=mine.device)
        self.all_gather_(everyone, mine)
        return {"image": everyone[:N, :3].reshape(1, N, 3), "depth": everyone[:N, 3].reshape(1, N)}

    def global_sum(self, local):
        """Value = sum over ranks, gradient = gradient of the local term (d total / d local = 1)."""
        if not self.enabled:
            return local
        tot = self.all_reduce_sum_(local.detach().clone())
        return tot + (local - local.detach())

    def global_norm_l2(self, diff):
        """|| concat_r diff_r ||_2 with the right gradient on every shard (torch.norm over the whole
        batch is not a sum of per-shard norms)."""
        if not self.enabled:
            return torch.norm(diff.float())  # zero-safe subgradient, as the reference's torch.norm (utils.py:947)
        s_local = (diff.float() ** 2).sum()
        s_tot = self.all_reduce_sum_(s_local.detach().clone())
        n = torch.sqrt(s_tot)
        return n + (s_local - s_local.detach()) / (2 * n.clamp_min(1e-20))

    def global_norm_l1(self, diff):
        return self.global_sum(diff.float().abs().sum())

    def global_mean(self, x):
        """mean over the global batch (equal shard sizes are not assumed)."""
        if not self.enabled:
            return x.float().mean()
        cnt = self.all_reduce_sum_(torch.tensor(float(x.numel()), device=x.device))
        return self.global_sum(x.float().sum()) / cnt


class FlatGrads:
    """All trainable parameters' gradients as views into ONE flat fp32 buffer, so the step's
    gradient exchange is a single all-reduce with no packing copies (autograd accumulates in place
    into pre-set .grad views)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            assert p.dtype == torch.float32
            p.grad = self._view(p, off)
            off += p.numel()

    def _view(self, p, off):
        # same strides as the parameter (VM factors are channels-last): fused AdamW requires params and
        # grads to share one memory layout, and autograd then accumulates without a re-layout
        return torch.as_strided(self.flat, p.size(), p.stride(), storage_offset=off)

    def zero_(self, full=False):
        self.flat.zero_()
        # re-attach: optimizers / zero_grad(set_to_none) may have dropped the views
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat[off:off + 1].data_ptr():
                p.grad = self._view(p, off)
            off += p.numel()


class _FlatOptGrads:
    """FlatGrads interface over FlatAdamW's own gradient buffer."""

    def __init__(self, opt):
        self.opt, self.flat, self.params = opt, opt.flat_g, opt.params

    def zero_(self, full=False):
        self.opt.zero_grad(full=full)


def _make_loss(kind, dp):
    """reference: Trainer.get_loss, utils.py:941-952 (normL2 is a Frobenius norm, NOT a mean)."""
    if kind == "L2":
        return lambda pred, gt: dp.global_mean((gt.float() - pred.float()) ** 2)
    if kind == "normL2":
        return lambda pred, gt: dp.global_norm_l2(pred - gt)
    if kind == "normL1":
        return lambda pred, gt: dp.global_norm_l1(pred - gt)
    raise ValueError("error loss_type")
