"""Training steps the metric is defined on: the distillation step (student vs frozen teacher on the
SAME samples) and the teacher step (L2 against ground-truth pixels).

Counterparts of ``Trainer.train_step`` / ``train_one_epoch`` in the reference
(distill_mutual/utils.py:753-934, 954-1189; just_train_tea/utils.py:540-640, 760-850) reduced to
what `train rays/s` and PSNR need: loss formulas, stage gating, AMP + GradScaler, AdamW, cosine
(student) / exponential (teacher) LR.  Data loading, logging, checkpoints, SSIM/LPIPS are out of scope.

Ray data parallelism (new work, the reference has none -- tools/details.md:24): every rank renders
its own rays against replicated models; one flat-bucket all-reduce (SUM) of the student gradient per
step over RCCL/xGMI; norm-type losses are made global by all-reducing the sum of squares first.
"""
import os

import torch
import torch.distributed as dist

from .capture import CarriedPrefix, SegmentedCapture
from .ray_dp import FlatGrads, RayDP, _FlatOptGrads, _make_loss


def pvd_forked_graphs_ok():
    """May a step be recorded as a hipGraph with two PARALLEL chains (next step's march / teacher forward forked next to
    this step's scatter + update)?  Only when the HIP runtime spreads its streams over the number of hardware queues the
    schedule was validated with (pvd_hip settles GPU_MAX_HW_QUEUES before the runtime starts);
    otherwise the steps are recorded back to back."""
    import pvd_hip
    return pvd_hip.forked_graphs_ok()


def psnr(pred, truth):
    """-10 log10(mean squared error) (reference: PSNRMeter.update, utils.py:500-507)."""
    mse = torch.mean((pred.float() - truth.float()) ** 2)
    return -10.0 * torch.log10(mse)


class _TrainerBase:
    def __init__(self, opt, model, lr, device, fp16=True, dp=None, eta_min=None, exp_decay=False):
        self.opt = opt
        self.device = torch.device(device)
        self.device_type = self.device.type
        self.fp16 = bool(fp16)
        self.dp = dp or RayDP()
        self.model = model
        params = model.get_params(lr)
        fused = self.device_type == "cuda"
        # reference: AdamW(betas=(0.9, 0.99), eps=1e-15), default weight decay (main_distill_mutual.py:334-339)
        self.flat_opt = fused and getattr(model.ops, "flat_adamw", None) is not None
        if self.flat_opt:  # parameters, gradients and moments in flat buffers, one HIP kernel per step
            self.optimizer = model.ops.flat_adamw(params, betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-2)
        else:
            if fused:  # device-side lr so that a captured step sees the schedule (LRScheduler fills tensor lrs in place)
                for g in params:
                    g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=self.device)
            self.optimizer = torch.optim.AdamW(params, betas=(0.9, 0.99), eps=1e-15, fused=fused, capturable=fused)
        amp = self.fp16 and self.device_type == "cuda"
        if self.flat_opt:
            # schedule evaluated inside the update kernel, inf check = one read-only pass over the flat gradient:
            # nothing of the scheduler / scaler bookkeeping is launched from the host per step
            from .flat_adamw import DeviceSchedule, FlatGradScaler
            self.scheduler = (DeviceSchedule(self.optimizer, "exp", opt.iters, 0.1) if exp_decay else
                              DeviceSchedule(self.optimizer, "cosine", opt.iters, eta_min or 5e-5))
            self.scaler = FlatGradScaler(self.device_type, enabled=amp)
        else:
            if exp_decay:  # teacher: 0.1^(iter/iters) (main_just_train_tea.py:293-296)
                self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lambda it: 0.1 ** min(it / opt.iters, 1))
            else:  # student: cosine to eta_min (main_distill_mutual.py:346-348)
                self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=opt.iters, eta_min=eta_min or 5e-5)
            self.scaler = torch.amp.GradScaler(self.device_type, enabled=amp)
        self._l1_folded = False
        if self.flat_opt and amp and not self.dp.enabled and model.model_type == "vm" and os.environ.get("PVD_INF_CHECK_RIDE", "1") != "0":
            # the scaler's inf check of the VM model's gradients rides on the launch that completes them (the table scatter + the head's
            # weight-gradient reduction, pvd_head_dw_rider.found_inf): no check kernel between the scatter and the update on the step's
            # chain.  Under ray-DP the check has to follow the exchange (an overflow on one rank skips the step on all): it stays.
            model._inf_check_in_backward = (self.optimizer.inf_flag(), self.optimizer.note_checked_by_backward)
        else:
            model.__dict__.pop("_inf_check_in_backward", None)  # (an earlier trainer's: its flag is not this one's)
        if self.flat_opt and model.model_type == "hash" and (not self.dp.enabled or os.environ.get("PVD_DP_HASH_WIRE", "f16") == "f16"):
            # the hash table's f16 scatter-add result goes straight into the update kernel.  Under ray-DP it crosses the links AS IT IS
            # (_exchange: the half table summed in half precision, the heads' fp32 gradients next to it): 21 instead of 42 MB per
            # step, and the arithmetic the reference has for this gradient -- ONE half-precision table that every sample of the batch
            # adds into (gridencoder.cu:297-304, grid.py:105-123).  PVD_DP_HASH_WIRE=f32: widened into the fp32 bucket first
            # (rounds 1-5).
            model.encoder.embeddings._pvd_half_grad_taker = self.optimizer.accept_half_grad
        if self.flat_opt:
            self.flat = _FlatOptGrads(self.optimizer)
            # the optimizer may hold deferred weight decay for table rows nothing reads (FlatAdamW.flush): whoever reads whole
            # tables -- state_dict / checkpoints, resampling -- gets them brought up to date first
            model._pvd_flush_params = self.optimizer.flush
            model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.optimizer.flush())
        else:
            self.flat = FlatGrads([p for g in self.optimizer.param_groups for p in g["params"]])
        self.global_step = 0

    def _l1_term(self, partials_only=False):
        """l1_reg_weight * density_loss() for the VM model (utils.py:1101-1104 / just_train_tea/utils.py:573-581).
        A parameter-only term: under ray-DP every rank adds 1/G of it.  With the flat optimizer its gradient is
        applied inside the update kernel (after the all-reduce, full weight) and only the value is computed here."""
        o, m = self.opt, self.model
        if self.flat_opt:
            if not self._l1_folded:
                self.optimizer.set_l1([*m.sigma_mat, *m.sigma_vec], o.l1_reg_weight)
                self._l1_folded = True
            if partials_only:
                return None
            return self.optimizer.l1_value(1.0 / self.dp.world_size)
        return m.density_loss() * (o.l1_reg_weight / self.dp.world_size)

    def _backward(self, loss):
        sc = self.scaler
        if sc.is_enabled() and loss.is_cuda:
            if sc._scale is None:
                sc.scale(loss.detach())  # lazily creates the device-side scale (and nothing else)
            # d(scale * loss) = scale: seed the backward with the scale instead of launching a multiply, a ones-fill
            # and MulBackward around it
            loss.backward(gradient=sc._scale.reshape(loss.shape).to(loss.dtype))
        else:
            sc.scale(loss).backward()

    compact_exchange = False  # DistillTrainer: the occupancy grid is frozen, the touched table rows are known up front

    def _marching_model(self):
        """The model whose occupancy grid decides where samples can be."""
        return self.model

    def _grad_compactor(self):
        """pvd/dp_compact.py: exchange only the table rows that occupied cells can touch (exact: the rest is zero on every
        rank).  Needs the dense L1 gradient out of the flat buffer (folded into the optimizer kernel, or off)."""
        o, m = self.opt, self.model
        if not self.compact_exchange or m.model_type not in ("vm", "tensors"):
            return None
        if m.model_type == "vm" and o.l1_reg_weight > 0.0 and not self.flat_opt:
            return None  # autograd writes w/n * sign(p) into every sigma-plane entry
        marcher = self._marching_model()
        c = getattr(self, "_compactor", None)
        if c is not None and c.occ_epoch != marcher.occ_epoch:
            c = None  # the marcher's occupancy grid was rewritten (renderer.note_occupancy_changed): the touched rows changed
        if c is None:
            from .dp_compact import GradCompactor
            offs = self.optimizer.offsets if self.flat_opt else None
            if offs is None:
                offs, acc = [], 0
                for p in self.flat.params:
                    offs.append(acc)
                    acc += p.numel()
            c = GradCompactor(m, self.flat.params, offs, self.device, marcher=marcher)
            c.agreed = True
            if self.dp.enabled and self.dp.capture is None and c.idx is not None:
                # every rank must move the same rows (same buffer sizes in the collective, nothing left out): compare
                # size and checksum of the index set once; on any disagreement ALL ranks use the dense exchange
                sig = torch.tensor([c.idx.numel(), float(c.idx.sum())], dtype=torch.float64, device=self.device)
                sig = torch.cat([sig, -sig])
                dist.all_reduce(sig, op=dist.ReduceOp.MAX, group=self.dp.group)
                c.agreed = bool(sig[0] == -sig[2]) and bool(sig[1] == -sig[3])
            self._compactor = c
        return c if (c.fraction < 0.7 and c.agreed) else None

    def _zero_grads(self, first_in_recording=False):
        """zero_grad.  With the flat optimizer and a frozen occupancy grid only the rows a sample can touch are ever
        written (the same set the compact exchange moves), so after one full clear only those are cleared and
        inf-checked (FlatAdamW.set_touched); PVD_TOUCHED_SET=0 turns that off.  first_in_recording: the first zero_grad recorded
        into a graph clears everything (FlatAdamW.zero_grad(full=True))."""
        if self.flat_opt:
            c = self._grad_compactor() if os.environ.get("PVD_TOUCHED_SET", "1") != "0" else None
            if c is not self.optimizer.touched:
                self.optimizer.set_touched(c)
        self.flat.zero_(full=first_in_recording)

    # ---- ray-DP exchange, round 6 form (recorded steps with the flat optimizer's two-part update and a compact set)
    _xlayouts = None
    _v2_off = False  # the touched set is not whole aligned groups of four in the optimizer's list order: the classic sequence stays
    _sharded_stale = False  # moments of the other ranks' rows are out of date on this rank (a sharded update ran)

    def _exchange_mode(self, c):
        """None = the classic sequence (gather -> all-reduce -> scatter + inf check -> update), else "allreduce" | "sharded":
        gather (zeroes the rows behind itself, looks at what it moves, raises the buffer's flag words) -> collective(s) -> part B of
        the update reads the buffer directly.  PVD_DP_EXCHANGE = allreduce (default) | sharded | classic."""
        mode = os.environ.get("PVD_DP_EXCHANGE", "allreduce")
        if (mode == "classic" or c is None or self._v2_off or not self.flat_opt or self.device_type != "cuda" or c is not self.optimizer.touched
                or not self.optimizer._outside_is_zero or os.environ.get("PVD_DP_WIRE", "f32") != "f32"
                or getattr(self, "_overlap_with_exchange", None) is not None or not self.optimizer.compact_ready()):
            return None
        return "sharded" if (mode == "sharded" and (self.dp.world_size > 1 or os.environ.get("PVD_DP_FORCE") == "1")) else "allreduce"

    def _xlayout(self, c, chunks, with_params):
        from .dp_compact import ExchangeLayout
        key = (id(c), chunks, with_params)
        if self._xlayouts is None or self._xlayouts[0] != key:
            assert not torch.cuda.is_current_stream_capturing(), "the exchange layout is built by the eager warm-up steps"
            self._xlayouts = (key, ExchangeLayout(c, self.optimizer._warm_B, chunks, self.device, with_params=with_params))
        return self._xlayouts[1]

    def prepare_exchange(self):
        """(before a recording, eagerly) build the exchange layout the recorded steps will use."""
        if not self.dp.enabled or not self.flat_opt:
            return
        c = self._grad_compactor()
        mode = self._exchange_mode(c)
        if mode is not None:
            try:
                self._xlayout(c, self.dp.world_size if mode == "sharded" else 1, mode == "sharded")
            except AssertionError:
                self._v2_off = True

    def _exchange_v2(self, c, mode):
        import pvd_hip
        o, dp = self.optimizer, self.dp
        sharded = mode == "sharded"
        L = self._xlayout(c, dp.world_size if sharded else 1, sharded)
        pvd_hip.segments_gather_zero_check(self.flat.flat, L.segs, L.xbuf, L.xbuf[L.slot:], L.chunk, L.chunks)
        clear = (L.xbuf[L.slot:], L.chunk, L.chunks)
        if not sharded:
            dp.all_reduce_sum_(L.xbuf)
            o.take_compact(L.xbuf, L.xbuf[L.slot:L.slot + 1], clear, (0, L.n_groups))
            return
        mine = L.chunk_of(L.xbuf, dp.rank)
        dp.reduce_scatter_sum_(mine, L.xbuf)  # (in place: this rank's chunk of the buffer receives the sum)
        pmine = L.chunk_of(L.pbuf, dp.rank)
        o.take_compact(mine, mine[L.slot:L.slot + 1], clear, L.rows_of(dp.rank), param_out=pmine)

        def after_update():
            dp.all_gather_(L.pbuf, pmine)
            pvd_hip.segments_op(pvd_hip.SEG_SCATTER, o.flat_p, L.segs, buf=L.pbuf)  # every rank's updated rows into the parameters
            pvd_hip.note_weights_changed(o.params)
        self._after_update = after_update
        self._sharded_stale = True

    def sync_sharded_state(self):
        """After sharded updates a rank holds current moments only for its own rows: before anything else updates all rows (an eager
        step, a recording in another form) or reads the optimizer state, every rank's rows are all-gathered -- one pass per moment."""
        if not self._sharded_stale:
            return
        import pvd_hip
        assert not torch.cuda.is_current_stream_capturing()
        L, o, dp = self._xlayouts[1], self.optimizer, self.dp
        o.flush()
        for buf in (o.flat_m, o.flat_v):
            pvd_hip.segments_op(pvd_hip.SEG_GATHER, buf, L.segs, buf=L.pbuf)
            dp.all_gather_(L.pbuf, L.chunk_of(L.pbuf, dp.rank).clone())
            pvd_hip.segments_op(pvd_hip.SEG_SCATTER, buf, L.segs, buf=L.pbuf)
        self._sharded_stale = False

    def _exchange(self):
        if self.dp.enabled:
            c = self._grad_compactor()
            mode = self._exchange_mode(c)
            if mode is not None:
                return self._exchange_v2(c, mode)
            if self._sharded_stale and not torch.cuda.is_current_stream_capturing():
                self.sync_sharded_state()
            ov = getattr(self, "_overlap_with_exchange", None)
            # PVD_DP_WIRE=f16 | bf16 (opt-in, NOT the reference's arithmetic): the gradient crosses the links in 16 bits -- half the
            # bytes of the one exchange that bounds the multi-GPU step.  f16 relies on the loss scale (AMP): an
            # overflow on the wire is an inf in the gradient, which the scaler's check (it runs after the exchange) answers by
            # skipping the step and halving the scale, as for any other overflow; bf16 keeps fp32's range at 8 bits of mantissa.
            wire = {"f16": torch.float16, "bf16": torch.bfloat16}.get(os.environ.get("PVD_DP_WIRE", "f32"))
            hg = getattr(self.optimizer, "_half_grad", None) if self.flat_opt else None
            if c is None and hg is not None:
                # a hash model: the table's gradient is the half-precision buffer the update will read; its fp32 range holds zero_grad's
                # zeros and stays at home.  Two collectives: the half table (SUM in half precision: an overflow is an inf the scaler's
                # check -- it follows the exchange -- answers like any other), and the rest of the fp32 bucket (the heads: a few KB,
                # always fp32 -- PVD_DP_WIRE has nothing to narrow here).  After a second backward before the step the fp32 range
                # carries that backward's gradient (FlatAdamW.accept_half_grad refused it): then the whole bucket crosses as well.
                lo, hi, g16 = hg
                self.dp.all_reduce_sum_(g16, overlap=ov)
                flat = self.flat.flat
                if self.optimizer._half_range_dirty:
                    self.dp.all_reduce_sum_(flat)
                else:
                    if lo > 0:
                        self.dp.all_reduce_sum_(flat[:lo])
                    if hi < flat.numel():
                        self.dp.all_reduce_sum_(flat[hi:])
            elif c is None:
                if wire is None:
                    self.dp.all_reduce_sum_(self.flat.flat, overlap=ov)  # one bucket, SUM (losses are already global objectives)
                else:
                    w16 = self.flat.flat.to(wire)
                    self.dp.all_reduce_sum_(w16, overlap=ov)
                    self.flat.flat.copy_(w16)
            else:
                buf = c.gather(self.flat.flat)
                if wire is None:
                    self.dp.all_reduce_sum_(buf, overlap=ov)
                else:
                    w16 = buf.to(wire)
                    self.dp.all_reduce_sum_(w16, overlap=ov)
                    buf = w16.to(torch.float32)
                # the exchanged rows are exactly what the scaler's check would look at (the optimizer's touched set IS this compactor):
                # they are looked at while they are put back, and the step has one dependent launch fewer
                checked = (self.flat_opt and c is self.optimizer.touched and self.optimizer._outside_is_zero
                           and bool(getattr(self.scaler, "_enabled", False)) and os.environ.get("PVD_INF_CHECK_RIDE", "1") != "0")
                c.scatter(self.flat.flat, buf, found_inf=self.optimizer.inf_flag() if checked else None)
                if checked:
                    self.optimizer.note_checked_by_backward()

    _after_update = None

    def _optimize(self):
        self.scaler.step(self.optimizer)
        self.scaler.update()
        after, self._after_update = self._after_update, None
        if after is not None:  # (a sharded update: all-gather of the updated rows)
            after()

    def _backward_and_step(self, loss):
        self._backward(loss)
        self._exchange()
        self._optimize()
        self.scheduler.step()
        self.global_step += 1

    # ---- hipGraph capture of the step (launch-bound otherwise: ~330 kernels of a few us each)
    steps_per_replay = 1

    def capture(self, body, warmup=3, steps_per_graph=1):
        """Capture `body()` (forward + loss + backward) and the optimizer into HIP graph(s).  steps_per_graph > 1 records
        that many consecutive steps into the one graph (a replay then costs one graph launch -- ~8 us of launch latency
        between replays in the step's timeline -- for several steps; `replay()` advances the step count accordingly).  Single GPU: one graph for
        the whole step.  Ray-DP: the capture is SEGMENTED at every collective (`SegmentedCapture`): the all-reduce of the
        four loss sums and the gradient exchange run eagerly between graph replays, so nothing depends on the
        communication library being capturable.  Everything that changes per step lives on the device (lr, loss rates,
        pose index, RNG state, loss scale), so a replay is a faithful step."""
        assert self.device_type == "cuda"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._zero_grads()
                out = body()
                self._backward(out[0])
                self._exchange()
                self._optimize()
                self.scheduler.step()
                self.global_step += 1
                del out  # drop the autograd graph of the warm-up pass before capturing
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        cap = SegmentedCapture(self.device)
        self.dp.capture = cap
        if self.dp.enabled and not self.dp.ingraph:
            steps_per_graph = 1  # graphs cut at the collectives: one step per chain
        self._graph_zeroes = False
        if max(1, int(steps_per_graph)) >= 2 and not self.dp.enabled:
            self._fold_launches(True)
        try:
            with cap:
                for k_rec in range(max(1, int(steps_per_graph))):
                    self._zero_grads(first_in_recording=k_rec == 0)
                    self._static_out = body()
                    if os.environ.get("PVD_TEST_FAIL_IN_CAPTURE") == "1":  # exercises the callers' eager fallback
                        raise RuntimeError("forced failure inside the capture (PVD_TEST_FAIL_IN_CAPTURE)")
                    self._backward(self._static_out[0])
                    self._exchange()  # (breaks the capture around its all-reduce under ray-DP; nothing otherwise)
                    self._optimize()
        finally:
            self.dp.capture = None
            self._fold_launches(False)
        self._cap = cap
        self.steps_per_replay = max(1, int(steps_per_graph))
        # (only a step that relies on a touched-row set is tied to the occupancy grid it was captured with)
        self._captured_occ_epoch = self._marching_model().occ_epoch if (self.flat_opt and self.optimizer.touched is not None) else None
        return self._static_out  # the warm-up steps above are real steps; the capture itself records without running

    def _fold_launches(self, on):
        """While several steps are recorded into one graph the update zeroes the gradients it has read (FlatAdamW.zero_in_step), so
        only the first step of the graph launches a zero_grad.  What the host knows about the gradients after such a recording
        is settled by replay() (a recording runs nothing)."""
        if self.flat_opt:
            self.optimizer.zero_in_step = bool(on)
            if on:
                self._graph_zeroes = False
                # the first recorded step must record its zero_grad whatever the previous replay left behind: a graph that
                # relies on "the last replay zeroed the gradients" accumulates stale ones after any eager step
                self.optimizer._zeroed_by_step = False
            else:
                # what the LAST recorded update did (it zeroes only in the touched-set / warm-list form, FlatAdamW.step) is what
                # a replay leaves behind; the recording itself ran nothing, so right now the host knows nothing
                self._graph_zeroes = bool(self.optimizer._zeroed_by_step)
                self.optimizer._zeroed_by_step = False

    def _after_failed_capture(self):
        """A recording that raised left nothing on the device (nothing runs while capturing) but may have left host-side
        bookkeeping half way through a step: settle it before recording again."""
        import traceback
        traceback.print_exc()
        torch.cuda.synchronize()
        self.dp.capture = None
        self._fold_launches(False)
        self._graph_zeroes = False
        self._after_update = None  # (a sharded update recorded half way: its all-gather belongs to a step that never ran)
        if self.flat_opt:
            self.optimizer._compact = None  # an exchange buffer handed over to an update that never came
            self.optimizer._half_grad = None  # a half-precision table gradient handed over by a backward whose update never came
            self.optimizer._part_a_owed = None  # (a two-part update recorded half way: nothing of it ran)
            self.optimizer.end_two_part(failed=True)

    def replay(self):
        assert getattr(self, "_captured_occ_epoch", None) in (None, self._marching_model().occ_epoch), \
            "the occupancy grid changed since the step was captured (touched-row set is stale): capture again"
        if self.flat_opt:
            import pvd_hip
            pvd_hip.note_weights_changed(self.optimizer.params)  # the captured optimizer kernel rewrites the parameters
        if self.flat_opt:
            self.optimizer.before_replay()  # (stale L1 partial sums of another launch shape)
        self._cap.replay()
        if self.flat_opt:
            if getattr(self, "_replay_owes_part_a", None) is not None:
                self.optimizer.note_carried_part_a(self._replay_owes_part_a)
            self.optimizer.note_device_steps(self.steps_per_replay)
            if getattr(self, "_graph_zeroes", False):
                self.optimizer._zeroed_by_step = True  # the graph's last update left the touched set clean
        for _ in range(self.steps_per_replay):
            self.scheduler.step()
        self.global_step += self.steps_per_replay
        return self._static_out


class DistillTrainer(_TrainerBase):
    """One distillation step = student render (marches, grads on) -> teacher render on the inherited
    samples (no grad) -> staged loss -> backward -> AdamW (reference: utils.py:804-824, 954-1189)."""

    compact_exchange = True  # the occupancy grid is never updated while distilling

    def __init__(self, opt, model_tea, model_stu, device, fp16=True, dp=None):
        for p in model_tea.parameters():
            p.requires_grad = False
        lr = opt.lr * (0.1 if opt.model_type == "mlp" else 1.0)  # main_distill_mutual.py:241-242
        super().__init__(opt, model_stu, lr, device, fp16, dp, eta_min=5e-5)
        self.model_tea = model_tea.train()  # `training` selects the train branch of run_cuda; teacher is frozen
        self.model_stu = model_stu.train()
        self.loss = _make_loss(opt.loss_type, self.dp)
        self.loss_rate_fea_sc = opt.loss_rate_fea_sc  # host shadow (only used for the > 0 tests)
        # device-side loss rates [rgb, fea, sigma, colour]; the feature rate decays on the device every step
        self.rates = torch.tensor([opt.loss_rate_rgb, opt.loss_rate_fea_sc, opt.loss_rate_sigma, opt.loss_rate_color],
                                  dtype=torch.float32, device=self.device)
        self.fea_rate = self.rates[1:2]
        self.fused_loss = getattr(model_stu.ops, "distill_loss", None)

    def render_kwargs(self):
        o = self.opt
        return dict(dt_gamma=o.dt_gamma, max_steps=o.max_steps)

    def _marching_model(self):
        # whoever renders first marches (renderer.py:365-411); the other model inherits its samples
        return self.model_stu if bool(getattr(self.opt, "render_stu_first", True)) else self.model_tea

    def prefetch(self, batch_fn):
        """The parameter-independent prefix of a step: batch, march, the frozen teacher's forward and compositing.  Its
        results feed compute_loss(pre=...); under ray-DP it is captured as its own graph and replayed for step k+1 while
        step k's gradient exchange is in flight."""
        return self.prefetch_teacher(self.prefetch_march(batch_fn))

    def prefetch_march(self, batch_fn):
        """First half of the prefix: the batch and the march (depend on the occupancy grid only)."""
        rays_o, rays_d, bg, *rest = batch_fn()
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
            inh, nf = self.model_stu.march(rays_o, rays_d, perturb=True, force_all_rays=False, nears_fars=rest[0] if rest else None,
                                           **self.render_kwargs())
        return dict(rays_o=rays_o, rays_d=rays_d, bg=bg, inh=inh, nf=nf)

    def prefetch_teacher(self, part):
        """Second half: the frozen teacher's forward and compositing on the marched samples (rebinds the teacher's
        feature_sigma_color: not before the objective of the step in flight has been issued)."""
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"), torch.no_grad():
            out_tea = self.model_tea.render(part["rays_o"], part["rays_d"], staged=False, bg_color=part["bg"], perturb=True,
                                            force_all_rays=False, inherited_params=part["inh"], nears_fars=part["nf"], premarched=True,
                                            **self.render_kwargs())
        tea = self.model_tea
        attrs = {k: getattr(tea, k, None) for k in ("feature_sigma_color", "sigma_l", "color_l")}
        return dict(part, out_tea=out_tea, tea_attrs={k: v for k, v in attrs.items() if torch.is_tensor(v)})

    def compute_loss(self, rays_o, rays_d, bg_color, nears_fars=None, pre=None):
        o, stu, tea = self.opt, self.model_stu, self.model_tea
        o.global_step = self.global_step
        self.__dict__.pop("_ride", None)  # (an ObjectiveRide belongs to the step that created it)
        kw = self.render_kwargs()
        kw_stu = dict(kw, nears_fars=nears_fars) if nears_fars is not None else kw  # the batch kernel already intersected the box
        if pre is not None:
            for k, v in pre.get("tea_attrs", {}).items():  # what the teacher's forward left on the model (this prefix's, not the latest)
                setattr(tea, k, v)
            out_tea = dict(pre["out_tea"])
            if out_tea.get("image") is not None and pre.get("replayed_ahead", False):
                out_tea["image"] = out_tea["image"].clone()  # the prefix graph overwrites its outputs one step ahead
            # the teacher's outputs exist already: the stage-3 objective can ride on the student's compositing launches
            # (ObjectiveRide: two launches fewer on the chain)
            ride = None
            # (under ray-DP too, since round 6: ONE all-reduce of the forward launch's partial sums -- a count that is the same on every
            # rank, ~20 KB -- sits between the two compositing launches and the backward launch finishes the objective from the summed
            # partials; PVD_DP_EXCHANGE=classic keeps the four separate objective launches of rounds 1-5)
            l1_on_host = o.l1_reg_weight > 0.0 and o.model_type == "vm" and not self.flat_opt
            dp_ride = self.dp.enabled and not l1_on_host and os.environ.get("PVD_DP_EXCHANGE", "allreduce") != "classic"
            if (self.fused_loss is not None and o.loss_type == "normL2" and (not self.dp.enabled or dp_ride) and torch.is_grad_enabled()
                    and self._stage_of(self.global_step) == 3 and out_tea.get("image") is not None
                    and min(o.loss_rate_color, o.loss_rate_sigma, self.loss_rate_fea_sc * 0.995, o.loss_rate_rgb) > 0.0
                    and torch.is_tensor(getattr(tea, "feature_sigma_color", None)) and torch.is_tensor(getattr(tea, "color_l", None))
                    and tea.feature_sigma_color.dim() == 2 and tea.feature_sigma_color.shape[-1] == 16
                    and tea.feature_sigma_color.dtype == torch.float32 and pre["rays_o"].is_cuda):
                from .losses import ObjectiveRide
                # the objective is finished inside the compositing backward's launch -- never when the L1 value is added to the loss
                # by the host afterwards (L1 on, VM student, non-flat optimizer: k_loss_final then stays a launch of its own)
                fin = not l1_on_host
                ride = ObjectiveRide(out_tea["image"], tea.feature_sigma_color, tea.color_l, rates_decay=self.rates if fin else None, fea_decay=0.995,
                                     fixed_parts=self.dp.enabled)
                self.dp_objective_rides = bool(self.dp.enabled)
            out_stu = stu.render(pre["rays_o"], pre["rays_d"], staged=False, bg_color=pre["bg"], perturb=True, force_all_rays=False,
                                 inherited_params=pre["inh"], nears_fars=pre["nf"], premarched=True, objective=ride, **kw)
            self._ride = ride if (ride is not None and ride.S is not None) else None
        else:
            out_tea = None
            if not bool(getattr(o, "render_stu_first", True)) and stu.cuda_ray:
                # the TEACHER marches and the student inherits its samples (utils.py:1020-1043, renderer.py:392-411)
                with torch.no_grad():
                    out_tea = tea.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False, **kw_stu)
                out_stu = stu.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                                     inherited_params=out_tea["inherited_params"], nears_fars=out_tea.get("nears_fars"), **kw)
            else:
                out_stu = stu.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False, **kw_stu)
            if out_tea is None:
                with torch.no_grad():
                    out_tea = tea.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                                         inherited_params=out_stu["inherited_params"], nears_fars=out_stu.get("nears_fars"), **kw)
        self.loss_rate_fea_sc *= 0.995  # decays every step (utils.py:1044)
        have_fea = stu.feature_sigma_color is not None and tea.feature_sigma_color is not None
        pred_stu, pred_tea = out_stu.get("image"), out_tea.get("image")
        fused = (self.fused_loss is not None and o.loss_type == "normL2" and have_fea and pred_stu is not None and pred_stu.is_cuda
                 and "stage1" not in out_stu and "stage2" not in out_stu
                 and min(o.loss_rate_color, o.loss_rate_sigma, self.loss_rate_fea_sc, o.loss_rate_rgb) > 0.0
                 and stu.feature_sigma_color.dtype == torch.float32 and tea.feature_sigma_color.dtype == torch.float32
                 # the fused objective reads [M,16] rows whose column 0 is sigma_l (geo_feat_dim = 15, the reference's default)
                 and stu.feature_sigma_color.dim() == 2 and stu.feature_sigma_color.shape[-1] == 16
                 and tea.feature_sigma_color.shape == stu.feature_sigma_color.shape)
        # models without a feature vector (the Plenoxel student): the same fused objective with the rows holding sigma_l alone
        # (rgb + sigma + colour terms; utils.py:1109-1176 with the feature term absent)
        fused_nofea = (not fused and not have_fea and self.fused_loss is not None and o.loss_type == "normL2" and pred_stu is not None
                       and pred_stu.is_cuda and "stage1" not in out_stu and "stage2" not in out_stu
                       and min(o.loss_rate_color, o.loss_rate_sigma, o.loss_rate_rgb) > 0.0
                       and getattr(stu, "sigma_l", None) is not None and getattr(tea, "sigma_l", None) is not None
                       and stu.sigma_l.dim() == 1 and tea.sigma_l.shape == stu.sigma_l.shape
                       and not (o.l1_reg_weight > 0.0 and o.model_type == "vm"))
        assert fused or not getattr(getattr(self, "_ride", None), "decayed", False), \
            "the compositing launch applied the feature rate's decay for a fused objective that is not being formed"
        if not fused:
            self.fea_rate.mul_(0.995) if not fused_nofea else None  # (the fused objective decays the device-side rate inside its own kernel)
        info = {}
        loss = 0.0
        if "stage1" in out_stu and self.loss_rate_fea_sc > 0.0 and have_fea:
            l_fea = self.loss(stu.feature_sigma_color, tea.feature_sigma_color)
            info["fea"] = l_fea.detach()
            return loss + self.fea_rate * l_fea, info, None, None
        if "stage2" in out_stu:
            l_col = self.loss(stu.color_l, tea.color_l)
            l_sig = self.loss(stu.sigma_l, tea.sigma_l)
            if o.loss_rate_color > 0.0:
                loss = loss + o.loss_rate_color * l_col
            if o.loss_rate_sigma > 0.0:
                loss = loss + o.loss_rate_sigma * l_sig
            if self.loss_rate_fea_sc > 0.0 and have_fea:
                loss = loss + self.fea_rate * self.loss(stu.feature_sigma_color, tea.feature_sigma_color)
            info.update(color=l_col.detach(), sigma=l_sig.detach())
            return loss, info, None, None

        if fused_nofea:
            l3, norms = self.fused_loss(pred_stu, pred_tea, stu.sigma_l.float().unsqueeze(-1), tea.sigma_l.float().unsqueeze(-1),
                                        stu.color_l.float(), tea.color_l.float(), self.rates, self.dp, fea_decay=0.995)
            info["rgb"] = norms[0]
            return l3, info, pred_stu, pred_tea
        if fused:
            # all four norm terms (utils.py:1109-1176), the feature-rate decay and the value of the L1 term: one objective
            extra = None
            if o.l1_reg_weight > 0.0 and o.model_type == "vm":
                if self.flat_opt:
                    self._l1_term(partials_only=True)
                    extra = self.optimizer.l1_partials(1.0 / self.dp.world_size)
            add_l1 = o.l1_reg_weight > 0.0 and o.model_type == "vm" and extra is None
            ride = self.__dict__.pop("_ride", None)
            rkw = {} if ride is None else {"ride": ride}
            l4, norms = self.fused_loss(pred_stu, pred_tea, stu.feature_sigma_color, tea.feature_sigma_color, stu.color_l.float(),
                                        tea.color_l.float(), self.rates, self.dp, fea_decay=0.995, extra=extra, **rkw)
            loss = l4  # (loss is still the python 0.0 here: no "0 + x" launch)
            if add_l1:
                loss = loss + self._l1_term()
            info["rgb"] = norms[0]
            return loss, info, pred_stu, pred_tea
        if o.loss_type == "normL2":
            l_rgb = self.dp.global_norm_l2(pred_tea - pred_stu)
        elif o.loss_type == "normL1":
            l_rgb = self.dp.global_norm_l1(pred_tea - pred_stu)
        else:
            l_rgb = self.dp.global_mean((pred_tea.float() - pred_stu.float()) ** 2)
        loss = loss + l_rgb * o.loss_rate_rgb
        if o.l1_reg_weight > 0.0 and o.model_type == "vm":
            loss = loss + self._l1_term()
        if self.loss_rate_fea_sc > 0.0 and have_fea:
            loss = loss + self.fea_rate * self.loss(stu.feature_sigma_color, tea.feature_sigma_color)
        if o.loss_rate_color > 0.0:
            loss = loss + o.loss_rate_color * self.loss(stu.color_l, tea.color_l)
        if o.loss_rate_sigma > 0.0:
            loss = loss + o.loss_rate_sigma * self.loss(stu.sigma_l, tea.sigma_l)
        info["rgb"] = l_rgb.detach()
        return loss, info, pred_stu, pred_tea

    def train_step(self, rays_o, rays_d, bg_color, nears_fars=None):
        o, stu = self.opt, self.model_stu
        if getattr(o, "update_stu_extra", False) and stu.cuda_ray and self.global_step % o.update_extra_interval == 0:
            # the reference's option (utils.py:786-796): the student's own occupancy grid follows its density while
            # distilling.  update_extra_state bumps stu.occ_epoch, which retires the touched-row set / compact exchange
            # (rebuilt by _zero_grads below from the new grid).
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
                stu.update_extra_state()
        self._zero_grads()
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
            loss, info, pred_stu, pred_tea = self.compute_loss(rays_o, rays_d, bg_color, nears_fars)
        self._backward_and_step(loss)
        return loss.detach(), info, pred_stu, pred_tea

    def capture_step(self, batch_fn, steps_per_graph=1):
        """Capture `batch_fn() -> (rays_o, rays_d, bg)` + the whole step into HIP graph(s); the stage
        (which loss terms exist) is frozen at capture time, so re-capture when the stage changes.  Three eager warm-up
        steps run first (real steps: global_step advances by 3)."""
        def body():
            rays_o, rays_d, bg, *rest = batch_fn()
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                return self.compute_loss(rays_o, rays_d, bg, *rest)
        # the 3 eager warm-up steps inside capture() / the pipelined recordings are real steps: they advance global_step, so the
        # stage that gets captured is the one AFTER them, and they must not straddle a stage boundary (the warm-up and the
        # captured step would then build different losses)
        warm = 3
        if self.flat_opt:
            self.optimizer.flush()  # (a part of the previous recording's last update may still be owed: FlatAdamW.carry_last)
            self.sync_sharded_state()  # (a sharded recording leaves the other ranks' rows' moments behind: current before anything else runs)
        self._replay_owes_part_a = None  # what replay() tells the optimizer afterwards belongs to the recording made below
        self.fused_spans = None
        stage = self._stage_of(self.global_step)
        assert self._stage_of(self.global_step + warm) == stage, \
            "capture_step: the %d warm-up steps cross a stage boundary (global_step %d); step eagerly past it first" % (warm, self.global_step)
        # the pipelined recordings march ahead with the STUDENT (prefetch_march); when the teacher marches
        # (render_stu_first = False) the touched-row set, the compact exchange and the cold bitmap belong to the teacher's
        # grid, so those steps are recorded back to back through compute_loss, which honours the flag
        stu_marches = bool(getattr(self.opt, "render_stu_first", True))
        if self.dp.enabled and self.dp.ingraph:
            # ONE graph for the whole step, both collectives recorded into it (no graph cuts, no eager calls per step); the
            # gradient exchange is then not overlapped with the next step's prefix -- a replayed child graph cannot be
            # recorded into a capture here (tools/probe_rccl_capture.py) -- which costs less than the ~60 us of fixed
            # overhead the three-graph form pays on every step
            try:
                pipe = os.environ.get("PVD_DP_PIPELINE", "1")  # 0: off; 1: when there is more than one rank; 2: always (tests)
                if stu_marches and stage == 3 and steps_per_graph > 1 and (pipe == "2" or (pipe == "1" and self.dp.world_size > 1)):
                    out = self._capture_ingraph_pipelined(batch_fn, body, steps_per_graph)
                else:
                    out = self.capture(body, steps_per_graph=steps_per_graph)
            except Exception:
                self._after_failed_capture()
                self.dp.ingraph = False  # fall back to graphs cut at the collectives
                # (the failed attempt's warm-up steps were real steps and are not repeated; one step per chain)
                out = self.capture(body, warmup=0, steps_per_graph=1)
                self.capture_fallback = "segmented"
        elif self.dp.enabled and stu_marches and stage == 3 and os.environ.get("PVD_DP_OVERLAP", "1") != "0":
            out = self._capture_pipelined(batch_fn, body)
        elif not self.dp.enabled and stage == 3 and steps_per_graph > 1 and stu_marches and pvd_forked_graphs_ok():
            # single GPU, several steps per graph: the same fork -- next step's batch / march / teacher forward (ALU- and
            # latency-bound) recorded next to this step's table scatter + inf check + AdamW inside the one graph
            try:
                out = self._capture_ingraph_pipelined(batch_fn, body, steps_per_graph)
            except Exception:
                self._after_failed_capture()
                self.pipelined_ingraph = False
                self.capture_fallback = "back-to-back"
                out = self.capture(body, warmup=0, steps_per_graph=steps_per_graph)  # the same steps recorded back to back (warm-ups done)
        else:
            out = self.capture(body, steps_per_graph=steps_per_graph)
        self._captured_stage = self._stage_of(self.global_step)
        return out

    def _capture_ingraph_pipelined(self, batch_fn, body, steps_per_graph):
        """Several steps per graph (single GPU, or ray-DP with the collectives recorded into the graph): the parameter-
        independent prefix of step k + 1 (batch, march, frozen teacher's forward and compositing) is recorded on a FORKED
        stream next to step k -- the fork at the START of the step, the join behind its update -- inside one graph: no extra
        graph launches, one fork / join pair per step.  The scatter waits on the memory side, the update on HBM and the
        exchange on xGMI while the prefix is instruction- and L2-bound, so the two chains share the chip (DESIGN section 6;
        0.379 -> 0.327 ms/step on one GPU).  The prefix next to the LAST step of the graph feeds the FIRST step of the next
        replay through a static home (`CarriedPrefix`), so every step has its prefix overlapped.  Same batches in the same
        order, same update rule as the back-to-back recording (tests/test_hip_graph.py, tests/test_hip_dp_graph.py).
        The other placements of the fork that were built and measured slower (before the backward / the update, between the
        head's and the table's backward, one fork per graph, two steps deep, the teacher held back) are recorded with their
        numbers in profiles/r03_fork_modes.txt, r04_fork_modes.txt, r04_fork_optimizer.txt and DESIGN sections 9.3 / 10.9."""
        assert self.device_type == "cuda" and (not self.dp.enabled or self.dp.ingraph)
        K = int(steps_per_graph)
        assert K >= 2, "with one step per graph the step's own scatter would still read the samples the carried prefix overwrites"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # eager warm-up steps (communicators, compactor, caches), as in capture()
            for _ in range(3):
                self._zero_grads()
                out = body()
                self._backward(out[0])
                self._exchange()
                self._optimize()
                self.scheduler.step()
                self.global_step += 1
                del out
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph_zeroes = False
        if not self.dp.enabled:
            self._fold_launches(True)
        with torch.cuda.stream(side):
            carried = CarriedPrefix(self.prefetch(batch_fn))  # prologue: the first replayed step's prefix, eagerly
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # Two-part AdamW (FlatAdamW.two_part, PVD_ADAMW_SPLIT=late, the default; =0: one launch): the update behind the table scatter
        # covers only what the backward can have written (touched rows, the heads); the L1-only / still-decaying rows -- half of the
        # update's bytes, read by nothing but their own update -- are updated ONE STEP LATER on the next step's branch (between its
        # march and its teacher forward since round 6; at the end of the branch before), next to the head backward and the table
        # scatter, which wait on the matrix cores and on the memory side's atomic units, not on HBM: 0.280 vs 0.296 ms/step
        # (profiles/r04_adamw_late_ab.txt; at the START of the branch it measured slower than one launch, profiles/r03_adamw_split_ab.txt).  Bit-identical parameters and moments (tests/test_hip_graph.py).  The objective
        # does not wait for it: the L1 VALUE it reports counts those rows one step late.  Under ray-DP with the collectives in the
        # graph the deferred rows are the ones no sample reaches -- not part of the exchange, updated from the step's scalars,
        # which are the same on every rank -- and the deferred part then runs under the exchange.
        late = (self.flat_opt and os.environ.get("PVD_ADAMW_SPLIT", "late") == "late" and (not self.dp.enabled or self.dp.ingraph))
        scaled = bool(getattr(self.scaler, "_enabled", False))
        self._replay_owes_part_a = None
        fh = getattr(getattr(self.model_stu, "ops", None), "fused_head", None)
        pack_ahead = getattr(fh, "prepack_train_image", None)
        # (measurement, bench.py) record_fused_spans: every recorded launch of the frozen hash teacher's lookup + head writes its own
        # extent {first workgroup's start, last one's end} into a row of fused_spans [K, 2] (pvd_hash_head_forward_fused_span)
        self.fused_spans = None
        if getattr(self, "record_fused_spans", False) and getattr(self.model_tea, "model_type", "") == "hash":
            import pvd_hip
            self.fused_spans = torch.tensor([list(pvd_hip.FUSED_SPAN_INIT)] * K, dtype=torch.int64, device=self.device)
        if late and not self.optimizer.begin_two_part(defer=True):
            late = False  # (no warm-group lists, or no rows of one of the two kinds: one launch)
        if late:
            # ... and the LAST step's deferred part rides on the next replay's first branch instead of trailing the graph
            self.optimizer.carry_last = True
        if self.dp.enabled and self.flat_opt:
            # round 6: the recorded steps' exchange feeds the update directly (gather that zeroes and checks -> collective -> part B, or
            # the single launch of a student without deferred rows); the layout is built here, outside the recording, and the steps
            # record no zero_grad of their own after the first
            self.prepare_exchange()
            if self._exchange_mode(self._grad_compactor()) is not None:
                self._fold_launches(True)
        cap = SegmentedCapture(self.device)
        self.dp.capture = cap
        branch = torch.cuda.Stream(self.device)
        try:
            with cap:
                main = torch.cuda.current_stream()
                pre = carried.pre
                try:
                    for k in range(K):
                        # the fork, first thing in the step: compute_loss below re-installs the teacher outputs of ITS prefix before
                        # it reads them, so the rebinding the next prefix does on the host is harmless
                        branch.wait_stream(main)
                        if self.fused_spans is not None:  # (measurement) this step's teacher launch leaves its extent in row k
                            self.model_tea._fused_span = self.fused_spans[k]
                        with torch.cuda.stream(branch):
                            if pack_ahead is not None and pack_ahead(self.model_stu):
                                # the student's f16 weight image for THIS step's head (5 us on the main chain otherwise): the
                                # weights are final (the fork follows the previous update), the head waits for it
                                packed = torch.cuda.Event()
                                packed.record(branch)
                                self.model_stu._before_head = lambda packed=packed: main.wait_event(packed)
                            marched = self.prefetch_march(batch_fn)
                            if late:
                                # BETWEEN the march and the teacher's forward: the deferred part (HBM streaming) then runs next to the
                                # student's head backward and the teacher's lookup next to the table scatter, instead of the lookup on
                                # the head backward (46 -> 42 us) and the deferred part on the scatter: 0.2683 -> 0.2650 ms/step; at the
                                # START of the branch 0.286 (profiles/r06_part_a_position_ab.txt).  No dependency edge either way.
                                self.optimizer.run_part_a()  # what the previous step's update still owes
                                if k == 0:
                                    self.optimizer.run_carried_part_a(scaled)  # ... and the previous replay's last step
                            pre_next = self.prefetch_teacher(marched)
                            if k + 1 == K:  # for the next replay
                                carried.store(pre_next)
                        self._zero_grads(first_in_recording=k == 0)
                        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                            self._static_out = self.compute_loss(None, None, None, pre=pre)
                        self._backward(self._static_out[0])
                        if os.environ.get("PVD_TEST_FAIL_IN_CAPTURE") in ("1", "forked") and k == 1:  # exercises the fall-backs
                            raise RuntimeError("forced failure inside the forked capture (PVD_TEST_FAIL_IN_CAPTURE)")
                        self._exchange()
                        if late and k + 1 == K:
                            self.optimizer._next_step_is_last = True
                        self._optimize()
                        main.wait_stream(branch)  # join
                        pre = pre_next
                    if late:
                        self.optimizer._part_a_owed = None  # the last step's: recorded on the next replay's first branch
                        self._replay_owes_part_a = scaled
                except Exception:
                    self.model_stu.__dict__.pop("_before_head", None)
                    self.model_stu.__dict__.pop("_train_image_ready", None)
                    main.wait_stream(branch)  # a capture can only be ended with its forked work joined
                    raise
        finally:
            self.dp.capture = None
            self._fold_launches(False)
            if late:
                self.optimizer.end_two_part()
        self._cap = cap
        self.steps_per_replay = K
        self._captured_occ_epoch = self._marching_model().occ_epoch if (self.flat_opt and self.optimizer.touched is not None) else None
        self.pipelined_ingraph = True
        self.pipeline_fork = "start"
        # how the recorded steps update: "0" one launch, "late" two parts (only if the optimizer did split: a student without L1-only
        # rows keeps the single launch)
        self.adamw_split = "late" if (late and getattr(self.optimizer, "_graph_is_two_part", False)) else "0"
        return self._static_out

    def _capture_pipelined(self, batch_fn, body):
        """Ray-DP: the next step's batch / march / teacher forward do not depend on this step's update, so they are a
        separate graph that is replayed while the gradient exchange is in flight:
            [student forward, sums] -> all-reduce(16 B) -> [loss, backward, gather] -> (all-reduce || PREFIX of step k+1)
            -> [scatter, inf check, AdamW]."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # eager warm-up steps (also build the compactor, the communicators, the caches)
            for _ in range(3):
                self._zero_grads()
                out = body()
                self._backward(out[0])
                self._exchange()
                self._optimize()
                self.scheduler.step()
                self.global_step += 1
                del out
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._g_prefix = torch.cuda.CUDAGraph()  # own memory pool: it is replayed out of capture order
        with torch.cuda.graph(self._g_prefix, capture_error_mode="thread_local"):
            self._pre = dict(self.prefetch(batch_fn), replayed_ahead=True)

        def body_pre():
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                return self.compute_loss(None, None, None, pre=self._pre)
        self._overlap_with_exchange = self._g_prefix.replay
        try:
            out = self.capture(body_pre, warmup=0)
        finally:
            self._overlap_with_exchange = None
        self._g_prefix.replay()  # prologue: the first replayed step needs its prefix
        return out

    def _stage_of(self, step):
        st = self.opt.stage_iters
        return 1 if step < st["stage1"] else (2 if step < st["stage2"] else 3)

    def replay_step(self):
        assert self._stage_of(self.global_step) == self._captured_stage, "stage changed: capture_step() again"
        assert self._stage_of(self.global_step + self.steps_per_replay - 1) == self._captured_stage, "the replay would cross a stage boundary"
        loss, info, pred_stu, pred_tea = self.replay()
        return loss.detach(), info, pred_stu, pred_tea


class TeacherTrainer(_TrainerBase):
    """Teacher step: render, MSE against ground-truth pixels (+ VM L1 reg), occupancy update every
    `update_extra_interval` steps (reference: just_train_tea/utils.py:573-581, 841-846)."""

    def __init__(self, opt, model, device, fp16=True, dp=None):
        lr = opt.lr * (0.1 if opt.model_type == "mlp" else 1.0)
        super().__init__(opt, model, lr, device, fp16, dp, exp_decay=True)
        self.model.train()

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color, error_sink=None):
        """error_sink (optional): called with the step's per-ray error [.., N] (mean over the channels of the squared difference, detached)
        -- what --error_map feeds back into the data provider's sampling weights (utils.py:1120-1129; BlenderScene.update_error)."""
        o, m = self.opt, self.model
        if m.cuda_ray and self.global_step % o.update_extra_interval == 0:
            self._update_occupancy()
        self._zero_grads()
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
            out = m.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                           dt_gamma=o.dt_gamma, max_steps=o.max_steps, num_steps=o.num_steps, upsample_steps=o.upsample_steps)
            pred = out["image"]
            loss = self._mse(pred, gt_rgb)
            if o.l1_reg_weight > 0.0 and o.model_type == "vm":
                loss = loss + self._l1_term()
        if error_sink is not None:
            error_sink(((pred.detach().float() - gt_rgb.float()) ** 2).mean(-1))
        self._backward_and_step(loss)
        return loss.detach(), pred

    def _update_occupancy(self):
        """update_extra_state (just_train_tea/utils.py:841-846), and under ray-DP the replicas agree on rank 0's grid afterwards
        (the update draws random cells; RayDP.sync_occupancy)."""
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
            self.model.update_extra_state()
        if self.dp.enabled:
            self.dp.sync_occupancy(self.model)

    def _mse(self, pred, gt):
        """mean over rays and channels of the squared difference (just_train_tea/utils.py:573-581: MSELoss(reduction='none'), .mean(-1),
        .mean()).  One GPU: one launch forward (value + gradient), one multiply backward, where subtract / square / mean and their
        three autograd nodes were six (the library's mse_loss: two + two)."""
        if self.dp.enabled or not pred.is_cuda:
            return self.dp.global_mean((pred.float() - gt.float()) ** 2)
        fused = getattr(self.model.ops, "mse_loss", None)
        if fused is not None and pred.dtype == torch.float32:
            return fused(pred, gt)  # value and gradient in one launch, one multiply backward (pvd_mse_forward)
        return torch.nn.functional.mse_loss(pred.float(), gt.float())

    # ---- a whole block of steps between two occupancy-grid updates as ONE captured graph
    def _block_body(self, batches):
        o, m = self.opt, self.model
        it = {"k": 0}

        def body():
            rays_o, rays_d, gt_rgb, bg_color = batches[it["k"] % len(batches)]
            it["k"] += 1
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
                out = m.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                               dt_gamma=o.dt_gamma, max_steps=o.max_steps, num_steps=o.num_steps, upsample_steps=o.upsample_steps)
                pred = out["image"]
                loss = self._mse(pred, gt_rgb)
                if o.l1_reg_weight > 0.0 and o.model_type == "vm":
                    loss = loss + self._l1_term()
            return loss, pred
        return body

    def _capture_block_pipelined(self, batches):
        """The block with step k + 1's march (near/far, count and write passes: ~50 us, instruction-bound, depends only on the
        occupancy grid, which is fixed inside a block) recorded on a forked stream next to step k's backward (the hash-grid
        scatter sits at the memory-side atomic rate) and update -- the same schedule as DistillTrainer's multi-step graph."""
        o, m = self.opt, self.model
        kw = dict(dt_gamma=o.dt_gamma, max_steps=o.max_steps)

        def march(b):
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                return m.march(b[0], b[1], perturb=True, force_all_rays=False, **kw)

        def loss_of(b, marched):
            inh, nf = marched
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                out = m.render(b[0], b[1], staged=False, bg_color=b[3], perturb=True, force_all_rays=False, inherited_params=inh,
                               nears_fars=nf, premarched=True, own_march=True, num_steps=o.num_steps, upsample_steps=o.upsample_steps, **kw)
                pred = out["image"]
                loss = self._mse(pred, b[2])
                if o.l1_reg_weight > 0.0 and o.model_type == "vm":
                    loss = loss + self._l1_term()
            return loss, pred
        torch.cuda.synchronize()
        cap = SegmentedCapture(self.device)
        self.dp.capture = cap
        branch = torch.cuda.Stream(self.device)
        K = len(batches)
        try:
            with cap:
                main = torch.cuda.current_stream()
                marched = march(batches[0])
                try:
                    for k in range(K):
                        self._zero_grads(first_in_recording=k == 0)
                        self._static_out = loss_of(batches[k], marched)
                        nxt = None
                        if k + 1 < K:
                            # (where this fork sits does not matter to the step: at its start, here, or between the head's backward
                            # and the table scatter -- 0.599 / 0.599 / 0.600 ms, profiles/r06_teacher_launch_diet.txt)
                            branch.wait_stream(main)
                            with torch.cuda.stream(branch):
                                nxt = march(batches[k + 1])
                        self._backward(self._static_out[0])
                        if os.environ.get("PVD_TEST_FAIL_IN_CAPTURE") in ("1", "forked") and k == 1:  # exercises the fall-back
                            raise RuntimeError("forced failure inside the forked teacher block (PVD_TEST_FAIL_IN_CAPTURE)")
                        self._exchange()
                        self._optimize()
                        if nxt is not None:
                            main.wait_stream(branch)
                            marched = nxt
                except Exception:
                    main.wait_stream(branch)  # a capture can only be ended with its forked work joined
                    raise
        finally:
            self.dp.capture = None
        self._cap = cap
        self.steps_per_replay = K
        self._captured_occ_epoch = None
        self.pipelined_block = True

    def capture_block(self, batches):
        """Capture `update_extra_interval` consecutive training steps (one per entry of `batches`: STATIC device tensors
        (rays_o, rays_d, gt_rgb, bg) that the caller refills in place) as one HIP graph.  The teacher's sample budget moves
        with every occupancy-grid update (mean_count, renderer.py:773-775); the captured steps therefore allocate a fixed
        number of sample rows (`fix_sample_alloc`) and read the budget rays are dropped against from device memory, so the
        graph survives the updates.  Call after at least one eager block (lazy initialisations, a measured mean_count)."""
        o, m = self.opt, self.model
        assert len(batches) == o.update_extra_interval == 16, "one batch per step of a block (the step counter has 16 slots)"
        assert m.cuda_ray and m.mean_count > 0 and self.global_step % o.update_extra_interval == 0
        m.fix_sample_alloc()
        self._block_batches = batches
        self.pipelined_block = False
        if not self.dp.enabled and pvd_forked_graphs_ok():
            step0, local0 = self.global_step, m.local_step
            try:
                self._capture_block_pipelined(batches)
            except Exception:
                # the same join-and-fall-back DistillTrainer.capture_step has: the 16 steps recorded back to back (nothing ran
                # during the failed recording, so the step counters are put back and no training step is lost or repeated)
                self._after_failed_capture()
                self.global_step, m.local_step = step0, local0
                self.pipelined_block = False
        if not self.pipelined_block:
            self.capture(self._block_body(batches), warmup=0, steps_per_graph=len(batches))
        self._block_alloc = m.sample_alloc

    def train_block(self):
        """The occupancy-grid update (eager: it sizes the next block's budget) followed by one replay = 16 training steps."""
        o, m = self.opt, self.model
        assert self.global_step % o.update_extra_interval == 0
        self._update_occupancy()
        aligned = m.mean_count + (128 - m.mean_count % 128)
        if getattr(m, "budget_exceeded", False) or aligned < 0.6 * m.sample_alloc:
            # the scene needs more rows than were captured (or far fewer: the padding rows cost time): capture again
            self.capture_block(self._block_batches)
        out = self.replay()
        m.local_step += self.steps_per_replay  # the replayed marches filled that many slots of the step counter
        return out[0].detach(), out[1]
