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
import gc
import math
import os

import torch
import torch.distributed as dist


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


def _tree_map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _tree_map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree_map(v, fn) for v in obj)
    return obj


def _tree_tensors(obj, out):
    _tree_map(obj, lambda t: out.append(t) or t)
    return out


class CarriedPrefix:
    """A static home for the results of DistillTrainer.prefetch across graph replays: the prefix recorded next to the LAST step
    of a multi-step graph feeds the FIRST step of the next replay.  Storage by storage (views of one buffer stay views of one
    buffer: sigma_l is column 0 of feature_sigma_color), same sizes / strides / offsets."""

    def __init__(self, pre):
        self._stores = {}  # data_ptr of a source storage -> flat uint8 tensor owning the static copy

        def home(t):
            st = t.untyped_storage()
            if st.data_ptr() not in self._stores:
                flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (st.nbytes(),), (1,))
                self._stores[st.data_ptr()] = flat.clone()
            dst = self._stores[st.data_ptr()].untyped_storage()
            return torch.empty(0, dtype=t.dtype, device=t.device).set_(dst, t.storage_offset(), t.size(), t.stride())
        self.pre = _tree_map(pre, home)
        self._layout = [(t.dtype, tuple(t.size()), tuple(t.stride()), t.storage_offset()) for t in _tree_tensors(pre, [])]

    def store(self, pre):
        """Copy a new prefix (same structure and layout) into the static home: one multi-tensor copy on the current stream."""
        new, old = _tree_tensors(pre, []), _tree_tensors(self.pre, [])
        assert [(t.dtype, tuple(t.size()), tuple(t.stride()), t.storage_offset()) for t in new] == self._layout, \
            "the prefix changed shape between steps of one capture"
        srcs, dsts, seen = [], [], set()
        for tn, to in zip(new, old):
            sn, so = tn.untyped_storage(), to.untyped_storage()
            if sn.data_ptr() in seen:
                continue
            seen.add(sn.data_ptr())
            assert sn.nbytes() == so.nbytes()
            srcs.append(torch.empty(0, dtype=torch.uint8, device=tn.device).set_(sn, 0, (sn.nbytes(),), (1,)))
            dsts.append(torch.empty(0, dtype=torch.uint8, device=to.device).set_(so, 0, (so.nbytes(),), (1,)))
        torch._foreach_copy_(dsts, srcs)


class SegmentedCapture:
    """A step as a chain of HIP graphs with eager host calls between them.  `break_for(fn)` ends the graph being
    captured, runs fn() eagerly (and remembers it), and starts the next graph in the same memory pool; `replay()` replays
    graph 0, calls fn 0, replays graph 1, ...  Used to keep collectives out of the graphs."""

    def __init__(self, device):
        self.device = device
        self.graphs, self.between = [], []
        self.pool = None
        self.active = False
        self._stream = torch.cuda.Stream(device)

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        # thread_local: a communication library's watchdog thread may touch the device while we capture
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()  # one private pool for all segments: tensors live across them
        g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self.graphs.append(g)
        self.active = True

    def _end(self):
        self.active = False
        self.graphs[-1].capture_end()

    def __enter__(self):
        # as torch.cuda.graph.__enter__ does: collect garbage and return cached blocks BEFORE the capture begins, and keep the
        # cyclic collector off while it is under way -- an earlier trainer's graphs (reference cycles: collected whenever the
        # collector happens to run) would otherwise be destroyed, and their private pools released, in the middle of this
        # capture.  PVD_CAPTURE_GC=0 restores the old behaviour (tools/flake_hunt.sh).
        self._gc_was_enabled = None
        torch.cuda.synchronize()
        if os.environ.get("PVD_CAPTURE_GC", "1") != "0":
            gc.collect()
            torch.cuda.empty_cache()
            self._gc_was_enabled = gc.isenabled()
            gc.disable()
        self._stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self._stream)
        self._ctx.__enter__()
        try:
            self._begin()
        except BaseException:
            self._ctx.__exit__(None, None, None)
            self._restore_gc()
            raise
        return self

    def _restore_gc(self):
        if self._gc_was_enabled:
            gc.enable()
        self._gc_was_enabled = None

    def __exit__(self, exc_type, exc, tb):
        try:
            if self.active:
                self._end()
        finally:
            self._ctx.__exit__(exc_type, exc, tb)
            self._restore_gc()
        torch.cuda.current_stream().wait_stream(self._stream)
        return False

    def break_for(self, fn, replay_fn=None):
        """fn runs now (between two captures); replay_fn (default: fn) is what runs at that point of every replay."""
        self._end()
        fn()
        self.between.append(replay_fn or fn)
        self._begin()

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.between):
                self.between[i]()


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


    def _two_shot_sum_(self, t):
        """SUM over ranks as reduce-scatter + all-gather built from ALL-TO-ALL exchanges (PVD_DP_EXCHANGE=twoshot, opt-in): rank r
        receives chunk r of every peer, adds the n chunks in rank order, and sends the sum to every peer.  On a fully connected
        node every rank then talks to its n - 1 peers at once over its own links -- 2 (S / n) / one link's bandwidth instead of
        a ring's 2 (n - 1) / n S / (its slowest hop) (tools/scale_model.py) -- and every
        element is summed by exactly ONE rank, so the replicas receive identical bits by construction."""
        n = self.world_size
        flat = t.reshape(-1)
        chunk = (flat.numel() + n - 1) // n
        send = flat
        if chunk * n != flat.numel():
            send = torch.zeros(chunk * n, dtype=flat.dtype, device=flat.device)
            send[:flat.numel()].copy_(flat)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)  # recv[k * chunk : (k + 1) * chunk] = rank k's chunk `rank`
        red = recv.view(n, chunk)[0].clone()
        for k in range(1, n):  # rank order: the same sum whichever rank forms it
            red.add_(recv.view(n, chunk)[k])
        out = torch.empty_like(send)
        dist.all_to_all_single(out, red.repeat(n), group=self.group)  # out[k * chunk : ...] = the sum rank k formed
        flat.copy_(out[:flat.numel()])
        return t

    def all_reduce_sum_(self, t, overlap=None):
        """overlap: a callable launching device work that does not depend on the result (e.g. replaying the graph of the
        next step's parameter-independent prefix); in a captured step it is issued while the collective is in flight."""
        if self.enabled:
            # (verified on gloo only, tests/test_dist_gloo.py.  NOT recorded into a graph and not taken in a one-rank world: an attempt to
            # capture RCCL's all-to-all in a forced one-rank world did not return within ten minutes on the GPU box)
            two_shot = (os.environ.get("PVD_DP_EXCHANGE", "allreduce") == "twoshot" and self.world_size > 1
                        and not (self.capture is not None and self.capture.active and self.ingraph)
                        and t.numel() >= int(os.environ.get("PVD_DP_TWOSHOT_MIN", "65536")))  # (scalars and short buffers: one latency-bound all-reduce)
            run = (lambda: self._two_shot_sum_(t)) if two_shot else (lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
            if two_shot:
                overlap = None  # (its exchanges are issued synchronously: the overlap callable would only follow them)
            if self.capture is not None and self.capture.active and self.ingraph:
                run()  # recorded as a node of the graph being captured
            elif self.capture is not None and self.capture.active:
                replay = None
                if overlap is not None:
                    def replay():
                        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                        overlap()
                        work.wait()
                self.capture.break_for(run, replay)  # collectives stay out of the graphs: eager, between two replays
            else:
                run()
        return t

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

    def zero_(self):
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

    def zero_(self):
        self.opt.zero_grad()


def _make_loss(kind, dp):
    """reference: Trainer.get_loss, utils.py:941-952 (normL2 is a Frobenius norm, NOT a mean)."""
    if kind == "L2":
        return lambda pred, gt: dp.global_mean((gt.float() - pred.float()) ** 2)
    if kind == "normL2":
        return lambda pred, gt: dp.global_norm_l2(pred - gt)
    if kind == "normL1":
        return lambda pred, gt: dp.global_norm_l1(pred - gt)
    raise ValueError("error loss_type")


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
        if self.flat_opt and not self.dp.enabled and model.model_type == "hash":
            # the hash table's f16 scatter-add result goes straight into the update kernel (under ray-DP it is widened into
            # the fp32 buffer first, so that the exchange sees it)
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
        import os
        o, m = self.opt, self.model
        if not self.compact_exchange or os.environ.get("PVD_DP_COMPACT", "1") == "0" or m.model_type not in ("vm", "tensors"):
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

    def _zero_grads(self):
        """zero_grad.  With the flat optimizer and a frozen occupancy grid only the rows a sample can touch are ever
        written (the same set the compact exchange moves), so after one full clear only those are cleared and
        inf-checked (FlatAdamW.set_touched); PVD_TOUCHED_SET=0 turns that off."""
        if self.flat_opt:
            import os
            c = self._grad_compactor() if os.environ.get("PVD_TOUCHED_SET", "1") != "0" else None
            if c is not self.optimizer.touched:
                self.optimizer.set_touched(c)
        self.flat.zero_()

    def _step_prologue(self):
        """zero_grad (+ the student's weight image) at the start of a step.  PVD_PROLOGUE_FORK=1: on a side stream, i.e. as a
        parallel branch of the captured step that joins after the march (compute_loss) -- these launches are a few us of
        latency each and depend on nothing the marcher does.  Measured: 0.411 vs 0.394 ms/step -- a fork / join pair inside a
        hipGraph costs more (~17 us) than the ~11 us of launches it hides; off by default."""
        fh = getattr(getattr(self, "model_stu", None), "ops", None)
        fh = getattr(fh, "fused_head", None)
        if os.environ.get("PVD_PROLOGUE_FORK", "0") != "1" or fh is None or not hasattr(fh, "prepack_train_image") or self.dp.enabled:
            self._zero_grads()
            return
        main = torch.cuda.current_stream()
        side = getattr(self, "_prologue_stream", None)
        if side is None:
            side = self._prologue_stream = torch.cuda.Stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._zero_grads()
            fh.prepack_train_image(self.model_stu)
        self._prologue_join = lambda: main.wait_stream(side)

    def _fork_prefix(self, launch):
        """Single GPU, pipelined capture: launch the NEXT step's parameter-independent prefix (its own graph) on a side
        stream, next to this step's inf check / AdamW on the main stream; joined before the next step starts."""
        side = self._pipe_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            launch()
        self._pipe_pending = True

    def _exchange(self):
        ov = getattr(self, "_overlap_with_exchange", None)
        if not self.dp.enabled and ov is not None and self.dp.capture is not None and self.dp.capture.active:
            self.dp.capture.break_for(lambda: None, lambda: self._fork_prefix(ov))  # cut the graph here: fork point
            return
        if self.dp.enabled:
            c = self._grad_compactor()
            ov = getattr(self, "_overlap_with_exchange", None)
            # PVD_DP_WIRE=f16 | bf16 (opt-in, NOT the reference's arithmetic): the gradient crosses the links in 16 bits -- half the
            # bytes of the one exchange that bounds the multi-GPU step.  f16 relies on the loss scale (AMP): an
            # overflow on the wire is an inf in the gradient, which the scaler's check (it runs after the exchange) answers by
            # skipping the step and halving the scale, as for any other overflow; bf16 keeps fp32's range at 8 bits of mantissa.
            wire = {"f16": torch.float16, "bf16": torch.bfloat16}.get(os.environ.get("PVD_DP_WIRE", "f32"))
            if c is None:
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
                c.scatter(self.flat.flat, buf)

    def _optimize(self):
        self.scaler.step(self.optimizer)
        self.scaler.update()

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
                for _ in range(max(1, int(steps_per_graph))):
                    self._step_prologue()
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
            self.optimizer.zero_in_step = bool(on) and os.environ.get("PVD_ADAMW_ZERO_IN_STEP", "1") != "0"
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
        self.__dict__.pop("_before_objective", None)
        self._fold_launches(False)
        self._graph_zeroes = False
        if self.flat_opt:
            self.optimizer._half_grad = None  # a half-precision table gradient handed over by a backward whose update never came
            self.optimizer._part_a_owed = None  # (a two-part update recorded half way: nothing of it ran)
            self.optimizer.end_two_part(failed=True)
        for mdl in (getattr(self, "model_stu", None), self.model):
            if mdl is not None and getattr(mdl, "_between_backwards", None) is not None:
                mdl._between_backwards = None

    def replay(self):
        assert getattr(self, "_captured_occ_epoch", None) in (None, self._marching_model().occ_epoch), \
            "the occupancy grid changed since the step was captured (touched-row set is stale): capture again"
        if self.flat_opt:
            import pvd_hip
            pvd_hip.note_weights_changed(self.optimizer.params)  # the captured optimizer kernel rewrites the parameters
        if getattr(self, "_pipe_pending", False):  # the prefix forked during the previous step feeds this one
            torch.cuda.current_stream().wait_stream(self._pipe_stream)
            self._pipe_pending = False
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
        import os
        # measured on MI355X: 0.565 ms/step with the two forwards as parallel graph branches vs 0.545 ms in sequence
        # (both forwards already occupy every CU; the fork/join costs more than the overlap gains) -> off by default
        self.overlap_teacher = self.device_type == "cuda" and os.environ.get("PVD_OVERLAP_TEACHER", "0") == "1"
        self._side = torch.cuda.Stream(self.device) if self.overlap_teacher else None

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
            # (ObjectiveRide: two launches fewer on the chain; PVD_OBJECTIVE_RIDE=0 keeps the separate launches)
            ride = None
            if (self.fused_loss is not None and o.loss_type == "normL2" and not self.dp.enabled and torch.is_grad_enabled()
                    and self._stage_of(self.global_step) == 3 and out_tea.get("image") is not None
                    and min(o.loss_rate_color, o.loss_rate_sigma, self.loss_rate_fea_sc * 0.995, o.loss_rate_rgb) > 0.0
                    and torch.is_tensor(getattr(tea, "feature_sigma_color", None)) and torch.is_tensor(getattr(tea, "color_l", None))
                    and tea.feature_sigma_color.dim() == 2 and tea.feature_sigma_color.shape[-1] == 16
                    and tea.feature_sigma_color.dtype == torch.float32 and pre["rays_o"].is_cuda
                    and os.environ.get("PVD_OBJECTIVE_RIDE", "1") != "0" and os.environ.get("PVD_LOSS_DEFER", "0") != "1"):
                from .losses import ObjectiveRide
                # PVD_OBJECTIVE_FINISH=0: k_loss_final stays a launch of its own between the passes
                # (and never when the L1 value is added to the loss by the host afterwards -- L1 on, VM student, non-flat
                # optimizer: the finished loss only exists after the backward launch, as with PVD_LOSS_DEFER)
                l1_on_host = o.l1_reg_weight > 0.0 and o.model_type == "vm" and not self.flat_opt
                fin = os.environ.get("PVD_OBJECTIVE_FINISH", "1") != "0" and not l1_on_host
                ride = ObjectiveRide(out_tea["image"], tea.feature_sigma_color, tea.color_l, rates_decay=self.rates if fin else None, fea_decay=0.995)
            out_stu = stu.render(pre["rays_o"], pre["rays_d"], staged=False, bg_color=pre["bg"], perturb=True, force_all_rays=False,
                                 inherited_params=pre["inh"], nears_fars=pre["nf"], premarched=True, objective=ride, **kw)
            self._ride = ride if (ride is not None and ride.S is not None) else None
        elif self.overlap_teacher and rays_o.is_cuda and stu.cuda_ray and bool(getattr(o, "render_stu_first", True)):
            # march once, then the frozen teacher's forward runs on a side stream next to the student's forward
            # (they share only the samples); in a captured step this becomes two parallel branches of the graph
            inh, nf = stu.march(rays_o, rays_d, perturb=True, force_all_rays=False, **kw)
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side), torch.no_grad():
                out_tea = tea.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                                     inherited_params=inh, nears_fars=nf, premarched=True, **kw)
            out_stu = stu.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                                 inherited_params=inh, nears_fars=nf, premarched=True, **kw)
            main.wait_stream(self._side)
        else:
            join = self.__dict__.pop("_prologue_join", None)
            out_tea = None
            if join is not None and stu.cuda_ray and bool(getattr(o, "render_stu_first", True)):
                # the step's prologue (zero_grad, weight image) was forked onto a side stream: march first, join, then the forward
                inh, nf = stu.march(rays_o, rays_d, perturb=True, force_all_rays=False, **kw_stu)
                join()
                out_stu = stu.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                                     inherited_params=inh, nears_fars=nf, premarched=True, **kw)
            else:
                if join is not None:
                    join()
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

        wait_l1 = self.__dict__.pop("_before_objective", None)
        if wait_l1 is not None:
            wait_l1()  # a deferred part of the previous step's update refreshes the L1 term's partial sums (see _capture_ingraph_pipelined)
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
            # PVD_LOSS_DEFER=1: nothing looks at the value of the objective before loss.backward() in a training step, so it can be
            # finished inside the backward launch (pvd_distill_loss_backward) instead of by a launch of its own.  Bit-identical
            # (tests/test_hip_fused_misc.py) and measured: 0.3945 vs 0.3930 ms/step -- the single-workgroup launch it removes
            # was hidden behind its neighbours in the replayed graph; off by default.
            defer = not add_l1 and torch.is_grad_enabled() and os.environ.get("PVD_LOSS_DEFER", "0") == "1"
            ride = self.__dict__.pop("_ride", None)
            rkw = {} if (ride is None or defer) else {"ride": ride}
            l4, norms = self.fused_loss(pred_stu, pred_tea, stu.feature_sigma_color, tea.feature_sigma_color, stu.color_l.float(),
                                        tea.color_l.float(), self.rates, self.dp, fea_decay=0.995, extra=extra, defer=defer, **rkw)
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
        # the 3 eager warm-up steps inside capture() / _capture_pipelined() are real steps: they advance global_step, so the
        # stage that gets captured is the one AFTER them, and they must not straddle a stage boundary (the warm-up and the
        # captured step would then build different losses)
        warm = 3
        if self.flat_opt:
            self.optimizer.flush()  # (a part of the previous recording's last update may still be owed: FlatAdamW.carry_last)
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
        elif not self.dp.enabled and stu_marches and stage == 3 and os.environ.get("PVD_PIPELINE", "0") == "1":
            self._pipe_stream = torch.cuda.Stream()
            out = self._capture_pipelined(batch_fn, body)  # fork point instead of a collective (see _exchange)
        elif (not self.dp.enabled and stage == 3 and steps_per_graph > 1 and os.environ.get("PVD_PIPELINE_INGRAPH", "1") != "0"
              and stu_marches and pvd_forked_graphs_ok()):
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
        stream next to step k's table scatter, gradient exchange and update, and joined before step k + 1's student forward --
        inside one graph: no extra graph launches, one fork / join pair per step.  The scatter waits on the memory side, the
        update on HBM and the exchange on xGMI while the prefix is instruction- and L2-bound, so the two chains share the chip
        (0.379 -> 0.327 ms/step on one GPU).  Same batches in the same order, same update rule as the
        back-to-back recording (tests/test_hip_graph.py, tests/test_hip_dp_graph.py)."""
        assert self.device_type == "cuda" and (not self.dp.enabled or self.dp.ingraph)
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
        K = max(1, int(steps_per_graph))
        # the prefix next to the LAST step of the graph feeds the FIRST step of the next replay through a static home, so that
        # every step has its prefix overlapped, whatever the number of steps per graph (K >= 2: with one step per graph the
        # step's own scatter would still be reading the samples the copy overwrites)
        carried = None
        # (default "start": 0.322 vs 0.330 ms/step at 20 steps per graph, level at 5, profiles/r03_fork_modes.txt)
        fork_mode = os.environ.get("PVD_PIPELINE_FORK", "start")
        per_graph = fork_mode == "graph" and K >= 2 and os.environ.get("PVD_PIPELINE_CARRY", "1") != "0"
        self._graph_zeroes = False
        if K >= 2 and not self.dp.enabled:
            self._fold_launches(True)
        if per_graph:
            # ONE fork / join pair per GRAPH instead of one per step (a pair costs the main chain ~10 us at the fork and ~9 us at
            # the join, profiles/r03_step_timeline.txt): the branch records the prefixes of all K steps of the NEXT replay back to
            # back, the main chain records the K steps of this replay on the prefixes carried over from the previous one; after
            # the join the K new prefixes move into the static homes (one multi-tensor copy per prefix).
            with torch.cuda.stream(side):
                homes = [CarriedPrefix(self.prefetch(batch_fn)) for _ in range(K)]  # prologue: the first replay's prefixes, eagerly
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            cap = SegmentedCapture(self.device)
            self.dp.capture = cap
            branch = torch.cuda.Stream(self.device)
            try:
                with cap:
                    main = torch.cuda.current_stream()
                    branch.wait_stream(main)
                    try:
                        with torch.cuda.stream(branch):
                            nxts = [self.prefetch(batch_fn) for _ in range(K)]
                        for k in range(K):
                            self._zero_grads()
                            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                                self._static_out = self.compute_loss(None, None, None, pre=homes[k].pre)
                            self._backward(self._static_out[0])
                            if os.environ.get("PVD_TEST_FAIL_IN_CAPTURE") in ("1", "forked") and k == 1:  # exercises the fall-backs
                                raise RuntimeError("forced failure inside the forked capture (PVD_TEST_FAIL_IN_CAPTURE)")
                            self._exchange()
                            self._optimize()
                    finally:
                        main.wait_stream(branch)  # a capture can only be ended with its forked work joined
                    for k in range(K):
                        homes[k].store(nxts[k])
            finally:
                self.dp.capture = None
                self._fold_launches(False)
            self._cap = cap
            self.steps_per_replay = K
            self._captured_occ_epoch = self._marching_model().occ_epoch if (self.flat_opt and self.optimizer.touched is not None) else None
            self.pipelined_ingraph = True
            self.pipeline_fork = "graph"
            return self._static_out
        if fork_mode == "deep" and K >= 3 and os.environ.get("PVD_PIPELINE_CARRY", "1") != "0":
            return self._capture_deep(batch_fn, K, side)
        if K >= 2 and os.environ.get("PVD_PIPELINE_CARRY", "1") != "0":
            with torch.cuda.stream(side):
                carried = CarriedPrefix(self.prefetch(batch_fn))  # prologue: the first replayed step's prefix, eagerly
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        # Two-part AdamW (FlatAdamW.two_part; single GPU, fork at "start"): the update behind the table scatter covers only what the
        # backward can have written (touched rows, the heads); the L1-only / still-decaying rows -- half of the update's bytes, read by
        # nothing but their own update -- are updated one step later on the forked branch, with the scalars the step recorded.
        # Bit-identical parameters and moments (tests/test_hip_graph.py, tests/test_hip_fused_misc.py).  WHERE on the branch decides:
        #   "1"    first thing, next to the student's forward; the objective waits for it (exact L1 value).  Slower than one launch:
        #          0.340 vs 0.325 ms/step (profiles/r03_adamw_split_ab.txt) -- a 90 MB stream on top of the forward's gathers.
        #   "late" (default) at the END of the branch, next to the head backward and the table scatter, which wait on the matrix cores
        #          and on the memory side's atomic units, not on HBM: 0.280 vs 0.296 ms/step (profiles/r04_adamw_late_ab.txt).
        #   "0"    one launch.
        split_mode = os.environ.get("PVD_ADAMW_SPLIT", "late")
        # (ray-DP with the collectives in the graph: the deferred rows are the ones no sample reaches -- they are not part of the
        # exchange, and their update needs nothing but the step's scalars, which are the same on every rank -- so "late" holds there
        # too, and the deferred part then runs under the exchange)
        split = (self.flat_opt and fork_mode == "start" and K >= 2 and split_mode in ("1", "late")
                 and (not self.dp.enabled or (self.dp.ingraph and split_mode == "late")))
        # "late": part A of step k - 1 is launched at the END of step k's branch (after the next prefix: next to step k's head backward
        # and table scatter, which wait on the matrix cores and on the memory side) instead of at its start; the objective does not
        # wait for it -- the L1 VALUE it reports then counts the L1-only rows one step late (parameters and gradients are unaffected:
        # those rows are read by nothing but their own update)
        late = split and split_mode == "late"
        # ... and the LAST step's part A rides on the next replay's first branch instead of trailing the graph (FlatAdamW.carry_last)
        carry_a = late and carried is not None and os.environ.get("PVD_ADAMW_CARRY", "1") != "0"
        scaled = bool(getattr(self.scaler, "_enabled", False))
        self._replay_owes_part_a = None
        fh = getattr(getattr(self.model_stu, "ops", None), "fused_head", None)
        pack_ahead = getattr(fh, "prepack_train_image", None) if os.environ.get("PVD_PACK_ON_BRANCH", "1") != "0" else None
        # (measurement, bench.py) record_fused_spans: every recorded launch of the frozen hash teacher's lookup + head writes its own
        # extent {first workgroup's start, last one's end} into a row of fused_spans [K, 2] (pvd_hash_head_forward_fused_span)
        self.fused_spans = None
        if getattr(self, "record_fused_spans", False) and getattr(self.model_tea, "model_type", "") == "hash":
            import pvd_hip
            self.fused_spans = torch.tensor([list(pvd_hip.FUSED_SPAN_INIT)] * K, dtype=torch.int64, device=self.device)
        if split and not self.optimizer.begin_two_part(defer=True):
            split = late = carry_a = False  # (no warm-group lists, or no rows of one of the two kinds: one launch)
        if split:
            self.optimizer.carry_last = carry_a
        cap = SegmentedCapture(self.device)
        self.dp.capture = cap
        branch = torch.cuda.Stream(self.device)
        try:
            with cap:
                main = torch.cuda.current_stream()
                pre = carried.pre if carried is not None else self.prefetch(batch_fn)
                try:
                    for k in range(K):
                        more = k + 1 < K or carried is not None  # a prefix to record next to this step
                        # where the next prefix branches off: "mid" = between the student's head backward and its table scatter
                        # (a VM student; anything else: as "backward"), "backward" = before this step's backward, "optimizer" =
                        # before its exchange + update.  (Not before compute_loss: it reads tea.feature_sigma_color, which the
                        # prefix rebinds.)
                        # "start" = before this step's forward: compute_loss re-installs the teacher outputs of ITS prefix first
                        # thing, so the rebinding is harmless once the fork's host code has run before it.
                        # "headbwd" = right before the student's head backward (after the objective's and the compositing backward).
                        # "start2" = two stages: the batch and the march (instruction-bound) fork at the start, next to the student's
                        # forward; the frozen teacher's forward (gathers) is held back until the student's head backward begins, so
                        # that it does not sit on top of the short latency-bound launches of the objective in between.
                        two_stage = fork_mode == "start2"
                        fork_at = "start" if two_stage else (fork_mode if fork_mode in ("mid", "backward", "optimizer", "start", "headbwd") else "start")
                        pre_next = None

                        def fork(k=k):
                            branch.wait_stream(main)
                            if self.fused_spans is not None:  # (measurement) this step's teacher launch leaves its extent in row k
                                self.model_tea._fused_span = self.fused_spans[k]
                            with torch.cuda.stream(branch):
                                if fork_at == "start" and pack_ahead is not None and pack_ahead(self.model_stu):
                                    # the student's f16 weight image for THIS step's head (5 us on the main chain otherwise): the
                                    # weights are final (the fork follows the previous update), the head waits for it
                                    packed = torch.cuda.Event()
                                    packed.record(branch)
                                    self.model_stu._before_head = lambda: main.wait_event(packed)
                                if split and not late and self.optimizer.run_part_a():  # what the previous step's update still owes
                                    done = torch.cuda.Event()
                                    done.record(branch)
                                    # (the objective adds up the L1 term from the partial sums this launch refreshes)
                                    self._before_objective = lambda: main.wait_event(done)
                                if two_stage:
                                    return self.prefetch_march(batch_fn)  # stage 1; stage 2 (teacher) follows from the hook below
                                nxt = self.prefetch(batch_fn)
                                if k + 1 == K:  # for the next replay
                                    carried.store(nxt)
                                if late:
                                    self.optimizer.run_part_a()
                                    if carry_a and k == 0:
                                        self.optimizer.run_carried_part_a(scaled)
                                return nxt

                        def teacher_stage(part, k=k):
                            branch.wait_stream(main)  # not before the main chain has come this far
                            with torch.cuda.stream(branch):
                                nxt = self.prefetch_teacher(part)
                                if k + 1 == K:
                                    carried.store(nxt)
                                return nxt
                        held = {}
                        if more and fork_at == "mid":  # between the student's head backward and its table scatter

                            def between(grad, held=held):
                                if "pre" not in held:
                                    held["pre"] = fork()
                                return None
                            self.model_stu._between_backwards = between
                        if more and fork_at == "headbwd":

                            def before_head(grad, held=held):
                                if "pre" not in held:
                                    held["pre"] = fork()
                                return None
                            self.model_stu._before_head_backward = before_head
                        if more and two_stage:
                            held["part"] = fork()

                            def before_head2(grad, held=held):
                                if "pre" not in held:
                                    held["pre"] = teacher_stage(held["part"])
                                return None
                            self.model_stu._before_head_backward = before_head2
                        elif more and fork_at == "start":
                            pre_next = fork()
                        elif split:
                            self.optimizer.run_part_a()  # no branch to put it on
                        self._zero_grads()
                        try:
                            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                                self._static_out = self.compute_loss(None, None, None, pre=pre)
                        finally:
                            if fork_at == "mid" and getattr(self.model_stu, "_between_backwards", None) is not None:
                                fork_at = "backward"  # the forward did not take the hook (not a fused VM student)
                            if fork_at == "headbwd" and getattr(self.model_stu, "_before_head_backward", None) is not None:
                                fork_at = "backward"
                            if two_stage and getattr(self.model_stu, "_before_head_backward", None) is not None:
                                held["pre"] = teacher_stage(held["part"])  # hook not taken: the teacher stage before the backward
                            self.model_stu._between_backwards = None
                            self.model_stu._before_head_backward = None
                        if more and fork_at == "backward":  # the next step's prefix depends on nothing this step computes
                            pre_next = fork()
                        self._backward(self._static_out[0])
                        if os.environ.get("PVD_TEST_FAIL_IN_CAPTURE") in ("1", "forked") and k == 1:  # exercises the fall-backs
                            raise RuntimeError("forced failure inside the forked capture (PVD_TEST_FAIL_IN_CAPTURE)")
                        if more and two_stage:
                            pre_next = held.get("pre")
                            if pre_next is None:  # the forward did not take the hook (not a fused VM student): stage 2 now
                                pre_next = teacher_stage(held["part"])
                        if more and fork_at in ("mid", "headbwd"):
                            pre_next = held.get("pre")
                        if more and pre_next is None:  # "optimizer", or a student without the hook point
                            pre_next = fork()
                        self._exchange()
                        if carry_a and k + 1 == K:
                            self.optimizer._next_step_is_last = True
                        self._optimize()
                        if pre_next is not None:  # join
                            main.wait_stream(branch)
                            pre = pre_next
                    if split and carry_a:
                        self.optimizer._part_a_owed = None  # the last step's: recorded on the next replay's first branch
                        self._replay_owes_part_a = scaled
                    elif split:
                        self.optimizer.run_part_a()  # the last step's: a replay leaves nothing owed
                except Exception:
                    self.model_stu._between_backwards = None
                    self.model_stu._before_head_backward = None
                    self.model_stu.__dict__.pop("_before_head", None)
                    self.model_stu.__dict__.pop("_train_image_ready", None)
                    self.__dict__.pop("_before_objective", None)
                    main.wait_stream(branch)  # a capture can only be ended with its forked work joined
                    raise
        finally:
            self.dp.capture = None
            self._fold_launches(False)
            if split:
                self.optimizer.end_two_part()
        self._cap = cap
        self.steps_per_replay = K
        self._captured_occ_epoch = self._marching_model().occ_epoch if (self.flat_opt and self.optimizer.touched is not None) else None
        self.pipelined_ingraph = True
        self.pipeline_fork = fork_mode
        # how the recorded steps update: "0" one launch, "1" / "late" two parts (only if the optimizer did split: a student without
        # L1-only rows keeps the single launch)
        self.adamw_split = split_mode if (split and getattr(self.optimizer, "_graph_is_two_part", False)) else "0"
        return self._static_out

    def _capture_deep(self, batch_fn, K, side):
        """PVD_PIPELINE_FORK=deep: the fork at the start of the step, the branch TWO steps deep -- first the frozen teacher's forward
        on the samples of step k + 1 (marched during step k - 1's branch: the gathers then run next to the student's
        instruction-bound forward instead of on top of its table scatter), then the batch and the march of step k + 2
        (instruction-bound, next to the atomics-bound scatter).  No additional edge in the graph: still one fork and one join per
        step.  Two static homes carry the state across replays (the complete prefix of the next replay's step 1, the marched
        samples of its step 2); K >= 3 so that nobody still reads a home when the last step's branch refills it."""
        with torch.cuda.stream(side):
            full_home = CarriedPrefix(self.prefetch(batch_fn))
            part_home = CarriedPrefix(self.prefetch_march(batch_fn))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        fh = getattr(getattr(self.model_stu, "ops", None), "fused_head", None)
        pack_ahead = getattr(fh, "prepack_train_image", None) if os.environ.get("PVD_PACK_ON_BRANCH", "1") != "0" else None
        # (measurement, bench.py) record_fused_spans: every recorded launch of the frozen hash teacher's lookup + head writes its own
        # extent {first workgroup's start, last one's end} into a row of fused_spans [K, 2] (pvd_hash_head_forward_fused_span)
        self.fused_spans = None
        if getattr(self, "record_fused_spans", False) and getattr(self.model_tea, "model_type", "") == "hash":
            import pvd_hip
            self.fused_spans = torch.tensor([list(pvd_hip.FUSED_SPAN_INIT)] * K, dtype=torch.int64, device=self.device)
        cap = SegmentedCapture(self.device)
        self.dp.capture = cap
        branch = torch.cuda.Stream(self.device)
        try:
            with cap:
                main = torch.cuda.current_stream()
                pre, part = full_home.pre, part_home.pre
                try:
                    for k in range(K):
                        branch.wait_stream(main)
                        with torch.cuda.stream(branch):
                            if pack_ahead is not None and pack_ahead(self.model_stu):
                                packed = torch.cuda.Event()
                                packed.record(branch)
                                self.model_stu._before_head = lambda packed=packed: main.wait_event(packed)
                            nxt_full = self.prefetch_teacher(part)
                            nxt_part = self.prefetch_march(batch_fn)
                            if k + 1 == K:  # for the next replay
                                full_home.store(nxt_full)
                                part_home.store(nxt_part)
                        self._zero_grads()
                        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16):
                            self._static_out = self.compute_loss(None, None, None, pre=pre)
                        self._backward(self._static_out[0])
                        if os.environ.get("PVD_TEST_FAIL_IN_CAPTURE") in ("1", "forked") and k == 1:  # exercises the fall-backs
                            raise RuntimeError("forced failure inside the forked capture (PVD_TEST_FAIL_IN_CAPTURE)")
                        self._exchange()
                        self._optimize()
                        main.wait_stream(branch)
                        pre, part = nxt_full, nxt_part
                except Exception:
                    self.model_stu.__dict__.pop("_before_head", None)
                    self.model_stu.__dict__.pop("_train_image_ready", None)
                    main.wait_stream(branch)  # a capture can only be ended with its forked work joined
                    raise
        finally:
            self.dp.capture = None
            self._fold_launches(False)
        self._cap = cap
        self.steps_per_replay = K
        self._captured_occ_epoch = self._marching_model().occ_epoch if (self.flat_opt and self.optimizer.touched is not None) else None
        self.pipelined_ingraph = True
        self.pipeline_fork = "deep"
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

    def train_step(self, rays_o, rays_d, gt_rgb, bg_color):
        o, m = self.opt, self.model
        if m.cuda_ray and self.global_step % o.update_extra_interval == 0:
            with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
                m.update_extra_state()
        self._zero_grads()
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
            out = m.render(rays_o, rays_d, staged=False, bg_color=bg_color, perturb=True, force_all_rays=False,
                           dt_gamma=o.dt_gamma, max_steps=o.max_steps, num_steps=o.num_steps, upsample_steps=o.upsample_steps)
            pred = out["image"]
            loss = self.dp.global_mean((pred.float() - gt_rgb.float()) ** 2)
            if o.l1_reg_weight > 0.0 and o.model_type == "vm":
                loss = loss + self._l1_term()
        self._backward_and_step(loss)
        return loss.detach(), pred

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
                loss = self.dp.global_mean((pred.float() - gt_rgb.float()) ** 2)
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
                loss = self.dp.global_mean((pred.float() - b[2].float()) ** 2)
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
                        self._zero_grads()
                        self._static_out = loss_of(batches[k], marched)
                        nxt = None
                        if k + 1 < K:
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
        if not self.dp.enabled and os.environ.get("PVD_TEACHER_PIPELINE", "1") != "0" and pvd_forked_graphs_ok():
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
        with torch.autocast(self.device_type, dtype=torch.float16, enabled=self.fp16 and self.device_type == "cuda"):
            m.update_extra_state()
        aligned = m.mean_count + (128 - m.mean_count % 128)
        if getattr(m, "budget_exceeded", False) or aligned < 0.6 * m.sample_alloc:
            # the scene needs more rows than were captured (or far fewer: the padding rows cost time): capture again
            self.capture_block(self._block_batches)
        out = self.replay()
        m.local_step += self.steps_per_replay  # the replayed marches filled that many slots of the step counter
        return out[0].detach(), out[1]
