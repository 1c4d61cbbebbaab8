"""AdamW on flat buffers: every parameter (and its gradient and both moments) is a strided view into one fp32
buffer per kind, so the update is one streaming HIP kernel (pvd_adamw_step) instead of a multi-tensor launch
per parameter group, the gradient exchange of ray-DP is one all-reduce, and zeroing the gradients one memset.
Drop-in for torch.optim.AdamW(betas, eps, weight_decay) as the reference constructs it
(main_distill_mutual.py:334-339), including GradScaler's unscale / skip-on-inf protocol and tensor learning
rates (LR schedulers fill them in place), so a captured HIP graph sees the schedule."""
import os

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

    # ---- fewer launches around the update (pvd_adamw_extras.zero_grad_after / arrivals)
    zero_in_step = False     # set by the trainer while it records several steps into one graph: the update zeroes what it read, so
    _zeroed_by_step = False  # ... the NEXT step's zero_grad() has nothing to launch
    _arrivals = None

    def _tail_in_kernel(self):
        """The arrival counter of the in-kernel tail (the last workgroup to arrive does what a one-thread launch did)."""
        if self._arrivals is None:
            self._arrivals = torch.zeros(65 * 32, dtype=torch.int32, device=self.flat_p.device)
        return self._arrivals

    # ---- two-part update (pvd_adamw_extras.snapshot / replay): set by the trainer while it records a pipelined multi-step graph
    two_part = False      # step() updates part B (+ tail) and owes part A
    defer_part_a = False  # ... and leaves it to the caller to run it (run_part_a) where it overlaps latency-bound kernels
    _part_a_owed = None
    _snapshot = None

    @torch.no_grad()
    def run_part_a(self):
        """The owed second part of a two-part update (no-op when nothing is owed): the groups whose gradient is structurally zero,
        with the scalars of the step that owes them.  Launched on the CURRENT stream."""
        owed = self._part_a_owed
        if owed is None:
            return False
        self._part_a_owed = None
        self._launch_part_a(owed)
        if self._owed_is_carried:  # a recording's last step, run here instead of by the next replay: that replay must find it done
            self._owed_is_carried = False
            if not torch.cuda.is_current_stream_capturing():
                owed[6][0:1].fill_(1.0)  # (the record's "skipped" flag: the replayed launch returns at once)
        return True

    def _launch_part_a(self, owed):
        cold, log, count, warm_a, scaled, l1n, snap = owed
        d = self.defaults
        pvd_hip.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.segment_ends, self.lr_dev, d["betas"][0], d["betas"][1],
                           d["eps"], d["weight_decay"], self.step_count, snap[2:3] if scaled else None, None,
                           l1_ranges=getattr(self, "_l1", None), l1_next=l1n, cold_bits=cold, lazy=(log, count, warm_a), replay=snap)
        if torch.cuda.is_current_stream_capturing():
            self._graph_is_two_part = True  # (note_device_steps: what a replay leaves in the L1 partial sums)
        pvd_hip.note_weights_changed(self.params)

    # A recording's LAST step has no later step of the same graph to carry its part A: with `carry_last` its record goes to a
    # buffer of its own and the NEXT replay's first branch launches it (run_carried_part_a, recorded once per graph); between
    # replays the host knows it as owed (note_carried_part_a), so a flush / an eager step runs it and marks the record as done.
    carry_last = False
    _next_step_is_last = False
    _owed_is_carried = False
    _snap_last = None

    def _snap_buffers(self):
        if self._snap_last is None:
            dev, n = self.flat_p.device, 4 + len(self.segment_ends)
            self._snapshots = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(2)]
            self._snap_last = torch.zeros(n, dtype=torch.float32, device=dev)
            self._snap_last[0] = 1.0  # nothing owed yet
            self._snapshot = self._snapshots[0]

    def _part_a_record(self, scaled, snap):
        lazy, st = self._lazy_state(), getattr(self, "_l1_track", None)
        return (self._cold_bits, lazy[0], lazy[1], self._warm_A, bool(scaled), (st["buf"][4096:], st["scale"]) if st is not None else None, snap)

    @torch.no_grad()
    def run_carried_part_a(self, scaled):
        """(while recording, on the branch of a graph's FIRST step) part A of the previous replay's last step."""
        self._snap_buffers()
        self._launch_part_a(self._part_a_record(scaled, self._snap_last))

    def note_carried_part_a(self, scaled):
        """(after a replay of a graph recorded with carry_last) its last step's part A is owed."""
        self._part_a_owed = self._part_a_record(scaled, self._snap_last)
        self._owed_is_carried = True

    def _two_part_now(self, lazy):
        return bool(self.two_part and lazy is not None and len(lazy) >= 3 and getattr(self, "_half_grad", None) is None
                    and getattr(self, "_warm_A", None) is not None and self._warm_A.numel() > 0 and self._warm_B.numel() > 0)

    _l1_layout, _graph_is_two_part = "one", False

    def _l1_layout_is(self, layout, capturing=False):
        """The L1 term's partial sums (l1_partials) are one entry per workgroup of whatever launch shape wrote them last: when the
        shape changes (one launch <-> two parts) they are started afresh, as after a change of the warm list."""
        if layout == self._l1_layout:
            return
        assert not capturing, "the form of the update must be settled before a capture begins (FlatAdamW.begin_two_part)"
        self._l1_layout = layout
        st = getattr(self, "_l1_track", None)
        if st is not None:
            st["buf"].zero_()
            total = self.l1_value(st["scale"])
            if layout == "two" and getattr(self, "_warm_A", None) is not None and self._warm_A.numel() > 0:
                # the two regions start out with THEIR rows' sums: the first part B then replaces region one with its rows' new sums while
                # region two still describes the deferred rows as they are in memory (their update comes a step later)
                a = (self._warm_A.long()[:, None] * 4 + torch.arange(4, device=self._warm_A.device)).reshape(-1)
                a_val = torch.zeros((), dtype=torch.float32, device=a.device)
                for b, e, c in self._l1:
                    sel = a[(a >= b) & (a < e)]
                    if sel.numel():
                        a_val = a_val + float(c * st["scale"]) * self.flat_p[sel].abs().sum()
                st["buf"][4096] = a_val
                st["buf"][0] = total - a_val
            else:
                st["buf"][0] = total

    def begin_two_part(self, defer):
        """Called by the trainer BEFORE it records steps whose update is split (eagerly: may launch).  Returns whether the recorded
        steps WILL split: they need the warm-group lists (deferred decay of the cold groups) and rows of both kinds."""
        self.flush()
        if self._cold_dirty:  # the lists are built by the first step after the touched set changed; the recording needs them now
            self._cold_bits = self._build_cold_bits()
            self._cold_dirty = False
        will = bool(self.touched is not None and self._cold_bits is not None and self._lazy_state() is not None
                    and getattr(self, "_warm_A", None) is not None
                    and self._warm_A.numel() > 0 and self._warm_B.numel() > 0)
        if not will:
            return False
        self._snap_buffers()  # (allocated and initialised OUTSIDE the recording)
        self._graph_is_two_part = False  # (set again by the first part A this recording launches)
        self.two_part, self.defer_part_a = True, bool(defer)
        self._l1_layout_is("two")
        return True

    def end_two_part(self, failed=False):
        """After the recording: eager steps go back to the single launch (the recorded graph keeps the two parts).  failed: the
        recording raised -- no graph with two parts exists, and whatever is recorded next (the trainer's back-to-back fall-back)
        is a single launch whose L1 partial sums must be laid out as one region BEFORE that capture begins."""
        self.two_part = self.defer_part_a = self.carry_last = self._next_step_is_last = False
        if failed:
            self._graph_is_two_part = False
            self._owed_is_carried = False
            self._l1_layout_is("one")

    # ---- ray-DP (round 6): the exchanged compact gradient goes straight into part B of a two-part update (no pass that puts the
    # summed rows back into flat_g; the gather that filled the buffer zeroed the rows and looked at them: no zero_grad, no check)
    _compact = None

    def compact_ready(self):
        """May the coming step() take its gradient from a compact exchange buffer?  The update must walk the touched set as a LIST of its
        own: part B of the two-part form, or the single launch of a model whose warm groups ARE the touched set (no L1-only rows, no
        moments still decaying elsewhere: the Plenoxel student) -- and nothing else may be pending."""
        lazy = self._lazy_state() if (self.touched is not None and self._outside_is_zero and not self._cold_dirty and self._cold_bits is not None) else None
        if lazy is None or getattr(self, "_half_grad", None) is not None or getattr(self, "_warm_B", None) is None or self._warm_B.numel() == 0:
            return False
        return bool(self._two_part_now((lazy[0], lazy[1], None)) or (not self.two_part and self._warm_A.numel() == 0))

    def take_compact(self, grad, flag, clear, rows, param_out=None):
        """grad: f32, four floats per list entry of rows = (first, one past last) of the part-B list; flag: f32 [1] view = the
        step's found_inf (global); clear = (tensor, stride, count): flag words the update's tail zeroes; param_out: the updated rows'
        copy for a sharded update's all-gather."""
        self._compact = dict(grad=grad, flag=flag, clear=clear, rows=rows, param_out=param_out)
        self._checked_by_backward = True  # (FlatGradScaler.step: no check launch -- the gather looked at every value it moved)

    touched = None  # set_touched(): the only entries of flat_g anything ever writes (pvd/dp_compact.py), or None = all
    _outside_is_zero = False

    def set_touched(self, touched):
        """touched: an object with zero(flat) / check_finite(flat, flag) over a fixed index set, under the caller's
        guarantee that no gradient is ever written outside that set.  zero_grad and the scaler's inf check then walk
        only the set (after one full zero)."""
        self.flush()  # deferred decays belong to the OLD cold set
        self.touched = touched
        self._outside_is_zero = False
        self._cold_bits, self._cold_dirty = None, touched is not None

    # ---- lazy weight decay of the cold groups (pvd_adamw_extras.lazy_log): nothing reads a cold parameter while it is cold, so
    # its per-step decay is logged on the device and replayed -- same arithmetic, same bits -- when the values are needed
    LAZY_CAPACITY = 1 << 15
    _lazy, _lazy_logged = None, 0

    def _lazy_state(self):
        if os.environ.get("PVD_ADAMW_LAZY", "1") == "0":
            return None
        if self._lazy is None:
            dev = self.flat_p.device
            self._lazy = (torch.zeros(self.LAZY_CAPACITY, len(self.segment_ends), dtype=torch.float32, device=dev),
                          torch.zeros(1, dtype=torch.int32, device=dev))
        return self._lazy

    def before_replay(self):
        """A graph whose updates are split writes the L1 partial sums in two regions; entries an eager single-launch step wrote
        outside them would be added to every replayed step's L1 value.  Start the sums afresh when the shape changes."""
        if self._graph_is_two_part:
            self._l1_layout_is("two")
        if self._owed_is_carried and self._part_a_owed is not None:  # the graph's first branch runs it
            self._part_a_owed, self._owed_is_carried = None, False

    def note_device_steps(self, n):
        """n update steps ran on the device without step() being called (a graph replay): keep the host's idea of the log's
        fill level current, and empty the log well before it is full."""
        if self._graph_is_two_part:
            self._l1_layout = "two"  # the replayed steps wrote the partial sums in two regions
        if self._lazy is not None and self._cold_bits is not None:
            self._lazy_logged += int(n)
            if self._lazy_logged > self.LAZY_CAPACITY - 1024 and not torch.cuda.is_current_stream_capturing():
                self.flush()

    @torch.no_grad()
    def flush(self):
        """Apply the deferred decays: after this every parameter holds what per-step updates would have left.  Called before
        anything reads whole tables (state_dict, checkpoints, a change of the touched set) -- and by whoever compares
        parameters."""
        self.run_part_a()  # (a two-part update whose second part is still owed)
        if self._lazy is None or self._cold_bits is None or self._lazy_logged == 0:
            return
        status = torch.zeros(1, dtype=torch.int32, device=self.flat_p.device)
        pvd_hip.adamw_lazy_flush(self.flat_p, self.segment_ends, self._cold_bits, self._lazy[0], self._lazy[1],
                                 self.defaults["weight_decay"], status)
        n = int(status[0])
        assert n >= 0, "the lazy-decay log overflowed: decays were lost"
        self._lazy_logged = 0
        pvd_hip.note_weights_changed(self.params)

    def state_dict(self):
        self.flush()
        return super().state_dict()

    def _build_cold_bits(self):
        """One bit per group of 4 parameters: set where the group lies outside the touched set, outside every L1 range (the
        regulariser's gradient reaches every entry there) and has zero moments right now (they stay zero: no gradient will
        ever arrive).  The update kernel then applies the weight decay alone to those groups -- bit-identical to the full
        expression with g = m = v = 0 -- without reading or writing g, m, v."""
        n = self.flat_p.numel()
        n4 = n // 4
        dev = self.flat_p.device
        warm = torch.zeros(n4 + 1, dtype=torch.bool, device=dev)
        warm[self.touched.idx >> 2] = True
        for b, e, _ in (getattr(self, "_l1", None) or []):
            warm[b >> 2:(e + 3) >> 2] = True
        warm = warm[:n4]
        warm |= (self.flat_m[:n4 * 4].view(n4, 4) != 0).any(1) | (self.flat_v[:n4 * 4].view(n4, 4) != 0).any(1)
        cold = ~warm
        words = (n + 127) // 128
        bits = torch.zeros(words * 32, dtype=torch.int64, device=dev)
        bits[:n4] = cold.to(torch.int64)
        packed = (bits.view(words, 32) << torch.arange(32, device=dev)).sum(1)
        packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32).contiguous()
        self.cold_fraction = float(cold.float().mean()) if n4 else 0.0
        # the warm groups as a list (the update kernel walks it when the cold groups' decay is deferred): full wavefronts of work
        self._warm_groups = (~cold).nonzero().squeeze(1).to(torch.int32).contiguous()
        # the two parts of a two-part update (step(): `two_part`): B = warm groups the step's backward can write (touched rows,
        # the MLP heads -- everything in the touched set), A = the other warm groups (L1-only rows, rows whose moments are still
        # decaying): their gradient is structurally zero and nothing reads them before the next step's objective
        in_touched = torch.zeros(n4 + 1, dtype=torch.bool, device=dev)
        in_touched[self.touched.idx >> 2] = True
        self._warm_B = ((~cold) & in_touched[:n4]).nonzero().squeeze(1).to(torch.int32).contiguous()
        self._warm_A = ((~cold) & ~in_touched[:n4]).nonzero().squeeze(1).to(torch.int32).contiguous()
        st = getattr(self, "_l1_track", None)
        if st is not None:  # per-workgroup partial sums: the launch shape changes with the list, start them afresh
            st["buf"].zero_()
            st["buf"][0] = self.l1_value(st["scale"])
            self._l1_layout = "one"  # (a two-part recording re-bases the two regions: _l1_layout_is)
        return packed if self.cold_fraction > 0.05 else None

    # ---- the inf check done by the backward itself (pvd_head_dw_rider.found_inf: the VM student's table scatter + weight-gradient
    # reduction look at everything they complete).  inf_flag() is the scaler's flag, handed to that launch; the launch's caller reports
    # back through note_checked_by_backward(); FlatGradScaler.step then launches no check of its own for THIS step.
    _checked_by_backward = False

    def inf_flag(self):
        flag = getattr(self, "_found_inf_flag", None)
        if flag is None:
            flag = self._found_inf_flag = torch.zeros(1, dtype=torch.float32, device=self.flat_g.device)
        return flag

    def note_checked_by_backward(self):
        self._checked_by_backward = True

    _fp32_range_still_zero = None  # (lo, hi): the fp32 gradient range of a parameter whose gradient arrived in half precision, untouched since the last zero_grad

    def zero_grad(self, set_to_none=False, full=False):
        """full = True: zero the whole buffer whatever the last step is known to have left (the FIRST zero_grad of a recording: what
        the host knows then describes the last executed step, not the state every replay will start from)."""
        self._checked_by_backward = False  # (a backward before this zero_grad says nothing about the gradients to come)
        self._half_grad = None
        self._half_range_dirty = False
        self._compact = None
        still_zero, self._fp32_range_still_zero = self._fp32_range_still_zero, None
        if self._zeroed_by_step and self.touched is not None and self._outside_is_zero:
            self._zeroed_by_step = False  # the previous step's update zeroed every group it read: the touched set is clean
        elif self.touched is not None and self._outside_is_zero:
            self._zeroed_by_step = False
            self.touched.zero(self.flat_g)
        elif still_zero is not None and not full and self.touched is None:
            # the hash table's gradient went to the update in half precision (accept_half_grad) and nothing wrote its fp32 range since
            # the last zero_grad: 42 MB of zeros need no second fill -- only what lies outside the range (the heads) is cleared
            self._zeroed_by_step = False
            lo, hi = still_zero
            if lo > 0:
                self.flat_g[:lo].zero_()
            if hi < self.flat_g.numel():
                self.flat_g[hi:].zero_()
        else:
            self._zeroed_by_step = False
            self.flat_g.zero_()
            self._outside_is_zero = self.touched is not None
        self.reattach()

    def check_finite(self, flag):
        """flag[0] = 1 if a gradient is inf / nan (read-only; never cleared here)."""
        hg = getattr(self, "_half_grad", None)
        if self.touched is not None and self._outside_is_zero:
            self.touched.check_finite(self.flat_g, flag)
        elif hg is not None and not self._half_range_dirty and hg[2].numel() % 8 == 0 and hg[0] % 4 == 0 and hg[1] % 4 == 0 and self.flat_g.numel() % 4 == 0:
            # the fp32 range of the parameter whose gradient came in half precision holds the zeros of zero_grad (accept_half_grad:
            # nothing adds into it): one launch over the rest of the fp32 buffer and the half buffer
            pvd_hip.check_finite_mixed(self.flat_g, hg[0], hg[1], hg[2], flag)
            return
        else:
            pvd_hip.check_finite(self.flat_g, flag)
        if hg is not None:
            pvd_hip.check_finite_f16(hg[2], flag)

    def reattach(self):
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = _view(self.flat_g, p, o)

    # ---- pieces of the training loop folded into the update kernel (pvd_adamw_step_ex)
    def set_schedule(self, kind, T, param):
        """Evaluate the lr schedule on the device inside step(): kind "cosine" (CosineAnnealingLR closed form,
        T = T_max, param = eta_min) or "exp" (LambdaLR(param ** min(t / T, 1))).  Base rates = the current ones."""
        self.base_lr = self.lr_dev.clone()
        self.sched_step = torch.zeros(1, dtype=torch.float32, device=self.lr_dev.device)
        self._schedule = ({"cosine": 1, "exp": 2}[kind], float(T), float(param), self.base_lr, self.sched_step)

    _cold_bits, _cold_dirty = None, False
    _half_range_dirty = False  # the fp32 gradient range of the parameter whose gradient came in half precision has been written since zero_grad

    def set_l1(self, tensors, weight):
        self._cold_dirty = self.touched is not None  # the regularised ranges are never cold
        """Fold weight * sum_t mean|t| (NeRFNetwork.density_loss) into the update: its gradient weight/numel * sign(p)
        is added to the unscaled gradient inside the kernel; l1_value() returns the term's value."""
        off = {id(p): o for p, o in zip(self.params, self.offsets)}
        self._l1 = []
        for t in tensors:
            o, n = off[id(t)], t.numel()
            assert o % 4 == 0 and n % 4 == 0, "L1 ranges must be multiples of 4 elements"
            self._l1.append((o, o + n, weight / n))
        self._l1_scratch = torch.empty(1024, dtype=torch.float32, device=self.flat_p.device)

    @torch.no_grad()
    def l1_value(self, scale=1.0):
        out = torch.empty(1, dtype=torch.float32, device=self.flat_p.device)
        ranges = self._l1 if scale == 1.0 else [(b, e, c * scale) for b, e, c in self._l1]
        pvd_hip.l1_ranges(self.flat_p, ranges, self._l1_scratch, out)
        return out[0]

    def accept_half_grad(self, param, g16):
        """A half-precision gradient of `param` (the hash table's scatter-add result) for the coming step: added inside the
        update kernel instead of a widen-and-add pass over the fp32 gradient.  Returns False if it cannot be taken (then
        the caller adds it into .grad itself)."""
        if not getattr(self, "half_grad_ok", True):
            return False
        hg = getattr(self, "_half_grad", None)
        if hg is not None:
            # a second backward before the step: the caller adds this gradient into the fp32 range the mixed inf check skips
            # (check_finite), so that range is no longer zeros -- the full check runs for this step
            self._half_range_dirty = True
            return False
        for p, o in zip(self.params, self.offsets):
            if p is param:
                if o % 4 or p.numel() % 8 or not p.is_contiguous() or g16.shape != p.shape:
                    return False
                self._half_grad = (o, o + p.numel(), g16.reshape(-1))
                return True
        return False

    @torch.no_grad()
    def l1_partials(self, scale=1.0):
        """Partial sums of the L1 term's value (to be added by the consumer, e.g. the fused distillation objective).  After
        the first call the update kernel itself keeps them current (it reads every parameter anyway), so a step costs no
        extra pass over the regularised tables."""
        st = getattr(self, "_l1_track", None)
        if st is None or st["scale"] != scale:
            buf = torch.zeros(8192, dtype=torch.float32, device=self.flat_p.device)  # [0, 4096): the update's workgroups (part B of a two-part update), [4096, 8192): part A's
            buf[0] = self.l1_value(scale)  # once; every update that goes through overwrites one entry per workgroup
            st = self._l1_track = dict(buf=buf, scale=scale)
        return st["buf"]

    @torch.no_grad()
    def step(self, closure=None):
        d = self.defaults
        st = getattr(self, "_l1_track", None)
        capturing = torch.cuda.is_current_stream_capturing()
        if self._cold_dirty and not capturing:
            self.flush()  # with the old bitmap
            self._cold_bits = self._build_cold_bits()
            self._cold_dirty = False
        cold = self._cold_bits if (self.touched is not None and self._outside_is_zero and not self._cold_dirty
                                   and getattr(self, "_half_grad", None) is None) else None
        lazy = self._lazy_state() if cold is not None else None
        if lazy is not None:
            lazy = (lazy[0], lazy[1], self._warm_groups)
        if lazy is None and self._lazy_logged and not capturing:
            self.flush()  # this step decays the cold groups itself: the logged decays come first
        if lazy is not None:
            if self._lazy_logged > self.LAZY_CAPACITY - 1024 and not capturing:
                self.flush()
            self._lazy_logged += 1
        self.run_part_a()  # (never two steps' worth owed)
        self._l1_layout_is("two" if self._two_part_now(lazy) else "one", capturing)
        two = self._two_part_now(lazy)
        # zero what the update reads (zero_in_step): only when everything that can be non-zero IS read -- the warm list contains
        # the touched set, outside of which the gradient buffer is known to be zero; a pending half-precision gradient lives in
        # its own buffer
        zero_after = bool(self.zero_in_step and getattr(self, "_half_grad", None) is None
                          and self.touched is not None and self._outside_is_zero and cold is not None and lazy is not None and len(lazy) >= 3)
        cg, self._compact = self._compact, None
        assert cg is None or two or (lazy is not None and len(lazy) >= 3 and self._warm_A.numel() == 0), \
            "a compact gradient was handed over (take_compact) but the update does not walk the touched set as a list of its own"
        if two:
            # part B (what the backward may have written) + the tail, which records the scalars this step used; part A is owed
            # the step's record: two buffers taken in turn (part A of step k may still be reading its own while step k + 1 writes), and
            # one of its own for a recording's last step (carry_last)
            self._snap_buffers()
            if self.carry_last and self._next_step_is_last:
                self._next_step_is_last = False
                self._snapshot = self._snap_last
            else:
                self._snapshot = self._snapshots[1] if self._snapshot is self._snapshots[0] else self._snapshots[0]
            pvd_hip.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.segment_ends, self.lr_dev, d["betas"][0], d["betas"][1],
                               d["eps"], d["weight_decay"], self.step_count, getattr(self, "grad_scale", None), getattr(self, "found_inf", None),
                               schedule=getattr(self, "_schedule", None), l1_ranges=getattr(self, "_l1", None),
                               amp_update=getattr(self, "amp_update", None), l1_next=(st["buf"], st["scale"]) if st is not None else None,
                               cold_bits=cold, lazy=(lazy[0], lazy[1], self._warm_B if cg is None else self._warm_B[cg["rows"][0]:cg["rows"][1]]),
                               snapshot=self._snapshot, zero_after=zero_after and cg is None, arrivals=self._tail_in_kernel(),
                               **({} if cg is None else dict(compact_grad=cg["grad"], compact_param_out=cg["param_out"], tail_clear=cg["clear"])))
            if cg is not None:
                zero_after = True  # (the gather that filled the buffer zeroed every touched row behind itself)
            self._part_a_owed = (cold, lazy[0], lazy[1], self._warm_A, getattr(self, "grad_scale", None) is not None,
                                 (st["buf"][4096:], st["scale"]) if st is not None else None, self._snapshot)
            if not self.defer_part_a:
                self.run_part_a()
        else:
            pvd_hip.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.segment_ends, self.lr_dev, d["betas"][0], d["betas"][1],
                               d["eps"], d["weight_decay"], self.step_count, getattr(self, "grad_scale", None), getattr(self, "found_inf", None),
                               schedule=getattr(self, "_schedule", None), l1_ranges=getattr(self, "_l1", None),
                               amp_update=getattr(self, "amp_update", None), half_grad=getattr(self, "_half_grad", None),
                               l1_next=(st["buf"], st["scale"]) if st is not None else None, cold_bits=cold,
                               lazy=lazy if cg is None else (lazy[0], lazy[1], self._warm_B[cg["rows"][0]:cg["rows"][1]]),
                               zero_after=zero_after and cg is None, arrivals=self._tail_in_kernel(),
                               **({} if cg is None else dict(compact_grad=cg["grad"], compact_param_out=cg["param_out"], tail_clear=cg["clear"])))
            if cg is not None:
                zero_after = True  # (the exchange's gather zeroed every touched row behind itself)
        self._zeroed_by_step = zero_after
        hg = getattr(self, "_half_grad", None)
        # (every zero_grad leaves the whole buffer zero; a step whose table gradient was taken in half precision and whose fp32 range
        # nobody wrote leaves that range zero)
        self._fp32_range_still_zero = (hg[0], hg[1]) if (hg is not None and not self._half_range_dirty) else None
        self._half_grad = None
        self._half_range_dirty = False
        pvd_hip.note_weights_changed(self.params)  # the kernel rewrites the parameters without bumping their autograd versions
        # (GradScaler sets grad_scale / found_inf right before step() and deletes them afterwards)


class DeviceSchedule:
    """Stand-in for a torch LR scheduler when FlatAdamW evaluates the schedule on the device: step() only counts."""

    def __init__(self, optimizer, kind, T, param):
        optimizer.set_schedule(kind, T, param)
        self.optimizer, self.last_epoch = optimizer, 0

    def step(self):
        self.last_epoch += 1

    def get_last_lr(self):
        return [float(v) for v in self.optimizer.lr_dev.tolist()]


class FlatGradScaler(torch.amp.GradScaler):
    """GradScaler whose inf check for a FlatAdamW is one read-only pass over the flat gradient buffer
    (pvd_check_finite) instead of the multi-tensor check-and-unscale-by-1 (which also rewrites every gradient)."""

    def step(self, optimizer, *args, **kwargs):
        """For a FlatAdamW the whole scaler protocol of a step is three launches on the device: inf check (read-only),
        the update kernel (unscales, skips on inf), and a tail that does what update() would do and clears the flag --
        instead of fill + check-and-rewrite + sum + update_scale issued from the host around the optimizer."""
        if not self._enabled or not isinstance(optimizer, FlatAdamW) or "closure" in kwargs:
            return super().step(optimizer, *args, **kwargs)
        from torch.amp.grad_scaler import OptState
        self._check_scale_growth_tracker("step")
        state = self._per_optimizer_states[id(optimizer)]
        if state["stage"] is OptState.STEPPED:
            raise RuntimeError("step() has already been called since the last update().")
        if state["stage"] is OptState.UNSCALED:
            return super().step(optimizer, *args, **kwargs)  # unscale_() was called explicitly: generic path
        flag = optimizer.inf_flag() if optimizer._compact is None else optimizer._compact["flag"]
        if optimizer._checked_by_backward:
            optimizer._checked_by_backward = False  # the backward's own launches looked at every gradient they completed
        else:
            optimizer.check_finite(flag)
        optimizer.grad_scale, optimizer.found_inf = self._scale, flag
        optimizer.amp_update = (self._scale, self._growth_tracker, self._growth_factor, self._backoff_factor, self._growth_interval)
        try:
            ret = optimizer.step(*args, **kwargs)
        finally:
            del optimizer.grad_scale, optimizer.found_inf, optimizer.amp_update
        state["stage"] = OptState.STEPPED
        self._update_done_on_device = True
        return ret

    def update(self, new_scale=None):
        if getattr(self, "_update_done_on_device", False) and new_scale is None:
            from collections import defaultdict
            from torch.amp.grad_scaler import _refresh_per_optimizer_state
            self._update_done_on_device = False
            self._per_optimizer_states = defaultdict(_refresh_per_optimizer_state)
            return
        super().update(new_scale)

    def _check_inf_per_device(self, optimizer):
        if not isinstance(optimizer, FlatAdamW):
            return super()._check_inf_per_device(optimizer)
        _scale, _ = self._check_scale_growth_tracker("_check_inf_per_device")
        found_inf = torch.full((), 0.0, dtype=torch.float32, device=_scale.device)
        optimizer.check_finite(found_inf.view(1))
        self._per_optimizer_states[id(optimizer)]["found_inf_per_device"] = {_scale.device: found_inf}
        return self._per_optimizer_states[id(optimizer)]["found_inf_per_device"]
