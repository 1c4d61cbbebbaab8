"""hipGraph recording helpers of the trainers: a step as a chain of graphs with eager host calls between them
(`SegmentedCapture`), and the static home that carries a forked prefix from one replay to the next (`CarriedPrefix`)."""
import gc
import os

import torch


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
            # eight bytes per element (a byte-wise multi-tensor copy moved the prefix's ~10 MB at 250 GB/s: 40 us on the branch of a
            # graph's last step), the last nbytes % 8 as bytes
            n8 = sn.nbytes() // 8
            if n8:
                srcs.append(torch.empty(0, dtype=torch.int64, device=tn.device).set_(sn, 0, (n8,), (1,)))
                dsts.append(torch.empty(0, dtype=torch.int64, device=to.device).set_(so, 0, (n8,), (1,)))
            if sn.nbytes() % 8:
                srcs.append(torch.empty(0, dtype=torch.uint8, device=tn.device).set_(sn, 8 * n8, (sn.nbytes() - 8 * n8,), (1,)))
                dsts.append(torch.empty(0, dtype=torch.uint8, device=to.device).set_(so, 8 * n8, (so.nbytes() - 8 * n8,), (1,)))
        for dt in (torch.int64, torch.uint8):  # (one multi-tensor launch per element type)
            d = [t for t in dsts if t.dtype == dt]
            if d:
                torch._foreach_copy_(d, [t for t in srcs if t.dtype == dt])


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
