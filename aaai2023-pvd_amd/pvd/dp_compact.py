"""Compact gradient exchange for ray data parallelism.

A table entry of the VM planes (or the Plenoxel volume) can only receive a gradient from a sample whose interpolation
footprint covers it, and samples exist only inside occupied cells of the density bitfield -- which is replicated and
frozen during distillation.  So the set of entries that can be non-zero on ANY rank is known up front: the projection
of the occupied cells onto each table, dilated by the footprint.  The all-reduce then only needs those rows (19 % of
the VM planes of the bench scene: 69 MB -> 13 MB per step), and it is exact: everything left out is zero on every rank.

Only valid when nothing else writes dense gradients into those tables: the L1 regulariser must be applied inside the
optimizer kernel (FlatAdamW.set_l1) or be off -- the trainer checks that."""
import torch


def _gather3(x):
    x = x & 0x49249249
    x = (x | (x >> 2)) & 0xC30C30C3
    x = (x | (x >> 4)) & 0x0F00F00F
    x = (x | (x >> 8)) & 0xFF0000FF
    x = (x | (x >> 16)) & 0x0000FFFF
    return x


def occupied_cells(bitfield, cascade, grid_size, bound):
    """[(lo [n,3], hi [n,3])] world-space boxes of the occupied cells (Morton order, raymarching.cu:58-83), all levels."""
    bits = bitfield.to(torch.int64)
    H3 = grid_size ** 3
    boxes = []
    for c in range(cascade):
        b = bits[c * H3 // 8:(c + 1) * H3 // 8]
        occ = ((b[:, None] >> torch.arange(8, device=b.device)) & 1).reshape(-1).nonzero().squeeze(-1)
        if occ.numel() == 0:
            continue
        cell = torch.stack([_gather3(occ), _gather3(occ >> 1), _gather3(occ >> 2)], dim=1).double()
        half = float(min(2 ** c, bound))
        lo = -half + 2 * half * cell / grid_size
        boxes.append((lo, lo + 2 * half / grid_size))
    return boxes


def footprint_mask(boxes, aabb, sizes, axes, margin=1):
    """Boolean mask over a table with `sizes` texels per table axis (fastest axis first); table axis k samples world
    axis axes[k] at grid_sample's align_corners=True coordinate.  Marks every texel a linear-interpolation footprint of
    a point inside one of the boxes can touch, plus `margin` texels (rounding at cell faces)."""
    dev = boxes[0][0].device if boxes else torch.device("cpu")
    nd = len(sizes)
    mask = torch.zeros(*reversed(sizes), dtype=torch.bool, device=dev)  # [.., H, W]
    for lo, hi in boxes:
        lo_t, hi_t = [], []
        for k in range(nd):
            a, n = axes[k], sizes[k]
            ext = float(aabb[a + 3] - aabb[a])
            u_lo = ((2 * (lo[:, a] - float(aabb[a])) / ext - 1) + 1) / 2 * (n - 1)
            u_hi = ((2 * (hi[:, a] - float(aabb[a])) / ext - 1) + 1) / 2 * (n - 1)
            lo_t.append((torch.floor(u_lo).long() - margin).clamp(0, n - 1))
            hi_t.append((torch.floor(u_hi).long() + 1 + margin).clamp(0, n - 1))
        span = [int((h - l).max().item()) + 1 for l, h in zip(lo_t, hi_t)]
        grids = torch.meshgrid(*[torch.arange(s, device=dev) for s in span], indexing="ij")
        for offs in zip(*[g.reshape(-1).tolist() for g in grids]):
            idx, ok = [], None
            for k in range(nd):
                t = lo_t[k] + offs[k]
                o = t <= hi_t[k]
                ok = o if ok is None else ok & o
                idx.append(t)
            sel = [i[ok] for i in idx]
            mask[tuple(reversed(sel))] = True
    return mask


class GradCompactor:
    """Index list (into the flat gradient buffer) of everything that has to be exchanged."""

    def __init__(self, model, params, offsets, device, marcher=None):
        """model: owner of the tables; marcher: the model whose occupancy grid the samples come from (the student itself, or
        the teacher when it renders first -- distill_mutual/renderer.py:365-411)."""
        marcher = marcher if marcher is not None else model
        aabb = [float(v) for v in model.aabb_train.tolist()]
        boxes = occupied_cells(marcher.density_bitfield.to(device), marcher.cascade, marcher.grid_size, marcher.bound)
        masked = {}  # id(param) -> (row mask flattened, channels)
        if model.model_type == "vm":
            for mats in (model.sigma_mat, model.color_mat):
                for i, p in enumerate(mats):
                    m0, m1 = model.mat_ids[i]
                    _, R, Hh, Ww = p.shape
                    masked[id(p)] = (footprint_mask(boxes, aabb, [Ww, Hh], [m0, m1]).reshape(-1), R)
        elif model.model_type == "tensors":
            p = model.tensor_volume[0]
            _, C, D, Hh, Ww = p.shape
            masked[id(p)] = (footprint_mask(boxes, aabb, [Ww, Hh, D], [0, 1, 2]).reshape(-1), C)
        parts, total, kept = [], 0, 0
        for p, off in zip(params, offsets):
            n = p.numel()
            total += n
            if id(p) in masked:
                rows, R = masked[id(p)]
                assert rows.numel() * R == n and p.stride(1) == 1, "masked tables must be stored channels-last"
                r = rows.nonzero().squeeze(-1)
                parts.append((off + r[:, None] * R + torch.arange(R, device=device)).reshape(-1))
            else:
                parts.append(torch.arange(off, off + n, device=device))
            kept += parts[-1].numel()
        self.fraction = kept / max(total, 1)
        self.idx = torch.cat(parts).to(torch.int64) if parts else None
        self.occ_epoch = marcher.occ_epoch  # the set is valid for this state of the marcher's occupancy grid only
        self.segs = segments_of(self.idx) if self.idx is not None else None  # the same set as a run table, for the HIP kernels

    # On the GPU the set is walked as a run table by pvd_segments_op (one workgroup per run, float4 moves) instead of an
    # int64 index per element; on the CPU (gloo tests) it is plain torch indexing.
    def gather(self, flat):
        if flat.is_cuda:
            import pvd_hip
            buf = torch.empty(self.idx.numel(), dtype=flat.dtype, device=flat.device)
            pvd_hip.segments_op(pvd_hip.SEG_GATHER, flat, self.segs, buf=buf)
            return buf
        return flat.index_select(0, self.idx)

    def scatter(self, flat, buf, found_inf=None):
        """flat[idx] = buf; found_inf (float [1], optional): set to 1 if buf holds an inf / nan -- the inf check of exactly the rows
        check_finite() would look at, in the same pass."""
        if flat.is_cuda:
            import pvd_hip
            if found_inf is not None:
                pvd_hip.segments_op(pvd_hip.SEG_SCATTER_CHECK, flat, self.segs, buf=buf, found_inf=found_inf)
            else:
                pvd_hip.segments_op(pvd_hip.SEG_SCATTER, flat, self.segs, buf=buf)
        else:
            flat.index_copy_(0, self.idx, buf)
            if found_inf is not None:
                found_inf.masked_fill_(~torch.isfinite(buf).all(), 1.0)

    def zero(self, flat):
        """flat[idx] = 0: all a zero_grad has to do once the rest of the buffer is known to be zero."""
        if flat.is_cuda:
            import pvd_hip
            pvd_hip.segments_op(pvd_hip.SEG_ZERO, flat, self.segs)
        else:
            flat.index_fill_(0, self.idx, 0.0)

    def check_finite(self, flat, found_inf):
        """found_inf[0] = 1 if flat[idx] holds an inf / nan (everything outside idx is zero)."""
        if flat.is_cuda:
            import pvd_hip
            pvd_hip.segments_op(pvd_hip.SEG_CHECK, flat, self.segs, found_inf=found_inf)
        else:
            found_inf.masked_fill_(~torch.isfinite(flat.index_select(0, self.idx)).all(), 1.0)


class ExchangeLayout:
    """The compact gradient as it crosses the links (round 6): `chunks` equal chunks of whole groups of four floats, each followed
    by ONE flag group whose first word is the step's inf flag -- the collective sums the flags with the data, so every rank ends up
    with the global verdict and GradScaler's check needs neither a launch nor a collective of its own.
        chunks == 1: one all-reduce of the buffer (every rank updates every row);
        chunks == world: reduce_scatter -> this rank's AdamW on its chunk's rows -> all_gather of the updated rows (sharded update).
    The run table walks the set in the optimizer's list order (group j of the touched set = list entry j of FlatAdamW's part-B
    list), cut at chunk boundaries; `xbuf` / `pbuf` are the persistent gradient / parameter exchange buffers."""

    def __init__(self, compactor, groups, chunks, device, with_params=False):
        idx = compactor.idx
        assert idx.numel() % 4 == 0, "the touched set must consist of whole groups of four parameters"
        g = idx.view(-1, 4)
        assert bool((g[:, 0] % 4 == 0).all()) and bool((g[:, 3] == g[:, 0] + 3).all()), "the touched set must consist of aligned groups of four"
        assert groups.numel() == g.shape[0] and bool((groups.long() == (g[:, 0] >> 2)).all()), \
            "the optimizer's part-B list must be the touched set, group for group"
        self.n_groups = G = int(g.shape[0])
        self.chunks = int(chunks)
        assert G >= self.chunks, "fewer touched groups than chunks: every rank of a sharded update needs rows of its own"
        self.gpc = (G + self.chunks - 1) // self.chunks  # groups per chunk
        self.chunk = 4 * self.gpc + 4                    # floats per chunk (data + the flag group)
        self.slot = 4 * self.gpc                         # the flag word's offset inside a chunk
        tables = []
        for r in range(self.chunks):
            part = idx[4 * r * self.gpc:4 * min((r + 1) * self.gpc, G)]
            t = segments_of(part)
            t[:, 1] += r * self.chunk
            tables.append(t)
        self.segs = torch.cat(tables).contiguous()
        self.xbuf = torch.zeros(self.chunks * self.chunk, dtype=torch.float32, device=device)
        self.pbuf = torch.zeros_like(self.xbuf) if with_params else None

    def rows_of(self, r):
        """(first, one past last) list entry of chunk r"""
        return r * self.gpc, min((r + 1) * self.gpc, self.n_groups)

    def chunk_of(self, buf, r):
        return buf[r * self.chunk:(r + 1) * self.chunk]


SEG_MAX = 4096  # elements per run-table entry = per workgroup


def segments_of(idx, seg_max=SEG_MAX):
    """Sorted unique element indices -> run table [n, 3] int32 (start, dst, len): maximal runs of consecutive indices, cut
    into pieces of at most seg_max elements; dst = position of the piece in the compact order."""
    n = idx.numel()
    dev = idx.device
    if n == 0:
        return torch.zeros(0, 3, dtype=torch.int32, device=dev)
    assert int(idx[-1]) < 2 ** 31
    brk = torch.ones(n, dtype=torch.bool, device=dev)
    brk[1:] = idx[1:] != idx[:-1] + 1
    run_pos = brk.nonzero().squeeze(-1)  # compact position where each run starts
    run_len = torch.diff(torch.cat([run_pos, torch.tensor([n], device=dev)]))
    pieces = (run_len + seg_max - 1) // seg_max
    run_of = torch.repeat_interleave(torch.arange(run_pos.numel(), device=dev), pieces)
    first = torch.cumsum(pieces, 0) - pieces  # index of each run's first piece
    k = torch.arange(run_of.numel(), device=dev) - first[run_of]  # piece number inside its run
    dst = run_pos[run_of] + k * seg_max
    length = torch.minimum(run_len[run_of] - k * seg_max, torch.tensor(seg_max, device=dev))
    start = idx[run_pos][run_of] + k * seg_max
    return torch.stack([start, dst, length], dim=1).to(torch.int32).contiguous()
