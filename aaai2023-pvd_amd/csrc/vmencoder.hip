// vmencoder.hip -- TensoRF "VM" (plane x line) feature lookup for gfx950 (MI355X), forward + backward.
//
// Replaces, for the VM student/teacher, the reference's twelve F.grid_sample calls and the
// elementwise products around them (distill_mutual/network.py:216-309: get_sigma_feat /
// get_color_feat, tables from init_one_vm :193-214).  For a sample x in [-1,1]^3 and axis triple
// i (mat_ids = [[0,1],[0,2],[1,2]], vec_ids = [2,1,0]):
//     plane_i[r] = bilinear(mat_i[r], (x[m0], x[m1]))      align_corners=True, zero padding
//     line_i[r]  = linear  (vec_i[r],  x[vec_id])
//     sigma_feat = sum_i sum_{r<16} plane_i[r] * line_i[r]
//     color_prod[i*48 + r] = plane_i[r] * line_i[r]          (r < 48)  -> basis_mat (144 -> 15)
//
// HBM layout: every factor keeps the reference's logical shape ([1,R,H,W] planes, [1,R,L,1] lines,
// so state-dicts line up) but is stored CHANNELS-LAST: physically [H][W][R] / [L][R].  The 16 sigma
// + 48 colour channels of one tap are then exactly one 64-lane wavefront: lane c < 16 owns sigma
// channel c, lane c >= 16 owns colour channel c-16, a tap is one fully coalesced 64 B + 192 B
// access instead of 64 cache lines (the reference's channel-major layout), and the backward's
// scatter-add is one contiguous 64-lane atomic per tap instead of 64 scattered ones.
// A wave walks a contiguous run of samples (samples of a ray are contiguous), so in the backward
// consecutive samples that hit the same texel are merged in registers before touching memory.
#include "pvd_device.h"
#include "head_dw_reduce.h"
#include "head_pack.h"
#include "vm_lookup.h"

#include <stdlib.h>

namespace pvd {

constexpr uint32_t kVmBlock = 256;
// Register windows.  Consecutive samples of a ray move by about half a texel, so the 2x2 plane footprint (and the
// 2-tap line footprint) of sample k+1 usually equals or overlaps that of sample k.  Each wave keeps the current
// footprint in registers -- the texel VALUES (both passes: 18 coalesced 256-byte gathers per sample otherwise, which
// made the forward L2-bandwidth-bound at 415 MB per launch) and, in the backward, the pending gradient SUMS.  When the
// footprint slides by one texel along one axis only the row/column that ENTERS is loaded and the one that LEAVES is
// flushed (one contiguous 64-lane atomic per texel, lane = channel), the overlap is shifted in registers; any other
// move reloads / flushes everything.  Every texel a ray crosses is thus read about once and receives about one
// atomic per contiguous visit instead of one per sample and tap.  Control flow is wave-uniform: every lane shares
// the sample's texel coordinates.
__device__ __forceinline__ void atom(float *__restrict__ p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool GRAD>
struct PlaneWin {
    int x0, y0;
    bool open;
    float v[4];  // texel values, index = dy * 2 + dx
    float a[4];  // pending gradient sums (GRAD only)

    __device__ __forceinline__ void fetch(int k, const float *__restrict__ mat, int W, int H, uint32_t R) {
        const int x = x0 + (k & 1), y = y0 + (k >> 1);
        v[k] = (x >= 0 && x < W && y >= 0 && y < H) ? mat[toff(y * W + x, R)] : 0.f;
    }
    __device__ __forceinline__ void flush(int k, float *__restrict__ gm, int W, int H, uint32_t R) {
        if (GRAD) {
            const int x = x0 + (k & 1), y = y0 + (k >> 1);
            if (x >= 0 && x < W && y >= 0 && y < H) atom(gm + toff(y * W + x, R), a[k]);
            a[k] = 0.f;
        }
    }
    // move the window to (nx, ny); mat / gm are the lane's channel pointers into the table / its gradient
    __device__ __forceinline__ void move(int nx, int ny, const float *__restrict__ mat, float *__restrict__ gm, int W, int H, uint32_t R) {
        if (!open) {
            open = true; x0 = nx; y0 = ny;
#pragma unroll
            for (int k = 0; k < 4; k++) { fetch(k, mat, W, H, R); a[k] = 0.f; }
            return;
        }
        const int dx = nx - x0, dy = ny - y0;
        if (dx == 0 && dy == 0) return;
        // static register indices in every branch (a runtime index would send the arrays to scratch)
        if (dy == 0 && dx == 1) slide<0, 1, 2, 3>(nx, ny, mat, gm, W, H, R);         // column 0 leaves, column 1 -> 0
        else if (dy == 0 && dx == -1) slide<1, 0, 3, 2>(nx, ny, mat, gm, W, H, R);   // column 1 leaves, column 0 -> 1
        else if (dx == 0 && dy == 1) slide<0, 2, 1, 3>(nx, ny, mat, gm, W, H, R);    // row 0 leaves, row 1 -> 0
        else if (dx == 0 && dy == -1) slide<2, 0, 3, 1>(nx, ny, mat, gm, W, H, R);   // row 1 leaves, row 0 -> 1
        else {
#pragma unroll
            for (int k = 0; k < 4; k++) flush(k, gm, W, H, R);
            x0 = nx; y0 = ny;
#pragma unroll
            for (int k = 0; k < 4; k++) fetch(k, mat, W, H, R);
        }
    }
    // texels LA, LB leave (flushed at the old position); SA, SB stay and move into LA, LB; SA, SB are re-fetched
    template <int LA, int SA, int LB, int SB>
    __device__ __forceinline__ void slide(int nx, int ny, const float *__restrict__ mat, float *__restrict__ gm, int W, int H, uint32_t R) {
        flush(LA, gm, W, H, R); flush(LB, gm, W, H, R);
        v[LA] = v[SA]; v[LB] = v[SB];
        if (GRAD) { a[LA] = a[SA]; a[LB] = a[SB]; a[SA] = 0.f; a[SB] = 0.f; }
        x0 = nx; y0 = ny;
        fetch(SA, mat, W, H, R); fetch(SB, mat, W, H, R);
    }
    __device__ __forceinline__ void close(float *__restrict__ gm, int W, int H, uint32_t R) {
        if (!open) return;
#pragma unroll
        for (int k = 0; k < 4; k++) flush(k, gm, W, H, R);
        open = false;
    }

    // ---- interior fast path: the move code comes precomputed with the sample (see WalkCtl), old and new footprints
    // are known to lie inside the table, so there are no comparisons and no bounds checks; `t` is a texel index.
    template <int K>
    __device__ __forceinline__ void fetch_at(int t, const float *__restrict__ mat, uint32_t R) { v[K] = mat[toff(t, R)]; }
    template <int K>
    __device__ __forceinline__ void flush_at(int t, float *__restrict__ gm, uint32_t R) {
        if (GRAD) { atom(gm + toff(t, R), a[K]); a[K] = 0.f; }
    }
    // The entering texels are LOADED before the leaving ones are flushed: memory operations complete in issue order
    // (one vmcnt), so a load issued behind the atomics would make the next use of the values wait for the atomics.
    __device__ __forceinline__ float ld(int t, const float *__restrict__ mat, uint32_t R) const { return mat[toff(t, R)]; }
    __device__ __forceinline__ void move_fast(uint32_t code, int nx, int ny, const float *__restrict__ mat, float *__restrict__ gm, int W, uint32_t R) {
        const int nb = ny * W + nx;  // new origin texel
        if (code == 1) {         // +x: column 0 leaves
            const float e1 = ld(nb + 1, mat, R), e3 = ld(nb + W + 1, mat, R);
            flush_at<0>(nb - 1, gm, R); flush_at<2>(nb - 1 + W, gm, R);
            v[0] = v[1]; v[2] = v[3]; v[1] = e1; v[3] = e3;
            if (GRAD) { a[0] = a[1]; a[2] = a[3]; a[1] = 0.f; a[3] = 0.f; }
        } else if (code == 2) {  // -x: column 1 leaves
            const float e0 = ld(nb, mat, R), e2 = ld(nb + W, mat, R);
            flush_at<1>(nb + 2, gm, R); flush_at<3>(nb + 2 + W, gm, R);
            v[1] = v[0]; v[3] = v[2]; v[0] = e0; v[2] = e2;
            if (GRAD) { a[1] = a[0]; a[3] = a[2]; a[0] = 0.f; a[2] = 0.f; }
        } else if (code == 3) {  // +y: row 0 leaves
            const float e2 = ld(nb + W, mat, R), e3 = ld(nb + W + 1, mat, R);
            flush_at<0>(nb - W, gm, R); flush_at<1>(nb - W + 1, gm, R);
            v[0] = v[2]; v[1] = v[3]; v[2] = e2; v[3] = e3;
            if (GRAD) { a[0] = a[2]; a[1] = a[3]; a[2] = 0.f; a[3] = 0.f; }
        } else if (code == 4) {  // -y: row 1 leaves
            const float e0 = ld(nb, mat, R), e1 = ld(nb + 1, mat, R);
            flush_at<2>(nb + 2 * W, gm, R); flush_at<3>(nb + 2 * W + 1, gm, R);
            v[2] = v[0]; v[3] = v[1]; v[0] = e0; v[1] = e1;
            if (GRAD) { a[2] = a[0]; a[3] = a[1]; a[0] = 0.f; a[1] = 0.f; }
        } else {                 // jump: everything leaves (the old footprint is interior, too)
            const int ob = y0 * W + x0;
            const float e0 = ld(nb, mat, R), e1 = ld(nb + 1, mat, R), e2 = ld(nb + W, mat, R), e3 = ld(nb + W + 1, mat, R);
            flush_at<0>(ob, gm, R); flush_at<1>(ob + 1, gm, R); flush_at<2>(ob + W, gm, R); flush_at<3>(ob + W + 1, gm, R);
            v[0] = e0; v[1] = e1; v[2] = e2; v[3] = e3;
        }
        x0 = nx; y0 = ny;
    }
};

template <bool GRAD>
struct LineWin {
    int l0;
    bool open;
    float v[2], a[2];
    __device__ __forceinline__ void fetch(int k, const float *__restrict__ vec, int L, uint32_t R) {
        const int l = l0 + k;
        v[k] = (l >= 0 && l < L) ? vec[toff(l, R)] : 0.f;
    }
    __device__ __forceinline__ void flush(int k, float *__restrict__ gv, int L, uint32_t R) {
        if (GRAD) {
            const int l = l0 + k;
            if (l >= 0 && l < L) atom(gv + toff(l, R), a[k]);
            a[k] = 0.f;
        }
    }
    __device__ __forceinline__ void move(int nl, const float *__restrict__ vec, float *__restrict__ gv, int L, uint32_t R) {
        if (!open) {
            open = true; l0 = nl;
            fetch(0, vec, L, R); fetch(1, vec, L, R);
            a[0] = a[1] = 0.f;
            return;
        }
        const int d = nl - l0;
        if (d == 0) return;
        if (d == 1) {
            flush(0, gv, L, R);
            v[0] = v[1]; a[0] = a[1]; a[1] = 0.f;
            l0 = nl;
            fetch(1, vec, L, R);
        } else if (d == -1) {
            flush(1, gv, L, R);
            v[1] = v[0]; a[1] = a[0]; a[0] = 0.f;
            l0 = nl;
            fetch(0, vec, L, R);
        } else {
            flush(0, gv, L, R); flush(1, gv, L, R);
            l0 = nl;
            fetch(0, vec, L, R); fetch(1, vec, L, R);
        }
    }
    __device__ __forceinline__ void close(float *__restrict__ gv, int L, uint32_t R) {
        if (!open) return;
        flush(0, gv, L, R); flush(1, gv, L, R);
        open = false;
    }
    // interior fast path (see PlaneWin::move_fast): code 1 = +1, 2 = -1, 3 = jump
    __device__ __forceinline__ void move_fast(uint32_t code, int nl, const float *__restrict__ vec, float *__restrict__ gv, uint32_t R) {
        if (code == 1) {
            const float e = vec[toff(nl + 1, R)];
            if (GRAD) { atom(gv + toff(nl - 1, R), a[0]); a[0] = a[1]; a[1] = 0.f; }
            v[0] = v[1]; v[1] = e;
        } else if (code == 2) {
            const float e = vec[toff(nl, R)];
            if (GRAD) { atom(gv + toff(nl + 2, R), a[1]); a[1] = a[0]; a[0] = 0.f; }
            v[1] = v[0]; v[0] = e;
        } else {
            const float e0 = vec[toff(nl, R)], e1 = vec[toff(nl + 1, R)];
            if (GRAD) {
                atom(gv + toff(l0, R), a[0]); atom(gv + toff(l0 + 1, R), a[1]);
                a[0] = 0.f; a[1] = 0.f;
            }
            v[0] = e0; v[1] = e1;
        }
        l0 = nl;
    }
};

// plane value and line value of one sample from the windows, accumulated in grid_sample's tap order
template <bool GRAD>
__device__ __forceinline__ float plane_value(const PlaneWin<GRAD> &w, const Tap1 &tx, const Tap1 &ty) {
    float pv = w.v[0] * (tx.w0 * ty.w0);  // nw
    pv += w.v[1] * (tx.w1 * ty.w0);       // ne
    pv += w.v[2] * (tx.w0 * ty.w1);       // sw
    pv += w.v[3] * (tx.w1 * ty.w1);       // se
    return pv;
}
template <bool GRAD>
__device__ __forceinline__ float line_value(const LineWin<GRAD> &w, const Tap1 &tl) {
    // the line is a [L,1] image sampled at x = 0: weights (1 * w0, 0 * w0, 1 * w1, 0 * w1)
    float lv = w.v[0] * tl.w0;
    lv += w.v[1] * tl.w1;
    return lv;
}

// The sampling state of a sample (3 axes x {texel, two weights}) is the same in all 64 lanes; computing it per lane
// made both kernels instruction-issue-bound on 64-fold redundant coordinate math and window logic (rocprofv3 PMC: 389
// wave-instructions per sample forward, 0.23 IPC per SIMD = the one-instruction-per-4-cycles ceiling).  Instead lane j
// evaluates sample s0 + j once -- including, by comparing with lane j-1, HOW each window moves (same / slide +-x / +-y
// / jump) and whether old and new footprints are interior -- and the walk broadcasts sample m's ten values from
// lane m - s0 into SGPRs (v_readlane).  The common case per window is then: extract 3 bits, branch, 2 loads.
struct WalkCtl {
    int i0[3];
    float w0[3], w1[3];
    uint32_t code;  // bits 3i..3i+2: plane i move (0 same, 1 +x, 2 -x, 3 +y, 4 -y, 5 jump); bits 9+2i..: line i move
                    // (0 same, 1 +1, 2 -1, 3 jump); bit 15+i: plane i takes the interior fast path; bit 18+i: line i
};
__device__ __forceinline__ WalkCtl precompute_ctl(const float *__restrict__ xyz, uint32_t s0, uint32_t s1, uint32_t lane, const VmTables &tb) {
    WalkCtl p;
    float xn[3] = {0.f, 0.f, 0.f};
    if (s0 + lane < s1) normalise(xyz, s0 + lane, tb, xn);
    // axis a is always sampled at resolution res[a]: planes use (W_i, H_i) = (res[m0], res[m1]), lines res[vec_id]
    const int size[3] = {(int)tb.W[0], (int)tb.H[0], (int)tb.L[0]};  // res[0], res[1], res[2]  (mat_0 = (x, y), vec_0 = z)
    bool inside[3];
    int prev[3];
    bool prev_inside[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const Tap1 t = tap1(xn[a], (uint32_t)size[a]);
        p.i0[a] = t.i0; p.w0[a] = t.w0; p.w1[a] = t.w1;
        inside[a] = t.i0 >= 0 && t.i0 + 1 < size[a];  // both taps of this axis in range
        prev[a] = __shfl_up(t.i0, 1, 64);
        prev_inside[a] = __shfl_up((int)inside[a], 1, 64) != 0;
    }
    uint32_t code = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int dx = p.i0[kM0[i]] - prev[kM0[i]], dy = p.i0[kM1[i]] - prev[kM1[i]];
        uint32_t c = (dx == 0 && dy == 0) ? 0u : (dy == 0 && dx == 1) ? 1u : (dy == 0 && dx == -1) ? 2u : (dx == 0 && dy == 1) ? 3u
                                                                                                   : (dx == 0 && dy == -1) ? 4u : 5u;
        const int dl = p.i0[kV[i]] - prev[kV[i]];
        uint32_t cl = dl == 0 ? 0u : dl == 1 ? 1u : dl == -1 ? 2u : 3u;
        bool fast = inside[kM0[i]] && inside[kM1[i]] && prev_inside[kM0[i]] && prev_inside[kM1[i]];
        bool fast_l = inside[kV[i]] && prev_inside[kV[i]];
        if (lane == 0) { c = 5u; cl = 3u; fast = false; fast_l = false; }  // first sample of the run: open the windows (generic path)
        code |= c << (3 * i) | cl << (9 + 2 * i) | (fast ? 1u : 0u) << (15 + i) | (fast_l ? 1u : 0u) << (18 + i);
    }
    p.code = code;
    return p;
}
__device__ __forceinline__ float bcast_f(float v, uint32_t lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)lane));
}
struct SampleTaps {
    Tap1 ax[3];
    uint32_t code;
};
__device__ __forceinline__ SampleTaps bcast_sample(const WalkCtl &p, uint32_t lane) {
    SampleTaps t;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        t.ax[a].i0 = __builtin_amdgcn_readlane(p.i0[a], (int)lane);
        t.ax[a].w0 = bcast_f(p.w0[a], lane);
        t.ax[a].w1 = bcast_f(p.w1[a], lane);
        t.ax[a].in0 = t.ax[a].in1 = true;  // bounds are checked where texels are touched
    }
    t.code = (uint32_t)__builtin_amdgcn_readlane((int)p.code, (int)lane);
    return t;
}
// move window set i to the sample's footprint
template <bool GRAD>
__device__ __forceinline__ void walk_move(PlaneWin<GRAD> &pw, LineWin<GRAD> &lw, const SampleTaps &t, int i, const float *__restrict__ mat,
                                          float *__restrict__ gm, const float *__restrict__ vec, float *__restrict__ gv, int W, int H, int L,
                                          uint32_t R, uint32_t Rv) {
    const Tap1 &tx = t.ax[kM0[i]], &ty = t.ax[kM1[i]], &tl = t.ax[kV[i]];
    const uint32_t c = (t.code >> (3 * i)) & 7u, cl = (t.code >> (9 + 2 * i)) & 3u;
    if (c) {
        if ((t.code >> (15 + i)) & 1u) pw.move_fast(c, tx.i0, ty.i0, mat, gm, W, R);
        else pw.move(tx.i0, ty.i0, mat, gm, W, H, R);
    }
    if (cl) {
        if ((t.code >> (18 + i)) & 1u) lw.move_fast(cl, tl.i0, vec, gv, Rv);
        else lw.move(tl.i0, vec, gv, L, Rv);
    }
}

template <typename T>
__global__ void __launch_bounds__(kVmBlock) k_vm_fwd(const float *__restrict__ xyz, uint32_t M, uint32_t chunk, VmTables tb,
                                                     float *__restrict__ sigma_feat, T *__restrict__ color_prod,
                                                     const int32_t *__restrict__ rows_dev, pvd_head_pack_rider pk = pvd_head_pack_rider{},
                                                     uint32_t lookup_blocks = 0xFFFFFFFFu) {
    if (blockIdx.x >= lookup_blocks) {  // the VM head's packed weight image riding on this launch: head_pack.h
        static_assert(kVmBlock == 256, "the pack workgroups stride by 256 threads");
        head_pack_elements<KIND_VM>(pk.Wa1, nullptr, pk.Wc1, pk.Wc2, pk.Wc3, reinterpret_cast<_Float16 *>(pk.image),
                                    (int)((blockIdx.x - lookup_blocks) * kVmBlock + threadIdx.x), (int)((gridDim.x - lookup_blocks) * kVmBlock));
        return;
    }
    if (rows_dev) M = min(M, (uint32_t)max(*rows_dev, 0));  // inference rounds: the row count lives on the device
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kVmBlock + threadIdx.x) >> 6;
    const uint32_t s0 = wave * chunk;
    if (s0 >= M) return;
    const uint32_t s1 = min(M, s0 + chunk);
    const uint32_t kind = lane < kRs ? 0u : 1u;       // 0 sigma, 1 colour
    const uint32_t R = tb.ms[kind], Rv = tb.vs[kind];
    const uint32_t ch = kind ? lane - kRs : lane;

    PlaneWin<false> pw[3];
    LineWin<false> lw[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { pw[i].open = false; lw[i].open = false; }

    const WalkCtl pre = precompute_ctl(xyz, s0, s1, lane, tb);  // chunk <= 64
    for (uint32_t m = s0; m < s1; m++) {
        const SampleTaps st = bcast_sample(pre, m - s0);
        float sig = 0.f;
        // all three factor sets move their windows FIRST (the entering texels' loads of all sets are in flight together: one
        // memory round trip per sample), then the products are formed; with move and product interleaved per set the wave
        // waited for three round trips in a row -- it has few companions on its SIMD to hide them
#pragma unroll
        for (int i = 0; i < 3; i++)
            walk_move<false>(pw[i], lw[i], st, i, tb.mat[kind][i] + ch, nullptr, tb.vec[kind][i] + ch, nullptr, (int)tb.W[i], (int)tb.H[i],
                             (int)tb.L[i], R, Rv);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const Tap1 tx = st.ax[kM0[i]], ty = st.ax[kM1[i]], tl = st.ax[kV[i]];
            const float prod = plane_value(pw[i], tx, ty) * line_value(lw[i], tl);
            if (kind) color_prod[(size_t)m * (3 * kRc) + i * kRc + ch] = (T)prod;
            else sig += prod;
        }
        // sum the 16 sigma lanes (lanes >= 16 carry 0)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) sig += __shfl_xor(sig, off, 64);
        if (lane == 0) sigma_feat[m] = sig;
    }
}

// factor sets [I0, I1) of one run of samples: (i, i + 1) = one plane/line pair per wave (all three in one wave -- a third of the waves,
// three times the serial work per sample -- measured slower and was removed)
template <typename T, int I0, int I1>
__device__ __forceinline__ void vm_bwd_body(const float *__restrict__ xyz, uint32_t M, uint32_t chunk, const VmTables &tb,
                                            const float *__restrict__ g_sigma, const T *__restrict__ g_prod, const VmGrads &gr,
                                            float *__restrict__ found_inf) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kVmBlock + threadIdx.x) >> 6;
    const uint32_t s0 = wave * chunk;
    if (s0 >= M) return;
    const uint32_t s1 = min(M, s0 + chunk);
    const uint32_t kind = lane < kRs ? 0u : 1u;
    const uint32_t R = tb.ms[kind], Rv = tb.vs[kind];
    const uint32_t ch = kind ? lane - kRs : lane;
    // (pvd_head_dw_rider.found_inf) every incoming gradient value this wave reads is looked at once: exponent all ones = inf / nan
    uint32_t bad = 0u;
    auto look = [&](float v) { bad |= (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u ? 1u : 0u; };

    PlaneWin<true> pw[3];
    LineWin<true> lw[3];
#pragma unroll
    for (int i = I0; i < I1; i++) { pw[i].open = false; lw[i].open = false; }

    const WalkCtl pre = precompute_ctl(xyz, s0, s1, lane, tb);  // chunk <= 64
    const float gs_lane = (s0 + lane < s1) ? g_sigma[s0 + lane] : 0.f;
    look(gs_lane);
    // the colour lanes' incoming gradients, one sample ahead of their use (a wave has few companions on its SIMD:
    // a load issued where it is needed costs its full latency on every sample)
    T g_next[3];
#pragma unroll
    for (int i = I0; i < I1; i++) g_next[i] = kind ? g_prod[(size_t)s0 * (3 * kRc) + i * kRc + ch] : (T)0;
    for (uint32_t m = s0; m < s1; m++) {
        const SampleTaps st = bcast_sample(pre, m - s0);
        const float gs = bcast_f(gs_lane, m - s0);
        T g_cur[3];
#pragma unroll
        for (int i = I0; i < I1; i++) { g_cur[i] = g_next[i]; look((float)g_cur[i]); }
        if (m + 1 < s1) {
#pragma unroll
            for (int i = I0; i < I1; i++) g_next[i] = kind ? g_prod[(size_t)(m + 1) * (3 * kRc) + i * kRc + ch] : (T)0;
        }
#pragma unroll
        for (int i = I0; i < I1; i++) {
            const Tap1 tx = st.ax[kM0[i]], ty = st.ax[kM1[i]], tl = st.ax[kV[i]];
            const float g = kind ? (float)g_cur[i] : gs;
            walk_move<true>(pw[i], lw[i], st, i, tb.mat[kind][i] + ch, gr.mat[kind][i] + ch, tb.vec[kind][i] + ch, gr.vec[kind][i] + ch,
                            (int)tb.W[i], (int)tb.H[i], (int)tb.L[i], R, Rv);
            const float gp = g * line_value(lw[i], tl);       // d loss / d plane value
            const float gl = g * plane_value(pw[i], tx, ty);  // d loss / d line value
            pw[i].a[0] += gp * (tx.w0 * ty.w0);
            pw[i].a[1] += gp * (tx.w1 * ty.w0);
            pw[i].a[2] += gp * (tx.w0 * ty.w1);
            pw[i].a[3] += gp * (tx.w1 * ty.w1);
            lw[i].a[0] += gl * tl.w0;
            lw[i].a[1] += gl * tl.w1;
        }
    }
#pragma unroll
    for (int i = I0; i < I1; i++) {
        pw[i].close(gr.mat[kind][i] + ch, (int)tb.W[i], (int)tb.H[i], R);
        lw[i].close(gr.vec[kind][i] + ch, (int)tb.L[i], Rv);
    }
    if (found_inf && __ballot(bad != 0u) != 0ull && lane == 0) found_inf[0] = 1.0f;
}

// blockIdx.y = factor set: three times the waves, a third of the serial work per sample in each
template <typename T>
__global__ void __launch_bounds__(kVmBlock) k_vm_bwd_split(const float *__restrict__ xyz, uint32_t M, uint32_t chunk, VmTables tb,
                                                           const float *__restrict__ g_sigma, const T *__restrict__ g_prod, VmGrads gr,
                                                           pvd_head_dw_rider hd = pvd_head_dw_rider{}) {
    if (blockIdx.y == 3) {  // (gridDim.y == 4) the VM head's weight-gradient reduction riding on this launch: head_dw_reduce.h
        static_assert(kVmBlock == 256, "head_vm_reduce_dw is written for 256 threads");
        for (uint32_t rb = blockIdx.x; rb < kVmHeadReduceBlocks * kReduceSlices; rb += gridDim.x)
            head_vm_reduce_dw(hd.partials, hd.nblocks, hd.gWa1, hd.gWc1, hd.gWc2, hd.gWc3, rb % kVmHeadReduceBlocks, rb / kVmHeadReduceBlocks,
                              hd.found_inf);
        return;
    }
    if (blockIdx.y == 0) vm_bwd_body<T, 0, 1>(xyz, M, chunk, tb, g_sigma, g_prod, gr, hd.found_inf);
    else if (blockIdx.y == 1) vm_bwd_body<T, 1, 2>(xyz, M, chunk, tb, g_sigma, g_prod, gr, hd.found_inf);
    else vm_bwd_body<T, 2, 3>(xyz, M, chunk, tb, g_sigma, g_prod, gr, hd.found_inf);
}

static uint32_t pick_chunk(uint32_t M, bool backward) {
    // A wave walks its run of samples serially, and every step that moves a window waits for a memory round trip
    // (load of the entering texels / the atomics' addresses): the launch lasts as long as ONE wave's chain, so runs
    // are kept short as long as the chip is not oversubscribed several times; longer runs only buy window reuse.
    if (backward) {  // three factor sets per run (blockIdx.y): long runs merge the most atomics; keep >= 3072 waves
        uint32_t chunk = 64;
        while (chunk > 16 && 3ull * M / chunk < 3072ull) chunk >>= 1;
        return chunk;
    }
    uint32_t chunk = 16;
    while (chunk < 64 && (uint64_t)M / chunk > 256u * 32u) chunk <<= 1;
    return chunk;
}

}  // namespace pvd

using namespace pvd;

extern "C" {

static int vm_forward_impl(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                           float *sigma_feat, void *color_prod, int prod_dtype, const int32_t *rows_dev, const uint32_t *texel_stride_host,
                           const pvd_head_pack_rider *pack, pvd_stream_t stream) {
    if (!xyz || !aabb_host || !tables_host || !res_host || !sigma_feat || !color_prod) return PVD_ERR_INVALID;
    VmTables tb;
    const int rc = fill_tables(tb, tables_host, res_host, aabb_host, texel_stride_host);
    if (rc != PVD_OK) return rc;
    const uint32_t chunk = pick_chunk(M, false);
    const uint32_t waves = div_up(M, chunk);
    const uint32_t lookup_blocks = div_up(waves * 64u, kVmBlock);
    // the rider's workgroups come LAST in the grid (dispatched behind the lookup's: they fill the launch's tail), one element each
    const uint32_t pack_blocks = pack ? div_up((uint32_t)kVmImageHalfs, kVmBlock) : 0u;
    const dim3 grid(lookup_blocks + pack_blocks), block(kVmBlock);
    const pvd_head_pack_rider pk = pack ? *pack : pvd_head_pack_rider{};
    if (prod_dtype == PVD_F32)
        hipLaunchKernelGGL((k_vm_fwd<float>), grid, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, sigma_feat, (float *)color_prod, rows_dev, pk,
                           lookup_blocks);
    else if (prod_dtype == PVD_F16)
        hipLaunchKernelGGL((k_vm_fwd<half_t>), grid, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, sigma_feat, (half_t *)color_prod, rows_dev, pk,
                           lookup_blocks);
    else
        return PVD_ERR_UNSUPPORTED;
    return check_launch();
}

int pvd_vm_forward(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                   float *sigma_feat, void *color_prod, int prod_dtype, const int32_t *rows_dev, const uint32_t *texel_stride_host,
                   pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    return vm_forward_impl(xyz, M, aabb_host, tables_host, res_host, sigma_feat, color_prod, prod_dtype, rows_dev, texel_stride_host, nullptr, stream);
}

int pvd_vm_forward_pack_rider(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                              float *sigma_feat, void *color_prod, int prod_dtype, const int32_t *rows_dev, const uint32_t *texel_stride_host,
                              const pvd_head_pack_rider *pack, pvd_stream_t stream) {
    if (!pack || pack->kind != 1 || !pack->Wa1 || !pack->Wc1 || !pack->Wc2 || !pack->Wc3 || !pack->image || M == 0)
        return PVD_ERR_INVALID;  // (an owed image cannot be dropped: with no rows there is no launch to ride on)
    return vm_forward_impl(xyz, M, aabb_host, tables_host, res_host, sigma_feat, color_prod, prod_dtype, rows_dev, texel_stride_host, pack, stream);
}

static int vm_backward_impl(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                            const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype, void *const *grad_tables_host,
                            const uint32_t *texel_stride_host, const pvd_head_dw_rider *rider, pvd_stream_t stream);

int pvd_vm_backward(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                    const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype, void *const *grad_tables_host,
                    const uint32_t *texel_stride_host, pvd_stream_t stream) {
    return vm_backward_impl(xyz, M, aabb_host, tables_host, res_host, grad_sigma_feat, grad_color_prod, prod_dtype, grad_tables_host,
                            texel_stride_host, nullptr, stream);
}

int pvd_vm_backward_rider(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                          const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype, void *const *grad_tables_host,
                          const uint32_t *texel_stride_host, const pvd_head_dw_rider *rider, pvd_stream_t stream) {
    if (!rider || !rider->partials || !rider->nblocks || !rider->gWa1 || !rider->gWc1 || !rider->gWc2 || !rider->gWc3 || M == 0)
        return PVD_ERR_INVALID;  // (an owed reduction cannot be dropped)
    return vm_backward_impl(xyz, M, aabb_host, tables_host, res_host, grad_sigma_feat, grad_color_prod, prod_dtype, grad_tables_host,
                            texel_stride_host, rider, stream);
}

static int vm_backward_impl(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                            const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype, void *const *grad_tables_host,
                            const uint32_t *texel_stride_host, const pvd_head_dw_rider *rider, pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!xyz || !aabb_host || !tables_host || !res_host || !grad_sigma_feat || !grad_color_prod || !grad_tables_host) return PVD_ERR_INVALID;
    VmTables tb;
    const int rc = fill_tables(tb, tables_host, res_host, aabb_host, texel_stride_host);
    if (rc != PVD_OK) return rc;
    VmGrads gr;
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 2; k++) {
            gr.mat[k][i] = (float *)grad_tables_host[k * 6 + i];
            gr.vec[k][i] = (float *)grad_tables_host[k * 6 + 3 + i];
            if (!gr.mat[k][i] || !gr.vec[k][i]) return PVD_ERR_INVALID;
        }
    const uint32_t chunk = pick_chunk(M, true);
    const uint32_t waves = div_up(M, chunk);
    const dim3 grid(div_up(waves * 64u, kVmBlock)), block(kVmBlock);
    const dim3 grid3(grid.x, rider ? 4 : 3);  // blockIdx.y = factor set (3 = the riding reduction)
    const pvd_head_dw_rider hd = rider ? *rider : pvd_head_dw_rider{};
    if (prod_dtype == PVD_F32)
        hipLaunchKernelGGL((k_vm_bwd_split<float>), grid3, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, grad_sigma_feat,
                           (const float *)grad_color_prod, gr, hd);
    else if (prod_dtype == PVD_F16)
        hipLaunchKernelGGL((k_vm_bwd_split<half_t>), grid3, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, grad_sigma_feat,
                           (const half_t *)grad_color_prod, gr, hd);
    else
        return PVD_ERR_UNSUPPORTED;
    return check_launch();
}

}  // extern "C"
