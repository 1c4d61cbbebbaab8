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

namespace pvd {

constexpr uint32_t kVmBlock = 256;
constexpr uint32_t kRs = 16;  // sigma_rank (network.py:79)
constexpr uint32_t kRc = 48;  // color_rank (network.py:80)

struct VmTables {
    const float *mat[2][3];  // [0] = sigma, [1] = colour; channels-last [H][W][R]
    const float *vec[2][3];  // channels-last [L][R]
    uint32_t W[3], H[3], L[3];
    float lo[3], inv_extent2[3];  // x_n = 2*(x-lo)/(hi-lo) - 1, kept as (2*(x-lo)) / (hi-lo) - 1
    float extent[3];
};

struct VmGrads {
    float *mat[2][3];
    float *vec[2][3];
};

typedef _Float16 half_t;

// per-axis sampling state, identical in every lane (grid_sampler_unnormalize, align_corners=True)
struct Tap1 {
    int i0;       // floor(pos); taps at i0 and i0+1
    float w0, w1; // (i0+1 - pos), (pos - i0)
    bool in0, in1;
};

__device__ __forceinline__ Tap1 tap1(float coord, uint32_t size) {
    const float pos = ((coord + 1.0f) / 2.0f) * (float)(size - 1);
    const float fl = floorf(pos);
    Tap1 t;
    t.i0 = (int)fl;
    t.w1 = pos - fl;
    t.w0 = (fl + 1.0f) - pos;
    t.in0 = t.i0 >= 0 && t.i0 < (int)size;
    t.in1 = t.i0 + 1 >= 0 && t.i0 + 1 < (int)size;
    return t;
}

__device__ __forceinline__ void normalise(const float *__restrict__ xyz, size_t m, const VmTables &tb, float (&xn)[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) xn[a] = (2.0f * (xyz[3 * m + a] - tb.lo[a])) / tb.extent[a] - 1.0f;  // network.py:345-350
}

constexpr int kM0[3] = {0, 0, 1}, kM1[3] = {1, 2, 2}, kV[3] = {2, 1, 0};

template <typename T>
__global__ void __launch_bounds__(kVmBlock) k_vm_fwd(const float *__restrict__ xyz, uint32_t M, uint32_t chunk, VmTables tb,
                                                     float *__restrict__ sigma_feat, T *__restrict__ color_prod) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kVmBlock + threadIdx.x) >> 6;
    const uint32_t s0 = wave * chunk;
    if (s0 >= M) return;
    const uint32_t s1 = min(M, s0 + chunk);
    const uint32_t kind = lane < kRs ? 0u : 1u;       // 0 sigma, 1 colour
    const uint32_t R = kind ? kRc : kRs;
    const uint32_t ch = kind ? lane - kRs : lane;

    for (uint32_t m = s0; m < s1; m++) {
        float xn[3];
        normalise(xyz, m, tb, xn);
        float sig = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const Tap1 tx = tap1(xn[kM0[i]], tb.W[i]), ty = tap1(xn[kM1[i]], tb.H[i]), tl = tap1(xn[kV[i]], tb.L[i]);
            const float *__restrict__ mat = tb.mat[kind][i] + ch;
            const float *__restrict__ vec = tb.vec[kind][i] + ch;
            const size_t W = tb.W[i];
            // 4 plane taps + 2 line taps, all independent loads
            const float nw = (tx.in0 && ty.in0) ? mat[((size_t)ty.i0 * W + tx.i0) * R] : 0.f;
            const float ne = (tx.in1 && ty.in0) ? mat[((size_t)ty.i0 * W + tx.i0 + 1) * R] : 0.f;
            const float sw = (tx.in0 && ty.in1) ? mat[((size_t)(ty.i0 + 1) * W + tx.i0) * R] : 0.f;
            const float se = (tx.in1 && ty.in1) ? mat[((size_t)(ty.i0 + 1) * W + tx.i0 + 1) * R] : 0.f;
            const float l0 = tl.in0 ? vec[(size_t)tl.i0 * R] : 0.f;
            const float l1 = tl.in1 ? vec[(size_t)(tl.i0 + 1) * R] : 0.f;
            // accumulate in grid_sample's tap order nw, ne, sw, se
            float pv = nw * (tx.w0 * ty.w0);
            pv += ne * (tx.w1 * ty.w0);
            pv += sw * (tx.w0 * ty.w1);
            pv += se * (tx.w1 * ty.w1);
            // the line is a [L,1] image sampled at x = 0: weights (1 * w0, 0 * w0, 1 * w1, 0 * w1)
            float lv = l0 * tl.w0;
            lv += l1 * tl.w1;
            const float prod = pv * lv;
            if (kind) color_prod[(size_t)m * (3 * kRc) + i * kRc + ch] = (T)prod;
            else sig += prod;
        }
        // sum the 16 sigma lanes (lanes >= 16 carry 0)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) sig += __shfl_xor(sig, off, 64);
        if (lane == 0) sigma_feat[m] = sig;
    }
}

// Gradient accumulation windows.  Consecutive samples of a ray move by a fraction of a texel, so the 2x2
// plane footprint (and the 2-tap line footprint) of sample k+1 usually overlaps that of sample k.  Each
// wave keeps the current footprint's partial sums in registers; when the footprint slides by one texel
// along one axis only the row/column that LEAVES is flushed (one contiguous 64-lane atomic per texel, lane =
// channel), the overlapping one is shifted in registers; any other move flushes everything.  Every texel a
// ray crosses thus receives about one atomic per contiguous visit instead of one per sample and tap --
// the memory-side atomic rate, not bandwidth, is what bounds this kernel (rocprofv3: WRITE_SIZE 153 MB/launch).
struct PlaneWin { int x0, y0; float a00, a01, a10, a11; bool open; };  // a[dy][dx]
struct LineWin { int l0; float a0, a1; bool open; };

__device__ __forceinline__ void atom(float *__restrict__ p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void flush_texel(float *__restrict__ gm, int x, int y, int W, int H, uint32_t R, float v) {
    if (x >= 0 && x < W && y >= 0 && y < H) atom(gm + ((long)y * W + x) * (long)R, v);
}

__device__ __forceinline__ void plane_move(PlaneWin &w, int nx, int ny, float *__restrict__ gm, int W, int H, uint32_t R) {
    if (!w.open) {
        w.open = true; w.x0 = nx; w.y0 = ny; w.a00 = w.a01 = w.a10 = w.a11 = 0.f;
        return;
    }
    const int dx = nx - w.x0, dy = ny - w.y0;
    if (dx == 0 && dy == 0) return;
    if (dy == 0 && dx == 1) {          // column x0 leaves
        flush_texel(gm, w.x0, w.y0, W, H, R, w.a00); flush_texel(gm, w.x0, w.y0 + 1, W, H, R, w.a10);
        w.a00 = w.a01; w.a10 = w.a11; w.a01 = 0.f; w.a11 = 0.f;
    } else if (dy == 0 && dx == -1) {  // column x0+1 leaves
        flush_texel(gm, w.x0 + 1, w.y0, W, H, R, w.a01); flush_texel(gm, w.x0 + 1, w.y0 + 1, W, H, R, w.a11);
        w.a01 = w.a00; w.a11 = w.a10; w.a00 = 0.f; w.a10 = 0.f;
    } else if (dx == 0 && dy == 1) {   // row y0 leaves
        flush_texel(gm, w.x0, w.y0, W, H, R, w.a00); flush_texel(gm, w.x0 + 1, w.y0, W, H, R, w.a01);
        w.a00 = w.a10; w.a01 = w.a11; w.a10 = 0.f; w.a11 = 0.f;
    } else if (dx == 0 && dy == -1) {  // row y0+1 leaves
        flush_texel(gm, w.x0, w.y0 + 1, W, H, R, w.a10); flush_texel(gm, w.x0 + 1, w.y0 + 1, W, H, R, w.a11);
        w.a10 = w.a00; w.a11 = w.a01; w.a00 = 0.f; w.a01 = 0.f;
    } else {
        flush_texel(gm, w.x0, w.y0, W, H, R, w.a00); flush_texel(gm, w.x0 + 1, w.y0, W, H, R, w.a01);
        flush_texel(gm, w.x0, w.y0 + 1, W, H, R, w.a10); flush_texel(gm, w.x0 + 1, w.y0 + 1, W, H, R, w.a11);
        w.a00 = w.a01 = w.a10 = w.a11 = 0.f;
    }
    w.x0 = nx; w.y0 = ny;
}
__device__ __forceinline__ void plane_close(PlaneWin &w, float *__restrict__ gm, int W, int H, uint32_t R) {
    if (!w.open) return;
    flush_texel(gm, w.x0, w.y0, W, H, R, w.a00); flush_texel(gm, w.x0 + 1, w.y0, W, H, R, w.a01);
    flush_texel(gm, w.x0, w.y0 + 1, W, H, R, w.a10); flush_texel(gm, w.x0 + 1, w.y0 + 1, W, H, R, w.a11);
    w.open = false;
}

__device__ __forceinline__ void flush_line_texel(float *__restrict__ gv, int l, int L, uint32_t R, float v) {
    if (l >= 0 && l < L) atom(gv + (long)l * R, v);
}
__device__ __forceinline__ void line_move(LineWin &w, int nl, float *__restrict__ gv, int L, uint32_t R) {
    if (!w.open) {
        w.open = true; w.l0 = nl; w.a0 = w.a1 = 0.f;
        return;
    }
    const int d = nl - w.l0;
    if (d == 0) return;
    if (d == 1) { flush_line_texel(gv, w.l0, L, R, w.a0); w.a0 = w.a1; w.a1 = 0.f; }
    else if (d == -1) { flush_line_texel(gv, w.l0 + 1, L, R, w.a1); w.a1 = w.a0; w.a0 = 0.f; }
    else { flush_line_texel(gv, w.l0, L, R, w.a0); flush_line_texel(gv, w.l0 + 1, L, R, w.a1); w.a0 = w.a1 = 0.f; }
    w.l0 = nl;
}
__device__ __forceinline__ void line_close(LineWin &w, float *__restrict__ gv, int L, uint32_t R) {
    if (!w.open) return;
    flush_line_texel(gv, w.l0, L, R, w.a0); flush_line_texel(gv, w.l0 + 1, L, R, w.a1);
    w.open = false;
}

template <typename T>
__global__ void __launch_bounds__(kVmBlock) k_vm_bwd(const float *__restrict__ xyz, uint32_t M, uint32_t chunk, VmTables tb,
                                                     const float *__restrict__ g_sigma, const T *__restrict__ g_prod, VmGrads gr) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kVmBlock + threadIdx.x) >> 6;
    const uint32_t s0 = wave * chunk;
    if (s0 >= M) return;
    const uint32_t s1 = min(M, s0 + chunk);
    const uint32_t kind = lane < kRs ? 0u : 1u;
    const uint32_t R = kind ? kRc : kRs;
    const uint32_t ch = kind ? lane - kRs : lane;

    PlaneWin pw[3];
    LineWin lw[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { pw[i].open = false; lw[i].open = false; }

    for (uint32_t m = s0; m < s1; m++) {
        float xn[3];
        normalise(xyz, m, tb, xn);
        const float gs = g_sigma[m];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const Tap1 tx = tap1(xn[kM0[i]], tb.W[i]), ty = tap1(xn[kM1[i]], tb.H[i]), tl = tap1(xn[kV[i]], tb.L[i]);
            const float g = kind ? (float)g_prod[(size_t)m * (3 * kRc) + i * kRc + ch] : gs;
            const float *__restrict__ mat = tb.mat[kind][i] + ch;
            const float *__restrict__ vec = tb.vec[kind][i] + ch;
            const size_t W = tb.W[i];
            const float nw = (tx.in0 && ty.in0) ? mat[((size_t)ty.i0 * W + tx.i0) * R] : 0.f;
            const float ne = (tx.in1 && ty.in0) ? mat[((size_t)ty.i0 * W + tx.i0 + 1) * R] : 0.f;
            const float sw = (tx.in0 && ty.in1) ? mat[((size_t)(ty.i0 + 1) * W + tx.i0) * R] : 0.f;
            const float se = (tx.in1 && ty.in1) ? mat[((size_t)(ty.i0 + 1) * W + tx.i0 + 1) * R] : 0.f;
            const float l0 = tl.in0 ? vec[(size_t)tl.i0 * R] : 0.f;
            const float l1 = tl.in1 ? vec[(size_t)(tl.i0 + 1) * R] : 0.f;
            float pv = nw * (tx.w0 * ty.w0);
            pv += ne * (tx.w1 * ty.w0);
            pv += sw * (tx.w0 * ty.w1);
            pv += se * (tx.w1 * ty.w1);
            float lv = l0 * tl.w0;
            lv += l1 * tl.w1;
            const float gp = g * lv;  // d loss / d plane value
            const float gl = g * pv;  // d loss / d line value

            // wave-uniform control flow: every lane shares the sample's texel coordinates
            plane_move(pw[i], tx.i0, ty.i0, gr.mat[kind][i] + ch, (int)tb.W[i], (int)tb.H[i], R);
            pw[i].a00 += gp * (tx.w0 * ty.w0);
            pw[i].a01 += gp * (tx.w1 * ty.w0);
            pw[i].a10 += gp * (tx.w0 * ty.w1);
            pw[i].a11 += gp * (tx.w1 * ty.w1);
            line_move(lw[i], tl.i0, gr.vec[kind][i] + ch, (int)tb.L[i], R);
            lw[i].a0 += gl * tl.w0;
            lw[i].a1 += gl * tl.w1;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        plane_close(pw[i], gr.mat[kind][i] + ch, (int)tb.W[i], (int)tb.H[i], R);
        line_close(lw[i], gr.vec[kind][i] + ch, (int)tb.L[i], R);
    }
}

static uint32_t pick_chunk(uint32_t M, bool backward) {
    // forward: short runs, many waves (pure latency hiding).  backward: longer runs so the accumulation
    // windows see more consecutive samples of a ray, while still filling 256 CUs x 16 waves.
    uint32_t chunk = 16;
    const uint32_t target_waves = backward ? 256u * 16u : 256u * 32u;
    while (chunk < 64 && (uint64_t)M / chunk > target_waves) chunk <<= 1;
    return chunk;
}

static int fill_tables(VmTables &tb, const void *const *tables, const uint32_t *res, const float *aabb) {
    // reference shapes (network.py:199-212): mat_i [1,R,res[m1],res[m0]], vec_i [1,R,res[vec_id],1]
    for (int i = 0; i < 3; i++) {
        tb.W[i] = res[kM0[i]];
        tb.H[i] = res[kM1[i]];
        tb.L[i] = res[kV[i]];
        if (tb.W[i] < 1 || tb.H[i] < 1 || tb.L[i] < 1) return PVD_ERR_INVALID;
        for (int k = 0; k < 2; k++) {
            tb.mat[k][i] = (const float *)tables[k * 6 + i];
            tb.vec[k][i] = (const float *)tables[k * 6 + 3 + i];
            if (!tb.mat[k][i] || !tb.vec[k][i]) return PVD_ERR_INVALID;
        }
        tb.lo[i] = aabb[i];
        tb.extent[i] = aabb[i + 3] - aabb[i];
        tb.inv_extent2[i] = 0.f;
    }
    return PVD_OK;
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_vm_forward(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                   float *sigma_feat, void *color_prod, int prod_dtype, pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!xyz || !aabb_host || !tables_host || !res_host || !sigma_feat || !color_prod) return PVD_ERR_INVALID;
    VmTables tb;
    const int rc = fill_tables(tb, tables_host, res_host, aabb_host);
    if (rc != PVD_OK) return rc;
    const uint32_t chunk = pick_chunk(M, false);
    const uint32_t waves = div_up(M, chunk);
    const dim3 grid(div_up(waves * 64u, kVmBlock)), block(kVmBlock);
    if (prod_dtype == PVD_F32)
        hipLaunchKernelGGL((k_vm_fwd<float>), grid, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, sigma_feat, (float *)color_prod);
    else if (prod_dtype == PVD_F16)
        hipLaunchKernelGGL((k_vm_fwd<half_t>), grid, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, sigma_feat, (half_t *)color_prod);
    else
        return PVD_ERR_UNSUPPORTED;
    return check_launch();
}

int pvd_vm_backward(const float *xyz, uint32_t M, const float *aabb_host, const void *const *tables_host, const uint32_t *res_host,
                    const float *grad_sigma_feat, const void *grad_color_prod, int prod_dtype, void *const *grad_tables_host,
                    pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!xyz || !aabb_host || !tables_host || !res_host || !grad_sigma_feat || !grad_color_prod || !grad_tables_host) return PVD_ERR_INVALID;
    VmTables tb;
    const int rc = fill_tables(tb, tables_host, res_host, aabb_host);
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
    if (prod_dtype == PVD_F32)
        hipLaunchKernelGGL((k_vm_bwd<float>), grid, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, grad_sigma_feat,
                           (const float *)grad_color_prod, gr);
    else if (prod_dtype == PVD_F16)
        hipLaunchKernelGGL((k_vm_bwd<half_t>), grid, block, 0, (hipStream_t)stream, xyz, M, chunk, tb, grad_sigma_feat,
                           (const half_t *)grad_color_prod, gr);
    else
        return PVD_ERR_UNSUPPORTED;
    return check_launch();
}

}  // extern "C"
