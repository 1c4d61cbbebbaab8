// plenoxel.hip -- dense-volume ("tensors" / Plenoxels) lookup + SH colour head for gfx950 (MI355X).
//
// Replaces, for the Plenoxel student/teacher, compute_plenoxel_fea + the head around it
// (distill_mutual/network.py:311-322, 383-409): a 3-D F.grid_sample (trilinear, align_corners=True,
// zero padding) of the [1, C, D, H, W] parameter at x in [-1,1]^3, C = 3*deg^2 + 1, then
//     sigma_l = clamp(h[0], clip_min, clip_max); sigma = trunc_exp(sigma_l)            (:391-399)
//     rgb[c]  = sigmoid(sum_k h[1 + c*deg^2 + k] * SH_k(d))                            (:401-406)
// and autograd through all of it.  Everything is fp32, as in the reference (grid_sample promotes to the
// widest input under autocast; the parameter and the coordinates are fp32).
//
// HBM layout: the parameter keeps its logical [1,C,D,H,W] shape (state-dict compatible) but is stored
// CHANNELS-LAST, physically [D][H][W][C]: the C = 28 channels of a voxel are one contiguous 112 B run
// instead of 28 cache lines 8 MiB apart.  A half-wavefront (32 lanes, lane = channel) owns one stream
// of consecutive samples -- samples of a ray are contiguous and advance 0.21 voxel per step -- and keeps
// the 2x2x2 footprint of the current cell in registers: when the cell is unchanged nothing is read
// (forward) or written (backward); when it slides by one along one axis only the entering face is
// loaded / the leaving face flushed (4 taps, each one contiguous <=128 B access or atomic); any other
// move reloads / flushes all 8.  The backward is bound by the memory-side atomic rate, which is why the
// merging matters (cf. vmencoder.hip).
#include "infer_persistent.h"
#include "pvd_device.h"

namespace pvd {

#include "sh_basis.inc"

constexpr uint32_t kPxBlock = 256;

struct PxVolume {
    const float *vol;  // [D][H][W][C]
    uint32_t D, H, W, C;
    uint32_t small;  // fewer than 2^24 voxels and 2^32 elements: voxel offsets by 24-bit multiplies (see px_offset)
    float lo[3], extent[3];  // x_n = 2*(x-lo)/(hi-lo) - 1 (network.py:384-388)
    float clip_min, clip_max;
};

// per-axis sampling state (grid_sampler_unnormalize, align_corners=True)
struct PxAxis {
    int i0;
    float w0, w1;
};
__device__ __forceinline__ PxAxis px_axis(float coord, uint32_t size) {
    const float pos = ((coord + 1.0f) / 2.0f) * (float)(size - 1);
    const float fl = floorf(pos);
    PxAxis t;
    t.i0 = (int)fl;
    t.w1 = pos - fl;
    t.w0 = (fl + 1.0f) - pos;
    return t;
}

struct PxSample {
    PxAxis ax[3];  // x (W), y (H), z (D)
    float w[8];    // tap t = bx | by << 1 | bz << 2, grid_sample's order tnw, tne, tsw, tse, bnw, ...
};
__device__ __forceinline__ PxSample px_locate(const float *__restrict__ xyz, size_t m, const PxVolume &v) {
    PxSample s;
    const uint32_t size[3] = {v.W, v.H, v.D};
#pragma unroll
    for (int a = 0; a < 3; a++) s.ax[a] = px_axis((2.0f * (xyz[3 * m + a] - v.lo[a])) / v.extent[a] - 1.0f, size[a]);
#pragma unroll
    for (int t = 0; t < 8; t++)
        s.w[t] = (((t & 1) ? s.ax[0].w1 : s.ax[0].w0) * ((t & 2) ? s.ax[1].w1 : s.ax[1].w0)) * ((t & 4) ? s.ax[2].w1 : s.ax[2].w0);
    return s;
}

// the 2x2x2 footprint a half-wave currently holds
struct PxWindow {
    int c[3];    // cell (x0, y0, z0)
    float a[8];  // forward: voxel values; backward: pending gradient sums
    bool open;
};

__device__ __forceinline__ bool px_inside(const PxVolume &v, int x, int y, int z) {
    return x >= 0 && x < (int)v.W && y >= 0 && y < (int)v.H && z >= 0 && z < (int)v.D;
}
__device__ __forceinline__ long px_offset(const PxVolume &v, int x, int y, int z) {
    // the usual volumes (128^3 x 28) index with three full-rate 24-bit multiply-adds; the 64-bit expression is a quarter-rate
    // v_mad_u64_u32 chain per access (~200 of them in each kernel's code)
    if (v.small) return (long)__umul24(__umul24(__umul24((uint32_t)z, v.H) + (uint32_t)y, v.W) + (uint32_t)x, v.C);
    return (((long)z * v.H + y) * v.W + x) * (long)v.C;
}

// FWD = true: a[] caches voxel values (load on entry); FWD = false: a[] accumulates gradients (atomic on exit).
// `p` is the lane's channel pointer into the volume (forward) or its gradient (backward).
template <bool FWD>
__device__ __forceinline__ void px_touch(const PxVolume &v, float *__restrict__ p, int x, int y, int z, float &a) {
    if (FWD) {
        a = px_inside(v, x, y, z) ? p[px_offset(v, x, y, z)] : 0.f;
    } else {
        if (px_inside(v, x, y, z)) __hip_atomic_fetch_add(p + px_offset(v, x, y, z), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = 0.f;
    }
}

template <bool FWD>
__device__ __forceinline__ void px_move(PxWindow &w, const PxVolume &v, float *__restrict__ p, int nx, int ny, int nz) {
    const int n[3] = {nx, ny, nz};
    if (!w.open) {
        w.open = true;
#pragma unroll
        for (int a = 0; a < 3; a++) w.c[a] = n[a];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (FWD) px_touch<true>(v, p, nx + (t & 1), ny + ((t >> 1) & 1), nz + (t >> 2), w.a[t]);
            else w.a[t] = 0.f;
        }
        return;
    }
    const int d[3] = {nx - w.c[0], ny - w.c[1], nz - w.c[2]};
    if (d[0] == 0 && d[1] == 0 && d[2] == 0) return;
    const int moved = (d[0] != 0) + (d[1] != 0) + (d[2] != 0);
    bool slid = false;
    if (moved == 1) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
            if (d[ax] == 1 || d[ax] == -1) {
                slid = true;
                const int bit = 1 << ax;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    // d = +1: face bit=0 leaves, face bit=1 becomes face 0, the new face 1 enters
                    const bool leaving = d[ax] == 1 ? !(t & bit) : (t & bit) != 0;
                    if (!leaving) continue;
                    const int stay = t ^ bit;
                    if (!FWD) px_touch<false>(v, p, w.c[0] + (t & 1), w.c[1] + ((t >> 1) & 1), w.c[2] + (t >> 2), w.a[t]);
                    w.a[t] = w.a[stay];
                    if (FWD) px_touch<true>(v, p, n[0] + (stay & 1), n[1] + ((stay >> 1) & 1), n[2] + (stay >> 2), w.a[stay]);
                    else w.a[stay] = 0.f;
                }
            }
        }
    }
    if (!slid) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (!FWD) px_touch<false>(v, p, w.c[0] + (t & 1), w.c[1] + ((t >> 1) & 1), w.c[2] + (t >> 2), w.a[t]);
            else px_touch<true>(v, p, nx + (t & 1), ny + ((t >> 1) & 1), nz + (t >> 2), w.a[t]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) w.c[a] = n[a];
}

__device__ __forceinline__ void px_close(PxWindow &w, const PxVolume &v, float *__restrict__ p) {
    if (!w.open) return;
#pragma unroll
    for (int t = 0; t < 8; t++) px_touch<false>(v, p, w.c[0] + (t & 1), w.c[1] + ((t >> 1) & 1), w.c[2] + (t >> 2), w.a[t]);
    w.open = false;
}

// SH_k(d) for the lane's own k (k = (channel - 1) mod DEG^2), bands l < DEG
template <int DEG>
__device__ __forceinline__ float px_sh_k(float dx, float dy, float dz, int k) {
    float o[DEG * DEG];
    pvd_sh_basis<DEG, false>(dx, dy, dz, [&](int i, float val) { o[i] = val; }, [&](int, float, float, float) {});
    float r = o[0];
#pragma unroll
    for (int i = 1; i < DEG * DEG; i++) r = (k == i) ? o[i] : r;
    return r;
}
template <int DEG>
__device__ __forceinline__ float px_sh_of_lane(const float *__restrict__ dirs, size_t m, int k) {
    return px_sh_k<DEG>(dirs[3 * m], dirs[3 * m + 1], dirs[3 * m + 2], k);
}

__device__ __forceinline__ float px_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct PxStream {
    uint32_t s0, s1, ch;
    bool lane_on;
};
__device__ __forceinline__ PxStream px_stream(uint32_t M, uint32_t chunk, uint32_t C) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kPxBlock + threadIdx.x) >> 6;
    const uint32_t stream = wave * 2 + (lane >> 5);
    PxStream st;
    st.ch = lane & 31u;
    st.s0 = min(M, stream * chunk);
    st.s1 = min(M, st.s0 + chunk);
    st.lane_on = st.ch < C;
    return st;
}

// reference: compute_plenoxel_fea + NeRFNetwork.forward "tensors" branch, network.py:311-322, 383-409
template <int DEG>
__global__ void __launch_bounds__(kPxBlock) k_plenoxel_fwd(const float *__restrict__ xyz, const float *__restrict__ dirs, uint32_t M,
                                                           uint32_t chunk, PxVolume v, float *__restrict__ feat,
                                                           float *__restrict__ h0_raw, float *__restrict__ sigma_l,
                                                           float *__restrict__ sigma, float *__restrict__ rgb) {
    constexpr int K2 = DEG * DEG;
    const PxStream st = px_stream(M, chunk, v.C);
    if (!st.lane_on) return;  // lanes >= C of each half idle (shuffles below only read active lanes)
    const int k = st.ch == 0 ? 0 : (int)(st.ch - 1) % K2;
    float *__restrict__ p = const_cast<float *>(v.vol) + st.ch;
    PxWindow w;
    w.open = false;
    for (uint32_t m = st.s0; m < st.s1; m++) {
        const PxSample s = px_locate(xyz, m, v);
        px_move<true>(w, v, p, s.ax[0].i0, s.ax[1].i0, s.ax[2].i0);
        float h = 0.f;
#pragma unroll
        for (int t = 0; t < 8; t++) h += w.a[t] * s.w[t];
        if (feat) feat[(size_t)m * v.C + st.ch] = h;
        if (!dirs) continue;
        // colour: lanes 1 + c*K2 + k multiply by SH_k, the K2 lanes of a colour are summed onto lane k == 0
        float part = st.ch == 0 ? 0.f : h * px_sh_of_lane<DEG>(dirs, m, k);
#pragma unroll
        for (int off = 1; off < K2; off <<= 1) {
            const float up = __shfl_down(part, off, 64);
            if (k + off < K2) part += up;
        }
        if (st.ch == 0) {
            const float cl = fminf(v.clip_max, fmaxf(v.clip_min, h));
            h0_raw[m] = h;
            sigma_l[m] = cl;
            sigma[m] = expf(cl);  // trunc_exp forward (tools/activation.py:11-13)
        } else if (k == 0) {
            rgb[(size_t)m * 3 + (st.ch - 1) / K2] = px_sigmoid(part);
        }
    }
}

// d loss / d volume, accumulated (+=) into a channels-last gradient buffer.
//   g_sigma, g_sigma_l, g_rgb: gradients of the forward's sigma / sigma_l / rgb (NULL = zero)
//   g_feat: gradient of the raw feature output (NULL = zero)
template <int DEG>
__global__ void __launch_bounds__(kPxBlock) k_plenoxel_bwd(const float *__restrict__ xyz, const float *__restrict__ dirs, uint32_t M,
                                                           uint32_t chunk, PxVolume v, const float *__restrict__ h0_raw,
                                                           const float *__restrict__ rgb,
                                                           const float *__restrict__ g_feat, const float *__restrict__ g_sigma,
                                                           const float *__restrict__ g_sigma_l, const float *__restrict__ g_rgb,
                                                           float *__restrict__ g_vol) {
    constexpr int K2 = DEG * DEG;
    const PxStream st = px_stream(M, chunk, v.C);
    if (!st.lane_on) return;
    const int k = st.ch == 0 ? 0 : (int)(st.ch - 1) % K2;
    const int c = st.ch == 0 ? 0 : (int)(st.ch - 1) / K2;
    float *__restrict__ p = g_vol + st.ch;
    PxWindow w;
    w.open = false;
    for (uint32_t m = st.s0; m < st.s1; m++) {
        float g = g_feat ? g_feat[(size_t)m * v.C + st.ch] : 0.f;
        if (dirs) {
            if (st.ch == 0) {
                const float raw = h0_raw[m];
                float gh = g_sigma_l ? g_sigma_l[m] : 0.f;
                // trunc_exp backward: g * exp(clamp(x, -12, 12)) (tools/activation.py:15-18); x is already inside the clip range
                if (g_sigma) gh += g_sigma[m] * expf(fminf(12.f, fmaxf(-12.f, fminf(v.clip_max, fmaxf(v.clip_min, raw)))));
                g += (raw >= v.clip_min && raw <= v.clip_max) ? gh : 0.f;  // clamp passes the gradient inside [min, max]
            } else if (g_rgb) {
                const float y = rgb[(size_t)m * 3 + c];
                g += (g_rgb[(size_t)m * 3 + c] * ((1.0f - y) * y)) * px_sh_of_lane<DEG>(dirs, m, k);
            }
        }
        const PxSample s = px_locate(xyz, m, v);
        px_move<false>(w, v, p, s.ax[0].i0, s.ax[1].i0, s.ax[2].i0);
#pragma unroll
        for (int t = 0; t < 8; t++) w.a[t] += g * s.w[t];
    }
    px_close(w, v, p);
}

// ---- the eval branch's round loop of a frozen Plenoxel model as ONE persistent launch (pvd_infer_image_plenoxel).
// The slot machinery is infer_persistent_loop's; the shading of a round's rows: a half-wave (lane = channel, C <= 32) per row, the
// eight corners of the row's cell requested unconditionally (clamped address, zero selected for corners outside the volume: the
// rows of a round are 1-3 samples of each of ~60 rays, there is no run for k_plenoxel_fwd's register window to follow), then
// k_plenoxel_fwd's own expressions in its order -- h = sum_t a_t w_t, the SH products summed over a colour's lanes by the same
// shuffle tree, clamp / exp / sigmoid -- so per row the values are pvd_plenoxel_forward's and per ray the round loop's, bit for bit
// (tests/test_hip_infer_rounds.py).  No weights, 7 KB of LDS, 163 VGPRs: three workgroups per CU.
#ifndef PVD_INFER_PX_UNROLL
#define PVD_INFER_PX_UNROLL 2  // 163 VGPRs, three workgroups per CU: 6.65 ms per 800 x 800 view (1: 9.12, 4: 8.8 at two per CU)
#endif
constexpr uint32_t kPxInfRays = 64, kPxInfRows = 128;
template <int DEG>
__global__ void __launch_bounds__(kPxBlock) k_infer_px_persistent(PxVolume v, InferImageArgs q) {
    constexpr int K2 = DEG * DEG;
    __shared__ float tile_mem[InferTile::floats<kPxInfRows, kPxBlock>()];
    InferTile T;
    T.carve<kPxInfRows, kPxBlock>(tile_mem);
    const uint32_t lane = threadIdx.x & 63u, ch = lane & 31u, hw = threadIdx.x >> 5;  // 8 half-waves per workgroup
    const bool lane_on = ch < v.C;
    const int k = ch == 0 ? 0 : (int)(ch - 1) % K2;
    const float *__restrict__ p = v.vol + (lane_on ? ch : 0u);
    __syncthreads();
    infer_persistent_loop<kPxInfRays, kPxInfRows, kPxBlock>(q, T, [&](uint32_t rows) {
        // U rows per half-wave and pass: the 8 U corner loads are all requested before the first row is formed (one memory round
        // trip per pass instead of one per row; the trip count is uniform -- the shuffles are executed by whole waves)
        constexpr uint32_t U = PVD_INFER_PX_UNROLL, kHalves = kPxBlock / 32;
        for (uint32_t r0 = 0; r0 < rows; r0 += U * kHalves) {
            PxAxis ax[U][3];
            float a[U][8];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t rw = min(r0 + u * kHalves + hw, rows - 1u);  // (a row past the last repeats it: nothing of it is stored)
                const PxSample s = px_locate(T.pos, rw, v);
#pragma unroll
                for (int c = 0; c < 3; c++) ax[u][c] = s.ax[c];
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int x = s.ax[0].i0 + (t & 1), y = s.ax[1].i0 + ((t >> 1) & 1), z = s.ax[2].i0 + (t >> 2);
                    a[u][t] = p[px_offset(v, min(max(x, 0), (int)v.W - 1), min(max(y, 0), (int)v.H - 1), min(max(z, 0), (int)v.D - 1))];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const bool live = r0 + u * kHalves + hw < rows;
                const uint32_t rw = min(r0 + u * kHalves + hw, rows - 1u);
                float h = 0.f;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int x = ax[u][0].i0 + (t & 1), y = ax[u][1].i0 + ((t >> 1) & 1), z = ax[u][2].i0 + (t >> 2);
                    // px_locate's weights, k_plenoxel_fwd's sum: h += a_t * w_t in tap order, corners outside the volume count as zero
                    const float w = (((t & 1) ? ax[u][0].w1 : ax[u][0].w0) * ((t & 2) ? ax[u][1].w1 : ax[u][1].w0)) * ((t & 4) ? ax[u][2].w1 : ax[u][2].w0);
                    h += (px_inside(v, x, y, z) ? a[u][t] : 0.f) * w;
                }
                const uint32_t slot = T.row_slot[rw];
                float part = ch == 0 ? 0.f : h * px_sh_k<DEG>(T.sdir[3 * slot], T.sdir[3 * slot + 1], T.sdir[3 * slot + 2], k);
#pragma unroll
                for (int off = 1; off < K2; off <<= 1) {
                    const float up = __shfl_down(part, off, 64);
                    if (k + off < K2) part += up;
                }
                if (live && lane_on) {
                    if (ch == 0) T.sig[rw] = expf(fminf(v.clip_max, fmaxf(v.clip_min, h)));
                    else if (k == 0) T.rgb[3 * rw + (ch - 1) / K2] = px_sigmoid(part);
                }
            }
        }
    });
}

static uint32_t px_chunk(uint32_t M) {
    uint32_t chunk = 16;
    while (chunk < 64 && (uint64_t)M / chunk > 2u * 256u * 16u) chunk <<= 1;
    return chunk;
}

static int px_fill(PxVolume &v, const float *vol, const uint32_t *dims, uint32_t C, const float *aabb, float cmin, float cmax) {
    if (!vol || !dims || !aabb) return PVD_ERR_INVALID;
    v.vol = vol;
    v.D = dims[0]; v.H = dims[1]; v.W = dims[2]; v.C = C;
    if (v.D < 1 || v.H < 1 || v.W < 1) return PVD_ERR_INVALID;
    const uint64_t voxels = (uint64_t)v.D * v.H * v.W;
    v.small = (voxels < (1ull << 24) && voxels * C < (1ull << 32) && C < (1u << 24)) ? 1u : 0u;
    for (int a = 0; a < 3; a++) { v.lo[a] = aabb[a]; v.extent[a] = aabb[a + 3] - aabb[a]; }
    v.clip_min = cmin; v.clip_max = cmax;
    return PVD_OK;
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_plenoxel_forward(const float *xyz, const float *dirs, uint32_t M, const float *aabb_host, const float *volume,
                         const uint32_t *dims_host, uint32_t C, uint32_t degree, float clip_min, float clip_max, float *feat,
                         float *h0_raw, float *sigma_l, float *sigma, float *rgb, pvd_stream_t stream) {
    if (degree < 1 || degree > 3 || C != 3 * degree * degree + 1) return PVD_ERR_UNSUPPORTED;  // C <= 32 lanes
    if (M == 0) return PVD_OK;
    if (!xyz || (!feat && !dirs) || (dirs && (!h0_raw || !sigma_l || !sigma || !rgb))) return PVD_ERR_INVALID;
    PxVolume v;
    const int rc = px_fill(v, volume, dims_host, C, aabb_host, clip_min, clip_max);
    if (rc != PVD_OK) return rc;
    const uint32_t chunk = px_chunk(M);
    const dim3 grid(div_up(div_up(M, 2 * chunk) * 64u, kPxBlock)), block(kPxBlock);
    hipStream_t s = (hipStream_t)stream;
    switch (degree) {
        case 1: hipLaunchKernelGGL((k_plenoxel_fwd<1>), grid, block, 0, s, xyz, dirs, M, chunk, v, feat, h0_raw, sigma_l, sigma, rgb); break;
        case 2: hipLaunchKernelGGL((k_plenoxel_fwd<2>), grid, block, 0, s, xyz, dirs, M, chunk, v, feat, h0_raw, sigma_l, sigma, rgb); break;
        default: hipLaunchKernelGGL((k_plenoxel_fwd<3>), grid, block, 0, s, xyz, dirs, M, chunk, v, feat, h0_raw, sigma_l, sigma, rgb); break;
    }
    return check_launch();
}

int pvd_plenoxel_backward(const float *xyz, const float *dirs, uint32_t M, const float *aabb_host, const uint32_t *dims_host, uint32_t C,
                          uint32_t degree, float clip_min, float clip_max, const float *h0_raw, const float *rgb,
                          const float *g_feat, const float *g_sigma, const float *g_sigma_l, const float *g_rgb, float *grad_volume,
                          pvd_stream_t stream) {
    if (degree < 1 || degree > 3 || C != 3 * degree * degree + 1) return PVD_ERR_UNSUPPORTED;
    if (M == 0) return PVD_OK;
    if (!xyz || !grad_volume || (dirs && (!h0_raw || !rgb)) || (!dirs && !g_feat)) return PVD_ERR_INVALID;
    PxVolume v;
    const int rc = px_fill(v, grad_volume, dims_host, C, aabb_host, clip_min, clip_max);
    if (rc != PVD_OK) return rc;
    const uint32_t chunk = px_chunk(M);
    const dim3 grid(div_up(div_up(M, 2 * chunk) * 64u, kPxBlock)), block(kPxBlock);
    hipStream_t s = (hipStream_t)stream;
    switch (degree) {
        case 1: hipLaunchKernelGGL((k_plenoxel_bwd<1>), grid, block, 0, s, xyz, dirs, M, chunk, v, h0_raw, rgb, g_feat, g_sigma, g_sigma_l, g_rgb, grad_volume); break;
        case 2: hipLaunchKernelGGL((k_plenoxel_bwd<2>), grid, block, 0, s, xyz, dirs, M, chunk, v, h0_raw, rgb, g_feat, g_sigma, g_sigma_l, g_rgb, grad_volume); break;
        default: hipLaunchKernelGGL((k_plenoxel_bwd<3>), grid, block, 0, s, xyz, dirs, M, chunk, v, h0_raw, rgb, g_feat, g_sigma, g_sigma_l, g_rgb, grad_volume); break;
    }
    return check_launch();
}

int pvd_infer_image_plenoxel(const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N, const uint8_t *bitfield,
                             float bound, float dt_gamma, uint32_t max_steps, uint32_t cascade, uint32_t grid_size, float sigma_scale,
                             const float *aabb_host, const float *volume, const uint32_t *dims_host, uint32_t C, uint32_t degree, float clip_min,
                             float clip_max, int32_t *workspace, float *weights_sum, float *depth, float *image_out, pvd_stream_t stream) {
    if (degree < 1 || degree > 3 || C != 3 * degree * degree + 1) return PVD_ERR_UNSUPPORTED;
    if (N == 0) return PVD_OK;
    if (!rays_o || !rays_d || !nears || !fars || !bitfield || !workspace || !weights_sum || !depth || !image_out) return PVD_ERR_INVALID;
    if (max_steps == 0 || cascade == 0 || grid_size == 0) return PVD_ERR_INVALID;
    PxVolume v;
    const int rc = px_fill(v, volume, dims_host, C, aabb_host, clip_min, clip_max);
    if (rc != PVD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    InferImageArgs q;
    const int prc = infer_prepare(q, rays_o, rays_d, nears, fars, N, bitfield, bound, dt_gamma, max_steps, cascade, grid_size, sigma_scale, workspace,
                                  weights_sum, depth, image_out, s);
    if (prc != PVD_OK) return prc;
    uint32_t blocks = div_up(N, kPxInfRays);
    if (blocks > 768u) blocks = 768u;  // persistent: 163 VGPRs = three workgroups per CU
    switch (degree) {
        case 1: hipLaunchKernelGGL((k_infer_px_persistent<1>), dim3(blocks), dim3(kPxBlock), 0, s, v, q); break;
        case 2: hipLaunchKernelGGL((k_infer_px_persistent<2>), dim3(blocks), dim3(kPxBlock), 0, s, v, q); break;
        default: hipLaunchKernelGGL((k_infer_px_persistent<3>), dim3(blocks), dim3(kPxBlock), 0, s, v, q); break;
    }
    return check_launch();
}

}  // extern "C"
