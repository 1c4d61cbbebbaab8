// vm_lookup.h -- what the two users of the TensoRF "VM" (plane x line) tables share: vmencoder.hip (pvd_vm_forward / pvd_vm_backward,
// the register-window walks over runs of samples) and fusedhead.hip (pvd_infer_image_vm: the lookup of a persistent render's LDS tile).
// Table layout, sampling rule and arithmetic: see vmencoder.hip's header (distill_mutual/network.py:193-309).
#pragma once

#include "pvd_device.h"
#include "grid_lookup.h"

namespace pvd {

constexpr uint32_t kRs = 16;  // sigma_rank (network.py:79)
constexpr uint32_t kRc = 48;  // color_rank (network.py:80)

struct VmTables {
    const float *mat[2][3];  // [0] = sigma, [1] = colour; channels-last [H][W][R]
    const float *vec[2][3];  // channels-last [L][R]
    uint32_t ms[2], vs[2];   // elements between consecutive texels of a plane / taps of a line (>= R: tables may interleave)
    uint32_t W[3], H[3], L[3];
    float lo[3], inv_extent2[3];  // x_n = 2*(x-lo)/(hi-lo) - 1, kept as (2*(x-lo)) / (hi-lo) - 1
    float extent[3];
};

struct VmGrads {
    float *mat[2][3];
    float *vec[2][3];
};

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

// texel index -> element offset at texel stride R: a 24-bit multiply (fill_tables refuses tables it does not reach)
__device__ __forceinline__ uint32_t toff(int t, uint32_t R) { return __umul24((uint32_t)t, R); }

// ---- one sample, every texel of its footprint requested unconditionally (clamped address, zero selected afterwards): the form
// the persistent render uses for the rows of its LDS tile -- rows of one round belong to up to 64 different rays, there is no run
// for a register window to follow.  Same values, same operation order as the walks (plane_value / line_value): bit-identical.
struct SampleCtl {   // lane j's view of row j of a batch: tap indices (raw, clamped), weights, in-range bits
    int c0[3], c1[3];
    float w0[3], w1[3];
    uint32_t in;     // bit 2 a + k: tap k of axis a in range
};
__device__ __forceinline__ SampleCtl sample_ctl(const float (&xn)[3], const VmTables &tb) {
    const int size[3] = {(int)tb.W[0], (int)tb.H[0], (int)tb.L[0]};  // res[0], res[1], res[2]
    SampleCtl r;
    r.in = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const Tap1 t = tap1(xn[a], (uint32_t)size[a]);
        r.w0[a] = t.w0; r.w1[a] = t.w1;
        r.c0[a] = min(max(t.i0, 0), size[a] - 1);
        r.c1[a] = min(max(t.i0 + 1, 0), size[a] - 1);
        r.in |= (t.in0 ? 1u : 0u) << (2 * a) | (t.in1 ? 1u : 0u) << (2 * a + 1);
    }
    return r;
}
struct Taps6 {
    float p[4], l[2];
};
__device__ __forceinline__ int rl_i(int v, uint32_t lane) { return __builtin_amdgcn_readlane(v, (int)lane); }
__device__ __forceinline__ float rl_f(float v, uint32_t lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)lane));
}
// factor set I of the sample in lane j of `c` (uniform j): the six loads of the calling lane's channel (mat / vec: the lane's table pointers)
template <int I>
__device__ __forceinline__ void issue6(const SampleCtl &c, uint32_t j, const float *__restrict__ mat, const float *__restrict__ vec, uint32_t R,
                                       uint32_t Rv, int W, Taps6 &o) {
    constexpr int AX = kM0[I], AY = kM1[I], AL = kV[I];
    const int x0 = rl_i(c.c0[AX], j), x1 = rl_i(c.c1[AX], j), y0 = rl_i(c.c0[AY], j) * W, y1 = rl_i(c.c1[AY], j) * W;
    const int l0 = rl_i(c.c0[AL], j), l1 = rl_i(c.c1[AL], j);
    o.p[0] = mat[toff(y0 + x0, R)];
    o.p[1] = mat[toff(y0 + x1, R)];
    o.p[2] = mat[toff(y1 + x0, R)];
    o.p[3] = mat[toff(y1 + x1, R)];
    o.l[0] = vec[toff(l0, Rv)];
    o.l[1] = vec[toff(l1, Rv)];
}
// plane value x line value of factor set I (zero padding; grid_sample's tap order)
template <int I>
__device__ __forceinline__ float finish6(const SampleCtl &c, uint32_t j, const Taps6 &t) {
    constexpr int AX = kM0[I], AY = kM1[I], AL = kV[I];
    const uint32_t in = (uint32_t)rl_i((int)c.in, j);
    const bool ix0 = (in >> (2 * AX)) & 1u, ix1 = (in >> (2 * AX + 1)) & 1u;
    const bool iy0 = (in >> (2 * AY)) & 1u, iy1 = (in >> (2 * AY + 1)) & 1u;
    const bool il0 = (in >> (2 * AL)) & 1u, il1 = (in >> (2 * AL + 1)) & 1u;
    const float wx0 = rl_f(c.w0[AX], j), wx1 = rl_f(c.w1[AX], j), wy0 = rl_f(c.w0[AY], j), wy1 = rl_f(c.w1[AY], j);
    const float wl0 = rl_f(c.w0[AL], j), wl1 = rl_f(c.w1[AL], j);
    const float v0 = (ix0 && iy0) ? t.p[0] : 0.f, v1 = (ix1 && iy0) ? t.p[1] : 0.f;
    const float v2 = (ix0 && iy1) ? t.p[2] : 0.f, v3 = (ix1 && iy1) ? t.p[3] : 0.f;
    float pv = v0 * (wx0 * wy0);
    pv += v1 * (wx1 * wy0);
    pv += v2 * (wx0 * wy1);
    pv += v3 * (wx1 * wy1);
    float lv = (il0 ? t.l[0] : 0.f) * wl0;
    lv += (il1 ? t.l[1] : 0.f) * wl1;
    return pv * lv;
}

// host: tables_host[12] / res / aabb / strides -> VmTables (pvd_vm_forward's contract, include/pvd_hip.h)
static inline int fill_tables(VmTables &tb, const void *const *tables, const uint32_t *res, const float *aabb, const uint32_t *stride) {
    // reference shapes (network.py:199-212): mat_i [1,R,res[m1],res[m0]], vec_i [1,R,res[vec_id],1]
    for (int i = 0; i < 3; i++) {
        tb.W[i] = res[kM0[i]];
        tb.H[i] = res[kM1[i]];
        tb.L[i] = res[kV[i]];
        if (tb.W[i] < 1 || tb.H[i] < 1 || tb.L[i] < 1) return PVD_ERR_INVALID;
        if ((uint64_t)tb.W[i] * tb.H[i] >= (1ull << 24) || tb.L[i] >= (1u << 24)) return PVD_ERR_UNSUPPORTED;  // toff(): 24-bit texel indices
        for (int k = 0; k < 2; k++) {
            tb.mat[k][i] = (const float *)tables[k * 6 + i];
            tb.vec[k][i] = (const float *)tables[k * 6 + 3 + i];
            if (!tb.mat[k][i] || !tb.vec[k][i]) return PVD_ERR_INVALID;
        }
        tb.lo[i] = aabb[i];
        tb.extent[i] = aabb[i + 3] - aabb[i];
        tb.inv_extent2[i] = 0.f;
    }
    // texel strides: {sigma planes, sigma lines, colour planes, colour lines}; NULL = densely packed channels-last tables
    const uint32_t dense[4] = {kRs, kRs, kRc, kRc};
    if (!stride) stride = dense;
    for (int k = 0; k < 2; k++) {
        tb.ms[k] = stride[2 * k];
        tb.vs[k] = stride[2 * k + 1];
        if (tb.ms[k] < dense[2 * k] || tb.vs[k] < dense[2 * k] || tb.ms[k] > 4096u || tb.vs[k] > 4096u) return PVD_ERR_INVALID;
        // toff() returns a 32-bit ELEMENT offset: texels x stride of every plane and line must fit
        for (int i = 0; i < 3; i++)
            if ((uint64_t)tb.W[i] * tb.H[i] * tb.ms[k] >= (1ull << 32) || (uint64_t)tb.L[i] * tb.vs[k] >= (1ull << 32)) return PVD_ERR_UNSUPPORTED;
    }
    return PVD_OK;
}


}  // namespace pvd
