// fusedhead.hip -- the sigma / colour head as ONE MFMA kernel per direction (gfx950, MI355X).
//
// Replaces, per sample, the chain the reference runs as ~5 cuBLAS GEMMs + ~15 elementwise launches
// (distill_mutual/network.py:335-437 under autocast):
//   hash model : enc[28] -> sigma_net (28->64 relu ->16) -> clamp h0 -> feature_sigma_color[16]
//   VM model   : prod[144] -> basis_mat (144->15) -> clamp ; sigma_feat clamp -> feature_sigma_color[16]
//   both       : sigma = exp(h0);  rgb = sigmoid(color_net([SH_4(d) (16), h1..15] : 31->64 relu->64 relu->3))
//
// MFMA formulation.  v_mfma_f32_16x16x16_f16 computes D = A.B + C with per-lane fragments
//   A[16x16]: lane l holds A[row = l&15][k = 4*(l>>4)+j]      (4 halfs)
//   B[16x16]: lane l holds B[k = 4*(l>>4)+j][col = l&15]      (4 halfs)
//   D[16x16]: lane l holds D[row = 4*(l>>4)+j][col = l&15]    (4 floats)
// We compute Y^T = W . X^T: A = a 16x16 tile of the weight matrix (rows = output neurons), B = 16
// features x 16 SAMPLES, so a D tile holds 4 consecutive neurons of one sample per lane -- which is
// exactly the B fragment the NEXT layer needs for its k-step.  Activations therefore never leave
// registers between layers: no LDS round trip, no shuffles.  A wavefront owns 16 samples at a time.
// Weights are read from the fp32 masters, rounded to f16 exactly like autocast's cast, and kept in LDS
// (<= 22 KB per workgroup) with +4 halfs of row padding (conflict-free 8-byte fragment reads).
// Layer outputs are rounded to f16 between layers (what a chain of autocast Linear layers does).
//
// Index bookkeeping that keeps the layers lane-aligned:
//   * the colour layer's input is [SH(16) | h1..h15]; we feed k-step 1 with the whole 16-row feature tile
//     (h0 = log-sigma included) and give column 16 a ZERO weight, columns 17..31 the weights of h1..h15;
//   * the VM basis layer (144->15) is stored with a zero ROW 0, so its output tile is [0, cf0..cf14] and
//     row 0 is then replaced by the clamped sigma feature: same tile shape as the hash model's h.
//
// The backward kernel (student) recomputes the forward in registers, chains W^T . dY the same way, and
// forms the weight gradients dW = dY . X^T with MFMAs whose operands are 16x16 transposes of the
// register tiles (through a 512-byte LDS scratch per wave); dW tiles live in accumulators for the whole
// launch and leave as one partial per wave, summed by a second tiny kernel straight into the gradients.
#include "dda.h"
#include "infer_persistent.h"
#include "grid_lookup.h"
#include "head_dw_reduce.h"
#include "head_pack.h"
#include "vm_lookup.h"

#include <stdlib.h>

namespace pvd {

#include "sh_basis.inc"

typedef half_t h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kHeadBlock = 256;  // 4 independent waves sharing the weights in LDS

__device__ __forceinline__ f4 mfma(h4 a, h4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ h4 to_h4(f4 v) {
    h4 r;
    r.x = (half_t)v.x; r.y = (half_t)v.y; r.z = (half_t)v.z; r.w = (half_t)v.w;
    return r;
}
__device__ __forceinline__ h4 relu_h4(h4 v) {
    const half_t z = (half_t)0.0f;
    h4 r;
    r.x = v.x > z ? v.x : z; r.y = v.y > z ? v.y : z; r.z = v.z > z ? v.z : z; r.w = v.w > z ? v.w : z;
    return r;
}

// ---- weights in LDS: row-major [rows_pad][cols_pad + kPad] halfs, built from fp32 masters
struct LdsMat {
    half_t *p;
    int stride;  // cols_pad + kPad
    __device__ __forceinline__ h4 afrag(int tile, int kstep, uint32_t lane) const {  // A fragment of W
        return *reinterpret_cast<const h4 *>(p + (16 * tile + (lane & 15)) * stride + 16 * kstep + 4 * (lane >> 4));
    }
};

// dst[r][c] = (r in [r0, r0+rows) and mapped col valid) ? half(src[r-r0][col_map(c)]) : 0
// col_shift: columns >= col_split of dst take src column (c - 1)   (the colour layer's zero column 16)
// f32 weight [rows][cols] in HBM -> f16 [rows_pad][cols_pad] in LDS (TRANSPOSED: stored [cols_pad][rows_pad]), with
// `row0` zero rows in front and an optional zero column inserted at `col_split`.  The global loads of kUnroll
// elements are issued before any of them is consumed: the weights are the kernel's fixed cost (every workgroup
// reads all of them), and a load -> convert -> store chain per element made that cost 60 us in the backward.
template <bool TRANSPOSED>
__device__ __forceinline__ void load_weight_impl(LdsMat m, const float *__restrict__ src, int rows, int cols, int rows_pad, int cols_pad,
                                                 int row0, int col_split, uint32_t tid, uint32_t nthreads) {
    constexpr int kUnroll = 8;
    const int total = rows_pad * cols_pad;
    for (int base = tid; base < total; base += nthreads * kUnroll) {
        float v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int i = base + u * (int)nthreads;
            const int r = i / cols_pad, c = i - r * cols_pad;
            const int sr = r - row0;
            int sc = c;
            bool ok = i < total && sr >= 0 && sr < rows;
            if (col_split >= 0) {
                if (c == col_split) ok = false;
                else if (c > col_split) sc = c - 1;
            }
            v[u] = (ok && sc < cols) ? src[(size_t)sr * cols + sc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const int i = base + u * (int)nthreads;
            if (i < total) {
                const int r = i / cols_pad, c = i - r * cols_pad;
                m.p[TRANSPOSED ? c * m.stride + r : r * m.stride + c] = (half_t)v[u];
            }
        }
    }
}
__device__ __forceinline__ void load_weight(LdsMat m, const float *__restrict__ src, int rows, int cols, int rows_pad, int cols_pad,
                                            int row0, int col_split, uint32_t tid, uint32_t nthreads) {
    load_weight_impl<false>(m, src, rows, cols, rows_pad, cols_pad, row0, col_split, tid, nthreads);
}
__device__ __forceinline__ void load_weight_T(LdsMat m, const float *__restrict__ src, int rows, int cols, int rows_pad, int cols_pad,
                                              int row0, int col_split, uint32_t tid, uint32_t nthreads) {
    load_weight_impl<true>(m, src, rows, cols, rows_pad, cols_pad, row0, col_split, tid, nthreads);
}

struct HeadArgs {
    // inputs
    const half_t *x0;        // hash: encoder output [14][M][2] f16 (level-major);  VM: products [M][144] f16
    const float *sigma_raw;  // VM: [M] raw sigma feature
    const float *dirs;       // [M][3]
    uint32_t M;
    // fp32 master weights, row-major [out][in]
    const float *Wa1;  // hash: sigma_net.0 [64][28]   VM: basis_mat [15][144]
    const float *Wa2;  // hash: sigma_net.1 [16][64]   VM: unused
    const float *Wc1;  // color_net.0 [64][31]
    const float *Wc2;  // color_net.1 [64][64]
    const float *Wc3;  // color_net.2 [3][64]
    const half_t *image;  // optional: the weights pre-packed by pvd_head_pack_weights (LDS image), else NULL
    float clip_sigma_min, clip_feat_min, clip_max;
    const int32_t *rows_dev;  // optional DEVICE row count: only the first min(M, *rows_dev) rows are processed (inference rounds)
    // outputs
    float *sigma;   // [M]
    float *rgb;     // [M][3]
    float *feat16;  // [M][16]  feature_sigma_color
};


// the 4 SH values k = 4*hi + j of a degree-4 basis for direction d (all 16 computed, 4 kept)
__device__ __forceinline__ h4 sh_frag(float x, float y, float z, uint32_t hi) {
    float o[16];
    pvd_sh_basis<4, false>(x, y, z, [&](int i, float v) { o[i] = v; }, [&](int, float, float, float) {});
    h4 r;
    r.x = (half_t)(hi == 0 ? o[0] : hi == 1 ? o[4] : hi == 2 ? o[8] : o[12]);
    r.y = (half_t)(hi == 0 ? o[1] : hi == 1 ? o[5] : hi == 2 ? o[9] : o[13]);
    r.z = (half_t)(hi == 0 ? o[2] : hi == 1 ? o[6] : hi == 2 ? o[10] : o[14]);
    r.w = (half_t)(hi == 0 ? o[3] : hi == 1 ? o[7] : hi == 2 ? o[11] : o[15]);
    return r;
}

// everything the forward produces for one 16-sample tile, in registers
struct TileFwd {
    f4 F;        // feature tile rows 4hi..4hi+3 (row 0 = clamped log-sigma), fp32 copy for the output
    h4 Fh;       // same as the next layer's B fragment
    f4 raw;      // VM: pre-clamp basis output (for the clamp's backward mask)
    float sig_raw;
    h4 sh;       // SH fragment
    h4 H1[4], H2[4];
    f4 out;      // colour layer 3 pre-activation (rows 0..2 valid)
    h4 X[2], Ha[4];  // hash: encoder-feature fragments and sigma_net hidden layer (kept for the backward)
};

template <int KIND>
struct HeadLds {
    LdsMat Wa1, Wa2, Wc1, Wc2, Wc3;
    static constexpr int a1_rows = KIND == KIND_HASH ? 64 : 16, a1_cols = KIND == KIND_HASH ? 32 : 144;
    static constexpr int halfs = a1_rows * (a1_cols + kPad) + (KIND == KIND_HASH ? 16 * (64 + kPad) : 0) + 64 * (32 + kPad) +
                                 64 * (64 + kPad) + 16 * (64 + kPad);
    __device__ __forceinline__ void carve(half_t *base) {
        Wa1 = {base, a1_cols + kPad}; base += a1_rows * (a1_cols + kPad);
        if (KIND == KIND_HASH) { Wa2 = {base, 64 + kPad}; base += 16 * (64 + kPad); }
        else Wa2 = {base, 0};
        Wc1 = {base, 32 + kPad}; base += 64 * (32 + kPad);
        Wc2 = {base, 64 + kPad}; base += 64 * (64 + kPad);
        Wc3 = {base, 64 + kPad};
    }
    __device__ __forceinline__ void load(const HeadArgs &a, uint32_t tid, uint32_t n) {
        if (KIND == KIND_HASH) {
            load_weight(Wa1, a.Wa1, 64, 28, 64, 32, 0, -1, tid, n);
            load_weight(Wa2, a.Wa2, 16, 64, 16, 64, 0, -1, tid, n);
        } else {
            load_weight(Wa1, a.Wa1, 15, 144, 16, 144, /*row0=*/1, -1, tid, n);  // zero row 0
        }
        load_weight(Wc1, a.Wc1, 64, 31, 64, 32, 0, /*zero column*/ 16, tid, n);
        load_weight(Wc2, a.Wc2, 64, 64, 64, 64, 0, -1, tid, n);
        load_weight(Wc3, a.Wc3, 3, 64, 16, 64, 0, -1, tid, n);
    }
};

// transposed copies for dX = W^T . dY (A fragments must be contiguous along the contracted index); the backward's
// LDS holds HeadLds followed by this
template <int KIND>
struct HeadLdsT {
    LdsMat Wa1T, Wa2T, Wc1T, Wc2T, Wc3T;
    static constexpr int halfs = (KIND == KIND_VM ? 144 * (16 + kPad) : 32 * (64 + kPad) + 64 * (16 + kPad)) + 32 * (64 + kPad) +
                                 64 * (64 + kPad) + 64 * (16 + kPad);
    __device__ __forceinline__ void carve(half_t *p) {
        if (KIND == KIND_VM) {
            Wa1T = {p, 16 + kPad}; p += 144 * (16 + kPad);  // basis^T [144][16]
            Wa2T = {p, 0};
        } else {
            Wa1T = {p, 64 + kPad}; p += 32 * (64 + kPad);   // sigma_net.0^T [32][64]
            Wa2T = {p, 16 + kPad}; p += 64 * (16 + kPad);   // sigma_net.1^T [64][16]
        }
        Wc1T = {p, 64 + kPad}; p += 32 * (64 + kPad);        // [32][64]
        Wc2T = {p, 64 + kPad}; p += 64 * (64 + kPad);        // [64][64]
        Wc3T = {p, 16 + kPad};                               // [64][16]
    }
    __device__ __forceinline__ void load(const HeadArgs &a, uint32_t tid, uint32_t n) {
        if (KIND == KIND_VM) {
            load_weight_T(Wa1T, a.Wa1, 15, 144, 16, 144, 1, -1, tid, n);
        } else {
            load_weight_T(Wa1T, a.Wa1, 64, 28, 64, 32, 0, -1, tid, n);
            load_weight_T(Wa2T, a.Wa2, 16, 64, 16, 64, 0, -1, tid, n);
        }
        load_weight_T(Wc1T, a.Wc1, 64, 31, 64, 32, 0, 16, tid, n);
        load_weight_T(Wc2T, a.Wc2, 64, 64, 64, 64, 0, -1, tid, n);
        load_weight_T(Wc3T, a.Wc3, 3, 64, 16, 64, 0, -1, tid, n);
    }
};

// every workgroup needs all the weights in LDS: with a packed image (pvd_head_pack_weights) that is ~10
// independent 16-byte loads per thread instead of a convert-and-scatter of the fp32 masters (which was the
// kernels' fixed cost: 11-20k cycles forward, 23-31k backward of ~50k for a single tile)
__device__ __forceinline__ void copy_image(half_t *__restrict__ lds, const half_t *__restrict__ image, int halfs, uint32_t tid, uint32_t n) {
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(image);
    uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(lds);
    for (int i = tid; i < halfs / 8; i += n) dst[i] = src[i];
}

// the same copy as LDS-DMA (gfx950 global_load_lds_dwordx4): wave w moves the 1-KB pieces w, w + 4, ... -- lane l's 16 bytes land
// at piece base + 16 l, which is the linear order of the image -- without passing through registers; completion is the
// caller's business (s_waitcnt vmcnt(0), which __syncthreads() carries while such a load is in flight)
template <uint32_t BLOCK = kHeadBlock>
__device__ __forceinline__ void copy_image_dma(half_t *__restrict__ lds, const half_t *__restrict__ image, int halfs, uint32_t tid) {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const int bytes = halfs * 2;
    for (int c = (int)wave; c * 1024 < bytes; c += (int)(BLOCK / 64)) {
        const int off = c * 1024 + (int)lane * 16;
        if (off + 16 <= bytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)image + off),
                                             (__attribute__((address_space(3))) void *)((char *)lds + c * 1024), 16, 0, 0);
    }
}

// the same DMA with a compile-time size, fully unrolled: a LOOP of LDS-DMA operations in front of other loads makes the
// compiler's wait-count insertion give up on graded waits (every later `s_waitcnt vmcnt(N)` becomes vmcnt(0)), a straight
// line of them does not
template <int HALFS, uint32_t BLOCK = kHeadBlock>
__device__ __forceinline__ void copy_image_dma_static(half_t *__restrict__ lds, const half_t *__restrict__ image, uint32_t tid) {
    constexpr int bytes = HALFS * 2, pieces_per_round = (int)(BLOCK / 64), rounds = (bytes + 1024 * pieces_per_round - 1) / (1024 * pieces_per_round);
    const int wave = (int)(tid >> 6), lane = (int)(tid & 63u);
#pragma unroll
    for (int r = 0; r < rounds; r++) {
        const int c = r * pieces_per_round + wave;
        const int off = c * 1024 + lane * 16;
        if (off + 16 <= bytes)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)image + off),
                                             (__attribute__((address_space(3))) void *)((char *)lds + c * 1024), 16, 0, 0);
    }
}

// the packed image: [HeadLds<KIND>][HeadLdsT<KIND>], exactly the backward kernel's LDS contents (head_pack.h: shared with the VM
// lookup's forward launch, which can carry the pack in extra workgroups)
template <int KIND>
__global__ void __launch_bounds__(256) k_head_pack(HeadArgs a, half_t *__restrict__ image) {
    head_pack_elements<KIND>(a.Wa1, a.Wa2, a.Wc1, a.Wc2, a.Wc3, image, (int)(blockIdx.x * 256 + threadIdx.x), (int)(gridDim.x * 256));
}
static_assert(kVmImageHalfs == HeadLds<KIND_VM>::halfs + HeadLdsT<KIND_VM>::halfs, "head_pack.h: the VM image's size");
static_assert(kHashImageHalfs == HeadLds<KIND_HASH>::halfs + HeadLdsT<KIND_HASH>::halfs, "head_pack.h: the hash image's size");

// the global-memory inputs of one tile.  Loaded one tile AHEAD of their use: at one wave per SIMD (the backward) nothing
// else hides the ~2k-cycle round trip, which was a quarter of the 17.5k cycles per tile.
template <int KIND>
struct TileIn {
    h4 x[KIND == KIND_VM ? 9 : 2];  // VM: the 144 products; hash: the 28 encoder features
    float sraw, dx, dy, dz;
};
template <int KIND>
__device__ __forceinline__ TileIn<KIND> load_tile_in(const HeadArgs &a, size_t b, bool valid, uint32_t lane) {
    const uint32_t hi = lane >> 4;
    TileIn<KIND> in;
    const h4 hz = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
    if (KIND == KIND_HASH) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            in.x[s] = hz;
            const int k0 = 16 * s + 4 * hi;  // features k0..k0+3 = levels k0/2, k0/2+1 (2 channels each)
            if (valid && k0 < 28) {
                const uint32_t lv = k0 >> 1;
                const uint32_t u0 = *reinterpret_cast<const uint32_t *>(a.x0 + ((size_t)lv * a.M + b) * 2);
                const uint32_t u1 = *reinterpret_cast<const uint32_t *>(a.x0 + ((size_t)(lv + 1) * a.M + b) * 2);
                uint32_t w[2] = {u0, u1};
                __builtin_memcpy(&in.x[s], w, 8);
            }
        }
        in.sraw = 0.f;
    } else {
#pragma unroll
        for (int s = 0; s < 9; s++) in.x[s] = valid ? *reinterpret_cast<const h4 *>(a.x0 + b * 144 + 16 * s + 4 * hi) : hz;
        in.sraw = (valid && hi == 0) ? a.sigma_raw[b] : 0.f;
    }
    in.dx = in.dy = in.dz = 0.f;
    if (valid) { in.dx = a.dirs[3 * b]; in.dy = a.dirs[3 * b + 1]; in.dz = a.dirs[3 * b + 2]; }
    return in;
}

// NT tiles at once: every weight fragment is read from LDS once and feeds NT independent MFMAs (the backward runs one wave per
// SIMD: nothing else hides an LDS round trip or a matrix-core result).  Per tile the operations and their order are those of a
// single tile: bit-identical outputs whatever NT.
template <int KIND, int NT>
__device__ __forceinline__ void head_forward_tiles(const HeadArgs &a, const HeadLds<KIND> &W, const TileIn<KIND> *__restrict__ in, uint32_t lane,
                                                   TileFwd *__restrict__ t) {
    const uint32_t hi = lane >> 4;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    // ---- stage A: features -> 16-row feature tile
    if (KIND == KIND_HASH) {
        h4 Ha[NT][4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            f4 acc[NT];
#pragma unroll
            for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const h4 A = W.Wa1.afrag(n, s, lane);
#pragma unroll
                for (int u = 0; u < NT; u++) acc[u] = mfma(A, in[u].x[s], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < NT; u++) Ha[u][n] = relu_h4(to_h4(acc[u]));
        }
        f4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h4 A = W.Wa2.afrag(0, s, lane);
#pragma unroll
            for (int u = 0; u < NT; u++) acc[u] = mfma(A, Ha[u][s], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NT; u++) {
            h4 Fh = to_h4(acc[u]);  // Linear output is f16
            t[u].raw = zero;
            t[u].sig_raw = (float)Fh.x;  // pre-clamp h0 (for the clamp's backward mask)
            if (hi == 0) {
                const float c = fminf(a.clip_max, fmaxf(a.clip_sigma_min, (float)Fh.x));
                Fh.x = (half_t)c;
            }
            t[u].Fh = Fh;
            t[u].F = (f4){(float)Fh.x, (float)Fh.y, (float)Fh.z, (float)Fh.w};
#pragma unroll
            for (int s = 0; s < 2; s++) t[u].X[s] = in[u].x[s];
#pragma unroll
            for (int n = 0; n < 4; n++) t[u].Ha[n] = Ha[u][n];
        }
    } else {
        f4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
        for (int s = 0; s < 9; s++) {
            const h4 A = W.Wa1.afrag(0, s, lane);
#pragma unroll
            for (int u = 0; u < NT; u++) acc[u] = mfma(A, in[u].x[s < (KIND == KIND_VM ? 9 : 2) ? s : 0], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NT; u++) {
            const h4 rawh = to_h4(acc[u]);  // basis_mat output, f16
            t[u].raw = (f4){(float)rawh.x, (float)rawh.y, (float)rawh.z, (float)rawh.w};
            f4 F;
            F.x = fminf(a.clip_max, fmaxf(a.clip_feat_min, t[u].raw.x));
            F.y = fminf(a.clip_max, fmaxf(a.clip_feat_min, t[u].raw.y));
            F.z = fminf(a.clip_max, fmaxf(a.clip_feat_min, t[u].raw.z));
            F.w = fminf(a.clip_max, fmaxf(a.clip_feat_min, t[u].raw.w));
            t[u].sig_raw = 0.f;
            if (hi == 0) {
                t[u].sig_raw = in[u].sraw;
                F.x = fminf(a.clip_max, fmaxf(a.clip_sigma_min, t[u].sig_raw));  // row 0 := clamped sigma feature (fp32)
            }
            t[u].F = F;
            t[u].Fh = to_h4(F);
        }
    }
    // ---- stage B: colour head
#pragma unroll
    for (int u = 0; u < NT; u++) t[u].sh = sh_frag(in[u].dx, in[u].dy, in[u].dz, hi);
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const h4 A0 = W.Wc1.afrag(n, 0, lane), A1 = W.Wc1.afrag(n, 1, lane);
#pragma unroll
        for (int u = 0; u < NT; u++) {
            f4 acc = zero;
            acc = mfma(A0, t[u].sh, acc);
            acc = mfma(A1, t[u].Fh, acc);  // column 16 (log-sigma) has zero weight
            t[u].H1[n] = relu_h4(to_h4(acc));
        }
    }
#pragma unroll
    for (int n = 0; n < 4; n++) {
        f4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h4 A = W.Wc2.afrag(n, s, lane);
#pragma unroll
            for (int u = 0; u < NT; u++) acc[u] = mfma(A, t[u].H1[s], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NT; u++) t[u].H2[n] = relu_h4(to_h4(acc[u]));
    }
    {
        f4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h4 A = W.Wc3.afrag(0, s, lane);
#pragma unroll
            for (int u = 0; u < NT; u++) acc[u] = mfma(A, t[u].H2[s], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NT; u++) t[u].out = acc[u];
    }
}
template <int KIND>
__device__ __forceinline__ void head_forward_tile(const HeadArgs &a, const HeadLds<KIND> &W, const TileIn<KIND> &in, uint32_t lane, TileFwd &t) {
    head_forward_tiles<KIND, 1>(a, W, &in, lane, &t);
}

__device__ __forceinline__ float sigmoid_h(float pre) {
    // Linear output rounded to f16, sigmoid evaluated in fp32, result rounded to f16 (torch.sigmoid on a half tensor)
    const float x = (float)(half_t)pre;
    return (float)(half_t)(1.0f / (1.0f + __expf(-x)));
}

template <int KIND>
__global__ void __launch_bounds__(kHeadBlock) k_head_fwd(HeadArgs a) {
    extern __shared__ __align__(16) half_t lds[];
    if (a.rows_dev) a.M = min(a.M, (uint32_t)max(*a.rows_dev, 0));
    if (blockIdx.x * (kHeadBlock / 64) * 16u >= a.M) return;  // nothing for this workgroup: skip the weight staging too
    HeadLds<KIND> W;
    W.carve(lds);
    const uint32_t lane = threadIdx.x & 63u, hi = lane >> 4;
    const uint32_t wave = (blockIdx.x * kHeadBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = gridDim.x * (kHeadBlock / 64);
    const uint32_t ntiles = div_up(a.M, 16u);
    // A cold launch is a chain of memory round trips of 1.5-2 us each (kernel arguments -> weight image -> first tile's inputs -> ...):
    // the launch costs 10 us whatever M (tools/bench_head.py, M = 64).  The packed image goes by LDS-DMA, every piece in flight at
    // once, and the first tile's inputs are requested BEHIND it in the same round trip (one wait for both).
    if (a.image) copy_image_dma_static<HeadLds<KIND>::halfs>(lds, a.image, threadIdx.x);
    TileIn<KIND> nxt = load_tile_in<KIND>(a, (size_t)wave * 16 + (lane & 15), (size_t)wave * 16 + (lane & 15) < a.M, lane);
    if (a.image) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the DMA has landed before the barrier releases the readers
    else W.load(a, threadIdx.x, kHeadBlock);
    __syncthreads();
    for (uint32_t tile = wave; tile < ntiles; tile += nwaves) {
        const size_t b = (size_t)tile * 16 + (lane & 15);
        const bool valid = b < a.M;
        const TileIn<KIND> in = nxt;
        if (tile + nwaves < ntiles) {
            const size_t bn = (size_t)(tile + nwaves) * 16 + (lane & 15);
            nxt = load_tile_in<KIND>(a, bn, bn < a.M, lane);
        }
        TileFwd t;
        head_forward_tile<KIND>(a, W, in, lane, t);
        if (valid) {
            *reinterpret_cast<f4 *>(a.feat16 + b * 16 + 4 * hi) = t.F;
            if (hi == 0) {
                a.sigma[b] = __expf(t.F.x);
                a.rgb[3 * b] = sigmoid_h(t.out.x);
                a.rgb[3 * b + 1] = sigmoid_h(t.out.y);
                a.rgb[3 * b + 2] = sigmoid_h(t.out.z);
            }
        }
    }
}

template <int KIND>
static int launch_head_fwd(const HeadArgs &a, hipStream_t s) {
    const uint32_t ntiles = div_up(a.M, 16u);
    uint32_t blocks = div_up(ntiles, kHeadBlock / 64);
    if (blocks > 768) blocks = 768;  // 256 CUs x 3 resident workgroups (148 VGPRs): every workgroup pays one weight load (1024: a second round of them --
                                     // 24.5 against 19.1 us from the fp32 masters, 11.8 against 11.4 from the packed image)
    const size_t lds_bytes = HeadLds<KIND>::halfs * sizeof(half_t);
    hipLaunchKernelGGL((k_head_fwd<KIND>), dim3(blocks), dim3(kHeadBlock), lds_bytes, s, a);
    return check_launch();
}

// ====================================================================== frozen hash model: lookup + head in ONE launch
//
// The teacher of a distillation run (and every inference / occupancy query of a hash model) needs no autograd state, so the
// [14][M][2] f16 encoder output that pvd_grid_encode_forward writes and pvd_head_forward reads back is pure overhead, and so
// is the second launch.  Here a workgroup owns 128 samples: phase 1 is the lanes-per-sample lookup (gridencoder.hip,
// k_grid_fwd_lps<2>: lane pair = corners x / x+1, DPP blend in the reference's corner order -- bit-identical values) looped
// over the 14 levels with the results going to an LDS feature tile [128][28] f16; phase 2 runs the MFMA head on that tile
// (16 samples per wave and pass, weights from the packed LDS image).  All workgroups of a launch are co-resident (726 at the
// bench size, <= 3 per CU) and walk the levels in step, so at any moment the whole chip gathers from one or two levels'
// tables (2 MiB each) -- the level-major sweep that keeps the lookup's working set inside the L2s -- without a grid of
// (level, block) workgroups; the position of a sample is read once instead of 14 times.
constexpr uint32_t kFusedTile = 128;        // samples per workgroup pass (= 256 threads / 2 lanes per sample)
constexpr int kFeatStride = 36;             // halfs per LDS feature row: 28 features + 4 zeros + 4 pad (8-byte B-fragment reads)

struct FusedLookup {
    const float *xyz;        // [M][3] positions in [-bound, bound]
    InputAffine aff;         // x01 = (x + add) / div   (GridEncoder.forward's mapping, grid.py:211)
    const uint32_t *grid;    // embeddings as packed f16 pairs [rows]
    const int32_t *offsets;  // [15]
    LevelScales scales;
    uint32_t gridtype;
    bool align_corners;
    // (measurement, pvd_hash_head_forward_fused_span) DEVICE [2] or NULL: span[0] = min over the launch's workgroups of their start,
    // span[1] = max of their end, in s_memrealtime ticks (100 MHz) -- how long THIS launch lasted where it ran, e.g. inside a
    // replayed graph next to other kernels, where no host-side event can be placed.  The caller initialises {~0, 0}.
    unsigned long long *span = nullptr;
};

// Round 4.  What the round-3 kernel (removed in round 5; DESIGN section 10.1, profiles/r04_fused_variants_ab.txt) really executed was
// FOURTEEN dependent memory round trips per workgroup, not two: `g.offsets` is a pointer inside a by-value struct, so the
// compiler cannot prove the table of level offsets invariant, fetches offsets[level + 1] with a VECTOR load in front of every
// level and waits for it -- and vmcnt retires in order, so that wait (s_waitcnt vmcnt(1) / vmcnt(0) in the ISA) also drains
// every gather of the levels before it.  On top, LevelIndex decided hashed / pow2 / wrap with three uniform branches in front
// of every one of the 56 loads of a lane (~90 instructions per load, 428 branches in the kernel).  Here:
//   * the 15 offsets are read ONCE per workgroup (lanes 0..14, one load) and broadcast into SGPRs with v_readlane;
//   * a level's shape is decided once (Level3, grid_lookup.h): hashed power-of-two levels and dense levels that cannot wrap
//     get straight-line index code (shared y P1 / z P2 terms: 3 VALU ops per corner), everything else goes out of line;
//   * the fractions are recomputed after the loads return instead of being held in 3 G registers;
// so the G levels of a group really are in flight together (G = 7: two round trips, 28 loads per lane; G = 14: one, 56).
// Same rows, same blend order: outputs bit-identical to the round-3 kernel and to lookup + head as two launches.
struct FusedRes {
    uint32_t res[14];  // (uint32_t)ceil((double)scale) + 1, on the host (what the kernels compute per level on the device)
};

// The 14-level lookup of ONE sample by its lane pair (x01 in [0, 1], `xb` = which x corner this lane fetches): gathers issued coarse
// to fine in groups of G levels, each level blended in the reference's corner order as soon as its own four loads have retired, the two
// features of a level written to `feat_row[2 level]` by the even lane.  Shared by k_hash_fwd_fused and k_infer_hash_persistent.
template <uint32_t G>
__device__ __forceinline__ void fused_lookup_sample(const FusedLookup &g, const FusedRes &gr, const uint32_t (&off)[15], const float (&x01)[3],
                                                bool inside, uint32_t xb, half_t *__restrict__ feat_row) {
    constexpr uint32_t D = 3, L = 14;
    const float half_or_0 = g.align_corners ? 0.0f : 0.5f;
#pragma unroll
    for (uint32_t l0 = 0; l0 < L; l0 += G) {
        uint32_t v[G][4];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) {
            const uint32_t level = l0 + j;
            if (level >= L) break;
            const float scale = g.scales.scale[level];
            Level3 lv;
            lv.init(off[level + 1] - off[level], gr.res[level], g.gridtype, g.align_corners);
            const uint32_t *__restrict__ table = g.grid + off[level];
            uint32_t cell[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) cell[d] = inside ? (uint32_t)floorf(fmaf(x01[d], scale, half_or_0)) : 0u;
            uint32_t row[4];
            level3_rows(lv, g.gridtype, g.align_corners, cell, xb, row);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) v[j][k] = table[row[k]];
            // G = 14: the levels are ISSUED coarse to fine and CONSUMED in the same order while the finer ones are still in
            // flight (vmcnt retires in order: level j is blended behind s_waitcnt vmcnt(4 (13 - j))), so the blend arithmetic
            // of the early levels runs under the fine levels' memory time.  The fences keep the scheduler from hoisting all
            // 56 address computations to the top (201 VGPRs) and from interleaving the blends (one vmcnt(0) for all).
            if (G == 14) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) {
            const uint32_t level = l0 + j;
            if (level >= L) break;
            const float scale = g.scales.scale[level];
            float fr[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                float xd = x01[d];
                asm volatile("" : "+v"(xd));  // recompute, do not keep: the same expression as above would be CSE'd into 3 live registers per level
                const float p = fmaf(xd, scale, half_or_0);
                fr[d] = p - (float)(uint32_t)floorf(p);
            }
            uint32_t acc = 0u;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t yb = k & 1u, zb = k >> 1;
                float wi = 1;  // the reference's product order: ((1 * wx) * wy) * wz
                wi *= xb ? fr[0] : 1 - fr[0];
                wi *= yb ? fr[1] : 1 - fr[1];
                wi *= zb ? fr[2] : 1 - fr[2];
                const uint32_t pr = weighted_pair(wi, v[j][k]);
                const uint32_t other = dpp_quad<0xB1>(pr);
                acc = pk_add(acc, xb ? other : pr);
                acc = pk_add(acc, xb ? pr : other);
            }
            if (xb == 0) *reinterpret_cast<uint32_t *>(feat_row + 2 * level) = inside ? acc : 0u;
            if (G == 14) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <uint32_t G, int DMA>
__global__ void __launch_bounds__(kHeadBlock) k_hash_fwd_fused(HeadArgs a, FusedLookup g, FusedRes gr) {
    extern __shared__ __align__(16) half_t lds[];
    constexpr uint32_t D = 3, L = 14;
#ifdef PVD_FUSED_PROFILE  // instrumented A/B build (tools/prof_fused_stamps.py): every 97th workgroup's waves leave 8 s_memtime stamps
    long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PVD_FSTAMP(k) do { stamp[k] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PVD_FSTAMP(k) do { } while (0)
#endif
    PVD_FSTAMP(0);
    // (measurement) the workgroup's start time, parked in LDS until the end: no register lives across the kernel for it, and no
    // global memory operation is issued here (the gathers' graded vmcnt waits would have to drain it)
    __shared__ unsigned long long span_t0;
    if (g.span && threadIdx.x == 0) span_t0 = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    if (a.rows_dev) a.M = min(a.M, (uint32_t)max(*a.rows_dev, 0));
    if (blockIdx.x * kFusedTile >= a.M) return;  // nothing for this workgroup: skip the weight staging too
    HeadLds<KIND_HASH> W;
    W.carve(lds);
    half_t *feat = lds + ((HeadLds<KIND_HASH>::halfs + 7) & ~7);  // [kFusedTile][kFeatStride]
    const uint32_t lane = threadIdx.x & 63u, hi = lane >> 4, wave = threadIdx.x >> 6;
    // level offsets: one vector load per wave (the FIRST load of the kernel: with the positions it is the critical path to the
    // first gather), then SGPRs for the rest of the kernel
    const int32_t offs_v = lane <= L ? g.offsets[lane] : 0;
    // The head's weights: with a packed image, 22 KB of L2 hits go straight into LDS (global_load_lds_dwordx4: no data
    // registers).  WHERE the DMA is issued matters more than it looks: while a global_load_lds is outstanding the compiler turns
    // every later `s_waitcnt vmcnt(N)` into vmcnt(0) (it does not assume DMA and register loads retire in order), so a DMA in
    // flight during the lookup serialises its graded waits.  DMA = 0: issued first and drained together with the positions, in
    // front of the first gather; DMA = 1: issued when the blend is done, its latency in front of the head.  (Register-staging
    // the image instead -- loads behind the last gather, ds_write in front of the barrier -- costs 20 VGPRs and the third wave
    // per SIMD.)  Without an image the fp32 masters are converted in place (tests).
    if (!a.image) W.load(a, threadIdx.x, kHeadBlock);
    else if (DMA == 0) copy_image_dma_static<HeadLds<KIND_HASH>::halfs>(lds, a.image, threadIdx.x);
    for (uint32_t i = threadIdx.x; i < kFusedTile * 2; i += kHeadBlock)  // features 28..31 of every row: zero for good
        *reinterpret_cast<uint32_t *>(feat + (i >> 1) * kFeatStride + 28 + 2 * (i & 1)) = 0u;
    const uint32_t xb = threadIdx.x & 1u, s_local = threadIdx.x >> 1;
    const uint32_t nchunks = div_up(a.M, kFusedTile);
    constexpr int kTilesPerWave = kFusedTile / 16 / (kHeadBlock / 64);
    uint32_t off[L + 1];
#pragma unroll
    for (uint32_t l = 0; l <= L; l++) off[l] = (uint32_t)__builtin_amdgcn_readlane(offs_v, (int)l);
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // ---------------- phase 1: 14-level lookup of sample b by the lane pair (b, xb)
        const uint32_t b = chunk * kFusedTile + s_local;
        float x01[D] = {0.f, 0.f, 0.f};
        bool inside = b < a.M;
        // Unconditional loads of a clamped row: a load under a run-time condition reaches its use through a phi and is waited for
        // at the end of the conditional block; rows past M are computed on row M - 1 and never stored.  The directions of the
        // (two) 16-sample tiles this wave will run the head on are requested here and used after the lookup.  (Requesting a
        // chunk's inputs one chunk ahead -- the first chunk's in front of the wait for the offsets -- was built and measured:
        // no change, the workgroup's first wait is ONE round trip of ~3-4 us either way: profiles/r04_fused_stamps.txt.)
        const Pos3 p = *reinterpret_cast<const Pos3 *>(g.xyz + (size_t)min(b, a.M - 1u) * D);
        float dir_pre[kTilesPerWave][3];
#pragma unroll
        for (int ti = 0; ti < kTilesPerWave; ti++) {
            const size_t bs = min((size_t)chunk * kFusedTile + (wave + ti * (kHeadBlock / 64)) * 16 + (lane & 15), (size_t)a.M - 1);
#pragma unroll
            for (int c = 0; c < 3; c++) dir_pre[ti][c] = a.dirs[3 * bs + c];
        }
        // a sample outside the box gathers the rows of cell (0, 0, 0) of every level and is zeroed afterwards (branch-free)
        if (inside) {  // (first use of the position: everything above is in flight by now)
            x01[0] = p.x; x01[1] = p.y; x01[2] = p.z;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if (g.aff.on) x01[d] = (x01[d] + g.aff.add) / g.aff.div;
                inside = inside && !(x01[d] < 0.0f) && !(x01[d] > 1.0f);
            }
        }
        PVD_FSTAMP(7);  // (everything the workgroup needs before its first gather has been requested)
        if (DMA == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // positions, directions, offsets AND the weight DMA: nothing in flight from here
        PVD_FSTAMP(1);
        fused_lookup_sample<G>(g, gr, off, x01, inside, xb, feat + s_local * kFeatStride);
        PVD_FSTAMP(2);
        if (DMA == 1 && a.image && chunk == blockIdx.x) copy_image_dma_static<HeadLds<KIND_HASH>::halfs>(lds, a.image, threadIdx.x);
        PVD_FSTAMP(3);
        __syncthreads();  // (also covers the weights / zero columns on the first pass)
        PVD_FSTAMP(4);
        // ---------------- phase 2: the head on the tile, 16 samples per wave and pass
        for (uint32_t t16 = wave; t16 < kFusedTile / 16; t16 += kHeadBlock / 64) {
            const uint32_t row = t16 * 16 + (lane & 15);
            const size_t bs = (size_t)chunk * kFusedTile + row;
            const bool valid = bs < a.M;
            TileIn<KIND_HASH> in;
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) in.x[s2] = *reinterpret_cast<const h4 *>(feat + row * kFeatStride + 16 * s2 + 4 * hi);
            in.sraw = 0.f;
            const int tsel = (int)((t16 - wave) / (kHeadBlock / 64));
            in.dx = dir_pre[tsel][0]; in.dy = dir_pre[tsel][1]; in.dz = dir_pre[tsel][2];
            TileFwd t;
            head_forward_tile<KIND_HASH>(a, W, in, lane, t);
            if (valid) {
                *reinterpret_cast<f4 *>(a.feat16 + bs * 16 + 4 * hi) = t.F;
                if (hi == 0) {
                    a.sigma[bs] = __expf(t.F.x);
                    a.rgb[3 * bs] = sigmoid_h(t.out.x);
                    a.rgb[3 * bs + 1] = sigmoid_h(t.out.y);
                    a.rgb[3 * bs + 2] = sigmoid_h(t.out.z);
                }
            }
        }
        PVD_FSTAMP(5);
        __syncthreads();  // the next chunk's lookup overwrites the tile
        PVD_FSTAMP(6);
#ifdef PVD_FUSED_PROFILE
        if (chunk % 97u == 0 && lane == 0) {
            long long *dst = reinterpret_cast<long long *>(a.rgb) + ((size_t)(chunk / 97u) * 4 + wave) * 8;
            for (int q = 0; q < 8; q++) dst[q] = stamp[q];
        }
#endif
    }
    if (g.span && threadIdx.x == 0) {
        atomicMin(g.span, span_t0);
        atomicMax(g.span + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ====================================================================== inference of a frozen hash model: ONE launch per image
//
// Replaces the round loop of run_cuda's eval branch (distill_mutual/renderer.py:450-543: march_rays -> model -> composite_rays ->
// compact_rays, ~54 rounds x 6 launches for an 800 x 800 view, every round looking up n_alive * n_step rows of which the rays that
// ended inside the round leave zero-filled ones) by two launches -- SURVEY section 8 f2:
//   k_infer_first_hit       thread per ray: the marcher's walk from `near` up to the ray's FIRST occupied probe (the empty space in
//                           front of the object: a chain of dependent bitfield loads per ray, cheap only when every ray of the image
//                           runs it at once); rays that never meet an occupied cell are not queued at all;
//   k_infer_hash_persistent a workgroup owns kInfRays ray SLOTS, one per thread; the ray's march position and its accumulators live
//                           in the thread's registers.  A local round:
//       refill   free slots take the next rays of the image from a device-side queue (one atomic per workgroup);
//       march    every live slot walks on (k_march_rays' loop, the same Dda::probe) until it holds n_step = max(min(kInfRows / live,
//                8), 1) samples -- the reference's rule (renderer.py:493) with the workgroup's own numbers -- or has spent its probes;
//       shade    the samples go into an LDS tile, densely packed; 14-level lookup + sigma / colour head on the tile's rows, results
//                left in LDS (fused_lookup_sample + head_forward_tile: the arithmetic of pvd_hash_head_forward_fused);
//       blend    every slot composites its samples in order (k_composite_rays' loop body) and retires when the ray ended or saturated.
// A ray's samples and its sums depend on nothing but the ray: the walk is the reference's walk paused and resumed (the running t is
// carried, never recomputed), the sums are the reference's sums in the reference's order -- the image is the round loop's, bit for
// bit (tests/test_hip_infer_rounds.py).  One thing the reference's loop does that this does not: it restarts a ray's march every
// round from the t its compositing reconstructed by adding up `t - last_t`; those differences are exact for consecutive samples, so
// the two t agree.  The reference stops ALL rays once the rounds' steps add up to max_steps; here a ray stops after max_steps samples.
// (InferImageArgs, kInfRows / kInfSteps / kInfProbes: infer_persistent.h -- shared with plenoxel.hip's k_infer_px_persistent)

// The walk in front of the object: from t = near to the first occupied probe (raymarching.cu:756-810 with nothing emitted).
__global__ void __launch_bounds__(kHeadBlock) k_infer_first_hit(InferImageArgs q, uint32_t N, float *__restrict__ t_first,
                                                                int32_t *__restrict__ ray_ids, int32_t *__restrict__ n_ids) {
    __shared__ uint32_t wave_cnt[kHeadBlock / 64];
    __shared__ uint32_t block_base;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * kHeadBlock; base < N; base += gridDim.x * kHeadBlock) {  // uniform per workgroup
        const uint32_t n = base + threadIdx.x;
        bool keep = false;
        if (n < N) {
            const float near = q.nears[n], far = q.fars[n];
            if (near < far) {
                Dda r;
                r.init(q.rays_o + 3 * (size_t)n, q.rays_d + 3 * (size_t)n, q.bound, q.dt_gamma, q.max_steps, q.C, q.H, q.grid);
                float t = near;
                while (t < far) {
                    float x, y, z, dt, tn;
                    if (r.probe(t, x, y, z, dt, tn)) { keep = true; break; }
                    t = tn;
                }
                t_first[n] = t;
            }
        }
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) wave_cnt[wid] = __popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < kHeadBlock / 64; w++) { const uint32_t c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
            block_base = tot ? (uint32_t)atomicAdd(n_ids, (int32_t)tot) : 0u;
        }
        __syncthreads();
        if (keep) ray_ids[block_base + wave_cnt[wid] + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)n;
        __syncthreads();
    }
}

// RAYS = ray slots per workgroup (threads 0 .. RAYS - 1 own one each; all 256 threads shade).  Fewer slots per workgroup = more
// workgroups per image = more waves per SIMD to hide the shading's latencies, and a queue that outlasts the first fill.
template <uint32_t RAYS, uint32_t ROWS>
__global__ void __launch_bounds__(kHeadBlock) k_infer_hash_persistent(HeadArgs a, FusedLookup g, FusedRes gr, InferImageArgs q) {
    extern __shared__ __align__(16) half_t lds[];
    constexpr uint32_t D = 3, L = 14;
    HeadLds<KIND_HASH> W;
    W.carve(lds);
    half_t *feat = lds + ((HeadLds<KIND_HASH>::halfs + 7) & ~7);                         // [ROWS][kFeatStride]
    float *pos = reinterpret_cast<float *>(feat + ROWS * kFeatStride);                 // [ROWS][3]
    float *sig = pos + 3 * ROWS;                                                       // [ROWS]
    float *rgb = sig + ROWS;                                                           // [ROWS][3]
    float *sdir = rgb + 3 * ROWS;                                                      // [kHeadBlock][3]: direction of the ray in slot s
    uint32_t *row_slot = reinterpret_cast<uint32_t *>(sdir + 3 * kHeadBlock);              // [ROWS]
    uint32_t *wcnt = row_slot + ROWS;                                                  // [8] scan scratch + queue hand-off
    const uint32_t tid = threadIdx.x, lane = tid & 63u, hi = lane >> 4, wave = tid >> 6;
    const int32_t offs_v = lane <= L ? g.offsets[lane] : 0;
    if (a.image) copy_image(lds, a.image, HeadLds<KIND_HASH>::halfs, tid, kHeadBlock);
    else W.load(a, tid, kHeadBlock);
    for (uint32_t i = tid; i < ROWS * 2; i += kHeadBlock)  // features 28..31 of every row: zero for good
        *reinterpret_cast<uint32_t *>(feat + (i >> 1) * kFeatStride + 28 + 2 * (i & 1)) = 0u;
    uint32_t off[L + 1];
#pragma unroll
    for (uint32_t l = 0; l <= L; l++) off[l] = (uint32_t)__builtin_amdgcn_readlane(offs_v, (int)l);
    const uint32_t n_ids = (uint32_t)max(*q.n_ids, 0);
    const uint32_t xb = tid & 1u, s_local = tid >> 1;
    // queue position -> ray: position * mul mod n_ids is a permutation only for a multiplier coprime to n_ids
    uint32_t mul = q.shuffle % max(n_ids, 1u);
    for (;; mul++) {
        uint32_t x = max(mul, 1u), y = max(n_ids, 1u);
        while (y) { const uint32_t r = x % y; x = y; y = r; }
        if (x == 1u || n_ids <= 1u) break;
    }
    mul = max(mul, 1u);

    // the slot's ray: t = what compositing has reached (the reference's rays_t / last_t), tt = where the walk stands, and the `cnt`
    // samples the walk has found since the last blend (position, dt, the walk's t behind the sample)
    int32_t index = -1;
    uint32_t taken = 0, cnt = 0;
    float t = 0.f, tt = 0.f, far = 0.f, ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    float sx[kInfSteps], sy[kInfSteps], sz[kInfSteps], sdt[kInfSteps], stt[kInfSteps];
    float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 1.f};
    bool queue_done = n_ids == 0;

    // exclusive prefix of `v` over the workgroup's threads and the total (two barriers)
    auto scan_of = [&](uint32_t v, uint32_t &total) -> uint32_t {
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
            if ((int)lane >= d) inc += up;
        }
        if (lane == 63) wcnt[wave] = inc;
        __syncthreads();
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < kHeadBlock / 64; w++) { const uint32_t c = wcnt[w]; before += w < wave ? c : 0u; tot += c; }
        __syncthreads();
        total = tot;
        return before + inc - v;
    };
    auto retire = [&]() {  // the ray is done: its pixel leaves the slot
        q.weights_sum[index] = ws;
        q.depth[index] = dep;
        q.image[3 * (size_t)index] = cr; q.image[3 * (size_t)index + 1] = cg; q.image[3 * (size_t)index + 2] = cb;
        index = -1; cnt = 0;
    };

    uint32_t n_rounds = 0, n_rows = 0, n_walk = 0;
#ifdef PVD_INFER_PROFILE  // phase times of workgroup 0 (100 MHz ticks): refill+scan, march, lookup, head, blend
    long long ph[5] = {0, 0, 0, 0, 0}, tm = (long long)__builtin_amdgcn_s_memrealtime();
#define PVD_ISTAMP(k) do { const long long now_ = (long long)__builtin_amdgcn_s_memrealtime(); ph[k] += now_ - tm; tm = now_; } while (0)
#else
#define PVD_ISTAMP(k) do { } while (0)
#endif
    __syncthreads();  // weights, zero columns
    for (;;) {
        // ---------------- refill: free slots take the next rays (when enough of them are free to be worth the round trip)
        uint32_t nfree;
        const uint32_t frank = scan_of(tid < RAYS && index < 0 ? 1u : 0u, nfree);
        if (!queue_done && (nfree >= RAYS / 8 || nfree == RAYS)) {
            if (tid == 0) wcnt[4] = (uint32_t)atomicAdd(q.queue, (int32_t)nfree);
            __syncthreads();
            const uint32_t base = wcnt[4];
            __syncthreads();
            if (base + nfree >= n_ids) queue_done = true;  // (uniform) the queue has been handed out completely
            if (tid < RAYS && index < 0 && base + frank < n_ids) {
                // queue position -> ray: a multiplicative shuffle of the (image-ordered) list, so that a workgroup's 256 rays come
                // from all over the image -- neighbouring pixels have similar path lengths, and a workgroup of long rays would
                // be the launch's tail (the queue is empty after the first fill whenever the image has fewer rays than slots)
                const int32_t id = q.ray_ids[(uint32_t)(((uint64_t)(base + frank) * mul) % n_ids)];
                index = id; taken = 0; cnt = 0;
#pragma unroll
                for (int c = 0; c < 3; c++) { ro[c] = q.rays_o[3 * (size_t)id + c]; rd[c] = q.rays_d[3 * (size_t)id + c]; sdir[3 * tid + c] = rd[c]; }
                t = q.nears[id]; far = q.fars[id]; tt = q.t_first[id];
                ws = dep = cr = cg = cb = 0.f;
            }
        }
        const uint32_t nlive = nfree < RAYS ? RAYS - nfree : 0u;  // (slots filled just now included below)
        uint32_t live_now;
        (void)scan_of(index >= 0 ? 1u : 0u, live_now);
        if (live_now == 0) {
            if (queue_done) break;
            continue;
        }
        (void)nlive;
        PVD_ISTAMP(0);
        // samples per slot this round: the reference's rule (renderer.py:493) with the workgroup's own numbers
        const uint32_t n_step = max(min(ROWS / live_now, kInfSteps), 1u);
        // ---------------- march: walk on (k_march_rays' loop, raymarching.cu:756-810, perturb = 0)
        if (index >= 0) {
            Dda r;
            r.init(ro, rd, q.bound, q.dt_gamma, q.max_steps, q.C, q.H, q.grid);
            for (uint32_t pb = 0; pb < n_step + kInfProbes && cnt < n_step && tt < far; pb++) {
                float x, y, z, dt, tn;
                if (r.probe(tt, x, y, z, dt, tn)) {
                    tt += dt;
#pragma unroll
                    for (uint32_t k = 0; k < kInfSteps; k++)  // (static register indices)
                        if (k == cnt) { sx[k] = x; sy[k] = y; sz[k] = z; sdt[k] = dt; stt[k] = tt; }
                    cnt++;
                } else {
                    tt = tn;
                }
            }
            if (cnt == 0 && !(tt < far)) retire();  // the walk left the box: no further sample (the round loop's dt == 0 row)
        }
        PVD_ISTAMP(1);
        uint32_t rows;
        const uint32_t row0 = scan_of(cnt, rows);
        n_rounds++; n_rows += rows;
        if (rows == 0) { n_walk++; continue; }  // walkers only
#pragma unroll
        for (uint32_t k = 0; k < kInfSteps; k++)
            if (k < cnt) {
                pos[3 * (row0 + k)] = sx[k]; pos[3 * (row0 + k) + 1] = sy[k]; pos[3 * (row0 + k) + 2] = sz[k];
                row_slot[row0 + k] = tid;
            }
        __syncthreads();
        // ---------------- shade: lookup (two lanes per row, 128 rows per pass) ...
        for (uint32_t p0 = 0; p0 < rows; p0 += kFusedTile) {  // uniform
            const uint32_t rw = p0 + s_local;
            bool inside = rw < rows;
            float x01[D] = {0.f, 0.f, 0.f};
            if (inside) {
#pragma unroll
                for (uint32_t d = 0; d < D; d++) {
                    x01[d] = pos[3 * rw + d];
                    if (g.aff.on) x01[d] = (x01[d] + g.aff.add) / g.aff.div;
                    inside = inside && !(x01[d] < 0.0f) && !(x01[d] > 1.0f);
                }
            }
            fused_lookup_sample<7>(g, gr, off, x01, inside, xb, feat + min(rw, ROWS - 1u) * kFeatStride);
        }
        __syncthreads();
        PVD_ISTAMP(2);
        // ... and the head, 16 rows per wave and pass; sigma / rgb stay in LDS.  (Two tiles per pass as two independent MFMA chains, and
        // all 14 levels of the lookup in flight: 210 VGPRs = two workgroups per CU instead of three, and no faster per round.)
        for (uint32_t t16 = wave; t16 * 16 < rows; t16 += kHeadBlock / 64) {
            const uint32_t rw = min(t16 * 16 + (lane & 15), ROWS - 1u);
            TileIn<KIND_HASH> in;
#pragma unroll
            for (int s2 = 0; s2 < 2; s2++) in.x[s2] = *reinterpret_cast<const h4 *>(feat + rw * kFeatStride + 16 * s2 + 4 * hi);
            in.sraw = 0.f;
            const uint32_t slot = rw < rows ? row_slot[rw] : 0u;
            in.dx = sdir[3 * slot]; in.dy = sdir[3 * slot + 1]; in.dz = sdir[3 * slot + 2];
            TileFwd tf;
            head_forward_tile<KIND_HASH>(a, W, in, lane, tf);
            if (hi == 0 && t16 * 16 + (lane & 15) < rows) {
                sig[rw] = __expf(tf.F.x);
                rgb[3 * rw] = sigmoid_h(tf.out.x); rgb[3 * rw + 1] = sigmoid_h(tf.out.y); rgb[3 * rw + 2] = sigmoid_h(tf.out.z);
            }
        }
        __syncthreads();
        PVD_ISTAMP(3);
        // ---------------- blend (k_composite_rays' loop, raymarching.cu:858-899; sigma scaled as renderer.py:528)
        if (cnt > 0) {
            bool done = false;
#pragma unroll
            for (uint32_t k = 0; k < kInfSteps; k++) {
                if (k < cnt && !done) {
                    const uint32_t rw = row0 + k;
                    const float alpha = 1.0f - __expf(-(q.sigma_scale * sig[rw]) * sdt[k]);
                    const float T = 1 - ws;
                    const float w = alpha * T;
                    ws += w;
                    t += stt[k] - t;  // deltas[1] = (walk's t behind the sample) - last_t, added to the ray's t (raymarching.cu:795, 877)
                    dep += w * t;
                    cr += w * rgb[3 * rw]; cg += w * rgb[3 * rw + 1]; cb += w * rgb[3 * rw + 2];
                    taken++;
                    if ((double)T < 1e-4 || taken >= q.max_steps) done = true;
                }
            }
            cnt = 0;
            if (done) retire();
        }
        __syncthreads();  // the next round rewrites the tile
        PVD_ISTAMP(4);
    }
#ifdef PVD_INFER_PROFILE
    if (tid == 0 && blockIdx.x == 0) for (int z = 0; z < 5; z++) q.stats[4 + z] = (int32_t)ph[z];
#endif
    if (tid == 0 && n_rounds > 1) {
        atomicAdd(q.stats + 0, (int32_t)n_rounds); atomicAdd(q.stats + 1, (int32_t)n_rows); atomicAdd(q.stats + 2, (int32_t)n_walk); atomicAdd(q.stats + 3, 1);
    }
}

// ---- the same persistent render for a frozen VM (TensoRF plane x line) model -- the student whose PSNR the metric reports
// (pvd_infer_image_vm).  Refill / march / blend are k_infer_hash_persistent's; the shading of a round's LDS tile differs:
//   lookup   wave w takes the rows w, w + 4, ...: lane j of the wave works out row j's sampling state once (normalised position ->
//            three axes x {clamped taps, weights, in-range bits}), then the wave goes through its rows with lane = channel (16 sigma +
//            48 colour), every texel of a row's three 2 x 2 + 2 footprints requested unconditionally and the next row's 18 loads in
//            flight while this row's products are formed (vm_lookup.h: rows of one round belong to up to 64 different rays -- there
//            is no run for k_vm_fwd's register windows to follow); products rounded to f16 into the tile's row [144], the sigma
//            feature summed over the 16 sigma lanes in k_vm_fwd's order;
//   head     k_head_fwd<VM>'s tile code on the LDS rows (basis_mat, clamps, colour head).
// Arithmetic per row = pvd_vm_forward + pvd_head_forward (bit for bit); per ray = the round loop's (tests/test_hip_infer_rounds.py).
#ifndef PVD_INFER_VM_DEPTH
#define PVD_INFER_VM_DEPTH 2
#endif
constexpr uint32_t kVmInfStride = 144 + 8;  // halfs per tile row: 76 dwords = 4 x odd -- the 16 rows of an 8-byte fragment read start on distinct bank quads
// ROWS sample rows per local round, shaded FT at a time (the feature tile is the big LDS consumer); OCC: waves per SIMD the register
// budget must allow.  <64, 128, 64, 3> -- 47 KB of LDS, 168 VGPRs, three workgroups per CU -- renders the 800 x 800 view in 7.6 ms where
// <64, 128, 128, 1> (67 KB, 192 VGPRs, two per CU) takes 8.8 and 64-row rounds at three per CU 8.4: the launch is a chain of barrier-
// separated phases with 64 of 256 threads marching, and lives on the workgroups a CU can switch between (four per CU: no further gain).
template <uint32_t RAYS, uint32_t ROWS, uint32_t FT, int OCC>
__global__ void __launch_bounds__(kHeadBlock, OCC) k_infer_vm_persistent(HeadArgs a, VmTables tb, InferImageArgs q) {
    extern __shared__ __align__(16) half_t lds[];
    HeadLds<KIND_VM> W;
    W.carve(lds);
    half_t *feat = lds + ((HeadLds<KIND_VM>::halfs + 7) & ~7);                          // [FT][kVmInfStride] plane x line products
    float *pos = reinterpret_cast<float *>(feat + FT * kVmInfStride);                  // [ROWS][3]
    float *sraw = pos + 3 * ROWS;                                                      // [ROWS] raw sigma feature
    float *sig = sraw + ROWS;                                                          // [ROWS]
    float *rgb = sig + ROWS;                                                           // [ROWS][3]
    float *sdir = rgb + 3 * ROWS;                                                      // [kHeadBlock][3]: direction of the ray in slot s
    uint32_t *row_slot = reinterpret_cast<uint32_t *>(sdir + 3 * kHeadBlock);          // [ROWS]
    uint32_t *wcnt = row_slot + ROWS;                                                  // [8] scan scratch + queue hand-off
    const uint32_t tid = threadIdx.x, lane = tid & 63u, hi = lane >> 4, wave = tid >> 6;
    if (a.image) { copy_image_dma_static<HeadLds<KIND_VM>::halfs>(lds, a.image, tid); __builtin_amdgcn_s_waitcnt(0x0f70); }
    else W.load(a, tid, kHeadBlock);
    for (uint32_t i = tid; i < FT * 4; i += kHeadBlock)  // the row padding (halfs 144..151) is never read; keep it defined
        *reinterpret_cast<uint32_t *>(feat + (i >> 2) * kVmInfStride + 144 + 2 * (i & 3)) = 0u;
    const uint32_t n_ids = (uint32_t)max(*q.n_ids, 0);
    // the lane's channel of the twelve tables
    const uint32_t kind = lane < kRs ? 0u : 1u, ch = kind ? lane - kRs : lane;
    const uint32_t R = tb.ms[kind], Rv = tb.vs[kind];
    const float *mat0 = tb.mat[kind][0] + ch, *mat1 = tb.mat[kind][1] + ch, *mat2 = tb.mat[kind][2] + ch;
    const float *vec0 = tb.vec[kind][0] + ch, *vec1 = tb.vec[kind][1] + ch, *vec2 = tb.vec[kind][2] + ch;
    // queue position -> ray: position * mul mod n_ids is a permutation only for a multiplier coprime to n_ids
    uint32_t mul = q.shuffle % max(n_ids, 1u);
    for (;; mul++) {
        uint32_t x = max(mul, 1u), y = max(n_ids, 1u);
        while (y) { const uint32_t r = x % y; x = y; y = r; }
        if (x == 1u || n_ids <= 1u) break;
    }
    mul = max(mul, 1u);

    int32_t index = -1;
    uint32_t taken = 0, cnt = 0;
    float t = 0.f, tt = 0.f, far = 0.f, ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    float sx[kInfSteps], sy[kInfSteps], sz[kInfSteps], sdt[kInfSteps], stt[kInfSteps];
    float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 1.f};
    bool queue_done = n_ids == 0;

    auto scan_of = [&](uint32_t v, uint32_t &total) -> uint32_t {  // exclusive prefix over the workgroup's threads and the total
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
            if ((int)lane >= d) inc += up;
        }
        if (lane == 63) wcnt[wave] = inc;
        __syncthreads();
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < kHeadBlock / 64; w++) { const uint32_t c = wcnt[w]; before += w < wave ? c : 0u; tot += c; }
        __syncthreads();
        total = tot;
        return before + inc - v;
    };
    auto retire = [&]() {
        q.weights_sum[index] = ws;
        q.depth[index] = dep;
        q.image[3 * (size_t)index] = cr; q.image[3 * (size_t)index + 1] = cg; q.image[3 * (size_t)index + 2] = cb;
        index = -1; cnt = 0;
    };

    uint32_t n_rounds = 0, n_rows = 0, n_walk = 0;
#ifdef PVD_INFER_PROFILE
    long long ph[5] = {0, 0, 0, 0, 0}, tm = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();  // weights
    for (;;) {
        // ---------------- refill
        uint32_t nfree;
        const uint32_t frank = scan_of(tid < RAYS && index < 0 ? 1u : 0u, nfree);
        if (!queue_done && (nfree >= RAYS / 8 || nfree == RAYS)) {
            if (tid == 0) wcnt[4] = (uint32_t)atomicAdd(q.queue, (int32_t)nfree);
            __syncthreads();
            const uint32_t base = wcnt[4];
            __syncthreads();
            if (base + nfree >= n_ids) queue_done = true;
            if (tid < RAYS && index < 0 && base + frank < n_ids) {
                const int32_t id = q.ray_ids[(uint32_t)(((uint64_t)(base + frank) * mul) % n_ids)];
                index = id; taken = 0; cnt = 0;
#pragma unroll
                for (int c = 0; c < 3; c++) { ro[c] = q.rays_o[3 * (size_t)id + c]; rd[c] = q.rays_d[3 * (size_t)id + c]; sdir[3 * tid + c] = rd[c]; }
                t = q.nears[id]; far = q.fars[id]; tt = q.t_first[id];
                ws = dep = cr = cg = cb = 0.f;
            }
        }
        uint32_t live_now;
        (void)scan_of(index >= 0 ? 1u : 0u, live_now);
        if (live_now == 0) {
            if (queue_done) break;
            continue;
        }
        const uint32_t n_step = max(min(ROWS / live_now, kInfSteps), 1u);  // the reference's rule (renderer.py:493), the workgroup's numbers
        PVD_ISTAMP(0);
        // ---------------- march (k_march_rays' loop, raymarching.cu:756-810, perturb = 0)
        if (index >= 0) {
            Dda r;
            r.init(ro, rd, q.bound, q.dt_gamma, q.max_steps, q.C, q.H, q.grid);
            for (uint32_t pb = 0; pb < n_step + kInfProbes && cnt < n_step && tt < far; pb++) {
                float x, y, z, dt, tn;
                if (r.probe(tt, x, y, z, dt, tn)) {
                    tt += dt;
#pragma unroll
                    for (uint32_t k = 0; k < kInfSteps; k++)
                        if (k == cnt) { sx[k] = x; sy[k] = y; sz[k] = z; sdt[k] = dt; stt[k] = tt; }
                    cnt++;
                } else {
                    tt = tn;
                }
            }
            if (cnt == 0 && !(tt < far)) retire();
        }
        uint32_t rows;
        const uint32_t row0 = scan_of(cnt, rows);
        n_rounds++; n_rows += rows;
        if (rows == 0) { n_walk++; continue; }
#pragma unroll
        for (uint32_t k = 0; k < kInfSteps; k++)
            if (k < cnt) {
                pos[3 * (row0 + k)] = sx[k]; pos[3 * (row0 + k) + 1] = sy[k]; pos[3 * (row0 + k) + 2] = sz[k];
                row_slot[row0 + k] = tid;
            }
        __syncthreads();
        PVD_ISTAMP(1);
        // ---------------- shade, FT rows of the round at a time (the feature tile holds FT rows; positions / results all ROWS):
        // the VM lookup of the pass's rows, wave w: rows w, w + 4, ... (lane j holds the state of its j-th row)
        for (uint32_t base = 0; base < rows; base += FT) {
        const uint32_t sub = min(FT, rows - base);
        {
            const uint32_t mine = sub > wave ? (sub - wave + 3u) / 4u : 0u;  // (uniform per wave; <= FT / 4 <= 64)
            float xn[3] = {0.f, 0.f, 0.f};
            if (lane < mine) normalise(pos, (size_t)(base + wave + 4u * lane), tb, xn);
            const SampleCtl ctl = sample_ctl(xn, tb);
            constexpr int kDepth = PVD_INFER_VM_DEPTH;  // register slots of 18 loads: depth - 1 rows' loads in flight while one row is formed
            Taps6 tp[kDepth][3];
            auto issue = [&](Taps6(&dst)[3], uint32_t j) __attribute__((always_inline)) {
                j = min(j, mine - 1u);  // (a row past the wave's last repeats it: same loads, same bytes written again)
                issue6<0>(ctl, j, mat0, vec0, R, Rv, (int)tb.W[0], dst[0]);
                issue6<1>(ctl, j, mat1, vec1, R, Rv, (int)tb.W[1], dst[1]);
                issue6<2>(ctl, j, mat2, vec2, R, Rv, (int)tb.W[2], dst[2]);
            };
            auto finish = [&](const Taps6(&src)[3], uint32_t j) __attribute__((always_inline)) {
                j = min(j, mine - 1u);
                const uint32_t rw = wave + 4u * j;
                const float p0 = finish6<0>(ctl, j, src[0]), p1 = finish6<1>(ctl, j, src[1]), p2 = finish6<2>(ctl, j, src[2]);
                float s = 0.f;
                if (kind) {
                    half_t *row = feat + rw * kVmInfStride + ch;
                    row[0] = (half_t)p0; row[kRc] = (half_t)p1; row[2 * kRc] = (half_t)p2;
                } else {
                    s += p0; s += p1; s += p2;
                }
                // k_vm_fwd's xor butterfly over the 16 sigma lanes, as row rotations (same operands, same sums)
                s += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s), 0x128, 0xf, 0xf, false));
                s += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s), 0x124, 0xf, 0xf, false));
                s += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s), 0x122, 0xf, 0xf, false));
                s += __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(s), 0x121, 0xf, 0xf, false));
                if (lane == 0) sraw[base + rw] = s;
            };
            if (mine > 0) {
#pragma unroll
                for (int dd = 0; dd < kDepth; dd++) issue(tp[dd], (uint32_t)dd);
                for (uint32_t j = 0; j < mine; j += kDepth) {
#pragma unroll
                    for (int dd = 0; dd < kDepth; dd++) {
                        finish(tp[dd], j + dd);
                        issue(tp[dd], j + dd + kDepth);  // this slot's registers are requested again at once
                    }
                }
            }
        }
        __syncthreads();
        PVD_ISTAMP(2);
        // ---------------- ... and the head, 16 rows per wave and pass; sigma / rgb stay in LDS
        for (uint32_t t16 = wave; t16 * 16 < sub; t16 += kHeadBlock / 64) {
            const uint32_t fr = min(t16 * 16 + (lane & 15), FT - 1u), rw = base + fr;  // row of the feature tile / of the round
            const bool live = t16 * 16 + (lane & 15) < sub;
            TileIn<KIND_VM> in;
            const h4 hz = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
#pragma unroll
            for (int s2 = 0; s2 < 9; s2++) in.x[s2] = live ? *reinterpret_cast<const h4 *>(feat + fr * kVmInfStride + 16 * s2 + 4 * hi) : hz;
            in.sraw = (live && hi == 0) ? sraw[rw] : 0.f;
            const uint32_t slot = live ? row_slot[rw] : 0u;
            in.dx = live ? sdir[3 * slot] : 0.f; in.dy = live ? sdir[3 * slot + 1] : 0.f; in.dz = live ? sdir[3 * slot + 2] : 0.f;
            TileFwd tf;
            head_forward_tile<KIND_VM>(a, W, in, lane, tf);
            if (hi == 0 && live) {
                sig[rw] = __expf(tf.F.x);
                rgb[3 * rw] = sigmoid_h(tf.out.x); rgb[3 * rw + 1] = sigmoid_h(tf.out.y); rgb[3 * rw + 2] = sigmoid_h(tf.out.z);
            }
        }
        __syncthreads();  // (the next pass rewrites the feature tile)
        PVD_ISTAMP(3);
        }
        // ---------------- blend (k_composite_rays' loop, raymarching.cu:858-899; sigma scaled as renderer.py:528)
        if (cnt > 0) {
            bool done = false;
#pragma unroll
            for (uint32_t k = 0; k < kInfSteps; k++) {
                if (k < cnt && !done) {
                    const uint32_t rw = row0 + k;
                    const float alpha = 1.0f - __expf(-(q.sigma_scale * sig[rw]) * sdt[k]);
                    const float T = 1 - ws;
                    const float w = alpha * T;
                    ws += w;
                    t += stt[k] - t;
                    dep += w * t;
                    cr += w * rgb[3 * rw]; cg += w * rgb[3 * rw + 1]; cb += w * rgb[3 * rw + 2];
                    taken++;
                    if ((double)T < 1e-4 || taken >= q.max_steps) done = true;
                }
            }
            cnt = 0;
            if (done) retire();
        }
        __syncthreads();  // the next round rewrites the tile
        PVD_ISTAMP(4);
    }
#ifdef PVD_INFER_PROFILE
    if (tid == 0 && blockIdx.x == 0) for (int z = 0; z < 5; z++) q.stats[4 + z] = (int32_t)ph[z];
#endif
    if (tid == 0 && n_rounds > 1) {
        atomicAdd(q.stats + 0, (int32_t)n_rounds); atomicAdd(q.stats + 1, (int32_t)n_rows); atomicAdd(q.stats + 2, (int32_t)n_walk); atomicAdd(q.stats + 3, 1);
    }
}

int infer_prepare(InferImageArgs &q, const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N,
                  const uint8_t *bitfield, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, float sigma_scale,
                  int32_t *workspace, float *weights_sum, float *depth, float *image_out, hipStream_t s) {
    if (hipMemsetAsync(workspace, 0, 2 * sizeof(int32_t), s) != hipSuccess) return PVD_ERR_LAUNCH;
    if (hipMemsetAsync(workspace + 2 + 2 * (size_t)N, 0, 10 * sizeof(int32_t), s) != hipSuccess) return PVD_ERR_LAUNCH;
    uint32_t hb = div_up(N, kHeadBlock);
    if (hb > 4096) hb = 4096;
    q.rays_o = rays_o; q.rays_d = rays_d; q.nears = nears; q.fars = fars; q.ray_ids = workspace + 2; q.n_ids = workspace; q.queue = workspace + 1;
    q.grid = bitfield; q.bound = bound; q.dt_gamma = dt_gamma; q.sigma_scale = sigma_scale; q.max_steps = max_steps; q.C = C; q.H = H;
    q.weights_sum = weights_sum; q.depth = depth; q.image = image_out;
    q.t_first = reinterpret_cast<const float *>(workspace + 2 + N);
    q.stats = workspace + 2 + 2 * (size_t)N;
    q.shuffle = 7919u;
    if (const char *e = getenv("PVD_INFER_SHUFFLE")) q.shuffle = (uint32_t)max(atoi(e), 1);
    hipLaunchKernelGGL(k_infer_first_hit, dim3(hb), dim3(kHeadBlock), 0, s, q, N, reinterpret_cast<float *>(workspace + 2 + N), workspace + 2, workspace);
    return PVD_OK;
}

template <uint32_t ROWS, uint32_t FT>
static size_t infer_vm_persistent_lds_bytes() {
    return (((size_t)HeadLds<KIND_VM>::halfs + 7) & ~(size_t)7) * sizeof(half_t) + (size_t)FT * kVmInfStride * sizeof(half_t) +
           sizeof(float) * (3 * ROWS + ROWS + ROWS + 3 * ROWS + 3 * kHeadBlock) + sizeof(uint32_t) * (ROWS + 8);
}

static size_t infer_persistent_lds_bytes() {
    return (((size_t)HeadLds<KIND_HASH>::halfs + 7) & ~(size_t)7) * sizeof(half_t) + (size_t)kInfRows * kFeatStride * sizeof(half_t) +
           sizeof(float) * (3 * kInfRows + kInfRows + 3 * kInfRows + 3 * kHeadBlock) + sizeof(uint32_t) * (kInfRows + 8);
}

// ------------------------------------------------------------------------------------------------------------------
// The frozen `mlp` model (vanilla NeRF trunk, network.py:154-182 / forward_nerf_mlp, then the same sigma / colour head) in one
// launch: positional encoding [M][64] f16 in (pvd_freq_encode), sigma / rgb / feature_sigma_color out.  The reference runs it
// as ~10 library GEMMs with their activations round-tripping HBM (94 MB per 256-wide layer at 93 k samples, 38 us each on
// MI355X); here a wave keeps kMlpTS 16-sample tiles of activations in registers for the whole network -- computing
// Y^T = W X^T makes a layer's D tiles the next layer's B fragments, as in the head -- and the workgroup streams the weights
// through LDS in chunks of 64 output rows (<= 41.6 KB), double-buffered by LDS-DMA, so that every A fragment read from LDS
// feeds kMlpTS MFMAs.  Bias in the accumulator's initial value, ReLU and the rounding to f16 on the accumulator: the
// autocast formulation's arithmetic (f16 GEMM, f32 accumulation, f16 result per layer).
#ifndef PVD_MLP_TS
#define PVD_MLP_TS 3
#endif
#ifndef PVD_MLP_WAVES
#define PVD_MLP_WAVES 4
#endif
constexpr int kMlpTS = PVD_MLP_TS;    // 16-sample tiles per wave (3: 48 samples; a workgroup of 4 waves = 192 samples)
constexpr uint32_t kMlpBlock = 64u * PVD_MLP_WAVES;
constexpr int kMlpW = 256;            // hidden width (the reference's nerf_layer_wide default, main_distill_mutual.py)
constexpr int kMlpIn = 64;            // positional encoding, padded (63 -> 64)
constexpr int kMlpChunkRows = 64;     // output rows per weight chunk = 4 MFMA tiles
constexpr int kMlpPad = 8;            // halfs of row padding: row stride (K + 8) / 2 dwords = 4 x odd for K = 64, 256 and 320, so the 16 lanes of
                                      // a 16-byte read pass (one row each) start on 16 different multiples of 4 banks: conflict-free
constexpr int kMlpMaxChunkHalfs = kMlpChunkRows * (kMlpIn + kMlpW + kMlpPad) + kMlpChunkRows;  // the skip layer's chunk: 21056 halfs

struct MlpArgs {
    const half_t *pts;     // [M][64] f16 positional encoding (zero-padded column 63)
    const half_t *wstream; // weight chunks in execution order (see mlp_chunk_halfs), each = rows x (K + kMlpPad) halfs, then rows bias halfs
    uint32_t n_before;     // hidden 256 -> 256 layers before the skip connection is concatenated (= args.skip)
    uint32_t n_after;      // hidden 256 -> 256 layers after the skip layer
};

__host__ __device__ constexpr int mlp_chunk_halfs(int rows, int K) { return rows * (K + kMlpPad) + rows; }

// DMA of one chunk into an LDS buffer (lane-linear 16-byte pieces, like copy_image_dma)
__device__ __forceinline__ void mlp_dma(half_t *__restrict__ buf, const half_t *__restrict__ src, int halfs, uint32_t tid) {
    copy_image_dma<kMlpBlock>(buf, src, halfs, tid);
}

// One chunk of NT output tiles: acc = bias; acc += W[chunk rows, :] . X^T over KP k-steps of `pts` then KX k-steps of `x`.
// Two k-steps per MFMA: v_mfma_f32_16x16x32_f16 fed with the two K = 16 fragments of A and of B side by side contracts the
// same k set (a permutation of the contraction index applied to both operands; checked by tools/probes/mfma32_probe.hip).
// This kernel, unlike the heads, is MFMA-bound, and the K = 32 form is the one gfx950 issues at full rate.
typedef half_t h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f4 mfma_k32(h4 a0, h4 a1, h4 b0, h4 b1, f4 c) {
    const h8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const h8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <int KP, int KX, int NT, bool RELU>
__device__ __forceinline__ void mlp_chunk(const half_t *__restrict__ buf, const h4 (&pts)[kMlpTS][kMlpIn / 16], const h4 (&x)[kMlpTS][kMlpW / 16],
                                          h4 (&out)[kMlpTS][NT], uint32_t lane) {
    constexpr int K = 16 * (KP + KX), stride = K + kMlpPad, KS = KP + KX;
    static_assert(KS % 2 == 0, "k-steps are taken in pairs");
    const uint32_t r = lane & 15u, hi = lane >> 4;
    const half_t *__restrict__ bias = buf + 16 * NT * stride;
    f4 acc[kMlpTS][NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const h4 b = *reinterpret_cast<const h4 *>(bias + 16 * nt + 4 * hi);
        const f4 bf = {(float)b.x, (float)b.y, (float)b.z, (float)b.w};
#pragma unroll
        for (int ts = 0; ts < kMlpTS; ts++) acc[ts][nt] = bf;
    }
    // A operands: the weight stream stores, for every row and every pair of k-steps, the 8 halfs a lane feeds to one K = 32
    // MFMA next to each other ([pair][hi][k-step of the pair][4]): one 16-byte LDS read per output tile and pair
    const half_t *__restrict__ arow = buf + r * stride + 8 * hi;
    h8 an[NT];  // the pair in flight; the next one is requested before this one is consumed
#pragma unroll
    for (int nt = 0; nt < NT; nt++) an[nt] = *reinterpret_cast<const h8 *>(arow + 16 * nt * stride);
#pragma unroll
    for (int k = 0; k < KS; k += 2) {
        h8 ac[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) ac[nt] = an[nt];
        if (k + 2 < KS) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) an[nt] = *reinterpret_cast<const h8 *>(arow + 16 * nt * stride + 16 * (k + 2));
        }
#pragma unroll
        for (int ts = 0; ts < kMlpTS; ts++) {
            const h4 b0 = k < KP ? pts[ts][k < KP ? k : 0] : x[ts][k >= KP ? k - KP : 0];
            const h4 b1 = k + 1 < KP ? pts[ts][k + 1 < KP ? k + 1 : 0] : x[ts][k + 1 >= KP ? k + 1 - KP : 0];
            const h8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[ts][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[nt], b, acc[ts][nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int ts = 0; ts < kMlpTS; ts++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) out[ts][nt] = RELU ? relu_h4(to_h4(acc[ts][nt])) : to_h4(acc[ts][nt]);
}

// The chunk pipeline: chunk q is computed from buffer q % 3 while chunks q + 1 and q + 2 are landing in the other two (one chunk
// of compute, ~0.7 us, is shorter than the ~1-2 us an LDS-DMA batch takes to arrive from L2).  Hand-made barrier: __syncthreads()
// would wait for EVERY outstanding load (vmcnt(0)), i.e. also for the chunks that were just requested.
template <int N>
__device__ __forceinline__ void mlp_wait_barrier() {
    // own DMA pieces up to the wanted chunk have landed (memory operations complete in order: at most N younger ones may still
    // be in flight), own LDS reads of the buffer about to be refilled are done; then meet the other waves
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ uint32_t mlp_dma_count(int halfs, uint32_t wave) {  // LDS-DMA instructions THIS wave issues for a chunk
    const int pieces = (halfs * 2 + 1023) / 1024, nw = (int)(kMlpBlock / 64);
    return pieces > (int)wave ? (uint32_t)(pieces - (int)wave + nw - 1) / (uint32_t)nw : 0u;
}
struct MlpStream {
    half_t *buf[3];
    const half_t *next;  // source of the next chunk to FETCH
    uint32_t q;          // chunks computed so far
    uint32_t young;      // this wave's DMA instructions of the most recently requested chunk (q + 1)
    // make chunk q usable, request chunk q + 2 (`next2_halfs`: its size, 0 = none)
    __device__ __forceinline__ const half_t *acquire(int next2_halfs, uint32_t tid) {
        // chunk q is complete once at most `young` younger DMA instructions of this wave are outstanding (rounded down to a
        // value that exists as an immediate: waiting for more is always safe)
        if (young >= 10u) mlp_wait_barrier<10>();
        else if (young >= 8u) mlp_wait_barrier<8>();
        else if (young >= 4u) mlp_wait_barrier<4>();
        else if (young >= 2u) mlp_wait_barrier<2>();
        else mlp_wait_barrier<0>();
        // every wave is past its reads of buffer (q + 2) % 3 (chunk q - 1): refill it
        young = 0u;
        if (next2_halfs > 0) {
            mlp_dma(buf[(q + 2u) % 3u], next, next2_halfs, tid);
            next += next2_halfs;
            young = mlp_dma_count(next2_halfs, tid >> 6);
        }
        const half_t *cur = buf[q % 3u];
        q++;
        return cur;
    }
};

// one 256-wide layer = 4 chunks; `after1` / `after2`: sizes of the first two chunks of what FOLLOWS this layer (0: none) -- the
// requests run two chunks ahead of the compute
template <int KP, int KX>
__device__ __forceinline__ void mlp_layer(MlpStream &st, const h4 (&pts)[kMlpTS][kMlpIn / 16], const h4 (&x)[kMlpTS][kMlpW / 16],
                                          h4 (&y)[kMlpTS][kMlpW / 16], int after1, int after2, uint32_t tid, uint32_t lane) {
    constexpr int own = mlp_chunk_halfs(kMlpChunkRows, 16 * (KP + KX));
    constexpr int nc = kMlpW / kMlpChunkRows;
#pragma unroll
    for (int c = 0; c < nc; c++) {
        const half_t *cur = st.acquire(c + 2 < nc ? own : (c + 2 == nc ? after1 : after2), tid);
        h4 o[kMlpTS][4];
        mlp_chunk<KP, KX, 4, true>(cur, pts, x, o, lane);
#pragma unroll
        for (int ts = 0; ts < kMlpTS; ts++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++) y[ts][4 * c + nt] = o[ts][nt];
    }
}

__global__ void __launch_bounds__(kMlpBlock, 1) k_mlp_fwd_fused(HeadArgs a, MlpArgs m) {
    extern __shared__ __align__(16) half_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, hi = lane >> 4, wave = tid >> 6;
    constexpr int kBufHalfs = (kMlpMaxChunkHalfs + 7) & ~7;
    MlpStream st;
    st.buf[0] = lds; st.buf[1] = lds + kBufHalfs; st.buf[2] = lds + 2 * kBufHalfs; st.q = 0;
    half_t *headw = lds + 3 * kBufHalfs;
    HeadLds<KIND_HASH> W;
    W.carve(headw);
    // the first chunk of the first layer and the head's weights start moving now
    constexpr int h_first = mlp_chunk_halfs(kMlpChunkRows, kMlpIn), h_hidden = mlp_chunk_halfs(kMlpChunkRows, kMlpW),
                  h_skip = mlp_chunk_halfs(kMlpChunkRows, kMlpIn + kMlpW), h_last = mlp_chunk_halfs(32, kMlpW);
    st.next = m.wstream;
    if (a.image) copy_image_dma<kMlpBlock>(headw, a.image, HeadLds<KIND_HASH>::halfs, tid);  // (older than every chunk: complete by the first acquire)
    else W.load(a, tid, kMlpBlock);
    mlp_dma(st.buf[0], st.next, h_first, tid);
    st.next += h_first;
    mlp_dma(st.buf[1], st.next, h_first, tid);
    st.next += h_first;
    st.young = mlp_dma_count(h_first, wave);
    // ---- inputs of this wave's kMlpTS tiles
    const size_t base = ((size_t)blockIdx.x * (kMlpBlock / 64) + wave) * (16 * kMlpTS);
    h4 pts[kMlpTS][kMlpIn / 16];
    float dir[kMlpTS][3];
    const h4 hz = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
#pragma unroll
    for (int ts = 0; ts < kMlpTS; ts++) {
        const size_t b = base + 16 * ts + (lane & 15u);
        const bool valid = b < a.M;
#pragma unroll
        for (int k = 0; k < kMlpIn / 16; k++) pts[ts][k] = valid ? *reinterpret_cast<const h4 *>(m.pts + b * kMlpIn + 16 * k + 4 * hi) : hz;
#pragma unroll
        for (int c = 0; c < 3; c++) dir[ts][c] = valid ? a.dirs[3 * b + c] : 0.f;
    }
    h4 xa[kMlpTS][kMlpW / 16], xb[kMlpTS][kMlpW / 16];
#pragma unroll
    for (int ts = 0; ts < kMlpTS; ts++)
#pragma unroll
        for (int k = 0; k < kMlpW / 16; k++) xa[ts][k] = hz;
    // ---- the trunk.  (`after` = size of the chunk that follows each layer: what its last acquire must start fetching)
    // (sizes of the two chunks after each layer: the next layer's first two, all of one kind -- every 256-row layer has four)
    mlp_layer<kMlpIn / 16, 0>(st, pts, xa, xb, m.n_before > 0 ? h_hidden : h_skip, m.n_before > 0 ? h_hidden : h_skip, tid, lane);  // 64 -> 256
    for (uint32_t l = 0; l < m.n_before; l++) {
#pragma unroll
        for (int ts = 0; ts < kMlpTS; ts++)
#pragma unroll
            for (int k = 0; k < kMlpW / 16; k++) xa[ts][k] = xb[ts][k];
        mlp_layer<0, kMlpW / 16>(st, pts, xa, xb, l + 1 < m.n_before ? h_hidden : h_skip, l + 1 < m.n_before ? h_hidden : h_skip, tid, lane);
    }
#pragma unroll
    for (int ts = 0; ts < kMlpTS; ts++)
#pragma unroll
        for (int k = 0; k < kMlpW / 16; k++) xa[ts][k] = xb[ts][k];
    mlp_layer<kMlpIn / 16, kMlpW / 16>(st, pts, xa, xb, m.n_after > 0 ? h_hidden : h_last, m.n_after > 0 ? h_hidden : 0, tid, lane);  // [pts | x] 320 -> 256
    for (uint32_t l = 0; l < m.n_after; l++) {
#pragma unroll
        for (int ts = 0; ts < kMlpTS; ts++)
#pragma unroll
            for (int k = 0; k < kMlpW / 16; k++) xa[ts][k] = xb[ts][k];
        mlp_layer<0, kMlpW / 16>(st, pts, xa, xb, l + 1 < m.n_after ? h_hidden : h_last, l + 1 < m.n_after ? h_hidden : 0, tid, lane);
    }
    // ---- last layer: 256 -> 28 (two tiles, no ReLU) = the head's input fragments
    const half_t *cur = st.acquire(0, tid);
    h4 feat[kMlpTS][2];
    mlp_chunk<0, kMlpW / 16, 2, false>(cur, pts, xb, feat, lane);
    // ---- sigma / colour head on each tile (weights: DMA'd at the start, visible since the first acquire's barrier)
#pragma unroll
    for (int ts = 0; ts < kMlpTS; ts++) {
        const size_t b = base + 16 * ts + (lane & 15u);
        const bool valid = b < a.M;
        TileIn<KIND_HASH> in;
        in.x[0] = feat[ts][0]; in.x[1] = feat[ts][1];
        in.sraw = 0.f; in.dx = dir[ts][0]; in.dy = dir[ts][1]; in.dz = dir[ts][2];
        TileFwd t;
        head_forward_tile<KIND_HASH>(a, W, in, lane, t);
        if (valid) {
            *reinterpret_cast<f4 *>(a.feat16 + b * 16 + 4 * hi) = t.F;
            if (hi == 0) {
                a.sigma[b] = __expf(t.F.x);
                a.rgb[3 * b] = sigmoid_h(t.out.x);
                a.rgb[3 * b + 1] = sigmoid_h(t.out.y);
                a.rgb[3 * b + 2] = sigmoid_h(t.out.z);
            }
        }
    }
}

static int launch_mlp_fwd_fused(const HeadArgs &a, const MlpArgs &m, hipStream_t s) {
    const uint32_t per_wg = (kMlpBlock / 64) * 16 * kMlpTS;
    const uint32_t blocks = div_up(a.M, per_wg);
    constexpr int kBufHalfs = (kMlpMaxChunkHalfs + 7) & ~7;
    const size_t lds_bytes = (3 * (size_t)kBufHalfs + (((size_t)HeadLds<KIND_HASH>::halfs + 7) & ~(size_t)7)) * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_mlp_fwd_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return PVD_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_mlp_fwd_fused, dim3(blocks), dim3(kMlpBlock), lds_bytes, s, a, m);
    return check_launch();
}

static int launch_hash_fwd_fused(const HeadArgs &a, const FusedLookup &g, hipStream_t s) {
    const uint32_t nchunks = div_up(a.M, kFusedTile);
    uint32_t blocks = nchunks;
    if (blocks > 256u * 4u) blocks = 256u * 4u;  // persistent beyond 4 workgroups per CU
    const size_t lds_bytes = (((size_t)HeadLds<KIND_HASH>::halfs + 7) & ~(size_t)7) * sizeof(half_t) + kFusedTile * kFeatStride * sizeof(half_t);
    // PVD_FUSED_DMA (measurement; read per launch so that one process can A/B): where the weight image's LDS-DMA is issued.  Same bits.
    int dma = 0;
    if (const char *e = getenv("PVD_FUSED_DMA")) dma = atoi(e) == 1 ? 1 : 0;
    FusedRes gr;
    for (uint32_t l = 0; l < 14; l++) gr.res[l] = (uint32_t)ceil((double)g.scales.scale[l]) + 1u;
    if (dma == 0) hipLaunchKernelGGL((k_hash_fwd_fused<14, 0>), dim3(blocks), dim3(kHeadBlock), lds_bytes, s, a, g, gr);
    else hipLaunchKernelGGL((k_hash_fwd_fused<14, 1>), dim3(blocks), dim3(kHeadBlock), lds_bytes, s, a, g, gr);
    return check_launch();
}


// ====================================================================== backward (training)

// accumulator tiles per wave: [stage-A weights][colour 1: 4x2][colour 2: 4x4][colour 3: 1x4]
//   VM   stage A = basis_mat  (1 x 9 tiles);   hash stage A = sigma_net.0 (4 x 2) then sigma_net.1 (1 x 4)
template <int KIND>
struct DwLayout {
    static constexpr int a = KIND == KIND_VM ? 9 : 12;
    static constexpr int c1 = a, c2 = a + 8, c3 = a + 24, tiles = a + 28;
    static constexpr int floats = tiles * 256;
};

struct HeadBwdArgs {
    HeadArgs f;
    const float *g_sigma;   // [M]      d loss / d sigma
    const float *g_rgb;     // [M][3]
    const float *g_rgb2;    // [M][3] or null: a second consumer's gradient of the same rgb output, added while loading
    const float *g_feat16;  // [M][16]
    float *g_sigma_raw;     // VM: [M]
    half_t *g_x0;           // VM: d loss / d products [M][144];  hash: d loss / d encoder output [14][M][2]
    float *partials;        // [blocks][DwLayout::floats]
};

// 16x16 transpose of a register tile through LDS: in = X[row 4hi+j][col l&15]  ->  out = X[row l&15][col 4hi+j].
// Lane (c, hi) parks its four values -- X[4hi .. 4hi+3][c] -- as ONE 8-byte chunk of row c of T = X^T (ds_write_b64), and
// ds_read_b64_tr_b16 hands every 16-lane group the [4][16] block T[4hi .. 4hi+3][0 .. 15] column-major: lane n of group hi receives
// T[4hi + j][n] = X[n][4hi + j], j = 0..3 (lane i of the group addresses the chunk (row 4hi + i/4, columns 4(i%4) ..)).  Two LDS
// instructions per tile instead of four 2-byte stores and a read.  The four chunks of row c sit at chunk position hi ^ (c >> 2): a
// 16-lane store group (one hi, c = 0..15) then covers all 32 store banks once, and a 32-lane read group (rows 8 hi' .. 8 hi' + 7,
// every chunk once) all 64 read banks once -- the row-major [16][16] halfs of rounds 2-4 put lanes c and c + 4 (stores) and rows
// r and r + 8 (reads) on one bank: SQ_LDS_BANK_CONFLICT 1.4 cycles per LDS instruction (profiles/r04_pmc_step_kernels.csv).
#ifndef PVD_HEAD_TR
#define PVD_HEAD_TR 1
#endif
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 transpose_tile(h4 v, half_t *__restrict__ scratch, uint32_t lane) {
    const uint32_t c = lane & 15, hi = lane >> 4;
#if PVD_HEAD_TR
    *reinterpret_cast<h4 *>(scratch + c * 16 + 4 * (hi ^ (c >> 2))) = v;
    __builtin_amdgcn_wave_barrier();
    const uint32_t row = 4 * hi + (c >> 2);
    const fp16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4_t *)(scratch + row * 16 + 4 * ((c & 3) ^ hi)));
    __builtin_amdgcn_wave_barrier();
    h4 r;
    __builtin_memcpy(&r, &t, sizeof(r));
    return r;
#else
    scratch[(4 * hi + 0) * 16 + c] = v.x;
    scratch[(4 * hi + 1) * 16 + c] = v.y;
    scratch[(4 * hi + 2) * 16 + c] = v.z;
    scratch[(4 * hi + 3) * 16 + c] = v.w;
    __builtin_amdgcn_wave_barrier();
    const h4 r = *reinterpret_cast<const h4 *>(scratch + c * 16 + 4 * hi);
    __builtin_amdgcn_wave_barrier();
    return r;
#endif
}

__device__ __forceinline__ h4 mask_relu(f4 g, h4 act) {  // dPre = dAct * (act > 0), rounded to f16
    const half_t z = (half_t)0.0f;
    h4 r;
    r.x = act.x > z ? (half_t)g.x : z; r.y = act.y > z ? (half_t)g.y : z;
    r.z = act.z > z ? (half_t)g.z : z; r.w = act.w > z ? (half_t)g.w : z;
    return r;
}

#ifndef PVD_HEAD_NT
#define PVD_HEAD_NT 1
#endif
typedef half_t h8 __attribute__((ext_vector_type(8)));
struct TrFrag { h4 v[PVD_HEAD_NT]; };    // the transposed register tiles of the NT tiles of a trip
struct TrFragIn { h4 v[PVD_HEAD_NT]; };
// dW tile += sum over the samples of all NT tiles: one K = 32 MFMA for two tiles, a chain of K = 16 ones otherwise
__device__ __forceinline__ f4 mfma_nt(const TrFrag &A, const TrFrag &B, f4 c) {
#if PVD_HEAD_NT == 2
    const h8 a8 = __builtin_shufflevector(A.v[0], A.v[1], 0, 1, 2, 3, 4, 5, 6, 7), b8 = __builtin_shufflevector(B.v[0], B.v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
#else
#pragma unroll
    for (int u = 0; u < PVD_HEAD_NT; u++) c = mfma(A.v[u], B.v[u], c);
    return c;
#endif
}

template <int KIND, int OCC>
__global__ void __launch_bounds__(kHeadBlock, OCC) k_head_bwd(HeadBwdArgs a) {
    extern __shared__ __align__(16) half_t lds[];
    HeadLds<KIND> W;
    W.carve(lds);
    HeadLdsT<KIND> T;
    T.carve(lds + HeadLds<KIND>::halfs);
    const LdsMat Wa1T = T.Wa1T, Wa2T = T.Wa2T, Wc1T = T.Wc1T, Wc2T = T.Wc2T, Wc3T = T.Wc3T;
    half_t *scratch = lds + HeadLds<KIND>::halfs + HeadLdsT<KIND>::halfs + (threadIdx.x >> 6) * 512;  // two 512-byte transpose pieces per wave
#ifdef PVD_HEAD_PROFILE
    long long *stamps = reinterpret_cast<long long *>(a.partials + (size_t)gridDim.x * DwLayout<KIND>::floats);
    int n_stamp = 0;
#define PVD_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0 && n_stamp < 30) stamps[n_stamp] = (long long)__builtin_readcyclecounter(); n_stamp++; } while (0)
#else
#define PVD_STAMP() do { } while (0)
#endif
    PVD_STAMP();
    const uint32_t lane = threadIdx.x & 63u, hi = lane >> 4;
    const uint32_t wave = (blockIdx.x * kHeadBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = gridDim.x * (kHeadBlock / 64);
    const uint32_t ntiles = div_up(a.f.M, 16u);
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const h4 hzero = {(half_t)0, (half_t)0, (half_t)0, (half_t)0};
    if (a.f.image) copy_image_dma_static<HeadLds<KIND>::halfs + HeadLdsT<KIND>::halfs>(lds, a.f.image, threadIdx.x);  // (see k_head_fwd)

    using LY = DwLayout<KIND>;
    f4 dWa[LY::a], dW1[8], dW2[16], dW3[4];
#pragma unroll
    for (int i = 0; i < LY::a; i++) dWa[i] = zero;
#pragma unroll
    for (int i = 0; i < 8; i++) dW1[i] = zero;
#pragma unroll
    for (int i = 0; i < 16; i++) dW2[i] = zero;
#pragma unroll
    for (int i = 0; i < 4; i++) dW3[i] = zero;

    struct TileGrad { float r, g, bl, sig; f4 feat; };
    auto load_grad = [&](size_t b, bool valid) {
        TileGrad q = {0.f, 0.f, 0.f, 0.f, zero};
        if (valid) {
            if (a.g_feat16) q.feat = *reinterpret_cast<const f4 *>(a.g_feat16 + b * 16 + 4 * hi);  // (NULL: no gradient reaches the feature rows)
            if (hi == 0) {
                q.r = a.g_rgb[3 * b]; q.g = a.g_rgb[3 * b + 1]; q.bl = a.g_rgb[3 * b + 2]; q.sig = a.g_sigma[b];
                if (a.g_rgb2) { q.r += a.g_rgb2[3 * b]; q.g += a.g_rgb2[3 * b + 1]; q.bl += a.g_rgb2[3 * b + 2]; }
            }
        }
        return q;
    };
    // NT = 2 tiles (32 consecutive samples) per trip: every weight fragment read from LDS feeds two independent MFMAs, the two
    // tiles' dependent chains (MFMA -> f16 -> mask -> MFMA ...) interleave, and a weight-gradient tile takes both tiles' samples in
    // ONE v_mfma_f32_16x16x32_f16 (k = 8 hi + 4 u + j  <->  sample 4 hi + j of tile u, the same for both operands).  One wave per
    // SIMD has nothing else to fill those latencies with: 9.2 k cycles per tile at NT = 1 (profiles/r05_head_bwd_stamps_tr.txt).
    constexpr int NT = PVD_HEAD_NT;
    const uint32_t ntrips = div_up(ntiles, (uint32_t)NT);
    TileIn<KIND> nxt[NT];
    TileGrad nxt_g[NT];
#pragma unroll
    for (int u = 0; u < NT; u++) {
        const size_t bf = ((size_t)wave * NT + u) * 16 + (lane & 15);
        nxt[u] = load_tile_in<KIND>(a.f, bf, bf < a.f.M, lane);
        nxt_g[u] = load_grad(bf, bf < a.f.M);
    }
    // (the first trip's inputs were requested behind the image's DMA: one round trip for both)
    if (a.f.image) {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the DMA has landed before the barrier releases the readers
    } else {
        W.load(a.f, threadIdx.x, kHeadBlock);
        T.load(a.f, threadIdx.x, kHeadBlock);
    }
    PVD_STAMP();
    __syncthreads();
    PVD_STAMP();
    for (uint32_t trip = wave; trip < ntrips; trip += nwaves) {
        size_t b[NT];
        bool valid[NT];
        TileIn<KIND> in[NT];
        TileGrad gin[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            b[u] = ((size_t)trip * NT + u) * 16 + (lane & 15);
            valid[u] = b[u] < a.f.M;
            in[u] = nxt[u];
            gin[u] = nxt_g[u];
        }
        if (trip + nwaves < ntrips) {  // next trip's inputs: in flight while this one computes
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const size_t bn = ((size_t)(trip + nwaves) * NT + u) * 16 + (lane & 15);
                nxt[u] = load_tile_in<KIND>(a.f, bn, bn < a.f.M, lane);
                nxt_g[u] = load_grad(bn, bn < a.f.M);
            }
        }
        TileFwd t[NT];
        PVD_STAMP();
        head_forward_tiles<KIND, NT>(a.f, W, in, lane, t);
        PVD_STAMP();

        // ---- d loss / d (colour layer 3 pre-activation): rows 0..2 live in the hi == 0 lanes
        h4 D3[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            D3[u] = hzero;
            if (valid[u] && hi == 0) {
                const float s0 = sigmoid_h(t[u].out.x), s1 = sigmoid_h(t[u].out.y), s2 = sigmoid_h(t[u].out.z);
                D3[u].x = (half_t)(gin[u].r * s0 * (1.0f - s0));
                D3[u].y = (half_t)(gin[u].g * s1 * (1.0f - s1));
                D3[u].z = (half_t)(gin[u].bl * s2 * (1.0f - s2));
            }
        }
        h4 D2[NT][4], D1[NT][4];
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const h4 A = Wc3T.afrag(n, 0, lane);
#pragma unroll
            for (int u = 0; u < NT; u++) D2[u][n] = mask_relu(mfma(A, D3[u], zero), t[u].H2[n]);
        }
#pragma unroll
        for (int n = 0; n < 4; n++) {
            f4 acc[NT];
#pragma unroll
            for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const h4 A = Wc2T.afrag(n, s, lane);
#pragma unroll
                for (int u = 0; u < NT; u++) acc[u] = mfma(A, D2[u][s], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < NT; u++) D1[u][n] = mask_relu(acc[u], t[u].H1[n]);
        }
        f4 dF[NT];  // rows 16..31 of the colour layer's input = the feature tile (row 16 has zero weights)
#pragma unroll
        for (int u = 0; u < NT; u++) dF[u] = zero;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const h4 A = Wc1T.afrag(1, s, lane);
#pragma unroll
            for (int u = 0; u < NT; u++) dF[u] = mfma(A, D1[u][s], dF[u]);
        }
        h4 Da[NT];  // gradient of stage A's output tile
        f4 dh[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            if (valid[u]) { dF[u].x += gin[u].feat.x; dF[u].y += gin[u].feat.y; dF[u].z += gin[u].feat.z; dF[u].w += gin[u].feat.w; }
            // row 0 = log-sigma: sigma = trunc_exp(F0) -> g * exp(clamp(F0, -12, 12)) (tools/activation.py:18), plus
            // whatever arrived through feature_sigma_color[:, 0]; then the sigma clamp's mask
            float g0 = 0.f;
            if (hi == 0 && valid[u]) {
                g0 = dF[u].x + gin[u].sig * __expf(fminf(12.f, fmaxf(-12.f, t[u].F.x)));
                g0 = (t[u].sig_raw >= a.f.clip_sigma_min && t[u].sig_raw <= a.f.clip_max) ? g0 : 0.f;
            }
            if (KIND == KIND_VM) {
                // clamp backward on the colour features; row 0 leaves through g_sigma_raw
                const float lo = a.f.clip_feat_min, hi_c = a.f.clip_max;
                f4 dcf;
                dcf.x = (t[u].raw.x >= lo && t[u].raw.x <= hi_c) ? dF[u].x : 0.f;
                dcf.y = (t[u].raw.y >= lo && t[u].raw.y <= hi_c) ? dF[u].y : 0.f;
                dcf.z = (t[u].raw.z >= lo && t[u].raw.z <= hi_c) ? dF[u].z : 0.f;
                dcf.w = (t[u].raw.w >= lo && t[u].raw.w <= hi_c) ? dF[u].w : 0.f;
                if (hi == 0) {
                    if (valid[u]) a.g_sigma_raw[b[u]] = g0;
                    dcf.x = 0.f;
                }
                Da[u] = to_h4(dcf);
            } else {
                // hash: only h0 is clamped (network.py:418-420); the tile is sigma_net.1's output
                dh[u] = dF[u];
                if (hi == 0) dh[u].x = g0;
                Da[u] = to_h4(dh[u]);
            }
        }
        if (KIND == KIND_VM) {
            // d loss / d products = Wb'^T . Da, written as the f16 [M][144] the VM backward reads
#pragma unroll
            for (int tk = 0; tk < 9; tk++) {
                const h4 A = Wa1T.afrag(tk, 0, lane);
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    const h4 g = to_h4(mfma(A, Da[u], zero));
                    if (valid[u]) *reinterpret_cast<h4 *>(a.g_x0 + b[u] * 144 + 16 * tk + 4 * hi) = g;
                }
            }
        }
        h4 Dh[NT][4];  // hash: gradient of sigma_net's hidden layer (pre-relu)
        if (KIND == KIND_HASH) {
#pragma unroll
            for (int n = 0; n < 4; n++) {
                const h4 A = Wa2T.afrag(n, 0, lane);
#pragma unroll
                for (int u = 0; u < NT; u++) Dh[u][n] = mask_relu(mfma(A, Da[u], zero), t[u].Ha[n]);
            }
            // d loss / d encoder features = W_s0^T . Dh, stored level-major [14][M][2] (what pvd_grid_encode_backward reads)
#pragma unroll
            for (int tk = 0; tk < 2; tk++) {
                f4 acc[NT];
#pragma unroll
                for (int u = 0; u < NT; u++) acc[u] = zero;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const h4 A = Wa1T.afrag(tk, s, lane);
#pragma unroll
                    for (int u = 0; u < NT; u++) acc[u] = mfma(A, Dh[u][s], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < NT; u++) {
                    const h4 g = to_h4(acc[u]);
                    const int k0 = 16 * tk + 4 * hi;
                    if (valid[u] && k0 < 28) {
                        uint32_t w2[2];
                        __builtin_memcpy(w2, &g, 8);
                        const uint32_t lv = k0 >> 1;
                        *reinterpret_cast<uint32_t *>(a.g_x0 + ((size_t)lv * a.f.M + b[u]) * 2) = w2[0];
                        *reinterpret_cast<uint32_t *>(a.g_x0 + ((size_t)(lv + 1) * a.f.M + b[u]) * 2) = w2[1];
                    }
                }
            }
        }
        PVD_STAMP();
        // ---- weight gradients: dW[n][k] += sum_samples dY[n][s] X[k][s]  (both operands transposed tiles; the NT tiles' samples
        // side by side along the contracted index)
        auto tr = [&](const h4 (&v)[NT]) __attribute__((always_inline)) {
            TrFrag r;
#pragma unroll
            for (int u = 0; u < NT; u++) r.v[u] = transpose_tile(v[u], scratch + (u & 1) * 256, lane);
            return r;
        };
        auto pick = [&](auto member) __attribute__((always_inline)) {  // the same member of every tile, as an array
            TrFragIn q;
#pragma unroll
            for (int u = 0; u < NT; u++) q.v[u] = member(u);
            return q;
        };
#define PVD_TR(expr) tr(pick([&](int u) __attribute__((always_inline)) { return (expr); }).v)
        {
            const TrFrag TD3 = PVD_TR(D3[u]);
#pragma unroll
            for (int tk = 0; tk < 4; tk++) dW3[tk] = mfma_nt(TD3, PVD_TR(t[u].H2[tk]), dW3[tk]);
        }
        {
            TrFrag TH1[4];
#pragma unroll
            for (int tk = 0; tk < 4; tk++) TH1[tk] = PVD_TR(t[u].H1[tk]);
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const TrFrag TD = PVD_TR(D2[u][tn]);
#pragma unroll
                for (int tk = 0; tk < 4; tk++) dW2[tn * 4 + tk] = mfma_nt(TD, TH1[tk], dW2[tn * 4 + tk]);
            }
        }
        {
            const TrFrag TSH = PVD_TR(t[u].sh), TF = PVD_TR(t[u].Fh);
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const TrFrag TD = PVD_TR(D1[u][tn]);
                dW1[tn * 2] = mfma_nt(TD, TSH, dW1[tn * 2]);
                dW1[tn * 2 + 1] = mfma_nt(TD, TF, dW1[tn * 2 + 1]);
            }
        }
        if (KIND == KIND_VM) {
            const TrFrag TD = PVD_TR(Da[u]);
#pragma unroll
            for (int tk = 0; tk < 9; tk++) dWa[tk] = mfma_nt(TD, PVD_TR(in[u].x[tk < (KIND == KIND_VM ? 9 : 2) ? tk : 0]), dWa[tk]);
        } else {
            // sigma_net.0 [64][32]: tiles tn*2+tk;  sigma_net.1 [16][64]: tiles 8+tk
            const TrFrag TX0 = PVD_TR(t[u].X[0]), TX1 = PVD_TR(t[u].X[1]);
#pragma unroll
            for (int tn = 0; tn < 4; tn++) {
                const TrFrag TD = PVD_TR(Dh[u][tn]);
                dWa[tn * 2] = mfma_nt(TD, TX0, dWa[tn * 2]);
                dWa[tn * 2 + 1] = mfma_nt(TD, TX1, dWa[tn * 2 + 1]);
            }
            const TrFrag TDa = PVD_TR(Da[u]);
#pragma unroll
            for (int tk = 0; tk < 4; tk++) dWa[8 + tk] = mfma_nt(TDa, PVD_TR(t[u].Ha[tk]), dWa[8 + tk]);
        }
#undef PVD_TR
    }

    PVD_STAMP();
    // ---- one partial per WORKGROUP: the four waves' accumulator tiles are summed in LDS in two rounds over TWO regions (waves 0 / 1
    // store, waves 2 / 3 add: two waves busy per round instead of one wave in each of four turns; the weights are dead by now; plain
    // 16-byte read-modify-writes -- LDS float atomics measured 118k cycles here), then the block writes region 0 + region 1 as
    // [tile][reg j][lane], coalesced.
    __syncthreads();
    float *__restrict__ red = reinterpret_cast<float *>(lds);
    const uint32_t wave_in_block = threadIdx.x >> 6;
    static_assert(kHeadBlock == 256, "the epilogue pairs waves (0, 2) and (1, 3)");
    for (uint32_t round = 0; round < 2; round++) {
        if ((wave_in_block >> 1) == round) {
            float *__restrict__ reg = red + (wave_in_block & 1u) * LY::floats;
            auto put = [&](int tile, f4 v) {
                float *q = reg + tile * 256 + lane;
                if (round != 0) { v.x += q[0]; v.y += q[64]; v.z += q[128]; v.w += q[192]; }
                q[0] = v.x; q[64] = v.y; q[128] = v.z; q[192] = v.w;
            };
#pragma unroll
            for (int i = 0; i < LY::a; i++) put(i, dWa[i]);
#pragma unroll
            for (int i = 0; i < 8; i++) put(LY::c1 + i, dW1[i]);
#pragma unroll
            for (int i = 0; i < 16; i++) put(LY::c2 + i, dW2[i]);
#pragma unroll
            for (int i = 0; i < 4; i++) put(LY::c3 + i, dW3[i]);
        }
        __syncthreads();
    }
    float *__restrict__ out = a.partials + (size_t)blockIdx.x * LY::floats;
    for (uint32_t i = threadIdx.x; i < (uint32_t)LY::floats; i += kHeadBlock) out[i] = red[i] + red[LY::floats + i];
    PVD_STAMP();
#undef PVD_STAMP
}

// Sum the per-wave partials and ACCUMULATE into the fp32 gradient buffers (real, un-padded layouts).
// blockIdx.x walks the real weight elements (consecutive threads read consecutive lanes of a tile:
// coalesced), blockIdx.y a slice of the waves; every slice adds its sum with one atomic per element.
template <int KIND>
__global__ void __launch_bounds__(256) k_head_reduce_dw(const float *__restrict__ partials, uint32_t nwaves, float *__restrict__ gWa1,
                                                       float *__restrict__ gWa2, float *__restrict__ gW1, float *__restrict__ gW2,
                                                       float *__restrict__ gW3) {
    using LY = DwLayout<KIND>;
    if (KIND == KIND_VM) {  // (the same code the table scatter's launch can run in extra workgroups: head_dw_reduce.h)
        static_assert(KIND != KIND_VM || LY::floats == (int)kVmHeadTileFloats, "head_dw_reduce.h mirrors DwLayout<KIND_VM>");
        head_vm_reduce_dw(partials, nwaves, gWa1, gW1, gW2, gW3, blockIdx.x, blockIdx.y);
        return;
    }
    const uint32_t nA1 = KIND == KIND_VM ? 15 * 144 : 64 * 28, nA2 = KIND == KIND_VM ? 0 : 16 * 64;
    const uint32_t n1 = 64 * 31, n2 = 64 * 64, n3 = 3 * 64;
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float *dst;
    uint32_t tile, n16, k;  // accumulator tile, row inside the 16-row tile, padded column
    if (i < nA1) {
        if (KIND == KIND_VM) {
            const uint32_t r = i / 144, c = i - r * 144;
            dst = gWa1 + i; n16 = r + 1; k = c; tile = k >> 4;
        } else {
            const uint32_t r = i / 28, c = i - r * 28;
            dst = gWa1 + i; n16 = r & 15; k = c; tile = (r >> 4) * 2 + (k >> 4);
        }
    } else if ((i -= nA1) < nA2) {
        const uint32_t r = i >> 6, c = i & 63;
        dst = gWa2 + i; n16 = r; k = c; tile = 8 + (k >> 4);
    } else if ((i -= nA2) < n1) {
        const uint32_t r = i / 31, c = i - r * 31;
        dst = gW1 + i; n16 = r & 15; k = c < 16 ? c : c + 1; tile = LY::c1 + (r >> 4) * 2 + (k >> 4);
    } else if ((i -= n1) < n2) {
        const uint32_t r = i >> 6, c = i & 63;
        dst = gW2 + i; n16 = r & 15; k = c; tile = LY::c2 + (r >> 4) * 4 + (k >> 4);
    } else if ((i -= n2) < n3) {
        const uint32_t r = i >> 6, c = i & 63;
        dst = gW3 + i; n16 = r; k = c; tile = LY::c3 + (k >> 4);
    } else {
        return;
    }
    const uint32_t off = (tile * 4 + (n16 & 3)) * 64 + (k & 15) + 16 * (n16 >> 2);
    const uint32_t per = div_up(nwaves, kReduceSlices);
    const uint32_t w0 = blockIdx.y * per, w1 = min(nwaves, w0 + per);
    float acc = 0.f;
#pragma unroll 4
    for (uint32_t w = w0; w < w1; w++) acc += partials[(size_t)w * LY::floats + off];
    if (w1 > w0) __hip_atomic_fetch_add(dst, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KIND>
static int launch_head_bwd(const HeadBwdArgs &a, uint32_t nwaves, float *gWa1, float *gWa2, float *gW1, float *gW2, float *gW3, hipStream_t s,
                           pvd_head_dw_rider *defer = nullptr) {
    size_t lds_halfs = HeadLds<KIND>::halfs + HeadLdsT<KIND>::halfs + (kHeadBlock / 64) * 512;
    if (lds_halfs < 4 * (size_t)DwLayout<KIND>::floats) lds_halfs = 4 * (size_t)DwLayout<KIND>::floats;  // the epilogue's two fp32 regions
    const uint32_t nblocks = nwaves / (kHeadBlock / 64);
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in (one workgroup per CU either way)
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_head_bwd<KIND, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(lds_halfs * sizeof(half_t))) != hipSuccess)
            return PVD_ERR_LAUNCH;
        attr_set = true;
    }
    // one wave per SIMD, all accumulators in registers (> 256 VGPRs); two waves with ~40-80 spilled registers measured slower for both
    // heads (teacher step 0.59 vs 0.73 ms) and were removed
    hipLaunchKernelGGL((k_head_bwd<KIND, 1>), dim3(nblocks), dim3(kHeadBlock), lds_halfs * sizeof(half_t), s, a);
    if (defer) {  // the reduction rides on the caller's next launch (pvd_vm_backward_rider)
        defer->partials = a.partials; defer->nblocks = nblocks; defer->gWa1 = gWa1; defer->gWc1 = gW1; defer->gWc2 = gW2; defer->gWc3 = gW3;
        defer->found_inf = nullptr;  // (the caller's to fill in: pvd_head_dw_rider)
        return check_launch();
    }
    const uint32_t nreal = (KIND == KIND_VM ? 15 * 144 : 64 * 28 + 16 * 64) + 64 * 31 + 64 * 64 + 3 * 64;
    hipLaunchKernelGGL((k_head_reduce_dw<KIND>), dim3(div_up(nreal, 256u), kReduceSlices), dim3(256), 0, s, a.partials, nblocks, gWa1, gWa2, gW1,
                       gW2, gW3);
    return check_launch();
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_head_image_halfs(int kind) {
    if (kind == KIND_VM) return HeadLds<KIND_VM>::halfs + HeadLdsT<KIND_VM>::halfs;
    if (kind == KIND_HASH) return HeadLds<KIND_HASH>::halfs + HeadLdsT<KIND_HASH>::halfs;
    return PVD_ERR_UNSUPPORTED;
}

int pvd_head_pack_weights(int kind, const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, void *image,
                          pvd_stream_t stream) {
    if (!Wa1 || !Wc1 || !Wc2 || !Wc3 || !image || (kind == KIND_HASH && !Wa2)) return PVD_ERR_INVALID;
    HeadArgs a = {};
    a.Wa1 = Wa1; a.Wa2 = Wa2; a.Wc1 = Wc1; a.Wc2 = Wc2; a.Wc3 = Wc3;
    if (kind == KIND_VM)
        hipLaunchKernelGGL((k_head_pack<KIND_VM>), dim3(div_up((uint32_t)(HeadLds<KIND_VM>::halfs + HeadLdsT<KIND_VM>::halfs), 256u)), dim3(256), 0,
                           (hipStream_t)stream, a, (half_t *)image);
    else if (kind == KIND_HASH)
        hipLaunchKernelGGL((k_head_pack<KIND_HASH>), dim3(div_up((uint32_t)(HeadLds<KIND_HASH>::halfs + HeadLdsT<KIND_HASH>::halfs), 256u)), dim3(256), 0,
                           (hipStream_t)stream, a, (half_t *)image);
    else return PVD_ERR_UNSUPPORTED;
    return check_launch();
}

int pvd_head_forward(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M, const float *Wa1,
                     const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min,
                     float clip_feat_min, float clip_max, float *sigma, float *rgb, float *feat16, const int32_t *rows_dev,
                     pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!x0 || !dirs || !Wa1 || !Wc1 || !Wc2 || !Wc3 || !sigma || !rgb || !feat16) return PVD_ERR_INVALID;
    HeadArgs a;
    a.x0 = (const half_t *)x0; a.sigma_raw = sigma_raw; a.dirs = dirs; a.M = M;
    a.Wa1 = Wa1; a.Wa2 = Wa2; a.Wc1 = Wc1; a.Wc2 = Wc2; a.Wc3 = Wc3;
    a.clip_sigma_min = clip_sigma_min; a.clip_feat_min = clip_feat_min; a.clip_max = clip_max;
    a.sigma = sigma; a.rgb = rgb; a.feat16 = feat16; a.image = (const half_t *)image;
    a.rows_dev = rows_dev;
    if (kind == KIND_HASH) {
        if (!Wa2) return PVD_ERR_INVALID;
        return launch_head_fwd<KIND_HASH>(a, (hipStream_t)stream);
    }
    if (kind == KIND_VM) {
        if (!sigma_raw) return PVD_ERR_INVALID;
        return launch_head_fwd<KIND_VM>(a, (hipStream_t)stream);
    }
    return PVD_ERR_UNSUPPORTED;
}

int pvd_hash_head_forward_fused(const float *xyz, float in_add, float in_div, const void *embeddings_f16, const int32_t *offsets, float S,
                                uint32_t H, uint32_t gridtype, int align_corners, const float *dirs, uint32_t M, const float *Wa1,
                                const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, const void *image,
                                float clip_sigma_min, float clip_max, float *sigma, float *rgb, float *feat16, const int32_t *rows_dev,
                                pvd_stream_t stream) {
    return pvd_hash_head_forward_fused_span(xyz, in_add, in_div, embeddings_f16, offsets, S, H, gridtype, align_corners, dirs, M, Wa1, Wa2, Wc1,
                                            Wc2, Wc3, image, clip_sigma_min, clip_max, sigma, rgb, feat16, rows_dev, nullptr, stream);
}

int pvd_hash_head_forward_fused_span(const float *xyz, float in_add, float in_div, const void *embeddings_f16, const int32_t *offsets, float S,
                                     uint32_t H, uint32_t gridtype, int align_corners, const float *dirs, uint32_t M, const float *Wa1,
                                     const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, const void *image,
                                     float clip_sigma_min, float clip_max, float *sigma, float *rgb, float *feat16, const int32_t *rows_dev,
                                     uint64_t *span, pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!xyz || !embeddings_f16 || !offsets || !dirs || !Wa1 || !Wa2 || !Wc1 || !Wc2 || !Wc3 || !sigma || !rgb || !feat16) return PVD_ERR_INVALID;
    if (!(in_div != 0.f)) return PVD_ERR_INVALID;
    HeadArgs a;
    a.x0 = nullptr; a.sigma_raw = nullptr; a.dirs = dirs; a.M = M;
    a.Wa1 = Wa1; a.Wa2 = Wa2; a.Wc1 = Wc1; a.Wc2 = Wc2; a.Wc3 = Wc3;
    a.clip_sigma_min = clip_sigma_min; a.clip_feat_min = clip_sigma_min; a.clip_max = clip_max;
    a.sigma = sigma; a.rgb = rgb; a.feat16 = feat16; a.image = (const half_t *)image;
    a.rows_dev = rows_dev;
    FusedLookup g;
    g.xyz = xyz; g.aff = {true, in_add, in_div}; g.grid = (const uint32_t *)embeddings_f16; g.offsets = offsets;
    g.scales = make_scales(14, S, H); g.gridtype = gridtype; g.align_corners = align_corners != 0;
    g.span = reinterpret_cast<unsigned long long *>(span);
    return launch_hash_fwd_fused(a, g, (hipStream_t)stream);
}

int pvd_infer_image_hash(const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N, const uint8_t *bitfield,
                         float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, float sigma_scale, float in_add, float in_div,
                         const void *embeddings_f16, const int32_t *offsets, float S, uint32_t H0, uint32_t gridtype, int align_corners,
                         const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, const void *image,
                         float clip_sigma_min, float clip_max, int32_t *workspace, float *weights_sum, float *depth, float *image_out,
                         pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    if (!rays_o || !rays_d || !nears || !fars || !bitfield || !embeddings_f16 || !offsets || !Wa1 || !Wa2 || !Wc1 || !Wc2 || !Wc3 || !workspace ||
        !weights_sum || !depth || !image_out)
        return PVD_ERR_INVALID;
    if (!(in_div != 0.f) || max_steps == 0 || C == 0 || H == 0) return PVD_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HeadArgs a;
    a.x0 = nullptr; a.sigma_raw = nullptr; a.dirs = nullptr; a.M = 0;
    a.Wa1 = Wa1; a.Wa2 = Wa2; a.Wc1 = Wc1; a.Wc2 = Wc2; a.Wc3 = Wc3;
    a.clip_sigma_min = clip_sigma_min; a.clip_feat_min = clip_sigma_min; a.clip_max = clip_max;
    a.sigma = nullptr; a.rgb = nullptr; a.feat16 = nullptr; a.image = (const half_t *)image; a.rows_dev = nullptr;
    FusedLookup g;
    g.xyz = nullptr; g.aff = {true, in_add, in_div}; g.grid = (const uint32_t *)embeddings_f16; g.offsets = offsets;
    g.scales = make_scales(14, S, H0); g.gridtype = gridtype; g.align_corners = align_corners != 0;
    FusedRes gr;
    for (uint32_t l = 0; l < 14; l++) gr.res[l] = (uint32_t)ceil((double)g.scales.scale[l]) + 1u;
    InferImageArgs q;
    const int prc = infer_prepare(q, rays_o, rays_d, nears, fars, N, bitfield, bound, dt_gamma, max_steps, C, H, sigma_scale, workspace, weights_sum,
                                  depth, image_out, s);
    if (prc != PVD_OK) return prc;
    const size_t lds_bytes = infer_persistent_lds_bytes();
    // 64 ray slots per workgroup, 256 sample rows per local round: three workgroups of 50 KB of LDS fit a CU (128 / 256 slots and
    // 128-row tiles measured slower, profiles/r04_render.txt, and were removed)
    uint32_t blocks = div_up(N, 64u);
    if (blocks > 768u) blocks = 768u;  // persistent
    hipLaunchKernelGGL((k_infer_hash_persistent<64, 256>), dim3(blocks), dim3(kHeadBlock), lds_bytes, s, a, g, gr, q);
    return check_launch();
}

int pvd_infer_image_vm(const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N, const uint8_t *bitfield,
                       float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, float sigma_scale, const float *aabb_host,
                       const void *const *tables_host, const uint32_t *res_host, const uint32_t *texel_stride_host, const float *Wb,
                       const float *Wc1, const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min, float clip_feat_min,
                       float clip_max, int32_t *workspace, float *weights_sum, float *depth, float *image_out, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    if (!rays_o || !rays_d || !nears || !fars || !bitfield || !aabb_host || !tables_host || !res_host || !Wb || !Wc1 || !Wc2 || !Wc3 ||
        !workspace || !weights_sum || !depth || !image_out)
        return PVD_ERR_INVALID;
    if (max_steps == 0 || C == 0 || H == 0) return PVD_ERR_INVALID;
    VmTables tb;
    const int rc = fill_tables(tb, tables_host, res_host, aabb_host, texel_stride_host);
    if (rc != PVD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    HeadArgs a;
    a.x0 = nullptr; a.sigma_raw = nullptr; a.dirs = nullptr; a.M = 0;
    a.Wa1 = Wb; a.Wa2 = nullptr; a.Wc1 = Wc1; a.Wc2 = Wc2; a.Wc3 = Wc3;
    a.clip_sigma_min = clip_sigma_min; a.clip_feat_min = clip_feat_min; a.clip_max = clip_max;
    a.sigma = nullptr; a.rgb = nullptr; a.feat16 = nullptr; a.image = (const half_t *)image; a.rows_dev = nullptr;
    InferImageArgs q;
    const int prc = infer_prepare(q, rays_o, rays_d, nears, fars, N, bitfield, bound, dt_gamma, max_steps, C, H, sigma_scale, workspace, weights_sum,
                                  depth, image_out, s);
    if (prc != PVD_OK) return prc;
    // 128 sample rows per local round, shaded through a 64-row feature tile in two passes: 47 KB of LDS and (by launch bounds) 168
    // VGPRs = THREE workgroups per CU (a 128-row feature tile -- 67 KB, ~190 VGPRs, two per CU -- and 64-row rounds were measured and
    // removed: profiles/r05_render_vm.txt)
    uint32_t blocks = div_up(N, 64u);
    if (blocks > 768u) blocks = 768u;  // persistent
    hipLaunchKernelGGL((k_infer_vm_persistent<64, 128, 64, 3>), dim3(blocks), dim3(kHeadBlock), (infer_vm_persistent_lds_bytes<128, 64>()), s, a, tb, q);
    return check_launch();
}

int pvd_mlp_head_forward_fused(const void *pts_f16, uint32_t M, const void *wstream_f16, uint32_t n_before, uint32_t n_after, const float *dirs,
                               const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3, const void *image,
                               float clip_sigma_min, float clip_max, float *sigma, float *rgb, float *feat16, pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!pts_f16 || !wstream_f16 || !dirs || !Wa1 || !Wa2 || !Wc1 || !Wc2 || !Wc3 || !sigma || !rgb || !feat16) return PVD_ERR_INVALID;
    if (n_before > 16 || n_after > 16) return PVD_ERR_UNSUPPORTED;
    HeadArgs a;
    a.x0 = nullptr; a.sigma_raw = nullptr; a.dirs = dirs; a.M = M;
    a.Wa1 = Wa1; a.Wa2 = Wa2; a.Wc1 = Wc1; a.Wc2 = Wc2; a.Wc3 = Wc3;
    a.clip_sigma_min = clip_sigma_min; a.clip_feat_min = clip_sigma_min; a.clip_max = clip_max;
    a.sigma = sigma; a.rgb = rgb; a.feat16 = feat16; a.image = (const half_t *)image;
    a.rows_dev = nullptr;
    MlpArgs m;
    m.pts = (const half_t *)pts_f16; m.wstream = (const half_t *)wstream_f16; m.n_before = n_before; m.n_after = n_after;
    return launch_mlp_fwd_fused(a, m, (hipStream_t)stream);
}

static uint32_t head_bwd_waves(int kind, uint32_t M) {
    const uint32_t ntrips = div_up(div_up(M, 16u), (uint32_t)PVD_HEAD_NT);  // a wave takes PVD_HEAD_NT tiles per trip
    uint32_t blocks = div_up(ntrips, kHeadBlock / 64);
    (void)kind;
    if (blocks > 256u) blocks = 256u;  // one workgroup per CU (persistent)
    if (blocks < 1) blocks = 1;
    return blocks * (kHeadBlock / 64);
}

int pvd_head_backward_workspace_floats(int kind, uint32_t M) {
    return (int)(head_bwd_waves(kind, M) / (kHeadBlock / 64) * (kind == KIND_VM ? DwLayout<KIND_VM>::floats : DwLayout<KIND_HASH>::floats));
}

static int head_backward_impl(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M, const float *Wa1, const float *Wa2,
                              const float *Wc1, const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min, float clip_feat_min,
                              float clip_max, const float *g_sigma, const float *g_rgb, const float *g_rgb2, const float *g_feat16,
                              float *g_sigma_raw, void *g_x0, float *gWa1, float *gWa2, float *gWc1, float *gWc2, float *gWc3, float *workspace,
                              pvd_head_dw_rider *defer, pvd_stream_t stream) {
    if (defer) { defer->partials = nullptr; defer->nblocks = 0; }
    if (M == 0) return PVD_OK;
    if (!x0 || !dirs || !Wa1 || !Wc1 || !Wc2 || !Wc3 || !g_sigma || !g_rgb || !g_x0 || !gWa1 || !gWc1 || !gWc2 || !gWc3 ||
        !workspace)
        return PVD_ERR_INVALID;
    HeadBwdArgs a;
    a.f.x0 = (const half_t *)x0; a.f.sigma_raw = sigma_raw; a.f.dirs = dirs; a.f.M = M;
    a.f.Wa1 = Wa1; a.f.Wa2 = Wa2; a.f.Wc1 = Wc1; a.f.Wc2 = Wc2; a.f.Wc3 = Wc3;
    a.f.clip_sigma_min = clip_sigma_min; a.f.clip_feat_min = clip_feat_min; a.f.clip_max = clip_max;
    a.f.sigma = nullptr; a.f.rgb = nullptr; a.f.feat16 = nullptr; a.f.image = (const half_t *)image;
    a.f.rows_dev = nullptr;
    a.g_sigma = g_sigma; a.g_rgb = g_rgb; a.g_rgb2 = g_rgb2; a.g_feat16 = g_feat16; a.g_sigma_raw = g_sigma_raw; a.g_x0 = (half_t *)g_x0;
    a.partials = workspace;
    const uint32_t nwaves = head_bwd_waves(kind, M);
    if (kind == KIND_VM) {
        if (!sigma_raw || !g_sigma_raw) return PVD_ERR_INVALID;
        return launch_head_bwd<KIND_VM>(a, nwaves, gWa1, nullptr, gWc1, gWc2, gWc3, (hipStream_t)stream, defer);
    }
    if (defer) return PVD_ERR_UNSUPPORTED;  // only the VM head's reduction has a launch to ride on
    if (kind == KIND_HASH) {
        if (!Wa2 || !gWa2) return PVD_ERR_INVALID;
        return launch_head_bwd<KIND_HASH>(a, nwaves, gWa1, gWa2, gWc1, gWc2, gWc3, (hipStream_t)stream);
    }
    return PVD_ERR_UNSUPPORTED;
}

int pvd_head_backward(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M, const float *Wa1, const float *Wa2,
                      const float *Wc1, const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min, float clip_feat_min,
                      float clip_max,
                      const float *g_sigma, const float *g_rgb, const float *g_rgb2, const float *g_feat16, float *g_sigma_raw, void *g_x0,
                      float *gWa1, float *gWa2, float *gWc1, float *gWc2, float *gWc3, float *workspace, pvd_stream_t stream) {
    return head_backward_impl(kind, x0, sigma_raw, dirs, M, Wa1, Wa2, Wc1, Wc2, Wc3, image, clip_sigma_min, clip_feat_min, clip_max, g_sigma, g_rgb,
                              g_rgb2, g_feat16, g_sigma_raw, g_x0, gWa1, gWa2, gWc1, gWc2, gWc3, workspace, nullptr, stream);
}

int pvd_head_backward_defer(int kind, const void *x0, const float *sigma_raw, const float *dirs, uint32_t M, const float *Wa1, const float *Wa2,
                            const float *Wc1, const float *Wc2, const float *Wc3, const void *image, float clip_sigma_min, float clip_feat_min,
                            float clip_max, const float *g_sigma, const float *g_rgb, const float *g_rgb2, const float *g_feat16,
                            float *g_sigma_raw, void *g_x0, float *gWa1, float *gWa2, float *gWc1, float *gWc2, float *gWc3, float *workspace,
                            pvd_head_dw_rider *rider_out, pvd_stream_t stream) {
    if (!rider_out) return PVD_ERR_INVALID;
    return head_backward_impl(kind, x0, sigma_raw, dirs, M, Wa1, Wa2, Wc1, Wc2, Wc3, image, clip_sigma_min, clip_feat_min, clip_max, g_sigma, g_rgb,
                              g_rgb2, g_feat16, g_sigma_raw, g_x0, gWa1, gWa2, gWc1, gWc2, gWc3, workspace, rider_out, stream);
}

}  // extern "C"
