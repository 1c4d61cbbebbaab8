// grid_lookup.h -- the index / blend arithmetic of the multiresolution hash grid (reference: gridencoder.cu:35-224), shared
// by gridencoder.hip (the _gridencoder entry points) and fusedhead.hip (lookup fused into the frozen teacher's head).
#pragma once

#include "pvd_device.h"

#include <math.h>

namespace pvd {

constexpr uint32_t kMaxLevels = 32;

struct LevelScales {
    float scale[kMaxLevels];
};

// C consecutive table elements moved as one naturally aligned access
template <typename T, uint32_t C>
struct alignas(sizeof(T) * C) FeatVec {
    T v[C];
};

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));

struct alignas(4) Pos3 {  // one sample position as a single 12-byte load
    float x, y, z;
};

// level-uniform indexing state (reference: get_grid_index, gridencoder.cu:54-72)
template <uint32_t D>
struct LevelIndex {
    uint32_t size;       // hashmap_size = offsets[l+1] - offsets[l]
    uint32_t stride[D];  // 0 for dimensions the dense loop never reaches
    bool hashed;
    bool pow2;

    __device__ __forceinline__ void init(uint32_t size_, uint32_t resolution, uint32_t gridtype, bool align_corners) {
        size = size_;
        uint32_t s = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if (s <= size) {
                stride[d] = s;
                s *= align_corners ? resolution : (resolution + 1);
            } else {
                stride[d] = 0;
            }
        }
        hashed = (gridtype == 0) && (s > size);
        pow2 = (size & (size - 1)) == 0;
    }

    __device__ __forceinline__ uint32_t operator()(const uint32_t (&pg)[D]) const {
        uint32_t index;
        if (hashed) {
            constexpr uint32_t primes[3] = {1u, 2654435761u, 805459861u};
            index = 0;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) index ^= pg[d] * primes[d];
            index = pow2 ? (index & (size - 1)) : (index % size);
        } else {
            index = 0;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) index += pg[d] * stride[d];
            if (index >= size) index %= size;  // only when the table was sized smaller than the kernel's resolution
        }
        return index;
    }
};

// ---- lean per-level geometry for the D = 3 kernels that keep SEVERAL levels' gathers in flight (fusedhead.hip).
// LevelIndex::operator() decides hashed / pow2 / wrap per CORNER (three uniform branches and a possible division in front of
// every load); here the decision is taken once per level and the two shapes the PVD table consists of get straight-line code:
//   kHashPow2   : hashed level, power-of-two size   -> ((x) ^ (y P1) ^ (z P2)) & (size - 1), y P1 and z P2 shared by the corners
//   kDensePlain : dense level that cannot wrap      -> x + y s1 + z s2
//   kGeneric    : everything else (tiled grids, non-power-of-two hash sizes, align_corners) through LevelIndex
// Same integers as LevelIndex for every corner, hence the same rows and bit-identical outputs.
enum : uint32_t { kLevelGeneric = 0, kLevelHashPow2 = 1, kLevelDensePlain = 2 };

struct Level3 {
    uint32_t size, s1, s2, mode, resolution;

    __device__ __forceinline__ void init(uint32_t size_, uint32_t resolution_, uint32_t gridtype, bool align_corners) {
        size = size_;
        resolution = resolution_;
        const uint32_t R = align_corners ? resolution_ : resolution_ + 1u;
        uint32_t s = 1;
        s1 = s2 = 0;
        if (s <= size) s *= R;                 // x: stride 1
        if (s <= size) { s1 = s; s *= R; }     // y
        if (s <= size) { s2 = s; s *= R; }     // z
        const bool hashed = gridtype == 0u && s > size;
        const bool pow2 = (size & (size - 1u)) == 0u;
        // dense and unable to wrap: all three strides real, R^3 <= size, and (no align_corners) every corner coordinate <= R - 1
        const bool dense_plain = !hashed && s2 != 0u && s <= size && !align_corners;
        mode = (hashed && pow2) ? kLevelHashPow2 : dense_plain ? kLevelDensePlain : kLevelGeneric;
    }
};

// the general index of one corner, out of line (never taken by the PVD table; keeps the unrolled level loops small: inlined,
// the fused kernel needs 173-183 VGPRs instead of 155, i.e. two waves per SIMD instead of three)
__device__ __noinline__ uint32_t level3_generic_index(uint32_t size, uint32_t resolution, uint32_t gridtype, bool align_corners, uint32_t x,
                                                      uint32_t y, uint32_t z) {
    LevelIndex<3> index;
    index.init(size, resolution, gridtype, align_corners);
    const uint32_t pg[3] = {x, y, z};
    return index(pg);
}

// rows of the four (y, z) corners k = yb + 2 zb of lane-corner x = cell[0] + xb
__device__ __forceinline__ void level3_rows(const Level3 &lv, uint32_t gridtype, bool align_corners, const uint32_t (&cell)[3], uint32_t xb,
                                            uint32_t (&row)[4]) {
    const uint32_t x = cell[0] + xb;
    if (lv.mode == kLevelHashPow2) {
        const uint32_t mask = lv.size - 1u;
        const uint32_t a0 = cell[1] * 2654435761u, a1 = a0 + 2654435761u;  // (y + 1) P1 = y P1 + P1 (mod 2^32)
        const uint32_t b0 = cell[2] * 805459861u, b1 = b0 + 805459861u;
        const uint32_t xa0 = x ^ a0, xa1 = x ^ a1;
        row[0] = (xa0 ^ b0) & mask; row[1] = (xa1 ^ b0) & mask; row[2] = (xa0 ^ b1) & mask; row[3] = (xa1 ^ b1) & mask;
    } else if (lv.mode == kLevelDensePlain) {
        const uint32_t base = x + cell[1] * lv.s1 + cell[2] * lv.s2;
        row[0] = base; row[1] = base + lv.s1; row[2] = base + lv.s2; row[3] = base + lv.s1 + lv.s2;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            row[k] = level3_generic_index(lv.size, lv.resolution, gridtype, align_corners, x, cell[1] + (k & 1u), cell[2] + (k >> 1));
    }
}

// optional input mapping x01 = (x + add) / div applied while reading the positions: GridEncoder.forward's
// (inputs + bound) / (2 * bound) (grid.py:211) without two elementwise launches; same two IEEE operations
struct InputAffine {
    bool on;
    float add, div;
};

template <uint32_t D>
__device__ __forceinline__ bool locate(const float *__restrict__ in, float scale, bool align_corners, float (&frac)[D], uint32_t (&cell)[D],
                                       InputAffine aff = {false, 0.f, 1.f}) {
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = in[d];
        if (aff.on) x[d] = (x[d] + aff.add) / aff.div;
        oob |= (x[d] < 0.0f) | (x[d] > 1.0f);
    }
    if (oob) return false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const float p = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);  // canonical fused form
        const float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        frac[d] = p - (float)cell[d];
    }
    return true;
}

// accumulate one weighted feature vector with the reference's scalar_t arithmetic:
//   f32: acc = fma(w, v, acc);   f16: acc = half(acc + half(w * float(v)))  (gridencoder.cu:166)
template <typename T>
__device__ __forceinline__ void axpy(T &acc, float w, T v);
template <>
__device__ __forceinline__ void axpy<float>(float &acc, float w, float v) { acc = fmaf(w, v, acc); }
// (half)(float product): the product must be rounded to f32 FIRST and then to f16, as c10::Half /
// __half conversions of a float expression do (gridencoder.cu:166,303).  hipcc otherwise selects
// v_fma_mixlo_f16 (one rounding of the exact product), which differs in the last half-ulp for
// about 1 value in 10^4; the empty asm pins the f32 intermediate.
__device__ __forceinline__ half_t half_of_product(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return (half_t)p;
}
template <>
__device__ __forceinline__ void axpy<half_t>(half_t &acc, float w, half_t v) { acc = acc + half_of_product(w, (float)v); }

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_quad(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}

__device__ __forceinline__ uint32_t weighted_pair(float w, uint32_t packed) {
    half_t v[2];
    __builtin_memcpy(v, &packed, 4);
    half_t r[2] = {half_of_product(w, (float)v[0]), half_of_product(w, (float)v[1])};
    uint32_t out;
    __builtin_memcpy(&out, r, 4);
    return out;
}

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    half2_t x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const half2_t s = x + y;  // v_pk_add_f16
    uint32_t out;
    __builtin_memcpy(&out, &s, 4);
    return out;
}

static inline LevelScales make_scales(uint32_t L, float S, uint32_t H) {
    LevelScales s;
    for (uint32_t l = 0; l < kMaxLevels; l++) s.scale[l] = 0.f;
    // per-level scale on the HOST with libm exp2f: device exp2 differs from libm/CUDA by an ulp or
    // two, which would move knife-edge samples across cells (SURVEY.md section 7 "hard parts").
    for (uint32_t l = 0; l < L; l++) s.scale[l] = exp2f((float)l * S) * (float)H - 1.0f;  // gridencoder.cu:126
    return s;
}

}  // namespace pvd
