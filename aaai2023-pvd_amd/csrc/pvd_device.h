// pvd_device.h -- shared device/host helpers for libpvd_hip.so (gfx950 only).
//
// Canonical floating-point mode (see DESIGN.md "Arithmetic contract"): the library is
// built with -ffp-contract=off, so a product is rounded before it is added unless the
// source says fmaf().  fmaf() is used exactly where the reference's compilers contract
// the reference source AND the value feeds an integer decision or a position (o + t*d, the
// start jitter, x*mip_rbound + 1 and (..)*mip_bound - x in the marchers -- the last three
// found in round 6 against the reference's own kernels built for gfx950, oracle/_ref --,
// x*scale + 0.5 and acc += w*v in the grid encoder); the CPU oracle mirrors this, which is
// what makes HIP-vs-oracle bit-exact for the marcher and the encoder forward.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pvd_hip.h"

namespace pvd {

constexpr int kWave = 64;  // CDNA wavefront

__host__ __device__ inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// thread-local last HIP error, reported through pvd_last_hip_error()
void set_last_error(hipError_t e);

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(e);
        return PVD_ERR_LAUNCH;
    }
    return PVD_OK;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float sign1f(float x) { return copysignf(1.0f, x); }

// ---- 10-bit x 3 Morton code (reference semantics: raymarching.cu:58-83) ----
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__device__ __forceinline__ uint32_t gather3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// ---- PCG32 (XSH-RR 64/32), the generator of raymarching/src/pcg32.h:44-170 ----
struct Pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t kMult = 0x5851f42d4c957f2dULL;

    __device__ __forceinline__ uint32_t next() {
        const uint64_t old = state;
        state = old * kMult + inc;
        const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
        const uint32_t rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    __device__ __forceinline__ void seed(uint64_t initstate, uint64_t initseq = 1) {
        state = 0;
        inc = (initseq << 1) | 1u;
        (void)next();
        state += initstate;
        (void)next();
    }
    // jump ahead by delta draws in O(log delta)
    __device__ __forceinline__ void advance(uint64_t delta) {
        uint64_t cur_mult = kMult, cur_plus = inc, acc_mult = 1, acc_plus = 0;
        while (delta) {
            if (delta & 1u) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
    __device__ __forceinline__ float next_float() {
        return __uint_as_float((next() >> 9) | 0x3f800000u) - 1.0f;
    }
};

}  // namespace pvd
