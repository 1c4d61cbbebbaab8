// head_dw_reduce.h -- the reduction of the VM head's weight-gradient partials (fusedhead.hip: k_head_bwd leaves one set of
// accumulator tiles per workgroup; k_head_reduce_dw sums them into the fp32 gradients), as a device function: the VM student's
// table scatter (vmencoder.hip: k_vm_bwd_split) can run it in extra workgroups of ITS launch -- both depend only on the head's
// backward -- so the step's chain has one dependent launch fewer (pvd_head_backward_defer / pvd_vm_backward_rider).
#pragma once

#include "pvd_device.h"

namespace pvd {

constexpr uint32_t kReduceSlices = 16;
// VM head: basis_mat 15 x 144, colour net 64 x 31, 64 x 64, 3 x 64; accumulator tiles per wave [9][4 x 2][4 x 4][1 x 4]
constexpr uint32_t kVmHeadReal = 15 * 144 + 64 * 31 + 64 * 64 + 3 * 64;
constexpr uint32_t kVmHeadTileFloats = (9 + 28) * 256;
constexpr uint32_t kVmHeadReduceBlocks = (kVmHeadReal + 255u) / 256u;  // x kReduceSlices slices

// one 256-thread workgroup: elements [256 bx, 256 bx + 256) of the concatenated gradients, waves of slice `slice`
__device__ __forceinline__ void head_vm_reduce_dw(const float *__restrict__ partials, uint32_t nwaves, float *__restrict__ gWa1,
                                                  float *__restrict__ gW1, float *__restrict__ gW2, float *__restrict__ gW3, uint32_t bx,
                                                  uint32_t slice, float *__restrict__ found_inf = nullptr) {
    constexpr uint32_t nA1 = 15 * 144, n1 = 64 * 31, n2 = 64 * 64, n3 = 3 * 64;
    constexpr uint32_t c1 = 9, c2 = 17, c3 = 33;
    uint32_t i = bx * 256 + threadIdx.x;
    float *dst;
    uint32_t tile, n16, k;  // accumulator tile, row inside the 16-row tile, padded column
    if (i < nA1) {
        const uint32_t r = i / 144, c = i - r * 144;
        dst = gWa1 + i; n16 = r + 1; k = c; tile = k >> 4;
    } else if ((i -= nA1) < n1) {
        const uint32_t r = i / 31, c = i - r * 31;
        dst = gW1 + i; n16 = r & 15; k = c < 16 ? c : c + 1; tile = c1 + (r >> 4) * 2 + (k >> 4);
    } else if ((i -= n1) < n2) {
        const uint32_t r = i >> 6, c = i & 63;
        dst = gW2 + i; n16 = r & 15; k = c; tile = c2 + (r >> 4) * 4 + (k >> 4);
    } else if ((i -= n2) < n3) {
        const uint32_t r = i >> 6, c = i & 63;
        dst = gW3 + i; n16 = r; k = c; tile = c3 + (k >> 4);
    } else {
        return;
    }
    const uint32_t off = (tile * 4 + (n16 & 3)) * 64 + (k & 15) + 16 * (n16 >> 2);
    const uint32_t per = div_up(nwaves, kReduceSlices);
    const uint32_t w0 = slice * per, w1 = min(nwaves, w0 + per);
    float acc = 0.f;
#pragma unroll 4
    for (uint32_t w = w0; w < w1; w++) acc += partials[(size_t)w * kVmHeadTileFloats + off];
    if (w1 > w0) __hip_atomic_fetch_add(dst, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (pvd_head_dw_rider.found_inf) a slice's sum that is inf / nan makes the weight gradient inf / nan
    if (found_inf && (__float_as_uint(acc) & 0x7f800000u) == 0x7f800000u) found_inf[0] = 1.0f;
}

}  // namespace pvd
