// occupancy.hip -- maintenance of the occupancy grid on the device (gfx950).
//
// Replaces the torch code of NeRFRenderer.update_extra_state (distill_mutual/renderer.py:647-775): per cascade
//   * which cells to query -- every cell while iter_density < 16, afterwards H^3/4 uniformly random cells plus H^3/4
//     cells drawn (with replacement) from the currently occupied ones (:700-730) -- and where inside them (cell centre
//     +- half a cell of jitter, :733-741),
//   * after the caller's density query: scatter into a scratch grid, decayed running maximum where both old and new
//     values are valid (:747-750),
//   * mean of the clamped grid, threshold = min(mean, density_thresh), packbits (:752-760),
// as five small kernels and NO host round trip (the reference -- and the torch formulation -- synchronise for nonzero(),
// for the mean and per 64^3 block of the full sweep).  Random numbers: PCG32 keyed by (seed, cell slot); the reference uses
// torch's generator, so only the distribution is reproduced, not the stream -- except through pvd_occ_sample_replay /
// pvd_occ_update_ordered, which take the draws as input and resolve duplicate cells like a sequential assignment: a run of the
// reference can then be replayed cell for cell (tests/test_hip_occupancy.py).
#include "pvd_device.h"

namespace pvd {

constexpr uint32_t kOccBlock = 256;

// occupied cells of one cascade, compacted (order is irrelevant: they are sampled uniformly).  A workgroup owns a contiguous stretch of
// the grid: it counts its occupied cells, reserves their range with ONE atomic, then writes them through a cursor in LDS.  (One returning
// atomic per WAVE on the one counter -- the first version -- is 32 768 of them for a 128^3 grid in which nearly every cell holds a positive
// running maximum: the memory side serialises them at ~12 ns each, 374 us per update in the teacher's trace; profiles/r06_occ_compact.txt.)
constexpr uint32_t kOccCompactGroups = 1024;
__global__ void __launch_bounds__(kOccBlock) k_occ_compact(const float *__restrict__ grid, uint32_t H3, int32_t *__restrict__ list,
                                                          uint32_t *__restrict__ count) {
    __shared__ uint32_t wave_cnt[kOccBlock / 64];
    __shared__ uint32_t cursor;
    const uint32_t per = (H3 + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(H3, lo + per);
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t mine = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kOccBlock) mine += grid[i] > 0.f ? 1u : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0) wave_cnt[wid] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < kOccBlock / 64; w++) tot += wave_cnt[w];
        cursor = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    for (uint32_t base = lo + wid * 64u; base < hi; base += kOccBlock) {  // (uniform per wave)
        const uint32_t i = base + lane;
        const bool occ = i < hi && grid[i] > 0.f;
        const uint64_t m = __ballot(occ);
        uint32_t at = 0;
        if (lane == 0 && m) at = atomicAdd(&cursor, (uint32_t)__popcll(m));  // LDS
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (occ) list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
    }
}

// slot i < n_uniform: a uniformly random cell; n_uniform <= i < n_uniform + n_occ: a random occupied cell (index -1 if
// there is none); full sweep: slot i = Morton index i.  Writes the Morton index and the jittered world position.
__global__ void __launch_bounds__(kOccBlock) k_occ_positions(uint32_t H, uint32_t n_uniform, uint32_t n_occ, int full, float bound_c,
                                                            uint64_t seed, const int32_t *__restrict__ list,
                                                            const uint32_t *__restrict__ count, int32_t *__restrict__ indices,
                                                            float *__restrict__ xyz) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= n_uniform + n_occ) return;
    Pcg32 g;
    g.seed(seed);
    g.advance(8ull * i);
    uint32_t cx, cy, cz;
    int32_t idx;
    if (full) {
        idx = (int32_t)i;
        cx = gather3(i); cy = gather3(i >> 1); cz = gather3(i >> 2);
    } else if (i < n_uniform) {
        cx = (uint32_t)(((uint64_t)g.next() * H) >> 32);
        cy = (uint32_t)(((uint64_t)g.next() * H) >> 32);
        cz = (uint32_t)(((uint64_t)g.next() * H) >> 32);
        idx = (int32_t)morton3(cx, cy, cz);
    } else {
        const uint32_t n = count[0];
        if (n == 0) {
            indices[i] = -1;
            xyz[3 * (size_t)i] = 0.f; xyz[3 * (size_t)i + 1] = 0.f; xyz[3 * (size_t)i + 2] = 0.f;
            return;
        }
        idx = list[(uint32_t)(((uint64_t)g.next() * n) >> 32)];
        cx = gather3((uint32_t)idx); cy = gather3((uint32_t)idx >> 1); cz = gather3((uint32_t)idx >> 2);
    }
    // xyzs = 2 * coords / (H - 1) - 1;  p = xyzs * (bound - half_grid_size) + (rand * 2 - 1) * half_grid_size  (:733-741)
    const float hgs = bound_c / (float)H;
    const float c[3] = {(float)cx, (float)cy, (float)cz};
    indices[i] = idx;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float u = 2.0f * c[a] / (float)(H - 1) - 1.0f;
        const float r = g.next_float();
        xyz[3 * (size_t)i + a] = u * (bound_c - hgs) + (r * 2.0f - 1.0f) * hgs;
    }
}

// The same positions from SUPPLIED draws (pvd_occ_sample_replay): what the reference's torch calls return for one cascade, in its
// order -- cells = randint(0, H, (n_uniform, 3)); picks = randint(0, #occupied, n_occ) into the ASCENDING list of occupied Morton
// indices (nonzero()); jitter = rand(n, 3) in [0, 1), row k of it belonging to the k-th queried point (full sweep: the meshgrid
// point (x * H + y) * H + z, renderer.py:700-712) -- so that a run of the reference can be replayed cell for cell.
__global__ void __launch_bounds__(kOccBlock) k_occ_positions_replay(uint32_t H, uint32_t n_uniform, uint32_t n_occ, int full, float bound_c,
                                                                   const int32_t *__restrict__ cells, const int32_t *__restrict__ list,
                                                                   const int32_t *__restrict__ picks, const float *__restrict__ jitter,
                                                                   int32_t *__restrict__ indices, float *__restrict__ xyz) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= n_uniform + n_occ) return;
    uint32_t cx, cy, cz, row = i;
    int32_t idx;
    if (full) {
        idx = (int32_t)i;
        cx = gather3(i); cy = gather3(i >> 1); cz = gather3(i >> 2);
        row = (cx * H + cy) * H + cz;
    } else if (i < n_uniform) {
        cx = (uint32_t)cells[3 * (size_t)i]; cy = (uint32_t)cells[3 * (size_t)i + 1]; cz = (uint32_t)cells[3 * (size_t)i + 2];
        idx = (int32_t)morton3(cx, cy, cz);
    } else {
        idx = list[picks[i - n_uniform]];
        cx = gather3((uint32_t)idx); cy = gather3((uint32_t)idx >> 1); cz = gather3((uint32_t)idx >> 2);
    }
    const float hgs = bound_c / (float)H;
    const float c[3] = {(float)cx, (float)cy, (float)cz};
    indices[i] = idx;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float u = 2.0f * c[a] / (float)(H - 1) - 1.0f;
        const float r = jitter[3 * (size_t)row + a];
        xyz[3 * (size_t)i + a] = u * (bound_c - hgs) + (r * 2.0f - 1.0f) * hgs;
    }
}

// tmp_grid[cas, indices] = sigmas with duplicates resolved the way a sequential assignment resolves them (the reference on the
// CPU): the LAST position holding a cell wins.  owner[cell] = max position (pass 1), the owner writes (pass 2).
__global__ void __launch_bounds__(kOccBlock) k_occ_fill_i32(int32_t *__restrict__ p, uint32_t n, int32_t v) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(kOccBlock) k_occ_owner(int32_t *__restrict__ owner, const int32_t *__restrict__ indices, uint32_t n) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t idx = indices[i];
    if (idx >= 0) atomicMax(owner + idx, (int32_t)i);
}
__global__ void __launch_bounds__(kOccBlock) k_occ_scatter_owned(float *__restrict__ tmp, const int32_t *__restrict__ owner,
                                                                const int32_t *__restrict__ indices, const float *__restrict__ sigmas,
                                                                float scale, uint32_t n) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t idx = indices[i];
    if (idx >= 0 && owner[idx] == (int32_t)i) tmp[idx] = sigmas[i] * scale;
}

__global__ void __launch_bounds__(kOccBlock) k_occ_fill(float *__restrict__ tmp, uint32_t n4, float v) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i < n4) reinterpret_cast<float4 *>(tmp)[i] = make_float4(v, v, v, v);
}

// tmp_grid[cas, indices] = sigmas (:743-745; duplicate indices: whichever write lands last, as in the reference)
__global__ void __launch_bounds__(kOccBlock) k_occ_scatter(float *__restrict__ tmp, const int32_t *__restrict__ indices,
                                                          const float *__restrict__ sigmas, float scale, uint32_t n) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t idx = indices[i];
    if (idx >= 0) tmp[idx] = sigmas[i] * scale;
}

// valid = (grid >= 0) & (tmp >= 0);  grid[valid] = max(grid * decay, tmp)   (:747-750)
__global__ void __launch_bounds__(kOccBlock) k_occ_ema(float *__restrict__ grid, const float *__restrict__ tmp, uint32_t n4, float decay) {
    const uint32_t i = blockIdx.x * kOccBlock + threadIdx.x;
    if (i >= n4) return;
    float4 g = reinterpret_cast<float4 *>(grid)[i];
    const float4 t = reinterpret_cast<const float4 *>(tmp)[i];
    float *gg = &g.x;
    const float *tt = &t.x;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (gg[k] >= 0.f && tt[k] >= 0.f) gg[k] = fmaxf(gg[k] * decay, tt[k]);
    reinterpret_cast<float4 *>(grid)[i] = g;
}

// partial sums of clamp(grid, min = 0) (:752)
__global__ void __launch_bounds__(kOccBlock) k_occ_mean_partial(const float *__restrict__ grid, uint32_t n4, float *__restrict__ partials) {
    __shared__ float sh[kOccBlock / 64];
    float acc = 0.f;
    for (uint32_t i = blockIdx.x * kOccBlock + threadIdx.x; i < n4; i += gridDim.x * kOccBlock) {
        const float4 g = reinterpret_cast<const float4 *>(grid)[i];
        acc += (fmaxf(g.x, 0.f) + fmaxf(g.y, 0.f)) + (fmaxf(g.z, 0.f) + fmaxf(g.w, 0.f));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (uint32_t w = 0; w < kOccBlock / 64; w++) s += sh[w];
        partials[blockIdx.x] = s;
    }
}
__global__ void __launch_bounds__(kOccBlock) k_occ_mean_final(const float *__restrict__ partials, uint32_t nparts, float inv_n,
                                                             float density_thresh, float *__restrict__ mean_thresh) {
    __shared__ float sh[kOccBlock / 64];
    float acc = 0.f;
    for (uint32_t i = threadIdx.x; i < nparts; i += kOccBlock) acc += partials[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (uint32_t w = 0; w < kOccBlock / 64; w++) s += sh[w];
        const float mean = s * inv_n;
        mean_thresh[0] = mean;
        mean_thresh[1] = fminf(mean, density_thresh);  // density_thresh = min(mean_density, density_thresh) (:757)
    }
}

// packbits with the threshold on the device (reference: kernel_packbits, raymarching.cu:269-291)
__global__ void __launch_bounds__(kOccBlock) k_occ_packbits(const float *__restrict__ grid, uint32_t N, const float *__restrict__ thresh_dev,
                                                           uint8_t *__restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kOccBlock + threadIdx.x;
    if (n >= N) return;
    const float thresh = thresh_dev[1];
    const float4 a = reinterpret_cast<const float4 *>(grid)[2 * (size_t)n];
    const float4 b = reinterpret_cast<const float4 *>(grid)[2 * (size_t)n + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;   bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;   bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;  bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;  bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

constexpr uint32_t kOccMeanBlocks = 1024;

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_occ_sample(const float *density_grid, uint32_t H, uint32_t n_uniform, uint32_t n_occupied, int full, float bound_c, uint64_t seed,
                   int32_t *occ_list, uint32_t *occ_count, int32_t *indices, float *xyz, pvd_stream_t stream) {
    if (!density_grid || !indices || !xyz || H < 2 || H > 1024) return PVD_ERR_INVALID;
    const uint32_t H3 = H * H * H;
    if (full) { n_uniform = H3; n_occupied = 0; }
    if (n_uniform + n_occupied == 0) return PVD_OK;
    if (n_occupied && (!occ_list || !occ_count)) return PVD_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_occupied) {
        (void)hipMemsetAsync(occ_count, 0, sizeof(uint32_t), s);
        const uint32_t groups = div_up(H3, kOccBlock) < kOccCompactGroups ? div_up(H3, kOccBlock) : kOccCompactGroups;
        hipLaunchKernelGGL(k_occ_compact, dim3(groups), dim3(kOccBlock), 0, s, density_grid, H3, occ_list, occ_count);
    }
    hipLaunchKernelGGL(k_occ_positions, dim3(div_up(n_uniform + n_occupied, kOccBlock)), dim3(kOccBlock), 0, s, H, n_uniform, n_occupied, full,
                       bound_c, seed, occ_list, occ_count, indices, xyz);
    return check_launch();
}

int pvd_occ_update(float *density_grid, float *tmp, const int32_t *indices, const float *sigmas, uint32_t n, uint32_t H, float sigma_scale,
                   float decay, pvd_stream_t stream) {
    if (!density_grid || !tmp || (n && (!indices || !sigmas)) || H < 2 || H > 1024) return PVD_ERR_INVALID;
    const uint32_t H3 = H * H * H;
    if (H3 & 3u) return PVD_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_occ_fill, dim3(div_up(H3 / 4, kOccBlock)), dim3(kOccBlock), 0, s, tmp, H3 / 4, -1.0f);
    if (n) hipLaunchKernelGGL(k_occ_scatter, dim3(div_up(n, kOccBlock)), dim3(kOccBlock), 0, s, tmp, indices, sigmas, sigma_scale, n);
    hipLaunchKernelGGL(k_occ_ema, dim3(div_up(H3 / 4, kOccBlock)), dim3(kOccBlock), 0, s, density_grid, tmp, H3 / 4, decay);
    return check_launch();
}

int pvd_occ_sample_replay(uint32_t H, uint32_t n_uniform, uint32_t n_occupied, int full, float bound_c, const int32_t *cells,
                          const int32_t *occ_list, const int32_t *picks, const float *jitter, int32_t *indices, float *xyz,
                          pvd_stream_t stream) {
    if (!jitter || !indices || !xyz || H < 2 || H > 1024) return PVD_ERR_INVALID;
    if (full) { n_uniform = H * H * H; n_occupied = 0; }
    if (n_uniform + n_occupied == 0) return PVD_OK;
    if ((!full && n_uniform && !cells) || (n_occupied && (!occ_list || !picks))) return PVD_ERR_INVALID;
    hipLaunchKernelGGL(k_occ_positions_replay, dim3(div_up(n_uniform + n_occupied, kOccBlock)), dim3(kOccBlock), 0, (hipStream_t)stream, H,
                       n_uniform, n_occupied, full, bound_c, cells, occ_list, picks, jitter, indices, xyz);
    return check_launch();
}

int pvd_occ_update_ordered(float *density_grid, float *tmp, int32_t *owner, const int32_t *indices, const float *sigmas, uint32_t n,
                           uint32_t H, float sigma_scale, float decay, pvd_stream_t stream) {
    if (!density_grid || !tmp || !owner || (n && (!indices || !sigmas)) || H < 2 || H > 1024) return PVD_ERR_INVALID;
    const uint32_t H3 = H * H * H;
    if (H3 & 3u) return PVD_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_occ_fill, dim3(div_up(H3 / 4, kOccBlock)), dim3(kOccBlock), 0, s, tmp, H3 / 4, -1.0f);
    if (n) {
        hipLaunchKernelGGL(k_occ_fill_i32, dim3(div_up(H3, kOccBlock)), dim3(kOccBlock), 0, s, owner, H3, -1);
        hipLaunchKernelGGL(k_occ_owner, dim3(div_up(n, kOccBlock)), dim3(kOccBlock), 0, s, owner, indices, n);
        hipLaunchKernelGGL(k_occ_scatter_owned, dim3(div_up(n, kOccBlock)), dim3(kOccBlock), 0, s, tmp, owner, indices, sigmas, sigma_scale, n);
    }
    hipLaunchKernelGGL(k_occ_ema, dim3(div_up(H3 / 4, kOccBlock)), dim3(kOccBlock), 0, s, density_grid, tmp, H3 / 4, decay);
    return check_launch();
}

int pvd_occ_finish(const float *density_grid, uint32_t n_cells, float density_thresh, float *mean_thresh, float *scratch, uint8_t *bitfield,
                   pvd_stream_t stream) {
    if (!density_grid || !mean_thresh || !scratch || !bitfield || n_cells == 0) return PVD_ERR_INVALID;
    if (n_cells & 7u) return PVD_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_occ_mean_partial, dim3(kOccMeanBlocks), dim3(kOccBlock), 0, s, density_grid, n_cells / 4, scratch);
    hipLaunchKernelGGL(k_occ_mean_final, dim3(1), dim3(kOccBlock), 0, s, scratch, kOccMeanBlocks, 1.0f / (float)n_cells, density_thresh,
                       mean_thresh);
    hipLaunchKernelGGL(k_occ_packbits, dim3(div_up(n_cells / 8, kOccBlock)), dim3(kOccBlock), 0, s, density_grid, n_cells / 8, mean_thresh,
                       bitfield);
    return check_launch();
}

}  // extern "C"
