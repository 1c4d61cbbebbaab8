// distill.hip -- the distillation objective of stage 3 as three launches instead of ~45.
//
// Reference (torch code): Trainer.train_step / get_loss, distill_mutual/utils.py:941-952, 1044-1189 with
// loss_type = "normL2":  loss = r_rgb ||I_tea - I_stu|| + r_fea ||F_stu - F_tea|| + r_sig ||s_stu - s_tea||
//                               + r_col ||c_stu - c_tea||       (Frobenius norms over ALL rows, padding included)
// where F = feature_sigma_color [M,16], s = F[:,0] (sigma_l), c = color_l [M,3], I = composited image [N,3].
//   k_sumsq4      : S[0..3] = the four sums of squares (one pass over the six tensors)
//   (all-reduce of S[4] under ray data parallelism happens between the two kernels, on the host side)
//   k_loss_final  : loss = sum_i r_i sqrt(S_i);  coef_i = r_i / sqrt(S_i)   (0 where S_i = 0: torch.norm's subgradient)
//   k_sumsq4_bwd  : the three student gradients in one pass, scaled by the upstream gradient on the device
#include "pvd_device.h"

namespace pvd {

constexpr uint32_t kLossBlock = 256;
constexpr uint32_t kSumsqMaxBlocks = 256;  // per-block partials after the four sums in S4 (callers provide room for 1024): one per
                                           // thread of whoever finishes them (k_loss_final, or EVERY workgroup of k_loss_final_bwd)

static_assert(kSumsqMaxBlocks <= kLossBlock, "k_loss_final_bwd reads one partial per thread");

__device__ __forceinline__ float block_sum(float v, float *__restrict__ sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (uint32_t w = 0; w < kLossBlock / 64; w++) r += sh[w];
    return r;
}

__global__ void __launch_bounds__(kLossBlock) k_sumsq4(const float *__restrict__ img_s, const float *__restrict__ img_t, uint32_t n_img,
                                                      const float *__restrict__ fea_s, const float *__restrict__ fea_t, uint32_t M,
                                                      const float *__restrict__ col_s, const float *__restrict__ col_t,
                                                      float *__restrict__ S, float *__restrict__ rates_decay, float fea_decay, uint32_t fea_width) {
    __shared__ float sh[kLossBlock / 64];
    const uint32_t tid = blockIdx.x * kLossBlock + threadIdx.x, stride = gridDim.x * kLossBlock;
    // the per-step decay of the feature rate (utils.py:1044) for callers that finish the objective with k_loss_final_bwd,
    // whose workgroups all READ the rates: nothing reads them during this launch
    if (rates_decay && tid == 0) rates_decay[1] *= fea_decay;
    float s_img = 0.f, s_fea = 0.f, s_sig = 0.f, s_col = 0.f;
    for (uint32_t i = tid; i < n_img; i += stride) { const float d = img_t[i] - img_s[i]; s_img += d * d; }
    if (fea_width == 1u) {  // models without a feature vector (Plenoxel student): the rows hold sigma_l alone, no feature term
        for (uint32_t i = tid; i < M; i += stride) { const float d = fea_s[i] - fea_t[i]; s_sig += d * d; }
    } else {
        // feature rows as float4 quarters: quarter q of row m; column 0 (q == 0, .x) is also the sigma term
        for (uint32_t i = tid; i < M * 4u; i += stride) {
            const float4 a = reinterpret_cast<const float4 *>(fea_s)[i], b = reinterpret_cast<const float4 *>(fea_t)[i];
            const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
            s_fea += dx * dx + dy * dy + dz * dz + dw * dw;
            if ((i & 3u) == 0) s_sig += dx * dx;
        }
    }
    for (uint32_t i = tid; i < M * 3u; i += stride) { const float d = col_s[i] - col_t[i]; s_col += d * d; }
    s_img = block_sum(s_img, sh); s_fea = block_sum(s_fea, sh); s_sig = block_sum(s_sig, sh); s_col = block_sum(s_col, sh);
    // per-block partials (4096 same-line atomics would serialise at ~12 ns each); reduced by k_sumsq4_reduce
    if (threadIdx.x == 0) reinterpret_cast<float4 *>(S + 4)[blockIdx.x] = make_float4(s_img, s_fea, s_sig, s_col);
}

__global__ void __launch_bounds__(kLossBlock) k_sumsq4_reduce(float *__restrict__ S, uint32_t nblocks) {
    __shared__ float sh[kLossBlock / 64];
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (uint32_t i = threadIdx.x; i < nblocks; i += kLossBlock) {
        const float4 v = reinterpret_cast<const float4 *>(S + 4)[i];
        a += v.x; b += v.y; c += v.z; d += v.w;
    }
    a = block_sum(a, sh); b = block_sum(b, sh); c = block_sum(c, sh); d = block_sum(d, sh);
    if (threadIdx.x == 0) { S[0] = a; S[1] = b; S[2] = c; S[3] = d; }
}

// One workgroup: (optionally) finish the sums of squares from the per-block partials, decay the feature rate the way
// train_step does before using it (utils.py:1044), add a parameter-only term given as partial sums (the L1 regulariser,
// whose gradient lives in the optimizer kernel), write loss / coefficients / norms.
__global__ void __launch_bounds__(kLossBlock) k_loss_final(float *__restrict__ S, uint32_t reduce_blocks, float *__restrict__ rates,
                                                          float fea_decay, const float *__restrict__ extra, uint32_t n_extra,
                                                          float *__restrict__ loss, float *__restrict__ coef, float *__restrict__ norms) {
    __shared__ float sh[kLossBlock / 64];
    float s4[4];
    if (reduce_blocks) {
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
        for (uint32_t i = threadIdx.x; i < reduce_blocks; i += kLossBlock) {
            const float4 v = reinterpret_cast<const float4 *>(S + 4)[i];
            a += v.x; b += v.y; c += v.z; d += v.w;
        }
        s4[0] = block_sum(a, sh); s4[1] = block_sum(b, sh); s4[2] = block_sum(c, sh); s4[3] = block_sum(d, sh);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) s4[i] = S[i];
    }
    float e = 0.f;
    for (uint32_t i = threadIdx.x; i < n_extra; i += kLossBlock) e += extra[i];
    e = block_sum(e, sh);
    if (threadIdx.x != 0) return;
    float total = e;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float r = rates[i];
        if (i == 1 && fea_decay != 1.0f) { r *= fea_decay; rates[1] = r; }
        const float n = sqrtf(s4[i]);
        S[i] = s4[i];
        norms[i] = n;
        total += r * n;
        coef[i] = n > 0.f ? r / n : 0.f;  // d (r ||x||) / dx = r x / ||x||
    }
    loss[0] = total;
}

__global__ void __launch_bounds__(kLossBlock) k_sumsq4_bwd(const float *__restrict__ img_s, const float *__restrict__ img_t, uint32_t n_img,
                                                          const float *__restrict__ fea_s, const float *__restrict__ fea_t, uint32_t M,
                                                          const float *__restrict__ col_s, const float *__restrict__ col_t,
                                                          const float *__restrict__ coef, const float *__restrict__ upstream,
                                                          float *__restrict__ g_img, float *__restrict__ g_fea, float *__restrict__ g_col,
                                                          uint32_t fea_width) {
    const uint32_t tid = blockIdx.x * kLossBlock + threadIdx.x, stride = gridDim.x * kLossBlock;
    const float up = upstream[0];
    const float c_img = coef[0] * up, c_fea = coef[1] * up, c_sig = coef[2] * up, c_col = coef[3] * up;
    for (uint32_t i = tid; i < n_img; i += stride) g_img[i] = c_img * (img_s[i] - img_t[i]);
    if (fea_width == 1u) {
        for (uint32_t i = tid; i < M; i += stride) g_fea[i] = c_sig * (fea_s[i] - fea_t[i]);
    } else {
        for (uint32_t i = tid; i < M * 4u; i += stride) {
            const float4 a = reinterpret_cast<const float4 *>(fea_s)[i], b = reinterpret_cast<const float4 *>(fea_t)[i];
            float4 g = make_float4(c_fea * (a.x - b.x), c_fea * (a.y - b.y), c_fea * (a.z - b.z), c_fea * (a.w - b.w));
            if ((i & 3u) == 0) g.x += c_sig * (a.x - b.x);
            reinterpret_cast<float4 *>(g_fea)[i] = g;
        }
    }
    for (uint32_t i = tid; i < M * 3u; i += stride) g_col[i] = c_col * (col_s[i] - col_t[i]);
}

// k_loss_final + k_sumsq4_bwd in one launch.  EVERY workgroup finishes the four sums (<= 256 partials: one per thread) and
// forms the coefficients for itself -- 4 KB of L2 hits and five block reductions against an 8 us single-workgroup launch in
// the middle of the step -- then writes its share of the gradients; workgroup 0 also publishes loss / norms / coefficients.
// The rates are only read here (the decay happened in k_sumsq4).
__global__ void __launch_bounds__(kLossBlock) k_loss_final_bwd(const float *__restrict__ img_s, const float *__restrict__ img_t, uint32_t n_img,
                                                              const float *__restrict__ fea_s, const float *__restrict__ fea_t, uint32_t M,
                                                              const float *__restrict__ col_s, const float *__restrict__ col_t,
                                                              float *__restrict__ S, uint32_t reduce_blocks, const float *__restrict__ rates,
                                                              const float *__restrict__ extra, uint32_t n_extra,
                                                              const float *__restrict__ upstream, float *__restrict__ loss,
                                                              float *__restrict__ coef_out, float *__restrict__ norms,
                                                              float *__restrict__ g_img, float *__restrict__ g_fea, float *__restrict__ g_col,
                                                              uint32_t fea_width) {
    __shared__ float sh[kLossBlock / 64];
    float s4[4];
    if (reduce_blocks) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x < reduce_blocks) v = reinterpret_cast<const float4 *>(S + 4)[threadIdx.x];  // reduce_blocks <= kLossBlock
        s4[0] = block_sum(v.x, sh); s4[1] = block_sum(v.y, sh); s4[2] = block_sum(v.z, sh); s4[3] = block_sum(v.w, sh);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) s4[i] = S[i];
    }
    float coef[4], nrm[4], total = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float r = rates[i];
        nrm[i] = sqrtf(s4[i]);
        total += r * nrm[i];
        coef[i] = nrm[i] > 0.f ? r / nrm[i] : 0.f;
    }
    if (blockIdx.x == 0) {  // the value of the objective (same summation order as k_loss_final: parameter-only term first)
        float e = 0.f;
        for (uint32_t i = threadIdx.x; i < n_extra; i += kLossBlock) e += extra[i];
        e = block_sum(e, sh);
        if (threadIdx.x == 0) {
            float t = e;
#pragma unroll
            for (int i = 0; i < 4; i++) { t += rates[i] * nrm[i]; S[i] = s4[i]; norms[i] = nrm[i]; coef_out[i] = coef[i]; }
            loss[0] = t;
        }
    }
    (void)total;
    const uint32_t tid = blockIdx.x * kLossBlock + threadIdx.x, stride = gridDim.x * kLossBlock;
    const float up = upstream[0];
    const float c_img = coef[0] * up, c_fea = coef[1] * up, c_sig = coef[2] * up, c_col = coef[3] * up;
    for (uint32_t i = tid; i < n_img; i += stride) g_img[i] = c_img * (img_s[i] - img_t[i]);
    if (fea_width == 1u) {
        for (uint32_t i = tid; i < M; i += stride) g_fea[i] = c_sig * (fea_s[i] - fea_t[i]);
    } else {
        for (uint32_t i = tid; i < M * 4u; i += stride) {
            const float4 a = reinterpret_cast<const float4 *>(fea_s)[i], b = reinterpret_cast<const float4 *>(fea_t)[i];
            float4 g = make_float4(c_fea * (a.x - b.x), c_fea * (a.y - b.y), c_fea * (a.z - b.z), c_fea * (a.w - b.w));
            if ((i & 3u) == 0) g.x += c_sig * (a.x - b.x);
            reinterpret_cast<float4 *>(g_fea)[i] = g;
        }
    }
    for (uint32_t i = tid; i < M * 3u; i += stride) g_col[i] = c_col * (col_s[i] - col_t[i]);
}

}  // namespace pvd

using namespace pvd;

namespace pvd {
// Mean squared error of two images and its gradient in ONE launch of one workgroup (the teacher's objective, just_train_tea/utils.py:
// 573-581: MSELoss(reduction='none'), .mean(-1), .mean() -- 12 288 elements): loss = sum (a - b)^2 / n, d = 2 (a - b) / n.  The library's
// formulation is four to six launches of ~5 us each (elementwise, mean, and their autograd nodes) on the step's dependent chain.
constexpr uint32_t kMseBlock = 1024;
__global__ void __launch_bounds__(kMseBlock) k_mse(const float *__restrict__ a, const float *__restrict__ b, uint32_t n, float *__restrict__ loss,
                                                   float *__restrict__ d) {
    __shared__ float sh[kMseBlock / 64];
    const float k = 2.0f / (float)n;
    float acc = 0.f;
    for (uint32_t i = threadIdx.x; i < n; i += kMseBlock) {
        const float e = a[i] - b[i];
        acc += e * e;
        d[i] = k * e;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
#pragma unroll
        for (uint32_t w = 0; w < kMseBlock / 64; w++) r += sh[w];
        loss[0] = r / (float)n;
    }
}
}  // namespace pvd

extern "C" {

int pvd_mse_forward(const float *pred, const float *target, uint32_t n, float *loss, float *dloss_dpred, pvd_stream_t stream) {
    if (!pred || !target || !loss || !dloss_dpred || n == 0) return PVD_ERR_INVALID;
    hipLaunchKernelGGL(k_mse, dim3(1), dim3(kMseBlock), 0, (hipStream_t)stream, pred, target, n, loss, dloss_dpred);
    return check_launch();
}

static uint32_t sumsq_blocks(uint32_t n_img, uint32_t M) {
    uint32_t blocks = div_up(M * 4u > n_img ? M * 4u : n_img, kLossBlock);
    if (blocks > kSumsqMaxBlocks) blocks = kSumsqMaxBlocks;
    return blocks < 1 ? 1 : blocks;
}

int pvd_distill_sumsq(const float *img_stu, const float *img_tea, uint32_t n_img, const float *fea_stu, const float *fea_tea, uint32_t M,
                      uint32_t fea_width, const float *col_stu, const float *col_tea, float *S4, int reduce, float *rates4_decay,
                      float fea_decay, pvd_stream_t stream) {
    if (!img_stu || !img_tea || !fea_stu || !fea_tea || !col_stu || !col_tea || !S4) return PVD_ERR_INVALID;
    if (fea_width != 16u && fea_width != 1u) return PVD_ERR_INVALID;  // rows are read as four float4, or hold sigma_l alone
    hipStream_t s = (hipStream_t)stream;
    const uint32_t blocks = sumsq_blocks(n_img, M);
    hipLaunchKernelGGL(k_sumsq4, dim3(blocks), dim3(kLossBlock), 0, s, img_stu, img_tea, n_img, fea_stu, fea_tea, M, col_stu, col_tea, S4,
                       rates4_decay, fea_decay, fea_width);
    if (reduce) hipLaunchKernelGGL(k_sumsq4_reduce, dim3(1), dim3(kLossBlock), 0, s, S4, blocks);
    return blocks > 0 ? check_launch() : PVD_ERR_INVALID;
}

int pvd_distill_loss_final(float *S4, uint32_t n_img, uint32_t M, int reduce, float *rates4, float fea_decay, const float *extra,
                           uint32_t n_extra, float *loss, float *coef4, float *norms4, pvd_stream_t stream) {
    if (!S4 || !rates4 || !loss || !coef4 || !norms4 || (n_extra && !extra)) return PVD_ERR_INVALID;
    // reduce: 0 = S4[0..4) are final, 1 = pvd_distill_sumsq's partials (their count follows from n_img / M), >= 2 = that many
    // partials as they are (pvd_composite_objective_forward: pvd_composite_objective_blocks(N, rows))
    const uint32_t nparts = reduce >= 2 ? (uint32_t)reduce : (reduce ? sumsq_blocks(n_img, M) : 0u);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(kLossBlock), 0, (hipStream_t)stream, S4, nparts, rates4,
                       fea_decay, extra, n_extra, loss, coef4, norms4);
    return check_launch();
}

int pvd_distill_sumsq_backward(const float *img_stu, const float *img_tea, uint32_t n_img, const float *fea_stu, const float *fea_tea,
                               uint32_t M, uint32_t fea_width, const float *col_stu, const float *col_tea, const float *coef4,
                               const float *upstream, float *g_img, float *g_fea, float *g_col, pvd_stream_t stream) {
    if (!img_stu || !img_tea || !fea_stu || !fea_tea || !col_stu || !col_tea || !coef4 || !upstream || !g_img || !g_fea || !g_col)
        return PVD_ERR_INVALID;
    if (fea_width != 16u && fea_width != 1u) return PVD_ERR_INVALID;
    uint32_t blocks = div_up(M * 4u > n_img ? M * 4u : n_img, kLossBlock);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sumsq4_bwd, dim3(blocks), dim3(kLossBlock), 0, (hipStream_t)stream, img_stu, img_tea, n_img, fea_stu, fea_tea, M,
                       col_stu, col_tea, coef4, upstream, g_img, g_fea, g_col, fea_width);
    return check_launch();
}

int pvd_distill_loss_backward(const float *img_stu, const float *img_tea, uint32_t n_img, const float *fea_stu, const float *fea_tea,
                              uint32_t M, uint32_t fea_width, const float *col_stu, const float *col_tea, float *S4, int reduce,
                              const float *rates4, const float *extra, uint32_t n_extra, const float *upstream, float *loss, float *coef4,
                              float *norms4, float *g_img, float *g_fea, float *g_col, pvd_stream_t stream) {
    if (!img_stu || !img_tea || !fea_stu || !fea_tea || !col_stu || !col_tea || !S4 || !rates4 || !upstream || !loss || !coef4 || !norms4 ||
        !g_img || !g_fea || !g_col || (n_extra && !extra))
        return PVD_ERR_INVALID;
    if (fea_width != 16u && fea_width != 1u) return PVD_ERR_INVALID;
    uint32_t blocks = div_up(M * 4u > n_img ? M * 4u : n_img, kLossBlock);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_loss_final_bwd, dim3(blocks), dim3(kLossBlock), 0, (hipStream_t)stream, img_stu, img_tea, n_img, fea_stu, fea_tea, M,
                       col_stu, col_tea, S4, reduce ? sumsq_blocks(n_img, M) : 0u, rates4, extra, n_extra, upstream, loss, coef4, norms4, g_img,
                       g_fea, g_col, fea_width);
    return check_launch();
}

}  // extern "C"
