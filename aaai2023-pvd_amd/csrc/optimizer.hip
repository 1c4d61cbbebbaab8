// optimizer.hip -- AdamW over ONE flat parameter / gradient / moment buffer (gfx950).
//
// The reference's optimiser is torch.optim.AdamW(betas=(0.9, 0.99), eps=1e-15) on ~20 tensors
// (main_distill_mutual.py:334-339).  The student's parameters, gradients and moments live here in four flat
// fp32 buffers (trainer.py), so the update is a single streaming pass -- 28 B/parameter, pure HBM bandwidth --
// instead of one multi-tensor launch per parameter group.  Same arithmetic as torch's fused AdamW kernel
// (decoupled weight decay, lerp first moment, bias corrections from a device-side step count, optional
// GradScaler unscale + skip-on-inf), so it is capturable in a HIP graph.
#include "pvd_device.h"

#include <math.h>

namespace pvd {

constexpr uint32_t kOptBlock = 256;
constexpr uint32_t kMaxSegments = 16;

struct AdamSegments {
    uint64_t end[kMaxSegments];  // segment k covers [end[k-1], end[k]); its learning rate is lr[k] (device)
    uint32_t count;
};

__global__ void k_adamw_count(float *__restrict__ step, const float *__restrict__ found_inf) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(found_inf && found_inf[0] != 0.f)) step[0] += 1.0f;
}

__global__ void __launch_bounds__(kOptBlock) k_adamw(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, uint64_t n, AdamSegments seg, const float *__restrict__ lr,
                                                    double beta1, double beta2, double eps, double weight_decay,
                                                    const float *__restrict__ step, const float *__restrict__ grad_scale,
                                                    const float *__restrict__ found_inf) {
    if (found_inf && found_inf[0] != 0.f) return;  // GradScaler: skip the whole step
    const double t = (double)step[0];
    const double bc1 = 1.0 - pow((double)beta1, t);
    const double bc2_sqrt = sqrt(1.0 - pow((double)beta2, t));
    const uint64_t n4 = n >> 2;
    for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kOptBlock) {
        const uint64_t e = i << 2;
        uint32_t k = 0;
        while (k + 1 < seg.count && e >= seg.end[k]) k++;  // segments are multiples of 4 elements (see trainer)
        const double lrk = (double)lr[k];
        const float step_size = (float)(lrk / bc1);
        float4 P = reinterpret_cast<float4 *>(p)[i], G = reinterpret_cast<const float4 *>(g)[i];
        float4 M = reinterpret_cast<float4 *>(m)[i], V = reinterpret_cast<float4 *>(v)[i];
        float *pp = &P.x, *gg = &G.x, *mm = &M.x, *vv = &V.x;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // torch's fused kernel keeps the hyper-parameters in double, so these expressions evaluate in fp64
            // and round once into the fp32 state (ATen fused_adam_utils.cuh: adam_math, ADAMW mode)
            const float grad = grad_scale ? (float)((double)gg[c] / (double)grad_scale[0]) : gg[c];
            float param = (float)((double)pp[c] - lrk * (double)weight_decay * (double)pp[c]);
            const float ea = (float)((double)mm[c] + (1.0 - (double)beta1) * ((double)grad - (double)mm[c]));
            const float es = (float)((double)beta2 * (double)vv[c] + (1.0 - (double)beta2) * (double)grad * (double)grad);
            const float denom = (float)((double)sqrtf(es) / bc2_sqrt + (double)eps);
            param -= step_size * ea / denom;
            pp[c] = param; mm[c] = ea; vv[c] = es;
        }
        reinterpret_cast<float4 *>(p)[i] = P;
        reinterpret_cast<float4 *>(m)[i] = M;
        reinterpret_cast<float4 *>(v)[i] = V;
    }
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_adamw_step(float *p, const float *g, float *m, float *v, uint64_t n, const uint64_t *segment_ends_host, uint32_t n_segments,
                   const float *lr, double beta1, double beta2, double eps, double weight_decay, float *step, const float *grad_scale,
                   const float *found_inf, pvd_stream_t stream) {
    if (n == 0) return PVD_OK;
    if (!p || !g || !m || !v || !segment_ends_host || !lr || !step) return PVD_ERR_INVALID;
    if (n_segments < 1 || n_segments > kMaxSegments || (n & 3u)) return PVD_ERR_UNSUPPORTED;
    AdamSegments seg;
    seg.count = n_segments;
    for (uint32_t k = 0; k < n_segments; k++) {
        if (segment_ends_host[k] & 3u) return PVD_ERR_UNSUPPORTED;
        seg.end[k] = segment_ends_host[k];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adamw_count, dim3(1), dim3(64), 0, s, step, found_inf);
    uint64_t blocks = (n / 4 + kOptBlock - 1) / kOptBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_adamw, dim3((uint32_t)blocks), dim3(kOptBlock), 0, s, p, g, m, v, n, seg, lr, beta1, beta2, eps, weight_decay, step,
                       grad_scale, found_inf);
    return check_launch();
}

}  // extern "C"
