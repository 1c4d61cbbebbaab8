// shencoder.hip -- real spherical-harmonics direction encoder for gfx950 (MI355X).
//
// Replaces the reference's _shencoder module (shencoder/src/shencoder.cu).  The basis is the
// polynomial family documented in tools/gen_sh_coeffs.py (same functions on R^3 as the
// reference's 64 hand-expanded polynomials, shencoder.cu:51-125), evaluated by Horner in z^2
// times a (x+iy)^m recurrence from the generated sh_basis.inc.  float32 only, like the
// reference wrapper (shencoder/sphere_harmonics.py:17 forces float32).
#include "pvd_device.h"

namespace pvd {

#include "sh_basis.inc"

constexpr uint32_t kShBlock = 256;

// reference: kernel_sh, shencoder.cu:27-356.  One thread per direction; the DEG*DEG outputs
// are built in registers and written as 16-byte rows.
template <int DEG, bool GRAD>
__global__ void __launch_bounds__(kShBlock) k_sh_fwd(const float *__restrict__ inputs, float *__restrict__ outputs, uint32_t B,
                                                     float *__restrict__ dy_dx) {
    constexpr int C2 = DEG * DEG;
    const uint32_t b = blockIdx.x * kShBlock + threadIdx.x;
    if (b >= B) return;
    const float x = inputs[3 * (size_t)b], y = inputs[3 * (size_t)b + 1], z = inputs[3 * (size_t)b + 2];
    float o[C2];
    float gx[GRAD ? C2 : 1], gy[GRAD ? C2 : 1], gz[GRAD ? C2 : 1];
    pvd_sh_basis<DEG, GRAD>(
        x, y, z, [&](int i, float v) { o[i] = v; },
        [&](int i, float vx, float vy, float vz) {
            if constexpr (GRAD) { gx[i] = vx; gy[i] = vy; gz[i] = vz; }
        });
    float *__restrict__ out = outputs + (size_t)b * C2;
    if constexpr (C2 % 4 == 0) {
#pragma unroll
        for (int i = 0; i < C2; i += 4) *reinterpret_cast<float4 *>(out + i) = make_float4(o[i], o[i + 1], o[i + 2], o[i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < C2; i++) out[i] = o[i];
    }
    if constexpr (GRAD) {  // dy_dx [B, 3, C2] (shencoder.cu:128-131)
        float *__restrict__ d = dy_dx + (size_t)b * 3 * C2;
#pragma unroll
        for (int i = 0; i < C2; i++) { d[i] = gx[i]; d[C2 + i] = gy[i]; d[2 * C2 + i] = gz[i]; }
    }
}

// reference: kernel_sh_backward, shencoder.cu:359-383 (accumulates with +=)
__global__ void __launch_bounds__(kShBlock) k_sh_bwd(const float *__restrict__ grad, uint32_t B, uint32_t C2,
                                                     const float *__restrict__ dy_dx, float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kShBlock + threadIdx.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    const float *__restrict__ g = grad + (size_t)b * C2;
    const float *__restrict__ dd = dy_dx + (size_t)b * 3 * C2 + (size_t)d * C2;
    float r = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) r += g[ch] * dd[ch];
    grad_inputs[t] = r;
}

template <int DEG>
static int launch_sh(const float *inputs, float *outputs, uint32_t B, bool calc, float *dy_dx, hipStream_t s) {
    const dim3 grid(div_up(B, kShBlock)), block(kShBlock);
    if (calc) hipLaunchKernelGGL((k_sh_fwd<DEG, true>), grid, block, 0, s, inputs, outputs, B, dy_dx);
    else hipLaunchKernelGGL((k_sh_fwd<DEG, false>), grid, block, 0, s, inputs, outputs, B, dy_dx);
    return check_launch();
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                          float *dy_dx, pvd_stream_t stream) {
    if (D != 3 || C < 1 || C > 8) return PVD_ERR_UNSUPPORTED;  // sphere_harmonics.py:75-78
    if (B == 0) return PVD_OK;
    if (!inputs || !outputs || (calc_grad_inputs && !dy_dx)) return PVD_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const bool calc = calc_grad_inputs != 0;
    switch (C) {
        case 1: return launch_sh<1>(inputs, outputs, B, calc, dy_dx, s);
        case 2: return launch_sh<2>(inputs, outputs, B, calc, dy_dx, s);
        case 3: return launch_sh<3>(inputs, outputs, B, calc, dy_dx, s);
        case 4: return launch_sh<4>(inputs, outputs, B, calc, dy_dx, s);
        case 5: return launch_sh<5>(inputs, outputs, B, calc, dy_dx, s);
        case 6: return launch_sh<6>(inputs, outputs, B, calc, dy_dx, s);
        case 7: return launch_sh<7>(inputs, outputs, B, calc, dy_dx, s);
        default: return launch_sh<8>(inputs, outputs, B, calc, dy_dx, s);
    }
}

int pvd_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C, const float *dy_dx,
                           float *grad_inputs, pvd_stream_t stream) {
    (void)inputs;
    if (D != 3 || C < 1 || C > 8) return PVD_ERR_UNSUPPORTED;
    if (B == 0) return PVD_OK;
    if (!grad || !dy_dx || !grad_inputs) return PVD_ERR_INVALID;
    hipLaunchKernelGGL(k_sh_bwd, dim3(div_up(B * 3, kShBlock)), dim3(kShBlock), 0, (hipStream_t)stream, grad, B, C * C, dy_dx, grad_inputs);
    return check_launch();
}

}  // extern "C"
