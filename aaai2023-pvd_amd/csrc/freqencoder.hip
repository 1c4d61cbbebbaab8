// freqencoder.hip -- NeRF positional encoding for gfx950 (MI355X).
//
// Torch code in the reference: FreqEncoder.forward, tools/encoding.py:6-49 -- [x, sin(f0 x), cos(f0 x), sin(f1 x), ...]
// with f = 2^linspace(0, max_freq_log2, N) -- one sin and one cos launch (plus a multiply each) per frequency and a
// concatenation: 40 launches of ~4.5 us for the 63-wide encoding of the `mlp` model (rocprofv3, round 2).  Here: one launch,
// rows written once, optionally as f16 (what the first Linear casts them to under autocast) and padded with zero columns to
// a row stride the GEMM likes.
#include "pvd_device.h"

namespace pvd {

constexpr uint32_t kFreqBlock = 256;
constexpr uint32_t kMaxFreqs = 16;

struct FreqBands {
    float f[kMaxFreqs];
};

// One thread per (row, frequency, input dimension): one sincosf, the sine and the cosine written to their two places in the
// row (a thread per output element -- a sin OR a cos each, an integer division each -- took 18.6 us for 92 k rows of 64).  The
// last threads of a row copy the inputs and clear the padding.
template <typename T>
__global__ void __launch_bounds__(kFreqBlock) k_freq_encode(const float *__restrict__ x, uint32_t M, uint32_t D, uint32_t n_freqs, FreqBands bands,
                                                           uint32_t include_input, T *__restrict__ out, uint32_t stride) {
    const uint32_t per_row = D * n_freqs + 1u;  // D * n_freqs sincos slots + one slot for inputs and padding
    const uint64_t i = (uint64_t)blockIdx.x * kFreqBlock + threadIdx.x;
    if (i >= (uint64_t)M * per_row) return;
    const uint32_t row = (uint32_t)(i / per_row), j = (uint32_t)(i - (uint64_t)row * per_row);
    T *__restrict__ o = out + (size_t)row * stride;
    const uint32_t base = include_input ? D : 0u;
    if (j < D * n_freqs) {
        const uint32_t k = j / D, d = j - k * D;
        const float a = x[(size_t)row * D + d] * bands.f[k];
        o[base + 2u * D * k + d] = (T)sinf(a);
        o[base + 2u * D * k + D + d] = (T)cosf(a);
    } else {
        if (include_input)
            for (uint32_t d = 0; d < D; d++) o[d] = (T)x[(size_t)row * D + d];
        for (uint32_t c = base + 2u * D * n_freqs; c < stride; c++) o[c] = (T)0.f;
    }
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_freq_encode(const float *x, uint32_t M, uint32_t D, const float *freq_bands_host, uint32_t n_freqs, int include_input, void *out,
                    int out_dtype, uint32_t row_stride, pvd_stream_t stream) {
    if (M == 0) return PVD_OK;
    if (!x || !out || (n_freqs && !freq_bands_host)) return PVD_ERR_INVALID;
    const uint32_t width = (include_input ? D : 0u) + 2u * D * n_freqs;
    if (D < 1 || n_freqs > kMaxFreqs || row_stride < width) return PVD_ERR_INVALID;
    FreqBands b;
    for (uint32_t k = 0; k < kMaxFreqs; k++) b.f[k] = k < n_freqs ? freq_bands_host[k] : 0.f;
    const uint64_t total = (uint64_t)M * (D * n_freqs + 1u);
    const dim3 grid((uint32_t)((total + kFreqBlock - 1) / kFreqBlock)), block(kFreqBlock);
    if (out_dtype == PVD_F32)
        hipLaunchKernelGGL((k_freq_encode<float>), grid, block, 0, (hipStream_t)stream, x, M, D, n_freqs, b, include_input ? 1u : 0u, (float *)out,
                           row_stride);
    else if (out_dtype == PVD_F16)
        hipLaunchKernelGGL((k_freq_encode<_Float16>), grid, block, 0, (hipStream_t)stream, x, M, D, n_freqs, b, include_input ? 1u : 0u,
                           (_Float16 *)out, row_stride);
    else
        return PVD_ERR_UNSUPPORTED;
    return check_launch();
}

}  // extern "C"
