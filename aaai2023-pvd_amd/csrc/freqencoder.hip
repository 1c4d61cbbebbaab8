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

template <typename T>
__global__ void __launch_bounds__(kFreqBlock) k_freq_encode(const float *__restrict__ x, uint32_t M, uint32_t D, uint32_t n_freqs, FreqBands bands,
                                                           uint32_t include_input, T *__restrict__ out, uint32_t stride) {
    // one thread per (row, output column): consecutive threads write consecutive columns of a row
    const uint32_t width = (include_input ? D : 0u) + 2u * D * n_freqs;
    const uint64_t i = (uint64_t)blockIdx.x * kFreqBlock + threadIdx.x;
    if (i >= (uint64_t)M * stride) return;
    const uint32_t row = (uint32_t)(i / stride), c = (uint32_t)(i - (uint64_t)row * stride);
    float v = 0.f;  // padding columns
    if (c < width) {
        uint32_t cc = c;
        if (include_input && cc < D) {
            v = x[(size_t)row * D + cc];
        } else {
            if (include_input) cc -= D;
            const uint32_t k = cc / (2u * D), r = cc - k * 2u * D;  // frequency, then sin block | cos block
            const float a = x[(size_t)row * D + (r < D ? r : r - D)] * bands.f[k];
            v = r < D ? sinf(a) : cosf(a);
        }
    }
    out[i] = (T)v;
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
    const uint64_t total = (uint64_t)M * row_stride;
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
