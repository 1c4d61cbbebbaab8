// LDS accumulate flavours on gfx950: cycles per 64-lane operation, lane = consecutive dword (the binned VM backward's pattern),
// `waves` waves of one workgroup hammering a 64 KB accumulator at pseudo-random rows.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o lds_atomic_probe lds_atomic_probe.hip && ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kRows = 256;  // x 64 floats = 64 KB
template <int MODE>
__global__ void __launch_bounds__(1024) k_probe(float *out, uint64_t *clk, int iters) {
    __shared__ float acc[kRows * 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < kRows * 64; k += blockDim.x) acc[k] = 0.f;
    __syncthreads();
    uint32_t r = wave * 2654435761u + 12345u;
    const float v = 1.0f + lane;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        r = r * 1664525u + 1013904223u;
        const uint32_t row = __builtin_amdgcn_readfirstlane(r >> 24) % kRows;
        float *p = acc + row * 64 + lane;
        if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(p), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2) *p += v;  // (racy across waves: rate only)
        else if (MODE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 4) {  // a CAS loop on the bit pattern
            uint32_t *q = reinterpret_cast<uint32_t *>(p);
            uint32_t old = *q, seen;
            do { seen = old; old = atomicCAS(q, seen, __float_as_uint(__uint_as_float(seen) + v)); } while (old != seen);
        } else if (MODE == 5) {  // fp64 accumulators (32 KB of the array = 64 rows)
            double *q = reinterpret_cast<double *>(acc) + (row & 63u) * 64 + lane;
            __hip_atomic_fetch_add(q, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (lane == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
    float s = 0.f;
    for (uint32_t k = threadIdx.x; k < kRows * 64; k += blockDim.x) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int waves, int blocks) {
    float *out; uint64_t *clk;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&clk, blocks * 16 * 8);
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_probe<MODE><<<blocks, waves * 64>>>(out, clk, iters);
    hipEventRecord(a);
    k_probe<MODE><<<blocks, waves * 64>>>(out, clk, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t h[16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s %2d waves x %4d workgroups: %7.1f us, %.1f counter ticks (100 MHz) per op per wave; per CU %.2f ns per 64-lane op\n", name, waves, blocks,
           ms * 1e3, (double)h[0] / iters, ms * 1e6 / ((double)iters * waves * ((blocks + 255) / 256)));
    hipFree(out); hipFree(clk);
}
int main() {
    for (int waves : {1, 4, 16}) {
        run<0>("ds float atomic add (workgroup)", waves, 256);
        run<3>("ds float atomic add (wavefront)", waves, 256);
        run<1>("ds u32 atomic add", waves, 256);
        run<2>("plain read-add-write", waves, 256);
        run<4>("CAS loop", waves, 256);
        run<5>("ds f64 atomic add", waves, 256);
    }
    return 0;
}
