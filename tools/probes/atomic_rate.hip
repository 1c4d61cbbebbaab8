// Memory-side rate of 64-lane fp32 atomics on gfx950, the VM backward's flush pattern (lane = consecutive dword of one 256-byte
// texel row, pseudo-random rows, nothing waits for the acknowledgements):
//   agent      global_atomic_add_f32, agent scope (what k_vm_bwd_* issue: executed past the L2, TCC_EA0_WRREQ = TCC_ATOMIC)
//   workgroup  the same instruction at workgroup scope (does the XCD's L2 execute it?  rate only: the sums of different XCDs
//              would not meet in memory)
//   xcd        workgroup scope into a copy of the table private to the wave's XCD (XCC_ID): the L2-side alternative made coherent
//   store      plain stores of the same rows (what the write path alone sustains)
// and the latency of ONE acknowledged atomic per wave (s_waitcnt vmcnt(0) after each) -- the round trip a walk that waits pays.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float *__restrict__ buf, uint32_t rows, uint32_t iters, uint32_t distinct, uint64_t *__restrict__ ticks) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
    uint32_t r = wave * 2654435761u + 12345u;
    const float v = 1.0f;
    float *base = buf;
    if (MODE == 2) base += (size_t)xcc_id() * rows * 64;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < iters; i++) {
        r = r * 1664525u + 1013904223u;
        uint32_t row = ((r >> 8) % distinct) * (rows / distinct);
        if (MODE == 8) row = (wave * iters + i) % rows;                                   // every wave streams its own consecutive rows
        if (MODE == 9) row = ((wave * 64u) % (rows - 64u)) + ((r >> 8) & 63u);              // random inside the wave's own 16 KB window
        if (MODE == 10) row = (((r >> 8) % (rows / 16u)) * 16u) + (i & 15u);               // runs of 16 consecutive rows (4 KB), random runs
        float *p = base + (size_t)row * 64 + lane;
        if (MODE == 0 || MODE >= 8) { if (MODE != 11 || lane < 16) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (MODE == 1 || MODE == 2) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 3) __builtin_nontemporal_store(v, p);
        else if (MODE == 4) { *p = v; }
        else if (MODE == 5) {  // one acknowledged atomic at a time
            __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        } else if (MODE == 6) {  // one acknowledged workgroup-scope atomic at a time
            __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_waitcnt(0x0f70);
        } else if (MODE == 7) {  // one load at a time (the round trip of a read that misses the L1)
            const float x = __builtin_nontemporal_load(p);
            if (x == 12345.678f) buf[0] = x;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && ticks) ticks[wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, float *buf, uint32_t rows, uint32_t waves, uint32_t iters, uint32_t distinct, uint64_t *ticks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t blocks = waves / 4;
    hipLaunchKernelGGL((k_rate<MODE>), dim3(blocks), dim3(256), 0, 0, buf, rows, iters, distinct, ticks);
    hipEventRecord(a);
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL((k_rate<MODE>), dim3(blocks), dim3(256), 0, 0, buf, rows, iters, distinct, ticks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    uint64_t h[8];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double ops = (double)waves * iters;
    printf("%-10s waves %6u x %4u ops, %6u distinct rows: %8.1f us  %7.2f G 64-byte requests/s  (%5.2f TB/s of 256-byte rows); wave 0: %.0f ns per op\n", name, waves,
           iters, distinct, ms * 1e3, ops * 4 / (ms * 1e-3) * 1e-9, ops * 256 / (ms * 1e-3) * 1e-12, (double)h[0] * 10.0 / iters);
    fflush(stdout);
}

int main() {
    const uint32_t rows = 270000;  // three 300 x 300 planes of 64 channels: 69 MB
    float *buf;
    uint64_t *ticks;
    hipMalloc(&buf, (size_t)rows * 64 * 4 * 8);  // eight copies for the per-XCD variant
    hipMemset(buf, 0, (size_t)rows * 64 * 4 * 8);
    hipMalloc(&ticks, 1 << 20);
    for (uint32_t waves : {4096u, 8192u, 16384u}) {
        for (uint32_t distinct : {270000u, 47000u}) {
            const uint32_t iters = 1400000u / waves * 4;  // ~1.4 M row operations per launch, the VM backward's count x 4
            run<0>("agent", buf, rows, waves, iters, distinct, ticks);
            run<1>("workgroup", buf, rows, waves, iters, distinct, ticks);
            run<2>("xcd-copy", buf, rows, waves, iters, distinct, ticks);
            run<3>("nt-store", buf, rows, waves, iters, distinct, ticks);
            run<4>("store", buf, rows, waves, iters, distinct, ticks);
        }
    }
    for (uint32_t waves : {8192u}) {
        const uint32_t iters = 1400000u / waves * 4;
        run<8>("seq-rows", buf, rows, waves, iters, 270000u, ticks);
        run<9>("16KB-win", buf, rows, waves, iters, 270000u, ticks);
        run<10>("4KB-runs", buf, rows, waves, iters, 270000u, ticks);
        run<11>("16-lanes", buf, rows, waves, iters, 270000u, ticks);  // one 64-byte request per instruction (the line printed counts 4)
    }
    for (uint32_t waves : {256u, 1024u, 4096u}) {
        run<5>("agent+wait", buf, rows, waves, 200, 270000u, ticks);
        run<6>("wg+wait", buf, rows, waves, 200, 270000u, ticks);
        run<7>("load+wait", buf, rows, waves, 200, 270000u, ticks);
    }
    return 0;
}
