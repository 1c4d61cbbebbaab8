// Does v_mfma_f32_16x16x32_f16 fed with two concatenated K=16 fragment pairs equal the two K=16 MFMAs?  (tools/probes: not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 half_t;
typedef half_t h4 __attribute__((ext_vector_type(4)));
typedef half_t h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const half_t *A, const half_t *B, float *D16, float *D32) {
    const unsigned lane = threadIdx.x;
    h4 a0, a1, b0, b1;
    for (int j = 0; j < 4; j++) {
        a0[j] = A[(lane & 15) * 32 + 4 * (lane >> 4) + j];        // A[m][k], k-step 0
        a1[j] = A[(lane & 15) * 32 + 16 + 4 * (lane >> 4) + j];   // k-step 1
        b0[j] = B[(4 * (lane >> 4) + j) * 16 + (lane & 15)];      // B[k][n]
        b1[j] = B[(16 + 4 * (lane >> 4) + j) * 16 + (lane & 15)];
    }
    f4 z = {0, 0, 0, 0};
    f4 d16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, z, 0, 0, 0), 0, 0, 0);
    const h8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const h8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    f4 d32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
    for (int j = 0; j < 4; j++) { D16[lane * 4 + j] = d16[j]; D32[lane * 4 + j] = d32[j]; }
}
int main() {
    half_t hA[16 * 32], hB[32 * 16];
    srand(1);
    for (auto &v : hA) v = (half_t)((rand() % 2001 - 1000) / 1000.0f);
    for (auto &v : hB) v = (half_t)((rand() % 2001 - 1000) / 1000.0f);
    half_t *A, *B; float *D16, *D32;
    hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB); hipMalloc(&D16, 1024); hipMalloc(&D32, 1024);
    hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(A, B, D16, D32);
    float h16[256], h32[256];
    hipMemcpy(h16, D16, 1024, hipMemcpyDeviceToHost); hipMemcpy(h32, D32, 1024, hipMemcpyDeviceToHost);
    double worst = 0, worst_ref = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
        const int m = 4 * (l >> 4) + j, n = l & 15;  // D layout of the 16x16 tile: row 4 hi + j, column lane & 15
        double ref = 0;
        for (int kk = 0; kk < 32; kk++) ref += (double)hA[m * 32 + kk] * (double)hB[kk * 16 + n];
        (void)ref;
        worst = fmax(worst, fabs(h16[l * 4 + j] - h32[l * 4 + j]));
    }
    // which (m, n) does each D slot hold?  compare against the reference in both conventions
    double e_a = 0, e_b = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
        double ra = 0, rb = 0;
        for (int kk = 0; kk < 32; kk++) {
            ra += (double)hA[(4 * (l >> 4) + j) * 32 + kk] * (double)hB[kk * 16 + (l & 15)];  // row = 4hi+j (from A), col = lane&15 (from B)
            rb += (double)hA[(l & 15) * 32 + kk] * (double)hB[kk * 16 + 4 * (l >> 4) + j];    // transposed convention
        }
        e_a = fmax(e_a, fabs(h16[l * 4 + j] - ra)); e_b = fmax(e_b, fabs(h16[l * 4 + j] - rb));
        worst_ref = fmax(worst_ref, fmin(fabs(h32[l * 4 + j] - ra), fabs(h32[l * 4 + j] - rb)));
    }
    printf("max |K16 pair - K32| = %.3g ; K16 vs reference: convention A %.3g, convention B %.3g ; K32 vs nearer reference %.3g\n", worst, e_a, e_b, worst_ref);
    return 0;
}
