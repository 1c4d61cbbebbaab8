// dda.h -- one probe of the occupancy-grid march (reference: the loop bodies of raymarching.cu:362-403, 430-482, 756-810), shared by
// raymarching.hip (the _raymarching entry points) and fusedhead.hip (the persistent inference kernel of a frozen hash model).
#pragma once

#include "pvd_device.h"

namespace pvd {

constexpr float kSqrt3 = 1.7320508075688772f;  // SQRT3(), raymarching.cu:21

// ------------------------------------------------------------------ DDA

struct Dda {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, dt_const, rH, Cf, Hf, cell_hi;
    uint32_t H, H3;
    const uint8_t *grid;
    // wave-uniform fast paths with identical results: one cascade (level is always 0) and a power-of-two H
    // (0.5 * v * H in fp64, rounded to fp32, equals the fp32 product v * (H / 2): a power-of-two scaling is exact)
    bool one_cascade, pow2_h;
    float mip_bound0, mip_rbound0, half_h;

    __device__ __forceinline__ void init(const float *o, const float *d, float bound_, float dt_gamma_, uint32_t max_steps,
                                         uint32_t C, uint32_t H_, const uint8_t *grid_) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = 1.0f / dx; rdy = 1.0f / dy; rdz = 1.0f / dz;
        bound = bound_; dt_gamma = dt_gamma_;
        dt_min = 2 * kSqrt3 / (float)max_steps;
        dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / (float)H_;
        // the step of a march with dt_gamma == 0: clamp(0, dt_min, dt_max) = fminf(dt_max, fmaxf(dt_min, 0)) (:36-38, :368) -- dt_min
        // unless max_steps is so small that dt_min exceeds dt_max (max_steps < H / 2^(C-1)): then dt_max (round 6, found against the
        // reference's own kernel)
        dt_const = clampf(0.0f, dt_min, dt_max);
        rH = 1.0f / (float)H_;
        Cf = (float)C; Hf = (float)H_; cell_hi = (float)(H_ - 1);
        H = H_; H3 = H_ * H_ * H_;
        grid = grid_;
        one_cascade = C == 1;
        pow2_h = (H_ & (H_ - 1)) == 0 && H_ >= 2;
        mip_bound0 = fminf(1.0f, bound_);
        mip_rbound0 = 1.0f / mip_bound0;
        half_h = 0.5f * (float)H_;
    }

    // One probe at parameter t (reference: the loop bodies at raymarching.cu:362-403,
    // 430-482, 756-810).  Returns true when the cell is occupied; otherwise t_next is the
    // first t, advanced in whole dt steps, at or past the cell's exit face.
    template <bool SKIP_TARGET_ONLY>
    __device__ __forceinline__ bool probe_impl(float t, float &x, float &y, float &z, float &dt, float &t_next) const {
        x = clampf(fmaf(t, dx, ox), -bound, bound);
        y = clampf(fmaf(t, dy, oy), -bound, bound);
        z = clampf(fmaf(t, dz, oz), -bound, bound);
        dt = clampf(t * dt_gamma, dt_min, dt_max);

        int level = 0;
        float mip_bound = mip_bound0, mip_rbound = mip_rbound0;
        if (!one_cascade) {
            const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
            int e_pos, e_dt;
            (void)frexpf(mx, &e_pos);
            const float dmx = (float)((double)(dt * Hf) * 0.5);  // double literal in the reference (:52)
            (void)frexpf(dmx, &e_dt);
            const int lvl_pos = (int)fminf(Cf - 1, fmaxf(0.0f, (float)e_pos));
            const int lvl_dt = (int)fminf(Cf - 1, fmaxf(0.0f, (float)e_dt));
            level = lvl_pos > lvl_dt ? lvl_pos : lvl_dt;
            mip_bound = fminf((float)(1 << level), bound);
            mip_rbound = 1.0f / mip_bound;
        }

        // nearest cell via fp64 temporaries, as the reference source promotes (:377-379)
        int nx, ny, nz;
        // x * mip_rbound + 1 is ONE fused operation in the reference's builds (v_fma in the ISA of its own kernel compiled for gfx950,
        // oracle/_ref; nvcc contracts it too): exact product, hence no change, for a power-of-two mip_bound; decides knife-edge cells
        // for others (bound 1.5: 2 rays of 2048 differed by a sample before round 6)
        if (pow2_h) {
            nx = (int)clampf(fmaf(x, mip_rbound, 1.0f) * half_h, 0.0f, cell_hi);
            ny = (int)clampf(fmaf(y, mip_rbound, 1.0f) * half_h, 0.0f, cell_hi);
            nz = (int)clampf(fmaf(z, mip_rbound, 1.0f) * half_h, 0.0f, cell_hi);
        } else {
            const double Hd = (double)H;
            nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * Hd), 0.0f, cell_hi);
            ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * Hd), 0.0f, cell_hi);
            nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * Hd), 0.0f, cell_hi);
        }

        const uint32_t index = (uint32_t)level * H3 + morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
        const bool occ = (grid[index >> 3] >> (index & 7u)) & 1u;
        if (occ) return true;

        // (...) * mip_bound - x: contracted likewise
        const float tx = fmaf(((nx + 0.5f + 0.5f * sign1f(dx)) * rH * 2 - 1), mip_bound, -x) * rdx;
        const float ty = fmaf(((ny + 0.5f + 0.5f * sign1f(dy)) * rH * 2 - 1), mip_bound, -y) * rdy;
        const float tz = fmaf(((nz + 0.5f + 0.5f * sign1f(dz)) * rH * 2 - 1), mip_bound, -z) * rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        if (SKIP_TARGET_ONLY) {  // the wave-parallel marcher resolves the skip itself
            t_next = tt;
            return false;
        }
        do {
            t += clampf(t * dt_gamma, dt_min, dt_max);
        } while (t < tt);
        t_next = t;
        return false;
    }

    __device__ __forceinline__ bool probe(float t, float &x, float &y, float &z, float &dt, float &t_next) const {
        return probe_impl<false>(t, x, y, z, dt, t_next);
    }
};

__device__ __forceinline__ float ray_t0(float near, float dt_min, uint32_t perturb, uint64_t seed, uint32_t n) {
    if (!perturb) return near;
    Pcg32 g;
    g.seed(seed);
    g.advance(n);
    // `t0 += dt_min * rng.next_float()` (:353, :751) is ONE fused operation in the reference's builds (v_fmac_f32 in the ISA of its own
    // kernels compiled for gfx950, oracle/_ref; nvcc contracts it too): about one ray in 2000 lands an ulp away otherwise
    return fmaf(dt_min, g.next_float(), near);
}


}  // namespace pvd
