/*
 * pvd_oracle.c -- CPU ORACLE (test infrastructure, see pvd_oracle.h header).
 *
 * Plain C restatement of the reference's CUDA kernels; citations are to
 * /root/reference/<file>:<line>.  Build: oracle/Makefile (gcc -O2
 * -ffp-contract=off, OpenMP optional).  Parity: pinned by the reference's own
 * raymarching / SH kernels built for gfx950 (oracle/_ref, round 6); the grid
 * encoder is PARITY UNPINNED BY THE REFERENCE -- see pvd_oracle.h.
 */
#include "pvd_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SQRT3_F 1.7320508075688772f
#define RPI_F 0.3183098861837907f

int pvdo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void pvdo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ */
/* small float helpers                                                 */
/* ------------------------------------------------------------------ */

static inline float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
static inline float sign1f(float x) { return copysignf(1.0f, x); }

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE binary16 <-> binary32, round-to-nearest-even (what __float2half_rn /
 * c10::Half's constructor do; gridencoder.cu:143,166,303 rely on it). */
uint16_t pvdo_f32_to_f16(float f) {
    uint32_t x = f32_bits(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint16_t h;
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) {
        h = 0x7e00u; /* NaN */
    } else if (x >= 0x47800000u) {
        h = 0x7c00u; /* >= 65536 (incl. inf) */
    } else if (x < 0x38800000u) {
        /* result is a half subnormal (or zero): let the FPU do the RNE shift */
        const float a = bits_f32(x) + 0.5f;
        h = (uint16_t)(f32_bits(a) - 0x3f000000u);
    } else {
        const uint32_t odd = (x >> 13) & 1u;
        x += ((uint32_t)(15 - 127) << 23) + 0xfffu + odd;
        h = (uint16_t)(x >> 13); /* [65520, 65536) carries into 0x7c00 = inf */
    }
    return (uint16_t)(sign | h);
}

float pvdo_f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t em = (uint32_t)h & 0x7fffu;
    uint32_t out;
    if (em >= 0x7c00u) {
        out = sign | 0x7f800000u | ((em & 0x3ffu) << 13);
    } else if (em >= 0x0400u) {
        out = sign | ((em << 13) + ((uint32_t)(127 - 15) << 23));
    } else {
        /* subnormal: value = em * 2^-24, exact in float */
        const float v = (float)em * 5.9604644775390625e-08f;
        out = sign | f32_bits(v);
    }
    return bits_f32(out);
}

/* half arithmetic as c10::Half does it: compute in float, round to half */
static inline uint16_t h_add(uint16_t a, uint16_t b) {
    return pvdo_f32_to_f16(pvdo_f16_to_f32(a) + pvdo_f16_to_f32(b));
}
static inline uint16_t h_sub(uint16_t a, uint16_t b) {
    return pvdo_f32_to_f16(pvdo_f16_to_f32(a) - pvdo_f16_to_f32(b));
}
static inline uint16_t h_mul(uint16_t a, uint16_t b) {
    return pvdo_f32_to_f16(pvdo_f16_to_f32(a) * pvdo_f16_to_f32(b));
}

/* ------------------------------------------------------------------ */
/* pcg32 -- raymarching/src/pcg32.h:44-170 (PCG-XSH-RR 64/32, O'Neill)  */
/* ------------------------------------------------------------------ */

#define PCG_MULT 0x5851f42d4c957f2dULL

typedef struct {
    uint64_t state, inc;
} pcg32_t;

static inline uint32_t pcg_next(pcg32_t *g) { /* pcg32.h:66-72 */
    const uint64_t old = g->state;
    g->state = old * PCG_MULT + g->inc;
    const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    const uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}

static inline void pcg_seed(pcg32_t *g, uint64_t initstate, uint64_t initseq) { /* pcg32.h:57-63 */
    g->state = 0;
    g->inc = (initseq << 1) | 1u;
    (void)pcg_next(g);
    g->state += initstate;
    (void)pcg_next(g);
}

static inline void pcg_advance(pcg32_t *g, uint64_t delta) { /* pcg32.h:149-170 */
    uint64_t cur_mult = PCG_MULT, cur_plus = g->inc, acc_mult = 1, acc_plus = 0;
    while (delta) {
        if (delta & 1u) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    g->state = acc_mult * g->state + acc_plus;
}

static inline float pcg_next_float(pcg32_t *g) { /* pcg32.h:107-116 */
    return bits_f32((pcg_next(g) >> 9) | 0x3f800000u) - 1.0f;
}

void pvdo_pcg32_stream(uint64_t seed, uint64_t initseq, int64_t advance,
                       uint32_t count, uint32_t *out_u32, float *out_f32) {
    pcg32_t g;
    pcg_seed(&g, seed, initseq);
    pcg_advance(&g, (uint64_t)advance);
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t u = pcg_next(&g);
        if (out_u32) out_u32[i] = u;
        if (out_f32) out_f32[i] = bits_f32((u >> 9) | 0x3f800000u) - 1.0f;
    }
}

/* ------------------------------------------------------------------ */
/* Morton -- raymarching.cu:58-83                                      */
/* ------------------------------------------------------------------ */

static inline uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
static inline uint32_t gather3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

void pvdo_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) { /* raymarching.cu:216-228 */
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}

void pvdo_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) { /* raymarching.cu:239-256 */
    for (uint32_t n = 0; n < N; n++) {
        /* the reference shifts the *signed* int (arithmetic shift), :251-255 */
        const int32_t ind = indices[n];
        coords[3 * n + 0] = (int32_t)gather3((uint32_t)(ind >> 0));
        coords[3 * n + 1] = (int32_t)gather3((uint32_t)(ind >> 1));
        coords[3 * n + 2] = (int32_t)gather3((uint32_t)(ind >> 2));
    }
}

void pvdo_packbits(const float *grid, uint32_t N, float thresh, uint8_t *bitfield) { /* raymarching.cu:269-291 */
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++)
            if (grid[(size_t)n * 8 + i] > thresh) bits |= (uint8_t)(1u << i);
        bitfield[n] = bits;
    }
}

/* ------------------------------------------------------------------ */
/* near/far, polar -- raymarching.cu:93-147, 164-200                    */
/* ------------------------------------------------------------------ */

void pvdo_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                             uint32_t N, float min_near, float *nears, float *fars) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
        float tn = -FLT_MAX, tf = FLT_MAX; /* running slab interval */
        int miss = 0;
        for (int a = 0; a < 3 && !miss; a++) {
            const float rd = 1.0f / d[a];
            float lo = (aabb[a] - o[a]) * rd;
            float hi = (aabb[a + 3] - o[a]) * rd;
            if (lo > hi) { const float s = lo; lo = hi; hi = s; }
            if (a == 0) {
                tn = lo; tf = hi; /* :115-117 */
            } else {
                if (tn > hi || lo > tf) { miss = 1; break; } /* :123, :135 */
                if (lo > tn) tn = lo;
                if (hi < tf) tf = hi;
            }
        }
        if (miss) {
            nears[n] = fars[n] = FLT_MAX; /* :124, :136 */
        } else {
            if (tn < min_near) tn = min_near; /* :143 */
            nears[n] = tn;
            fars[n] = tf;
        }
    }
}

void pvdo_polar_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
        /* |o + t d| = radius, larger root (:186-190) */
        const float A = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float Bh = o[0] * d[0] + o[1] * d[1] + o[2] * d[2];
        const float Cc = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - radius * radius;
        const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
        const float x = o[0] + t * d[0], y = o[1] + t * d[1], z = o[2] + t * d[2];
        const float theta = atan2f(sqrtf(x * x + z * z), y); /* y is up (:194) */
        const float phi = atan2f(z, x);
        coords[2 * n + 0] = 2 * theta * RPI_F - 1;
        coords[2 * n + 1] = phi * RPI_F;
    }
}

/* ------------------------------------------------------------------ */
/* occupancy-grid DDA shared by the three marchers                      */
/* raymarching.cu:362-403 (count), :430-482 (write), :756-810 (infer)    */
/* ------------------------------------------------------------------ */

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float bound, dt_gamma, dt_min, dt_max, rH, Cf, Hf;
    uint32_t C, H;
    const uint8_t *grid;
} dda_t;

static inline void dda_init(dda_t *r, const float *o, const float *d, float bound, float dt_gamma,
                            uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid) {
    r->ox = o[0]; r->oy = o[1]; r->oz = o[2];
    r->dx = d[0]; r->dy = d[1]; r->dz = d[2];
    r->rdx = 1.0f / d[0]; r->rdy = 1.0f / d[1]; r->rdz = 1.0f / d[2];
    r->bound = bound; r->dt_gamma = dt_gamma;
    r->dt_min = 2 * SQRT3_F / (float)max_steps;                      /* :346 */
    r->dt_max = 2 * SQRT3_F * (float)(1 << (C - 1)) / (float)H;      /* :347 */
    r->rH = 1.0f / (float)H;
    r->Cf = (float)C; r->Hf = (float)H;
    r->C = C; r->H = H; r->grid = grid;
}

static inline int frexp_exponent(float v) { int e; (void)frexpf(v, &e); return e; }

/* One probe at parameter t.  Returns 1 if the cell is occupied (and fills the
 * clamped sample position + dt); otherwise returns 0 and *t_next is the first
 * t (advanced in whole dt steps) at or beyond the cell's exit face. */
static inline int dda_probe(const dda_t *r, float t, float *px, float *py, float *pz, float *pdt, float *t_next) {
    /* canonical: o + t*d is a fused multiply-add (nvcc contracts :364-366) */
    const float x = clampf(fmaf(t, r->dx, r->ox), -r->bound, r->bound);
    const float y = clampf(fmaf(t, r->dy, r->oy), -r->bound, r->bound);
    const float z = clampf(fmaf(t, r->dz, r->oz), -r->bound, r->bound);
    const float dt = clampf(t * r->dt_gamma, r->dt_min, r->dt_max); /* :368 */

    /* mip level (:44-56, :371) */
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    const int lvl_pos = (int)fminf(r->Cf - 1, fmaxf(0.0f, (float)frexp_exponent(mx)));
    const float dmx = (float)((double)(dt * r->Hf) * 0.5); /* double literal, :52 */
    const int lvl_dt = (int)fminf(r->Cf - 1, fmaxf(0.0f, (float)frexp_exponent(dmx)));
    const int level = lvl_pos > lvl_dt ? lvl_pos : lvl_dt;

    const float mip_bound = fminf((float)(1 << level), r->bound);
    const float mip_rbound = 1.0f / mip_bound;

    /* nearest cell: double temporaries then float clamp then trunc (:377-379) */
    const float hi = (float)(r->H - 1);
    /* x * mip_rbound + 1 is a float expression the reference's compilers contract (checked in the ISA of the reference's own kernel built
     * for gfx950, oracle/_ref; nvcc's -fmad=true does the same): a fused multiply-add.  For a power-of-two mip_bound the product is
     * exact and nothing changes; for others (bound 1.5) it decides knife-edge cells (round 6: 2 rays of 2048 differed without it). */
    const int nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)r->H), 0.0f, hi);
    const int ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)r->H), 0.0f, hi);
    const int nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)r->H), 0.0f, hi);

    const uint32_t index = (uint32_t)level * r->H * r->H * r->H + morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const int occ = (r->grid[index >> 3] >> (index & 7u)) & 1;

    *px = x; *py = y; *pz = z; *pdt = dt;
    if (occ) return 1;

    /* distance to the exit face of this cell (:393-397) */
    /* (...) * mip_bound - x: contracted likewise (exact product for a power-of-two mip_bound) */
    const float tx = fmaf(((nx + 0.5f + 0.5f * sign1f(r->dx)) * r->rH * 2 - 1), mip_bound, -x) * r->rdx;
    const float ty = fmaf(((ny + 0.5f + 0.5f * sign1f(r->dy)) * r->rH * 2 - 1), mip_bound, -y) * r->rdy;
    const float tz = fmaf(((nz + 0.5f + 0.5f * sign1f(r->dz)) * r->rH * 2 - 1), mip_bound, -z) * r->rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do { /* always at least one step (:399-401) */
        t += clampf(t * r->dt_gamma, r->dt_min, r->dt_max);
    } while (t < tt);
    *t_next = t;
    return 0;
}

void pvdo_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                           float bound, float dt_gamma, uint32_t max_steps,
                           uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                           const float *nears, const float *fars,
                           float *xyzs, float *dirs, float *deltas,
                           int32_t *rays, int32_t *counter, uint32_t perturb) {
    uint32_t *steps = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
    float *t0s = (float *)malloc(sizeof(float) * (N ? N : 1));

    /* pass 1: count occupied steps per ray (:357-403) */
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        dda_t r;
        dda_init(&r, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t0 = nears[n];
        if (perturb) { /* :351-354, rng = pcg32{42} (:488) */
            pcg32_t g;
            pcg_seed(&g, 42u, 1u);
            pcg_advance(&g, (uint64_t)n);
            t0 = fmaf(r.dt_min, pcg_next_float(&g), t0);  /* `t0 += dt_min * rng.next_float()` is ONE fused operation in the reference's build (v_fmac_f32 in oracle/_ref's ISA; nvcc contracts it too) */
        }
        float t = t0;
        uint32_t num = 0;
        while (t < far && num < max_steps) {
            float x, y, z, dt, tn;
            if (dda_probe(&r, t, &x, &y, &z, &dt, &tn)) { num++; t += dt; }
            else t = tn;
        }
        steps[n] = num;
        t0s[n] = t0;
    }

    /* slot allocation: the reference uses atomicAdd(counter, num_steps) and
     * atomicAdd(counter+1, 1) (:408-409); the in-order execution of those
     * atomics gives exactly this exclusive prefix sum. */
    uint32_t point_base = (uint32_t)counter[0];
    const uint32_t ray_base = (uint32_t)counter[1];
    uint32_t *offs = (uint32_t *)malloc(sizeof(uint32_t) * (N ? N : 1));
    for (uint32_t n = 0; n < N; n++) { offs[n] = point_base; point_base += steps[n]; }
    counter[0] = (int32_t)point_base;
    counter[1] = (int32_t)(ray_base + N);

    /* pass 2: re-march and write (:413-482) */
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t num = steps[n], off = offs[n];
        int32_t *row = rays + 3 * (size_t)n; /* row n == ray n (counter[1] is zeroed by every caller) */
        row[0] = (int32_t)n; row[1] = (int32_t)off; row[2] = (int32_t)num;
        if (num == 0) continue;
        if (off + num >= M) continue; /* strict, :419 */

        dda_t r;
        dda_init(&r, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[n];
        float t = t0s[n], last_t = t;
        float *px = xyzs + 3 * (size_t)off, *pd = dirs + 3 * (size_t)off, *pl = deltas + 2 * (size_t)off;
        uint32_t step = 0;
        while (t < far && step < num) {
            float x, y, z, dt, tn;
            if (dda_probe(&r, t, &x, &y, &z, &dt, &tn)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t; /* includes skipped gaps (:463-465) */
                last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            } else {
                t = tn;
            }
        }
    }
    free(steps); free(t0s); free(offs);
}

/* ------------------------------------------------------------------ */
/* compositing -- raymarching.cu:504-582, 606-686                       */
/* ------------------------------------------------------------------ */

void pvdo_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, uint32_t M, uint32_t N,
                                       float *weights_sum, float *depth, float *image) {
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[3 * (size_t)n];
        const uint32_t offset = (uint32_t)rays[3 * (size_t)n + 1];
        const uint32_t num = (uint32_t)rays[3 * (size_t)n + 2];
        float r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0, T = 1.0f;
        if (!(num == 0 || offset + num >= M)) { /* :525 */
            for (uint32_t s = 0; s < num; s++) {
                const size_t i = (size_t)offset + s;
                const float alpha = 1.0f - expf(-sigmas[i] * deltas[2 * i]);
                const float w = alpha * T;
                r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
                t += deltas[2 * i + 1];
                d += w * t;
                ws += w;
                T *= 1.0f - alpha;
            }
        }
        weights_sum[index] = ws; depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}

void pvdo_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                        const float *sigmas, const float *rgbs, const float *deltas,
                                        const int32_t *rays, const float *weights_sum, const float *image,
                                        uint32_t M, uint32_t N, float *grad_sigmas, float *grad_rgbs) {
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[3 * (size_t)n];
        const uint32_t offset = (uint32_t)rays[3 * (size_t)n + 1];
        const uint32_t num = (uint32_t)rays[3 * (size_t)n + 2];
        if (num == 0 || offset + num >= M) continue; /* :629 */
        const float gws = grad_weights_sum[index];
        const float *gi = grad_image + 3 * (size_t)index;
        const float rF = image[3 * (size_t)index], gF = image[3 * (size_t)index + 1], bF = image[3 * (size_t)index + 2];
        const float wsF = weights_sum[index];
        float r = 0, g = 0, b = 0, ws = 0, T = 1.0f;
        for (uint32_t s = 0; s < num; s++) {
            const size_t i = (size_t)offset + s;
            const float alpha = 1.0f - expf(-sigmas[i] * deltas[2 * i]);
            const float w = alpha * T;
            r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
            ws += w;
            T *= 1.0f - alpha; /* T is the post-update transmittance in :668-673 */
            grad_rgbs[3 * i] = gi[0] * w; grad_rgbs[3 * i + 1] = gi[1] * w; grad_rgbs[3 * i + 2] = gi[2] * w;
            grad_sigmas[i] = deltas[2 * i] * (gi[0] * (T * rgbs[3 * i] - (rF - r)) +
                                              gi[1] * (T * rgbs[3 * i + 1] - (gF - g)) +
                                              gi[2] * (T * rgbs[3 * i + 2] - (bF - b)) +
                                              gws * (T - (wsF - ws)));
        }
    }
}

/* ------------------------------------------------------------------ */
/* inference trio -- raymarching.cu:704-811, 825-909, 921-939           */
/* ------------------------------------------------------------------ */

void pvdo_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                     const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                     uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                     const float *nears, const float *fars,
                     float *xyzs, float *dirs, float *deltas, uint32_t perturb) {
    (void)nears;
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        dda_t r;
        dda_init(&r, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[index];
        float t = rays_t[n];
        if (perturb) { /* rng = pcg32{perturb}, advance(n) (:749-752, :816) */
            pcg32_t g;
            pcg_seed(&g, (uint64_t)perturb, 1u);
            pcg_advance(&g, (uint64_t)n);
            t = fmaf(r.dt_min, pcg_next_float(&g), t);  /* fused, as above (:751) */
        }
        float last_t = t;
        float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step, *pl = deltas + 2 * (size_t)n * n_step;
        uint32_t step = 0;
        while (t < far && step < n_step) {
            float x, y, z, dt, tn;
            if (dda_probe(&r, t, &x, &y, &z, &dt, &tn)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                t += dt;
                pl[0] = dt; pl[1] = t - last_t; last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            } else {
                t = tn;
            }
        }
    }
}

void pvdo_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, float *rays_t,
                         const float *sigmas, const float *rgbs, const float *deltas,
                         float *weights_sum, float *depth, float *image) {
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int32_t index = rays_alive[n];
        float t = rays_t[n];
        float ws = weights_sum[index], d = depth[index];
        float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
        uint32_t step = 0;
        while (step < n_step) {
            const size_t i = (size_t)n * n_step + step;
            if (deltas[2 * i] == 0) break; /* :862 */
            const float alpha = 1.0f - expf(-sigmas[i] * deltas[2 * i]);
            const float T = 1 - ws; /* :872 */
            const float w = alpha * T;
            ws += w;
            t += deltas[2 * i + 1];
            d += w * t;
            r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
            if (T < 1e-4) break; /* double literal compare, after accumulating (:886) */
            step++;
        }
        rays_t[n] = (step < n_step) ? -1.0f : t; /* :898-902 */
        weights_sum[index] = ws; depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}

void pvdo_compact_rays(uint32_t n_alive, int32_t *rays_alive, const int32_t *rays_alive_old,
                       float *rays_t, const float *rays_t_old, int32_t *alive_counter) {
    int32_t k = alive_counter[0];
    for (uint32_t n = 0; n < n_alive; n++) {
        if (rays_t_old[n] >= 0) { /* :934 */
            rays_alive[k] = rays_alive_old[n];
            rays_t[k] = rays_t_old[n];
            k++;
        }
    }
    alive_counter[0] = k;
}

/* ------------------------------------------------------------------ */
/* grid encoder -- gridencoder.cu:35-343                                */
/* ------------------------------------------------------------------ */

#define PVDO_MAX_LEVELS 32

void pvdo_grid_level_params(uint32_t L, float S, uint32_t H, float *scales, uint32_t *resolutions) {
    for (uint32_t l = 0; l < L; l++) {
        /* gridencoder.cu:126-127; host libm exp2f is the canonical value, the
         * product computes the same table on the host and passes it down. */
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;
        scales[l] = scale;
        resolutions[l] = (uint32_t)ceil((double)scale) + 1u;
    }
}

static inline uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners,
                                  uint32_t hashmap_size, uint32_t resolution, const uint32_t *pg) {
    /* gridencoder.cu:54-72 with ch = 0 */
    static const uint32_t primes[3] = {1u, 2654435761u, 805459861u}; /* :42 */
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pg[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        index = 0;
        for (uint32_t d = 0; d < D; d++) index ^= pg[d] * primes[d];
    }
    return (index % hashmap_size) * C;
}

static inline int grid_locate(const float *in, uint32_t D, float scale, int align_corners, float *frac, uint32_t *cell) {
    for (uint32_t d = 0; d < D; d++)
        if (in[d] < 0 || in[d] > 1) return 0; /* :99-105 */
    for (uint32_t d = 0; d < D; d++) {
        /* canonical: x*scale + 0.5 fused (nvcc contracts :135) */
        const float p = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
        const float fl = floorf(p);
        cell[d] = (uint32_t)fl;
        frac[d] = p - (float)cell[d];
    }
    return 1;
}

static int grid_shape_ok(uint32_t D, uint32_t C, uint32_t L) {
    if (!(D == 2 || D == 3)) return 0;                  /* gridencoder.cu:367-371 */
    if (!(C == 1 || C == 2 || C == 4 || C == 8)) return 0; /* :350-356 */
    if (L == 0 || L > PVDO_MAX_LEVELS) return 0;
    return 1;
}

int pvdo_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                             void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             float S, uint32_t H, int calc_grad_inputs, void *dy_dx,
                             uint32_t gridtype, int align_corners, int dtype) {
    if (!grid_shape_ok(D, C, L)) return -1;
    float scales[PVDO_MAX_LEVELS];
    uint32_t ress[PVDO_MAX_LEVELS];
    pvdo_grid_level_params(L, S, H, scales, ress);
    const uint32_t ncorner = 1u << D;

#pragma omp parallel for collapse(2) schedule(static)
    for (uint32_t level = 0; level < L; level++) {
        for (uint32_t b = 0; b < B; b++) {
            const size_t gbase = (size_t)(uint32_t)offsets[level] * C;
            const uint32_t hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
            const float scale = scales[level];
            const uint32_t res = ress[level];
            const float *in = inputs + (size_t)b * D;
            const size_t obase = ((size_t)level * B + b) * C;            /* [L,B,C], :96 */
            const size_t dbase = (size_t)b * D * L * C + (size_t)level * D * C; /* [B,L,D,C], :113 */
            float frac[3];
            uint32_t cell[3];

            if (!grid_locate(in, D, scale, align_corners, frac, cell)) {
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (dtype == 0) ((float *)outputs)[obase + ch] = 0.0f;
                    else ((uint16_t *)outputs)[obase + ch] = 0;
                }
                if (calc_grad_inputs)
                    for (uint32_t k = 0; k < D * C; k++) {
                        if (dtype == 0) ((float *)dy_dx)[dbase + k] = 0.0f;
                        else ((uint16_t *)dy_dx)[dbase + k] = 0;
                    }
                continue;
            }

            float accf[8] = {0};
            uint16_t acch[8] = {0};
            for (uint32_t idx = 0; idx < ncorner; idx++) { /* :146-170 */
                float w = 1;
                uint32_t pg[3];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx >> d) & 1u) { w *= frac[d]; pg[d] = cell[d] + 1; }
                    else { w *= 1 - frac[d]; pg[d] = cell[d]; }
                }
                const size_t gi = gbase + grid_index(D, C, gridtype, align_corners, hsize, res, pg);
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (dtype == 0) {
                        accf[ch] = fmaf(w, ((const float *)embeddings)[gi + ch], accf[ch]);
                    } else {
                        /* scalar_t = at::Half: product rounded to half, then half add (:166) */
                        const uint16_t p = pvdo_f32_to_f16(w * pvdo_f16_to_f32(((const uint16_t *)embeddings)[gi + ch]));
                        acch[ch] = h_add(acch[ch], p);
                    }
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) {
                if (dtype == 0) ((float *)outputs)[obase + ch] = accf[ch];
                else ((uint16_t *)outputs)[obase + ch] = acch[ch];
            }

            if (calc_grad_inputs) { /* :180-223 */
                for (uint32_t gd = 0; gd < D; gd++) {
                    float gaf[8] = {0};
                    uint16_t gah[8] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pg[3];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                            if ((idx >> nd) & 1u) { w *= frac[d]; pg[d] = cell[d] + 1; }
                            else { w *= 1 - frac[d]; pg[d] = cell[d]; }
                        }
                        pg[gd] = cell[gd];
                        const size_t il = gbase + grid_index(D, C, gridtype, align_corners, hsize, res, pg);
                        pg[gd] = cell[gd] + 1;
                        const size_t ir = gbase + grid_index(D, C, gridtype, align_corners, hsize, res, pg);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            if (dtype == 0) {
                                const float *g = (const float *)embeddings;
                                gaf[ch] = fmaf(w, g[ir + ch] - g[il + ch], gaf[ch]);
                            } else {
                                const uint16_t *g = (const uint16_t *)embeddings;
                                const uint16_t diff = h_sub(g[ir + ch], g[il + ch]);
                                const uint16_t p = pvdo_f32_to_f16(w * pvdo_f16_to_f32(diff));
                                gah[ch] = h_add(gah[ch], p);
                            }
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) {
                        if (dtype == 0) ((float *)dy_dx)[dbase + gd * C + ch] = gaf[ch];
                        else ((uint16_t *)dy_dx)[dbase + gd * C + ch] = gah[ch];
                    }
                }
            }
        }
    }
    return 0;
}

int pvdo_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                              const int32_t *offsets, void *grad_embeddings,
                              uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                              int calc_grad_inputs, const void *dy_dx, void *grad_inputs,
                              uint32_t gridtype, int align_corners, int dtype) {
    (void)embeddings;
    if (!grid_shape_ok(D, C, L)) return -1;
    float scales[PVDO_MAX_LEVELS];
    uint32_t ress[PVDO_MAX_LEVELS];
    pvdo_grid_level_params(L, S, H, scales, ress);
    const uint32_t ncorner = 1u << D;

    /* scatter-add, gridencoder.cu:227-314.  Levels own disjoint table regions,
     * so parallelising over levels keeps the summation order (ascending b)
     * deterministic -- one legal ordering of the reference's atomics. */
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t level = 0; level < L; level++) {
        const size_t gbase = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hsize = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = scales[level];
        const uint32_t res = ress[level];
        for (uint32_t b = 0; b < B; b++) {
            float frac[3];
            uint32_t cell[3];
            if (!grid_locate(inputs + (size_t)b * D, D, scale, align_corners, frac, cell)) continue; /* :254-259 */
            const size_t qbase = ((size_t)level * B + b) * C; /* grad is [L,B,C], :247 */
            for (uint32_t idx = 0; idx < ncorner; idx++) {
                float w = 1;
                uint32_t pg[3];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx >> d) & 1u) { w *= frac[d]; pg[d] = cell[d] + 1; }
                    else { w *= 1 - frac[d]; pg[d] = cell[d]; }
                }
                const size_t gi = gbase + grid_index(D, C, gridtype, align_corners, hsize, res, pg);
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (dtype == 0) {
                        ((float *)grad_embeddings)[gi + ch] += w * ((const float *)grad)[qbase + ch]; /* :310 */
                    } else {
                        /* __half2 atomicAdd of (half)(w*g) (:303-304); C==1 in half is
                         * a no-op stub in the reference (:22-26) -- the oracle
                         * accumulates it properly instead, see DESIGN.md. */
                        uint16_t *gg = (uint16_t *)grad_embeddings;
                        const uint16_t v = pvdo_f32_to_f16(w * pvdo_f16_to_f32(((const uint16_t *)grad)[qbase + ch]));
                        gg[gi + ch] = h_add(gg[gi + ch], v);
                    }
                }
            }
        }
    }

    if (calc_grad_inputs) { /* kernel_input_backward, :317-343 */
#pragma omp parallel for schedule(static)
        for (uint32_t t = 0; t < B * D; t++) {
            const uint32_t b = t / D, d = t - b * D;
            if (dtype == 0) {
                const float *g = (const float *)grad, *dd = (const float *)dy_dx + (size_t)b * L * D * C;
                float r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t ch = 0; ch < C; ch++)
                        r = fmaf(g[((size_t)l * B + b) * C + ch], dd[(size_t)l * D * C + d * C + ch], r);
                ((float *)grad_inputs)[t] = r;
            } else {
                const uint16_t *g = (const uint16_t *)grad, *dd = (const uint16_t *)dy_dx + (size_t)b * L * D * C;
                uint16_t r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t ch = 0; ch < C; ch++)
                        r = h_add(r, h_mul(g[((size_t)l * B + b) * C + ch], dd[(size_t)l * D * C + d * C + ch]));
                ((uint16_t *)grad_inputs)[t] = r;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* SH encoder -- shencoder.cu:27-383                                    */
/* ------------------------------------------------------------------ */
/*
 * The reference hard-codes, for band l and order m, the polynomial
 *     Y_lm(x,y,z) = K_lm * Q_l^|m|(z) * { Re (x+iy)^m      (m > 0)
 *                                        { 1               (m = 0)
 *                                        { Im (x+iy)^|m|   (m < 0)
 * with Q_l^m = d^m/dz^m P_l (Legendre), K_lm = (-1)^m sqrt(2)^(m!=0)
 * sqrt((2l+1)/(4 pi) (l-|m|)!/(l+|m|)!), stored at index l*l + l + m
 * (shencoder.cu:51-125; e.g. :56 l=2,m=0 is 0.946 z^2 - 0.315, i.e. r = 1 is
 * assumed, so the value at non-unit inputs is this polynomial's, which is
 * what padding rows d = 0 see).  The oracle builds the same polynomials from
 * the recurrences in double precision and rounds once.
 */

#define SH_MAXDEG 8

typedef struct {
    double q[SH_MAXDEG][SH_MAXDEG][SH_MAXDEG]; /* q[l][m][k]: coeff of z^k in K_lm * Q_l^m(z) */
    int ready;
} sh_table_t;

static sh_table_t g_sh;

static void sh_build_table(void) {
    double P[SH_MAXDEG][SH_MAXDEG];
    memset(P, 0, sizeof(P));
    P[0][0] = 1.0;
    if (SH_MAXDEG > 1) P[1][1] = 1.0;
    for (int n = 1; n + 1 < SH_MAXDEG; n++) /* (n+1) P_{n+1} = (2n+1) z P_n - n P_{n-1} */
        for (int k = 0; k < SH_MAXDEG; k++) {
            const double a = (k > 0) ? (2 * n + 1) * P[n][k - 1] : 0.0;
            P[n + 1][k] = (a - n * P[n - 1][k]) / (n + 1);
        }
    const double pi = 3.14159265358979323846;
    for (int l = 0; l < SH_MAXDEG; l++) {
        double d[SH_MAXDEG];
        memcpy(d, P[l], sizeof(d));
        for (int m = 0; m <= l; m++) {
            if (m > 0) { /* differentiate once more */
                for (int k = 0; k + 1 < SH_MAXDEG; k++) d[k] = (k + 1) * d[k + 1];
                d[SH_MAXDEG - 1] = 0;
            }
            double ratio = 1.0; /* (l-m)!/(l+m)! */
            for (int i = l - m + 1; i <= l + m; i++) ratio /= i;
            double K = sqrt((2 * l + 1) / (4 * pi) * ratio);
            if (m > 0) K *= sqrt(2.0) * ((m & 1) ? -1.0 : 1.0);
            for (int k = 0; k < SH_MAXDEG; k++) g_sh.q[l][m][k] = K * d[k];
        }
    }
    g_sh.ready = 1;
}

static inline double poly_eval(const double *c, double z) {
    double r = 0;
    for (int k = SH_MAXDEG - 1; k >= 0; k--) r = r * z + c[k];
    return r;
}
static inline double poly_deriv(const double *c, double z) {
    double r = 0;
    for (int k = SH_MAXDEG - 1; k >= 1; k--) r = r * z + k * c[k];
    return r;
}

int pvdo_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t C,
                           int calc_grad_inputs, float *dy_dx) {
    if (D != 3 || C < 1 || C > SH_MAXDEG) return -1;
    if (!g_sh.ready) sh_build_table();
    const uint32_t C2 = C * C;
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        const double x = inputs[3 * (size_t)b], y = inputs[3 * (size_t)b + 1], z = inputs[3 * (size_t)b + 2];
        double A[SH_MAXDEG], Bm[SH_MAXDEG]; /* Re/Im (x+iy)^m */
        A[0] = 1; Bm[0] = 0;
        for (int m = 1; m < (int)C; m++) {
            A[m] = x * A[m - 1] - y * Bm[m - 1];
            Bm[m] = x * Bm[m - 1] + y * A[m - 1];
        }
        float *out = outputs + (size_t)b * C2;
        float *gx = calc_grad_inputs ? dy_dx + (size_t)b * 3 * C2 : 0; /* [B,3,C2], shencoder.cu:128-131 */
        float *gy = gx ? gx + C2 : 0, *gz = gy ? gy + C2 : 0;
        for (int l = 0; l < (int)C; l++) {
            for (int m = 0; m <= l; m++) {
                const double q = poly_eval(g_sh.q[l][m], z);
                const double dq = poly_deriv(g_sh.q[l][m], z);
                const int ip = l * l + l + m, in = l * l + l - m;
                if (m == 0) {
                    out[ip] = (float)q;
                    if (gx) { gx[ip] = 0; gy[ip] = 0; gz[ip] = (float)dq; }
                } else {
                    out[ip] = (float)(q * A[m]);
                    out[in] = (float)(q * Bm[m]);
                    if (gx) {
                        /* d/dx (x+iy)^m = m (x+iy)^(m-1);  d/dy = i m (x+iy)^(m-1) */
                        gx[ip] = (float)(q * m * A[m - 1]);  gy[ip] = (float)(-q * m * Bm[m - 1]); gz[ip] = (float)(dq * A[m]);
                        gx[in] = (float)(q * m * Bm[m - 1]); gy[in] = (float)(q * m * A[m - 1]);   gz[in] = (float)(dq * Bm[m]);
                    }
                }
            }
        }
    }
    return 0;
}

int pvdo_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t C,
                            const float *dy_dx, float *grad_inputs) {
    (void)inputs;
    if (D != 3 || C < 1 || C > SH_MAXDEG) return -1;
    const uint32_t C2 = C * C;
#pragma omp parallel for schedule(static)
    for (uint32_t t = 0; t < B * D; t++) { /* shencoder.cu:359-383: accumulates with += */
        const uint32_t b = t / D, d = t - b * D;
        const float *g = grad + (size_t)b * C2, *dd = dy_dx + (size_t)b * D * C2 + (size_t)d * C2;
        float r = grad_inputs[t];
        for (uint32_t ch = 0; ch < C2; ch++) r += g[ch] * dd[ch];
        grad_inputs[t] = r;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* The sigma / colour head under torch.autocast(float16) -- distill_mutual/network.py:413-437 (hash / mlp: sigma_net,
 * clamp, trunc_exp, color_net, sigmoid) and :344-381 (vm: basis_mat inside get_color_feat :290-309, the clamps, the
 * same colour head).  What autocast does to that code on the device the reference trains on:
 *   - every nn.Linear (all of them bias-free, network.py:111-152) takes its input and its fp32 master weight rounded to
 *     f16, accumulates in fp32 and stores f16; F.relu and torch.clamp act on those f16 values;
 *   - trunc_exp (tools/activation.py: custom_fwd(cast_inputs=torch.float32)) and the SH direction encoding
 *     (shencoder/sphere_harmonics.py: custom_fwd(cast_inputs=torch.float32)) run in fp32;
 *   - torch.cat([enc_d (f32), geo_feat (f16)]) promotes to f32 and the next Linear rounds it to f16 again;
 *   - torch.sigmoid on the last Linear's f16 output: evaluated in fp32, stored f16;
 *   - vm: sigma_feat comes out of grid_sample in fp32 (an autocast-to-fp32 op) and is clamped in fp32; basis_mat's
 *     input (the plane x line products) is rounded to f16 by the Linear.
 * This is the arithmetic of the timed configuration's fused MFMA head (fusedhead.hip: head_forward_tile), restated
 * with a plain ascending-k fp32 sum per output -- the matrix cores add the same exact products in another order, so
 * the two agree up to the rounding of an fp32 sum, i.e. up to one f16 ulp on the rare entry that lands on a tie.
 *   kind 0 (hash / mlp trunk): x0 = encoder output [M][28] f16; Wa1 = sigma_net.0 [64][28], Wa2 = sigma_net.1 [16][64]
 *   kind 1 (vm)              : x0 = products [M][144] f16, sigma_raw [M] f32; Wa1 = basis_mat [15][144]
 *   Wc1 [64][31], Wc2 [64][64], Wc3 [3][64]; dirs [M][3] unit vectors.
 * Outputs: sigma [M] f32 = exp(feature 0), rgb [M][3] (f16 values widened), feat16 [M][16] = feature_sigma_color. */
static inline float h_round(float v) { return pvdo_f16_to_f32(pvdo_f32_to_f16(v)); }

static void linear_f16(const float *W, int rows, int cols, const float *x_h /* f16 values */, float *y_h, int relu) {
    for (int o = 0; o < rows; o++) {
        float acc = 0.0f;
        for (int k = 0; k < cols; k++) acc += h_round(W[(size_t)o * cols + k]) * x_h[k]; /* exact products, fp32 sum */
        float r = h_round(acc);
        if (relu && !(r > 0.0f)) r = 0.0f;
        y_h[o] = r;
    }
}

int pvdo_head_forward_amp(int kind, const uint16_t *x0, const float *sigma_raw, const float *dirs, uint32_t M,
                          const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3,
                          float clip_sigma_min, float clip_feat_min, float clip_max,
                          float *sigma, float *rgb, float *feat16) {
    if (kind != 0 && kind != 1) return -1;
    float *sh = (float *)malloc((size_t)M * 16 * sizeof(float));
    if (!sh) return -2;
    if (pvdo_sh_encode_forward(dirs, sh, M, 3, 4, 0, 0) != 0) { free(sh); return -1; } /* network.py:104 (degree 4), fp32 */
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < M; b++) {
        float F[16]; /* feature_sigma_color */
        if (kind == 0) {
            float x[28], h1[64], h[16];
            for (int k = 0; k < 28; k++) x[k] = pvdo_f16_to_f32(x0[(size_t)b * 28 + k]);
            linear_f16(Wa1, 64, 28, x, h1, 1);  /* network.py:414-417 */
            linear_f16(Wa2, 16, 64, h1, h, 0);
            h[0] = h_round(fminf(clip_max, fmaxf(clip_sigma_min, h[0]))); /* :418-420, on the f16 tensor */
            for (int k = 0; k < 16; k++) F[k] = h[k];
        } else {
            float x[144], cf[15];
            for (int k = 0; k < 144; k++) x[k] = pvdo_f16_to_f32(x0[(size_t)b * 144 + k]);
            linear_f16(Wa1, 15, 144, x, cf, 0); /* basis_mat, network.py:308 */
            F[0] = fminf(clip_max, fmaxf(clip_sigma_min, sigma_raw[b]));             /* :357-360 (fp32) */
            for (int k = 0; k < 15; k++) F[1 + k] = fminf(clip_max, fmaxf(clip_feat_min, cf[k])); /* :361-363 */
        }
        for (int k = 0; k < 16; k++) feat16[(size_t)b * 16 + k] = F[k];
        sigma[b] = expf(F[0]); /* trunc_exp forward, fp32 (:425 / :372) */
        float in[31], c1[64], c2[64], c3[3];
        for (int k = 0; k < 16; k++) in[k] = h_round(sh[(size_t)b * 16 + k]);  /* torch.cat -> f32 -> the Linear's f16 cast */
        for (int k = 0; k < 15; k++) in[16 + k] = h_round(F[1 + k]);
        linear_f16(Wc1, 64, 31, in, c1, 1); /* :429-434 */
        linear_f16(Wc2, 64, 64, c1, c2, 1);
        linear_f16(Wc3, 3, 64, c2, c3, 0);
        for (int k = 0; k < 3; k++) rgb[(size_t)b * 3 + k] = h_round(1.0f / (1.0f + expf(-c3[k]))); /* :435 */
    }
    free(sh);
    return 0;
}

/* ------------------------------------------------------------------ */
/* The TensoRF "VM" plane x line lookup -- distill_mutual/network.py:216-309 (get_sigma_feat / get_color_feat over the tables of
 * init_one_vm :193-214) with x normalised as :345-350.  F.grid_sample(align_corners=True, zero padding, bilinear) restated as
 * PyTorch evaluates it: ix = ((x + 1) / 2) (W - 1); the four taps nw, ne, sw, se with weights (ix_se - ix)(iy_se - iy), ... ,
 * out = nw_val * nw + ne_val * ne + sw_val * sw + se_val * se accumulated in that order, a tap outside the image contributing
 * nothing.  The line factor is a [L, 1] image sampled at x = 0: two taps, weights 1 * w0 and 1 * w1.
 * Tables in the REFERENCE's layout: mat_i [R][H_i][W_i] with W_i = res[mat_ids[i][0]], H_i = res[mat_ids[i][1]]; vec_i [R][L_i].
 * color_prod [M][144] = (plane x line) per channel: no sum is formed, so it is the value the HIP lookup must produce bit for bit.
 * sigma_feat [M] = sum over the 3 x 16 products: torch.sum's association is an implementation detail (and the HIP kernel's is a
 * butterfly over lanes), so the oracle forms it in double and rounds once; compare within fp32 summation noise. */
static inline float bilinear_tap(const float *img, int W, int H, int x, int y) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}
static float grid_sample_2d(const float *img, int W, int H, float gx, float gy) {
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    float out = 0.0f;
    int any = 0;
    /* (a tap outside contributes nothing; the first tap inside starts the sum -- 0 + v*w == v*w exactly either way) */
    const int inx0 = x0 >= 0 && x0 < W, inx1 = x0 + 1 >= 0 && x0 + 1 < W, iny0 = y0 >= 0 && y0 < H, iny1 = y0 + 1 >= 0 && y0 + 1 < H;
    if (inx0 && iny0) { out = bilinear_tap(img, W, H, x0, y0) * (wx0 * wy0); any = 1; }
    if (inx1 && iny0) { const float t = bilinear_tap(img, W, H, x0 + 1, y0) * (wx1 * wy0); out = any ? out + t : t; any = 1; }
    if (inx0 && iny1) { const float t = bilinear_tap(img, W, H, x0, y0 + 1) * (wx0 * wy1); out = any ? out + t : t; any = 1; }
    if (inx1 && iny1) { const float t = bilinear_tap(img, W, H, x0 + 1, y0 + 1) * (wx1 * wy1); out = any ? out + t : t; any = 1; }
    return out;
}

int pvdo_vm_forward(const float *xyz, uint32_t M, const float *aabb, const float *const *tables /* [12]: sigma_mat[3], sigma_vec[3],
                    color_mat[3], color_vec[3] */, const uint32_t *res, float *sigma_feat, float *color_prod) {
    static const int mat_ids[3][2] = {{0, 1}, {0, 2}, {1, 2}}, vec_ids[3] = {2, 1, 0};
    const int R[2] = {16, 48};
#pragma omp parallel for schedule(static)
    for (uint32_t m = 0; m < M; m++) {
        float xn[3];
        for (int a = 0; a < 3; a++) xn[a] = (2.0f * (xyz[3 * (size_t)m + a] - aabb[a])) / (aabb[a + 3] - aabb[a]) - 1.0f; /* :345-350 */
        double sig = 0.0;
        for (int k = 0; k < 2; k++) {
            for (int i = 0; i < 3; i++) {
                const int W = (int)res[mat_ids[i][0]], H = (int)res[mat_ids[i][1]], L = (int)res[vec_ids[i]];
                const float *mat = tables[6 * k + i], *vec = tables[6 * k + 3 + i];
                for (int r = 0; r < R[k]; r++) {
                    const float pv = grid_sample_2d(mat + (size_t)r * W * H, W, H, xn[mat_ids[i][0]], xn[mat_ids[i][1]]);
                    const float lv = grid_sample_2d(vec + (size_t)r * L, 1, L, 0.0f, xn[vec_ids[i]]); /* [L, 1] image at x = 0 */
                    const float prod = pv * lv;
                    if (k == 0) sig += (double)prod;
                    else color_prod[(size_t)m * 144 + i * 48 + r] = prod;
                }
            }
        }
        sigma_feat[m] = (float)sig;
    }
    return 0;
}
