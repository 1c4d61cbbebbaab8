// raymarching.hip -- occupancy-grid ray marching and compositing for gfx950 (MI355X).
//
// Replaces the reference's _raymarching module (raymarching/src/raymarching.cu); every
// entry point cites the reference function it stands in for.  Written for CDNA4:
// 64-wide wavefronts, 256-thread workgroups, ballot/scan compaction instead of
// per-ray global atomics, and a deterministic prefix-sum slot allocator.
#include "dda.h"

#include <float.h>

namespace pvd {

constexpr uint32_t kBlock = 256;
constexpr float kRPi = 0.3183098861837907f;

// ------------------------------------------------------------------ utils

// reference: kernel_near_far_from_aabb, raymarching.cu:93-147
__device__ __forceinline__ void near_far_of(const float o[3], const float d[3], const float *__restrict__ aabb, float min_near, float &near,
                                            float &far) {
    float tn = 0.f, tf = 0.f;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float rd = 1.0f / d[a];
        float lo = (aabb[a] - o[a]) * rd;
        float hi = (aabb[a + 3] - o[a]) * rd;
        if (lo > hi) { const float s = lo; lo = hi; hi = s; }
        if (a == 0) {
            tn = lo; tf = hi;
        } else if (!miss) {
            if (tn > hi || lo > tf) miss = true;
            else {
                if (lo > tn) tn = lo;
                if (hi < tf) tf = hi;
            }
        }
    }
    if (miss) {
        near = FLT_MAX; far = FLT_MAX;
    } else {
        near = tn < min_near ? min_near : tn;
        far = tf;
    }
}

__global__ void __launch_bounds__(kBlock) k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                     const float *__restrict__ aabb, uint32_t N, float min_near,
                                                     float *__restrict__ nears, float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float o[3] = {rays_o[3 * (size_t)n], rays_o[3 * (size_t)n + 1], rays_o[3 * (size_t)n + 2]};
    const float d[3] = {rays_d[3 * (size_t)n], rays_d[3 * (size_t)n + 1], rays_d[3 * (size_t)n + 2]};
    near_far_of(o, d, aabb, min_near, nears[n], fars[n]);
}

// reference: kernel_polar_from_ray, raymarching.cu:164-200
__global__ void __launch_bounds__(kBlock) k_polar(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                  float radius, uint32_t N, float *__restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[3 * (size_t)n], oy = rays_o[3 * (size_t)n + 1], oz = rays_o[3 * (size_t)n + 2];
    const float dx = rays_d[3 * (size_t)n], dy = rays_d[3 * (size_t)n + 1], dz = rays_d[3 * (size_t)n + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float Bh = ox * dx + oy * dy + oz * dz;
    const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[2 * (size_t)n] = 2 * theta * kRPi - 1;
    coords[2 * (size_t)n + 1] = phi * kRPi;
}


// reference: get_rays, distill_mutual/utils.py:324-404 (pixel-centre directions through K^-1, normalise,
// rotate by the camera-to-world pose).  One thread per ray instead of ~20 elementwise launches.
__device__ __forceinline__ void ray_of_pixel(const float *__restrict__ pose, float fx, float fy, float cx, float cy, int64_t k, uint32_t W,
                                             float o[3], float d[3]) {
    const float i = (float)(k % W) + 0.5f, j = (float)(k / W) + 0.5f;
    const float x = (i - cx) / fx, y = (j - cy) / fy, z = 1.0f;
    const float inv = 1.0f / sqrtf(x * x + y * y + z * z);
    const float dx = x * inv, dy = y * inv, dz = z * inv;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        d[r] = dx * pose[4 * r] + dy * pose[4 * r + 1] + dz * pose[4 * r + 2];
        o[r] = pose[4 * r + 3];
    }
}

__global__ void __launch_bounds__(kBlock) k_get_rays(const float *__restrict__ pose, float fx, float fy, float cx, float cy,
                                                     const int64_t *__restrict__ inds, uint32_t W, uint32_t N,
                                                     float *__restrict__ rays_o, float *__restrict__ rays_d) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    float o[3], d[3];
    ray_of_pixel(pose, fx, fy, cx, cy, inds ? inds[n] : (int64_t)n, W, o, d);
#pragma unroll
    for (int r = 0; r < 3; r++) { rays_o[3 * (size_t)n + r] = o[r]; rays_d[3 * (size_t)n + r] = d[r]; }
}

// One training batch in one launch (the reference's data side: Trainer.train_one_epoch -> get_rays with
// N random pixels of one pose and a random background colour per ray, utils.py:354, 987-995; then run_cuda's
// near_far_from_aabb): pose = poses[state[0]], pixel ids and background from a PCG32 stream keyed by (seed, batch
// counter, ray), rays, near/far.  The last workgroup to finish advances state = {pose index, batch counter, done}.
__global__ void __launch_bounds__(kBlock) k_make_ray_batch(const float *__restrict__ poses, uint32_t P, long long *__restrict__ state,
                                                           uint64_t seed, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W,
                                                           uint32_t N, const float *__restrict__ aabb, float min_near,
                                                           int64_t *__restrict__ inds, float *__restrict__ rays_o,
                                                           float *__restrict__ rays_d, float *__restrict__ bg, float *__restrict__ nears,
                                                           float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const long long pose_idx = state[0], batch = state[1];
    if (n < N) {
        Pcg32 g;
        g.seed(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(batch + 1));
        g.advance(4ull * n);
        const int64_t k = (int64_t)(((uint64_t)g.next() * (uint64_t)(H * W)) >> 32);  // uniform in [0, H*W)
        float o[3], d[3];
        ray_of_pixel(poses + 16 * (size_t)pose_idx, fx, fy, cx, cy, k, W, o, d);
        if (inds) inds[n] = k;
#pragma unroll
        for (int r = 0; r < 3; r++) { rays_o[3 * (size_t)n + r] = o[r]; rays_d[3 * (size_t)n + r] = d[r]; }
        if (bg) {
#pragma unroll
            for (int r = 0; r < 3; r++) bg[3 * (size_t)n + r] = g.next_float();
        }
        near_far_of(o, d, aabb, min_near, nears[n], fars[n]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long *>(state + 2), 1ull);
        if (done == gridDim.x - 1) {  // every workgroup has read the state
            state[0] = (pose_idx + 1) % (long long)P;
            state[1] = batch + 1;
            state[2] = 0;
        }
    }
}

// reference: kernel_morton3D / kernel_morton3D_invert, raymarching.cu:216-256
__global__ void __launch_bounds__(kBlock) k_morton3D(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1], (uint32_t)coords[3 * (size_t)n + 2]);
}

__global__ void __launch_bounds__(kBlock) k_morton3D_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];  // signed shifts, as the reference does
    coords[3 * (size_t)n] = (int32_t)gather3((uint32_t)(ind >> 0));
    coords[3 * (size_t)n + 1] = (int32_t)gather3((uint32_t)(ind >> 1));
    coords[3 * (size_t)n + 2] = (int32_t)gather3((uint32_t)(ind >> 2));
}

// reference: kernel_packbits, raymarching.cu:269-291.  One thread per output byte; the
// eight cells are fetched as two 16-byte loads (the grid is 32-byte aligned per byte).
__global__ void __launch_bounds__(kBlock) k_packbits(const float *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4 *>(grid)[2 * (size_t)n];
    const float4 b = reinterpret_cast<const float4 *>(grid)[2 * (size_t)n + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;   bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;   bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;  bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;  bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// ------------------------------------------------------------------ march_rays_train
// Three launches: count -> scan -> write.  The `rays` table itself is the scratch between
// them (column 2 = count after pass 1, column 1 = offset after the scan), so the ABI needs
// no workspace.  reference: kernel_march_rays_train, raymarching.cu:313-483.

// Single-workgroup exclusive scan of the per-ray counts (N is 4096 in training; a
// 640k-ray full image is 625 trips of a 1024-wide scan).  Also bumps the caller's counter
// the way the reference's two atomics do (:408-409).
constexpr uint32_t kScanBlock = 1024;
__global__ void __launch_bounds__(kScanBlock) k_march_scan(int32_t *__restrict__ rays, uint32_t N, int32_t *__restrict__ counter) {
    __shared__ uint32_t wave_sums[kScanBlock / kWave];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (tid == 0) carry_s = (uint32_t)counter[0];
    __syncthreads();
    for (uint32_t base = 0; base < N; base += kScanBlock) {
        const uint32_t n = base + tid;
        const uint32_t v = (n < N) ? (uint32_t)rays[3 * (size_t)n + 2] : 0u;
        // inclusive wave scan
        uint32_t s = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t up = __shfl_up(s, off, kWave);
            if ((int)lane >= off) s += up;
        }
        if (lane == 63) wave_sums[wid] = s;
        __syncthreads();
        uint32_t wave_prefix = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kScanBlock / kWave; w++) {
            const uint32_t ws = wave_sums[w];
            if (w < wid) wave_prefix += ws;
            total += ws;
        }
        const uint32_t carry = carry_s;
        if (n < N) {
            rays[3 * (size_t)n] = (int32_t)n;
            rays[3 * (size_t)n + 1] = (int32_t)(carry + wave_prefix + s - v);
        }
        __syncthreads();
        if (tid == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (tid == 0) {
        counter[0] = (int32_t)carry_s;
        counter[1] += (int32_t)N;
    }
}

// The sample budget of a launch: M rows are allocated and (re)initialised; rays are dropped against the LOGICAL budget, which a
// caller that replays a captured step may keep in device memory (the running mean of the sample count, refreshed by
// update_extra_state without re-capturing): min(M, *budget).  NULL: the budget is M, as in the reference.
__device__ __forceinline__ uint32_t logical_budget(uint32_t M, const int32_t *__restrict__ budget) {
    return budget ? min(M, (uint32_t)max(*budget, 0)) : M;
}

// ------------------------------------------------------------------ wave-per-ray marcher (dt_gamma == 0)
// With dt_gamma == 0 every advance of t -- an occupied step or one trip of the skip loop -- adds the
// same constant dt, so all t a ray ever visits lie on ONE lattice t_{k+1} = fl(t_k + dt), and the
// reference's loop only selects which lattice points are probed.  A wavefront owns a ray: its 64 lanes
// probe 64 consecutive lattice points at once (position, mip level, Morton index, bit test, exit-face
// distance), then the sequential control flow is replayed on the ballots: runs of occupied lanes are
// emitted wholesale, an empty lane jumps to the first lane whose t is not below its exit distance.
// Output ranks come from popcounts of the emit mask -- no atomics, bit-identical samples.

// Exact k-fold application of t <- fl(t + dt) for t >= 0, dt > 0: inside one binade the rounded
// increment is constant after the first step (a tie, when dt's remainder is exactly half an ulp, is
// resolved by the parity of the previous sum and lands on an even mantissa, after which parity repeats),
// so the walk is integer arithmetic on the bit pattern; the first TWO steps after entering a binade (the
// entering one rounds at the new ulp, the next one may be the parity-dependent tie) and every crossing
// are done with the hardware adder.  Checked against serial accumulation on 2e4 random (t, dt, n) incl.
// forced ties and crossings (tests/test_oracle_pins.py::test_lattice_advance_model).
__device__ __forceinline__ float lattice_advance(float t, float dt, uint32_t n) {
    while (n > 0) {
        const float t1 = t + dt;  // may enter a new binade (generic rounding): hardware adder
        n--;
        if (n == 0) return t1;
        const float t2 = t1 + dt;  // first step inside t1's binade: in the tie case its rounding depends on t1's parity
        const uint32_t b1 = __float_as_uint(t1), b2 = __float_as_uint(t2);
        if ((b1 >> 23) != (b2 >> 23)) { t = t1; continue; }
        n--;
        if (n == 0) return t2;
        const float t3 = t2 + dt;  // from here on the increment is the binade's steady one
        const uint32_t b3 = __float_as_uint(t3);
        if ((b2 >> 23) != (b3 >> 23)) { t = t2; continue; }
        const uint32_t c = b3 - b2;
        if (c == 0) return t2;  // dt below half an ulp: t no longer moves (the reference would spin, too)
        const uint32_t last = ((b2 >> 23) + 1u) << 23;   // first pattern of the next binade
        const uint32_t kmax = (last - 1u - b2) / c;      // steps that stay inside this binade
        const uint32_t k = n < kmax ? n : kmax;
        t = __uint_as_float(b2 + k * c);
        n -= k;
    }
    return t;
}

// dt_gamma > 0: the step grows with t, dt(t) = clamp(t * dt_gamma, dt_min, dt_max), but an occupied step and every trip of the
// skip loop still advance t the SAME way, t <- fl(t + dt(t)) (raymarching.cu:368, 395-398): the points a ray can visit are
// again one fixed sequence from t0, only not an arithmetic one.  Lane k's point of the chunk that starts at t_base: the
// recurrence run k times -- 63 dependent steps per chunk, every lane the same, each keeping its own; nothing closed-form
// reproduces the per-step roundings.  `far` lets the loop stop once the rest of the chunk is past the ray's end.
__device__ __forceinline__ float lattice_points_gamma(float t_base, uint32_t lane, float dt_gamma, float dt_min, float dt_max, float far) {
    float cur = t_base, mine = t_base;
    for (uint32_t i = 1; i < 64; i++) {
        cur = cur + clampf(cur * dt_gamma, dt_min, dt_max);
        mine = lane >= i ? cur : mine;
        if (!(cur < far)) break;  // lanes >= i hold a value >= far: invalid points, whatever they are exactly
    }
    return mine;
}

__device__ __forceinline__ uint64_t lanes_from(uint32_t lane) { return lane >= 64 ? 0ull : (~0ull << lane); }
// __builtin_amdgcn_readlane is an INT builtin: a float argument would be value-converted (truncated)
__device__ __forceinline__ float readlane_f(float v, uint32_t lane) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), (int)lane));
}

// Chunk records: the count pass keeps, for every 64-point chunk of the lattice in which samples were emitted, the
// emit mask and the chunk's first lattice point, in a caller-provided workspace; the write pass rebuilds the samples
// from them (positions are a function of t alone) instead of probing the occupancy grid a second time.
constexpr uint32_t kMarchMaxRecords = 15;
struct MarchChunk { uint64_t mask; float t_base; uint32_t pad; };
struct MarchRayRecords {            // one per ray in the workspace: 256 bytes
    uint32_t n, overflow, pad[2];
    MarchChunk chunk[kMarchMaxRecords];
};
struct MarchRecord {                // register-side view while marching
    MarchChunk *chunk;
    uint32_t n;
    bool overflow;
};
static_assert(sizeof(MarchRayRecords) == 256, "workspace stride");

template <bool WRITE>
__device__ __forceinline__ uint32_t march_ray_wave(const Dda &r, float t0, float far, uint32_t limit, uint32_t lane,
                                                  float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                  uint32_t *n_chunks = nullptr, MarchRecord *__restrict__ rec = nullptr) {
    const bool grows = r.dt_gamma != 0.0f;  // (wave-uniform) dt = clamp(t * dt_gamma, ..): the lattice is not arithmetic
    float t_base = t0;
    bool pending = false;
    float pending_tt = 0.f;
    uint32_t emitted = 0;
    float last_t = t0;
    if (limit == 0) return 0;
    // steady increment (in units of the bit pattern) of the binade t_base lies in, 0 = unknown: while the whole chunk
    // (and the next base) stays inside the binade, lane k's lattice point is bits(t_base) + k * lat_c (see lattice_advance)
    uint32_t lat_c = 0;
    for (;;) {
        if (n_chunks) (*n_chunks)++;
        const uint32_t bb = __float_as_uint(t_base);
        const bool fast = !grows && lat_c != 0 && bb + 64u * lat_c < (((bb >> 23) + 1u) << 23);
        const float t = grows ? lattice_points_gamma(t_base, lane, r.dt_gamma, r.dt_min, r.dt_max, far)
                              : (fast ? __uint_as_float(bb + lane * lat_c) : lattice_advance(t_base, r.dt_const, lane));
        const float t_after = t + clampf(t * r.dt_gamma, r.dt_min, r.dt_max);
        const bool valid = t < far;
        float x = 0, y = 0, z = 0, dtp = 0, tt = 0;
        bool occ = false;
        if (valid) occ = r.probe_impl<true>(t, x, y, z, dtp, tt);
        const uint64_t valid_mask = __ballot(valid);
        const uint64_t occ_mask = __ballot(valid && occ);
        // Where an empty lane's skip lands: the first lane k > lane with !(t_k < tt), 64 if none in this chunk.  The
        // lattice is non-decreasing in k, so every lane finds its own target by a 6-step binary search over the wave
        // (ds_bpermute) -- in parallel -- and the serial replay below only follows these pointers with one v_readlane
        // per skip instead of a readlane + compare + ballot + find-first per skip (PMC: 390 of the 584 instructions
        // per chunk were that scalar loop).
        uint32_t jump = 64;
        {
            uint32_t lo = lane + 1, hi = 64;
#pragma unroll
            for (int s = 0; s < 6; s++) {
                const uint32_t mid = (lo + hi) >> 1;  // <= 63 whenever lo < hi
                const float tm = __shfl(t, (int)(mid & 63u), 64);
                const bool go_right = lo < hi && tm < tt;
                const bool go_left = lo < hi && !(tm < tt);
                lo = go_right ? mid + 1 : lo;
                hi = go_left ? mid : hi;
            }
            jump = lo;
        }

        uint32_t cur = 0;
        bool done = false;
        uint64_t emit_mask = 0;
        if (pending) {
            const uint64_t ge = __ballot(!(t < pending_tt));
            if (ge == 0) {
                cur = 64;
                if (!((valid_mask >> 63) & 1ull)) done = true;  // the whole rest of the lattice is past `far`
            } else {
                cur = (uint32_t)__ffsll((long long)ge) - 1u;
                pending = false;
            }
        }
        while (cur < 64 && !done) {
            if (!((valid_mask >> cur) & 1ull)) { done = true; break; }  // t >= far
            if ((occ_mask >> cur) & 1ull) {
                const uint64_t stop = ~occ_mask & lanes_from(cur);
                const uint32_t e = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
                uint32_t run = e - cur;
                const uint32_t room = limit - emitted;
                if (run >= room) { run = room; done = true; }
                emit_mask |= lanes_from(cur) & ~lanes_from(cur + run);
                emitted += run;
                cur = e;
            } else {
                const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)jump, (int)cur);
                if (j >= 64) {
                    pending = true;
                    pending_tt = readlane_f(tt, cur);
                    cur = 64;
                    if (!((valid_mask >> 63) & 1ull)) done = true;
                } else {
                    cur = j;
                }
            }
        }
        if (WRITE && emit_mask) {
            const uint64_t below = emit_mask & ((1ull << lane) - 1ull);
            const uint32_t n_before = emitted - (uint32_t)__popcll(emit_mask);
            // t after the previous emitted sample: the previous emitting lane's t_after, or the carry
            const int prev_lane = below ? 63 - __clzll((long long)below) : 0;
            const float prev_after = __shfl(t_after, prev_lane, 64);
            if ((emit_mask >> lane) & 1ull) {
                const size_t k = n_before + (uint32_t)__popcll(below);
                xyzs[3 * k] = x; xyzs[3 * k + 1] = y; xyzs[3 * k + 2] = z;
                dirs[3 * k] = r.dx; dirs[3 * k + 1] = r.dy; dirs[3 * k + 2] = r.dz;
                deltas[2 * k] = dtp;
                deltas[2 * k + 1] = t_after - (below ? prev_after : last_t);
            }
            last_t = readlane_f(t_after, 63u - (uint32_t)__clzll((long long)emit_mask));
        }
        if (rec && emit_mask) {  // what the write pass needs to re-create this chunk's samples without probing again
            if (rec->n < kMarchMaxRecords) {
                if (lane == 0) { rec->chunk[rec->n].mask = emit_mask; rec->chunk[rec->n].t_base = t_base; }
                rec->n++;
            } else {
                rec->overflow = true;
            }
        }
        if (done) break;
        t_base = readlane_f(t_after, 63);
        if (!fast && !grows) {
            // the last three lattice points in one binade: the one before last is at least the first step after
            // entering it, so the last difference is the binade's steady increment
            const uint32_t b61 = __float_as_uint(readlane_f(t_after, 61)), b62 = __float_as_uint(readlane_f(t_after, 62));
            const uint32_t b63 = __float_as_uint(t_base);
            lat_c = ((b61 >> 23) == (b63 >> 23) && (b62 >> 23) == (b63 >> 23)) ? b63 - b62 : 0u;
        }
    }
    return emitted;
}

constexpr uint32_t kRaysPerBlock = kBlock / kWave;

__global__ void __launch_bounds__(kBlock) k_march_count_wave(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                             const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                                                             uint32_t N, uint32_t C, uint32_t H, const float *__restrict__ nears,
                                                             const float *__restrict__ fars, int32_t *__restrict__ rays, uint32_t perturb,
                                                             MarchRayRecords *__restrict__ records, const int32_t *__restrict__ counter,
                                                             uint32_t fresh) {
    // the caller's running sample offset (fresh: the counter is scratch, the offset is zero)
    if (records && blockIdx.x == 0 && threadIdx.x == 0) records[N].n = fresh ? 0u : (uint32_t)counter[0];
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
#ifdef PVD_MARCH_PROFILE
    const long long prof_t0 = __builtin_readcyclecounter();
    uint32_t prof_chunks = 0;
#endif
    Dda r;
    r.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid);
    const float t0 = ray_t0(nears[n], r.dt_min, perturb, 42u, n);
    const float far = fars[n];
    uint32_t num;
    if (t0 >= 0.0f && t0 < far) {
        MarchRecord rec = {records ? records[n].chunk : nullptr, 0u, false};
#ifdef PVD_MARCH_PROFILE
        num = march_ray_wave<false>(r, t0, far, max_steps, lane, nullptr, nullptr, nullptr, &prof_chunks, records ? &rec : nullptr);
#else
        num = march_ray_wave<false>(r, t0, far, max_steps, lane, nullptr, nullptr, nullptr, nullptr, records ? &rec : nullptr);
#endif
        if (records && lane == 0) { records[n].n = rec.n; records[n].overflow = rec.overflow ? 1u : 0u; }
    } else {  // empty ray, or a start the lattice walk does not cover (t0 < 0, NaN): serial walk, every lane the same
        float t = t0;
        num = 0;
        while (t < far && num < max_steps) {
            float x, y, z, dt, tn;
            if (r.probe(t, x, y, z, dt, tn)) { num++; t += dt; }
            else t = tn;
        }
        if (records && lane == 0) { records[n].n = 0; records[n].overflow = num ? 1u : 0u; }  // write pass walks it serially again
    }
    if (lane == 0) rays[3 * (size_t)n + 2] = (int32_t)num;
#ifdef PVD_MARCH_PROFILE
    if (lane == 0) { rays[3 * (size_t)n] = (int32_t)(__builtin_readcyclecounter() - prof_t0); rays[3 * (size_t)n + 1] = (int32_t)prof_chunks; }
#endif
}

__global__ void __launch_bounds__(kBlock) k_march_write_wave(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                             const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                                                             uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                             const float *__restrict__ nears, const float *__restrict__ fars,
                                                             float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                             const int32_t *__restrict__ rays, uint32_t perturb,
                                                             const int32_t *__restrict__ budget) {
    const uint32_t n = blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t off = (uint32_t)rays[3 * (size_t)n + 1];
    const uint32_t num = (uint32_t)rays[3 * (size_t)n + 2];
    if (num == 0) return;
    if (off + num >= logical_budget(M, budget)) return;  // strict (:419)
    Dda r;
    r.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid);
    const float t0 = ray_t0(nears[n], r.dt_min, perturb, 42u, n);
    const float far = fars[n];
    float *px = xyzs + 3 * (size_t)off, *pd = dirs + 3 * (size_t)off, *pl = deltas + 2 * (size_t)off;
    if (t0 >= 0.0f) {
        march_ray_wave<true>(r, t0, far, num, lane, px, pd, pl);
    } else if (lane == 0) {
        float t = t0, last_t = t0;
        uint32_t step = 0;
        while (t < far && step < num) {
            float x, y, z, dt, tn;
            if (r.probe(t, x, y, z, dt, tn)) {
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


// Write pass from the chunk records, with the exclusive scan of the counts folded in (N small: every workgroup sums the
// counts of the rays before its own -- a few KB from L2 -- instead of a separate single-workgroup scan launch).
__global__ void __launch_bounds__(kBlock) k_march_write_records(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                               const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                                                               uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                               const float *__restrict__ nears, const float *__restrict__ fars,
                                                               float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                               int32_t *__restrict__ rays, uint32_t perturb,
                                                               const MarchRayRecords *__restrict__ records, int32_t *__restrict__ counter,
                                                               uint32_t fresh, const int32_t *__restrict__ budget) {
    __shared__ uint32_t part[kBlock / kWave], part_all[kBlock / kWave];
    const uint32_t Mlog = logical_budget(M, budget);  // rays are dropped against this; [.., M) is what gets initialised
    const uint32_t n0 = blockIdx.x * kRaysPerBlock;
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    // ---- exclusive prefix of the counts of rays [0, n0)  (fresh: also the total, for the tail every workgroup helps to clear)
    uint32_t acc = 0, all = 0;
    const uint32_t n_sum = fresh ? N : n0;
    for (uint32_t i = threadIdx.x; i < n_sum; i += kBlock) {
        const uint32_t c = (uint32_t)rays[3 * (size_t)i + 2];
        all += c;
        acc += i < n0 ? c : 0u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { acc += __shfl_xor(acc, off, kWave); all += __shfl_xor(all, off, kWave); }
    if (lane == 0) { part[wid] = acc; part_all[wid] = all; }
    __syncthreads();
    uint32_t prefix = records[N].n;  // running offset the caller passed in counter[0]
    uint32_t grand = prefix;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / kWave; w++) { prefix += part[w]; grand += part_all[w]; }
    if (fresh && grand < Mlog) {
        // outputs arrive uninitialised: nobody writes [grand, M) (no ray was dropped, or grand >= Mlog), clear it here, spread
        // over the launch.  (When a ray IS dropped, grand >= Mlog and the first dropped ray clears [its offset, M) below.)
        const size_t stride = (size_t)gridDim.x * kBlock, first = (size_t)blockIdx.x * kBlock + threadIdx.x;
        for (size_t i = 3 * (size_t)grand + first; i < 3 * (size_t)M; i += stride) { xyzs[i] = 0.f; dirs[i] = 0.f; }
        for (size_t i = 2 * (size_t)grand + first; i < 2 * (size_t)M; i += stride) deltas[i] = 0.f;
    }
    const uint32_t n = n0 + wid;
    uint32_t cnt[kRaysPerBlock];
#pragma unroll
    for (uint32_t w = 0; w < kRaysPerBlock; w++) cnt[w] = (n0 + w < N) ? (uint32_t)rays[3 * (size_t)(n0 + w) + 2] : 0u;
    uint32_t off = prefix;
#pragma unroll
    for (uint32_t w = 0; w < kRaysPerBlock; w++) off += (w < wid) ? cnt[w] : 0u;
    if (n0 + kRaysPerBlock >= N && threadIdx.x == 0) {  // the last workgroup publishes the totals (reference: the two atomics, :408-409)
        uint32_t total = prefix;
#pragma unroll
        for (uint32_t w = 0; w < kRaysPerBlock; w++) total += cnt[w];
        counter[0] = (int32_t)total;
        counter[1] = fresh ? (int32_t)N : counter[1] + (int32_t)N;
    }
    if (n >= N) return;
    const uint32_t num = cnt[wid];
    if (lane == 0) { rays[3 * (size_t)n] = (int32_t)n; rays[3 * (size_t)n + 1] = (int32_t)off; }
    if (num == 0) return;
    if (off + num >= Mlog) {  // strict (:419): dropped
        if (fresh && off < Mlog) {  // the first dropped ray (every later one starts at or beyond Mlog): the rest of the buffers stays zero
            for (size_t i = 3 * (size_t)off + lane; i < 3 * (size_t)M; i += kWave) { xyzs[i] = 0.f; dirs[i] = 0.f; }
            for (size_t i = 2 * (size_t)off + lane; i < 2 * (size_t)M; i += kWave) deltas[i] = 0.f;
        }
        return;
    }
    Dda r;
    r.init(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid);
    const float t0 = ray_t0(nears[n], r.dt_min, perturb, 42u, n);
    float *px = xyzs + 3 * (size_t)off, *pd = dirs + 3 * (size_t)off, *pl = deltas + 2 * (size_t)off;
    const MarchRayRecords &rr = records[n];
    if (rr.overflow) {  // more chunks than the record holds, or a start the lattice walk does not cover: march again
        const float far = fars[n];
        if (t0 >= 0.0f) {
            march_ray_wave<true>(r, t0, far, num, lane, px, pd, pl);
        } else if (lane == 0) {
            float t = t0, last_t = t0;
            uint32_t step = 0;
            while (t < far && step < num) {
                float x, y, z, dt, tn;
                if (r.probe(t, x, y, z, dt, tn)) {
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
        return;
    }
    float last_t = t0;
    uint32_t emitted = 0;
    for (uint32_t c = 0; c < rr.n; c++) {
        const uint64_t emit_mask = rr.chunk[c].mask;
        const float t = r.dt_gamma != 0.0f ? lattice_points_gamma(rr.chunk[c].t_base, lane, r.dt_gamma, r.dt_min, r.dt_max, 3.0e38f)
                                           : lattice_advance(rr.chunk[c].t_base, r.dt_const, lane);
        const float t_after = t + clampf(t * r.dt_gamma, r.dt_min, r.dt_max);
        const uint64_t below = emit_mask & ((1ull << lane) - 1ull);
        const int prev_lane = below ? 63 - __clzll((long long)below) : 0;
        const float prev_after = __shfl(t_after, prev_lane, 64);
        if ((emit_mask >> lane) & 1ull) {
            const size_t k = emitted + (uint32_t)__popcll(below);
            xyzs[3 * (size_t)off + 3 * k] = clampf(fmaf(t, r.dx, r.ox), -bound, bound);
            xyzs[3 * (size_t)off + 3 * k + 1] = clampf(fmaf(t, r.dy, r.oy), -bound, bound);
            xyzs[3 * (size_t)off + 3 * k + 2] = clampf(fmaf(t, r.dz, r.oz), -bound, bound);
            pd[3 * k] = r.dx; pd[3 * k + 1] = r.dy; pd[3 * k + 2] = r.dz;
            pl[2 * k] = clampf(t * r.dt_gamma, r.dt_min, r.dt_max);
            pl[2 * k + 1] = t_after - (below ? prev_after : last_t);
        }
        emitted += (uint32_t)__popcll(emit_mask);
        last_t = readlane_f(t_after, 63u - (uint32_t)__clzll((long long)emit_mask));
    }
}


// ------------------------------------------------------------------ composite (train), one wavefront per ray
// Lanes own consecutive samples of the ray; transmittance is an exclusive prefix product and the depth
// parameter / running colour sums are prefix sums across the wave (shuffle scans), carried between
// 64-sample chunks.  Loads and stores are contiguous per ray.  Same formulas as the serial kernels below
// (which remain the reference restatement); only the floating-point association differs.

__device__ __forceinline__ float wave_incl_scan_add(float v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(v, off, 64);
        if ((int)lane >= off) v += up;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_scan_mul(float v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(v, off, 64);
        if ((int)lane >= off) v *= up;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Optional fused epilogue of run_cuda (distill_mutual/renderer.py:445-446):
//   image += (1 - weights_sum) * bg_color ;  depth = clamp(depth - near, 0) / (far - near + eps)
struct CompositeEpilogue {
    const float *bg;      // [N,3] per-ray background, or null -> bg_scalar
    float bg_scalar;
    const float *nears, *fars;
    float depth_eps;
    // backward only: grad buffers arrive uninitialised and `rays` is a table of pvd_march_rays_train (offsets = exclusive
    // prefix sum of the counts in row order, from 0): the kernel clears every slot no ray owns
    uint32_t fresh;
    const int32_t *budget;  // logical sample budget in device memory (see logical_budget), or NULL
};

// The stage-3 distillation objective riding on the student's compositing launches (pvd_composite_objective_*): the forward
// launch also forms the four sums of squares of Trainer.train_step's normL2 terms (utils.py:1109-1176) -- the image term by
// the workgroups that composite the rays, the feature / sigma / colour terms by extra workgroups of the same launch -- and the
// backward launch forms the image gradient coef * (I_stu - I_tea) on the fly and writes the feature / colour gradients from
// extra workgroups.  Two of the five short dependent launches between the student's head forward and head backward go away;
// the arithmetic per element is that of k_sumsq4 / k_sumsq4_bwd (distill.hip).
struct CompositeObjective {
    const float *img_t;           // [N,3] teacher image, by ray index
    const float *fea_s, *fea_t;   // [rows,16] feature_sigma_color (column 0 = sigma_l)
    const float *col_s, *col_t;   // [rows,3] color_l
    uint32_t rows;
    uint32_t ray_blocks;          // workgroups [0, ray_blocks) composite rays, the others walk the feature rows
    float *partials;              // forward: float4 per workgroup {sum (I_t - I_s)^2, sum dF^2, sum dsigma^2, sum dc^2}
    const float *coef, *upstream; // backward: r_i / ||.||_i [4] (k_loss_final), upstream gradient [1]
    float *g_fea, *g_col;         // backward outputs [rows,16], [rows,3]
    // the objective FINISHED by the backward launch (no k_loss_final between the passes): every workgroup reduces the forward
    // launch's partial sums for itself and forms the coefficients; workgroup 0 publishes loss / norms / coefficients / sums.
    // The feature rate's per-step decay (utils.py:1044) is then applied by ONE thread of the forward launch (rates_decay),
    // which nothing else reads meanwhile.
    float *rates_decay;           // forward: rates_decay[1] *= fea_decay (or NULL)
    float fea_decay;
    const float *rates;           // backward: != NULL = finish here (coef is then an OUTPUT)
    const float *extra;           // partial sums of a parameter-only term of the loss value (L1 regulariser), or NULL
    uint32_t n_extra, nparts;
    float *sums, *loss, *coef_out, *norms;
};

__device__ __forceinline__ float objective_block_sum(float v, float *__restrict__ sh) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / kWave; w++) r += sh[w];
    return r;
}

// (backward, ob.rates != NULL) the four coefficients from the partial sums; same arithmetic as k_loss_final
__device__ __forceinline__ void objective_finish(const CompositeObjective &ob, float (&c4)[4], float *__restrict__ sh) {
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (uint32_t i = threadIdx.x; i < ob.nparts; i += kBlock) {
        const float4 v = reinterpret_cast<const float4 *>(ob.partials)[i];
        a += v.x; b += v.y; c += v.z; d += v.w;
    }
    float s4[4] = {objective_block_sum(a, sh), objective_block_sum(b, sh), objective_block_sum(c, sh), objective_block_sum(d, sh)};
    float nrm[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        nrm[i] = sqrtf(s4[i]);
        c4[i] = nrm[i] > 0.f ? ob.rates[i] / nrm[i] : 0.f;  // d (r ||x||) / dx = r x / ||x||
    }
    if (blockIdx.x == 0) {  // the value of the objective (k_loss_final's summation order: parameter-only term first)
        float e = 0.f;
        for (uint32_t i = threadIdx.x; i < ob.n_extra; i += kBlock) e += ob.extra[i];
        e = objective_block_sum(e, sh);
        if (threadIdx.x == 0) {
            float t = e;
#pragma unroll
            for (int i = 0; i < 4; i++) { t += ob.rates[i] * nrm[i]; ob.sums[i] = s4[i]; ob.norms[i] = nrm[i]; ob.coef_out[i] = c4[i]; }
            ob.loss[0] = t;
        }
    }
}

// workgroup `blk` of `nblk` over the feature rows: the sums of k_sumsq4 without its image term
__device__ __forceinline__ void objective_row_sums(const CompositeObjective &ob, uint32_t blk, uint32_t nblk, float *__restrict__ sh) {
    const uint32_t tid = blk * kBlock + threadIdx.x, stride = nblk * kBlock;
    float s_fea = 0.f, s_sig = 0.f, s_col = 0.f;
    for (uint32_t i = tid; i < ob.rows * 4u; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(ob.fea_s)[i], b = reinterpret_cast<const float4 *>(ob.fea_t)[i];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        s_fea += dx * dx + dy * dy + dz * dz + dw * dw;
        if ((i & 3u) == 0) s_sig += dx * dx;
    }
    for (uint32_t i = tid; i < ob.rows * 3u; i += stride) { const float d = ob.col_s[i] - ob.col_t[i]; s_col += d * d; }
    s_fea = objective_block_sum(s_fea, sh); s_sig = objective_block_sum(s_sig, sh); s_col = objective_block_sum(s_col, sh);
    if (threadIdx.x == 0) reinterpret_cast<float4 *>(ob.partials)[ob.ray_blocks + blk] = make_float4(0.f, s_fea, s_sig, s_col);
}

// ... and the gradients of k_sumsq4_bwd without the image's
__device__ __forceinline__ void objective_row_grads(const CompositeObjective &ob, const float (&c4)[4], uint32_t blk, uint32_t nblk) {
    const uint32_t tid = blk * kBlock + threadIdx.x, stride = nblk * kBlock;
    const float up = ob.upstream[0];
    const float c_fea = c4[1] * up, c_sig = c4[2] * up, c_col = c4[3] * up;
    for (uint32_t i = tid; i < ob.rows * 4u; i += stride) {
        const float4 a = reinterpret_cast<const float4 *>(ob.fea_s)[i], b = reinterpret_cast<const float4 *>(ob.fea_t)[i];
        float4 g = make_float4(c_fea * (a.x - b.x), c_fea * (a.y - b.y), c_fea * (a.z - b.z), c_fea * (a.w - b.w));
        if ((i & 3u) == 0) g.x += c_sig * (a.x - b.x);
        reinterpret_cast<float4 *>(ob.g_fea)[i] = g;
    }
    for (uint32_t i = tid; i < ob.rows * 3u; i += stride) ob.g_col[i] = c_col * (ob.col_s[i] - ob.col_t[i]);
}

// reference: kernel_composite_rays_train_forward, raymarching.cu:504-582
template <bool EPI, bool OBJ = false>
__global__ void __launch_bounds__(kBlock) k_composite_fwd_wave(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                               const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                               uint32_t M, uint32_t N, float *__restrict__ weights_sum,
                                                               float *__restrict__ depth, float *__restrict__ image, CompositeEpilogue ep,
                                                               CompositeObjective ob = CompositeObjective{}) {
    __shared__ float obj_sh[kBlock / kWave];
    if (OBJ && blockIdx.x >= ob.ray_blocks) {
        if (ob.rates_decay && blockIdx.x == ob.ray_blocks && threadIdx.x == 0) ob.rates_decay[1] *= ob.fea_decay;
        objective_row_sums(ob, blockIdx.x - ob.ray_blocks, gridDim.x - ob.ray_blocks, obj_sh);
        return;
    }
    const uint32_t n = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (!OBJ && n >= N) return;
    const bool live = n < N;  // (OBJ: the whole workgroup stays for the reduction of the image term)
    const uint32_t index = live ? (uint32_t)rays[3 * (size_t)n] : 0u;
    const uint32_t offset = live ? (uint32_t)rays[3 * (size_t)n + 1] : 0u;
    const uint32_t num = live ? (uint32_t)rays[3 * (size_t)n + 2] : 0u;
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    if (!(num == 0 || offset + num >= logical_budget(M, EPI ? ep.budget : nullptr))) {
        float T_carry = 1.0f, t_carry = 0.0f;
        for (uint32_t base = 0; base < num; base += 64) {
            const uint32_t s = base + lane;
            const bool in = s < num;
            const size_t i = (size_t)offset + s;
            float alpha = 0.f, dl1 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (in) {
                const float2 dl = reinterpret_cast<const float2 *>(deltas)[i];
                alpha = 1.0f - __expf(-sigmas[i] * dl.x);
                dl1 = dl.y;
                c0 = rgbs[3 * i]; c1 = rgbs[3 * i + 1]; c2 = rgbs[3 * i + 2];
            }
            const float incl = wave_incl_scan_mul(1.0f - alpha, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float w = alpha * (T_carry * excl);
            const float t = t_carry + wave_incl_scan_add(dl1, lane);
            r += w * c0; g += w * c1; b += w * c2;
            d += w * t;
            ws += w;
            T_carry *= __shfl(incl, 63, 64);
            t_carry = __shfl(t, 63, 64);
        }
        r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    }
    float d2 = 0.f;  // (OBJ) this ray's share of sum (I_tea - I_stu)^2
    if (lane == 0 && live) {
        if (EPI) {
            const float t = 1.0f - ws;
            const float b0 = ep.bg ? ep.bg[3 * (size_t)index] : ep.bg_scalar, b1 = ep.bg ? ep.bg[3 * (size_t)index + 1] : ep.bg_scalar,
                        b2 = ep.bg ? ep.bg[3 * (size_t)index + 2] : ep.bg_scalar;
            r = r + t * b0; g = g + t * b1; b = b + t * b2;
            const float near = ep.nears[index], far = ep.fars[index];
            d = fmaxf(d - near, 0.0f) / (far - near + ep.depth_eps);
        }
        weights_sum[index] = ws;
        depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
        if (OBJ) {
            const float e0 = ob.img_t[3 * (size_t)index] - r, e1 = ob.img_t[3 * (size_t)index + 1] - g, e2 = ob.img_t[3 * (size_t)index + 2] - b;
            d2 = e0 * e0 + e1 * e1 + e2 * e2;
        }
    }
    if (OBJ) {
        const float s_img = objective_block_sum(d2, obj_sh);
        if (threadIdx.x == 0) reinterpret_cast<float4 *>(ob.partials)[blockIdx.x] = make_float4(s_img, 0.f, 0.f, 0.f);
    }
}

// reference: kernel_composite_rays_train_backward, raymarching.cu:606-686.  With EPI the incoming gradient is
// w.r.t. the blended image: d blended / d ws = -bg, and `image` holds the blended colours (un-blended here).
template <bool EPI, bool OBJ = false>
__global__ void __launch_bounds__(kBlock) k_composite_bwd_wave(const float *__restrict__ grad_ws, const float *__restrict__ grad_image,
                                                               const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                               const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                               const float *__restrict__ weights_sum, const float *__restrict__ image,
                                                               uint32_t M, uint32_t N, float *__restrict__ grad_sigmas,
                                                               float *__restrict__ grad_rgbs, CompositeEpilogue ep,
                                                               CompositeObjective ob = CompositeObjective{}) {
    float c4[4] = {0.f, 0.f, 0.f, 0.f};
    if (OBJ) {
        __shared__ float fin_sh[kBlock / kWave];
        if (ob.rates) objective_finish(ob, c4, fin_sh);  // (all threads of every workgroup: before anybody leaves)
        else { c4[0] = ob.coef[0]; c4[1] = ob.coef[1]; c4[2] = ob.coef[2]; c4[3] = ob.coef[3]; }
    }
    if (OBJ && blockIdx.x >= ob.ray_blocks) {
        objective_row_grads(ob, c4, blockIdx.x - ob.ray_blocks, gridDim.x - ob.ray_blocks);
        return;
    }
    const uint32_t n = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3 * (size_t)n];
    const uint32_t offset = (uint32_t)rays[3 * (size_t)n + 1];
    const uint32_t num = (uint32_t)rays[3 * (size_t)n + 2];
    if (EPI && ep.fresh) {
        // slots nobody owns: [end of the last ray, M) when no ray was dropped (else that end is >= M), spread over the launch
        const uint32_t end = (uint32_t)rays[3 * (size_t)(N - 1) + 1] + (uint32_t)rays[3 * (size_t)(N - 1) + 2];
        for (size_t i = (size_t)end + (size_t)n * kWave + lane; i < M; i += (size_t)N * kWave) {
            grad_sigmas[i] = 0.f; grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f;
        }
        // ... and [offset, M) of the first dropped ray (every later ray starts at or beyond the budget)
        if (num != 0 && offset + num >= logical_budget(M, ep.budget))
            for (size_t i = (size_t)offset + lane; i < M; i += kWave) {
                grad_sigmas[i] = 0.f; grad_rgbs[3 * i] = 0.f; grad_rgbs[3 * i + 1] = 0.f; grad_rgbs[3 * i + 2] = 0.f;
            }
    }
    if (num == 0 || offset + num >= logical_budget(M, EPI ? ep.budget : nullptr)) return;
    float gws = grad_ws ? grad_ws[index] : 0.0f;
    float rF = image[3 * (size_t)index], gF = image[3 * (size_t)index + 1], bF = image[3 * (size_t)index + 2];
    float g0, g1, g2;
    if (OBJ) {  // d (r_rgb ||I_tea - I_stu||) / d I_stu, as k_sumsq4_bwd writes it: coef * upstream * (I_stu - I_tea)
        const float c_img = c4[0] * ob.upstream[0];
        g0 = c_img * (rF - ob.img_t[3 * (size_t)index]); g1 = c_img * (gF - ob.img_t[3 * (size_t)index + 1]);
        g2 = c_img * (bF - ob.img_t[3 * (size_t)index + 2]);
    } else {
        g0 = grad_image[3 * (size_t)index]; g1 = grad_image[3 * (size_t)index + 1]; g2 = grad_image[3 * (size_t)index + 2];
    }
    const float wsF = weights_sum[index];
    if (EPI) {
        const float b0 = ep.bg ? ep.bg[3 * (size_t)index] : ep.bg_scalar, b1 = ep.bg ? ep.bg[3 * (size_t)index + 1] : ep.bg_scalar,
                    b2 = ep.bg ? ep.bg[3 * (size_t)index + 2] : ep.bg_scalar;
        const float t = 1.0f - wsF;
        rF -= t * b0; gF -= t * b1; bF -= t * b2;
        gws -= g0 * b0 + g1 * b1 + g2 * b2;
    }
    float T_carry = 1.0f, r_carry = 0.f, g_carry = 0.f, b_carry = 0.f, ws_carry = 0.f;
    for (uint32_t base = 0; base < num; base += 64) {
        const uint32_t s = base + lane;
        const bool in = s < num;
        const size_t i = (size_t)offset + s;
        float alpha = 0.f, dl0 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (in) {
            dl0 = deltas[2 * i];
            alpha = 1.0f - __expf(-sigmas[i] * dl0);
            c0 = rgbs[3 * i]; c1 = rgbs[3 * i + 1]; c2 = rgbs[3 * i + 2];
        }
        const float incl = wave_incl_scan_mul(1.0f - alpha, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = alpha * (T_carry * excl);
        const float T = T_carry * incl;  // post-update transmittance (:660, :668-673)
        const float r = r_carry + wave_incl_scan_add(w * c0, lane);
        const float g = g_carry + wave_incl_scan_add(w * c1, lane);
        const float b = b_carry + wave_incl_scan_add(w * c2, lane);
        const float ws = ws_carry + wave_incl_scan_add(w, lane);
        if (in) {
            grad_rgbs[3 * i] = g0 * w; grad_rgbs[3 * i + 1] = g1 * w; grad_rgbs[3 * i + 2] = g2 * w;
            grad_sigmas[i] = dl0 * (g0 * (T * c0 - (rF - r)) + g1 * (T * c1 - (gF - g)) + g2 * (T * c2 - (bF - b)) + gws * (T - (wsF - ws)));
        }
        T_carry = __shfl(T, 63, 64);
        r_carry = __shfl(r, 63, 64); g_carry = __shfl(g, 63, 64); b_carry = __shfl(b, 63, 64); ws_carry = __shfl(ws, 63, 64);
    }
}

// ------------------------------------------------------------------ inference trio

// reference: kernel_march_rays, raymarching.cu:704-811
__global__ void __launch_bounds__(kBlock) k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                       const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                       const float *__restrict__ rays_d, float bound, float dt_gamma,
                                                       uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *__restrict__ grid,
                                                       const float *__restrict__ fars, float *__restrict__ xyzs,
                                                       float *__restrict__ dirs, float *__restrict__ deltas, uint32_t perturb) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    Dda r;
    r.init(rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, bound, dt_gamma, max_steps, C, H, grid);
    const float far = fars[index];
    float t = ray_t0(rays_t[n], r.dt_min, perturb, (uint64_t)perturb, n);  // seed = perturb (:816), advance(n) (:750)
    float last_t = t;
    float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step, *pl = deltas + 2 * (size_t)n * n_step;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        float x, y, z, dt, tn;
        if (r.probe(t, x, y, z, dt, tn)) {
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

// reference: kernel_composite_rays, raymarching.cu:825-909
__global__ void __launch_bounds__(kBlock) k_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                           float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                                           const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                           float *__restrict__ weights_sum, float *__restrict__ depth,
                                                           float *__restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    float t = rays_t[n];
    float ws = weights_sum[index], d = depth[index];
    float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
    uint32_t step = 0;
    while (step < n_step) {
        const size_t i = (size_t)n * n_step + step;
        const float dl0 = deltas[2 * i];
        if (dl0 == 0) break;  // end-of-ray marker (:862)
        const float alpha = 1.0f - __expf(-sigmas[i] * dl0);
        const float T = 1 - ws;  // (:872)
        const float w = alpha * T;
        ws += w;
        t += deltas[2 * i + 1];
        d += w * t;
        r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
        if ((double)T < 1e-4) break;  // after accumulating this sample (:886)
        step++;
    }
    rays_t[n] = (step < n_step) ? -1.0f : t;
    weights_sum[index] = ws;
    depth[index] = d;
    image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
}

// reference: kernel_compact_rays, raymarching.cu:921-939.  Wave-ballot compaction: one
// global atomic per workgroup instead of one per surviving ray; order inside a workgroup is
// preserved, order of workgroups is whatever the atomic gives (as in the reference).
__global__ void __launch_bounds__(kBlock) k_compact_rays(uint32_t n_alive, int32_t *__restrict__ rays_alive,
                                                         const int32_t *__restrict__ rays_alive_old, float *__restrict__ rays_t,
                                                         const float *__restrict__ rays_t_old, int32_t *__restrict__ alive_counter) {
    __shared__ uint32_t wave_cnt[kBlock / kWave];
    __shared__ uint32_t block_base;
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    float t = -1.0f;
    int32_t id = 0;
    if (n < n_alive) { t = rays_t_old[n]; id = rays_alive_old[n]; }
    const bool keep = (n < n_alive) && (t >= 0);
    const unsigned long long mask = __ballot(keep);
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wid] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < kBlock / kWave; w++) { const uint32_t c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
        block_base = tot ? (uint32_t)atomicAdd(alive_counter, (int32_t)tot) : 0u;
    }
    __syncthreads();
    if (keep) {
        const uint32_t dst = block_base + wave_cnt[wid] + rank;
        rays_alive[dst] = id;
        rays_t[dst] = t;
    }
}

// ------------------------------------------------------------------ inference rounds without host round trips
//
// The reference's inference loop (renderer.py:450-543) reads the number of surviving rays back to the host after every
// round (`alive_counter.item()`, :488): n_alive sizes the next launches and picks n_step = max(min(N / n_alive, 8), 1).
// Here that state lives on the device -- `st` below -- and every kernel of a round takes its extent from it, in
// persistent grid-stride loops, so a round is a fixed launch sequence the host can issue without looking:
//     [k_infer_compact]  k_infer_begin  k_infer_march  <model forward on st.rows rows>  k_infer_composite
// The host checks st only every few rounds (to stop, and to shrink the launch sizes).  Per-ray results are those of the
// reference loop: same n_step rule, same march / composite arithmetic (k_march_rays / k_composite_rays bodies).
struct InferState {
    int32_t cnt[2];      // alive counts: round i uses cnt[i & 1] (written by the compaction of its own start)
    int32_t n_alive;     // this round
    int32_t n_step;
    int32_t rows;        // n_alive * n_step: the rows of xyzs / dirs / deltas / sigmas / rgbs in use
    int32_t steps_done;  // sum of n_step over the rounds so far (the reference's `step`, :483-541)
    int32_t rounds;
    int32_t pad;
};

constexpr uint32_t kInferBlocks = 1024;  // persistent grids: 4 workgroups per CU

__global__ void k_infer_begin(InferState *__restrict__ st, uint32_t parity, uint32_t N, uint32_t max_steps) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int32_t n = st->cnt[parity];
    if ((uint32_t)st->steps_done >= max_steps) n = 0;  // `while step < max_steps` (:483)
    const int32_t n_step = n > 0 ? max(min((int32_t)(N / (uint32_t)n), 8), 1) : 0;  // (:493)
    st->n_alive = n;
    st->n_step = n_step;
    st->rows = n * n_step;
    st->steps_done += n_step;
    st->rounds += n > 0 ? 1 : 0;
    st->cnt[parity ^ 1u] = 0;  // target of the next round's compaction
}

__global__ void __launch_bounds__(kBlock) k_infer_compact(const InferState *__restrict__ st_in, InferState *__restrict__ st, uint32_t parity,
                                                          int32_t *__restrict__ rays_alive, const int32_t *__restrict__ rays_alive_old,
                                                          float *__restrict__ rays_t, const float *__restrict__ rays_t_old) {
    __shared__ uint32_t wave_cnt[kBlock / kWave];
    __shared__ uint32_t block_base;
    const uint32_t n_old = (uint32_t)st_in->n_alive;  // the round that just finished
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * kBlock; base < n_old; base += gridDim.x * kBlock) {  // uniform per workgroup
        const uint32_t n = base + threadIdx.x;
        float t = -1.0f;
        int32_t id = 0;
        if (n < n_old) { t = rays_t_old[n]; id = rays_alive_old[n]; }
        const bool keep = (n < n_old) && (t >= 0);
        const unsigned long long mask = __ballot(keep);
        const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < kBlock / kWave; w++) { const uint32_t c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
            block_base = tot ? (uint32_t)atomicAdd(&st->cnt[parity], (int32_t)tot) : 0u;
        }
        __syncthreads();
        if (keep) {
            const uint32_t dst = block_base + wave_cnt[wid] + rank;
            rays_alive[dst] = id;
            rays_t[dst] = t;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock) k_infer_march(const InferState *__restrict__ st, const int32_t *__restrict__ rays_alive,
                                                        const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                        const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                                        uint32_t C, uint32_t H, const uint8_t *__restrict__ grid, const float *__restrict__ fars,
                                                        float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                        uint32_t perturb) {
    const uint32_t n_alive = (uint32_t)st->n_alive, n_step = (uint32_t)st->n_step;
    for (uint32_t n = blockIdx.x * kBlock + threadIdx.x; n < n_alive; n += gridDim.x * kBlock) {
        const int32_t index = rays_alive[n];
        Dda r;
        r.init(rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, bound, dt_gamma, max_steps, C, H, grid);
        const float far = fars[index];
        float t = ray_t0(rays_t[n], r.dt_min, perturb, (uint64_t)perturb, n);
        float last_t = t;
        float *px = xyzs + 3 * (size_t)n * n_step, *pd = dirs + 3 * (size_t)n * n_step, *pl = deltas + 2 * (size_t)n * n_step;
        uint32_t step = 0;
        while (t < far && step < n_step) {
            float x, y, z, dt, tn;
            if (r.probe(t, x, y, z, dt, tn)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
                t += dt;
                pl[0] = dt; pl[1] = t - last_t; last_t = t;
                px += 3; pd += 3; pl += 2; step++;
            } else {
                t = tn;
            }
        }
        // the reference hands over zero-filled buffers (raymarching.py:421-426): dt == 0 marks the end of a ray
        for (; step < n_step; step++) {
            px[0] = px[1] = px[2] = 0.f; pd[0] = pd[1] = pd[2] = 0.f; pl[0] = pl[1] = 0.f;
            px += 3; pd += 3; pl += 2;
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_infer_composite(const InferState *__restrict__ st, const int32_t *__restrict__ rays_alive,
                                                            float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                                            const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                            float sigma_scale, float *__restrict__ weights_sum, float *__restrict__ depth,
                                                            float *__restrict__ image) {
    const uint32_t n_alive = (uint32_t)st->n_alive, n_step = (uint32_t)st->n_step;
    for (uint32_t n = blockIdx.x * kBlock + threadIdx.x; n < n_alive; n += gridDim.x * kBlock) {
        const int32_t index = rays_alive[n];
        float t = rays_t[n];
        float ws = weights_sum[index], d = depth[index];
        float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
        uint32_t step = 0;
        while (step < n_step) {
            const size_t i = (size_t)n * n_step + step;
            const float dl0 = deltas[2 * i];
            if (dl0 == 0) break;
            const float alpha = 1.0f - __expf(-(sigma_scale * sigmas[i]) * dl0);  // density_scale * sigmas (renderer.py:528)
            const float T = 1 - ws;
            const float w = alpha * T;
            ws += w;
            t += deltas[2 * i + 1];
            d += w * t;
            r += w * rgbs[3 * i]; g += w * rgbs[3 * i + 1]; b += w * rgbs[3 * i + 2];
            if ((double)T < 1e-4) break;
            step++;
        }
        rays_t[n] = (step < n_step) ? -1.0f : t;
        weights_sum[index] = ws;
        depth[index] = d;
        image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
    }
}

}  // namespace pvd

// ====================================================================== C ABI
using namespace pvd;

#define PVD_REQUIRE(cond) \
    do { if (!(cond)) return PVD_ERR_INVALID; } while (0)

extern "C" {

int pvd_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                           float *nears, float *fars, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(rays_o && rays_d && aabb && nears && fars);
    hipLaunchKernelGGL(k_near_far, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N, min_near, nears, fars);
    return check_launch();
}

int pvd_get_rays(const float *pose, float fx, float fy, float cx, float cy, const int64_t *inds, uint32_t W, uint32_t N,
                 float *rays_o, float *rays_d, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(pose && rays_o && rays_d && W > 0);
    hipLaunchKernelGGL(k_get_rays, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pose, fx, fy, cx, cy, inds, W, N, rays_o, rays_d);
    return check_launch();
}

int pvd_make_ray_batch(const float *poses, uint32_t P, int64_t *state, uint64_t seed, float fx, float fy, float cx, float cy, uint32_t H,
                       uint32_t W, uint32_t N, const float *aabb, float min_near, int64_t *inds, float *rays_o, float *rays_d, float *bg,
                       float *nears, float *fars, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(poses && P > 0 && state && aabb && rays_o && rays_d && nears && fars && H > 0 && W > 0);
    PVD_REQUIRE((uint64_t)H * W < (1ull << 32));
    hipLaunchKernelGGL(k_make_ray_batch, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, poses, P, (long long *)state, seed, fx,
                       fy, cx, cy, H, W, N, aabb, min_near, inds, rays_o, rays_d, bg, nears, fars);
    return check_launch();
}

int pvd_polar_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(rays_o && rays_d && coords);
    hipLaunchKernelGGL(k_polar, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, rays_o, rays_d, radius, N, coords);
    return check_launch();
}

int pvd_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(coords && indices);
    hipLaunchKernelGGL(k_morton3D, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, coords, N, indices);
    return check_launch();
}

int pvd_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(coords && indices);
    hipLaunchKernelGGL(k_morton3D_invert, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, indices, N, coords);
    return check_launch();
}

int pvd_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(grid && bitfield);
    PVD_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 15u) == 0);
    hipLaunchKernelGGL(k_packbits, dim3(div_up(N, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, grid, N, density_thresh, bitfield);
    return check_launch();
}

constexpr uint32_t kMarchFusedScanMaxRays = 16384;  // beyond this the per-workgroup prefix (O(N) reads each) stops paying

size_t pvd_march_workspace_bytes(uint32_t N) { return ((size_t)N + 1) * sizeof(MarchRayRecords); }

int pvd_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                         const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                         uint32_t perturb, pvd_stream_t stream) {
    return pvd_march_rays_train_ws(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                   counter, perturb, nullptr, 0, 0u, nullptr, stream);
}

int pvd_march_rays_train_ws(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                            uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                            const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                            uint32_t perturb, void *workspace, size_t workspace_bytes, uint32_t flags, const int32_t *budget_dev,
                            pvd_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool fresh = (flags & PVD_MARCH_FRESH) != 0;
    if (N == 0) {
        if (fresh && M && xyzs && dirs && deltas && counter) {
            (void)hipMemsetAsync(xyzs, 0, 3 * (size_t)M * sizeof(float), s); (void)hipMemsetAsync(dirs, 0, 3 * (size_t)M * sizeof(float), s);
            (void)hipMemsetAsync(deltas, 0, 2 * (size_t)M * sizeof(float), s); (void)hipMemsetAsync(counter, 0, 2 * sizeof(int32_t), s);
        }
        return PVD_OK;
    }
    PVD_REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter);
    PVD_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 1024 && max_steps >= 1);
    // one wavefront per ray, for a constant step (dt_gamma = 0: all BASELINE configs with bound 1) and for a growing one alike
    // (the lattice of dt_gamma > 0 is a recurrence instead of an arithmetic sequence; thread-per-ray kernels left 94 % of the SIMDs
    // idle at 4096 rays and were removed in round 5)
    MarchRayRecords *records = (workspace && workspace_bytes >= pvd_march_workspace_bytes(N) && N <= kMarchFusedScanMaxRays)
                                   ? (MarchRayRecords *)workspace : nullptr;
    if (fresh && !records) {  // only the record path initialises what it does not write: do it up front for the others
        (void)hipMemsetAsync(xyzs, 0, 3 * (size_t)M * sizeof(float), s); (void)hipMemsetAsync(dirs, 0, 3 * (size_t)M * sizeof(float), s);
        (void)hipMemsetAsync(deltas, 0, 2 * (size_t)M * sizeof(float), s); (void)hipMemsetAsync(counter, 0, 2 * sizeof(int32_t), s);
    }
    const dim3 g(div_up(N, kRaysPerBlock)), b(kBlock);
    hipLaunchKernelGGL(k_march_count_wave, g, b, 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, rays, perturb,
                       records, counter, records && fresh ? 1u : 0u);
    if (records) {  // two launches: the write pass rebuilds the samples from the chunk records and scans the counts itself
        hipLaunchKernelGGL(k_march_write_records, g, b, 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs,
                           dirs, deltas, rays, perturb, records, counter, fresh ? 1u : 0u, budget_dev);
        return check_launch();
    }
#ifdef PVD_MARCH_PROFILE
    return check_launch();
#endif
    hipLaunchKernelGGL(k_march_scan, dim3(1), dim3(kScanBlock), 0, s, rays, N, counter);
    hipLaunchKernelGGL(k_march_write_wave, g, b, 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs,
                       deltas, rays, perturb, budget_dev);
    return check_launch();
}

int pvd_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                                     uint32_t M, uint32_t N, float *weights_sum, float *depth, float *image,
                                     pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && depth && image);
    hipLaunchKernelGGL(k_composite_fwd_wave<false>, dim3(div_up(N, kBlock / kWave)), dim3(kBlock), 0, (hipStream_t)stream, sigmas, rgbs, deltas,
                       rays, M, N, weights_sum, depth, image, CompositeEpilogue{});
    return check_launch();
}

int pvd_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                      const float *rgbs, const float *deltas, const int32_t *rays, const float *weights_sum,
                                      const float *image, uint32_t M, uint32_t N, float *grad_sigmas, float *grad_rgbs,
                                      pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs);
    hipLaunchKernelGGL(k_composite_bwd_wave<false>, dim3(div_up(N, kBlock / kWave)), dim3(kBlock), 0, (hipStream_t)stream, grad_weights_sum,
                       grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs, CompositeEpilogue{});
    return check_launch();
}

int pvd_composite_rays_train_bg_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M,
                                        uint32_t N, const float *bg, float bg_scalar, const float *nears, const float *fars,
                                        float depth_eps, float *weights_sum, float *depth, float *image, const int32_t *budget_dev,
                                        pvd_stream_t stream) {
    if (N == 0) return PVD_OK;
    PVD_REQUIRE(sigmas && rgbs && deltas && rays && nears && fars && weights_sum && depth && image);
    const CompositeEpilogue ep{bg, bg_scalar, nears, fars, depth_eps, 0u, budget_dev};
    hipLaunchKernelGGL(k_composite_fwd_wave<true>, dim3(div_up(N, kBlock / kWave)), dim3(kBlock), 0, (hipStream_t)stream, sigmas, rgbs, deltas,
                       rays, M, N, weights_sum, depth, image, ep);
    return check_launch();
}

int pvd_composite_rays_train_bg_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas, const float *rgbs,
                                         const float *deltas, const int32_t *rays, const float *weights_sum, const float *image,
                                         uint32_t M, uint32_t N, const float *bg, float bg_scalar, float *grad_sigmas, float *grad_rgbs,
                                         uint32_t flags, const int32_t *budget_dev, pvd_stream_t stream) {
    const bool fresh = (flags & PVD_MARCH_FRESH) != 0;
    if (N == 0) {
        if (fresh && M && grad_sigmas && grad_rgbs) {
            (void)hipMemsetAsync(grad_sigmas, 0, (size_t)M * sizeof(float), (hipStream_t)stream);
            (void)hipMemsetAsync(grad_rgbs, 0, 3 * (size_t)M * sizeof(float), (hipStream_t)stream);
        }
        return PVD_OK;
    }
    PVD_REQUIRE(grad_image && sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs);
    const CompositeEpilogue ep{bg, bg_scalar, nullptr, nullptr, 0.f, fresh ? 1u : 0u, budget_dev};
    hipLaunchKernelGGL(k_composite_bwd_wave<true>, dim3(div_up(N, kBlock / kWave)), dim3(kBlock), 0, (hipStream_t)stream, grad_weights_sum,
                       grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs, ep);
    return check_launch();
}

// ---- the student's compositing with the stage-3 objective riding along (see CompositeObjective)
static uint32_t objective_row_blocks(uint32_t rows, uint32_t cap) {
    uint32_t b = div_up(rows * 4u, kBlock);
    if (b > cap) b = cap;
    return b < 1 ? 1 : b;
}

uint32_t pvd_composite_objective_blocks(uint32_t N, uint32_t rows) { return div_up(N, kBlock / kWave) + objective_row_blocks(rows, 256u); }
// PVD_OBJECTIVE_FIXED_PARTS: the number of partial sums depends on N alone (the feature rows always take 256 workgroups; those beyond
// the rows write zeros), so ranks whose sample counts differ leave buffers of ONE size -- which ray-DP all-reduces between the launches
uint32_t pvd_composite_objective_blocks_fixed(uint32_t N) { return div_up(N, kBlock / kWave) + 256u; }

int pvd_composite_objective_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M, uint32_t N,
                                    const float *bg, float bg_scalar, const float *nears, const float *fars, float depth_eps,
                                    float *weights_sum, float *depth, float *image, const int32_t *budget_dev, const float *img_tea,
                                    const float *fea_stu, const float *fea_tea, const float *col_stu, const float *col_tea, uint32_t rows,
                                    float *S4, float *rates4_decay, float fea_decay, uint32_t flags, pvd_stream_t stream) {
    PVD_REQUIRE(N > 0 && rows > 0);
    PVD_REQUIRE(sigmas && rgbs && deltas && rays && nears && fars && weights_sum && depth && image);
    PVD_REQUIRE(img_tea && fea_stu && fea_tea && col_stu && col_tea && S4);
    const CompositeEpilogue ep{bg, bg_scalar, nears, fars, depth_eps, 0u, budget_dev};
    CompositeObjective ob{};
    ob.img_t = img_tea; ob.fea_s = fea_stu; ob.fea_t = fea_tea; ob.col_s = col_stu; ob.col_t = col_tea; ob.rows = rows;
    ob.ray_blocks = div_up(N, kBlock / kWave);
    ob.partials = S4 + 4;  // the layout pvd_distill_loss_final reduces: S4[0..4) the sums, then one float4 per workgroup
    ob.rates_decay = rates4_decay; ob.fea_decay = fea_decay;
    const uint32_t blocks = ob.ray_blocks + ((flags & PVD_OBJECTIVE_FIXED_PARTS) ? 256u : objective_row_blocks(rows, 256u));
    hipLaunchKernelGGL((k_composite_fwd_wave<true, true>), dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, sigmas, rgbs, deltas, rays, M, N,
                       weights_sum, depth, image, ep, ob);
    return check_launch();
}

int pvd_composite_objective_backward(const float *grad_weights_sum, const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, const float *weights_sum, const float *image, uint32_t M, uint32_t N,
                                     const float *bg, float bg_scalar, float *grad_sigmas, float *grad_rgbs, uint32_t flags,
                                     const int32_t *budget_dev, const float *img_tea, const float *fea_stu, const float *fea_tea,
                                     const float *col_stu, const float *col_tea, uint32_t rows, float *coef4, const float *upstream,
                                     float *g_fea, float *g_col, const float *rates4, const float *extra, uint32_t n_extra, float *S4,
                                     float *loss, float *norms4, pvd_stream_t stream) {
    PVD_REQUIRE(N > 0 && rows > 0);
    PVD_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && image && grad_sigmas && grad_rgbs);
    PVD_REQUIRE(img_tea && fea_stu && fea_tea && col_stu && col_tea && coef4 && upstream && g_fea && g_col);
    PVD_REQUIRE(!rates4 || (S4 && loss && norms4 && (!n_extra || extra)));
    const bool fresh = (flags & PVD_MARCH_FRESH) != 0;
    const CompositeEpilogue ep{bg, bg_scalar, nullptr, nullptr, 0.f, fresh ? 1u : 0u, budget_dev};
    CompositeObjective ob{};
    ob.img_t = img_tea; ob.fea_s = fea_stu; ob.fea_t = fea_tea; ob.col_s = col_stu; ob.col_t = col_tea; ob.rows = rows;
    ob.ray_blocks = div_up(N, kBlock / kWave);
    ob.coef = coef4; ob.upstream = upstream; ob.g_fea = g_fea; ob.g_col = g_col;
    if (rates4) {  // finish the objective here: the forward launch's partial sums sit at S4 + 4
        ob.rates = rates4; ob.extra = extra; ob.n_extra = n_extra;
        ob.nparts = (flags & PVD_OBJECTIVE_FIXED_PARTS) ? pvd_composite_objective_blocks_fixed(N) : pvd_composite_objective_blocks(N, rows);
        ob.partials = S4 + 4; ob.sums = S4; ob.loss = loss; ob.coef_out = coef4; ob.norms = norms4;
    }
    const uint32_t blocks = ob.ray_blocks + objective_row_blocks(rows, 2048u);
    hipLaunchKernelGGL((k_composite_bwd_wave<true, true>), dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, grad_weights_sum,
                       (const float *)nullptr, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs, ep, ob);
    return check_launch();
}

int pvd_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                   const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                   const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                   uint32_t perturb, pvd_stream_t stream) {
    (void)nears;
    if (n_alive == 0) return PVD_OK;
    PVD_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas);
    PVD_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 1024 && max_steps >= 1);
    hipLaunchKernelGGL(k_march_rays, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n_alive, n_step, rays_alive, rays_t,
                       rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, perturb);
    return check_launch();
}

int pvd_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, float *rays_t, const float *sigmas,
                       const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image,
                       pvd_stream_t stream) {
    if (n_alive == 0) return PVD_OK;
    PVD_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image);
    hipLaunchKernelGGL(k_composite_rays, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n_alive, n_step, rays_alive,
                       rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return check_launch();
}

int pvd_compact_rays(uint32_t n_alive, int32_t *rays_alive, const int32_t *rays_alive_old, float *rays_t,
                     const float *rays_t_old, int32_t *alive_counter, pvd_stream_t stream) {
    if (n_alive == 0) return PVD_OK;
    PVD_REQUIRE(rays_alive && rays_alive_old && rays_t && rays_t_old && alive_counter);
    hipLaunchKernelGGL(k_compact_rays, dim3(div_up(n_alive, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n_alive, rays_alive,
                       rays_alive_old, rays_t, rays_t_old, alive_counter);
    return check_launch();
}

static uint32_t infer_blocks(uint32_t upper) {
    uint32_t b = div_up(upper ? upper : 1u, kBlock);
    return b < kInferBlocks ? b : kInferBlocks;
}

int pvd_infer_round_begin(int32_t *state, uint32_t parity, uint32_t N, uint32_t max_steps, pvd_stream_t stream) {
    PVD_REQUIRE(state && parity < 2 && N > 0);
    hipLaunchKernelGGL(k_infer_begin, dim3(1), dim3(64), 0, (hipStream_t)stream, (InferState *)state, parity, N, max_steps);
    return check_launch();
}

int pvd_infer_compact(int32_t *state, uint32_t parity, uint32_t n_upper, int32_t *rays_alive, const int32_t *rays_alive_old, float *rays_t,
                      const float *rays_t_old, pvd_stream_t stream) {
    PVD_REQUIRE(state && parity < 2 && rays_alive && rays_alive_old && rays_t && rays_t_old);
    hipLaunchKernelGGL(k_infer_compact, dim3(infer_blocks(n_upper)), dim3(kBlock), 0, (hipStream_t)stream, (const InferState *)state,
                       (InferState *)state, parity, rays_alive, rays_alive_old, rays_t, rays_t_old);
    return check_launch();
}

int pvd_infer_march(const int32_t *state, uint32_t n_upper, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                    const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                    const float *fars, float *xyzs, float *dirs, float *deltas, uint32_t perturb, pvd_stream_t stream) {
    PVD_REQUIRE(state && rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && deltas);
    PVD_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 1024 && max_steps >= 1);
    hipLaunchKernelGGL(k_infer_march, dim3(infer_blocks(n_upper)), dim3(kBlock), 0, (hipStream_t)stream, (const InferState *)state, rays_alive,
                       rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, perturb);
    return check_launch();
}

int pvd_infer_composite(const int32_t *state, uint32_t n_upper, const int32_t *rays_alive, float *rays_t, const float *sigmas,
                        const float *rgbs, const float *deltas, float sigma_scale, float *weights_sum, float *depth, float *image,
                        pvd_stream_t stream) {
    PVD_REQUIRE(state && rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image);
    hipLaunchKernelGGL(k_infer_composite, dim3(infer_blocks(n_upper)), dim3(kBlock), 0, (hipStream_t)stream, (const InferState *)state,
                       rays_alive, rays_t, sigmas, rgbs, deltas, sigma_scale, weights_sum, depth, image);
    return check_launch();
}

}  // extern "C"
