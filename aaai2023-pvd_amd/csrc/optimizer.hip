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
#include <stdlib.h>

namespace pvd {

constexpr uint32_t kOptBlock = 256;
constexpr uint32_t kMaxSegments = 16;

struct AdamSegments {
    uint64_t end[kMaxSegments];  // segment k covers [end[k-1], end[k]); its learning rate is lr[k] (device)
    uint32_t count;
};

struct AdamExtras {
    // learning-rate schedule evaluated on the device (so a captured step sees it without host-side scheduler ops)
    int32_t sched_kind;   // 0: lr[] used as is; 1: cosine annealing; 2: exponential decay
    float sched_T;        // T_max (cosine) / iters (exponential)
    float sched_param;    // eta_min (cosine) / total factor (exponential: lr = base * factor^min(t/T, 1))
    const float *base_lr; // DEVICE [segments]
    float *sched_step;    // DEVICE scalar: scheduler ticks so far (advanced every call, also on skipped steps)
    // L1 regulariser folded into the update: inside range r the gradient gets coef[r] * sign(p)
    uint32_t n_l1;
    uint64_t l1_begin[kMaxSegments], l1_end[kMaxSegments];
    float l1_coef[kMaxSegments];
    // a second, half-precision gradient for one range (the hash table's scatter-add result, which the reference widens and
    // adds into the fp32 gradient in a separate pass, grid.py:105-136): added while the fp32 gradient is read
    const _Float16 *g16;
    uint64_t g16_begin, g16_end;
    // value of the L1 term AFTER this update (= at the next step's forward), as one partial sum per workgroup
    float *l1_next;  // DEVICE [>= gridDim.x] or NULL
    float l1_next_scale;
    // one bit per group of 4 parameters, set = "cold": gradient and both moments are known to be zero and stay zero (rows
    // of a table no sample can reach, outside every L1 / g16 range).  The update of such a group is weight decay alone --
    // the same bits the full expression yields for g = m = v = 0 -- so g, m, v are neither read nor written for it.
    const uint32_t *cold;  // DEVICE [ceil(n / 128)] or NULL
    // Lazy weight decay of the cold groups.  Nothing reads a cold parameter while it is cold (no sample reaches it), so instead
    // of p <- p - lr wd p on every step the tail kernel LOGS the rates of the step (lazy_log[count][segment]) and
    // pvd_adamw_lazy_flush replays the logged decays one after the other, in the same arithmetic, when somebody needs the
    // values (checkpoint, occupancy change): the same bits, 0 instead of 8 B/parameter per step.
    float *lazy_log;        // DEVICE [lazy_capacity][segments] or NULL
    uint32_t *lazy_count;   // DEVICE scalar: logged steps
    uint32_t lazy_capacity;
    // with lazy_log: the groups that are NOT cold, ascending (DEVICE [n_warm]) -- the update then walks this list instead of all
    // groups: full wavefronts of warm groups instead of wavefronts in which the cold lanes idle
    const uint32_t *warm;
    uint32_t n_warm;
    uint32_t warm_nograd_from;  // list entries from here on have a structurally zero gradient (0 = none)
    // two-part update (include/pvd_hip.h): the tail records {found_inf, step before, scale before, 0, lr used ...} here
    float *snapshot;
    // fewer launches (include/pvd_hip.h): zero every visited gradient group after reading it; the last workgroup to arrive does the tail
    uint32_t zero_g;
    uint32_t *arrivals;
    float *tail_scale;      // GradScaler state for the in-kernel tail (NULL: no scaler)
    int32_t *tail_tracker;
    double tail_growth, tail_backoff;
    int32_t tail_interval;
    uint32_t tail_segments;
    // ray data parallelism (include/pvd_hip.h: compact_grad / compact_param_out / tail_clear): the gradient of warm-list entry j is
    // gc[4 j .. 4 j + 4) (the exchanged compact buffer) instead of g[4 i ..]; the updated parameters of entry j are also written to
    // pc[4 j ..] (what a sharded update all-gathers); the tail zeroes tail_clear[k * tail_clear_stride], k < tail_clear_n
    const float *gc;
    float *pc;
    float *tail_clear;
    uint32_t tail_clear_stride, tail_clear_n;
};

// The schedule, evaluated where it is needed (every workgroup of k_adamw, then once more by the tail that publishes it):
//   cosine: torch.optim.lr_scheduler.CosineAnnealingLR's closed form (main_distill_mutual.py:346-348)
//   exponential: LambdaLR(0.1 ** min(iter / iters, 1)) (main_just_train_tea.py:293-296)
__device__ __forceinline__ float scheduled_lr(const AdamExtras &ex, const float *__restrict__ lr, uint32_t k) {
    if (ex.sched_kind == 0) return lr[k];
    const double t = (double)ex.sched_step[0], base = (double)ex.base_lr[k];
    double v;
    if (ex.sched_kind == 1) v = (double)ex.sched_param + (base - (double)ex.sched_param) * (1.0 + cos(M_PI * t / (double)ex.sched_T)) * 0.5;
    else v = base * pow((double)ex.sched_param, fmin(t / (double)ex.sched_T, 1.0));
    return (float)v;
}

// One thread, AFTER the update: advance the step count (unless the GradScaler found an inf) and the scheduler tick, publish
// the learning rates the update used; then GradScaler.update() on the device (torch's amp_update_scale_cuda_kernel: back
// off on inf, grow after `interval` clean steps) and clear the inf flag for the next step -- one launch instead of a
// counting launch before the update plus four host-issued scaler ops after it.
__device__ __forceinline__ void adamw_tail_body(float *__restrict__ step, float *__restrict__ found_inf, const AdamExtras &ex,
                                                float *__restrict__ lr, uint32_t n_segments, float *__restrict__ scale,
                                                int32_t *__restrict__ tracker, double growth, double backoff, int32_t interval);

__global__ void k_adamw_tail(float *__restrict__ step, float *__restrict__ found_inf, AdamExtras ex, float *__restrict__ lr, uint32_t n_segments,
                             float *__restrict__ scale, int32_t *__restrict__ tracker, double growth, double backoff, int32_t interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    adamw_tail_body(step, found_inf, ex, lr, n_segments, scale, tracker, growth, backoff, interval);
}

__device__ __forceinline__ void adamw_tail_body(float *__restrict__ step, float *__restrict__ found_inf, const AdamExtras &ex,
                                                float *__restrict__ lr, uint32_t n_segments, float *__restrict__ scale,
                                                int32_t *__restrict__ tracker, double growth, double backoff, int32_t interval) {
    const bool inf = found_inf && found_inf[0] != 0.f;
    if (ex.snapshot) {  // what the update above used, for the deferred part of a two-part update
        ex.snapshot[0] = inf ? 1.f : 0.f;
        ex.snapshot[1] = step[0];
        ex.snapshot[2] = scale ? scale[0] : 1.f;
        ex.snapshot[3] = 0.f;
        for (uint32_t k = 0; k < n_segments; k++) ex.snapshot[4 + k] = scheduled_lr(ex, lr, k);
    }
    if (!inf) step[0] += 1.0f;
    if (ex.sched_kind != 0)
        for (uint32_t k = 0; k < n_segments; k++) lr[k] = scheduled_lr(ex, lr, k);
    if (ex.lazy_log && !inf) {  // the rates this (applied) step used, for the cold groups' deferred decay
        const uint32_t c = ex.lazy_count[0];
        if (c < ex.lazy_capacity) {
            for (uint32_t k = 0; k < n_segments; k++) ex.lazy_log[(size_t)c * n_segments + k] = lr[k];
            ex.lazy_count[0] = c + 1u;
        } else {
            ex.lazy_count[0] = 0xFFFFFFFFu;  // overflow: the host flushes long before (FlatAdamW); poisoned so that a flush fails loudly
        }
    }
    if (ex.sched_kind != 0) ex.sched_step[0] += 1.0f;
    if (!scale) {
        for (uint32_t k = 0; k < ex.tail_clear_n; k++) ex.tail_clear[(size_t)k * ex.tail_clear_stride] = 0.f;
        return;
    }
    if (inf) {
        scale[0] = (float)((double)scale[0] * backoff);
        tracker[0] = 0;
    } else {
        const int32_t ok = tracker[0] + 1;
        if (ok == interval) {
            const float ns = (float)((double)scale[0] * growth);
            if (!isinf(ns)) scale[0] = ns;
            tracker[0] = 0;
        } else {
            tracker[0] = ok;
        }
    }
    found_inf[0] = 0.f;
    for (uint32_t k = 0; k < ex.tail_clear_n; k++) ex.tail_clear[(size_t)k * ex.tail_clear_stride] = 0.f;
}

// The tail inside the update (ex.arrivals): a workgroup announces itself when it is DONE (all of them read the step's scalars
// first thing, long before); the last one to be done knows that nobody will read them again and advances them.  Relaxed
// counters: nothing but the order of each workgroup's own reads and its increment matters.  Returning atomics on ONE address
// serialise at the memory side (~12 ns each: 4096 workgroups announcing on one counter made the launch 16-20 us longer, at
// the start of the workgroups as well as at their end, profiles/r03_fold_launches_ab.txt), so the count is kept in two levels:
// kArriveGroups counters, each on a 128-byte line of its own, for the workgroups of one residue class, and a top counter
// that the last workgroup of every class bumps.  ex.arrivals = (1 + kArriveGroups) x 32 uint32, zero before the first call.
constexpr uint32_t kArriveGroups = 64, kArriveStride = 32;
__device__ __forceinline__ void adamw_announce(const AdamExtras &ex, float *__restrict__ step, float *__restrict__ found_inf, float *__restrict__ lr) {
    if (!ex.arrivals || threadIdx.x != 0) return;
    const uint32_t grp = blockIdx.x % kArriveGroups;
    const uint32_t in_grp = (gridDim.x - grp + kArriveGroups - 1) / kArriveGroups;  // workgroups of this residue class
    uint32_t *c = ex.arrivals + (size_t)(1 + grp) * kArriveStride;
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != in_grp - 1) return;
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t groups = gridDim.x < kArriveGroups ? gridDim.x : kArriveGroups;
    if (__hip_atomic_fetch_add(ex.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != groups - 1) return;
    __hip_atomic_store(ex.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    adamw_tail_body(step, found_inf, ex, lr, ex.tail_segments, ex.tail_scale, ex.tail_tracker, ex.tail_growth, ex.tail_backoff, ex.tail_interval);
}

__global__ void __launch_bounds__(kOptBlock) k_adamw(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, uint64_t n, AdamSegments seg, float *__restrict__ lr,
                                                    double beta1, double beta2, double eps, double weight_decay,
                                                    float *__restrict__ step, const float *__restrict__ grad_scale,
                                                    float *__restrict__ found_inf, AdamExtras ex) {
    // GradScaler: an inf skips the whole step (l1_next keeps describing the parameters)
    const bool skip = found_inf && found_inf[0] != 0.f;
    __shared__ float l1_sh[kOptBlock / 64];
    __shared__ float lr_sh[kMaxSegments];
    __shared__ double bc_sh[2];
    if (threadIdx.x < seg.count) lr_sh[threadIdx.x] = scheduled_lr(ex, lr, threadIdx.x);
    const double t = (double)step[0] + 1.0;  // the tail advances the stored count after the update
    const float gscale = grad_scale ? grad_scale[0] : 1.0f;
    // the bias corrections: two fp64 pow() per THREAD were ~7 us of the launch (a few hundred instructions each, every thread the
    // same two values); one lane of the last wave computes them while the first lanes evaluate the schedule
    if (threadIdx.x == kOptBlock - 1) {
        bc_sh[0] = 1.0 - pow((double)beta1, t);
        bc_sh[1] = sqrt(1.0 - pow((double)beta2, t));
    }
    __syncthreads();  // (every scalar of the step has been READ by this workgroup beyond this point)
    if (skip && !ex.zero_g && !ex.pc) { adamw_announce(ex, step, found_inf, lr); return; }
    float l1_acc = 0.f;
    const double bc1 = bc_sh[0];
    const double bc2_sqrt = bc_sh[1];
    const uint64_t n4 = ex.warm ? (uint64_t)ex.n_warm : n >> 2;
    // walking the warm list: the NEXT group's index is fetched while this group is updated (index -> data is a dependent pair of
    // round trips otherwise), and the cold bit is not looked up -- the list holds exactly the groups whose bit is clear
    const uint64_t jstride = (uint64_t)gridDim.x * kOptBlock;
    uint64_t j = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x;
    uint64_t i_next = (ex.warm && j < n4) ? (uint64_t)ex.warm[j] : j;
    const bool test_cold = ex.cold && !ex.warm;
    for (; j < n4; j += jstride) {
        const uint64_t i = i_next;
        if (j + jstride < n4) i_next = ex.warm ? (uint64_t)ex.warm[j + jstride] : j + jstride;
        const uint64_t e = i << 2;
        const bool no_grad = ex.warm_nograd_from && j >= (uint64_t)ex.warm_nograd_from;  // the buffer holds zeros there, for good
        if (skip) {  // (zero_g) the skipped step's gradients must not reach the next step
            if (ex.zero_g && !no_grad && !(test_cold && ((ex.cold[i >> 5] >> (uint32_t)(i & 31u)) & 1u))) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ex.pc) reinterpret_cast<float4 *>(ex.pc)[j] = reinterpret_cast<const float4 *>(p)[i];  // (the all-gather moves the unchanged values)
            continue;
        }
        uint32_t k = 0;
        while (k + 1 < seg.count && e >= seg.end[k]) k++;  // segments are multiples of 4 elements (see trainer)
        float l1 = 0.f;  // ranges are multiples of 4 elements too
        for (uint32_t r = 0; r < ex.n_l1; r++)
            if (e >= ex.l1_begin[r] && e < ex.l1_end[r]) l1 = ex.l1_coef[r];
        const double lrk = (double)lr_sh[k];
        if (test_cold && ((ex.cold[i >> 5] >> (uint32_t)(i & 31u)) & 1u)) {
            if (ex.lazy_log) continue;  // decay deferred: logged by the tail kernel, replayed by pvd_adamw_lazy_flush
            // g = m = v = 0: ea = es = 0, denom = eps, param -= step_size * 0 / eps leaves param as decayed below
            float4 P = reinterpret_cast<float4 *>(p)[i];
            P.x = (float)((double)P.x - lrk * (double)weight_decay * (double)P.x);
            P.y = (float)((double)P.y - lrk * (double)weight_decay * (double)P.y);
            P.z = (float)((double)P.z - lrk * (double)weight_decay * (double)P.z);
            P.w = (float)((double)P.w - lrk * (double)weight_decay * (double)P.w);
            reinterpret_cast<float4 *>(p)[i] = P;
            continue;
        }
        const float step_size = (float)(lrk / bc1);
        // the moments are touched exactly once per step: stream them past the caches (nt) so that the Infinity Cache keeps
        // the parameters and gradients the other kernels of the step come back to
        typedef float f4v __attribute__((ext_vector_type(4)));
        float4 P = reinterpret_cast<float4 *>(p)[i], G = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!no_grad) {
            G = ex.gc ? reinterpret_cast<const float4 *>(ex.gc)[j] : reinterpret_cast<const float4 *>(g)[i];
            if (ex.zero_g) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#ifndef PVD_ADAMW_NT
#define PVD_ADAMW_NT 1
#endif
#if PVD_ADAMW_NT
        const f4v Mn = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(m) + i);
        const f4v Vn = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(v) + i);
#else
        const f4v Mn = reinterpret_cast<const f4v *>(m)[i];
        const f4v Vn = reinterpret_cast<const f4v *>(v)[i];
#endif
        float4 M = make_float4(Mn.x, Mn.y, Mn.z, Mn.w), V = make_float4(Vn.x, Vn.y, Vn.z, Vn.w);
        if (ex.g16 && e >= ex.g16_begin && e < ex.g16_end) {  // 4 halfs = one 8-byte load (range start is a multiple of 4)
            typedef _Float16 h4v __attribute__((ext_vector_type(4)));
            const h4v h = *reinterpret_cast<const h4v *>(ex.g16 + (e - ex.g16_begin));
            G.x += (float)h.x; G.y += (float)h.y; G.z += (float)h.z; G.w += (float)h.w;
        }
        float *pp = &P.x, *gg = &G.x, *mm = &M.x, *vv = &V.x;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // torch's fused kernel keeps the hyper-parameters in double, so these expressions evaluate in fp64
            // and round once into the fp32 state (ATen fused_adam_utils.cuh: adam_math, ADAMW mode)
            float grad = grad_scale ? (float)((double)gg[c] / (double)gscale) : gg[c];
            if (l1 != 0.f) grad += l1 * (pp[c] > 0.f ? 1.0f : (pp[c] < 0.f ? -1.0f : 0.0f));  // d(l1 * |p|)/dp
            float param = (float)((double)pp[c] - lrk * (double)weight_decay * (double)pp[c]);
            const float ea = (float)((double)mm[c] + (1.0 - (double)beta1) * ((double)grad - (double)mm[c]));
            const float es = (float)((double)beta2 * (double)vv[c] + (1.0 - (double)beta2) * (double)grad * (double)grad);
            const float denom = (float)((double)sqrtf(es) / bc2_sqrt + (double)eps);
            param -= step_size * ea / denom;
            pp[c] = param; mm[c] = ea; vv[c] = es;
            l1_acc += l1 * fabsf(param);
        }
        reinterpret_cast<float4 *>(p)[i] = P;
        if (ex.pc) reinterpret_cast<float4 *>(ex.pc)[j] = P;
#if PVD_ADAMW_NT
        __builtin_nontemporal_store((f4v){M.x, M.y, M.z, M.w}, reinterpret_cast<f4v *>(m) + i);
        __builtin_nontemporal_store((f4v){V.x, V.y, V.z, V.w}, reinterpret_cast<f4v *>(v) + i);
#else
        reinterpret_cast<f4v *>(m)[i] = (f4v){M.x, M.y, M.z, M.w};
        reinterpret_cast<f4v *>(v)[i] = (f4v){V.x, V.y, V.z, V.w};
#endif
    }
    if (ex.l1_next && !skip) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) l1_acc += __shfl_xor(l1_acc, off, 64);
        if ((threadIdx.x & 63u) == 0) l1_sh[threadIdx.x >> 6] = l1_acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sacc = 0.f;
            for (uint32_t w = 0; w < kOptBlock / 64; w++) sacc += l1_sh[w];
            ex.l1_next[blockIdx.x] = sacc * ex.l1_next_scale;
        }
    }
    adamw_announce(ex, step, found_inf, lr);
}

// Replay of the deferred weight decay: every cold group takes the logged steps one after the other, p <- (float)(p - lr wd p)
// in the update kernel's own arithmetic (fp64 expression, one rounding per step), so the result is bit for bit what the
// per-step decay would have left.  The log is read with wave-uniform (scalar) loads.
__global__ void __launch_bounds__(kOptBlock) k_adamw_lazy_flush(float *__restrict__ p, uint64_t n, AdamSegments seg, const uint32_t *__restrict__ cold,
                                                               const float *__restrict__ log, const uint32_t *__restrict__ count,
                                                               double weight_decay) {
    const uint32_t steps = count[0];
    if (steps == 0u || steps == 0xFFFFFFFFu) return;
    const uint64_t n4 = n >> 2;
    for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kOptBlock) {
        if (!((cold[i >> 5] >> (uint32_t)(i & 31u)) & 1u)) continue;
        const uint64_t e = i << 2;
        uint32_t k = 0;
        while (k + 1 < seg.count && e >= seg.end[k]) k++;
        float4 P = reinterpret_cast<float4 *>(p)[i];
        for (uint32_t s = 0; s < steps; s++) {
            const double f = (double)log[(size_t)s * seg.count + k] * weight_decay;
            P.x = (float)((double)P.x - f * (double)P.x);
            P.y = (float)((double)P.y - f * (double)P.y);
            P.z = (float)((double)P.z - f * (double)P.z);
            P.w = (float)((double)P.w - f * (double)P.w);
        }
        reinterpret_cast<float4 *>(p)[i] = P;
    }
}
__global__ void k_adamw_lazy_reset(uint32_t *__restrict__ count, int32_t *__restrict__ status) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (status) status[0] = count[0] == 0xFFFFFFFFu ? -1 : (int32_t)count[0];
        count[0] = 0u;
    }
}

// found_inf[0] = 1 if any element is inf / nan (never cleared here): the read-only half of
// torch._amp_foreach_non_finite_check_and_unscale_, which GradScaler.step runs with a scale of 1 for optimizers
// that unscale inside their own kernel.
__global__ void __launch_bounds__(kOptBlock) k_check_finite_f16(const _Float16 *__restrict__ g, uint64_t n8, float *__restrict__ found_inf) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * kOptBlock) {
        const uint4 w = reinterpret_cast<const uint4 *>(g)[i];  // 8 halfs; exponent field 0x7c00 all ones = inf / nan
        const uint32_t q[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; k++) bad |= ((q[k] & 0x7c00u) == 0x7c00u) | ((q[k] & 0x7c000000u) == 0x7c000000u);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0) found_inf[0] = 1.0f;
}

__global__ void __launch_bounds__(kOptBlock) k_check_finite(const float *__restrict__ g, uint64_t n4, float *__restrict__ found_inf) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * kOptBlock) {
        const float4 G = reinterpret_cast<const float4 *>(g)[i];
        // finite <=> exponent field not all ones
        const uint32_t a = __float_as_uint(G.x), b = __float_as_uint(G.y), c = __float_as_uint(G.z), d = __float_as_uint(G.w);
        bad |= ((a & 0x7f800000u) == 0x7f800000u) | ((b & 0x7f800000u) == 0x7f800000u) | ((c & 0x7f800000u) == 0x7f800000u) |
               ((d & 0x7f800000u) == 0x7f800000u);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0) found_inf[0] = 1.0f;
}

// both of the above in ONE launch, for a flat fp32 gradient one of whose ranges lives in a half-precision buffer instead (the hash
// table's scatter-add result, pvd_adamw_extras.g16): the fp32 groups outside [skip_b4, skip_e4), then the half groups
__global__ void __launch_bounds__(kOptBlock) k_check_finite_mixed(const float *__restrict__ g, uint64_t n4, uint64_t skip_b4, uint64_t skip_e4,
                                                                 const _Float16 *__restrict__ g16, uint64_t n8, float *__restrict__ found_inf) {
    const uint64_t gap = skip_e4 - skip_b4, n4_eff = n4 - gap;
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < n4_eff + n8; i += (uint64_t)gridDim.x * kOptBlock) {
        if (i < n4_eff) {
            const float4 G = reinterpret_cast<const float4 *>(g)[i < skip_b4 ? i : i + gap];
            const uint32_t a = __float_as_uint(G.x), b = __float_as_uint(G.y), c = __float_as_uint(G.z), d = __float_as_uint(G.w);
            bad |= ((a & 0x7f800000u) == 0x7f800000u) | ((b & 0x7f800000u) == 0x7f800000u) | ((c & 0x7f800000u) == 0x7f800000u) |
                   ((d & 0x7f800000u) == 0x7f800000u);
        } else {
            const uint4 w = reinterpret_cast<const uint4 *>(g16)[i - n4_eff];
            const uint32_t q[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; k++) bad |= ((q[k] & 0x7c00u) == 0x7c00u) | ((q[k] & 0x7c000000u) == 0x7c000000u);
        }
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0) found_inf[0] = 1.0f;
}

// ---- segment-table operations over the flat gradient buffer.  The rows of a VM plane / Plenoxel volume that can receive a
// gradient at all are known from the occupancy grid (harness: pvd/dp_compact.py); zeroing, the inf check and the ray-DP
// gather/scatter then only touch those rows.  segs[s] = {start (flat), dst (compact), len}, one workgroup per segment.
// 5 (ray-DP, what goes on the wire): gather, ZERO the source behind it (the next step's zero_grad), and look at what it moves -- a
//    workgroup that sees an inf / nan stores 1 into found_inf[k * slot_stride], k < n_slots (one flag word per chunk of the exchange
//    buffer: summed over the ranks by the collective itself, they are the step's global found_inf on every rank)
enum { kSegZero = 0, kSegGather = 1, kSegScatter = 2, kSegCheck = 3, kSegScatterCheck = 4, kSegGatherZeroCheck = 5 };  // 4: scatter, looking at what it moves
template <int OP>
__global__ void __launch_bounds__(kOptBlock) k_segments(float *__restrict__ flat, float *__restrict__ buf, const uint32_t *__restrict__ segs,
                                                        uint32_t n_segs, float *__restrict__ found_inf, uint32_t slot_stride = 0, uint32_t n_slots = 1) {
    bool bad = false;
    for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
        const uint32_t start = segs[3 * s], dst = segs[3 * s + 1], len = segs[3 * s + 2];
        if (((start | dst | len) & 3u) == 0) {
            float4 *f4 = reinterpret_cast<float4 *>(flat + start);
            float4 *b4 = reinterpret_cast<float4 *>(buf + dst);
            for (uint32_t i = threadIdx.x; i < (len >> 2); i += kOptBlock) {
                if (OP == kSegZero) f4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (OP == kSegGather) b4[i] = f4[i];
                if (OP == kSegScatter) f4[i] = b4[i];
                if (OP == kSegGatherZeroCheck) {
                    const float4 G = f4[i];
                    b4[i] = G;
                    f4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const uint32_t a = __float_as_uint(G.x), b = __float_as_uint(G.y), c = __float_as_uint(G.z), d = __float_as_uint(G.w);
                    bad |= ((a & 0x7f800000u) == 0x7f800000u) | ((b & 0x7f800000u) == 0x7f800000u) | ((c & 0x7f800000u) == 0x7f800000u) |
                           ((d & 0x7f800000u) == 0x7f800000u);
                }
                if (OP == kSegCheck || OP == kSegScatterCheck) {
                    const float4 G = OP == kSegCheck ? f4[i] : b4[i];
                    if (OP == kSegScatterCheck) f4[i] = G;
                    const uint32_t a = __float_as_uint(G.x), b = __float_as_uint(G.y), c = __float_as_uint(G.z), d = __float_as_uint(G.w);
                    bad |= ((a & 0x7f800000u) == 0x7f800000u) | ((b & 0x7f800000u) == 0x7f800000u) | ((c & 0x7f800000u) == 0x7f800000u) |
                           ((d & 0x7f800000u) == 0x7f800000u);
                }
            }
        } else {
            for (uint32_t i = threadIdx.x; i < len; i += kOptBlock) {
                if (OP == kSegZero) flat[start + i] = 0.f;
                if (OP == kSegGather) buf[dst + i] = flat[start + i];
                if (OP == kSegScatter) flat[start + i] = buf[dst + i];
                if (OP == kSegCheck) bad |= (__float_as_uint(flat[start + i]) & 0x7f800000u) == 0x7f800000u;
                if (OP == kSegScatterCheck) {
                    const float v = buf[dst + i];
                    flat[start + i] = v;
                    bad |= (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
                }
                if (OP == kSegGatherZeroCheck) {
                    const float v = flat[start + i];
                    buf[dst + i] = v;
                    flat[start + i] = 0.f;
                    bad |= (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u;
                }
            }
        }
    }
    if ((OP == kSegCheck || OP == kSegScatterCheck) && __ballot(bad) != 0ull && (threadIdx.x & 63u) == 0) found_inf[0] = 1.0f;
    if (OP == kSegGatherZeroCheck && __ballot(bad) != 0ull && (threadIdx.x & 63u) == 0)
        for (uint32_t k = 0; k < n_slots; k++) found_inf[(size_t)k * slot_stride] = 1.0f;
}

// partials[block] = sum over the block's elements of coef[range] * |p|
__global__ void __launch_bounds__(kOptBlock) k_l1_partial(const float *__restrict__ p, AdamExtras ex, float *__restrict__ partials) {
    __shared__ float red[kOptBlock / 64];
    float acc = 0.f;
    for (uint32_t r = 0; r < ex.n_l1; r++) {
        const uint64_t b4 = ex.l1_begin[r] >> 2, e4 = ex.l1_end[r] >> 2;
        float a = 0.f;
        for (uint64_t i = b4 + (uint64_t)blockIdx.x * kOptBlock + threadIdx.x; i < e4; i += (uint64_t)gridDim.x * kOptBlock) {
            const float4 P = reinterpret_cast<const float4 *>(p)[i];
            a += (fabsf(P.x) + fabsf(P.y)) + (fabsf(P.z) + fabsf(P.w));
        }
        acc += ex.l1_coef[r] * a;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (uint32_t w = 0; w < kOptBlock / 64; w++) s += red[w];
        partials[blockIdx.x] = s;
    }
}
__global__ void __launch_bounds__(kOptBlock) k_l1_final(const float *__restrict__ partials, uint32_t n, float *__restrict__ out) {
    __shared__ float red[kOptBlock / 64];
    float acc = 0.f;
    for (uint32_t i = threadIdx.x; i < n; i += kOptBlock) acc += partials[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (uint32_t w = 0; w < kOptBlock / 64; w++) s += red[w];
        out[0] = s;
    }
}

constexpr uint32_t kL1Blocks = 1024;

static int fill_l1(AdamExtras &ex, const uint64_t *begin, const uint64_t *end, const float *coef, uint32_t n) {
    if (n > kMaxSegments || (n && (!begin || !end || !coef))) return PVD_ERR_INVALID;
    ex.n_l1 = n;
    for (uint32_t r = 0; r < n; r++) {
        if ((begin[r] & 3u) || (end[r] & 3u) || end[r] < begin[r]) return PVD_ERR_UNSUPPORTED;
        ex.l1_begin[r] = begin[r]; ex.l1_end[r] = end[r]; ex.l1_coef[r] = coef[r];
    }
    return PVD_OK;
}

}  // namespace pvd

using namespace pvd;

extern "C" {

int pvd_adamw_step_ex(float *p, const float *g, float *m, float *v, uint64_t n, const uint64_t *segment_ends_host, uint32_t n_segments,
                      float *lr, double beta1, double beta2, double eps, double weight_decay, float *step, const float *grad_scale,
                      const float *found_inf, const pvd_adamw_extras *extras_host, pvd_stream_t stream) {
    if (n == 0) return PVD_OK;
    if (!p || !g || !m || !v || !segment_ends_host || !lr || !step) return PVD_ERR_INVALID;
    if (n_segments < 1 || n_segments > kMaxSegments || (n & 3u)) return PVD_ERR_UNSUPPORTED;
    AdamSegments seg;
    seg.count = n_segments;
    for (uint32_t k = 0; k < n_segments; k++) {
        if (segment_ends_host[k] & 3u) return PVD_ERR_UNSUPPORTED;
        seg.end[k] = segment_ends_host[k];
    }
    AdamExtras ex;
    ex.sched_kind = 0; ex.sched_T = 1.f; ex.sched_param = 0.f; ex.base_lr = nullptr; ex.sched_step = nullptr; ex.n_l1 = 0;
    ex.g16 = nullptr; ex.g16_begin = ex.g16_end = 0; ex.l1_next = nullptr; ex.l1_next_scale = 1.f; ex.cold = nullptr;
    ex.lazy_log = nullptr; ex.lazy_count = nullptr; ex.lazy_capacity = 0; ex.warm = nullptr; ex.n_warm = 0; ex.warm_nograd_from = 0; ex.snapshot = nullptr;
    ex.zero_g = 0; ex.arrivals = nullptr; ex.tail_scale = nullptr; ex.tail_tracker = nullptr; ex.tail_growth = ex.tail_backoff = 0.0;
    ex.tail_interval = 1; ex.tail_segments = n_segments;
    ex.gc = nullptr; ex.pc = nullptr; ex.tail_clear = nullptr; ex.tail_clear_stride = 0; ex.tail_clear_n = 0;
    const float *replay = nullptr;
    if (extras_host) {
        const pvd_adamw_extras &h = *extras_host;
        if (h.sched_kind < 0 || h.sched_kind > 2) return PVD_ERR_UNSUPPORTED;
        if (h.sched_kind != 0 && (!h.base_lr || !h.sched_step || !(h.sched_T > 0.f))) return PVD_ERR_INVALID;
        ex.sched_kind = h.sched_kind; ex.sched_T = h.sched_T; ex.sched_param = h.sched_param;
        ex.base_lr = h.base_lr; ex.sched_step = h.sched_step;
        const int rc = fill_l1(ex, h.l1_begin_host, h.l1_end_host, h.l1_coef_host, h.n_l1);
        if (rc != PVD_OK) return rc;
        if (h.l1_next) { ex.l1_next = h.l1_next; ex.l1_next_scale = h.l1_next_scale; }
        ex.cold = h.cold_bits;
        if (h.lazy_log) {
            if (!h.cold_bits || !h.lazy_count || h.lazy_capacity < 1) return PVD_ERR_INVALID;
            ex.lazy_log = h.lazy_log; ex.lazy_count = h.lazy_count; ex.lazy_capacity = h.lazy_capacity;
            if (h.warm_groups) {
                ex.warm = h.warm_groups; ex.n_warm = h.n_warm_groups;
                ex.warm_nograd_from = (h.warm_zero_grad_from && h.warm_zero_grad_from < h.n_warm_groups) ? h.warm_zero_grad_from : 0u;
            }
        } else if (h.warm_groups) {
            return PVD_ERR_INVALID;  // without the deferred decay the cold groups need their update every step
        }
        if (h.g16) {
            if ((h.g16_begin & 3u) || (h.g16_end & 3u) || h.g16_end < h.g16_begin || h.g16_end > n) return PVD_ERR_UNSUPPORTED;
            ex.g16 = (const _Float16 *)h.g16; ex.g16_begin = h.g16_begin; ex.g16_end = h.g16_end;
        }
        ex.snapshot = h.snapshot;
        replay = h.replay;
        if (replay && (!ex.warm || h.snapshot || h.g16 || h.arrivals)) return PVD_ERR_INVALID;  // the deferred part walks a list and records nothing
        ex.zero_g = h.zero_grad_after ? 1u : 0u;
        ex.arrivals = h.arrivals;
        if (h.compact_grad || h.compact_param_out) {
            // list entry j <-> compact group j: only a walk of a warm list can use them, and not the deferred part (it has no gradient)
            if (!ex.warm || replay || ex.warm_nograd_from || h.g16) return PVD_ERR_INVALID;
            ex.gc = h.compact_grad; ex.pc = h.compact_param_out;
        }
        if (h.tail_clear_n) {
            if (!h.tail_clear || replay) return PVD_ERR_INVALID;
            ex.tail_clear = h.tail_clear; ex.tail_clear_stride = h.tail_clear_stride; ex.tail_clear_n = h.tail_clear_n;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (replay) {
        // the deferred part: the scalars of the recorded step (found_inf, step count, scale, learning rates as published), no tail
        ex.sched_kind = 0;
        uint64_t blocks_a = ((uint64_t)ex.n_warm + kOptBlock - 1) / kOptBlock;
        // the deferred part runs NEXT TO other kernels: 512 workgroups leave them their wave slots (0.2803 vs 0.2830 ms/step with
        // 4096, profiles/r04_adamw_late_ab.txt)
        if (blocks_a > 512u) blocks_a = 512u;
        if (blocks_a < 1) blocks_a = 1;
        hipLaunchKernelGGL(k_adamw, dim3((uint32_t)blocks_a), dim3(kOptBlock), 0, s, p, const_cast<float *>(g), m, v, n, seg,
                           const_cast<float *>(replay + 4), beta1, beta2, eps, weight_decay, const_cast<float *>(replay + 1),
                           grad_scale ? replay + 2 : nullptr, const_cast<float *>(replay), ex);
        return check_launch();
    }
    uint64_t blocks = ((ex.warm ? (uint64_t)ex.n_warm : n / 4) + kOptBlock - 1) / kOptBlock;
    // 16 workgroups per CU; every workgroup writes one partial into l1_next[blockIdx.x], whose contract is ">= 4096 floats" (pvd_hip.h)
    if (blocks > 4096u) blocks = 4096u;
    if (blocks < 1) blocks = 1;
    const bool amp = extras_host && extras_host->amp_scale;
    if (amp && (!extras_host->amp_growth_tracker || !found_inf || extras_host->amp_interval < 1)) return PVD_ERR_INVALID;
    if (ex.arrivals) {  // the tail runs inside the update (last workgroup to arrive)
        ex.tail_scale = amp ? extras_host->amp_scale : nullptr; ex.tail_tracker = amp ? extras_host->amp_growth_tracker : nullptr;
        ex.tail_growth = amp ? extras_host->amp_growth : 0.0; ex.tail_backoff = amp ? extras_host->amp_backoff : 0.0;
        ex.tail_interval = amp ? extras_host->amp_interval : 1; ex.tail_segments = n_segments;
    }
    hipLaunchKernelGGL(k_adamw, dim3((uint32_t)blocks), dim3(kOptBlock), 0, s, p, const_cast<float *>(g), m, v, n, seg, lr, beta1, beta2, eps,
                       weight_decay, step, grad_scale, const_cast<float *>(found_inf), ex);
    if (ex.arrivals) return check_launch();
    hipLaunchKernelGGL(k_adamw_tail, dim3(1), dim3(64), 0, s, step, const_cast<float *>(found_inf), ex, lr, n_segments,
                       amp ? extras_host->amp_scale : nullptr, amp ? extras_host->amp_growth_tracker : nullptr,
                       amp ? extras_host->amp_growth : 0.0, amp ? extras_host->amp_backoff : 0.0, amp ? extras_host->amp_interval : 1);
    return check_launch();
}

int pvd_adamw_lazy_flush(float *p, uint64_t n, const uint64_t *segment_ends_host, uint32_t n_segments, const uint32_t *cold_bits,
                         const float *lazy_log, uint32_t *lazy_count, double weight_decay, int32_t *status, pvd_stream_t stream) {
    if (!p || !segment_ends_host || !cold_bits || !lazy_log || !lazy_count) return PVD_ERR_INVALID;
    if (n_segments < 1 || n_segments > kMaxSegments || (n & 3u)) return PVD_ERR_UNSUPPORTED;
    AdamSegments seg;
    seg.count = n_segments;
    for (uint32_t k = 0; k < n_segments; k++) seg.end[k] = segment_ends_host[k];
    hipStream_t s = (hipStream_t)stream;
    uint64_t blocks = (n / 4 + kOptBlock - 1) / kOptBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (n) hipLaunchKernelGGL(k_adamw_lazy_flush, dim3((uint32_t)blocks), dim3(kOptBlock), 0, s, p, n, seg, cold_bits, lazy_log, lazy_count, weight_decay);
    hipLaunchKernelGGL(k_adamw_lazy_reset, dim3(1), dim3(64), 0, s, lazy_count, status);
    return check_launch();
}

int pvd_adamw_step(float *p, const float *g, float *m, float *v, uint64_t n, const uint64_t *segment_ends_host, uint32_t n_segments,
                   const float *lr, double beta1, double beta2, double eps, double weight_decay, float *step, const float *grad_scale,
                   const float *found_inf, pvd_stream_t stream) {
    return pvd_adamw_step_ex(p, g, m, v, n, segment_ends_host, n_segments, const_cast<float *>(lr), beta1, beta2, eps, weight_decay, step,
                             grad_scale, found_inf, nullptr, stream);
}

int pvd_check_finite(const float *g, uint64_t n, float *found_inf, pvd_stream_t stream) {
    if (n == 0) return PVD_OK;
    if (!g || !found_inf) return PVD_ERR_INVALID;
    if (n & 3u) return PVD_ERR_UNSUPPORTED;
    uint64_t blocks = (n / 4 + kOptBlock - 1) / kOptBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_check_finite, dim3((uint32_t)blocks), dim3(kOptBlock), 0, (hipStream_t)stream, g, n >> 2, found_inf);
    return check_launch();
}

int pvd_check_finite_f16(const void *g, uint64_t n, float *found_inf, pvd_stream_t stream) {
    if (n == 0) return PVD_OK;
    if (!g || !found_inf) return PVD_ERR_INVALID;
    if (n & 7u) return PVD_ERR_UNSUPPORTED;
    uint64_t blocks = (n / 8 + kOptBlock - 1) / kOptBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_check_finite_f16, dim3((uint32_t)blocks), dim3(kOptBlock), 0, (hipStream_t)stream, (const _Float16 *)g, n >> 3,
                       found_inf);
    return check_launch();
}

int pvd_check_finite_mixed(const float *g, uint64_t n, uint64_t skip_begin, uint64_t skip_end, const void *g16, uint64_t n16, float *found_inf,
                           pvd_stream_t stream) {
    if (!g || !g16 || !found_inf) return PVD_ERR_INVALID;
    if ((n & 3u) || (skip_begin & 3u) || (skip_end & 3u) || (n16 & 7u) || skip_begin > skip_end || skip_end > n) return PVD_ERR_UNSUPPORTED;
    const uint64_t items = (n - (skip_end - skip_begin)) / 4 + n16 / 8;
    if (items == 0) return PVD_OK;
    uint64_t blocks = (items + kOptBlock - 1) / kOptBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_check_finite_mixed, dim3((uint32_t)blocks), dim3(kOptBlock), 0, (hipStream_t)stream, g, n >> 2, skip_begin >> 2,
                       skip_end >> 2, (const _Float16 *)g16, n16 >> 3, found_inf);
    return check_launch();
}

int pvd_segments_gather_zero_check(float *flat, float *buf, const uint32_t *segs, uint32_t n_segs, float *slots, uint32_t slot_stride,
                                   uint32_t n_slots, pvd_stream_t stream) {
    if (n_segs == 0) return PVD_OK;
    if (!flat || !buf || !segs || !slots || n_slots < 1 || n_slots > 64) return PVD_ERR_INVALID;
    const dim3 grid(n_segs < 65535u * 16u ? n_segs : 65535u * 16u), block(kOptBlock);
    hipLaunchKernelGGL(k_segments<kSegGatherZeroCheck>, grid, block, 0, (hipStream_t)stream, flat, buf, segs, n_segs, slots, slot_stride, n_slots);
    return check_launch();
}

int pvd_segments_op(int op, float *flat, float *buf, const uint32_t *segs, uint32_t n_segs, float *found_inf, pvd_stream_t stream) {
    if (op < kSegZero || op > kSegScatterCheck) return PVD_ERR_INVALID;
    if (n_segs == 0) return PVD_OK;
    if (!flat || !segs) return PVD_ERR_INVALID;
    if ((op == kSegGather || op == kSegScatter || op == kSegScatterCheck) && !buf) return PVD_ERR_INVALID;
    if ((op == kSegCheck || op == kSegScatterCheck) && !found_inf) return PVD_ERR_INVALID;
    const dim3 grid(n_segs < 65535u * 16u ? n_segs : 65535u * 16u), block(kOptBlock);
    hipStream_t s = (hipStream_t)stream;
    switch (op) {
        case kSegZero: hipLaunchKernelGGL(k_segments<kSegZero>, grid, block, 0, s, flat, buf, segs, n_segs, found_inf, 0u, 1u); break;
        case kSegGather: hipLaunchKernelGGL(k_segments<kSegGather>, grid, block, 0, s, flat, buf, segs, n_segs, found_inf, 0u, 1u); break;
        case kSegScatter: hipLaunchKernelGGL(k_segments<kSegScatter>, grid, block, 0, s, flat, buf, segs, n_segs, found_inf, 0u, 1u); break;
        case kSegScatterCheck: hipLaunchKernelGGL(k_segments<kSegScatterCheck>, grid, block, 0, s, flat, buf, segs, n_segs, found_inf, 0u, 1u); break;
        default: hipLaunchKernelGGL(k_segments<kSegCheck>, grid, block, 0, s, flat, buf, segs, n_segs, found_inf, 0u, 1u); break;
    }
    return check_launch();
}

int pvd_l1_ranges(const float *p, const uint64_t *begin_host, const uint64_t *end_host, const float *coef_host, uint32_t n_ranges,
                  float *scratch, float *out, pvd_stream_t stream) {
    if (!p || !scratch) return PVD_ERR_INVALID;
    AdamExtras ex;
    ex.sched_kind = 0; ex.sched_T = 1.f; ex.sched_param = 0.f; ex.base_lr = nullptr; ex.sched_step = nullptr;
    ex.g16 = nullptr; ex.g16_begin = ex.g16_end = 0; ex.l1_next = nullptr; ex.l1_next_scale = 1.f; ex.cold = nullptr;
    ex.lazy_log = nullptr; ex.lazy_count = nullptr; ex.lazy_capacity = 0; ex.warm = nullptr; ex.n_warm = 0; ex.warm_nograd_from = 0; ex.snapshot = nullptr;
    ex.zero_g = 0; ex.arrivals = nullptr; ex.tail_scale = nullptr; ex.tail_tracker = nullptr; ex.tail_growth = ex.tail_backoff = 0.0;
    ex.tail_interval = 1; ex.tail_segments = 0;
    ex.gc = nullptr; ex.pc = nullptr; ex.tail_clear = nullptr; ex.tail_clear_stride = 0; ex.tail_clear_n = 0;
    const int rc = fill_l1(ex, begin_host, end_host, coef_host, n_ranges);
    if (rc != PVD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_l1_partial, dim3(kL1Blocks), dim3(kOptBlock), 0, s, p, ex, scratch);
    if (out) hipLaunchKernelGGL(k_l1_final, dim3(1), dim3(kOptBlock), 0, s, scratch, kL1Blocks, out);
    return check_launch();
}

}  // extern "C"
