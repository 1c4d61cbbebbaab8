// infer_persistent.h -- what the persistent inference renders share (fusedhead.hip: hash and VM models; plenoxel.hip: the dense
// volume): the launch's argument block, the host-side preparation (first-hit pass + ray queue), and the slot machinery of a
// workgroup -- refill / march / blend around a model-specific shading of the round's sample rows -- as a template
// (`infer_persistent_loop`).  k_infer_px_persistent is written on the template; the hash and VM kernels, which came first and are
// tuned around their LDS tiles, carry the same loop inline.  Reference: the eval branch of run_cuda, distill_mutual/renderer.py:450-543.
#pragma once

#include "dda.h"
#include "pvd_device.h"

namespace pvd {

struct InferImageArgs {
    const float *rays_o, *rays_d;  // [N][3]
    const float *nears, *fars;     // [N]
    const int32_t *ray_ids;        // [*n_ids] rays that meet an occupied cell, any order
    const float *t_first;          // [N] the marcher's t at the ray's first occupied probe
    const int32_t *n_ids;          // device count
    int32_t *queue;                // device counter, zero at launch
    uint32_t shuffle;              // multiplier of the queue -> ray permutation (host: PVD_INFER_SHUFFLE, default 7919; 1 = image order)
    int32_t *stats;                // [4] zero at launch: local rounds, rows shaded, walk-only rounds, workgroups that took rays
    const uint8_t *grid;           // density bitfield
    float bound, dt_gamma, sigma_scale;
    uint32_t max_steps, C, H;
    float *weights_sum, *depth, *image;  // [N], [N], [N][3]: written for the rays in ray_ids (zero-filled by the caller)
};

constexpr uint32_t kInfRows = 256;         // sample rows per local round (LDS tile): the most any variant uses
constexpr uint32_t kInfSteps = 8;          // samples a slot may hold per round (the reference's cap on n_step, renderer.py:493)
constexpr uint32_t kInfProbes = 6;         // probes per slot and round beyond the samples it is looking for

// host (fusedhead.hip): zero the workspace's counters, fill `q` and launch the first-hit pass that builds the ray queue.
// workspace: [0] number of rays that meet an occupied cell, [1] the queue's head, [2 .. 2 + N) their ids, [2 + N .. 2 + 2 N) t_first
// (float), [2 + 2 N .. 12 + 2 N) statistics of the launch.
int infer_prepare(InferImageArgs &q, const float *rays_o, const float *rays_d, const float *nears, const float *fars, uint32_t N,
                  const uint8_t *bitfield, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, float sigma_scale,
                  int32_t *workspace, float *weights_sum, float *depth, float *image_out, hipStream_t s);

// LDS of the slot machinery: sample rows of the round and what the shading leaves for the blend
struct InferTile {
    float *pos;          // [ROWS][3]
    float *sig;          // [ROWS]
    float *rgb;          // [ROWS][3]
    float *sdir;         // [BLOCK][3]: direction of the ray in slot s
    uint32_t *row_slot;  // [ROWS]
    uint32_t *wcnt;      // [8] scan scratch + queue hand-off
    template <uint32_t ROWS, uint32_t BLOCK>
    __device__ __forceinline__ void carve(float *base) {
        pos = base; sig = pos + 3 * ROWS; rgb = sig + ROWS; sdir = rgb + 3 * ROWS;
        row_slot = reinterpret_cast<uint32_t *>(sdir + 3 * BLOCK); wcnt = row_slot + ROWS;
    }
    template <uint32_t ROWS, uint32_t BLOCK>
    static constexpr uint32_t floats() { return 3 * ROWS + ROWS + 3 * ROWS + 3 * BLOCK + ROWS + 8; }
};

// The local rounds of one workgroup of BLOCK threads owning RAYS ray slots (threads 0 .. RAYS - 1 one each; the ray's march position
// and accumulators in registers).  shade(rows): called by every thread once the round's `rows` sample rows are in T.pos / T.row_slot
// (barrier passed); leaves T.sig / T.rgb of those rows; the loop puts the barrier behind it.  Per ray the walk is the reference's walk
// paused and resumed and the sums are the reference's sums in its order (k_march_rays / k_composite_rays, raymarching.cu:756-899).
template <uint32_t RAYS, uint32_t ROWS, uint32_t BLOCK, class Shade>
__device__ __forceinline__ void infer_persistent_loop(const InferImageArgs &q, const InferTile &T, Shade shade) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_ids = (uint32_t)max(*q.n_ids, 0);
    // queue position -> ray: position * mul mod n_ids is a permutation only for a multiplier coprime to n_ids
    uint32_t mul = q.shuffle % max(n_ids, 1u);
    for (;; mul++) {
        uint32_t x = max(mul, 1u), y = max(n_ids, 1u);
        while (y) { const uint32_t r = x % y; x = y; y = r; }
        if (x == 1u || n_ids <= 1u) break;
    }
    mul = max(mul, 1u);

    // the slot's ray: t = what compositing has reached (the reference's rays_t / last_t), tt = where the walk stands, and the `cnt`
    // samples the walk has found since the last blend (position, dt, the walk's t behind the sample)
    int32_t index = -1;
    uint32_t taken = 0, cnt = 0;
    float t = 0.f, tt = 0.f, far = 0.f, ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    float sx[kInfSteps], sy[kInfSteps], sz[kInfSteps], sdt[kInfSteps], stt[kInfSteps];
    float ro[3] = {0.f, 0.f, 0.f}, rd[3] = {0.f, 0.f, 1.f};
    bool queue_done = n_ids == 0;

    auto scan_of = [&](uint32_t v, uint32_t &total) -> uint32_t {  // exclusive prefix over the workgroup's threads and the total
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
            if ((int)lane >= d) inc += up;
        }
        if (lane == 63) T.wcnt[wave] = inc;
        __syncthreads();
        uint32_t before = 0, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < BLOCK / 64; w++) { const uint32_t c = T.wcnt[w]; before += w < wave ? c : 0u; tot += c; }
        __syncthreads();
        total = tot;
        return before + inc - v;
    };
    auto retire = [&]() {  // the ray is done: its pixel leaves the slot
        q.weights_sum[index] = ws;
        q.depth[index] = dep;
        q.image[3 * (size_t)index] = cr; q.image[3 * (size_t)index + 1] = cg; q.image[3 * (size_t)index + 2] = cb;
        index = -1; cnt = 0;
    };

    uint32_t n_rounds = 0, n_rows = 0, n_walk = 0;
    for (;;) {
        // ---------------- refill: free slots take the next rays (when enough of them are free to be worth the round trip)
        uint32_t nfree;
        const uint32_t frank = scan_of(tid < RAYS && index < 0 ? 1u : 0u, nfree);
        if (!queue_done && (nfree >= RAYS / 8 || nfree == RAYS)) {
            if (tid == 0) T.wcnt[4] = (uint32_t)atomicAdd(q.queue, (int32_t)nfree);
            __syncthreads();
            const uint32_t base = T.wcnt[4];
            __syncthreads();
            if (base + nfree >= n_ids) queue_done = true;
            if (tid < RAYS && index < 0 && base + frank < n_ids) {
                const int32_t id = q.ray_ids[(uint32_t)(((uint64_t)(base + frank) * mul) % n_ids)];
                index = id; taken = 0; cnt = 0;
#pragma unroll
                for (int c = 0; c < 3; c++) { ro[c] = q.rays_o[3 * (size_t)id + c]; rd[c] = q.rays_d[3 * (size_t)id + c]; T.sdir[3 * tid + c] = rd[c]; }
                t = q.nears[id]; far = q.fars[id]; tt = q.t_first[id];
                ws = dep = cr = cg = cb = 0.f;
            }
        }
        uint32_t live_now;
        (void)scan_of(index >= 0 ? 1u : 0u, live_now);
        if (live_now == 0) {
            if (queue_done) break;
            continue;
        }
        const uint32_t n_step = max(min(ROWS / live_now, kInfSteps), 1u);  // the reference's rule (renderer.py:493), the workgroup's numbers
        // ---------------- march (k_march_rays' loop, raymarching.cu:756-810, perturb = 0)
        if (index >= 0) {
            Dda r;
            r.init(ro, rd, q.bound, q.dt_gamma, q.max_steps, q.C, q.H, q.grid);
            for (uint32_t pb = 0; pb < n_step + kInfProbes && cnt < n_step && tt < far; pb++) {
                float x, y, z, dt, tn;
                if (r.probe(tt, x, y, z, dt, tn)) {
                    tt += dt;
#pragma unroll
                    for (uint32_t k = 0; k < kInfSteps; k++)
                        if (k == cnt) { sx[k] = x; sy[k] = y; sz[k] = z; sdt[k] = dt; stt[k] = tt; }
                    cnt++;
                } else {
                    tt = tn;
                }
            }
            if (cnt == 0 && !(tt < far)) retire();
        }
        uint32_t rows;
        const uint32_t row0 = scan_of(cnt, rows);
        n_rounds++; n_rows += rows;
        if (rows == 0) { n_walk++; continue; }
#pragma unroll
        for (uint32_t k = 0; k < kInfSteps; k++)
            if (k < cnt) {
                T.pos[3 * (row0 + k)] = sx[k]; T.pos[3 * (row0 + k) + 1] = sy[k]; T.pos[3 * (row0 + k) + 2] = sz[k];
                T.row_slot[row0 + k] = tid;
            }
        __syncthreads();
        // ---------------- shade
        shade(rows);
        __syncthreads();
        // ---------------- blend (k_composite_rays' loop, raymarching.cu:858-899; sigma scaled as renderer.py:528)
        if (cnt > 0) {
            bool done = false;
#pragma unroll
            for (uint32_t k = 0; k < kInfSteps; k++) {
                if (k < cnt && !done) {
                    const uint32_t rw = row0 + k;
                    const float alpha = 1.0f - __expf(-(q.sigma_scale * T.sig[rw]) * sdt[k]);
                    const float Tr = 1 - ws;
                    const float w = alpha * Tr;
                    ws += w;
                    t += stt[k] - t;
                    dep += w * t;
                    cr += w * T.rgb[3 * rw]; cg += w * T.rgb[3 * rw + 1]; cb += w * T.rgb[3 * rw + 2];
                    taken++;
                    if ((double)Tr < 1e-4 || taken >= q.max_steps) done = true;
                }
            }
            cnt = 0;
            if (done) retire();
        }
        __syncthreads();  // the next round rewrites the tile
    }
    if (tid == 0 && n_rounds > 1) {
        atomicAdd(q.stats + 0, (int32_t)n_rounds); atomicAdd(q.stats + 1, (int32_t)n_rows); atomicAdd(q.stats + 2, (int32_t)n_walk); atomicAdd(q.stats + 3, 1);
    }
}

}  // namespace pvd
