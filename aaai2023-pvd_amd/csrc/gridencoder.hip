// gridencoder.hip -- multiresolution hash / tiled grid encoder for gfx950 (MI355X).
//
// Replaces the reference's _gridencoder module (gridencoder/src/gridencoder.cu).
// Layout in HBM: embeddings [sum_l size_l, C] (T = f32 or f16), level l owns rows
// [offsets[l], offsets[l+1]); outputs [L, B, C]; dy_dx [B, L, D, C]; grad [L, B, C].
//
// Work decomposition: workgroup = 256 consecutive points of ONE level (blockIdx.y = level),
// so everything that depends only on the level (table base, size, strides, hash-or-dense,
// per-level scale) is wave-uniform and lives in SGPRs, a level's table stays hot in the
// XCD-local L2 (2 MiB per hashed level in f16) while neighbouring workgroups sweep it, and
// the [L,B,C] store is a fully coalesced C*sizeof(T)-per-lane write.  One feature vector
// (C elements) is one global load / one packed atomic.
#include "grid_lookup.h"
#include "head_pack.h"

namespace pvd {

constexpr uint32_t kGridBlock = 256;
// XCD-aware (level, point-block) schedule.  MI355X has 8 XCDs with private 4 MiB L2s and dispatches
// workgroup i to XCD i % 8.  With a plain (point-block, level) grid every XCD sweeps every level, so each
// level's table (2 MiB per hashed level in f16) is pulled into all eight L2s: 8 x 21 MB of fabric traffic per
// launch, which is what bounds the kernel at training-batch sizes (~1e5 samples).  Instead the big levels are
// dealt out one per XCD ("exclusive": that XCD alone walks all points of that level, so the table is fetched
// into one L2 only), and what is left (big levels that do not fill a round of 8, plus the small coarse levels)
// is split by point-block across all XCDs.  Every XCD gets the same number of workgroups.  The mapping relies
// on i % 8 placement for speed only; any placement gives the same results.
struct LevelSchedule {
    uint32_t nb;             // point blocks per level
    uint32_t n_excl_rounds;  // exclusive levels per XCD
    uint32_t n_shared;       // levels split across XCDs
    uint32_t total_blocks;
    uint8_t excl[kMaxLevels];    // [round * 8 + xcd]
    uint8_t shared[kMaxLevels];

    __device__ __forceinline__ bool locate(uint32_t id, uint32_t &level, uint32_t &pblock) const {
        const uint32_t xcd = id & 7u, slot = id >> 3;
        const uint32_t n_excl_slots = n_excl_rounds * nb;
        if (slot < n_excl_slots) {
            const uint32_t round = slot / nb;
            level = excl[round * 8 + xcd];
            pblock = slot - round * nb;
            return true;
        }
        const uint32_t g = (slot - n_excl_slots) * 8 + xcd;
        if (g >= n_shared * nb) return false;
        const uint32_t k = g / nb;
        level = shared[k];
        pblock = g - k * nb;
        return true;
    }
};

static int g_grid_variant = 0;           // 1 = XCD-aware schedule, 0 = plain level-major order (default: measured faster)
static int g_grid_points_per_thread = 1;  // forward without dy_dx: 1, 2 or 4
static int g_grid_pair = 0;               // f16, D = 3, C = 2 without dy_dx: paired x / x+1 gathers (k_grid_fwd_pair); off: measured
                                          // 15 % faster on uniformly random points but 4-10 % slower on ray-coherent samples
static int g_grid_lps = 2;                // f16, D = 3, C = 2 without dy_dx: 2 / 4 = lanes per sample (k_grid_fwd_lps, default 2: measured
                                          // 15-20 % faster than thread-per-sample at the bench size, more on random points), 0 = thread per sample
static int g_grid_persist = 4096;         // k_grid_fwd_lps: workgroups of the persistent launch (0 = one per work item)
static int g_grid_bwd_lps = 1;            // backward, f16 / D = 3 / C = 2: two lanes per sample (k_grid_bwd_lps2); 0 = k_grid_bwd_coarse
static int g_grid_affine = 0;             // k_grid_fwd_lps: XCD-affine item order (LpsSchedule)
static uint32_t g_grid_hash_rows = 1u << 19;  // rows of a hashed level, for the table-size ordering of the affine schedule
static uint32_t g_grid_level_mask = 0;    // measurement only: if non-zero, the backward scatters just these levels
static float g_grid_coarse_scale = 1e30f;  // backward: levels with scale below this merge runs of equal rows per wave (0 = off);
                                           // measured best on every level (tools/bench_grid_bwd.py: 179 vs 205 vs 850 us, f16, 9e4 samples)

template <uint32_t D>
static LevelSchedule make_schedule(const LevelScales &sc, uint32_t L, uint32_t nb, size_t row_bytes) {
    LevelSchedule s;
    s.nb = nb;
    uint32_t big[kMaxLevels], n_big = 0, n_shared = 0;
    for (uint32_t l = 0; l < L; l++) {
        // table footprint from the level's resolution (dense (res+1)^D rows, capped by any hash size at 2^19+ rows)
        const double res = ceil((double)sc.scale[l]) + 2.0;
        double rows = 1.0;
        for (uint32_t d = 0; d < D; d++) rows *= res;
        const bool is_big = g_grid_variant == 1 && rows * (double)row_bytes >= 512.0 * 1024.0;
        if (is_big) big[n_big++] = l;
        else s.shared[n_shared++] = (uint8_t)l;
    }
    s.n_excl_rounds = n_big / 8;
    for (uint32_t i = 0; i < s.n_excl_rounds * 8; i++) s.excl[i] = (uint8_t)big[i];
    for (uint32_t i = s.n_excl_rounds * 8; i < n_big; i++) s.shared[n_shared++] = (uint8_t)big[i];
    s.n_shared = n_shared;
    const uint32_t per_xcd = s.n_excl_rounds * nb + div_up(n_shared * nb, 8u);
    s.total_blocks = per_xcd * 8;
    return s;
}

static thread_local InputAffine g_input_affine = {false, 0.f, 1.f};
// (pvd_grid_encode_forward_affine_pack) the hash head's packed weight image riding on the next forward launch of this thread
static thread_local pvd_head_pack_rider g_pack_rider = {};

// reference: kernel_grid, gridencoder.cu:75-224
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kGridBlock) k_grid_fwd(const float *__restrict__ inputs, const T *__restrict__ grid,
                                                         const int32_t *__restrict__ offsets, T *__restrict__ outputs,
                                                         uint32_t B, uint32_t L, LevelScales scales, LevelSchedule sched, uint32_t gridtype,
                                                         bool align_corners, bool calc_grad_inputs, T *__restrict__ dy_dx,
                                                         uint32_t level_mask, InputAffine aff) {
    using Vec = FeatVec<T, C>;
    uint32_t level, pblock;
    if (!sched.locate(blockIdx.x, level, pblock)) return;
    if (level_mask && !((level_mask >> level) & 1u)) return;  // measurement only (pvd_grid_set_variant)
    const uint32_t b = pblock * kGridBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const float scale = scales.scale[level];
    LevelIndex<D> index;
    index.init((uint32_t)offsets[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);
    const Vec *__restrict__ table = reinterpret_cast<const Vec *>(grid) + off0;
    Vec *__restrict__ out = reinterpret_cast<Vec *>(outputs) + ((size_t)level * B + b);
    T *__restrict__ dd = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;  // [B, L, D, C]

    float frac[D];
    uint32_t cell[D];
    if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, frac, cell, aff)) {
        Vec zero;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) zero.v[c] = (T)0;
        *out = zero;
        if (calc_grad_inputs) {
#pragma unroll
            for (uint32_t k = 0; k < D * C; k++) dd[k] = (T)0;
        }
        return;
    }

    // issue all 2^D gathers first (independent loads), then blend in the reference's corner order
    Vec corner[1u << D];
    float w[1u << D];
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float wi = 1;
        uint32_t pg[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx >> d) & 1u) { wi *= frac[d]; pg[d] = cell[d] + 1; }
            else { wi *= 1 - frac[d]; pg[d] = cell[d]; }
        }
        w[idx] = wi;
        corner[idx] = table[index(pg)];
    }
    Vec acc;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) acc.v[c] = (T)0;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) axpy<T>(acc.v[c], w[idx], corner[idx].v[c]);
    }
    *out = acc;

    if (calc_grad_inputs) {  // d out / d x, gridencoder.cu:180-223
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            T g[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) g[c] = (T)0;
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                float wi = scale;
                uint32_t pg[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                    if ((idx >> nd) & 1u) { wi *= frac[d]; pg[d] = cell[d] + 1; }
                    else { wi *= 1 - frac[d]; pg[d] = cell[d]; }
                }
                pg[gd] = cell[gd];
                const Vec lo = table[index(pg)];
                pg[gd] = cell[gd] + 1;
                const Vec hi = table[index(pg)];
#pragma unroll
                for (uint32_t c = 0; c < C; c++) axpy<T>(g[c], wi, (T)(hi.v[c] - lo.v[c]));
            }
#pragma unroll
            for (uint32_t c = 0; c < C; c++) dd[gd * C + c] = g[c];
        }
    }
}


// Forward without dy_dx for the tables of this code base (f16, D = 3, C = 2): the x and x+1 corners of a (y, z) pair
// come from ONE 8-byte access wherever they are adjacent in the table.  PMC (TCP_TOTAL_CACHE_ACCESSES 8.4 M, 0.7 per
// cycle per CU; TCP->TCC requests 1.7 M = 5 TB/s, L2 64 % hits) shows the lookup is bound by the L1's one-tag-lookup-
// per-cycle rate -- every 4-byte gather costs a full lookup -- not by L2 or HBM bandwidth.  Dense levels: index(x+1) =
// index(x) + 1 always (one dwordx2, 4-byte aligned).  Hashed levels (power-of-two size): index(x+1) = index(x) ^ 1 exactly
// when x is even -- the hash of x is x * 1 -- so the aligned pair holds both corners; for odd x the second corner is a
// separate, exec-masked load.  112 -> ~74 lookups per sample; same values, same blend order: bit-identical outputs.
// Measured: 15 % faster on uniformly random points (where every lane is its own line), but no faster -- 4-10 % slower --
// on ray-coherent samples, whose lanes already share lines; the training path keeps k_grid_fwd (knob: bit 1).
struct __attribute__((packed, aligned(4))) PairU32 {
    uint32_t a, b;
};

__global__ void __launch_bounds__(kGridBlock) k_grid_fwd_pair(const float *__restrict__ inputs, const uint32_t *__restrict__ grid,
                                                              const int32_t *__restrict__ offsets, uint32_t *__restrict__ outputs,
                                                              uint32_t B, uint32_t L, LevelScales scales, LevelSchedule sched,
                                                              uint32_t gridtype, bool align_corners, uint32_t level_mask, InputAffine aff) {
    constexpr uint32_t D = 3;
    uint32_t level, pblock;
    if (!sched.locate(blockIdx.x, level, pblock)) return;
    if (level_mask && !((level_mask >> level) & 1u)) return;
    const uint32_t b = pblock * kGridBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const float scale = scales.scale[level];
    LevelIndex<D> index;
    index.init((uint32_t)offsets[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);
    const uint32_t *__restrict__ table = grid + off0;
    uint32_t *__restrict__ out = outputs + ((size_t)level * B + b);

    float frac[D];
    uint32_t cell[D];
    if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, frac, cell, aff)) {
        *out = 0u;
        return;
    }
    uint32_t corner[8];
    const bool dense_plain = !index.hashed && gridtype == 0 && index.stride[1] != 0 && index.stride[2] != 0 &&
                             (cell[0] + 1) + (cell[1] + 1) * index.stride[1] + (cell[2] + 1) * index.stride[2] < index.size;
    if (index.hashed && index.pow2) {
        const uint32_t mask = index.size - 1u;
#pragma unroll
        for (uint32_t yz = 0; yz < 4; yz++) {
            const uint32_t h = ((cell[1] + (yz & 1u)) * 2654435761u) ^ ((cell[2] + (yz >> 1)) * 805459861u);
            const uint32_t i0 = (cell[0] ^ h) & mask, i1 = ((cell[0] + 1u) ^ h) & mask;
            const PairU32 pr = *reinterpret_cast<const PairU32 *>(table + (i0 & ~1u));
            corner[2 * yz] = (i0 & 1u) ? pr.b : pr.a;
            uint32_t c1 = (i0 & 1u) ? pr.a : pr.b;
            if ((i0 ^ i1) != 1u) c1 = table[i1];  // x odd: the neighbour hashes elsewhere
            corner[2 * yz + 1] = c1;
        }
    } else if (dense_plain) {
#pragma unroll
        for (uint32_t yz = 0; yz < 4; yz++) {
            const uint32_t i0 = cell[0] + (cell[1] + (yz & 1u)) * index.stride[1] + (cell[2] + (yz >> 1)) * index.stride[2];
            const PairU32 pr = *reinterpret_cast<const PairU32 *>(table + i0);
            corner[2 * yz] = pr.a;
            corner[2 * yz + 1] = pr.b;
        }
    } else {
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            const uint32_t pg[D] = {cell[0] + (idx & 1u), cell[1] + ((idx >> 1) & 1u), cell[2] + (idx >> 2)};
            corner[idx] = table[index(pg)];
        }
    }
    half_t acc0 = (half_t)0, acc1 = (half_t)0;
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float wi = 1;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) wi *= ((idx >> d) & 1u) ? frac[d] : 1 - frac[d];
        half_t v[2];
        __builtin_memcpy(v, &corner[idx], 4);
        axpy<half_t>(acc0, wi, v[0]);
        axpy<half_t>(acc1, wi, v[1]);
    }
    half_t r[2] = {acc0, acc1};
    uint32_t packed;
    __builtin_memcpy(&packed, r, 4);
    *out = packed;
}

// ---- "lanes per sample" forward (f16, D = 3, C = 2, no dy_dx): the corners of ONE sample are spread over LPS adjacent
// lanes, so that corners which sit in the same cache line are fetched by the SAME load instruction and the L1 serves the
// line once.  Why: the lookup is bound by the L1's tag-lookup rate (PMC, see k_grid_fwd_pair above): a 64-lane dword gather
// costs one lookup per DISTINCT line, and at the fine hashed levels every lane of the thread-per-sample kernel is its own
// line in all 8 of its loads.  But the hash of the x coordinate is x itself (prime 1): for fixed (y, z) the entries of
// x = 16a .. 16a+15 are one aligned 64-byte line (xor-permuted inside it), so corners x and x+1 share a line 15 times out of
// 16.  LPS = 2: lane pair = (x, x+1), 4 loads per lane (the (y, z) combinations) -> ~4.25 lines per sample and level instead
// of 8.  LPS = 4: lane quad = (x, y) bits, 2 loads per lane (z) -> the same lines in half the instructions.
// The blend keeps the reference's order: every lane rounds its own products (same f32-then-f16 rounding), the partner
// lanes' products come over by DPP quad permutes, and the sum runs over corners 0..7 in sequence as packed f16 adds (both
// channels at once; per channel the same IEEE add).  Bit-identical to k_grid_fwd.
// Work items of the lanes-per-sample kernel: (level, point block).  Two orders:
//   plain  : item id = level * nb + pblock, workgroup g takes ids g, g + G, ... (level-major sweep by the whole chip);
//   affine : every XCD owns a contiguous stretch of the level-major item list (levels sorted by table size, so a big
//            level's table is walked by ONE XCD -- at most two -- and is fetched into one L2 instead of all eight).
//            Workgroup g belongs to XCD g % 8 (the hardware's round-robin placement; a different placement costs speed
//            only) and is the (g / 8)-th of its XCD's G / 8 workgroups.  MEASURED AND REJECTED (kept as an A/B knob): 25 us
//            against 18 us -- the lookup is bound by the L2s' REQUEST rate (PMC: ~1.3 M line requests per launch, ~9 G/s per
//            XCD), and pinning a fine level to one XCD piles a third of all requests onto one L2.
struct LpsSchedule {
    uint32_t nb, total;
    uint32_t affine;               // 0 / 1
    uint32_t xcd_begin[9];         // affine: item range [xcd_begin[x], xcd_begin[x + 1]) of the ORDERED list
    uint8_t level_order[kMaxLevels];  // ordered list position -> level

    __device__ __forceinline__ uint32_t first(uint32_t g, uint32_t G, uint32_t &step, uint32_t &end) const {
        if (!affine) { step = G; end = total; return g; }
        const uint32_t x = g & 7u;
        step = G >> 3;
        end = xcd_begin[x + 1];
        return xcd_begin[x] + (g >> 3);
    }
    __device__ __forceinline__ void decode(uint32_t i, uint32_t &level, uint32_t &pblock) const {
        const uint32_t k = i / nb;
        pblock = i - k * nb;
        level = affine ? level_order[k] : k;
    }
};

template <uint32_t LPS>
__global__ void __launch_bounds__(kGridBlock) k_grid_fwd_lps(const float *__restrict__ inputs, const uint32_t *__restrict__ grid,
                                                             const int32_t *__restrict__ offsets, uint32_t *__restrict__ outputs,
                                                             uint32_t B, uint32_t L, LevelScales scales, LpsSchedule sched,
                                                             uint32_t gridtype, bool align_corners, uint32_t level_mask, InputAffine aff,
                                                             pvd_head_pack_rider pk = pvd_head_pack_rider{}, uint32_t lookup_blocks = 0xFFFFFFFFu) {
    constexpr uint32_t D = 3;
    constexpr uint32_t SPB = kGridBlock / LPS;  // samples per workgroup pass
    constexpr uint32_t NL = 8 / LPS;            // loads per lane
    if (blockIdx.x >= lookup_blocks) {  // the hash head's packed weight image riding on this launch (head_pack.h), at the end of the grid
        head_pack_elements<KIND_HASH>(pk.Wa1, pk.Wa2, pk.Wc1, pk.Wc2, pk.Wc3, reinterpret_cast<_Float16 *>(pk.image),
                                      (int)((blockIdx.x - lookup_blocks) * kGridBlock + threadIdx.x), (int)((gridDim.x - lookup_blocks) * kGridBlock));
        return;
    }
    const uint32_t q = threadIdx.x & (LPS - 1);
    const uint32_t xb = q & 1u, yq = (q >> 1) & 1u;
    const uint32_t s_in_block = threadIdx.x / LPS;
    uint32_t step, end;
    uint32_t i = sched.first(blockIdx.x, lookup_blocks == 0xFFFFFFFFu ? gridDim.x : lookup_blocks, step, end);
    // the position of the NEXT item's sample is loaded while this item's gathers are in flight (a wave otherwise pays two
    // dependent memory round trips per item: position, then corners)
    auto fetch = [&](uint32_t item, uint32_t &level, uint32_t &b, Pos3 &p) {
        uint32_t pblock;
        sched.decode(item, level, pblock);
        b = pblock * SPB + s_in_block;
        if (b < B) p = *reinterpret_cast<const Pos3 *>(inputs + (size_t)b * D);  // one 12-byte load
    };
    uint32_t level = 0, b = 0;
    Pos3 pos = {0.f, 0.f, 0.f};
    if (i < end) fetch(i, level, b, pos);
    while (i < end) {
        const uint32_t cur_level = level, cur_b = b;
        const Pos3 cur = pos;
        i += step;
        if (i < end) fetch(i, level, b, pos);
        if (level_mask && !((level_mask >> cur_level) & 1u)) continue;
        if (cur_b >= B) continue;  // (all LPS lanes of a sample leave together)
        const uint32_t off0 = (uint32_t)offsets[cur_level];
        const float scale = scales.scale[cur_level];
        LevelIndex<D> index;
        index.init((uint32_t)offsets[cur_level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);
        const uint32_t *__restrict__ table = grid + off0;
        uint32_t *__restrict__ out = outputs + ((size_t)cur_level * B + cur_b);
        float frac[D];
        uint32_t cell[D];
        const float xin[D] = {cur.x, cur.y, cur.z};
        if (!locate<D>(xin, scale, align_corners, frac, cell, aff)) {
            if (q == 0) *out = 0u;
            continue;
        }
        uint32_t v[NL];
        float w[NL];
#pragma unroll
        for (uint32_t k = 0; k < NL; k++) {
            const uint32_t yb = LPS == 2 ? (k & 1u) : yq, zb = LPS == 2 ? (k >> 1) : k;
            const uint32_t pg[D] = {cell[0] + xb, cell[1] + yb, cell[2] + zb};
            float wi = 1;  // the reference's product order: ((1 * wx) * wy) * wz
            wi *= xb ? frac[0] : 1 - frac[0];
            wi *= yb ? frac[1] : 1 - frac[1];
            wi *= zb ? frac[2] : 1 - frac[2];
            w[k] = wi;
            v[k] = table[index(pg)];
        }
        uint32_t acc = 0u;  // two f16 zeros
#pragma unroll
        for (uint32_t k = 0; k < NL; k++) {
            const uint32_t p = weighted_pair(w[k], v[k]);
            if (LPS == 2) {  // corners 2k (even lane) and 2k+1 (odd lane)
                const uint32_t other = dpp_quad<0xB1>(p);  // quad_perm [1,0,3,2]
                acc = pk_add(acc, xb ? other : p);
                acc = pk_add(acc, xb ? p : other);
            } else {  // corners 4k + {0,1,2,3} = the quad's lanes in order
                acc = pk_add(acc, dpp_quad<0x00>(p));
                acc = pk_add(acc, dpp_quad<0x55>(p));
                acc = pk_add(acc, dpp_quad<0xAA>(p));
                acc = pk_add(acc, dpp_quad<0xFF>(p));
            }
        }
        if (q == 0) *out = acc;
    }
}

static LpsSchedule make_lps_schedule(const LevelScales &sc, const int32_t *offsets_host_or_null, uint32_t L, uint32_t nb, bool affine,
                                     uint32_t hash_rows) {
    LpsSchedule s;
    s.nb = nb;
    s.total = L * nb;
    s.affine = affine ? 1u : 0u;
    // level order: biggest tables first (rows from the level's resolution, capped by the hash size)
    double rows[kMaxLevels];
    for (uint32_t l = 0; l < L; l++) {
        const double res = ceil((double)sc.scale[l]) + 2.0;
        rows[l] = res * res * res;
        if (rows[l] > (double)hash_rows) rows[l] = (double)hash_rows;
        s.level_order[l] = (uint8_t)l;
    }
    for (uint32_t a = 0; a + 1 < L; a++)  // stable selection sort, L <= 32
        for (uint32_t c = a + 1; c < L; c++)
            if (rows[s.level_order[c]] > rows[s.level_order[a]]) {
                const uint8_t t = s.level_order[c];
                for (uint32_t m = c; m > a; m--) s.level_order[m] = s.level_order[m - 1];
                s.level_order[a] = t;
            }
    (void)offsets_host_or_null;
    for (uint32_t x = 0; x <= 8; x++) s.xcd_begin[x] = (uint32_t)(((uint64_t)s.total * x) / 8u);
    return s;
}

// Forward without dy_dx, P points per thread (strided by the workgroup so every pass stays coalesced): the
// P x 2^D gathers of a thread are independent, so more of them are in flight per wave and the launch needs
// P x fewer workgroups (at ~1e5 samples the plain grid is ~2.4 waves of workgroups: tail-bound).
template <typename T, uint32_t D, uint32_t C, uint32_t P>
__global__ void __launch_bounds__(kGridBlock) k_grid_fwd_multi(const float *__restrict__ inputs, const T *__restrict__ grid,
                                                               const int32_t *__restrict__ offsets, T *__restrict__ outputs, uint32_t B,
                                                               uint32_t L, LevelScales scales, LevelSchedule sched, uint32_t gridtype,
                                                               bool align_corners) {
    using Vec = FeatVec<T, C>;
    uint32_t level, pblock;
    if (!sched.locate(blockIdx.x, level, pblock)) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const float scale = scales.scale[level];
    LevelIndex<D> index;
    index.init((uint32_t)offsets[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);
    const Vec *__restrict__ table = reinterpret_cast<const Vec *>(grid) + off0;
    Vec *__restrict__ out = reinterpret_cast<Vec *>(outputs) + (size_t)level * B;

    Vec corner[P][1u << D];
    float w[P][1u << D];
    bool live[P], inside[P];
#pragma unroll
    for (uint32_t q = 0; q < P; q++) {
        const uint32_t b = (pblock * P + q) * kGridBlock + threadIdx.x;
        live[q] = b < B;
        inside[q] = false;
        float frac[D];
        uint32_t cell[D];
        if (live[q]) inside[q] = locate<D>(inputs + (size_t)b * D, scale, align_corners, frac, cell);
#pragma unroll
        for (uint32_t idx = 0; idx < (1u << D); idx++) {
            float wi = 1;
            uint32_t pg[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                if ((idx >> d) & 1u) { wi *= frac[d]; pg[d] = cell[d] + 1; }
                else { wi *= 1 - frac[d]; pg[d] = cell[d]; }
            }
            w[q][idx] = wi;
            if (inside[q]) corner[q][idx] = table[index(pg)];
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < P; q++) {
        if (!live[q]) continue;
        Vec acc;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) acc.v[c] = (T)0;
        if (inside[q]) {
#pragma unroll
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
#pragma unroll
                for (uint32_t c = 0; c < C; c++) axpy<T>(acc.v[c], w[q][idx], corner[q][idx].v[c]);
            }
        }
        out[(pblock * P + q) * kGridBlock + threadIdx.x] = acc;
    }
}

// packed / scalar atomic accumulate of one weighted gradient vector
template <typename T, uint32_t C>
__device__ __forceinline__ void scatter_add(T *__restrict__ dst, float w, const FeatVec<T, C> &g);

template <uint32_t C>
__device__ __forceinline__ void scatter_add_f32(float *__restrict__ dst, float w, const FeatVec<float, C> &g) {
#pragma unroll
    for (uint32_t c = 0; c < C; c++)
        __hip_atomic_fetch_add(dst + c, w * g.v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_atomic_add_f32
}

template <uint32_t C>
__device__ __forceinline__ void scatter_add_f16(half_t *__restrict__ dst, float w, const FeatVec<half_t, C> &g) {
    if constexpr (C % 2 == 0) {
        // (half)(w*g) pairs -> global_atomic_pk_add_f16 (reference: __half2 atomicAdd, gridencoder.cu:303-304)
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {
            half2_t v;
            v.x = half_of_product(w, (float)g.v[c]);
            v.y = half_of_product(w, (float)g.v[c + 1]);
            __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t *)(dst + c), v);
        }
    } else {
        // C == 1 in half: the reference's at::Half atomicAdd is an empty stub (:22-26), i.e. the
        // gradient is silently dropped.  We accumulate it instead with a 32-bit CAS on the
        // containing word (documented deviation).
        const half_t add = half_of_product(w, (float)g.v[0]);
        const uintptr_t a = reinterpret_cast<uintptr_t>(dst);
        uint32_t *word = reinterpret_cast<uint32_t *>(a & ~(uintptr_t)3);
        const uint32_t shift = (a & 2u) ? 16u : 0u;
        uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), assumed;
        do {
            assumed = old;
            const uint16_t cur_bits = (uint16_t)(assumed >> shift);
            half_t cur;
            __builtin_memcpy(&cur, &cur_bits, 2);
            const half_t sum = cur + add;
            uint16_t sum_bits;
            __builtin_memcpy(&sum_bits, &sum, 2);
            const uint32_t next = (assumed & ~(0xffffu << shift)) | ((uint32_t)sum_bits << shift);
            old = assumed;
            __hip_atomic_compare_exchange_strong(word, &old, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (old != assumed);
    }
}

template <>
__device__ __forceinline__ void scatter_add<float, 1>(float *d, float w, const FeatVec<float, 1> &g) { scatter_add_f32<1>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<float, 2>(float *d, float w, const FeatVec<float, 2> &g) { scatter_add_f32<2>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<float, 4>(float *d, float w, const FeatVec<float, 4> &g) { scatter_add_f32<4>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<float, 8>(float *d, float w, const FeatVec<float, 8> &g) { scatter_add_f32<8>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<half_t, 1>(half_t *d, float w, const FeatVec<half_t, 1> &g) { scatter_add_f16<1>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<half_t, 2>(half_t *d, float w, const FeatVec<half_t, 2> &g) { scatter_add_f16<2>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<half_t, 4>(half_t *d, float w, const FeatVec<half_t, 4> &g) { scatter_add_f16<4>(d, w, g); }
template <>
__device__ __forceinline__ void scatter_add<half_t, 8>(half_t *d, float w, const FeatVec<half_t, 8> &g) { scatter_add_f16<8>(d, w, g); }

// reference: kernel_grid_backward, gridencoder.cu:227-314
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kGridBlock) k_grid_bwd(const T *__restrict__ grad, const float *__restrict__ inputs,
                                                         const int32_t *__restrict__ offsets, T *__restrict__ grad_grid,
                                                         uint32_t B, uint32_t L, LevelScales scales, LevelSchedule sched, uint32_t gridtype,
                                                         bool align_corners, uint32_t level_mask) {
    using Vec = FeatVec<T, C>;
    uint32_t level, pblock;
    if (!sched.locate(blockIdx.x, level, pblock)) return;
    if (level_mask && !((level_mask >> level) & 1u)) return;
    const uint32_t b = pblock * kGridBlock + threadIdx.x;
    if (b >= B) return;
    const uint32_t off0 = (uint32_t)offsets[level];
    const float scale = scales.scale[level];
    LevelIndex<D> index;
    index.init((uint32_t)offsets[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);

    float frac[D];
    uint32_t cell[D];
    if (!locate<D>(inputs + (size_t)b * D, scale, align_corners, frac, cell)) return;  // grad stays 0 (:254-259)
    const Vec g = reinterpret_cast<const Vec *>(grad)[(size_t)level * B + b];
    T *__restrict__ table = grad_grid + (size_t)off0 * C;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float wi = 1;
        uint32_t pg[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx >> d) & 1u) { wi *= frac[d]; pg[d] = cell[d] + 1; }
            else { wi *= 1 - frac[d]; pg[d] = cell[d]; }
        }
        scatter_add<T, C>(table + (size_t)index(pg) * C, wi, g);
    }
}

// Coarse levels: the samples of a wave are consecutive along a ray and a coarse cell is many steps wide
// (level 0: ~40 steps), so whole runs of lanes scatter into the SAME rows -- which is what makes the plain
// kernel slow there (rocprof/bench_grid_bwd: level 0 alone 527 us of 806 us: same-address atomics serialise at
// the memory side).  Here every corner's contributions are first summed across each run of equal row index
// with a segmented wave scan (runs are identified by a ballot of head flags, so only contiguous equal keys
// merge), and only the last lane of a run issues the atomic: one atomic per (run, corner) instead of per
// (sample, corner).  The run sum is formed in fp32 and rounded once.
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kGridBlock) k_grid_bwd_coarse(const T *__restrict__ grad, const float *__restrict__ inputs,
                                                                const int32_t *__restrict__ offsets, T *__restrict__ grad_grid, uint32_t B,
                                                                uint32_t L, LevelScales scales, uint32_t level_lo, uint32_t level_hi,
                                                                uint32_t gridtype, bool align_corners, uint32_t level_mask) {
    using Vec = FeatVec<T, C>;
    const uint32_t level = level_lo + blockIdx.y;
    if (level >= level_hi) return;
    if (level_mask && !((level_mask >> level) & 1u)) return;
    const uint32_t b = blockIdx.x * kGridBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t off0 = (uint32_t)offsets[level];
    const float scale = scales.scale[level];
    LevelIndex<D> index;
    index.init((uint32_t)offsets[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);
    float frac[D];
    uint32_t cell[D];
    const bool live = b < B && locate<D>(inputs + (size_t)b * D, scale, align_corners, frac, cell);
    float g[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) g[c] = 0.f;
    if (live) {
        const Vec gv = reinterpret_cast<const Vec *>(grad)[(size_t)level * B + b];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) g[c] = (float)gv.v[c];
    }
    T *__restrict__ table = grad_grid + (size_t)off0 * C;
#pragma unroll
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float wi = 1;
        uint32_t pg[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((idx >> d) & 1u) { wi *= live ? frac[d] : 0.f; pg[d] = live ? cell[d] + 1 : 0u; }
            else { wi *= live ? 1 - frac[d] : 0.f; pg[d] = live ? cell[d] : 0u; }
        }
        const uint32_t key = live ? index(pg) : 0xffffffffu;
        float v[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) v[c] = wi * g[c];
        // runs of equal keys: head flags -> run ids
        const uint32_t prev = __shfl_up(key, 1, 64);
        const bool head = lane == 0 || prev != key;
        const unsigned long long heads = __ballot(head);
        const uint32_t run = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull));
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t r_up = __shfl_up(run, off, 64);
            const bool same = (int)lane >= off && r_up == run;
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                const float up = __shfl_up(v[c], off, 64);
                if (same) v[c] += up;
            }
        }
        const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
        if (live && tail) {
            Vec sum;
#pragma unroll
            for (uint32_t c = 0; c < C; c++) sum.v[c] = (T)v[c];
            scatter_add<T, C>(table + (size_t)key * C, 1.0f, sum);
        }
    }
}

// Scatter-add for the tables of this code base (f16, D = 3, C = 2) with TWO lanes per sample, as in the forward
// (k_grid_fwd_lps): lane pair = corners x / x+1 of the same (y, z), whose rows sit in one cache line 15 times out of 16 (the
// hash of x is x).  The memory side sees one request per distinct line of a wave-instruction, and a 4-byte packed-f16 atomic
// is a whole request of its own otherwise: the fabric's request rate (~35 G/s measured), not bytes, bounds this
// kernel.  Runs of consecutive samples that hit the same row are still merged first (k_grid_bwd_coarse), now among the lanes
// of equal parity: segmented scan with strides 2, 4, .., 32, the run's last lane issues the atomic.
__global__ void __launch_bounds__(kGridBlock) k_grid_bwd_lps2(const uint32_t *__restrict__ grad, const float *__restrict__ inputs,
                                                              const int32_t *__restrict__ offsets, half_t *__restrict__ grad_grid, uint32_t B,
                                                              uint32_t L, LevelScales scales, uint32_t gridtype, bool align_corners,
                                                              uint32_t level_mask, InputAffine aff) {
    constexpr uint32_t D = 3;
    const uint32_t level = blockIdx.y;
    if (level >= L) return;
    if (level_mask && !((level_mask >> level) & 1u)) return;
    const uint32_t t = blockIdx.x * kGridBlock + threadIdx.x;
    const uint32_t b = t >> 1, xb = t & 1u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t off0 = (uint32_t)offsets[level];
    const float scale = scales.scale[level];
    LevelIndex<D> index;
    index.init((uint32_t)offsets[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, gridtype, align_corners);
    float frac[D];
    uint32_t cell[D];
    const bool live = b < B && locate<D>(inputs + (size_t)b * D, scale, align_corners, frac, cell, aff);
    float g0 = 0.f, g1 = 0.f;
    if (live) {
        half_t gv[2];
        const uint32_t packed = grad[(size_t)level * B + b];
        __builtin_memcpy(gv, &packed, 4);
        g0 = (float)gv[0];
        g1 = (float)gv[1];
    }
    half_t *__restrict__ table = grad_grid + (size_t)off0 * 2;
    const unsigned long long parity = xb ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t yb = k & 1u, zb = k >> 1;
        float wi = 1;
        wi *= live ? (xb ? frac[0] : 1 - frac[0]) : 0.f;
        wi *= live ? (yb ? frac[1] : 1 - frac[1]) : 0.f;
        wi *= live ? (zb ? frac[2] : 1 - frac[2]) : 0.f;
        uint32_t key = 0xffffffffu;
        if (live) {
            const uint32_t pg[D] = {cell[0] + xb, cell[1] + yb, cell[2] + zb};
            key = index(pg);
        }
        float v0 = wi * g0, v1 = wi * g1;
        const uint32_t prev = __shfl_up(key, 2, 64);
        const bool head = lane < 2 || prev != key;
        const unsigned long long heads = __ballot(head);
        const uint32_t run = (uint32_t)__popcll(heads & parity & ((2ull << lane) - 1ull));
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) {
            const uint32_t r_up = __shfl_up(run, off, 64);
            const bool same = (int)lane >= off && r_up == run;
            const float u0 = __shfl_up(v0, off, 64), u1 = __shfl_up(v1, off, 64);
            if (same) { v0 += u0; v1 += u1; }
        }
        const bool tail = lane >= 62 || ((heads >> (lane + 2)) & 1ull);
        if (live && tail) {
            FeatVec<half_t, 2> sum;
            sum.v[0] = (half_t)v0;
            sum.v[1] = (half_t)v1;
            scatter_add<half_t, 2>(table + (size_t)key * 2, 1.0f, sum);
        }
    }
}

// reference: kernel_input_backward, gridencoder.cu:317-343
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kGridBlock) k_grid_input_bwd(const T *__restrict__ grad, const T *__restrict__ dy_dx,
                                                               T *__restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * kGridBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *__restrict__ dd = dy_dx + (size_t)b * L * D * C;
    T r = (T)0;
    for (uint32_t l = 0; l < L; l++) {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const T gv = grad[((size_t)l * B + b) * C + c];
            const T dv = dd[(size_t)l * D * C + d * C + c];
            if constexpr (sizeof(T) == 4) r = fmaf(gv, dv, r);
            else r = r + (T)(gv * dv);  // half: product rounded, then half add
        }
    }
    grad_inputs[t] = r;
}

template <typename T, uint32_t D, uint32_t C>
static int launch_fwd(const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t L, float S,
                      uint32_t H, bool calc, void *dy_dx, uint32_t gridtype, bool align, hipStream_t s) {
    const LevelScales sc = make_scales(L, S, H);
    const InputAffine aff = g_input_affine;
    if (!calc && g_grid_pair && D == 3 && C == 2 && sizeof(T) == 2) {
        const LevelSchedule sched = make_schedule<D>(sc, L, div_up(B, kGridBlock), sizeof(T) * C);
        hipLaunchKernelGGL(k_grid_fwd_pair, dim3(sched.total_blocks), dim3(kGridBlock), 0, s, inputs, (const uint32_t *)emb, offsets,
                           (uint32_t *)outputs, B, L, sc, sched, gridtype, align, g_grid_level_mask, aff);
        return check_launch();
    }
    if (!calc && g_grid_lps && D == 3 && C == 2 && sizeof(T) == 2) {
        const uint32_t lps = (uint32_t)g_grid_lps;
        const uint32_t nb = div_up(B, kGridBlock / lps), total = L * nb;
        uint32_t blocks = g_grid_persist > 0 ? (total < (uint32_t)g_grid_persist ? total : (uint32_t)g_grid_persist) : total;
        const bool affine = g_grid_affine && blocks >= 8;
        if (affine) blocks &= ~7u;
        const LpsSchedule sched = make_lps_schedule(sc, nullptr, L, nb, affine, g_grid_hash_rows);
        if (lps == 2 && g_pack_rider.image) {
            const pvd_head_pack_rider pk = g_pack_rider;
            g_pack_rider = pvd_head_pack_rider{};  // (taken)
            hipLaunchKernelGGL((k_grid_fwd_lps<2>), dim3(blocks + div_up((uint32_t)kHashImageHalfs, kGridBlock)), dim3(kGridBlock), 0, s, inputs,
                               (const uint32_t *)emb, offsets, (uint32_t *)outputs, B, L, sc, sched, gridtype, align, g_grid_level_mask, aff, pk, blocks);
        } else if (lps == 2)
            hipLaunchKernelGGL((k_grid_fwd_lps<2>), dim3(blocks), dim3(kGridBlock), 0, s, inputs, (const uint32_t *)emb, offsets, (uint32_t *)outputs,
                               B, L, sc, sched, gridtype, align, g_grid_level_mask, aff);
        else
            hipLaunchKernelGGL((k_grid_fwd_lps<4>), dim3(blocks), dim3(kGridBlock), 0, s, inputs, (const uint32_t *)emb, offsets, (uint32_t *)outputs,
                               B, L, sc, sched, gridtype, align, g_grid_level_mask, aff);
        return check_launch();
    }
    const uint32_t P = (calc || aff.on) ? 1u : (uint32_t)g_grid_points_per_thread;
    if (P > 1 && sizeof(T) * C <= 8) {
        const LevelSchedule sched = make_schedule<D>(sc, L, div_up(B, kGridBlock * P), sizeof(T) * C);
        if (P == 2)
            hipLaunchKernelGGL((k_grid_fwd_multi<T, D, C, 2>), dim3(sched.total_blocks), dim3(kGridBlock), 0, s, inputs, (const T *)emb, offsets,
                               (T *)outputs, B, L, sc, sched, gridtype, align);
        else
            hipLaunchKernelGGL((k_grid_fwd_multi<T, D, C, 4>), dim3(sched.total_blocks), dim3(kGridBlock), 0, s, inputs, (const T *)emb, offsets,
                               (T *)outputs, B, L, sc, sched, gridtype, align);
        return check_launch();
    }
    const LevelSchedule sched = make_schedule<D>(sc, L, div_up(B, kGridBlock), sizeof(T) * C);
    hipLaunchKernelGGL((k_grid_fwd<T, D, C>), dim3(sched.total_blocks), dim3(kGridBlock), 0, s, inputs, (const T *)emb, offsets, (T *)outputs, B,
                       L, sc, sched, gridtype, align, calc, (T *)dy_dx, g_grid_level_mask, aff);
    return check_launch();
}

template <typename T, uint32_t D, uint32_t C>
static int launch_bwd(const void *grad, const float *inputs, const int32_t *offsets, void *grad_emb, uint32_t B, uint32_t L, float S,
                      uint32_t H, bool calc, const void *dy_dx, void *grad_inputs, uint32_t gridtype, bool align, hipStream_t s) {
    const LevelScales sc = make_scales(L, S, H);
    if (g_grid_bwd_lps && !calc && D == 3 && C == 2 && sizeof(T) == 2) {
        hipLaunchKernelGGL(k_grid_bwd_lps2, dim3(div_up(2u * B, kGridBlock), L), dim3(kGridBlock), 0, s, (const uint32_t *)grad, inputs, offsets,
                           (half_t *)grad_emb, B, L, sc, gridtype, align, g_grid_level_mask, g_input_affine);
        return check_launch();
    }
    if (g_input_affine.on) return PVD_ERR_UNSUPPORTED;  // (pvd_grid_encode_backward_affine: the f16 / D 3 / C 2 scatter only)
    // levels whose cells are wider than a few marching steps go through the run-merging kernel
    uint32_t n_coarse = 0;
    if (g_grid_coarse_scale > 0.f)
        while (n_coarse < L && sc.scale[n_coarse] < g_grid_coarse_scale) n_coarse++;
    if (n_coarse > 0)
        hipLaunchKernelGGL((k_grid_bwd_coarse<T, D, C>), dim3(div_up(B, kGridBlock), n_coarse), dim3(kGridBlock), 0, s, (const T *)grad, inputs,
                           offsets, (T *)grad_emb, B, L, sc, 0u, n_coarse, gridtype, align, g_grid_level_mask);
    uint32_t fine_mask = g_grid_level_mask;
    if (n_coarse > 0) {
        const uint32_t not_coarse = ~((1u << n_coarse) - 1u) & ((L >= 32 ? 0u : (1u << L)) - 1u);
        fine_mask = g_grid_level_mask ? (g_grid_level_mask & not_coarse) : not_coarse;
    }
    if (n_coarse == 0 || fine_mask != 0) {
        const LevelSchedule sched = make_schedule<D>(sc, L, div_up(B, kGridBlock), sizeof(T) * C);
        hipLaunchKernelGGL((k_grid_bwd<T, D, C>), dim3(sched.total_blocks), dim3(kGridBlock), 0, s, (const T *)grad, inputs, offsets,
                           (T *)grad_emb, B, L, sc, sched, gridtype, align, fine_mask);
    }
    if (calc)
        hipLaunchKernelGGL((k_grid_input_bwd<T, D, C>), dim3(div_up(B * D, kGridBlock)), dim3(kGridBlock), 0, s, (const T *)grad,
                           (const T *)dy_dx, (T *)grad_inputs, B, L);
    return check_launch();
}

#define PVD_GRID_DISPATCH(FN, ...)                                              \
    do {                                                                        \
        if (D == 3) {                                                           \
            switch (C) {                                                        \
                case 1: return FN<T, 3, 1>(__VA_ARGS__);                        \
                case 2: return FN<T, 3, 2>(__VA_ARGS__);                        \
                case 4: return FN<T, 3, 4>(__VA_ARGS__);                        \
                case 8: return FN<T, 3, 8>(__VA_ARGS__);                        \
            }                                                                   \
        } else if (D == 2) {                                                    \
            switch (C) {                                                        \
                case 1: return FN<T, 2, 1>(__VA_ARGS__);                        \
                case 2: return FN<T, 2, 2>(__VA_ARGS__);                        \
                case 4: return FN<T, 2, 4>(__VA_ARGS__);                        \
                case 8: return FN<T, 2, 8>(__VA_ARGS__);                        \
            }                                                                   \
        }                                                                       \
        return PVD_ERR_UNSUPPORTED; /* gridencoder.cu:355,370 */               \
    } while (0)

template <typename T>
static int fwd_t(const float *inputs, const void *emb, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D, uint32_t C,
                 uint32_t L, float S, uint32_t H, bool calc, void *dy_dx, uint32_t gridtype, bool align, hipStream_t s) {
    PVD_GRID_DISPATCH(launch_fwd, inputs, emb, offsets, outputs, B, L, S, H, calc, dy_dx, gridtype, align, s);
}

template <typename T>
static int bwd_t(const void *grad, const float *inputs, const int32_t *offsets, void *grad_emb, uint32_t B, uint32_t D, uint32_t C,
                 uint32_t L, float S, uint32_t H, bool calc, const void *dy_dx, void *grad_inputs, uint32_t gridtype, bool align,
                 hipStream_t s) {
    PVD_GRID_DISPATCH(launch_bwd, grad, inputs, offsets, grad_emb, B, L, S, H, calc, dy_dx, grad_inputs, gridtype, align, s);
}

}  // namespace pvd

using namespace pvd;

extern "C" {

// tuning knob for A/B measurements (tools/bench_grid.py); not part of the drop-in surface
int pvd_grid_set_variant(int v) {  // bit 0: XCD-aware schedule; bit 1: paired gathers; bits 4..7: points per thread (0 -> 1, else 2 or 4)
    const int old = g_grid_variant | (g_grid_points_per_thread << 4);
    g_grid_variant = v & 1;
    g_grid_pair = (v & 2) ? 1 : 0;  // bit 1: the paired-gather kernel
    const int ppt = (v >> 4) & 15;
    g_grid_points_per_thread = ppt == 2 ? 2 : (ppt >= 4 ? 4 : 1);
    g_grid_level_mask = ((uint32_t)v >> 8) & 0x1fffffu;  // bits 8..28: backward level mask (measurement only)
    g_grid_coarse_scale = (v & (1 << 30)) ? 0.f : ((v & (1 << 29)) ? 300.f : 1e30f);  // bit 30: no run merging; bit 29: coarse levels only
    return old;
}

int pvd_grid_set_fwd_kernel(int lanes_per_sample, int persistent_blocks) {
    if (lanes_per_sample != 0 && lanes_per_sample != 2 && lanes_per_sample != 4) return PVD_ERR_INVALID;
    const int old = g_grid_lps | (g_grid_persist << 4);
    g_grid_lps = lanes_per_sample;
    g_grid_affine = (persistent_blocks & (1 << 30)) ? 1 : 0;  // bit 30: XCD-affine item order
    g_grid_bwd_lps = (persistent_blocks & (1 << 29)) ? 0 : 1;  // bit 29: backward through k_grid_bwd_coarse (A/B)
    persistent_blocks &= ~(1 << 29);
    persistent_blocks &= ~(1 << 30);
    g_grid_persist = persistent_blocks > 0 ? persistent_blocks : 0;
    return old;
}

int pvd_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void *dy_dx,
                            uint32_t gridtype, int align_corners, int dtype, pvd_stream_t stream) {
    if (L == 0 || L > kMaxLevels) return PVD_ERR_UNSUPPORTED;
    if (B == 0) return PVD_OK;
    if (!inputs || !embeddings || !offsets || !outputs || (calc_grad_inputs && !dy_dx)) return PVD_ERR_INVALID;
    if (dtype == PVD_F32)
        return fwd_t<float>(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs != 0, dy_dx, gridtype,
                            align_corners != 0, (hipStream_t)stream);
    if (dtype == PVD_F16)
        return fwd_t<half_t>(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs != 0, dy_dx, gridtype,
                             align_corners != 0, (hipStream_t)stream);
    return PVD_ERR_UNSUPPORTED;
}

int pvd_grid_encode_forward_affine(const float *inputs, float in_add, float in_div, const void *embeddings, const int32_t *offsets,
                                   void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                   int align_corners, int dtype, pvd_stream_t stream) {
    if (!(in_div != 0.f)) return PVD_ERR_INVALID;
    g_input_affine = {true, in_add, in_div};
    const int rc = pvd_grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, 0, nullptr, gridtype, align_corners, dtype,
                                           stream);
    g_input_affine = {false, 0.f, 1.f};
    return rc;
}

__global__ void __launch_bounds__(256) k_grid_pack_only(pvd_head_pack_rider pk) {
    head_pack_elements<KIND_HASH>(pk.Wa1, pk.Wa2, pk.Wc1, pk.Wc2, pk.Wc3, reinterpret_cast<_Float16 *>(pk.image), (int)(blockIdx.x * 256 + threadIdx.x),
                                  (int)(gridDim.x * 256));
}

int pvd_grid_encode_forward_affine_pack(const float *inputs, float in_add, float in_div, const void *embeddings, const int32_t *offsets,
                                        void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                        int align_corners, int dtype, const pvd_head_pack_rider *pack, pvd_stream_t stream) {
    if (!pack || pack->kind != 0 || !pack->Wa1 || !pack->Wa2 || !pack->Wc1 || !pack->Wc2 || !pack->Wc3 || !pack->image) return PVD_ERR_INVALID;
    g_pack_rider = *pack;
    const int rc = pvd_grid_encode_forward_affine(inputs, in_add, in_div, embeddings, offsets, outputs, B, D, C, L, S, H, gridtype, align_corners,
                                                  dtype, stream);
    if (g_pack_rider.image) {  // the launch that ran (another kernel variant, B == 0, an error) did not take it: pack in a launch of its own
        const pvd_head_pack_rider pk = g_pack_rider;
        g_pack_rider = pvd_head_pack_rider{};
        if (rc != PVD_OK) return rc;
        hipLaunchKernelGGL(k_grid_pack_only, dim3(div_up((uint32_t)kHashImageHalfs, 256u)), dim3(256), 0, (hipStream_t)stream, pk);
        return check_launch();
    }
    return rc;
}

int pvd_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets,
                             void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             int calc_grad_inputs, const void *dy_dx, void *grad_inputs, uint32_t gridtype, int align_corners,
                             int dtype, pvd_stream_t stream) {
    (void)embeddings;
    if (L == 0 || L > kMaxLevels) return PVD_ERR_UNSUPPORTED;
    if (B == 0) return PVD_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings || (calc_grad_inputs && (!dy_dx || !grad_inputs))) return PVD_ERR_INVALID;
    if (dtype == PVD_F32)
        return bwd_t<float>(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs != 0, dy_dx, grad_inputs,
                            gridtype, align_corners != 0, (hipStream_t)stream);
    if (dtype == PVD_F16)
        return bwd_t<half_t>(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs != 0, dy_dx, grad_inputs,
                             gridtype, align_corners != 0, (hipStream_t)stream);
    return PVD_ERR_UNSUPPORTED;
}

int pvd_grid_encode_backward_affine(const void *grad, const float *inputs, float in_add, float in_div, const void *embeddings,
                                    const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                    uint32_t H, uint32_t gridtype, int align_corners, int dtype, pvd_stream_t stream) {
    if (!(in_div != 0.f)) return PVD_ERR_INVALID;
    if (!(dtype == PVD_F16 && D == 3 && C == 2)) return PVD_ERR_UNSUPPORTED;
    g_input_affine = {true, in_add, in_div};
    const int rc = pvd_grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, 0, nullptr, nullptr, gridtype,
                                            align_corners, dtype, stream);
    g_input_affine = {false, 0.f, 1.f};
    return rc;
}

}  // extern "C"
