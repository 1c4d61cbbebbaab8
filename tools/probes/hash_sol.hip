// hash_sol.hip -- speed-of-light probes for the hash-grid lookup of the frozen teacher (k_hash_fwd_fused, csrc/fusedhead.hip).
//
// Same samples, same index arithmetic (csrc/grid_lookup.h), same lane mapping (two lanes per sample, lane pair = corners
// x / x+1, four y-z corners per lane) and therefore the SAME ADDRESS STREAM as the product kernel -- but nothing else: no
// blend, no LDS tile, no MFMA head, one dword written per lane.  What these kernels take is what the memory system charges
// for the gathers alone; variants change one thing at a time:
//   G        levels whose gathers are in flight together (product: 7)
//   L0..L1   which levels are gathered (coarse / fine split)
//   row_mask rows & mask: shrink every level's table to a cache-resident piece (the all-hit floor: TA / L1 / L2-hit rate)
//   TILE     samples per workgroup (product: 128)
//   PIPE     persistent workgroups, the next (chunk, group) item's loads issued before the current item's are consumed
//   FLAVOUR  plain / nontemporal / agent-scope (sc1, L1-bypassing) loads, for all levels or the fine ones only
//   k_sol_stream: a coalesced read of the lookup's ALGORITHMIC byte count (516 B/sample) -- what "fraction of the HBM
//            roofline" means for a launch of this size.
// Measurement tool (tools/hash_sol.py); not part of libpvd_hip.so.
#include "../../aaai2023-pvd_amd/csrc/grid_lookup.h"

#include <hip/hip_runtime.h>

using namespace pvd;

struct SolArgs {
    const float *xyz;      // [M][3] in [0, 1]
    const uint32_t *grid;  // packed f16 pairs
    int32_t offs[15];
    float scale[14];
    uint32_t M;
    uint32_t *out;         // [2 M]
    uint32_t row_mask;     // 0xffffffff: the real tables
    uint32_t chunk_perm;   // 1: chunk c is processed by workgroup slot (c % 8) * ceil(n / 8) + c / 8 (XCD-contiguous ranges)
    uint32_t *stage;       // split map with staging: corner values [(level, corner)][2 M] (round 6), or nullptr
};

template <int FLAVOUR>
__device__ __forceinline__ uint32_t load_row(const uint32_t *p) {
    if (FLAVOUR == 1) return __builtin_nontemporal_load(p);
    if (FLAVOUR == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

template <uint32_t G, uint32_t L0, int FLAVOUR = 0, uint32_t FINE_FROM = 0>
__device__ __forceinline__ void issue(const SolArgs &a, uint32_t g0, uint32_t gend, const float (&x01)[3], bool inside, uint32_t xb,
                                      uint32_t (&v)[G][4]) {
#pragma unroll
    for (uint32_t j = 0; j < G; j++) {
        const uint32_t level = g0 + j;
        if (level >= gend) break;
        const uint32_t off0 = (uint32_t)a.offs[level];
        const float scale = a.scale[level];
        Level3 lv;  // the product kernel's index code (csrc/grid_lookup.h)
        lv.init((uint32_t)a.offs[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, 0u, false);
        const uint32_t *__restrict__ table = a.grid + off0;
        uint32_t cell[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) cell[d] = inside ? (uint32_t)floorf(fmaf(x01[d], scale, 0.5f)) : 0u;
        uint32_t row[4];
        level3_rows(lv, 0u, false, cell, xb, row);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            v[j][k] = level >= FINE_FROM ? load_row<FLAVOUR>(table + (row[k] & a.row_mask)) : load_row<0>(table + (row[k] & a.row_mask));
    }
}

// KEEP = loads that may stay in flight (the next item's, issued after this one's): the wait + memory clobber is what makes
// "G levels per round trip" real -- without it the compiler hoists all 56 loads of a sample to the top whatever G says
template <uint32_t G, uint32_t KEEP>
__device__ __forceinline__ uint32_t fold(uint32_t g0, uint32_t gend, const uint32_t (&v)[G][4], uint32_t acc) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
#pragma unroll
    for (uint32_t j = 0; j < G; j++) {
        if (g0 + j >= gend) break;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) acc += v[j][k];
    }
    return acc;
}

__device__ __forceinline__ uint32_t chunk_of(const SolArgs &a, uint32_t slot, uint32_t nchunks) {
    if (!a.chunk_perm) return slot;
    // workgroup b runs on XCD b % 8: give XCD k the contiguous chunk range [k * per, (k + 1) * per)
    const uint32_t per = (nchunks + 7u) / 8u;
    return (slot & 7u) * per + (slot >> 3);
}

// one workgroup pass = TILE samples x 2 lanes; levels [L0, L1) in groups of G
template <uint32_t TILE, uint32_t G, uint32_t L0, uint32_t L1, bool PIPE, int FLAVOUR = 0, uint32_t FINE_FROM = 0>
__global__ void __launch_bounds__(2 * TILE) k_sol_gather(SolArgs a) {
    const uint32_t xb = threadIdx.x & 1u, s_local = threadIdx.x >> 1;
    const uint32_t nchunks = (a.M + TILE - 1) / TILE;
    const uint32_t nslots = a.chunk_perm ? ((nchunks + 7u) / 8u) * 8u : nchunks;
    constexpr uint32_t NG = (L1 - L0 + G - 1) / G;
    if (!PIPE) {
        for (uint32_t slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
            const uint32_t chunk = chunk_of(a, slot, nchunks);
            if (chunk >= nchunks) continue;
            const uint32_t b = chunk * TILE + s_local;
            float x01[3] = {0.f, 0.f, 0.f};
            bool inside = b < a.M;
            if (inside) {
                const Pos3 p = *reinterpret_cast<const Pos3 *>(a.xyz + (size_t)b * 3);
                x01[0] = p.x; x01[1] = p.y; x01[2] = p.z;
                for (uint32_t d = 0; d < 3; d++) inside = inside && !(x01[d] < 0.0f) && !(x01[d] > 1.0f);
            }
            uint32_t acc = 0;
#pragma unroll
            for (uint32_t g = 0; g < NG; g++) {
                uint32_t v[G][4];
                issue<G, L0, FLAVOUR, FINE_FROM>(a, L0 + g * G, L1, x01, inside, xb, v);
                acc = fold<G, 0>(L0 + g * G, L1, v, acc);
            }
            if (b < a.M) a.out[2 * b + xb] = acc;
        }
    } else {
        // items = (slot, group) in order; item i + 1's loads are issued before item i's values are consumed
        uint32_t vA[G][4], vB[G][4];
        uint32_t slot = blockIdx.x;
        uint32_t chunk = slot < nslots ? chunk_of(a, slot, nchunks) : nchunks;
        float x01[3] = {0.f, 0.f, 0.f};
        bool inside = false;
        uint32_t b = chunk * TILE + s_local;
        auto load_pos = [&]() {
            b = chunk * TILE + s_local;
            inside = chunk < nchunks && b < a.M;
            x01[0] = x01[1] = x01[2] = 0.f;
            if (inside) {
                const Pos3 p = *reinterpret_cast<const Pos3 *>(a.xyz + (size_t)b * 3);
                x01[0] = p.x; x01[1] = p.y; x01[2] = p.z;
                for (uint32_t d = 0; d < 3; d++) inside = inside && !(x01[d] < 0.0f) && !(x01[d] > 1.0f);
            }
        };
        load_pos();
        issue<G, L0, FLAVOUR, FINE_FROM>(a, L0, L1, x01, inside, xb, vA);
        uint32_t acc = 0;
        while (slot < nslots) {
            // groups 1 .. NG-1 of this chunk, then group 0 of the next chunk, each issued one item ahead
#pragma unroll
            for (uint32_t g = 0; g < NG; g++) {
                const bool last = g + 1 == NG;
                uint32_t out_b = b;
                bool out_ok = chunk < nchunks && b < a.M;
                if (last) {
                    slot += gridDim.x;
                    chunk = slot < nslots ? chunk_of(a, slot, nchunks) : nchunks;
                    load_pos();
                }
                const uint32_t gn = last ? 0u : g + 1;
                if (g & 1u) {
                    issue<G, L0, FLAVOUR, FINE_FROM>(a, L0 + gn * G, L1, x01, inside, xb, vA);
                    acc = fold<G, 4 * G>(L0 + g * G, L1, vB, acc);
                } else {
                    issue<G, L0, FLAVOUR, FINE_FROM>(a, L0 + gn * G, L1, x01, inside, xb, vB);
                    acc = fold<G, 4 * G>(L0 + g * G, L1, vA, acc);
                }
                if (last) {
                    if (out_ok) a.out[2 * out_b + xb] = acc;
                    acc = 0;
                    if (NG & 1u) {  // odd group count: the next chunk's group 0 sits in the "other" buffer; move it (register renaming)
#pragma unroll
                        for (uint32_t j = 0; j < G; j++)
#pragma unroll
                            for (uint32_t k = 0; k < 4; k++) vA[j][k] = vB[j][k];
                    }
                }
            }
        }
    }
}

// The split map (VERDICT r4 "next" 4): the four finest levels (2 MB each: 8 MB against an XCD's 4 MB of L2, every L2 pulling its own
// copy through the fabric) served by XCD PAIRS -- pair p gathers only the rows with ((row >> 10) & 3) == p, for ALL samples (workgroup
// b runs on XCD b % 8: pair (b % 8) >> 1; the pair's two XCDs take alternate chunks).  An L2 then holds a quarter of every fine
// level (2 MB in all).  Four passes over the samples' index arithmetic, a quarter of the loads each; what a product kernel would
// still have to add: the corner values written to a staging buffer and read back by the blending pass.
// Round 6 (VERDICT r5 "next" 3): the level range is a parameter (levels [L0, L0 + NL): 10-13 as in round 5, 7-13 = every level that
// is larger than an L2's share), and with a.stage the owning pair WRITES each corner value it loaded to the staging buffer the
// blending pass reads back -- what a bit-exact product needs: the reference blends corner by corner in f16 in a fixed order, so the
// pairs can only deliver VALUES, and a sample's 8 corners of a level hash to rows of different pairs (different XCDs: LDS cannot carry
// them, the staging goes through memory).
template <uint32_t TILE, uint32_t L0, uint32_t NL>
__global__ void __launch_bounds__(2 * TILE) k_sol_split(SolArgs a) {
    const uint32_t xb = threadIdx.x & 1u, s_local = threadIdx.x >> 1;
    const uint32_t nchunks = (a.M + TILE - 1) / TILE;
    const uint32_t xcd = blockIdx.x & 7u, pair = xcd >> 1, half = xcd & 1u;
    // the pair's workgroups: blockIdx.x >> 3 enumerates the workgroups of this XCD; chunks c with c % 2 == half belong to this XCD
    for (uint32_t c = (blockIdx.x >> 3) * 2u + half; c < nchunks; c += (gridDim.x >> 3) * 2u) {
        const uint32_t b = c * TILE + s_local;
        float x01[3] = {0.f, 0.f, 0.f};
        bool inside = b < a.M;
        if (inside) {
            const Pos3 p = *reinterpret_cast<const Pos3 *>(a.xyz + (size_t)b * 3);
            x01[0] = p.x; x01[1] = p.y; x01[2] = p.z;
            for (uint32_t d = 0; d < 3; d++) inside = inside && !(x01[d] < 0.0f) && !(x01[d] > 1.0f);
        }
        uint32_t acc = 0;
        uint32_t v[NL][4];
        bool mine[NL][4];
#pragma unroll
        for (uint32_t j = 0; j < NL; j++) {
            const uint32_t level = L0 + j;
            const uint32_t off0 = (uint32_t)a.offs[level];
            const float scale = a.scale[level];
            Level3 lv;
            lv.init((uint32_t)a.offs[level + 1] - off0, (uint32_t)ceil((double)scale) + 1u, 0u, false);
            const uint32_t *__restrict__ table = a.grid + off0;
            uint32_t cell[3];
#pragma unroll
            for (uint32_t d = 0; d < 3; d++) cell[d] = inside ? (uint32_t)floorf(fmaf(x01[d], scale, 0.5f)) : 0u;
            uint32_t row[4];
            level3_rows(lv, 0u, false, cell, xb, row);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t r = row[k] & a.row_mask;
                mine[j][k] = ((r >> 10) & 3u) == pair;
                v[j][k] = mine[j][k] ? table[r] : 0u;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.stage) {  // the owner's values to the staging planes: plane (j, k), element 2 b + xb
            if (b < a.M) {
#pragma unroll
                for (uint32_t j = 0; j < NL; j++)
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++)
                        if (mine[j][k]) a.stage[(size_t)(j * 4u + k) * 2u * a.M + 2u * b + xb] = v[j][k];
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < NL; j++)
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) acc += v[j][k];
            if (b < a.M && acc == 0x12345678u) a.out[2 * b + xb] = acc;  // (keeps the loads)
        }
    }
}

// ... and the blending pass's memory side: the coarse levels [0, L0) gathered as the product does, the NL fine levels' corner values
// read back from the staging planes (coalesced), one dword written per lane
template <uint32_t TILE, uint32_t L0, uint32_t NL>
__global__ void __launch_bounds__(2 * TILE) k_sol_blend_side(SolArgs a) {
    const uint32_t xb = threadIdx.x & 1u, s_local = threadIdx.x >> 1;
    const uint32_t nchunks = (a.M + TILE - 1) / TILE;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t b = chunk * TILE + s_local;
        float x01[3] = {0.f, 0.f, 0.f};
        bool inside = b < a.M;
        if (inside) {
            const Pos3 p = *reinterpret_cast<const Pos3 *>(a.xyz + (size_t)b * 3);
            x01[0] = p.x; x01[1] = p.y; x01[2] = p.z;
            for (uint32_t d = 0; d < 3; d++) inside = inside && !(x01[d] < 0.0f) && !(x01[d] > 1.0f);
        }
        uint32_t acc = 0;
        uint32_t v[L0][4];
        issue<L0, 0>(a, 0u, L0, x01, inside, xb, v);
        uint32_t w[NL][4];
        const uint32_t bb = b < a.M ? b : 0u;
#pragma unroll
        for (uint32_t j = 0; j < NL; j++)
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) w[j][k] = a.stage[(size_t)(j * 4u + k) * 2u * a.M + 2u * bb + xb];
        acc = fold<L0, 0>(0u, L0, v, acc);
#pragma unroll
        for (uint32_t j = 0; j < NL; j++)
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) acc += w[j][k];
        if (b < a.M) a.out[2 * b + xb] = acc;
    }
}

extern "C" int sol_split(const float *xyz, const void *grid, const int32_t *offs_host, const float *scale_host, uint32_t M, uint32_t *out,
                         uint32_t row_mask, uint32_t blocks, void *stream) {
    SolArgs a;
    a.xyz = xyz; a.grid = (const uint32_t *)grid; a.M = M; a.out = out; a.row_mask = row_mask; a.chunk_perm = 0; a.stage = nullptr;
    for (int i = 0; i < 15; i++) a.offs[i] = offs_host[i];
    for (int i = 0; i < 14; i++) a.scale[i] = scale_host[i];
    hipLaunchKernelGGL((k_sol_split<128, 10, 4>), dim3(blocks & ~7u), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

// levels 7-13; what: 0 = the pair-owned gathers alone, 1 = + staging writes (stage: 28 planes of 2 M dwords), 2 = the blending pass's
// memory side (coarse levels gathered + staged values read back)
extern "C" int sol_split7(int what, const float *xyz, const void *grid, const int32_t *offs_host, const float *scale_host, uint32_t M, uint32_t *out,
                          uint32_t *stage, uint32_t row_mask, uint32_t blocks, void *stream) {
    SolArgs a;
    a.xyz = xyz; a.grid = (const uint32_t *)grid; a.M = M; a.out = out; a.row_mask = row_mask; a.chunk_perm = 0;
    a.stage = what == 0 ? nullptr : stage;
    for (int i = 0; i < 15; i++) a.offs[i] = offs_host[i];
    for (int i = 0; i < 14; i++) a.scale[i] = scale_host[i];
    if (what == 2) {
        const uint32_t nchunks = (M + 127u) / 128u;
        hipLaunchKernelGGL((k_sol_blend_side<128, 7, 7>), dim3(blocks ? blocks : nchunks), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL((k_sol_split<128, 7, 7>), dim3(blocks & ~7u), dim3(256), 0, (hipStream_t)stream, a);
    }
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) k_sol_stream(const uint4 *__restrict__ src, size_t n16, uint32_t *out) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x;  // never true in practice: keeps the loads
}

__global__ void k_sol_empty(uint32_t *out) {
    if (out == nullptr) __builtin_trap();
}

template <uint32_t TILE, uint32_t G, uint32_t L0, uint32_t L1, bool PIPE, int FLAVOUR = 0, uint32_t FINE_FROM = 0>
static int launch(const SolArgs &a, uint32_t blocks, hipStream_t s) {
    const uint32_t nchunks = (a.M + TILE - 1) / TILE;
    if (blocks == 0) blocks = a.chunk_perm ? ((nchunks + 7u) / 8u) * 8u : nchunks;
    hipLaunchKernelGGL((k_sol_gather<TILE, G, L0, L1, PIPE, FLAVOUR, FINE_FROM>), dim3(blocks), dim3(2 * TILE), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" int sol_gather(int variant, const float *xyz, const void *grid, const int32_t *offs_host, const float *scale_host, uint32_t M,
                          uint32_t *out, uint32_t row_mask, uint32_t chunk_perm, uint32_t blocks, void *stream) {
    SolArgs a;
    a.xyz = xyz; a.grid = (const uint32_t *)grid; a.M = M; a.out = out; a.row_mask = row_mask; a.chunk_perm = chunk_perm; a.stage = nullptr;
    for (int i = 0; i < 15; i++) a.offs[i] = offs_host[i];
    for (int i = 0; i < 14; i++) a.scale[i] = scale_host[i];
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
    case 0: return launch<128, 7, 0, 14, false>(a, blocks, s);   // the product's structure
    case 1: return launch<128, 14, 0, 14, false>(a, blocks, s);  // one round trip
    case 2: return launch<128, 4, 0, 14, false>(a, blocks, s);
    case 3: return launch<128, 2, 0, 14, false>(a, blocks, s);
    case 4: return launch<128, 1, 0, 14, false>(a, blocks, s);   // level by level
    case 5: return launch<128, 5, 0, 10, false>(a, blocks, s);   // levels 0..9 only (cells >= a step: neighbours share lines)
    case 6: return launch<128, 4, 10, 14, false>(a, blocks, s);  // levels 10..13 only (every sample its own cell)
    case 7: return launch<128, 7, 0, 7, false>(a, blocks, s);    // the product's first group alone
    case 8: return launch<128, 7, 7, 14, false>(a, blocks, s);   // the product's second group alone
    case 9: return launch<64, 7, 0, 14, false>(a, blocks, s);    // half-size workgroups
    case 10: return launch<256, 7, 0, 14, false>(a, blocks, s);  // double-size workgroups
    case 11: return launch<128, 7, 0, 14, true>(a, blocks, s);   // persistent + one item ahead
    case 12: return launch<128, 4, 0, 14, true>(a, blocks, s);
    case 13: return launch<128, 2, 0, 14, true>(a, blocks, s);
    case 14: return launch<64, 7, 0, 14, true>(a, blocks, s);
    case 15: return launch<64, 14, 0, 14, false>(a, blocks, s);
    case 16: return launch<128, 14, 0, 14, false, 1, 0>(a, blocks, s);   // nontemporal loads, all levels
    case 17: return launch<128, 14, 0, 14, false, 1, 7>(a, blocks, s);   // nontemporal loads, levels 7..13
    case 18: return launch<128, 14, 0, 14, false, 2, 0>(a, blocks, s);   // sc1 (L1-bypassing) loads, all levels
    case 19: return launch<128, 14, 0, 14, false, 2, 7>(a, blocks, s);   // sc1 loads, levels 7..13
    case 20: return launch<128, 14, 0, 14, false, 2, 10>(a, blocks, s);  // sc1 loads, levels 10..13
    default: return -1;
    }
}

extern "C" int sol_stream(const void *src, size_t bytes, uint32_t *out, uint32_t blocks, void *stream) {
    hipLaunchKernelGGL(k_sol_stream, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, bytes / 16, out);
    return (int)hipGetLastError();
}

extern "C" int sol_empty(uint32_t *out, uint32_t blocks, void *stream) {
    hipLaunchKernelGGL(k_sol_empty, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
