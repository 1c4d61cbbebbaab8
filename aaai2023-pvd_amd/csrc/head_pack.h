// head_pack.h -- the heads' packed f16 weight image ([HeadLds<KIND>][HeadLdsT<KIND>] of fusedhead.hip: the forward matrices, then their
// transposes, row-padded), built from the fp32 masters, as a device function: k_head_pack (fusedhead.hip) runs it as a launch of its
// own; the VM student's lookup (vmencoder.hip: k_vm_fwd) can run it in extra workgroups of ITS launch -- the head's forward follows the
// lookup on the step's chain and both read the weights the update just wrote -- so the chain has neither a pack launch nor a
// cross-queue wait for one (pvd_vm_forward_pack_rider).
#pragma once

#include "pvd_device.h"

namespace pvd {

constexpr int kPad = 4;  // halfs of LDS row padding
constexpr int KIND_HASH = 0, KIND_VM = 1;
// halfs of the VM head's image (fusedhead.hip asserts it against HeadLds / HeadLdsT)
constexpr int kVmImageHalfs = (16 * (144 + kPad) + 64 * (32 + kPad) + 64 * (64 + kPad) + 16 * (64 + kPad)) +
                              (144 * (16 + kPad) + 32 * (64 + kPad) + 64 * (64 + kPad) + 64 * (16 + kPad));

// halfs of the hash head's image
constexpr int kHashImageHalfs = (64 * (32 + kPad) + 16 * (64 + kPad) + 64 * (32 + kPad) + 64 * (64 + kPad) + 16 * (64 + kPad)) +
                                (32 * (64 + kPad) + 64 * (16 + kPad) + 32 * (64 + kPad) + 64 * (64 + kPad) + 64 * (16 + kPad));

// One image element per thread and ONE round of loads: the ten matrices (five weights, plain and transposed) used to be ten loops one
// after the other, i.e. ten dependent memory round trips for 43 k elements (7.6 us in the step's timeline).
struct PackSeg {
    int begin;        // first image element (halfs) of this matrix
    int rows_dst, stride;  // destination rows and row stride (cols_pad + kPad, or rows_pad + kPad when transposed)
    const float *src;
    int rows, cols, rows_pad, cols_pad, row0, col_split, transposed;
};

__device__ __forceinline__ float pack_value(const PackSeg &g, int local) {
    const int dr = local / g.stride, dc = local - dr * g.stride;
    const int r = g.transposed ? dc : dr, c = g.transposed ? dr : dc;  // logical (padded) row / column of the weight
    if (r >= g.rows_pad || c >= g.cols_pad) return 0.f;                // row padding
    const int sr = r - g.row0;
    int sc = c;
    if (g.col_split >= 0) {
        if (c == g.col_split) return 0.f;
        if (c > g.col_split) sc = c - 1;
    }
    return (sr >= 0 && sr < g.rows && sc < g.cols) ? g.src[(size_t)sr * g.cols + sc] : 0.f;
}

// image elements first, first + step, ... of the KIND head's image (Wa2: hash head only)
template <int KIND>
__device__ __forceinline__ void head_pack_elements(const float *Wa1, const float *Wa2, const float *Wc1, const float *Wc2, const float *Wc3,
                                                   _Float16 *__restrict__ image, int first, int step) {
    // (matrix, rows, cols, rows_pad, cols_pad, row0, col_split) in the order HeadLds / HeadLdsT carve them
    PackSeg seg[10];
    int n = 0, pos = 0;
    auto add = [&](const float *src, int rows, int cols, int rows_pad, int cols_pad, int row0, int split, int transposed) {
        const int rows_dst = transposed ? cols_pad : rows_pad, stride = (transposed ? rows_pad : cols_pad) + kPad;
        seg[n++] = PackSeg{pos, rows_dst, stride, src, rows, cols, rows_pad, cols_pad, row0, split, transposed};
        pos += rows_dst * stride;
    };
    for (int t = 0; t < 2; t++) {
        if (KIND == KIND_HASH) {
            add(Wa1, 64, 28, 64, 32, 0, -1, t);
            add(Wa2, 16, 64, 16, 64, 0, -1, t);
        } else {
            add(Wa1, 15, 144, 16, 144, 1, -1, t);  // zero row 0
        }
        add(Wc1, 64, 31, 64, 32, 0, 16, t);  // zero column 16
        add(Wc2, 64, 64, 64, 64, 0, -1, t);
        add(Wc3, 3, 64, 16, 64, 0, -1, t);
    }
    const int total = pos;  // == HeadLds<KIND>::halfs + HeadLdsT<KIND>::halfs
    for (int i = first; i < total; i += step) {
        PackSeg g = seg[0];  // (unrolled selection: the segment table stays in scalar registers, no indexed private array)
#pragma unroll
        for (int q = 1; q < 10; q++)
            if (q < n && i >= seg[q].begin) g = seg[q];
        image[i] = (_Float16)pack_value(g, i - g.begin);
    }
}

}  // namespace pvd
