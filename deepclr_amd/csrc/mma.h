// fp32 MFMA building blocks shared by the dense-MLP kernels (gfx950).
//
// Instruction: v_mfma_f32_32x32x2_f32 -- exact binary32 (a k-ordered fmaf chain), 64 cycles per
// issue per SIMD = the f32 peak of the chip (MI355X_MICROARCH.md, "Matrix cores"). The pose tolerance
// (1e-4 on the 4x4, BASELINE.json) rules out plain bf16 inputs for the 131->...->1024 chains.
//
// Operand maps (lane l, i/j = l & 31, h = l >> 5):
//   A (32 x 2):  A[i][h]         B (2 x 32):  B[h][j]
//   C/D (32x32): 16 regs, column j = l & 31, row = (r & 3) + 8*(r >> 2) + 4*h
//
// K is consumed in groups of 8: MFMA q (0..3) of a group multiplies k = 8*g + 4*h + q, so one
// 16-byte read per lane feeds four MFMAs on either side:
//   activations  X[row][8g + 4h .. +3]                      (ds_read_b128 from a row-major LDS tile)
//   weights      packed[(ntile*KG + g)*64 + lane] (float4)  (one coalesced 1 KiB load per wave)
// The sum over k is thereby re-ordered inside each group of 8; both operands use the same order.
#pragma once
#include "common.h"

typedef float dclr_f32x16 __attribute__((ext_vector_type(16)));

// LDS row stride (floats) for a tile holding `kp` columns: kp rounded to 8, plus 4, so that
// stride/4 is odd and every 16-lane group of a ds_read_b128 hits 16 distinct 16-byte slots.
__host__ __device__ constexpr int dclr_lds_stride(int kp) { return ((kp + 7) / 8) * 8 + 4; }

__device__ __forceinline__ dclr_f32x16 dclr_zero16() {
    dclr_f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// One k-group (8 values of K) for MT row tiles x NT column tiles.
//   a_lds  : this lane's row inside row-tile 0, already offset by 4*h  (tile t adds t*32*stride)
//   w_lane : packed weights of column-tile 0, group g, this lane; column-tile u adds u*ntile_stride float4
template <int MT, int NT>
__device__ __forceinline__ void dclr_mma_group(dclr_f32x16 (&acc)[MT][NT], const float *a_lds, int stride,
                                               int g, const float4 *w_lane, int ntile_stride) {
    float4 b[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) b[u] = w_lane[(size_t)u * ntile_stride];
    float4 a[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4 *>(a_lds + t * 32 * stride + 8 * g);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
        }
}

// Whole K loop (kg k-groups) for MT x NT tiles with the packed-weight fragments of group g+1 requested
// before the MFMAs of group g are issued (a fragment load is an L2 round trip; asking for it right
// before use leaves the matrix pipe idle whenever the co-resident waves stall the same way).
template <int MT, int NT>
__device__ __forceinline__ void dclr_mma_step(dclr_f32x16 (&acc)[MT][NT], const float *a_lds, int stride, int g_lds,
                                              const float4 (&b)[NT]) {
    float4 a[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4 *>(a_lds + t * 32 * stride + 8 * g_lds);
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[u].x, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[u].y, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[u].z, acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[u].w, acc[t][u], 0, 0, 0);
        }
}

// Two named fragment sets, unrolled by two: while one set feeds the MFMAs the load of the other is in
// flight (written as a rotating single set, hipcc folds the prefetch back into a load-wait-use loop).
template <int MT, int NT>
__device__ __forceinline__ void dclr_mma_panel(dclr_f32x16 (&acc)[MT][NT], const float *a_lds, int stride, int g_lds0,
                                               int kg, const float4 *w_lane, int ntile_stride) {
    float4 b0[NT], b1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) b0[u] = w_lane[(size_t)u * ntile_stride];
    int g = 0;
    for (; g + 2 <= kg; g += 2) {
#pragma unroll
        for (int u = 0; u < NT; ++u) b1[u] = w_lane[(size_t)u * ntile_stride + (size_t)(g + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);                 // keep the prefetch ahead of the MFMAs it hides behind
        dclr_mma_step<MT, NT>(acc, a_lds, stride, g_lds0 + g, b0);
        const int gn = g + 2 < kg ? g + 2 : g + 1;
#pragma unroll
        for (int u = 0; u < NT; ++u) b0[u] = w_lane[(size_t)u * ntile_stride + (size_t)gn * 64];
        __builtin_amdgcn_sched_barrier(0);
        dclr_mma_step<MT, NT>(acc, a_lds, stride, g_lds0 + g + 1, b1);
    }
    if (g < kg) dclr_mma_step<MT, NT>(acc, a_lds, stride, g_lds0 + g, b0);
}

// Row of accumulator register r for lane-half h inside a 32-row tile.
__device__ __forceinline__ int dclr_acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- 16x16x4 variant (v_mfma_f32_16x16x4_f32: 32-cycle issue, same FLOP rate, 4 accumulator registers).
// Operand maps (lane l, r/c = l & 15, kq = l >> 4):  A[r][kq]  B[kq][c]  C/D: column c, row 4*kq + reg.
// K is consumed in groups of 16: MFMA q (0..3) multiplies k = 16*g + 4*kq + q.
//   activations  X[row][16g + 4kq .. +3]                       (ds_read_b128)
//   weights      packed16[(ntile*KG16 + g)*64 + lane] (float4) (one coalesced 1 KiB load per wave)
typedef float dclr_f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int NT>
__device__ __forceinline__ void dclr_mma16_group(dclr_f32x4 (&acc)[MT][NT], const float *a_lds, int stride, int g,
                                                 const float4 *w_lane, int ntile_stride) {
    float4 b[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) b[u] = w_lane[(size_t)u * ntile_stride];
    float4 a[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4 *>(a_lds + t * 16 * stride + 16 * g);
    // q outermost: back-to-back MFMAs hit different accumulators (dependent latency 40 > issue 32 cycles)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const float av = q == 0 ? a[t].x : (q == 1 ? a[t].y : (q == 2 ? a[t].z : a[t].w));
                const float bw = q == 0 ? b[u].x : (q == 1 ? b[u].y : (q == 2 ? b[u].z : b[u].w));
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, acc[t][u], 0, 0, 0);
            }
}

template <int MT, int NT>
__device__ __forceinline__ void dclr_mma16_step(dclr_f32x4 (&acc)[MT][NT], const float *a_lds, int stride, int g,
                                                const float4 (&b)[NT]) {
    float4 a[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4 *>(a_lds + t * 16 * stride + 16 * g);
    // q outermost: back-to-back MFMAs hit different accumulators (dependent latency 40 > issue 32 cycles)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const float av = q == 0 ? a[t].x : (q == 1 ? a[t].y : (q == 2 ? a[t].z : a[t].w));
                const float bw = q == 0 ? b[u].x : (q == 1 ? b[u].y : (q == 2 ? b[u].z : b[u].w));
                acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, acc[t][u], 0, 0, 0);
            }
}

// Whole K loop for the 16x16x4 layout, double-buffered like dclr_mma_panel.
template <int MT, int NT>
__device__ __forceinline__ void dclr_mma16_panel(dclr_f32x4 (&acc)[MT][NT], const float *a_lds, int stride, int kg,
                                                 const float4 *w_lane, int ntile_stride) {
    float4 b0[NT], b1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) b0[u] = w_lane[(size_t)u * ntile_stride];
    int g = 0;
    for (; g + 2 <= kg; g += 2) {
#pragma unroll
        for (int u = 0; u < NT; ++u) b1[u] = w_lane[(size_t)u * ntile_stride + (size_t)(g + 1) * 64];
        __builtin_amdgcn_sched_barrier(0);                 // keep the prefetch ahead of the MFMAs it hides behind
        dclr_mma16_step<MT, NT>(acc, a_lds, stride, g, b0);
        const int gn = g + 2 < kg ? g + 2 : g + 1;
#pragma unroll
        for (int u = 0; u < NT; ++u) b0[u] = w_lane[(size_t)u * ntile_stride + (size_t)gn * 64];
        __builtin_amdgcn_sched_barrier(0);
        dclr_mma16_step<MT, NT>(acc, a_lds, stride, g + 1, b1);
    }
    if (g < kg) dclr_mma16_step<MT, NT>(acc, a_lds, stride, g, b0);
}
