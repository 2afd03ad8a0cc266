// Split-fp16 MFMA building blocks (gfx950): fp32-accurate products at the fp16 matrix rate.
//
// Every fp32 operand x is carried as two halves  hi = f16(x),  lo = f16((x - hi) * 2^11)  and a product
// is evaluated as  a*b ~= a_hi*b_hi + 2^-11 * (a_hi*b_lo + a_lo*b_hi)  (the lo*lo term, 2^-22 relative,
// is dropped). hi carries 11 significant bits and the scaled lo the next 11, so the pair holds x to
// ~2^-22; the three f16 products are exact in the f32 accumulator. Measured on MI355X against fp64
// (scratch/mfma16_probe.hip, K = 512, all-positive data = worst case for accumulation bias): relative
// rms error 1.2e-7 for this scheme vs 3.0e-7 for the chained v_mfma_f32_32x32x2_f32 it replaces, i.e.
// the result is at least as close to the exact product as the fp32 matrix pipe's.
// Rate: v_mfma_f32_32x32x16_f16 / 16x16x32_f16 retire 16x the MACs per cycle of the f32 forms; at
// three instructions per product the contraction runs at 16/3 = 5.3x the fp32 matrix peak.
// Range: |x| must stay below 65504 (f16 max); larger magnitudes saturate (dclr_split clamps).
//
// Two accumulators per tile: `acc` takes hi*hi, `acc2` takes both cross terms at scale 2^11;
// result = acc + acc2 * 2^-11.
//
// Fragment layout (both instruction shapes): a lane holds 8 consecutive k of one row/column,
//   32x32x16: index = lane & 31, k = 16 g + 8 (lane >> 5) + q
//   16x16x32: index = lane & 15, k = 32 g + 8 (lane >> 4) + q
// so in memory one "k-octet" is 16 bytes of hi followed by 16 bytes of lo:
//   activations (LDS)  row r, octet o:  byte r * stride + 32 o  (hi)  /  + 16  (lo)
//   weights (global)   plane hi then plane lo, each  [(tile * KG + g) * 64 + lane]  16-byte fragments
// The A and B operand maps are symmetric, so the same fragments serve W*X^T (output lane = point,
// registers = 4 consecutive channels: what the next layer's octets need) and X*W^T (output lane =
// channel, registers = points: what a max over points needs).
#pragma once
#include "mma.h"

typedef _Float16 dclr_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dclr_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 dclr_h2 __attribute__((ext_vector_type(2)));

constexpr float DCLR_SPLIT_SCALE = 2048.f;
constexpr float DCLR_SPLIT_INV = 1.f / 2048.f;
constexpr float DCLR_F16_MAX = 65504.f;

__device__ __forceinline__ void dclr_split(float v, _Float16 &hi, _Float16 &lo) {
    const float c = fminf(fmaxf(v, -DCLR_F16_MAX), DCLR_F16_MAX);
    hi = (_Float16)c;
    lo = (_Float16)fminf(fmaxf((v - (float)hi) * DCLR_SPLIT_SCALE, -DCLR_F16_MAX), DCLR_F16_MAX);
}

typedef float dclr_f2 __attribute__((ext_vector_type(2)));

// Two values at a time (v_cvt_pk_f16_f32, v_pk_add/mul_f32): 4 vector instructions per value instead of ~10.
// The _relu form clamps to [0, 65504] with one v_med3_f32 -- the ReLU of the layer comes for free.
__device__ __forceinline__ void dclr_split2_clamped(dclr_f2 c, dclr_h2 &hi, dclr_h2 &lo) {
    hi = __builtin_convertvector(c, dclr_h2);
    const dclr_f2 r = (c - __builtin_convertvector(hi, dclr_f2)) * DCLR_SPLIT_SCALE;
    lo = __builtin_convertvector(r, dclr_h2);
}
__device__ __forceinline__ void dclr_split2_relu(float v0, float v1, dclr_h2 &hi, dclr_h2 &lo) {
    const dclr_f2 c = {__builtin_amdgcn_fmed3f(v0, 0.f, DCLR_F16_MAX), __builtin_amdgcn_fmed3f(v1, 0.f, DCLR_F16_MAX)};
    dclr_split2_clamped(c, hi, lo);
}
__device__ __forceinline__ void dclr_split2(float v0, float v1, dclr_h2 &hi, dclr_h2 &lo) {
    const dclr_f2 c = {__builtin_amdgcn_fmed3f(v0, -DCLR_F16_MAX, DCLR_F16_MAX),
                       __builtin_amdgcn_fmed3f(v1, -DCLR_F16_MAX, DCLR_F16_MAX)};
    dclr_split2_clamped(c, hi, lo);
}

// The same two, remembering in `peak` the largest value (magnitude) that entered the clamp: one v_max3_f32 per two values.
// A kernel compares `peak` with DCLR_F16_MAX once per phase and reports through dclr_report_overflow -- the clamp is silent
// otherwise, and activations beyond 65504 would turn into wrong poses without a message.
__device__ __forceinline__ void dclr_split2_relu(float v0, float v1, dclr_h2 &hi, dclr_h2 &lo, float &peak) {
    peak = fmaxf(fmaxf(v0, v1), peak);
    dclr_split2_relu(v0, v1, hi, lo);
}
__device__ __forceinline__ void dclr_split2(float v0, float v1, dclr_h2 &hi, dclr_h2 &lo, float &peak) {
    peak = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), peak);
    dclr_split2(v0, v1, hi, lo);
}
// flag: NULL, or one word any later reader polls (device memory, or host memory mapped into the device's address space:
// a plain system-scope store of the constant 1, so it needs no atomic the link may lack). Sticky: never cleared here.
__device__ __forceinline__ void dclr_report_overflow(uint32_t *flag, float peak) {
    if (flag != nullptr && peak > DCLR_F16_MAX) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// LDS row stride in BYTES for kp values per row (kp % 16 == 0): 4 kp + 16, i.e. (stride / 16) odd, so the
// 16 or 32 rows addressed by one ds_read_b128 fall into distinct 16-byte bank groups.
__host__ __device__ constexpr int dclr_split_stride(int kp) { return 4 * kp + 16; }

__device__ __forceinline__ dclr_h8 dclr_lds_h8(const char *p) { return *reinterpret_cast<const dclr_h8 *>(p); }
__device__ __forceinline__ dclr_h8 dclr_frag_h8(const float4 *p) {
    const float4 v = *p;
    return __builtin_bit_cast(dclr_h8, v);
}

__device__ __forceinline__ dclr_f32x16 dclr_mfma32(dclr_h8 a, dclr_h8 b, dclr_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ dclr_f32x4 dclr_mfma16(dclr_h8 a, dclr_h8 b, dclr_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
