// Shared device/host helpers for libdeepclr_amd (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/deepclr_amd.h"

#define DCLR_WAVE 64

#define DCLR_REQUIRE(cond)             \
    do {                               \
        if (!(cond)) return DCLR_E_INVALID; \
    } while (0)

static inline int dclr_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DCLR_OK : -(1000 + (int)e);
}

// ---- library-internal forms of three level-2 entry points (not exported): the split-f16 kernels with the word that
// receives 1 when an activation left the f16 range (overflow, NULL = not reported), and -- flow embedding -- a buffer of
// zero_count floats the kernel clears on its way (the head's column maxima: saves the fill launch between the two).
#define DCLR_INTERNAL __attribute__((visibility("hidden")))
DCLR_INTERNAL int dclr_x_head_conv_fused_f16(int m, int n_layers, int k_in, const int *k_host, const int *n_host,
                                             const void *const *w_packed_host, const float *const *bias_host,
                                             const float *x, int ldx, float *colmax, int rows_per_group,
                                             uint32_t *overflow, dclr_stream_t stream);
DCLR_INTERNAL int dclr_x_flow_embedding_fused_f16(int pairs, int npoint, int k, float radius, const float *f_rows,
                                                  const int32_t *knn_idx, const float *pt, const float *ps,
                                                  const float *w1a, const float *b1, const void *w2p, const float *b2,
                                                  const void *w3p, const float *b3, float *e_rows, float *zero,
                                                  long long zero_count, uint32_t *overflow, dclr_stream_t stream);

// dclr_fc with the overflow word as `poison`: set -> the outputs are written as NaN (the last layer of dclr_merge_forward)
DCLR_INTERNAL int dclr_x_fc(int m, int n, int k, const float *x, const float *w, const float *bias, int act, float *y,
                            const uint32_t *poison, dclr_stream_t stream);

// ---- frozen distance recipe (include/deepclr_amd.h): (dx*dx + dy*dy) + dz*dz, no contraction.
// The whole library is built with -ffp-contract=off; MLP code asks for FMA explicitly (fmaf).
__device__ __forceinline__ float dclr_sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

// ---- DPP wave reductions over u32 (result broadcast through an SGPR).
// `old` is the operation's identity, so lanes a DPP control leaves unwritten contribute nothing and
// hipcc folds each step into one v_max_u32_dpp / v_min_u32_dpp (passing `old = v` costs 4 instructions
// per step instead of 1 on the serial path of the sampling loop).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dclr_dpp(uint32_t identity, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)src, CTRL, ROW_MASK, 0xf, false);
}

#define DCLR_DPP_ROW_SHR(n) (0x110 + (n))
#define DCLR_DPP_ROW_BCAST15 0x142
#define DCLR_DPP_ROW_BCAST31 0x143

__device__ __forceinline__ uint32_t dclr_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t dclr_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Inclusive "scan to the right" with an idempotent op: lane 15 of each row, then lane 63, end up
// with the row / wave result.
__device__ __forceinline__ uint32_t dclr_row16_max_lanes(uint32_t v) {
    v = dclr_umax(v, dclr_dpp<DCLR_DPP_ROW_SHR(1), 0xf>(0u, v));
    v = dclr_umax(v, dclr_dpp<DCLR_DPP_ROW_SHR(2), 0xf>(0u, v));
    v = dclr_umax(v, dclr_dpp<DCLR_DPP_ROW_SHR(4), 0xf>(0u, v));
    v = dclr_umax(v, dclr_dpp<DCLR_DPP_ROW_SHR(8), 0xf>(0u, v));
    return v;
}
__device__ __forceinline__ uint32_t dclr_row16_min_lanes(uint32_t v) {
    v = dclr_umin(v, dclr_dpp<DCLR_DPP_ROW_SHR(1), 0xf>(0xFFFFFFFFu, v));
    v = dclr_umin(v, dclr_dpp<DCLR_DPP_ROW_SHR(2), 0xf>(0xFFFFFFFFu, v));
    v = dclr_umin(v, dclr_dpp<DCLR_DPP_ROW_SHR(4), 0xf>(0xFFFFFFFFu, v));
    v = dclr_umin(v, dclr_dpp<DCLR_DPP_ROW_SHR(8), 0xf>(0xFFFFFFFFu, v));
    return v;
}

__device__ __forceinline__ uint32_t dclr_wave_max_u32(uint32_t v) {
    v = dclr_row16_max_lanes(v);
    v = dclr_umax(v, dclr_dpp<DCLR_DPP_ROW_BCAST15, 0xa>(0u, v));
    v = dclr_umax(v, dclr_dpp<DCLR_DPP_ROW_BCAST31, 0xc>(0u, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t dclr_wave_min_u32(uint32_t v) {
    v = dclr_row16_min_lanes(v);
    v = dclr_umin(v, dclr_dpp<DCLR_DPP_ROW_BCAST15, 0xa>(0xFFFFFFFFu, v));
    v = dclr_umin(v, dclr_dpp<DCLR_DPP_ROW_BCAST31, 0xc>(0xFFFFFFFFu, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Same over the first 16 lanes only (one DPP row): lane 15 holds the result.
__device__ __forceinline__ uint32_t dclr_row16_max_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)dclr_row16_max_lanes(v), 15);
}
__device__ __forceinline__ uint32_t dclr_row16_min_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)dclr_row16_min_lanes(v), 15);
}

// Wave max over f32 for values >= 0 (bit pattern order == value order).
__device__ __forceinline__ float dclr_wave_max_nonneg(float v) {
    return __uint_as_float(dclr_wave_max_u32(__float_as_uint(v)));
}

// Read-only, wave-uniform parameter tables (MLP weights): viewing them through the constant address
// space lets the compiler keep s_load (scalar operands) even when workgroup fences sit between
// loads -- a fence otherwise counts as a clobber and demotes them to hoisted vector loads.
typedef const float __attribute__((address_space(4))) *dclr_const_f32p;
__device__ __forceinline__ dclr_const_f32p dclr_as_const(const float *p) { return (dclr_const_f32p)p; }

// Arguments of a non-inlined device function arrive in VGPRs; these re-establish wave uniformity.
__device__ __forceinline__ int dclr_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const float *dclr_uniform(const float *p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const float *)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int dclr_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ uint32_t dclr_lanemask_lt_popc(uint64_t mask) {
    // number of set bits of `mask` strictly below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// ---- batches that travel together without being concatenated (include/deepclr_amd.h, dclr_*_batched): the clouds of a
// call are `batches` batches of 2 * per clouds each ([templates | sources], the reference's batch layout), batch i at
// base + i * stride floats; the call numbers them [templates of every batch | sources of every batch]. batches <= 1: plain.
struct DclrCloudView {
    int per, batches;
    long long stride;
};
__device__ __forceinline__ size_t dclr_cloud_offset(const DclrCloudView &v, size_t c, size_t cloud_floats) {
    if (v.batches <= 1) return c * cloud_floats;
    const size_t half_n = (size_t)v.per * v.batches;        // templates (or sources) in the call
    const size_t half = c / half_n, r = c % half_n;
    return (r / v.per) * (size_t)v.stride + (half * v.per + r % v.per) * cloud_floats;
}
