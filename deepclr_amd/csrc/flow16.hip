// Fused flow embedding on split-fp16 MFMA (mma16f.h).
//
// Same operator and workgroup shape as flow_kernel in flow.hip (reference: MotionEmbeddingBase.forward,
// /root/reference/deepclr/models/deepclr.py:201-231): 4 template points per workgroup, row tile t =
// neighbours 4t..4t+3 of the 4 points, layer 1 by linearity on the vector pipe. Layers 2 and 3 run on
// v_mfma_f32_16x16x32_f16 with hi/lo-split operands (three instructions per product, f32 accumulation):
//   layer 2 computes W2 * H1^T -- accumulator lane = neighbour row, registers = 4 consecutive channels,
//           split and stored in place as half-octets of layer 3's input;
//   layer 3 computes H2 * W3^T -- lane = channel, lane-quarter = template point, registers = its 4
//           neighbours of the tile, so mask + max over the k neighbours stays in registers.
#ifdef DCLR_ABLATION
#include <stdlib.h>
#endif

#include "mma16f.h"

namespace {

constexpr int F16_G = 4;                        // template points per workgroup
constexpr int F16_C = 128;                      // hidden width of layers 1 and 2
constexpr int F16_OUT = 256;
constexpr int F16_STRIDE = dclr_split_stride(F16_C);    // 528 bytes per row
constexpr int F16_KG = F16_C / 32;              // 4 k-steps of 32

// Byte offset of k-octet o = 4 g + kq (hi 16 B, lo 16 B) inside an LDS row. NOT 32 o: ds_read_b128 is served in the lane
// groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- rows 0-3 and 12-15 of one lane-quarter together with rows
// 4-11 of the NEXT quarter -- so quarters kq and kq ^ 1 must sit a multiple of 256 B apart for the 16 lanes of a group
// to cover 16 different 16-byte slots (row stride = 33 slots). With consecutive octets (32 B apart) every operand read
// was 2-way conflicted: 5.2 M conflict cycles of 10.5 M LDS cycles (profiles/r01_e_pmc_split_fp16.txt).
__host__ __device__ constexpr int f16_octet_offset(int o) { return 256 * (o & 1) + 128 * ((o >> 1) & 1) + 32 * (o >> 2); }

// One pass over K for T row tiles x 2 channel tiles. TRANSPOSED: weights are the A operand.
// ABL (timing-only ablations, results wrong): bit 0 = no gather round trips in phase A (rows built from constants),
// bit 1 = weight fragments always from k-step 0 of tile 0 (no weight streaming), bit 2 = no phase A at all, bit 3 = no weight
// loads at all (80 pairs: 497 us; ABL 1 / 2 / 8 / 9 / 4 / 12: 465 / 457 / 414 / 385 / 366 / 298)
template <int T, bool TRANSPOSED, int ABL = 0>
__device__ __forceinline__ void flow16_panel(dclr_f32x4 (&acc)[T][2], dclr_f32x4 (&acc2)[T][2], const char *a_lane,
                                             const float4 *wh_lane, const float4 *wl_lane, int tile_stride) {
    dclr_h8 h0[2], l0[2], h1[2], l1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if constexpr (ABL & 8) {                                  // timing probe: no weight loads at all
            h0[u] = __builtin_bit_cast(dclr_h8, make_float4(1.f, 2.f, (float)u, 3.f));
            l0[u] = h0[u];
        } else {
            h0[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride);
            l0[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride);
        }
    }
    auto step = [&](int g, const dclr_h8 (&wh)[2], const dclr_h8 (&wl)[2]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const char *p = a_lane + t * 16 * F16_STRIDE + 32 * g;
            const dclr_h8 ah = dclr_lds_h8(p), al = dclr_lds_h8(p + 16);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[t][u] = TRANSPOSED ? dclr_mfma16(wh[u], ah, acc[t][u]) : dclr_mfma16(ah, wh[u], acc[t][u]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc2[t][u] = TRANSPOSED ? dclr_mfma16(wl[u], ah, acc2[t][u]) : dclr_mfma16(ah, wl[u], acc2[t][u]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc2[t][u] = TRANSPOSED ? dclr_mfma16(wh[u], al, acc2[t][u]) : dclr_mfma16(al, wh[u], acc2[t][u]);
        }
    };
#pragma unroll
    for (int g = 0; g < F16_KG; g += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (ABL & 10) { h1[u] = l0[u]; l1[u] = h0[u]; }
            else {
            h1[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride + (size_t)(g + 1) * 64);
            l1[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride + (size_t)(g + 1) * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        step(g, h0, l0);
        if (g + 2 < F16_KG) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if constexpr (ABL & 10) { h0[u] = l1[u]; l0[u] = h1[u]; }
                else {
                h0[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride + (size_t)(g + 2) * 64);
                l0[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride + (size_t)(g + 2) * 64);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        step(g + 1, h1, l1);
    }
}

template <int T, int ABL = 0>
__global__ __launch_bounds__(256, T <= 5 ? 3 : 2) void flow16_kernel(int pairs, int npoint, int k, float radius,
                                                     const float *__restrict__ f_rows,
                                                     const int32_t *__restrict__ knn_idx,
                                                     const float *__restrict__ pt, const float *__restrict__ ps,
                                                     const float *__restrict__ w1a, const float *__restrict__ b1,
                                                     const float4 *__restrict__ w2p, const float *__restrict__ b2,
                                                     const float4 *__restrict__ w3p, const float *__restrict__ b3,
                                                     float *__restrict__ e_rows, float *__restrict__ zero,
                                                     long long zero_count, uint32_t *overflow) {
    __shared__ __attribute__((aligned(16))) char tile[T * 16 * F16_STRIDE];
    __shared__ uint32_t vbits[F16_G];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, c16 = lane & 15;
    const size_t total = (size_t)pairs * npoint;
    // The workgroups of one pair gather from that pair's source rows (ps: 512 bytes per point, 20-30 rows per template
    // point). Consecutive block ids are dealt round-robin over the 8 XCDs, so every XCD's L2 ended up fetching every pair's
    // source rows (545 MB per 80-pair launch against 225 MB algorithmic). With a multiple of 8 pairs and whole workgroups
    // per pair, block L works on pair (L / 8 / blocks_per_pair) * 8 + L % 8 instead: one pair, one L2. Speed only.
    size_t blk = blockIdx.x;
    if ((pairs & 7) == 0 && npoint % F16_G == 0) {
        const unsigned per_pair = (unsigned)(npoint / F16_G);
        const unsigned xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        blk = (size_t)((i / per_pair) * 8u + xcd) * per_pair + i % per_pair;
    }
    const size_t g0 = blk * F16_G;
    // the buffer the NEXT launch on this stream accumulates into with atomic maxima (the head's column maxima)
    if (zero != nullptr)
        for (long long i = (long long)blockIdx.x * 256 + tid; i < zero_count; i += (long long)gridDim.x * 256) zero[i] = 0.f;

    // ---- phase A: wave w builds the layer-1 rows of template point g0 + w (lane = channels 2l, 2l+1) ----
    {
        const int p = wave;
        const size_t gp = g0 + p;
        const bool live = gp < total && !(ABL & 4);                     // wave-uniform
        uint32_t bits = 0;
        int s_done = 0;
        float peak = 0.f;
        // channels 2 lane, 2 lane + 1 sit in octet lane / 4 at half positions 2 (lane % 4), + 1
        char *const slot = tile + f16_octet_offset(lane >> 2) + 4 * (lane & 3);
        if (live) {
            // Three dependent L2 round trips for the whole point instead of one or two per neighbour:
            // (1) the k neighbour indices, one per lane; (2) lane s fetches neighbour s's position, so the
            // position differences, their norms and the radius mask are computed for all neighbours at once;
            // (3) the k source halves of layer 1 (float2 per lane and neighbour), all requested before the
            // first is used.
            const size_t pair = gp / npoint;
            const float *trow = f_rows + gp * DCLR_F_STRIDE;             // template clouds come first
            const float tx = trow[64], ty = trow[65], tz = trow[66];
            const size_t src0 = (pairs + pair) * (size_t)npoint;         // first row of the source cloud
            // a slot the search left unfilled (-1: fewer than k candidates within the 1e10 start distance of the slots, or
            // NaN coordinates; upstream fails at its .view(2, G, k) there) reads row 0 and is masked like a neighbour
            // beyond the radius: no address ever leaves the source cloud
            const int raw_nb = (ABL & 1) ? (lane & 15) : (lane < k ? knn_idx[gp * k + lane] : 0);
            const int my_nb = raw_nb < 0 ? 0 : raw_nb;
            const float4 nbp = (ABL & 1) ? make_float4(tx + lane, ty, tz, 0.f)
                                         : *reinterpret_cast<const float4 *>(f_rows + (src0 + my_nb) * DCLR_F_STRIDE + 64);
            const float my_dx = nbp.x - tx, my_dy = nbp.y - ty, my_dz = nbp.z - tz;
            const float norm = sqrtf(my_dx * my_dx + my_dy * my_dy + my_dz * my_dz);
            bits = (uint32_t)__ballot(lane < k && raw_nb >= 0 && (!(radius > 0.f) || norm < radius));
            constexpr int KMAX = 4 * T;
            float2 psv[KMAX];
            const float *psrow = ps + pair * (size_t)npoint * F16_C + 2 * lane;
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int nb = __builtin_amdgcn_readlane(my_nb, s < k ? s : 0);     // s >= k: a harmless repeat
                psv[s] = (ABL & 1) ? make_float2(0.01f * nb, 0.02f * lane) : *reinterpret_cast<const float2 *>(psrow + (size_t)nb * F16_C);
            }
            const float2 ptv = *reinterpret_cast<const float2 *>(pt + gp * F16_C + 2 * lane);
            const float2 bv = *reinterpret_cast<const float2 *>(b1 + 2 * lane);
            const float wa0 = w1a[(2 * lane) * 3 + 0], wa1 = w1a[(2 * lane) * 3 + 1], wa2 = w1a[(2 * lane) * 3 + 2];
            const float wb0 = w1a[(2 * lane + 1) * 3 + 0], wb1 = w1a[(2 * lane + 1) * 3 + 1],
                        wb2 = w1a[(2 * lane + 1) * 3 + 2];
            const float base0 = ptv.x + bv.x, base1 = ptv.y + bv.y;
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                if (s < k) {                                             // wave-uniform
                    const float dx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_dx), s));
                    const float dy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_dy), s));
                    const float dz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_dz), s));
                    float v0 = base0 + psv[s].x, v1 = base1 + psv[s].y;
                    v0 = fmaf(wa0, dx, v0); v0 = fmaf(wa1, dy, v0); v0 = fmaf(wa2, dz, v0);
                    v1 = fmaf(wb0, dx, v1); v1 = fmaf(wb1, dy, v1); v1 = fmaf(wb2, dz, v1);
                    dclr_h2 hi, lo;
                    dclr_split2_relu(v0, v1, hi, lo, peak);
                    const int row = (s >> 2) * 16 + 4 * p + (s & 3);
                    *reinterpret_cast<dclr_h2 *>(slot + row * F16_STRIDE) = hi;
                    *reinterpret_cast<dclr_h2 *>(slot + row * F16_STRIDE + 16) = lo;
                }
            }
            s_done = k;
        }
        for (int s = s_done; s < 4 * T; ++s) {                          // padding rows (k % 4 != 0, or no point)
            const int row = (s >> 2) * 16 + 4 * p + (s & 3);
            *reinterpret_cast<uint32_t *>(slot + row * F16_STRIDE) = 0u;
            *reinterpret_cast<uint32_t *>(slot + row * F16_STRIDE + 16) = 0u;
        }
        if (lane == 0) vbits[p] = bits;
        dclr_report_overflow(overflow, peak);
    }
    __syncthreads();

    const char *a_lane = tile + c16 * F16_STRIDE + f16_octet_offset(kq);   // octet 4 g + kq of row c16: + 32 g (+ tile offset)

    // ---- phase B: layer 2 (128 -> 128), wave w owns channel tiles 2w, 2w+1 ---------------------------
    {
        // registers i of tile u = channels (2 wave + u) * 16 + 4 kq + i: the accumulators start at the bias
        dclr_f32x4 acc[T][2], acc2[T][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 bv = *reinterpret_cast<const float4 *>(b2 + (2 * wave + u) * 16 + 4 * kq);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                acc[t][u][0] = bv.x; acc[t][u][1] = bv.y; acc[t][u][2] = bv.z; acc[t][u][3] = bv.w;
                acc2[t][u] = 0.f;
            }
        }
        const float4 *wh = w2p + (size_t)(2 * wave) * F16_KG * 64 + lane;
        flow16_panel<T, true, ABL>(acc, acc2, a_lane, wh, wh + (size_t)(F16_C / 16) * F16_KG * 64, F16_KG * 64);
        __syncthreads();                                   // every wave has consumed the layer-1 rows
        float peak = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ch = (2 * wave + u) * 16 + 4 * kq;   // registers i = channels ch + i of neighbour row c16
#pragma unroll
            for (int t = 0; t < T; ++t) {
                dclr_h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    dclr_h2 a, b;
                    dclr_split2_relu(fmaf(acc2[t][u][i], DCLR_SPLIT_INV, acc[t][u][i]),
                                     fmaf(acc2[t][u][i + 1], DCLR_SPLIT_INV, acc[t][u][i + 1]), a, b, peak);
                    hi[i] = a[0]; hi[i + 1] = a[1]; lo[i] = b[0]; lo[i + 1] = b[1];
                }
                char *dst = tile + (t * 16 + c16) * F16_STRIDE + f16_octet_offset(ch >> 3) + 2 * (ch & 7);
                *reinterpret_cast<dclr_h4 *>(dst) = hi;
                *reinterpret_cast<dclr_h4 *>(dst + 16) = lo;
            }
        }
        dclr_report_overflow(overflow, peak);
    }
    __syncthreads();

    // ---- phase C: layer 3 (128 -> 256) + radius mask + max over neighbours; wave w owns channel tiles
    //      4w .. 4w+3, two at a time ------------------------------------------------------------------
    {
        // Rows outside the radius are excluded from the maximum (reference: their outputs are zeroed, and the
        // ReLU floor of the maximum is zero anyway). Done at accumulator start-up: a masked row begins at
        // -3e38 instead of the bias, so it can never win -- the epilogue is then one fma and one max per value.
        const uint32_t vb = vbits[kq];                     // lane-quarter kq holds template point kq
        const size_t gp = g0 + kq;
        bool keep[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) keep[t][i] = ((vb >> (4 * t + i)) & 1u) != 0;   // row 4 kq + i of tile t: neighbour 4t + i
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int tile0 = 4 * wave + 2 * half;
            dclr_f32x4 acc[T][2], acc2[T][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float bv = b3[(tile0 + u) * 16 + c16];
#pragma unroll
                for (int t = 0; t < T; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][u][i] = keep[t][i] ? bv : -3.0e38f;
                    acc2[t][u] = 0.f;
                }
            }
            const float4 *wh = w3p + (size_t)tile0 * F16_KG * 64 + lane;
            flow16_panel<T, false, ABL>(acc, acc2, a_lane, wh, wh + (size_t)(F16_OUT / 16) * F16_KG * 64, F16_KG * 64);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float mx = 0.f;                            // ReLU output floor
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaf(acc2[t][u][i], DCLR_SPLIT_INV, acc[t][u][i]));
                if (gp < total) e_rows[gp * DCLR_E_STRIDE + (tile0 + u) * 16 + c16] = mx;
            }
        }
    }
    // template xyz + zero padding (columns 256..263): 4 points x 8 columns = the first 32 threads
    if (tid < 32) {
        const size_t gp = g0 + (tid >> 3);
        const int c = tid & 7;
        if (gp < total) e_rows[gp * DCLR_E_STRIDE + F16_OUT + c] = c < 3 ? f_rows[gp * DCLR_F_STRIDE + 64 + c] : 0.f;
    }
}

template <int T, int ABL = 0>
void flow16_launch(int pairs, int npoint, int k, float radius, const float *f_rows, const int32_t *knn_idx,
                   const float *pt, const float *ps, const float *w1a, const float *b1, const void *w2p,
                   const float *b2, const void *w3p, const float *b3, float *e_rows, hipStream_t stream,
                   float *zero = nullptr, long long zero_count = 0, uint32_t *overflow = nullptr) {
    const size_t total = (size_t)pairs * npoint;
    hipLaunchKernelGGL((flow16_kernel<T, ABL>), dim3((unsigned)((total + F16_G - 1) / F16_G)), dim3(256), 0, stream, pairs,
                       npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, reinterpret_cast<const float4 *>(w2p), b2,
                       reinterpret_cast<const float4 *>(w3p), b3, e_rows, zero, zero_count, overflow);
}

// ---- the same operator on v_mfma_f32_32x32x16_f16 tiles (round 6) -------------------------------------------------------
// The 16x16x32 form above holds its SIMD's vector issue for 8 of an MFMA's 16 cycles; a 32x32x16 MFMA does twice the work
// for the same 8 (of 32), and a block of 4 points needs half as many of them. Rows: template point p's neighbour s sits in
// row 4 T p + s of the block (T = ceil(k / 4)), 16 T rows padded to RT = ceil(T / 2) tiles of 32 -- k = 30: four points =
// four whole tiles; k = 20: 80 rows + 16 zero rows.
// Radius mask and padding cost nothing in the matrix phases: a neighbour outside the radius (reference: its output column
// is zeroed, deepclr.py:220-223, and the ReLU floor of the maximum over k is zero anyway, deepclr.py:225) and the padding
// rows s >= k are built as COPIES of the point's first neighbour inside the radius -- a duplicate never changes a maximum
// -- and a point with no neighbour inside the radius writes zeros. So every accumulator starts at zero, the bias is added
// after the maximum (max(x_r) + b == max(x_r + b) in every rounding mode) and no per-row predicate exists.
//   layer 2: W2 * H1^T  -- wave w owns channels 32 w .. 32 w + 31 of all RT row tiles; lane = row, registers = channels
//            8 g4 + 4 h + i: half-octets of layer 3's input, split and stored in place;
//   layer 3: H2 * W3^T  -- wave w owns channel tiles 2 w, 2 w + 1, one at a time; lane = channel, registers = rows
//            8 g4 + 4 h + i of each tile: four consecutive rows of ONE point (4 T is a multiple of 4).
constexpr int F32_KG = F16_C / 16;               // 8 k-steps of 16
#ifndef DCLR_FLOW32_ABL
#define DCLR_FLOW32_ABL 0        // timing builds only (results wrong): bit 0 = no phase A (zero rows), bit 1 = no weight loads
#endif

#ifdef DCLR_FLOW_STAMPS          // measurement builds only: cycle stamps of every 64th workgroup (scratch/flow_stamps.py)
constexpr int F32_NSTAMP = 10, F32_STAMP_BLOCKS = 1024;
__device__ unsigned long long g_flow_stamps[F32_STAMP_BLOCKS][4][F32_NSTAMP];
#define F32_STAMP(i) do { unsigned long long v_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_) :: "memory"); st_[i] = v_; } while (0)
#else
#define F32_STAMP(i) do { } while (0)
#endif

template <int RT, bool TRANSPOSED>
__device__ __forceinline__ void flow32_panel(dclr_f32x16 (&acc)[RT], dclr_f32x16 (&acc2)[RT], const char *a_lane,
                                             const float4 *wh_lane, const float4 *wl_lane) {
    // weight fragments three k-steps ahead in four rotating sets, activation fragments one step ahead in two (fixed names
    // through full unrolling, see gemm16.hip); a step's loads and reads are dealt between its 3 RT MFMAs
    dclr_h8 wh[4], wl[4], ah[2][RT], al[2][RT];
    auto read_act = [&](int g, dclr_h8 (&h)[RT], dclr_h8 (&l)[RT]) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            h[t] = dclr_lds_h8(a_lane + t * 32 * F16_STRIDE + 64 * g);
            l[t] = dclr_lds_h8(a_lane + t * 32 * F16_STRIDE + 64 * g + 16);
        }
    };
#if DCLR_FLOW32_ABL & 2
#define F32_FRAG(p) __builtin_bit_cast(dclr_h8, make_float4(1.f, 2.f, 3.f, (float)((size_t)(p) & 255)))
#else
#define F32_FRAG(p) dclr_frag_h8(p)
#endif
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        wh[g] = F32_FRAG(wh_lane + (size_t)g * 64);
        wl[g] = F32_FRAG(wl_lane + (size_t)g * 64);
    }
    // AHEAD costs 8 RT registers: with three workgroups per CU (RT <= 3, 168 registers) it spills, and three waves per SIMD
    // cover an LDS round trip anyway
    constexpr bool AHEAD = RT >= 4;
    if constexpr (AHEAD) read_act(0, ah[0], al[0]);
#pragma unroll
    for (int g = 0; g < F32_KG; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (g + 3 < F32_KG) {
            wh[(g + 3) & 3] = F32_FRAG(wh_lane + (size_t)(g + 3) * 64);
            wl[(g + 3) & 3] = F32_FRAG(wl_lane + (size_t)(g + 3) * 64);
        }
        if constexpr (AHEAD) {
            if (g + 1 < F32_KG) read_act(g + 1, ah[(g + 1) & 1], al[(g + 1) & 1]);
        } else {
            __builtin_amdgcn_sched_barrier(0);
            read_act(g, ah[0], al[0]);
        }
        const dclr_h8 h = wh[g & 3], l = wl[g & 3];
        const dclr_h8 (&xh)[RT] = ah[AHEAD ? (g & 1) : 0], (&xl)[RT] = al[AHEAD ? (g & 1) : 0];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = TRANSPOSED ? dclr_mfma32(h, xh[t], acc[t]) : dclr_mfma32(xh[t], h, acc[t]);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc2[t] = TRANSPOSED ? dclr_mfma32(l, xh[t], acc2[t]) : dclr_mfma32(xh[t], l, acc2[t]);
#pragma unroll
        for (int t = 0; t < RT; ++t) acc2[t] = TRANSPOSED ? dclr_mfma32(h, xl[t], acc2[t]) : dclr_mfma32(xl[t], h, acc2[t]);
        if (AHEAD && g + 1 < F32_KG) {
            // one global load, then RT LDS reads, after every third of the step's MFMAs
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
                if (g + 3 < F32_KG) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RT, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int T>
__global__ __launch_bounds__(256, T <= 6 ? 3 : 2) void flow32_kernel(int pairs, int npoint, int k, float radius,
                                                     const float *__restrict__ f_rows,
                                                     const int32_t *__restrict__ knn_idx,
                                                     const float *__restrict__ pt, const float *__restrict__ ps,
                                                     const float *__restrict__ w1a, const float *__restrict__ b1,
                                                     const float4 *__restrict__ w2p, const float *__restrict__ b2,
                                                     const float4 *__restrict__ w3p, const float *__restrict__ b3,
                                                     float *__restrict__ e_rows, float *__restrict__ zero,
                                                     long long zero_count, uint32_t *overflow) {
    constexpr int RT = (T + 1) / 2, ROWS = 32 * RT, KP = 4 * T;           // KP rows per template point
    __shared__ __attribute__((aligned(16))) char tile[ROWS * F16_STRIDE];
    __shared__ uint32_t vbits[F16_G];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const size_t total = (size_t)pairs * npoint;
#ifdef DCLR_FLOW_STAMPS
    unsigned long long st_[F32_NSTAMP] = {};
#endif
    F32_STAMP(0);
    size_t blk = blockIdx.x;                                              // one pair, one L2 (see flow16_kernel)
    if ((pairs & 7) == 0 && npoint % F16_G == 0) {
        const unsigned per_pair = (unsigned)(npoint / F16_G);
        const unsigned xcd = blockIdx.x & 7u, i = blockIdx.x >> 3;
        blk = (size_t)((i / per_pair) * 8u + xcd) * per_pair + i % per_pair;
    }
    const size_t g0 = blk * F16_G;
    if (zero != nullptr)
        for (long long i = (long long)blockIdx.x * 256 + tid; i < zero_count; i += (long long)gridDim.x * 256) zero[i] = 0.f;

    // ---- phase A: wave p builds the KP layer-1 rows of template point g0 + p (lane = channels 2 lane, 2 lane + 1) ----
    {
        const int p = wave;
        const size_t gp = g0 + p;
        uint32_t bits = 0;
        float peak = 0.f;
        char *const slot = tile + p * KP * F16_STRIDE + 32 * (lane >> 2) + 4 * (lane & 3);
        if (gp < total && !(DCLR_FLOW32_ABL & 1)) {                       // wave-uniform
            // Two dependent L2 round trips behind the neighbour list (one index per lane): lane s fetches neighbour s's
            // position (offsets, norms and the radius mask for all neighbours at once) while the source halves of layer 1
            // (float2 per lane and row) are already on their way -- they depend on the list only.
            const size_t pair = gp / npoint;
            const float *trow = f_rows + gp * DCLR_F_STRIDE;             // template clouds come first
            const float tx = trow[64], ty = trow[65], tz = trow[66];
            const size_t src0 = (pairs + pair) * (size_t)npoint;         // first row of the source cloud
            // a slot the search left unfilled (-1: fewer than k candidates, or NaN coordinates; upstream fails at its
            // .view(2, G, k) there) reads row 0 and counts as outside the radius: no address leaves the source cloud
            const int raw_nb = lane < k ? knn_idx[gp * k + lane] : -1;
            const int my_nb = raw_nb < 0 ? 0 : raw_nb;
            const float4 nbp = *reinterpret_cast<const float4 *>(f_rows + (src0 + my_nb) * DCLR_F_STRIDE + 64);
            float2 psv[KP];
            const float *psrow = ps + pair * (size_t)npoint * F16_C + 2 * lane;
#pragma unroll
            for (int s = 0; s < KP; ++s) {
                const int nb = __builtin_amdgcn_readlane(my_nb, s);       // lanes >= k hold 0: a row nobody uses
                psv[s] = *reinterpret_cast<const float2 *>(psrow + (size_t)nb * F16_C);
            }
            const float2 ptv = *reinterpret_cast<const float2 *>(pt + gp * F16_C + 2 * lane);
            const float2 bv = *reinterpret_cast<const float2 *>(b1 + 2 * lane);
            const float wa0 = w1a[(2 * lane) * 3 + 0], wa1 = w1a[(2 * lane) * 3 + 1], wa2 = w1a[(2 * lane) * 3 + 2];
            const float wb0 = w1a[(2 * lane + 1) * 3 + 0], wb1 = w1a[(2 * lane + 1) * 3 + 1],
                        wb2 = w1a[(2 * lane + 1) * 3 + 2];
            const float base0 = ptv.x + bv.x, base1 = ptv.y + bv.y;
            const float my_dx = nbp.x - tx, my_dy = nbp.y - ty, my_dz = nbp.z - tz;
            const float norm = sqrtf(my_dx * my_dx + my_dy * my_dy + my_dz * my_dz);
            bits = (uint32_t)__ballot(raw_nb >= 0 && (!(radius > 0.f) || norm < radius));      // k <= 32: the low word
            if (bits != 0) {                                              // wave-uniform
                // rows inside the radius first; then every other row (outside, unfilled, beyond k) as a copy of the first
                // of them
                const int keep = __builtin_ctz(bits);
                dclr_h2 keep_hi = {(_Float16)0.f, (_Float16)0.f}, keep_lo = keep_hi;
#pragma unroll
                for (int s = 0; s < KP; ++s) {
                    if ((bits >> s) & 1u) {                               // wave-uniform
                        const float dx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_dx), s));
                        const float dy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_dy), s));
                        const float dz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_dz), s));
                        float v0 = base0 + psv[s].x, v1 = base1 + psv[s].y;
                        v0 = fmaf(wa0, dx, v0); v0 = fmaf(wa1, dy, v0); v0 = fmaf(wa2, dz, v0);
                        v1 = fmaf(wb0, dx, v1); v1 = fmaf(wb1, dy, v1); v1 = fmaf(wb2, dz, v1);
                        dclr_h2 hi, lo;
                        dclr_split2_relu(v0, v1, hi, lo, peak);
                        if (s == keep) { keep_hi = hi; keep_lo = lo; }
                        *reinterpret_cast<dclr_h2 *>(slot + s * F16_STRIDE) = hi;
                        *reinterpret_cast<dclr_h2 *>(slot + s * F16_STRIDE + 16) = lo;
                    }
                }
                if (bits != (uint32_t)((1ull << KP) - 1ull)) {
#pragma unroll
                    for (int s = 0; s < KP; ++s) {
                        if (!((bits >> s) & 1u)) {
                            *reinterpret_cast<dclr_h2 *>(slot + s * F16_STRIDE) = keep_hi;
                            *reinterpret_cast<dclr_h2 *>(slot + s * F16_STRIDE + 16) = keep_lo;
                        }
                    }
                }
            }
        }
        if (bits == 0) {                                                  // no point, or nothing inside the radius
#pragma unroll
            for (int s = 0; s < KP; ++s) {
                *reinterpret_cast<uint32_t *>(slot + s * F16_STRIDE) = 0u;
                *reinterpret_cast<uint32_t *>(slot + s * F16_STRIDE + 16) = 0u;
            }
        }
        if constexpr (ROWS > F16_G * KP) {                                // the 16 rows no point owns: zeros (finite sums; their
                                                                          // layer-2 rows are relu(b2), their layer-3 rows ignored)
            static_assert(ROWS - F16_G * KP == 16, "tail rows");
            char *dst = tile + (F16_G * KP + (tid >> 4)) * F16_STRIDE + 32 * (tid & 15);
            *reinterpret_cast<float4 *>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(dst + 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (lane == 0) vbits[p] = bits;
        dclr_report_overflow(overflow, peak);
    }
    F32_STAMP(1);
    __syncthreads();
    F32_STAMP(2);

    const char *a_lane = tile + j * F16_STRIDE + 32 * h;                  // octet 2 g + h of row j: + 64 g (+ tile offset)

    // ---- phase B: layer 2 (128 -> 128), wave w owns channel tile w ------------------------------------------------
    {
        dclr_f32x16 acc[RT], acc2[RT];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bv = *reinterpret_cast<const float4 *>(b2 + 32 * wave + 8 * g4 + 4 * h);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                acc[t][4 * g4 + 0] = bv.x; acc[t][4 * g4 + 1] = bv.y; acc[t][4 * g4 + 2] = bv.z; acc[t][4 * g4 + 3] = bv.w;
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) acc2[t] = dclr_zero16();
        const float4 *wh = w2p + (size_t)wave * F32_KG * 64 + lane;
        flow32_panel<RT, true>(acc, acc2, a_lane, wh, wh + (size_t)(F16_C / 32) * F32_KG * 64);
        F32_STAMP(3);
        __syncthreads();                                   // every wave has consumed the layer-1 rows
        F32_STAMP(4);
        float peak = 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                dclr_h4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    dclr_h2 a, c;
                    dclr_split2_relu(fmaf(acc2[t][4 * g4 + i], DCLR_SPLIT_INV, acc[t][4 * g4 + i]),
                                     fmaf(acc2[t][4 * g4 + i + 1], DCLR_SPLIT_INV, acc[t][4 * g4 + i + 1]), a, c, peak);
                    hi[i] = a[0]; hi[i + 1] = a[1]; lo[i] = c[0]; lo[i + 1] = c[1];
                }
                char *dst = tile + (32 * t + j) * F16_STRIDE + 32 * (4 * wave + g4) + 8 * h;
                *reinterpret_cast<dclr_h4 *>(dst) = hi;
                *reinterpret_cast<dclr_h4 *>(dst + 16) = lo;
            }
        }
        dclr_report_overflow(overflow, peak);
    }
    F32_STAMP(5);
    __syncthreads();
    F32_STAMP(6);

    // ---- phase C: layer 3 (128 -> 256) + max over each point's rows; wave w owns channel tiles 2 w, 2 w + 1 --------
    {
        uint32_t vb[F16_G];
#pragma unroll
        for (int p = 0; p < F16_G; ++p) vb[p] = vbits[p];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int ct = 2 * wave + half;
            dclr_f32x16 acc[RT], acc2[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) { acc[t] = dclr_zero16(); acc2[t] = dclr_zero16(); }
            const float4 *wh = w3p + (size_t)ct * F32_KG * 64 + lane;
            flow32_panel<RT, false>(acc, acc2, a_lane, wh, wh + (size_t)(F16_OUT / 32) * F32_KG * 64);
            if (half == 0) F32_STAMP(7); else F32_STAMP(8);
            float mx[F16_G];
#pragma unroll
            for (int p = 0; p < F16_G; ++p) mx[p] = -3.0e38f;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int r0 = 32 * t + 8 * g4;                       // rows r0 + 4 h + i; compile-time after unrolling
                    const int p0 = r0 / KP, p1 = (r0 + 4) / KP;
                    if (p0 >= F16_G) continue;
                    float m4 = fmaf(acc2[t][4 * g4], DCLR_SPLIT_INV, acc[t][4 * g4]);
#pragma unroll
                    for (int i = 1; i < 4; ++i) m4 = fmaxf(m4, fmaf(acc2[t][4 * g4 + i], DCLR_SPLIT_INV, acc[t][4 * g4 + i]));
                    if (p0 == p1) mx[p0] = fmaxf(mx[p0], m4);
                    else {
                        mx[p0] = fmaxf(mx[p0], h == 0 ? m4 : -3.0e38f);
                        if (p1 < F16_G) mx[p1] = fmaxf(mx[p1], h == 1 ? m4 : -3.0e38f);
                    }
                }
            }
            const float bv = b3[32 * ct + j];
#pragma unroll
            for (int p = 0; p < F16_G; ++p) {
                float m = fmaxf(mx[p], __shfl_xor(mx[p], 32));
                m = vb[p] != 0 ? fmaxf(m + bv, 0.f) : 0.f;               // bias and ReLU commute with the maximum
                const size_t gp = g0 + p;
                if ((p >> 1) == h && gp < total) e_rows[gp * DCLR_E_STRIDE + 32 * ct + j] = m;
            }
        }
    }
    if (tid < 32) {                                                       // template xyz + zero padding (columns 256..263)
        const size_t gp = g0 + (tid >> 3);
        const int c = tid & 7;
        if (gp < total) e_rows[gp * DCLR_E_STRIDE + F16_OUT + c] = c < 3 ? f_rows[gp * DCLR_F_STRIDE + 64 + c] : 0.f;
    }
#ifdef DCLR_FLOW_STAMPS
    F32_STAMP(9);
    if ((blockIdx.x & 63u) == 17u && (blockIdx.x >> 6) < F32_STAMP_BLOCKS && lane == 0)
        for (int i = 0; i < F32_NSTAMP; ++i) g_flow_stamps[blockIdx.x >> 6][wave][i] = st_[i];
#endif
}

template <int T>
void flow32_launch(int pairs, int npoint, int k, float radius, const float *f_rows, const int32_t *knn_idx,
                   const float *pt, const float *ps, const float *w1a, const float *b1, const void *w2p,
                   const float *b2, const void *w3p, const float *b3, float *e_rows, hipStream_t stream,
                   float *zero, long long zero_count, uint32_t *overflow) {
    const size_t total = (size_t)pairs * npoint;
    hipLaunchKernelGGL((flow32_kernel<T>), dim3((unsigned)((total + F16_G - 1) / F16_G)), dim3(256), 0, stream, pairs,
                       npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, reinterpret_cast<const float4 *>(w2p), b2,
                       reinterpret_cast<const float4 *>(w3p), b3, e_rows, zero, zero_count, overflow);
}

}  // namespace

#ifdef DCLR_FLOW_STAMPS
extern "C" int dclr_debug_flow_stamps(unsigned long long *host_out, int blocks) {
    if (blocks > F32_STAMP_BLOCKS) blocks = F32_STAMP_BLOCKS;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_flow_stamps), sizeof(unsigned long long) * blocks * 4 * F32_NSTAMP);
}
#endif

// Which MFMA tile the split-f16 flow kernel of this build runs k neighbours on, i.e. the `width` to pack its layer-2 /
// layer-3 weights with (dclr_pack_weight_f16): 32 (v_mfma_f32_32x32x16_f16, flow32_kernel) from 29 neighbours up -- four
// points x 32 rows are whole tiles there: 1168-1226 us against 1230-1316 per 256 x 512 points at k = 29..32 -- and 16
// (v_mfma_f32_16x16x32_f16, flow16_kernel) below, where the rows pad: k = 25..28 (112 rows in 128) 1154-1178 against
// 1120-1135, k = 21..24 (96 in 96, but three tiles of 32 against six of 16 per wave) 932-940 against 896-930, k = 20 (80 rows
// in 96) 632-658 against 517-541 per 80 x 1024 points (profiles/NOTES.md, round 6).
// -DDCLR_FLOW_TILE16 / -DDCLR_FLOW_TILE32 (A/B builds) force one form for every k.
static bool flow_uses_tile32(int k) {
#if defined(DCLR_FLOW_TILE16)
    return false;
#elif defined(DCLR_FLOW_TILE32)
    return true;
#else
    return (k + 3) / 4 >= 8;
#endif
}
extern "C" int dclr_flow_f16_tile(int k) { return flow_uses_tile32(k) ? 32 : 16; }

extern "C" int dclr_flow_embedding_fused_f16(int pairs, int npoint, int k, float radius, const float *f_rows,
                                             const int32_t *knn_idx, const float *pt, const float *ps,
                                             const float *w1a, const float *b1, const void *w2p, const float *b2,
                                             const void *w3p, const float *b3, float *e_rows, dclr_stream_t stream) {
    return dclr_x_flow_embedding_fused_f16(pairs, npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, w2p, b2, w3p, b3, e_rows,
                                           nullptr, 0, nullptr, stream);
}

int dclr_x_flow_embedding_fused_f16(int pairs, int npoint, int k, float radius, const float *f_rows, const int32_t *knn_idx,
                                    const float *pt, const float *ps, const float *w1a, const float *b1, const void *w2p,
                                    const float *b2, const void *w3p, const float *b3, float *e_rows, float *zero,
                                    long long zero_count, uint32_t *overflow, dclr_stream_t stream) {
    DCLR_REQUIRE(zero == nullptr || zero_count > 0);
    DCLR_REQUIRE(pairs > 0 && npoint > 0 && f_rows && knn_idx && pt && ps && w1a && b1 && w2p && b2 && w3p &&
                 b3 && e_rows);
    if (k < 1 || k > 32) return DCLR_E_UNSUPPORTED;
    DCLR_REQUIRE(((uintptr_t)w2p & 15) == 0 && ((uintptr_t)w3p & 15) == 0 && ((uintptr_t)pt & 7) == 0 &&
                 ((uintptr_t)ps & 7) == 0 && ((uintptr_t)b2 & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
#ifdef DCLR_ABLATION
    // Measurement builds only (-DDCLR_ABLATION, a separate .so selected with DCLR_LIB; scratch/flow_probe.py): the
    // timing-only variants return WRONG rows. The product library is compiled without them and reads no environment.
    static const int abl = getenv("DCLR_FLOW_ABL") ? atoi(getenv("DCLR_FLOW_ABL")) : 0;      // k = 20 only
    if (abl != 0 && (k + 3) / 4 == 5) {
#define DCLR_FLOW16_ABL(A) case A: flow16_launch<5, A>(pairs, npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, w2p, b2, w3p, b3, e_rows, st); break
        switch (abl) { DCLR_FLOW16_ABL(1); DCLR_FLOW16_ABL(2); DCLR_FLOW16_ABL(3); DCLR_FLOW16_ABL(4); DCLR_FLOW16_ABL(6); DCLR_FLOW16_ABL(8); DCLR_FLOW16_ABL(9); DCLR_FLOW16_ABL(12); default: break; }
#undef DCLR_FLOW16_ABL
        return dclr_launch_status();
    }
#endif
#define DCLR_FLOW16_CASE(T) case T: flow16_launch<T>(pairs, npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, w2p, b2, w3p, b3, e_rows, st, zero, zero_count, overflow); break
#define DCLR_FLOW32_CASE(T) case T: flow32_launch<T>(pairs, npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, w2p, b2, w3p, b3, e_rows, st, zero, zero_count, overflow); break
    if (flow_uses_tile32(k)) {
        switch ((k + 3) / 4) {
            DCLR_FLOW32_CASE(1); DCLR_FLOW32_CASE(2); DCLR_FLOW32_CASE(3); DCLR_FLOW32_CASE(4);
            DCLR_FLOW32_CASE(5); DCLR_FLOW32_CASE(6); DCLR_FLOW32_CASE(7); DCLR_FLOW32_CASE(8);
            default: return DCLR_E_UNSUPPORTED;
        }
    } else {
        switch ((k + 3) / 4) {
            DCLR_FLOW16_CASE(1); DCLR_FLOW16_CASE(2); DCLR_FLOW16_CASE(3); DCLR_FLOW16_CASE(4);
            DCLR_FLOW16_CASE(5); DCLR_FLOW16_CASE(6); DCLR_FLOW16_CASE(7); DCLR_FLOW16_CASE(8);
            default: return DCLR_E_UNSUPPORTED;
        }
    }
#undef DCLR_FLOW16_CASE
#undef DCLR_FLOW32_CASE
    return dclr_launch_status();
}
