// Fused flow embedding for gfx950.
//
// Replaces MotionEmbeddingBase.forward (/root/reference/deepclr/models/deepclr.py:201-231) after the
// kNN grouping (142-173), append_features=True: for every template point, its k source neighbours are
// turned into rows [pos_diff(3) | template feat(64) | source feat(64)], pushed through
// Conv1dMultiLayer 131->128->128->256 (ReLU after every layer, helper.py:37-38), rows with
// |pos_diff| >= radius are zeroed (deepclr.py:220-223) and the rows are max-pooled (225).
// The reference materialises (G, 131, k) and three activation tensors; here one workgroup keeps
// the rows of four template points in LDS from gather to max.
//
// Layer 1 is split by linearity:  W1 x = W1a pos_diff + W1b feat_t + W1c feat_s. The two feature
// products are per point, not per (point, neighbour), and arrive precomputed in pt / ps
// (dclr_linear); per row only a 512-byte gather, three FMAs per channel and the ReLU remain.
// Layers 2 and 3 run on fp32 MFMA: each template point is one 32-row tile (rows >= k are padding),
// wave w owns 32 (layer 2) / 64 (layer 3) output channels for all four tiles, so the max over
// the k neighbours is an in-register maximum plus one cross-half exchange.
#include "mma.h"

namespace {

constexpr int FL_G = 4;                         // template points (= 32-row tiles) per workgroup
constexpr int FL_C = 128;                       // hidden width of layers 1 and 2
constexpr int FL_OUT = 256;
constexpr int FL_STRIDE = dclr_lds_stride(FL_C);   // 132
constexpr int FL_KG = FL_C / 8;                 // 16 k-groups

__global__ __launch_bounds__(256, 2) void flow_kernel(int pairs, int npoint, int k, float radius,
                                                   const float *__restrict__ f_rows,
                                                   const int32_t *__restrict__ knn_idx,
                                                   const float *__restrict__ pt, const float *__restrict__ ps,
                                                   const float *__restrict__ w1a, const float *__restrict__ b1,
                                                   const float4 *__restrict__ w2p, const float *__restrict__ b2,
                                                   const float4 *__restrict__ w3p, const float *__restrict__ b3,
                                                   float *__restrict__ e_rows) {
    __shared__ __attribute__((aligned(16))) float tile[FL_G * 32 * FL_STRIDE];
    __shared__ uint32_t vbits[FL_G];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const size_t total = (size_t)pairs * npoint;
    const size_t g0 = (size_t)blockIdx.x * FL_G;

    // ---- phase A: wave w gathers the rows of template point g0 + w -------------------------------
    {
        const size_t gp = g0 + wave;
        const bool live = gp < total;                                   // wave-uniform
        float *rows = &tile[wave * 32 * FL_STRIDE];
        uint32_t bits = 0;
        if (live) {
            const size_t pair = gp / npoint;
            const float *trow = f_rows + gp * DCLR_F_STRIDE;             // template clouds come first
            const float tx = trow[64], ty = trow[65], tz = trow[66];
            const float2 ptv = *reinterpret_cast<const float2 *>(pt + gp * FL_C + 2 * lane);
            const float2 bv = *reinterpret_cast<const float2 *>(b1 + 2 * lane);
            const float wa0 = w1a[(2 * lane) * 3 + 0], wa1 = w1a[(2 * lane) * 3 + 1], wa2 = w1a[(2 * lane) * 3 + 2];
            const float wb0 = w1a[(2 * lane + 1) * 3 + 0], wb1 = w1a[(2 * lane + 1) * 3 + 1],
                        wb2 = w1a[(2 * lane + 1) * 3 + 2];
            const float base0 = ptv.x + bv.x, base1 = ptv.y + bv.y;
            const int my_nb = lane < k ? knn_idx[gp * k + lane] : 0;
            const size_t src0 = (pairs + pair) * (size_t)npoint;         // first row of the source cloud
#pragma unroll 4
            for (int r = 0; r < k; ++r) {
                const int nb = __builtin_amdgcn_readlane(my_nb, r);
                const float *srow = f_rows + (src0 + nb) * DCLR_F_STRIDE;
                const float dx = srow[64] - tx, dy = srow[65] - ty, dz = srow[66] - tz;
                const float2 psv = *reinterpret_cast<const float2 *>(ps + (pair * npoint + nb) * FL_C + 2 * lane);
                float v0 = base0 + psv.x, v1 = base1 + psv.y;
                v0 = fmaf(wa0, dx, v0); v0 = fmaf(wa1, dy, v0); v0 = fmaf(wa2, dz, v0);
                v1 = fmaf(wb0, dx, v1); v1 = fmaf(wb1, dy, v1); v1 = fmaf(wb2, dz, v1);
                *reinterpret_cast<float2 *>(&rows[r * FL_STRIDE + 2 * lane]) =
                    make_float2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
                const float norm = sqrtf(dx * dx + dy * dy + dz * dz);
                if (!(radius > 0.f) || norm < radius) bits |= 1u << r;
            }
        }
        const int r_first_pad = live ? k : 0;
        for (int r = r_first_pad; r < 32; ++r)
            *reinterpret_cast<float2 *>(&rows[r * FL_STRIDE + 2 * lane]) = make_float2(0.f, 0.f);
        if (lane == 0) vbits[wave] = bits;
    }
    __syncthreads();

    const float *a_lds = &tile[j * FL_STRIDE + 4 * h];

    // ---- phase B: layer 2 (128 -> 128), wave w owns channels 32w .. 32w+31 ------------------------
    {
        dclr_f32x16 acc[FL_G][1];
#pragma unroll
        for (int t = 0; t < FL_G; ++t) acc[t][0] = dclr_zero16();
        const float4 *wl = w2p + (size_t)wave * FL_KG * 64 + lane;
#pragma unroll 4
        for (int g = 0; g < FL_KG; ++g) dclr_mma_group<FL_G, 1>(acc, a_lds, FL_STRIDE, g, wl + g * 64, 0);
        __syncthreads();                                   // every wave has consumed layer-1 rows
        const int col = wave * 32 + j;
        const float bv = b2[col];
#pragma unroll
        for (int t = 0; t < FL_G; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(t * 32 + dclr_acc_row(r, h)) * FL_STRIDE + col] = fmaxf(acc[t][0][r] + bv, 0.f);
    }
    __syncthreads();

    // ---- phase C: layer 3 (128 -> 256) + radius mask + max over neighbours -----------------------
    {
        dclr_f32x16 acc[FL_G][2];
#pragma unroll
        for (int t = 0; t < FL_G; ++t) { acc[t][0] = dclr_zero16(); acc[t][1] = dclr_zero16(); }
        const float4 *wl = w3p + (size_t)(2 * wave) * FL_KG * 64 + lane;
#pragma unroll 2
        for (int g = 0; g < FL_KG; ++g) dclr_mma_group<FL_G, 2>(acc, a_lds, FL_STRIDE, g, wl + g * 64, FL_KG * 64);
#pragma unroll
        for (int t = 0; t < FL_G; ++t) {
            const size_t gp = g0 + t;
            const uint32_t bits = vbits[t];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int col = (2 * wave + u) * 32 + j;
                const float bv = b3[col];
                float mx = 0.f;                            // ReLU output floor; masked rows contribute 0
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[t][u][r] + bv;
                    mx = ((bits >> dclr_acc_row(r, h)) & 1u) ? fmaxf(mx, v) : mx;
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (h == 0 && gp < total) e_rows[gp * DCLR_E_STRIDE + col] = mx;
            }
        }
    }
    // template xyz + zero padding (columns 256..263) of this wave's point
    {
        const size_t gp = g0 + wave;
        if (gp < total && lane < 8) {
            const float *trow = f_rows + gp * DCLR_F_STRIDE;
            e_rows[gp * DCLR_E_STRIDE + FL_OUT + lane] = lane < 3 ? trow[64 + lane] : 0.f;
        }
    }
}

}  // namespace

extern "C" int dclr_flow_embedding_fused(int pairs, int npoint, int k, float radius, const float *f_rows,
                                         const int32_t *knn_idx, const float *pt, const float *ps,
                                         const float *w1a, const float *b1, const float *w2p, const float *b2,
                                         const float *w3p, const float *b3, float *e_rows, dclr_stream_t stream) {
    DCLR_REQUIRE(pairs > 0 && npoint > 0 && f_rows && knn_idx && pt && ps && w1a && b1 && w2p && b2 && w3p &&
                 b3 && e_rows);
    if (k < 1 || k > 32) return DCLR_E_UNSUPPORTED;
    DCLR_REQUIRE(((uintptr_t)w2p & 15) == 0 && ((uintptr_t)w3p & 15) == 0 && ((uintptr_t)pt & 7) == 0 &&
                 ((uintptr_t)ps & 7) == 0);
    const size_t total = (size_t)pairs * npoint;
    hipLaunchKernelGGL(flow_kernel, dim3((unsigned)((total + FL_G - 1) / FL_G)), dim3(256), 0, (hipStream_t)stream,
                       pairs, npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1,
                       reinterpret_cast<const float4 *>(w2p), b2, reinterpret_cast<const float4 *>(w3p), b3, e_rows);
    return dclr_launch_status();
}
