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
// Layers 2 and 3 run on fp32 MFMA (v_mfma_f32_16x16x4_f32). A workgroup serves 4 template points;
// MFMA row tile t holds neighbours 4t..4t+3 of the 4 points (row 4p + i), so k = 20 fills 5 tiles
// exactly (a tile per point would pad 20 rows to 32). In the 16x16 accumulator layout lane-quarter p
// holds point p and its 4 registers are the 4 neighbours: the max over the k neighbours is an
// in-register maximum over 4 registers x T tiles, no cross-lane traffic at all. 42 KB of LDS per
// workgroup at k = 20: three workgroups per CU overlap each other's gather, barriers and MFMA phases.
#include "mma.h"

namespace {

constexpr int FL_G = 4;                         // template points per workgroup
constexpr int FL_C = 128;                       // hidden width of layers 1 and 2
constexpr int FL_OUT = 256;
constexpr int FL_STRIDE = dclr_lds_stride(FL_C);   // 132
constexpr int FL_KG = FL_C / 16;                // 8 k-groups of 16

// T = ceil(k / 4) row tiles of 16 rows.
template <int T>
__global__ __launch_bounds__(256, T <= 5 ? 3 : 2) void flow_kernel(int pairs, int npoint, int k, float radius,
                                                   const float *__restrict__ f_rows,
                                                   const int32_t *__restrict__ knn_idx,
                                                   const float *__restrict__ pt, const float *__restrict__ ps,
                                                   const float *__restrict__ w1a, const float *__restrict__ b1,
                                                   const float4 *__restrict__ w2p, const float *__restrict__ b2,
                                                   const float4 *__restrict__ w3p, const float *__restrict__ b3,
                                                   float *__restrict__ e_rows) {
    __shared__ __attribute__((aligned(16))) float tile[T * 16 * FL_STRIDE];
    __shared__ uint32_t vbits[FL_G];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, c16 = lane & 15;
    const size_t total = (size_t)pairs * npoint;
    const size_t g0 = (size_t)blockIdx.x * FL_G;

    // ---- phase A: wave w gathers the rows of template point g0 + w --------------------------------
    {
        const int p = wave;
        const size_t gp = g0 + p;
        const bool live = gp < total;                                   // wave-uniform
        uint32_t bits = 0;
        int s_done = 0;
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
            // an unfilled search slot (-1) reads row 0 and is masked like a neighbour beyond the radius (see flow16.hip)
            const int raw_nb = lane < k ? knn_idx[gp * k + lane] : 0;
            const uint32_t filled = (uint32_t)__ballot(lane < k && raw_nb >= 0);
            const int my_nb = raw_nb < 0 ? 0 : raw_nb;
            const size_t src0 = (pairs + pair) * (size_t)npoint;         // first row of the source cloud
#pragma unroll 4
            for (int s = 0; s < k; ++s) {
                const int nb = __builtin_amdgcn_readlane(my_nb, s);
                const float *srow = f_rows + (src0 + nb) * DCLR_F_STRIDE;
                const float dx = srow[64] - tx, dy = srow[65] - ty, dz = srow[66] - tz;
                const float2 psv = *reinterpret_cast<const float2 *>(ps + (pair * npoint + nb) * FL_C + 2 * lane);
                float v0 = base0 + psv.x, v1 = base1 + psv.y;
                v0 = fmaf(wa0, dx, v0); v0 = fmaf(wa1, dy, v0); v0 = fmaf(wa2, dz, v0);
                v1 = fmaf(wb0, dx, v1); v1 = fmaf(wb1, dy, v1); v1 = fmaf(wb2, dz, v1);
                const int row = (s >> 2) * 16 + 4 * p + (s & 3);
                *reinterpret_cast<float2 *>(&tile[row * FL_STRIDE + 2 * lane]) =
                    make_float2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
                const float norm = sqrtf(dx * dx + dy * dy + dz * dz);
                if (!(radius > 0.f) || norm < radius) bits |= 1u << s;
            }
            bits &= filled;
            s_done = k;
        }
        for (int s = s_done; s < 4 * T; ++s) {                          // padding rows (k % 4 != 0, or no point)
            const int row = (s >> 2) * 16 + 4 * p + (s & 3);
            *reinterpret_cast<float2 *>(&tile[row * FL_STRIDE + 2 * lane]) = make_float2(0.f, 0.f);
        }
        if (lane == 0) vbits[p] = bits;
    }
    __syncthreads();

    const float *a_lds = &tile[c16 * FL_STRIDE + 4 * kq];

    // ---- phase B: layer 2 (128 -> 128), wave w owns channels 32w .. 32w+31 (two 16-column tiles) ---
    {
        dclr_f32x4 acc[T][2];
#pragma unroll
        for (int t = 0; t < T; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; }
        const float4 *wl = w2p + (size_t)(2 * wave) * FL_KG * 64 + lane;
        dclr_mma16_panel<T, 2>(acc, a_lds, FL_STRIDE, FL_KG, wl, FL_KG * 64);
        __syncthreads();                                   // every wave has consumed layer-1 rows
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = (2 * wave + u) * 16 + c16;
            const float bv = b2[col];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    tile[(t * 16 + 4 * kq + i) * FL_STRIDE + col] = fmaxf(acc[t][u][i] + bv, 0.f);
        }
    }
    __syncthreads();

    // ---- phase C: layer 3 (128 -> 256) + radius mask + max over neighbours; wave w owns channels
    //      64w .. 64w+63 (four 16-column tiles) ------------------------------------------------------
    {
        dclr_f32x4 acc[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = 0.f;
        const float4 *wl = w3p + (size_t)(4 * wave) * FL_KG * 64 + lane;
        dclr_mma16_panel<T, 4>(acc, a_lds, FL_STRIDE, FL_KG, wl, FL_KG * 64);
        const uint32_t vb = vbits[kq];                     // lane-quarter kq holds template point kq
        const size_t gp = g0 + kq;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int col = (4 * wave + u) * 16 + c16;
            const float bv = b3[col];
            float mx = 0.f;                                // ReLU output floor; masked rows contribute 0
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[t][u][i] + bv;     // row 4*kq + i of tile t: neighbour 4t + i
                    mx = ((vb >> (4 * t + i)) & 1u) ? fmaxf(mx, v) : mx;
                }
            if (gp < total) e_rows[gp * DCLR_E_STRIDE + col] = mx;
        }
    }
    // template xyz + zero padding (columns 256..263): 4 points x 8 columns = the first 32 threads
    if (tid < 32) {
        const size_t gp = g0 + (tid >> 3);
        const int c = tid & 7;
        if (gp < total) e_rows[gp * DCLR_E_STRIDE + FL_OUT + c] = c < 3 ? f_rows[gp * DCLR_F_STRIDE + 64 + c] : 0.f;
    }
}

template <int T>
void flow_launch(int pairs, int npoint, int k, float radius, const float *f_rows, const int32_t *knn_idx,
                 const float *pt, const float *ps, const float *w1a, const float *b1, const float *w2p,
                 const float *b2, const float *w3p, const float *b3, float *e_rows, hipStream_t stream) {
    const size_t total = (size_t)pairs * npoint;
    hipLaunchKernelGGL((flow_kernel<T>), dim3((unsigned)((total + FL_G - 1) / FL_G)), dim3(256), 0, stream, pairs,
                       npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, reinterpret_cast<const float4 *>(w2p), b2,
                       reinterpret_cast<const float4 *>(w3p), b3, e_rows);
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
    hipStream_t st = (hipStream_t)stream;
#define DCLR_FLOW_CASE(T) case T: flow_launch<T>(pairs, npoint, k, radius, f_rows, knn_idx, pt, ps, w1a, b1, w2p, b2, w3p, b3, e_rows, st); break
    switch ((k + 3) / 4) {
        DCLR_FLOW_CASE(1); DCLR_FLOW_CASE(2); DCLR_FLOW_CASE(3); DCLR_FLOW_CASE(4);
        DCLR_FLOW_CASE(5); DCLR_FLOW_CASE(6); DCLR_FLOW_CASE(7); DCLR_FLOW_CASE(8);
        default: return DCLR_E_UNSUPPORTED;
    }
#undef DCLR_FLOW_CASE
    return dclr_launch_status();
}
