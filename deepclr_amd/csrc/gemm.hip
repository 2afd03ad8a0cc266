// Dense MLP layers for gfx950: weight packing, Y = act(X W^T + b) on fp32 MFMA, small FC tail.
//
// Replaces the cuDNN/cuBLAS 1x1 Conv1d / Linear calls behind the reference's helper modules
// (/root/reference/deepclr/models/helper.py:11-65: affine + ReLU after every layer) for the pose head
// (OutputSimple.forward, /root/reference/deepclr/models/deepclr.py:284-294) and for the per-point
// part of the flow embedding's first layer. Activations are point-major rows, so a 1x1 conv is a
// plain row-major GEMM and the max over points is a column maximum.
#include "mma.h"

namespace {

// ---- weight packing ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(int n_out, int k_in, const float *__restrict__ w,
                                                          const int32_t *__restrict__ kmap, int kp, int np,
                                                          float *__restrict__ packed) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)np * kp) return;
    const int q = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const size_t grp = e >> 8;                 // ntile * KG + g
    const int kg = kp / 8;
    const int g = (int)(grp % kg), ntile = (int)(grp / kg);
    const int n = ntile * 32 + (lane & 31);
    const int k = g * 8 + 4 * (lane >> 5) + q;
    int col = kmap ? kmap[k] : (k < k_in ? k : -1);
    float v = 0.f;
    if (n < n_out && col >= 0 && col < k_in) v = w[(size_t)n * k_in + col];
    packed[e] = v;
}

// Same, for the 16x16x4 MFMA layout (mma.h): 16-column tiles, K groups of 16.
__global__ __launch_bounds__(256) void pack_weight16_kernel(int n_out, int k_in, const float *__restrict__ w,
                                                            const int32_t *__restrict__ kmap, int kp, int np,
                                                            float *__restrict__ packed) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)np * kp) return;
    const int q = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const size_t grp = e >> 8;                 // ntile * KG16 + g
    const int kg = kp / 16;
    const int g = (int)(grp % kg), ntile = (int)(grp / kg);
    const int n = ntile * 16 + (lane & 15);
    const int k = g * 16 + 4 * (lane >> 4) + q;
    int col = kmap ? kmap[k] : (k < k_in ? k : -1);
    float v = 0.f;
    if (n < n_out && col >= 0 && col < k_in) v = w[(size_t)n * k_in + col];
    packed[e] = v;
}

// ---- Y = act(X W^T + b) ----------------------------------------------------------------------------
// Workgroup: 64 rows x 128 columns, 4 waves; wave w owns column tile w (32 columns) and both 32-row
// tiles. X is staged through LDS in 32-wide K chunks (double buffered, one barrier per chunk);
// packed weights come straight from L2 as coalesced 1 KiB fragments, prefetched one chunk ahead.
constexpr int LIN_BM = 64, LIN_BK = 32, LIN_WAVES = 4;
constexpr int LIN_STRIDE = dclr_lds_stride(LIN_BK);       // 36 floats

// MT = 32-row tiles per workgroup (1 or 2): layers whose grid would otherwise leave CUs with a single
// workgroup (nothing to overlap its barriers and load latencies with) use the 32-row variant.
template <int MT>
__global__ __launch_bounds__(LIN_WAVES * 64) void linear_kernel(int m, int n, int kp, const float *__restrict__ x,
                                                                int ldx, const float4 *__restrict__ wp,
                                                                const float *__restrict__ bias, int relu,
                                                                float *__restrict__ y, int ldy,
                                                                float *__restrict__ colmax, int rows_per_group,
                                                                int m_split, const float4 *__restrict__ wp_hi,
                                                                float *__restrict__ y_hi) {
    constexpr int BM = 32 * MT;
    __shared__ __attribute__((aligned(16))) float tile[2][BM * LIN_STRIDE];
    if ((int)blockIdx.x * BM >= m_split) {                // second problem of a pair: rows m_split.. use their own
        wp = wp_hi;                                       // weights and write to their own output (row 0 = m_split)
        y = y_hi - (size_t)m_split * ldy;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int ntile = blockIdx.y * LIN_WAVES + wave;
    const int n_tiles = (n + 31) / 32;
    const bool active = ntile < n_tiles;                  // wave-uniform
    const int kg_total = kp / 8;
    const int n_chunks = (kp + LIN_BK - 1) / LIN_BK;

    // staging role: thread -> (row, 16-byte column) of the BM x 32 chunk, MT rows per thread
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const float *xs0 = x + (size_t)(m0 + srow) * ldx + scol;
    const float *xs1 = xs0 + (size_t)32 * ldx;

    auto fetch = [&](int chunk, float4 &r0, float4 &r1) {
        const int kc = chunk * LIN_BK + scol;
        r0 = r1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kc < kp) {
            r0 = *reinterpret_cast<const float4 *>(xs0 + chunk * LIN_BK);
            if constexpr (MT == 2) r1 = *reinterpret_cast<const float4 *>(xs1 + chunk * LIN_BK);
        }
    };
    auto stash = [&](int buf, const float4 &r0, const float4 &r1) {
        *reinterpret_cast<float4 *>(&tile[buf][srow * LIN_STRIDE + scol]) = r0;
        if constexpr (MT == 2) *reinterpret_cast<float4 *>(&tile[buf][(srow + 32) * LIN_STRIDE + scol]) = r1;
    };

    dclr_f32x16 acc[MT][1];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t][0] = dclr_zero16();

    const float4 *w_lane = wp + ((size_t)(active ? ntile : 0) * kg_total) * 64 + lane;

    float4 r0, r1;
    fetch(0, r0, r1);
    stash(0, r0, r1);
    __syncthreads();

    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) fetch(c + 1, r0, r1);
        if (active) {
            const int g0 = c * (LIN_BK / 8);
            const int g1 = g0 + LIN_BK / 8 < kg_total ? g0 + LIN_BK / 8 : kg_total;
            const float *a_lds = &tile[buf][j * LIN_STRIDE + 4 * h];
            dclr_mma_panel<MT, 1>(acc, a_lds, LIN_STRIDE, 0, g1 - g0, w_lane + (size_t)g0 * 64, 0);
        }
        if (c + 1 < n_chunks) stash(buf ^ 1, r0, r1);
        __syncthreads();
    }

    if (!active) return;
    const int col = ntile * 32 + j;
    const float bv = (bias && col < n) ? bias[col] : 0.f;
    if (colmax) {
        float mx = 0.f;                                   // relu is required: values >= 0
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[t][0][r] + bv);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (h == 0 && col < n)
            atomicMax(reinterpret_cast<unsigned int *>(colmax + (size_t)(m0 / rows_per_group) * n + col),
                      __float_as_uint(mx));
        return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + t * 32 + dclr_acc_row(r, h);
            float v = acc[t][0][r] + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (col < n) y[(size_t)row * ldy + col] = v;
        }
}

// ---- fused conv chain of the pose head ----------------------------------------------------------------
// One workgroup keeps 32 points through ALL 1x1-conv layers (reference: OutputSimple.forward,
// deepclr.py:286-287: Conv1dMultiLayer then max over points): activations ping-pong between two LDS
// buffers, packed weights stream from L2 (every CU walks the layers in step, so a layer's weights are
// L2-hot), the last layer's column maximum goes out through atomic max. Versus one launch per layer
// this removes the activation round trips (~50 MB per batch), four launches and the tail effect of the
// narrow first layers. 8 waves: two per SIMD, so one wave's LDS/L2 waits hide behind the other's MFMAs.
constexpr int HEAD_WAVES = 8, HEAD_ROWS = 32, HEAD_MAX_LAYERS = 8, HEAD_MAX_WIDTH = 512;
constexpr int HEAD_BUF = HEAD_ROWS * dclr_lds_stride(HEAD_MAX_WIDTH);        // floats per activation buffer

struct HeadParams {
    int n_layers;
    int k[HEAD_MAX_LAYERS];                 // padded input width of layer l (multiple of 8)
    int n[HEAD_MAX_LAYERS];                 // output width (multiple of 32)
    const float4 *w[HEAD_MAX_LAYERS];       // packed (n, k) fragments
    const float *b[HEAD_MAX_LAYERS];
};

__global__ __launch_bounds__(HEAD_WAVES * 64) void head_fused_kernel(HeadParams prm, const float *__restrict__ x,
                                                                     int ldx, float *__restrict__ colmax,
                                                                     int rows_per_group) {
    __shared__ __attribute__((aligned(16))) float act[2][HEAD_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.x * HEAD_ROWS;

    // stage the 32 input rows (k[0] columns, 16-byte pieces)
    {
        const int stride = dclr_lds_stride(prm.k[0]);
        const int pieces = prm.k[0] / 4;
        for (int e = tid; e < HEAD_ROWS * pieces; e += HEAD_WAVES * 64) {
            const int r = e / pieces, c4 = e - r * pieces;
            *reinterpret_cast<float4 *>(&act[0][r * stride + 4 * c4]) =
                *reinterpret_cast<const float4 *>(x + (size_t)(m0 + r) * ldx + 4 * c4);
        }
    }
    __syncthreads();

    for (int l = 0; l < prm.n_layers; ++l) {
        const float *in = act[l & 1];
        float *out = act[(l & 1) ^ 1];
        const int kp = prm.k[l], n = prm.n[l], kg = kp / 8;
        const int in_stride = dclr_lds_stride(kp), out_stride = dclr_lds_stride(n);
        const bool last = l == prm.n_layers - 1;
        const float *a_lds = in + j * in_stride + 4 * h;
        const int n_tiles = n / 32;
        // wave w owns column tiles w, w + 8, ...; two at a time share one pass over K
        for (int t0 = wave; t0 < n_tiles; t0 += 2 * HEAD_WAVES) {
            const bool two = t0 + HEAD_WAVES < n_tiles;                 // wave-uniform
            dclr_f32x16 acc[1][2];
            acc[0][0] = dclr_zero16();
            acc[0][1] = dclr_zero16();
            const float4 *w_lane = prm.w[l] + (size_t)t0 * kg * 64 + lane;
            if (two) {
                dclr_mma_panel<1, 2>(acc, a_lds, in_stride, 0, kg, w_lane, HEAD_WAVES * kg * 64);
            } else {
                dclr_f32x16 acc1[1][1];
                acc1[0][0] = dclr_zero16();
                dclr_mma_panel<1, 1>(acc1, a_lds, in_stride, 0, kg, w_lane, 0);
                acc[0][0] = acc1[0][0];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                const int col = (t0 + u * HEAD_WAVES) * 32 + j;
                const float bv = prm.b[l][col];
                if (!last) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        out[dclr_acc_row(r, h) * out_stride + col] = fmaxf(acc[0][u][r] + bv, 0.f);
                } else {
                    float mx = 0.f;                                     // ReLU floor
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[0][u][r] + bv);
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    if (h == 0)
                        atomicMax(reinterpret_cast<unsigned int *>(colmax + (size_t)(m0 / rows_per_group) * n + col),
                                  __float_as_uint(mx));
                }
            }
        }
        __syncthreads();
    }
}

// ---- FC tail: a handful of rows (one per scan pair) -------------------------------------------------
// One wave per output column, 4 columns per workgroup. The <= FC_ROWS input rows are staged in LDS
// once per workgroup; a wave keeps the whole weight row in flight (K/64 coalesced loads issued back
// to back -- a serial load-use loop costs one L2 round trip per 64 weights) and reduces the per-lane
// partial sums with DPP adds.
constexpr int FC_WAVES = 4, FC_ROWS = 8, FC_MAX_K = 1024, FC_CHUNKS = FC_MAX_K / 64;

__device__ __forceinline__ float fc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(FC_WAVES * 64) void fc_kernel(int m, int n, int k, const float *__restrict__ x,
                                                           const float *__restrict__ w,
                                                           const float *__restrict__ bias, int act,
                                                           float *__restrict__ y, const uint32_t *poison) {
    __shared__ float xs[FC_ROWS][FC_MAX_K];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // poison (the LAST layer of the fused dense stages on the split-f16 path): the sticky word the kernels before this one
    // set when an activation was clamped at the f16 range. Set -> the outputs are written as NaN: a clamped forward hands
    // out poses that cannot be mistaken for results, whether or not the host looks at the word again (it does, at its
    // next entry into the model). One scalar load, requested first, needed last.
    const bool bad = poison != nullptr && __hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
    const int col = blockIdx.x * FC_WAVES + wave;
    const bool live = col < n;                                       // wave-uniform
    float wv[FC_CHUNKS];
    if (live) {
        const float *wr = w + (size_t)col * k;
#pragma unroll
        for (int c = 0; c < FC_CHUNKS; ++c) wv[c] = c * 64 + lane < k ? wr[c * 64 + lane] : 0.f;
    }
    const float bv = (live && bias) ? bias[col] : 0.f;
    // blockIdx.y strides over the row blocks: many rows (large batches of small clouds) spread over the grid
    // instead of queueing inside one workgroup per column group
    for (int r0 = blockIdx.y * FC_ROWS; r0 < m; r0 += gridDim.y * FC_ROWS) {
        const int rows = m - r0 < FC_ROWS ? m - r0 : FC_ROWS;
        __syncthreads();
        if ((k & 3) == 0 && ((uintptr_t)x & 15) == 0) {              // 16 bytes per lane and request (a quarter of the instructions)
            for (int r = 0; r < FC_ROWS; ++r)
                for (int kk = 4 * tid; kk < k; kk += FC_WAVES * 64 * 4)
                    *reinterpret_cast<float4 *>(&xs[r][kk]) =
                        r < rows ? *reinterpret_cast<const float4 *>(x + (size_t)(r0 + r) * k + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int r = 0; r < FC_ROWS; ++r)
                for (int kk = tid; kk < k; kk += FC_WAVES * 64)
                    xs[r][kk] = r < rows ? x[(size_t)(r0 + r) * k + kk] : 0.f;
        }
        __syncthreads();
        if (!live) continue;
        float acc[FC_ROWS];
#pragma unroll
        for (int r = 0; r < FC_ROWS; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < FC_CHUNKS; ++c) {
            if (c * 64 < k) {                                        // uniform
#pragma unroll
                for (int r = 0; r < FC_ROWS; ++r) acc[r] = fmaf(wv[c], xs[r][c * 64 + lane < k ? c * 64 + lane : 0], acc[r]);
            }
        }
        // The eight rows' wave sums in ten exchanges instead of 48: the butterfly (xor 32, 16, ..., 1) of fc_wave_sum, but a
        // lane keeps only half of its rows at each of the first three steps and hands the other half to its partner --
        // the same pairs are added at every level (a + b == b + a bit for bit), so every row's sum is the one
        // fc_wave_sum(acc[r]) returns. Afterwards lane 32 b5 + 16 b4 + 8 b3 (+ 0..7) holds row 4 b5 + 2 b4 + b3.
        static_assert(FC_ROWS == 8, "three halving steps");
        float h4[4], h2[2], v;
        {
            const bool up = (lane & 32) != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) h4[i] = (up ? acc[4 + i] : acc[i]) + __shfl_xor(up ? acc[i] : acc[4 + i], 32);
        }
        {
            const bool up = (lane & 16) != 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) h2[i] = (up ? h4[2 + i] : h4[i]) + __shfl_xor(up ? h4[i] : h4[2 + i], 16);
        }
        {
            const bool up = (lane & 8) != 0;
            v = (up ? h2[1] : h2[0]) + __shfl_xor(up ? h2[0] : h2[1], 8);
        }
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        {
            const int r = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
            v += bv;
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = col == 0 ? 1.f / (1.f + expf(-v)) : (col < 4 ? tanhf(v) : v);
            else if (act == 3) v = col == 3 ? 1.f / (1.f + expf(-v)) : (col > 3 ? tanhf(v) : v);
            if (bad) v = __builtin_nanf("");
            if ((lane & 7) == 0 && r < rows) y[(size_t)(r0 + r) * n + col] = v;
        }
    }
}

}  // namespace

extern "C" int dclr_pack_weight(int n_out, int k_in, const float *w, const int32_t *kmap, int kp, int np,
                                float *packed, dclr_stream_t stream) {
    DCLR_REQUIRE(n_out > 0 && k_in > 0 && w && packed && kp > 0 && np > 0);
    DCLR_REQUIRE(kp % 8 == 0 && np % 32 == 0 && np >= n_out && (kmap || kp >= k_in));
    const size_t total = (size_t)np * kp;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_out, k_in, w, kmap, kp, np, packed);
    return dclr_launch_status();
}

extern "C" int dclr_pack_weight16(int n_out, int k_in, const float *w, const int32_t *kmap, int kp, int np,
                                  float *packed, dclr_stream_t stream) {
    DCLR_REQUIRE(n_out > 0 && k_in > 0 && w && packed && kp > 0 && np > 0);
    DCLR_REQUIRE(kp % 16 == 0 && np % 16 == 0 && np >= n_out && (kmap || kp >= k_in));
    const size_t total = (size_t)np * kp;
    hipLaunchKernelGGL(pack_weight16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_out, k_in, w, kmap, kp, np, packed);
    return dclr_launch_status();
}

extern "C" int dclr_linear(int m, int n, int kp, const float *x, int ldx, const float *w_packed,
                           const float *bias, int relu, float *y, int ldy, float *colmax,
                           int rows_per_group, dclr_stream_t stream) {
    DCLR_REQUIRE(m > 0 && n > 0 && kp > 0 && x && w_packed && (y || colmax));
    DCLR_REQUIRE(m % LIN_BM == 0 && kp % 8 == 0 && ldx % 4 == 0 && ldx >= kp);
    DCLR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_packed & 15) == 0);
    if (colmax) DCLR_REQUIRE(relu && rows_per_group > 0 && rows_per_group % LIN_BM == 0 && m % rows_per_group == 0);
    else DCLR_REQUIRE(ldy >= n);
    const int n_tiles = (n + 31) / 32;
    const unsigned gy = (n_tiles + LIN_WAVES - 1) / LIN_WAVES;
    DCLR_REQUIRE(gy <= 65535);
    if ((size_t)(m / LIN_BM) * gy < 1024)       // fewer than 4 workgroups per CU: halve the row tile
        hipLaunchKernelGGL((linear_kernel<1>), dim3(m / 32, gy), dim3(LIN_WAVES * 64), 0, (hipStream_t)stream, m, n, kp,
                           x, ldx, reinterpret_cast<const float4 *>(w_packed), bias, relu, y, ldy, colmax, rows_per_group,
                           0x7FFFFFFF, nullptr, nullptr);
    else
        hipLaunchKernelGGL((linear_kernel<2>), dim3(m / LIN_BM, gy), dim3(LIN_WAVES * 64), 0, (hipStream_t)stream, m, n,
                           kp, x, ldx, reinterpret_cast<const float4 *>(w_packed), bias, relu, y, ldy, colmax,
                           rows_per_group, 0x7FFFFFFF, nullptr, nullptr);
    return dclr_launch_status();
}

// Two products over one row range in one launch: rows [0, m_each) with w_a -> y_a, rows [m_each, 2 m_each) with
// w_b -> y_b (the template / source halves of flow layer 1). No bias, no activation.
extern "C" int dclr_linear_pair(int m_each, int n, int kp, const float *x, int ldx, const float *w_a, const float *w_b,
                                float *y_a, float *y_b, int ldy, dclr_stream_t stream) {
    DCLR_REQUIRE(m_each > 0 && n > 0 && kp > 0 && x && w_a && w_b && y_a && y_b);
    DCLR_REQUIRE(m_each % LIN_BM == 0 && kp % 8 == 0 && ldx % 4 == 0 && ldx >= kp && ldy >= n);
    DCLR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_a & 15) == 0 && ((uintptr_t)w_b & 15) == 0);
    const int n_tiles = (n + 31) / 32;
    const unsigned gy = (n_tiles + LIN_WAVES - 1) / LIN_WAVES;
    DCLR_REQUIRE(gy <= 65535);
    hipLaunchKernelGGL((linear_kernel<1>), dim3(2 * m_each / 32, gy), dim3(LIN_WAVES * 64), 0, (hipStream_t)stream,
                       2 * m_each, n, kp, x, ldx, reinterpret_cast<const float4 *>(w_a), nullptr, 0, y_a, ldy, nullptr, 0,
                       m_each, reinterpret_cast<const float4 *>(w_b), y_b);
    return dclr_launch_status();
}

extern "C" int dclr_head_conv_fused(int m, int n_layers, const int *k_host, const int *n_host,
                                    const float *const *w_packed_host, const float *const *bias_host, const float *x,
                                    int ldx, float *colmax, int rows_per_group, dclr_stream_t stream) {
    DCLR_REQUIRE(m > 0 && n_layers >= 1 && k_host && n_host && w_packed_host && bias_host && x && colmax);
    DCLR_REQUIRE(m % HEAD_ROWS == 0 && rows_per_group > 0 && rows_per_group % HEAD_ROWS == 0 && m % rows_per_group == 0);
    DCLR_REQUIRE(ldx % 4 == 0 && ldx >= k_host[0] && ((uintptr_t)x & 15) == 0);
    if (n_layers > HEAD_MAX_LAYERS) return DCLR_E_UNSUPPORTED;
    HeadParams prm{};
    prm.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        DCLR_REQUIRE(w_packed_host[l] && bias_host[l] && k_host[l] > 0 && n_host[l] > 0);
        DCLR_REQUIRE(k_host[l] % 8 == 0 && n_host[l] % 32 == 0 && ((uintptr_t)w_packed_host[l] & 15) == 0);
        if (l > 0) DCLR_REQUIRE(k_host[l] == n_host[l - 1]);
        if (k_host[l] > HEAD_MAX_WIDTH || (l + 1 < n_layers && n_host[l] > HEAD_MAX_WIDTH)) return DCLR_E_UNSUPPORTED;
        prm.k[l] = k_host[l];
        prm.n[l] = n_host[l];
        prm.w[l] = reinterpret_cast<const float4 *>(w_packed_host[l]);
        prm.b[l] = bias_host[l];
    }
    hipLaunchKernelGGL(head_fused_kernel, dim3(m / HEAD_ROWS), dim3(HEAD_WAVES * 64), 0, (hipStream_t)stream, prm, x, ldx,
                       colmax, rows_per_group);
    return dclr_launch_status();
}

extern "C" int dclr_fc(int m, int n, int k, const float *x, const float *w, const float *bias, int act,
                       float *y, dclr_stream_t stream) {
    return dclr_x_fc(m, n, k, x, w, bias, act, y, nullptr, stream);
}

int dclr_x_fc(int m, int n, int k, const float *x, const float *w, const float *bias, int act, float *y,
              const uint32_t *poison, dclr_stream_t stream) {
    DCLR_REQUIRE(m > 0 && n > 0 && k > 0 && x && w && y && act >= 0 && act <= 3);
    if (k > FC_MAX_K) return DCLR_E_UNSUPPORTED;
    const int row_blocks = (m + FC_ROWS - 1) / FC_ROWS;
    hipLaunchKernelGGL(fc_kernel, dim3((n + FC_WAVES - 1) / FC_WAVES, row_blocks < 64 ? row_blocks : 64), dim3(FC_WAVES * 64), 0,
                       (hipStream_t)stream, m, n, k, x, w, bias, act, y, poison);
    return dclr_launch_status();
}
