// Pose-head conv chain on split-fp16 MFMA (mma16f.h): weight packing and the fused five-layer kernel.
//
// Same operator as head_fused_kernel in gemm.hip (reference: OutputSimple.forward,
// /root/reference/deepclr/models/deepclr.py:284-287: Conv1dMultiLayer 259->256->256->512->512->1024 with
// ReLU after every layer, then max over points); the contraction runs on v_mfma_f32_32x32x16_f16 with
// every operand split into f16 hi/lo halves, three instructions per product, f32 accumulation.
#include "mma16f.h"

namespace {

// ---- weight packing ------------------------------------------------------------------------------
// w (n_out, k_in) f32 row-major -> hi plane | lo plane, each np * kp halves in fragment order
// [(tile * KG + g) * 64 + lane][q], tile width = `width` (16 or 32) outputs, 64 / width k-octets per step.
__global__ __launch_bounds__(256) void pack_weight_f16_kernel(int n_out, int k_in, const float *__restrict__ w,
                                                              const int32_t *__restrict__ kmap, int kp, int np,
                                                              int width, _Float16 *__restrict__ packed) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)np * kp) return;
    const int q = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const size_t grp = e >> 9;                              // tile * KG + g
    const int octets = 64 / width, kstep = 8 * octets;
    const int kg = kp / kstep;
    const int g = (int)(grp % kg), tile = (int)(grp / kg);
    const int n = tile * width + (lane % width);
    const int k = g * kstep + 8 * (lane / width) + q;
    const int col = kmap ? kmap[k] : (k < k_in ? k : -1);
    float v = 0.f;
    if (n < n_out && col >= 0 && col < k_in) v = w[(size_t)n * k_in + col];
    _Float16 hi, lo;
    dclr_split(v, hi, lo);
    packed[e] = hi;
    packed[(size_t)np * kp + e] = lo;
}

// ---- fused conv chain ----------------------------------------------------------------------------------
// Workgroup = 32 points through all layers, 8 waves (two per SIMD). Activations live in LDS as k-octets
// (16 B hi | 16 B lo), ping-ponging between two buffers. Hidden layers compute W * X^T: the accumulator
// lane is a point and its registers are 4 x 4 consecutive output channels, which are split and stored as
// half-octets of the next layer's input (two ds_write_b64 per 4 channels). The last layer computes
// X * W^T: lane = channel, registers = the 32 points, so the max over points is in-register plus one
// cross-half exchange, then one atomic max per channel.
// Per k-step (16 values) a wave issues 4 weight-fragment loads (its 2 column tiles x hi/lo) for 6 MFMAs of
// 32 cycles: with 8 waves that asks for ~85 B/clk/CU from L2, above the ~64 B/clk a CU's vector memory
// path delivers, so the kernel is bound by weight delivery (4.2 MB per workgroup), not by the matrix pipe.
constexpr int H16_WAVES = 8, H16_ROWS = 32, H16_MAX_LAYERS = 8, H16_MAX_WIDTH = 512;
constexpr int H16_BUF = H16_ROWS * dclr_split_stride(H16_MAX_WIDTH);        // bytes per activation buffer

struct Head16Params {
    int n_layers;
    int k_in;                               // valid input columns of x (multiple of 8)
    int k[H16_MAX_LAYERS];                  // padded input width of layer l (multiple of 16)
    int n[H16_MAX_LAYERS];                  // output width (multiple of 32)
    const float4 *w[H16_MAX_LAYERS];        // packed hi plane; lo plane follows at n * k / 8 fragments
    const float *b[H16_MAX_LAYERS];
};

template <bool LAST, int NT>
__device__ __forceinline__ void head16_panel(dclr_f32x16 (&acc)[2], dclr_f32x16 (&acc2)[2], const char *a_lane,
                                             int kg, const float4 *wh_lane, const float4 *wl_lane, int tile_stride) {
    // two named weight-fragment sets: set 1 is in flight while set 0 feeds the MFMAs (see mma.h)
    dclr_h8 h0[NT], l0[NT], h1[NT], l1[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        h0[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride);
        l0[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride);
    }
    auto step = [&](int g, const dclr_h8 (&wh)[NT], const dclr_h8 (&wl)[NT]) {
        const dclr_h8 ah = dclr_lds_h8(a_lane + 64 * g), al = dclr_lds_h8(a_lane + 64 * g + 16);
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = LAST ? dclr_mfma32(ah, wh[u], acc[u]) : dclr_mfma32(wh[u], ah, acc[u]);
#pragma unroll
        for (int u = 0; u < NT; ++u) acc2[u] = LAST ? dclr_mfma32(ah, wl[u], acc2[u]) : dclr_mfma32(wl[u], ah, acc2[u]);
#pragma unroll
        for (int u = 0; u < NT; ++u) acc2[u] = LAST ? dclr_mfma32(al, wh[u], acc2[u]) : dclr_mfma32(wh[u], al, acc2[u]);
    };
    int g = 0;
    for (; g + 2 <= kg; g += 2) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            h1[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride + (size_t)(g + 1) * 64);
            l1[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride + (size_t)(g + 1) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        step(g, h0, l0);
        const int gn = g + 2 < kg ? g + 2 : g + 1;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            h0[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride + (size_t)gn * 64);
            l0[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride + (size_t)gn * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        step(g + 1, h1, l1);
    }
    if (g < kg) step(g, h0, l0);
}

__global__ __launch_bounds__(H16_WAVES * 64) void head16_kernel(Head16Params prm, const float *__restrict__ x, int ldx,
                                                                float *__restrict__ colmax, int rows_per_group) {
    __shared__ __attribute__((aligned(16))) char act[2][H16_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.x * H16_ROWS;

    // stage the 32 input rows: one k-octet (8 floats -> 16 B hi + 16 B lo) per thread and step
    {
        const int stride = dclr_split_stride(prm.k[0]);
        const int octets = prm.k[0] / 8, valid = prm.k_in / 8;
        for (int e = tid; e < H16_ROWS * octets; e += H16_WAVES * 64) {
            const int r = e / octets, o = e - r * octets;
            dclr_h8 hi, lo;
            if (o < valid) {
                const float4 v0 = *reinterpret_cast<const float4 *>(x + (size_t)(m0 + r) * ldx + 8 * o);
                const float4 v1 = *reinterpret_cast<const float4 *>(x + (size_t)(m0 + r) * ldx + 8 * o + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    _Float16 a, b;
                    dclr_split(v[q], a, b);
                    hi[q] = a; lo[q] = b;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) { hi[q] = (_Float16)0.f; lo[q] = (_Float16)0.f; }
            }
            char *dst = &act[0][r * stride + 32 * o];
            *reinterpret_cast<dclr_h8 *>(dst) = hi;
            *reinterpret_cast<dclr_h8 *>(dst + 16) = lo;
        }
    }
    __syncthreads();

    for (int l = 0; l < prm.n_layers; ++l) {
        const char *in = act[l & 1];
        char *out = act[(l & 1) ^ 1];
        const int kp = prm.k[l], n = prm.n[l], kg = kp / 16;
        const int in_stride = dclr_split_stride(kp), out_stride = dclr_split_stride(n);
        const bool last = l == prm.n_layers - 1;
        const char *a_lane = in + j * in_stride + 32 * h;
        const int n_tiles = n / 32;
        const size_t plane = (size_t)n_tiles * kg * 64;                 // fragments per plane
        for (int t0 = wave; t0 < n_tiles; t0 += 2 * H16_WAVES) {
            const bool two = t0 + H16_WAVES < n_tiles;                  // wave-uniform
            dclr_f32x16 acc[2] = {dclr_zero16(), dclr_zero16()}, acc2[2] = {dclr_zero16(), dclr_zero16()};
            const float4 *wh = prm.w[l] + (size_t)t0 * kg * 64 + lane;
            const float4 *wl = wh + plane;
            const int ts = H16_WAVES * kg * 64;
            if (last) {
                if (two) head16_panel<true, 2>(acc, acc2, a_lane, kg, wh, wl, ts);
                else head16_panel<true, 1>(acc, acc2, a_lane, kg, wh, wl, ts);
            } else {
                if (two) head16_panel<false, 2>(acc, acc2, a_lane, kg, wh, wl, ts);
                else head16_panel<false, 1>(acc, acc2, a_lane, kg, wh, wl, ts);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                const int tt = t0 + u * H16_WAVES;
                if (!last) {
                    // lane = point j; registers 4 g4 + i = channel 32 tt + 8 g4 + 4 h + i
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int ch = 32 * tt + 8 * g4 + 4 * h;
                        const float4 bv = *reinterpret_cast<const float4 *>(prm.b[l] + ch);
                        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                        dclr_h4 hi, lo;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = fmaxf(fmaf(acc2[u][4 * g4 + i], DCLR_SPLIT_INV, acc[u][4 * g4 + i]) + bb[i], 0.f);
                            _Float16 a, b;
                            dclr_split(v, a, b);
                            hi[i] = a; lo[i] = b;
                        }
                        char *dst = out + j * out_stride + 32 * (4 * tt + g4) + 8 * h;
                        *reinterpret_cast<dclr_h4 *>(dst) = hi;
                        *reinterpret_cast<dclr_h4 *>(dst + 16) = lo;
                    }
                } else {
                    // lane = channel 32 tt + j; registers = points
                    const int col = 32 * tt + j;
                    float mx = -3.0e38f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaf(acc2[u][r], DCLR_SPLIT_INV, acc[u][r]));
                    mx = fmaxf(mx + prm.b[l][col], 0.f);                // bias and ReLU commute with the maximum
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

}  // namespace

extern "C" int dclr_pack_weight_f16(int n_out, int k_in, const float *w, const int32_t *kmap, int kp, int width,
                                    void *packed, dclr_stream_t stream) {
    DCLR_REQUIRE(n_out > 0 && k_in > 0 && w && packed && (width == 16 || width == 32));
    const int kstep = 8 * (64 / width);
    DCLR_REQUIRE(kp > 0 && kp % kstep == 0 && (kmap || kp >= k_in));
    const int np = (n_out + width - 1) / width * width;
    const size_t total = (size_t)np * kp;
    hipLaunchKernelGGL(pack_weight_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_out, k_in, w, kmap, kp, np, width, reinterpret_cast<_Float16 *>(packed));
    return dclr_launch_status();
}

extern "C" int dclr_head_conv_fused_f16(int m, int n_layers, int k_in, const int *k_host, const int *n_host,
                                        const void *const *w_packed_host, const float *const *bias_host,
                                        const float *x, int ldx, float *colmax, int rows_per_group,
                                        dclr_stream_t stream) {
    DCLR_REQUIRE(m > 0 && n_layers >= 1 && k_host && n_host && w_packed_host && bias_host && x && colmax);
    DCLR_REQUIRE(m % H16_ROWS == 0 && rows_per_group > 0 && rows_per_group % H16_ROWS == 0 && m % rows_per_group == 0);
    DCLR_REQUIRE(k_in > 0 && k_in % 8 == 0 && ldx % 4 == 0 && ldx >= k_in && k_in <= k_host[0] && ((uintptr_t)x & 15) == 0);
    if (n_layers > H16_MAX_LAYERS) return DCLR_E_UNSUPPORTED;
    Head16Params prm{};
    prm.n_layers = n_layers;
    prm.k_in = k_in;
    for (int l = 0; l < n_layers; ++l) {
        DCLR_REQUIRE(w_packed_host[l] && bias_host[l] && k_host[l] > 0 && n_host[l] > 0);
        DCLR_REQUIRE(k_host[l] % 16 == 0 && n_host[l] % 32 == 0 && ((uintptr_t)w_packed_host[l] & 15) == 0 &&
                     ((uintptr_t)bias_host[l] & 15) == 0);
        if (l > 0) DCLR_REQUIRE(k_host[l] == n_host[l - 1]);
        if (k_host[l] > H16_MAX_WIDTH || (l + 1 < n_layers && n_host[l] > H16_MAX_WIDTH)) return DCLR_E_UNSUPPORTED;
        prm.k[l] = k_host[l];
        prm.n[l] = n_host[l];
        prm.w[l] = reinterpret_cast<const float4 *>(w_packed_host[l]);
        prm.b[l] = bias_host[l];
    }
    hipLaunchKernelGGL(head16_kernel, dim3(m / H16_ROWS), dim3(H16_WAVES * 64), 0, (hipStream_t)stream, prm, x, ldx,
                       colmax, rows_per_group);
    return dclr_launch_status();
}
