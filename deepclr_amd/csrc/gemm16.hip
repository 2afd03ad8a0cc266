// Pose-head conv chain on split-fp16 MFMA (mma16f.h): weight packing and the fused five-layer kernel.
//
// Same operator as head_fused_kernel in gemm.hip (reference: OutputSimple.forward,
// /root/reference/deepclr/models/deepclr.py:284-287: Conv1dMultiLayer 259->256->256->512->512->1024 with
// ReLU after every layer, then max over points); the contraction runs on v_mfma_f32_32x32x16_f16 with
// every operand split into f16 hi/lo halves, three instructions per product, f32 accumulation.
#include "mma16f.h"

#ifndef H16_PIPE
#define H16_PIPE 3               // 3 (default since round 4): the activation fragments of step g + 1 are read while the MFMAs of
                                 // step g run, and a step's loads / reads are dealt between its MFMAs (518 -> 496 us per 80 pairs);
                                 // 1 = the read-ahead alone (515); 0 = round 3's loop (loads and reads in front of the MFMAs)
#endif
#ifndef H16_ABL
#define H16_ABL 0                // timing probes of scratch/head_abl.sh only; the product library is built with 0
#endif

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
// Workgroup = 32 * MT points through all layers, 8 waves (two per SIMD). Activations live in ONE LDS buffer
// as k-octets (16 B hi | 16 B lo) and are overwritten in place: a layer's outputs stay in the accumulators
// until every wave has finished reading its inputs (wave w owns output tiles w and w + 8: at most
// 2 x MT tiles = the whole layer for widths <= 512), then they are split and stored over them.
// Hidden layers compute W * X^T: the accumulator lane is a point and its registers are 4 x 4 consecutive
// output channels, i.e. half-octets of the next layer's input (two ds_write_b64 per 4 channels). The last
// layer computes X * W^T: lane = channel, registers = points, so the max over points is in-register plus one
// cross-half exchange, then one atomic max per channel.
// Weight delivery bounds the 32-point form (4.2 MB per workgroup through a ~64 B/clk vector memory path:
// ~30 us, against ~22 us of MFMA); with MT = 2 the same stream feeds twice the MFMAs, the kernel turns
// matrix-bound, needs only 128 CUs for 8192 rows and so keeps its speed while the sampler holds CUs.
constexpr int H16_WAVES = 8, H16_MAX_LAYERS = 8, H16_MAX_WIDTH = 512;

struct Head16Params {
    int n_layers;
    int k_in;                               // valid input columns of x (multiple of 8)
    int k[H16_MAX_LAYERS];                  // padded input width of layer l (multiple of 16)
    int n[H16_MAX_LAYERS];                  // output width (multiple of 32)
    const float4 *w[H16_MAX_LAYERS];        // packed hi plane; lo plane follows at n * k / 8 fragments
    const float *b[H16_MAX_LAYERS];
    uint32_t *overflow;                     // NULL, or the word that receives 1 when an activation left the f16 range
};

// K loop for NT weight tiles x MT point tiles. LAST: activations are the A operand (X * W^T).
template <bool LAST, int NT, int MT>
__device__ __forceinline__ void head16_panel(dclr_f32x16 (&acc)[2][MT], dclr_f32x16 (&acc2)[2][MT], const char *a_lane,
                                             int tile_bytes, int kg, const float4 *wh_lane, const float4 *wl_lane,
                                             int tile_stride) {
    // Four weight-fragment sets in rotation, loads issued three k-steps ahead: a step is only 6-12 MFMAs
    // (200-400 cycles), an L2 round trip under load is several times that, and the second wave of the SIMD
    // stalls on its own fragments at about the same moments -- one step of lead leaves the matrix pipe idle
    // about half the time. (Fixed set names through full unrolling: a rotating index makes hipcc fold the
    // prefetch back into load-wait-use, see mma.h.)
    dclr_h8 wh[4][NT], wl[4][NT];
    auto fetch = [&](int g, dclr_h8 (&h)[NT], dclr_h8 (&l)[NT]) {
#if H16_ABL & 1                  // timing probe: every step re-reads the first fragment (no stream from L2)
        g = 0;
#endif
#pragma unroll
        for (int u = 0; u < NT; ++u) {
#if H16_ABL & 2                  // timing probe: no weight loads at all
            h[u] = __builtin_bit_cast(dclr_h8, make_float4((float)g, 1.f, 2.f, (float)u));
            l[u] = h[u];
#else
            h[u] = dclr_frag_h8(wh_lane + (size_t)u * tile_stride + (size_t)g * 64);
            l[u] = dclr_frag_h8(wl_lane + (size_t)u * tile_stride + (size_t)g * 64);
#endif
        }
    };
#if H16_PIPE
    // (H16_PIPE) the activation fragments of step g + 1 are read from LDS while the MFMAs of step g run (two register sets),
    // and the step's loads and reads are dealt between its MFMAs instead of standing in front of them
    dclr_h8 ah2[2][MT], al2[2][MT];
    auto read_act = [&](int g, dclr_h8 (&h)[MT], dclr_h8 (&l)[MT]) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            h[t] = dclr_lds_h8(a_lane + t * tile_bytes + 64 * g);
            l[t] = dclr_lds_h8(a_lane + t * tile_bytes + 64 * g + 16);
        }
    };
    auto mfmas = [&](const dclr_h8 (&h)[NT], const dclr_h8 (&l)[NT], const dclr_h8 (&xh)[MT], const dclr_h8 (&xl)[MT]) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[u][t] = LAST ? dclr_mfma32(xh[t], h[u], acc[u][t]) : dclr_mfma32(h[u], xh[t], acc[u][t]);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc2[u][t] = LAST ? dclr_mfma32(xh[t], l[u], acc2[u][t]) : dclr_mfma32(l[u], xh[t], acc2[u][t]);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc2[u][t] = LAST ? dclr_mfma32(xl[t], h[u], acc2[u][t]) : dclr_mfma32(h[u], xl[t], acc2[u][t]);
    };
    fetch(0, wh[0], wl[0]);
    fetch(kg > 1 ? 1 : 0, wh[1], wl[1]);
    fetch(kg > 2 ? 2 : 0, wh[2], wl[2]);
    read_act(0, ah2[0], al2[0]);
    for (int g = 0; g < kg; g += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (g + i < kg) {                                          // wave-uniform
                const int ahead = g + i + 3 < kg ? g + i + 3 : kg - 1;  // clamped: the load stays unconditional
                const int nxt = g + i + 1 < kg ? g + i + 1 : kg - 1;
                __builtin_amdgcn_sched_barrier(0);
                fetch(ahead, wh[(i + 3) & 3], wl[(i + 3) & 3]);
                read_act(nxt, ah2[(i + 1) & 1], al2[(i + 1) & 1]);
                mfmas(wh[i], wl[i], ah2[i & 1], al2[i & 1]);
#if H16_PIPE & 2
                // deal the 2 NT loads and 2 MT reads between the 3 NT MT MFMAs
#pragma unroll
                for (int q = 0; q < 2 * NT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, (3 * NT * MT) / (2 * NT + 1), 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, (2 * MT + 2 * NT - 1) / (2 * NT), 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT * MT, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#else
    auto step = [&](int g, const dclr_h8 (&h)[NT], const dclr_h8 (&l)[NT]) {
        dclr_h8 ah[MT], al[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#if H16_ABL & 4                  // timing probe: no activation reads from LDS
            ah[t] = __builtin_bit_cast(dclr_h8, make_float4((float)g, 1.f, 2.f, (float)t));
            al[t] = ah[t];
#else
            ah[t] = dclr_lds_h8(a_lane + t * tile_bytes + 64 * g);
            al[t] = dclr_lds_h8(a_lane + t * tile_bytes + 64 * g + 16);
#endif
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc[u][t] = LAST ? dclr_mfma32(ah[t], h[u], acc[u][t]) : dclr_mfma32(h[u], ah[t], acc[u][t]);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc2[u][t] = LAST ? dclr_mfma32(ah[t], l[u], acc2[u][t]) : dclr_mfma32(l[u], ah[t], acc2[u][t]);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int t = 0; t < MT; ++t)
                acc2[u][t] = LAST ? dclr_mfma32(al[t], h[u], acc2[u][t]) : dclr_mfma32(h[u], al[t], acc2[u][t]);
    };
    fetch(0, wh[0], wl[0]);
    fetch(kg > 1 ? 1 : 0, wh[1], wl[1]);
    fetch(kg > 2 ? 2 : 0, wh[2], wl[2]);
    for (int g = 0; g < kg; g += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (g + i < kg) {                                          // wave-uniform
                const int ahead = g + i + 3 < kg ? g + i + 3 : kg - 1;  // clamped: the load stays unconditional
                fetch(ahead, wh[(i + 3) & 3], wl[(i + 3) & 3]);
                __builtin_amdgcn_sched_barrier(0);
                step(g + i, wh[i], wl[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#endif
}

template <int MT>
__global__ __launch_bounds__(H16_WAVES * 64) void head16_kernel(Head16Params prm, const float *__restrict__ x, int ldx,
                                                                float *__restrict__ colmax, int rows_per_group) {
    constexpr int ROWS = 32 * MT;
    __shared__ __attribute__((aligned(16))) char act[ROWS * dclr_split_stride(H16_MAX_WIDTH)];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.x * ROWS;

    // stage the input rows: one k-octet (8 floats -> 16 B hi + 16 B lo) per thread and step
    {
        const int stride = dclr_split_stride(prm.k[0]);
        const int octets = prm.k[0] / 8, valid = prm.k_in / 8;
        float peak = 0.f;
        for (int e = tid; e < ROWS * octets; e += H16_WAVES * 64) {
            const int r = e / octets, o = e - r * octets;
            dclr_h8 hi, lo;
            if (o < valid) {
                const float4 v0 = *reinterpret_cast<const float4 *>(x + (size_t)(m0 + r) * ldx + 8 * o);
                const float4 v1 = *reinterpret_cast<const float4 *>(x + (size_t)(m0 + r) * ldx + 8 * o + 4);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    dclr_h2 a, b;
                    dclr_split2(v[q], v[q + 1], a, b, peak);
                    hi[q] = a[0]; hi[q + 1] = a[1]; lo[q] = b[0]; lo[q + 1] = b[1];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) { hi[q] = (_Float16)0.f; lo[q] = (_Float16)0.f; }
            }
            char *dst = &act[r * stride + 32 * o];
            *reinterpret_cast<dclr_h8 *>(dst) = hi;
            *reinterpret_cast<dclr_h8 *>(dst + 16) = lo;
        }
        dclr_report_overflow(prm.overflow, peak);
    }
    __syncthreads();

    for (int l = 0; l < prm.n_layers; ++l) {
        const int kp = prm.k[l], n = prm.n[l], kg = kp / 16;
        const int in_stride = dclr_split_stride(kp), out_stride = dclr_split_stride(n);
        const bool last = l == prm.n_layers - 1;
        const char *a_lane = act + j * in_stride + 32 * h;
        const int n_tiles = n / 32;
        const size_t plane = (size_t)n_tiles * kg * 64;                 // fragments per plane
        const int ts = H16_WAVES * kg * 64;
        if (!last) {
            // hidden layer: this wave's tiles are w and w + 8 (n <= 512), results held until everyone has read
            const int t0 = wave;
            const bool any = t0 < n_tiles, two = t0 + H16_WAVES < n_tiles;      // wave-uniform
            // registers 4 g4 + i of tile tt = channels 32 tt + 8 g4 + 4 h + i: the accumulators start at the bias
            dclr_f32x16 acc[2][MT], acc2[2][MT];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int tt = t0 + u * H16_WAVES < n_tiles ? t0 + u * H16_WAVES : 0;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 bv = *reinterpret_cast<const float4 *>(prm.b[l] + 32 * tt + 8 * g4 + 4 * h);
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        acc[u][t][4 * g4 + 0] = bv.x; acc[u][t][4 * g4 + 1] = bv.y;
                        acc[u][t][4 * g4 + 2] = bv.z; acc[u][t][4 * g4 + 3] = bv.w;
                    }
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) acc2[u][t] = dclr_zero16();
            }
            if (any) {
                const float4 *wh = prm.w[l] + (size_t)t0 * kg * 64 + lane;
                if (two) head16_panel<false, 2, MT>(acc, acc2, a_lane, 32 * in_stride, kg, wh, wh + plane, ts);
                else head16_panel<false, 1, MT>(acc, acc2, a_lane, 32 * in_stride, kg, wh, wh + plane, ts);
            }
            __syncthreads();                                   // layer input fully consumed: overwrite in place
            if (any) {
                float peak = 0.f;                              // lives through this epilogue only (the K loop sits at 249 registers)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !two) break;
                    const int tt = t0 + u * H16_WAVES;
                    // lane = point j of row tile t; registers 4 g4 + i = channel 32 tt + 8 g4 + 4 h + i
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            dclr_h4 hi, lo;
#pragma unroll
                            for (int i = 0; i < 4; i += 2) {
                                dclr_h2 a, b;
                                dclr_split2_relu(fmaf(acc2[u][t][4 * g4 + i], DCLR_SPLIT_INV, acc[u][t][4 * g4 + i]),
                                                 fmaf(acc2[u][t][4 * g4 + i + 1], DCLR_SPLIT_INV, acc[u][t][4 * g4 + i + 1]),
                                                 a, b, peak);
                                hi[i] = a[0]; hi[i + 1] = a[1]; lo[i] = b[0]; lo[i + 1] = b[1];
                            }
                            char *dst = act + (32 * t + j) * out_stride + 32 * (4 * tt + g4) + 8 * h;
                            *reinterpret_cast<dclr_h4 *>(dst) = hi;
                            *reinterpret_cast<dclr_h4 *>(dst + 16) = lo;
                        }
                    }
                }
                dclr_report_overflow(prm.overflow, peak);
            }
            __syncthreads();
        } else {
            for (int t0 = wave; t0 < n_tiles; t0 += 2 * H16_WAVES) {
                const bool two = t0 + H16_WAVES < n_tiles;                  // wave-uniform
                dclr_f32x16 acc[2][MT], acc2[2][MT];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < MT; ++t) { acc[u][t] = dclr_zero16(); acc2[u][t] = dclr_zero16(); }
                const float4 *wh = prm.w[l] + (size_t)t0 * kg * 64 + lane;
                if (two) head16_panel<true, 2, MT>(acc, acc2, a_lane, 32 * in_stride, kg, wh, wh + plane, ts);
                else head16_panel<true, 1, MT>(acc, acc2, a_lane, 32 * in_stride, kg, wh, wh + plane, ts);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !two) break;
                    // lane = channel 32 tt + j; registers = points
                    const int col = 32 * (t0 + u * H16_WAVES) + j;
                    float mx = -3.0e38f;
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaf(acc2[u][t][r], DCLR_SPLIT_INV, acc[u][t][r]));
                    mx = fmaxf(mx + prm.b[l][col], 0.f);                // bias and ReLU commute with the maximum
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    if (h == 0)
                        atomicMax(reinterpret_cast<unsigned int *>(colmax + (size_t)(m0 / rows_per_group) * n + col),
                                  __float_as_uint(mx));
                }
            }
        }
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
    return dclr_x_head_conv_fused_f16(m, n_layers, k_in, k_host, n_host, w_packed_host, bias_host, x, ldx, colmax,
                                      rows_per_group, nullptr, stream);
}

int dclr_x_head_conv_fused_f16(int m, int n_layers, int k_in, const int *k_host, const int *n_host,
                               const void *const *w_packed_host, const float *const *bias_host, const float *x, int ldx,
                               float *colmax, int rows_per_group, uint32_t *overflow, dclr_stream_t stream) {
    DCLR_REQUIRE(m > 0 && n_layers >= 1 && k_host && n_host && w_packed_host && bias_host && x && colmax);
    DCLR_REQUIRE(m % 32 == 0 && rows_per_group > 0 && rows_per_group % 32 == 0 && m % rows_per_group == 0);
    DCLR_REQUIRE(k_in > 0 && k_in % 8 == 0 && ldx % 4 == 0 && ldx >= k_in && k_in <= k_host[0] && ((uintptr_t)x & 15) == 0);
    if (n_layers > H16_MAX_LAYERS) return DCLR_E_UNSUPPORTED;
    Head16Params prm{};
    prm.n_layers = n_layers;
    prm.k_in = k_in;
    prm.overflow = overflow;
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
    // 64 points per workgroup once that still gives every other CU a workgroup (and groups stay whole)
    if (m >= 64 * 128 && rows_per_group % 64 == 0)
        hipLaunchKernelGGL(head16_kernel<2>, dim3(m / 64), dim3(H16_WAVES * 64), 0, (hipStream_t)stream, prm, x, ldx,
                           colmax, rows_per_group);
    else
        hipLaunchKernelGGL(head16_kernel<1>, dim3(m / 32), dim3(H16_WAVES * 64), 0, (hipStream_t)stream, prm, x, ldx,
                           colmax, rows_per_group);
    return dclr_launch_status();
}
