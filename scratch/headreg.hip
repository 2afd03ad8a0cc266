// Pose-head conv chain, register-resident form (split-fp16 MFMA, mma16f.h), for the reference architecture
// 259 -> 256 -> 256 -> 512 -> 512 -> 1024 (+ max over points); reference: OutputSimple.forward,
// /root/reference/deepclr/models/deepclr.py:284-287 (Conv1dMultiLayer, ReLU after every layer, then max over points),
// layer widths from /root/reference/models/kitti_00-06/model_config.yaml (identical in the ModelNet config).
//
// head16_kernel (gemm16.hip) keeps the activations of 64 points in LDS and splits the output channels over its
// waves: two barriers per layer, in-place split-and-store epilogues, 2 waves per SIMD (matrix pipe 45 % busy).
// Here the roles are swapped:
//   * a wave owns 16 points for the whole chain and keeps their activations IN REGISTERS: with
//     v_mfma_f32_16x16x32_f16 computing W * X^T the accumulator of a 16-channel tile has the point on the lane
//     and 4 consecutive channels in its registers, which is -- for a weight matrix whose K order is permuted to
//     match (baked into the packing) -- exactly the B operand of the next layer: no LDS round trip, no barrier
//     between layers, no cross-lane movement;
//   * the weights (the A operand, needed by all four waves) stream through LDS: the packed matrix of all five
//     layers is ONE linear sequence of 16 KB stages (8 channel tiles x one 32-deep k-step, hi and lo planes), moved
//     by LDS-DMA (global_load_lds_dwordx4) into a 7-slot ring five stages ahead of use and read back as MFMA
//     fragments with ds_read_b128. One s_barrier per stage orders the ring; it never drains at a layer boundary.
// Workgroup = 4 waves (one per SIMD, up to 512 VGPRs each) = 64 points; per stage and wave 16 ds_read_b128 and
// 24 MFMAs, the reads of the next half stage issued before the MFMAs of the current one.
#include <stdlib.h>

#include "mma16f.h"

namespace {

constexpr int HR_WAVES = 4, HR_ROWS = 64;
constexpr int HR_STAGE = 16384;                          // bytes per stage: 8 tiles x (hi, lo) x 64 lanes x 16 B
constexpr int HR_D = 7, HR_R = HR_D + 2;                 // stages in flight ahead of the one being read; ring slots
constexpr int HR_NL = 5;
constexpr int HR_K0 = 288;                               // layer-0 input columns, padded to k-steps of 32 (264 valid)
constexpr int HR_STAGES = 2 * 9 + 2 * 8 + 4 * 8 + 4 * 16 + 8 * 16;     // 258
constexpr int HR_BIAS = 256 + 256 + 512 + 512 + 1024;    // 2560 floats
constexpr int HR_LDS = HR_R * HR_STAGE + HR_BIAS * 4 + 1024 * 4;       // 161,792 of 163,840 bytes

struct HrPackParams {
    const float *w[HR_NL];      // row-major (n, k_in) f32
    int k_in[HR_NL];            // reference input width of each layer
    int n[HR_NL];
    const int32_t *kmap0;       // HR_K0 entries: row column -> reference input column of layer 0 (-1 = zero)
};

__host__ __device__ constexpr int hr_ks(int l) { return l == 0 ? 9 : (l < 3 ? 8 : 16); }
__host__ __device__ constexpr int hr_ntg(int l) { return l < 2 ? 2 : (l < 4 ? 4 : 8); }

// Packed stream: stage (layer l, tile group tg, k-step s) in consumption order; inside a stage
// [tile t][plane hi/lo][lane][8 halves]: lane = (channel & 15) + 16 * k-group g, element j. The logical input index
// of (s, g, j) is 32 s + 8 g + j for layer 0 (activations loaded from rows) and 32 s + 16 (j >> 2) + 4 g + (j & 3)
// for the other layers (activations taken from the accumulators of tiles 2 s and 2 s + 1, see the kernel).
__global__ __launch_bounds__(256) void hr_pack_kernel(HrPackParams p, int total_stages, _Float16 *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)total_stages * (HR_STAGE / 2)) return;
    const int stage = (int)(e >> 13), r = (int)(e & 8191);
    const int t = r >> 10, hl = (r >> 9) & 1, lane = (r >> 3) & 63, j = r & 7;
    float v = 0.f;
    if (stage < HR_STAGES) {
        int l = 0, base = 0;
        while (stage >= base + hr_ks(l) * hr_ntg(l)) { base += hr_ks(l) * hr_ntg(l); ++l; }
        const int rel = stage - base, tg = rel / hr_ks(l), s = rel % hr_ks(l);
        const int n = 128 * tg + 16 * t + (lane & 15), g = lane >> 4;
        const int kk = l == 0 ? 32 * s + 8 * g + j : 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
        const int col = l == 0 ? p.kmap0[kk] : kk;
        if (n < p.n[l] && col >= 0 && col < p.k_in[l]) v = p.w[l][(size_t)n * p.k_in[l] + col];
    }
    _Float16 hi, lo;
    dclr_split(v, hi, lo);
    out[e] = hl ? lo : hi;
}

struct HrFrag {
    dclr_h8 hi, lo;
};

typedef __attribute__((address_space(1))) const void *hr_gptr;
typedef __attribute__((address_space(3))) void *hr_lptr;

struct HrRing {
    const char *wq;      // packed stream + this wave's share of a stage + lane * 16
    char *ring;          // LDS ring base
    int slot;            // ring slot of the stage whose first half is already in `xa`
    int dma_slot;        // where the next DMA goes
    int dma_stage;       // the next stage to fetch (stages are consumed in stream order; past the end: the zero stage)
    unsigned long long t_last, t_a, t_dma, t_b;     // ABL & 4 (diagnostic build): cycle sums per stage segment
};

__device__ unsigned long long hr_dbg[16];
#define HR_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")

// this wave's quarter of the next stage -> ring slot dma_slot (4 x 1 KB, one LDS-DMA instruction each)
__device__ __forceinline__ void hr_dma(HrRing &r, int wave) {
    const int stage = r.dma_stage < HR_STAGES ? r.dma_stage : HR_STAGES;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((hr_gptr)(r.wq + (size_t)stage * HR_STAGE + q * 1024),
                                         (hr_lptr)(r.ring + r.dma_slot * HR_STAGE + (4 * wave + q) * 1024), 16, 0, 0);
    r.dma_stage += 1;
    r.dma_slot = r.dma_slot + 1 == HR_R ? 0 : r.dma_slot + 1;
}

__device__ __forceinline__ dclr_f32x4 hr_mfma(dclr_h8 a, dclr_h8 b, dclr_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// Twelve MFMAs with eight LDS fragment reads slotted behind the first four (two each): the reads feed the NEXT half
// stage, so nothing in this block waits for them, and the eight MFMAs that follow (128 cycles of matrix pipe) cover
// their latency before the next block's first MFMA asks for them.
__device__ __forceinline__ void hr_interleave_hint() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS reads
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
}

// One stage: 8 tiles x one k-step against the B fragment (bh, bl). On entry xa holds tiles 0..3 of the stage;
// on exit tiles 0..3 of the next one.
template <int ABL>      // timing-only ablations (results wrong): bit 0 = no DMA after the prologue, bit 1 = no LDS fragment reads
__device__ __forceinline__ void hr_stage(HrRing &r, dclr_h8 (&xa)[4][2], const dclr_h8 &bh, const dclr_h8 &bl,
                                         dclr_f32x4 (&acc)[8], dclr_f32x4 (&acc2)[8], int lane, int wave) {
    dclr_h8 ya[4][2];
    const char *rd = r.ring + r.slot * HR_STAGE + lane * 16;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = hr_mfma(xa[t][0], bh, acc[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = hr_mfma(xa[t][1], bh, acc2[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if constexpr (ABL & 2) { ya[t][0] = xa[t][1]; ya[t][1] = xa[t][0]; }
        else {
            ya[t][0] = dclr_lds_h8(rd + (8 + 2 * t) * 1024);
            ya[t][1] = dclr_lds_h8(rd + (9 + 2 * t) * 1024);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[t] = hr_mfma(xa[t][0], bl, acc2[t]);
    hr_interleave_hint();
    __builtin_amdgcn_sched_barrier(0);
    // stage st + 1 has landed once every wave has seen its own quarter arrive: all but the youngest
    // 4 * (HR_D - 1) DMA instructions (stages st + 2 .. st + HR_D) are done
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (HR_D - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (ABL & 4) {
        unsigned long long t;
        HR_STAMP(t);
        r.t_a += t - r.t_last;
        r.t_last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
    // every wave is past the first half of stage st, hence done with stage st - 1: its slot takes stage st + 1 + HR_D
    if constexpr (!(ABL & 1)) hr_dma(r, wave);
    r.slot = r.slot + 1 == HR_R ? 0 : r.slot + 1;
    const char *rn = r.ring + r.slot * HR_STAGE + lane * 16;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL & 4) {
        unsigned long long t;
        HR_STAMP(t);
        r.t_dma += t - r.t_last;
        r.t_last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[4 + t] = hr_mfma(ya[t][0], bh, acc[4 + t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[4 + t] = hr_mfma(ya[t][1], bh, acc2[4 + t]);
    dclr_h8 xn[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if constexpr (ABL & 2) { xn[t][0] = ya[t][1]; xn[t][1] = ya[t][0]; }
        else {
            xn[t][0] = dclr_lds_h8(rn + (2 * t) * 1024);
            xn[t][1] = dclr_lds_h8(rn + (2 * t + 1) * 1024);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[4 + t] = hr_mfma(ya[t][0], bl, acc2[4 + t]);
    hr_interleave_hint();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) { xa[t][0] = xn[t][0]; xa[t][1] = xn[t][1]; }
    if constexpr (ABL & 4) {
        unsigned long long t;
        HR_STAMP(t);
        r.t_b += t - r.t_last;
        r.t_last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LDS atomic max behind the compiler's back: hipcc orders every LDS access it cannot prove disjoint from an LDS-DMA
// destination behind `s_waitcnt vmcnt(0)`, which would drain the weight ring 256 times per workgroup.
__device__ __forceinline__ void hr_lds_max_u32(float *p, uint32_t v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)p;
    asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// One layer: NTG tile groups of 128 output channels, KS k-steps each. Hidden layers leave the next layer's B
// fragments in `out` (4 per tile group); the last layer folds each tile group into the wave's column maxima.
template <int KS, int NTG, bool LAST, int ABL>
__device__ __forceinline__ void hr_layer(HrRing &r, dclr_h8 (&xa)[4][2], const HrFrag (&in)[KS], HrFrag (&out)[LAST ? 1 : 4 * NTG],
                                         const float *bias_s, float *cm_wave, int lane, int wave) {
    const int g = lane >> 4;
#pragma unroll 1
    for (int tg = 0; tg < NTG; ++tg) {
        dclr_f32x4 acc[8], acc2[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // registers i of tile t = channel 128 tg + 16 t + 4 g + i: hidden layers start at the bias
            if constexpr (LAST) acc[t] = dclr_f32x4{0.f, 0.f, 0.f, 0.f};
            else acc[t] = *reinterpret_cast<const dclr_f32x4 *>(bias_s + 128 * tg + 16 * t + 4 * g);
            acc2[t] = dclr_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (ABL & 4) {
            unsigned long long t;
            HR_STAMP(t);
            r.t_last = t;                       // whatever ran since the last stage (epilogue, bias loads) is not a stage
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) hr_stage<ABL>(r, xa, in[s].hi, in[s].lo, acc, acc2, lane, wave);
        if constexpr (LAST) {
            // lane = point, registers = channels: bias and ReLU first (they commute with the maximum), then the values are
            // non-negative and the maximum over this wave's 16 points is four one-instruction u32 DPP steps per value;
            // lane 15 of each row then folds its 32 results into the workgroup's column maxima (one masked block)
            uint32_t mrow[8][4];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const dclr_f32x4 bv = *reinterpret_cast<const dclr_f32x4 *>(bias_s + 128 * tg + 16 * t + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    mrow[t][i] = dclr_row16_max_lanes(__float_as_uint(fmaxf(fmaf(acc2[t][i], DCLR_SPLIT_INV, acc[t][i]) + bv[i], 0.f)));
            }
            if ((lane & 15) == 15) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hr_lds_max_u32(cm_wave + 128 * tg + 16 * t + 4 * g + i, mrow[t][i]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < NTG; ++c) {
                if (tg == c) {                                   // wave-uniform: static register indices per branch
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // next layer's k-step 4 c + q = channels of tiles 2 q (elements 0..3) and 2 q + 1 (4..7)
                        dclr_h8 hi, lo;
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const int t = 2 * q + hf;
#pragma unroll
                            for (int i = 0; i < 4; i += 2) {
                                dclr_h2 a, b;
                                dclr_split2_relu(fmaf(acc2[t][i], DCLR_SPLIT_INV, acc[t][i]),
                                                 fmaf(acc2[t][i + 1], DCLR_SPLIT_INV, acc[t][i + 1]), a, b);
                                hi[4 * hf + i] = a[0]; hi[4 * hf + i + 1] = a[1];
                                lo[4 * hf + i] = b[0]; lo[4 * hf + i + 1] = b[1];
                            }
                        }
                        out[4 * c + q].hi = hi;
                        out[4 * c + q].lo = lo;
                    }
                }
            }
        }
    }
}

template <int ABL>
__global__ __launch_bounds__(HR_WAVES * 64, 1) void head_reg_kernel(const float *__restrict__ x, int ldx, int k_in,
                                                                    const char *__restrict__ wq,
                                                                    const float *__restrict__ bias,
                                                                    float *__restrict__ colmax, int rows_per_group) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *ring = lds;
    float *bias_s = reinterpret_cast<float *>(lds + HR_R * HR_STAGE);
    float *cm = bias_s + HR_BIAS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pt = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * HR_ROWS;

    // the weight stream starts at once: stages 0 .. HR_D
    HrRing r;
    r.wq = wq + (4 * wave) * 1024 + lane * 16;
    r.ring = ring;
    r.dma_stage = 0;
    r.t_a = r.t_dma = r.t_b = 0;
    unsigned long long t_begin = 0;
    if constexpr (ABL & 4) { HR_STAMP(t_begin); }
    r.t_last = t_begin;
    r.slot = 0;
    r.dma_slot = 0;
#pragma unroll
    for (int s = 0; s <= HR_D; ++s) hr_dma(r, wave);

    // biases -> LDS, this lane's share of its point's input row -> split B fragments of layer 0
    for (int i = tid; i < HR_BIAS; i += HR_WAVES * 64) bias_s[i] = bias[i];
    for (int i = tid; i < 1024; i += HR_WAVES * 64) cm[i] = 0.f;
    HrFrag a0[9];
    {
        const float *row = x + (size_t)(m0 + 16 * wave + pt) * ldx;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int c0 = 32 * s + 8 * g;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (c0 + 8 <= k_in) {
                const float4 v0 = *reinterpret_cast<const float4 *>(row + c0);
                const float4 v1 = *reinterpret_cast<const float4 *>(row + c0 + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            }
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                dclr_h2 a, b;
                dclr_split2(v[q], v[q + 1], a, b);
                a0[s].hi[q] = a[0]; a0[s].hi[q + 1] = a[1];
                a0[s].lo[q] = b[0]; a0[s].lo[q + 1] = b[1];
            }
        }
    }
    // stage 0 landed (the plain loads above have drained the counter anyway) and the biases are visible
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * HR_D) : "memory");
    __syncthreads();
    dclr_h8 xa[4][2];
    {
        const char *rn = ring + lane * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xa[t][0] = dclr_lds_h8(rn + (2 * t) * 1024);
            xa[t][1] = dclr_lds_h8(rn + (2 * t + 1) * 1024);
        }
    }
    float *cm_wave = cm;
    HrFrag a1[8], a2[8], a3[16], a4[16], none[1];
    hr_layer<9, 2, false, ABL>(r, xa, a0, a1, bias_s, cm_wave, lane, wave);
    hr_layer<8, 2, false, ABL>(r, xa, a1, a2, bias_s + 256, cm_wave, lane, wave);
    hr_layer<8, 4, false, ABL>(r, xa, a2, a3, bias_s + 512, cm_wave, lane, wave);
    hr_layer<16, 4, false, ABL>(r, xa, a3, a4, bias_s + 1024, cm_wave, lane, wave);
    hr_layer<16, 8, true, ABL>(r, xa, a4, none, bias_s + 1536, cm_wave, lane, wave);
    // drain the DMAs of the padding stages (nobody reads them) before the ring's LDS could be reused, then fold the
    // four waves' column maxima: bias, ReLU, one atomic max per channel
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (ABL & 4) {
        unsigned long long t_end;
        HR_STAMP(t_end);
        if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 3)) {
            unsigned long long *d = hr_dbg + (wave == 0 ? 0 : 8);
            d[0] = t_end - t_begin; d[1] = r.t_a; d[2] = r.t_dma; d[3] = r.t_b;
        }
    }
    float *dst = colmax + (size_t)(m0 / rows_per_group) * 1024;
    for (int c = tid; c < 1024; c += HR_WAVES * 64)
        atomicMax(reinterpret_cast<unsigned int *>(dst + c), __float_as_uint(cm[c]));
}

}  // namespace

// diagnostic builds only (DCLR_HR_ABL=4/5): the cycle sums the stamped kernel left behind, 16 values
extern "C" int dclr_head_reg_debug(unsigned long long *out_host) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(hr_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}

extern "C" long long dclr_head_reg_packed_bytes(void) { return (long long)(HR_STAGES + 1) * HR_STAGE; }

// Pack the five conv layers of the reference head (259 -> 256 -> 256 -> 512 -> 512 -> 1024) for
// dclr_head_conv_reg_f16. w_host[l]: DEVICE pointers to row-major f32 weights (n[l], k_in[l]) with
// k_in = {259, 256, 256, 512, 512}; kmap0: 288 i32 (device): column of rows E -> reference input column of
// layer 0 (-1 = zero). packed: dclr_head_reg_packed_bytes() bytes, 16-byte aligned.
extern "C" int dclr_head_reg_pack(int n_layers, const int *k_in_host, const int *n_host, const float *const *w_host,
                                  const int32_t *kmap0, void *packed, dclr_stream_t stream) {
    DCLR_REQUIRE(k_in_host && n_host && w_host && kmap0 && packed && ((uintptr_t)packed & 15) == 0);
    static const int want_n[HR_NL] = {256, 256, 512, 512, 1024};
    if (n_layers != HR_NL) return DCLR_E_UNSUPPORTED;
    HrPackParams p{};
    for (int l = 0; l < HR_NL; ++l) {
        DCLR_REQUIRE(w_host[l] != nullptr && k_in_host[l] > 0);
        if (n_host[l] != want_n[l] || (l > 0 && k_in_host[l] != want_n[l - 1]) || (l == 0 && k_in_host[0] > HR_K0))
            return DCLR_E_UNSUPPORTED;
        p.w[l] = w_host[l];
        p.k_in[l] = k_in_host[l];
        p.n[l] = n_host[l];
    }
    p.kmap0 = kmap0;
    const int total = HR_STAGES + 1;                       // one zero stage behind the stream: what the ring fetches past the end
    const size_t elems = (size_t)total * (HR_STAGE / 2);
    hipLaunchKernelGGL(hr_pack_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, total,
                       reinterpret_cast<_Float16 *>(packed));
    return dclr_launch_status();
}

// x rows (m, ldx) with k_in valid leading columns (k_in % 8 == 0, k_in <= 288) -> column maxima per group of
// rows_per_group rows into colmax (m / rows_per_group, 1024), zero-filled by the caller. packed: dclr_head_reg_pack;
// bias: the five layers' biases back to back (2560 floats). m and rows_per_group multiples of 64.
extern "C" int dclr_head_conv_reg_f16(int m, int k_in, const void *packed, const float *bias, const float *x, int ldx,
                                      float *colmax, int rows_per_group, dclr_stream_t stream) {
    DCLR_REQUIRE(m > 0 && packed && bias && x && colmax && rows_per_group > 0);
    DCLR_REQUIRE(m % HR_ROWS == 0 && rows_per_group % HR_ROWS == 0 && m % rows_per_group == 0);
    DCLR_REQUIRE(k_in > 0 && k_in % 8 == 0 && k_in <= HR_K0 && ldx % 4 == 0 && ldx >= k_in && ((uintptr_t)x & 15) == 0 &&
                 ((uintptr_t)packed & 15) == 0);
    // more than 64 KB of dynamic LDS has to be granted once per device (idempotent; no other state is kept)
    static bool granted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return dclr_launch_status();
    static const int abl = getenv("DCLR_HR_ABL") ? atoi(getenv("DCLR_HR_ABL")) : 0;        // measurement switches
    auto launch = [&](auto kern) -> int {
        if (!granted[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, HR_LDS) !=
                hipSuccess)
                return dclr_launch_status();
            granted[dev] = true;
        }
        hipLaunchKernelGGL(kern, dim3(m / HR_ROWS), dim3(HR_WAVES * 64), HR_LDS, (hipStream_t)stream, x, ldx, k_in,
                           reinterpret_cast<const char *>(packed), bias, colmax, rows_per_group);
        return dclr_launch_status();
    };
    if (abl == 4) return launch(head_reg_kernel<4>);
    if (abl == 5) return launch(head_reg_kernel<5>);
    if (abl == 1) return launch(head_reg_kernel<1>);
    if (abl == 2) return launch(head_reg_kernel<2>);
    if (abl == 3) return launch(head_reg_kernel<3>);
    return launch(head_reg_kernel<0>);
}
