// How fast can a CU pull a shared, L2-resident weight stream -- alone and beside MFMAs?
//
// The fused pose head (csrc/gemm16.hip) and the flow embedding (csrc/flow16.hip) re-stream their packed weights from L2
// once per workgroup: 4.2 MB per 64 rows (head), 240 KB per 80 rows (flow). This probe reproduces only that access
// pattern: every workgroup (8 waves, as the head) walks the SAME buffer front to back with one 16-byte load per lane and
// step, three steps ahead, and optionally issues the head's MFMA work per loaded fragment pair (6 x v_mfma_f32_32x32x16_f16
// per 2 x 16 B per lane: three split products for two row tiles). It prints, per configuration, the time per pass, the
// aggregate L2 -> CU rate and the rate per CU -- the ceiling the two kernels' 0.44 of the MFMA peak sits under:
// 6 MFMAs x 32 cycles per 2 KB per wave = 42.7 B / cycle / CU = 102 GB/s per CU at 2.4 GHz for a busy matrix pipe.
//
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream_probe profiles/l2_stream_probe.hip && ./l2_stream_probe > profiles/r05_l2_stream_probe.csv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// shader-clock cycles (s_memtime) against the constant 100 MHz counter (s_memrealtime): the clock the kernel really ran at
__device__ unsigned long long stamps[4];
__device__ __forceinline__ void stamp(int i) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[i] = __builtin_readcyclecounter(); stamps[i + 1] = wall_clock64(); }
}

// The same work with the roles split between waves: waves 0..3 of the workgroup only load (two fragments per step each: the
// same bytes per workgroup), waves 4..7 only issue MFMAs (twice as many each: the same matrix work per workgroup), on
// operands that never leave their registers. If loads and MFMAs of DIFFERENT waves overlap where those of one wave do not,
// this runs in max(load time, MFMA time) instead of their sum.
template <int MFMA_PER_PAIR>
__global__ __launch_bounds__(512) void split_kernel(const float4 *__restrict__ w, int steps, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float r = 0.f;
    stamp(0);
    if (wave < 4) {
        const float4 *p = w + (size_t)(2 * wave) * 64 + lane;
        const size_t stride = 8 * 64;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q[4][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) { q[i][0] = p[(size_t)i * stride]; q[i][1] = p[(size_t)i * stride + 64]; }
        for (int s = 0; s < steps; s += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ahead = s + i + 3 < steps ? s + i + 3 : steps - 1;
                q[(i + 3) & 3][0] = p[(size_t)ahead * stride];
                q[(i + 3) & 3][1] = p[(size_t)ahead * stride + 64];
                x.x += q[i][0].x + q[i][1].x; x.y += q[i][0].y + q[i][1].y;
            }
        }
        r = x.x + x.y;
    } else {
        f16v acc[6] = {};                                            // six independent chains: MFMA latency never shows
        h8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.0137f * ((lane * 7 + i * 13) % 61) - 0.4f);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int m = 0; m < MFMA_PER_PAIR; ++m)                  // two fragments' worth per step and wave
                acc[m % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[m % 6], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i] + acc[4][i] + acc[5][i];
    }
    if (r == 123.456f) sink[0] = r;
    stamp(2);
}

template <int MFMA_PER_PAIR>
__global__ __launch_bounds__(512) void stream_kernel(const float4 *__restrict__ w, size_t frags_per_wave_step, int steps,
                                                     float *__restrict__ sink) {
    // wave v of the workgroup reads fragments [step * 8 + v] * 64 + lane: the 8 waves cover 8 KB per step, in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 *p = w + (size_t)wave * 64 + lane;
    stamp(0);
    f16v acc[6] = {};                                                // six independent chains (the head keeps eight)
    float4 q[4];
    const size_t stride = 8 * 64;                                   // float4 per step of the workgroup
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = p[(size_t)(i < steps ? i : 0) * stride];
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < steps; s += 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ahead = s + i + 3 < steps ? s + i + 3 : steps - 1;
            q[(i + 3) & 3] = p[(size_t)ahead * stride];
            const float4 v = q[i];
            if constexpr (MFMA_PER_PAIR > 0) {
                const h8 a = __builtin_bit_cast(h8, v);
#pragma unroll
                for (int m = 0; m < MFMA_PER_PAIR / 2; ++m)          // MFMA_PER_PAIR per TWO fragments (hi + lo planes)
                    acc[(3 * (i & 1) + m) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[(3 * (i & 1) + m) % 6], 0, 0, 0);
            } else {
                x.x += v.x; x.y += v.y; x.z += v.z; x.w += v.w;
            }
        }
    }
    float r = x.x + x.y + x.z + x.w;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i] + acc[4][i] + acc[5][i];
    if (r == 123.456f) sink[0] = r;                                  // keeps the loads and MFMAs alive
    stamp(2);
}

int main() {
    const size_t bytes = 4200 * 1024;                                // the head's packed weights: 4.2 MB
    const int steps = (int)(bytes / (8 * 64 * 16)) / 4 * 4;          // 8 KB per workgroup step
    float4 *w; float *sink;
    hipMalloc(&w, bytes); hipMalloc(&sink, 4);
    {   // random f16 operands: zero-filled operands let the chip hold a ~20 % higher clock under MFMA load (MI355X_MICROARCH.md, DVFS)
        std::vector<_Float16> h(bytes / 2);
        unsigned x = 12345u;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 16) % 2001 - 1000) * 0.001f); }
        hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice);
    }
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("mfma_per_fragment_pair,workgroups,us_per_launch,aggregate_TBps,GBps_per_CU,shader_clock_GHz_of_workgroup_0,mfma_pipe_busy_at_that_clock\n");
    for (int variant = 0; variant < 5; ++variant) {
        for (int wgs : {cus, 5 * cus}) {                             // one per CU; the head's 1,280 at 80 pairs
            float best = 1e30f;
            const int warm = 1 + (int)(1.0e6 / (variant == 0 ? 200.0 : 600.0) * cus / wgs / 5.0 * 5.0);   // ~1 s of back-to-back launches first
            for (int rep = -warm; rep < 7; ++rep) {
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(wgs), dim3(512), 0, 0, w, 0, steps, sink);
                if (variant == 1) hipLaunchKernelGGL(stream_kernel<6>, dim3(wgs), dim3(512), 0, 0, w, 0, steps, sink);
                if (variant == 2) hipLaunchKernelGGL(stream_kernel<12>, dim3(wgs), dim3(512), 0, 0, w, 0, steps, sink);
                if (variant == 3) hipLaunchKernelGGL(split_kernel<6>, dim3(wgs), dim3(512), 0, 0, w, steps, sink);
                if (variant == 4) hipLaunchKernelGGL(split_kernel<12>, dim3(wgs), dim3(512), 0, 0, w, steps, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 0 && ms < best) best = ms;
            }
            const double moved = (double)wgs * steps * 8 * 64 * 16;   // bytes through the CUs' vector memory path
            const double rate = moved / (best * 1e-3);
            unsigned long long st[4];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(stamps), sizeof(st));
            const double ghz = (double)(st[2] - st[0]) / ((double)(st[3] - st[1]) * 10.0);          // cycles per 10 ns tick
            const int per_pair = variant == 0 ? 0 : (variant == 1 || variant == 3) ? 6 : 12;
            // matrix-pipe time of the launch: MFMAs per SIMD x 32 cycles (8 passes of 4) at the measured clock
            const double mfma_s = (double)wgs / cus * steps * 8 / 2 * per_pair / 4.0 * 32.0 / (ghz * 1e9);
            printf("%s,%d,%.1f,%.2f,%.1f,%.2f,%.2f\n", variant == 0 ? "0" : variant == 1 ? "6" : variant == 2 ? "12" : variant == 3 ? "6 (loader waves | MFMA waves)" : "12 (loader waves | MFMA waves)", wgs, best * 1e3, rate / 1e12,
                   rate / 1e9 / cus, ghz, mfma_s / (best * 1e-3));
        }
    }
    return 0;
}
