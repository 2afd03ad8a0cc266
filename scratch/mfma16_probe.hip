// Probe: accumulation behaviour and rate of v_mfma_f32_32x32x16_f16 on gfx950 (diagnostic, not product).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// C (32x32) = A (32xK) * B (Kx32), fp32 inputs split into fp16 hi + scaled lo, 3 products
__global__ void split_gemm(int K, const float *A, const float *B, float *C, int mode) {
    const int l = threadIdx.x, i = l & 31, kb = l >> 5;
    f16v acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        h8 ah, al, bh, bl;
        for (int q = 0; q < 8; ++q) {
            const float a = A[i * K + k0 + 8 * kb + q], b = B[(k0 + 8 * kb + q) * 32 + i];
            const _Float16 a_h = (_Float16)a, b_h = (_Float16)b;
            ah[q] = a_h; bh[q] = b_h;
            al[q] = (_Float16)((a - (float)a_h) * 2048.f);
            bl[q] = (_Float16)((b - (float)b_h) * 2048.f);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        if (mode >= 1) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
        C[row * 32 + i] = acc[r] + acc2[r] * (1.0f / 2048.f);
    }
}

// fp32 MFMA reference chain
typedef float f16v_;
__global__ void f32_gemm(int K, const float *A, const float *B, float *C) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + h], B[(k0 + h) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

// rate: back-to-back MFMAs on registers, 4 waves per CU x 256 CUs x 2 (two waves per SIMD)
template <int SHAPE>
__global__ void rate_kernel(int iters, float *out) {
    h8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.001f * (threadIdx.x + q)); b[q] = (_Float16)(0.002f * (threadIdx.x ^ q)); }
    f16v c0, c1, c2;
    for (int r = 0; r < 16; ++r) { c0[r] = 0; c1[r] = 0; c2[r] = 0; }
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int K = 512;
    std::mt19937 rng(3);
    std::normal_distribution<float> g(0.f, 1.f);
    std::vector<float> A(32 * K), B(K * 32), C(1024), Cs(1024), C1(1024);
    for (auto &v : A) v = fabsf(g(rng));            // post-ReLU-like: all positive -> no cancellation, bias visible
    for (auto &v : B) v = 0.1f * fabsf(g(rng));
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    f32_gemm<<<1, 64>>>(K, dA, dB, dC); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    split_gemm<<<1, 64>>>(K, dA, dB, dC, 1); hipMemcpy(Cs.data(), dC, 4096, hipMemcpyDeviceToHost);
    split_gemm<<<1, 64>>>(K, dA, dB, dC, 0); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
    double e32 = 0, es = 0, e1 = 0, b32 = 0, bs = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[k * 32 + j];
            const double d32 = (C[i * 32 + j] - ref) / ref, ds = (Cs[i * 32 + j] - ref) / ref, d1 = (C1[i * 32 + j] - ref) / ref;
            e32 += d32 * d32; es += ds * ds; e1 += d1 * d1; b32 += d32; bs += ds;
        }
    printf("relative error vs fp64, K=%d, positive data: f32 mfma rms %.3e bias %.3e | f16 split(3) rms %.3e bias %.3e | f16 hi only rms %.3e\n",
           K, sqrt(e32 / 1024), b32 / 1024, sqrt(es / 1024), bs / 1024, sqrt(e1 / 1024));
    hipEvent_t e0, e1v; hipEventCreate(&e0); hipEventCreate(&e1v);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        rate_kernel<0><<<256 * 2, 256>>>(iters, dC);
        hipEventRecord(e1v); hipEventSynchronize(e1v);
        float ms; hipEventElapsedTime(&ms, e0, e1v);
        const double flop = 2.0 * 32 * 32 * 16 * 3.0 * iters * (256.0 * 2 * 4);
        printf("f16 32x32x16 rate: %.1f TFLOP/s (%.2f ms)\n", flop / ms / 1e9, ms);
    }
    return 0;
}
