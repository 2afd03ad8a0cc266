// Level-1 grouping operators for gfx950: gather, ball query, group, and the backward of gather / group.
//
// Replace gather_points_wrapper_fast / ball_query_wrapper_fast / group_points_wrapper_fast
// (/root/reference/extern/pointnet2.patch:275-288, 101-116, 160-174) and, for the training step
// (/root/reference/deepclr/engine/engines.py:57-84 differentiates through them), group_points_grad_wrapper_fast /
// gather_points_grad_wrapper_fast (pointnet2.patch:144-158, 290-304). Behaviour follows
// oracle/primitives.c (the wrapped kernel bodies are not in the reference tree).
#include "common.h"

namespace {

// out[b,c,j] = points[b,c,idx[b,j]] -- one thread per output element, j fastest (coalesced stores).
__global__ __launch_bounds__(256) void gather_points_kernel(int c, int n, int npoints,
                                                            const float *__restrict__ points,
                                                            const int32_t *__restrict__ idx,
                                                            float *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int ci = blockIdx.y, bi = blockIdx.z;
    if (j >= npoints) return;
    const int k = idx[(size_t)bi * npoints + j];
    out[((size_t)bi * c + ci) * npoints + j] = points[((size_t)bi * c + ci) * n + k];
}

// out[b,c,j,s] = points[b,c,idx[b,j,s]] -- thread per (j,s) element; the index row is read once
// per channel from L2, stores are fully coalesced.
__global__ __launch_bounds__(256) void group_points_kernel(int c, int n, int npoints, int nsample,
                                                           const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_cloud = (size_t)npoints * nsample;
    const int bi = blockIdx.y;
    if (e >= per_cloud) return;
    const int k = idx[(size_t)bi * per_cloud + e];
    const float *src = points + (size_t)bi * c * n;
    float *dst = out + (size_t)bi * c * per_cloud;
    for (int ci = 0; ci < c; ++ci) dst[(size_t)ci * per_cloud + e] = src[(size_t)ci * n + k];
}

// Ball query: one WAVE per centroid. The wave sweeps the cloud 64 points at a time (coalesced),
// ballots the in-radius lanes and uses the prefix popcount to keep hits in ascending point order,
// which is what the published one-thread-per-centroid serial scan produces.
constexpr int BQ_WAVES = 4;

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(int n, int m, float radius2,
                                                                   int nsample,
                                                                   const float *__restrict__ new_xyz,
                                                                   const float *__restrict__ xyz,
                                                                   int32_t *__restrict__ idx) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = blockIdx.x * BQ_WAVES + wave;
    const int bi = blockIdx.y;
    if (j >= m) return;                                     // wave-uniform
    const float *c = new_xyz + ((size_t)bi * m + j) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    const float *p = xyz + (size_t)bi * n * 3;
    int32_t *o = idx + ((size_t)bi * m + j) * nsample;

    int cnt = 0, first = -1;
    for (int base = 0; base < n && cnt < nsample; base += 64) {
        const int k = base + lane;
        bool hit = false;
        if (k < n) hit = dclr_sqdist(cx, cy, cz, p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]) < radius2;
        const uint64_t mask = __ballot(hit);
        if (mask == 0) continue;
        if (first < 0) first = base + __builtin_ctzll(mask);
        const int pos = cnt + (int)dclr_lanemask_lt_popc(mask);
        if (hit && pos < nsample) o[pos] = k;
        cnt += __builtin_popcountll(mask);
    }
    if (cnt == 0) return;                                   // row keeps the caller's zeros
    if (cnt > nsample) cnt = nsample;
    for (int s = cnt + lane; s < nsample; s += 64) o[s] = first;
}

// ---- backward of gather / group: grad_points[b,c,k] += sum over the entries e with idx[b,e] == k of grad_out[b,c,e] ----
// The published kernels issue one atomicAdd per entry. A ball-query index row repeats its first hit in every unused slot
// (up to nsample - 1 times), so most entries of a row hit ONE address, and same-address float atomics serialise at the
// memory side (MI355X_MICROARCH.md, global float atomics: "every workgroup into ONE row: 14x slower"). Here a wave takes 64
// consecutive entries, sums each run of adjacent equal indices in registers (segmented scan over the wave) and issues one
// atomic per run: a padded row costs a few atomics per 64 slots instead of 64. Entries with distinct indices (furthest point
// samples, the real hits of a row) are runs of one and cost what they did. Sums are f32 and, as upstream, depend on the
// order in which atomics of different waves land (last bits may differ between runs).
__global__ __launch_bounds__(256) void scatter_add_runs_kernel(int c, int n, size_t per_cloud,
                                                               const float *__restrict__ grad_out,
                                                               const int32_t *__restrict__ idx,
                                                               float *__restrict__ grad_points) {
    const int lane = threadIdx.x & 63;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int bi = blockIdx.y;
    const bool live = e < per_cloud;
    const int k = live ? idx[(size_t)bi * per_cloud + e] : -1;
    // run structure inside the wave: a lane starts a run when its index differs from its left neighbour's
    const int left = __shfl_up(k, 1);
    const bool head = lane == 0 || k != left;
    const uint64_t heads = __ballot(head);
    const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int start = 63 - __builtin_clzll(heads & upto);                  // first lane of this lane's run
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);        // last lane of its run: holds the run's sum
    const float *src = grad_out + (size_t)bi * c * per_cloud;
    float *dst = grad_points + (size_t)bi * c * n;
    for (int ci = 0; ci < c; ++ci) {
        float v = live ? src[(size_t)ci * per_cloud + e] : 0.f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float up = __shfl_up(v, d);
            if (lane - d >= start) v += up;
        }
        if (tail && k >= 0 && k < n) atomicAdd(dst + (size_t)ci * n + k, v);
    }
}

}  // namespace

extern "C" int dclr_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int32_t *idx,
                                       float *grad_points, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && c > 0 && n > 0 && npoints > 0 && grad_out && idx && grad_points);
    DCLR_REQUIRE(b <= 65535);
    hipLaunchKernelGGL(scatter_add_runs_kernel, dim3((unsigned)((npoints + 255) / 256), b), dim3(256), 0, (hipStream_t)stream,
                       c, n, (size_t)npoints, grad_out, idx, grad_points);
    return dclr_launch_status();
}

extern "C" int dclr_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                      const int32_t *idx, float *grad_points, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && c > 0 && n > 0 && npoints > 0 && nsample > 0 && grad_out && idx && grad_points);
    DCLR_REQUIRE(b <= 65535);
    const size_t per_cloud = (size_t)npoints * nsample;
    hipLaunchKernelGGL(scatter_add_runs_kernel, dim3((unsigned)((per_cloud + 255) / 256), b), dim3(256), 0, (hipStream_t)stream,
                       c, n, per_cloud, grad_out, idx, grad_points);
    return dclr_launch_status();
}

extern "C" int dclr_gather_points(int b, int c, int n, int npoints, const float *points,
                                  const int32_t *idx, float *out, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && c > 0 && n > 0 && npoints > 0 && points && idx && out);
    DCLR_REQUIRE(c <= 65535 && b <= 65535);
    hipLaunchKernelGGL(gather_points_kernel, dim3((npoints + 255) / 256, c, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, npoints, points, idx, out);
    return dclr_launch_status();
}

extern "C" int dclr_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                 const int32_t *idx, float *out, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && c > 0 && n > 0 && npoints > 0 && nsample > 0 && points && idx && out);
    DCLR_REQUIRE(b <= 65535);
    const size_t per_cloud = (size_t)npoints * nsample;
    hipLaunchKernelGGL(group_points_kernel, dim3((unsigned)((per_cloud + 255) / 256), b), dim3(256), 0,
                       (hipStream_t)stream, c, n, npoints, nsample, points, idx, out);
    return dclr_launch_status();
}

extern "C" int dclr_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                               const float *xyz, int32_t *idx, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0 && new_xyz && xyz && idx);
    DCLR_REQUIRE(b <= 65535);
    const float radius2 = radius * radius;      // binary32 product, as in the published kernel
    hipLaunchKernelGGL(ball_query_kernel, dim3((m + BQ_WAVES - 1) / BQ_WAVES, b), dim3(BQ_WAVES * 64), 0,
                       (hipStream_t)stream, n, m, radius2, nsample, new_xyz, xyz, idx);
    return dclr_launch_status();
}
