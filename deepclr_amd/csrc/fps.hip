// Furthest point sampling for gfx950.
//
// Replaces furthest_point_sampling_wrapper (/root/reference/extern/pointnet2.patch:306-320); the
// kernel body it wrapped is not in the reference tree, so the behaviour (start at index 0,
// running-minimum update, strict-maximum selection, tie order of the T-thread halving tree) follows
// oracle/primitives.c, which restates the published algorithm.
//
// Design (one workgroup per cloud, the sampling rounds are inherently serial):
//   * every thread keeps its points AND their running minima in VGPRs for the whole kernel, so a
//     round touches no memory except one 24-byte LDS slot per wave;
//   * one barrier per round: each wave reduces (max distance, then min tie key) with DPP row
//     operations, its winning lane publishes {dist, key, k, x, y, z} to a double-buffered LDS slot,
//     and after the barrier every wave redundantly reduces the <=16 slots in one DPP row, ending
//     with the winner's coordinates in SGPRs for the next round;
//   * the tie key restates the published tree order: among equal maxima the point with the
//     smallest bit-reversed (k mod T), then the smallest k / T, wins -- independent of how points
//     are laid out over threads here;
//   * sampled indices are collected in LDS and written once at the end (no global store, hence no
//     vmcnt wait, inside the serial loop).
#include "common.h"

namespace {

struct FpsSlot {
    uint32_t best;  // f32 bits of the wave's maximum (>= 0)
    uint32_t key;   // tie key (lower wins)
    int32_t k;
    float x, y, z;
};

__device__ __forceinline__ uint32_t fps_tiekey(uint32_t k, uint32_t tmask, uint32_t log2t) {
    uint32_t r = k & tmask;
    uint32_t br = log2t ? (__brev(r) >> (32 - log2t)) : 0u;
    return (br << 20) | (k >> log2t);
}

template <int N>
struct VecOf {
    typedef float type __attribute__((ext_vector_type(N)));
};
template <>
struct VecOf<1> {
    typedef float type;
};

template <int P>
__device__ __forceinline__ float vec_get(const typename VecOf<P>::type &v, int i) {
    if constexpr (P == 1) return v; else return v[i];
}
template <int P>
__device__ __forceinline__ void vec_set(typename VecOf<P>::type &v, int i, float x) {
    if constexpr (P == 1) v = x; else v[i] = x;
}

// Visit order of a thread's points when T == 1024: ascending tie key.
// k = t + WGS*j, R = 1024/WGS residues per thread: k mod 1024 = t + WGS*(j mod R), k / 1024 = j / R.
template <int WGS, int P>
__host__ __device__ constexpr int fps_visit(int jj) {
    constexpr int R = 1024 / WGS;
    constexpr int Q = (P >= R) ? P / R : 1;
    if (P < R) return jj;                       // fewer points than residues: keys already ascend
    int ri = jj / Q, q = jj % Q;
    int r = 0;                                  // bit-reverse ri over log2(R) bits
    for (int b = 1, rb = R >> 1; b < R; b <<= 1, rb >>= 1)
        if (ri & b) r |= rb;
    return r + R * q;
}

// Cross-wave stage shared by both kernels: returns the winner of this round in (k, x, y, z).
template <int NW>
__device__ __forceinline__ void fps_combine(const FpsSlot *slots, int lane, int32_t &wk, float &wx,
                                            float &wy, float &wz) {
    uint32_t b = 0u, key = 0xFFFFFFFFu;
    int32_t k = 0;
    float x = 0.f, y = 0.f, z = 0.f;
    if (lane < NW) {
        FpsSlot s = slots[lane];
        b = s.best; key = s.key; k = s.k; x = s.x; y = s.y; z = s.z;
    }
    const uint32_t mx = dclr_row16_max_u32(b);
    const uint32_t kx = dclr_row16_min_u32(b == mx ? key : 0xFFFFFFFFu);
    const uint64_t win = __ballot(lane < NW && b == mx && key == kx);
    const int wl = __builtin_ctzll(win);
    wk = __builtin_amdgcn_readlane(k, wl);
    wx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), wl));
    wy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(y), wl));
    wz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(z), wl));
}

// ------------------------------------------------------------------------------------------------
// Kernel A: points and running minima in registers. n <= WGS * P.
// ------------------------------------------------------------------------------------------------
template <int WGS, int P>
__global__ __launch_bounds__(WGS) void fps_reg_kernel(int n, int pstride, int m,
                                                      const float *__restrict__ pts,
                                                      float *__restrict__ temp,
                                                      int32_t *__restrict__ idx, uint32_t tmask,
                                                      uint32_t log2t) {
    constexpr int NW = WGS / 64;
    typedef typename VecOf<P>::type vec;
    __shared__ FpsSlot slots[2][16];
    extern __shared__ int32_t picked[];          // m entries

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    pts += (size_t)blockIdx.x * n * pstride;
    idx += (size_t)blockIdx.x * m;
    if (temp) temp += (size_t)blockIdx.x * n;

    vec px, py, pz, td;
#pragma unroll
    for (int jj = 0; jj < P; ++jj) {
        const int k = t + WGS * fps_visit<WGS, P>(jj);
        float x = 0.f, y = 0.f, z = 0.f, d = -2.0f;   // -2: padding lanes can never beat best = -1
        if (k < n) {
            x = pts[(size_t)k * pstride + 0];
            y = pts[(size_t)k * pstride + 1];
            z = pts[(size_t)k * pstride + 2];
            d = temp ? temp[k] : 1e10f;
        }
        vec_set<P>(px, jj, x); vec_set<P>(py, jj, y); vec_set<P>(pz, jj, z); vec_set<P>(td, jj, d);
    }

    float cx = pts[0], cy = pts[1], cz = pts[2];
    if (t == 0) picked[0] = 0;

    for (int r = 1; r < m; ++r) {
        float best = -1.0f;
        int bjj = 0;
#pragma unroll
        for (int jj = 0; jj < P; ++jj) {
            const float d = dclr_sqdist(vec_get<P>(px, jj), vec_get<P>(py, jj), vec_get<P>(pz, jj), cx, cy, cz);
            const float o = vec_get<P>(td, jj);
            const float d2 = d < o ? d : o;
            vec_set<P>(td, jj, d2);
            const bool gt = d2 > best;
            bjj = gt ? jj : bjj;
            best = gt ? d2 : best;
        }
        const uint32_t bb = best < 0.f ? 0u : __float_as_uint(best);
        const int bk = t + WGS * fps_visit<WGS, P>(bjj);
        const uint32_t key = best < 0.f ? 0xFFFFFFFFu : fps_tiekey((uint32_t)bk, tmask, log2t);

        const uint32_t wmax = dclr_wave_max_u32(bb);
        const uint32_t wkey = dclr_wave_min_u32(bb == wmax ? key : 0xFFFFFFFFu);
        const uint64_t win = __ballot(bb == wmax && key == wkey);
        const int wl = __builtin_ctzll(win);
        const int wjj = __builtin_amdgcn_readlane(bjj, wl);      // uniform -> indexed VGPR read
        const float sx = vec_get<P>(px, wjj), sy = vec_get<P>(py, wjj), sz = vec_get<P>(pz, wjj);
        FpsSlot *slot = &slots[r & 1][0];
        if (lane == wl) {
            FpsSlot s;
            s.best = wmax; s.key = wkey; s.k = bk; s.x = sx; s.y = sy; s.z = sz;
            slot[wave] = s;
        }
        __syncthreads();
        int32_t wk;
        fps_combine<NW>(slot, lane, wk, cx, cy, cz);
        if (t == 0) picked[r] = wk;
    }

    __syncthreads();
    for (int i = t; i < m; i += WGS) idx[i] = picked[i];
    if (temp) {
#pragma unroll
        for (int jj = 0; jj < P; ++jj) {
            const int k = t + WGS * fps_visit<WGS, P>(jj);
            if (k < n) temp[k] = vec_get<P>(td, jj);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Kernel B: large clouds. Coordinates are re-read (L2-resident) every round; running minima live in
// registers (REG_P > 0, n <= 1024*REG_P) or in the caller's temp buffer (REG_P == 0, any n).
// 1024 threads, T == 1024, a thread's points k = t + 1024*j already ascend in tie key.
// ------------------------------------------------------------------------------------------------
template <int REG_P>
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int pstride, int m,
                                                          const float *__restrict__ pts,
                                                          float *__restrict__ temp,
                                                          int32_t *__restrict__ idx) {
    constexpr int WGS = 1024, NW = 16;
    constexpr int PR = REG_P > 0 ? REG_P : 1;
    __shared__ FpsSlot slots[2][16];
    extern __shared__ int32_t picked[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    pts += (size_t)blockIdx.x * n * pstride;
    idx += (size_t)blockIdx.x * m;
    if (temp) temp += (size_t)blockIdx.x * n;

    float td[PR];
    if constexpr (REG_P > 0) {
#pragma unroll
        for (int j = 0; j < PR; ++j) {
            const int k = t + WGS * j;
            td[j] = k < n ? (temp ? temp[k] : 1e10f) : -2.0f;
        }
    }
    float cx = pts[0], cy = pts[1], cz = pts[2];
    if (t == 0) picked[0] = 0;

    for (int r = 1; r < m; ++r) {
        float best = -1.0f;
        int bk = 0;
        if constexpr (REG_P > 0) {
#pragma unroll
            for (int j = 0; j < PR; ++j) {
                const int k = t + WGS * j;
                const int kc = k < n ? k : 0;
                const float *p = pts + (size_t)kc * pstride;
                const float d = dclr_sqdist(p[0], p[1], p[2], cx, cy, cz);
                const float o = td[j];
                const float d2 = d < o ? d : o;      // padding: o = -2 stays -2
                td[j] = d2;
                const bool gt = d2 > best;
                bk = gt ? k : bk;
                best = gt ? d2 : best;
            }
        } else {
            for (int k = t; k < n; k += WGS) {
                const float *p = pts + (size_t)k * pstride;
                const float d = dclr_sqdist(p[0], p[1], p[2], cx, cy, cz);
                const float o = temp[k];
                const float d2 = d < o ? d : o;
                temp[k] = d2;
                const bool gt = d2 > best;
                bk = gt ? k : bk;
                best = gt ? d2 : best;
            }
        }
        const uint32_t bb = best < 0.f ? 0u : __float_as_uint(best);
        const uint32_t key = best < 0.f ? 0xFFFFFFFFu : fps_tiekey((uint32_t)bk, 1023u, 10u);
        const uint32_t wmax = dclr_wave_max_u32(bb);
        const uint32_t wkey = dclr_wave_min_u32(bb == wmax ? key : 0xFFFFFFFFu);
        const uint64_t win = __ballot(bb == wmax && key == wkey);
        const int wl = __builtin_ctzll(win);
        FpsSlot *slot = &slots[r & 1][0];
        if (lane == wl) {
            FpsSlot s;
            s.best = wmax; s.key = wkey; s.k = bk; s.x = 0.f; s.y = 0.f; s.z = 0.f;
            slot[wave] = s;
        }
        __syncthreads();
        int32_t wk;
        float ux, uy, uz;
        fps_combine<NW>(slot, lane, wk, ux, uy, uz);
        const float *p = pts + (size_t)wk * pstride;   // uniform address: scalar load
        cx = p[0]; cy = p[1]; cz = p[2];
        if (t == 0) picked[r] = wk;
    }

    __syncthreads();
    for (int i = t; i < m; i += WGS) idx[i] = picked[i];
    if constexpr (REG_P > 0) {
        if (temp) {
#pragma unroll
            for (int j = 0; j < PR; ++j) {
                const int k = t + WGS * j;
                if (k < n) temp[k] = td[j];
            }
        }
    }
}

int fps_block(int n) {
    int t = 1;
    while (t * 2 <= n && t * 2 <= 1024) t *= 2;
    return t;
}

template <int WGS, int P>
void launch_reg(int b, int n, int pstride, int m, const float *pts, float *temp, int32_t *idx,
                hipStream_t s) {
    const int T = fps_block(n);
    int log2t = 0;
    while ((1 << log2t) < T) ++log2t;
    hipLaunchKernelGGL((fps_reg_kernel<WGS, P>), dim3(b), dim3(WGS), (size_t)m * sizeof(int32_t), s, n,
                       pstride, m, pts, temp, idx, (uint32_t)(T - 1), (uint32_t)log2t);
}

int fps_dispatch(int b, int n, int pstride, int m, const float *pts, float *temp, int32_t *idx,
                 hipStream_t s) {
    DCLR_REQUIRE(b > 0 && n > 0 && m > 0 && pstride >= 3 && pts && idx);
    if ((size_t)m * sizeof(int32_t) > 96 * 1024) return DCLR_E_UNSUPPORTED;   // picked[] lives in LDS
    if (n <= 1024) launch_reg<1024, 1>(b, n, pstride, m, pts, temp, idx, s);
    else if (n <= 2048) launch_reg<1024, 2>(b, n, pstride, m, pts, temp, idx, s);
    else if (n <= 4096) launch_reg<1024, 4>(b, n, pstride, m, pts, temp, idx, s);
    else if (n <= 8192) launch_reg<1024, 8>(b, n, pstride, m, pts, temp, idx, s);
    else if (n <= 16384) launch_reg<1024, 16>(b, n, pstride, m, pts, temp, idx, s);
    else if (n <= 32768)
        hipLaunchKernelGGL((fps_stream_kernel<32>), dim3(b), dim3(1024), (size_t)m * sizeof(int32_t), s, n,
                           pstride, m, pts, temp, idx);
    else if (n <= 65536)
        hipLaunchKernelGGL((fps_stream_kernel<64>), dim3(b), dim3(1024), (size_t)m * sizeof(int32_t), s, n,
                           pstride, m, pts, temp, idx);
    else if (temp)
        hipLaunchKernelGGL((fps_stream_kernel<0>), dim3(b), dim3(1024), (size_t)m * sizeof(int32_t), s, n,
                           pstride, m, pts, temp, idx);
    else
        return DCLR_E_UNSUPPORTED;
    return dclr_launch_status();
}

}  // namespace

extern "C" int dclr_furthest_point_sampling(int b, int n, int m, const float *points, float *temp,
                                            int32_t *idx, dclr_stream_t stream) {
    DCLR_REQUIRE(temp != nullptr);
    return fps_dispatch(b, n, 3, m, points, temp, idx, (hipStream_t)stream);
}

extern "C" int dclr_fps_clouds(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                               dclr_stream_t stream) {
    DCLR_REQUIRE(c >= 3);
    return fps_dispatch(b, n, c, m, clouds, nullptr, idx, (hipStream_t)stream);
}
