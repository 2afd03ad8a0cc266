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
//     vmcnt wait, inside the serial loop);
//   * (clouds of 1025..16384 points) the kernel first sorts the cloud by a coarse Morton key in LDS,
//     so that each wave owns a spatially compact 1/16 of the cloud, and from then on a region whose
//     bounding box is farther from the new sample than its largest running minimum skips the
//     round entirely: with the same rounded operation sequence, distance-to-box <= distance-to-point
//     (rounding is monotone), so no running minimum of that wave could change and its cached
//     candidate stays exact. The test runs per group of 4 register slots (64 compact regions of 256
//     points per cloud); typically a handful of the 64 groups do work in a round.
#include <stdlib.h>

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
        // (branch-free per element: a predicated load per point put 64 branches and 14 spilled registers into this prologue)
        if (temp) {
#pragma unroll
            for (int j = 0; j < PR; ++j) {
                const int k = t + WGS * j;
                const float v = temp[k < n ? k : 0];
                td[j] = k < n ? v : -2.0f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PR; ++j) td[j] = t + WGS * j < n ? 1e10f : -2.0f;
        }
    }
    float cx = pts[0], cy = pts[1], cz = pts[2];
    if (t == 0) picked[0] = 0;

    for (int r = 1; r < m; ++r) {
        float best = -1.0f;
        int bk = 0;
        if constexpr (REG_P > 0) {
            // The points' addresses do not change from round to round: hoisted out of the sampling loop they were 2 REG_P
            // registers beside the REG_P minima, and at 128 registers per lane the 32- and 64-point forms spilled 252 / 904
            // bytes. `tr` hides the thread index from that hoisting; eight (64-point form: four) points' loads are in flight at a time.
            int tr = t;
            asm volatile("" : "+v"(tr));
#pragma unroll
            for (int j = 0; j < PR; ++j) {
                if (j % (REG_P > 32 ? 4 : 8) == 0 && j > 0) __builtin_amdgcn_sched_barrier(0);
                const int k = tr + WGS * j;
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

// ------------------------------------------------------------------------------------------------
// Kernel A': as kernel A, plus spatial sorting and per-wave pruning. 1024 threads, T == 1024,
// 1024 < n <= 1024 * P, P a power of two.
// ------------------------------------------------------------------------------------------------
// x, y, z of point k of an interleaved cloud. Four floats per point on a 16-byte aligned cloud (the KITTI layout): ONE
// 16-byte request per lane instead of three 4-byte ones -- the setup passes that fetch points by sorted index are bound by
// the number of memory requests when 160 clouds are in flight.
__device__ __forceinline__ void fps_load_xyz(const float *__restrict__ pts, size_t k, int pstride, bool vec4, float &x,
                                             float &y, float &z) {
    if (vec4) {                                           // wave-uniform
        const float4 v = *reinterpret_cast<const float4 *>(pts + k * 4);
        x = v.x; y = v.y; z = v.z;
    } else {
        const float *p = pts + k * pstride;
        x = p[0]; y = p[1]; z = p[2];
    }
}

// Tie key for T == 1024 in 16 bits (n <= 65536): same order as fps_tiekey(k, 1023, 10), invertible.
__device__ __forceinline__ uint32_t fps_tk1024(uint32_t k) { return ((__brev(k & 1023u) >> 22) << 6) | (k >> 10); }
__device__ __forceinline__ uint32_t fps_tk1024_inv(uint32_t tk) { return (__brev(tk >> 6) >> 22) | ((tk & 63u) << 10); }

// Wave minimum / maximum of floats (no NaNs) through the DPP reductions of common.h: the bit pattern is mapped to an unsigned
// key of the same order (sign bit flipped for positives, all bits for negatives). Six dependent ds_bpermute shuffles per
// reduction were 12 us of the 16384-point sampler's setup once the slice boxes (24 reductions per group) were added.
__device__ __forceinline__ uint32_t fps_ordered_key(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float fps_from_ordered_key(uint32_t k) {
    return __uint_as_float(k ^ (((k >> 31) - 1u) | 0x80000000u));
}
__device__ __forceinline__ float fps_shfl_min(float v) { return fps_from_ordered_key(dclr_wave_min_u32(fps_ordered_key(v))); }
__device__ __forceinline__ float fps_shfl_max(float v) { return fps_from_ordered_key(dclr_wave_max_u32(fps_ordered_key(v))); }

// Rounded lower bound of dclr_sqdist(p, c) over all p inside the box [lo, hi] (same operation
// sequence as dclr_sqdist; see the header comment for why it is a bound on the ROUNDED distance).
__device__ __forceinline__ float fps_box_lower_bound(float lx, float ly, float lz, float hx, float hy, float hz,
                                                     float cx, float cy, float cz) {
    const float dx = fmaxf(fmaxf(lx - cx, cx - hx), 0.f);
    const float dy = fmaxf(fmaxf(ly - cy, cy - hy), 0.f);
    const float dz = fmaxf(fmaxf(lz - cz, cz - hz), 0.f);
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

#ifdef FPS_DEBUG
__device__ unsigned long long fps_dbg[16];
__device__ unsigned long long fps_dbg_setup[8];       // cloud 0, wave 0: cycle stamps along the setup of fps_pruned_kernel
__device__ unsigned int fps_grp[64];
__device__ unsigned long long fps_bucket[16][6][3];   // per wave, per marks-in-round bucket (0..4, 5 = rewrites>=1...): rounds, cycles, rewrites         // table mode: rounds in which group q was marked (cloud 0)     // [0] active (wave, round) count, [1..] cycle sums (wave 0)
#define FPS_STAMP(v) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); } while (0)
#endif

#ifndef FPS_J
#define FPS_J 4           // samples a barrier round of the table mode may accept (5 and 6 measured equal: the leader pays
#endif                    // ~370 cycles per candidate for 0.4 more samples per round); per-wave candidates keep 3

struct FpsCand {          // 16 bytes: one ds_write_b128 / ds_read_b128
    int32_t k;
    float x, y, z;
};

// The 12-bit sorting cell. The counting sort only decides which wave / group owns which point (any order yields the
// same samples), but the tighter the groups' boxes, the fewer groups a sample touches -- here and in set abstraction,
// which scans the same groups. The 12 bits are dealt to the axes by extent: each next bit halves the axis whose cells
// are currently the widest, and the key interleaves the bits in that same order (a k-d-like space-filling order).
// A LiDAR scan (160 x 160 x 4 m) gets 6 + 6 + 0 bits = 2.5 m cells; the former 4 + 4 + 4 bits with one common cell size
// gave it 10 m cells, 256 of the 4096 bins in use, and a dense cell spanned several groups with identical boxes.
struct FpsGrid {
    float scale[3];       // cells per unit length
    int bits[3];
    uint32_t order;       // 2 bits per key bit, most significant key bit first: the axis it comes from
};

__device__ __forceinline__ FpsGrid fps_make_grid(const float (&ext)[3]) {
    FpsGrid g;
    float width[3] = {ext[0], ext[1], ext[2]};
    g.bits[0] = g.bits[1] = g.bits[2] = 0;
    g.order = 0u;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int a = (width[1] > width[0] ? (width[2] > width[1] ? 2 : 1) : (width[2] > width[0] ? 2 : 0));
        g.order = (g.order << 2) | (uint32_t)a;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g.bits[c] += a == c ? 1 : 0;
            width[c] = a == c ? width[c] * 0.5f : width[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) g.scale[c] = ext[c] > 0.f ? ((float)(1 << g.bits[c]) - 0.001f) / ext[c] : 0.f;
    return g;
}

// The cell coordinates of a point (offsets from the cloud's minimum corner) ...
__device__ __forceinline__ void fps_cell_coords(const FpsGrid &g, float dx, float dy, float dz, uint32_t (&q)[3]) {
    const float d[3] = {dx, dy, dz};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v = (int)(d[c] * g.scale[c]);
        const int top = (1 << g.bits[c]) - 1;
        q[c] = (uint32_t)(v < 0 ? 0 : (v > top ? top : v));
    }
}

// ... and their bits dealt into the key. Every key bit comes from exactly one axis, so
// key(q0, q1, q2) = key(q0, 0, 0) | key(0, q1, 0) | key(0, 0, q2): with at most 8 bits per axis the kernels look the three
// parts up in a 3 x 256 table built once per cloud (fps_cell_lut) instead of running this 12-step loop per point (85
// instructions per point: 40 % of the 16384-point sampler's setup).
__device__ __forceinline__ uint32_t fps_cell_key(const FpsGrid &g, uint32_t q0, uint32_t q1, uint32_t q2) {   // (scalars: an array here ends up in scratch, indexed by the axis)
    int left[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) left[c] = g.bits[c];
    uint32_t key = 0u;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int a = (int)((g.order >> (2 * (11 - i))) & 3u);
        // the axis' next most significant unused bit
        const int sh = (a == 0 ? left[0] : a == 1 ? left[1] : left[2]) - 1;
        const uint32_t qa = a == 0 ? q0 : a == 1 ? q1 : q2;
        key = (key << 1) | ((qa >> sh) & 1u);
        left[0] -= a == 0 ? 1 : 0; left[1] -= a == 1 ? 1 : 0; left[2] -= a == 2 ? 1 : 0;
    }
    return key;
}

__device__ __forceinline__ uint32_t fps_cell(const FpsGrid &g, float dx, float dy, float dz) {
    uint32_t q[3];
    fps_cell_coords(g, dx, dy, dz, q);
    return fps_cell_key(g, q[0], q[1], q[2]);
}

// Table of the per-axis key parts (see fps_cell_key); false if an axis has more than 8 bits (a cloud stretched along a
// line: the callers then keep the loop). All threads of the workgroup call it; ends with a barrier when it returns true.
template <int WGS>
__device__ __forceinline__ bool fps_cell_lut(const FpsGrid &g, uint16_t (*lut)[256], int t) {
    if (g.bits[0] > 8 || g.bits[1] > 8 || g.bits[2] > 8) return false;      // the same for every thread
    for (int e = t; e < 3 * 256; e += WGS) {
        const int c = e >> 8;
        const uint32_t v = (uint32_t)(e & 255);
        lut[c][e & 255] = (uint16_t)fps_cell_key(g, c == 0 ? v : 0u, c == 1 ? v : 0u, c == 2 ? v : 0u);   // entries beyond 2^bits are never read
    }
    __syncthreads();
    return true;
}

// WGS threads, P points per thread (WGS * P = padded cloud size, a power of two), G groups per wave.
template <int WGS, int P, int G, int MODE>      // MODE 0: one sample per barrier round; 1: several (per-wave candidates);
                                                // 3: several, per-GROUP candidate table behind a leader wave (default)
__global__ __launch_bounds__(WGS) void fps_pruned_kernel(int n, int pstride, int m,
                                                         const float *__restrict__ pts,
                                                         float *__restrict__ temp,
                                                         int32_t *__restrict__ idx,
                                                         float4 *__restrict__ group_pts,
                                                         float *__restrict__ group_box, DclrCloudView view,
                                                         float *__restrict__ slice_box) {
    constexpr int NW = WGS / 64, NP = WGS * P, BINS = 4096, S = P / G;
    static_assert(P % G == 0 && G <= 16 && NW <= 16 && BINS % WGS == 0, "layout");
    typedef typename VecOf<P>::type vec;
    __shared__ unsigned long long cell[3];                 // per-round 64-bit arg-max cells, rotated
    __shared__ FpsCand cand[2][16];                        // per-wave candidate payload, by round parity
    __shared__ unsigned long long wpk[2][16];              // MULTI: per-wave packed candidate, by round parity
    __shared__ uint32_t wru[2][16];                        // MULTI: per-wave runner-up (largest other running minimum)
    __shared__ float plist[FPS_J][4];                      // MULTI: the samples accepted for the next round
    __shared__ int plist_n;
    __shared__ uint4 gtab[64][2];                          // MODE 3: one entry per group: {value, key, runner-up, index}, {x, y, z, -}
    __shared__ float red[6][16];
    __shared__ uint32_t wsum[16];
    // BINS counters (u32), NP sorted indices / tie keys (u16: n <= 16384 and keys < 0xFFFF), then NP cell ids (u16)
    // whose storage `picked` (i32 x m) takes over after the sort: 16 + 32 + 32 KB at NP = 16384
    extern __shared__ uint32_t dyn_lds[];
    uint32_t *hist = dyn_lds;
    uint16_t *sbuf = reinterpret_cast<uint16_t *>(dyn_lds + BINS);
    int32_t *picked = reinterpret_cast<int32_t *>(dyn_lds + BINS + NP / 2);   // shares storage with `cellof`

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    pts += dclr_cloud_offset(view, blockIdx.x, (size_t)n * pstride);
    const bool vec4 = pstride == 4 && ((uintptr_t)pts & 15) == 0;          // wave-uniform
    idx += (size_t)blockIdx.x * m;
    if (temp) temp += (size_t)blockIdx.x * n;
    if (group_pts) group_pts += (size_t)blockIdx.x * NP;
    if (group_box) group_box += (size_t)blockIdx.x * NW * G * 8;
    if (slice_box) slice_box += (size_t)blockIdx.x * NW * G * S * 8;

#ifdef FPS_DEBUG
    if (blockIdx.x == 0 && t == 0) { unsigned long long ts_; FPS_STAMP(ts_); fps_dbg_setup[0] = ts_; }
#endif
    // ---- 1. bounding box of the cloud --------------------------------------------------------------
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    float first_x[P], first_y[P], first_z[P];              // the thread's P points stay in registers for the cell pass (step 2)
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = t + WGS * j;
        first_x[j] = first_y[j] = first_z[j] = 0.f;
        if (k < n) {
            fps_load_xyz(pts, (size_t)k, pstride, vec4, first_x[j], first_y[j], first_z[j]);
            lo[0] = fminf(lo[0], first_x[j]); lo[1] = fminf(lo[1], first_y[j]); lo[2] = fminf(lo[2], first_z[j]);
            hi[0] = fmaxf(hi[0], first_x[j]); hi[1] = fmaxf(hi[1], first_y[j]); hi[2] = fmaxf(hi[2], first_z[j]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = fps_shfl_min(lo[a]), h = fps_shfl_max(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
#pragma unroll
    for (int u = 0; u < BINS / WGS; ++u) hist[t + WGS * u] = 0u;
    if (t < 3) cell[t] = 0ull;
    if (t < 32) cand[t >> 4][t & 15] = FpsCand{0, 0.f, 0.f, 0.f};
    __syncthreads();
    float ext[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        lo[a] = l;
        ext[a] = h - l;
    }
#ifdef FPS_DEBUG
    if (blockIdx.x == 0 && t == 0) { unsigned long long ts_; FPS_STAMP(ts_); fps_dbg_setup[1] = ts_; }
#endif
    // ---- 2. counting sort by a 12-bit cell (bits dealt to the axes by extent, FpsGrid). The sort
    //         only decides which wave owns which point; any order yields the same samples. -----------
    const FpsGrid grid = fps_make_grid(ext);
    __shared__ uint16_t cell_lut[3][256];
    const bool use_lut = fps_cell_lut<WGS>(grid, cell_lut, t);
    uint16_t *cellof = sbuf + NP;                          // scratch list of cell ids, dead before `picked` is used
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = t + WGS * j;
        if (k < n) {
            uint32_t q3[3];
            fps_cell_coords(grid, first_x[j] - lo[0], first_y[j] - lo[1], first_z[j] - lo[2], q3);
            const uint32_t mc = use_lut ? (uint32_t)cell_lut[0][q3[0]] | cell_lut[1][q3[1]] | cell_lut[2][q3[2]]
                                        : fps_cell_key(grid, q3[0], q3[1], q3[2]);
            atomicAdd(&hist[mc], 1u);
            cellof[k] = (uint16_t)mc;
        }
    }
    __syncthreads();
    {   // exclusive prefix over the counters: BPT per thread, wave scan, NW wave totals
        constexpr int BPT = BINS / WGS;
        uint32_t c[BPT], mine = 0;
#pragma unroll
        for (int u = 0; u < BPT; ++u) { c[u] = hist[BPT * t + u]; mine += c[u]; }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int u = 0; u < BPT; ++u) { hist[BPT * t + u] = run; run += c[u]; }
    }
    __syncthreads();
    // With the groups exported (group_pts: the sorted point list set abstraction reads -- position p of it IS sorted position
    // p) every thread writes its points there NOW, from the registers of step 1, at the positions the sort hands out; step 3
    // then reads each group's points back in order, coalesced and mostly from L2, instead of gathering 16 bytes per point from
    // all over the cloud by sorted index (64-byte sectors for 16-byte requests: that gather was 190 of the 281 MB a
    // 160-cloud launch moved, and two dependent round trips of the setup).
    const bool via_export = group_pts != nullptr;          // wave-uniform
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = t + WGS * j;
        if (k < n) {
            const uint32_t pos = atomicAdd(&hist[cellof[k]], 1u);
            sbuf[pos] = (uint16_t)k;
            if (via_export) group_pts[pos] = make_float4(first_x[j], first_y[j], first_z[j], __uint_as_float((uint32_t)k));
        }
    }
    if (via_export) {
        for (int pos = n + t; pos < NP; pos += WGS)        // padding slots: far away, no index
            group_pts[pos] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __uint_as_float(0xFFFFFFFFu));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __syncthreads();
    if (via_export) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef FPS_DEBUG
    if (blockIdx.x == 0 && t == 0) { unsigned long long ts_; FPS_STAMP(ts_); fps_dbg_setup[2] = ts_; }
#endif
    // ---- 3. this thread's points: wave w owns sorted positions [w*64*P, (w+1)*64*P); its P register
    //         slots form G groups of S consecutive slots, i.e. G spatially compact runs of 64*S points.
    // Inside each group of a thread the tie keys ascend, so a strict ">" scan over the group keeps the
    // right point on equal distances; groups and lanes are merged with an explicit key comparison.
    // The sorted keys go back to LDS (same positions, now in slot order): the hot loop never needs
    // them in registers, only the winner's key is fetched once per round.
    // MODE 3 deals the sorted groups out by how often they will be touched instead. Samples spread evenly in SPACE, so a
    // group is marked in proportion to the size of its box: the sparse outskirts' groups in (nearly) every round, the
    // dense core's hardly ever (measured: 20 to 300 of 324 rounds). A round lasts as long as its busiest wave, so the
    // groups are ranked by box size (half the surface area) and dealt in serpentine order: wave w gets ranks w,
    // 2 NW - 1 - w, 2 NW + w, ... -- every wave one large, one small and two medium groups, and neighbours in space
    // (adjacent ranks are mostly adjacent regions) on different waves and SIMDs. Costs one extra pass over the points.
    // Since the second session of round 3 the default is the plain round-robin deal in key order again (group g of wave w =
    // sorted group g NW + w): the ranking needs a pass of its own over the points, fetched by sorted index -- 26k cycles per
    // cloud alone, 25 us of a 750 us launch with 160 clouds in flight -- and never bought more than it cost (737 vs 729 us
    // when it was introduced); the slice boxes come out of the registers of step 3 instead. 752 -> 720 us per 160 clouds.
#ifndef FPS_ROUND_ROBIN
#define FPS_ROUND_ROBIN 1                       // MODE 3 deals its groups round-robin in key order; 0 (A/B builds): by box size, below
#endif
    constexpr bool MAPPED = MODE == 3;                     // groups owned through the dq[] map (else: a wave's contiguous share)
    constexpr bool DEALT = MODE == 3 && !FPS_ROUND_ROBIN;  // ... and the map comes from the box-size ranking
    constexpr bool SLICES_IN_STEP3 = MAPPED && !DEALT;     // slice boxes from the registers of step 3 (no pass of their own)
    int dq[G];                                             // sorted group of this wave's group g (wave-uniform)
#pragma unroll
    for (int g = 0; g < G; ++g) dq[g] = (MAPPED && !DEALT) ? g * NW + wave : wave * G + g;
    // Extent of sorted group q (the wave's contiguous share, before dealing) and, on request, the boxes of its S slices
    // (slice i = sorted positions [(q S + i) 64, + 64): a compact sub-cell; set abstraction tests them one by one)
    auto group_extent = [&](int q, float (&d)[3]) {
        float glo3[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ghi3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int pos = (q * S + i) * 64 + lane;
            float lo3[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            if (pos < n) {
                fps_load_xyz(pts, (size_t)sbuf[pos], pstride, vec4, lo3[0], lo3[1], lo3[2]);
#pragma unroll
                for (int a = 0; a < 3; ++a) hi3[a] = lo3[a];
            }
            if (slice_box != nullptr && S > 1) {            // wave-uniform
                float sb[6];
#pragma unroll
                for (int a = 0; a < 3; ++a) { sb[a] = fps_shfl_min(lo3[a]); sb[3 + a] = fps_shfl_max(hi3[a]); }
                if (lane < 8)
                    slice_box[(size_t)(q * S + i) * 8 + lane] =
                        lane == 0 ? sb[0] : lane == 1 ? sb[1] : lane == 2 ? sb[2] : lane == 3 ? sb[3]
                        : lane == 4 ? sb[4] : lane == 5 ? sb[5] : 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) { glo3[a] = fminf(glo3[a], sb[a]); ghi3[a] = fmaxf(ghi3[a], sb[3 + a]); }
            } else {
#pragma unroll
                for (int a = 0; a < 3; ++a) { glo3[a] = fminf(glo3[a], lo3[a]); ghi3[a] = fmaxf(ghi3[a], hi3[a]); }
            }
        }
        const bool reduced = slice_box != nullptr && S > 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) d[a] = reduced ? ghi3[a] - glo3[a] : fps_shfl_max(ghi3[a]) - fps_shfl_min(glo3[a]);
    };
    if constexpr (!DEALT && !SLICES_IN_STEP3) {
        if (slice_box != nullptr && S > 1) {
#pragma unroll
            for (int g = 0; g < G; ++g) { float d[3]; group_extent(wave * G + g, d); }
        }
    }
    if constexpr (DEALT) {
        constexpr int NGR = NW * G;
        static_assert(NGR <= 64, "one group per lane");
        float *ghot = red[0];                              // 64 floats of the 96 in `red` (free again after step 1)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int q = wave * G + g;                    // provisional: the wave's contiguous share
            float d[3];
            group_extent(q, d);
            if (lane == 0) ghot[q] = d[0] >= 0.f ? d[0] * d[1] + d[1] * d[2] + d[2] * d[0] : -1.0f;   // empty group: last
        }
        __syncthreads();
        const float h = lane < NGR ? ghot[lane] : -2.0f;
        int rank = 0;
#pragma unroll 8
        for (int p2 = 0; p2 < NGR; ++p2) {
            const float hp = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(h), p2));
            rank += (hp > h || (hp == h && p2 < lane)) ? 1 : 0;
        }
        const int level = rank / NW, along = rank % NW;
        const int owner = (level & 1) ? NW - 1 - along : along;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint64_t mine = __ballot(lane < NGR && level == g && owner == wave);
            dq[g] = mine != 0 ? __builtin_ctzll(mine) : wave * G + g;
        }
        __syncthreads();                                   // `red` is read again below only after further barriers; be safe
    }
#ifdef FPS_DEBUG
    if (blockIdx.x == 0 && t == 0) { unsigned long long ts_; FPS_STAMP(ts_); fps_dbg_setup[3] = ts_; }
#endif
    auto deal = [&](int g) -> int {                        // g is a compile-time constant in every unrolled caller
        int q = dq[0];
#pragma unroll
        for (int u = 1; u < G; ++u) q = g == u ? dq[u] : q;
        return q;
    };
    auto slot_pos = [&](int jj, int ln) -> int {           // position in sbuf of lane ln's slot jj (this wave); jj a constant
        if constexpr (MAPPED) return (deal(jj / S) * S + (jj % S)) * 64 + ln;
        else return wave * 64 * P + jj * 64 + ln;
    };
    auto slot_pos_g = [&](int g, int jj, int ln) -> int {  // the same for a slot of group g (a constant) given at run time
        return (deal(g) * S + (jj - g * S)) * 64 + ln;
    };
    vec px, py, pz, td;
    float glo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ghi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};   // lane g: box of group g
    float gmaxv = 0.f;                                     // lane g: upper bound of group g's largest running minimum;
                                                           // 0 for an empty group (its box bound is +inf: never active)
    float gbest[G];                                        // per lane: largest running minimum inside the group
    int gjj[G];                                            // ... and the slot holding it (first one in key order)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        uint32_t tkg[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int pos = slot_pos(g * S + i, lane);
            // (key << 2 | run): the sort below orders a lane's slots by tie key; the run a point came from -- 64 consecutive
            // positions of the sorted cloud, a compact sub-cell -- decides where the EXPORTED copy of it goes
            tkg[i] = ((pos < n ? fps_tk1024(sbuf[pos]) : 0xFFFFu) << 2) | (uint32_t)i;
        }
#pragma unroll
        for (int k2 = 2; k2 <= S; k2 <<= 1)
#pragma unroll
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1)
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const int l = i ^ j2;
                    if (l > i) {
                        const uint32_t a = tkg[i], b = tkg[l];
                        const uint32_t mn = a < b ? a : b, mxv = a < b ? b : a;
                        const bool asc = (i & k2) == 0;
                        tkg[i] = asc ? mn : mxv;
                        tkg[l] = asc ? mxv : mn;
                    }
                }
        float blo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        bool any = false;
        uint32_t run[S];
#pragma unroll
        for (int i = 0; i < S; ++i) { run[i] = tkg[i] & 3u; tkg[i] >>= 2; }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const int jj = g * S + i;
            sbuf[slot_pos(jj, lane)] = (uint16_t)tkg[i];   // own positions only: no cross-thread hazard
            float x = 0.f, y = 0.f, z = 0.f, d = -2.0f;   // -2: padding can never beat best = -1
            if (tkg[i] != 0xFFFFu) {
                const uint32_t k = fps_tk1024_inv(tkg[i]);
                if (via_export) {                           // sorted position of this slot's point: run run[i] of the group
                    const float4 v = group_pts[(size_t)deal(g) * (64 * S) + run[i] * 64 + lane];
                    x = v.x; y = v.y; z = v.z;
                } else {
                    fps_load_xyz(pts, (size_t)k, pstride, vec4, x, y, z);
                }
                d = temp ? temp[k] : 1e10f;
                blo[0] = fminf(blo[0], x); blo[1] = fminf(blo[1], y); blo[2] = fminf(blo[2], z);
                bhi[0] = fmaxf(bhi[0], x); bhi[1] = fmaxf(bhi[1], y); bhi[2] = fmaxf(bhi[2], z);
                any = true;
            }
            vec_set<P>(px, jj, x); vec_set<P>(py, jj, y); vec_set<P>(pz, jj, z); vec_set<P>(td, jj, d);
            // (the regrouped copy for the set-abstraction fast path -- slice r of exported group q = the r-th 64 consecutive
            // sorted positions of the group, their boxes in slice_box -- was written in step 2, padding included)
        }
        if constexpr (SLICES_IN_STEP3 && S > 1) {
            if (slice_box != nullptr) {                     // wave-uniform: a lane holds one point of each of the S slices (runs)
#pragma unroll
                for (int r = 0; r < S; ++r) {
                    float c3[3] = {vec_get<P>(px, g * S), vec_get<P>(py, g * S), vec_get<P>(pz, g * S)};
                    bool ok = tkg[0] != 0xFFFFu;
#pragma unroll
                    for (int i = 1; i < S; ++i) {
                        const bool mine = run[i] == (uint32_t)r;
                        c3[0] = mine ? vec_get<P>(px, g * S + i) : c3[0];
                        c3[1] = mine ? vec_get<P>(py, g * S + i) : c3[1];
                        c3[2] = mine ? vec_get<P>(pz, g * S + i) : c3[2];
                        ok = mine ? tkg[i] != 0xFFFFu : ok;
                    }
                    float sb[6];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        sb[a] = fps_shfl_min(ok ? c3[a] : 3.0e38f);
                        sb[3 + a] = fps_shfl_max(ok ? c3[a] : -3.0e38f);
                    }
                    if (lane < 8)
                        slice_box[(size_t)(deal(g) * S + r) * 8 + lane] =
                            lane == 0 ? sb[0] : lane == 1 ? sb[1] : lane == 2 ? sb[2] : lane == 3 ? sb[3]
                            : lane == 4 ? sb[4] : lane == 5 ? sb[5] : 0.f;
                }
            }
        }
        float box[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = fps_shfl_min(blo[a]), h = fps_shfl_max(bhi[a]);
            if (lane == g) { glo[a] = l; ghi[a] = h; }
            box[a] = l; box[3 + a] = h;
        }
        if (group_box && lane < 8)
            group_box[(size_t)deal(g) * 8 + lane] =
                lane == 0 ? box[0] : lane == 1 ? box[1] : lane == 2 ? box[2] : lane == 3 ? box[3]
                : lane == 4 ? box[4] : lane == 5 ? box[5] : 0.f;
        gbest[g] = any ? 0.f : -1.0f;
        gjj[g] = g * S;
        const bool group_any = __ballot(any) != 0;                               // all lanes vote: NOT inside `lane == g &&`
        if (lane == g && group_any) gmaxv = __uint_as_float(0x7F800000u);        // +inf forces the first update
    }
    __syncthreads();                                       // `cellof` is dead: `picked` (same storage) may be written

#ifdef FPS_DEBUG
    if (blockIdx.x == 0 && t == 0) { unsigned long long ts_; FPS_STAMP(ts_); fps_dbg_setup[4] = ts_; }
#endif
    float cx = pts[0], cy = pts[1], cz = pts[2];
    if (t == 0) picked[0] = 0;

    unsigned long long c_packed = (unsigned long long)wave;      // best 0, worst key: never wins against a real point
    int32_t c_k = 0;
    float c_x = 0.f, c_y = 0.f, c_z = 0.f;
    int fresh = 0;                                                // rounds for which cand[parity][wave] is still stale

#ifdef FPS_DEBUG
    unsigned long long t_setup;
    FPS_STAMP(t_setup);
    unsigned long long acc_upd = 0, acc_bar = 0, acc_comb = 0, n_act = 0;
    unsigned long long dbg_box = 0, dbg_upd = 0, dbg_sel = 0, dbg_pub = 0, dbg_idle = 0, dbg_nact = 0;
#endif
    if constexpr (MODE == 3) {
        // ---- several samples per barrier round, one table entry per GROUP, everything global done by a leader wave ----
        // The CU is issue-bound here (16 waves share 4 SIMDs and every instruction of every wave counts), so a round is
        // laid out for the fewest instructions in total:
        //   table    every group (64 S points) keeps an exact entry in LDS: its largest running minimum b_g, that point's
        //            tie key, index and coordinates, and the runner-up u_g (largest running minimum among the group's
        //            OTHER points); its bounding box sits beside it;
        //   leader   (wave 0, lane = group) reads the <= 64 entries. p1 = best entry overall is sample r. The best
        //            remaining entry c (value v) is sample r + 1 as well if sqdist(c, p1) >= v (p1 leaves it untouched) and
        //            u_{g(p1)} < v (nothing else in p1's group can reach it; running minima only decrease): every other
        //            point is already ordered behind c by its group's arg-max (ties by key). The same test against every
        //            sample accepted so far admits c as sample r + j. The J candidates are extracted first (J chained
        //            wave maxima), then all tests run side by side. The leader ALSO tests every accepted sample against
        //            all group boxes at once (one lane per group: a group can change iff the rounded lower bound of its
        //            distance to the sample is below b_g) and publishes one 64-bit mask per sample;
        //   workers  a wave looks up its groups' bits: none set (the usual case) -> straight to the next barrier.
        //            Otherwise each marked group takes exactly the samples that marked it, then one pass finds the lane's
        //            best and second-best slot, one top-2 reduction over the wave gives b_g and u_g, and lane 0 rewrites
        //            the entry.
        // Against per-wave candidates (MODE 1): 3.1 instead of 2.4 samples per round on the bench clouds (the runner-up of
        // 64 S points instead of 1024 stands in the way less often); no wave but the leader runs box tests (they were
        // 40 % of the instructions of a round); the groups are dealt out round-robin, so the handful a sample touches
        // are updated on different SIMDs; and the picks are not one serial chain of test -> pick -> test.
        constexpr int J = FPS_J, NG = NW * G;
        static_assert(NG <= 64 && J <= 8, "one table entry per lane, one (j, i) pair of candidates per lane");
        __shared__ float gbox[64][8];                     // box of group q (min xyz, max xyz), written once
        __shared__ float4 rb_s[8];                        // the round's samples: x, y, z, -
        __shared__ unsigned long long rb_m[8];            // ... and the groups each of them can change
        float gval = gmaxv;                               // lane g < G: b_g of this wave's group g
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (lane == g) {
                const int q = dq[g];
                // a non-empty group starts at +inf (gmaxv): sample 0 then marks it; an empty one stays at 0 (bound +inf: never marked)
                gtab[q][0] = make_uint4(__float_as_uint(gval), 0xFFFFu, 0u, 0u);
                gtab[q][1] = make_uint4(0u, 0u, 0u, 0u);
                gbox[q][0] = glo[0]; gbox[q][1] = glo[1]; gbox[q][2] = glo[2];
                gbox[q][3] = ghi[0]; gbox[q][4] = ghi[1]; gbox[q][5] = ghi[2];
            }
        }
        if (t < 8) {
            // round 1: sample 0 = point 0, marked for every group (the table is not built yet)
            rb_s[t] = make_float4(cx, cy, cz, 0.f);
            rb_m[t] = t == 0 ? ~0ull : 0ull;
        }
        __syncthreads();
        // bits of this wave's groups in a mask: group g of this wave is table entry deal(g)
        unsigned long long own = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) own |= 1ull << deal(g);
        int sr = 0;
        int cur_wl[G], cur_jj[G];                         // per group of this wave: lane and slot holding the entry's point ...
        uint32_t cur_b[G];                                // ... and b_g as last published (wave-uniform; ~0: none yet)
#pragma unroll
        for (int g = 0; g < G; ++g) { cur_wl[g] = 0; cur_jj[g] = g * S; cur_b[g] = 0xFFFFFFFFu; }
#ifdef FPS_DEBUG
        unsigned long long mu = 0, mb = 0, mc = 0, mt = 0, mg = 0, mr = 0, mg_prev = 0, mr_prev = 0, ml1 = 0, ml2 = 0, ml3 = 0;
#endif
        for (int r = 0;;) {                                               // r: samples picked so far
#ifdef FPS_DEBUG
            unsigned long long q0, q1, q2, q3;
            FPS_STAMP(q0);
#endif
            // -- workers ------------------------------------------------------------------------------------------
            const unsigned long long mmask = rb_m[lane & 7];              // lane j < J: the groups sample j can change
            const float4 smp = rb_s[lane & 7];                            // ... and the sample (read now: one LDS round trip, not two)
            const uint32_t live = (uint32_t)__ballot(lane < J && mmask != 0ull);       // the samples picked last (each marks its own group)
            r += __builtin_popcount(live);
            if (r >= m) break;                                            // all picked (the last ones are never applied: see `temp`)
            const uint32_t mine = (uint32_t)__ballot(lane < J && (mmask & own) != 0ull);
            if (mine != 0) {                                              // wave-uniform: some sample reaches a group of this wave
#ifdef FPS_DEBUG
                mt += 1;
#endif
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const unsigned long long gbit = 1ull << deal(g);
                    const uint32_t sel = (uint32_t)__ballot(lane < J && (mmask & gbit) != 0ull);
                    if (sel == 0) continue;                               // wave-uniform
#ifdef FPS_DEBUG
                    mg += 1;
#endif
                    // the samples that marked this group, one after the other over the S slots of the lane's cell
                    // (two slots per instruction: v_pk_add_f32 / v_pk_mul_f32 are the IEEE operations of dclr_sqdist in the
                    // same order, so the distances are bit-identical)
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    for (uint32_t rem = sel; rem != 0; rem &= rem - 1) {
                        const int j = __builtin_ctz(rem);
                        const float sx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(smp.x), j));
                        const float sy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(smp.y), j));
                        const float sz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(smp.z), j));
                        if constexpr (S % 2 == 0) {
                            const f2 c2x = {sx, sx}, c2y = {sy, sy}, c2z = {sz, sz};
#pragma unroll
                            for (int i = 0; i < S; i += 2) {
                                const int jj = g * S + i;
                                const f2 ax = {vec_get<P>(px, jj), vec_get<P>(px, jj + 1)};
                                const f2 ay = {vec_get<P>(py, jj), vec_get<P>(py, jj + 1)};
                                const f2 az = {vec_get<P>(pz, jj), vec_get<P>(pz, jj + 1)};
                                const f2 dx = ax - c2x, dy = ay - c2y, dz = az - c2z;
                                const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                                const f2 d = (xx + yy) + zz;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    float d2;
                                    asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d[h]), "v"(vec_get<P>(td, jj + h)));
                                    vec_set<P>(td, jj + h, d2);
                                }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < S; ++i) {
                                const int jj = g * S + i;
                                const float d = dclr_sqdist(vec_get<P>(px, jj), vec_get<P>(py, jj), vec_get<P>(pz, jj), sx, sy, sz);
                                float d2;
                                asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(vec_get<P>(td, jj)));
                                vec_set<P>(td, jj, d2);
                            }
                        }
                    }
                    // Did the group's best point keep its value? Running minima only decrease, so then b_g and the entry's
                    // point stand; the runner-up in the table may now be too large, which only makes the leader's test
                    // (v > u_g) more cautious. Most updates nibble at a group's fringe and end here.
                    // (the entry's own SLOT is checked, not its lane's maximum: on tie-heavy clouds another slot of the lane
                    // may hold the same value, and then the entry has to move to that point)
                    {
                        float ev = vec_get<P>(td, g * S);
#pragma unroll
                        for (int i = 1; i < S; ++i) ev = cur_jj[g] == g * S + i ? vec_get<P>(td, g * S + i) : ev;   // uniform selects
                        const uint32_t now = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ev), cur_wl[g]);
                        if (now == cur_b[g]) continue;                    // wave-uniform
                    }
#ifdef FPS_DEBUG
                    mr += 1;
#endif
                    // this lane's best and second-best slot (slots ascend in tie key: strict > keeps the first)
                    float best = -1.0f, sec = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
                    int bjj = g * S;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const int jj = g * S + i;
                        const float v = vec_get<P>(td, jj);
                        const bool gt = v > best;
                        sec = fmaxf(sec, gt ? best : v);
                        bjj = gt ? jj : bjj;
                        bx = gt ? vec_get<P>(px, jj) : bx; by = gt ? vec_get<P>(py, jj) : by; bz = gt ? vec_get<P>(pz, jj) : bz;
                        best = gt ? v : best;
                    }
                    // the group: top-2 over the wave of (best, second) per lane (padding and exhausted cells count as 0)
                    uint32_t m1 = best < 0.f ? 0u : __float_as_uint(best), m2 = sec < 0.f ? 0u : __float_as_uint(sec);
                    const uint32_t mybest = m1;
#define FPS_TOP2_STEP(CTRL, RM)                                                                     \
                    {                                                                               \
                        const uint32_t o1 = dclr_dpp<CTRL, RM>(0u, m1), o2 = dclr_dpp<CTRL, RM>(0u, m2); \
                        const uint32_t lo = dclr_umin(m1, o1);                                      \
                        m1 = dclr_umax(m1, o1);                                                     \
                        m2 = dclr_umax(dclr_umax(m2, o2), lo);                                      \
                    }
                    FPS_TOP2_STEP(DCLR_DPP_ROW_SHR(1), 0xf)
                    FPS_TOP2_STEP(DCLR_DPP_ROW_SHR(2), 0xf)
                    FPS_TOP2_STEP(DCLR_DPP_ROW_SHR(4), 0xf)
                    FPS_TOP2_STEP(DCLR_DPP_ROW_SHR(8), 0xf)
                    FPS_TOP2_STEP(DCLR_DPP_ROW_BCAST15, 0xa)
                    FPS_TOP2_STEP(DCLR_DPP_ROW_BCAST31, 0xc)
#undef FPS_TOP2_STEP
                    const uint32_t gm1 = (uint32_t)__builtin_amdgcn_readlane((int)m1, 63);
                    const uint32_t gm2 = (uint32_t)__builtin_amdgcn_readlane((int)m2, 63);
                    const bool real = best >= 0.f;
                    const uint64_t hit = __ballot(real && mybest == gm1);
                    int wl, wjj;
                    uint32_t wkey;
                    if ((hit & (hit - 1)) == 0) {                         // one lane holds the maximum (the usual case)
                        wl = hit != 0 ? __builtin_ctzll(hit) : 0;
                        wjj = __builtin_amdgcn_readlane(bjj, wl);
                        wkey = sbuf[slot_pos_g(g, wjj, wl)];
                    } else {                                              // exact tie: smallest tie key among the holders
                        uint32_t key = 0xFFFFFFFFu;
                        if (real && mybest == gm1) key = sbuf[slot_pos_g(g, bjj, lane)];
                        wkey = dclr_wave_min_u32(key);
                        wl = __builtin_ctzll(__ballot(key == wkey));
                        wjj = __builtin_amdgcn_readlane(bjj, wl);
                    }
                    const float wx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bx), wl));
                    const float wy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(by), wl));
                    const float wz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bz), wl));
                    cur_wl[g] = wl; cur_jj[g] = wjj; cur_b[g] = gm1;
                    if (lane == 0) {
                        gtab[deal(g)][0] = make_uint4(gm1, wkey, gm2, fps_tk1024_inv(wkey));
                        gtab[deal(g)][1] = make_uint4(__float_as_uint(wx), __float_as_uint(wy), __float_as_uint(wz), 0u);
                    }
                }
            }
#ifdef FPS_DEBUG
            FPS_STAMP(q1);
#endif
            __syncthreads();
#ifdef FPS_DEBUG
            FPS_STAMP(q2);
#endif
            // -- leader: the next samples from the table, and the groups each of them can change ----------------------
            if (wave == 0) {
                const int le = lane < NG ? lane : 0;
                const uint4 e0 = gtab[le][0], e1 = gtab[le][1];
                const float4 blo = *reinterpret_cast<const float4 *>(&gbox[le][0]);      // min x y z, max x
                const float2 bhi = *reinterpret_cast<const float2 *>(&gbox[le][4]);      // max y z
                const uint32_t val = lane < NG ? e0.x : 0u;               // b_g (the box test below needs it unmasked)
                uint32_t v = val;
                const uint32_t tk = lane < NG ? e0.y : 0xFFFFFFFFu;
                int wid[J];
                uint32_t mv[J];
                unsigned long long reach[J];                              // groups candidate j can change (lane = group)
                bool tie = false;
                auto box_test = [&](int w) -> unsigned long long {
                    const float sx = __uint_as_float(__builtin_amdgcn_readlane((int)e1.x, w));
                    const float sy = __uint_as_float(__builtin_amdgcn_readlane((int)e1.y, w));
                    const float sz = __uint_as_float(__builtin_amdgcn_readlane((int)e1.z, w));
                    const float lbv = fps_box_lower_bound(blo.x, blo.y, blo.z, blo.w, bhi.x, bhi.y, sx, sy, sz);
                    // (its own group is always marked: with b_g = 0, an exhausted cloud, the box test marks nothing, and the
                    // workers count the accepted samples by their non-empty masks)
                    return __ballot(lane < NG && lbv < __uint_as_float(val)) | (1ull << w);
                };
#pragma unroll
                for (int j = 0; j < J; ++j) {                             // J chained wave maxima: candidates in value order
                    mv[j] = dclr_wave_max_u32(v);
                    const uint64_t holders = __ballot(v == mv[j]);
                    tie = tie || (holders & (holders - 1)) != 0;
                    wid[j] = __builtin_ctzll(holders);
                    v = lane == wid[j] ? 0u : v;
                    // the box test of candidate j sits here so that its arithmetic fills the wait states of the next
                    // reduction's cross-lane steps (one straight run of instructions; a rejected candidate's mask is dropped)
                    reach[j] = box_test(wid[j]);
                }
#ifdef FPS_DEBUG
                unsigned long long l1_; FPS_STAMP(l1_); ml1 += l1_ - q2;
#endif
                if (tie) {
                    // two entries share a value (duplicate points, lattices, an exhausted cloud): again, by (value, key).
                    // Kept out of the loop above so that the usual round is one straight run of instructions in which the
                    // box tests and crossbar reads below fill the wait states of the reductions.
                    v = val;
#pragma unroll 1
                    for (int j = 0; j < J; ++j) {
                        const uint32_t mx = dclr_wave_max_u32(v);
                        const uint32_t kmin = dclr_wave_min_u32(v == mx ? tk : 0xFFFFFFFFu);
                        const int w = __builtin_ctzll(__ballot(v == mx && tk == kmin));
                        const unsigned long long rw = box_test(w);
#pragma unroll
                        for (int u = 0; u < J; ++u) { if (u == j) { wid[u] = w; mv[u] = mx; reach[u] = rw; } }
                        v = lane == w ? 0u : v;
                    }
                }
                // The tests of candidate j against the earlier ones, one (j, i) pair per lane: lane 4 j + i fetches both
                // entries through the LDS crossbar (ds_bpermute) and evaluates "v_j > u_i and sqdist(c_j, c_i) >= v_j" --
                // one distance computation for all pairs instead of one per pair on wave-uniform operands.
                const int lj = lane >> 3, li = lane & 7;                   // lane 8 j + i: candidate j against candidate i
                int src_j = wid[0], src_i = wid[0];
#pragma unroll
                for (int u = 1; u < J; ++u) { src_j = lj == u ? wid[u] : src_j; src_i = li == u ? wid[u] : src_i; }
                const float xj = __shfl(__uint_as_float(e1.x), src_j), yj = __shfl(__uint_as_float(e1.y), src_j),
                            zj = __shfl(__uint_as_float(e1.z), src_j);
                const float xi = __shfl(__uint_as_float(e1.x), src_i), yi = __shfl(__uint_as_float(e1.y), src_i),
                            zi = __shfl(__uint_as_float(e1.z), src_i);
                const uint32_t vj = (uint32_t)__shfl((int)val, src_j), ui = (uint32_t)__shfl((int)e0.z, src_i);
                const int kj = __shfl((int)e0.w, src_j);
                const uint32_t dji = __float_as_uint(dclr_sqdist(xj, yj, zj, xi, yi, zi));
                const bool pair_bad = lj < J && li < lj && !(vj > ui && dji >= vj);
                const unsigned long long bad = __ballot(pair_bad);         // bits 8 j .. 8 j + 7: candidate j fails a test
#ifdef FPS_DEBUG
                unsigned long long l2_; FPS_STAMP(l2_); ml2 += l2_ - l1_;
#endif
                // candidate j joins iff every earlier one did; the level-1 contract (temp = minima over the first m - 1
                // samples) gives the final sample a round of its own
                int cnt = 1;
                bool open = true;
#pragma unroll
                for (int j = 1; j < J; ++j) {
                    const bool ok = r + j < m && !(temp != nullptr && r + j == m - 1) && mv[j] != 0u && ((bad >> (8 * j)) & 0xFFull) == 0ull;
                    open = open && ok;
                    cnt += open ? 1 : 0;
                }
                unsigned long long am[J];
#pragma unroll
                for (int j = 0; j < J; ++j) am[j] = j < cnt ? reach[j] : 0ull;
#ifdef FPS_DEBUG_MARKS                        // (a global read-modify-write inside the leader: off when the leader is being timed)
                if (blockIdx.x == 0) {
                    unsigned long long all = 0;
#pragma unroll
                    for (int u = 0; u < J; ++u) all |= am[u];
                    if ((all >> lane) & 1ull) fps_grp[lane] += 1;
                }
#endif
                if (lj < J && li == 0) {                                  // lanes 0, 8, 16, ... publish candidates 0, 1, 2, ...
                    unsigned long long om = am[0];
#pragma unroll
                    for (int u = 1; u < J; ++u) om = lj == u ? am[u] : om;
                    rb_s[lj] = make_float4(xj, yj, zj, 0.f);
                    rb_m[lj] = om;
                    if (lj < cnt) picked[r + lj] = kj;
                }
#ifdef FPS_DEBUG
                unsigned long long l3_; FPS_STAMP(l3_); ml3 += l3_ - l2_;
#endif
            }
            __syncthreads();
            sr += 1;
#ifdef FPS_DEBUG
            FPS_STAMP(q3);
            mu += q1 - q0; mb += q2 - q1; mc += q3 - q2;
            if (lane == 0 && blockIdx.x == 0) {
                const int nm = (int)(mg - mg_prev) > 4 ? 4 : (int)(mg - mg_prev);
                fps_bucket[wave][nm][0] += 1; fps_bucket[wave][nm][1] += q1 - q0; fps_bucket[wave][nm][2] += mr - mr_prev;
            }
            mg_prev = mg; mr_prev = mr;
#endif
        }
        if (group_box && t == 0) group_box[6] = (float)sr;                // diagnostics: barrier rounds this cloud took
#ifdef FPS_DEBUG
        if (lane == 0 && blockIdx.x == 0) {
            fps_dbg[11] = (unsigned long long)sr;
            if (wave == 3) { fps_dbg[12] = mu; fps_dbg[13] = mg; fps_dbg[14] = mb; fps_dbg[15] = mc; fps_dbg[10] = mt; fps_dbg[7] = mr; }
            if (wave == 0) { fps_dbg[1] = mu; fps_dbg[2] = mb; fps_dbg[3] = mc; fps_dbg[5] = mt; fps_dbg[6] = mg; fps_dbg[8] = mr;
                             fps_dbg[0] = ml1; fps_dbg[4] = ml2; fps_dbg[9] = ml3; }   // leader: table + J maxima + box tests | pair tests | publish
        }
#endif
    } else
    if constexpr (MODE == 1) {
        // ---- several samples per barrier round -----------------------------------------------------------
        // Sample r+1 is the point with the largest running minimum AFTER sample r has been applied. Let every
        // wave w publish its best point c_w (value b_w, tie key) and its runner-up value u_w = the largest
        // running minimum among its other points. With p1 the best candidate overall, the best candidate p2 of
        // the OTHER waves is the next sample as well, provided (i) sqdist(p2, p1) >= td[p2] (its value does not
        // change when p1 is applied) and (ii) td[p2] > u_w1 (nothing left in p1's wave can reach it; running
        // minima only decrease). Every other point is already ordered behind p2: in p2's wave and in the
        // remaining waves by the per-wave arg-max (ties by key), in p1's wave by (ii). The same argument admits
        // p3 after p1, p2, and so on. All waves evaluate the test on the same 16 published entries, so they agree
        // without another exchange; then each applies the accepted samples one after the other. The exchange
        // (select, publish, barrier, combine) -- two thirds of a round -- is paid once per batch of samples.
        constexpr int J = 3;
        if (t < 32) { wpk[t >> 4][t & 15] = (unsigned long long)(t & 15); wru[t >> 4][t & 15] = 0u; }
        __syncthreads();
        float pcx[J] = {cx}, pcy[J] = {cy}, pcz[J] = {cz};
        int np = 1;
        uint32_t c_ru = 0u;
        int sr = 0;
#ifdef FPS_DEBUG
        unsigned long long mu = 0, ms = 0, mb = 0, mc = 0, mt = 0;
#endif
        for (int r = 1; r < m;) {
#ifdef FPS_DEBUG
            unsigned long long q0, q1, q2, q3, q4;
            FPS_STAMP(q0);
#endif
            uint32_t touched = 0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (j >= np) break;                                       // wave-uniform
                const float sx = pcx[j], sy = pcy[j], sz = pcz[j];
                const float lbv = fps_box_lower_bound(glo[0], glo[1], glo[2], ghi[0], ghi[1], ghi[2], sx, sy, sz);
                const uint32_t act = (uint32_t)__ballot(lbv < gmaxv);
                if (act == 0) continue;                                   // wave-uniform
                touched |= act;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (act & (1u << g)) {
                        float best = -1.0f;
                        int bjj = g * S;
                        if constexpr (S % 2 == 0) {
                            typedef float f2 __attribute__((ext_vector_type(2)));
                            const f2 c2x = {sx, sx}, c2y = {sy, sy}, c2z = {sz, sz};
#pragma unroll
                            for (int i = 0; i < S; i += 2) {
                                const int jj = g * S + i;
                                const f2 ax = {vec_get<P>(px, jj), vec_get<P>(px, jj + 1)};
                                const f2 ay = {vec_get<P>(py, jj), vec_get<P>(py, jj + 1)};
                                const f2 az = {vec_get<P>(pz, jj), vec_get<P>(pz, jj + 1)};
                                const f2 dx = ax - c2x, dy = ay - c2y, dz = az - c2z;
                                const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                                const f2 d = (xx + yy) + zz;
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    float d2;
                                    asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d[h]), "v"(vec_get<P>(td, jj + h)));
                                    vec_set<P>(td, jj + h, d2);
                                    const bool gt = d2 > best;
                                    bjj = gt ? jj + h : bjj;
                                    best = gt ? d2 : best;
                                }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < S; ++i) {
                                const int jj = g * S + i;
                                const float d = dclr_sqdist(vec_get<P>(px, jj), vec_get<P>(py, jj), vec_get<P>(pz, jj), sx, sy, sz);
                                float d2;
                                asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(vec_get<P>(td, jj)));
                                vec_set<P>(td, jj, d2);
                                const bool gt = d2 > best;
                                bjj = gt ? jj : bjj;
                                best = gt ? d2 : best;
                            }
                        }
                        gbest[g] = best; gjj[g] = bjj;
                    }
                }
            }
#ifdef FPS_DEBUG
            FPS_STAMP(q1);
#endif
            if (touched != 0) {
                if ((sr & 7) == 1) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float gm = __uint_as_float(dclr_wave_max_u32(gbest[g] < 0.f ? 0u : __float_as_uint(gbest[g])));
                        gmaxv = lane == g ? gm : gmaxv;
                    }
                }
                float lbest = gbest[0];
#pragma unroll
                for (int g = 1; g < G; ++g) lbest = fmaxf(lbest, gbest[g]);
                const uint32_t wmax = dclr_wave_max_u32(lbest < 0.f ? 0u : __float_as_uint(lbest));
                const float wmaxf = __uint_as_float(wmax);
                int hits = 0, hjj = 0;
#pragma unroll
                for (int g = G - 1; g >= 0; --g) {
                    const bool eq = gbest[g] == wmaxf;
                    hits += eq ? 1 : 0;
                    hjj = eq ? gjj[g] : hjj;
                }
                const uint64_t lanes_hit = __ballot(hits > 0);
                int wl, wjj;
                uint32_t wkey;
                if (__builtin_popcountll(lanes_hit) == 1 && __ballot(hits > 1) == 0) {
                    wl = __builtin_ctzll(lanes_hit);
                    wjj = __builtin_amdgcn_readlane(hjj, wl);
                    wkey = sbuf[slot_pos(wjj, wl)];
                } else {
                    uint32_t key = 0xFFFFFFFFu;
                    int kjj = 0;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        uint32_t kg = 0xFFFFu;
                        if (gbest[g] == wmaxf) kg = sbuf[slot_pos(gjj[g], lane)];
                        const bool take = gbest[g] == wmaxf && kg < key;
                        key = take ? kg : key;
                        kjj = take ? gjj[g] : kjj;
                    }
                    wkey = dclr_wave_min_u32(key);
                    wl = __builtin_ctzll(__ballot(key == wkey));
                    wjj = __builtin_amdgcn_readlane(kjj, wl);
                }
                c_packed = ((unsigned long long)wmax << 32) | ((unsigned long long)(0xFFFFu - wkey) << 16) |
                           (unsigned long long)wave;
                // the winning lane stores the payload itself, for both parities: wave 0 finished reading the other
                // parity's entries before the second barrier of the previous round
                if (lane == wl) {
                    const FpsCand c{(int32_t)fps_tk1024_inv(wkey), vec_get<P>(px, wjj), vec_get<P>(py, wjj), vec_get<P>(pz, wjj)};
                    cand[0][wave] = c;
                    cand[1][wave] = c;
                }
                // runner-up of the wave: every other lane's best; in the winner's lane the other groups' bests
                // and the other slots of the winner's (lane, group) cell
                const int wcell = wjj / S;                            // wave-uniform
                float other = -1.0f;
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const int jj = wcell * S + i;                     // wave-uniform register index
                    const float v = vec_get<P>(td, jj);
                    other = jj != wjj ? fmaxf(other, v) : other;
                }
#pragma unroll
                for (int g = 0; g < G; ++g) other = g != wcell ? fmaxf(other, gbest[g]) : other;
                const float alt = lane == wl ? other : lbest;
                c_ru = dclr_wave_max_u32(alt < 0.f ? 0u : __float_as_uint(alt));
            }
#ifdef FPS_DEBUG
            FPS_STAMP(q2);
#endif
            const int par = sr & 1;
            if (lane == 0) {
                wpk[par][wave] = c_packed;
                wru[par][wave] = c_ru;
            }
            __syncthreads();
#ifdef FPS_DEBUG
            FPS_STAMP(q3);
#endif
            // wave 0 alone reads the 16 published entries (lane & 15 = wave) and derives the list of samples; the
            // others wait at a second barrier and read the list (all 16 waves evaluating it redundantly cost
            // more: four waves per SIMD competing for the same issue slots, ~1300 cycles per round)
            if (wave == 0) {
                const unsigned long long e = wpk[par][lane & 15];
                const uint32_t e_ru = wru[par][lane & 15];
                const FpsCand w = cand[par][lane & 15];
                uint32_t e_hi = (uint32_t)(e >> 32), e_lo = (uint32_t)e;
                uint32_t ru_acc[J];
                float qx[J], qy[J], qz[J];
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    // level-1 contract: temp ends as the minima over the first m - 1 samples, so with temp the final
                    // sample gets a round of its own (the samples accepted last are never applied)
                    if (r + j >= m || (temp != nullptr && j > 0 && r + j == m - 1)) break;   // uniform
                    const uint32_t m_hi = dclr_row16_max_u32(e_hi);
                    // the wave holding it: unique unless two waves tie on the value (then the key field decides)
                    const uint32_t holders = (uint32_t)__ballot(e_hi == m_hi) & 0xFFFFu;
                    int wid;
                    if ((holders & (holders - 1)) == 0) wid = __builtin_ctz(holders);
                    else wid = (int)(dclr_row16_max_u32(e_hi == m_hi ? e_lo : 0u) & 15u);
                    const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.x), wid));
                    const float y = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.y), wid));
                    const float z = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.z), wid));
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < j; ++i) {
                        const uint32_t dist = __float_as_uint(dclr_sqdist(x, y, z, qx[i], qy[i], qz[i]));
                        ok = ok && m_hi > ru_acc[i] && dist >= m_hi;
                    }
                    if (!ok) break;
                    qx[j] = x; qy[j] = y; qz[j] = z;
                    ru_acc[j] = (uint32_t)__builtin_amdgcn_readlane((int)e_ru, wid);
                    if (lane == 0) {
                        picked[r + j] = __builtin_amdgcn_readlane(w.k, wid);
                        plist[j][0] = x; plist[j][1] = y; plist[j][2] = z;
                    }
                    cnt = j + 1;
                    const bool mine = (lane & 15) == wid;
                    e_hi = mine ? 0u : e_hi;
                    e_lo = mine ? 0u : e_lo;
                }
                if (lane == 0) plist_n = cnt;
            }
            __syncthreads();
            np = plist_n;
#pragma unroll
            for (int j = 0; j < J; ++j) { pcx[j] = plist[j][0]; pcy[j] = plist[j][1]; pcz[j] = plist[j][2]; }
            r += np;
            sr += 1;
#ifdef FPS_DEBUG
            FPS_STAMP(q4);
            mu += q1 - q0; ms += q2 - q1; mb += q3 - q2; mc += q4 - q3; mt += touched != 0 ? 1 : 0;
#endif
        }
        if (group_box && t == 0) group_box[6] = (float)sr;                // diagnostics: barrier rounds this cloud took
#ifdef FPS_DEBUG
        if (lane == 0 && blockIdx.x == 0) { fps_dbg[11] = (unsigned long long)sr; if (wave == 3) { fps_dbg[12] = mu; fps_dbg[13] = ms; fps_dbg[14] = mb; fps_dbg[15] = mc; fps_dbg[10] = mt; } }
#endif
    } else {
    int c3 = 1;                                                   // r % 3
    for (int r = 1; r < m; ++r) {
#ifdef FPS_DEBUG
        unsigned long long s0, s1, s2, s3;
        FPS_STAMP(s0);
#endif
        // lanes 0..G-1 test the G group boxes at once: group g needs work iff the rounded lower bound
        // of its distance to the new sample is below its largest running minimum
        const float lbv = fps_box_lower_bound(glo[0], glo[1], glo[2], ghi[0], ghi[1], ghi[2], cx, cy, cz);
        const uint32_t act = (uint32_t)__ballot(lbv < gmaxv);
#ifdef FPS_DEBUG
        unsigned long long a0 = 0, a1 = 0, a2 = 0;
        FPS_STAMP(a0);
#endif
        if (act != 0) {                                           // wave-uniform
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (act & (1u << g)) {
#ifdef FPS_DEBUG
                    n_act += 1;
#endif
                    float best = -1.0f;
                    int bjj = g * S;
                    if constexpr (S % 2 == 0) {
                        // two slots per instruction (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations in the
                        // same order as dclr_sqdist, so the distances are bit-identical)
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const f2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
                        for (int i = 0; i < S; i += 2) {
                            const int jj = g * S + i;
                            const f2 ax = {vec_get<P>(px, jj), vec_get<P>(px, jj + 1)};
                            const f2 ay = {vec_get<P>(py, jj), vec_get<P>(py, jj + 1)};
                            const f2 az = {vec_get<P>(pz, jj), vec_get<P>(pz, jj + 1)};
                            const f2 dx = ax - c2x, dy = ay - c2y, dz = az - c2z;
                            const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                            const f2 d = (xx + yy) + zz;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                float d2;                   // plain v_min_f32: no canonicalising v_max in front of it
                                asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d[h]), "v"(vec_get<P>(td, jj + h)));
                                vec_set<P>(td, jj + h, d2);
                                const bool gt = d2 > best;
                                bjj = gt ? jj + h : bjj;
                                best = gt ? d2 : best;
                            }
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < S; ++i) {
                            const int jj = g * S + i;
                            const float d = dclr_sqdist(vec_get<P>(px, jj), vec_get<P>(py, jj), vec_get<P>(pz, jj), cx, cy, cz);
                            float d2;                       // plain v_min_f32: no canonicalising v_max in front of it
                            asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(vec_get<P>(td, jj)));
                            vec_set<P>(td, jj, d2);
                            const bool gt = d2 > best;
                            bjj = gt ? jj : bjj;
                            best = gt ? d2 : best;
                        }
                    }
                    gbest[g] = best; gjj[g] = bjj;
                }
            }
            // Running minima only decrease, so a stale group maximum stays a valid (conservative) bound
            // for the test above; the bounds are tightened every 8th round instead of on every update.
            if ((r & 7) == 1) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float gm = __uint_as_float(dclr_wave_max_u32(gbest[g] < 0.f ? 0u : __float_as_uint(gbest[g])));
                    gmaxv = lane == g ? gm : gmaxv;
                }
            }
#ifdef FPS_DEBUG
            FPS_STAMP(a1);
#endif
            float lbest = gbest[0];
#pragma unroll
            for (int g = 1; g < G; ++g) lbest = fmaxf(lbest, gbest[g]);
            const uint32_t wmax = dclr_wave_max_u32(lbest < 0.f ? 0u : __float_as_uint(lbest));
            const float wmaxf = __uint_as_float(wmax);
            // how many (lane, group) candidates carry the wave maximum? exactly one unless distances tie
            int hits = 0, hjj = 0;
#pragma unroll
            for (int g = G - 1; g >= 0; --g) {
                const bool eq = gbest[g] == wmaxf;
                hits += eq ? 1 : 0;
                hjj = eq ? gjj[g] : hjj;
            }
            const uint64_t lanes_hit = __ballot(hits > 0);
            int wl, wjj;
            uint32_t wkey;
            if (__builtin_popcountll(lanes_hit) == 1 && __ballot(hits > 1) == 0) {
                wl = __builtin_ctzll(lanes_hit);
                wjj = __builtin_amdgcn_readlane(hjj, wl);
                wkey = sbuf[slot_pos(wjj, wl)];                                           // uniform address
            } else {
                // exact tie (duplicate points, lattices, exhausted cloud): smallest tie key among all
                // candidates. A group's cached slot already is its lowest-key maximum (ascending keys,
                // strict ">"), so only groups and lanes have to be merged here.
                uint32_t key = 0xFFFFFFFFu;
                int kjj = 0;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    uint32_t kg = 0xFFFFu;
                    if (gbest[g] == wmaxf) kg = sbuf[slot_pos(gjj[g], lane)];
                    const bool take = gbest[g] == wmaxf && kg < key;
                    key = take ? kg : key;
                    kjj = take ? gjj[g] : kjj;
                }
                wkey = dclr_wave_min_u32(key);
                wl = __builtin_ctzll(__ballot(key == wkey));
                wjj = __builtin_amdgcn_readlane(kjj, wl);
            }
            c_packed = ((unsigned long long)wmax << 32) | ((unsigned long long)(0xFFFFu - wkey) << 16) |
                       (unsigned long long)wave;
            c_k = (int32_t)fps_tk1024_inv(wkey);
            c_x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vec_get<P>(px, wjj)), wl));
            c_y = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vec_get<P>(py, wjj)), wl));
            c_z = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vec_get<P>(pz, wjj)), wl));
            fresh = 2;
#ifdef FPS_DEBUG
            FPS_STAMP(a2);
            if (wave == 0 && blockIdx.x == 0) { dbg_box += a0 - s0; dbg_upd += a1 - a0; dbg_sel += a2 - a1; dbg_nact += 1; }
#endif
        }
        const int par = r & 1;
        if (lane == 0) {
            if (fresh > 0) cand[par][wave] = FpsCand{c_k, c_x, c_y, c_z};
            atomicMax(&cell[c3], c_packed);
            if (wave == 0) cell[c3 == 2 ? 0 : c3 + 1] = 0ull;      // next round's cell; its readers passed the last barrier
        }
        fresh = fresh > 0 ? fresh - 1 : 0;
#ifdef FPS_DEBUG
        FPS_STAMP(s1);
        if (act != 0) dbg_pub += s1 - a2; else dbg_idle += s1 - s0;
#endif
        __syncthreads();
#ifdef FPS_DEBUG
        FPS_STAMP(s2);
#endif
        // both LDS reads are issued together: every lane fetches one wave's payload, the cell picks the lane
        const unsigned long long top = cell[c3];
        const FpsCand w = cand[par][lane & 15];
        const int wid = (int)__builtin_amdgcn_readfirstlane((uint32_t)top) & 15;
        const int32_t wk = __builtin_amdgcn_readlane(w.k, wid);
        cx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.x), wid));
        cy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.y), wid));
        cz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.z), wid));
        if (t == 0) picked[r] = wk;
        c3 = c3 == 2 ? 0 : c3 + 1;
#ifdef FPS_DEBUG
        FPS_STAMP(s3);
        acc_upd += s1 - s0; acc_bar += s2 - s1; acc_comb += s3 - s2;
#endif
    }
#ifdef FPS_DEBUG
    if (lane == 0) atomicAdd(&fps_dbg[0], n_act);
    if (t == 0 && blockIdx.x == 0) {
        unsigned long long t_end;
        FPS_STAMP(t_end);
        fps_dbg[1] = acc_upd; fps_dbg[2] = acc_bar; fps_dbg[3] = acc_comb; fps_dbg[4] = t_end - t_setup;
        fps_dbg[5] = dbg_box; fps_dbg[6] = dbg_upd; fps_dbg[7] = dbg_sel; fps_dbg[8] = dbg_pub; fps_dbg[9] = dbg_idle;
        fps_dbg[10] = dbg_nact;
    }
#endif

    }   // single-sample rounds

    __syncthreads();
    for (int i = t; i < m; i += WGS) idx[i] = picked[i];
    if (temp) {
#pragma unroll
        for (int jj = 0; jj < P; ++jj)
            if (sbuf[slot_pos(jj, lane)] != 0xFFFFu) temp[fps_tk1024_inv(sbuf[slot_pos(jj, lane)])] = vec_get<P>(td, jj);
    }
}

// Leader step of a several-samples round (see fps_pruned_kernel, MODE 1): one wave reads the 16 published per-wave
// entries (lane & 15 = wave) and accepts the best candidate p1, then the best of the other waves p2 while
// sqdist(p2, accepted) >= td[p2] and td[p2] exceeds the runner-up of every accepted sample's wave, and so on (<= J).
// Writes picked[r..], plist[j] = coordinates, *plist_n = count.
template <int J>
__device__ __forceinline__ void fps_accept_samples(int par, int r, int m, bool last_alone, int lane,
                                                   unsigned long long (*wpk)[16], uint32_t (*wru)[16], FpsCand (*cand)[16],
                                                   int32_t *picked, float (*plist)[4], int *plist_n) {
    const unsigned long long e = wpk[par][lane & 15];
    const uint32_t e_ru = wru[par][lane & 15];
    const FpsCand w = cand[par][lane & 15];
    uint32_t e_hi = (uint32_t)(e >> 32), e_lo = (uint32_t)e;
    uint32_t ru_acc[J];
    float qx[J], qy[J], qz[J];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        if (r + j >= m || (last_alone && j > 0 && r + j == m - 1)) break;   // uniform; last_alone: see fps_pruned_kernel
        const uint32_t m_hi = dclr_row16_max_u32(e_hi);
        // the wave holding it: unique unless two waves tie on the value (then the key field decides)
        const uint32_t holders = (uint32_t)__ballot(e_hi == m_hi) & 0xFFFFu;
        int wid;
        if ((holders & (holders - 1)) == 0) wid = __builtin_ctz(holders);
        else wid = (int)(dclr_row16_max_u32(e_hi == m_hi ? e_lo : 0u) & 15u);
        const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.x), wid));
        const float y = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.y), wid));
        const float z = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.z), wid));
        bool ok = true;
#pragma unroll
        for (int i = 0; i < j; ++i) {
            const uint32_t dist = __float_as_uint(dclr_sqdist(x, y, z, qx[i], qy[i], qz[i]));
            ok = ok && m_hi > ru_acc[i] && dist >= m_hi;
        }
        if (!ok) break;
        qx[j] = x; qy[j] = y; qz[j] = z;
        ru_acc[j] = (uint32_t)__builtin_amdgcn_readlane((int)e_ru, wid);
        if (lane == 0) {
            picked[r + j] = __builtin_amdgcn_readlane(w.k, wid);
            plist[j][0] = x; plist[j][1] = y; plist[j][2] = z;
        }
        cnt = j + 1;
        const bool mine = (lane & 15) == wid;
        e_hi = mine ? 0u : e_hi;
        e_lo = mine ? 0u : e_lo;
    }
    if (lane == 0) *plist_n = cnt;
}

// ------------------------------------------------------------------------------------------------
// Kernel B: clouds too large for one CU's registers (16384 < n <= 65536). Same sampling rule, same
// spatial pruning as kernel A', but the sorted points and their running minima live in a global
// workspace (L2-resident, ~1.3 MB per cloud) and only the groups a round can change are touched:
// a wave owns NG groups of 256 points (64 lanes x 4 slots), lane g keeps group g's box and an upper
// bound of its largest running minimum, each lane keeps the largest running minimum of its 4 slots per
// group (registers) and which slot holds it (2 bits per group). A round then reads and writes ~10 % of
// the cloud instead of all of it (fps_stream_kernel: 1 MB per round through one CU's memory path).
// ------------------------------------------------------------------------------------------------
// The same leader step with the acceptance tests side by side (the form fps_pruned_kernel MODE 3 uses): the J best
// published candidates are extracted first in value order (ties by key), then lane 8 j + i tests candidate j against
// candidate i < j through the LDS crossbar -- one distance evaluation for all pairs instead of a serial
// extract -> test -> extract chain with wave-uniform operands. Candidate j joins iff every earlier one did and all its
// tests pass; same picks as fps_accept_samples.
template <int J>
__device__ __forceinline__ void fps_accept_samples_par(int par, int r, int m, bool last_alone, int lane,
                                                       unsigned long long (*wpk)[16], uint32_t (*wru)[16], FpsCand (*cand)[16],
                                                       int32_t *picked, float (*plist)[4], int *plist_n) {
    static_assert(J <= 8, "one (j, i) pair per lane");
    const unsigned long long e = wpk[par][lane & 15];
    const uint32_t e_ru = wru[par][lane & 15];
    const FpsCand w = cand[par][lane & 15];
    uint32_t e_hi = (uint32_t)(e >> 32), e_lo = (uint32_t)e;
    int wid[J];
    uint32_t mv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        mv[j] = dclr_row16_max_u32(e_hi);
        const uint32_t holders = (uint32_t)__ballot(e_hi == mv[j]) & 0xFFFFu;
        if ((holders & (holders - 1)) == 0) wid[j] = __builtin_ctz(holders);
        else wid[j] = (int)(dclr_row16_max_u32(e_hi == mv[j] ? e_lo : 0u) & 15u);
        const bool mine = (lane & 15) == wid[j];
        e_hi = mine ? 0u : e_hi;
        e_lo = mine ? 0u : e_lo;
    }
    const int lj = lane >> 3, li = lane & 7;                              // lane 8 j + i: candidate j against candidate i
    int src_j = wid[0], src_i = wid[0];
    uint32_t vj = mv[0];
#pragma unroll
    for (int u = 1; u < J; ++u) {
        src_j = lj == u ? wid[u] : src_j;
        src_i = li == u ? wid[u] : src_i;
        vj = lj == u ? mv[u] : vj;
    }
    const float xj = __shfl(w.x, src_j), yj = __shfl(w.y, src_j), zj = __shfl(w.z, src_j);
    const float xi = __shfl(w.x, src_i), yi = __shfl(w.y, src_i), zi = __shfl(w.z, src_i);
    const uint32_t ui = (uint32_t)__shfl((int)e_ru, src_i);
    const int kj = __shfl(w.k, src_j);
    const uint32_t dji = __float_as_uint(dclr_sqdist(xj, yj, zj, xi, yi, zi));
    const bool pair_bad = lj < J && li < lj && !(vj > ui && dji >= vj);
    const unsigned long long bad = __ballot(pair_bad);                    // bits 8 j .. 8 j + 7: candidate j fails a test
    int cnt = 1;
    bool open = true;
#pragma unroll
    for (int j = 1; j < J; ++j) {
        const bool ok = r + j < m && !(last_alone && r + j == m - 1) && ((bad >> (8 * j)) & 0xFFull) == 0ull;
        open = open && ok;
        cnt += open ? 1 : 0;
    }
    if (lj < J && li == 0 && lj < cnt) {                                  // lanes 0, 8, 16, ...: candidates 0, 1, 2, ...
        picked[r + lj] = kj;
        plist[lj][0] = xj; plist[lj][1] = yj; plist[lj][2] = zj;
    }
    if (lane == 0) *plist_n = cnt;
}

// Kernel B: the global group (256 consecutive sorted positions) that is wave w's g-th: neighbours in sorted order go to
// different waves. (Rotating the assignment from one coarse sorting cell to the next -- 16 g + (w - g) mod 16 -- moved the
// per-cloud imbalance to other waves and changed nothing: 1471 vs 1456 us per 16 clouds.)
__device__ __forceinline__ int fps_paged_group(int g, int wave) { return g * 16 + wave; }

#ifdef FPS_DEBUG
__device__ unsigned long long fps_dbg2[16][8];     // cloud 0, per wave: cycles in box tests, visits, selection, wait-A, leader+B; visits; batches; rounds
#endif
template <int NG, int MODE>                       // MODE as in fps_pruned_kernel: 0 one sample per barrier round, 1 several
__global__ __launch_bounds__(1024) void fps_paged_kernel(int n, int pstride, int m, const float *__restrict__ pts,
                                                         int32_t *__restrict__ idx, float4 *__restrict__ spts_all,
                                                         float *__restrict__ std_all, uint32_t *__restrict__ sidx_all,
                                                         uint16_t *__restrict__ cell_all, float *__restrict__ group_box,
                                                         float *__restrict__ temp, DclrCloudView view) {
    constexpr int WGS = 1024, NW = 16, P = 4 * NG, NP = WGS * P, BINS = 4096;
    typedef typename VecOf<NG>::type gvec;
    __shared__ unsigned long long cell[3];
    __shared__ FpsCand cand[2][16];
    __shared__ unsigned long long wpk[2][16];              // MODE 1: per-wave packed candidate, by round parity
    __shared__ uint32_t wru[2][16];                        // MODE 1: per-wave runner-up value
    __shared__ __attribute__((aligned(16))) float plist[4][4];   // MODE 1: the samples accepted for the next round
    __shared__ int plist_n;
    __shared__ float red[6][16];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t hist[BINS];
    extern __shared__ int32_t picked[];

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    pts += dclr_cloud_offset(view, blockIdx.x, (size_t)n * pstride);
    const bool vec4 = pstride == 4 && ((uintptr_t)pts & 15) == 0;          // wave-uniform
    idx += (size_t)blockIdx.x * m;
    float4 *spts = spts_all + (size_t)blockIdx.x * NP;
    float *std_ = std_all + (size_t)blockIdx.x * NP;
    uint32_t *sidx = sidx_all + (size_t)blockIdx.x * NP;
    uint16_t *cellof = cell_all + (size_t)blockIdx.x * NP;
    if (group_box) group_box += (size_t)blockIdx.x * NW * NG * 8;
    if (temp) temp += (size_t)blockIdx.x * n;

    // ---- 1. bounding box ------------------------------------------------------------------------
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = t; k < n; k += WGS) {
        float vx_, vy_, vz_;
        fps_load_xyz(pts, (size_t)k, pstride, vec4, vx_, vy_, vz_);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = a == 0 ? vx_ : a == 1 ? vy_ : vz_;
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = fps_shfl_min(lo[a]), h = fps_shfl_max(hi[a]);
        if (lane == 0) { red[a][wave] = l; red[3 + a][wave] = h; }
    }
    for (int u = t; u < BINS; u += WGS) hist[u] = 0u;
    if (t < 3) cell[t] = 0ull;
    if (t < 32) cand[t >> 4][t & 15] = FpsCand{0, 0.f, 0.f, 0.f};
    __syncthreads();
    float ext[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[a][0], h = red[3 + a][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        lo[a] = l;
        ext[a] = h - l;
    }
    // ---- 2. counting sort by a 12-bit cell, bits dealt to the axes by extent (any order yields the same samples) ----
    const FpsGrid grid = fps_make_grid(ext);
    __shared__ uint16_t cell_lut[3][256];
    const bool use_lut = fps_cell_lut<WGS>(grid, cell_lut, t);
    for (int k = t; k < n; k += WGS) {
        float px_, py_, pz_;
        fps_load_xyz(pts, (size_t)k, pstride, vec4, px_, py_, pz_);
        uint32_t q3[3];
        fps_cell_coords(grid, px_ - lo[0], py_ - lo[1], pz_ - lo[2], q3);
        const uint32_t mc = use_lut ? (uint32_t)cell_lut[0][q3[0]] | cell_lut[1][q3[1]] | cell_lut[2][q3[2]]
                                    : fps_cell_key(grid, q3[0], q3[1], q3[2]);
        atomicAdd(&hist[mc], 1u);
        cellof[k] = (uint16_t)mc;
    }
    __syncthreads();
    {
        constexpr int BPT = BINS / WGS;
        uint32_t c[BPT], mine = 0;
#pragma unroll
        for (int u = 0; u < BPT; ++u) { c[u] = hist[BPT * t + u]; mine += c[u]; }
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int u = 0; u < BPT; ++u) { hist[BPT * t + u] = run; run += c[u]; }
    }
    __syncthreads();
    for (int k = t; k < n; k += WGS) sidx[atomicAdd(&hist[cellof[k]], 1u)] = (uint32_t)k;   // own cellof entries
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- 3. groups: 256 consecutive sorted positions each; wave w owns groups w, 16 + w, 32 + w, ... (its
    //         group g = global group 16 g + w): a sample touches a few spatially adjacent groups and every touched
    //         group costs its wave a dependent round trip to the workspace, so neighbours go to different waves.
    //         Slot i of lane l = position base + 64 i + l, slots of a lane in ascending tie-key order ------
    float glo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ghi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};   // lane g: box of group g
    float gmaxv = 0.f;
    gvec gbest;                                            // per lane: largest running minimum of the 4 slots
    gvec gsec;                                             // MODE 1: the largest among the lane's other 3 slots
    uint32_t gslot = 0;                                    // 2 bits per group: the slot holding it
    // MODE 1 keeps the running minima on the CU: slots 0..2 of every group in registers (element g of a 16-vector,
    // indexed dynamically: register-relative moves), slot 3 in LDS (64 KB) -- 64 registers of minima beside a round's
    // working set do not fit 128. The workspace copy, its read and write-back per visited group (a third of the
    // kernel's HBM traffic) are gone; the workspace holds only the read-only sorted coordinates.
    constexpr bool REGTD = MODE == 1;
    typedef float fps_v16 __attribute__((ext_vector_type(16)));
    fps_v16 tdr[3];
    __shared__ float tdl3[REGTD ? 16 : 1][REGTD ? WGS : 1];
#define FPS_TD_GET(i_, g_) ((i_) == 0 ? tdr[0][g_] : (i_) == 1 ? tdr[1][g_] : (i_) == 2 ? tdr[2][g_] : tdl3[REGTD ? (g_) : 0][REGTD ? t : 0])
#define FPS_TD_SET(i_, g_, v_)                                                       \
    {                                                                               \
        if ((i_) == 0) tdr[0][g_] = (v_);                                           \
        else if ((i_) == 1) tdr[1][g_] = (v_);                                      \
        else if ((i_) == 2) tdr[2][g_] = (v_);                                      \
        else tdl3[REGTD ? (g_) : 0][REGTD ? t : 0] = (v_);                          \
    }
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const int base = fps_paged_group(g, wave) * 256 + lane;
        uint32_t tk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = base + 64 * i;
            tk[i] = pos < n ? fps_tk1024(sidx[pos]) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k2 = 2; k2 <= 4; k2 <<= 1)
#pragma unroll
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int l = i ^ j2;
                    if (l > i) {
                        const uint32_t a = tk[i], b = tk[l];
                        const uint32_t mn = a < b ? a : b, mxv = a < b ? b : a;
                        const bool asc = (i & k2) == 0;
                        tk[i] = asc ? mn : mxv;
                        tk[l] = asc ? mxv : mn;
                    }
                }
        float blo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        bool any = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // padding: far away (set abstraction reads these groups too: distance +inf, never inside a ball) and
            // running minimum -2, which no update raises and which never beats best = -1
            float x = 3.0e38f, y = 3.0e38f, z = 3.0e38f, d = -2.0f;
            uint32_t k = 0xFFFFFFFFu;
            if (tk[i] != 0xFFFFFFFFu) {
                k = fps_tk1024_inv(tk[i]);
                fps_load_xyz(pts, (size_t)k, pstride, vec4, x, y, z);
                d = temp ? temp[k] : 1e10f;
                blo[0] = fminf(blo[0], x); blo[1] = fminf(blo[1], y); blo[2] = fminf(blo[2], z);
                bhi[0] = fmaxf(bhi[0], x); bhi[1] = fmaxf(bhi[1], y); bhi[2] = fmaxf(bhi[2], z);
                any = true;
            }
            spts[base + 64 * i] = make_float4(x, y, z, __uint_as_float(k));
            if constexpr (REGTD) FPS_TD_SET(i, g, d)
            else std_[base + 64 * i] = d;
        }
        float box[6];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float l = fps_shfl_min(blo[a]), h = fps_shfl_max(bhi[a]);
            if ((lane & (NG - 1)) == g) { glo[a] = l; ghi[a] = h; }          // (replicated: lane j NG + g tests group g against sample j)
            box[a] = l; box[3 + a] = h;
        }
        if (group_box && lane < 8)
            group_box[(size_t)fps_paged_group(g, wave) * 8 + lane] =
                lane == 0 ? box[0] : lane == 1 ? box[1] : lane == 2 ? box[2] : lane == 3 ? box[3]
                : lane == 4 ? box[4] : lane == 5 ? box[5] : 0.f;
        vec_set<NG>(gbest, g, any ? 0.f : -1.0f);
        if constexpr (MODE == 1) vec_set<NG>(gsec, g, -1.0f);
        const bool group_any = __ballot(any) != 0;                               // all lanes vote: NOT inside `lane == g &&`
        if ((lane & (NG - 1)) == g && group_any) gmaxv = __uint_as_float(0x7F800000u);   // +inf forces the first update
    }
    // every thread re-reads only what it wrote itself (same positions): no further fence needed

    if constexpr (MODE == 1) {
        // ---- several samples per barrier round: the acceptance scheme of fps_pruned_kernel MODE 1, on GROUP-level
        //      state. The workspace of a launch is larger than an XCD's L2, so every dependent access to it costs
        //      ~1 us: a round makes exactly one -- the touched groups are read two at a time, ALL accepted samples are
        //      applied (a sample whose box bound spares a group cannot change it: min is idempotent there), and while
        //      the points are in registers the wave reduces the group's best point, its tie key and its second-best
        //      value into lane g. Selecting the wave's candidate and runner-up is then a 16-lane reduction over
        //      registers, with no fetch of the winner. gmaxv is the group's exact largest running minimum here. -------
#ifndef FPS_PAGED_J
#define FPS_PAGED_J 4             // samples a barrier round may accept (3: 424 rounds, 4: 370; 1-2 % with the side-by-side leader)
#endif
#ifndef FPS_PAGED_B
#define FPS_PAGED_B 1          // groups in flight per wave: 2 would need 24 bytes of scratch beside the register-resident minima
#endif
        constexpr int J = FPS_PAGED_J, B = FPS_PAGED_B;
        if (t < 32) { wpk[t >> 4][t & 15] = (unsigned long long)(t & 15); wru[t >> 4][t & 15] = 0u; }
        __syncthreads();
        static_assert(J * NG <= 64, "one (sample, group) pair per lane in the box tests");
        float pcx[J] = {pts[0]}, pcy[J] = {pts[1]}, pcz[J] = {pts[2]};
        const int tj = lane / NG;                              // the sample this lane tests its group lane % NG against
        float tsx = pts[0], tsy = pts[1], tsz = pts[2];
        int np = 1;
        if (t == 0) picked[0] = 0;
        uint32_t gsv = 0u, gkey = 0xFFFFFFFFu;                 // lane g: second-best value bits, tie key of the best point
        float gx = 0.f, gy = 0.f, gz = 0.f;                    // lane g: the best point of group g
        int32_t gk = 0;
        uint32_t gstate = 0u;                                  // lane g: (lane << 3 | slot << 1) of the best point | entry valid
        unsigned long long c_packed = (unsigned long long)wave;
        uint32_t c_ru = 0u;
        int sr = 0;
#ifdef FPS_DEBUG
        unsigned long long a_box = 0, a_vis = 0, a_sel = 0, a_w1 = 0, a_lead = 0, a_batches = 0, a_touch = 0;
#endif
        for (int r = 1; r < m;) {
#ifdef FPS_DEBUG
            unsigned long long q0, q1, q2, q3, q4, q5;
            FPS_STAMP(q0);
#endif
            // All of the round's samples against all of this wave's groups in ONE pass: lane j NG + g tests group g against
            // sample j (boxes and maxima are replicated over the lane groups). One bound computation per round instead of
            // one per sample: the kernel is bound by instruction issue (16 waves on 4 SIMDs), and every wave runs this
            // every round.
            uint32_t act = 0, actj[J];
            {
                const float lbv = fps_box_lower_bound(glo[0], glo[1], glo[2], ghi[0], ghi[1], ghi[2], tsx, tsy, tsz);
                const unsigned long long hit = __ballot(tj < np && lbv < gmaxv);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    actj[j] = (uint32_t)(hit >> (j * NG)) & ((1u << NG) - 1u);
                    act |= actj[j];
                }
            }
#ifdef FPS_DEBUG
            FPS_STAMP(q1);
            q2 = q1;
            a_batches += (__builtin_popcount(act) + B - 1) / B; a_touch += __builtin_popcount(act);
#endif
            if (act != 0) {                                               // wave-uniform
                for (uint32_t rem = act; rem != 0;) {
                    int gs[B];
                    bool live[B];
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        live[u] = rem != 0;
                        gs[u] = live[u] ? __builtin_ctz(rem) : gs[0];    // absent: a clamped repeat, results unused
                        if (live[u]) rem &= rem - 1;
                    }
                    float4 q[B][4];
                    float o[B][4];
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        if (!live[u]) break;                              // wave-uniform
                        const int base = fps_paged_group(gs[u], wave) * 256 + lane;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { q[u][i] = spts[base + 64 * i]; o[u][i] = FPS_TD_GET(i, gs[u]); }
                    }
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        if (!live[u]) break;                              // wave-uniform
                        const int g = gs[u];
#pragma unroll
                        for (int j = 0; j < J; ++j) {
                            if (((actj[j] >> g) & 1u) == 0) continue;     // wave-uniform: sample j cannot change this group
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float d = dclr_sqdist(q[u][i].x, q[u][i].y, q[u][i].z, pcx[j], pcy[j], pcz[j]);
                                asm("v_min_f32 %0, %1, %2" : "=v"(o[u][i]) : "v"(d), "v"(o[u][i]));
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) FPS_TD_SET(i, g, o[u][i])
#ifndef FPS_PAGED_NOLAZY
                        // Did the group's best point keep its value? Running minima only decrease, so then the group's
                        // entry (b_g, best point, key) stands; its runner-up may now be too large, which only makes the
                        // leader's test (v > u) more cautious. 40 % of the visits end here (as in fps_pruned_kernel MODE 3).
                        {
                            const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)gstate, g);   // lane << 3 | slot << 1 | valid
                            const uint32_t old_b = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(gmaxv), g);
                            const int hs = (int)((st >> 1) & 3u);
                            const float hv = hs == 0 ? o[u][0] : hs == 1 ? o[u][1] : hs == 2 ? o[u][2] : o[u][3];   // uniform selects
                            const uint32_t now = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(hv), (int)(st >> 3));
                            if ((st & 1u) != 0 && now == old_b) continue;                              // wave-uniform
                        }
#endif
                        // this lane: best slot (slots ascend in tie-key order, strict > keeps the first) and the rest
                        float best = -1.0f, sec = -1.0f;
                        float4 bq = q[u][0];
                        int bslot = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool gt = o[u][i] > best;
                            bslot = gt ? i : bslot;
                            sec = fmaxf(sec, gt ? best : o[u][i]);
                            bq.x = gt ? q[u][i].x : bq.x; bq.y = gt ? q[u][i].y : bq.y;
                            bq.z = gt ? q[u][i].z : bq.z; bq.w = gt ? q[u][i].w : bq.w;
                            best = gt ? o[u][i] : best;
                        }
                        // the group: largest value, its holder (ties by key), the largest value of everything else
                        const uint32_t bbits = best < 0.f ? 0u : __float_as_uint(best);
                        const uint32_t gmx = dclr_wave_max_u32(bbits);
                        const bool holder = best >= 0.f && bbits == gmx;
                        const uint64_t hl = __ballot(holder);
                        int wl;
                        if ((hl & (hl - 1)) == 0) {
                            wl = __builtin_ctzll(hl);
                        } else {
                            const uint32_t key = holder ? fps_tk1024(__float_as_uint(bq.w)) : 0xFFFFFFFFu;
                            const uint32_t kmin = dclr_wave_min_u32(key);
                            wl = __builtin_ctzll(__ballot(key == kmin));
                        }
                        const float alt = lane == wl ? sec : best;
                        const uint32_t g2 = dclr_wave_max_u32(alt < 0.f ? 0u : __float_as_uint(alt));
                        const float wx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bq.x), wl));
                        const float wy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bq.y), wl));
                        const float wz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(bq.z), wl));
                        const int32_t wk = __builtin_amdgcn_readlane(__float_as_int(bq.w), wl);
                        const int wslot = __builtin_amdgcn_readlane(bslot, wl);
                        if ((lane & (NG - 1)) == g) gmaxv = __uint_as_float(gmx);          // every replica (the box tests read it)
                        if (lane == g) {
                            gsv = g2;
                            gkey = fps_tk1024((uint32_t)wk);
                            gx = wx; gy = wy; gz = wz; gk = wk;
                            gstate = ((uint32_t)wl << 3) | ((uint32_t)wslot << 1) | (hl != 0 ? 1u : 0u);
                        }
                    }
                }
#ifdef FPS_DEBUG
                FPS_STAMP(q2);
#endif
                // the wave's candidate: largest group maximum, ties by key; runner-up: the other groups' maxima and the
                // winner group's second-best
                const bool has = lane < NG && gkey != 0xFFFFFFFFu;
                const uint32_t vbits = has ? __float_as_uint(gmaxv) : 0u;
                const uint32_t wmax = dclr_row16_max_u32(vbits);
                const bool hold = has && vbits == wmax;
                const uint32_t kmin = dclr_row16_min_u32(hold ? gkey : 0xFFFFFFFFu);
                const int wgp = __builtin_ctzll(__ballot(hold && gkey == kmin));
                c_packed = ((unsigned long long)wmax << 32) | ((unsigned long long)(0xFFFFu - kmin) << 16) |
                           (unsigned long long)wave;
                // both parities: wave 0 finished reading the other parity's entries before the second barrier of the
                // previous round
                if (lane == wgp) {
                    const FpsCand c{gk, gx, gy, gz};
                    cand[0][wave] = c;
                    cand[1][wave] = c;
                }
                c_ru = dclr_row16_max_u32(has ? (lane == wgp ? gsv : vbits) : 0u);
            }
            const int par = sr & 1;
            if (lane == 0) {
                wpk[par][wave] = c_packed;
                wru[par][wave] = c_ru;
            }
#ifdef FPS_DEBUG
            FPS_STAMP(q3);
#endif
            __syncthreads();
#ifdef FPS_DEBUG
            FPS_STAMP(q4);
#endif
#ifdef FPS_PAGED_SERIAL_LEADER
            if (wave == 0) fps_accept_samples<J>(par, r, m, temp != nullptr, lane, wpk, wru, cand, picked, plist, &plist_n);
#else
            if (wave == 0) fps_accept_samples_par<J>(par, r, m, temp != nullptr, lane, wpk, wru, cand, picked, plist, &plist_n);
#endif
            __syncthreads();
#ifdef FPS_DEBUG
            FPS_STAMP(q5);
            a_box += q1 - q0; a_vis += q2 - q1; a_sel += q3 - q2; a_w1 += q4 - q3; a_lead += q5 - q4;
#endif
            np = plist_n;
#pragma unroll
            for (int j = 0; j < J; ++j) { pcx[j] = plist[j][0]; pcy[j] = plist[j][1]; pcz[j] = plist[j][2]; }
            {
                const float4 ts = *reinterpret_cast<const float4 *>(plist[tj < J ? tj : 0]);
                tsx = ts.x; tsy = ts.y; tsz = ts.z;
            }
            r += np;
            sr += 1;
        }
#ifdef FPS_DEBUG
        if (blockIdx.x == 0 && lane == 0) {
            fps_dbg2[wave][0] = a_box; fps_dbg2[wave][1] = a_vis; fps_dbg2[wave][2] = a_sel; fps_dbg2[wave][3] = a_w1;
            fps_dbg2[wave][4] = a_lead; fps_dbg2[wave][5] = a_touch; fps_dbg2[wave][6] = a_batches;
            fps_dbg2[wave][7] = (unsigned long long)sr;
        }
#endif
        if (group_box && t == 0) group_box[6] = (float)sr;                // diagnostics: barrier rounds this cloud took
    } else {
    float cx = pts[0], cy = pts[1], cz = pts[2];
    if (t == 0) picked[0] = 0;
    unsigned long long c_packed = (unsigned long long)wave;
    int32_t c_k = 0;
    float c_x = 0.f, c_y = 0.f, c_z = 0.f;
    int fresh = 0;
    int c3 = 1;
    for (int r = 1; r < m; ++r) {
        const float lbv = fps_box_lower_bound(glo[0], glo[1], glo[2], ghi[0], ghi[1], ghi[2], cx, cy, cz);
        uint32_t act = (uint32_t)__ballot(lane < NG && lbv < gmaxv);
        if (act != 0) {                                           // wave-uniform
            for (uint32_t rem = act; rem != 0; rem &= rem - 1) {
                const int g = __builtin_ctz(rem);
                const int base = fps_paged_group(g, wave) * 256 + lane;
                float4 q[4];
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { q[i] = spts[base + 64 * i]; o[i] = std_[base + 64 * i]; }
                float best = -1.0f;
                uint32_t bs = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = dclr_sqdist(q[i].x, q[i].y, q[i].z, cx, cy, cz);
                    float d2;
                    asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(o[i]));
                    std_[base + 64 * i] = d2;
                    const bool gt = d2 > best;
                    bs = gt ? (uint32_t)i : bs;
                    best = gt ? d2 : best;
                }
                vec_set<NG>(gbest, g, best);
                gslot = (gslot & ~(3u << (2 * g))) | (bs << (2 * g));
            }
            if ((r & 7) == 1) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float v = vec_get<NG>(gbest, g);
                    const float gm = __uint_as_float(dclr_wave_max_u32(v < 0.f ? 0u : __float_as_uint(v)));
                    gmaxv = lane == g ? gm : gmaxv;
                }
            }
            // this lane's candidate: first group (lowest g) holding its largest running minimum
            float lbest = vec_get<NG>(gbest, 0);
#pragma unroll
            for (int g = 1; g < NG; ++g) lbest = fmaxf(lbest, vec_get<NG>(gbest, g));
            int hits = 0, hg = 0;
#pragma unroll
            for (int g = NG - 1; g >= 0; --g) {
                const bool eq = vec_get<NG>(gbest, g) == lbest;
                hits += eq ? 1 : 0;
                hg = eq ? g : hg;
            }
            // fetch the candidate point (position, original index) while the wave reduction runs
            const float4 mine4 = spts[fps_paged_group(hg, wave) * 256 + 64 * (int)((gslot >> (2 * hg)) & 3u) + lane];
            const uint32_t wmax = dclr_wave_max_u32(lbest < 0.f ? 0u : __float_as_uint(lbest));
            const float wmaxf = __uint_as_float(wmax);
            const bool holder = lbest == wmaxf;
            const uint64_t lanes_hit = __ballot(holder);
            float4 win4;
            int wl;
            if (__builtin_popcountll(lanes_hit) == 1 && __ballot(holder && hits > 1) == 0) {
                wl = __builtin_ctzll(lanes_hit);
                win4 = mine4;
            } else {
                // exact tie: smallest tie key among all (lane, group) candidates carrying the maximum
                uint32_t key = 0xFFFFFFFFu;
                float4 k4 = mine4;
#pragma unroll 1
                for (int g = 0; g < NG; ++g) {
                    if (__ballot(vec_get<NG>(gbest, g) == wmaxf) == 0) continue;      // wave-uniform
                    float4 c4 = mine4;
                    uint32_t kg = 0xFFFFFFFFu;
                    if (vec_get<NG>(gbest, g) == wmaxf) {
                        c4 = spts[fps_paged_group(g, wave) * 256 + 64 * (int)((gslot >> (2 * g)) & 3u) + lane];
                        kg = fps_tk1024(__float_as_uint(c4.w));
                    }
                    const bool take = kg < key;
                    key = take ? kg : key;
                    k4 = take ? c4 : k4;
                }
                const uint32_t wk = dclr_wave_min_u32(key);
                wl = __builtin_ctzll(__ballot(key == wk));
                win4 = k4;
            }
            c_x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(win4.x), wl));
            c_y = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(win4.y), wl));
            c_z = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(win4.z), wl));
            c_k = __builtin_amdgcn_readlane(__float_as_int(win4.w), wl);
            const uint32_t wkey = fps_tk1024((uint32_t)c_k);
            c_packed = ((unsigned long long)wmax << 32) | ((unsigned long long)(0xFFFFu - wkey) << 16) |
                       (unsigned long long)wave;
            fresh = 2;
        }
        const int par = r & 1;
        if (lane == 0) {
            if (fresh > 0) cand[par][wave] = FpsCand{c_k, c_x, c_y, c_z};
            atomicMax(&cell[c3], c_packed);
            if (wave == 0) cell[c3 == 2 ? 0 : c3 + 1] = 0ull;
        }
        fresh = fresh > 0 ? fresh - 1 : 0;
        __syncthreads();
        const unsigned long long top = cell[c3];
        const FpsCand w = cand[par][lane & 15];
        const int wid = (int)__builtin_amdgcn_readfirstlane((uint32_t)top) & 15;
        const int32_t wk = __builtin_amdgcn_readlane(w.k, wid);
        cx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.x), wid));
        cy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.y), wid));
        cz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w.z), wid));
        if (t == 0) picked[r] = wk;
        c3 = c3 == 2 ? 0 : c3 + 1;
    }
    }   // single-sample rounds
    __syncthreads();
    for (int i = t; i < m; i += WGS) idx[i] = picked[i];
    if (temp) {                                                // level-1 contract: the running minima go back to temp
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int base = fps_paged_group(g, wave) * 256 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t k = __float_as_uint(spts[base + 64 * i].w);
                if (k != 0xFFFFFFFFu) temp[k] = REGTD ? FPS_TD_GET(i, g) : std_[base + 64 * i];
            }
        }
    }
#undef FPS_TD_GET
#undef FPS_TD_SET
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

template <int WGS, int P, int G>
void launch_pruned(int b, int n, int pstride, int m, const float *pts, float *temp, int32_t *idx, float4 *group_pts,
                   float *group_box, hipStream_t s, DclrCloudView view, float *slice_box) {
    constexpr int NP = WGS * P;
    const size_t tail = (size_t)NP * 2 > (size_t)m * 4 ? (size_t)NP * 2 : (size_t)m * 4;   // cell ids, then picked[]
    const size_t lds = (size_t)4096 * 4 + (size_t)NP * 2 + tail;
    // A/B switches: DCLR_FPS_SINGLE = one sample per barrier round, DCLR_FPS_WAVECAND = several with per-wave candidates
    // A/B switches (same samples): DCLR_FPS_SINGLE = one sample per barrier round, DCLR_FPS_WAVECAND = several with per-wave
    // candidates (round 2's kernel); default = several with the per-group table
    static const int mode = getenv("DCLR_FPS_SINGLE") ? 0 : getenv("DCLR_FPS_WAVECAND") ? 1 : 3;
    if (mode == 0)
        hipLaunchKernelGGL((fps_pruned_kernel<WGS, P, G, 0>), dim3(b), dim3(WGS), lds, s, n, pstride, m, pts, temp, idx,
                           group_pts, group_box, view, slice_box);
    else if (mode == 1)
        hipLaunchKernelGGL((fps_pruned_kernel<WGS, P, G, 1>), dim3(b), dim3(WGS), lds, s, n, pstride, m, pts, temp, idx,
                           group_pts, group_box, view, slice_box);
    else
        hipLaunchKernelGGL((fps_pruned_kernel<WGS, P, G, 3>), dim3(b), dim3(WGS), lds, s, n, pstride, m, pts, temp, idx,
                           group_pts, group_box, view, slice_box);
}

// Spatial groups the pruned kernel forms (and can export): NW waves x G groups of 64 * (P / G) points.
bool fps_group_layout(int n, int *n_groups, int *group_size) {
    if (n <= 1024 || n > 65536) return false;
    if (n > 16384) { *n_groups = n <= 32768 ? 128 : 256; *group_size = 256; return true; }   // workspace kernel: 16 waves x NG
    if (n <= 2048) { *n_groups = 32; *group_size = 64; return true; }      // 8 waves x 4 groups of one slot
    const int p = n <= 2048 ? 2 : n <= 4096 ? 4 : n <= 8192 ? 8 : 16;
    const int g = p >= 4 ? 4 : p;
    *n_groups = 16 * g;
    *group_size = 64 * (p / g);
    return true;
}

int fps_dispatch(int b, int n, int pstride, int m, const float *pts, float *temp, int32_t *idx,
                 hipStream_t s, float4 *group_pts = nullptr, float *group_box = nullptr,
                 DclrCloudView view = DclrCloudView{0, 1, 0}, float *slice_box = nullptr) {
    DCLR_REQUIRE(b > 0 && n > 0 && m > 0 && pstride >= 3 && pts && idx);
    if ((size_t)m * sizeof(int32_t) > 64 * 1024) return DCLR_E_UNSUPPORTED;   // picked[] lives in LDS
    static const bool plain = getenv("DCLR_FPS_PLAIN") != nullptr;             // A/B switch for measurements
    if (!plain && n > 1024 && n <= 16384) {
        // 2048 points: 8 waves x 4 points per lane and two clouds per CU (422 vs 445 us for 512 clouds; 4 waves x 8
        // points, six clouds per CU: 490 -- a round costs the same ~4.6 k cycles whatever the wave count)
        if (n <= 2048) launch_pruned<512, 4, 4>(b, n, pstride, m, pts, temp, idx, group_pts, group_box, s, view, slice_box);
        else if (n <= 4096) launch_pruned<1024, 4, 4>(b, n, pstride, m, pts, temp, idx, group_pts, group_box, s, view, slice_box);
        else if (n <= 8192) launch_pruned<1024, 8, 4>(b, n, pstride, m, pts, temp, idx, group_pts, group_box, s, view, slice_box);
        else launch_pruned<1024, 16, 4>(b, n, pstride, m, pts, temp, idx, group_pts, group_box, s, view, slice_box);
        return dclr_launch_status();
    }
    if (group_pts || group_box || slice_box || view.batches > 1) return DCLR_E_UNSUPPORTED;
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

extern "C" int dclr_fps_clouds(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                               dclr_stream_t stream) {
    DCLR_REQUIRE(c >= 3);
    return fps_dispatch(b, n, c, m, clouds, nullptr, idx, (hipStream_t)stream);
}

// Workspace of the large-cloud sampler (kernel B): per cloud 1024 * P sorted points (float4), their running
// minima (float), the sorted index list (u32) and the cell ids (u16); 0 where no workspace is used.
static size_t fps_ws_bytes_per_cloud(int n) {
    if (n <= 16384 || n > 65536) return 0;
    const size_t np = n <= 32768 ? 32768 : 65536;
    return np * (16 + 4 + 4 + 2);
}

extern "C" long long dclr_fps_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return DCLR_E_INVALID;
    return (long long)(fps_ws_bytes_per_cloud(n) * (size_t)b);
}

// Workspace sampler (16384 < n <= 65536). `spts` = the sorted points (float4 x, y, z, index bits; 1024 * P per cloud): a
// slice of the workspace, or the caller's group_pts buffer when the groups are exported for set abstraction.
static int fps_launch_paged(int b, int n, int c, int m, const float *clouds, int32_t *idx, float4 *spts, char *rest,
                            float *group_box, float *temp, hipStream_t stream, DclrCloudView view = DclrCloudView{0, 1, 0}) {
    if ((size_t)m * sizeof(int32_t) > 32 * 1024) return DCLR_E_UNSUPPORTED;   // picked[] shares LDS with the histogram
    const size_t np = n <= 32768 ? 32768 : 65536;
    float *stdv = reinterpret_cast<float *>(rest);
    uint32_t *sidx = reinterpret_cast<uint32_t *>(rest + (size_t)b * np * 4);
    uint16_t *cells = reinterpret_cast<uint16_t *>(rest + (size_t)b * np * 8);
    static const int mode = getenv("DCLR_FPS_SINGLE") ? 0 : 1;                 // A/B switch: one sample per barrier round
#define FPS_PAGED(NG_, MODE_)                                                                                         \
    hipLaunchKernelGGL((fps_paged_kernel<NG_, MODE_>), dim3(b), dim3(1024), (size_t)m * sizeof(int32_t), stream, n, c, \
                       m, clouds, idx, spts, stdv, sidx, cells, group_box, temp, view)
    if (np == 32768) { if (mode) FPS_PAGED(8, 1); else FPS_PAGED(8, 0); }
    else             { if (mode) FPS_PAGED(16, 1); else FPS_PAGED(16, 0); }
#undef FPS_PAGED
    return dclr_launch_status();
}

// Level 1 (the reference wrapper's signature has no workspace argument): clouds of 16385..65536 points take the
// workspace kernel with a stream-ordered scratch allocation (hipMallocAsync / hipFreeAsync on the caller's stream).
extern "C" int dclr_furthest_point_sampling(int b, int n, int m, const float *points, float *temp,
                                            int32_t *idx, dclr_stream_t stream) {
    DCLR_REQUIRE(temp != nullptr);
    const size_t need = (b > 0 && n > 0) ? fps_ws_bytes_per_cloud(n) * (size_t)b : 0;
    if (need == 0 || (size_t)m * sizeof(int32_t) > 32 * 1024 || getenv("DCLR_FPS_PLAIN"))
        return fps_dispatch(b, n, 3, m, points, temp, idx, (hipStream_t)stream);
    DCLR_REQUIRE(m > 0 && points && idx);
    void *ws = nullptr;
    if (hipMallocAsync(&ws, need, (hipStream_t)stream) != hipSuccess || ws == nullptr) {
        (void)hipGetLastError();
        return fps_dispatch(b, n, 3, m, points, temp, idx, (hipStream_t)stream);   // no scratch: the global-temp kernel
    }
    const size_t np = n <= 32768 ? 32768 : 65536;
    char *w = static_cast<char *>(ws);
    const int rc = fps_launch_paged(b, n, 3, m, points, idx, reinterpret_cast<float4 *>(w), w + (size_t)b * np * 16, nullptr,
                                    temp, (hipStream_t)stream);
    const hipError_t fe = hipFreeAsync(ws, (hipStream_t)stream);
    return rc != DCLR_OK ? rc : fe == hipSuccess ? DCLR_OK : -(1000 + (int)fe);
}

extern "C" int dclr_fps_clouds_ws(int b, int n, int c, int m, const float *clouds, int32_t *idx, void *workspace,
                                  long long workspace_bytes, dclr_stream_t stream) {
    DCLR_REQUIRE(c >= 3 && b > 0 && n > 0 && m > 0 && clouds && idx);
    const size_t need = fps_ws_bytes_per_cloud(n) * (size_t)b;
    if (need == 0 || getenv("DCLR_FPS_PLAIN")) return fps_dispatch(b, n, c, m, clouds, nullptr, idx, (hipStream_t)stream);
    DCLR_REQUIRE(workspace && workspace_bytes >= (long long)need && ((uintptr_t)workspace & 15) == 0);
    const size_t np = n <= 32768 ? 32768 : 65536;
    char *w = static_cast<char *>(workspace);
    return fps_launch_paged(b, n, c, m, clouds, idx, reinterpret_cast<float4 *>(w), w + (size_t)b * np * 16, nullptr,
                            nullptr, (hipStream_t)stream);
}

extern "C" int dclr_fps_group_layout(int n, int *n_groups, int *group_size) {
    DCLR_REQUIRE(n_groups && group_size);
    return fps_group_layout(n, n_groups, group_size) ? DCLR_OK : DCLR_E_UNSUPPORTED;
}

extern "C" int dclr_fps_clouds_grouped_ws(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                                          float *group_pts, float *group_box, void *workspace, long long workspace_bytes,
                                          dclr_stream_t stream) {
    DCLR_REQUIRE(c >= 3 && b > 0 && n > 16384 && n <= 65536 && m > 0 && clouds && idx && group_pts && group_box &&
                 ((uintptr_t)group_pts & 15) == 0);
    if (getenv("DCLR_FPS_PLAIN")) return DCLR_E_UNSUPPORTED;
    const size_t np = n <= 32768 ? 32768 : 65536;
    DCLR_REQUIRE(workspace && workspace_bytes >= (long long)((size_t)b * np * 10) && ((uintptr_t)workspace & 15) == 0);
    return fps_launch_paged(b, n, c, m, clouds, idx, reinterpret_cast<float4 *>(group_pts), static_cast<char *>(workspace),
                            group_box, nullptr, (hipStream_t)stream);
}

extern "C" int dclr_fps_clouds_grouped(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                                       float *group_pts, float *group_box, dclr_stream_t stream) {
    DCLR_REQUIRE(c >= 3 && group_pts && group_box && ((uintptr_t)group_pts & 15) == 0);
    int ng, gs;
    if (n > 16384) return DCLR_E_UNSUPPORTED;                     // larger clouds: dclr_fps_clouds_grouped_ws
    if (!fps_group_layout(n, &ng, &gs) || getenv("DCLR_FPS_PLAIN")) return DCLR_E_UNSUPPORTED;
    return fps_dispatch(b, n, c, m, clouds, nullptr, idx, (hipStream_t)stream, reinterpret_cast<float4 *>(group_pts),
                        group_box);
}

// The grouped sampler over batches that are NOT concatenated (DclrCloudView, common.h): what the pipelined runner launches
// for the batches of one sampling group. workspace: as dclr_fps_clouds_grouped_ws for n > 16384, ignored otherwise.
extern "C" int dclr_fps_clouds_grouped_batched(int b, int n, int c, int m, const float *clouds, int pairs_per_batch,
                                               int n_batches, long long batch_stride, int32_t *idx, float *group_pts,
                                               float *group_box, float *slice_box, void *workspace,
                                               long long workspace_bytes, dclr_stream_t stream) {
    DCLR_REQUIRE(c >= 3 && b > 0 && m > 0 && clouds && idx && group_pts && group_box && ((uintptr_t)group_pts & 15) == 0);
    DCLR_REQUIRE(pairs_per_batch > 0 && n_batches > 0 && batch_stride >= 0 && b == 2 * pairs_per_batch * n_batches);
    if (getenv("DCLR_FPS_PLAIN")) return DCLR_E_UNSUPPORTED;
    const DclrCloudView view{pairs_per_batch, n_batches, batch_stride};
    int ng, gs;
    if (!fps_group_layout(n, &ng, &gs)) return DCLR_E_UNSUPPORTED;
    if (n > 16384) {
        if (slice_box) return DCLR_E_UNSUPPORTED;           // the workspace kernel exports no slice boxes
        const size_t np = n <= 32768 ? 32768 : 65536;
        DCLR_REQUIRE(workspace && workspace_bytes >= (long long)((size_t)b * np * 10) && ((uintptr_t)workspace & 15) == 0);
        return fps_launch_paged(b, n, c, m, clouds, idx, reinterpret_cast<float4 *>(group_pts), static_cast<char *>(workspace),
                                group_box, nullptr, (hipStream_t)stream, view);
    }
    if (slice_box && gs <= 64) return DCLR_E_UNSUPPORTED;   // one slice per group: the group box is the slice box
    return fps_dispatch(b, n, c, m, clouds, nullptr, idx, (hipStream_t)stream, reinterpret_cast<float4 *>(group_pts),
                        group_box, view, slice_box);
}
