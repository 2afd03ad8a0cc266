// Batched k-nearest-neighbour search for gfx950.
//
// Replaces torch_cluster.knn as DeepCLR calls it (/root/reference/deepclr/models/deepclr.py:164-166):
// for every query (template point) the k nearest candidates (source points) of the same pair,
// ascending squared distance, equal distances in ascending candidate index -- the order the
// published torch-cluster 1.5.9 GPU kernel's strict-">" insertion list yields (oracle/primitives.c).
//
// A workgroup stages one candidate cloud in LDS; each wave then answers a few queries: lane l owns
// candidates c*64 + l (CPL distance evaluations per lane); a ballot-counted threshold lets ~25-40 candidates
// through, they are compacted to one per lane and ranked against each other; the general fallback extracts the
// k nearest in k rounds of a two-step DPP reduction (min distance bits, then min index among equals).
#include "common.h"

namespace {

constexpr int KNN_WAVES = 4;     // waves per workgroup
constexpr int KNN_QRUN = 2;      // queries per wave
constexpr int KNN_MAX_NX = 4096; // candidates staged in LDS (3 * 4 bytes each)
constexpr uint32_t KNN_INF = 0x7F800000u;
constexpr float KNN_SLOT_INIT = 1e10f;   // torch-cluster 1.5.9: dist = torch::full(..., 1e10), col = -1
constexpr int KNN_LIMIT = 40;    // most lanes a selection threshold may let through (fast path of knn_block)

struct KnnXyzSource {            // (clouds*n, 3) packed points
    const float *base;
    __device__ __forceinline__ void load(size_t cloud, int n, int i, float &x, float &y, float &z) const {
        const float *p = base + (cloud * n + i) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
};

struct KnnRowSource {            // feature rows F: xyz at columns 64..66 of a 68-float row
    const float *base;
    __device__ __forceinline__ void load(size_t cloud, int n, int i, float &x, float &y, float &z) const {
        // columns 64..67 of a row = x y z 0: one aligned 16-byte load (rows are 272 bytes)
        const float4 v = *reinterpret_cast<const float4 *>(base + (cloud * n + i) * DCLR_F_STRIDE + 64);
        x = v.x; y = v.y; z = v.z;
    }
};

// Workgroup: the candidate cloud is staged once into LDS (structure of arrays), then every wave
// answers KNN_QRUN queries. Small register footprint -> many waves per SIMD hide the serial
// k-round selection latency.
template <int CPL, typename Src, typename OutFn>
__device__ __forceinline__ void knn_block(const Src &cand, size_t cand_cloud, int nx, const Src &query,
                                          size_t query_cloud, int ny, int k, int block, OutFn out) {
    extern __shared__ float knn_lds[];                      // x[nxp] y[nxp] z[nxp], nxp = 64 * CPL
    constexpr int NXP = 64 * CPL;
    float *sx = knn_lds, *sy = knn_lds + NXP, *sz = knn_lds + 2 * NXP;
    uint32_t *knn_compact = reinterpret_cast<uint32_t *>(knn_lds + 3 * NXP);    // per wave: 64 distances + 64 indices
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NXP; i += KNN_WAVES * 64) {
        // padding slots lie 3e18 away: their squared distance (~1e37, finite) fails the "< 1e10" test below like any
        // candidate the published kernel would never insert
        float x = 3.0e18f, y = 3.0e18f, z = 3.0e18f;
        if (i < nx) cand.load(cand_cloud, nx, i, x, y, z);
        sx[i] = x; sy[i] = y; sz[i] = z;
    }
    __syncthreads();
    const int q0 = (block * KNN_WAVES + wave) * KNN_QRUN;
#pragma unroll 1
    for (int q = q0; q < q0 + KNN_QRUN && q < ny; ++q) {
        float qx, qy, qz;
        query.load(query_cloud, ny, q, qx, qy, qz);          // wave-uniform
        uint32_t d[CPL];
        if constexpr (CPL % 2 == 0) {
            // two candidates per instruction (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations in the same order
            // as dclr_sqdist -- (candidate - query), accumulated x, y, z: the published kernel's order)
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#pragma unroll
            for (int c = 0; c < CPL; c += 2) {
                const int i = c * 64 + lane;
                const f2 ax = {sx[i], sx[i + 64]}, ay = {sy[i], sy[i + 64]}, az = {sz[i], sz[i + 64]};
                const f2 dx = ax - q2x, dy = ay - q2y, dz = az - q2z;
                const f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const f2 dd = (xx + yy) + zz;
                // the published kernel's slots start at 1e10 and take a candidate only if "slot > distance": a candidate
                // at 1e10 or beyond (or NaN) is never inserted -- its slot stays -1
                d[c] = dd[0] < KNN_SLOT_INIT ? __float_as_uint(dd[0]) : KNN_INF;
                d[c + 1] = dd[1] < KNN_SLOT_INIT ? __float_as_uint(dd[1]) : KNN_INF;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int i = c * 64 + lane;
                const float dd = dclr_sqdist(sx[i], sy[i], sz[i], qx, qy, qz);
                d[c] = dd < KNN_SLOT_INIT ? __float_as_uint(dd) : KNN_INF;
            }
        }
        // Fast path: shrink the problem to <= 64 candidates, one per lane, then run the k arg-min rounds on those.
        //  1. threshold: a value tau with k <= #lanes(lane minimum <= tau) <= KNN_LIMIT, found by a few pivot
        //     tries (pivot = some lane's minimum, count = one ballot). Every lane counted holds a candidate
        //     <= tau, so at least k candidates pass; the k nearest overall all pass (they are <= the k-th
        //     smallest lane minimum <= tau).
        //  2. all candidates <= tau are compacted (wave prefix sum, LDS) to one per lane -- typically 25-40 of
        //     the 1024; more than 64 (heavy ties) or no suitable pivot falls back to the general selection.
        //  3. every survivor computes its rank among the survivors (ties: lowest candidate index) and the first k store.
        bool fast = false;
        {
            uint32_t lmin = d[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) lmin = lmin < d[c] ? lmin : d[c];
            uint32_t lo = 0u, hi = 0xFFFFFFFFu, tau = 0u;
            bool first = true, found = false;
#pragma unroll 1
            for (int t = 0; t < 12 && !found && k <= KNN_LIMIT; ++t) {         // (k > KNN_LIMIT: straight to the k-round selection)
                const uint64_t pool = __ballot(first || (lmin > lo && lmin < hi));
                if (pool == 0) break;
                const uint32_t piv = (uint32_t)__builtin_amdgcn_readlane((int)lmin, __builtin_ctzll(pool));
                const int c = __builtin_popcountll(__ballot(lmin <= piv));
                first = false;
                if (c < k) lo = piv;
                else if (c > KNN_LIMIT) hi = piv;
                else { tau = piv; found = true; }
            }
            if (found && tau < KNN_INF) {
                // count and compact in one pass: a ballot per candidate slice gives the slice's survivors their places
                // (most slices of a lane's 16 hold none); the order of the survivors is irrelevant, they are ranked below
                uint32_t *cd_lds = knn_compact + wave * 128, *ci_lds = cd_lds + 64;
                int total = 0;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const bool pass = d[c] <= tau;
                    const uint64_t mask = __ballot(pass);
                    if (mask != 0) {                                 // wave-uniform
                        const int pos = total + (int)dclr_lanemask_lt_popc(mask);
                        if (pass && pos < 64) { cd_lds[pos] = d[c]; ci_lds[pos] = (uint32_t)(c * 64 + lane); }
                        total += __builtin_popcountll(mask);
                    }
                }
                if (total <= 64) {                                   // wave-uniform
                    // same wave wrote and reads: LDS operations of a wave complete in order
                    const uint32_t cd = lane < total ? cd_lds[lane] : 0xFFFFFFFFu;
                    const uint32_t ci = lane < total ? ci_lds[lane] : 0xFFFFFFFFu;
                    // Selection by rank: a survivor's output slot is the number of survivors ordered before it
                    // (distance, then candidate index) -- `total` steps of two lane reads and one 64-bit compare,
                    // instead of k dependent wave arg-min rounds (~7 cross-lane steps each); every lane whose rank is
                    // below k then stores its own result.
                    const uint64_t key = ((uint64_t)cd << 32) | ci;
                    int rank = 0;
#pragma unroll 1
                    for (int j = 0; j < total; ++j) {
                        const uint64_t kj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)cd, j) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)ci, j);
                        rank += kj < key ? 1 : 0;
                    }
                    if (lane < total && rank < k) out(q, rank, (int)ci);
                    fast = true;
                }
            }
        }
        if (fast) continue;
        // Each lane keeps its two nearest unused candidates (m1 <= m2, ties in index order); a round is then one
        // wave arg-min over m1 plus a pop in the winning lane. Only when a lane has been popped twice -- k picks
        // spread over 64 lanes: about once per query -- are the pairs rebuilt from the distance registers
        // (rescanning all CPL registers every round, as a plain selection does, is ~3x the instructions).
        uint64_t used = 0;                                   // bit c: this lane's candidate c is already output
        uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
        int c1 = 0, c2 = 0, have = 0;                        // have: how many of (m1, m2) are still valid
#pragma unroll 1
        for (int s = 0; s < k; ++s) {
            if (__ballot(have == 0) != 0) {                  // wave-uniform
                m1 = 0xFFFFFFFFu; m2 = 0xFFFFFFFFu; c1 = 0; c2 = 0;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const uint32_t v = ((used >> c) & 1ull) ? 0xFFFFFFFFu : d[c];
                    const bool lt1 = v < m1, lt2 = v < m2;
                    c2 = lt1 ? c1 : (lt2 ? c : c2);
                    m2 = lt1 ? m1 : (lt2 ? v : m2);
                    c1 = lt1 ? c : c1;
                    m1 = lt1 ? v : m1;
                }
                have = 2;
            }
            const uint32_t li = (uint32_t)(c1 * 64 + lane);
            const uint32_t wmin = dclr_wave_min_u32(m1);
            // one lane holds the minimum unless distances tie exactly: only then a second reduction picks the
            // lowest candidate index among the holders
            const uint64_t holders = __ballot(m1 == wmin);
            uint32_t widx;
            if (__builtin_popcountll(holders) == 1)
                widx = (uint32_t)__builtin_amdgcn_readlane((int)li, __builtin_ctzll(holders));
            else
                widx = dclr_wave_min_u32(m1 == wmin ? li : 0xFFFFFFFFu);
            if (lane == 0) out(q, s, wmin >= KNN_INF ? -1 : (int)widx);
            if (lane == (int)(widx & 63)) {
                used |= 1ull << (widx >> 6);
                m1 = m2; c1 = c2; m2 = 0xFFFFFFFFu;
                have -= 1;
            }
        }
    }
}

template <int CPL>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_xyz_kernel(int nx, int ny, int k,
                                                                 const float *__restrict__ x,
                                                                 const float *__restrict__ y,
                                                                 int64_t *__restrict__ row,
                                                                 int64_t *__restrict__ col) {
    const size_t bi = blockIdx.y;
    KnnXyzSource cs{x}, qs{y};
    knn_block<CPL>(cs, bi, nx, qs, bi, ny, k, (int)blockIdx.x, [&](int q, int s, int ci) {
        const size_t gq = bi * ny + q;
        row[gq * k + s] = ci < 0 ? -1 : (int64_t)gq;
        col[gq * k + s] = ci < 0 ? -1 : (int64_t)(bi * nx + ci);
    });
}

template <int CPL>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_rows_kernel(int pairs, int npoint, int k,
                                                                  const float *__restrict__ f_rows,
                                                                  int32_t *__restrict__ knn_idx) {
    // (the blocks of one pair stage the same candidate cloud: with a multiple of 8 pairs a pair's blocks are given ids
    // that the hardware deals to ONE XCD -- block ids b and b + 8 share an L2 -- instead of all eight; speed only)
    size_t bi = blockIdx.y;
    int bx = blockIdx.x;
    if ((gridDim.y & 7u) == 0u) {
        const unsigned linear = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = linear & 7u, i = linear >> 3;
        bi = (size_t)(i / gridDim.x) * 8u + xcd;
        bx = (int)(i % gridDim.x);
    }
    KnnRowSource src{f_rows};
    knn_block<CPL>(src, bi + pairs, npoint, src, bi, npoint, k, bx, [&](int q, int s, int ci) {
        knn_idx[(bi * npoint + q) * k + s] = ci;
    });
}

template <template <int> class Launcher, typename... Args>
int knn_dispatch(int nx, Args... args) {
    if (nx <= 64) return Launcher<1>::go(args...);
    if (nx <= 128) return Launcher<2>::go(args...);
    if (nx <= 256) return Launcher<4>::go(args...);
    if (nx <= 512) return Launcher<8>::go(args...);
    if (nx <= 1024) return Launcher<16>::go(args...);
    if (nx <= 2048) return Launcher<32>::go(args...);
    if (nx <= KNN_MAX_NX) return Launcher<64>::go(args...);
    return DCLR_E_UNSUPPORTED;
}

template <int CPL>
struct XyzLauncher {
    static int go(int b, int nx, int ny, int k, const float *x, const float *y, int64_t *row, int64_t *col,
                  hipStream_t s) {
        const int per_wg = KNN_WAVES * KNN_QRUN;
        hipLaunchKernelGGL((knn_xyz_kernel<CPL>), dim3((ny + per_wg - 1) / per_wg, b), dim3(KNN_WAVES * 64),
                           (size_t)3 * 64 * CPL * sizeof(float) + KNN_WAVES * 128 * sizeof(uint32_t), s, nx, ny, k, x, y, row, col);
        return dclr_launch_status();
    }
};

template <int CPL>
struct RowsLauncher {
    static int go(int pairs, int npoint, int k, const float *f_rows, int32_t *knn_idx, hipStream_t s) {
        const int per_wg = KNN_WAVES * KNN_QRUN;
        hipLaunchKernelGGL((knn_rows_kernel<CPL>), dim3((npoint + per_wg - 1) / per_wg, pairs),
                           dim3(KNN_WAVES * 64), (size_t)3 * 64 * CPL * sizeof(float) + KNN_WAVES * 128 * sizeof(uint32_t), s, pairs, npoint, k, f_rows,
                           knn_idx);
        return dclr_launch_status();
    }
};

}  // namespace

extern "C" int dclr_knn(int b, int nx, int ny, int k, const float *x, const float *y, int64_t *row,
                        int64_t *col, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && nx > 0 && ny > 0 && x && y && row && col && b <= 65535);
    DCLR_REQUIRE(k >= 1 && nx >= k);           // (k > 40: the k-round selection; any k up to the candidate count)
    return knn_dispatch<XyzLauncher>(nx, b, nx, ny, k, x, y, row, col, (hipStream_t)stream);
}

extern "C" int dclr_knn_rows(int pairs, int npoint, int k, const float *f_rows, int32_t *knn_idx,
                             dclr_stream_t stream) {
    DCLR_REQUIRE(pairs > 0 && npoint > 0 && f_rows && knn_idx && pairs <= 65535);
    DCLR_REQUIRE(k >= 1 && npoint >= k);
    return knn_dispatch<RowsLauncher>(npoint, pairs, npoint, k, f_rows, knn_idx, (hipStream_t)stream);
}
