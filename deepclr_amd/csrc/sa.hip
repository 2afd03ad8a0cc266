// Fused multi-scale set abstraction for gfx950.
//
// Replaces, for the reference's SetAbstraction.forward (/root/reference/deepclr/models/deepclr.py:88-94)
// and the absent PointnetSAModuleMSG it drives (constructed at deepclr.py:63-70, use_xyz=True, bn=False),
// the chain  gather centroid -> per scale [ball_query -> group xyz/features -> subtract centroid ->
// 1x1 conv (c->16->16->32, ReLU each) -> max over nsample] -> concat.  The reference design
// materialises (clouds, {4,16,16,32}, npoint, nsample) tensors (~856 MB per KITTI pair);
// here nothing but the 64 pooled features per centroid ever leaves the CU.
//
// Workgroup = 4 waves x 4 centroids of one cloud.
//   sweep   the cloud streams once per workgroup through an LDS tile (256 points, next tile prefetched in registers,
//           padded to float4); every wave tests each 64-point slice against its 4 centroids (scalar
//           coordinates) and both radii. In-radius lanes are ballot-compacted, in ascending point
//           order and capped at nsample per (centroid, scale) -- exactly the index set the published
//           ball query keeps -- into one small per-wave, per-scale ring of (centroid, point) entries.
//   drain   whenever a ring holds 64 entries (and once at the end for the remainder) the wave calls
//           sa_drain: the shared MLP with lane = entry and all weights as scalar operands, a
//           64 x 32 transpose through LDS and a segmented maximum per centroid slot, folded by the
//           caller into per-(scale, centroid) running maxima held by lane = (row parity, channel).
//   groups  when the sampling kernel exported its spatial partition (<= 64 compact groups per cloud
//           with tight boxes), each centroid first tests the group boxes (rounding-safe lower bound
//           of the distance, as in fps.hip) and scans only the groups its largest ball can reach --
//           a dozen 64-point slices instead of N/64. Hits then arrive out of index order, which is
//           irrelevant while a neighbourhood stays within its nsample cap (the max does not care);
//           a centroid whose count exceeds a cap (or the ring) is redone wave-locally: a radix select over the
//           hits of its candidate groups finds the nsample-th smallest point index, which is all the index order
//           decides. The exhaustive in-order sweep serves calls without groups only.
// Slots that would only repeat the first hit are skipped: max() over a multiset equals max() over
// its support, so the result is identical. A centroid with no hit reproduces the published
// behaviour (zero-filled index row => every slot is point 0).
#include <stdlib.h>
#include "mma16f.h"

namespace {

constexpr int SA_WAVES = 4;
constexpr int SA_CPW = 4;                       // centroids per wave
constexpr int SA_TILE = 256;                    // points per LDS tile (one tile, two barriers per tile: the sweep is the
                                                // rare path; 37 KB of LDS in all keep four workgroups on a CU)
constexpr int SA_RING = 512;                    // ring capacity: < 64 left over + one centroid's neighbours (fast path) or
                                                // + 4 slices staged between drain checks (sweep); power of two
constexpr int SA_MAX_SCALES = 2;
#ifndef SA_SLICE_STEP
#define SA_SLICE_STEP 6                         // slices fetched together on the slice path (8: 32 bytes of scratch beside the drain)
#endif
constexpr int SA_H1 = 16, SA_H2 = 16, SA_OUT = 32;

struct SaParams {
    int n, npoint, n_scales;
    float radius2[SA_MAX_SCALES];
    float radius2_max;
    int nsample[SA_MAX_SCALES];
    const float *mlp[SA_MAX_SCALES];
    const float4 *group_pts;                    // optional spatial groups from the sampling kernel (or null)
    const float *group_box;
    int n_groups, group_size;
    DclrCloudView view;                         // how the call's clouds lie in memory (dclr_sa_msg_fused_batched)
    const float *slice_box;                     // optional (<= 64 groups of > 64 points): boxes of the groups' 64-point slices
    uint32_t *overflow;                         // NULL, or the word that receives 1 when a split-f16 activation was clamped
};

template <int C>
__device__ __forceinline__ float4 sa_load_point(const float *__restrict__ cloud, int k) {
    if constexpr (C == 4) {
        return *reinterpret_cast<const float4 *>(cloud + (size_t)k * 4);
    } else {
        const float *p = cloud + (size_t)k * 3;
        return make_float4(p[0], p[1], p[2], 0.f);
    }
}

constexpr int SA_MLP_FLOATS = SA_H1 * 4 + SA_H1 + SA_H2 * SA_H1 + SA_H2 + SA_OUT * SA_H2 + SA_OUT;   // 896 at c = 4

// Workgroup-shared state. Namespace scope so that the (deliberately not inlined) drain routine can
// address it; both template instances of the kernel use the same layout.
__shared__ float4 sa_tile[1][SA_TILE];
__shared__ uint32_t sa_ring[SA_WAVES][SA_MAX_SCALES][SA_RING];
__shared__ __attribute__((aligned(16))) float sa_obuf[SA_WAVES][64 * 4 + 64];   // drain staging: 64 inputs (float4) + 64 centroid tags
__shared__ float sa_cxyz[SA_WAVES][SA_CPW][4];
__shared__ uint32_t *sa_overflow;               // SaParams.overflow, where the out-of-line drain routine finds it
__shared__ float sa_c16[SA_WAVES * SA_CPW][4];      // the workgroup's 16 centroids (groups path: waves pull them one at a time)
__shared__ int sa_next;                              // next centroid of the workgroup to be taken
__shared__ int sa_crowd[SA_WAVES * SA_CPW];          // crowded centroids of the workgroup (redone by the four waves together)
__shared__ int sa_ncrowd;
__shared__ int sa_cnt[SA_MAX_SCALES];                // their hit counts, summed over the waves

// running maxima per (wave, scale, centroid slot, channel): non-negative floats, compared as u32
__shared__ uint32_t sa_acc[SA_WAVES][SA_MAX_SCALES][SA_CPW][SA_OUT];
__shared__ int sa_tot[SA_WAVES][SA_CPW][SA_MAX_SCALES];
__shared__ __attribute__((aligned(16))) float sa_w[SA_MAX_SCALES][SA_MLP_FLOATS];       // true neighbour counts of centroids done on the fast path

#ifdef SA_DEBUG
__device__ unsigned long long sa_dbg_w[16384][8];   // per wave, cycles: [0] total, [1] fast path incl. drains, [2] drains,
                                                   // [3] #drains, [4] sweep, [5] 1, [6] drain: point load, [7] drain: MLP + fold
__shared__ unsigned long long sa_dbg_l[4][2];
__device__ int getenv_dbg2 = 0;                    // harness switch: 1 = slice-path stamps in slots 4, 6, 7
#define SA_STAMP(v) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); } while (0)
#endif

// One pass of the shared MLP over `take` (<= 64) ring entries of scale `s`; the results are folded into
// sa_acc. Kept out of line on purpose (one copy, called from three places).
//
// The three layers run on v_mfma_f32_16x16x4_f32 as W * H^T with the entries as COLUMNS (four tiles of 16
// entries): the accumulator of lane (entry e = lane & 15, quarter kq = lane >> 4) holds channels
// 4 kq .. 4 kq + 3 of entry e, which is exactly the B operand the next layer needs when step j of its K loop
// is given k = 4 kq + j -- so the activations never leave the registers between layers; only the weight
// fragments (A operand, lane (m, kq) holds W[m][4 kq .. 4 kq + 3]) are laid out for that k order.
// 52 MFMAs per pass; the scalar version (lane = entry, 832 FMAs with LDS-broadcast weights) took ~20k
// cycles per pass and dominated the kernel wherever neighbourhoods are dense.
//
// F16: layers 2 and 3 (K = 16: one k-step of v_mfma_f32_16x16x16_f16, whose operand maps are the f32 form's with
// four consecutive k per lane -- lane (e, kq) holds k = 4 kq .. 4 kq + 3 -- so the accumulator of one layer is
// still the B operand of the next) run on split-f16 operands (mma16f.h: hi*hi + 2^-11 (hi*lo + lo*hi), f32
// accumulation, f32-grade results): 3 + 6 f16 instructions instead of 4 + 8 f32 ones at a quarter of the matrix-pipe
// time each. With dense neighbourhoods (LiDAR near field, ModelNet) four waves per SIMD queue on one matrix pipe and
// the drain was bound by it (1,470 cycles per 16-entry tile against 416 of pipe time). Layer 1 (K = 3 or 4) stays on
// the f32 instruction: one issue either way.
template <int C, bool F16>
__device__ __noinline__ void sa_drain(const float *cloud, int wave, int s, int head, int take) {
    cloud = dclr_uniform(cloud);
    wave = dclr_uniform(wave); s = dclr_uniform(s); head = dclr_uniform(head); take = dclr_uniform(take);
    const int lane = dclr_lane(), e16 = lane & 15, kq = lane >> 4;
#ifdef SA_DEBUG
    unsigned long long g0, g1, g2, g3;
    SA_STAMP(g0);
#endif
    // stage the inputs (lane = entry): relative position + feature, and the centroid slot (-1: no entry)
    float *stg = sa_obuf[wave];
    int *tag = reinterpret_cast<int *>(stg + 256);
    {
        const bool valid = lane < take;
        const uint32_t e = valid ? sa_ring[wave][s][(head + lane) & (SA_RING - 1)] : 0u;
        const int c = (int)(e >> 16), k = (int)(e & 0xFFFFu);
#ifdef SA_ABL_GATHER          // timing probe only (wrong rows): the drained entries' points from 64 fixed addresses (cache hits)
        const float4 p = sa_load_point<C>(cloud, lane);
#else
        const float4 p = sa_load_point<C>(cloud, k);
#endif
        *reinterpret_cast<float4 *>(stg + 4 * lane) =
            make_float4(p.x - sa_cxyz[wave][c][0], p.y - sa_cxyz[wave][c][1], p.z - sa_cxyz[wave][c][2], p.w);
        tag[lane] = valid ? c : -1;
    }
#ifdef SA_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SA_STAMP(g1);
#endif
    const float *w1 = &sa_w[s][0];
    const float *b1 = w1 + SA_H1 * C, *w2 = b1 + SA_H1, *b2 = w2 + SA_H2 * SA_H1;
    const float *w3 = b2 + SA_H2, *b3 = w3 + SA_OUT * SA_H2;
    const float a1 = kq < C ? w1[e16 * C + kq] : 0.f;                        // W1[m][k = kq]
    float a2[4], a3[2][4], c1[4], c2[4], c3[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a2[j] = w2[e16 * SA_H1 + 4 * kq + j];
        a3[0][j] = w3[e16 * SA_H2 + 4 * kq + j];
        a3[1][j] = w3[(16 + e16) * SA_H2 + 4 * kq + j];
        c1[j] = b1[4 * kq + j]; c2[j] = b2[4 * kq + j];
        c3[0][j] = b3[4 * kq + j]; c3[1][j] = b3[16 + 4 * kq + j];
    }
    // split-f16 weight fragments (unused and dropped by the compiler in the f32 instance)
    dclr_h4 a2h, a2l, a3h[2], a3l[2];
    if constexpr (F16) {
        auto split4 = [](const float (&v)[4], dclr_h4 &hi, dclr_h4 &lo) {
            dclr_h2 h0, l0, h1, l1;
            dclr_split2(v[0], v[1], h0, l0);
            dclr_split2(v[2], v[3], h1, l1);
            hi = dclr_h4{h0[0], h0[1], h1[0], h1[1]};
            lo = dclr_h4{l0[0], l0[1], l1[0], l1[1]};
        };
        split4(a2, a2h, a2l);
        split4(a3[0], a3h[0], a3l[0]);
        split4(a3[1], a3h[1], a3l[1]);
    }
    uint32_t *acc = &sa_acc[wave][s][0][0];
    uint64_t clamped = 0;                                                    // lanes that saw an activation beyond the f16 range (F16; scalar registers)
#ifdef SA_DEBUG
    SA_STAMP(g2);
#endif
    // raw third-layer outputs (before bias and ReLU) of the 16 entries of tile t
    auto tile_mlp = [&](int t, dclr_f32x4 (&h3)[2]) {
        const float x = stg[4 * (16 * t + e16) + kq];                        // input component kq of entry 16 t + e16
        dclr_f32x4 h1 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
        h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, x, h1, 0, 0, 0);
        h3[0] = dclr_f32x4{0.f, 0.f, 0.f, 0.f};
        h3[1] = dclr_f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (F16) {
            // relu(h1 + b1) as hi / lo halves = the B operand of layer 2 (k = 4 kq + i)
            dclr_h2 p0, q0, p1, q1;
            float peak = 0.f;                                                // of this tile only: the kernel has no register to spare
            dclr_split2_relu(h1[0] + c1[0], h1[1] + c1[1], p0, q0, peak);
            dclr_split2_relu(h1[2] + c1[2], h1[3] + c1[3], p1, q1, peak);
            const dclr_h4 b1h = {p0[0], p0[1], p1[0], p1[1]}, b1l = {q0[0], q0[1], q1[0], q1[1]};
            dclr_f32x4 x2 = {0.f, 0.f, 0.f, 0.f};
            h2 = dclr_f32x4{c2[0], c2[1], c2[2], c2[3]};                     // the accumulator starts at the bias
            h2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a2h, b1h, h2, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a2h, b1l, x2, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a2l, b1h, x2, 0, 0, 0);
            dclr_split2_relu(fmaf(x2[0], DCLR_SPLIT_INV, h2[0]), fmaf(x2[1], DCLR_SPLIT_INV, h2[1]), p0, q0, peak);
            dclr_split2_relu(fmaf(x2[2], DCLR_SPLIT_INV, h2[2]), fmaf(x2[3], DCLR_SPLIT_INV, h2[3]), p1, q1, peak);
#ifndef DCLR_SA_NO_OVF
            clamped |= __ballot(peak > DCLR_F16_MAX);
#endif
            const dclr_h4 b2h = {p0[0], p0[1], p1[0], p1[1]}, b2l = {q0[0], q0[1], q1[0], q1[1]};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dclr_f32x4 x3 = {0.f, 0.f, 0.f, 0.f};
                h3[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a3h[u], b2h, h3[u], 0, 0, 0);
                x3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a3h[u], b2l, x3, 0, 0, 0);
                x3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a3l[u], b2h, x3, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) h3[u][i] = fmaf(x3[i], DCLR_SPLIT_INV, h3[u][i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) h1[i] = fmaxf(h1[i] + c1[i], 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) h2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j], h1[j], h2, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) h2[i] = fmaxf(h2[i] + c2[i], 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h3[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[0][j], h2[j], h3[0], 0, 0, 0);
                h3[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[1][j], h2[j], h3[1], 0, 0, 0);
            }
        }
    };
    // Fold: max over entries of relu(h + b3) = relu(max over entries of h + b3) (monotone), so consecutive tiles of ONE
    // centroid -- the rule with dense neighbourhoods, the ring is filled centroid by centroid -- only update a running
    // per-lane maximum (8 v_max). The reduction over the 16 lanes of a DPP row and the LDS atomic max (one lane per
    // quarter: sixteen lanes on one LDS word serialise) are paid once per run of tiles instead of once per tile (they
    // were 32 DPP steps + 8 atomics of ~110 instructions per tile). A tile that mixes centroids (or holds padding
    // lanes, tag -1) ends the run and folds per lane.
    int run_cen = -1;                                                        // wave-uniform
    dclr_f32x4 rmax[2];
    auto flush = [&]() {
        if (run_cen >= 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t v = dclr_row16_max_lanes(__float_as_uint(fmaxf(rmax[u][i] + c3[u][i], 0.f)));
                    if (e16 == 15) atomicMax(acc + run_cen * SA_OUT + 16 * u + 4 * kq + i, v);
                }
        }
        run_cen = -1;
    };
    auto fold = [&](int t, const dclr_f32x4 (&h3)[2]) {
        const int cen = tag[16 * t + e16];
        const int cen0 = __builtin_amdgcn_readfirstlane(cen);
        if (__ballot(cen != cen0) == 0 && cen0 >= 0) {
            if (cen0 != run_cen) {
                flush();
                run_cen = cen0;
                rmax[0] = h3[0]; rmax[1] = h3[1];
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) rmax[u][i] = fmaxf(rmax[u][i], h3[u][i]);
            }
        } else {
            flush();
            if (cen >= 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        atomicMax(acc + cen * SA_OUT + 16 * u + 4 * kq + i, __float_as_uint(fmaxf(h3[u][i] + c3[u][i], 0.f)));
            }
        }
    };
    const int n_tiles = (take + 15) >> 4;
#pragma unroll 1
    for (int t = 0; t < n_tiles; ++t) {
        dclr_f32x4 ha[2];
        tile_mlp(t, ha);
        fold(t, ha);
    }
    flush();
    if constexpr (F16) {
        if (clamped != 0) dclr_report_overflow(sa_overflow, 2.f * DCLR_F16_MAX);     // the LDS word is read only then
    }
#ifdef SA_DEBUG
    SA_STAMP(g3);
    if (lane == 0) { sa_dbg_l[wave][0] += g1 - g0; sa_dbg_l[wave][1] += g3 - g2; }
#endif
}

// Rounded lower bound of dclr_sqdist(c, p) over all p in the box (same operation order; rounding is monotone).
__device__ __forceinline__ float sa_box_lower_bound(float lx, float ly, float lz, float hx, float hy, float hz,
                                                    float cx, float cy, float cz) {
    const float dx = fmaxf(fmaxf(lx - cx, cx - hx), 0.f);
    const float dy = fmaxf(fmaxf(ly - cy, cy - hy), 0.f);
    const float dz = fmaxf(fmaxf(lz - cz, cz - hz), 0.f);
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}


template <int C, int NCH, bool F16, bool SL = false>     // SL: slice boxes present (NCH == 1): the scan works slice by slice
__global__ __launch_bounds__(SA_WAVES * 64, 4) void sa_msg_kernel(SaParams prm,
                                                               const float *__restrict__ clouds,
                                                               const int32_t *__restrict__ fps_idx,
                                                               float *__restrict__ out_rows,
                                                               int32_t *__restrict__ counts) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (cloud, centroid block). The workgroups of ONE cloud all scan that cloud's groups (256 KB of sorted
    // points): dealt round-robin over the 8 XCDs as the hardware does with consecutive block ids, every XCD's L2 fetches
    // every cloud (measured 443 MB per 160-cloud launch against 88 MB of clouds + groups + rows). With a multiple of 8
    // clouds, block L goes to cloud (L / 8 / blocks_per_cloud) * 8 + L % 8: a cloud's blocks share one L2. Speed only --
    // any placement gives the same rows (MI355X_MICROARCH.md: block b and b + 8 share an XCD, not guaranteed).
    size_t bi = blockIdx.y;
    int bx = blockIdx.x;
    if ((gridDim.y & 7u) == 0u) {
        const unsigned linear = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned xcd = linear & 7u, i = linear >> 3;
        bi = (size_t)(i / gridDim.x) * 8u + xcd;
        bx = (int)(i % gridDim.x);
    }
    const int j0 = (bx * SA_WAVES + wave) * SA_CPW;                  // this wave's first centroid
    const float *cloud = dclr_uniform(clouds + dclr_cloud_offset(prm.view, bi, (size_t)prm.n * C));   // scalar registers

    // The workgroup's 16 centroids go to LDS once. Without groups (exhaustive sweep) wave w keeps centroids 4 w .. 4 w + 3
    // for the whole kernel; with groups the waves PULL centroids one at a time (sa_next): a workgroup lives as long as its
    // slowest wave, and with a fixed four centroids per wave that was the wave that drew the crowded ones (LiDAR-density
    // clouds: per-wave cycles median 61 k, p90 124 k -- the maximum of four is twice the mean).
    const int jw0 = bx * SA_WAVES * SA_CPW;                          // the workgroup's first centroid
    const int n_wg = prm.npoint - jw0 < SA_WAVES * SA_CPW ? prm.npoint - jw0 : SA_WAVES * SA_CPW;
    if (tid < SA_WAVES * SA_CPW) {
        const int jc = tid < n_wg ? jw0 + tid : prm.npoint - 1;
        const float4 q = sa_load_point<C>(cloud, fps_idx[bi * prm.npoint + jc]);
        sa_c16[tid][0] = q.x; sa_c16[tid][1] = q.y; sa_c16[tid][2] = q.z; sa_c16[tid][3] = 0.f;
    }
    if (tid == 0) { sa_next = 0; sa_ncrowd = 0; sa_overflow = prm.overflow; }
    int cnt[SA_CPW][SA_MAX_SCALES];
    int jrow[SA_CPW];                                                // centroid (row) index of slot c, -1: slot unused
    const int n_live = prm.npoint - j0 < SA_CPW ? (prm.npoint - j0 > 0 ? prm.npoint - j0 : 0) : SA_CPW;
#pragma unroll
    for (int c = 0; c < SA_CPW; ++c) {
        const bool live = c < n_live;
        jrow[c] = live ? j0 + c : -1;
        cnt[c][0] = live ? 0 : prm.nsample[0];                       // a slot past the end starts "full": the sweep never
        cnt[c][1] = live ? 0 : prm.nsample[1];                       // records a hit for it
    }

    int qhead[SA_MAX_SCALES] = {0, 0}, qn[SA_MAX_SCALES] = {0, 0};
    for (int i = lane; i < SA_MAX_SCALES * SA_CPW * SA_OUT; i += 64)
        (&sa_acc[wave][0][0][0])[i] = 0u;                              // post-ReLU values are >= 0
    {
        const int mlp_floats = SA_H1 * C + SA_H1 + SA_H2 * SA_H1 + SA_H2 + SA_OUT * SA_H2 + SA_OUT;
        for (int s = 0; s < prm.n_scales; ++s)
            for (int i = tid; i < mlp_floats; i += SA_WAVES * 64) sa_w[s][i] = prm.mlp[s][i];
    }
    // The boxes of the cloud's 64-point slices (<= 256), as f16 pairs {min, max} per axis, rounded OUTWARD (a larger box
    // keeps the bound a bound): 3 KB in the sweep's tile, which is idle whenever there are groups (its first 1 KB serves
    // the crowded-centroid histogram). One 32-byte load per thread, overlapped with the centroid fetches above.
    uint32_t *sa_sbx = reinterpret_cast<uint32_t *>(&sa_tile[0][0]) + 256;       // [3][256]: axis, slice
    constexpr bool use_slices = SL && NCH == 1;
    // More than 64 groups (the workspace sampler's 128 / 256): the GROUP boxes are staged the same way (f16, outward), so
    // that a centroid's box tests read LDS instead of 24 floats per lane from L2 -- one dependent round trip per centroid.
    const bool stage_group_boxes = NCH > 1 && prm.group_box != nullptr && prm.n_groups <= 256;
    if ((use_slices && prm.slice_box != nullptr) || stage_group_boxes) {   // (groups of a single slice: no slice boxes)
        const int n_slices = stage_group_boxes ? prm.n_groups : prm.n_groups * (prm.group_size / 64);
        if (tid < n_slices) {
            const float4 *sb = reinterpret_cast<const float4 *>((stage_group_boxes ? prm.group_box : prm.slice_box) +
                                                                (bi * n_slices + tid) * 8);
            const float4 b0 = sb[0], b1 = sb[1];                                   // min x y z, max x | max y z
            const float lo3[3] = {b0.x, b0.y, b0.z}, hi3[3] = {b0.w, b1.x, b1.y};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                _Float16 hl = (_Float16)lo3[a], hh = (_Float16)hi3[a];
                uint16_t bl = __builtin_bit_cast(uint16_t, hl), bh = __builtin_bit_cast(uint16_t, hh);
                // one step towards -inf / +inf where the nearest f16 landed inside the box
                if ((float)hl > lo3[a]) bl = (bl & 0x7FFFu) == 0 ? (uint16_t)0x8001u : (bl & 0x8000u) ? (uint16_t)(bl + 1) : (uint16_t)(bl - 1);
                if ((float)hh < hi3[a]) bh = (bh & 0x7FFFu) == 0 ? (uint16_t)0x0001u : (bh & 0x8000u) ? (uint16_t)(bh - 1) : (uint16_t)(bh + 1);
                sa_sbx[a * 256 + tid] = (uint32_t)bl | ((uint32_t)bh << 16);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SA_CPW; ++c)
        if (lane < 4) sa_cxyz[wave][c][lane] = sa_c16[wave * SA_CPW + c][lane];

#ifdef SA_DEBUG
    if (lane == 0) { sa_dbg_l[wave][0] = 0; sa_dbg_l[wave][1] = 0; }
    unsigned long long t_begin, t_fast = 0, t_drain = 0, n_drain = 0, t_sweep = 0;
    unsigned long long t_pre = 0, t_scan = 0, t_rows = 0;      // slice path: pull + box / slice tests; fetch + scan; row writes
    SA_STAMP(t_begin);
#endif
    // rows of the slots in use: the pooled features, the centroid, the counts; then the slots are free again
    auto write_rows = [&]() {
#pragma unroll
        for (int c = 0; c < SA_CPW; ++c) {
            if (jrow[c] < 0) continue;                                // wave-uniform
            float *orow = out_rows + (bi * prm.npoint + jrow[c]) * DCLR_F_STRIDE;
            // lane = s * 32 + channel; columns of an absent scale stay zero
            orow[lane] = (lane >> 5) < prm.n_scales ? __uint_as_float(sa_acc[wave][lane >> 5][c][lane & 31]) : 0.f;
            if (lane < 4) orow[64 + lane] = lane < 3 ? sa_cxyz[wave][c][lane] : 0.f;
            if (counts && lane < prm.n_scales)
                counts[(bi * prm.npoint + jrow[c]) * prm.n_scales + lane] = lane == 0 ? cnt[c][0] : cnt[c][1];
        }
    };
    auto drain_all = [&](bool final_pass) {
#pragma unroll
        for (int s = 0; s < SA_MAX_SCALES; ++s) {
            if (s >= prm.n_scales) break;
            while (qn[s] >= 64 || (final_pass && qn[s] > 0)) {
                const int take = qn[s] < 64 ? qn[s] : 64;
#ifdef SA_DEBUG
                unsigned long long d0, d1;
                SA_STAMP(d0);
#endif
                sa_drain<C, F16>(cloud, wave, s, qhead[s], take);
#ifdef SA_DEBUG
                SA_STAMP(d1);
                t_drain += d1 - d0; n_drain += 1;
#endif
                qhead[s] += take;
                qn[s] -= take;
            }
        }
    };

    uint32_t done = 0;                                     // bit c: slot c was finished on the groups path, counts in sa_tot
    // The slots in use are complete: counts, the published zero-hit behaviour, the last (partial) drains, the rows.
    auto finish_slots = [&]() {
#pragma unroll
        for (int c = 0; c < SA_CPW; ++c)
            if ((done >> c) & 1u) { cnt[c][0] = sa_tot[wave][c][0]; cnt[c][1] = sa_tot[wave][c][1]; }
        // published behaviour for a centroid without any hit: its (zero-filled) index row means point 0
#pragma unroll
        for (int c = 0; c < SA_CPW; ++c)
#pragma unroll
            for (int s = 0; s < SA_MAX_SCALES; ++s) {
                if (s >= prm.n_scales) break;
                if (jrow[c] >= 0 && cnt[c][s] == 0) {
                    if (lane == 0) sa_ring[wave][s][(qhead[s] + qn[s]) & (SA_RING - 1)] = (uint32_t)c << 16;
                    qn[s] += 1;
                }
            }
        drain_all(true);
#ifdef SA_DEBUG
        unsigned long long r0, r1; SA_STAMP(r0);
#endif
        write_rows();
#ifdef SA_DEBUG
        SA_STAMP(r1); t_rows += r1 - r0;
#endif
    };

    // ---- fast path over the sampling kernel's spatial groups ---------------------------------------
    // The waves of the workgroup pull its centroids one at a time; a wave keeps up to SA_CPW of them in its slots
    // (coordinates, counts and running maxima through LDS, tagged by slot in the ring) and finishes the slots -- last
    // drains, rows -- when all are taken and once at the end.
    const bool need_sweep = prm.group_pts == nullptr;
    if (prm.group_pts != nullptr) {
#pragma unroll
        for (int u = 0; u < SA_CPW; ++u) { jrow[u] = -1; cnt[u][0] = 0; cnt[u][1] = 0; }
        const float4 *gp = prm.group_pts + bi * (size_t)prm.n_groups * prm.group_size;
        // lane g owns the boxes of groups g, 64 + g, ... (NCH chunks of 64 groups: 1 for the register sampler's <= 64
        // groups, 2 / 4 for the workspace sampler's 128 / 256)
        // (with one chunk the boxes stay in registers for the wave's four centroids; with 2 / 4 chunks they are re-read per
        // centroid -- 6 L2-resident loads per chunk -- rather than held across the drain calls: 128 registers per wave)
        auto load_box = [&](int ch, float (&bb)[6]) {
            if (NCH > 1 && stage_group_boxes) {               // wave-uniform
                const int g = ch * 64 + lane;
                const bool have = g < prm.n_groups;
                const uint32_t bxw = sa_sbx[have ? g : 0], byw = sa_sbx[256 + (have ? g : 0)], bzw = sa_sbx[512 + (have ? g : 0)];
                auto lo_of = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); };
                auto hi_of = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); };
                bb[0] = have ? lo_of(bxw) : 3.0e38f; bb[1] = have ? lo_of(byw) : 3.0e38f; bb[2] = have ? lo_of(bzw) : 3.0e38f;
                bb[3] = have ? hi_of(bxw) : -3.0e38f; bb[4] = have ? hi_of(byw) : -3.0e38f; bb[5] = have ? hi_of(bzw) : -3.0e38f;
                return;
            }
            const bool have = ch * 64 + lane < prm.n_groups;
            const float *gb = prm.group_box + (bi * prm.n_groups + (have ? ch * 64 + lane : 0)) * 8;
#pragma unroll
            for (int a = 0; a < 3; ++a) { bb[a] = have ? gb[a] : 3.0e38f; bb[3 + a] = have ? gb[3 + a] : -3.0e38f; }
        };
        float box0[6];
        if constexpr (NCH == 1) load_box(0, box0);
        const int slices = prm.group_size / 64;
        int c = 0;                                             // next free slot
#pragma unroll 1
        for (;;) {
#ifdef SA_DEBUG
            unsigned long long p0, p1, p2; SA_STAMP(p0); p1 = p0; p2 = p0;
#endif
            int pulled = 0;
            if (lane == 0) pulled = atomicAdd(&sa_next, 1);
            pulled = __builtin_amdgcn_readfirstlane(pulled);
            if (pulled >= n_wg) break;
            if (c == SA_CPW) {                                 // all slots taken: finish them, start over
                finish_slots();
                for (int i = lane; i < SA_MAX_SCALES * SA_CPW * SA_OUT; i += 64) (&sa_acc[wave][0][0][0])[i] = 0u;
#pragma unroll
                for (int u = 0; u < SA_CPW; ++u) { jrow[u] = -1; cnt[u][0] = 0; cnt[u][1] = 0; }
                done = 0;
                c = 0;
            }
            if (lane < 4) sa_cxyz[wave][c][lane] = sa_c16[pulled][lane];
#pragma unroll
            for (int u = 0; u < SA_CPW; ++u) jrow[u] = u == c ? jw0 + pulled : jrow[u];
            const float cx = sa_c16[pulled][0], cy = sa_c16[pulled][1], cz = sa_c16[pulled][2];
            uint64_t gm[NCH];          // groups the largest ball can reach, per chunk (an absent group's bound is +inf)
            if constexpr (NCH == 1) {
                gm[0] = __ballot(sa_box_lower_bound(box0[0], box0[1], box0[2], box0[3], box0[4], box0[5], cx, cy, cz) <
                                 prm.radius2_max);
            } else {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    float bb[6];
                    load_box(ch, bb);
                    gm[ch] = __ballot(sa_box_lower_bound(bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], cx, cy, cz) < prm.radius2_max);
                }
            }
            // One pass: neighbours are staged for the MLP as they are found (any order) and counted. If a cap
            // turns out to be exceeded -- index order then decides which nsample neighbours count -- or the
            // ring would overflow, the centroid's entries are taken back (nothing of it has been drained:
            // drains happen between centroids only) and it is left to the in-order sweep.
            // Two candidate groups (<= 8 slices of 64 points) per step, all loads issued before the first
            // use: one slice at a time the scan is a chain of ~250 dependent L2 round trips per wave.
            int n1[SA_MAX_SCALES] = {0, 0};
            const int q0[SA_MAX_SCALES] = {qn[0], qn[1]};
            auto load_pair = [&](uint64_t &m, int g0, float4 (&q)[8], int &nq) {
                const int ga = g0 + __builtin_ctzll(m);
                m &= m - 1;
                const bool two = m != 0;
                const int gbb = two ? g0 + __builtin_ctzll(m) : ga;
                if (two) m &= m - 1;
                const float4 *pa = gp + (size_t)ga * prm.group_size + lane;
                const float4 *pb = gp + (size_t)gbb * prm.group_size + lane;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ic = it < slices ? it : slices - 1;           // clamped: loads stay unconditional
                    q[it] = pa[ic * 64];
                    q[4 + it] = pb[ic * 64];
                }
                nq = two ? 8 : 4;
            };
            bool over = false;
            int ch = 0;
            // One loaded slice of 64 candidate points against the centroid: hits are counted and staged (any order)
            auto scan_slice = [&](const float4 &qq) {
                // padding slots of a group hold x = y = z = 3e38: their distance is +inf, never a hit
                const float d2 = dclr_sqdist(cx, cy, cz, qq.x, qq.y, qq.z);
                if (__ballot(d2 < prm.radius2_max) == 0) return;           // most slices: nothing inside the largest ball
#pragma unroll
                for (int s = 0; s < SA_MAX_SCALES; ++s) {
                    if (s >= prm.n_scales) break;
                    const bool hit = d2 < prm.radius2[s];
                    const uint64_t mask = __ballot(hit);
                    if (mask != 0) {
                        const int add = __builtin_popcountll(mask);
                        n1[s] += add;
                        if (qn[s] + 64 > SA_RING) {            // a slice adds <= 64 entries: never onto live ones
                            over = true;                       // crowded centroid (> ~450 neighbours): in-order sweep
                        } else {
                            const int pre = (int)dclr_lanemask_lt_popc(mask);
                            if (hit)
                                sa_ring[wave][s][(qhead[s] + qn[s] + pre) & (SA_RING - 1)] =
                                    ((uint32_t)c << 16) | (__float_as_uint(qq.w) & 0xFFFFu);
                            qn[s] += add;
                        }
                    }
                }
            };
            if constexpr (use_slices) {
                {
                    // Slice level: the reachable groups' slices are tested against the ball (lane 4 h + i: slice i of the
                    // h-th reachable group, boxes from LDS) and only slices that can hold a neighbour are fetched, six at
                    // a time whatever groups they belong to -- on the bench clouds 5.8 slices per centroid instead of the
                    // 20 of its 5 reachable groups: one dependent fetch step instead of three.
                    // (groups of ONE slice -- clouds of up to 4096 points -- skip the slice test: lane g = group g, and the
                    // reachable groups are fetched six at a time instead of two)
                    const int lps = slices == 4 ? 2 : slices == 2 ? 1 : 0;   // log2(slices per group)
                    for (uint64_t mg = gm[0]; mg != 0 && !over;) {
                        int myg = -1;
                        bool shit = false;
                        if (lps == 0) {                                     // wave-uniform
                            myg = lane;
                            shit = ((mg >> lane) & 1ull) != 0;
                            mg = 0;
                        } else {
                            const int hh = lane >> lps, per = 64 >> lps;   // groups per round of 64 lanes
                            for (int h = 0; h < per && mg != 0; ++h) {
                                const int g = __builtin_ctzll(mg);
                                mg &= mg - 1;
                                myg = hh == h ? g : myg;
                            }
                        }
                        if (lps != 0 && myg >= 0) {
                            const int sl = (myg << lps) | (lane & (slices - 1));
                            const uint32_t bxw = sa_sbx[sl], byw = sa_sbx[256 + sl], bzw = sa_sbx[512 + sl];
                            auto lo_of = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu)); };
                            auto hi_of = [](uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); };
                            shit = sa_box_lower_bound(lo_of(bxw), lo_of(byw), lo_of(bzw), hi_of(bxw), hi_of(byw), hi_of(bzw),
                                                      cx, cy, cz) < prm.radius2_max;
                        }
#ifdef SA_DEBUG
                        SA_STAMP(p1);
#endif
                        for (uint64_t sm = __ballot(shit); sm != 0 && !over;) {
                            float4 q[SA_SLICE_STEP];
                            int nq = 0;
#pragma unroll
                            for (int u = 0; u < SA_SLICE_STEP; ++u) {
                                if (sm != 0) {                             // wave-uniform
                                    const int L = __builtin_ctzll(sm);
                                    sm &= sm - 1;
                                    const int g = __builtin_amdgcn_readlane(myg, L);
                                    q[u] = gp[(size_t)g * prm.group_size + (L & (slices - 1)) * 64 + lane];
                                    nq = u + 1;
                                }
                            }
#pragma unroll
                            for (int u = 0; u < SA_SLICE_STEP; ++u)
                                if (u < nq) scan_slice(q[u]);
                        }
                    }
                }
            }
#ifdef SA_DEBUG
            SA_STAMP(p2);
            if (use_slices) { t_pre += p1 - p0; t_scan += p2 - p1; }
#endif
            for (uint64_t m = use_slices ? 0ull : gm[0]; !over;) {
                if constexpr (NCH > 1) {
                    while (m == 0 && ch + 1 < NCH) {           // next chunk of 64 groups (wave-uniform)
                        ++ch;
                        m = ch == 1 ? gm[1] : ch == 2 ? gm[NCH > 2 ? 2 : 0] : gm[NCH > 3 ? 3 : 0];
                    }
                }
                if (m == 0) break;
                float4 q[8];
                int nq;
                load_pair(m, ch * 64, q, nq);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if ((u & 3) >= slices || u >= nq) continue;            // clamped duplicate of another slice (uniform)
                    // padding slots of a group hold x = y = z = 3e38: their distance is +inf, never a hit
                    const float d2 = dclr_sqdist(cx, cy, cz, q[u].x, q[u].y, q[u].z);
                    if (__ballot(d2 < prm.radius2_max) == 0) continue;     // most slices: nothing inside the largest ball
#pragma unroll
                    for (int s = 0; s < SA_MAX_SCALES; ++s) {
                        if (s >= prm.n_scales) break;
                        const bool hit = d2 < prm.radius2[s];
                        const uint64_t mask = __ballot(hit);
                        if (mask != 0) {
                            const int add = __builtin_popcountll(mask);
                            n1[s] += add;
                            if (qn[s] + 64 > SA_RING) {        // a slice adds <= 64 entries: never onto live ones
                                over = true;                   // crowded centroid (> ~450 neighbours): in-order sweep
                            } else {
                                const int pre = (int)dclr_lanemask_lt_popc(mask);
                                if (hit)
                                    sa_ring[wave][s][(qhead[s] + qn[s] + pre) & (SA_RING - 1)] =
                                        ((uint32_t)c << 16) | (__float_as_uint(q[u].w) & 0xFFFFu);
                                qn[s] += add;
                            }
                        }
                    }
                }
            }
            if (over || n1[0] > prm.nsample[0] || (prm.n_scales > 1 && n1[1] > prm.nsample[1])) {
                // Crowded centroid: the ring would overflow (> ~450 neighbours) or a cap is exceeded, and then INDEX order
                // decides which nsample neighbours count. Its entries are taken back (none has been drained) and the
                // centroid goes on the workgroup's list: once the other centroids are done, the four waves redo it TOGETHER.
                qn[0] = q0[0]; qn[1] = q0[1];
                if (lane == 0) sa_crowd[atomicAdd(&sa_ncrowd, 1)] = pulled;
#pragma unroll
                for (int u = 0; u < SA_CPW; ++u) jrow[u] = u == c ? -1 : jrow[u];
                continue;                                      // the slot is free again
            }
            if (qn[0] >= 64 || qn[1] >= 64) drain_all(false);
            done |= 1u << c;
            if (lane == 0) { sa_tot[wave][c][0] = n1[0]; sa_tot[wave][c][1] = n1[1]; }
            ++c;
        }
        // ---- crowded centroids, the four waves together ------------------------------------------------------------
        // Every hit of such a centroid lies in its candidate groups (any order); the waves split those groups
        // (every fourth one each) and
        //   1. count the hits per scale (shared counters);
        //   2. per scale over its cap find the nsample-th smallest point index T among the hits by a two-level radix
        //      select on the 16-bit index (one shared 256-bin histogram per level: the point indices of a cloud are
        //      distinct, so the second level lands on T exactly);
        //   3. stage their share of the hits with index <= T in their own ring (slot 0), draining whenever it fills;
        //   4. wave 0 folds the four partial maxima and writes the row.
        // Exact, and one crowded centroid costs its workgroup a quarter of what it cost the wave that drew it (the
        // LiDAR near field: 2-6 % of the centroids, each worth 10-20 ordinary ones). Round 2 sent the whole workgroup
        // through all N points instead.
#ifdef SA_DEBUG
        unsigned long long f0, f1, f2; SA_STAMP(f0);
#endif
        finish_slots();                                        // the ordinary centroids of this wave: last drains, rows
        done = 0;
#pragma unroll
        for (int u = 0; u < SA_CPW; ++u) { jrow[u] = -1; cnt[u][0] = 0; cnt[u][1] = 0; }
#ifdef SA_DEBUG
        SA_STAMP(f1);
#endif
        __syncthreads();
#ifdef SA_DEBUG
        SA_STAMP(f2);
        if (lane == 0) { sa_dbg_l[wave][0] = f1 - f0; sa_dbg_l[wave][1] = f2 - f1; }       // last finish, wait for the other waves
#endif
        const int n_crowd = sa_ncrowd;                         // the same in every wave from here on
        uint32_t *shist = reinterpret_cast<uint32_t *>(&sa_tile[0][0]);       // 256 shared bins (the sweep's tile is idle here)
#pragma unroll 1
        for (int kc = 0; kc < n_crowd; ++kc) {
            const int pc = sa_crowd[kc];
            const float cx = sa_c16[pc][0], cy = sa_c16[pc][1], cz = sa_c16[pc][2];
            uint64_t gm[NCH];
            if constexpr (NCH == 1) {
                gm[0] = __ballot(sa_box_lower_bound(box0[0], box0[1], box0[2], box0[3], box0[4], box0[5], cx, cy, cz) <
                                 prm.radius2_max);
            } else {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    float bb[6];
                    load_box(ch, bb);
                    gm[ch] = __ballot(sa_box_lower_bound(bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], cx, cy, cz) < prm.radius2_max);
                }
            }
            // this wave's share of the candidate groups: every SA_WAVES-th one, then only those in which a scan found a hit
            {
                int turn = 0;
#pragma unroll
                for (int chn = 0; chn < NCH; ++chn) {
                    uint64_t keep = 0;
                    for (uint64_t mm = gm[chn]; mm != 0; mm &= mm - 1) {
                        if ((turn++ & (SA_WAVES - 1)) == wave) keep |= mm & (0 - mm);
                    }
                    gm[chn] = keep;
                }
            }
            uint64_t ghit[NCH];
            auto scan = [&](auto &&per_slice) {
#pragma unroll 1
                for (int chn = 0; chn < NCH; ++chn) {
                    ghit[chn] = 0;
#pragma unroll 1
                    for (uint64_t mm = gm[chn]; mm != 0; mm &= mm - 1) {
                        const float4 *pg = gp + (size_t)(chn * 64 + __builtin_ctzll(mm)) * prm.group_size + lane;
                        constexpr int SB = NCH >= 4 ? 2 : 4;           // slices per round trip (register budget: 128 per wave)
#pragma unroll 1
                        for (int it0 = 0; it0 < slices; it0 += SB) {
                            float4 qq[SB];
#pragma unroll
                            for (int it = 0; it < SB; ++it) qq[it] = pg[(it0 + it < slices ? it0 + it : slices - 1) * 64];
#pragma unroll
                            for (int it = 0; it < SB; ++it) {
                                if (it0 + it >= slices) break;         // wave-uniform
                                const float d2 = dclr_sqdist(cx, cy, cz, qq[it].x, qq[it].y, qq[it].z);
                                if (__ballot(d2 < prm.radius2_max) == 0) continue;
                                ghit[chn] |= mm & (0 - mm);            // lowest set bit = this group
                                per_slice(d2, __float_as_uint(qq[it].w) & 0xFFFFu);
                            }
                        }
                    }
                }
            };
            // 1. counts
            if (tid < SA_MAX_SCALES) sa_cnt[tid] = 0;
            __syncthreads();
            {
                int n0 = 0, n1c = 0;
                scan([&](float d2, uint32_t) {
                    n0 += __builtin_popcountll(__ballot(d2 < prm.radius2[0]));
                    if (prm.n_scales > 1) n1c += __builtin_popcountll(__ballot(d2 < prm.radius2[1]));
                });
                if (lane == 0) { atomicAdd(&sa_cnt[0], n0); atomicAdd(&sa_cnt[1], n1c); }
#pragma unroll
                for (int chn = 0; chn < NCH; ++chn) gm[chn] = ghit[chn];      // later passes: only the groups that hold a hit
            }
            __syncthreads();
            int tot[SA_MAX_SCALES] = {sa_cnt[0], sa_cnt[1]};
            // 2. thresholds
            uint32_t thr[SA_MAX_SCALES] = {0xFFFFu, 0xFFFFu};
#pragma unroll 1
            for (int s = 0; s < prm.n_scales; ++s) {
                if (tot[s] <= prm.nsample[s]) continue;                       // the same in every wave
                uint32_t prefix = 0;                     // high byte of T once known
                int need = prm.nsample[s];               // rank of T among the hits still in play (1-based)
#pragma unroll 1
                for (int level = 0; level < 2; ++level) {
                    shist[tid] = 0u;                                          // 256 threads, 256 bins
                    __syncthreads();
                    const float r2 = prm.radius2[s];
                    scan([&](float d2, uint32_t k) {
                        const bool in = d2 < r2 && (level == 0 || (k >> 8) == prefix);
                        if (in) atomicAdd(&shist[level == 0 ? (k >> 8) : (k & 255u)], 1u);
                    });
                    __syncthreads();
                    // every wave alike: lane l owns bins 4 l .. 4 l + 3; exclusive prefix over the lanes, then inside the lane
                    uint32_t cb[4], mine = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { cb[u] = shist[4 * lane + u]; mine += cb[u]; }
                    uint32_t incl = mine;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const uint32_t up = __shfl_up(incl, off);
                        if (lane >= off) incl += up;
                    }
                    const uint32_t excl = incl - mine;
                    const int wl = __builtin_ctzll(__ballot(excl < (uint32_t)need && (uint32_t)need <= incl));
                    uint32_t run = excl, bin = 0, before = 0;
                    bool found = false;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!found && run + cb[u] >= (uint32_t)need) { bin = 4 * lane + u; before = run; found = true; }
                        run += cb[u];
                    }
                    const uint32_t wbin = (uint32_t)__builtin_amdgcn_readlane((int)bin, wl);
                    need -= (int)__builtin_amdgcn_readlane((int)before, wl);
                    if (level == 0) prefix = wbin; else thr[s] = (prefix << 8) | wbin;
                    __syncthreads();                                          // every wave has read the bins
                }
                tot[s] = prm.nsample[s];
            }
            // 3. this wave's share of the hits, slot 0
            for (int i = lane; i < SA_MAX_SCALES * SA_CPW * SA_OUT; i += 64) (&sa_acc[wave][0][0][0])[i] = 0u;
            if (lane < 4) sa_cxyz[wave][0][lane] = sa_c16[pc][lane];
            scan([&](float d2, uint32_t k) {
#pragma unroll
                for (int s = 0; s < SA_MAX_SCALES; ++s) {
                    if (s >= prm.n_scales) break;
                    const bool hit = d2 < prm.radius2[s] && k <= thr[s];
                    const uint64_t mask = __ballot(hit);
                    if (mask != 0) {
                        if (qn[s] + 64 > SA_RING) drain_all(false);               // leaves < 64 entries in each ring
                        const int pre = (int)dclr_lanemask_lt_popc(mask);
                        if (hit) sa_ring[wave][s][(qhead[s] + qn[s] + pre) & (SA_RING - 1)] = k;      // slot 0
                        qn[s] += __builtin_popcountll(mask);
                    }
                }
            });
            drain_all(true);
            __syncthreads();
            // 4. the row
            if (wave == 0) {
                float *orow = out_rows + (bi * prm.npoint + jw0 + pc) * DCLR_F_STRIDE;
                uint32_t v = 0u;
                if ((lane >> 5) < prm.n_scales) {
#pragma unroll
                    for (int w = 0; w < SA_WAVES; ++w) v = dclr_umax(v, sa_acc[w][lane >> 5][0][lane & 31]);
                }
                orow[lane] = __uint_as_float(v);
                if (lane < 4) orow[64 + lane] = lane < 3 ? sa_c16[pc][lane] : 0.f;
                if (counts && lane < prm.n_scales)
                    counts[(bi * prm.npoint + jw0 + pc) * prm.n_scales + lane] = lane == 0 ? tot[0] : tot[1];
            }
            __syncthreads();                                                  // slot 0 and the counters are free again
        }
    }

#ifdef SA_DEBUG
    { unsigned long long t1; SA_STAMP(t1); t_fast = t1 - t_begin; }
#endif
    // ---- exhaustive in-order sweep: calls without the sampler's groups (n <= 1024 or no grouped kernel) -----
    if (need_sweep) {                                      // the same for every wave of the grid (a launch argument)
#ifdef SA_DEBUG
        unsigned long long w0; SA_STAMP(w0);
#endif
        float ccx[SA_CPW], ccy[SA_CPW], ccz[SA_CPW];                    // this wave's centroids (wave-uniform)
#pragma unroll
        for (int c = 0; c < SA_CPW; ++c) {
            const int src = wave * SA_CPW + c;
            ccx[c] = sa_c16[src][0]; ccy[c] = sa_c16[src][1]; ccz[c] = sa_c16[src][2];
        }
        const int n_tiles = (prm.n + SA_TILE - 1) / SA_TILE;
        constexpr int PER_THREAD = SA_TILE / (SA_WAVES * 64);           // points staged per thread
        float4 stage[PER_THREAD];
        auto fetch = [&](int t) {
#pragma unroll
            for (int u = 0; u < PER_THREAD; ++u) {
                const int k = t * SA_TILE + u * (SA_WAVES * 64) + tid;
                stage[u] = k < prm.n ? sa_load_point<C>(cloud, k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto stash = [&](int buf) {
#pragma unroll
            for (int u = 0; u < PER_THREAD; ++u) sa_tile[buf][u * (SA_WAVES * 64) + tid] = stage[u];
        };
        fetch(0);
        stash(0);
        for (int t = 0; t < n_tiles; ++t) {
            __syncthreads();                               // tile t is in sa_tile
            const int buf = 0;
            if (t + 1 < n_tiles) fetch(t + 1);
            for (int it = 0; it < SA_TILE / 64; ++it) {
                const int k = t * SA_TILE + it * 64 + lane;
                const float4 p = sa_tile[buf][it * 64 + lane];
                const bool inb = k < prm.n;
                float d2c[SA_CPW];
#pragma unroll
                for (int c = 0; c < SA_CPW; ++c) d2c[c] = dclr_sqdist(ccx[c], ccy[c], ccz[c], p.x, p.y, p.z);
                // common case: no lane is inside the largest ball of any of the wave's centroids
                float dmin = d2c[0];
#pragma unroll
                for (int c = 1; c < SA_CPW; ++c) dmin = fminf(dmin, d2c[c]);
                if (__ballot(inb && dmin < prm.radius2_max) == 0) continue;
#pragma unroll
                for (int c = 0; c < SA_CPW; ++c) {
                    const float d2 = d2c[c];
#pragma unroll
                    for (int s = 0; s < SA_MAX_SCALES; ++s) {
                        if (s >= prm.n_scales) break;
                        const bool hit = inb && d2 < prm.radius2[s];
                        const uint64_t mask = __ballot(hit);
                        if (__builtin_expect(mask != 0 && cnt[c][s] < prm.nsample[s], 0)) {   // wave-uniform, rare
                            const int room = prm.nsample[s] - cnt[c][s];
                            const int pre = (int)dclr_lanemask_lt_popc(mask);
                            int nt = __builtin_popcountll(mask);
                            nt = nt < room ? nt : room;
                            if (hit && pre < room)
                                sa_ring[wave][s][(qhead[s] + qn[s] + pre) & (SA_RING - 1)] = ((uint32_t)c << 16) | (uint32_t)k;
                            qn[s] += nt;
                            cnt[c][s] += nt;
                        }
                    }
                    if ((c & 3) == 3 && __builtin_expect(qn[0] >= 64 || qn[1] >= 64, 0)) drain_all(false);
                }
            }
            __syncthreads();                               // every wave is done with tile t
            if (t + 1 < n_tiles) stash(0);
        }
#ifdef SA_DEBUG
        { unsigned long long w1; SA_STAMP(w1); t_sweep = w1 - w0; }
#endif
    }
    finish_slots();
#ifdef SA_DEBUG
    if (lane == 0) {
        unsigned long long t_end; SA_STAMP(t_end);
        unsigned long long *o = sa_dbg_w[(blockIdx.y * gridDim.x + blockIdx.x) * SA_WAVES + wave];
        o[0] = t_end - t_begin; o[1] = t_fast; o[2] = t_drain; o[3] = n_drain; o[4] = t_sweep; o[5] = 1ull;
        o[6] = sa_dbg_l[wave][0]; o[7] = sa_dbg_l[wave][1];
        if (t_pre != 0 && getenv_dbg2) { o[4] = t_rows; o[6] = t_pre; o[7] = t_scan; }  // slice path: reported in place of sweep / drain detail
    }
#endif
}

__global__ __launch_bounds__(256) void rows_to_channels_kernel(int npoint, int nfeat, int xyz_col, int stride,
                                                               const float *__restrict__ rows,
                                                               float *__restrict__ channels) {
    // channels (b, 3 + nfeat, npoint): channel 0..2 = xyz (row columns xyz_col..+2), then features
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y;
    const size_t bi = blockIdx.z;
    if (p >= npoint) return;
    const int col = ch < 3 ? xyz_col + ch : ch - 3;
    channels[(bi * (3 + nfeat) + ch) * npoint + p] = rows[(bi * npoint + p) * stride + col];
}

__global__ __launch_bounds__(256) void channels_to_rows_kernel(int npoint, int nfeat, int xyz_col, int stride,
                                                               const float *__restrict__ channels,
                                                               float *__restrict__ rows) {
    const int col = blockIdx.x * 256 + threadIdx.x;      // one thread per row element
    const int p = blockIdx.y;
    const size_t bi = blockIdx.z;
    if (col >= stride) return;
    float v = 0.f;
    if (col < nfeat) v = channels[(bi * (3 + nfeat) + 3 + col) * npoint + p];
    else if (col >= xyz_col && col < xyz_col + 3) v = channels[(bi * (3 + nfeat) + (col - xyz_col)) * npoint + p];
    rows[(bi * npoint + p) * stride + col] = v;
}

}  // namespace

static int sa_launch(bool f16, int b, int n, int c, int npoint, const float *clouds, const int32_t *fps_idx, int n_scales,
                     const float *radii_host, const int *nsamples_host, const float *const *mlp_host_ptrs, float *out_rows,
                     int32_t *counts, const float *group_pts, const float *group_box, dclr_stream_t stream,
                     DclrCloudView view = DclrCloudView{0, 1, 0}, const float *slice_box = nullptr, uint32_t *overflow = nullptr) {
    DCLR_REQUIRE(b > 0 && n > 0 && npoint > 0 && clouds && fps_idx && radii_host && nsamples_host &&
                 mlp_host_ptrs && out_rows && b <= 65535);
    if (n_scales < 1 || n_scales > SA_MAX_SCALES || (c != 3 && c != 4)) return DCLR_E_UNSUPPORTED;
    if (n > 65536) return DCLR_E_UNSUPPORTED;                 // ring entries hold 16-bit point indices
    SaParams prm{};
    prm.n = n; prm.npoint = npoint; prm.n_scales = n_scales;
    prm.view = view;
    prm.overflow = overflow;
    for (int s = 0; s < n_scales; ++s) {
        DCLR_REQUIRE(nsamples_host[s] > 0 && mlp_host_ptrs[s]);
        prm.radius2[s] = radii_host[s] * radii_host[s];
        prm.radius2_max = s == 0 || prm.radius2[s] > prm.radius2_max ? prm.radius2[s] : prm.radius2_max;
        prm.nsample[s] = nsamples_host[s];
        prm.mlp[s] = mlp_host_ptrs[s];
    }
    if (group_pts || group_box) {
        DCLR_REQUIRE(group_pts && group_box && ((uintptr_t)group_pts & 15) == 0);
        if (dclr_fps_group_layout(n, &prm.n_groups, &prm.group_size) != DCLR_OK || prm.n_groups > 256) return DCLR_E_INVALID;
        prm.group_pts = reinterpret_cast<const float4 *>(group_pts);
        prm.group_box = group_box;
        if (slice_box) {
            // slice boxes: <= 64 groups of 128 or 256 points (<= 256 slices: one per thread, 3 KB of LDS)
            if (prm.n_groups > 64 || (prm.group_size != 128 && prm.group_size != 256)) return DCLR_E_UNSUPPORTED;
            DCLR_REQUIRE(((uintptr_t)slice_box & 15) == 0);
            prm.slice_box = slice_box;
        }
    } else {
        DCLR_REQUIRE(slice_box == nullptr);
    }
    constexpr int per_wg = SA_WAVES * SA_CPW;
    dim3 grid((npoint + per_wg - 1) / per_wg, b);
    const int nch = prm.n_groups <= 64 ? 1 : 4;      // chunks of 64 group boxes per lane (128 groups: two of the four stay empty)
    // slice-granular fetches: with slice boxes, or where a group IS one slice (A/B: DCLR_SA_PAIRS=1 keeps two groups per step)
    static const bool pairs_only = getenv("DCLR_SA_PAIRS") != nullptr;
    const bool slice_path = prm.slice_box != nullptr || (prm.group_pts && nch == 1 && prm.group_size == 64 && !pairs_only);
#define SA_LAUNCH(C_, NCH_, F_, SL_)                                                                                     \
    hipLaunchKernelGGL((sa_msg_kernel<C_, NCH_, F_, SL_>), grid, dim3(SA_WAVES * 64), 0, (hipStream_t)stream, prm, clouds, \
                       fps_idx, out_rows, counts)
#define SA_LAUNCH_C(C_)                                                                                                  \
    do {                                                                                                                 \
        if (slice_path) { if (f16) SA_LAUNCH(C_, 1, true, true); else SA_LAUNCH(C_, 1, false, true); }                   \
        else if (f16) { if (nch == 1) SA_LAUNCH(C_, 1, true, false); else SA_LAUNCH(C_, 4, true, false); }               \
        else          { if (nch == 1) SA_LAUNCH(C_, 1, false, false); else SA_LAUNCH(C_, 4, false, false); }             \
    } while (0)
    if (c == 4) SA_LAUNCH_C(4); else SA_LAUNCH_C(3);
#undef SA_LAUNCH_C
#undef SA_LAUNCH
    return dclr_launch_status();
}

extern "C" int dclr_sa_msg_fused(int b, int n, int c, int npoint, const float *clouds,
                                 const int32_t *fps_idx, int n_scales, const float *radii_host,
                                 const int *nsamples_host, const float *const *mlp_host_ptrs,
                                 float *out_rows, int32_t *counts, const float *group_pts, const float *group_box,
                                 dclr_stream_t stream) {
    return sa_launch(false, b, n, c, npoint, clouds, fps_idx, n_scales, radii_host, nsamples_host, mlp_host_ptrs, out_rows,
                     counts, group_pts, group_box, stream);
}

extern "C" int dclr_sa_msg_fused_f16(int b, int n, int c, int npoint, const float *clouds,
                                     const int32_t *fps_idx, int n_scales, const float *radii_host,
                                     const int *nsamples_host, const float *const *mlp_host_ptrs,
                                     float *out_rows, int32_t *counts, const float *group_pts, const float *group_box,
                                     dclr_stream_t stream) {
    return sa_launch(true, b, n, c, npoint, clouds, fps_idx, n_scales, radii_host, nsamples_host, mlp_host_ptrs, out_rows,
                     counts, group_pts, group_box, stream);
}

extern "C" int dclr_sa_msg_fused_batched(int f16, int b, int n, int c, int npoint, const float *clouds, int pairs_per_batch,
                                         int n_batches, long long batch_stride, const int32_t *fps_idx, int n_scales,
                                         const float *radii_host, const int *nsamples_host,
                                         const float *const *mlp_host_ptrs, float *out_rows, int32_t *counts,
                                         const float *group_pts, const float *group_box, const float *slice_box,
                                         dclr_stream_t stream) {
    return dclr_sa_msg_fused_batched_ov(f16, b, n, c, npoint, clouds, pairs_per_batch, n_batches, batch_stride, fps_idx, n_scales,
                                        radii_host, nsamples_host, mlp_host_ptrs, out_rows, counts, group_pts, group_box,
                                        slice_box, nullptr, stream);
}

extern "C" int dclr_sa_msg_fused_batched_ov(int f16, int b, int n, int c, int npoint, const float *clouds,
                                            int pairs_per_batch, int n_batches, long long batch_stride, const int32_t *fps_idx,
                                            int n_scales, const float *radii_host, const int *nsamples_host,
                                            const float *const *mlp_host_ptrs, float *out_rows, int32_t *counts,
                                            const float *group_pts, const float *group_box, const float *slice_box,
                                            uint32_t *overflow, dclr_stream_t stream) {
    DCLR_REQUIRE(pairs_per_batch > 0 && n_batches > 0 && batch_stride >= 0 && b == 2 * pairs_per_batch * n_batches);
    return sa_launch(f16 != 0, b, n, c, npoint, clouds, fps_idx, n_scales, radii_host, nsamples_host, mlp_host_ptrs, out_rows,
                     counts, group_pts, group_box, stream, DclrCloudView{pairs_per_batch, n_batches, batch_stride}, slice_box,
                     overflow);
}

extern "C" int dclr_rows_to_channels(int b, int npoint, int nfeat, int xyz_col, int stride, const float *rows,
                                     float *channels, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && npoint > 0 && nfeat > 0 && xyz_col >= nfeat && stride >= xyz_col + 3 && rows &&
                 channels && b <= 65535);
    hipLaunchKernelGGL(rows_to_channels_kernel, dim3((npoint + 255) / 256, 3 + nfeat, b), dim3(256), 0,
                       (hipStream_t)stream, npoint, nfeat, xyz_col, stride, rows, channels);
    return dclr_launch_status();
}

extern "C" int dclr_channels_to_rows(int b, int npoint, int nfeat, int xyz_col, int stride, const float *channels,
                                     float *rows, dclr_stream_t stream) {
    DCLR_REQUIRE(b > 0 && npoint > 0 && nfeat > 0 && xyz_col >= nfeat && stride >= xyz_col + 3 && rows &&
                 channels && b <= 65535);
    DCLR_REQUIRE(npoint <= 65535);
    hipLaunchKernelGGL(channels_to_rows_kernel, dim3((stride + 255) / 256, npoint, b), dim3(256), 0,
                       (hipStream_t)stream, npoint, nfeat, xyz_col, stride, channels, rows);
    return dclr_launch_status();
}
