/*
 * oracle/primitives.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement, in plain C, of the third-party primitives DeepCLR's hot path
 * calls and whose sources are ABSENT from /root/reference:
 *
 *   - sshaoshuai/Pointnet2.PyTorch (un-vendored submodule, commit unpinned:
 *     /root/reference/.gitmodules:1-4; only the C++ wrapper signatures are in
 *     tree, /root/reference/extern/pointnet2.patch:101-116,160-174,275-288,306-320)
 *   - rusty1s/pytorch_cluster, torch-cluster==1.5.9
 *     (/root/reference/requirements.txt:24; call site
 *     /root/reference/deepclr/models/deepclr.py:164-166)
 *
 * PARITY UNPINNED at this level: the reference tree holds no kernel body, no
 * golden vector and no known-answer test for any of these functions
 * (SURVEY.md section 8c). What follows restates the *published* algorithm of
 * those libraries' GPU kernels (one logical GPU thread == one loop iteration
 * here), including their tie rules. The only in-tree cross-check is the numpy
 * FPS in /root/reference/deepclr/data/transforms/transforms.py:47-59, which
 * tests/test_oracle.py compares against on tie-free data.
 *
 * Frozen floating-point recipe (both here and in the HIP kernels):
 *   d = (dx*dx + dy*dy) + dz*dz   in IEEE binary32, one rounding per operation,
 *   NO fused multiply-add.  Build with -ffp-contract=off (see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    /* (a-b) per component; squares summed left to right; no contraction. */
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

/* Largest power of two <= work_size, clamped to [1, 1024] -- the published
 * launch rule ("opt_n_threads") that fixes the FPS tie order. */
int dclr_oracle_fps_block(int n) {
    int t = 1;
    while (t * 2 <= n && t * 2 <= 1024) t *= 2;
    return t;
}

/*
 * Furthest point sampling. Signature follows
 * furthest_point_sampling_wrapper(b, n, m, points, temp, idx)
 * (/root/reference/extern/pointnet2.patch:306-320): caller pre-fills temp
 * with 1e10 and owns idx (b, m) int32.
 *
 * Published kernel, restated: T = dclr_oracle_fps_block(n) logical threads per
 * cloud; idx[0] = 0; for every further sample each thread tid walks
 * k = tid, tid+T, ... updating temp[k] = min(d(k, last), temp[k]) and keeps its
 * first strict maximum (best starts at -1, besti at 0); the T candidates are
 * merged by a halving tree in which slot t absorbs slot t+s and the lower slot
 * keeps ties (v2 > v1 ? i2 : i1).
 */
void dclr_oracle_fps(int b, int n, int m, const float *points, float *temp, int32_t *idx) {
    if (m <= 0 || n <= 0) return;
    const int T = dclr_oracle_fps_block(n);
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < b; ++bi) {
        const float *p = points + (size_t)bi * n * 3;
        float *tp = temp + (size_t)bi * n;
        int32_t *out = idx + (size_t)bi * m;
        float *dists = (float *)malloc(sizeof(float) * T);
        int *dists_i = (int *)malloc(sizeof(int) * T);
        int old = 0;
        out[0] = 0;
        for (int j = 1; j < m; ++j) {
            const float x1 = p[old * 3 + 0], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
            for (int tid = 0; tid < T; ++tid) {
                int besti = 0;
                float best = -1.0f;
                for (int k = tid; k < n; k += T) {
                    float d = sqdist3(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], x1, y1, z1);
                    float d2 = d < tp[k] ? d : tp[k];
                    tp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = T / 2; s >= 1; s /= 2) {
                for (int tid = 0; tid < s; ++tid) {
                    float v1 = dists[tid], v2 = dists[tid + s];
                    int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2;
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            out[j] = old;
        }
        free(dists);
        free(dists_i);
    }
}

/* gather_points_wrapper_fast(b, c, n, npoints, points(b,c,n), idx(b,npoints), out(b,c,npoints))
 * (/root/reference/extern/pointnet2.patch:275-288): out[b,c,j] = points[b,c,idx[b,j]]. */
void dclr_oracle_gather_points(int b, int c, int n, int npoints, const float *points,
                               const int32_t *idx, float *out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int j = 0; j < npoints; ++j)
                out[((size_t)bi * c + ci) * npoints + j] =
                    points[((size_t)bi * c + ci) * n + idx[(size_t)bi * npoints + j]];
}

/*
 * ball_query_wrapper_fast(b, n, m, radius, nsample, new_xyz(b,m,3), xyz(b,n,3), idx(b,m,nsample))
 * (/root/reference/extern/pointnet2.patch:101-116). idx arrives zero-filled.
 * Published kernel, restated: per centroid walk k = 0..n-1; a point is a hit
 * when d2 < radius*radius (radius squared in binary32); the first hit fills
 * every slot, later hits overwrite slot cnt; stop at nsample hits.
 */
void dclr_oracle_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                            const float *xyz, int32_t *idx) {
    const float radius2 = radius * radius;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int j = 0; j < m; ++j) {
            const float *c = new_xyz + ((size_t)bi * m + j) * 3;
            const float *p = xyz + (size_t)bi * n * 3;
            int32_t *o = idx + ((size_t)bi * m + j) * nsample;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                float d2 = sqdist3(c[0], c[1], c[2], p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2]);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    }
}

/* group_points_wrapper_fast(b, c, n, npoints, nsample, points(b,c,n), idx(b,npoints,nsample),
 * out(b,c,npoints,nsample)) (/root/reference/extern/pointnet2.patch:160-174). */
void dclr_oracle_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                              const int32_t *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *src = points + ((size_t)bi * c + ci) * n;
            float *dst = out + ((size_t)bi * c + ci) * npoints * nsample;
            const int32_t *ix = idx + (size_t)bi * npoints * nsample;
            for (size_t e = 0; e < (size_t)npoints * nsample; ++e) dst[e] = src[ix[e]];
        }
}

/*
 * torch_cluster.knn(x, y, k, batch_x, batch_y) for equally sized, sorted
 * batches -- the only form DeepCLR uses (deepclr.py:149-155,164-166).
 * x: (b*nx, 3) candidates, y: (b*ny, 3) queries. Published 1.5.9 GPU kernel,
 * restated: per query, distance accumulates from 0 over the 3 dims of (x - y);
 * a k-slot list (init 1e10 / -1, as upstream's torch::full) is kept ascending by insertion with a strict
 * "slot > new" test, so equal distances keep the lower candidate index first and a
 * candidate at squared distance 1e10 or beyond (or NaN) is never inserted.
 * row/col are (b*ny*k) int64 with GLOBAL (flattened) indices; unfilled slots
 * are -1 (the Python side drops them, which DeepCLR's .view(2, G, k) cannot
 * survive, so callers need nx >= k).
 */
void dclr_oracle_knn(int b, int nx, int ny, int k, const float *x, const float *y, int64_t *row,
                     int64_t *col) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int qy = 0; qy < ny; ++qy) {
            const size_t gy = (size_t)bi * ny + qy;
            float dist[4096];                        /* k <= 4096 (the HIP search stages at most 4096 candidates) */
            int64_t *r = row + gy * k, *c = col + gy * k;
            for (int s = 0; s < k; ++s) { dist[s] = 1e10f; c[s] = -1; r[s] = -1; }
            for (int qx = 0; qx < nx; ++qx) {
                const size_t gx = (size_t)bi * nx + qx;
                float d = sqdist3(x[gx * 3 + 0], x[gx * 3 + 1], x[gx * 3 + 2],
                                  y[gy * 3 + 0], y[gy * 3 + 1], y[gy * 3 + 2]);
                for (int s1 = 0; s1 < k; ++s1) {
                    if (dist[s1] > d) {
                        for (int s2 = k - 1; s2 > s1; --s2) { dist[s2] = dist[s2 - 1]; c[s2] = c[s2 - 1]; }
                        dist[s1] = d;
                        c[s1] = (int64_t)gx;
                        break;
                    }
                }
            }
            for (int s = 0; s < k; ++s) if (c[s] >= 0) r[s] = (int64_t)gy;
        }
    }
}

void dclr_oracle_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int dclr_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
