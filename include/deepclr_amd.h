/*
 * deepclr_amd.h -- C ABI of libdeepclr_amd.so (gfx950 / MI355X).
 *
 * The library is the drop-in boundary for DeepCLR's forward hot path
 * (BASELINE.json north_star). Conventions, all entry points:
 *
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory
 *     unless its name ends in _host;
 *   - the caller allocates every output and workspace; the library never
 *     synchronises and keeps no global state (re-entrant). It allocates in
 *     ONE place: dclr_furthest_point_sampling, whose signature (the
 *     reference's) has no workspace argument, takes the scratch of the
 *     16385..65536-point kernel from the stream-ordered allocator
 *     (hipMallocAsync / hipFreeAsync on `stream`); every other entry point,
 *     dclr_fps_clouds_ws / _grouped_ws included, works in caller memory only;
 *   - `stream` is a hipStream_t (NULL = default stream); work is enqueued
 *     asynchronously on it, exactly like the reference's wrappers enqueue on
 *     at::cuda::getCurrentCUDAStream() (/root/reference/extern/pointnet2.patch:113,155,285,317);
 *   - return value: 0 = enqueued; DCLR_E_* < 0 = rejected before any launch;
 *     -(1000 + hipError_t) = the HIP runtime refused the launch. The reference
 *     wrappers return a constant 1 that callers ignore and signal misuse through
 *     TORCH_CHECK -> RuntimeError (pointnet2.patch:97-99); the Python host layer
 *     (deepclr_amd/lib.py) turns any non-zero code into RuntimeError likewise.
 *
 * Frozen distance recipe shared with the oracle (oracle/primitives.c):
 *   d = (dx*dx + dy*dy) + dz*dz in binary32, one rounding per operation, no FMA.
 */
#ifndef DEEPCLR_AMD_H
#define DEEPCLR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCLR_OK             0
#define DCLR_E_INVALID     (-1)   /* NULL pointer, non-positive size, size relation violated */
#define DCLR_E_UNSUPPORTED (-2)   /* valid request outside what the kernels are built for     */

typedef void *dclr_stream_t;      /* hipStream_t */

int         dclr_version(void);                 /* 1000*major + minor. 0.2: DclrMergeArgs / DclrCloudArgs start with
                                                 * struct_size and carry `overflow`; dclr_flow_embedding_fused_f16 reads the
                                                 * weight packing dclr_flow_f16_tile(k) names; dclr_knn takes any k */
const char *dclr_error_string(int code);        /* static string, never NULL */
/* Device address of a page-locked host allocation (hipHostMalloc / hipHostRegister, e.g. a pinned torch tensor), asked
 * of the runtime this library runs on: what a host puts into DclrMergeArgs.overflow to poll the flag without a
 * device synchronisation. */
int         dclr_host_device_pointer(void *host, void **device);

/* ------------------------------------------------------------------------------------------
 * Level 1 -- operator-for-operator replacements of the reference's native extension
 * `pointnet2_cuda` and of `torch_cluster.knn` (same argument order and meaning).
 * ---------------------------------------------------------------------------------------- */

/* Replaces furthest_point_sampling_wrapper(b, n, m, points, temp, idx)
 * (/root/reference/extern/pointnet2.patch:306-320).
 * points (b,n,3) f32; temp (b,n) f32 pre-filled by the caller (1e10); idx (b,m) i32.
 * idx[.,0] = 0; temp holds the final running minimum distances on return (over the first m - 1
 * samples, as the reference's kernel leaves it). Any n >= 1, m >= 1 (m > n keeps emitting index 0
 * once every point is taken). Clouds of 16385..65536 points run the workspace kernel on a scratch
 * buffer allocated and freed in stream order (hipMallocAsync / hipFreeAsync on `stream`). */
int dclr_furthest_point_sampling(int b, int n, int m, const float *points, float *temp,
                                 int32_t *idx, dclr_stream_t stream);

/* Replaces gather_points_wrapper_fast(b, c, n, npoints, points, idx, out)
 * (/root/reference/extern/pointnet2.patch:275-288).
 * points (b,c,n) f32; idx (b,npoints) i32; out (b,c,npoints): out[b,c,j] = points[b,c,idx[b,j]]. */
int dclr_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                       float *out, dclr_stream_t stream);

/* Replaces ball_query_wrapper_fast(b, n, m, radius, nsample, new_xyz, xyz, idx)
 * (/root/reference/extern/pointnet2.patch:101-116).
 * new_xyz (b,m,3), xyz (b,n,3) f32; idx (b,m,nsample) i32, zero-filled by the caller.
 * Per centroid: the first nsample point indices k (ascending) with d2 < radius*radius; unused
 * slots repeat the first hit; a centroid with no hit leaves its row untouched. */
int dclr_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int32_t *idx, dclr_stream_t stream);

/* Replaces group_points_wrapper_fast(b, c, n, npoints, nsample, points, idx, out)
 * (/root/reference/extern/pointnet2.patch:160-174).
 * points (b,c,n); idx (b,npoints,nsample); out (b,c,npoints,nsample) = points[b,c,idx[b,j,s]]. */
int dclr_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int32_t *idx, float *out, dclr_stream_t stream);

/* Replaces gather_points_grad_wrapper_fast(b, c, n, npoints, grad_out, idx, grad_points)
 * (/root/reference/extern/pointnet2.patch:290-304): the backward of dclr_gather_points.
 * grad_out (b,c,npoints) f32; idx (b,npoints) i32; grad_points (b,c,n) f32, zero-filled by the caller (the
 * reference's autograd Function allocates it zeroed): grad_points[b,c,idx[b,j]] += grad_out[b,c,j]. */
int dclr_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out, const int32_t *idx,
                            float *grad_points, dclr_stream_t stream);

/* Replaces group_points_grad_wrapper_fast(b, c, n, npoints, nsample, grad_out, idx, grad_points)
 * (/root/reference/extern/pointnet2.patch:144-158): the backward of dclr_group_points.
 * grad_out (b,c,npoints,nsample); idx (b,npoints,nsample); grad_points (b,c,n), zero-filled by the caller:
 * grad_points[b,c,idx[b,j,s]] += grad_out[b,c,j,s]. Runs of adjacent equal indices (a ball-query row repeats its
 * first hit in every unused slot) are summed in registers and cost one atomic each. Out-of-range indices are ignored. */
int dclr_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                           const int32_t *idx, float *grad_points, dclr_stream_t stream);

/* Replaces torch_cluster.knn(x, y, k, batch_x, batch_y) for the equally sized, sorted batches
 * DeepCLR builds (/root/reference/deepclr/models/deepclr.py:149-155,164-166).
 * x (b*nx,3) candidates, y (b*ny,3) queries; row, col (b*ny*k) i64: row = global query index,
 * col = global candidate index, each query's k entries contiguous, ascending distance, equal
 * distances in ascending candidate index. Needs 1 <= k <= nx <= 4096 (k <= 40 takes the rank selection, larger k one
 * wave arg-min round per neighbour). */
int dclr_knn(int b, int nx, int ny, int k, const float *x, const float *y, int64_t *row,
             int64_t *col, dclr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Level 2 -- the fused forward path. Internal activations are point-major rows:
 *   cloud-feature rows  F (clouds*npoint, 68):  [feat 0..63 | x y z | 0]
 *   flow-embedding rows E (pairs*npoint, 264):  [feat 0..255 | x y z | 0 x5]
 * and MLP weights are pre-packed once into MFMA fragment order (dclr_pack_weight).
 * ---------------------------------------------------------------------------------------- */

#define DCLR_F_STRIDE 68
#define DCLR_E_STRIDE 264

/* FPS on interleaved clouds (b, n, c) f32, c in {3,4,...}: xyz = first 3 of each point. No temp
 * buffer (running minima live in registers); idx (b,m) i32. Same result as
 * dclr_furthest_point_sampling on the xyz slice with temp = 1e10. */
int dclr_fps_clouds(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                    dclr_stream_t stream);

/* The same sampling, additionally exporting the spatial partition the kernel builds for its own
 * pruning: every cloud is split into n_groups compact groups of group_size points
 * (dclr_fps_group_layout; available for 1024 < n <= 65536, otherwise DCLR_E_UNSUPPORTED; clouds above
 * 16384 points go through dclr_fps_clouds_grouped_ws).
 * group_pts (b, n_groups*group_size, 4) f32: x y z and the point index as raw u32 bits (0xFFFFFFFF =
 * padding slot, coordinates 3e38); group_box (b, n_groups, 8) f32: min xyz, max xyz, 0, 0 -- except
 * group_box[cloud][0][6], which receives the number of barrier rounds the cloud's sampling took (diagnostics:
 * (m - 1) / rounds = samples per round).
 * dclr_sa_msg_fused uses them to skip the exhaustive ball-query sweep. */
int dclr_fps_group_layout(int n, int *n_groups, int *group_size);
/* Large clouds (16384 < n <= 65536): same samples through a spatially pruned kernel whose sorted points and
 * running minima live in a caller-provided workspace of dclr_fps_workspace_bytes(b, n) bytes (16-byte
 * aligned; 0 bytes = size not covered, the call then equals dclr_fps_clouds). */
long long dclr_fps_workspace_bytes(int b, int n);
int dclr_fps_clouds_ws(int b, int n, int c, int m, const float *clouds, int32_t *idx, void *workspace,
                       long long workspace_bytes, dclr_stream_t stream);
int dclr_fps_clouds_grouped(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                            float *group_pts, float *group_box, dclr_stream_t stream);
/* The grouped form for 16384 < n <= 65536: the kernel's sorted point list IS group_pts (128 / 256 groups of 256
 * points), so the workspace only has to hold the rest (any dclr_fps_workspace_bytes(b, n) buffer is enough). */
int dclr_fps_clouds_grouped_ws(int b, int n, int c, int m, const float *clouds, int32_t *idx,
                               float *group_pts, float *group_box, void *workspace, long long workspace_bytes,
                               dclr_stream_t stream);

/* Set abstraction, multi-scale grouping, fused (reference: SetAbstraction.forward,
 * /root/reference/deepclr/models/deepclr.py:88-94, -> PointnetSAModuleMSG with use_xyz=True, bn=False):
 * per sampled centroid and scale: ball query -> [xyz - centroid, features] -> 1x1-conv MLP
 * (c -> 16 -> 16 -> 32, ReLU each) -> max over the neighbourhood. Neighbourhoods are never
 * materialised; slots that only repeat the first hit are skipped (max is idempotent).
 * clouds (b,n,c), c in {3,4}; fps_idx (b,npoint); n_scales in {1,2}; radii_host/nsamples_host are
 * HOST arrays read at call time; mlp[s] is a device array [W1(16,c) b1(16) W2(16,16) b2(16) W3(32,16) b3(32)];
 * out rows F (b*npoint, 68), scale s at columns 32*s..; counts (b,npoint,n_scales) i32 or NULL
 * receives min(hits, nsample) per centroid (diagnostics / parity tests). group_pts / group_box: the
 * optional outputs of dclr_fps_clouds_grouped for the same clouds (both NULL: exhaustive sweep). */
int dclr_sa_msg_fused(int b, int n, int c, int npoint, const float *clouds, const int32_t *fps_idx,
                      int n_scales, const float *radii_host, const int *nsamples_host,
                      const float *const *mlp_host_ptrs, float *out_rows, int32_t *counts,
                      const float *group_pts, const float *group_box, dclr_stream_t stream);

/* Layout conversion between rows F/E and the reference's channel-major tensors:
 * channels (b, 3 + nfeat, npoint) with xyz in channels 0..2  <->  rows (b*npoint, stride) with the
 * nfeat feature columns first and xyz at columns xyz_col..xyz_col+2 (64 for F, 256 for E); every
 * other row column is written as zero by dclr_channels_to_rows. */
int dclr_rows_to_channels(int b, int npoint, int nfeat, int xyz_col, int stride, const float *rows,
                          float *channels, dclr_stream_t stream);
int dclr_channels_to_rows(int b, int npoint, int nfeat, int xyz_col, int stride, const float *channels,
                          float *rows, dclr_stream_t stream);

/* Pack a row-major weight W (n_out, k_in) into MFMA fragment order for dclr_linear / the fused
 * kernels. kmap (kp) i32 DEVICE array or NULL: packed K position -> source column (-1 = zero);
 * NULL = identity, zero padded. kp = padded K (multiple of 8), np = padded N (multiple of 32);
 * packed holds np*kp floats. */
int dclr_pack_weight(int n_out, int k_in, const float *w, const int32_t *kmap, int kp, int np,
                     float *packed, dclr_stream_t stream);

/* The same for the fused flow-embedding kernel's 16-column tile layout (v_mfma_f32_16x16x4_f32):
 * kp multiple of 16, np multiple of 16. */
int dclr_pack_weight16(int n_out, int k_in, const float *w, const int32_t *kmap, int kp, int np,
                       float *packed, dclr_stream_t stream);

/* Y = act(X * W^T + bias): X rows (m, ldx) using its first kp columns, packed W (np, kp), bias (n) or
 * NULL, Y rows (m, ldy) first n columns. relu != 0 applies max(.,0). Requires m % 64 == 0,
 * ldx % 4 == 0 and 16-byte aligned X. If colmax != NULL nothing is written to Y; instead
 * colmax (m / rows_per_group, n) receives the per-group column maxima of the activated output
 * through atomic max -- the caller zero-fills it, relu must be set, rows_per_group % 64 == 0. */
int dclr_linear(int m, int n, int kp, const float *x, int ldx, const float *w_packed, const float *bias,
                int relu, float *y, int ldy, float *colmax, int rows_per_group, dclr_stream_t stream);

/* Two dclr_linear products in one launch: rows [0, m_each) of x with w_a -> y_a, rows [m_each, 2 m_each) with w_b ->
 * y_b; no bias, no activation (the template / source halves of flow layer 1). */
int dclr_linear_pair(int m_each, int n, int kp, const float *x, int ldx, const float *w_a, const float *w_b,
                     float *y_a, float *y_b, int ldy, dclr_stream_t stream);

/* The whole 1x1-conv chain of the pose head in one launch (reference: OutputSimple.forward,
 * /root/reference/deepclr/models/deepclr.py:286-287): x rows (m, ldx) -> n_layers x [affine + ReLU] ->
 * column maxima per group of rows_per_group rows into colmax (m / rows_per_group, n_last), which the caller
 * zero-fills. k_host / n_host: padded input / output width per layer (k multiple of 8, n multiple of 32,
 * k[l] == n[l-1], hidden widths <= 512); w_packed_host[l]: dclr_pack_weight(n[l], ., kp = k[l], np = n[l]);
 * the four *_host arrays are HOST arrays of sizes / device pointers read at call time. m and
 * rows_per_group multiples of 32. */
int dclr_head_conv_fused(int m, int n_layers, const int *k_host, const int *n_host,
                         const float *const *w_packed_host, const float *const *bias_host, const float *x,
                         int ldx, float *colmax, int rows_per_group, dclr_stream_t stream);

/* kNN on feature rows: queries = template clouds 0..pairs-1, candidates = source clouds
 * pairs..2*pairs-1 of F (xyz at columns 64..66). knn_idx (pairs, npoint, k) i32 local candidate
 * indices, ordered as dclr_knn orders them. */
int dclr_knn_rows(int pairs, int npoint, int k, const float *f_rows, int32_t *knn_idx,
                  dclr_stream_t stream);

/* Flow embedding, fused (reference: MotionEmbeddingBase.forward,
 * /root/reference/deepclr/models/deepclr.py:201-231, append_features=True, 131->128->128->256):
 * layer 1 is split as W1 = [W1a | W1b | W1c] over [pos_diff | template feat | source feat];
 * pt/ps (pairs*npoint,128) hold W1b*feat_t and W1c*feat_s (dclr_linear, no bias, no relu).
 * Per template point: gather its k neighbours, h1 = relu(pt + ps[nb] + W1a*pos_diff + b1),
 * two MFMA layers, zero rows with |pos_diff| >= radius (radius <= 0 disables), max over k.
 * w1a (128,3) row-major, b1 (128); w2p/w3p packed with dclr_pack_weight16 (128,128)/(256,128);
 * out rows E. k <= 32. */
int dclr_flow_embedding_fused(int pairs, int npoint, int k, float radius, const float *f_rows,
                              const int32_t *knn_idx, const float *pt, const float *ps,
                              const float *w1a, const float *b1, const float *w2p, const float *b2,
                              const float *w3p, const float *b3, float *e_rows, dclr_stream_t stream);

/* Fully connected tail on a handful of rows: y (m,n) = act(x (m,k) * w (n,k)^T + bias).
 * act: 0 none, 1 relu, 2 dual-quaternion head (sigmoid on column 0, tanh on 1..3; reference
 * OutputSimple._output_activation, deepclr.py:279-281), 3 quaternion head (sigmoid col 3, tanh 4..6). */
int dclr_fc(int m, int n, int k, const float *x, const float *w, const float *bias, int act, float *y,
            dclr_stream_t stream);

/* ---- split-fp16 matrix path (deepclr_amd/csrc/mma16f.h) ------------------------------------------------
 * The same operators as dclr_head_conv_fused / dclr_flow_embedding_fused with every f32 operand carried as
 * f16 hi + f16 lo * 2^-11 and each product evaluated as hi*hi + 2^-11 (hi*lo + lo*hi) on the f16 matrix
 * instructions with f32 accumulation: f32-accurate results (measured closer to fp64 than the f32 matrix
 * path) at 16/3 of its rate. Operands must stay below 65504 in magnitude.
 * dclr_pack_weight_f16: w (n_out, k_in) row-major -> hi plane | lo plane (np * kp halves each), fragment
 *   order for `width`-column tiles (32: kp % 16 == 0; 16: kp % 32 == 0); kmap as in
 *   dclr_pack_weight. packed: 4 * np * kp bytes, 16-byte aligned.
 * dclr_head_conv_fused_f16: x rows hold k_in valid f32 columns (k_in % 8 == 0, k_in <= k[0]); k[l] % 16 == 0.
 * dclr_flow_embedding_fused_f16: w2p / w3p packed with width dclr_flow_f16_tile(k), kp 128 (ABI 0.2: 32 from 29
 *   neighbours up, where the kernel runs on v_mfma_f32_32x32x16_f16 tiles, 16 below; ABI 0.1 read width 16 for every k). */
int dclr_flow_f16_tile(int k);
int dclr_pack_weight_f16(int n_out, int k_in, const float *w, const int32_t *kmap, int kp, int width, void *packed,
                         dclr_stream_t stream);
int dclr_head_conv_fused_f16(int m, int n_layers, int k_in, const int *k_host, const int *n_host,
                             const void *const *w_packed_host, const float *const *bias_host, const float *x,
                             int ldx, float *colmax, int rows_per_group, dclr_stream_t stream);
int dclr_flow_embedding_fused_f16(int pairs, int npoint, int k, float radius, const float *f_rows,
                                  const int32_t *knn_idx, const float *pt, const float *ps, const float *w1a,
                                  const float *b1, const void *w2p, const float *b2, const void *w3p,
                                  const float *b3, float *e_rows, dclr_stream_t stream);
/* dclr_sa_msg_fused with layers 2 and 3 of the shared MLP (16 -> 16 -> 32) on split-f16 operands; same arguments, same
 * weight buffers (f32: the kernel splits its fragments itself), same neighbour sets and counts; features agree with the
 * f32 form to f32 rounding. Activations must stay below 65504. */
int dclr_sa_msg_fused_f16(int b, int n, int c, int npoint, const float *clouds, const int32_t *fps_idx,
                          int n_scales, const float *radii_host, const int *nsamples_host,
                          const float *const *mlp_host_ptrs, float *out_rows, int32_t *counts,
                          const float *group_pts, const float *group_box, dclr_stream_t stream);

/* ---- batches that travel together, not concatenated ------------------------------------------------------
 * The reference runs one batch per call (/root/reference/deepclr/models/deepclr.py:488-503: x = [templates | sources] of
 * ONE batch). The pipelined runner puts several batches into one sampling launch and one dense launch; their clouds had
 * to be concatenated first as [templates of every batch | sources of every batch] (42 MB copied per ten KITTI batches).
 * These two entries read the batches where they lie instead: the b = 2 * pairs_per_batch * n_batches clouds of the call are
 * n_batches batches of 2 * pairs_per_batch clouds each ([templates | sources], (2 B, n, c) contiguous), batch i at
 * clouds + i * batch_stride floats (0: the same batch every time; any constant stride, e.g. a ring of staging buffers);
 * cloud j of the call = template / source (j / (B g)) of batch (j % (B g)) / B, i.e. all outputs (idx, groups, rows) are in
 * the concatenated order. n_batches = 1 is the plain call. Grouped sampler: workspace as dclr_fps_clouds_grouped_ws for
 * n > 16384 (ignored otherwise); set abstraction: f16 != 0 selects dclr_sa_msg_fused_f16. */
int dclr_fps_clouds_grouped_batched(int b, int n, int c, int m, const float *clouds, int pairs_per_batch, int n_batches,
                                    long long batch_stride, int32_t *idx, float *group_pts, float *group_box,
                                    float *slice_box, void *workspace, long long workspace_bytes, dclr_stream_t stream);
int dclr_sa_msg_fused_batched(int f16, int b, int n, int c, int npoint, const float *clouds, int pairs_per_batch,
                              int n_batches, long long batch_stride, const int32_t *fps_idx, int n_scales,
                              const float *radii_host, const int *nsamples_host, const float *const *mlp_host_ptrs,
                              float *out_rows, int32_t *counts, const float *group_pts, const float *group_box,
                              const float *slice_box, dclr_stream_t stream);
/* The same with the word the split-f16 layers (f16 != 0) set to 1 when an activation exceeds 65504 and is clamped
 * (NULL: not reported; semantics of DclrMergeArgs.overflow below). ABI 0.2. */
int dclr_sa_msg_fused_batched_ov(int f16, int b, int n, int c, int npoint, const float *clouds, int pairs_per_batch,
                                 int n_batches, long long batch_stride, const int32_t *fps_idx, int n_scales,
                                 const float *radii_host, const int *nsamples_host, const float *const *mlp_host_ptrs,
                                 float *out_rows, int32_t *counts, const float *group_pts, const float *group_box,
                                 const float *slice_box, uint32_t *overflow, dclr_stream_t stream);
/* slice_box (optional, groups of more than 64 points, n <= 16384): (b, n_groups * group_size / 64, 8) f32, the boxes
 * (min xyz, max xyz, 0, 0) of the 64-point slices of every exported group -- a group's points are exported slice by slice,
 * slice r = the r-th 64 consecutive points of the spatially sorted cloud inside the group. Set abstraction tests a
 * centroid's ball against the slices of the groups it reaches and loads only those that can hold a neighbour (a quarter of
 * the points on the bench clouds). */

/* ---- the dense stages of one batch in one call ----------------------------------------------------------
 * Rows F of [templates..., sources...] -> pose outputs y (pairs, n_out): the launches DeepCLR.forward makes
 * after set abstraction (reference: deepclr.py:502-506: merge layers = flow embedding, then the pose head)
 * enqueued back to back on one stream -- per-point layer-1 products (2 x dclr_linear), dclr_knn_rows, the
 * fused flow embedding, the fused conv chain + max over points, the fully connected tail. Exists because the
 * host, not the GPU, bounds the step once these take ~0.25 ms: one foreign call and no allocation per batch.
 * All buffers are the caller's; the workspace may be reused by the next call on the same stream.
 * events: NULL, or DCLR_MERGE_EVENTS hipEvent_t handles recorded on `stream` before the first launch and after
 * each stage (slot order: start, pt, ps, knn, flow, head, then one per fully connected layer).
 * Range of the split-fp16 path (precision == 1): operands beyond +-65504 are clamped; the kernels report it through
 * `overflow` (below) instead of returning wrong poses silently. */
#define DCLR_MERGE_MAX_LAYERS 8
#define DCLR_MERGE_MAX_FC 4
#define DCLR_MERGE_EVENTS (6 + DCLR_MERGE_MAX_FC)
typedef struct DclrMergeArgs {
    uint32_t struct_size;                   /* sizeof(DclrMergeArgs) of the header the caller was built against: a call with
                                             * another size is rejected (DCLR_E_INVALID) instead of being read past its end
                                             * (ABI 0.2; 0.1 had no such member and no `overflow`) */
    int pairs, npoint, k, precision;        /* precision: 0 = f32 matrix path, 1 = split-fp16 (f16x2) */
    int stages;                             /* bit 0: layer-1 halves + kNN (fill pt, ps, knn_idx); bit 1: flow embedding,
                                             * head, fully connected tail (consume them); 3 = everything */
    float radius;
    int n_head_layers, head_k_in, n_fc;
    int head_k[DCLR_MERGE_MAX_LAYERS], head_n[DCLR_MERGE_MAX_LAYERS];
    int fc_k[DCLR_MERGE_MAX_FC], fc_n[DCLR_MERGE_MAX_FC], fc_act[DCLR_MERGE_MAX_FC];
    const float *f_rows;                    /* (2 * pairs * npoint, DCLR_F_STRIDE) */
    const float *wt, *ws;                   /* dclr_pack_weight of W1's template / source feature blocks (kp 64) */
    const float *w1a, *b1;                  /* (128, 3) position block of layer 1, bias */
    const void *w2, *w3;                    /* flow layers 2, 3: dclr_pack_weight16 (f32) or dclr_pack_weight_f16 */
    const float *b2, *b3;
    const void *head_w[DCLR_MERGE_MAX_LAYERS];      /* dclr_pack_weight (f32) or dclr_pack_weight_f16 (width 32) */
    const float *head_b[DCLR_MERGE_MAX_LAYERS];
    const float *fc_w[DCLR_MERGE_MAX_FC], *fc_b[DCLR_MERGE_MAX_FC];       /* row-major (n, k) */
    float *pt, *ps;                         /* workspace (pairs * npoint, 128) each */
    int32_t *knn_idx;                       /* workspace (pairs, npoint, k) */
    float *e_rows;                          /* workspace (pairs * npoint, DCLR_E_STRIDE) */
    float *colmax;                          /* workspace (pairs, head_n[last]) */
    float *fc_tmp[2];                       /* workspace (pairs, max fc width) each */
    float *y;                               /* out (pairs, fc_n[last]) */
    uint32_t *overflow;                     /* NULL, or one word (device memory, or host memory mapped into the device's
                                             * address space) that the split-fp16 kernels set to 1 when an activation
                                             * exceeds 65504 and is clamped (precision == 1); sticky: the caller clears it.
                                             * While it is set the last fully connected layer writes y as NaN (ABI 0.2): a
                                             * clamped forward never hands out plausible-looking poses */
} DclrMergeArgs;
int dclr_merge_forward(const DclrMergeArgs *args, void *const *events, dclr_stream_t stream);

/* ---- the per-cloud stages of one launch group in one call ------------------------------------------------
 * Clouds -> rows F (-> the per-point halves of flow layer 1 and the kNN lists): the launches DeepCLR.forward makes
 * before the dense stages (reference: deepclr.py:488-521 cloud_features -> SetAbstraction, then the grouping half of
 * MotionEmbedding, deepclr.py:149-171) enqueued back to back on one stream -- dclr_fps_clouds_grouped_batched,
 * dclr_sa_msg_fused_batched and, when `merge` is given, dclr_merge_forward with stages = 1 on the rows just written
 * (merge->f_rows and merge->stages are ignored: the call uses f_rows and 1). Exists for the same reason as
 * dclr_merge_forward: at ~0.2 ms of GPU time per batch the ~0.3 ms of host time the separate calls cost at every
 * sampling launch is exposed whenever the pipeline starts from an idle chip.
 * fps_idx, group_pts, group_box, slice_box, workspace: the caller's scratch (shapes as in the batched entries; they may be
 * reused by the next call on the same stream); f_rows (b * npoint, DCLR_F_STRIDE) is the product.
 * events: NULL, or DCLR_CLOUD_EVENTS hipEvent_t handles recorded on `stream` (start, after sampling, after set
 * abstraction); merge_events: NULL or as dclr_merge_forward (only the stage-1 slots are recorded). */
#define DCLR_CLOUD_MAX_SCALES 4
#define DCLR_CLOUD_EVENTS 3
typedef struct DclrCloudArgs {
    uint32_t struct_size;                   /* sizeof(DclrCloudArgs), checked like DclrMergeArgs.struct_size */
    int b, n, c, npoint;                    /* clouds of the call (all batches), points per cloud, columns, samples */
    int pairs_per_batch, n_batches;         /* as dclr_fps_clouds_grouped_batched: b == 2 * pairs_per_batch * n_batches */
    long long batch_stride;
    int f16, n_scales;                      /* f16 != 0: set abstraction layers 2, 3 on split-f16 operands */
    float radii[DCLR_CLOUD_MAX_SCALES];
    int nsamples[DCLR_CLOUD_MAX_SCALES];
    const float *mlp[DCLR_CLOUD_MAX_SCALES];        /* device: [W1 b1 W2 b2 W3 b3] per scale, as dclr_sa_msg_fused */
    const float *clouds;                    /* first batch */
    int32_t *fps_idx;                       /* scratch (b, npoint) */
    float *group_pts, *group_box, *slice_box;       /* scratch; slice_box may be NULL */
    void *workspace;                        /* scratch for n > 16384, else ignored */
    long long workspace_bytes;
    float *f_rows;                          /* out */
    const DclrMergeArgs *merge;             /* optional: stage 1 of dclr_merge_forward on f_rows (pairs = b / 2) */
    uint32_t *overflow;                     /* as DclrMergeArgs.overflow, for the split-f16 set-abstraction layers (f16 != 0);
                                             * NULL: merge->overflow when `merge` is given, else not reported */
} DclrCloudArgs;
int dclr_cloud_forward(const DclrCloudArgs *args, void *const *events, void *const *merge_events, dclr_stream_t stream);

/* ---- scan preparation (reference: CPU transforms run per sample before the model) -----------------------
 * One order-preserving pass over a raw scan raw (n_raw, c_raw): keep rows start, start+nth, ...
 * (SystematicErasing, /root/reference/deepclr/data/transforms/transforms.py:244-268), of those the rows with
 * min_range <= max(|x|,|y|) <= max_range (RangeSelection, transforms.py:90-110; min 0 and max +inf = keep all),
 * and of each row the first c_out columns (TruncateDimension, transforms.py:271-282).
 * out: capacity ceil((n_raw-start)/nth) rows of c_out floats; count: 1 int (rows written);
 * block_counts: scratch of dclr_prepare_cloud_blocks(n_raw, nth, start) ints. */
int dclr_prepare_cloud_blocks(int n_raw, int nth, int start);
int dclr_prepare_cloud(int n_raw, int c_raw, const float *raw, int nth, int start, float min_range, float max_range,
                       int c_out, float *out, int32_t *count, int32_t *block_counts, dclr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCLR_AMD_H */
