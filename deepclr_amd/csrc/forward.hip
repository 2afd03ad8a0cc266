// The dense stages of one batch behind a single entry point (host-side launch sequencing only; the
// kernels are the ones of knn.hip, gemm.hip / gemm16.hip and flow.hip / flow16.hip).
//
// Reference call order: DeepCLR.forward -> merge layers (/root/reference/deepclr/models/deepclr.py:502-506):
// MotionEmbedding (kNN grouping 149-171, shared MLP + mask + max 201-231), then OutputSimple (284-294).
#include "common.h"

extern "C" int dclr_merge_forward(const DclrMergeArgs *a, void *const *events, dclr_stream_t stream) {
    DCLR_REQUIRE(a != nullptr && a->struct_size == sizeof(DclrMergeArgs));     // a caller built against another header
    DCLR_REQUIRE(a->pairs > 0 && a->npoint > 0 && a->n_head_layers >= 1 && a->n_head_layers <= DCLR_MERGE_MAX_LAYERS &&
                 a->n_fc >= 1 && a->n_fc <= DCLR_MERGE_MAX_FC && (a->precision == 0 || a->precision == 1) &&
                 a->stages >= 1 && a->stages <= 3);
    DCLR_REQUIRE(a->f_rows && a->pt && a->ps && a->knn_idx);
    DCLR_REQUIRE(!(a->stages & 2) || (a->e_rows && a->colmax && a->y && a->fc_tmp[0] && a->fc_tmp[1]));
    hipStream_t st = (hipStream_t)stream;
    int slot = 0;
    auto mark = [&]() {
        if (events && events[slot]) (void)hipEventRecord((hipEvent_t)events[slot], st);
        ++slot;
    };
    const int rows = a->pairs * a->npoint;
    int rc;
    mark();
    if (a->stages & 1) {
        // per-point halves of flow layer 1: W1b * feat_t (templates), W1c * feat_s (sources), one launch
        rc = dclr_linear_pair(rows, 128, 64, a->f_rows, DCLR_F_STRIDE, a->wt, a->ws, a->pt, a->ps, 128, stream);
        if (rc != DCLR_OK) return rc;
        mark();
        mark();
        rc = dclr_knn_rows(a->pairs, a->npoint, a->k, a->f_rows, a->knn_idx, stream);
        if (rc != DCLR_OK) return rc;
        mark();
    } else {
        slot += 2;
        mark();                             // slot 3 doubles as the start of the flow-embedding span
    }
    if (!(a->stages & 2)) return DCLR_OK;
    const int n_last = a->head_n[a->n_head_layers - 1];
    const long long n_colmax = (long long)a->pairs * n_last;
    if (a->precision == 1)          // the split-f16 flow kernel also clears the head's column maxima (no fill launch)
        rc = dclr_x_flow_embedding_fused_f16(a->pairs, a->npoint, a->k, a->radius, a->f_rows, a->knn_idx, a->pt, a->ps,
                                             a->w1a, a->b1, a->w2, a->b2, a->w3, a->b3, a->e_rows, a->colmax, n_colmax,
                                             a->overflow, stream);
    else
        rc = dclr_flow_embedding_fused(a->pairs, a->npoint, a->k, a->radius, a->f_rows, a->knn_idx, a->pt, a->ps, a->w1a,
                                       a->b1, (const float *)a->w2, a->b2, (const float *)a->w3, a->b3, a->e_rows, stream);
    if (rc != DCLR_OK) return rc;
    mark();
    if (a->precision != 1) {
        const hipError_t ms = hipMemsetAsync(a->colmax, 0, (size_t)n_colmax * sizeof(float), st);
        if (ms != hipSuccess) {
            (void)hipGetLastError();                        // consume the sticky copy; the code below carries the error
            return -(1000 + (int)ms);
        }
    }
    if (a->precision == 1)
        rc = dclr_x_head_conv_fused_f16(rows, a->n_head_layers, a->head_k_in, a->head_k, a->head_n, a->head_w, a->head_b,
                                        a->e_rows, DCLR_E_STRIDE, a->colmax, a->npoint, a->overflow, stream);
    else
        rc = dclr_head_conv_fused(rows, a->n_head_layers, a->head_k, a->head_n, (const float *const *)a->head_w, a->head_b,
                                  a->e_rows, DCLR_E_STRIDE, a->colmax, a->npoint, stream);
    if (rc != DCLR_OK) return rc;
    mark();
    const float *x = a->colmax;
    for (int l = 0; l < a->n_fc; ++l) {
        float *out = l == a->n_fc - 1 ? a->y : a->fc_tmp[l & 1];
        // the last layer writes NaN poses when the sticky overflow word is set (a clamped activation in this or an earlier
        // forward the host has not acknowledged yet): never a plausible-looking wrong pose
        const uint32_t *poison = (l == a->n_fc - 1 && a->precision == 1) ? a->overflow : nullptr;
        rc = dclr_x_fc(a->pairs, a->fc_n[l], a->fc_k[l], x, a->fc_w[l], a->fc_b[l], a->fc_act[l], out, poison, stream);
        if (rc != DCLR_OK) return rc;
        mark();
        x = out;
    }
    return DCLR_OK;
}

// The per-cloud stages of one launch group behind a single entry point (reference call order: DeepCLR.cloud_features ->
// SetAbstraction, /root/reference/deepclr/models/deepclr.py:510-521,86-93; then the grouping half of MotionEmbedding,
// deepclr.py:149-171, which needs nothing but the rows just written).
extern "C" int dclr_cloud_forward(const DclrCloudArgs *a, void *const *events, void *const *merge_events,
                                  dclr_stream_t stream) {
    DCLR_REQUIRE(a != nullptr && a->struct_size == sizeof(DclrCloudArgs));
    DCLR_REQUIRE(a->merge == nullptr || a->merge->struct_size == sizeof(DclrMergeArgs));
    DCLR_REQUIRE(a->b > 0 && a->n > 0 && a->c >= 3 && a->npoint > 0 && a->n_scales >= 1 && a->n_scales <= DCLR_CLOUD_MAX_SCALES);
    DCLR_REQUIRE(a->clouds && a->fps_idx && a->group_pts && a->group_box && a->f_rows);
    hipStream_t st = (hipStream_t)stream;
    int slot = 0;
    auto mark = [&]() {
        if (events && events[slot]) (void)hipEventRecord((hipEvent_t)events[slot], st);
        ++slot;
    };
    mark();
    int rc = dclr_fps_clouds_grouped_batched(a->b, a->n, a->c, a->npoint, a->clouds, a->pairs_per_batch, a->n_batches,
                                             a->batch_stride, a->fps_idx, a->group_pts, a->group_box, a->slice_box,
                                             a->workspace, a->workspace_bytes, stream);
    if (rc != DCLR_OK) return rc;
    mark();
    // (the split-f16 set-abstraction layers report a clamped activation to the same word as the dense stages)
    rc = dclr_sa_msg_fused_batched_ov(a->f16, a->b, a->n, a->c, a->npoint, a->clouds, a->pairs_per_batch, a->n_batches,
                                     a->batch_stride, a->fps_idx, a->n_scales, a->radii, a->nsamples, a->mlp, a->f_rows, nullptr,
                                     a->group_pts, a->group_box, a->slice_box, a->overflow ? a->overflow : a->merge ? a->merge->overflow : nullptr,
                                     stream);
    if (rc != DCLR_OK) return rc;
    mark();
    if (a->merge == nullptr) return DCLR_OK;
    DclrMergeArgs m = *a->merge;
    DCLR_REQUIRE(2 * m.pairs == a->b && m.npoint == a->npoint);
    m.f_rows = a->f_rows;
    m.stages = 1;
    return dclr_merge_forward(&m, merge_events, stream);
}
