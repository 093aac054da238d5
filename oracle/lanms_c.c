/*
 * Plain-C restatement of the reference's locality-aware NMS and matrix NMS for box counts where
 * the literal NumPy statement (oracle/nms.py) would need N x N float matrices of gigabytes.
 * TEST INFRASTRUCTURE ONLY - never linked into the shipped library.
 *
 * Follows utils/image/bounding_box/locality_aware_nms.py:59-126 (self_merging, _self_merging_body,
 * _merge_boxes) and non_max_suppression.py:191-220 (self_suppression), entry by entry of the masked
 * IoU matrix, evaluated row by row instead of being stored.  Boxes are expected in rank order
 * (best first), exactly like the reference after _sort_boxes.  IoU = metrics.py:30-46:
 * union = area_i - inter + area_j with i the better-ranked box, 0/0 -> 0.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float iou_ij(const float* bi, const float* bj) {
    /* bi = better-ranked row box, bj = column box */
    float area1 = (bi[3] - bi[1]) * (bi[2] - bi[0]);
    float area2 = (bj[3] - bj[1]) * (bj[2] - bj[0]);
    float xmin = bi[0] > bj[0] ? bi[0] : bj[0];
    float ymin = bi[1] > bj[1] ? bi[1] : bj[1];
    float xmax = bi[2] < bj[2] ? bi[2] : bj[2];
    float ymax = bi[3] < bj[3] ? bi[3] : bj[3];
    float iw = xmax - xmin, ih = ymax - ymin;
    if (iw < 0.0f) iw = 0.0f;
    if (ih < 0.0f) ih = 0.0f;
    float inter = iw * ih;
    float uni = area1 - inter + area2;
    return uni != 0.0f ? inter / uni : 0.0f;
}

static int is_zero(const float* b) { return b[0] == 0.0f && b[1] == 0.0f && b[2] == 0.0f && b[3] == 0.0f; }

/* boxes: (n,4) float32, rank order, modified in place.  method 0 = union, 1 = average.
 * max_iter <= 0 means n.  thr must be > 0.  Returns the number of iterations run. */
int oracle_lanms_merge(float* boxes, int n, float thr, int method, int max_iter) {
    if (max_iter <= 0) max_iter = n;
    unsigned char* cso = (unsigned char*)malloc((size_t)n);
    unsigned char* supp = (unsigned char*)malloc((size_t)n);
    float* merged = (float*)malloc((size_t)n * 4 * sizeof(float));
    int it = 0, cond = 1;
    while (cond && it < max_iter) {
        /* can_suppress_others[i] = (max_k iou[k,i] < thr), iou masked to k < i and iou >= thr */
        for (int i = 0; i < n; ++i) {
            cso[i] = 1;
            supp[i] = 0;
            if (is_zero(boxes + 4 * i)) continue;
            for (int k = 0; k < i; ++k) {
                if (is_zero(boxes + 4 * k)) continue;
                if (iou_ij(boxes + 4 * k, boxes + 4 * i) >= thr) { cso[i] = 0; break; }
            }
        }
        /* merging_mask[i,j] = iou[i,j] * cso[i] >= thr ; suppressed[j] = any_i ; merged rows */
        for (int i = 0; i < n; ++i) {
            const float* bi = boxes + 4 * i;
            float mnx = INFINITY, mny = INFINITY, mxx = 0.0f, mxy = 0.0f;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            int cnt = 0;
            if (cso[i] && !is_zero(bi)) {
                for (int j = i + 1; j < n; ++j) {
                    const float* bj = boxes + 4 * j;
                    if (is_zero(bj)) continue;
                    if (iou_ij(bi, bj) >= thr) {
                        supp[j] = 1;
                        if (bj[0] < mnx) mnx = bj[0];
                        if (bj[1] < mny) mny = bj[1];
                        if (bj[2] > mxx) mxx = bj[2];
                        if (bj[3] > mxy) mxy = bj[3];
                        s0 = s0 + bj[0]; s1 = s1 + bj[1]; s2 = s2 + bj[2]; s3 = s3 + bj[3];
                        cnt++;
                    }
                }
            }
            float* m = merged + 4 * i;
            if (method == 0) {
                m[0] = bi[0] < mnx ? bi[0] : mnx;
                m[1] = bi[1] < mny ? bi[1] : mny;
                m[2] = bi[2] > mxx ? bi[2] : mxx;
                m[3] = bi[3] > mxy ? bi[3] : mxy;
            } else {
                float c = (float)(1 + cnt);
                m[0] = (bi[0] + s0) / c; m[1] = (bi[1] + s1) / c; m[2] = (bi[2] + s2) / c; m[3] = (bi[3] + s3) / c;
            }
        }
        /* loop condition: any(iou[suppressed rows] >= thr) */
        cond = 0;
        for (int i = 0; i < n && !cond; ++i) {
            if (!supp[i] || is_zero(boxes + 4 * i)) continue;
            for (int j = i + 1; j < n; ++j) {
                if (is_zero(boxes + 4 * j)) continue;
                if (iou_ij(boxes + 4 * i, boxes + 4 * j) >= thr) { cond = 1; break; }
            }
        }
        for (int i = 0; i < n; ++i) {
            if (supp[i]) {
                boxes[4 * i] = boxes[4 * i + 1] = boxes[4 * i + 2] = boxes[4 * i + 3] = 0.0f;
            } else {
                memcpy(boxes + 4 * i, merged + 4 * i, 4 * sizeof(float));
            }
        }
        ++it;
    }
    free(cso); free(supp); free(merged);
    return it;
}

/* Matrix NMS (self_suppression): rows of the masked IoU matrix are zeroed when a box that no
 * remaining row overlaps ("can suppress others") overlaps them; the loop runs while the matrix sum
 * drops by more than thr; a box is suppressed when a remaining row overlaps it.
 * keep[j] receives any(box_j > 0) && !suppressed[j].  Returns the number of iterations. */
int oracle_matrix_nms(const float* boxes, int n, float thr, unsigned char* keep) {
    unsigned char* row_alive = (unsigned char*)malloc((size_t)n);
    unsigned char* cso = (unsigned char*)malloc((size_t)n);
    unsigned char* kill = (unsigned char*)malloc((size_t)n);
    /* an all-zero box has IoU 0 with everything: its row is empty from the start */
    for (int i = 0; i < n; ++i) row_alive[i] = !is_zero(boxes + 4 * i);
    double sum = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n && row_alive[i]; ++j) {
            float v = iou_ij(boxes + 4 * i, boxes + 4 * j);
            if (v >= thr) sum += v;
        }
    int cond = 1, it = 0;
    while (cond) {
        for (int i = 0; i < n; ++i) {
            cso[i] = 1;
            if (is_zero(boxes + 4 * i)) continue;
            for (int k = 0; k < i; ++k)
                if (row_alive[k] && iou_ij(boxes + 4 * k, boxes + 4 * i) >= thr) { cso[i] = 0; break; }
        }
        for (int i = 0; i < n; ++i) {
            kill[i] = 0;
            if (is_zero(boxes + 4 * i)) continue;
            for (int k = 0; k < i; ++k)
                if (row_alive[k] && cso[k] && iou_ij(boxes + 4 * k, boxes + 4 * i) >= thr) { kill[i] = 1; break; }
        }
        for (int i = 0; i < n; ++i)
            if (kill[i]) row_alive[i] = 0;
        double sum_new = 0.0;
        for (int i = 0; i < n; ++i) {
            if (!row_alive[i]) continue;
            for (int j = i + 1; j < n; ++j) {
                float v = iou_ij(boxes + 4 * i, boxes + 4 * j);
                if (v >= thr) sum_new += v;
            }
        }
        cond = (sum - sum_new) > (double)thr;
        sum = sum_new;
        ++it;
    }
    for (int j = 0; j < n; ++j) {
        int suppressed = 0;
        for (int i = 0; i < j && !is_zero(boxes + 4 * j); ++i)
            if (row_alive[i] && iou_ij(boxes + 4 * i, boxes + 4 * j) >= thr) { suppressed = 1; break; }
        const float* b = boxes + 4 * j;
        int positive = (b[0] > 0.0f) || (b[1] > 0.0f) || (b[2] > 0.0f) || (b[3] > 0.0f);
        keep[j] = (unsigned char)(positive && !suppressed);
    }
    free(row_alive); free(cso); free(kill);
    return it;
}
