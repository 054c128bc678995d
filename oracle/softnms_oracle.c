/* TEST INFRASTRUCTURE ONLY - see wt_oracle.h.  CPU restatement of the reference's soft-NMS / NMS /
 * weighted-fusion ensemble path (float64 throughout, like the reference's torch.float64 CPU tensors).
 * Compile with -ffp-contract=off.
 */
#include "wt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* torch.sort semantics for NaN: NaN is the largest value (sorted last in ascending order) */
static int gt_nan_last(double a, double b)
{
    const int an = a != a, bn = b != b;
    if (an || bn) return an && !bn;
    return a > b;
}

/* stable ascending argsort (scores.sort(0), box_utils.py:324; tie order is unspecified upstream - stable here) */
static void argsort_ascending(const double* v, int n, int64_t* idx)
{
    int i, j;
    for (i = 0; i < n; ++i) idx[i] = i;
    /* insertion sort keeps this restatement obviously stable; n is at most a few thousand in tests */
    if (n < 64) {
        for (i = 1; i < n; ++i) {
            int64_t k = idx[i];
            for (j = i - 1; j >= 0 && gt_nan_last(v[idx[j]], v[k]); --j) idx[j + 1] = idx[j];
            idx[j + 1] = k;
        }
        return;
    }
    /* bottom-up merge sort (stable) */
    int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    int width;
    for (width = 1; width < n; width *= 2) {
        for (i = 0; i < n; i += 2 * width) {
            int a = i, b = i + width < n ? i + width : n, e = i + 2 * width < n ? i + 2 * width : n;
            int p = a, q = b, o = a;
            while (p < b && q < e) tmp[o++] = gt_nan_last(v[idx[p]], v[idx[q]]) ? idx[q++] : idx[p++];
            while (p < b) tmp[o++] = idx[p++];
            while (q < e) tmp[o++] = idx[q++];
        }
        memcpy(idx, tmp, sizeof(int64_t) * (size_t)n);
    }
    free(tmp);
}

static double clamp01(double v) { return v < 0. ? 0. : (v > 1. ? 1. : v); }   /* NaN passes through */

/* box_utils.py:307-395, soft branch */
int wto_softnms(const double* boxes4, const double* scores, int n, double overlap, double cut,
                double conf_thresh, int top_k, int64_t* keep, double* out_scores, int* n_keep)
{
    *n_keep = 0;
    if (n <= 0) return 0;
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    double* ss = (double*)malloc(sizeof(double) * (size_t)n);
    double* area = (double*)malloc(sizeof(double) * (size_t)n);
    int i, j, L = n;
    argsort_ascending(scores, n, idx);
    if (top_k > 0 && top_k < n) {                                   /* box_utils.py:325-327 */
        memmove(idx, idx + (n - top_k), sizeof(int64_t) * (size_t)top_k);
        L = top_k;
    }
    for (i = 0; i < L; ++i) ss[i] = scores[idx[i]];
    for (i = 0; i < n; ++i) area[i] = (boxes4[4 * i + 2] - boxes4[4 * i + 0]) * (boxes4[4 * i + 3] - boxes4[4 * i + 1]);
    while (L > 1) {
        const int64_t b = idx[L - 1];
        keep[*n_keep] = b; out_scores[*n_keep] = ss[L - 1]; ++(*n_keep);
        --L;
        const double bx1 = boxes4[4 * b], by1 = boxes4[4 * b + 1], bx2 = boxes4[4 * b + 2], by2 = boxes4[4 * b + 3];
        for (j = 0; j < L; ++j) {
            const int64_t o = idx[j];
            double xx1 = boxes4[4 * o]; if (xx1 < bx1) xx1 = bx1;          /* clamp(min=) */
            double yy1 = boxes4[4 * o + 1]; if (yy1 < by1) yy1 = by1;
            double xx2 = boxes4[4 * o + 2]; if (xx2 > bx2) xx2 = bx2;      /* clamp(max=) */
            double yy2 = boxes4[4 * o + 3]; if (yy2 > by2) yy2 = by2;
            double w = xx2 - xx1; if (w < 0.) w = 0.;
            double h = yy2 - yy1; if (h < 0.) h = 0.;
            double inter = w * h;
            double uni = (area[o] - inter) + area[b];                       /* box_utils.py:366 association */
            double iou = inter / uni;
            double wgt = clamp01((cut - iou) / (cut - overlap));            /* box_utils.py:373 */
            ss[j] = ss[j] * wgt;
        }
        int o2 = 0;
        for (j = 0; j < L; ++j) if (ss[j] >= conf_thresh) { idx[o2] = idx[j]; ss[o2] = ss[j]; ++o2; }   /* :379-381 */
        L = o2;
    }
    if (L > 0) { keep[*n_keep] = idx[L - 1]; out_scores[*n_keep] = ss[L - 1]; ++(*n_keep); }
    free(idx); free(ss); free(area);
    return 0;
}

/* box_utils.py:329-333 hard branch -> torchvision.ops.nms (absent here; restated: descending score,
 * suppress j when IoU(i,j) > thr, IoU = inter / (area_i + area_j - inter), areas without +1). */
int wto_hardnms(const double* boxes4, const double* scores, int n, double overlap, int top_k,
                int64_t* keep, double* out_scores, int* n_keep)
{
    *n_keep = 0;
    if (n <= 0) return 0;
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    char* dead = (char*)calloc((size_t)n, 1);
    int i, j, L = n;
    argsort_ascending(scores, n, idx);
    if (top_k > 0 && top_k < n) { memmove(idx, idx + (n - top_k), sizeof(int64_t) * (size_t)top_k); L = top_k; }
    for (i = L - 1; i >= 0; --i) {
        if (dead[i]) continue;
        const int64_t b = idx[i];
        keep[*n_keep] = b; out_scores[*n_keep] = scores[b]; ++(*n_keep);
        const double ab = (boxes4[4 * b + 2] - boxes4[4 * b]) * (boxes4[4 * b + 3] - boxes4[4 * b + 1]);
        for (j = i - 1; j >= 0; --j) {
            if (dead[j]) continue;
            const int64_t o = idx[j];
            double xx1 = fmax(boxes4[4 * o], boxes4[4 * b]), yy1 = fmax(boxes4[4 * o + 1], boxes4[4 * b + 1]);
            double xx2 = fmin(boxes4[4 * o + 2], boxes4[4 * b + 2]), yy2 = fmin(boxes4[4 * o + 3], boxes4[4 * b + 3]);
            double w = fmax(xx2 - xx1, 0.), h = fmax(yy2 - yy1, 0.);
            double inter = w * h;
            double ao = (boxes4[4 * o + 2] - boxes4[4 * o]) * (boxes4[4 * o + 3] - boxes4[4 * o + 1]);
            double iou = inter / (ab + ao - inter);
            if (iou > overlap) dead[j] = 1;
        }
    }
    free(idx); free(dead);
    return 0;
}

/* nn/tta.py:8-19 nms_detections (+ box_utils.py:32-35 point_form, :65-69 center_size) */
int wto_nms_detections(const double* dets5, int n, double iou_thresh, int soft, double cut,
                       double* out5, int* n_out)
{
    *n_out = 0;
    if (n <= 0) return 0;
    double* boxes = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    double* sc = (double*)malloc(sizeof(double) * (size_t)n);
    int64_t* keep = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    double* ks = (double*)malloc(sizeof(double) * (size_t)n);
    int i, nk = 0;
    for (i = 0; i < n; ++i) {
        const double* d = dets5 + 5 * (size_t)i;
        sc[i] = d[0];
        boxes[4 * i + 0] = d[1] - d[3] * 0.5;
        boxes[4 * i + 1] = d[2] - d[4] * 0.5;
        boxes[4 * i + 2] = d[1] + d[3] * 0.5;
        boxes[4 * i + 3] = d[2] + d[4] * 0.5;
    }
    if (soft) wto_softnms(boxes, sc, n, iou_thresh, cut, 0., 0, keep, ks, &nk);
    else wto_hardnms(boxes, sc, n, iou_thresh, 0, keep, ks, &nk);
    for (i = 0; i < nk; ++i) {
        const double* b = boxes + 4 * keep[i];
        double* o = out5 + 5 * (size_t)i;
        o[0] = ks[i];
        o[1] = (b[0] + b[2]) * 0.5;
        o[2] = (b[1] + b[3]) * 0.5;
        o[3] = b[2] - b[0];
        o[4] = b[3] - b[1];
    }
    *n_out = nk;
    free(boxes); free(sc); free(keep); free(ks);
    return 0;
}

/* box_utils.py:126-140 jaccard_bbox on centre-form boxes [cx,cy,w,h] (+ :72-92 intersect, :114-123 iou_bbox) */
static double jaccard_center(const double a[4], const double b[4])
{
    double ax1 = a[0] - a[2] * 0.5, ay1 = a[1] - a[3] * 0.5, ax2 = a[0] + a[2] * 0.5, ay2 = a[1] + a[3] * 0.5;
    double bx1 = b[0] - b[2] * 0.5, by1 = b[1] - b[3] * 0.5, bx2 = b[0] + b[2] * 0.5, by2 = b[1] + b[3] * 0.5;
    double mx = (ax2 < bx2 || ax2 != ax2) ? ax2 : bx2;     /* torch.min / torch.max propagate NaN */
    if (bx2 != bx2) mx = bx2;
    double my = (ay2 < by2 || ay2 != ay2) ? ay2 : by2;
    if (by2 != by2) my = by2;
    double nx = (ax1 > bx1 || ax1 != ax1) ? ax1 : bx1;
    if (bx1 != bx1) nx = bx1;
    double ny = (ay1 > by1 || ay1 != ay1) ? ay1 : by1;
    if (by1 != by1) ny = by1;
    double iw = mx - nx; if (iw < 0.) iw = 0.;
    double ih = my - ny; if (ih < 0.) ih = 0.;
    double inter = iw * ih;
    double area_a = a[2] * a[3];
    double area_b = b[2] * b[3];
    double uni = (area_a + area_b) - inter;
    return inter / uni;
}

/* nn/tta.py:22-66 merge_detections */
int wto_merge_detections(const double* dets5, const int* sizes, int k_inputs, double nms_thresh,
                         double* out5, int cap, int* n_out)
{
    int total = 0, k, i, j, m = 0;
    for (k = 0; k < k_inputs; ++k) total += sizes[k];
    *n_out = 0;
    if (total > cap) return 4;
    double* res = out5;                                   /* results grow in place, at most `total` rows */
    double* oth = (double*)malloc(sizeof(double) * 5 * (size_t)(total + 1));
    double* iou = (double*)malloc(sizeof(double) * (size_t)(total + 1));
    int* am = (int*)malloc(sizeof(int) * (size_t)(total + 1));
    double* snap = (double*)malloc(sizeof(double) * 5 * (size_t)(total + 1));
    const double K = (double)k_inputs;
    int off = 0;
    /* tta.py:34-36 */
    m = sizes[0];
    for (i = 0; i < m; ++i) {
        const double* d = dets5 + 5 * (size_t)i;
        double* r = res + 5 * (size_t)i;
        r[0] = d[0] / K;
        for (j = 1; j < 5; ++j) r[j] = d[j] * r[0];
    }
    off = sizes[0];
    for (k = 1; k < k_inputs; ++k) {
        const int no = sizes[k];
        if (no > 0) {
            for (i = 0; i < no; ++i) {                       /* tta.py:40-41 */
                const double* d = dets5 + 5 * (size_t)(off + i);
                double* o = oth + 5 * (size_t)i;
                o[0] = d[0] / K;
                for (j = 1; j < 5; ++j) o[j] = d[j] * o[0];
            }
            if (m > 0) {
                for (i = 0; i < no; ++i) {                   /* tta.py:44-47: per `other`, max IoU over results */
                    double ob[4], best = 0.; int bi = 0, first = 1, r;
                    for (j = 0; j < 4; ++j) ob[j] = oth[5 * i + 1 + j] / oth[5 * i];
                    for (r = 0; r < m; ++r) {
                        double mb[4], v;
                        for (j = 0; j < 4; ++j) mb[j] = res[5 * r + 1 + j] / res[5 * r];
                        v = jaccard_center(mb, ob);
                        if (first) { best = v; bi = r; first = 0; }
                        else if (best == best && (v > best || v != v)) { best = v; bi = r; }   /* first max; NaN wins */
                    }
                    iou[i] = best; am[i] = bi;
                }
                /* tta.py:50-55: results[idx_matched] += o_matched (buffered: duplicates keep the last one) */
                memcpy(snap, res, sizeof(double) * 5 * (size_t)m);
                for (i = 0; i < no; ++i)
                    if (iou[i] >= nms_thresh)
                        for (j = 0; j < 5; ++j) res[5 * am[i] + j] = snap[5 * am[i] + j] + oth[5 * i + j];
                /* tta.py:57-59: append unmatched */
                int m0 = m;
                for (i = 0; i < no; ++i)
                    if (iou[i] < nms_thresh) { memcpy(res + 5 * (size_t)m, oth + 5 * (size_t)i, sizeof(double) * 5); ++m; }
                (void)m0;
            } else {                                         /* tta.py:60-62 */
                memcpy(res, oth, sizeof(double) * 5 * (size_t)no);
                m = no;
            }
        }
        off += no;
    }
    for (i = 0; i < m; ++i) for (j = 1; j < 5; ++j) res[5 * i + j] = res[5 * i + j] / res[5 * i];   /* tta.py:65 */
    *n_out = m;
    free(oth); free(iou); free(am); free(snap);
    return 0;
}

/* ensemble.py:50-64 (before the min_score / astype(int) / round post-processing, which is host formatting) */
int wto_ensemble_groups(const double* dets5, const int64_t* group_offsets, const int32_t* input_sizes,
                        int64_t n_groups, int k_inputs, int method, double iou_thresh, double cut,
                        double* out5, int64_t* out_counts)
{
    int64_t g;
    int rc = 0;
    for (g = 0; g < n_groups && rc == 0; ++g) {
        const int64_t o0 = group_offsets[g];
        const int n = (int)(group_offsets[g + 1] - o0);
        double* in = (double*)malloc(sizeof(double) * 5 * (size_t)(n + 1));
        double* out = out5 + 5 * (size_t)o0;
        int i, no = 0;
        for (i = 0; i < n; ++i) {                            /* ensemble.py:19-22 lxly2cxcy */
            const double* d = dets5 + 5 * (size_t)(o0 + i);
            in[5 * i] = d[0];
            in[5 * i + 1] = d[1] + d[3] / 2;
            in[5 * i + 2] = d[2] + d[4] / 2;
            in[5 * i + 3] = d[3];
            in[5 * i + 4] = d[4];
        }
        if (method == 0) {
            int* sz = (int*)malloc(sizeof(int) * (size_t)k_inputs);
            for (i = 0; i < k_inputs; ++i) sz[i] = input_sizes[g * k_inputs + i];
            rc = wto_merge_detections(in, sz, k_inputs, iou_thresh, out, n, &no);
            free(sz);
        } else {
            rc = wto_nms_detections(in, n, iou_thresh, method == 2, cut, out, &no);
        }
        for (i = 0; i < no; ++i) {                           /* ensemble.py:25-28 cxcy2lxly */
            out[5 * i + 1] = out[5 * i + 1] - out[5 * i + 3] / 2;
            out[5 * i + 2] = out[5 * i + 2] - out[5 * i + 4] / 2;
        }
        out_counts[g] = no;
        free(in);
    }
    return rc;
}
