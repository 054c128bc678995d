/* libwaymotrack - C ABI of the MI355X-native detect -> ensemble -> SORT hot path.
 *
 * The reference (xuyuan/waymo_2d_tracking) is pure Python and has no FFI layer: its "operator API" is a
 * set of Python call signatures.  Each entry point below replaces one of those
 * call sites; the Python shims in waymo_2d_tracking_amd/ keep the reference's names and argument
 * meaning and reach this library through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions: C linkage, plain pointers and sizes, no exceptions; every function returns an int status
 * (WT_OK == 0) and wt_last_error() gives the message of the calling thread's last failure.  All compute
 * runs in HIP kernels on the current HIP device (gfx950); there is NO CPU fallback: without a usable GPU
 * the functions return WT_ERR_NO_DEVICE.
 *   *_host entry points take host buffers (they stage through device memory internally);
 *   *_dev  entry points take device pointers plus a hipStream_t (passed as void*), never synchronise,
 *          and use only caller-provided workspace (query the size with the matching *_workspace call).
 * Paths below are relative to the reference repository root.
 */
#ifndef WAYMOTRACK_H
#define WAYMOTRACK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WT_OK 0
#define WT_ERR_INVALID 1    /* bad argument */
#define WT_ERR_NO_DEVICE 2  /* no HIP device / wrong architecture */
#define WT_ERR_HIP 3        /* HIP runtime error */
#define WT_ERR_CAPACITY 4   /* caller buffer or workspace too small */
#define WT_ERR_NUMERIC 5    /* iteration guard hit inside a kernel (e.g. NaN cost matrix) */

const char* wt_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int wt_abi_version(void);
/* Number of visible HIP devices and name/arch of the current one ("gfx950..."). */
int wt_device_info(int* n_devices, char* arch, int arch_cap, int* n_cu);

/* =================================================================================================
 * SORT  (tracking/sort/sort.py, tracking/sort/tracker_sort.py, tracking/utils.py)
 * ================================================================================================= */

/* Process-global track-ID counter: KalmanBoxTracker.count (tracking/sort/sort.py:86,140-141). */
typedef struct wt_idctr wt_idctr;
wt_idctr* wt_idctr_create(int64_t start);
int64_t wt_idctr_get(const wt_idctr* c);
void wt_idctr_set(wt_idctr* c, int64_t value);
void wt_idctr_destroy(wt_idctr* c);

/* Sort(max_age, min_hits) - tracking/sort/sort.py:234-242.  State lives in device memory.
 * ctr may be NULL (private counter starting at 0). */
typedef struct wt_sort wt_sort;
int wt_sort_create(int max_age, int min_hits, wt_idctr* ctr, wt_sort** out);
void wt_sort_destroy(wt_sort* s);
/* Sort.update(dets, iou_threshold) - tracking/sort/sort.py:244-296.
 * dets5: (n,5) float32 rows [x1,y1,x2,y2,score] (n may be 0: "must be called once per frame even with empty
 * detections", :248).  out6: up to cap rows [x1,y1,x2,y2,id+1,confidence] float64, newest track first. */
int wt_sort_update_host(wt_sort* s, const float* dets5, int n, double iou_threshold,
                        double* out6, int cap, int* n_out);
/* Number of live tracks (len(self.trackers)) after the last update. */
int wt_sort_num_tracks(const wt_sort* s);
/* Debug/test hook: current track list in list order (ids, state x[7], covariance P[49], row-major). */
int wt_sort_state_host(wt_sort* s, int cap, int64_t* ids, double* x7, double* P49, int* n_tracks);

/* MultiClassTrackerSort(max_age, min_hits) - tracking/sort/tracker_sort.py:10-51: one Sort per class, created the first
 * time the class is seen; every known class is updated once per call, in first-seen order.
 *   dets6          : (n,6) float64 rows [x1,y1,x2,y2,confidence,class] (class = integer >= 1), cast to float32 like
 *                    np.array(..., dtype=np.float32) (:45)
 *   iou_thresholds : iou_thresholds[class-1] (:49); a class beyond n_thresholds is WT_ERR_INVALID (IndexError there)
 *   out6           : rows [x1,y1,x2,y2,id+1,confidence] of all classes, grouped by class in first-seen order;
 *   out_classes[k] / out_counts[k] : class id and row count of group k (k < *n_classes <= class_cap)
 * wt_mct_tracker returns the class's Sort (borrowed; NULL if the class has not been seen) for wt_sort_state_host. */
typedef struct wt_mct wt_mct;
int wt_mct_create(int max_age, int min_hits, wt_idctr* ctr, wt_mct** out);
void wt_mct_destroy(wt_mct* m);
int wt_mct_num_classes(const wt_mct* m);
wt_sort* wt_mct_tracker(wt_mct* m, int class_id);
int wt_mct_track_host(wt_mct* m, const double* dets6, int n, const double* iou_thresholds, int n_thresholds,
                      double* out6, int cap, int32_t* out_classes, int32_t* out_counts, int class_cap, int* n_classes);

/* associate_detections_to_trackers(detections, trackers, iou_threshold) - tracking/sort/sort.py:193-230
 * (IoU matrix :33-47,201-205 + sklearn 0.22.2 linear_assignment :206 + threshold filter :218-224).
 * dets5 (n,5) f32, trks4 (t,4) f64.  matches: (det,trk) pairs sorted by det; unmatched lists in the
 * reference's order.  Buffers: matches 2*min(n,t), unmatched_dets n, unmatched_trks t ints. */
int wt_associate_host(const float* dets5, int n, const double* trks4, int t, double iou_threshold,
                      int* matches, int* n_matches, int* unmatched_dets, int* n_unmatched_dets,
                      int* unmatched_trks, int* n_unmatched_trks);

/* linear_assignment(X) of scikit-learn 0.22.2 (call site tracking/sort/sort.py:206) on a float32 cost matrix
 * (n_rows x n_cols, row-major).  pairs: (row,col) sorted by row, 2*min(n_rows,n_cols) ints. */
int wt_linear_assignment_f32_host(const float* cost, int n_rows, int n_cols, int* pairs, int* n_pairs);

/* Batched tracking of every (segment,camera) stream of a detections file:
 * read_data_file filters (tracking/utils.py:79,86) + track_sort (tracking/utils.py:25-60) +
 * MultiClassTrackerSort.track (tracking/sort/tracker_sort.py:22-51) in the stream order of
 * tracking/track.py:43-47, including the global ID order of tracking/sort/sort.py:86.
 *
 * Detections are SoA, sorted by (stream, frame) with the input order preserved inside a frame:
 *   x,y,w,h,score float64 (JSON numbers), category int32 in 1..n_classes.
 * frame_det_offsets (n_frames+1) and stream_frame_offsets (n_streams+1) are CSR offsets; frames of a stream
 * are in ascending frame-id order; a frame whose detections are all filtered still ticks the trackers.
 * clip_w/clip_h: per-stream image size (tracking/utils.py:11-22); <= 0 disables clipping, the w/h<1 drop and
 * the confidence clip.  score_threshold / iou_threshold: n_classes entries indexed by category-1.
 * id_base: value of the global ID counter before this call (rank offset in multi-GPU runs).
 * Outputs hold at most n_dets rows, in the reference's output order:
 *   out_frame (global frame index), out_category, out_bbox4 [x1,y1,w,h], out_score, out_object_id.
 * n_births = number of track IDs consumed. */
typedef struct wt_track_params {
    int32_t max_age;
    int32_t min_hits;
    int32_t n_classes;
    int32_t reserved;
    const double* score_threshold;   /* host pointers, n_classes entries */
    const double* iou_threshold;
} wt_track_params;

int wt_track_streams_host(int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                          const double* score, const int32_t* category,
                          int64_t n_frames, const int64_t* frame_det_offsets,
                          int32_t n_streams, const int64_t* stream_frame_offsets,
                          const double* clip_w, const double* clip_h,
                          const wt_track_params* params, int64_t id_base,
                          int64_t* out_frame, int32_t* out_category, double* out_bbox4, double* out_score,
                          int64_t* out_object_id, int64_t* n_out, int64_t* n_births);

/* Device-resident form.  All array arguments are device pointers except params (host struct with host
 * threshold arrays) ; max_frame_dets = max detections in any frame (sizes the per-tracker state).
 * n_out_dev / n_births_dev: device int64 scalars.  Stream-ordered on `stream`; no host synchronisation. */
size_t wt_track_streams_workspace(int64_t n_dets, int64_t n_frames, int32_t n_streams, int64_t max_frame_dets,
                                  const wt_track_params* params);
int wt_track_streams_dev(int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                         const double* score, const int32_t* category,
                         int64_t n_frames, const int64_t* frame_det_offsets,
                         int32_t n_streams, const int64_t* stream_frame_offsets,
                         const double* clip_w, const double* clip_h, int64_t max_frame_dets,
                         const wt_track_params* params, int64_t id_base,
                         int64_t* out_frame, int32_t* out_category, double* out_bbox4, double* out_score,
                         int64_t* out_object_id, int64_t* n_out_dev, int64_t* n_births_dev,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Streaming form of the same tracker (online detect -> track: tracking/utils.py:29-36 keeps ONE MultiClassTrackerSort per
 * stream alive while the frames arrive).  The trackers of n_streams streams live in a caller-owned device block `state`;
 * every wt_track_chunk_dev call feeds the next frames of each stream (same SoA / CSR layout as wt_track_streams_dev; a
 * stream may have zero frames in a chunk) and continues exactly where the previous chunk stopped, so that the rows of all
 * chunks together equal ONE wt_track_streams_dev call over the concatenated frames (tests/test_gpu_sort.py).
 *   out_frame     : frame index inside THIS chunk's CSR;
 *   out_local_id  : 0-based ordinal of the track's birth inside its stream (over all chunks).  The reference's object id
 *                   (process-global counter, sort.py:86,140-141) is id_base + births of the streams in front + ordinal + 1
 *                   and is known once those streams are complete: wt_track_global_ids_dev does that conversion for any
 *                   set of collected rows (row_stream = stream index of each row; stream_birth_prefix = n_streams + 1
 *                   int64 of device scratch, receives the exclusive prefix sum of births per stream).
 * max_frame_dets and params must be the same for every call on one state. */
size_t wt_track_state_bytes(int32_t n_streams, int64_t max_frame_dets, const wt_track_params* params);
int wt_track_state_init_dev(void* state, size_t state_bytes, int32_t n_streams, int64_t max_frame_dets,
                            const wt_track_params* params, void* stream);
size_t wt_track_chunk_workspace(int64_t n_dets, int64_t n_frames, int32_t n_streams, int64_t max_frame_dets,
                                const wt_track_params* params);
int wt_track_chunk_dev(void* state, size_t state_bytes,
                       int64_t n_dets, const double* x, const double* y, const double* w, const double* h,
                       const double* score, const int32_t* category,
                       int64_t n_frames, const int64_t* frame_det_offsets,
                       int32_t n_streams, const int64_t* stream_frame_offsets,
                       const double* clip_w, const double* clip_h, int64_t max_frame_dets,
                       const wt_track_params* params,
                       int64_t* out_frame, int32_t* out_category, double* out_bbox4, double* out_score,
                       int64_t* out_local_id, int64_t* n_out_dev, int64_t* n_births_dev,
                       void* workspace, size_t workspace_bytes, void* stream);
int wt_track_global_ids_dev(const void* state, size_t state_bytes, int32_t n_streams, int64_t max_frame_dets,
                            const wt_track_params* params, int64_t n_rows, const int32_t* row_stream,
                            const int64_t* local_id, int64_t id_base, int64_t* out_object_id,
                            int64_t* stream_birth_prefix, void* stream);

/* Native COCO-JSON I/O around the tracking stage (host code).
 * wt_detfile_read = json.load + read_data_file (tracking/utils.py:63-96: "annotations" wrapper, w/h < 1 and
 * per-class score filters, frame keys kept for fully filtered frames, missing score = 1.0) + the stream / frame
 * ordering of tracking/track.py:43-47 and utils.py:31, straight into the SoA / CSR layout of wt_track_streams_*.
 * The accessors return pointers owned by the handle (valid until wt_detfile_free). */
typedef struct wt_detfile wt_detfile;
int wt_detfile_read(const char* path, const double* score_threshold, int n_classes, wt_detfile** out);
void wt_detfile_free(wt_detfile* f);
int64_t wt_detfile_num_dets(const wt_detfile* f);
int64_t wt_detfile_num_frames(const wt_detfile* f);
int32_t wt_detfile_num_streams(const wt_detfile* f);
const double* wt_detfile_x(const wt_detfile* f);
const double* wt_detfile_y(const wt_detfile* f);
const double* wt_detfile_w(const wt_detfile* f);
const double* wt_detfile_h(const wt_detfile* f);
const double* wt_detfile_score(const wt_detfile* f);
const int32_t* wt_detfile_category(const wt_detfile* f);
const int64_t* wt_detfile_frame_det_offsets(const wt_detfile* f);
const int64_t* wt_detfile_stream_frame_offsets(const wt_detfile* f);
const int64_t* wt_detfile_frame_ids(const wt_detfile* f);
const char* wt_detfile_segment(const wt_detfile* f, int32_t stream);
const char* wt_detfile_camera(const wt_detfile* f, int32_t stream);
/* json.dump of the tracking rows (tracking/utils.py:52-58, tracking/track.py:50), byte-compatible with Python:
 * [{"image_id": "<segment>/<frame>/<camera>", "bbox": [x1, y1, w, h], "score": s, "category_id": c,
 *   "object_id": "<id>"}, ...]; floats in Python repr form.  Rows as produced by wt_track_streams_*. */
int wt_tracks_write_json(const char* path, const wt_detfile* f, int64_t n, const int64_t* frame, const int32_t* category,
                         const double* bbox4, const double* score, const int64_t* object_id);
/* Generic detection JSON = the wire format between inference, ensemble and tracking (detnet/data/coco.py:229-252,
 * detnet/ensemble.py:59-63,79,159-160): a list of {"image_id": str, "category_id": int, "bbox": [x, y, w, h], "score": float}.
 * wt_detjson_read = json.load of such a file into columns (image ids interned in first-appearance order; every entry must carry
 * the four keys, like convert_submission's det['score'] / det['bbox'] lookups, ensemble.py:35-45).
 * wt_detections_write_json = json.dump of rows whose boxes are integers (coco.py:250 int(v), ensemble.py:62 astype(int)) and whose
 * scores were rounded by the caller; key order image_id, category_id, bbox, score; strings escaped like json.dumps (ensure_ascii).
 * image ids: UTF-8 blob + (n_images + 1) offsets. */
typedef struct wt_detjson wt_detjson;
int wt_detjson_read(const char* path, wt_detjson** out);
void wt_detjson_free(wt_detjson* f);
int64_t wt_detjson_num_rows(const wt_detjson* f);
int32_t wt_detjson_num_images(const wt_detjson* f);
const int32_t* wt_detjson_image(const wt_detjson* f);
const int32_t* wt_detjson_category(const wt_detjson* f);
const double* wt_detjson_x(const wt_detjson* f);
const double* wt_detjson_y(const wt_detjson* f);
const double* wt_detjson_w(const wt_detjson* f);
const double* wt_detjson_h(const wt_detjson* f);
const double* wt_detjson_score(const wt_detjson* f);
const char* wt_detjson_image_id(const wt_detjson* f, int32_t i);
int wt_detections_write_json(const char* path, int64_t n, const int32_t* image_index, int32_t n_images, const char* image_id_blob,
                             const int64_t* image_id_offsets, const int32_t* category, const int64_t* bbox4, const double* score);
/* Python repr() of a double (shortest round-trip digits); returns the length or -1 if cap is too small. */
int wt_format_double(double v, char* out, int cap);

/* =================================================================================================
 * soft-NMS / NMS / weighted-fusion ensemble
 * (detnet/utils/box_utils.py, detnet/nn/tta.py, detnet/ensemble.py)
 * ================================================================================================= */

/* nms(boxes, scores, overlap, top_k, soft=True, conf_thresh, soft_nms_cut) - detnet/utils/box_utils.py:307-395.
 * boxes4 (n,4) xyxy float64.  keep: kept indices in descending original-score order, out_scores: decayed
 * scores; both need n entries. */
int wt_softnms_f64_host(const double* boxes4, const double* scores, int n, double overlap, double cut,
                        double conf_thresh, int top_k, int64_t* keep, double* out_scores, int* n_keep);
/* Hard branch (detnet/utils/box_utils.py:329-333 -> torchvision.ops.nms): greedy, IoU > overlap suppresses. */
int wt_hardnms_f64_host(const double* boxes4, const double* scores, int n, double overlap, int top_k,
                        int64_t* keep, double* out_scores, int* n_keep);

/* ensemble(image_id, detections, category_ids) - detnet/ensemble.py:50-64 - for G (image,category) groups at
 * once: lxly2cxcy (:19-22) -> merge_func -> cxcy2lxly (:25-28).  Rows are [score,x_left,y_top,w,h] float64.
 *   method 0: merge_detections weighted fusion (detnet/nn/tta.py:22-66)
 *   method 1: nms_detections hard NMS          (detnet/nn/tta.py:8-19, soft=False)
 *   method 2: nms_detections linear soft-NMS   (detnet/nn/tta.py:8-19, soft=True, soft_nms_cut)
 *   method | 16: rows are [score,cx,cy,w,h] on input and output, i.e. the bare merge_func call signature of
 *                detnet/nn/tta.py:8,22 without the ensemble.py:19-28 conversions.
 * group_offsets (G+1) rows CSR; inside a group rows are the K inputs concatenated in input-file order and
 * input_sizes (G*K) gives the rows each input contributed (only method 0 reads it; may be NULL otherwise).
 * out5 has the input's row capacity: group g writes out_counts[g] rows starting at row group_offsets[g].
 * The min_score filter / astype(int) / round(score,5) of ensemble.py:59-62 is left to the caller. */
int wt_ensemble_groups_host(const double* dets5, const int64_t* group_offsets, const int32_t* input_sizes,
                            int64_t n_groups, int k_inputs, int method, double iou_thresh, double soft_nms_cut,
                            double* out5, int64_t* out_counts);
size_t wt_ensemble_groups_workspace(int64_t n_rows, int64_t n_groups, int64_t max_group_rows);
int wt_ensemble_groups_dev(const double* dets5, const int64_t* group_offsets, const int32_t* input_sizes,
                           int64_t n_rows, int64_t n_groups, int64_t max_group_rows, int k_inputs, int method,
                           double iou_thresh, double soft_nms_cut, double* out5, int64_t* out_counts,
                           void* workspace, size_t workspace_bytes, void* stream);

/* --- Waymo Open Dataset protobuf emit (csrc/waymo_proto.hip; host code) -----------------------------------
 * metrics.Objects - and with submission != 0 the Submission envelope around it - written straight from columns: replaces
 * the per-object message building of /root/reference/coco_to_waymo.py:16-82 (create_pd_object / create_pb_submission) and
 * generate_prediction_for_metrics.py:43-80 (metrics_mode = 1: explicit zero z / height / heading, num_lidar_points_in_box =
 * 100).  context / id: UTF-8 blob + n+1 offsets; has_id NULL = every row has an id when id_offsets is given; score NULL = not
 * set (ground truth); det_level / trk_level NULL or 0 = not set; authors = n_authors NUL-terminated strings back to back.
 * Field numbers are recalled from the public waymo-open-dataset .proto files (the package is not in the image). */
int wt_waymo_objects_write(const char* path, int64_t n, const char* context_blob, const int64_t* context_offsets,
                           const int64_t* frame_timestamp_micros, const int32_t* camera_name, const double* bbox_xywh,
                           const double* score, const int32_t* label_type, const char* id_blob, const int64_t* id_offsets,
                           const uint8_t* has_id, const int32_t* det_level, const int32_t* trk_level, int metrics_mode,
                           int submission, int task, const char* account_name, const char* unique_method_name,
                           const char* authors, int n_authors, const char* affiliation, const char* description,
                           int sensor_type, int64_t* bytes_written);

#ifdef __cplusplus
}
#endif
#endif
