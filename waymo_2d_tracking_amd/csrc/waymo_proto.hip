// Waymo Open Dataset `metrics.Objects` / `Submission` protobuf emit (host code, no GPU): the wire format after
// the path - what /root/reference/coco_to_waymo.py:16-82 and generate_prediction_for_metrics.py:43-80 build object by object
// through the waymo_open_dataset Python protos, written here straight from columns.
//
// The waymo-open-dataset package is not in the image (nor vendored by the reference), so the schema below is RECALLED from
// the public .proto files (label.proto, dataset.proto, metrics.proto, submission.proto) - "parity unpinned" for the field
// numbers; the ENCODING is pinned: tests/test_waymo_proto.py builds the same schema with google.protobuf's dynamic messages
// and compares bytes.  proto2 semantics: a field that was set is written even when it is zero / empty; fields are written in
// field-number order; embedded messages are length-prefixed.
//
//   Label.Box    { center_x=1 center_y=2 center_z=3 width=4 length=5 height=6 heading=7 : double }
//   Label        { box=1 type=3(enum) id=4(string) detection_difficulty_level=5 tracking_difficulty_level=6
//                  num_lidar_points_in_box=7(int32) }
//   Object       { object=1(Label) score=2(float) context_name=4 frame_timestamp_micros=5(int64) camera_name=6(enum) }
//   Objects      { objects=1 (repeated Object) }
//   Submission   { task=1 account_name=2 unique_method_name=3 authors=4(repeated) affiliation=5 description=6 method_link=7
//                  sensor_type=8 number_past_frames_exclude_current=9 number_future_frames_exclude_current=10
//                  inference_results=11(Objects) }
#include "common.h"
#include <cstdio>
#include <string>
#include <vector>

namespace {

struct Buf {
    std::string s;
    void varint(uint64_t v) {
        while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; }
        s.push_back((char)v);
    }
    void tag(int field, int wire) { varint((uint64_t)field << 3 | (uint64_t)wire); }
    void f64(int field, double v) { tag(field, 1); s.append(reinterpret_cast<const char*>(&v), 8); }
    void f32(int field, float v) { tag(field, 5); s.append(reinterpret_cast<const char*>(&v), 4); }
    void i64(int field, int64_t v) { tag(field, 0); varint((uint64_t)v); }       // int32 / int64 / enum: negative = 10 bytes
    void bytes(int field, const char* p, size_t n) { tag(field, 2); varint(n); s.append(p, n); }
    void msg(int field, const Buf& m) { bytes(field, m.s.data(), m.s.size()); }
};

}  // namespace

extern "C" {

/* Serialise n objects (and, with `submission` != 0, the Submission envelope) to `path`.
 *   context / id : UTF-8 blobs + (n+1) offsets; id_offsets NULL = no object ids (detection); has_id (nullable) marks the
 *                  rows whose JSON entry carried an 'object_id' key (NULL = all of them)
 *   bbox_xywh    : n x 4 doubles (COCO [x, y, w, h]) -> center_x = x + w * 0.5, center_y = y + h * 0.5, length = w, width = h
 *   score        : n doubles or NULL (ground truth); stored as float like `o.score = score`
 *   label_type   : Label.Type values (the reference assigns category_id unchanged: 1 vehicle 2 pedestrian 3 sign 4 cyclist)
 *   det_level / trk_level : Label.DifficultyLevel or NULL; 0 = not set
 *   metrics_mode : generate_prediction_for_metrics.py - also writes center_z = height = heading = 0 and
 *                  num_lidar_points_in_box = 100 (:66-77)
 *   submission   : task (1 DETECTION_2D / 3 TRACKING_2D), account, method, authors (n_authors NUL-terminated strings back to
 *                  back), affiliation, description (NULL = not set), sensor_type 3 = CAMERA_ALL
 * *bytes_written receives the file size. */
int wt_waymo_objects_write(const char* path, int64_t n, const char* context_blob, const int64_t* context_offsets,
                           const int64_t* frame_timestamp_micros, const int32_t* camera_name, const double* bbox_xywh,
                           const double* score, const int32_t* label_type, const char* id_blob, const int64_t* id_offsets,
                           const uint8_t* has_id,
                           const int32_t* det_level, const int32_t* trk_level, int metrics_mode, int submission, int task,
                           const char* account_name, const char* unique_method_name, const char* authors, int n_authors,
                           const char* affiliation, const char* description, int sensor_type, int64_t* bytes_written) {
    if (!path || n < 0 || (n > 0 && (!context_blob || !context_offsets || !frame_timestamp_micros || !camera_name || !bbox_xywh ||
                                      !label_type))) {
        wt::set_error("wt_waymo_objects_write: bad argument");
        return WT_ERR_INVALID;
    }
    Buf objects;
    objects.s.reserve((size_t)n * 96 + 64);
    Buf box, label, obj;
    for (int64_t i = 0; i < n; ++i) {
        const double* b = bbox_xywh + 4 * i;
        box.s.clear();
        box.f64(1, b[0] + b[2] * 0.5);
        box.f64(2, b[1] + b[3] * 0.5);
        if (metrics_mode) box.f64(3, 0.0);
        box.f64(4, b[3]);                                  // width  = bbox[3]
        box.f64(5, b[2]);                                  // length = bbox[2]
        if (metrics_mode) { box.f64(6, 0.0); box.f64(7, 0.0); }
        label.s.clear();
        label.msg(1, box);
        label.i64(3, label_type[i]);
        if (id_offsets && (!has_id || has_id[i]))
            label.bytes(4, id_blob + id_offsets[i], (size_t)(id_offsets[i + 1] - id_offsets[i]));
        if (det_level && det_level[i]) label.i64(5, det_level[i]);
        if (trk_level && trk_level[i]) label.i64(6, trk_level[i]);
        if (metrics_mode) label.i64(7, 100);
        obj.s.clear();
        obj.msg(1, label);
        if (score) obj.f32(2, (float)score[i]);
        obj.bytes(4, context_blob + context_offsets[i], (size_t)(context_offsets[i + 1] - context_offsets[i]));
        obj.i64(5, frame_timestamp_micros[i]);
        obj.i64(6, camera_name[i]);
        objects.msg(1, obj);
    }
    const std::string* out = &objects.s;
    Buf sub;
    if (submission) {
        if (!account_name || !unique_method_name || n_authors < 0 || (n_authors > 0 && !authors) || !affiliation) {
            wt::set_error("wt_waymo_objects_write: submission fields missing");
            return WT_ERR_INVALID;
        }
        sub.s.reserve(objects.s.size() + 512);
        sub.i64(1, task);
        sub.bytes(2, account_name, strlen(account_name));
        sub.bytes(3, unique_method_name, strlen(unique_method_name));
        const char* a = authors;
        for (int k = 0; k < n_authors; ++k) { const size_t len = strlen(a); sub.bytes(4, a, len); a += len + 1; }
        sub.bytes(5, affiliation, strlen(affiliation));
        if (description) sub.bytes(6, description, strlen(description));
        sub.bytes(7, "", 0);                                // method_link = ""
        sub.i64(8, sensor_type);
        sub.i64(9, 0);
        sub.i64(10, 0);
        sub.msg(11, objects);
        out = &sub.s;
    }
    FILE* fp = fopen(path, "wb");
    if (!fp) { wt::set_error("cannot open %s for writing", path); return WT_ERR_INVALID; }
    const size_t w = fwrite(out->data(), 1, out->size(), fp);
    if (fclose(fp) != 0 || w != out->size()) { wt::set_error("short write to %s", path); return WT_ERR_INVALID; }
    if (bytes_written) *bytes_written = (int64_t)out->size();
    return WT_OK;
}

}  // extern "C"
