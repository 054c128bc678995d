// Native COCO-JSON I/O on both sides of the tracking stage (host code, no GPU):
//   * wt_detfile_read  = json.load + read_data_file (tracking/utils.py:63-96) + the packing loop of track.py:43-47 in
//     one pass: the detections land directly in the SoA / CSR layout wt_track_streams_* consumes;
//   * wt_tracks_write_json = json.dump of the rows built in tracking/utils.py:52-58, byte-compatible with Python
//     (float repr = shortest round-trip digits, Python's fixed/exponent switch, ", " / ": " separators).
// At Waymo scale (1e6-1e7 rows) the Python dict churn of the reference dominates the CLI wall time once SORT itself
// runs on the GPU; this keeps the file -> HBM -> file path native.
#include "common.h"
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>

namespace {

struct Parser {
    const char* p;
    const char* end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool lit(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    bool fail(const char* m) { if (err.empty()) { err = m; err += " at byte " + std::to_string((long)(p - (end - 0))); } return false; }
    bool string(std::string& out) {
        ws();
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (p + 1 >= end) return fail("bad escape");
                const char e = p[1];
                p += 2;
                switch (e) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': {
                        if (p + 4 > end) return fail("bad \\u escape");
                        unsigned cp = (unsigned)strtoul(std::string(p, 4).c_str(), nullptr, 16);
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += e;
                }
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool number(double& v) {
        ws();
        const char* s = p;
        if (p < end && (*p == '-' || *p == '+')) ++p;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) ++p;
        if (s == p) {
            if (end - p >= 3 && !strncmp(p, "NaN", 3)) { p += 3; v = NAN; return true; }
            if (end - p >= 8 && !strncmp(p, "Infinity", 8)) { p += 8; v = INFINITY; return true; }
            return fail("expected number");
        }
        auto r = std::from_chars(s + (*s == '+' ? 1 : 0), p, v);
        if (r.ec != std::errc()) {
            if (end - s >= 9 && !strncmp(s, "-Infinity", 9)) { p = s + 9; v = -INFINITY; return true; }
            return fail("bad number");
        }
        return true;
    }
    bool skip() {          // any JSON value
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '"') { std::string t; return string(t); }
        if (*p == '{') {
            ++p;
            if (lit('}')) return true;
            do { std::string k; if (!string(k) || !lit(':') || !skip()) return false; } while (lit(','));
            return lit('}') || fail("expected }");
        }
        if (*p == '[') {
            ++p;
            if (lit(']')) return true;
            do { if (!skip()) return false; } while (lit(','));
            return lit(']') || fail("expected ]");
        }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return true; }
        double d;
        return number(d);
    }
};

struct Rec {
    int stream;
    int64_t frame;
    double x, y, w, h, score;
    int32_t cat;
    bool keep;
};

}  // namespace

struct wt_detfile {
    std::vector<double> x, y, w, h, score;
    std::vector<int32_t> category;
    std::vector<int64_t> frame_det_offsets, stream_frame_offsets, frame_ids;
    std::vector<std::string> segment, camera;     // per stream
};


// generic detection JSON (list of {image_id, category_id, bbox [x, y, w, h], score}): ensemble inputs (detnet/ensemble.py:79)
struct wt_detjson {
    std::vector<int32_t> image, category;
    std::vector<double> x, y, w, h, score;
    std::vector<std::string> image_ids;            // first-appearance order
};

// json.dumps(str) with ensure_ascii=True
static void py_json_string(const std::string& s, std::string& out) {
    out += '"';
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s.data());
    const unsigned char* e = p + s.size();
    char tmp[16];
    while (p < e) {
        unsigned c = *p;
        if (c == '"') { out += "\\\""; ++p; }
        else if (c == '\\') { out += "\\\\"; ++p; }
        else if (c == '\n') { out += "\\n"; ++p; }
        else if (c == '\r') { out += "\\r"; ++p; }
        else if (c == '\t') { out += "\\t"; ++p; }
        else if (c == '\b') { out += "\\b"; ++p; }
        else if (c == '\f') { out += "\\f"; ++p; }
        else if (c < 0x20 || c == 0x7F) { snprintf(tmp, sizeof(tmp), "\\u%04x", c); out += tmp; ++p; }     // json: everything outside ' '..'~'

        else if (c < 0x80) { out += (char)c; ++p; }
        else {
            // UTF-8 -> code point -> \uXXXX (surrogate pair above the BMP)
            unsigned cp = 0xFFFD; int len = 1;
            if ((c & 0xE0) == 0xC0 && p + 1 < e) { cp = ((c & 0x1F) << 6) | (p[1] & 0x3F); len = 2; }
            else if ((c & 0xF0) == 0xE0 && p + 2 < e) { cp = ((c & 0x0F) << 12) | ((p[1] & 0x3F) << 6) | (p[2] & 0x3F); len = 3; }
            else if ((c & 0xF8) == 0xF0 && p + 3 < e) { cp = ((c & 0x07) << 18) | ((p[1] & 0x3F) << 12) | ((p[2] & 0x3F) << 6) | (p[3] & 0x3F); len = 4; }
            if (cp >= 0x10000) {
                cp -= 0x10000;
                snprintf(tmp, sizeof(tmp), "\\u%04x\\u%04x", 0xD800 + (cp >> 10), 0xDC00 + (cp & 0x3FF));
            } else {
                snprintf(tmp, sizeof(tmp), "\\u%04x", cp);
            }
            out += tmp;
            p += len;
        }
    }
    out += '"';
}

// Python float repr (shortest round-trip digits; exponent form iff decpt <= -4 or decpt > 16)
static int py_float_repr(double v, char* out) {
    if (v != v) return sprintf(out, "NaN");
    if (std::isinf(v)) return sprintf(out, v > 0 ? "Infinity" : "-Infinity");
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);
    *r.ptr = 0;
    const char* s = buf;
    char* o = out;
    if (*s == '-') { *o++ = '-'; ++s; }
    char digits[32];
    int nd = 0;
    const char* e = strchr(s, 'e');
    for (const char* q = s; q < e; ++q) if (*q != '.') digits[nd++] = *q;
    const int exp10 = atoi(e + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd;
    const int decpt = exp10 + 1;
    if (decpt > -4 && decpt <= 16) {
        if (decpt <= 0) {
            *o++ = '0'; *o++ = '.';
            for (int i = 0; i < -decpt; ++i) *o++ = '0';
            for (int i = 0; i < nd; ++i) *o++ = digits[i];
        } else if (decpt >= nd) {
            for (int i = 0; i < nd; ++i) *o++ = digits[i];
            for (int i = 0; i < decpt - nd; ++i) *o++ = '0';
            *o++ = '.'; *o++ = '0';
        } else {
            for (int i = 0; i < decpt; ++i) *o++ = digits[i];
            *o++ = '.';
            for (int i = decpt; i < nd; ++i) *o++ = digits[i];
        }
    } else {
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; for (int i = 1; i < nd; ++i) *o++ = digits[i]; }
        o += sprintf(o, "e%c%02d", exp10 < 0 ? '-' : '+', exp10 < 0 ? -exp10 : exp10);
    }
    *o = 0;
    return (int)(o - out);
}

extern "C" {

int wt_detfile_read(const char* path, const double* score_threshold, int n_classes, wt_detfile** out) {
    if (!path || !out || !score_threshold || n_classes < 1) { wt::set_error("wt_detfile_read: bad argument"); return WT_ERR_INVALID; }
    *out = nullptr;
    FILE* fp = fopen(path, "rb");
    if (!fp) { wt::set_error("cannot open %s", path); return WT_ERR_INVALID; }
    fseek(fp, 0, SEEK_END);
    const long size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::string text((size_t)size, '\0');
    const size_t got = fread(&text[0], 1, (size_t)size, fp);
    fclose(fp);
    if (got != (size_t)size) { wt::set_error("short read on %s", path); return WT_ERR_INVALID; }
    Parser ps{text.data(), text.data() + text.size(), {}};
    // [ ... ]  or  { ..., "annotations": [ ... ], ... }   (utils.py:66-67)
    bool wrapped = false;
    if (ps.lit('{')) {
        wrapped = true;
        bool found = false;
        if (!ps.lit('}')) {
            do {
                std::string k;
                if (!ps.string(k) || !ps.lit(':')) break;
                if (k == "annotations") { found = true; break; }
                if (!ps.skip()) break;
            } while (ps.lit(','));
        }
        if (!found) { wt::set_error("%s: no top-level list / \"annotations\" (%s)", path, ps.err.c_str()); return WT_ERR_INVALID; }
    }
    if (!ps.lit('[')) { wt::set_error("%s: expected a JSON list", path); return WT_ERR_INVALID; }
    std::vector<Rec> recs;
    std::unordered_map<std::string, int> seg_index;
    std::vector<std::string> seg_names;
    std::vector<std::vector<std::pair<std::string, int>>> seg_cams;      // per segment: (camera, stream id) in first-seen order
    std::vector<std::pair<int, int>> stream_seg_cam;                      // stream id -> (segment, camera slot)
    std::string key, image_id;
    if (!ps.lit(']')) {
        do {
            if (!ps.lit('{')) { ps.fail("expected {"); break; }
            Rec r{};
            r.score = 1.0;                                                 // utils.py:83 "assume ground truth"
            r.cat = 0;
            bool have_id = false, have_bbox = false;
            if (!ps.lit('}')) {
                do {
                    if (!ps.string(key) || !ps.lit(':')) break;
                    if (key == "image_id") { if (!ps.string(image_id)) break; have_id = true; }
                    else if (key == "category_id") { double d; if (!ps.number(d)) break; r.cat = (int32_t)d; }
                    else if (key == "score") { if (!ps.number(r.score)) break; }
                    else if (key == "bbox") {
                        double b[4];
                        if (!ps.lit('[')) { ps.fail("expected ["); break; }
                        bool ok = true;
                        for (int i = 0; i < 4 && ok; ++i) { ok = ps.number(b[i]) && (i == 3 || ps.lit(',')); }
                        if (!ok || !ps.lit(']')) { ps.fail("bbox must have 4 numbers"); break; }
                        r.x = b[0]; r.y = b[1]; r.w = b[2]; r.h = b[3];
                        have_bbox = true;
                    } else if (!ps.skip()) break;
                } while (ps.lit(','));
                if (!ps.err.empty() || !ps.lit('}')) { ps.fail("expected }"); break; }
            }
            if (!have_id || !have_bbox) { ps.fail("entry without image_id / bbox"); break; }
            // image_id = "<segment>/<frame>/<camera>"  (utils.py:71-72)
            const size_t a = image_id.find('/');
            const size_t b = a == std::string::npos ? a : image_id.find('/', a + 1);
            if (b == std::string::npos || image_id.find('/', b + 1) != std::string::npos) { ps.fail("image_id is not segment/frame/camera"); break; }
            const std::string seg = image_id.substr(0, a), cam = image_id.substr(b + 1);
            r.frame = strtoll(image_id.c_str() + a + 1, nullptr, 10);
            auto it = seg_index.find(seg);
            int si;
            if (it == seg_index.end()) { si = (int)seg_names.size(); seg_index.emplace(seg, si); seg_names.push_back(seg); seg_cams.emplace_back(); }
            else si = it->second;
            int stream = -1;
            for (auto& c : seg_cams[si]) if (c.first == cam) { stream = c.second; break; }
            if (stream < 0) { stream = (int)stream_seg_cam.size(); seg_cams[si].emplace_back(cam, stream); stream_seg_cam.emplace_back(si, (int)seg_cams[si].size() - 1); }
            r.stream = stream;
            // filters of utils.py:79,86 (the frame key exists even when the entry is dropped)
            r.keep = !(r.w < 1 || r.h < 1);
            if (r.keep) {
                if (r.cat < 1 || r.cat > n_classes) { ps.fail("category_id outside 1..n_classes"); break; }
                if (r.score < score_threshold[r.cat - 1]) r.keep = false;
            }
            recs.push_back(r);
        } while (ps.lit(','));
        if (ps.err.empty() && !ps.lit(']')) ps.fail("expected ]");
    }
    (void)wrapped;
    if (!ps.err.empty()) { wt::set_error("%s: %s", path, ps.err.c_str()); return WT_ERR_INVALID; }
    // stream order of track.py:43-47: segments in first-appearance order, cameras in first-appearance order inside
    std::vector<int> order;            // output stream index -> parse-time stream id
    for (size_t si = 0; si < seg_names.size(); ++si) for (auto& c : seg_cams[si]) order.push_back(c.second);
    std::vector<int> rank_of(order.size());
    for (size_t i = 0; i < order.size(); ++i) rank_of[order[i]] = (int)i;
    std::vector<size_t> idx(recs.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
        const int ra = rank_of[recs[a].stream], rb = rank_of[recs[b].stream];
        if (ra != rb) return ra < rb;
        return recs[a].frame < recs[b].frame;
    });
    wt_detfile* f = new wt_detfile;
    f->frame_det_offsets.push_back(0);
    f->stream_frame_offsets.push_back(0);
    int cur_stream = -1;
    int64_t cur_frame = 0;
    bool have_frame = false;
    for (size_t k = 0; k < idx.size(); ++k) {
        const Rec& r = recs[idx[k]];
        const int rs = rank_of[r.stream];
        if (rs != cur_stream || !have_frame || r.frame != cur_frame) {
            if (have_frame) f->frame_det_offsets.push_back((int64_t)f->x.size());
            while (cur_stream < rs) {                        // close finished streams
                if (cur_stream >= 0) f->stream_frame_offsets.push_back((int64_t)f->frame_ids.size());
                ++cur_stream;
            }
            f->frame_ids.push_back(r.frame);
            cur_frame = r.frame;
            have_frame = true;
        }
        if (r.keep) {
            f->x.push_back(r.x); f->y.push_back(r.y); f->w.push_back(r.w); f->h.push_back(r.h);
            f->score.push_back(r.score); f->category.push_back(r.cat);
        }
    }
    if (have_frame) f->frame_det_offsets.push_back((int64_t)f->x.size());
    if (cur_stream >= 0) f->stream_frame_offsets.push_back((int64_t)f->frame_ids.size());
    for (size_t i = 0; i < order.size(); ++i) {
        const auto& sc = stream_seg_cam[order[i]];
        f->segment.push_back(seg_names[sc.first]);
        f->camera.push_back(seg_cams[sc.first][sc.second].first);
    }
    *out = f;
    return WT_OK;
}

void wt_detfile_free(wt_detfile* f) { delete f; }
int64_t wt_detfile_num_dets(const wt_detfile* f) { return f ? (int64_t)f->x.size() : 0; }
int64_t wt_detfile_num_frames(const wt_detfile* f) { return f ? (int64_t)f->frame_ids.size() : 0; }
int32_t wt_detfile_num_streams(const wt_detfile* f) { return f ? (int32_t)f->segment.size() : 0; }
const double* wt_detfile_x(const wt_detfile* f) { return f->x.data(); }
const double* wt_detfile_y(const wt_detfile* f) { return f->y.data(); }
const double* wt_detfile_w(const wt_detfile* f) { return f->w.data(); }
const double* wt_detfile_h(const wt_detfile* f) { return f->h.data(); }
const double* wt_detfile_score(const wt_detfile* f) { return f->score.data(); }
const int32_t* wt_detfile_category(const wt_detfile* f) { return f->category.data(); }
const int64_t* wt_detfile_frame_det_offsets(const wt_detfile* f) { return f->frame_det_offsets.data(); }
const int64_t* wt_detfile_stream_frame_offsets(const wt_detfile* f) { return f->stream_frame_offsets.data(); }
const int64_t* wt_detfile_frame_ids(const wt_detfile* f) { return f->frame_ids.data(); }
const char* wt_detfile_segment(const wt_detfile* f, int32_t stream) { return f->segment[(size_t)stream].c_str(); }
const char* wt_detfile_camera(const wt_detfile* f, int32_t stream) { return f->camera[(size_t)stream].c_str(); }

int wt_tracks_write_json(const char* path, const wt_detfile* f, int64_t n, const int64_t* frame, const int32_t* category,
                         const double* bbox4, const double* score, const int64_t* object_id) {
    if (!path || !f || n < 0) { wt::set_error("wt_tracks_write_json: bad argument"); return WT_ERR_INVALID; }
    FILE* fp = fopen(path, "wb");
    if (!fp) { wt::set_error("cannot open %s for writing", path); return WT_ERR_INVALID; }
    std::string buf;
    buf.reserve(1 << 20);
    buf += '[';
    char num[64];
    int32_t stream = 0;
    const int32_t ns = (int32_t)f->segment.size();
    for (int64_t i = 0; i < n; ++i) {
        const int64_t fr = frame[i];
        if (fr < 0 || fr >= (int64_t)f->frame_ids.size()) { fclose(fp); wt::set_error("row %lld: frame index out of range", (long long)i); return WT_ERR_INVALID; }
        while (stream + 1 < ns && f->stream_frame_offsets[(size_t)stream + 1] <= fr) ++stream;   // rows are stream-ordered
        while (stream > 0 && f->stream_frame_offsets[(size_t)stream] > fr) --stream;
        if (i) buf += ", ";
        buf += "{\"image_id\": \"";
        buf += f->segment[(size_t)stream];
        buf += '/';
        buf += std::to_string((long long)f->frame_ids[(size_t)fr]);
        buf += '/';
        buf += f->camera[(size_t)stream];
        buf += "\", \"bbox\": [";
        for (int q = 0; q < 4; ++q) {
            py_float_repr(bbox4[4 * i + q], num);
            if (q) buf += ", ";
            buf += num;
        }
        buf += "], \"score\": ";
        py_float_repr(score[i], num);
        buf += num;
        buf += ", \"category_id\": ";
        buf += std::to_string((int)category[i]);
        buf += ", \"object_id\": \"";
        buf += std::to_string((long long)object_id[i]);
        buf += "\"}";
        if (buf.size() > (1 << 20) - 512) { fwrite(buf.data(), 1, buf.size(), fp); buf.clear(); }
    }
    buf += ']';
    fwrite(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    return WT_OK;
}

/* ---- generic detection JSON: reader for the ensemble inputs, writer for detection / ensemble outputs ---- */
int wt_detjson_read(const char* path, wt_detjson** out) {
    if (!path || !out) { wt::set_error("wt_detjson_read: bad argument"); return WT_ERR_INVALID; }
    *out = nullptr;
    FILE* fp = fopen(path, "rb");
    if (!fp) { wt::set_error("cannot open %s", path); return WT_ERR_INVALID; }
    fseek(fp, 0, SEEK_END);
    const long size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::string text((size_t)size, '\0');
    const size_t got = fread(&text[0], 1, (size_t)size, fp);
    fclose(fp);
    if (got != (size_t)size) { wt::set_error("short read on %s", path); return WT_ERR_INVALID; }
    Parser ps{text.data(), text.data() + text.size(), {}};
    if (!ps.lit('[')) { wt::set_error("%s: expected a JSON list", path); return WT_ERR_INVALID; }
    wt_detjson* f = new wt_detjson;
    std::unordered_map<std::string, int> index;
    std::string key, image_id;
    if (!ps.lit(']')) {
        do {
            if (!ps.lit('{')) { ps.fail("expected {"); break; }
            double b[4] = {0, 0, 0, 0}, score = 0, cat = 0;
            bool have_id = false, have_bbox = false, have_score = false, have_cat = false;
            if (!ps.lit('}')) {
                do {
                    if (!ps.string(key) || !ps.lit(':')) break;
                    if (key == "image_id") { if (!ps.string(image_id)) break; have_id = true; }
                    else if (key == "category_id") { if (!ps.number(cat)) break; have_cat = true; }
                    else if (key == "score") { if (!ps.number(score)) break; have_score = true; }
                    else if (key == "bbox") {
                        if (!ps.lit('[')) { ps.fail("expected ["); break; }
                        bool ok = true;
                        for (int i = 0; i < 4 && ok; ++i) ok = ps.number(b[i]) && (i == 3 || ps.lit(','));
                        if (!ok || !ps.lit(']')) { ps.fail("bbox must have 4 numbers"); break; }
                        have_bbox = true;
                    } else if (!ps.skip()) break;
                } while (ps.lit(','));
                if (!ps.err.empty() || !ps.lit('}')) { ps.fail("expected }"); break; }
            }
            if (!have_id || !have_bbox || !have_score || !have_cat) { ps.fail("entry without image_id / category_id / bbox / score"); break; }
            auto it = index.find(image_id);
            int ii;
            if (it == index.end()) { ii = (int)f->image_ids.size(); index.emplace(image_id, ii); f->image_ids.push_back(image_id); }
            else ii = it->second;
            f->image.push_back(ii); f->category.push_back((int32_t)cat);
            f->x.push_back(b[0]); f->y.push_back(b[1]); f->w.push_back(b[2]); f->h.push_back(b[3]); f->score.push_back(score);
        } while (ps.lit(','));
        if (ps.err.empty() && !ps.lit(']')) ps.fail("expected ]");
    }
    if (!ps.err.empty()) { wt::set_error("%s: %s", path, ps.err.c_str()); delete f; return WT_ERR_INVALID; }
    *out = f;
    return WT_OK;
}
void wt_detjson_free(wt_detjson* f) { delete f; }
int64_t wt_detjson_num_rows(const wt_detjson* f) { return f ? (int64_t)f->x.size() : 0; }
int32_t wt_detjson_num_images(const wt_detjson* f) { return f ? (int32_t)f->image_ids.size() : 0; }
const int32_t* wt_detjson_image(const wt_detjson* f) { return f->image.data(); }
const int32_t* wt_detjson_category(const wt_detjson* f) { return f->category.data(); }
const double* wt_detjson_x(const wt_detjson* f) { return f->x.data(); }
const double* wt_detjson_y(const wt_detjson* f) { return f->y.data(); }
const double* wt_detjson_w(const wt_detjson* f) { return f->w.data(); }
const double* wt_detjson_h(const wt_detjson* f) { return f->h.data(); }
const double* wt_detjson_score(const wt_detjson* f) { return f->score.data(); }
const char* wt_detjson_image_id(const wt_detjson* f, int32_t i) { return f->image_ids[(size_t)i].c_str(); }

int wt_detections_write_json(const char* path, int64_t n, const int32_t* image_index, int32_t n_images, const char* image_id_blob,
                             const int64_t* image_id_offsets, const int32_t* category, const int64_t* bbox4, const double* score) {
    if (!path || n < 0 || n_images < 0 || (n && (!image_index || !category || !bbox4 || !score || !image_id_blob || !image_id_offsets))) {
        wt::set_error("wt_detections_write_json: bad argument");
        return WT_ERR_INVALID;
    }
    FILE* fp = fopen(path, "wb");
    if (!fp) { wt::set_error("cannot open %s for writing", path); return WT_ERR_INVALID; }
    std::vector<std::string> ids((size_t)n_images);             // escaped once per image
    for (int32_t i = 0; i < n_images; ++i)
        py_json_string(std::string(image_id_blob + image_id_offsets[i], (size_t)(image_id_offsets[i + 1] - image_id_offsets[i])), ids[(size_t)i]);
    std::string buf;
    buf.reserve(1 << 20);
    buf += '[';
    char num[64];
    for (int64_t i = 0; i < n; ++i) {
        const int32_t im = image_index[i];
        if (im < 0 || im >= n_images) { fclose(fp); wt::set_error("row %lld: image index out of range", (long long)i); return WT_ERR_INVALID; }
        if (i) buf += ", ";
        buf += "{\"image_id\": ";
        buf += ids[(size_t)im];
        buf += ", \"category_id\": ";
        buf += std::to_string((int)category[i]);
        buf += ", \"bbox\": [";
        for (int q = 0; q < 4; ++q) {
            if (q) buf += ", ";
            buf += std::to_string((long long)bbox4[4 * i + q]);
        }
        buf += "], \"score\": ";
        py_float_repr(score[i], num);
        buf += num;
        buf += '}';
        if (buf.size() > (1 << 20) - 512) { fwrite(buf.data(), 1, buf.size(), fp); buf.clear(); }
    }
    buf += ']';
    fwrite(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    return WT_OK;
}

int wt_format_double(double v, char* out, int cap) {
    char tmp[64];
    const int n = py_float_repr(v, tmp);
    if (n + 1 > cap) return -1;
    memcpy(out, tmp, (size_t)n + 1);
    return n;
}

}  // extern "C"
