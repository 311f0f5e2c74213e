// gs_ply.cpp — Inria PLY codec and Gaussian <-> PlyGaussianPod conversions (host side).
// Restates src/source_format/ply.rs (PlyGaussianPod, PLY_PROPERTIES, read_header, read_gaussians,
// write_to) and src/gaussian.rs:70-125 (from_ply / to_ply).  Differences from the reference are
// deliberate and listed in DESIGN.md §7: the fast path requires EXACTLY the 62 Inria properties
// (the reference's zip-based check also accepts a strict prefix or extra trailing properties, which
// then mis-strides the body), and elements that precede `vertex` are skipped instead of being
// read as vertices.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gs_convert.h"
#include "gs_internal.h"

static_assert(sizeof(gs_ply_gaussian_pod) == 248, "PlyGaussianPod is 62 f32");

static const char *k_props[62] = {
    "x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2",
    "f_rest_0", "f_rest_1", "f_rest_2", "f_rest_3", "f_rest_4", "f_rest_5", "f_rest_6", "f_rest_7",
    "f_rest_8", "f_rest_9", "f_rest_10", "f_rest_11", "f_rest_12", "f_rest_13", "f_rest_14",
    "f_rest_15", "f_rest_16", "f_rest_17", "f_rest_18", "f_rest_19", "f_rest_20", "f_rest_21",
    "f_rest_22", "f_rest_23", "f_rest_24", "f_rest_25", "f_rest_26", "f_rest_27", "f_rest_28",
    "f_rest_29", "f_rest_30", "f_rest_31", "f_rest_32", "f_rest_33", "f_rest_34", "f_rest_35",
    "f_rest_36", "f_rest_37", "f_rest_38", "f_rest_39", "f_rest_40", "f_rest_41", "f_rest_42",
    "f_rest_43", "f_rest_44", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2",
    "rot_3"};

extern "C" const char *gs_ply_property_name(uint32_t index) { return index < 62 ? k_props[index] : nullptr; }

// ---- Gaussian::from_ply / to_ply (src/gaussian.rs:70-125) ------------------------------------

// threaded over the records (50 M vertices took 80 s on one thread); the per-record arithmetic is
// gs_convert.h's, shared with the device kernel
template <class F>
static void ply_parallel_for(size_t n, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t threads = n < 32768 ? 1 : (hw ? (hw > 32 ? 32 : hw) : 4);
    if (const char *e = std::getenv("GS3D_HOST_THREADS")) threads = std::atoi(e) > 0 ? (size_t)std::atoi(e) : threads;
    if (threads <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (n + threads - 1) / threads;
    for (size_t t = 0; t < threads; t++) {
        const size_t a = t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        pool.emplace_back([=] { fn(a, b); });
    }
    for (auto &th : pool) th.join();
}

extern "C" void gs_gaussian_from_ply(const gs_ply_gaussian_pod *in, size_t n, gs_gaussian *out) {
    static_assert(sizeof(gs_gaussian) == gs::CV_GAUSSIAN_WORDS * 4, "gs_gaussian layout");
    ply_parallel_for(n, [=](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            uint32_t pw[gs::PLY_WORDS], gw[gs::CV_GAUSSIAN_WORDS];
            std::memcpy(pw, &in[i], sizeof(pw));
            gs::ply_to_gaussian_words(pw, gw, [](float v) { return std::sqrt(v); });
            std::memcpy(&out[i], gw, sizeof(gw));
        }
    });
}

extern "C" float gs_expf(float x) { return gs::gs_expf(x); }

extern "C" void gs_gaussian_to_ply(const gs_gaussian *in, size_t n, gs_ply_gaussian_pod *out) {
    for (size_t i = 0; i < n; i++) {
        const gs_gaussian &g = in[i];
        gs_ply_gaussian_pod &p = out[i];
        std::memcpy(p.pos, g.pos, 12);
        p.rot[0] = g.rot[3];
        p.rot[1] = g.rot[0];
        p.rot[2] = g.rot[1];
        p.rot[3] = g.rot[2];
        for (int k = 0; k < 3; k++) p.scale[k] = std::log(g.scale[k]);
        float rgba[4];
        for (int k = 0; k < 4; k++) rgba[k] = (float)g.color[k] / 255.0f;
        for (int k = 0; k < 3; k++) p.color[k] = (rgba[k] - 0.5f) / 0.2820948f;
        p.alpha = -std::log(1.0f / rgba[3] - 1.0f);
        for (int k = 0; k < 15; k++) {
            p.sh[k] = g.sh[3 * k + 0];
            p.sh[k + 15] = g.sh[3 * k + 1];
            p.sh[k + 30] = g.sh[3 * k + 2];
        }
        p.normal[0] = 0.0f;
        p.normal[1] = 0.0f;
        p.normal[2] = 1.0f;
    }
}

// ---- header ------------------------------------------------------------------------------------

enum Enc { ASCII, LE, BE };
enum Ty { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };

static Ty parse_type(const std::string &t) {
    if (t == "char" || t == "int8") return T_I8;
    if (t == "uchar" || t == "uint8") return T_U8;
    if (t == "short" || t == "int16") return T_I16;
    if (t == "ushort" || t == "uint16") return T_U16;
    if (t == "int" || t == "int32") return T_I32;
    if (t == "uint" || t == "uint32") return T_U32;
    if (t == "float" || t == "float32") return T_F32;
    if (t == "double" || t == "float64") return T_F64;
    return T_BAD;
}
static size_t type_size(Ty t) {
    switch (t) {
    case T_I8: case T_U8: return 1;
    case T_I16: case T_U16: return 2;
    case T_I32: case T_U32: case T_F32: return 4;
    case T_F64: return 8;
    default: return 0;
    }
}

struct Prop {
    std::string name;
    Ty type;       // scalar type, or element type of a list
    bool is_list;
    Ty count_type;
    int field;     // index into the 62 floats, -1 = unknown property (ignored with a warning)
};
struct Element {
    std::string name;
    size_t count;
    std::vector<Prop> props;
};
struct Header {
    Enc enc;
    std::vector<Element> elements;
    size_t body;   // offset of the first body byte
};

static std::vector<std::string> split_ws(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\r')) i++;
        size_t j = i;
        while (j < s.size() && s[j] != ' ' && s[j] != '\t' && s[j] != '\r') j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

static int field_of(const std::string &name) {
    for (int k = 0; k < 62; k++)
        if (name == k_props[k]) return k;
    return -1;
}

static gs_status parse_header(const uint8_t *b, size_t len, Header &h) {
    size_t pos = 0;
    int line_no = 0;
    bool have_format = false, ended = false;
    while (pos < len) {
        size_t eol = pos;
        while (eol < len && b[eol] != '\n') eol++;
        if (eol >= len) break;
        std::string line((const char *)b + pos, eol - pos);
        pos = eol + 1;
        std::vector<std::string> tok = split_ws(line);
        if (line_no++ == 0) {
            if (tok.size() != 1 || tok[0] != "ply")
                return gs_fail(GS_ERR_PLY, 0, 0, 0, "PLY magic number not found");
            continue;
        }
        if (tok.empty() || tok[0] == "comment" || tok[0] == "obj_info") continue;
        if (tok[0] == "format") {
            if (tok.size() < 3) return gs_fail(GS_ERR_PLY, line_no, 0, 0, "malformed PLY format line");
            if (tok[1] == "ascii") h.enc = ASCII;
            else if (tok[1] == "binary_little_endian") h.enc = LE;
            else if (tok[1] == "binary_big_endian") h.enc = BE;
            else return gs_fail(GS_ERR_PLY, line_no, 0, 0, "unknown PLY format '%s'", tok[1].c_str());
            have_format = true;
        } else if (tok[0] == "element") {
            if (tok.size() != 3) return gs_fail(GS_ERR_PLY, line_no, 0, 0, "malformed PLY element line");
            Element e;
            e.name = tok[1];
            e.count = (size_t)std::strtoull(tok[2].c_str(), nullptr, 10);
            h.elements.push_back(e);
        } else if (tok[0] == "property") {
            if (h.elements.empty()) return gs_fail(GS_ERR_PLY, line_no, 0, 0, "PLY property before any element");
            Prop p;
            p.is_list = tok.size() == 5 && tok[1] == "list";
            if (p.is_list) {
                p.count_type = parse_type(tok[2]);
                p.type = parse_type(tok[3]);
                p.name = tok[4];
                if (p.count_type == T_BAD || p.count_type == T_F32 || p.count_type == T_F64)
                    return gs_fail(GS_ERR_PLY, line_no, 0, 0, "bad PLY list count type");
            } else {
                if (tok.size() != 3) return gs_fail(GS_ERR_PLY, line_no, 0, 0, "malformed PLY property line");
                p.count_type = T_BAD;
                p.type = parse_type(tok[1]);
                p.name = tok[2];
            }
            if (p.type == T_BAD) return gs_fail(GS_ERR_PLY, line_no, 0, 0, "unknown PLY property type");
            p.field = field_of(p.name);
            h.elements.back().props.push_back(p);
        } else if (tok[0] == "end_header") {
            ended = true;
            break;
        } else {
            return gs_fail(GS_ERR_PLY, line_no, 0, 0, "unexpected PLY header line '%s'", tok[0].c_str());
        }
    }
    if (!ended || !have_format) return gs_fail(GS_ERR_PLY, 0, 0, 0, "incomplete PLY header");
    h.body = pos;
    return GS_OK;
}

// ---- body --------------------------------------------------------------------------------------

static bool read_scalar(const uint8_t *b, size_t len, size_t &pos, Ty t, bool big, double &out) {
    size_t sz = type_size(t);
    if (len - pos < sz || pos > len) return false;
    uint8_t tmp[8];
    for (size_t k = 0; k < sz; k++) tmp[k] = big ? b[pos + sz - 1 - k] : b[pos + k];
    pos += sz;
    switch (t) {
    case T_I8: out = (int8_t)tmp[0]; break;
    case T_U8: out = tmp[0]; break;
    case T_I16: { int16_t v; std::memcpy(&v, tmp, 2); out = v; } break;
    case T_U16: { uint16_t v; std::memcpy(&v, tmp, 2); out = v; } break;
    case T_I32: { int32_t v; std::memcpy(&v, tmp, 4); out = v; } break;
    case T_U32: { uint32_t v; std::memcpy(&v, tmp, 4); out = v; } break;
    case T_F32: { float v; std::memcpy(&v, tmp, 4); out = v; } break;
    case T_F64: { double v; std::memcpy(&v, tmp, 8); out = v; } break;
    default: return false;
    }
    return true;
}

static const char *k_eof = "failed to fill whole buffer";   // std::io::ErrorKind::UnexpectedEof

extern "C" gs_status gs_ply_read(const void *bytes, size_t len, gs_ply_gaussian_pod *out,
                                 size_t capacity, size_t *count_out, int32_t *is_inria_out) {
    if (!bytes || !count_out) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    const uint8_t *b = (const uint8_t *)bytes;
    Header h;
    h.enc = ASCII;
    h.body = 0;
    gs_status rc = parse_header(b, len, h);
    if (rc != GS_OK) return rc;
    const Element *vertex = nullptr;
    size_t vi = 0;
    for (size_t i = 0; i < h.elements.size(); i++)
        if (h.elements[i].name == "vertex") {
            vertex = &h.elements[i];
            vi = i;
            break;
        }
    if (!vertex) return gs_fail(GS_ERR_PLY, 0, 0, 0, "Gaussian vertex element not found in PLY header");
    bool inria = h.enc == LE && vertex->props.size() == 62 && vi == 0;
    for (size_t k = 0; inria && k < 62; k++)
        inria = !vertex->props[k].is_list && vertex->props[k].type == T_F32 && vertex->props[k].field == (int)k;
    {   // The header's vertex count is untrusted and callers size their allocation from it: a count the
        // remaining bytes cannot possibly hold is the reference's UnexpectedEof, reported up front.
        size_t min_bytes = 0;
        for (const Prop &p : vertex->props)
            min_bytes += h.enc == ASCII ? 2 : (size_t)type_size(p.is_list ? p.count_type : p.type);
        if (min_bytes == 0) min_bytes = h.enc == ASCII ? 1 : 0;
        const size_t avail = len > h.body ? len - h.body : 0;
        if (min_bytes && vertex->count > avail / min_bytes + (h.enc == ASCII ? 1 : 0))
            return gs_fail(GS_ERR_PLY, vertex->count, avail, min_bytes, "%s", k_eof);
    }
    *count_out = vertex->count;
    if (is_inria_out) *is_inria_out = inria ? 1 : 0;
    if (!out) return GS_OK;
    size_t n = vertex->count < capacity ? vertex->count : capacity;
    size_t pos = h.body;
    if (inria) {   // ply.rs:333-338: read_exact of 248 bytes per vertex
        if (len - pos < n * sizeof(gs_ply_gaussian_pod)) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
        std::memcpy(out, b + pos, n * sizeof(gs_ply_gaussian_pod));
        return GS_OK;
    }
    const bool big = h.enc == BE;
    // skip the elements in front of `vertex`
    for (size_t ei = 0; ei < vi; ei++) {
        const Element &e = h.elements[ei];
        for (size_t r = 0; r < e.count; r++) {
            if (h.enc == ASCII) {
                while (pos < len && b[pos] != '\n') pos++;
                if (pos >= len) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                pos++;
            } else {
                for (const Prop &p : e.props) {
                    double v;
                    if (p.is_list) {
                        if (!read_scalar(b, len, pos, p.count_type, big, v)) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                        size_t bytes_ = (size_t)v * type_size(p.type);
                        if (len - pos < bytes_) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                        pos += bytes_;
                    } else if (!read_scalar(b, len, pos, p.type, big, v)) {
                        return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                    }
                }
            }
        }
    }
    for (size_t r = 0; r < n; r++) {
        float f[62];
        std::memset(f, 0, sizeof(f));   // PlyGaussianPod::zeroed()
        if (h.enc == ASCII) {           // ply.rs:347-372
            size_t eol = pos;
            while (eol < len && b[eol] != '\n') eol++;
            std::string line((const char *)b + pos, eol - pos);
            pos = eol < len ? eol + 1 : eol;
            std::vector<std::string> tok;
            {   // the reference splits on single spaces and trims each piece
                size_t i = 0;
                while (i <= line.size()) {
                    size_t j = line.find(' ', i);
                    if (j == std::string::npos) j = line.size();
                    tok.push_back(line.substr(i, j - i));
                    i = j + 1;
                }
            }
            for (size_t k = 0; k < vertex->props.size(); k++) {
                const char *msg = "Gaussian element property invalid or missing in PLY";
                if (k >= tok.size()) return gs_fail(GS_ERR_PLY, r, k, 0, "%s", msg);
                std::string t = tok[k];
                while (!t.empty() && (t.back() == '\r' || t.back() == ' ' || t.back() == '\t')) t.pop_back();
                if (t.empty()) return gs_fail(GS_ERR_PLY, r, k, 0, "%s", msg);
                char *end = nullptr;
                float v = std::strtof(t.c_str(), &end);
                if (end == t.c_str() || *end != 0) return gs_fail(GS_ERR_PLY, r, k, 0, "%s", msg);
                int fld = vertex->props[k].field;
                if (fld >= 0) f[fld] = v;
            }
        } else {                        // ply.rs:373-378 via ply-rs: typed binary properties
            for (const Prop &p : vertex->props) {
                double v;
                if (p.is_list) {
                    if (!read_scalar(b, len, pos, p.count_type, big, v)) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                    size_t bytes_ = (size_t)v * type_size(p.type);
                    if (len - pos < bytes_) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                    pos += bytes_;
                    continue;
                }
                if (!read_scalar(b, len, pos, p.type, big, v)) return gs_fail(GS_ERR_PLY, 0, 0, 0, "%s", k_eof);
                // set_property: only Property::Float is accepted (ply.rs:108-115)
                if (p.type == T_F32 && p.field >= 0) f[p.field] = (float)v;
            }
        }
        std::memcpy(&out[r], f, sizeof(f));
    }
    return GS_OK;
}

extern "C" gs_status gs_ply_write(const gs_ply_gaussian_pod *pods, size_t n, void *out,
                                  size_t capacity, size_t *bytes_out) {
    if (!bytes_out || (n && !pods)) return gs_fail(GS_ERR_INVALID_ARGUMENT, 0, 0, 0, "null argument");
    std::string hdr = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(n) + "\n";
    for (int k = 0; k < 62; k++) hdr += std::string("property float ") + k_props[k] + "\n";
    hdr += "end_header\n";
    size_t total = hdr.size() + n * sizeof(gs_ply_gaussian_pod);
    *bytes_out = total;
    if (!out) return GS_OK;
    if (capacity < total)
        return gs_fail(GS_ERR_INVALID_ARGUMENT, capacity, total, 0, "output buffer too small: %zu < %zu", capacity, total);
    std::memcpy(out, hdr.data(), hdr.size());
    if (n) std::memcpy((uint8_t *)out + hdr.size(), pods, n * sizeof(gs_ply_gaussian_pod));
    return GS_OK;
}
