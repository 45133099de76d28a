// scene_io.cpp — the on-disk formats either side of the hot path, host only.
//
//   pt_scene_load   SceneDescriptor::load + to_data      src/render/mod.rs:92-110, 304-318
//                   (serde_json, externally tagged enums; JSON number -> f64 -> `as f32`)
//   pt_load_off     load_off                              src/render/load_off.rs:8-85
//   pt_write_ppm    the P3 writer of render()             src/render/mod.rs:1043-1076
//   pt_gamma_*      gamma_correction / to_int_with_...    src/render/mod.rs:57-63
//
// Where the reference panics (unwrap) this code returns PT_ERR_PARSE / PT_ERR_IO with a message.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/ptrace.h"
#include "../csrc/pt_host.h"

namespace {

// ------------------------------------------------------------------ minimal JSON value + parser
struct JValue;
using JPtr = std::unique_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;  // insertion order; serde takes the last duplicate? (first here)
    const JValue *get(const char *key) const {
        if (kind != Obj) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    explicit JParser(const std::string &s) : p(s.data()), end(s.data() + s.size()) {}
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }
    bool lit(const char *s) {
        size_t n = strlen(s);
        if ((size_t)(end - p) >= n && memcmp(p, s, n) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    bool parse_string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                ++p;
                if (p >= end) return fail("bad escape");
                switch (*p) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) {
                            char c = p[i];
                            cp <<= 4;
                            if (c >= '0' && c <= '9') cp |= (unsigned)(c - '0');
                            else if (c >= 'a' && c <= 'f') cp |= (unsigned)(c - 'a' + 10);
                            else if (c >= 'A' && c <= 'F') cp |= (unsigned)(c - 'A' + 10);
                            else return fail("bad \\u escape");
                        }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) {
                            out += (char)(0xC0 | (cp >> 6));
                            out += (char)(0x80 | (cp & 0x3F));
                        } else {
                            out += (char)(0xE0 | (cp >> 12));
                            out += (char)(0x80 | ((cp >> 6) & 0x3F));
                            out += (char)(0x80 | (cp & 0x3F));
                        }
                        break;
                    }
                    default: return fail("bad escape");
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse_value(JValue &v, int depth) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of input");
        if (*p == '{') {
            ++p;
            v.kind = JValue::Obj;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return true;
            }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                JPtr child(new JValue());
                if (!parse_value(*child, depth + 1)) return false;
                v.obj.emplace_back(std::move(key), std::move(child));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            ++p;
            v.kind = JValue::Arr;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return true;
            }
            for (;;) {
                JPtr child(new JValue());
                if (!parse_value(*child, depth + 1)) return false;
                v.arr.push_back(std::move(child));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') {
            v.kind = JValue::Str;
            return parse_string(v.str);
        }
        if (lit("true")) {
            v.kind = JValue::Bool;
            v.b = true;
            return true;
        }
        if (lit("false")) {
            v.kind = JValue::Bool;
            v.b = false;
            return true;
        }
        if (lit("null")) {
            v.kind = JValue::Null;
            return true;
        }
        // number: JSON grammar, value through strtod (serde_json: decimal -> f64)
        const char *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || !(*p >= '0' && *p <= '9')) return fail("unexpected character");
        while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && *p == '.') {
            ++p;
            if (p >= end || !(*p >= '0' && *p <= '9')) return fail("bad number");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || !(*p >= '0' && *p <= '9')) return fail("bad exponent");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        std::string tok(s, p);
        v.kind = JValue::Num;
        v.num = strtod(tok.c_str(), nullptr);
        return true;
    }
};

struct Ctx {
    std::string err;
    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }
};

bool get_f32(Ctx &c, const JValue *v, const char *what, float *out) {
    if (!v || v->kind != JValue::Num) return c.fail(std::string("missing or non-numeric field `") + what + "`");
    *out = (float)v->num;  // f64 -> f32, as serde's f32 visitor does
    return true;
}

bool get_vec3(Ctx &c, const JValue *v, const char *what, float out[3]) {
    if (!v || v->kind != JValue::Arr || v->arr.size() != 3)
        return c.fail(std::string("field `") + what + "` must be an array of 3 numbers");
    for (int i = 0; i < 3; ++i)
        if (!get_f32(c, v->arr[i].get(), what, &out[i])) return false;
    return true;
}

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

std::vector<std::string> split_ws(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) ++j;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

// Rust str::parse::<usize>(): optional '+', decimal digits only
bool parse_usize(const std::string &t, uint64_t *out) {
    size_t i = 0;
    if (i < t.size() && t[i] == '+') ++i;
    if (i >= t.size()) return false;
    uint64_t v = 0;
    for (; i < t.size(); ++i) {
        if (t[i] < '0' || t[i] > '9') return false;
        if (v > (UINT64_MAX - 9) / 10) return false;
        v = v * 10 + (uint64_t)(t[i] - '0');
    }
    *out = v;
    return true;
}

// Rust str::parse::<f32>(): correctly rounded decimal -> f32 (strtof), no hex, whole token
bool parse_f32(const std::string &t, float *out) {
    if (t.empty()) return false;
    for (char ch : t)
        if (ch == 'x' || ch == 'X' || ch == 'p' || ch == 'P') return false;
    char *e = nullptr;
    errno = 0;
    float v = strtof(t.c_str(), &e);
    if (e != t.c_str() + t.size()) return false;
    *out = v;
    return true;
}

int load_off_impl(const std::string &path, float scale, uint32_t flags, std::vector<pt_triangle> &out,
                  std::string &err) {
    std::ifstream f(path);
    if (!f) {
        err = "cannot open " + path;
        return PT_ERR_IO;
    }
    bool eof = false;
    auto get_line = [&](std::string &line) -> bool {  // skips blank and '#' lines (load_off.rs:12-20)
        for (;;) {
            std::string raw;
            if (!std::getline(f, raw)) {
                eof = true;
                return false;
            }
            line = trim(raw);
            if (!line.empty() && line[0] != '#') return true;
        }
    };
    std::string line;
    if (!get_line(line) || line != "OFF") {
        err = eof ? "unexpected end of file" : "Invalid header";
        return PT_ERR_PARSE;
    }
    if (!get_line(line)) {
        err = "unexpected end of file";
        return PT_ERR_PARSE;
    }
    std::vector<std::string> tk = split_ws(line);
    uint64_t nv = 0, nf = 0, ne = 0;
    if (tk.size() != 3 || !parse_usize(tk[0], &nv) || !parse_usize(tk[1], &nf) || !parse_usize(tk[2], &ne)) {
        err = "Invalid element counts";
        return PT_ERR_PARSE;
    }
    if (nv > (1u << 28) || nf > (1u << 28)) {
        err = "mesh too large";
        return PT_ERR_PARSE;
    }
    std::vector<pt::vec3> verts;
    verts.reserve((size_t)nv);
    for (uint64_t i = 0; i < nv; ++i) {
        if (!get_line(line)) {
            err = "unexpected end of file";
            return PT_ERR_PARSE;
        }
        tk = split_ws(line);
        float c[3];
        if (tk.size() != 3 || !parse_f32(tk[0], &c[0]) || !parse_f32(tk[1], &c[1]) || !parse_f32(tk[2], &c[2])) {
            err = "Invalid vertex coordinates";
            return PT_ERR_PARSE;
        }
        verts.push_back(pt::mk(c[0], c[1], c[2]) * scale);  // Vec3::new(..) * scale, load_off.rs:52
    }
    out.clear();
    out.reserve((size_t)nf);
    for (uint64_t i = 0; i < nf; ++i) {
        if (!get_line(line)) {
            err = "unexpected end of file";
            return PT_ERR_PARSE;
        }
        tk = split_ws(line);
        uint64_t idx[4];
        if (tk.size() < 4 || !parse_usize(tk[0], &idx[0]) || !parse_usize(tk[1], &idx[1]) ||
            !parse_usize(tk[2], &idx[2]) || !parse_usize(tk[3], &idx[3]) ||
            (idx[0] != 3 && !(flags & PT_LOAD_TRIANGULATE)) || idx[0] < 3) {
            err = "Invalid face: " + line;  // only triangles are supported (load_off.rs:73-76)
            return PT_ERR_PARSE;
        }
        // extension beyond the reference (no oracle there): a convex polygon v0..vk-1 becomes the fan
        // (v0,v1,v2), (v0,v2,v3), ...; a triangle is the one-element fan, i.e. exactly the reference's case
        std::vector<uint64_t> poly;
        if (tk.size() < 1 + idx[0]) {
            err = "Invalid face: " + line;
            return PT_ERR_PARSE;
        }
        for (uint64_t k = 0; k < idx[0]; ++k) {
            uint64_t vi = 0;
            if (!parse_usize(tk[1 + (size_t)k], &vi) || vi >= nv) {
                err = "face index out of range: " + line;
                return PT_ERR_PARSE;
            }
            poly.push_back(vi);
        }
        for (size_t k = 1; k + 1 < poly.size(); ++k) {
            pt_triangle t;
            const pt::vec3 a = verts[poly[0]], b = verts[poly[k]], c = verts[poly[k + 1]];
            t.a[0] = a.x, t.a[1] = a.y, t.a[2] = a.z;
            t.b[0] = b.x, t.b[1] = b.y, t.b[2] = b.z;
            t.c[0] = c.x, t.c[1] = c.y, t.c[2] = c.z;
            out.push_back(t);
        }
    }
    return PT_OK;
}

}  // namespace

// what SceneObjectDescriptorType keeps beyond the flattened object (needed to save the scene again)
struct ObjDesc {
    int variant = 0;  // 0 Sphere, 1 MeshFile, 2 Mesh
    std::string path;
    float scale = 0.0f;
    std::vector<pt_triangle> bounding_box;  // Mesh.bounding_box as loaded (12 triangles; unused by the tracer)
};

struct pt_scene {
    std::string id;
    pt_camera camera{};
    std::vector<pt_object> objects;
    std::vector<pt_triangle> triangles;
    std::vector<ObjDesc> desc;
};

namespace {

// f32 -> text exactly as serde_json does (ryu's shortest round-trip digits, `format32` layout)
std::string fmt_f32(float v) {
    if (!std::isfinite(v)) return "null";
    char buf[64];
    int prec = 0;
    for (; prec < 9; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec, (double)v);
        if (strtof(buf, nullptr) == v) break;
    }
    snprintf(buf, sizeof buf, "%.*e", prec, (double)v);
    // buf = [-]d[.ddd]e[+-]xx
    std::string t = buf;
    const bool neg = t[0] == '-';
    if (neg) t.erase(0, 1);
    const size_t epos = t.find('e');
    std::string digits = t.substr(0, epos);
    const int e10 = atoi(t.c_str() + epos + 1);
    digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    if (v == 0.0f) digits = "0";
    const int len = (int)digits.size();
    const int kk = e10 + 1;       // position of the decimal point relative to the first digit
    const int k = kk - len;       // exponent of the last digit
    std::string out;
    if (0 <= k && kk <= 13) {
        out = digits + std::string((size_t)k, '0') + ".0";
    } else if (0 < kk && kk <= 13) {
        out = digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    } else if (-6 < kk && kk <= 0) {
        out = "0." + std::string((size_t)(-kk), '0') + digits;
    } else if (len == 1) {
        out = digits + "e" + std::to_string(kk - 1);
    } else {
        out = digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    }
    return neg ? "-" + out : out;
}

std::string json_escape(const std::string &in) {
    std::string o = "\"";
    for (unsigned char ch : in) {
        switch (ch) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (ch < 0x20) {
                    char b[8];
                    snprintf(b, sizeof b, "\\u%04x", ch);
                    o += b;
                } else {
                    o += (char)ch;
                }
        }
    }
    return o + "\"";
}

// serde_json::to_string_pretty layout: 2-space indent, one array element per line
struct Pretty {
    std::string out;
    int depth = 0;
    void nl() {
        out += '\n';
        out.append((size_t)depth * 2, ' ');
    }
    void vec3(const float v[3]) {
        out += '[';
        ++depth;
        for (int i = 0; i < 3; ++i) {
            nl();
            out += fmt_f32(v[i]);
            if (i < 2) out += ',';
        }
        --depth;
        nl();
        out += ']';
    }
    void key(const char *k) {
        nl();
        out += '"';
        out += k;
        out += "\": ";
    }
    void triangles(const pt_triangle *t, size_t n) {
        if (n == 0) {
            out += "[]";
            return;
        }
        out += '[';
        ++depth;
        for (size_t i = 0; i < n; ++i) {
            nl();
            out += '{';
            ++depth;
            key("a");
            vec3(t[i].a);
            out += ',';
            key("b");
            vec3(t[i].b);
            out += ',';
            key("c");
            vec3(t[i].c);
            --depth;
            nl();
            out += '}';
            if (i + 1 < n) out += ',';
        }
        --depth;
        nl();
        out += ']';
    }
};

// bounding_box_to_triangles (mod.rs:501-536) over the AABB of a triangle list, for meshes built through the API
std::vector<pt_triangle> box_triangles(const pt_triangle *t, size_t n) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i)
        for (const float *v : {t[i].a, t[i].b, t[i].c})
            for (int c = 0; c < 3; ++c) {
                if (v[c] < lo[c]) lo[c] = v[c];
                if (v[c] > hi[c]) hi[c] = v[c];
            }
    const float vx[8][3] = {{lo[0], lo[1], lo[2]}, {hi[0], lo[1], lo[2]}, {hi[0], hi[1], lo[2]}, {lo[0], hi[1], lo[2]},
                            {lo[0], lo[1], hi[2]}, {hi[0], lo[1], hi[2]}, {hi[0], hi[1], hi[2]}, {lo[0], hi[1], hi[2]}};
    const int idx[12][3] = {{0, 1, 2}, {0, 2, 3}, {4, 6, 5}, {4, 7, 6}, {0, 4, 5}, {0, 5, 1},
                            {3, 2, 6}, {3, 6, 7}, {1, 5, 6}, {1, 6, 2}, {0, 3, 7}, {0, 7, 4}};
    std::vector<pt_triangle> out(12);
    for (int k = 0; k < 12; ++k) {
        memcpy(out[(size_t)k].a, vx[idx[k][0]], 12);
        memcpy(out[(size_t)k].b, vx[idx[k][1]], 12);
        memcpy(out[(size_t)k].c, vx[idx[k][2]], 12);
    }
    return out;
}

}  // namespace

extern "C" {

int pt_load_off(const char *path, float scale, pt_triangle **tris, uint32_t *n_tris) {
    return pt_load_off_ex(path, scale, 0u, tris, n_tris);
}

int pt_load_off_ex(const char *path, float scale, uint32_t flags, pt_triangle **tris, uint32_t *n_tris) {
    if (!path || !tris || !n_tris) {
        pt::set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    std::vector<pt_triangle> v;
    std::string err;
    int rc = load_off_impl(path, scale, flags, v, err);
    if (rc) {
        pt::set_error(err);
        return rc;
    }
    *n_tris = (uint32_t)v.size();
    *tris = (pt_triangle *)malloc(sizeof(pt_triangle) * (v.empty() ? 1 : v.size()));
    if (!*tris) {
        pt::set_error("out of memory");
        return PT_ERR_INVALID;
    }
    if (!v.empty()) memcpy(*tris, v.data(), sizeof(pt_triangle) * v.size());
    return PT_OK;
}

void pt_free(void *p) { free(p); }

int pt_scene_load(const char *path, const char *base_dir, pt_scene **out) {
    return pt_scene_load_ex(path, base_dir, 0u, out);
}

int pt_scene_load_ex(const char *path, const char *base_dir, uint32_t flags, pt_scene **out) {
    if (!path || !out) {
        pt::set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    *out = nullptr;
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        pt::set_error(std::string("cannot open ") + path);
        return PT_ERR_IO;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    JParser jp(text);
    JValue root;
    if (!jp.parse_value(root, 0)) {
        pt::set_error("JSON: " + jp.err);
        return PT_ERR_PARSE;
    }
    jp.ws();
    if (jp.p != jp.end) {
        pt::set_error("JSON: trailing characters");
        return PT_ERR_PARSE;
    }
    Ctx c;
    std::unique_ptr<pt_scene> sc(new pt_scene());
    const JValue *jid = root.get("id");
    const JValue *jobjs = root.get("objects");
    const JValue *jcam = root.get("camera");
    if (!jid || jid->kind != JValue::Str || !jobjs || jobjs->kind != JValue::Arr || !jcam || jcam->kind != JValue::Obj) {
        pt::set_error("scene: missing `id`, `objects` or `camera`");
        return PT_ERR_PARSE;
    }
    sc->id = jid->str;
    if (!get_vec3(c, jcam->get("position"), "camera.position", sc->camera.position) ||
        !get_vec3(c, jcam->get("direction"), "camera.direction", sc->camera.direction) ||
        !get_f32(c, jcam->get("focal_length"), "camera.focal_length", &sc->camera.focal_length) ||
        !get_f32(c, jcam->get("sensor_width"), "camera.sensor_width", &sc->camera.sensor_width) ||
        !get_f32(c, jcam->get("aspect_ratio"), "camera.aspect_ratio", &sc->camera.aspect_ratio)) {
        pt::set_error("scene: " + c.err);
        return PT_ERR_PARSE;
    }
    const std::string base = base_dir ? base_dir : ".";
    for (size_t i = 0; i < jobjs->arr.size(); ++i) {
        const JValue &jo = *jobjs->arr[i];
        const std::string where = "objects[" + std::to_string(i) + "]";
        pt_object o;
        memset(&o, 0, sizeof o);
        ObjDesc od;
        const JValue *jt = jo.get("type_"), *jm = jo.get("material");
        if (jo.kind != JValue::Obj || !jt || !jm || jm->kind != JValue::Obj ||
            !get_vec3(c, jo.get("position"), "position", o.position) ||
            !get_vec3(c, jm->get("color"), "material.color", o.color) ||
            !get_vec3(c, jm->get("emmission"), "material.emmission", o.emission)) {
            pt::set_error("scene: " + where + ": " + (c.err.empty() ? "missing `type_` or `material`" : c.err));
            return PT_ERR_PARSE;
        }
        const JValue *jr = jm->get("reflect_type");
        if (!jr || jr->kind != JValue::Str) {
            pt::set_error("scene: " + where + ": missing material.reflect_type");
            return PT_ERR_PARSE;
        }
        if (jr->str == "Diffuse") o.reflect_type = PT_DIFFUSE;
        else if (jr->str == "Specular") o.reflect_type = PT_SPECULAR;
        else if (jr->str == "Refract") o.reflect_type = PT_REFRACT;
        else {
            pt::set_error("scene: " + where + ": unknown reflect_type `" + jr->str + "`");
            return PT_ERR_PARSE;
        }
        // externally tagged enum: {"Sphere":{..}} | {"MeshFile":{..}} | {"Mesh":{..}}
        if (jt->kind != JValue::Obj || jt->obj.size() != 1 || jt->obj[0].second->kind != JValue::Obj) {
            pt::set_error("scene: " + where + ": `type_` must be a single-key object");
            return PT_ERR_PARSE;
        }
        const std::string &tag = jt->obj[0].first;
        const JValue &body = *jt->obj[0].second;
        if (tag == "Sphere") {
            o.kind = PT_SPHERE;
            if (!get_f32(c, body.get("radius"), "Sphere.radius", &o.radius)) {
                pt::set_error("scene: " + where + ": " + c.err);
                return PT_ERR_PARSE;
            }
        } else if (tag == "MeshFile") {
            o.kind = PT_MESH;
            od.variant = 1;
            const JValue *jp_ = body.get("path");
            float scale = 0.0f;
            if (!jp_ || jp_->kind != JValue::Str || !get_f32(c, body.get("scale"), "MeshFile.scale", &scale)) {
                pt::set_error("scene: " + where + ": MeshFile needs `path` and `scale`");
                return PT_ERR_PARSE;
            }
            od.path = jp_->str;
            od.scale = scale;
            std::vector<pt_triangle> tl;
            std::string err;
            const std::string full = (jp_->str.size() && jp_->str[0] == '/') ? jp_->str : base + "/" + jp_->str;
            int rc = load_off_impl(full, scale, flags, tl, err);
            if (rc) {
                pt::set_error("scene: " + where + ": " + err);
                return rc;
            }
            if (tl.empty()) {
                pt::set_error("scene: " + where + ": mesh file has no triangles");
                return PT_ERR_PARSE;
            }
            o.tri_offset = (uint32_t)sc->triangles.size();
            o.tri_count = (uint32_t)tl.size();
            pt::host::mesh_bounding_sphere(tl.data(), (uint32_t)tl.size(), o.bs_center, &o.bs_radius);  // Mesh::new
            sc->triangles.insert(sc->triangles.end(), tl.begin(), tl.end());
        } else if (tag == "Mesh") {
            o.kind = PT_MESH;
            const JValue *jtri = body.get("triangles"), *jbs = body.get("bounding_sphere");
            // `bounding_box` must be present for serde to accept the Mesh, but the tracer never reads it
            const JValue *jbb = body.get("bounding_box");
            if (!jtri || jtri->kind != JValue::Arr || !jbs || jbs->kind != JValue::Obj || !jbb || jbb->kind != JValue::Arr ||
                !get_vec3(c, jbs->get("position"), "bounding_sphere.position", o.bs_center) ||
                !get_f32(c, jbs->get("radius"), "bounding_sphere.radius", &o.bs_radius)) {
                pt::set_error("scene: " + where + ": Mesh needs `triangles`, `bounding_sphere`, `bounding_box`" +
                              (c.err.empty() ? "" : " (" + c.err + ")"));
                return PT_ERR_PARSE;
            }
            od.variant = 2;
            for (const auto &jb : jbb->arr) {
                pt_triangle t;
                if (jb->kind != JValue::Obj || !get_vec3(c, jb->get("a"), "bounding_box.a", t.a) ||
                    !get_vec3(c, jb->get("b"), "bounding_box.b", t.b) || !get_vec3(c, jb->get("c"), "bounding_box.c", t.c)) {
                    pt::set_error("scene: " + where + ": " + (c.err.empty() ? "bad bounding_box triangle" : c.err));
                    return PT_ERR_PARSE;
                }
                od.bounding_box.push_back(t);
            }
            o.tri_offset = (uint32_t)sc->triangles.size();
            o.tri_count = (uint32_t)jtri->arr.size();
            for (const auto &jtv : jtri->arr) {
                pt_triangle t;
                if (jtv->kind != JValue::Obj || !get_vec3(c, jtv->get("a"), "triangle.a", t.a) ||
                    !get_vec3(c, jtv->get("b"), "triangle.b", t.b) || !get_vec3(c, jtv->get("c"), "triangle.c", t.c)) {
                    pt::set_error("scene: " + where + ": " + (c.err.empty() ? "bad triangle" : c.err));
                    return PT_ERR_PARSE;
                }
                sc->triangles.push_back(t);
            }
        } else {
            pt::set_error("scene: " + where + ": unknown variant `" + tag + "`");
            return PT_ERR_PARSE;
        }
        sc->objects.push_back(o);
        sc->desc.push_back(od);
    }
    *out = sc.release();
    return PT_OK;
}

// SceneData::to_descriptor + SceneDescriptor::save (mod.rs:112-117, 127-149): serde_json::to_string_pretty
int pt_scene_save(const pt_scene *s, const char *path) {
    if (!s || !path) {
        pt::set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    Pretty w;
    w.out += '{';
    ++w.depth;
    w.key("id");
    w.out += json_escape(s->id);
    w.out += ',';
    w.key("objects");
    if (s->objects.empty()) {
        w.out += "[]";
    } else {
        w.out += '[';
        ++w.depth;
        for (size_t i = 0; i < s->objects.size(); ++i) {
            const pt_object &o = s->objects[i];
            const ObjDesc &d = s->desc[i];
            w.nl();
            w.out += '{';
            ++w.depth;
            w.key("type_");
            w.out += '{';
            ++w.depth;
            if (d.variant == 0) {
                w.key("Sphere");
                w.out += '{';
                ++w.depth;
                w.key("radius");
                w.out += fmt_f32(o.radius);
                --w.depth;
                w.nl();
                w.out += '}';
            } else if (d.variant == 1) {
                w.key("MeshFile");
                w.out += '{';
                ++w.depth;
                w.key("path");
                w.out += json_escape(d.path);
                w.out += ',';
                w.key("scale");
                w.out += fmt_f32(d.scale);
                --w.depth;
                w.nl();
                w.out += '}';
            } else {
                w.key("Mesh");
                w.out += '{';
                ++w.depth;
                w.key("triangles");
                w.triangles(s->triangles.data() + o.tri_offset, o.tri_count);
                w.out += ',';
                w.key("bounding_sphere");
                w.out += '{';
                ++w.depth;
                w.key("position");
                w.vec3(o.bs_center);
                w.out += ',';
                w.key("radius");
                w.out += fmt_f32(o.bs_radius);
                --w.depth;
                w.nl();
                w.out += "},";
                w.key("bounding_box");
                const std::vector<pt_triangle> box =
                    d.bounding_box.empty() ? box_triangles(s->triangles.data() + o.tri_offset, o.tri_count) : d.bounding_box;
                w.triangles(box.data(), box.size());
                --w.depth;
                w.nl();
                w.out += '}';
            }
            --w.depth;
            w.nl();
            w.out += "},";
            w.key("position");
            w.vec3(o.position);
            w.out += ',';
            w.key("material");
            w.out += '{';
            ++w.depth;
            w.key("color");
            w.vec3(o.color);
            w.out += ',';
            w.key("emmission");
            w.vec3(o.emission);
            w.out += ',';
            w.key("reflect_type");
            w.out += o.reflect_type == PT_DIFFUSE ? "\"Diffuse\"" : (o.reflect_type == PT_SPECULAR ? "\"Specular\"" : "\"Refract\"");
            --w.depth;
            w.nl();
            w.out += '}';
            --w.depth;
            w.nl();
            w.out += '}';
            if (i + 1 < s->objects.size()) w.out += ',';
        }
        --w.depth;
        w.nl();
        w.out += ']';
    }
    w.out += ',';
    w.key("camera");
    w.out += '{';
    ++w.depth;
    w.key("position");
    w.vec3(s->camera.position);
    w.out += ',';
    w.key("direction");
    w.vec3(s->camera.direction);
    w.out += ',';
    w.key("focal_length");
    w.out += fmt_f32(s->camera.focal_length);
    w.out += ',';
    w.key("sensor_width");
    w.out += fmt_f32(s->camera.sensor_width);
    w.out += ',';
    w.key("aspect_ratio");
    w.out += fmt_f32(s->camera.aspect_ratio);
    --w.depth;
    w.nl();
    w.out += '}';
    --w.depth;
    w.nl();
    w.out += '}';
    FILE *f = fopen(path, "wb");
    if (!f) {
        pt::set_error(std::string("cannot create ") + path);
        return PT_ERR_IO;
    }
    const size_t wr = fwrite(w.out.data(), 1, w.out.size(), f);
    if (fclose(f) != 0 || wr != w.out.size()) {
        pt::set_error(std::string("short write to ") + path);
        return PT_ERR_IO;
    }
    return PT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The scenes the reference builds in code when scenes/ holds no *.json (setup_scenes, scenes.rs:43-318; written to
// disk by load_scene_ids, scenes.rs:28-38).  Same values, same f32 arithmetic (e.g. the light's emission is
// (0.98, 1, 0.9) * 0.9 evaluated in f32), expressed as tables.
}  // extern "C"

namespace {

struct BuiltinSphere {
    float pos[3], radius, color[3], emission[3];
    uint32_t reflect;
};
struct BuiltinQuad {  // single_quad_mesh(size_x, size_y, axis, flip), scenes.rs:322-367
    float pos[3], size_x, size_y;
    int axis;
    bool flip;
    float color[3], emission[3];
};

constexpr float kBoxX = 2.6f, kBoxY = 2.0f, kBoxZ = 8.8f;  // BOX, scenes.rs:45-49

void add_sphere(pt_scene &sc, const BuiltinSphere &b) {
    pt_object o;
    memset(&o, 0, sizeof o);
    o.kind = PT_SPHERE;
    memcpy(o.position, b.pos, 12);
    o.radius = b.radius;
    memcpy(o.color, b.color, 12);
    memcpy(o.emission, b.emission, 12);
    o.reflect_type = b.reflect;
    sc.objects.push_back(o);
    sc.desc.push_back(ObjDesc());
}

// two triangles over the rectangle [-size_x, size_x] x [-size_y, size_y] spanned by axes (axis+1)%3 and (axis+2)%3
void add_quad(pt_scene &sc, const BuiltinQuad &q) {
    float v[4][3];
    const int i1 = (q.axis + 1) % 3, i2 = (q.axis + 2) % 3;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            float *p = v[2 * i + j];
            p[0] = p[1] = p[2] = 0.0f;
            p[i1] = i == 0 ? -q.size_x : q.size_x;
            p[i2] = j == 0 ? -q.size_y : q.size_y;
        }
    static const int order[2][2][3] = {{{0, 2, 1}, {1, 2, 3}}, {{0, 1, 2}, {2, 1, 3}}};  // [flip][triangle][corner]
    pt_triangle t[2];
    for (int k = 0; k < 2; ++k) {
        memcpy(t[k].a, v[order[q.flip][k][0]], 12);
        memcpy(t[k].b, v[order[q.flip][k][1]], 12);
        memcpy(t[k].c, v[order[q.flip][k][2]], 12);
    }
    pt_object o;
    memset(&o, 0, sizeof o);
    o.kind = PT_MESH;
    memcpy(o.position, q.pos, 12);
    memcpy(o.color, q.color, 12);
    memcpy(o.emission, q.emission, 12);
    o.reflect_type = PT_DIFFUSE;
    o.tri_offset = (uint32_t)sc.triangles.size();
    o.tri_count = 2;
    pt::host::mesh_bounding_sphere(t, 2, o.bs_center, &o.bs_radius);  // Mesh::new, mod.rs:450-499
    sc.triangles.push_back(t[0]);
    sc.triangles.push_back(t[1]);
    sc.objects.push_back(o);
    ObjDesc d;
    d.variant = 2;
    sc.desc.push_back(d);
}

// the six walls and the ceiling light shared by "cornell" and "mesh" (scenes.rs:51-123)
void add_cornell_box(pt_scene &sc) {
    const float light = 0.9f;
    const BuiltinQuad walls[7] = {
        {{kBoxX, 0.0f, 0.0f}, kBoxY, kBoxZ, 0, true, {0.85f, 0.25f, 0.25f}, {0.0f, 0.0f, 0.0f}},    // right, red
        {{-kBoxX, 0.0f, 0.0f}, kBoxY, kBoxZ, 0, false, {0.25f, 0.35f, 0.85f}, {0.0f, 0.0f, 0.0f}},  // left, blue
        {{0.0f, kBoxY, 0.0f}, kBoxZ, kBoxX, 1, true, {0.8f, 0.8f, 0.8f}, {0.0f, 0.0f, 0.0f}},       // top
        {{0.0f, -kBoxY, 0.0f}, kBoxZ, kBoxX, 1, false, {0.7f, 0.7f, 0.7f}, {0.0f, 0.0f, 0.0f}},     // bottom
        {{0.0f, 0.0f, -kBoxZ}, kBoxX, kBoxY, 2, true, {0.95f, 0.95f, 0.95f}, {0.0f, 0.0f, 0.0f}},   // back
        {{0.0f, 0.0f, kBoxZ}, kBoxX, kBoxY, 2, true, {0.05f, 0.05f, 0.05f}, {0.0f, 0.0f, 0.0f}},    // front
        {{0.0f, kBoxY - 0.04f, 0.0f}, kBoxZ, kBoxX, 1, true, {0.98f, 1.0f, 0.9f},
         {0.98f * light, 1.0f * light, 0.9f * light}},                                               // ceiling light
    };
    for (const BuiltinQuad &w : walls) add_quad(sc, w);
}

// CameraData::new (mod.rs:178-186): direction.normalize() = v * (1 / length) in glam
pt_camera builtin_camera(float px, float py, float pz, float dx, float dy, float dz) {
    pt_camera c;
    c.position[0] = px;
    c.position[1] = py;
    c.position[2] = pz;
    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float inv = 1.0f / len;
    c.direction[0] = dx * inv;
    c.direction[1] = dy * inv;
    c.direction[2] = dz * inv;
    c.focal_length = 0.035f;
    c.sensor_width = 0.036f;
    c.aspect_ratio = 3.0f / 2.0f;
    return c;
}

const char *const kBuiltinIds[] = {"single-sphere", "cartesian", "two-spheres", "three-spheres", "cornell", "mesh"};

}  // namespace

extern "C" {

uint32_t pt_builtin_scene_count(void) { return (uint32_t)(sizeof kBuiltinIds / sizeof kBuiltinIds[0]); }

const char *pt_builtin_scene_id(uint32_t i) { return i < pt_builtin_scene_count() ? kBuiltinIds[i] : nullptr; }

int pt_scene_builtin(const char *id, const char *base_dir, pt_scene **out) {
    if (!id || !out) {
        pt::set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    *out = nullptr;
    std::unique_ptr<pt_scene> sc(new pt_scene());
    sc->id = id;
    sc->camera = builtin_camera(0.0f, -kBoxY + 1.8f, kBoxZ - 1.0f, 0.0f, -0.06f, -1.0f);  // default_camera
    const std::string name = id;
    const uint32_t D = PT_DIFFUSE;
    if (name == "single-sphere") {
        add_sphere(*sc, {{0, 0, 0}, 1.0f, {1, 1, 1}, {0.98f * 15.0f, 15.0f, 0.9f * 15.0f}, D});
    } else if (name == "cartesian") {
        add_sphere(*sc, {{0, 0, 0}, 0.3f, {0.9f, 0.9f, 0.9f}, {0, 0, 0}, D});
        add_sphere(*sc, {{1, 0, 0}, 0.3f, {0.8f, 0, 0}, {0, 0, 0}, D});
        add_sphere(*sc, {{-1, 0, 0}, 0.3f, {0, 0, 0.8f}, {0, 0, 0}, D});
        add_sphere(*sc, {{0, 1, 0}, 0.3f, {0, 0.8f, 0}, {0, 0, 0}, D});
    } else if (name == "two-spheres") {
        add_sphere(*sc, {{0, 0, 0}, 1.0f, {1, 0, 0}, {0, 0, 0}, D});
        add_sphere(*sc, {{0, 0, 10}, 1.0f, {0, 0, 0}, {10, 10, 10}, D});
    } else if (name == "three-spheres") {
        add_sphere(*sc, {{0, 0, -3}, 1.0f, {1.0f, 0.2f, 0.2f}, {0, 0, 0}, D});
        add_sphere(*sc, {{4, 2, 0}, 1.0f, {0, 0, 0}, {20, 10, 10}, D});
        add_sphere(*sc, {{-6, -2, 0}, 1.0f, {0, 0, 0}, {5, 9, 20}, D});
    } else if (name == "cornell") {
        const float y = -kBoxY + 0.8f;
        add_sphere(*sc, {{-1.3f, y, -1.3f}, 0.8f, {0.999f, 0.999f, 0.999f}, {0, 0, 0}, PT_SPECULAR});
        add_sphere(*sc, {{1.3f, y, -0.2f}, 0.8f, {0.999f, 0.999f, 0.999f}, {0, 0, 0}, PT_REFRACT});
        add_sphere(*sc, {{0.08f, y, -0.8f}, 0.5f, {0.999f, 0.999f, 0.999f}, {0.98f * 2.0f, 1.0f * 2.0f, 0.9f * 2.0f}, D});
        add_sphere(*sc, {{-0.08f, y, 0.7f}, 0.5f, {0.4f, 0.9f, 0.49f}, {0, 0, 0}, D});
        add_cornell_box(*sc);
    } else if (name == "mesh") {
        const char *rel = "meshes/mctri.off";
        const float scale = 0.16f;
        std::vector<pt_triangle> tl;
        std::string err;
        int rc = load_off_impl(std::string(base_dir ? base_dir : ".") + "/" + rel, scale, 0u, tl, err);
        if (rc) {
            pt::set_error("builtin scene `mesh`: " + err);
            return rc;
        }
        if (tl.empty()) {
            pt::set_error("builtin scene `mesh`: mesh file has no triangles");
            return PT_ERR_PARSE;
        }
        pt_object o;
        memset(&o, 0, sizeof o);
        o.kind = PT_MESH;
        o.position[0] = -0.8f;
        o.position[1] = -kBoxY + 0.5f;
        o.position[2] = 0.0f;
        o.color[0] = 234.0f / 255.0f;
        o.color[1] = 1.0f;
        o.color[2] = 0.0f;
        o.reflect_type = PT_DIFFUSE;
        o.tri_offset = 0;
        o.tri_count = (uint32_t)tl.size();
        pt::host::mesh_bounding_sphere(tl.data(), o.tri_count, o.bs_center, &o.bs_radius);
        sc->triangles = tl;
        sc->objects.push_back(o);
        ObjDesc d;
        d.variant = 1;
        d.path = rel;
        d.scale = scale;
        sc->desc.push_back(d);
        add_cornell_box(*sc);
        sc->camera = builtin_camera(0.9f, -kBoxY + 1.8f, kBoxZ - 1.0f, -0.09f, -0.06f, -1.0f);
    } else {
        pt::set_error("unknown builtin scene `" + name + "`");
        return PT_ERR_INVALID;
    }
    *out = sc.release();
    return PT_OK;
}

// camera edits from a host (the GUI moves the camera, then saves: src/main.rs:245-251)
int pt_scene_set_camera(pt_scene *s, const pt_camera *cam) {
    if (!s || !cam) {
        pt::set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    s->camera = *cam;
    return PT_OK;
}

void pt_scene_free(pt_scene *s) { delete s; }
const char *pt_scene_id(const pt_scene *s) { return s ? s->id.c_str() : ""; }
const pt_camera *pt_scene_camera(const pt_scene *s) { return s ? &s->camera : nullptr; }
const pt_object *pt_scene_objects(const pt_scene *s, uint32_t *n) {
    if (n) *n = s ? (uint32_t)s->objects.size() : 0;
    return s ? s->objects.data() : nullptr;
}
const pt_triangle *pt_scene_triangles(const pt_scene *s, uint32_t *n) {
    if (n) *n = s ? (uint32_t)s->triangles.size() : 0;
    return s ? s->triangles.data() : nullptr;
}
const pt_triangle *pt_scene_bounding_box(const pt_scene *cs, uint32_t object) {
    pt_scene *s = const_cast<pt_scene *>(cs);  // the computed box of a MeshFile / API mesh is cached in its descriptor
    if (!s || object >= s->objects.size() || s->objects[object].kind != PT_MESH) return nullptr;
    if (s->desc.size() < s->objects.size()) s->desc.resize(s->objects.size());
    ObjDesc &d = s->desc[object];
    if (d.bounding_box.size() != 12) {
        const pt_object &o = s->objects[object];
        d.bounding_box = box_triangles(s->triangles.data() + o.tri_offset, o.tri_count);
    }
    return d.bounding_box.data();
}

// SipHash-c-d (Aumasson & Bernstein), little-endian message words, 64-bit tag
uint64_t pt_siphash(uint32_t c_rounds, uint32_t d_rounds, uint64_t k0, uint64_t k1, const uint8_t *data, size_t len) {
    uint64_t s0 = k0 ^ 0x736f6d6570736575ull, s1 = k1 ^ 0x646f72616e646f6dull;
    uint64_t s2 = k0 ^ 0x6c7967656e657261ull, s3 = k1 ^ 0x7465646279746573ull;
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    auto round = [&]() {
        s0 += s1, s2 += s3;
        s1 = rotl(s1, 13) ^ s0, s3 = rotl(s3, 16) ^ s2;
        s0 = rotl(s0, 32);
        s2 += s1, s0 += s3;
        s1 = rotl(s1, 17) ^ s2, s3 = rotl(s3, 21) ^ s0;
        s2 = rotl(s2, 32);
    };
    const size_t whole = len / 8;
    for (size_t w = 0; w < whole; ++w) {
        uint64_t mword = 0;
        for (int k = 7; k >= 0; --k) mword = (mword << 8) | data[8 * w + (size_t)k];
        s3 ^= mword;
        for (uint32_t r = 0; r < c_rounds; ++r) round();
        s0 ^= mword;
    }
    uint64_t last = (uint64_t)(len & 0xff) << 56;
    for (size_t k = 0; k < (len & 7); ++k) last |= (uint64_t)data[8 * whole + k] << (8 * k);
    s3 ^= last;
    for (uint32_t r = 0; r < c_rounds; ++r) round();
    s0 ^= last;
    s2 ^= 0xff;
    for (uint32_t r = 0; r < d_rounds; ++r) round();
    return s0 ^ s1 ^ s2 ^ s3;
}

uint64_t pt_image_hash(const float *rgb, size_t n_floats) {
    // v.x.to_bits().hash(..) for every component in order == the byte stream of the f32 array (mod.rs:919-923)
    return pt_siphash(1, 3, 0, 0, reinterpret_cast<const uint8_t *>(rgb), n_floats * sizeof(float));
}

float pt_gamma_correction(float x) {
    const float c = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);  // f32::clamp keeps NaN
    return powf(c, 1.0f / 2.2f);
}

uint32_t pt_to_int_with_gamma_correction(float x) {
    const float v = 255.0f * pt_gamma_correction(x) + 0.5f;
    if (!(v > 0.0f)) return 0;  // `as usize`: NaN and negatives -> 0
    return (uint32_t)v;
}

int pt_write_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height, uint32_t spp,
                 const char *scene_id, uint64_t seconds) {
    if (!path || !rgb || !scene_id) {
        pt::set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    FILE *f = fopen(path, "wb");
    if (!f) {
        pt::set_error(std::string("cannot create ") + path);
        return PT_ERR_IO;
    }
    std::string buf;
    buf.reserve((size_t)width * height * 12 + 256);
    char tmp[160];
    buf += "P3\n";
    buf += "# samplesPerPixel: " + std::to_string(spp) + ", resolution_y: " + std::to_string(height) +
           ", scene_id: " + scene_id + "\n";
    buf += "# rendering time: " + std::to_string(seconds) + " s\n";
    buf += std::to_string(width) + " " + std::to_string(height) + "\n255\n";
    const size_t npx = (size_t)width * height;
    for (size_t i = npx; i-- > 0;) {  // pixels.iter().rev(), mod.rs:1065
        int k = snprintf(tmp, sizeof tmp, "%u %u %u ", pt_to_int_with_gamma_correction(rgb[3 * i]),
                         pt_to_int_with_gamma_correction(rgb[3 * i + 1]), pt_to_int_with_gamma_correction(rgb[3 * i + 2]));
        buf.append(tmp, (size_t)k);
    }
    const size_t wr = fwrite(buf.data(), 1, buf.size(), f);
    const int cl = fclose(f);
    if (wr != buf.size() || cl != 0) {
        pt::set_error(std::string("short write to ") + path);
        return PT_ERR_IO;
    }
    return PT_OK;
}

}  // extern "C"
