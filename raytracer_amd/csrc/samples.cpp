// samples.cpp - librt_samples.so: host-side scene construction (no GPU code).
//
// Host mirror of the reference's scene-building surface so that the reference's
// sample scenes drive the MI355X render core unchanged:
//   Scene::new/add_* + Entity builder   raytracer-impl/src/implementation.rs:232-256,340-370
//   Camera::new                         raytracer-impl/src/implementation.rs:462-480
//   CameraConfiguration::make_camera    raytracer-samples/src/scene.rs:27-48
//   OBJ / MTL ingest                    raytracer-obj/src/format.rs:168-313, builder.rs:39-104
//   scene factories                     raytracer-samples/src/samples.rs, scene_*.rs, lib.rs:18-37
// Everything is f32 in the reference's operation order (build with
// -ffp-contract=off) because sphere positions, camera bases and rotations feed
// straight into the parity-checked path.
#include "../../include/rt_samples.h"
#include "image_decode.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

thread_local std::string g_error;
thread_local std::string g_warning;

struct Vec { float x, y, z; };
inline Vec mk(float x, float y, float z) { return Vec{x, y, z}; }
inline Vec add(Vec a, Vec b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec sub(Vec a, Vec b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec scale(Vec a, float f) { return mk(a.x * f, a.y * f, a.z * f); }
inline Vec divide(Vec a, float f) { return mk(a.x / f, a.y / f, a.z / f); }
inline float dotp(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec crossp(Vec a, Vec b) { return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x); }
inline float norm(Vec a) { return sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z)); }
inline Vec normalized(Vec a) { float l = norm(a); return l == 0.0f ? a : divide(a, l); }   // types.rs:41-44
inline bool is_zero(Vec a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
inline Vec rotated(Vec p, Vec axis, float theta) {                                          // types.rs:191-201
    Vec t1 = scale(p, cosf(theta));
    Vec t2 = scale(crossp(axis, p), sinf(theta));
    Vec t3 = scale(scale(axis, dotp(axis, p)), 1.0f - cosf(theta));
    return add(add(t1, t2), t3);
}
inline RtV3 out3(Vec a) { return RtV3{a.x, a.y, a.z}; }
inline Vec in3(const float* p) { return mk(p[0], p[1], p[2]); }
inline Vec in3(RtV3 p) { return mk(p.x, p.y, p.z); }

const float kPi = 3.14159265358979323846f;
const Vec kX = {1, 0, 0}, kY = {0, 1, 0}, kZ = {0, 0, 1}, kZero = {0, 0, 0}, kOne = {1, 1, 1};

inline float deg_to_rad(float deg) { return (deg / 180.0f) * kPi; }                         // samples util.rs:7-9
inline Vec lerp_v3(Vec p1, Vec p2, float d) { return add(p1, scale(sub(p2, p1), d)); }      // util.rs:15-18
inline float lerp_f32(float p1, float p2, float d) { return p1 + (p2 - p1) * d; }           // util.rs:20-23
inline float ease_in(float t, float s) { return powf(t, s); }                               // util.rs:25-28
inline float ease_out(float t, float s) { return 1.0f - powf(1.0f - t, s); }                // util.rs:30-33
inline float ease_in_out(float t, float s) { return lerp_f32(ease_in(t, s), ease_out(t, s), t); } // util.rs:35-37

// rand_xorshift 0.4.0 seeded like create_rng_from_seed (samples util.rs:45-60):
// the u128 is written big-endian into the 16 seed bytes, which from_seed reads
// as four little-endian u32.
struct SceneRng {
    uint32_t s[4];
    explicit SceneRng(unsigned __int128 seed) {
        uint8_t bytes[16];
        for (int off = 0; off < 16; off++) bytes[off] = (uint8_t)((seed >> ((15 - off) * 8)) & 0xff);
        bool all_zero = true;
        for (int i = 0; i < 4; i++) {
            s[i] = (uint32_t)bytes[4 * i] | ((uint32_t)bytes[4 * i + 1] << 8) | ((uint32_t)bytes[4 * i + 2] << 16) | ((uint32_t)bytes[4 * i + 3] << 24);
            all_zero = all_zero && s[i] == 0;
        }
        if (all_zero) s[0] = s[1] = s[2] = s[3] = 0x0BAD5EEDu;
    }
    uint32_t next_u32() {
        uint32_t t = s[0] ^ (s[0] << 11);
        s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
        s[3] = s[3] ^ (s[3] >> 19) ^ (t ^ (t >> 8));
        return s[3];
    }
    float random_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }           // rand 0.9 StandardUniform
    // rand 0.9 UniformFloat<f32>::sample_single for low..high
    float random_range(float low, float high) {
        float scale_ = high - low;
        for (;;) {
            uint32_t bits = (next_u32() >> 9) | 0x3F800000u;
            float v12; memcpy(&v12, &bits, 4);
            float res = (v12 - 1.0f) * scale_ + low;
            if (res < high) return res;
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------
// The builder: owns every array an RtSceneDesc points at.
// ---------------------------------------------------------------------------
struct RtSceneBuilder {
    RtCamera camera{};
    uint32_t sky = RT_SKY_DAY;
    std::vector<RtEntity> entities;
    std::vector<RtV3> translations;
    std::vector<RtRotation> rotations;
    std::vector<RtSphere> spheres;
    std::vector<RtPlane> planes;
    std::vector<RtMeshObject> mesh_objects;
    std::vector<std::vector<RtMeshTri>> mesh_tris;
    std::vector<RtMesh> meshes;
    std::vector<RtMaterial> materials;
    std::vector<RtTexture> textures;
    std::vector<int32_t> tex_set_entries;
    std::vector<RtV3> texels;
    std::vector<RtLight> lights;
    RtSceneDesc desc{};

    // --- Scene::add_* ---
    uint32_t add_material(const RtMaterial& m) { materials.push_back(m); return (uint32_t)materials.size() - 1; }
    uint32_t add_texture(const RtTexture& t) { textures.push_back(t); return (uint32_t)textures.size() - 1; }
    // rgba_to_v3, color_map.rs:9-14: channel as f32 / 255.0, alpha dropped
    uint32_t add_color_map_rgba8(uint32_t w, uint32_t h, const uint8_t* rgba) {
        RtTexture t{}; t.kind = RT_TEX_COLOR_MAP; t.width = w; t.height = h; t.texel_first = texels.size(); t.child0 = t.child1 = -1;
        texels.reserve(texels.size() + (size_t)w * h);
        for (uint64_t i = 0; i < (uint64_t)w * h; i++)
            texels.push_back(RtV3{(float)rgba[4 * i] / 255.0f, (float)rgba[4 * i + 1] / 255.0f, (float)rgba[4 * i + 2] / 255.0f});
        return add_texture(t);
    }
    void add_light(const RtLight& l) { lights.push_back(l); }
    uint32_t add_mesh(std::vector<RtMeshTri> tris) { mesh_tris.push_back(std::move(tris)); return (uint32_t)mesh_tris.size() - 1; }

    const RtSceneDesc* finish() {
        meshes.resize(mesh_tris.size());
        for (size_t i = 0; i < mesh_tris.size(); i++) meshes[i] = RtMesh{mesh_tris[i].data(), mesh_tris[i].size()};
        desc.abi_version = RT_ABI_VERSION;
        desc.camera = camera; desc.sky = sky;
        desc.entities = entities.data(); desc.entity_count = entities.size();
        desc.translations = translations.data(); desc.translation_count = translations.size();
        desc.rotations = rotations.data(); desc.rotation_count = rotations.size();
        desc.spheres = spheres.data(); desc.sphere_count = spheres.size();
        desc.planes = planes.data(); desc.plane_count = planes.size();
        desc.mesh_objects = mesh_objects.data(); desc.mesh_object_count = mesh_objects.size();
        desc.meshes = meshes.data(); desc.mesh_count = meshes.size();
        desc.materials = materials.data(); desc.material_count = materials.size();
        desc.textures = textures.data(); desc.texture_count = textures.size();
        desc.tex_set_entries = tex_set_entries.data(); desc.tex_set_entry_count = tex_set_entries.size();
        desc.texels = texels.data(); desc.texel_count = texels.size();
        desc.lights = lights.data(); desc.light_count = lights.size();
        return &desc;
    }
};

namespace {

// --- materials.rs builders -------------------------------------------------
RtMaterial lambertian(float reflectivity = 0.0f) { return RtMaterial{RT_MAT_LAMBERTIAN, reflectivity, 0.0f, 0.0f, 0.0f}; }
RtMaterial specular(float reflectiveness = 1.0f, float fuzz = 0.0f) { return RtMaterial{RT_MAT_SPECULAR, reflectiveness, fuzz, 0.0f, 0.0f}; }
RtMaterial dielectric(float reflectivity = 1.0f, float opacity = 0.0f, float ref_index = 1.5f) {
    return RtMaterial{RT_MAT_DIELECTRIC, reflectivity, 0.0f, opacity, ref_index};
}
// --- texture.rs builders -----------------------------------------------------
RtTexture color_texture(Vec c) { RtTexture t{}; t.kind = RT_TEX_COLOR; t.color = out3(c); t.child0 = t.child1 = -1; return t; }
Vec rgb(int r, int g, int b) { return mk((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f); }  // samples.rs:57-61
// --- lights.rs builders -------------------------------------------------------
RtLight point_light(Vec origin, float intensity) {                                   // lights.rs:16-22 (default 100)
    RtLight l{}; l.kind = RT_LIGHT_POINT; l.origin = out3(origin); l.color = out3(kOne); l.intensity = intensity; return l;
}
RtLight lamp_light(Vec origin, Vec direction, float intensity, float angle_deg) {    // lights.rs:61-69
    RtLight l{}; l.kind = RT_LIGHT_LAMP; l.origin = out3(origin); l.direction = out3(normalized(direction));
    l.color = out3(kOne); l.intensity = intensity; l.angle_deg = angle_deg; return l;
}
RtLight directional_light(Vec direction, float intensity) {                          // lights.rs:119-125
    RtLight l{}; l.kind = RT_LIGHT_DIRECTIONAL; l.direction = out3(normalized(direction)); l.color = out3(kOne); l.intensity = intensity; return l;
}

// Camera::new, implementation.rs:462-480
RtCamera camera_new(Vec look_from, Vec look_at, float v_fov, float aspect_ratio, float lens_radius, float focus_dist) {
    Vec v_up = mk(0.0f, 1.0f, 0.0f);
    float theta = v_fov * kPi / 180.0f;
    float half_height = tanf(theta / 2.0f);
    float half_width = aspect_ratio * half_height;
    Vec w = normalized(sub(look_from, look_at));
    Vec u = normalized(crossp(v_up, w));
    Vec v = crossp(w, u);
    RtCamera c{};
    Vec llc = sub(sub(sub(look_from, scale(scale(u, half_width), focus_dist)), scale(scale(v, half_height), focus_dist)), scale(w, focus_dist));
    c.lower_left_corner = out3(llc);
    c.horizontal = out3(scale(u, 2.0f * half_width * focus_dist));
    c.vertical = out3(scale(v, 2.0f * half_height * focus_dist));
    c.origin = out3(look_from); c.u = out3(u); c.v = out3(v); c.lens_radius = lens_radius;
    return c;
}

// CameraConfiguration::make_camera, raytracer-samples/src/scene.rs:27-48
RtCamera make_camera(const RtCameraConfig& cfg, Vec look_to, Vec default_look_from) {
    Vec p = sub(default_look_from, look_to);
    Vec v_axis = kY;
    p = rotated(p, v_axis, deg_to_rad(cfg.angle_adjust_h));
    Vec w = normalized(sub(kZero, p));
    Vec h_axis = normalized(crossp(v_axis, w));
    p = rotated(p, h_axis, deg_to_rad(cfg.angle_adjust_v));
    Vec look_from = add(p, look_to);
    float dist_to_focus = norm(sub(look_from, look_to)) + cfg.focus_dist_adjust;
    return camera_new(look_from, look_to, cfg.fov, cfg.width / cfg.height, cfg.lens_radius, dist_to_focus);
}

// Plane::new, plane.rs:29-48
RtPlane plane_new(Vec normal, uint32_t mat, uint32_t tex) {
    normal = normalized(normal);
    Vec u_basis = normalized(crossp(normal, kX));
    if (is_zero(u_basis)) u_basis = normalized(crossp(normal, kY));
    Vec v_basis = normalized(crossp(normal, u_basis));
    RtPlane p{}; p.normal = out3(normal); p.u_basis = out3(u_basis); p.v_basis = out3(v_basis);
    p.radius = 0.0f; p.has_radius = 0; p.mat_id = mat; p.tex_id = tex;
    return p;
}

// Entity builder, implementation.rs:232-256
struct EntityB {
    uint32_t kind = RT_HITABLE_SPHERE;
    RtSphere sphere{}; RtPlane plane{}; RtMeshObject mesh{};
    int32_t id_ = -1;
    std::vector<RtV3> tr; std::vector<RtRotation> ro;
    static EntityB of(RtSphere s) { EntityB e; e.kind = RT_HITABLE_SPHERE; e.sphere = s; return e; }
    static EntityB of(RtPlane p) { EntityB e; e.kind = RT_HITABLE_PLANE; e.plane = p; return e; }
    static EntityB of(RtMeshObject m) { EntityB e; e.kind = RT_HITABLE_MESH; e.mesh = m; return e; }
    EntityB id(uint32_t v) { id_ = (int32_t)v; return *this; }
    EntityB translate(Vec o) { tr.push_back(out3(o)); return *this; }
    EntityB rotate(Vec axis, float theta) { ro.push_back(RtRotation{out3(axis), theta}); return *this; }
};

int add_entity(RtSceneBuilder& b, const EntityB& e) {                                // Scene::add_entity
    RtEntity d{};
    d.hitable_kind = e.kind; d.id = e.id_;
    switch (e.kind) {
        case RT_HITABLE_SPHERE: d.hitable_index = (uint32_t)b.spheres.size(); b.spheres.push_back(e.sphere); break;
        case RT_HITABLE_PLANE: d.hitable_index = (uint32_t)b.planes.size(); b.planes.push_back(e.plane); break;
        default: d.hitable_index = (uint32_t)b.mesh_objects.size(); b.mesh_objects.push_back(e.mesh); break;
    }
    d.translation_first = (uint32_t)b.translations.size(); d.translation_count = (uint32_t)e.tr.size();
    b.translations.insert(b.translations.end(), e.tr.begin(), e.tr.end());
    d.rotation_first = (uint32_t)b.rotations.size(); d.rotation_count = (uint32_t)e.ro.size();
    b.rotations.insert(b.rotations.end(), e.ro.begin(), e.ro.end());
    b.entities.push_back(d);
    return (int)b.entities.size() - 1;
}

RtSphere sphere(float radius, uint32_t mat, uint32_t tex) { return RtSphere{radius, mat, tex}; }
RtMeshObject mesh_object(uint32_t mesh, uint32_t mat, uint32_t tex) { return RtMeshObject{mesh, mat, tex}; }

// ---------------------------------------------------------------------------
// OBJ / MTL ingest: raytracer-obj/src/format.rs + builder.rs
// ---------------------------------------------------------------------------
std::string trimmed(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// try_parse_elements (format.rs:94-107): split on EVERY whitespace char; an
// empty piece (double space) fails the parse, as does a surplus piece.
bool split_exact(const std::string& data, size_t n, std::vector<std::string>& parts) {
    parts.clear();
    std::string cur;
    for (char c : data) {
        if (isspace((unsigned char)c)) { parts.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    parts.push_back(cur);
    return parts.size() == n;
}
bool parse_f32(const std::string& s, float& out) {
    if (s.empty()) return false;
    char* end = nullptr; out = strtof(s.c_str(), &end);
    return end && *end == '\0';
}
bool parse_usize(const std::string& s, size_t& out) {
    if (s.empty()) return false;
    size_t i = (s[0] == '+') ? 1 : 0;
    if (i >= s.size()) return false;
    size_t v = 0;
    for (; i < s.size(); i++) { if (s[i] < '0' || s[i] > '9') return false; v = v * 10 + (size_t)(s[i] - '0'); }
    out = v; return true;
}
template <size_t N> bool parse_floats(const std::string& data, float (&out)[N]) {
    std::vector<std::string> parts;
    if (!split_exact(data, N, parts)) return false;
    for (size_t i = 0; i < N; i++) if (!parse_f32(parts[i], out[i])) return false;
    return true;
}

struct ObjVertex { size_t vertex_index = 0; bool has_uv = false; size_t uv_index = 0; };
bool parse_obj_vertex(const std::string& s, ObjVertex& v) {                          // format.rs:61-85  `v/vt?/vn?`
    std::vector<std::string> parts; std::string cur;
    for (char c : s) { if (c == '/') { parts.push_back(cur); cur.clear(); } else cur.push_back(c); }
    parts.push_back(cur);
    if (parts.empty() || parts.size() > 3) return false;
    if (!parse_usize(parts[0], v.vertex_index)) return false;
    v.has_uv = false;
    if (parts.size() >= 2 && !parts[1].empty()) { if (!parse_usize(parts[1], v.uv_index)) return false; v.has_uv = true; }
    if (parts.size() == 3 && !parts[2].empty()) { size_t n; if (!parse_usize(parts[2], n)) return false; }
    return true;
}

struct ObjFace { ObjVertex a, b, c; bool has_mtl; std::string mtl; };
struct ObjMaterial { std::string name; Vec ambient = kZero, specular = kZero, diffuse = kZero; bool has_map = false; std::string map; };

struct ObjData {
    bool has_mtllib = false; std::string mtllib;
    std::vector<Vec> vertices; std::vector<RtV2> uv;
    std::vector<ObjFace> faces;   // in group order == file order (groups only regroup consecutive faces)
};

bool split_directive(const std::string& line, std::string& key, std::string& rest) {  // str::split_once(' ')
    size_t p = line.find(' ');
    if (p == std::string::npos) return false;
    key = line.substr(0, p); rest = line.substr(p + 1); return true;
}

int parse_obj(const std::string& path, ObjData& out) {                               // format.rs:168-225
    std::ifstream f(path);
    if (!f) { g_error = "load_obj: expected obj file at path " + path; return RT_E_IO; }
    std::string raw; bool has_mtl = false; std::string mtl; size_t line_no = 0;
    while (std::getline(f, raw)) {
        std::string line = trimmed(raw);
        size_t this_line = line_no++;
        if (!line.empty() && line[0] == '#') continue;
        std::string key, data;
        if (!split_directive(line, key, data)) continue;
        if (key == "mtllib") { out.has_mtllib = true; out.mtllib = trimmed(data); }
        else if (key == "usemtl") { has_mtl = true; mtl = trimmed(data); }
        else if (key == "v") {
            float v[3];
            if (!parse_floats(data, v)) { g_error = "Unable to parse vertex on line " + std::to_string(this_line) + ": " + data; return RT_E_IO; }
            out.vertices.push_back(mk(v[0], v[1], v[2]));
        } else if (key == "vt") {
            float v[2];
            if (!parse_floats(data, v)) { g_error = "Unable to parse texture vertex on line " + std::to_string(this_line) + ": " + data; return RT_E_IO; }
            out.uv.push_back(RtV2{v[0], v[1]});
        } else if (key == "f") {
            std::vector<std::string> parts; ObjFace face; face.has_mtl = has_mtl; face.mtl = mtl;
            if (!split_exact(data, 3, parts) || !parse_obj_vertex(parts[0], face.a) || !parse_obj_vertex(parts[1], face.b) || !parse_obj_vertex(parts[2], face.c)) {
                g_error = "Unable to parse face on line " + std::to_string(this_line) + ": " + data; return RT_E_IO;
            }
            out.faces.push_back(face);
        }
        // "g" only regroups consecutive faces (format.rs:190-193) and build_mesh
        // concatenates every group in order, so groups need no representation;
        // "vn", "o", "s" are ignored (format.rs:207-209,217).
    }
    return RT_OK;
}

int parse_mtl(const std::string& path, std::vector<ObjMaterial>& out) {              // format.rs:264-313
    std::ifstream f(path);
    if (!f) { g_error = "load_mtl: expected mtl file at path " + path; return RT_E_IO; }
    std::string raw; bool open = false; ObjMaterial cur; size_t line_no = 0;
    while (std::getline(f, raw)) {
        std::string line = trimmed(raw);
        size_t this_line = line_no++;
        if (!line.empty() && line[0] == '#') continue;
        std::string key, data;
        if (!split_directive(line, key, data)) continue;
        float c[3];
        if (key == "newmtl") { if (open) out.push_back(cur); cur = ObjMaterial{}; cur.name = trimmed(data); open = true; }
        else if (key == "Ka" || key == "Ks" || key == "Kd") {
            if (!parse_floats(data, c)) { g_error = "Unable to parse " + key + " on line " + std::to_string(this_line) + ": " + data; return RT_E_IO; }
            // NOTE format.rs:284-300 stores colours in the parse state even before the first newmtl
            (key == "Ka" ? cur.ambient : key == "Ks" ? cur.specular : cur.diffuse) = mk(c[0], c[1], c[2]);
        } else if (key == "map_Kd") { cur.has_map = true; cur.map = trimmed(data); }
    }
    if (open) out.push_back(cur);
    return RT_OK;
}

// load_obj_builder(path).build_mesh(), builder.rs:39-134.  The reference orders
// the MeshTextureSet by HashSet iteration (nondeterministic, builder.rs:44-47);
// here: order of first use.  tex_key -> texture NAME is what is stable.
int load_obj_mesh(RtSceneBuilder& b, const std::string& path, int* texture_set_out) {
    ObjData obj;
    int rc = parse_obj(path, obj);
    if (rc) return rc;
    std::map<std::string, ObjMaterial> materials;
    if (obj.has_mtllib) {
        size_t slash = path.find_last_of('/');
        std::string mtl_path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + obj.mtllib;
        std::vector<ObjMaterial> mats;
        rc = parse_mtl(mtl_path, mats);
        if (rc) return rc;
        for (auto& m : mats) materials[m.name] = m;     // HashMap insert: later duplicates win
    }
    std::vector<std::string> names;                       // texture set, first-use order
    for (const ObjFace& f : obj.faces) {
        if (!f.has_mtl) continue;
        bool seen = false;
        for (auto& n : names) if (n == f.mtl) { seen = true; break; }
        // a material the MTL does not define is skipped with a warning (builder.rs:53-56)
        if (!seen && materials.count(f.mtl)) names.push_back(f.mtl);
    }
    std::vector<RtMeshTri> tris; tris.reserve(obj.faces.size());
    for (const ObjFace& f : obj.faces) {
        int32_t key = -1;
        if (f.has_mtl) for (size_t i = 0; i < names.size(); i++) if (names[i] == f.mtl) { key = (int32_t)i; break; }
        auto vertex = [&](size_t i, Vec& o) { if (i == 0 || i > obj.vertices.size()) return false; o = obj.vertices[i - 1]; return true; };
        auto uvof = [&](const ObjVertex& v) { return (v.has_uv && v.uv_index >= 1 && v.uv_index <= obj.uv.size()) ? obj.uv[v.uv_index - 1] : RtV2{0.0f, 0.0f}; };
        Vec a, bb, c;
        if (!vertex(f.a.vertex_index, a) || !vertex(f.b.vertex_index, bb) || !vertex(f.c.vertex_index, c)) { g_error = "vertex by index"; return RT_E_IO; }
        RtMeshTri t{}; t.a = out3(a); t.b = out3(bb); t.c = out3(c);
        t.a_uv = uvof(f.a); t.b_uv = uvof(f.b); t.c_uv = uvof(f.c); t.tex_key = key;
        tris.push_back(t);
    }
    if (tris.empty()) { g_error = "[ObjMeshBuilder::inner_build_mesh] expected at least one face"; return RT_E_IO; }
    // colour maps of EVERY material of the MTL, used or not; a file that is missing or
    // does not decode fails the whole load (builder.rs:121-131).  One exception, not in
    // the reference: a git-lfs pointer (the checked-in state of the reference's .bmp
    // files) is skipped with a warning, leaving the material's flat diffuse colour.
    std::map<std::string, int32_t> color_maps;            // map file name -> TexId (shared, builder.rs:127)
    {
        size_t slash = path.find_last_of('/');
        const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
        for (auto& kv : materials) {
            const ObjMaterial& m = kv.second;
            if (!m.has_map || color_maps.count(m.map)) continue;
            rtimg::Image img; std::string err;
            if (!rtimg::decode_file(dir + m.map, img, err)) {
                if (img.lfs_pointer) { g_warning = "WARNING: colour map skipped (" + err + ")"; continue; }
                g_error = err; return RT_E_IO;
            }
            color_maps[m.map] = (int32_t)b.add_color_map_rgba8(img.width, img.height, img.rgba.data());
        }
    }
    int mesh = (int)b.add_mesh(std::move(tris));
    // MeshTextureSet -> one MESH_SET texture over MESH textures
    std::vector<int32_t> entries;
    for (auto& n : names) {
        const ObjMaterial& m = materials[n];
        RtTexture t{}; t.kind = RT_TEX_MESH; t.color = out3(m.diffuse); t.color2 = out3(m.ambient); t.child0 = -1; t.child1 = -1;
        if (m.has_map) { auto it = color_maps.find(m.map); if (it != color_maps.end()) t.child0 = it->second; }   // else: "Unable to find color map" (builder.rs:62-65)
        entries.push_back((int32_t)b.add_texture(t));
    }
    RtTexture set{}; set.kind = RT_TEX_MESH_SET; set.child0 = (int32_t)b.tex_set_entries.size(); set.child1 = (int32_t)entries.size();
    b.tex_set_entries.insert(b.tex_set_entries.end(), entries.begin(), entries.end());
    int set_id = (int)b.add_texture(set);
    if (texture_set_out) *texture_set_out = set_id;
    return mesh;
}

// ---------------------------------------------------------------------------
// Scene factories
// ---------------------------------------------------------------------------
struct Params {
    std::map<std::string, float> values;
    float get(const std::string& k, float dflt) const { auto it = values.find(k); return it == values.end() ? dflt : it->second; }
};
Params parse_params(const char* s) {
    Params p; if (!s) return p;
    std::stringstream ss(s); std::string item;
    while (std::getline(ss, item, ';')) {
        size_t eq = item.find('=');
        if (eq == std::string::npos) continue;
        p.values[trimmed(item.substr(0, eq))] = strtof(item.substr(eq + 1).c_str(), nullptr);
    }
    return p;
}

struct Ctx { const RtCameraConfig& cam; std::string mesh_dir; Params params; };

// positioning helpers, samples.rs:19-49
Vec North(float f) { return scale(mk(1, 0, 0), f); }
Vec South(float f) { return scale(mk(-1, 0, 0), f); }
Vec East(float f) { return scale(mk(0, 0, 1), f); }
Vec West(float f) { return scale(mk(0, 0, -1), f); }
Vec Up(float f) { return scale(mk(0, 1, 0), f); }
Vec Down(float f) { return scale(mk(0, -1, 0), f); }
// position!(A, B, C) = A + (B + C): right-nested sum (samples.rs:45-49)
Vec pos(Vec a) { return a; }
Vec pos(Vec a, Vec b) { return add(a, b); }
Vec pos(Vec a, Vec b, Vec c) { return add(a, add(b, c)); }

int load_mesh(RtSceneBuilder& b, const Ctx& cx, const char* sub, int* tex_set) {
    return load_obj_mesh(b, cx.mesh_dir + "/" + sub, tex_set);
}

// samples.rs:70-106
void make_matte(RtSceneBuilder& s, SceneRng& rng, uint32_t& mat, uint32_t& tex) {
    float r = rng.random_f32() * rng.random_f32();
    float g = rng.random_f32() * rng.random_f32();
    float b = rng.random_f32() * rng.random_f32();
    mat = s.add_material(lambertian()); tex = s.add_texture(color_texture(mk(r, g, b)));
}
void make_metal(RtSceneBuilder& s, SceneRng& rng, uint32_t& mat, uint32_t& tex) {
    float r = 0.5f * (1.0f + rng.random_f32());
    float g = 0.5f * (1.0f + rng.random_f32());
    float b = 0.5f * (1.0f + rng.random_f32());
    float fuzz = 0.5f * rng.random_f32();
    mat = s.add_material(specular(1.0f, fuzz)); tex = s.add_texture(color_texture(mk(r, g, b)));
}
void make_glass(RtSceneBuilder& s, SceneRng& rng, uint32_t& mat, uint32_t& tex) {
    float r = 0.5f * (1.0f + rng.random_f32());
    float g = 0.5f * (1.0f + rng.random_f32());
    float b = 0.5f * (1.0f + rng.random_f32());
    mat = s.add_material(dielectric(1.0f, 0.0f, 1.5f)); tex = s.add_texture(color_texture(mk(r, g, b)));
}

// samples.rs:108-183 "Random Spheres" (the `weekend` scene)
int random_sphere_scene(RtSceneBuilder& s, const Ctx& cx) {
    Vec look_from = mk(13.0f, 2.0f, 3.0f), look_to = mk(0.0f, 0.0f, 0.0f);
    s.camera = make_camera(cx.cam, look_to, look_from);
    SceneRng rng((unsigned __int128)3178901564u);
    s.sky = RT_SKY_DAY;
    Vec lamp_direction = sub(kZero, mk(4.0f, 100.0f, 4.0f));
    s.add_light(directional_light(lamp_direction, 0.5f));
    // CheckerTexture::new(10, Color(0.4,0.5,0.4), Color(0.9,0.8,0.9)); the two
    // child textures are not scene textures in the reference (texture.rs:20-24)
    uint32_t odd = s.add_texture(color_texture(mk(0.4f, 0.5f, 0.4f)));
    uint32_t even = s.add_texture(color_texture(mk(0.9f, 0.8f, 0.9f)));
    RtTexture checker{}; checker.kind = RT_TEX_CHECKER; checker.size = 10.0f; checker.child0 = (int32_t)odd; checker.child1 = (int32_t)even;
    uint32_t world_tex = s.add_texture(checker);
    uint32_t world_mat = s.add_material(lambertian());
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(mk(0.0f, -1000.0f, 0.0f)));

    Vec lam_center = mk(-4.0f, 1.0f, 0.0f);
    uint32_t lam_tex = s.add_texture(color_texture(mk(0.8f, 0.2f, 0.1f)));
    uint32_t lam_mat = s.add_material(lambertian());
    add_entity(s, EntityB::of(sphere(1.0f, lam_mat, lam_tex)).translate(lam_center));

    Vec hollow_center = mk(0.0f, 1.0f, 0.0f);
    uint32_t hollow_tex = s.add_texture(color_texture(mk(0.95f, 0.95f, 0.95f)));
    uint32_t hollow_mat = s.add_material(dielectric(1.0f, 0.0f, 1.5f));
    add_entity(s, EntityB::of(sphere(1.0f, hollow_mat, hollow_tex)).translate(hollow_center));

    Vec metal_center = mk(4.0f, 1.0f, 0.0f);
    uint32_t metal_tex = s.add_texture(color_texture(mk(0.8f, 0.8f, 0.8f)));
    uint32_t metal_mat = s.add_material(specular(1.0f, 0.0f));
    add_entity(s, EntityB::of(sphere(1.0f, metal_mat, metal_tex)).translate(metal_center));

    Vec centers[3] = {lam_center, hollow_center, metal_center};
    for (int a = -11; a < 11; a++) {
        for (int bz = -11; bz < 11; bz++) {
            float cxp = (float)a + 0.9f * rng.random_f32();
            float czp = (float)bz + 0.9f * rng.random_f32();
            Vec center = mk(cxp, 0.2f, czp);
            bool too_close = false;
            for (const Vec& p : centers) if (norm(sub(center, p)) < 1.5f) { too_close = true; break; }
            if (too_close) continue;
            uint32_t mat, tex;
            float v = rng.random_f32();
            if (v < 0.8f) make_matte(s, rng, mat, tex);
            else if (v < 0.95f) make_metal(s, rng, mat, tex);
            else make_glass(s, rng, mat, tex);
            add_entity(s, EntityB::of(sphere(0.2f, mat, tex)).translate(center));
        }
    }
    return RT_OK;
}

void add_cardinal_markers(RtSceneBuilder& s) {                                       // samples.rs:185-196
    uint32_t mat = s.add_material(lambertian());
    uint32_t red = s.add_texture(color_texture(rgb(128, 0, 0)));
    uint32_t green = s.add_texture(color_texture(rgb(0, 128, 0)));
    uint32_t blue = s.add_texture(color_texture(rgb(0, 0, 128)));
    uint32_t white = s.add_texture(color_texture(rgb(255, 255, 255)));
    add_entity(s, EntityB::of(sphere(0.25f, mat, red)).translate(North(2.0f)));
    add_entity(s, EntityB::of(sphere(0.25f, mat, green)).translate(East(2.0f)));
    add_entity(s, EntityB::of(sphere(0.25f, mat, blue)).translate(West(2.0f)));
    add_entity(s, EntityB::of(sphere(0.25f, mat, white)).translate(South(2.0f)));
}
void add_coordinates_marker(RtSceneBuilder& s) {                                     // samples.rs:198-207
    uint32_t mat = s.add_material(lambertian());
    uint32_t red = s.add_texture(color_texture(rgb(128, 0, 0)));
    uint32_t green = s.add_texture(color_texture(rgb(0, 128, 0)));
    uint32_t blue = s.add_texture(color_texture(rgb(0, 0, 128)));
    add_entity(s, EntityB::of(sphere(0.05f, mat, red)).translate(mk(1, 0, 0)));
    add_entity(s, EntityB::of(sphere(0.05f, mat, green)).translate(mk(0, 1, 0)));
    add_entity(s, EntityB::of(sphere(0.05f, mat, blue)).translate(mk(0, 0, 1)));
}

int simple_scene(RtSceneBuilder& s, const Ctx& cx) {                                 // samples.rs:209-300
    Vec look_from = pos(South(6.0f), East(1.5f), Up(3.0f)), look_to = pos(Up(1.0f));
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    Vec lamp_pos = pos(Up(20.0f), North(4.0f));
    s.add_light(lamp_light(lamp_pos, sub(kZero, lamp_pos), 80.0f, 12.0f));
    Vec lamp_pos2 = pos(Up(10.0f), East(4.0f));
    s.add_light(lamp_light(lamp_pos2, sub(kZero, lamp_pos2), 80.0f, 20.0f));
    add_cardinal_markers(s);
    uint32_t world_mat = s.add_material(lambertian()), world_tex = s.add_texture(color_texture(rgb(200, 200, 200)));
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(Down(1000.0f)));
    uint32_t wall_mat = s.add_material(lambertian(1.0f)), wall_tex = s.add_texture(color_texture(rgb(200, 200, 200)));
    Vec wall_pos = North(4.5f), wall_facing = sub(wall_pos, kZero);
    add_entity(s, EntityB::of(plane_new(wall_facing, wall_mat, wall_tex)).translate(wall_pos));
    uint32_t plastic_mat = s.add_material(lambertian()), plastic_tex = s.add_texture(color_texture(rgb(226, 226, 226)));
    Vec plastic_pos = Up(1.0f);
    add_entity(s, EntityB::of(sphere(1.0f, plastic_mat, plastic_tex)).translate(plastic_pos));
    uint32_t glass_mat = s.add_material(dielectric()), glass_tex = s.add_texture(color_texture(rgb(130, 255, 140)));
    add_entity(s, EntityB::of(sphere(1.0f, glass_mat, glass_tex)).translate(pos(Up(1.0f), South(2.0f), East(2.0f))));
    uint32_t sg_mat = s.add_material(dielectric(0.98f, 0.01f, 1.5f)), sg_tex = s.add_texture(color_texture(rgb(66, 206, 245)));
    add_entity(s, EntityB::of(sphere(0.5f, sg_mat, sg_tex)).translate(lerp_v3(plastic_pos, lamp_pos2, 0.2f)));
    uint32_t metal_mat = s.add_material(specular(0.91f, 0.001f)), metal_tex = s.add_texture(color_texture(rgb(147, 154, 186)));
    add_entity(s, EntityB::of(sphere(1.0f, metal_mat, metal_tex)).translate(pos(Up(1.0f), North(2.0f), West(2.0f))).id(1));
    uint32_t sm_mat = s.add_material(specular(0.4f, 0.01f)), sm_tex = s.add_texture(color_texture(mk(0.8f, 0.1f, 0.1f)));
    Vec sm_start = pos(West(3.5f), North(1.0f)), sm_end = pos(West(2.5f), South(3.5f));
    for (int i = 0; i < 6; i++) {
        float t = (float)i / 6.0f;
        add_entity(s, EntityB::of(sphere(lerp_f32(0.5f, 0.05f, ease_out(t, 2.0f)), sm_mat, sm_tex)).translate(lerp_v3(sm_start, sm_end, ease_out(t, 2.0f))));
    }
    uint32_t sp_mat = s.add_material(dielectric()), sp_tex = s.add_texture(color_texture(mk(0.1f, 0.9f, 0.1f)));
    Vec sp_start = pos(West(2.5f), South(0.5f)), sp_end = pos(West(0.5f), South(2.5f));
    for (int i = 0; i < 12; i++) {
        float t = (float)i / 12.0f;
        Vec p = mk(lerp_f32(sp_start.x, sp_end.x, ease_out(t, 2.0f)), 0.0f, lerp_f32(sp_start.z, sp_end.z, ease_in(t, 2.0f)));
        add_entity(s, EntityB::of(sphere(lerp_f32(0.10f, 0.02f, ease_in_out(t, 2.0f)), sp_mat, sp_tex)).translate(p));
    }
    return RT_OK;
}

int planes_scene(RtSceneBuilder& s, const Ctx& cx) {                                 // samples.rs:302-332
    Vec look_from = pos(South(6.0f), East(1.5f), Up(3.0f)), look_to = pos(Up(1.0f));
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_DAY;
    Vec lamp_pos = pos(Up(6.0f), East(5.0f));
    s.add_light(lamp_light(lamp_pos, sub(Up(3.0f), lamp_pos), 80.0f, 20.0f));
    add_cardinal_markers(s);
    uint32_t world_mat = s.add_material(lambertian(0.01f)), world_tex = s.add_texture(color_texture(rgb(255, 255, 255)));
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(Down(1000.0f)));
    uint32_t plane_mat = s.add_material(specular(0.8f, 0.02f)), plane_tex = s.add_texture(color_texture(rgb(240, 240, 240)));
    Vec plane_pos = West(1.0f);
    add_entity(s, EntityB::of(plane_new(sub(kZero, plane_pos), plane_mat, plane_tex)).translate(plane_pos));
    return RT_OK;
}

int hall_of_mirrors(RtSceneBuilder& s, const Ctx& cx) {                              // samples.rs:334-376
    Vec look_from = pos(Up(1.0f), South(2.0f), East(1.5f)), look_to = pos(Up(0.5f));
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_DAY;
    Vec lamp_pos = Up(10.0f);
    s.add_light(lamp_light(lamp_pos, sub(kZero, lamp_pos), 80.0f, 20.0f));
    add_cardinal_markers(s);
    add_coordinates_marker(s);
    uint32_t world_mat = s.add_material(lambertian(0.01f)), world_tex = s.add_texture(color_texture(rgb(255, 255, 255)));
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(Down(1000.0f)));
    uint32_t plane_mat = s.add_material(specular(0.98f, 0.01f)), plane_tex = s.add_texture(color_texture(kOne));
    Vec cardinals[4] = {North(3.0f), South(3.0f), East(3.0f), West(3.0f)};
    for (Vec o : cardinals) {
        RtPlane p = plane_new(sub(kZero, o), plane_mat, plane_tex); p.has_radius = 1; p.radius = 30.0f;
        add_entity(s, EntityB::of(p).translate(o));
    }
    return RT_OK;
}

RtMeshTri tri_abc(Vec a, Vec b, Vec c) { RtMeshTri t{}; t.a = out3(a); t.b = out3(b); t.c = out3(c); t.tex_key = -1; return t; }

int triangle_world(RtSceneBuilder& s, const Ctx& cx) {                               // samples.rs:378-436
    Vec look_from = pos(Up(5.0f), South(6.0f), East(1.5f)), look_to = Up(0.0f);
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    Vec lamp_pos = pos(Up(20.0f), North(4.0f));
    s.add_light(lamp_light(lamp_pos, sub(kZero, lamp_pos), 80.0f, 12.0f));
    Vec lamp_pos2 = pos(Up(10.0f), East(4.0f));
    s.add_light(lamp_light(lamp_pos2, sub(kZero, lamp_pos2), 80.0f, 20.0f));
    add_cardinal_markers(s);
    uint32_t world_mat = s.add_material(lambertian()), world_tex = s.add_texture(color_texture(rgb(200, 200, 200)));
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(Down(1000.0f)));
    uint32_t m1 = s.add_material(lambertian(0.0f)), t1 = s.add_texture(color_texture(rgb(200, 100, 80)));
    uint32_t mesh1 = s.add_mesh({tri_abc(pos(Up(0.3f), South(1.0f)), pos(Up(0.6f), East(1.0f)), pos(Up(0.8f), West(1.0f)))});
    add_entity(s, EntityB::of(mesh_object(mesh1, m1, t1)).translate(kZero));
    uint32_t m2 = s.add_material(lambertian(0.0f)), t2 = s.add_texture(color_texture(rgb(100, 100, 200)));
    uint32_t mesh2 = s.add_mesh({tri_abc(pos(Up(0.4f), North(1.0f)), pos(Up(0.8f), South(1.0f)), pos(Up(0.6f), East(1.0f)))});
    add_entity(s, EntityB::of(mesh_object(mesh2, m2, t2)).translate(Up(1.0f)));
    return RT_OK;
}

int mesh_demo(RtSceneBuilder& s, const Ctx& cx) {                                    // samples.rs:438-503
    Vec look_from = pos(Up(1.5f), South(4.0f), East(4.0f)), look_to = Up(1.0f);
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    Vec lamp_pos = pos(Up(5.0f), East(4.0f));
    s.add_light(lamp_light(lamp_pos, sub(kZero, lamp_pos), 80.0f, 20.0f));
    Vec lamp_pos2 = pos(Up(3.0f), West(6.0f), North(1.5f));
    s.add_light(lamp_light(lamp_pos2, sub(kZero, lamp_pos2), 80.0f, 20.0f));
    uint32_t world_mat = s.add_material(lambertian()), world_tex = s.add_texture(color_texture(rgb(200, 200, 200)));
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(Down(1000.0f)).id(0));
    uint32_t cube_mat = s.add_material(lambertian(0.0f)), cube_tex = s.add_texture(color_texture(rgb(36, 193, 89)));
    int cube = load_mesh(s, cx, "simple/cube.obj", nullptr); if (cube < 0) return cube;
    add_entity(s, EntityB::of(mesh_object((uint32_t)cube, cube_mat, cube_tex)).translate(pos(South(1.5f), West(1.5f))).rotate(kY, kPi / 4.0f).id(1));
    uint32_t thing_mat = s.add_material(specular(0.8f, 0.02f)), thing_tex = s.add_texture(color_texture(rgb(89, 172, 255)));
    int thing = load_mesh(s, cx, "simple/thing.obj", nullptr); if (thing < 0) return thing;
    add_entity(s, EntityB::of(mesh_object((uint32_t)thing, thing_mat, thing_tex)).translate(pos(North(1.5f), East(1.5f))).id(2));
    uint32_t suz_mat = s.add_material(dielectric(0.0f, 0.2f, 0.8f)), suz_tex = s.add_texture(color_texture(rgb(255, 137, 58)));
    int suz = load_mesh(s, cx, "simple/suzanne.obj", nullptr); if (suz < 0) return suz;
    add_entity(s, EntityB::of(mesh_object((uint32_t)suz, suz_mat, suz_tex)).translate(kZero).id(3));
    return RT_OK;
}

int mega_cube(RtSceneBuilder& s, const Ctx& cx) {                                    // samples.rs:597-629
    float d = 300.0f;
    Vec look_from = pos(Up(d), South(d), East(d)), look_to = kZero;
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    s.add_light(directional_light(sub(look_to, look_from), 0.9f));
    uint32_t mat = s.add_material(lambertian()), tex = s.add_texture(color_texture(mk(1.0f, 0.41f, 0.70f)));
    for (int x = -100; x <= 100; x += 25) for (int y = -100; y <= 100; y += 25) for (int z = -100; z <= 100; z += 25)
        add_entity(s, EntityB::of(sphere(10.0f, mat, tex)).translate(mk((float)x, (float)y, (float)z)));
    return RT_OK;
}

int fleet(RtSceneBuilder& s, const Ctx& cx) {                                        // samples.rs:631-665
    float d = 100.0f;
    Vec look_from = pos(Up(d), North(d), East(d)), look_to = kZero;
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    s.add_light(point_light(look_from, 2000.0f));
    int tex_set; int mesh = load_mesh(s, cx, "Interceptor-T/Heavyinterceptor.obj", &tex_set); if (mesh < 0) return mesh;
    uint32_t mat = s.add_material(lambertian());
    EntityB proto = EntityB::of(mesh_object((uint32_t)mesh, mat, (uint32_t)tex_set)).rotate(kZ, deg_to_rad(90.0f));
    for (int x = -600; x <= 0; x += 60) for (int y = -600; y <= 0; y += 60) for (int z = -600; z <= 0; z += 60) {
        EntityB e = proto; add_entity(s, e.translate(mk((float)x, (float)y, (float)z)));
    }
    return RT_OK;
}

int capsule(RtSceneBuilder& s, const Ctx& cx) {                                      // samples.rs:556-595
    Vec look_from = pos(Up(10.0f), South(10.0f)), look_to = Up(4.0f);
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    Vec lamp_pos = pos(Up(20.0f), East(20.0f));
    s.add_light(directional_light(sub(kZero, lamp_pos), 0.5f));
    uint32_t world_mat = s.add_material(lambertian()), world_tex = s.add_texture(color_texture(rgb(200, 200, 200)));
    add_entity(s, EntityB::of(sphere(1000.0f, world_mat, world_tex)).translate(Down(1000.0f)).id(0));
    int tex_set; int mesh = load_mesh(s, cx, "capsule/capsule.obj", &tex_set); if (mesh < 0) return mesh;
    uint32_t mat = s.add_material(lambertian());
    add_entity(s, EntityB::of(mesh_object((uint32_t)mesh, mat, (uint32_t)tex_set)).translate(Up(4.0f)));
    return RT_OK;
}

int point_cloud(RtSceneBuilder& s, const Ctx& cx) {                                  // scene_point_cloud.rs:31-78
    float dist = cx.params.get("Camera Distance", 100.0f);
    Vec look_from = add(kZero, scale(kOne, dist)), look_to = kZero;
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    Vec lamp_pos = rotated(look_from, kY, deg_to_rad(15.0f));
    s.add_light(lamp_light(look_from, sub(look_to, lamp_pos), cx.params.get("Spotlight Intensity", 1300.0f), cx.params.get("Spotlight Beam Angle", 60.0f)));
    uint32_t mat = s.add_material(lambertian());
    float radius = cx.params.get("Cloud Point Diameter", 0.5f);
    SceneRng rng((unsigned __int128)432789012409ull);
    float x_len = cx.params.get("Cloud Width", 100.0f), z_len = cx.params.get("Cloud Height", 100.0f), y_len = cx.params.get("Cloud Depth", 100.0f);
    size_t count = (size_t)cx.params.get("Cloud Point Count", 100000.0f);
    for (size_t i = 0; i < count; i++) {
        float a = rng.random_f32(), b = rng.random_f32(), c = rng.random_f32();
        float x = (a * x_len) - (x_len / 2.0f), y = (b * z_len) - (z_len / 2.0f), z = (c * y_len) - (y_len / 2.0f);
        uint32_t tex = s.add_texture(color_texture(mk(a, b, c)));
        add_entity(s, EntityB::of(sphere(radius, mat, tex)).translate(mk(x, y, z)));
    }
    return RT_OK;
}

int dreadnaught(RtSceneBuilder& s, const Ctx& cx) {                                  // scene_dreadnaught.rs:32-68
    float dist = cx.params.get("Camera Distance", 800.0f);
    Vec look_to = kZero;
    Vec look_from = add(add(add(kZero, scale(kY, dist)), scale(kZ, dist)), scale(kX, dist));
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    s.add_light(point_light(add(look_from, scale(kY, 1000.0f)), cx.params.get("Global Light Intensity", 20.0f)));
    Vec spot = add(look_from, divide(scale(kY, dist), 10.0f));
    s.add_light(lamp_light(spot, sub(look_to, spot), cx.params.get("Spotlight Intensity", 80.0f), cx.params.get("Spotlight Beam Angle", 10.0f)));
    uint32_t mat = s.add_material(lambertian());
    int tex_set; int mesh = load_mesh(s, cx, "Dreadnaught/Dreadnaught.obj", &tex_set); if (mesh < 0) return mesh;
    add_entity(s, EntityB::of(mesh_object((uint32_t)mesh, mat, (uint32_t)tex_set))
                      .rotate(kZ, deg_to_rad(cx.params.get("Dreadnaught Roll", 0.0f)))
                      .rotate(kX, deg_to_rad(cx.params.get("Dreadnaught Pitch", 0.0f)))
                      .rotate(kY, deg_to_rad(cx.params.get("Dreadnaught Yaw", 0.0f))));
    return RT_OK;
}

int dootdoot(RtSceneBuilder& s, const Ctx& cx) {                                     // scene_dootdoot.rs:36-86
    float dist = cx.params.get("Camera Distance", 70.0f);
    Vec look_to = kZero;
    Vec look_from = add(look_to, rotated(scale(kOne, dist), kY, kPi));
    s.camera = make_camera(cx.cam, look_to, look_from);
    s.sky = cx.params.get("Sky", 0.0f) == 0.0f ? RT_SKY_BLACK : RT_SKY_DAY;
    s.add_light(point_light(add(look_from, scale(kY, 100.0f)), cx.params.get("Global Light Intensity", 1100.0f)));
    uint32_t mat = s.add_material(lambertian());
    int tex_set; int mesh = load_mesh(s, cx, "skeleton/SKELETON.obj", &tex_set); if (mesh < 0) return mesh;
    EntityB proto = EntityB::of(mesh_object((uint32_t)mesh, mat, (uint32_t)tex_set))
                        .rotate(kZ, deg_to_rad(cx.params.get("Doot Doot Roll", 0.0f)))
                        .rotate(kX, deg_to_rad(cx.params.get("Doot Doot Pitch", 0.0f)))
                        .rotate(kY, deg_to_rad(cx.params.get("Doot Doot Yaw", 0.0f)));
    float spacing = cx.params.get("Spacing", 30.0f);
    size_t count = (size_t)cx.params.get("Count", 10.0f);
    float half_w = (float)count * spacing / 2.0f;
    bool spin = cx.params.get("Spin em round", 0.0f) != 0.0f;
    SceneRng rng((unsigned __int128)3178901564u);
    for (size_t xi = 0; xi < count; xi++) for (size_t zi = 0; zi < count; zi++) {
        float x = ((float)xi * spacing) - half_w, z = ((float)zi * spacing) - half_w;
        float rot = spin ? rng.random_range(0.0f, 2.0f) * kPi : 0.0f;
        EntityB e = proto; add_entity(s, e.rotate(kY, rot).translate(mk(x, 0.0f, z)));
    }
    return RT_OK;
}

int spaceships(RtSceneBuilder& s, const Ctx& cx) {                                   // samples.rs:505-554
    Vec look_from = pos(Up(350.0f), South(400.0f), East(700.0f)), look_to = pos(Down(50.0f), West(350.0f));
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    Vec lamp_pos = pos(Up(1000.0f), East(1000.0f));
    s.add_light(directional_light(sub(kZero, lamp_pos), 0.9f));
    float world_radius = 10000.0f;
    uint32_t world_mat = s.add_material(lambertian()), world_tex = s.add_texture(color_texture(rgb(200, 200, 200)));
    add_entity(s, EntityB::of(sphere(world_radius, world_mat, world_tex)).translate(pos(Down(world_radius), Down(1000.0f))).id(0));
    int dest_set; int dest = load_mesh(s, cx, "Destroyer-K/Standarddestroyer.obj", &dest_set); if (dest < 0) return dest;
    uint32_t dest_mat = s.add_material(lambertian());
    EntityB dest_mesh = EntityB::of(mesh_object((uint32_t)dest, dest_mat, (uint32_t)dest_set)).rotate(kZ, deg_to_rad(90.0f));
    int int_set; int intc = load_mesh(s, cx, "Interceptor-T/Heavyinterceptor.obj", &int_set); if (intc < 0) return intc;
    uint32_t int_mat = s.add_material(lambertian());
    EntityB int_mesh = EntityB::of(mesh_object((uint32_t)intc, int_mat, (uint32_t)int_set)).rotate(kZ, deg_to_rad(90.0f));
    Vec int_origin = add(look_to, pos(Up(200.0f), East(300.0f), South(30.0f)));
    { EntityB e = int_mesh; add_entity(s, e.translate(add(int_origin, pos(East(15.0f))))); }
    { EntityB e = int_mesh; add_entity(s, e.translate(add(int_origin, pos(East(2.0f), North(80.0f), Down(30.0f))))); }
    { EntityB e = int_mesh; add_entity(s, e.translate(add(int_origin, pos(East(1.0f), South(65.0f), Down(15.0f))))); }
    add_entity(s, dest_mesh.translate(look_to));
    return RT_OK;
}

// scene_interceptor_spin.rs:32-77.  The reference takes the angle from the wall clock;
// "Time Ms" (not a control of the reference) pins it for reproducible renders.
int interceptor_spin(RtSceneBuilder& s, const Ctx& cx) {
    float ms_per_rotation = cx.params.get("Seconds Per Rotation", 10.0f) * 1000.0f;
    unsigned __int128 ms_since_epoch;
    if (cx.params.values.count("Time Ms")) ms_since_epoch = (unsigned __int128)cx.params.get("Time Ms", 0.0f);
    else ms_since_epoch = (unsigned __int128)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    unsigned __int128 period = (unsigned __int128)ms_per_rotation;      // `as u128` truncates (saturating)
    if (period == 0) { g_error = "Seconds Per Rotation: attempt to calculate the remainder with a divisor of zero"; return RT_E_INVALID; }
    float rot_rads = ((float)(ms_since_epoch % period) / ms_per_rotation) * kPi * 2.0f;
    float dist = cx.params.get("Camera Distance", 100.0f);
    Vec look_to = kZero, look_from = scale(kOne, dist);
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_BLACK;
    s.add_light(point_light(add(look_from, scale(kY, 1000.0f)), cx.params.get("Global Light Intensity", 20.0f)));
    Vec spot = add(look_from, divide(scale(kY, dist), 10.0f));
    s.add_light(lamp_light(spot, sub(look_to, spot), cx.params.get("Spotlight Intensity", 1200.0f), cx.params.get("Spotlight Beam Angle", 60.0f)));
    int tex_set; int mesh = load_mesh(s, cx, "Interceptor-T/Heavyinterceptor.obj", &tex_set); if (mesh < 0) return mesh;
    uint32_t mat = s.add_material(lambertian());
    add_entity(s, EntityB::of(mesh_object((uint32_t)mesh, mat, (uint32_t)tex_set)).rotate(kY, rot_rads).rotate(kZ, deg_to_rad(90.0f)));
    return RT_OK;
}

// CheckerTexture::new(size, ColorTexture(odd), ColorTexture(even)), texture.rs:20-30
RtTexture checker_texture(RtSceneBuilder& s, float size, Vec odd, Vec even) {
    RtTexture t{}; t.kind = RT_TEX_CHECKER; t.size = size;
    t.child0 = (int32_t)s.add_texture(color_texture(odd)); t.child1 = (int32_t)s.add_texture(color_texture(even));
    return t;
}

// scene_entity_transform_test.rs:88-118
float hue_to_rgb(float p, float q, float t) {
    if (t < 0.0f) t += 1.0f;
    if (t > 1.0f) t -= 1.0f;
    if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
    if (t < 1.0f / 2.0f) return q;
    if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
    return p;
}
Vec hsl_to_rgb(float h, float sat, float l) {
    if (sat == 0.0f) return mk(1.0f, 1.0f, 1.0f);
    float q = l < 0.5f ? l * (1.0f + sat) : l + sat - l * sat;
    float p = 2.0f * l - q;
    return mk(hue_to_rgb(p, q, h + 1.0f / 3.0f), hue_to_rgb(p, q, h), hue_to_rgb(p, q, h - 1.0f / 3.0f));
}

int entity_transform_test(RtSceneBuilder& s, const Ctx& cx) {                        // scene_entity_transform_test.rs:34-85
    Vec look_from = add(add(kZero, scale(mk(0, 0, -1), 100.0f)), scale(kY, 100.0f)), look_to = kZero;
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_DAY;
    uint32_t lamb = s.add_material(lambertian());
    uint32_t checker = s.add_texture(checker_texture(s, cx.params.get("Plane Checker Scale", 1.0f), mk(1.0f, 0.5f, 0.5f), mk(0.5f, 1.0f, 0.5f)));
    Vec lamp_pos = scale(kY, cx.params.get("Lamp Height", 10.0f));
    s.add_light(lamp_light(lamp_pos, sub(look_to, lamp_pos), cx.params.get("Lamp Intensity", 400.0f), cx.params.get("Lamp Angle", 120.0f)));
    add_entity(s, EntityB::of(plane_new(kY, lamb, checker)).translate(scale(mk(0, -1, 0), 0.5f)).translate(scale(kX, 0.5f))
                      .rotate(kY, deg_to_rad(cx.params.get("Plane Rotation Deg", 0.0f))).id(1));
    float additional_rot = cx.params.get("Swirl Rotation Deg", 0.0f);
    SceneRng rng((unsigned __int128)138219031u);
    for (int i = 0; i < 1000; i++) {
        float dist = rng.random_f32(), angle = rng.random_f32();
        Vec translation = rotated(scale(scale(kX, dist), 50.0f), kY, (angle * 2.0f * kPi) + deg_to_rad(additional_rot));
        uint32_t tex = s.add_texture(color_texture(hsl_to_rgb(angle, dist, 0.6f)));
        add_entity(s, EntityB::of(sphere(1.0f, lamb, tex)).translate(translation));
    }
    return RT_OK;
}

int uv_test(RtSceneBuilder& s, const Ctx& cx) {                                      // scene_uv_test.rs:33-106
    Vec look_from = add(add(kZero, scale(mk(0, 0, -1), 2.0f)), scale(kY, 2.0f)), look_to = kZero;
    s.camera = make_camera(cx.cam, look_to, look_from); s.sky = RT_SKY_DAY;
    uint32_t lamb = s.add_material(lambertian());
    Vec lamp_pos = add(look_from, scale(kY, 5.0f));
    s.add_light(lamp_light(lamp_pos, sub(look_to, lamp_pos), 80.0f, 60.0f));
    uint32_t plane_tex = s.add_texture(checker_texture(s, cx.params.get("Plane Checker Scale", 1.0f), mk(1.0f, 0.5f, 0.5f), mk(0.5f, 1.0f, 0.5f)));
    Vec plane_origin = add(add(add(look_to, scale(kY, cx.params.get("Plane Offset Y", -0.5f))), scale(kX, cx.params.get("Plane Offset X", 0.0f))),
                           scale(kZ, cx.params.get("Plane Offset Z", 0.0f)));
    add_entity(s, EntityB::of(plane_new(kY, lamb, plane_tex)).translate(plane_origin).rotate(kY, deg_to_rad(cx.params.get("Plane Rotation Deg", 0.0f))).id(1));
    // load_color_map(mesh_path!("simple/test.bmp"))?: no skipping here, a file that does not decode fails the scene
    rtimg::Image img; std::string err;
    if (!rtimg::decode_file(cx.mesh_dir + "/simple/test.bmp", img, err)) { g_error = err; return RT_E_IO; }
    uint32_t mesh_tex = s.add_color_map_rgba8(img.width, img.height, img.rgba.data());
    int mesh = load_mesh(s, cx, "simple/plane.obj", nullptr); if (mesh < 0) return mesh;
    float ry = cx.params.get("Mesh Rotation Y Deg", 0.0f), rx = cx.params.get("Mesh Rotation X Deg", -40.0f);
    add_entity(s, EntityB::of(mesh_object((uint32_t)mesh, lamb, mesh_tex)).translate(add(look_to, scale(kY, 0.5f)))
                      .rotate(kY, deg_to_rad(180.0f)).rotate(kY, deg_to_rad(ry)).rotate(kX, deg_to_rad(rx)).id(2));
    RtTexture uvt{}; uvt.kind = RT_TEX_UV_TEST; uvt.child0 = uvt.child1 = -1;
    uint32_t uv_tex = s.add_texture(uvt);
    add_entity(s, EntityB::of(mesh_object((uint32_t)mesh, lamb, uv_tex)).translate(add(add(add(look_to, scale(kY, 0.5f)), scale(kX, 1.0f)), scale(mk(0, 0, -1), 0.5f)))
                      .rotate(kY, deg_to_rad(180.0f + 45.0f)).rotate(kY, deg_to_rad(ry)).rotate(kX, deg_to_rad(rx)).id(3));
    add_entity(s, EntityB::of(mesh_object((uint32_t)mesh, lamb, uv_tex)).translate(add(add(add(look_to, scale(kY, 0.5f)), scale(mk(-1, 0, 0), 1.0f)), scale(mk(0, 0, -1), 0.5f)))
                      .rotate(kY, deg_to_rad(180.0f - 45.0f)).rotate(kY, deg_to_rad(ry)).rotate(kX, deg_to_rad(rx)).id(4));
    return RT_OK;
}

typedef int (*Factory)(RtSceneBuilder&, const Ctx&);
struct Named { const char* name; Factory fn; };
// lib.rs:18-37, same order and names
const Named kFactories[] = {
    {"Random Spheres", random_sphere_scene}, {"Simple", simple_scene}, {"Planes", planes_scene},
    {"Mirrors", hall_of_mirrors}, {"Triangles", triangle_world}, {"Mesh", mesh_demo}, {"Capsule", capsule},
    {"Point Cloud", point_cloud}, {"Mega Cube", mega_cube}, {"Spaceships", spaceships}, {"Fleet", fleet},
    {"Interceptor Spin", interceptor_spin}, {"Dreadnaught", dreadnaught}, {"Doot Doot", dootdoot},
    {"Entity Transforms Test", entity_transform_test}, {"UV Test", uv_test},
};
const int kFactoryCount = (int)(sizeof(kFactories) / sizeof(kFactories[0]));

}  // namespace

extern "C" {

const char* rt_samples_last_error(void) { return g_error.c_str(); }

int rt_make_camera(const RtCameraConfig* cfg, const float* look_to, const float* look_from, RtCamera* out) {
    if (!cfg || !look_to || !look_from || !out) { g_error = "rt_make_camera: null argument"; return RT_E_INVALID; }
    *out = make_camera(*cfg, in3(look_to), in3(look_from));
    return RT_OK;
}
int rt_camera_new(const float* look_from, const float* look_at, float v_fov, float aspect, float lens_radius, float focus_dist, RtCamera* out) {
    if (!look_from || !look_at || !out) { g_error = "rt_camera_new: null argument"; return RT_E_INVALID; }
    *out = camera_new(in3(look_from), in3(look_at), v_fov, aspect, lens_radius, focus_dist);
    return RT_OK;
}

RtSceneBuilder* rt_builder_new(const RtCamera* camera, uint32_t sky) {
    if (!camera) { g_error = "rt_builder_new: null camera"; return nullptr; }
    RtSceneBuilder* b = new RtSceneBuilder(); b->camera = *camera; b->sky = sky; return b;
}
void rt_builder_destroy(RtSceneBuilder* b) { delete b; }
int rt_builder_add_material(RtSceneBuilder* b, const RtMaterial* m) { if (!b || !m) return RT_E_INVALID; return (int)b->add_material(*m); }
int rt_builder_add_texture(RtSceneBuilder* b, const RtTexture* t) { if (!b || !t) return RT_E_INVALID; return (int)b->add_texture(*t); }
int rt_builder_add_color_map(RtSceneBuilder* b, uint32_t w, uint32_t h, const float* rgbp) {
    if (!b || !rgbp || w == 0 || h == 0) return RT_E_INVALID;
    RtTexture t{}; t.kind = RT_TEX_COLOR_MAP; t.width = w; t.height = h; t.texel_first = b->texels.size(); t.child0 = t.child1 = -1;
    for (uint64_t i = 0; i < (uint64_t)w * h; i++) b->texels.push_back(RtV3{rgbp[3 * i], rgbp[3 * i + 1], rgbp[3 * i + 2]});
    return (int)b->add_texture(t);
}
int rt_builder_load_color_map(RtSceneBuilder* b, const char* path) {
    if (!b || !path) return RT_E_INVALID;
    rtimg::Image img; std::string err;
    if (!rtimg::decode_file(path, img, err)) { g_error = err; return RT_E_IO; }
    return (int)b->add_color_map_rgba8(img.width, img.height, img.rgba.data());
}
int rt_image_decode(const char* path, uint32_t* width, uint32_t* height, uint8_t* rgba_out, uint64_t capacity) {
    if (!path || !width || !height) return RT_E_INVALID;
    rtimg::Image img; std::string err;
    if (!rtimg::decode_file(path, img, err)) { g_error = err; return RT_E_IO; }
    *width = img.width; *height = img.height;
    if (rgba_out) {
        if (capacity < img.rgba.size()) { g_error = "rt_image_decode: capacity too small"; return RT_E_INVALID; }
        memcpy(rgba_out, img.rgba.data(), img.rgba.size());
    }
    return RT_OK;
}
const char* rt_samples_last_warning(void) { return g_warning.c_str(); }
int rt_builder_add_light(RtSceneBuilder* b, const RtLight* l) { if (!b || !l) return RT_E_INVALID; b->add_light(*l); return (int)b->lights.size() - 1; }
int rt_builder_add_mesh(RtSceneBuilder* b, const RtMeshTri* tris, uint64_t count) {
    if (!b || !tris || count == 0) return RT_E_INVALID;
    return (int)b->add_mesh(std::vector<RtMeshTri>(tris, tris + count));
}
int rt_builder_load_obj(RtSceneBuilder* b, const char* path, int* tex_set) {
    if (!b || !path) return RT_E_INVALID;
    return load_obj_mesh(*b, path, tex_set);
}
int rt_builder_add_entity(RtSceneBuilder* b, uint32_t kind, const RtSphere* sp, const RtPlane* pl, const RtMeshObject* mo, int32_t id,
                          const RtV3* tr, uint32_t ntr, const RtRotation* ro, uint32_t nro) {
    if (!b) return RT_E_INVALID;
    EntityB e;
    if (kind == RT_HITABLE_SPHERE && sp) e = EntityB::of(*sp);
    else if (kind == RT_HITABLE_PLANE && pl) e = EntityB::of(*pl);
    else if (kind == RT_HITABLE_MESH && mo) e = EntityB::of(*mo);
    else { g_error = "rt_builder_add_entity: hitable missing"; return RT_E_INVALID; }
    e.id_ = id;
    for (uint32_t i = 0; i < ntr; i++) e.tr.push_back(tr[i]);
    for (uint32_t i = 0; i < nro; i++) e.ro.push_back(ro[i]);
    return add_entity(*b, e);
}
int rt_plane_new(const float* normal, uint32_t mat, uint32_t tex, int has_radius, float radius, RtPlane* out) {
    if (!normal || !out) return RT_E_INVALID;
    *out = plane_new(in3(normal), mat, tex);
    out->has_radius = has_radius ? 1 : 0; out->radius = radius;
    return RT_OK;
}
const RtSceneDesc* rt_builder_desc(RtSceneBuilder* b) { return b ? b->finish() : nullptr; }

int rt_samples_count(void) { return kFactoryCount; }
const char* rt_samples_name(int i) { return (i >= 0 && i < kFactoryCount) ? kFactories[i].name : nullptr; }

RtSceneBuilder* rt_samples_create(const char* name, const RtCameraConfig* camera, const char* mesh_dir, const char* params) {
    if (!name || !camera) { g_error = "rt_samples_create: null argument"; return nullptr; }
    for (int i = 0; i < kFactoryCount; i++) {
        if (strcmp(kFactories[i].name, name) != 0) continue;
        RtSceneBuilder* b = new RtSceneBuilder();
        Ctx cx{*camera, mesh_dir ? mesh_dir : "", parse_params(params)};
        int rc = kFactories[i].fn(*b, cx);
        if (rc < 0) { delete b; return nullptr; }
        b->finish();
        return b;
    }
    g_error = std::string("unknown sample scene: ") + name;
    return nullptr;
}

}  // extern "C"
