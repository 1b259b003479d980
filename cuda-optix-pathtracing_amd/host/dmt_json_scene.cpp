// dmt_json_scene.cpp -- JSON scene front-end: the reference's scene description, flattened to the
// megakernel's upload arrays.
//
// Schema, defaults, clamps and rejection rules follow the reference's parser, src/core/private/core-parser.cpp:
//   top level :1341-1455 (all nine keys required), camera :739-812, film :256-304, textures :306-396,
//   materials :398-737, transforms :814-905 (M = T * R * S), lights :907-1078, envlight :1080-1125,
//   objects :1127-1260, world :1262-1335 (nested transform names; "instances" / "lights" arrays; members are
//   visited in KEY order because nlohmann objects are sorted maps).
// Geometry of the primitives: src/core/private/core-trianglemesh.cpp:101-188 (unitCube, unitPlane).
// Light construction: src/core/private/core-light.cpp:14-80 (position = translation column, spot axis =
// normalize(M * (0,1,0)), cos clamps), defaults src/core/public/core-light.h:134-140 (radius 1e-3).
// Image bytes -> float: src/core/private/core-parser.cpp:156-167 (v / 255, no gamma).
// Matrices are glm's (column-major, column vectors; the reference calls glm::translate / rotate / scale on
// identity, src/core/private/cudautils/cudautils-transform.cu:28-85; glm is fetched by cmake, not vendored).
//
// What the reference feeds with this is its CPU renderer's scene graph.  Here the result is the megakernel's
// flat scene, so three mappings are this build's own and are documented in DESIGN.md: materials become ONE packed
// BSDF record each ("oren-nayar-dielectric" -> Oren-Nayar; else metallic <= 0 -> GGX dielectric, >= 1 -> GGX
// conductor, in between both, core-material.cpp:272-286; alpha_y = roughness, alpha_x = anisotropy * roughness, :262-263), instances are
// flattened (every vertex transformed on the host), and image textures (albedo of Oren-Nayar materials, roughness, normal maps) are sampled per hit by the *_tex kernels
// (dmt_upload_textures); a fractional or textured 'metallic' makes a record PAIR blended per hit (packMaterial).  FBX objects go through this build's own binary reader (dmt_fbx.cpp).
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <set>
#include <sstream>

#include "dmt_json.hpp"
#include "dmt_scene.hpp"

namespace dmt_host {
namespace {

using json::Value;

// ---- glm-style 4x4, column-major: m[col * 4 + row] ---------------------------------------------------
struct Mat4 {
  float m[16];
};
Mat4 identity() {
  Mat4 r{};
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
  return r;
}
Mat4 mul(Mat4 const& a, Mat4 const& b) {  // glm operator*: column c of the result = sum_k a[k] * b[c][k], k ascending
  Mat4 r{};
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row)
      r.m[c * 4 + row] = a.m[0 * 4 + row] * b.m[c * 4 + 0] + a.m[1 * 4 + row] * b.m[c * 4 + 1] +
                         a.m[2 * 4 + row] * b.m[c * 4 + 2] + a.m[3 * 4 + row] * b.m[c * 4 + 3];
  return r;
}
Vec3 normalized(Vec3 v) {  // glm::normalize: v * inversesqrt(dot(v, v))
  float const inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  return Vec3{v.x * inv, v.y * inv, v.z * inv};
}
Mat4 translate(Vec3 v) {
  Mat4 r = identity();
  r.m[12] = v.x, r.m[13] = v.y, r.m[14] = v.z;
  return r;
}
Mat4 scale(Vec3 s) {
  Mat4 r = identity();
  r.m[0] = s.x, r.m[5] = s.y, r.m[10] = s.z;
  return r;
}
Mat4 rotate(float degrees, Vec3 axisIn) {  // glm::rotate(identity, radians(degrees), axis)
  float const a = degrees * 0.01745329251994329576923690768489f;
  float const c = std::cos(a), s = std::sin(a);
  Vec3 const axis = normalized(axisIn);
  Vec3 const t{(1.f - c) * axis.x, (1.f - c) * axis.y, (1.f - c) * axis.z};
  Mat4 r = identity();
  r.m[0] = c + t.x * axis.x, r.m[1] = t.x * axis.y + s * axis.z, r.m[2] = t.x * axis.z - s * axis.y;
  r.m[4] = t.y * axis.x - s * axis.z, r.m[5] = c + t.y * axis.y, r.m[6] = t.y * axis.z + s * axis.x;
  r.m[8] = t.z * axis.x + s * axis.y, r.m[9] = t.z * axis.y - s * axis.x, r.m[10] = c + t.z * axis.z;
  return r;
}
Vec3 xformPoint(Mat4 const& M, Vec3 p) {  // M * (p, 1)
  return Vec3{M.m[0] * p.x + M.m[4] * p.y + M.m[8] * p.z + M.m[12], M.m[1] * p.x + M.m[5] * p.y + M.m[9] * p.z + M.m[13],
              M.m[2] * p.x + M.m[6] * p.y + M.m[10] * p.z + M.m[14]};
}
Vec3 xformVector(Mat4 const& M, Vec3 v) {
  return Vec3{M.m[0] * v.x + M.m[4] * v.y + M.m[8] * v.z, M.m[1] * v.x + M.m[5] * v.y + M.m[9] * v.z,
              M.m[2] * v.x + M.m[6] * v.y + M.m[10] * v.z};
}

std::string directoryOf(std::string const& path) {
  size_t const p = path.find_last_of('/');
  return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

struct Fail {
  std::string msg;
};
[[noreturn]] void fail(std::string const& m) { throw Fail{m}; }

bool extractVec3(Value const& v, Vec3& out) {  // extractVec3f: array of exactly three numbers
  if (!v.isArray() || v.array.size() != 3) return false;
  for (auto const& e : v.array)
    if (!e.isNumber()) return false;
  out = Vec3{float(v.array[0].number), float(v.array[1].number), float(v.array[2].number)};
  return true;
}
void onlyKeys(Value const& obj, std::set<std::string> const& allowed, std::string const& what) {
  for (auto const& kv : obj.object)
    if (!allowed.count(kv.first)) fail(what + " has extraneous key '" + kv.first + "'");
}

struct Material {
  Vec3 diffuse{1, 1, 1};
  float metallic = 0.f, roughness = 0.f, ior = 1.4f;
  Vec3 eta{0.18299f, 0.42108f, 1.37340f}, etak{3.42420f, 2.34590f, 1.77040f};
  float anisotropy = 1.f;  // (the reference leaves 0 when "ggx-anisotropy" is absent, which makes alpha_x = 0)
  Vec3 reflectanceTint{1, 1, 1}, transmittanceTint{1, 1, 1};
  bool orenNayar = false;
  int32_t diffuseTex = -1, roughnessTex = -1, normalTex = -1, metallicTex = -1;  // indices into State::textureList
  // 0 < metallic < 1, or a metallic map: the material becomes TWO records (dielectric tagged BS_GGX_BLEND, then conductor)
  bool blend() const { return !orenNayar && (metallicTex >= 0 || (metallic > 0.f && metallic < 1.f)); }
  int records() const { return blend() ? 2 : 1; }
};
struct LightProto {
  bool spot = false;
  Vec3 intensity{1, 1, 1};
  float cosTheta0 = 0.f, cosThetaE = 0.f;
};
struct Mesh {
  std::vector<Triangle> tris;
  std::vector<float> uv;  // six per triangle
  uint32_t material = 0;
};
struct Texture {
  std::string type;
  int32_t first = 0, width = 0, height = 0;  // texels in State::texels
};

// TriangleMesh::unitCube / unitPlane: positions and triangle order of core-trianglemesh.cpp:120-188
std::vector<Triangle> unitCube() {
  float const h = 0.5f;
  Vec3 const P[8] = {{-h, -h, h}, {-h, -h, -h}, {h, -h, -h}, {h, -h, h}, {-h, h, h}, {-h, h, -h}, {h, h, -h}, {h, h, h}};
  int const T[12][3] = {{1, 4, 5}, {1, 0, 4}, {2, 0, 1}, {2, 3, 0}, {6, 3, 2}, {6, 7, 3},
                        {0, 3, 4}, {4, 3, 7}, {7, 5, 4}, {7, 6, 5}, {1, 5, 6}, {2, 1, 6}};
  std::vector<Triangle> out;
  for (auto const& t : T) out.push_back(Triangle{P[t[0]], P[t[1]], P[t[2]]});
  return out;
}
// per-corner texture coordinates of unitCube / unitPlane (core-trianglemesh.cpp:103-156,174-185), six floats per triangle
std::vector<float> unitCubeUv() {
  float const t = 0.125f;
  float const U[14][2] = {{3 * t, 0.f},   {5 * t, 0.f},   {5 * t, 2 * t}, {5 * t, 4 * t}, {7 * t, 4 * t}, {7 * t, 6 * t}, {5 * t, 6 * t},
                          {5 * t, 1.f},   {3 * t, 1.f},   {3 * t, 6 * t}, {t, 6 * t},     {t, 4 * t},     {3 * t, 4 * t}, {3 * t, 2 * t}};
  int const I[12][3] = {{6, 4, 5}, {6, 3, 4}, {9, 3, 6}, {9, 12, 3}, {9, 12, 10}, {9, 11, 12},
                        {3, 12, 2}, {2, 12, 13}, {13, 1, 2}, {13, 0, 1}, {6, 7, 8}, {9, 6, 8}};
  std::vector<float> out;
  for (auto const& tri : I)
    for (int c : tri) out.push_back(U[c][0]), out.push_back(U[c][1]);
  return out;
}
std::vector<float> unitPlaneUv() { return {0, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1}; }
std::vector<Triangle> unitPlane() {
  float const h = 0.5f;
  Vec3 const P[4] = {{-h, -h, 0}, {h, -h, 0}, {-h, h, 0}, {h, h, 0}};
  return {Triangle{P[0], P[3], P[2]}, Triangle{P[0], P[1], P[3]}};
}

// core-material.cpp:272-286: metallic <= 0 -> the dielectric lobe alone, >= 1 -> the conductor alone, in between (or a
// metallic map) both, blended per hit: the dielectric record tagged BS_GGX_BLEND followed by the conductor record
void packMaterial(Material const& m, std::vector<Packed32>& out) {
  if (m.orenNayar) {
    out.push_back(makeOrenNayar(m.diffuse, m.roughness));
    return;
  }
  float const alphay = m.roughness, alphax = m.anisotropy * m.roughness;  // core-material.cpp:262-263
  if (m.blend()) {
    out.push_back(makeGGXBlendDielectric(m.reflectanceTint, m.transmittanceTint, 0.f, m.ior, alphax, alphay, m.metallic));
    out.push_back(makeGGXConductor(m.eta, m.etak, 0.f, alphax, alphay));
  } else if (m.metallic >= 1.f) {
    out.push_back(makeGGXConductor(m.eta, m.etak, 0.f, alphax, alphay));
  } else {
    out.push_back(makeGGXDielectric(m.reflectanceTint, m.transmittanceTint, 0.f, m.ior, alphax, alphay));
  }
}

struct State {
  std::map<std::string, uint32_t> materials, objects;
  std::map<std::string, uint32_t> textures;  // name -> index into textureList
  std::vector<Texture> textureList;
  std::vector<uint8_t> texels;               // RGBA8, all textures back to back
  std::map<std::string, LightProto> lights;
  std::map<std::string, Mat4> transforms;
  std::vector<Material> materialList;
  uint32_t recordCount = 0;  // packed BSDF records so far (a blend material takes two)
  std::vector<Mesh> meshes;
  std::vector<Mat4> stack;
};

void parseCamera(Value const& cam, JsonScene& out, Vec3& dir, Vec3& pos) {
  if (!cam.isObject() || cam.size() > 4) fail("'camera' should be an object with at most 4 members");  // :741
  if (!cam.contains("focalLength") || !cam.contains("sensorSize") || !cam.contains("direction"))
    fail("'camera' needs 'focalLength', 'sensorSize' and 'direction'");
  if (!cam.at("sensorSize").isNumber() || float(cam.at("sensorSize").number) <= 0.f) fail("camera 'sensorSize' should be a positive number");
  if (!cam.at("focalLength").isNumber() || float(cam.at("focalLength").number) <= 0.f) fail("camera 'focalLength' should be a positive number");
  if (!extractVec3(cam.at("direction"), dir)) fail("camera 'direction' should be an array of three numbers");
  out.scene.camera.focal_length = float(cam.at("focalLength").number);  // millimetres, as DeviceCamera holds them
  out.scene.camera.sensor_size = float(cam.at("sensorSize").number);
  pos = Vec3{0, 0, 0};
  if (cam.contains("position") && !extractVec3(cam.at("position"), pos)) fail("camera 'position' should be an array of three numbers");
  if (cam.contains("max-depth")) {
    if (!cam.at("max-depth").isInteger() || int(cam.at("max-depth").number) <= 0) fail("camera 'max-depth' should be a positive integer");
    out.maxDepth = int(cam.at("max-depth").number);
  }
}

void parseFilm(Value const& film, JsonScene& out) {
  if (!film.isObject() || film.size() > 3) fail("'film' should be an object with at most 3 members");
  if (!film.contains("resolutionX") || !film.contains("resolutionY")) fail("'film' needs 'resolutionX' and 'resolutionY'");
  if (!film.at("resolutionX").isNumber() || !film.at("resolutionY").isNumber()) fail("film resolution should be numeric");
  if (film.contains("samples")) {
    if (!film.at("samples").isInteger() || int(film.at("samples").number) <= 0) fail("'samples' Should be positive integer");
    out.samplesPerPixel = int(film.at("samples").number);
  }
  out.scene.camera.width = int(film.at("resolutionX").number);
  out.scene.camera.height = int(film.at("resolutionY").number);
  if (out.scene.camera.width <= 0 || out.scene.camera.height <= 0) fail("film resolution should be positive");
}

void parseTexture(Value const& t, State& st, std::string const& baseDir) {
  if (!t.isObject()) fail("The 'textures' array should contain only objects");
  onlyKeys(t, {"name", "type", "path"}, "Texture object");
  if (!t.contains("name") || !t.at("name").isString()) fail("Texture object should contain attribute 'name' and it should be a string");
  if (!t.contains("type") || !t.at("type").isString()) fail("Texture object should contain attribute 'type' and it should be a string");
  if (!t.contains("path") || !t.at("path").isString()) fail("Texture object should contain attribute 'path' and it should be a string");
  std::string const name = t.at("name").string, type = t.at("type").string;
  if (st.textures.count(name)) fail("texture " + name + " already exists");
  if (type != "diffuse" && type != "normal" && type != "metallic" && type != "roughness") fail("texture: unrecognized type '" + type + "'");
  // 8-bit PNGs (what the reference's scenes ship); channel rule of core-parser.cpp:373-386: diffuse / normal are
  // 3-channel images, metallic / roughness 1-channel ones
  std::vector<uint8_t> img;
  int w = 0, h = 0, ch = 0;
  std::string perr;
  if (!readPng8(baseDir + "/" + t.at("path").string, img, w, h, ch, &perr)) fail("Error loading texture '" + name + "': " + perr);
  bool const rgb = type == "diffuse" || type == "normal";
  if (rgb ? ch < 3 : (ch != 1 && ch != 2)) fail("Error loading texture '" + name + "': metallic, roughness expect 1 channel; diffuse, normal expect 3 channels");
  Texture tex;
  tex.type = type, tex.first = int32_t(st.texels.size() / 4), tex.width = w, tex.height = h;
  st.texels.reserve(st.texels.size() + size_t(w) * size_t(h) * 4);
  for (size_t i = 0; i < size_t(w) * size_t(h); ++i) {
    uint8_t const* p = &img[i * size_t(ch)];
    uint8_t const r = p[0], g = rgb ? p[1] : p[0], b = rgb ? p[2] : p[0];
    st.texels.insert(st.texels.end(), {r, g, b, uint8_t(255)});
  }
  st.textures[name] = uint32_t(st.textureList.size());
  st.textureList.push_back(tex);
}

// number, or the name of a texture of type `key` (returned through `tex`; the scalar then is the type's neutral value)
float scalarOrTexture(Value const& v, char const* key, State const& st, int32_t* tex) {
  if (v.isNumber()) return clampf(float(v.number), 0.f, 1.f);
  if (v.isString()) {
    if (!st.textures.count(v.string)) fail(std::string("'") + key + "' texture name should be an existing named texture");
    Texture const& t = st.textureList[st.textures.at(v.string)];
    if (t.type != key) fail(std::string("'") + key + "' material texture should point to a '" + key + "' texture");
    *tex = int32_t(st.textures.at(v.string));
    return 0.5f;
  }
  fail(std::string("material '") + key + "' should be either texture name or number");
}

void parseMaterial(Value const& m, State& st) {
  if (!m.isObject()) fail("The 'materials' array should contain only objects");
  if (!m.contains("name") || !m.at("name").isString()) fail("The material should have a unique 'name' *string*field in the materials namespace");
  std::string const name = m.at("name").string;
  if (st.materials.count(name)) fail("Duplicate material name '" + name + "'");
  onlyKeys(m, {"name", "diffuse", "metallic", "normal", "roughness", "ior", "eta", "etak", "ggx-anisotropy", "ggx-dielectric",
               "oren-nayar-dielectric"}, "material '" + name + "'");
  if (!m.contains("diffuse")) fail("material should specify a 'diffuse' either as RGB or texture name");
  if (!m.contains("metallic")) fail("material should specify a 'metallic' either as float or texture name");
  if (!m.contains("roughness")) fail("material should specify a 'roughness' either as float or texture name");
  Material mat;
  if (m.at("diffuse").isArray()) {
    Vec3 d;
    if (!extractVec3(m.at("diffuse"), d)) fail("'diffuse' constant expected to be RGB value");
    // byte3FromRGB(clamp01): the reference stores the constant as 8-bit
    auto q = [](float v) { return float(uint8_t(clampf(v, 0.f, 1.f) * 255.f)) / 255.f; };
    mat.diffuse = Vec3{q(d.x), q(d.y), q(d.z)};
  } else if (m.at("diffuse").isString()) {
    if (!st.textures.count(m.at("diffuse").string)) fail("'diffuse' texture name should be an existing named texture");
    if (st.textureList[st.textures.at(m.at("diffuse").string)].type != "diffuse") fail("'diffuse' material texture should point to a 'diffuse' texture");
    mat.diffuseTex = int32_t(st.textures.at(m.at("diffuse").string));
  } else {
    fail("material 'diffuse' should be either texture name or RGB");
  }
  if (m.contains("normal")) {  // core-parser.cpp:507-534
    if (!m.at("normal").isString()) fail("material 'normal' should be an RGB texture name");
    if (!st.textures.count(m.at("normal").string)) fail("'normal' texture name should be an existing named texture");
    if (st.textureList[st.textures.at(m.at("normal").string)].type != "normal") fail("'normal' material texture should point to a 'normal' texture");
    mat.normalTex = int32_t(st.textures.at(m.at("normal").string));
  }
  mat.roughness = scalarOrTexture(m.at("roughness"), "roughness", st, &mat.roughnessTex);
  mat.metallic = scalarOrTexture(m.at("metallic"), "metallic", st, &mat.metallicTex);
  if (m.contains("ior")) {
    if (!m.at("ior").isNumber()) fail("material 'ior' should be a number");
    mat.ior = std::fmax(float(m.at("ior").number), 1.f);
  }
  bool const hasEta = m.contains("eta"), hasEtak = m.contains("etak");
  if (hasEta != hasEtak) fail("material should specify both 'eta' and 'etak' or neither");
  if (hasEta) {
    Vec3 e, k;
    if (!extractVec3(m.at("eta"), e) || !extractVec3(m.at("etak"), k)) fail("'eta' and 'etak' expected to be RGB values");
    mat.eta = Vec3{std::fmax(e.x, 0.f), std::fmax(e.y, 0.f), std::fmax(e.z, 0.f)};
    mat.etak = Vec3{std::fmax(k.x, 0.f), std::fmax(k.y, 0.f), std::fmax(k.z, 0.f)};
  }
  if (m.contains("ggx-anisotropy")) {
    Value const& a = m.at("ggx-anisotropy");
    if (!a.isNumber() || float(a.number) < 0.f || float(a.number) > 1.f) fail("'ggx-anisotropy' should be a number in [0, 1]");
    float const t = clampf(float(a.number), 0.f, 1.f);
    mat.anisotropy = t == 0.f ? 1.f : (8.f - 1.f) * t + 1.f;  // fl::lerp(t, 1, 8)
  }
  bool dielectric = false;
  if (m.contains("ggx-dielectric")) {
    dielectric = true;
    Value const& g = m.at("ggx-dielectric");
    if (!g.isObject() && !g.isNull()) fail("'ggx-dielectric' should be an object or null");
    if (g.isObject()) {
      onlyKeys(g, {"reflectance-tint", "transmittance-tint"}, "'ggx-dielectric'");
      Vec3 t;
      if (g.contains("reflectance-tint")) {
        if (!extractVec3(g.at("reflectance-tint"), t)) fail("'reflectance-tint' expected to be RGB value");
        mat.reflectanceTint = Vec3{clampf(t.x, 0, 1), clampf(t.y, 0, 1), clampf(t.z, 0, 1)};
      }
      if (g.contains("transmittance-tint")) {
        if (!extractVec3(g.at("transmittance-tint"), t)) fail("'transmittance-tint' expected to be RGB value");
        mat.transmittanceTint = Vec3{clampf(t.x, 0, 1), clampf(t.y, 0, 1), clampf(t.z, 0, 1)};
      }
    }
  }
  if (m.contains("oren-nayar-dielectric")) {
    if (dielectric) fail("material should specify only one of 'ggx-dielectric' and 'oren-nayar-dielectric'");
    Value const& o = m.at("oren-nayar-dielectric");
    if (!o.isObject() && !o.isNull()) fail("'oren-nayar-dielectric' should be an object or null");
    if (o.isObject() && o.contains("multiscatter-multiplier")) {
      Value const& j = o.at("multiscatter-multiplier");
      if (!j.isNumber() || float(j.number) <= 0.f) fail("'multiscatter-multiplier' should be a positive number");
    }
    mat.orenNayar = true;
  }
  if (mat.orenNayar && mat.metallicTex >= 0) fail("material '" + name + "': a 'metallic' texture on an 'oren-nayar-dielectric' material is not supported");
  st.materials[name] = st.recordCount;  // index of the material's (first) packed record
  st.recordCount += uint32_t(mat.records());
  st.materialList.push_back(mat);
}

void parseObject(Value const& o, State& st, std::string const& baseDir) {
  if (!o.isObject()) fail("The 'objects' array should contain only objects");
  if (!o.contains("name") || !o.at("name").isString()) fail("object should have a 'name' string");
  std::string const name = o.at("name").string;
  if (st.objects.count(name)) fail("Duplicate object name '" + name + "'");
  if (!o.contains("material") || !o.at("material").isString()) fail("object '" + name + "' should have a 'material' string");
  if (!st.materials.count(o.at("material").string)) fail("object '" + name + "': unknown material '" + o.at("material").string + "'");
  if (!o.contains("type") || !o.at("type").isString()) fail("object '" + name + "' should have a 'type' string");
  std::string const type = o.at("type").string;
  Mesh mesh;
  mesh.material = st.materials.at(o.at("material").string);
  if (type == "fbx" || type == "FBX") {
    onlyKeys(o, {"name", "type", "material", "path"}, "object '" + name + "'");
    if (!o.contains("path") || !o.at("path").isString()) fail("object '" + name + "' should have a 'path' string");
    std::string ferr;
    if (!readFbxMesh(baseDir + "/" + o.at("path").string, mesh.tris, &ferr, &mesh.uv)) fail("object '" + name + "': " + ferr);
  } else if (type == "primitive") {
    onlyKeys(o, {"name", "type", "material", "shape"}, "object '" + name + "'");
    if (!o.contains("shape") || !o.at("shape").isString()) fail("object '" + name + "' should have a 'shape' string");
    std::string const shape = o.at("shape").string;
    if (shape == "cube") mesh.tris = unitCube(), mesh.uv = unitCubeUv();
    else if (shape == "plane") mesh.tris = unitPlane(), mesh.uv = unitPlaneUv();
    else fail("object '" + name + "': unrecognized shape '" + shape + "'");
  } else {
    fail("object '" + name + "': unrecognized type '" + type + "'");
  }
  st.objects[name] = uint32_t(st.meshes.size());
  st.meshes.push_back(std::move(mesh));
}

void parseLight(Value const& l, State& st) {
  if (!l.isObject()) fail("The 'lights' array should contain only objects");
  if (!l.contains("name") || !l.at("name").isString()) fail("light should have a 'name' string");
  std::string const name = l.at("name").string;
  if (st.lights.count(name)) fail("Duplicate light name '" + name + "'");
  if (!l.contains("type") || !l.at("type").isString()) fail("light '" + name + "' should have a 'type' string");
  std::string const type = l.at("type").string;
  LightProto p;
  if (type == "point") {
    onlyKeys(l, {"name", "type", "radiant-intensity"}, "light '" + name + "'");
    if (l.contains("radiant-intensity") && !extractVec3(l.at("radiant-intensity"), p.intensity)) fail("'radiant-intensity' expected to be RGB value");
  } else if (type == "spot") {
    onlyKeys(l, {"name", "type", "radiant-intensity", "cone-angle", "falloff-percentage"}, "light '" + name + "'");
    p.spot = true;
    float coneAngle = 60.f, falloff = 10.f;
    if (l.contains("radiant-intensity")) {
      Vec3 t;
      if (!extractVec3(l.at("radiant-intensity"), t)) fail("'radiant-intensity' expected to be RGB value");
      p.intensity = Vec3{std::fmax(t.x, 0.f), std::fmax(t.y, 0.f), std::fmax(t.z, 0.f)};
    }
    if (l.contains("cone-angle")) {
      if (!l.at("cone-angle").isNumber()) fail("'cone-angle' should be a number");
      coneAngle = clampf(float(l.at("cone-angle").number), 10.f, 120.f);
    }
    if (l.contains("falloff-percentage")) {
      if (!l.at("falloff-percentage").isNumber()) fail("'falloff-percentage' should be a number");
      falloff = clampf(float(l.at("falloff-percentage").number), 1.f, 80.f);
    }
    float const kPi = 3.14159265358979323846f;
    float const c0 = std::cos(coneAngle * (1.f - falloff / 100.f) * kPi / 180.f), ce = std::cos(coneAngle * kPi / 180.f);
    p.cosTheta0 = std::fmax(c0, ce);           // makeSpotLight, core-light.cpp:34-35
    p.cosThetaE = std::fmin(p.cosTheta0, ce);
  } else {
    fail("light '" + name + "': unrecognized type '" + type + "'");
  }
  st.lights[name] = p;
}

void parseTransform(Value const& t, State& st) {
  if (!t.isObject()) fail("The 'transforms' array should contain only objects");
  onlyKeys(t, {"name", "srt"}, "transform");
  if (!t.contains("name") || !t.at("name").isString() || t.at("name").string.empty()) fail("transform should have a non-empty 'name' string");
  std::string const name = t.at("name").string;
  if (st.transforms.count(name)) fail("Duplicate transform name '" + name + "'");
  Vec3 axis{0, 0, 1}, translation{0, 0, 0}, s{1, 1, 1};
  float degrees = 0.f;
  if (t.contains("srt")) {
    Value const& srt = t.at("srt");
    if (srt.contains("rotate-axis")) {
      Vec3 a;
      if (!extractVec3(srt.at("rotate-axis"), a)) fail("'rotate-axis' should be an array of three numbers");
      axis = normalized(a);
    }
    if (srt.contains("rotate-degrees")) {
      if (!srt.at("rotate-degrees").isNumber()) fail("'rotate-degrees' should be a number");
      degrees = float(srt.at("rotate-degrees").number);
    }
    if (srt.contains("translation-vector") && !extractVec3(srt.at("translation-vector"), translation))
      fail("'translation-vector' should be an array of three numbers");
    if (srt.contains("scale")) {
      if (srt.at("scale").isNumber()) s = Vec3{float(srt.at("scale").number), float(srt.at("scale").number), float(srt.at("scale").number)};
      else if (!extractVec3(srt.at("scale"), s)) fail("'scale' should be a number or an array of three numbers");
    }
  }
  st.transforms[name] = mul(mul(translate(translation), rotate(degrees, axis)), scale(s));
}

void emitTriangle(Scene& sc, Triangle const& t, uint32_t material) {
  sc.xs.insert(sc.xs.end(), {t.v0.x, t.v1.x, t.v2.x, 0.f});
  sc.ys.insert(sc.ys.end(), {t.v0.y, t.v1.y, t.v2.y, 0.f});
  sc.zs.insert(sc.zs.end(), {t.v0.z, t.v1.z, t.v2.z, 0.f});
  sc.matId.push_back(material);
}

void walkWorld(Value const& node, State& st, Scene& sc) {  // parseWorldTranform, core-parser.cpp:1262-1335
  if (!node.isObject()) fail("'world' nodes should be objects");
  for (auto const& kv : node.object) {
    std::string const& key = kv.first;
    Value const& value = kv.second;
    if (st.transforms.count(key)) {
      st.stack.push_back(st.transforms.at(key));
      walkWorld(value, st, sc);
      st.stack.pop_back();
    }
    Mat4 cur = st.stack.back();
    for (size_t i = st.stack.size() - 1; i-- > 0;) cur = mul(st.stack[i], cur);
    if (key == "instances") {
      if (!value.isArray()) fail("'instances' should be an array of object names");
      for (auto const& o : value.array) {
        if (!o.isString() || !st.objects.count(o.string)) fail("'instances' refers to an unknown object");
        Mesh const& mesh = st.meshes[st.objects.at(o.string)];
        for (size_t k = 0; k < mesh.tris.size(); ++k) {
          Triangle const& t = mesh.tris[k];
          emitTriangle(sc, Triangle{xformPoint(cur, t.v0), xformPoint(cur, t.v1), xformPoint(cur, t.v2)}, mesh.material);
          for (int j = 0; j < 6; ++j) sc.triUv.push_back(6 * k + size_t(j) < mesh.uv.size() ? mesh.uv[6 * k + size_t(j)] : 0.f);
        }
      }
    } else if (key == "lights") {
      if (!value.isArray()) fail("'lights' should be an array of light names");
      for (auto const& l : value.array) {
        if (!l.isString() || !st.lights.count(l.string)) fail("'lights' refers to an unknown light");
        LightProto const& p = st.lights.at(l.string);
        Vec3 const pos{cur.m[12], cur.m[13], cur.m[14]};
        if (p.spot) sc.lights.push_back(makeSpotLight(p.intensity, pos, normalized(xformVector(cur, Vec3{0, 1, 0})), p.cosTheta0, p.cosThetaE, 1e-3f));
        else sc.lights.push_back(makePointLight(p.intensity, pos, 1e-3f));
      }
    }
  }
}

}  // namespace

// ---- PNG reader: 8-bit grey / RGB / RGBA, non-interlaced (what the env maps of the reference's scenes are) ------
bool readPng8(std::string const& path, std::vector<uint8_t>& img, int& width, int& height, int& channels, std::string* error) {
  auto bad = [&](char const* m) {
    if (error) *error = path + ": " + m;
    return false;
  };
  std::ifstream f(path, std::ios::binary);
  if (!f) return bad("cannot open");
  std::vector<unsigned char> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static unsigned char const sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return bad("not a PNG file");
  auto be32 = [&](size_t o) { return (uint32_t(file[o]) << 24) | (uint32_t(file[o + 1]) << 16) | (uint32_t(file[o + 2]) << 8) | uint32_t(file[o + 3]); };
  uint32_t w = 0, h = 0;
  channels = 0;
  std::vector<unsigned char> idat;
  for (size_t o = 8; o + 12 <= file.size();) {
    uint32_t const len = be32(o);
    if (o + 12 + len > file.size()) return bad("truncated chunk");
    char const* type = reinterpret_cast<char const*>(&file[o + 4]);
    unsigned char const* data = &file[o + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) return bad("bad IHDR");
      w = be32(o + 8), h = be32(o + 12);
      int const depth = data[8], color = data[9], interlace = data[12];
      if (depth != 8 || interlace != 0) return bad("only 8-bit non-interlaced PNGs are supported");
      channels = color == 0 ? 1 : (color == 2 ? 3 : (color == 6 ? 4 : (color == 4 ? 2 : 0)));
      if (!channels) return bad("palette PNGs are not supported");
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    o += 12 + size_t(len);
  }
  if (!w || !h || idat.empty()) return bad("missing IHDR / IDAT");
  size_t const stride = size_t(w) * size_t(channels);
  std::vector<unsigned char> raw((stride + 1) * size_t(h));
  uLongf rawLen = uLongf(raw.size());
  if (uncompress(raw.data(), &rawLen, idat.data(), uLong(idat.size())) != Z_OK || rawLen != raw.size()) return bad("zlib stream does not match the image size");
  img.assign(stride * size_t(h), 0);
  for (size_t y = 0; y < h; ++y) {  // undo the per-row filters (PNG spec 9.2)
    unsigned char const* in = &raw[y * (stride + 1)];
    unsigned char* out = &img[y * stride];
    unsigned char const* up = y ? &img[(y - 1) * stride] : nullptr;
    int const filter = in[0];
    for (size_t x = 0; x < stride; ++x) {
      int const a = x >= size_t(channels) ? out[x - size_t(channels)] : 0, b = up ? up[x] : 0;
      int const c = (up && x >= size_t(channels)) ? up[x - size_t(channels)] : 0;
      int pred = 0;
      switch (filter) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          int const p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return bad("bad row filter");
      }
      out[x] = static_cast<unsigned char>(in[1 + x] + pred);
    }
  }
  width = int(w), height = int(h);
  return true;
}
bool readPngRgb(std::string const& path, std::vector<float>& rgb, int& width, int& height, std::string* error) {
  std::vector<uint8_t> img;
  int channels = 0;
  if (!readPng8(path, img, width, height, channels, error)) return false;
  size_t const n = size_t(width) * size_t(height);
  rgb.resize(n * 3);
  for (size_t i = 0; i < n; ++i) {  // loadImageAsRGB: byte / 255, grey replicated
    unsigned char const* p = &img[i * size_t(channels)];
    if (channels >= 3) rgb[3 * i] = p[0] / 255.0f, rgb[3 * i + 1] = p[1] / 255.0f, rgb[3 * i + 2] = p[2] / 255.0f;
    else rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = p[0] / 255.0f;
  }
  return true;
}

bool loadJsonScene(std::string const& path, JsonScene& out, std::string* error) {
  try {
    std::ifstream f(path);
    if (!f) fail("cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    std::string const text = ss.str();
    Value data;
    std::string err;
    if (!json::Reader(text).parse(data, err)) fail(err);
    if (!data.isObject()) fail("JSON value should be an object");
    static char const* const keys[] = {"camera", "film", "textures", "materials", "objects", "lights", "envlight", "transforms", "world"};
    for (char const* k : keys)
      if (!data.contains(k)) fail("JSON value is lacking some keys");
    for (auto const& kv : data.object) {
      bool known = false;
      for (char const* k : keys) known = known || kv.first == k;
      if (!known) fail("JSON value has extraneous key '" + kv.first + "'");
    }
    out = JsonScene{};
    State st;
    Vec3 dir, pos;
    parseCamera(data.at("camera"), out, dir, pos);
    parseFilm(data.at("film"), out);
    out.scene.camera.dir[0] = dir.x, out.scene.camera.dir[1] = dir.y, out.scene.camera.dir[2] = dir.z;
    out.scene.camera.pos[0] = pos.x, out.scene.camera.pos[1] = pos.y, out.scene.camera.pos[2] = pos.z;
    out.scene.camera.spp = out.samplesPerPixel;
    if (!data.at("textures").isArray()) fail("'textures' should be a JSON array");
    for (auto const& t : data.at("textures").array) parseTexture(t, st, directoryOf(path));
    if (!data.at("materials").isArray()) fail("'materials' should be a JSON array");
    for (auto const& m : data.at("materials").array) parseMaterial(m, st);
    if (!data.at("objects").isArray()) fail("'objects' should be a JSON array");
    for (auto const& o : data.at("objects").array) parseObject(o, st, directoryOf(path));
    if (!data.at("lights").isArray()) fail("'lights' should be a JSON array");
    for (auto const& l : data.at("lights").array) parseLight(l, st);
    if (!data.at("envlight").isString()) fail("'envlight' should be a path string");
    {
      std::string const envPath = directoryOf(path) + "/" + data.at("envlight").string;
      std::string perr;
      if (!readPngRgb(envPath, out.scene.envRgb, out.scene.envWidth, out.scene.envHeight, &perr)) fail("envlight: " + perr);
    }
    if (!data.at("transforms").isArray()) fail("'transforms' should be a JSON array");
    for (auto const& t : data.at("transforms").array) parseTransform(t, st);
    if (!data.at("world").isObject()) fail("'world' should be an object");
    for (auto const& kv : data.at("world").object) {  // top level: only transform names open a subtree (:1427-1437)
      if (!st.transforms.count(kv.first)) continue;
      st.stack.push_back(st.transforms.at(kv.first));
      walkWorld(kv.second, st, out.scene);
      st.stack.pop_back();
    }
    for (Material const& m : st.materialList) packMaterial(m, out.scene.bsdfs);
    // image textures: only when some material uses one (otherwise the plain kernels run and nothing is uploaded)
    bool textured = false;
    for (Material const& m : st.materialList)
      textured = textured || m.diffuseTex >= 0 || m.roughnessTex >= 0 || m.normalTex >= 0 || m.metallicTex >= 0;
    if (textured) {
      out.scene.texRgba = std::move(st.texels);
      for (Texture const& t : st.textureList) out.scene.texDesc.insert(out.scene.texDesc.end(), {t.first, t.width, t.height});
      for (Material const& m : st.materialList) {  // one row per packed RECORD
        uint32_t bits;
        memcpy(&bits, &m.anisotropy, 4);
        out.scene.matTex.insert(out.scene.matTex.end(), {uint32_t(m.diffuseTex), uint32_t(m.roughnessTex), uint32_t(m.normalTex), bits});
        // the conductor half of a blend: same roughness map and normal map; its first slot carries the METALLIC map
        if (m.blend()) out.scene.matTex.insert(out.scene.matTex.end(), {uint32_t(m.metallicTex), uint32_t(m.roughnessTex), uint32_t(m.normalTex), bits});
      }
    } else {
      out.scene.triUv.clear();
    }
    return true;
  } catch (Fail const& e) {
    if (error) *error = e.msg;
    return false;
  } catch (std::exception const& e) {
    if (error) *error = e.what();
    return false;
  }
}

}  // namespace dmt_host
