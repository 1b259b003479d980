// dmt_scene.cpp -- packers, procedural meshes and the hard-coded scene; see dmt_scene.hpp.
#include "dmt_scene.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

namespace dmt_host {
namespace {

constexpr float PI = 3.14159265358979323846f;

Vec3 add(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
Vec3 scale(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
Vec3 crossp(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
Vec3 unit(Vec3 a) {
  float const inv = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  return {a.x * inv, a.y * inv, a.z * inv};
}
float clamp01(float v) { return fmaxf(0, fminf(v, 1)); }

// little-endian field writer over a 32-byte record
struct Fields {
  Packed32 rec;
  void u16(int off, uint16_t v) { memcpy(rec.bytes + off, &v, 2); }
  void u32(int off, uint32_t v) { memcpy(rec.bytes + off, &v, 4); }
  void f32(int off, float v) { memcpy(rec.bytes + off, &v, 4); }
  void half(int off, float v) { u16(off, float_to_half_bits(v)); }
  void half3(int off, Vec3 v) { half(off, v.x), half(off + 2, v.y), half(off + 4, v.z); }
  void vec3(int off, Vec3 v) { f32(off, v.x), f32(off + 4, v.y), f32(off + 8, v.z); }
  float readHalf(int off) const {
    uint16_t v;
    memcpy(&v, rec.bytes + off, 2);
    return half_bits_to_float(v);
  }
};

// record field offsets (bsdf.cuh:18-73, light.cuh:10-49)
namespace bsdf_off {
constexpr int weight = 0, type = 6;
constexpr int onMulti = 14, onRough = 20, onA = 22, onB = 24;
constexpr int gEnergy = 8, gPhi0 = 12, gAx = 14, gAy = 16;
constexpr int dEta = 18, dRefl = 20, dTrans = 26;
constexpr int cEta = 18, cKappa = 24;
}  // namespace bsdf_off
namespace light_off {
constexpr int intensity = 0, type = 6, pos = 8;
constexpr int pointRadius = 20;
constexpr int spotDir = 20, spotCos0 = 24, spotCosE = 26, spotRadius = 28;
constexpr int dirDir = 8, dirOmc = 12;
}  // namespace light_off
enum BsdfType : uint16_t { kOrenNayar = 0, kGGXDielectric = 1, kGGXConductor = 2, kLambert = 3 };
enum LightType : uint16_t { kPoint = 0, kSpot = 1, kEnv = 2, kDirectional = 3 };

Fields newBsdf(BsdfType t, Vec3 albedo) {
  Fields f;
  f.half3(bsdf_off::weight, albedo);
  f.u16(bsdf_off::type, t);
  return f;
}
void ggxCommon(Fields& f, float ax, float ay, float phi0) {  // bsdf.cu:436-450
  float const top = 65535.f;
  f.u16(bsdf_off::gAx, static_cast<uint16_t>(fminf(fmaxf(ax * top, 0.f), top)));
  f.u16(bsdf_off::gAy, static_cast<uint16_t>(fminf(fmaxf(ay * top, 0.f), top)));
  f.f32(bsdf_off::gEnergy, 1.f);
  f.u16(bsdf_off::gPhi0, static_cast<uint16_t>(fminf(fmaxf(phi0 / (2.f * PI) * top, 0.f), top)));
}
Fields newLight(LightType t, Vec3 color) {
  Fields f;
  f.half3(light_off::intensity, color);
  f.u16(light_off::type, t);
  return f;
}

float signPm1(float v) { return std::signbit(v) ? -1.f : 1.f; }
// the reference clamps to [0,1] before rounding, so a component packs to 0 or 1 (encoding.cu:17-21)
uint32_t octaComponent(float v) {
  return static_cast<uint32_t>(roundf(fmaxf(fminf((v + 1) * 0.5f * 65535.f, 1.f), 0.f)));
}

}  // namespace

// ---- codecs -------------------------------------------------------------------------------------
uint16_t float_to_half_bits(float f) {  // encoding.cu:78-116: add half an ulp, truncate (ties up)
  uint32_t bits;
  memcpy(&bits, &f, 4);
  uint32_t const s = (bits >> 16) & 0x8000u;
  int32_t e = static_cast<int32_t>((bits >> 23) & 0xFFu) - 112;
  uint32_t m = bits & 0x7FFFFFu;
  if (e >= 31) return static_cast<uint16_t>(s | 0x7C00u | (m ? 0x200u : 0u));
  if (e <= 0) {
    if (e < -10) return static_cast<uint16_t>(s);
    m |= 0x800000u;
    uint32_t const sh = static_cast<uint32_t>(14 - e);
    return static_cast<uint16_t>(s | static_cast<uint16_t>((m >> sh) + ((m >> (sh - 1)) & 1u)));
  }
  m += 0x1000u;
  if (m & 0x800000u) {
    m = 0;
    if (++e >= 31) return static_cast<uint16_t>(s | 0x7C00u);
  }
  return static_cast<uint16_t>(s | static_cast<uint16_t>(e << 10) | static_cast<uint16_t>(m >> 13));
}
float half_bits_to_float(uint16_t h) {  // encoding.cu:124-155
  uint32_t const s = static_cast<uint32_t>(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, out;
  if (e == 0) {
    if (m == 0) {
      out = s;
    } else {
      e = 113;
      while (!(m & 0x400u)) m <<= 1, --e;
      out = s | (e << 23) | ((m & 0x3FFu) << 13);
    }
  } else if (e == 31) {
    out = s | 0x7F800000u | (m << 13);
  } else {
    out = s | ((e + 112) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &out, 4);
  return f;
}
uint32_t octaFromDir(Vec3 d) {  // encoding.cu:26-37
  float const l1 = fabsf(d.x) + fabsf(d.y) + fabsf(d.z);
  float x = d.x / l1, y = d.y / l1;
  float const z = d.z / l1;
  bool const flip = z < 0.f;
  x = flip * (1.f - fabsf(y)) * signPm1(x) + !flip * x;
  y = flip * (1.f - fabsf(x)) * signPm1(y) + !flip * y;
  return octaComponent(y) << 16 | octaComponent(x);
}
Vec3 dirFromOcta(uint32_t o) {  // encoding.cu:39-60
  float const fx = static_cast<float>(o & 0xFFFFu) / 65535.f * 2.f - 1.f;
  float const fy = static_cast<float>((o >> 16) & 0xFFFFu) / 65535.f * 2.f - 1.f;
  Vec3 n{fx, fy, 1.f - fabsf(fx) - fabsf(fy)};
  bool const flip = n.z < 0.f;
  n.x = flip * (1.f - fabsf(fy)) * signPm1(fx) + !flip * fx;
  n.y = flip * (1.f - fabsf(fx)) * signPm1(fy) + !flip * fy;
  return unit(n);
}

// ---- packers ------------------------------------------------------------------------------------
Packed32 makeLambert() { return newBsdf(kLambert, {1, 1, 1}).rec; }
Packed32 makeOrenNayar(Vec3 color, float roughness) {
  float const k = (PI / 2.f) - 2.f / 3.f;
  Fields f = newBsdf(kOrenNayar, {clamp01(color.x), clamp01(color.y), clamp01(color.z)});
  f.half(bsdf_off::onRough, fmaxf(0, fminf(roughness, PI / 2.f)));
  float const sigma = f.readHalf(bsdf_off::onRough);  // terms are derived from the STORED halves
  f.half(bsdf_off::onA, 1.f / (PI + k * sigma));
  float const a = f.readHalf(bsdf_off::onA);
  f.half(bsdf_off::onB, a * sigma);
  f.half3(bsdf_off::onMulti, {1.f, 1.f, 1.f});
  return f.rec;
}
Packed32 makeGGXDielectric(Vec3 reflectanceTint, Vec3 transmittanceTint, float phi0, float eta, float alphax,
                           float alphay) {
  Fields f = newBsdf(kGGXDielectric, {1, 1, 1});
  ggxCommon(f, alphax, alphay, phi0);
  f.half(bsdf_off::dEta, eta);
  f.half3(bsdf_off::dRefl, reflectanceTint);
  f.half3(bsdf_off::dTrans, transmittanceTint);
  return f.rec;
}
// This build's own record (pt_device.hpp BS_GGX_BLEND): the dielectric half of a material that is "metallic" by a
// fraction; the conductor record of the same material follows it in the array.  The fraction sits in the first half of
// the weight field, which prepareBSDF overwrites for GGX records anyway.
Packed32 makeGGXBlendDielectric(Vec3 reflectanceTint, Vec3 transmittanceTint, float phi0, float eta, float alphax, float alphay,
                                float metallic) {
  Fields f;
  f.rec = makeGGXDielectric(reflectanceTint, transmittanceTint, phi0, eta, alphax, alphay);
  f.u16(bsdf_off::type, 4);
  f.half(bsdf_off::weight, metallic);
  return f.rec;
}
Packed32 makeGGXConductor(Vec3 eta, Vec3 kappa, float phi0, float alphax, float alphay) {
  Fields f = newBsdf(kGGXConductor, {1, 1, 1});
  ggxCommon(f, alphax, alphay, phi0);
  f.half3(bsdf_off::cEta, eta);
  f.half3(bsdf_off::cKappa, kappa);
  return f.rec;
}
Packed32 makePointLight(Vec3 color, Vec3 position, float radius) {
  Fields f = newLight(kPoint, color);
  f.vec3(light_off::pos, position);
  f.half(light_off::pointRadius, radius);
  return f.rec;
}
Packed32 makeSpotLight(Vec3 color, Vec3 position, Vec3 direction, float cosTheta0, float cosThetaE, float radius) {
  Fields f = newLight(kSpot, color);
  f.vec3(light_off::pos, position);
  f.u32(light_off::spotDir, octaFromDir(direction));
  f.half(light_off::spotCos0, cosTheta0);
  f.half(light_off::spotCosE, cosThetaE);
  f.half(light_off::spotRadius, radius);
  return f.rec;
}
Packed32 makeDirectionalLight(Vec3 color, Vec3 direction, float oneMinusCosAngle) {
  Fields f = newLight(kDirectional, color);
  f.u32(light_off::dirDir, octaFromDir(direction));
  f.half(light_off::dirOmc, oneMinusCosAngle);
  return f.rec;
}
Packed32 makeEnvironmentalLight(Vec3 color) { return newLight(kEnv, color).rec; }

// ---- meshes -------------------------------------------------------------------------------------
std::vector<Triangle> generateSphereMesh(Vec3 c, float radius, int latSubdiv, int lonSubdiv) {
  std::vector<Triangle> out;
  Vec3 const top = add(c, {0, radius, 0}), bottom = add(c, {0, -radius, 0});
  for (int i = 0; i < latSubdiv; ++i) {
    float const t0 = PI * float(i) / latSubdiv, t1 = PI * float(i + 1) / latSubdiv;
    float const y0 = radius * cosf(t0), y1 = radius * cosf(t1);
    float const r0 = radius * sinf(t0), r1 = radius * sinf(t1);
    for (int j = 0; j < lonSubdiv; ++j) {
      float const a0 = 2.f * PI * float(j) / lonSubdiv;
      float const a1 = 2.f * PI * float((j + 1) % lonSubdiv) / lonSubdiv;
      Vec3 const p00 = add(c, {r0 * cosf(a0), y0, r0 * sinf(a0)});
      Vec3 const p01 = add(c, {r0 * cosf(a1), y0, r0 * sinf(a1)});
      Vec3 const p10 = add(c, {r1 * cosf(a0), y1, r1 * sinf(a0)});
      Vec3 const p11 = add(c, {r1 * cosf(a1), y1, r1 * sinf(a1)});
      if (i == 0) {
        out.push_back({top, p10, p11});
      } else if (i == latSubdiv - 1) {
        out.push_back({p00, bottom, p01});
      } else {
        out.push_back({p00, p10, p01});
        out.push_back({p01, p10, p11});
      }
    }
  }
  return out;
}
std::vector<Triangle> generateCube(Vec3 center, Vec3 s) {
  Vec3 corner[8];
  for (int i = 0; i < 8; ++i)
    corner[i] = add(center, {((i & 1) ? 0.5f : -0.5f) * s.x, ((i & 2) ? 0.5f : -0.5f) * s.y,
                             ((i & 4) ? 0.5f : -0.5f) * s.z});
  int const faces[6][4] = {{0, 1, 3, 2}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 3, 7, 5}};
  std::vector<Triangle> out;
  for (auto const& q : faces) {
    out.push_back({corner[q[0]], corner[q[1]], corner[q[2]]});
    out.push_back({corner[q[0]], corner[q[2]], corner[q[3]]});
  }
  return out;
}
std::vector<Triangle> generatePlane(Vec3 center, Vec3 normal, float width, float height) {
  Vec3 const n = unit(normal);
  float const ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
  Vec3 const major = (ax <= ay && ax <= az) ? Vec3{1, 0, 0} : (ay <= ax && ay <= az) ? Vec3{0, 1, 0} : Vec3{0, 0, 1};
  Vec3 t = unit(crossp(major, n));
  Vec3 b = crossp(n, t);
  t = scale(t, width * 0.5f);
  b = scale(b, height * 0.5f);
  Vec3 const p0 = sub(sub(center, t), b), p1 = sub(add(center, t), b);
  Vec3 const p2 = add(add(center, t), b), p3 = add(sub(center, t), b);
  return {{p0, p2, p1}, {p0, p3, p2}};
}

// ---- scene --------------------------------------------------------------------------------------
void Scene::addModel(std::vector<Triangle> const& mesh, uint32_t materialIndex) {
  uint32_t const mat = firstMesh_ ? 0u : materialIndex;  // host_utils.cu:139-152
  firstMesh_ = false;
  for (Triangle const& t : mesh) {
    xs.insert(xs.end(), {t.v0.x, t.v1.x, t.v2.x, 0.f});
    ys.insert(ys.end(), {t.v0.y, t.v1.y, t.v2.y, 0.f});
    zs.insert(zs.end(), {t.v0.z, t.v1.z, t.v2.z, 0.f});
    matId.push_back(mat);
  }
}

static dmt_camera defaultCamera() {  // DeviceCamera defaults, types.cuh:101-109
  dmt_camera c{};
  c.dir[1] = 1.f;
  c.width = 16, c.height = 16, c.spp = 2;
  c.focal_length = 20.f, c.sensor_size = 36.f;
  return c;
}

Scene cornellBox() {
  Scene s;
  Vec3 const white{0.9f, 170.f / 204.f, 160.f / 204.f};
  Vec3 const orange{1.f, .7f, .3f};
  struct Wall {
    Vec3 center, normal;
    Packed32 bsdf;
  };
  s.addModel(generateSphereMesh({-1.2f, 2.f, -0.25f}, 0.5f, 2, 4), 0);
  s.bsdfs.push_back(makeOrenNayar(orange, .7f));
  s.addModel(generateSphereMesh({1.2f, 2.4f, -0.25f}, 0.5f, 2, 4), 1);
  s.bsdfs.push_back(makeGGXDielectric({0.02f, 0.07f, 0.01f}, {0.95f, 0.95f, 0.87f}, 1.f, 1.44f, .5f, .7f));
  Wall const walls[5] = {
      {{0, 4, 0}, {0, -1, 0}, makeOrenNayar(white, .5f)},             // far
      {{0, 2, -.5f}, {0, 0, 1}, makeOrenNayar(orange, .7f)},          // floor
      {{0, 2, 2}, {0, 0, -1}, makeOrenNayar(white, .5f)},             // ceiling
      {{-2, 2, 0}, {1, 0, 0}, makeOrenNayar({1.f, 0.01f, 0.01f}, .6f)},  // left
      {{2, 2, 0}, {-1, 0, 0}, makeOrenNayar({0.01f, 1.f, 0.01f}, .6f)},  // right
  };
  uint32_t mat = 2;
  for (Wall const& w : walls) {
    s.addModel(generatePlane(w.center, w.normal, 4, 4), mat++);
    s.bsdfs.push_back(w.bsdf);
  }
  s.lights.push_back(makeSpotLight({2.f, 2.f, 2.f}, {0, 1.8f, 1.7f}, {0, 0, -1}, cosf(PI / 6), cosf(PI / 3), 0.01f));
  s.infiniteLights.push_back(makeEnvironmentalLight({0.1f, 0.1f, 0.1f}));
  s.camera = defaultCamera();
  s.camera.width = 256, s.camera.height = 256, s.camera.spp = 4;
  return s;
}

Scene randomTriangleScene(size_t count, uint64_t seed, float extent) {
  Scene s = cornellBox();
  s.xs.clear(), s.ys.clear(), s.zs.clear(), s.matId.clear();
  uint64_t state = seed;
  auto next01 = [&state]() {  // splitmix64 -> [0,1) float with 24 random bits
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return float(z >> 40) * (1.0f / 16777216.0f);
  };
  s.xs.reserve(4 * count), s.ys.reserve(4 * count), s.zs.reserve(4 * count), s.matId.reserve(count);
  for (size_t i = 0; i < count; ++i) {
    // extent 1 = SURVEY 8(d)'s recipe: centroids uniform in the cube [-10,10] x [5,25] x [-10,10] in front of the camera
    Vec3 const c{(next01() * 20.f - 10.f) * extent, next01() * 20.f * extent + 5.f, (next01() * 20.f - 10.f) * extent};
    Vec3 v[3];
    for (Vec3& p : v) p = add(c, {next01() * 0.3f - 0.15f, next01() * 0.3f - 0.15f, next01() * 0.3f - 0.15f});
    s.xs.insert(s.xs.end(), {v[0].x, v[1].x, v[2].x, 0.f});
    s.ys.insert(s.ys.end(), {v[0].y, v[1].y, v[2].y, 0.f});
    s.zs.insert(s.zs.end(), {v[0].z, v[1].z, v[2].z, 0.f});
    s.matId.push_back(uint32_t(i % 7));
  }
  s.lights.clear();
  s.lights.push_back(makeSpotLight({2.f, 2.f, 2.f}, {0, 5.f + 10.f * extent, 10.f * extent + 2.f}, {0, 0, -1}, cosf(PI / 6), cosf(PI / 3), 0.01f));
  s.camera.width = 1024, s.camera.height = 1024;
  return s;
}

// ---- writers ------------------------------------------------------------------------------------
void filmToRgb8(float const* mean4, float const* m24, size_t n, uint8_t* meanRgb, uint8_t* stdErrRgb) {
  for (size_t i = 0; i < n; ++i) {
    float const count = m24[4 * i + 3];
    for (int c = 0; c < 3; ++c) {
      if (meanRgb) meanRgb[3 * i + c] = static_cast<uint8_t>(fminf(fmaxf(mean4[4 * i + c], 0.f) * 255.f, 255.f));
      if (stdErrRgb) {
        float const se = sqrtf(fmaxf(m24[4 * i + c], 0.f)) / count;
        stdErrRgb[3 * i + c] = static_cast<uint8_t>(fminf(fmaxf(se, 0.f) * 255.f, 255.f));
      }
    }
  }
}

bool writePngRgb8(std::string const& path, uint8_t const* rgb, uint32_t width, uint32_t height, std::string* error) {
  auto failWith = [&](char const* m) {
    if (error) *error = std::string(m) + ": " + path;
    return false;
  };
  // filter byte 0 (None) in front of every scanline, one zlib stream
  std::vector<uint8_t> raw;
  raw.reserve(size_t(height) * (size_t(width) * 3 + 1));
  for (uint32_t y = 0; y < height; ++y) {
    raw.push_back(0);
    raw.insert(raw.end(), rgb + size_t(y) * width * 3, rgb + size_t(y + 1) * width * 3);
  }
  uLongf zlen = compressBound(uLong(raw.size()));
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), uLong(raw.size()), 6) != Z_OK) return failWith("zlib compress failed");
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) return failWith("cannot open for writing");
  auto be32 = [](uint8_t* p, uint32_t v) { p[0] = uint8_t(v >> 24), p[1] = uint8_t(v >> 16), p[2] = uint8_t(v >> 8), p[3] = uint8_t(v); };
  auto chunk = [&](char const* tag, uint8_t const* data, uint32_t len) {
    uint8_t hdr[8];
    be32(hdr, len);
    memcpy(hdr + 4, tag, 4);
    uint32_t crc = uint32_t(crc32(0L, hdr + 4, 4));
    if (len) crc = uint32_t(crc32(crc, data, len));
    uint8_t tail[4];
    be32(tail, crc);
    return fwrite(hdr, 1, 8, fp) == 8 && (len == 0 || fwrite(data, 1, len, fp) == len) && fwrite(tail, 1, 4, fp) == 4;
  };
  static uint8_t const sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  uint8_t ihdr[13];
  be32(ihdr, width), be32(ihdr + 4, height);
  ihdr[8] = 8, ihdr[9] = 2, ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;  // 8-bit RGB
  bool ok = fwrite(sig, 1, 8, fp) == 8 && chunk("IHDR", ihdr, 13) && chunk("IDAT", z.data(), uint32_t(zlen)) &&
            chunk("IEND", nullptr, 0);
  ok = (fclose(fp) == 0) && ok;
  return ok ? true : failWith("short write");
}

bool writeMeanAndMSERowMajor(float const* mean4, float const* m24, uint32_t width, uint32_t height,
                             std::string const& baseName, std::string* error) {
  size_t const n = size_t(width) * height;
  std::vector<uint8_t> a(3 * n), b(3 * n);
  filmToRgb8(mean4, m24, n, a.data(), b.data());
  return writePngRgb8(baseName + ".png", a.data(), width, height, error) &&
         writePngRgb8(baseName + "_sqrt_mse.png", b.data(), width, height, error);
}

int uploadScene(dmt_ctx* ctx, Scene const& s) {
  int rc = dmt_upload_triangles(ctx, s.xs.data(), s.ys.data(), s.zs.data(), s.matId.data(), s.triangleCount());
  if (rc) return rc;
  rc = dmt_upload_bsdfs(ctx, s.bsdfs.data(), uint32_t(s.bsdfs.size()));
  if (rc) return rc;
  rc = dmt_upload_lights(ctx, s.lights.data(), uint32_t(s.lights.size()), s.infiniteLights.data(),
                         uint32_t(s.infiniteLights.size()));
  if (rc) return rc;
  rc = dmt_set_camera(ctx, &s.camera);
  if (rc) return rc;
  rc = dmt_upload_area_lights(ctx, s.areaTri.data(), s.areaLe.data(), uint32_t(s.areaTri.size()));
  if (rc) return rc;
  rc = s.envRgb.empty() ? dmt_clear_envmap(ctx) : dmt_upload_envmap(ctx, s.envRgb.data(), s.envWidth, s.envHeight, s.envQuat, s.envScale);
  if (rc) return rc;
  if (s.texDesc.empty())
    return dmt_upload_textures(ctx, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0);
  return dmt_upload_textures(ctx, s.texRgba.data(), s.texRgba.size() / 4, s.texDesc.data(), uint32_t(s.texDesc.size() / 3), s.matTex.data(),
                             uint32_t(s.matTex.size() / 4), s.triUv.data(), s.triUv.size() / 6);
}

}  // namespace dmt_host
