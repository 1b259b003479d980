// dmt_host_c.cpp -- flat C view of the host-side scene library (for ctypes / other FFIs).
#include <cstring>
#include <new>

#include "dmt_scene.hpp"

using namespace dmt_host;

extern "C" {

struct dmt_host_scene {
  Scene s;
};

dmt_host_scene* dmt_host_scene_cornell_box() {
  auto* h = new (std::nothrow) dmt_host_scene();
  if (h) h->s = cornellBox();
  return h;
}
dmt_host_scene* dmt_host_scene_random_triangles(uint64_t count, uint64_t seed) {
  auto* h = new (std::nothrow) dmt_host_scene();
  if (h) h->s = randomTriangleScene(size_t(count), seed);
  return h;
}
dmt_host_scene* dmt_host_scene_random_triangles_ex(uint64_t count, uint64_t seed, float extent) {
  if (!(extent > 0.f)) return nullptr;
  auto* h = new (std::nothrow) dmt_host_scene();
  if (h) h->s = randomTriangleScene(size_t(count), seed, extent);
  return h;
}
// JSON front-end; on failure returns null and copies the message (NUL-terminated, truncated) into err
dmt_host_scene* dmt_host_scene_load_json(const char* path, int* max_depth, int* samples_per_pixel, char* err, uint64_t err_cap) {
  auto* h = new (std::nothrow) dmt_host_scene();
  if (!h || !path) return delete h, nullptr;
  JsonScene js;
  std::string msg;
  if (!loadJsonScene(path, js, &msg)) {
    if (err && err_cap) {
      size_t const n = msg.size() < err_cap - 1 ? msg.size() : size_t(err_cap - 1);
      memcpy(err, msg.data(), n);
      err[n] = 0;
    }
    delete h;
    return nullptr;
  }
  h->s = std::move(js.scene);
  if (max_depth) *max_depth = js.maxDepth;
  if (samples_per_pixel) *samples_per_pixel = js.samplesPerPixel;
  return h;
}
// first mesh of a binary FBX file as a triangle list (9 floats per triangle); returns the triangle count, -1 on
// failure (message in err).  Call with out == null to get the count.
int64_t dmt_host_read_fbx(const char* path, float* out9, uint64_t cap_triangles, char* err, uint64_t err_cap) {
  std::vector<Triangle> tris;
  std::string msg;
  if (!path || !readFbxMesh(path, tris, &msg)) {
    if (err && err_cap) {
      size_t const n = msg.size() < err_cap - 1 ? msg.size() : size_t(err_cap - 1);
      memcpy(err, msg.data(), n);
      err[n] = 0;
    }
    return -1;
  }
  if (out9)
    for (size_t i = 0; i < tris.size() && i < cap_triangles; ++i) {
      Vec3 const v[3] = {tris[i].v0, tris[i].v1, tris[i].v2};
      for (int k = 0; k < 3; ++k) out9[9 * i + 3 * size_t(k)] = v[k].x, out9[9 * i + 3 * size_t(k) + 1] = v[k].y, out9[9 * i + 3 * size_t(k) + 2] = v[k].z;
    }
  return int64_t(tris.size());
}
// PBRT-v4 subset front-end; same protocol as dmt_host_scene_load_json
dmt_host_scene* dmt_host_scene_load_pbrt(const char* path, int* max_depth, int* samples_per_pixel, char* err, uint64_t err_cap) {
  auto* h = new (std::nothrow) dmt_host_scene();
  if (!h || !path) return delete h, nullptr;
  PbrtScene ps;
  std::string msg;
  if (!loadPbrtScene(path, ps, &msg)) {
    if (err && err_cap) {
      size_t const n = msg.size() < err_cap - 1 ? msg.size() : size_t(err_cap - 1);
      memcpy(err, msg.data(), n);
      err[n] = 0;
    }
    delete h;
    return nullptr;
  }
  h->s = std::move(ps.scene);
  if (max_depth) *max_depth = ps.maxDepth;
  if (samples_per_pixel) *samples_per_pixel = ps.samplesPerPixel;
  return h;
}
uint32_t dmt_host_scene_area_light_count(const dmt_host_scene* h) { return uint32_t(h->s.areaTri.size()); }
const uint32_t* dmt_host_scene_area_tri(const dmt_host_scene* h) { return h->s.areaTri.data(); }
const float* dmt_host_scene_area_le(const dmt_host_scene* h) { return h->s.areaLe.data(); }
const float* dmt_host_scene_env_rgb(const dmt_host_scene* h, int* width, int* height) {
  if (width) *width = h->s.envWidth;
  if (height) *height = h->s.envHeight;
  return h->s.envRgb.empty() ? nullptr : h->s.envRgb.data();
}
// image textures: counts, then the four arrays in dmt_upload_textures' layout
uint32_t dmt_host_scene_texture_count(const dmt_host_scene* h) { return uint32_t(h->s.texDesc.size() / 3); }
uint64_t dmt_host_scene_texel_count(const dmt_host_scene* h) { return uint64_t(h->s.texRgba.size() / 4); }
const uint8_t* dmt_host_scene_tex_rgba(const dmt_host_scene* h) { return h->s.texRgba.data(); }
const int32_t* dmt_host_scene_tex_desc(const dmt_host_scene* h) { return h->s.texDesc.data(); }
const uint32_t* dmt_host_scene_mat_tex(const dmt_host_scene* h) { return h->s.matTex.data(); }
const float* dmt_host_scene_tri_uv(const dmt_host_scene* h) { return h->s.triUv.data(); }
void dmt_host_scene_destroy(dmt_host_scene* h) { delete h; }

uint64_t dmt_host_scene_triangle_count(const dmt_host_scene* h) { return h->s.triangleCount(); }
uint32_t dmt_host_scene_bsdf_count(const dmt_host_scene* h) { return uint32_t(h->s.bsdfs.size()); }
uint32_t dmt_host_scene_light_count(const dmt_host_scene* h) { return uint32_t(h->s.lights.size()); }
uint32_t dmt_host_scene_infinite_light_count(const dmt_host_scene* h) { return uint32_t(h->s.infiniteLights.size()); }
const float* dmt_host_scene_xs(const dmt_host_scene* h) { return h->s.xs.data(); }
const float* dmt_host_scene_ys(const dmt_host_scene* h) { return h->s.ys.data(); }
const float* dmt_host_scene_zs(const dmt_host_scene* h) { return h->s.zs.data(); }
const uint32_t* dmt_host_scene_mat_ids(const dmt_host_scene* h) { return h->s.matId.data(); }
const void* dmt_host_scene_bsdfs(const dmt_host_scene* h) { return h->s.bsdfs.data(); }
const void* dmt_host_scene_lights(const dmt_host_scene* h) { return h->s.lights.data(); }
const void* dmt_host_scene_infinite_lights(const dmt_host_scene* h) { return h->s.infiniteLights.data(); }
dmt_camera* dmt_host_scene_camera(dmt_host_scene* h) { return &h->s.camera; }
void dmt_host_scene_set_resolution(dmt_host_scene* h, int width, int height) {
  h->s.camera.width = width, h->s.camera.height = height;
}
int dmt_host_scene_upload(const dmt_host_scene* h, dmt_ctx* ctx) { return uploadScene(ctx, h->s); }

// packers, one 32-byte record out
static void put(Packed32 const& p, void* out32) { memcpy(out32, p.bytes, 32); }
static Vec3 v(const float* p) { return {p[0], p[1], p[2]}; }
void dmt_host_make_lambert(void* out32) { put(makeLambert(), out32); }
void dmt_host_make_oren_nayar(const float* color3, float roughness, void* out32) { put(makeOrenNayar(v(color3), roughness), out32); }
void dmt_host_make_ggx_dielectric(const float* r3, const float* t3, float phi0, float eta, float ax, float ay, void* out32) {
  put(makeGGXDielectric(v(r3), v(t3), phi0, eta, ax, ay), out32);
}
void dmt_host_make_ggx_conductor(const float* eta3, const float* k3, float phi0, float ax, float ay, void* out32) {
  put(makeGGXConductor(v(eta3), v(k3), phi0, ax, ay), out32);
}
void dmt_host_make_point_light(const float* c3, const float* p3, float radius, void* out32) { put(makePointLight(v(c3), v(p3), radius), out32); }
void dmt_host_make_spot_light(const float* c3, const float* p3, const float* d3, float cos0, float cosE, float radius, void* out32) {
  put(makeSpotLight(v(c3), v(p3), v(d3), cos0, cosE, radius), out32);
}
void dmt_host_make_directional_light(const float* c3, const float* d3, float omc, void* out32) {
  put(makeDirectionalLight(v(c3), v(d3), omc), out32);
}
void dmt_host_make_environmental_light(const float* c3, void* out32) { put(makeEnvironmentalLight(v(c3)), out32); }
uint16_t dmt_host_float_to_half_bits(float f) { return float_to_half_bits(f); }
float dmt_host_half_bits_to_float(uint16_t h) { return half_bits_to_float(h); }

void dmt_host_film_to_rgb8(const float* mean4, const float* m24, uint64_t pixels, uint8_t* meanRgb, uint8_t* stdErrRgb) {
  filmToRgb8(mean4, m24, size_t(pixels), meanRgb, stdErrRgb);
}
int dmt_host_write_mean_and_mse(const float* mean4, const float* m24, uint32_t width, uint32_t height, const char* baseName) {
  return writeMeanAndMSERowMajor(mean4, m24, width, height, baseName) ? 0 : 1;
}

}  // extern "C"
