// dmt_scene.hpp -- host-side scene description above the C ABI (include/dmt_hip.h).
//
// Mirrors the reference's host interface for the megakernel path so that a caller of
// cornellBox / make* / generate* (CC/public/cuda-core/host_utils.cuh:259-261,
// CC/public/cuda-core/bsdf.cuh:94-101, CC/public/cuda-core/light.cuh:83-93,
// CC/public/cuda-core/host_scene.cuh:26-48; CC = examples/triangles/cuda-core) finds the same
// names, argument meaning and byte-identical packed records.  IEEE fp32 throughout.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dmt_hip.h"

namespace dmt_host {

struct Vec3 {
  float x = 0.f, y = 0.f, z = 0.f;
};

struct Packed32 {  // one BSDF or Light record as uploaded (32 bytes)
  uint8_t bytes[32] = {};
};

struct Triangle {
  Vec3 v0, v1, v2;
};

// storage codecs of the packed records (CC/private/encoding.cu)
uint16_t float_to_half_bits(float f);
float half_bits_to_float(uint16_t h);
uint32_t octaFromDir(Vec3 dir);
Vec3 dirFromOcta(uint32_t octa);

// BSDF packers (CC/private/bsdf.cu:669-717,817-844)
Packed32 makeLambert();
Packed32 makeOrenNayar(Vec3 color, float roughness);
Packed32 makeGGXDielectric(Vec3 reflectanceTint, Vec3 transmittanceTint, float phi0, float eta,
                           float alphax, float alphay);
Packed32 makeGGXConductor(Vec3 eta, Vec3 kappa, float phi0, float alphax, float alphay);
Packed32 makeGGXBlendDielectric(Vec3 reflectanceTint, Vec3 transmittanceTint, float phi0, float eta, float alphax, float alphay,
                                float metallic);
// light packers (CC/private/light.cu:271-307)
Packed32 makePointLight(Vec3 color, Vec3 position, float radius);
Packed32 makeSpotLight(Vec3 color, Vec3 position, Vec3 direction, float cosTheta0, float cosThetaE,
                       float radius);
Packed32 makeDirectionalLight(Vec3 color, Vec3 direction, float oneMinusCosAngle);
Packed32 makeEnvironmentalLight(Vec3 color);

// procedural meshes (CC/private/host_scene.cu:7-119)
std::vector<Triangle> generateSphereMesh(Vec3 center, float radius, int latSubdiv, int lonSubdiv);
std::vector<Triangle> generateCube(Vec3 center, Vec3 scale);
std::vector<Triangle> generatePlane(Vec3 center, Vec3 normal, float width, float height);

// HostTriangleScene + everything else the upload helpers take, already flattened to the
// TriangleSoup layout of triSoupFromTriangles (CC/private/host_utils.cu:118-187)
struct Scene {
  std::vector<float> xs, ys, zs;  // 4 floats per triangle {c0, c1, c2, 0}
  std::vector<uint32_t> matId;
  std::vector<Packed32> bsdfs, lights, infiniteLights;
  dmt_camera camera{};
  // optional env-map light (A18; the JSON front-end's "envlight"): RGB floats, envHeight x envWidth x 3
  std::vector<float> envRgb;
  int envWidth = 0, envHeight = 0;
  float envQuat[4] = {0.f, 0.f, 0.f, 1.f};  // lightFromRender, x y z w
  float envScale = 1.f;
  // optional emissive triangles (the PBRT front-end's AreaLightSource): indices into the arrays above
  // and rgb radiance per entry
  std::vector<uint32_t> areaTri;
  std::vector<float> areaLe;
  // optional image textures (the JSON front-end's "textures"), in dmt_upload_textures' layout: RGBA8 texels
  // of all textures back to back, {first texel, width, height} per texture, {diffuse, roughness, normal texture or
  // 0xFFFFFFFF, anisotropy float bits} per BSDF, {u0 v0 u1 v1 u2 v2} per triangle
  std::vector<uint8_t> texRgba;
  std::vector<int32_t> texDesc;
  std::vector<uint32_t> matTex;
  std::vector<float> triUv;

  size_t triangleCount() const { return matId.size(); }
  // addModel + the material walk of triSoupFromTriangles: the FIRST mesh always gets material 0
  void addModel(std::vector<Triangle> const& mesh, uint32_t materialIndex);

 private:
  bool firstMesh_ = true;
};

// the reference's only scene (CC/private/host_utils.cu:402-469): 26 triangles, 7 materials,
// one spot light, one constant environment; 256x256, camera at the origin looking down +y
Scene cornellBox();

// BASELINE config 4 / SURVEY 8d: deterministic random triangle soup in front of the Cornell camera
// (splitmix64 seed 0x5EED1234, centroids in [-10,10]^3 shifted to y in [5,25], edges in
// [-0.15,0.15]^3, material = index mod 7 over the Cornell BSDFs, spot light at (0,15,12), env 0.1)
// extent: edge length of the cube of centroids in units of the recipe's 20 (triangle size unchanged): count x extent^-3 is the density
Scene randomTriangleScene(size_t triangleCount, uint64_t seed = 0x5EED1234ull, float extent = 1.f);

// JSON scene front-end: the reference's scene description (src/core/private/core-parser.cpp:256-1455;
// keys camera / film / textures / materials / objects / lights / envlight / transforms / world) flattened to the
// megakernel's upload arrays.  Returns false and a message for anything the reference's parser rejects, and for
// what this path cannot represent yet (image textures, normal maps).
struct JsonScene {
  Scene scene;
  int maxDepth = 5;          // camera "max-depth" (core-types.h:30)
  int samplesPerPixel = 1;   // film "samples"   (core-types.h:28)
};
bool loadJsonScene(std::string const& path, JsonScene& out, std::string* error = nullptr);
// PBRT-v4 subset front-end (scenes/cornell-box.pbrt): Film, Sampler, LookAt, Camera "perspective",
// Attribute blocks, Translate / Rotate / Scale / Transform, named "diffuse" materials, "trianglemesh" shapes and
// "diffuse" area lights (host/dmt_pbrt_scene.cpp)
struct PbrtScene {
  Scene scene;
  int maxDepth = 5;         // pbrt's PathIntegrator default
  int samplesPerPixel = 16;  // pbrt's sampler default
};
bool loadPbrtScene(std::string const& path, PbrtScene& out, std::string* error = nullptr);
// binary FBX (Kaydara 7100+): first mesh, fan-triangulated, Model TRS and unit scale applied (host/dmt_fbx.cpp); uv6: optional,
// six floats per triangle from the mesh's first LayerElementUV (zeros when the file has none)
bool readFbxMesh(std::string const& path, std::vector<Triangle>& out, std::string* error = nullptr, std::vector<float>* uv6 = nullptr);
// 8-bit grey / RGB / RGBA non-interlaced PNG -> RGB floats, byte / 255 as the reference's loadImageAsRGB
// (core-parser.cpp:156-167)
bool readPngRgb(std::string const& path, std::vector<float>& rgb, int& width, int& height, std::string* error = nullptr);
// the same decoder, bytes as stored: channels = 1 (grey), 2, 3 or 4 per pixel
bool readPng8(std::string const& path, std::vector<uint8_t>& pixels, int& width, int& height, int& channels, std::string* error = nullptr);

// 8-bit images of the film, exactly as the reference's writers quantise them
// (CC/private/host_utils.cu:475-497): u8 = (uint8)min(max(v,0)*255, 255), linear, truncating;
// standard-error image = sqrt(M2)/N
void filmToRgb8(float const* mean4, float const* m24, size_t pixelCount, uint8_t* meanRgb,
                uint8_t* stdErrRgb);
// writes <base>.png and <base>_sqrt_mse.png (writeMeanAndMSERowMajor, host_utils.cu:246-269)
bool writeMeanAndMSERowMajor(float const* mean4, float const* m24, uint32_t width, uint32_t height,
                             std::string const& baseName, std::string* error = nullptr);
bool writePngRgb8(std::string const& path, uint8_t const* rgb, uint32_t width, uint32_t height,
                  std::string* error = nullptr);

// upload a Scene through the C ABI (triSoupFromTriangles + deviceBSDF + deviceLights +
// deviceCamera + allocateDeviceConstantMemory in the reference, megakernel/main.cu:110-117)
int uploadScene(dmt_ctx* ctx, Scene const& scene);

}  // namespace dmt_host
