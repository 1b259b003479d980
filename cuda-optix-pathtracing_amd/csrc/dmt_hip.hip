// dmt_hip.hip -- gfx950 megakernel + C ABI (include/dmt_hip.h).
//
// Kernel design (MI355X-first, not a translation of T/megakernel/megakernel.cu):
//   * work item = one 8x8-pixel tile x all samples of the pass; waves pull items from a global
//     atomic counter (persistent threads + work stealing) instead of a static grid-stride loop;
//   * one lane = one pixel; when a lane's path ends it folds the radiance into its Welford
//     registers and immediately regenerates the next sample of the same pixel, so a wave never
//     waits for its longest path (the reference reconverges the warp after every sample);
//   * everything the reference does between two ray casts (BSDF prepare, light sample, NEE
//     weight, BSDF sample, Russian roulette) is evaluated BEFORE the shadow ray is traced, so the
//     shadow ray of bounce k and the closest-hit ray of bounce k+1 go through ONE pass over the
//     triangle array: two independent Moeller-Trumbore chains per lane (ILP) and each triangle is
//     fetched once.  Radiance is still accumulated in the reference's order.
//   * brute-force mode keeps the reference's "loop over every triangle" semantics: the loop
//     index is wave-uniform, so triangle records arrive through scalar loads (s_load_dwordx4)
//     and live in SGPRs -- no LDS or VGPR traffic in the hot loop.
//   * film state (mean, M2, N) stays in registers for the whole item; one 32-byte read and one
//     32-byte write per pixel per pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dmt_hip.h"
#include "pt_device.hpp"
#include "bvh_device.hpp"
#include "envmap.hpp"
#include "light_tree.hpp"
#include "light_tree_ref.hpp"

using namespace dmt;

namespace {

struct RenderParams {
  SceneView scene;
  BvhView bvh;
  CameraXf cam;
  SamplerParams sp;
  float4* __restrict__ mean;
  float4* __restrict__ m2;
  uint32_t* __restrict__ counter;
  int width, height;
  int x0, y0, x1, y1;     // pixel region
  int tx0, ty0, rtx;      // tile grid of the region: origin (in tiles) and tiles per row
  uint32_t numItems;      // owned tiles << subShift
  uint32_t subShift;      // an owned 8x8 tile is scheduled as 1 << subShift row bands (1, 2 or 4)
  int rank, world;
  uint32_t sampleOffset, spp;
  uint32_t chunkSpp;       // samples per work item
  uint32_t numChunks;      // ceil(spp / chunkSpp)
  float* stage;            // staging slabs of finished samples: [wave][kSlabsPerWave][chunkSpp][64] float3
  uint32_t* link;          // [numChunks][numItems] hand-over word of (chunk, tile): 0, kFoldReady or slab + 1 (item_complete)
  uint32_t* slabBusy;      // [wave][kSlabsPerWave] 1 while a handed-over slab waits for its folder
  unsigned long long* schedDiag;  // kSchedDiagWords counters, accumulated over launches (dmt_sched_diag)
  int maxDepth;
  int shadeThreshold;         // BVH megakernel: shade when this many lanes of the wave have finished their rays (bvhShadeThreshold)
  EnvView env;                // A18 env map (w == 0: none); read by the *_env kernels only
  // SURVEY 8f-3 emissive triangles; read by the *_area kernels only
  uint32_t const* areaOf;     // [triCount] index into areaTri / areaLe, 0xFFFFFFFF = not emissive
  uint32_t const* areaTri;    // [areaCount] ORIGINAL triangle index
  float const* areaLe;        // [areaCount] rgb radiance
  uint32_t areaCount;
  // SURVEY 8f-1 image textures; read by the *_tex kernels only (layout: dmt_upload_textures)
  uint32_t const* texRgba;    // RGBA8 texels of every texture, back to back
  int32_t const* texDesc;     // [texture] {first texel, width, height}
  uint32_t const* matTex;     // [bsdf] {diffuse, roughness, normal texture or 0xFFFFFFFF, anisotropy as float bits}
  float const* triUv;         // [triangle] {u0, v0, u1, v1, u2, v2}
  LightTreeNode const* lightTree;  // light BVH over `lights` (light_tree.hpp); read by the *_ltree kernels only
  LightTreeRefNode const* lightTreeRef;  // the reference-semantics tree (light_tree_ref.hpp); read by the *_ltree2 kernels only
  unsigned long long* stats;  // stats build only: samples, closest rays, shadow rays, node visits, triangle tests, bounces
};

// Per-lane state.  A lane carries (a) the path it is currently extending and (b) at most one
// pending shadow ray.  The shadow ray normally belongs to the current path, but when a path ends
// with its last shadow ray still untraced the lane parks that sample's radiance in Lfin and starts
// the NEXT sample at once: the old shadow ray and the new camera ray share the next triangle pass,
// and the parked sample is finalised (in order) right after it.
// All kernels that trace paths take a RenderParams as their FIRST by-value argument, i.e. at offset 0 of
// the kernarg segment.  Device code never touches that parameter object directly: it reads the fields it
// needs, where it needs them, through the kernarg pointer behind an opaque barrier (`kargs`).  Left to
// itself the compiler hoists all ~90 argument dwords to the top of the kernel and keeps them in SGPRs
// across the triangle loop; that overflows the 102-SGPR file and the spill code lands INSIDE the hot
// loop (measured: 97 ms -> 160 ms per launch).  An s_load from the scalar cache at the point of use
// costs nothing next to the ~2-3k instructions of a shading step.
typedef RenderParams const DMT_CONST_AS* KArgs;
DMT_DEV KArgs kargs_base() { return (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); }
DMT_DEV KArgs kargs(KArgs p) {
  asm volatile("" : "+s"(p));
  return p;
}
DMT_DEV SceneView load_scene(KArgs k) {
  k = kargs(k);
  SceneView s;
  s.tris = k->scene.tris, s.post = k->scene.post, s.bsdfs = k->scene.bsdfs, s.lights = k->scene.lights;
  s.infLights = k->scene.infLights, s.triCount = k->scene.triCount, s.bsdfCount = k->scene.bsdfCount;
  s.lightCount = k->scene.lightCount, s.infLightCount = k->scene.infLightCount;
  return s;
}
DMT_DEV BvhView load_bvh(KArgs k) {
  k = kargs(k);
  BvhView b;
  b.nodes = k->bvh.nodes, b.pairs = k->bvh.pairs, b.overflow = k->bvh.overflow, b.overflowStride = k->bvh.overflowStride;
  return b;
}

DMT_DEV EnvView load_env(KArgs k) {
  k = kargs(k);
  EnvView e;
  e.func = k->env.func, e.cdf = k->env.cdf, e.rowInt = k->env.rowInt, e.mFunc = k->env.mFunc, e.mCdf = k->env.mCdf;
  e.rgb = k->env.rgb, e.mInt = k->env.mInt, e.w = k->env.w, e.h = k->env.h;
  e.qx = k->env.qx, e.qy = k->env.qy, e.qz = k->env.qz, e.qw = k->env.qw;
  return e;
}

// Diagnostic build only (make variant DEFS=-DDMT_SECTION_TIMING=1): wave-level shader-clock cycles spent in each section of
// the brute-force megakernel, summed over all waves; read back with dmt_diag_section_cycles (tools/diag_sections.py).
#ifndef DMT_SECTION_TIMING
#define DMT_SECTION_TIMING 0
#endif
#if DMT_SECTION_TIMING
__device__ unsigned long long g_sect[16];
__shared__ unsigned long long s_sectLast[kLdsThreads / 64];
__shared__ unsigned long long s_sectAcc[kLdsThreads / 64][16];
#endif
DMT_DEV void sect_mark(int i) {  // everything since the previous mark belongs to section i
#if DMT_SECTION_TIMING
  unsigned long long const now = __builtin_readcyclecounter();
  unsigned long long const m = __ballot(1);
  if (int(__ffsll((long long)m)) - 1 == int(threadIdx.x & 63u)) {
    uint32_t const w = threadIdx.x >> 6;
    s_sectAcc[w][i] += now - s_sectLast[w];
    s_sectLast[w] = now;
  }
#endif
}
struct PathState {
  RayPair rp;       // .x = current path's ray, .y = pending shadow ray
  f3 beta, L;
  int depth;
  bool lastT;
  bool active;      // has a closest-hit ray to trace
  bool hasShadow;   // shadow ray / smax / C valid
  bool finPending;  // the pending shadow ray belongs to the finished sample parked in Lfin
  float smax;
  Sampler rng;
  uint32_t sidx;    // where the current sample's radiance goes (staging index, megakernel only)
  float lastPdf;    // env-map kernels only: pdf / delta flag of the bounce that produced the current ray
  bool lastSpecular;
};
// cold per-lane values in LDS, [field][thread]: pending NEE contribution C (0..2) and the parked
// radiance Lfin of a finished sample (3..5) with its staging index (6); each is touched once per ray pass at most
__shared__ float s_cold[7 * kLdsThreads];
DMT_DEV void put_C(f3 v) {
  float* const c = s_cold + threadIdx.x;
  c[0 * kLdsThreads] = v.x, c[1 * kLdsThreads] = v.y, c[2 * kLdsThreads] = v.z;
}
DMT_DEV f3 get_C() {
  float const* const c = s_cold + threadIdx.x;
  return mk3(c[0 * kLdsThreads], c[1 * kLdsThreads], c[2 * kLdsThreads]);
}
DMT_DEV void put_Lfin(f3 v) {
  float* const c = s_cold + threadIdx.x;
  c[3 * kLdsThreads] = v.x, c[4 * kLdsThreads] = v.y, c[5 * kLdsThreads] = v.z;
}
DMT_DEV f3 get_Lfin() {
  float const* const c = s_cold + threadIdx.x;
  return mk3(c[3 * kLdsThreads], c[4 * kLdsThreads], c[5 * kLdsThreads]);
}
DMT_DEV void put_finIdx(uint32_t i) { s_cold[6 * kLdsThreads + threadIdx.x] = __uint_as_float(i); }
DMT_DEV uint32_t get_finIdx() { return __float_as_uint(s_cold[6 * kLdsThreads + threadIdx.x]); }
DMT_DEV f3 ray_dir(PathState const& st) { return mk3(st.rp.dx.x, st.rp.dy.x, st.rp.dz.x); }
DMT_DEV f3 ray_org(PathState const& st) { return mk3(st.rp.ox.x, st.rp.oy.x, st.rp.oz.x); }
DMT_DEV void swap_rays(PathState& st) {  // path ray <-> pending shadow ray (megakernel_body_bvh)
  auto sw = [](v2f& p) { float const t = p.x; p.x = p.y, p.y = t; };
  sw(st.rp.ox), sw(st.rp.oy), sw(st.rp.oz), sw(st.rp.dx), sw(st.rp.dy), sw(st.rp.dz);
}
DMT_DEV void set_ray(PathState& st, f3 o, f3 d) {
  st.rp.ox.x = o.x, st.rp.oy.x = o.y, st.rp.oz.x = o.z;
  st.rp.dx.x = d.x, st.rp.dy.x = d.y, st.rp.dz.x = d.z;
}
DMT_DEV void set_shadow_ray(PathState& st, f3 o, f3 d) {
  st.rp.ox.y = o.x, st.rp.oy.y = o.y, st.rp.oz.y = o.z;
  st.rp.dx.y = d.x, st.rp.dy.y = d.y, st.rp.dz.y = d.z;
}

DMT_DEV void path_begin(PathState& st, CameraXf const& cam, SamplerParams const& sp, int px, int py,
                        int32_t pixBase, uint32_t s) {
  int32_t const hidx = pixBase + int32_t(s) * (sp.scale0 * sp.scale1);
  st.rng.start(uint32_t(hidx));
  Ray const r = camera_ray(cam, sp, px, py, hidx);
  set_ray(st, r.o, r.d);
  st.beta = mk3(1, 1, 1);
  st.L = mk3(0, 0, 0);
  st.depth = 0;
  st.lastT = false;
  st.active = true;
}

// Everything between two ray casts (T/megakernel/megakernel.cu:135-295).  Returns true when the
// path ends.  May leave a pending shadow ray (st.hasShadow) whose contribution st.C is added once
// visibility is known.
// ---- emissive triangles.  No reference implementation exists; semantics are pbrt-v4's, which the
// reference's scenes/cornell-box.pbrt is written for: DiffuseAreaLight, one-sided on n = normalize(cross(p1 - p0,
// p2 - p0)); uniform point sampling (SampleUniformTriangle); light chosen uniformly among [point/spot lights...,
// emissive triangles...]; power-heuristic MIS between light and BSDF sampling; emission seen directly by camera rays
// and after delta bounces.
struct AreaSampleDev {
  f3 wi;
  float dist, pdf;
  bool ok;
};
DMT_DEV AreaSampleDev area_sample(TriPost const& P, f3 p, f2 u) {
  AreaSampleDev r;
  r.ok = false, r.dist = 0.f, r.pdf = 0.f, r.wi = mk3(0, 0, 0);
  f3 const p0 = mk3(P.p0x, P.p0y, P.p0z), p1 = mk3(P.p1x, P.p1y, P.p1z), p2 = mk3(P.p2x, P.p2y, P.p2z);
  float b0, b1;
  if (u.x < u.y) {
    b0 = u.x / 2;
    b1 = u.y - b0;
  } else {
    b1 = u.y / 2;
    b0 = u.x - b1;
  }
  f3 const q = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
  f3 const c = cross(p1 - p0, p2 - p0);
  float const len = sqrtf(dot(c, c));
  if (!(len > 0.f)) return r;
  f3 const d = q - p;
  float const d2 = dot(d, d);
  if (!(d2 > 0.f)) return r;
  r.dist = sqrtf(d2);
  r.wi = d / r.dist;
  float const cosL = -dot(c / len, r.wi);
  if (!(cosL > 0.f)) return r;  // one-sided
  r.pdf = d2 / (cosL * (0.5f * len));
  r.ok = true;
  return r;
}
DMT_DEV float area_pdf(TriPost const& P, f3 rayD, float t) {
  f3 const p0 = mk3(P.p0x, P.p0y, P.p0z), p1 = mk3(P.p1x, P.p1y, P.p1z), p2 = mk3(P.p2x, P.p2y, P.p2z);
  f3 const c = cross(p1 - p0, p2 - p0);
  float const len = sqrtf(dot(c, c));
  if (!(len > 0.f)) return 0.f;
  float const cosL = -dot(c / len, rayD);
  if (!(cosL > 0.f)) return 0.f;
  return (t * t) / (cosL * (0.5f * len));
}

// ---- image textures of JSON materials.  The reference's megakernel path has none; semantics follow its CPU
// renderer (src/core/private/core-material.cpp:20-56,180-240; core-texture.cu:895-915): bilinear lookup at MIP level 0
// (this path carries no ray differentials, and the reference's isotropic fallback picks level 0 when its differentials
// vanish), mirror wrap, byte / 255, normal maps through Frame::fromZ(ng) after 10-bit quantisation.  The sampled albedo
// / roughness PATCH the packed record exactly as the host packers would have built it (makeOrenNayar, ggxCommon), so a
// textured and an untextured material take the same route through bsdf_prepare.
DMT_DEV f3 tex_texel(uint32_t const* rgba, int32_t first, int32_t w, int32_t h, int s, int t) {
  auto mirror = [](int c, int size) {
    int const p = size * 2;
    c %= p;
    if (c < 0) c += p;
    return c < size ? c : (p - c - 1);
  };
  uint32_t const px = rgba[size_t(first) + size_t(mirror(t, h)) * size_t(w) + size_t(mirror(s, w))];
  return mk3(float(px & 0xFFu) / 255.f, float((px >> 8) & 0xFFu) / 255.f, float((px >> 16) & 0xFFu) / 255.f);
}
DMT_DEV f3 tex_bilinear(KArgs k, int32_t tex, float s, float t, bool isNormal) {
  KArgs const ka = kargs(k);
  int32_t const* const d = ka->texDesc + 3 * tex;
  int32_t const first = d[0], w = d[1], h = d[2];
  uint32_t const* const rgba = ka->texRgba;
  float const x = s * float(w) - 0.5f, y = t * float(h) - 0.5f;
  float const fx = floorf(x), fy = floorf(y);
  int const x0 = int(fx), y0 = int(fy);
  float const tx = x - fx, ty = y - fy;
  f3 const c00 = tex_texel(rgba, first, w, h, x0, y0), c10 = tex_texel(rgba, first, w, h, x0 + 1, y0);
  f3 const c01 = tex_texel(rgba, first, w, h, x0, y0 + 1), c11 = tex_texel(rgba, first, w, h, x0 + 1, y0 + 1);
  f3 const cx0 = c00 * (1.f - tx) + c10 * tx, cx1 = c01 * (1.f - tx) + c11 * tx;
  f3 c = cx0 * (1.f - ty) + cx1 * ty;
  if (isNormal) c.x = c.x * 2.f - 1.f, c.y = c.y * 2.f - 1.f;
  return c;
}
// metallic fraction of a BS_GGX_BLEND material at the hit: the record's constant, or the material's 1-channel metallic map
// (core-material.cpp:209-216), whose index sits in the first texture slot of the pair's SECOND row
DMT_DEV float blend_metallic(KArgs k, Rec32 const& rec, uint32_t matId, int tri, float bu, float bv) {
  KArgs const ka = kargs(k);
  float m = h2f(lo16(rec.w[0]));
  if (ka->matTex != nullptr) {
    int32_t const texM = int32_t(ka->matTex[4 * (matId + 1u)]);
    if (texM >= 0) {
      float const* const uv = ka->triUv + 6 * size_t(tri);
      float const w0 = 1.f - bu - bv;
      m = tex_bilinear(k, texM, w0 * uv[0] + bu * uv[2] + bv * uv[4], w0 * uv[1] + bu * uv[3] + bv * uv[5], false).x;
    }
  }
  return m;
}
// patches `rec` from the material's textures at the hit and returns the shading normal (ng when there is no normal map)
DMT_DEV f3 apply_material_textures(KArgs k, Rec32& rec, uint32_t matId, int tri, float bu, float bv, f3 ng) {
  KArgs const ka = kargs(k);
  uint32_t const* const m = ka->matTex + 4 * matId;
  int32_t const texD = int32_t(m[0]), texR = int32_t(m[1]), texN = int32_t(m[2]);
  if (texD < 0 && texR < 0 && texN < 0) return ng;
  float const aniso = __uint_as_float(m[3]);
  float const* const uv = ka->triUv + 6 * size_t(tri);
  float const w0 = 1.f - bu - bv;
  float const s = w0 * uv[0] + bu * uv[2] + bv * uv[4], t = w0 * uv[1] + bu * uv[3] + bv * uv[5];
  uint32_t const type = hi16(rec.w[1]);
  if (texD >= 0 && type == BS_OREN) {
    f3 const c = tex_bilinear(k, texD, s, t, false);
    rec.w[0] = f2h(fmaxf(0.f, fminf(c.x, 1.f))) | (f2h(fmaxf(0.f, fminf(c.y, 1.f))) << 16);
    rec.w[1] = (rec.w[1] & 0xFFFF0000u) | f2h(fmaxf(0.f, fminf(c.z, 1.f)));
  }
  if (texR >= 0) {
    float const rough = fmaxf(0.f, fminf(tex_bilinear(k, texR, s, t, false).x, 1.f));
    if (type == BS_OREN) {  // makeOrenNayar, CC/private/bsdf.cu:817-844: terms derived from the STORED halves
      float const kk = (kPi / 2.f) - 2.f / 3.f;
      uint32_t const hr = f2h(fmaxf(0.f, fminf(rough, kPi / 2.f)));
      float const sigma = h2f(hr);
      uint32_t const ha = f2h(1.f / (kPi + kk * sigma));
      uint32_t const hb = f2h(h2f(ha) * sigma);
      rec.w[5] = hr | (ha << 16);
      rec.w[6] = (rec.w[6] & 0xFFFF0000u) | hb;
    } else if (type == BS_GGX_DIEL || type == BS_GGX_COND) {  // ggxCommon: alpha_y = roughness, alpha_x = anisotropy * roughness
      float const top = 65535.f;
      uint32_t const ax = uint32_t(fminf(fmaxf(aniso * rough * top, 0.f), top)), ay = uint32_t(fminf(fmaxf(rough * top, 0.f), top));
      rec.w[3] = (rec.w[3] & 0x0000FFFFu) | (ax << 16);
      rec.w[4] = (rec.w[4] & 0xFFFF0000u) | ay;
    }
  }
  if (texN < 0) return ng;
  f3 n = tex_bilinear(k, texN, s, t, true);
  auto quant = [](float v) { return float(int(v * 1023.f + 0.5f)) / 1023.f; };
  n = normalize(mk3(quant(n.x), quant(n.y), quant(n.z)));
  f3 tx, ty;
  gram_schmidt(ng, tx, ty);
  f3 const ns = tx * n.x + ty * n.y + ng * n.z;
  float const l2 = dot(ns, ns);
  return (l2 > 0.f && l2 < kInf) ? ns / sqrtf(l2) : ng;
}

DMT_DEV void trace_pair_brute(KArgs k, PathState const& st, bool doC, bool doS, int& bestTri, float& bu, float& bv, bool& occluded);

template <bool ENV = false, bool AREA = false, int TEX = false, int LTREE = false, bool BVH = false>
DMT_DEV bool path_shade(KArgs k, PathState& st, int bestTri, float bu, float bv) {
  SceneView const sc = load_scene(k);
  int const maxDepth = kargs(k)->maxDepth;
  if constexpr (ENV) {
    if (bestTri < 0) {  // A18: the env map seen by a path ray, MIS against NEE (core-render.cpp:154-163)
      EnvView const env = load_env(k);
      float pdfLight = 0.f;
      f3 const Le = env_eval_dir(env, ray_dir(st), pdfLight);
      if (st.depth == 0 || st.lastSpecular)
        st.L = st.L + st.beta * Le;
      else
        st.L = st.L + st.beta * (st.lastPdf / (st.lastPdf + pdfLight)) * Le;
      return true;
    }
  }
  if (bestTri < 0) {  // miss: constant environment, no MIS (megakernel.cu:135-151)
    if (sc.infLightCount > 0) {
      uint32_t const li = pick_index(st.rng.get1D(), sc.infLightCount);
      Rec32 const light = sc.infLights[li];
      float const pmf = 1.f / float(sc.infLightCount);
      if (light_type(light) == LT_ENV) st.L = st.L + st.beta * light_intensity(light) / pmf;
    }
    sect_mark(4);
    return true;
  }
  f3 const rd = ray_dir(st);
  Hit const hit = hit_finish(sc.post[bestTri], bu, bv, rd);
  uint32_t nAll = sc.lightCount;  // lights the NEE chooses among
  if constexpr (AREA) {
    KArgs const ka = kargs(k);
    nAll += ka->areaCount;
    uint32_t const ai = ka->areaOf[bestTri];
    if (ai != 0xFFFFFFFFu) {  // emitted radiance of the surface the path ray hit
      f3 const o = mk3(st.rp.ox.x, st.rp.oy.x, st.rp.oz.x);
      float const pl = area_pdf(sc.post[bestTri], rd, dot(hit.pos - o, rd));
      if (pl > 0.f) {
        f3 const Le = mk3(ka->areaLe[3 * ai], ka->areaLe[3 * ai + 1], ka->areaLe[3 * ai + 2]);
        if (st.depth == 0 || st.lastSpecular) {
          st.L = st.L + st.beta * Le;
        } else {
          float const a = st.lastPdf, b = pl * (ENV ? 0.5f : 1.f) / float(nAll);
          st.L = st.L + st.beta * Le * ((a * a) / (a * a + b * b));
        }
      }
    }
  }
  sect_mark(4);
  if (st.depth >= maxDepth) return true;  // :154-158

  f3 const wo = -rd;
  Rec32 rec = sc.bsdfs[hit.matId];
  f3 ns = hit.normal;  // shading normal: the geometric one unless a normal map says otherwise
  // fractional "metallic" (BS_GGX_BLEND, JSON scenes): both lobes of the material are prepared, evaluated and sampled and the
  // results blended as the reference's CPU renderer does (core-material.cpp:275-286, :383-394).  TEX instantiations only.
  Rec32 rec2{};
  float mix = 0.f;
  bool blend = false;
  if constexpr (TEX >= 2) {
    if (hi16(rec.w[1]) == BS_GGX_BLEND) {
      mix = blend_metallic(k, rec, hit.matId, bestTri, bu, bv);
      rec.w[1] = (rec.w[1] & 0x0000FFFFu) | (uint32_t(BS_GGX_DIEL) << 16);
      rec2 = sc.bsdfs[hit.matId + 1u];
      if (mix >= 1.f) rec = rec2;  // :273  the conductor alone
      else blend = mix > 0.f;      // :272  metallic <= 0: the dielectric alone
    }
  }
  if constexpr (TEX) {
    if (TEX < 2 || kargs(k)->matTex != nullptr) {
      ns = apply_material_textures(k, rec, hit.matId, bestTri, bu, bv, hit.normal);
      if (blend) (void)apply_material_textures(k, rec2, hit.matId + 1u, bestTri, bu, bv, hit.normal);  // same roughness map
    }
  }
#if DMT_SECTION_TIMING
  {  // material mix of the lanes that shade in this pass: [11] passes with a GGX lane, [12] GGX lanes, [13] passes, [14] lanes
    bool const ggx = hi16(rec.w[1]) != BS_OREN;
    unsigned long long const all = __ballot(1), mg = __ballot(ggx);
    if (int(__ffsll((long long)all)) - 1 == int(threadIdx.x & 63u)) {
      unsigned long long* const acc = s_sectAcc[threadIdx.x >> 6];
      acc[11] += mg ? 1u : 0u, acc[12] += uint32_t(__popcll(mg)), acc[13] += 1u, acc[14] += uint32_t(__popcll(all));
    }
  }
#endif
  Bsdf const b = bsdf_prepare(rec, ns, wo);  // :165-166
  Bsdf b2{};
  if constexpr (TEX >= 2) {
    if (blend) b2 = bsdf_prepare(rec2, ns, wo);
  }
  // f * weight and pdf of the material towards wi.  Blend: f = lerp(fD, fC, metallic), the RGB overload (a, b, t) of
  // cudautils-color.cuh:112-114; pdf = lerp(pdfD, pdfC, metallic) resolves to the FLOAT overload dmt::lerp(float x, float a,
  // float b) = (1 - x) a + x b (cudautils-vecmath.cuh:750-752), i.e. the reference computes (1 - pdfD) pdfC + pdfD metallic.
  // Kept as written, on the sampling side too (core-material.cpp:282).
  auto blend_pdf = [&](float pdfD, float pdfC) { return (1.f - pdfD) * pdfC + pdfD * mix; };
  auto eval_material = [&](f3 wi, float& pdf) {
    f3 f = eval_bsdf(b, wo, wi, ns, hit.normal, pdf) * b.weight;
    if constexpr (TEX >= 2) {
      if (blend) {
        float pdfC = 0.f;
        f3 const fC = eval_bsdf(b2, wo, wi, ns, hit.normal, pdfC) * b2.weight;
        f = f * (1.f - mix) + fC * mix;
        pdf = blend_pdf(pdf, pdfC);
      }
    }
    return f;
  };
  sect_mark(5);

  // next-event estimation (:170-241)
  float uLight = st.rng.get1D();
  f2 const uLight2 = st.rng.get2D();
  bool envNee = false;
  if constexpr (ENV) {  // A18: env map with probability 1/2, the light list otherwise (core-render.cpp:290-299)
    envNee = uLight < 0.5f;
    uLight = envNee ? uLight : (uLight - 0.5f) * 2.f;
    if (envNee) {
      EnvView const env = load_env(k);
      EnvSampleDev const es = env_sample(env, uLight2);
      if (es.ok) {
        float bsdfPdf = 0.f;
        f3 const f = eval_material(es.wi, bsdfPdf);
        f3 const Le = env_eval_uv(env, es.uv);
        if (!is_zero(f) && max3(Le) > 0.f) {  // core-render.cpp:357-369: Le f / (pdfLight pmf + pdfBsdf), pmf = 1/2
          put_C(st.beta * (Le * f / (es.pdf * 0.5f + bsdfPdf)));
          set_shadow_ray(st, offset_ray_origin(hit.pos, hit.error, hit.normal, es.wi), es.wi);
          st.smax = kInf;
          st.hasShadow = true;
        }
      }
    }
  }
  bool areaNee = false;
  if constexpr (AREA) {
    uint32_t const li = pick_index(uLight, nAll);
    areaNee = !envNee && li >= sc.lightCount;
    if (areaNee) {
      KArgs const ka = kargs(k);
      uint32_t const ai = li - sc.lightCount;
      AreaSampleDev const as = area_sample(sc.post[ka->areaTri[ai]], hit.pos, uLight2);
      if (as.ok) {
        float bsdfPdf = 0.f;
        f3 const f = eval_material(as.wi, bsdfPdf);
        if (!is_zero(f)) {
          f3 const Le = mk3(ka->areaLe[3 * ai], ka->areaLe[3 * ai + 1], ka->areaLe[3 * ai + 2]);
          float const a = as.pdf * (ENV ? 0.5f : 1.f) / float(nAll), bb = bsdfPdf;
          put_C(st.beta * (Le * f * (((a * a) / (a * a + bb * bb)) / a)));
          set_shadow_ray(st, offset_ray_origin(hit.pos, hit.error, hit.normal, as.wi), as.wi);
          st.smax = as.dist * 0.999f;
          st.hasShadow = true;
        }
      }
    }
  }
  bool treeNee = false;
  if constexpr (LTREE == 2) {
    // The reference's light tree with its own semantics (light_tree_ref.hpp): a cut of up to FOUR tree nodes, one light drawn
    // below each, one shadow ray per light (core-render.cpp:296-370).  This loop carries one pending shadow ray per lane, so
    // all but the LAST contributing light are tested for visibility right here (a whole any-hit traversal per ray; the
    // lane's traversal stack is free while it shades) and added in the reference's order; the last one rides with the next
    // closest-hit ray as usual.  Opt-in mode: the divergence of these in-line traversals is its price.
    treeNee = !envNee && sc.lightCount > 1;
    if (treeNee) {
      LightTreeRefSelection const sel = ltr_select(kargs(k)->lightTreeRef, hit.pos.x, hit.pos.y, hit.pos.z, hit.normal.x, hit.normal.y,
                                                   hit.normal.z, uLight, ENV ? 0.5f : 1.f);
      bool pending = false;
      f3 pendC = mk3(0, 0, 0), pendO = mk3(0, 0, 0), pendD = mk3(0, 0, 0);
      float pendMax = 0.f;
      for (uint32_t i = 0; i < sel.count; ++i) {
        Rec32 const light = sc.lights[sel.indices[i]];
        float const pmf = sel.pmfs[i];
        LightSample const ls = sample_light(light, hit.pos, uLight2, st.lastT, hit.normal);
        if (!ls.valid()) continue;
        float bsdfPdf = 0.f;
        f3 const f = eval_material(ls.direction, bsdfPdf);
        if (is_zero(f)) continue;
        f3 const Le = eval_light(light, ls);
        f3 C;
        if (ls.delta) {
          C = st.beta * Le * f / pmf;
        } else {
          float const w = sqr(pmf * ls.pdf) / sqr(pmf * ls.pdf + bsdfPdf);
          C = Le * f * st.beta * w;
        }
        if (pending) {  // an earlier light is waiting: resolve it now, keep this one pending
          bool occ;
          if constexpr (BVH) {
            occ = bvh_any(load_bvh(k), true, pendO, pendD, pendMax, blockIdx.x * blockDim.x + threadIdx.x);
          } else {
            PathState tmp = st;
            set_shadow_ray(tmp, pendO, pendD);
            tmp.smax = pendMax;
            int bt;
            float tu, tvv;
            trace_pair_brute(k, tmp, false, true, bt, tu, tvv, occ);
          }
          if (!occ) st.L = st.L + pendC;
        }
        pending = true, pendC = C, pendMax = ls.distance;
        pendO = offset_ray_origin(hit.pos, hit.error, hit.normal, ls.direction), pendD = ls.direction;
      }
      if (pending) {
        put_C(pendC);
        set_shadow_ray(st, pendO, pendD);
        st.smax = pendMax;
        st.hasShadow = true;
      }
    }
  }
  if (!envNee && !areaNee && !treeNee && sc.lightCount > 0) {
    uint32_t li = 0;
    float pmf = 0.f;
    bool picked = true;
    if constexpr (LTREE == 1) {  // importance-driven choice (light_tree.hpp) instead of the uniform pick
      float treePmf = 0.f;
      int const sel = lt_select(kargs(k)->lightTree, hit.pos.x, hit.pos.y, hit.pos.z, hit.normal.x, hit.normal.y, hit.normal.z, uLight, treePmf);
      picked = sel >= 0;
      li = picked ? uint32_t(sel) : 0u;
      pmf = (ENV ? 0.5f : 1.f) * treePmf;
    } else {
      li = pick_index(uLight, AREA ? nAll : sc.lightCount);
      pmf = (ENV ? 0.5f : 1.f) / float(AREA ? nAll : sc.lightCount);
    }
    Rec32 const light = sc.lights[li];
    LightSample const ls = sample_light(light, hit.pos, uLight2, st.lastT, hit.normal);
    sect_mark(6);
    if (picked && ls.valid()) {
      float bsdfPdf = 0.f;
      f3 const f = eval_material(ls.direction, bsdfPdf);
      if (!is_zero(f)) {
        f3 const Le = eval_light(light, ls);
        if (ls.delta) {
          put_C(st.beta * Le * f / pmf);
        } else {  // power heuristic, no division by the light pdf (:233-238)
          float const w = sqr(pmf * ls.pdf) / sqr(pmf * ls.pdf + bsdfPdf);
          put_C(Le * f * st.beta * w);
        }
        set_shadow_ray(st, offset_ray_origin(hit.pos, hit.error, hit.normal, ls.direction), ls.direction);
        st.smax = ls.distance;
        st.hasShadow = true;
      }
      // f == 0: the reference traces the shadow ray and then adds nothing; not traced here
    }
  }

  // bounce (:247-295); get2D before get1D = left-to-right argument evaluation
  sect_mark(7);
  f2 const u2 = st.rng.get2D();
  float const uc = st.rng.get1D();
  BsdfSample bs = sample_bsdf(b, wo, ns, hit.normal, u2, uc);
  if constexpr (TEX >= 2) {
    if (blend) {  // core-material.cpp:275-286: both lobes sampled with the same numbers; direction and flags of the conductor's
      BsdfSample sC = sample_bsdf(b2, wo, ns, hit.normal, u2, uc);
      sC.f = bs.f * (1.f - mix) + sC.f * mix;
      sC.pdf = blend_pdf(bs.pdf, sC.pdf);
      sC.eta = 1.f;
      bs = sC;
    }
  }
  sect_mark(8);
  if (!bs.valid()) return true;
  st.lastT = bs.refract;
  if constexpr (ENV || AREA) st.lastPdf = bs.pdf, st.lastSpecular = bs.delta;
  set_ray(st, offset_ray_origin(hit.pos, hit.error, hit.normal, bs.wi), bs.wi);
  st.beta = st.beta * (bs.f * fabsf(dot(bs.wi, hit.normal)) / bs.pdf);
  float const rrBeta = max3(st.beta * bs.eta);
  if (rrBeta < 1 && st.depth > 1) {
    float const q = fmaxf(0.f, 1.f - rrBeta);
    if (st.rng.get1D() < q) return true;
    st.beta = st.beta / (1 - q);
  }
  ++st.depth;
  sect_mark(9);
  return false;
}

DMT_DEV TriS load_tri(TriIsect const DMT_CONST_AS* tris, uint32_t i) {
  TriS T;
  T.p0x = tris[i].p0x, T.p0y = tris[i].p0y, T.p0z = tris[i].p0z;
  T.e0x = tris[i].e0x, T.e0y = tris[i].e0y, T.e0z = tris[i].e0z;
  T.e1x = tris[i].e1x, T.e1y = tris[i].e1y, T.e1z = tris[i].e1z;
  return T;
}

// One pass over the triangle array for the lane's ray pair: closest hit for .x, any-hit for .y.
// Brute force = the reference's semantics.  The loop index is wave-uniform and the array is read
// through the constant address space, so each triangle arrives by s_load into SGPRs, prefetched one
// triangle ahead in a two-register ping-pong (no SGPR copies), and the VALU work is packed fp32 over the
// two rays: no VGPR, LDS or vector-memory traffic in the loop.
struct BruteHit {
  float bt, bu, bv;
  int tri;
  bool occluded;
};
// nine named scalars per triangle (never a struct, see mt_core9)
#define DMT_TRI_DECL(P) float P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7, P##8
#define DMT_TRI_LOAD(P, idx)                                                                          \
  P##0 = tris[idx].p0x, P##1 = tris[idx].p0y, P##2 = tris[idx].p0z, P##3 = tris[idx].e0x, P##4 = tris[idx].e0y, \
  P##5 = tris[idx].e0z, P##6 = tris[idx].e1x, P##7 = tris[idx].e1y, P##8 = tris[idx].e1z
#define DMT_TRI_TEST(P, idx)                                                                                   \
  do {                                                                                                         \
    MTPair m;                                                                                                  \
    mt_core9<v2f>(P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7, P##8, st.rp.ox, st.rp.oy, st.rp.oz, st.rp.dx, \
                  st.rp.dy, st.rp.dz, m.det, m.t, m.u, m.v);                                                   \
    bool const v1 = mt_valid(m.det.x, m.t.x, m.u.x, m.v.x);                                                    \
    bool const v2 = mt_valid(m.det.y, m.t.y, m.u.y, m.v.y);                                                    \
    if (doC && v1 && m.t.x < h.bt) { /* strict <: lowest index wins ties (megakernel.cu:126) */                 \
      h.bt = m.t.x;                                                                                            \
      h.tri = int(idx);                                                                                        \
      h.bu = m.u.x;                                                                                            \
      h.bv = m.v.x;                                                                                            \
    }                                                                                                          \
    if (doS && v2 && m.t.y < st.smax) h.occluded = true; /* :210-211 */                                        \
  } while (0)

DMT_DEV void trace_pair_brute(KArgs k, PathState const& st, bool doC, bool doS, int& bestTri, float& bu,
                              float& bv, bool& occluded) {
  BruteHit h{kInf, 0.f, 0.f, -1, false};
  k = kargs(k);
  auto const* tris = to_const_as(k->scene.tris);
  uint32_t const n = k->scene.triCount;
  uint32_t const last = n ? n - 1 : 0;
  DMT_TRI_DECL(a);
  DMT_TRI_DECL(b);
  DMT_TRI_LOAD(a, 0);  // the array always holds >= 1 record (devAlloc)
  for (uint32_t i = 0; i < n;) {
    uint32_t const ib = i + 1 < last ? i + 1 : last;
    DMT_TRI_LOAD(b, ib);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch s_loads above the arithmetic
    DMT_TRI_TEST(a, i);
    if (++i >= n) break;
    uint32_t const ia = i + 1 < last ? i + 1 : last;
    DMT_TRI_LOAD(a, ia);
    __builtin_amdgcn_sched_barrier(0);
    DMT_TRI_TEST(b, i);
    ++i;
  }
  bestTri = h.tri, bu = h.bu, bv = h.bv, occluded = h.occluded;
}

struct LaneStats {  // stats build only
  uint32_t samples = 0, closest = 0, shadow = 0, bounces = 0;
  TraversalCounters tc;
  // loop profile of the BVH kernel: wave-level iterations (counted by every lane, /64 on the host) and the lanes
  // that did useful work in them
  uint32_t itNode = 0, itLeaf = 0, itShade = 0, itOuter = 0, itPrep = 0, lanesLeaf = 0, lanesShade = 0, lanesPrep = 0;
};
template <bool STATS = false>
DMT_DEV void trace_pair_bvh(KArgs k, PathState const& st, bool doC, bool doS, uint32_t gtid, int& bestTri,
                            float& bu, float& bv, bool& occluded, LaneStats* ls = nullptr) {
  BvhView const bvh = load_bvh(k);
  float bt;
  if constexpr (STATS) ls->closest += doC ? 1u : 0u, ls->shadow += doS ? 1u : 0u;
  bvh_closest<STATS>(bvh, doC, mk3(st.rp.ox.x, st.rp.oy.x, st.rp.oz.x), mk3(st.rp.dx.x, st.rp.dy.x, st.rp.dz.x),
                     gtid, bestTri, bt, bu, bv, STATS ? &ls->tc : nullptr);
  occluded = bvh_any<STATS>(bvh, doS, mk3(st.rp.ox.y, st.rp.oy.y, st.rp.oz.y),
                            mk3(st.rp.dx.y, st.rp.dy.y, st.rp.dz.y), st.smax, gtid, STATS ? &ls->tc : nullptr);
}

// One "ray pass" of a lane: trace (closest + pending shadow), resolve the shadow ray, shade.
// sink(L, sidx) is called once per completed sample with the index the sample was started with.
template <bool ENV = false, bool AREA = false, int TEX = false, int LTREE = false, bool BVH = false, class Sink>
DMT_DEV void lane_finish(KArgs k, PathState& st, bool doC, bool doS, int bestTri, float bu, float bv, bool occluded,
                         Sink&& sink);

template <bool BVH, bool STATS = false, bool ENV = false, bool AREA = false, int TEX = false, int LTREE = false, class Sink>
DMT_DEV void lane_step(KArgs k, uint32_t gtid, PathState& st, Sink&& sink, LaneStats* ls = nullptr) {
  bool const doC = st.active;
  bool const doS = st.hasShadow;
  int bestTri;
  float bu, bv;
  bool occluded;
  if constexpr (BVH)
    trace_pair_bvh<STATS>(k, st, doC, doS, gtid, bestTri, bu, bv, occluded, ls);
  else
    trace_pair_brute(k, st, doC, doS, bestTri, bu, bv, occluded);
  sect_mark(2);
  if constexpr (STATS) ls->bounces += (doC && bestTri >= 0 && st.depth < kargs(k)->maxDepth) ? 1u : 0u;
  lane_finish<ENV, AREA, TEX, LTREE, BVH>(k, st, doC, doS, bestTri, bu, bv, occluded, sink);
}

// Second half of a ray pass: resolve the shadow ray (in the reference's accumulation order), then shade.
template <bool ENV, bool AREA, int TEX, int LTREE, bool BVH, class Sink>
DMT_DEV void lane_finish(KArgs k, PathState& st, bool doC, bool doS, int bestTri, float bu, float bv, bool occluded,
                         Sink&& sink) {
  if (doS) {
    st.hasShadow = false;
    if (st.finPending) {  // the shadow ray of an already finished sample
      f3 Lfin = get_Lfin();
      if (!occluded) Lfin = Lfin + get_C();
      st.finPending = false;
      sink(Lfin, get_finIdx());
    } else if (!occluded) {
      st.L = st.L + get_C();  // NEE of the previous bounce, added before anything of this bounce
    }
  }
  sect_mark(3);
  if (doC) {
    if (path_shade<ENV, AREA, TEX, LTREE, BVH>(k, st, bestTri, bu, bv)) {
      st.active = false;
      if (st.hasShadow) {  // last NEE still untraced: park the sample, the lane may start the next
        put_Lfin(st.L);
        put_finIdx(st.sidx);
        st.finPending = true;
      } else {
        sink(st.L, st.sidx);
      }
    }
  }
  sect_mark(10);
}

// the lane's NEXT sample, prepared ahead of need (sampler values + camera ray), [field][thread]
__shared__ float s_prep[15 * kLdsThreads];  // 0-7 sampler values, 8-13 camera ray, 14 staging index

// Starting a sample costs ~2k instructions (8 scrambled radical inverses + camera ray).  Paths end at
// different times, so doing it on demand would run that code for a handful of lanes on almost every
// pass.  Instead every lane keeps its next sample PREPARED in LDS; a finished lane swaps it in (a few
// LDS moves) and the preparation of the following one is batched: it runs when at least half the wave
// needs one, or when some lane would otherwise starve.  Sample values are pure functions of
// (pixel, sample), so preparing early changes nothing.
// Cold kernel arguments (camera matrices, sampler parameters: 38 dwords) are only needed here, once per
// sample.  Left to itself the compiler hoists their kernarg loads to the top of the kernel and keeps
// them in SGPRs across the triangle loop, which overflows the SGPR file and spills INTO the hot loop.
// Reading them through an opaque copy of the kernarg pointer keeps the s_loads at the point of use.
struct ColdArgs {
  CameraXf cam;
  SamplerParams sp;
};
DMT_DEV ColdArgs load_cold_args(KArgs Pk) {
  Pk = kargs(Pk);
  ColdArgs c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.cam.cfr[i] = Pk->cam.cfr[i], c.cam.rfc[i] = Pk->cam.rfc[i];
  c.sp.scale0 = Pk->sp.scale0, c.sp.scale1 = Pk->sp.scale1, c.sp.exp0 = Pk->sp.exp0, c.sp.exp1 = Pk->sp.exp1;
  c.sp.inv0 = Pk->sp.inv0, c.sp.inv1 = Pk->sp.inv1;
  return c;
}

DMT_DEV void prepare_sample(KArgs Pk, int px, int py, int32_t pixBase, uint32_t s) {
  ColdArgs const cold = load_cold_args(Pk);
  CameraXf const& cam = cold.cam;
  SamplerParams const& sp = cold.sp;
  float* const prep = s_prep + threadIdx.x;
  int32_t const hidx = pixBase + int32_t(s) * (sp.scale0 * sp.scale1);
  sampler_values(uint32_t(hidx), prep);
  Ray const r = camera_ray(cam, sp, px, py, hidx);
  prep[8 * kLdsThreads] = r.o.x, prep[9 * kLdsThreads] = r.o.y, prep[10 * kLdsThreads] = r.o.z;
  prep[11 * kLdsThreads] = r.d.x, prep[12 * kLdsThreads] = r.d.y, prep[13 * kLdsThreads] = r.d.z;
}
DMT_DEV void path_begin_prepared(PathState& st) {
  float const* const prep = s_prep + threadIdx.x;
  float* const u = s_sampler_u + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k) u[k * kLdsThreads] = prep[k * kLdsThreads];
  set_ray(st, mk3(prep[8 * kLdsThreads], prep[9 * kLdsThreads], prep[10 * kLdsThreads]),
          mk3(prep[11 * kLdsThreads], prep[12 * kLdsThreads], prep[13 * kLdsThreads]));
  st.rng.dim = 2;
  st.beta = mk3(1, 1, 1);
  st.L = mk3(0, 0, 0);
  st.depth = 0;
  st.lastT = false;
  st.active = true;
  st.sidx = __float_as_uint(prep[14 * kLdsThreads]);
}

#ifndef DMT_MIN_WAVES_PER_SIMD
#define DMT_MIN_WAVES_PER_SIMD 4
#endif
#ifndef DMT_MIN_WAVES_PER_SIMD_BVH
#define DMT_MIN_WAVES_PER_SIMD_BVH 3
#endif
constexpr uint32_t kMaxChunkSpp = 512;  // staging: 384 KB per slab at most
#ifndef DMT_SLABS_PER_WAVE
#define DMT_SLABS_PER_WAVE 4
#endif
constexpr int kSlabsPerWave = DMT_SLABS_PER_WAVE;  // staging slabs per wave: two live items + two handed over
#ifndef DMT_PREP_THRESHOLD
#define DMT_PREP_THRESHOLD 64
#endif
struct TileArgs {  // what a wave needs when it picks up a new work item
  float4* mean;
  float4* m2;
  uint32_t* counter;
  int width, x0, y0, x1, y1, tx0, ty0, rtx;
  uint32_t numItems, subShift, sampleOffset, spp, chunkSpp, numChunks;
  uint32_t* link;
  uint32_t* slabBusy;
  float* stage;
  int rank, world;
};
DMT_DEV TileArgs load_tile_args(KArgs k) {
  k = kargs(k);
  TileArgs t;
  t.mean = k->mean, t.m2 = k->m2, t.counter = k->counter, t.width = k->width;
  t.x0 = k->x0, t.y0 = k->y0, t.x1 = k->x1, t.y1 = k->y1, t.tx0 = k->tx0, t.ty0 = k->ty0, t.rtx = k->rtx;
  t.numItems = k->numItems, t.subShift = k->subShift, t.sampleOffset = k->sampleOffset, t.spp = k->spp, t.rank = k->rank, t.world = k->world;
  t.chunkSpp = k->chunkSpp, t.numChunks = k->numChunks, t.link = k->link, t.slabBusy = k->slabBusy, t.stage = k->stage;
  return t;
}

// ---- work items ------------------------------------------------------------------------------------
// Work item = (sample chunk c, owned tile t), handed out chunk-major from one atomic counter: all tiles
// of chunk 0, then chunk 1, ...  An item is n samples x 64 pixels = up to 64 n UNITS (pixel, sample).
//
// * Lanes are not tied to pixels.  A lane that needs work takes the item's next unit (wave-wide ballot +
//   prefix count on a wave-uniform cursor; unit u -> pixel u mod 64, sample u div 64, so a batch of 64
//   requests is one sample of every pixel).  Path lengths differ between pixels (glass vs wall) and between
//   samples; with lane == pixel every item ran at the pace of its slowest pixel, now the wave stays full
//   until the item runs out of units.
// * A finished sample's radiance goes to one of the wave's kSlabsPerWave staging SLABS in global memory
//   ([sample][pixel] float3, written through so that any wave can read it back).  The film is the reference's
//   Welford update (SMEMLayout::updateSample, T/megakernel/megakernel.cuh:59-79) applied to a pixel's samples
//   IN INDEX ORDER, so chunk c of a tile must be folded after chunk c-1 -- while chunks c and c+1 are traced
//   CONCURRENTLY by different waves (a small frame, or one GPU's share of a frame split eight ways, still fills
//   the machine).  Nobody waits for that order; the fold is HANDED OVER instead (item_complete):
//     - every (chunk, tile) has one hand-over word `link`, zero at launch.  Exactly two parties touch it, each
//       with ONE atomic exchange: the wave that finishes tracing chunk c writes "slab + 1", the wave that has
//       put chunk c-1 into the film writes kFoldReady.  Exchanges on one word are totally ordered, so exactly
//       one of the two sees the other's value, and that one folds chunk c: the finisher if the predecessor was
//       already in the film, else the predecessor's folder, which then goes on to chunk c+1 the same way
//       (fold_chain).  Chunk 0 is folded by its finisher.
//     - a slab that was handed over stays busy until its folder clears `slabBusy`; its owner meanwhile uses
//       another of its slabs.
//   The film is therefore bit-identical for every chunk size and schedule, and no wave ever spins on another
//   wave's progress while it holds work: the only wait left is a wave with NO live item whose slabs are all
//   handed over and not folded yet (sched_retire) -- it holds nothing anybody needs, polls for a bounded wall
//   clock time and then EXITS (the remaining items are fetched by the other waves; counted in schedDiag).
//   Round 2's protocol had the finisher spin on a per-tile completion counter instead.  That is deadlock-free
//   only if every wave that has fetched an item keeps running; it gave up (error flag, 71 s per step) when four
//   processes' persistent kernels shared one GPU (DESIGN.md 7 has the record).
// * A wave holds up to TWO live items (sequence numbers cur and cur+1, LDS slots seq & 1): when item cur has
//   no units left, free lanes draw from item cur+1 (fetched at that moment) while the last paths of cur drain;
//   when cur's last sample is staged the wave completes it (lane i folds pixel i; whatever lane i is tracing
//   meanwhile stays in its registers).  Small items are therefore cheap, which keeps the end-of-launch tail
//   short when a frame is split over 8 GPUs.
__shared__ uint32_t s_desc[kLdsThreads / 64][2][8];  // per wave, per slot: chunk, tile item, px0, py0, s0, first float of the slab, nInside, slab
__shared__ int32_t s_pixbase[2 * kLdsThreads];       // per slot, per pixel: Halton pixel base, -1 = outside the region
__shared__ uint32_t s_pixmap[2 * kLdsThreads];       // per slot: j-th pixel inside the region

constexpr uint32_t kSlotBit = 0x80000000u;  // staging index = slot bit | (sample * 64 + pixel)
constexpr uint32_t kFoldReady = 0xFFFFFFFFu;  // link word: the tile's previous chunk is in the film
#ifndef DMT_SLAB_WAIT_MS
#define DMT_SLAB_WAIT_MS 250  // a wave without a live item and without a free slab polls this long, then exits
#endif
constexpr int kSchedDiagWords = 8;  // folds, handed over, folded for another wave, stalls, early exits, max stall ticks

// wave-level bookkeeping (all wave-uniform)
struct WaveSched {
  uint32_t cur = 0, fetched = 0;          // live items are [cur, fetched), at most two
  uint32_t alloc = 0;                     // item units are drawn from
  uint32_t nextUnit = 0, totalUnits = 0;  // cursor / size of item `alloc`
  bool exhausted = false;                 // the launch has no more items
  uint32_t slabFree = (1u << kSlabsPerWave) - 1u;  // own slabs that are free
  uint32_t slabPend = 0;                           // own slabs handed over, not known to be folded yet
};
// launch diagnostics (dmt_sched_diag; the host checks folds == items): one atomic per EVENT, from lane 0 -- an event is at
// most once per work item (~10^3 path samples), and counters kept in the wave's registers cost spills in the hot loop
enum { SD_FOLDS = 0, SD_HANDED = 1, SD_CHAINED = 2, SD_STALLS = 3, SD_EXITS = 4, SD_MAXSTALL = 5 };
DMT_DEV void sched_count(unsigned long long* D, int lane, int what, unsigned long long n = 1ull) {
  if (lane == 0) atomicAdd(&D[what], n);
}
// per-lane bookkeeping
struct LaneSched {
  bool prepared = false;  // a unit is waiting in s_prep
  bool prepSlot = false;  // ... of the item in this slot
};

// pixel origin and row count of a tile item (an owned tile is scheduled as 1 << subShift bands of 8 >> subShift
// rows: more, smaller items when this GPU has fewer tiles than resident waves; lanes beyond the band are "outside")
struct ItemGeom {
  int px0, py0;
  uint32_t rows;
};
DMT_DEV ItemGeom item_geom(TileArgs const& T, uint32_t item) {
  uint32_t const band = item & ((1u << T.subShift) - 1u);
  uint32_t const rows = 8u >> T.subShift;
  uint32_t const j = uint32_t(T.rank) + (item >> T.subShift) * uint32_t(T.world);
  return {(T.tx0 + int(j % uint32_t(T.rtx))) * 8, (T.ty0 + int(j / uint32_t(T.rtx))) * 8 + int(band * rows), rows};
}
DMT_DEV bool item_lane_inside(TileArgs const& T, ItemGeom const& g, int lane) {
  int const px = g.px0 + (lane & 7), py = g.py0 + (lane >> 3);
  return uint32_t(lane >> 3) < g.rows && px >= T.x0 && px < T.x1 && py >= T.y0 && py < T.y1;
}
DMT_DEV uint32_t chunk_samples(TileArgs const& T, uint32_t chunk) {
  uint32_t const s0 = chunk * T.chunkSpp;
  return s0 + T.chunkSpp < T.spp ? T.chunkSpp : T.spp - s0;
}

// index of this wave in the launch, as a value the compiler knows to be wave-uniform (gtid >> 6 lives in a VGPR; slab
// bookkeeping derived from it would be treated as divergent: VALU bit scans, exec-masked branches around item_fetch)
DMT_DEV uint32_t wave_index(uint32_t) { return blockIdx.x * (kLdsThreads / 64) + uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6))); }

// which of the wave's handed-over slabs have been folded meanwhile?
DMT_DEV void slab_refresh(TileArgs const& T, uint32_t gtid, int lane, WaveSched& W) {
  if (W.slabPend == 0u) return;
  bool const mine = lane < kSlabsPerWave && ((W.slabPend >> lane) & 1u) != 0u;
  uint32_t busy = 1u;
  if (mine) busy = __hip_atomic_load(&T.slabBusy[wave_index(gtid) * kSlabsPerWave + uint32_t(lane)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t const freed = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(__ballot(mine && busy == 0u)))));
  W.slabFree |= freed, W.slabPend &= ~freed;
}

// fetch the next work item into slot seq & 1; false = the launch has no more items.  The caller has checked that the
// wave has a free slab (W.slabFree != 0; slab_refresh runs in sched_retire only, to keep it out of the hot loop).
DMT_DEV bool item_fetch(KArgs Pk, uint32_t gtid, int lane, uint32_t seq, WaveSched& W, uint32_t& units) {
  TileArgs const T = load_tile_args(Pk);
  uint32_t work = 0;
  if (lane == 0) work = atomicAdd(T.counter, 1u);
  work = uint32_t(__builtin_amdgcn_readfirstlane(int(work)));
  if (work >= T.numItems * T.numChunks) return false;
  uint32_t const slabK = uint32_t(__builtin_ctz(W.slabFree));
  W.slabFree &= ~(1u << slabK);
  uint32_t const chunk = work / T.numItems;
  uint32_t const item = work - chunk * T.numItems;
  ItemGeom const g = item_geom(T, item);
  int const px = g.px0 + (lane & 7), py = g.py0 + (lane >> 3);
  bool const inside = item_lane_inside(T, g, lane);
  uint32_t const s0 = T.sampleOffset + chunk * T.chunkSpp;
  uint32_t const n = chunk_samples(T, chunk);
  uint32_t const slot = seq & 1u;
  unsigned long long const insideMask = __ballot(inside);
  uint32_t const nInside = uint32_t(__popcll(insideMask));
  uint32_t* const d = s_desc[threadIdx.x >> 6][slot];  // wave-uniform values: every lane stores the same words
  d[0] = chunk, d[1] = item, d[2] = uint32_t(g.px0), d[3] = uint32_t(g.py0), d[4] = s0, d[6] = nInside;
  d[7] = wave_index(gtid) * kSlabsPerWave + slabK;
  d[5] = d[7] * T.chunkSpp * 192u;  // first float of the slab (< 2^32: at most 16 384 slabs of kMaxChunkSpp * 192 floats)
  uint32_t const wbase = slot * kLdsThreads + (threadIdx.x & ~63u);
  s_pixbase[wbase + lane] = inside ? halton_pixel_base(load_cold_args(Pk).sp, px, py) : -1;
  if (inside) s_pixmap[wbase + uint32_t(__popcll(insideMask & ((1ull << lane) - 1ull)))] = uint32_t(lane);
  units = nInside * n;
  return true;
}

// Staged radiance goes to the wave's slab with PLAIN stores (they merge in this XCD's L2; the owner folds its own slab
// from there).  Only when a slab is handed over is it made visible to the other XCDs (slab_publish).
DMT_DEV void stage_sample(KArgs Pk, uint32_t sidx, f3 L) {
  uint32_t const first = s_desc[threadIdx.x >> 6][sidx >> 31][5];
  float* const p = kargs(Pk)->stage + (first + (sidx & ~kSlotBit) * 3u);
  p[0] = L.x, p[1] = L.y, p[2] = L.z;
}
// Hand-over: the folder may run on another XCD, whose L2 is a different one, and this XCD's L2 holds the slab as dirty
// lines.  Writing back the whole L2 (release fence = buffer_wbl2) would cost far more than the slab is worth, so the
// owner re-stores its n x 64 x 3 floats with agent-scope stores (global_store ... sc1: written through to memory), lane i
// the samples of pixel i; the folder reads them with agent-scope loads.  48 loads + 48 stores per lane for 16 samples,
// paid only by chunks that finish before their predecessor (none on a frame with more tiles than resident waves).
DMT_DEV void slab_publish(TileArgs const& T, int lane, uint32_t slab, uint32_t n) {
  float* p = T.stage + size_t(slab) * size_t(T.chunkSpp) * 192u + uint32_t(lane) * 3u;
#pragma unroll 4
  for (uint32_t k = 0; k < n; ++k, p += 192) {
    float const x = p[0], y = p[1], z = p[2];
    uint32_t* const q = reinterpret_cast<uint32_t*>(p);
    __hip_atomic_store(q + 0, __float_as_uint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, __float_as_uint(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 2, __float_as_uint(z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// prepare unit u of item `seq` in this lane's s_prep
DMT_DEV void prepare_unit(KArgs Pk, uint32_t seq, uint32_t u, LaneSched& Ls) {
  uint32_t const slot = seq & 1u;
  uint32_t const* const d = s_desc[threadIdx.x >> 6][slot];
  uint32_t const nInside = d[6];
  uint32_t k, j;
  if (nInside == 64u) k = u >> 6, j = u & 63u;
  else k = u / nInside, j = u - k * nInside;
  uint32_t const wbase = slot * kLdsThreads + (threadIdx.x & ~63u);
  uint32_t const pixel = s_pixmap[wbase + j];
  prepare_sample(Pk, int(d[2]) + int(pixel & 7u), int(d[3]) + int(pixel >> 3), s_pixbase[wbase + pixel], d[4] + k);
  s_prep[14 * kLdsThreads + threadIdx.x] = __uint_as_float((slot << 31) | (k * 64u + pixel));
  Ls.prepared = true, Ls.prepSlot = slot != 0u;
}

// Film words that another wave of this launch may read or have written (the tile's previous / next chunk) are
// moved with agent-scope relaxed atomics (global_load/store ... sc1: coherent across the XCDs' L2s) and
// ordered against the hand-over word by s_waitcnt.  The obvious alternative, plain accesses between
// acquire/release FENCES, costs a buffer_inv / buffer_wbl2 of the XCD's whole L2 per item.
DMT_DEV float4 film_load(float4 const* p) {
  unsigned long long const* const q = reinterpret_cast<unsigned long long const*>(p);
  unsigned long long const a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long const b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float(uint32_t(a)), __uint_as_float(uint32_t(a >> 32)), __uint_as_float(uint32_t(b)),
                     __uint_as_float(uint32_t(b >> 32)));
}
DMT_DEV void film_store(float4* p, float4 v) {
  unsigned long long* const q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q, (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One sample into a pixel's running statistics: SMEMLayout::updateSample, T/megakernel/megakernel.cuh:59-79.  ONE function
// for every fold in this library (megakernel items, wavefront passes), so that every path to the film rounds alike.
DMT_DEV void welford_update(f3& mean, f3& M2, float& N, f3 L) {
  N = N + 1.0f;
  f3 const delta = L - mean;
  mean = mean + delta / N;
  f3 const delta2 = L - mean;
  M2 = M2 + delta * delta2;
}

// Put chunk `chunk` of tile item `item` (staged in `slab`) into the film -- the caller knows that chunk - 1 is there --
// and then every directly following chunk that has already been handed over.  Lane i folds pixel i of the tile.
DMT_DEV void fold_chain(TileArgs const& T, unsigned long long* D, uint32_t gtid, int lane, uint32_t item, uint32_t chunk, uint32_t slab) {
  ItemGeom const g = item_geom(T, item);
  bool const inside = item_lane_inside(T, g, lane);
  size_t const pidx = size_t(g.px0 + (lane & 7)) + size_t(g.py0 + (lane >> 3)) * size_t(T.width);
  f3 mean = mk3(0, 0, 0), M2 = mk3(0, 0, 0);
  float N = 0.f;
  uint32_t folded = 0;
  if (inside) {
    float4 const m = film_load(T.mean + pidx);  // SMEMLayout::startSample, megakernel.cuh:45-57
    float4 const v = film_load(T.m2 + pidx);
    mean = mk3(m.x, m.y, m.z), M2 = mk3(v.x, v.y, v.z), N = v.w;
  }
  for (;;) {
    uint32_t const n = chunk_samples(T, chunk);
    bool const own = slab / kSlabsPerWave == wave_index(gtid);  // wave-uniform
    float const* p = T.stage + size_t(slab) * size_t(T.chunkSpp) * 192u + uint32_t(lane) * 3u;
    if (inside) {
      if (own) {
#pragma unroll 4
        for (uint32_t k = 0; k < n; ++k, p += 192) welford_update(mean, M2, N, mk3(p[0], p[1], p[2]));
      } else {  // another wave's slab: read it where it was written through to
#pragma unroll 2
        for (uint32_t k = 0; k < n; ++k, p += 192) {
          uint32_t const* const q = reinterpret_cast<uint32_t const*>(p);
          uint32_t const x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          uint32_t const y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          uint32_t const z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          welford_update(mean, M2, N, mk3(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z)));
        }
      }
      film_store(T.mean + pidx, make_float4(mean.x, mean.y, mean.z, 0.f));  // endSample, megakernel.cuh:81-85
      film_store(T.m2 + pidx, make_float4(M2.x, M2.y, M2.z, N));
    }
    ++folded;
    if (++chunk == T.numChunks) {  // (a foreign slab of the last chunk still has to be released)
      if (!own) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&T.slabBusy[slab], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slab read, film stores written through -> release the slab, publish
    uint32_t next = 0;
    if (lane == 0) {
      if (!own) __hip_atomic_store(&T.slabBusy[slab], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      next = __hip_atomic_exchange(&T.link[size_t(chunk) * T.numItems + item], kFoldReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    next = uint32_t(__builtin_amdgcn_readfirstlane(int(next)));
    if (next == 0u) break;  // chunk not finished yet: its finisher will find kFoldReady and fold it
    slab = next - 1u;       // finished and handed over: this wave folds it (the running statistics are in registers)
  }
  sched_count(D, lane, SD_FOLDS, folded);
  if (folded > 1u) sched_count(D, lane, SD_CHAINED, folded - 1u);
}

// Item `seq` of this wave is completely staged: fold it, or hand it over to the folder of the tile's previous chunk.
DMT_DEV void item_complete(KArgs Pk, uint32_t gtid, int lane, uint32_t seq, WaveSched& W) {
  TileArgs const T = load_tile_args(Pk);
  uint32_t const* const d = s_desc[threadIdx.x >> 6][seq & 1u];
  uint32_t const chunk = uint32_t(__builtin_amdgcn_readfirstlane(int(d[0])));
  uint32_t const item = uint32_t(__builtin_amdgcn_readfirstlane(int(d[1])));
  uint32_t const slab = uint32_t(__builtin_amdgcn_readfirstlane(int(d[7])));
  uint32_t const slabBit = 1u << (slab - wave_index(gtid) * kSlabsPerWave);
  if (chunk > 0u) {
    // Is the previous chunk in the film already (the usual case when the frame has more tiles than resident waves)?
    // Then this wave folds, and nothing has to be published.  A stale "no" only costs an unnecessary publish.
    uint32_t seen = 0;
    if (lane == 0) seen = __hip_atomic_load(&T.link[size_t(chunk) * T.numItems + item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seen = uint32_t(__builtin_amdgcn_readfirstlane(int(seen)));
    if (seen != kFoldReady) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own staging stores have landed (in this XCD's L2)
      slab_publish(T, lane, slab, chunk_samples(T, chunk));
      if (lane == 0) __hip_atomic_store(&T.slabBusy[slab], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slab is in memory and marked busy -> offer it
      uint32_t old = 0;
      if (lane == 0) old = __hip_atomic_exchange(&T.link[size_t(chunk) * T.numItems + item], slab + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = uint32_t(__builtin_amdgcn_readfirstlane(int(old)));
      if (old != kFoldReady) {  // the previous chunk is not in the film yet: its folder takes this slab
        W.slabPend |= slabBit;
        sched_count(kargs(Pk)->schedDiag, lane, SD_HANDED);
        return;
      }
    }  // (kFoldReady seen or received: nobody else touches this word any more)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own staging stores have landed
  fold_chain(T, kargs(Pk)->schedDiag, gtid, lane, item, chunk, slab);
  W.slabFree |= slabBit;
}

// Hand the next units of the wave's items to the lanes that ask for one (`want`) and prepare them.  This is the ONE place
// where work items are fetched (item_fetch is large, and the kernel already fills most of the instruction cache): a wave
// starts with no item, and a wave whose last live item was retired comes here because all its lanes are starving.
DMT_DEV void sched_draw(KArgs Pk, uint32_t gtid, int lane, WaveSched& W, LaneSched& Ls, bool want) {
  if (__builtin_expect(W.nextUnit == W.totalUnits, 0)) {  // the item units are drawn from is used up (or there is none yet): fetch the next if there is room
    if (!W.exhausted && W.fetched - W.cur < 2u && W.slabFree != 0u) {
      uint32_t units = 0;
      if (item_fetch(Pk, gtid, lane, W.fetched, W, units)) W.alloc = W.fetched, ++W.fetched, W.nextUnit = 0, W.totalUnits = units;
      else W.exhausted = true;
    }
  }
  uint32_t const avail = W.totalUnits - W.nextUnit;
  if (avail == 0u) return;
  unsigned long long const m = __ballot(want);
  uint32_t const rank = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
  uint32_t const cnt = uint32_t(__popcll(m));
  if (want && rank < avail) prepare_unit(Pk, W.alloc, W.nextUnit + rank, Ls);
  W.nextUnit += cnt < avail ? cnt : avail;
}

// item cur: all units drawn and none of them still in a lane -> complete it; false = the wave leaves the launch
DMT_DEV bool sched_retire(KArgs Pk, uint32_t gtid, int lane, WaveSched& W) {
  item_complete(Pk, gtid, lane, W.cur, W);
  ++W.cur;
  if (W.slabPend != 0u) slab_refresh(load_tile_args(Pk), gtid, lane, W);  // once per completed item: which handed-over slabs are back?
  if (W.cur == W.fetched) {  // no live item: the next sched_draw fetches one
    if (W.exhausted) return false;
    if (W.slabFree == 0u) {
      // Every slab of this wave is handed over and waits for a folder.  The wave holds nothing anybody needs; in a
      // healthy launch a slab comes back within one item's tracing time.  Poll for a bounded WALL CLOCK time, then leave:
      // the other waves fetch the remaining items, and a free wave slot lets a switched-out wave run again.
      TileArgs const T = load_tile_args(Pk);
      unsigned long long* const D = kargs(Pk)->schedDiag;
      sched_count(D, lane, SD_STALLS);
      unsigned long long const t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      unsigned long long waited = 0;
      do {
        __builtin_amdgcn_s_sleep(64);
        slab_refresh(T, gtid, lane, W);
        waited = __builtin_amdgcn_s_memrealtime() - t0;
      } while (W.slabFree == 0u && waited <= (unsigned long long)(DMT_SLAB_WAIT_MS) * 100000ull);
      if (lane == 0) atomicMax(&D[SD_MAXSTALL], waited);
      if (W.slabFree == 0u) {
        sched_count(D, lane, SD_EXITS);
        return false;
      }
    }
  }
  return true;
}
// does this lane still hold a sample of the item in slot `curSlot`?
DMT_DEV bool lane_holds(PathState const& st, LaneSched const& Ls, bool curSlot) {
  return (st.active && ((st.sidx >> 31) != 0u) == curSlot) || (Ls.prepared && Ls.prepSlot == curSlot) ||
         (st.finPending && ((get_finIdx() >> 31) != 0u) == curSlot);
}

template <bool STATS>
DMT_DEV void flush_stats(KArgs Pk, LaneStats const& ls) {
  if constexpr (STATS) {
    unsigned long long* const stats = kargs(Pk)->stats;
    atomicAdd(&stats[0], (unsigned long long)ls.samples);
    atomicAdd(&stats[1], (unsigned long long)ls.closest);
    atomicAdd(&stats[2], (unsigned long long)ls.shadow);
    atomicAdd(&stats[3], (unsigned long long)ls.tc.nodes);
    atomicAdd(&stats[4], (unsigned long long)ls.tc.tris);
    atomicAdd(&stats[5], (unsigned long long)ls.bounces);
    atomicAdd(&stats[6], (unsigned long long)ls.itNode);
    atomicAdd(&stats[7], (unsigned long long)ls.itLeaf);
    atomicAdd(&stats[8], (unsigned long long)ls.itShade);
    atomicAdd(&stats[9], (unsigned long long)ls.itOuter);
    atomicAdd(&stats[10], (unsigned long long)ls.itPrep);
    atomicAdd(&stats[11], (unsigned long long)ls.lanesLeaf);
    atomicAdd(&stats[12], (unsigned long long)ls.lanesShade);
    atomicAdd(&stats[13], (unsigned long long)ls.lanesPrep);
    atomicAdd(&stats[14], (unsigned long long)ls.tc.deadNodes);
    atomicAdd(&stats[15], (unsigned long long)ls.tc.overflowPushes);
  }
}

template <bool BVH, bool STATS = false, bool ENV = false, bool AREA = false, int TEX = false, int LTREE = false>
DMT_DEV void megakernel_body() {
  KArgs const Pk = kargs_base();
  LaneStats ls;
  int const lane = int(threadIdx.x) & 63;
  uint32_t const gtid = blockIdx.x * blockDim.x + threadIdx.x;
  WaveSched W;
  LaneSched Ls;
  PathState st{};
  auto sink = [&](f3 L, uint32_t sidx) { stage_sample(Pk, sidx, L); };
#if DMT_SECTION_TIMING
  if (lane < 16) s_sectAcc[threadIdx.x >> 6][lane] = 0;
  sect_mark(15);
#endif
  {
    for (;;) {
      bool const needPrep = !Ls.prepared;
      bool const starving = !st.active && needPrep;
      if (__any(starving) || __popcll(__ballot(needPrep)) >= DMT_PREP_THRESHOLD) sched_draw(Pk, gtid, lane, W, Ls, needPrep);
      if (W.cur == W.fetched) break;  // nothing live and nothing fetched: the launch has no more items
      if (!st.active && Ls.prepared) {
        path_begin_prepared(st);
        Ls.prepared = false;
        if constexpr (STATS) ++ls.samples;
      }
      sect_mark(0);
      if (W.alloc != W.cur || W.nextUnit == W.totalUnits) {  // item cur has no units left: is it complete?
        if (!__any(lane_holds(st, Ls, (W.cur & 1u) != 0u))) {
          bool const more = sched_retire(Pk, gtid, lane, W);  // fold it, or hand it over (never waits for another wave's item)
          sect_mark(1);
          if (!more) break;
          continue;
        }
      }
      sect_mark(1);
      lane_step<BVH, STATS, ENV, AREA, TEX, LTREE>(Pk, gtid, st, sink, STATS ? &ls : nullptr);
    }
  }
#if DMT_SECTION_TIMING
  if (lane < 16) atomicAdd(&g_sect[lane], s_sectAcc[threadIdx.x >> 6][lane]);
#endif
  flush_stats<STATS>(Pk, ls);
}

#ifndef DMT_BVH_NODE_WEIGHT
#define DMT_BVH_NODE_WEIGHT 1  // a node step is chosen when nNode * NODE_WEIGHT >= nLeaf * LEAF_WEIGHT: a leaf step (one
#define DMT_BVH_LEAF_WEIGHT 2  // pair test) costs about half a node step, so it pays from half as many lanes (measured best)
#endif
#ifndef DMT_BVH_SHADE_THRESHOLD
#define DMT_BVH_SHADE_THRESHOLD 32
#endif
// BVH flavour of the megakernel.  Same items, same sample order, same film as megakernel_body, but the
// traversal is asynchronous per lane (bvh_device.hpp: trav_step): every loop iteration advances each
// traversing lane by one node or leaf, lanes that have finished both of their rays wait for shading, and
// shading runs for all waiting lanes at once when at least DMT_BVH_SHADE_THRESHOLD of them are waiting (or
// nobody is traversing).  Incoherent rays take very different numbers of steps; with a pass-synchronous
// loop the wave ran at 14 % lane utilisation.
#ifndef DMT_BVH_DUMMY_LDS
#define DMT_BVH_DUMMY_LDS 0  // occupancy experiments: extra LDS bytes per block (fewer resident blocks per CU)
#endif
template <bool STATS = false, bool ENV = false, bool AREA = false, int TEX = false, int LTREE = false>
DMT_DEV void megakernel_body_bvh() {
  KArgs const Pk = kargs_base();
#if DMT_BVH_DUMMY_LDS > 0
  __shared__ volatile char s_dummy[DMT_BVH_DUMMY_LDS];
  if (threadIdx.x == 0) s_dummy[blockIdx.x % DMT_BVH_DUMMY_LDS] = 1;
#endif
  LaneStats ls;
  int const lane = int(threadIdx.x) & 63;
  uint32_t const gtid = blockIdx.x * blockDim.x + threadIdx.x;
  WaveSched W;
  LaneSched Ls;
  PathState st{};
  auto sink = [&](f3 L, uint32_t sidx) { stage_sample(Pk, sidx, L); };
  Traversal tv{};
  tv.phase = TR_IDLE;
  if constexpr (STATS) tv.stack.ovfCount = &ls.tc.overflowPushes;
  {
    for (;;) {
      // A. draw + prepare units, start samples (only lanes between rounds start one)
      bool const idle = tv.phase == TR_IDLE;
      bool const needPrep = !Ls.prepared;
      bool const starving = idle && !st.active && needPrep;
      if (__any(starving) || __popcll(__ballot(needPrep)) >= DMT_PREP_THRESHOLD) {
        if constexpr (STATS) ++ls.itPrep, ls.lanesPrep += needPrep ? 1u : 0u;
        sched_draw(Pk, gtid, lane, W, Ls, needPrep);
      }
      if (W.cur == W.fetched) break;  // nothing live and nothing fetched: the launch has no more items
      if constexpr (STATS) ++ls.itOuter;
      if (idle && !st.active && Ls.prepared) {
        path_begin_prepared(st);
        Ls.prepared = false;
        if constexpr (STATS) ++ls.samples;
      }
      // B. start a round: pending shadow ray first, then the closest-hit ray.  The ray being traversed is ALWAYS st.rp's
      //    .x half (the traversal keeps no copy of it): a round with a shadow ray swaps the halves, and swaps them back when
      //    the shadow ray is done, so that the closest-hit ray is in .x again when the lane shades.
      if (idle && (st.active || st.hasShadow)) {
        // (what the round consists of is st.active / st.hasShadow themselves: nothing changes them before the lane shades)
        tv.bestTri = -1, tv.bu = 0.f, tv.bv = 0.f;
        if (st.hasShadow) {
          tv.phase = TR_SHADOW;
          swap_rays(st);
        } else {
          tv.phase = TR_CLOSEST;
        }
        trav_set_ray(tv, ray_org(st), ray_dir(st), st.hasShadow ? st.smax : kInf);
        if constexpr (STATS) ls.closest += st.active ? 1u : 0u, ls.shadow += st.hasShadow ? 1u : 0u;
      }
      // R. item cur has no units left: is it complete?
      if (W.alloc != W.cur || W.nextUnit == W.totalUnits) {
        if (!__any(lane_holds(st, Ls, (W.cur & 1u) != 0u))) {
          if (!sched_retire(Pk, gtid, lane, W)) break;
          continue;
        }
      }
      // C. traversal.  Lanes sit on an inner node, on a leaf, or have finished their ray.  Each iteration runs ONE
      //    kind of step -- node or leaf, whichever serves more lanes per instruction (a leaf step costs about half a
      //    node step) -- so a step always serves a good share of the traversing lanes (a plain while-while loop kept running node steps for the last few lanes
      //    that were still descending: 11 % lane utilisation in node steps).  The loop ends when enough lanes
      //    wait for shading.
      BvhView const bvh = load_bvh(Pk);
      int const shadeThreshold = kargs(Pk)->shadeThreshold > 1 ? kargs(Pk)->shadeThreshold : 1;  // (0 would never let the wave traverse)
      // The words of the node a lane stands on are fetched AHEAD: when a step leaves the lane on an inner node, its loads are
      // issued right there, and the step selection, the other kind of step for the other lanes and the loop overhead run
      // under their latency (1 M triangles: 488 -> 517 Msamples/s, 16 M: 440 -> 466).  Fetched anew here for every lane on a
      // node, so that nothing of it is live while the wave shades.  Measured and lost: a leaf's pair words ahead as well (430 in
      // shared registers, 384 in registers of their own) and the hit nearest child ahead of the stack round trip (449); DESIGN 4.2.5.
      NodeWords nd{};
      if ((tv.phase == TR_CLOSEST || tv.phase == TR_SHADOW) && !(tv.cur & kBvhLeafFlag)) nd = node_fetch(bvh, tv.cur);
      for (;;) {
        bool traversing = tv.phase == TR_CLOSEST || tv.phase == TR_SHADOW;
        if (traversing && tv.cur == kBvhEmpty) {  // ray finished: next ray of the round, or done
          if (tv.phase == TR_SHADOW) {
            swap_rays(st);         // .x = the closest-hit ray again
            st.smax = tv.tlim;     // the shadow ray's verdict (negative = occluded) outlives the closest-hit traversal here
          }
          if (tv.phase == TR_SHADOW && st.active) {
            tv.phase = TR_CLOSEST;
            trav_set_ray(tv, ray_org(st), ray_dir(st), kInf);
            nd = node_fetch(bvh, tv.cur);  // the root
          } else {
            tv.phase = TR_DONE;
            traversing = false;
          }
        }
        bool const onNode = traversing && !(tv.cur & kBvhLeafFlag);
        bool const onLeaf = traversing && (tv.cur & kBvhLeafFlag) != 0u;  // cur != kBvhEmpty here
        int const nNode = __popcll(__ballot(onNode)), nLeaf = __popcll(__ballot(onLeaf));
        if (nNode + nLeaf == 0) break;
        if (__popcll(__ballot(tv.phase == TR_DONE)) >= shadeThreshold) break;
        // (parking a found leaf and carrying on with node steps, Aila & Laine's speculative traversal, was measured here in
        //  round 2 and removed: 8 % fewer wave iterations but 5 % slower, the extra dependent LDS pop lengthens every node step)
#ifdef DMT_BVH_BOTH_STEPS  // experiment: every traversing lane advances every iteration (node and leaf code both run)
        if constexpr (STATS) ++ls.itNode, ++ls.itLeaf, ls.lanesLeaf += onLeaf ? 1u : 0u;
        if (onNode) trav_node<STATS>(bvh, tv, STATS ? &ls.tc : nullptr);
        if (onLeaf) trav_leaf<STATS>(bvh, tv, ray_org(st), ray_dir(st), STATS ? &ls.tc : nullptr);
#else
        if (nNode * DMT_BVH_NODE_WEIGHT >= nLeaf * DMT_BVH_LEAF_WEIGHT) {
          if constexpr (STATS) ++ls.itNode;
          if (onNode) {
            trav_node<STATS>(bvh, tv, nd, STATS ? &ls.tc : nullptr);
            if (!(tv.cur & kBvhLeafFlag)) nd = node_fetch(bvh, tv.cur);
          }
        } else {
          if constexpr (STATS) ++ls.itLeaf, ls.lanesLeaf += onLeaf ? 1u : 0u;
          if (onLeaf) {
            trav_leaf<STATS>(bvh, tv, ray_org(st), ray_dir(st), STATS ? &ls.tc : nullptr);
            if (!(tv.cur & kBvhLeafFlag)) nd = node_fetch(bvh, tv.cur);
          }
        }
#endif
      }
      // D. resolve + shade every lane that has finished its round
      if constexpr (STATS) ++ls.itShade, ls.lanesShade += tv.phase == TR_DONE ? 1u : 0u;
      if (tv.phase == TR_DONE) {
        if constexpr (STATS) ls.bounces += (st.active && tv.bestTri >= 0 && st.depth < kargs(Pk)->maxDepth) ? 1u : 0u;
        lane_finish<ENV, AREA, TEX, LTREE, true>(Pk, st, st.active, st.hasShadow, tv.bestTri, tv.bu, tv.bv, st.smax < 0.f, sink);
        tv.phase = TR_IDLE;
      }
    }
  }
  flush_stats<STATS>(Pk, ls);
}

// brute force: the reference's semantics, every triangle tested (small scenes, parity mode)
__global__ void __launch_bounds__(256, DMT_MIN_WAVES_PER_SIMD) k_megakernel(RenderParams P) { megakernel_body<false>(); }
// BVH traversal (large scenes); 16 KB more LDS per block for the traversal stacks
__global__ void __launch_bounds__(256, DMT_MIN_WAVES_PER_SIMD_BVH) k_megakernel_bvh(RenderParams P) { megakernel_body_bvh<false>(); }
// same kernel with per-lane work counters (node visits, triangle tests, rays, bounces): feeds the
// algorithmic-bytes model of the BVH path; never on the timed path
__global__ void __launch_bounds__(256, 2) k_megakernel_bvh_stats(RenderParams P) { megakernel_body_bvh<true>(); }
// A18: the same two kernels with the env-map light compiled in (dmt_upload_envmap selects them).  Separate
// instantiations, so that the register allocation of the default kernels is not touched.
__global__ void __launch_bounds__(256, 4) k_megakernel_env(RenderParams P) { megakernel_body<false, false, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_env(RenderParams P) { megakernel_body_bvh<false, true>(); }
__global__ void __launch_bounds__(256, 2) k_megakernel_bvh_stats_env(RenderParams P) { megakernel_body_bvh<true, true>(); }
// emissive triangles compiled in (dmt_upload_area_lights selects them)
__global__ void __launch_bounds__(256, 4) k_megakernel_area(RenderParams P) { megakernel_body<false, false, false, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_area(RenderParams P) { megakernel_body_bvh<false, false, true>(); }
// image textures compiled in (dmt_upload_textures selects them); with or without the env map
__global__ void __launch_bounds__(256, 4) k_megakernel_tex(RenderParams P) { megakernel_body<false, false, false, false, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_tex(RenderParams P) { megakernel_body_bvh<false, false, false, true>(); }
__global__ void __launch_bounds__(256, 4) k_megakernel_env_tex(RenderParams P) { megakernel_body<false, false, true, false, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_env_tex(RenderParams P) { megakernel_body_bvh<false, true, false, true>(); }
// SURVEY 8f-1 fractional / textured "metallic": the texture kernels plus the two-lobe blend (a second prepared BSDF per lane:
// one wave per SIMD fewer than the texture kernels, whose register allocation stays untouched)
__global__ void __launch_bounds__(256, 3) k_megakernel_blend(RenderParams P) { megakernel_body<false, false, false, false, 2>(); }
__global__ void __launch_bounds__(256, 2) k_megakernel_bvh_blend(RenderParams P) { megakernel_body_bvh<false, false, false, 2>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_env_blend(RenderParams P) { megakernel_body<false, false, true, false, 2>(); }
__global__ void __launch_bounds__(256, 2) k_megakernel_bvh_env_blend(RenderParams P) { megakernel_body_bvh<false, true, false, 2>(); }
// light tree compiled in (dmt_set_light_sampling(DMT_LIGHTS_TREE) selects them); with or without the env map
__global__ void __launch_bounds__(256, 4) k_megakernel_ltree(RenderParams P) { megakernel_body<false, false, false, false, false, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_ltree(RenderParams P) { megakernel_body_bvh<false, false, false, false, true>(); }
__global__ void __launch_bounds__(256, 4) k_megakernel_env_ltree(RenderParams P) { megakernel_body<false, false, true, false, false, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_env_ltree(RenderParams P) { megakernel_body_bvh<false, true, false, false, true>(); }
// both optional light kinds at once
// the reference-semantics light tree (light_tree_ref.hpp): cuts of up to four lights per bounce
__global__ void __launch_bounds__(256, 3) k_megakernel_ltree2(RenderParams P) { megakernel_body<false, false, false, false, false, 2>(); }
__global__ void __launch_bounds__(256, 2) k_megakernel_bvh_ltree2(RenderParams P) { megakernel_body_bvh<false, false, false, false, 2>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_env_ltree2(RenderParams P) { megakernel_body<false, false, true, false, false, 2>(); }
__global__ void __launch_bounds__(256, 2) k_megakernel_bvh_env_ltree2(RenderParams P) { megakernel_body_bvh<false, true, false, false, 2>(); }
__global__ void __launch_bounds__(256, 4) k_megakernel_env_area(RenderParams P) { megakernel_body<false, false, true, true>(); }
__global__ void __launch_bounds__(256, 3) k_megakernel_bvh_env_area(RenderParams P) { megakernel_body_bvh<false, true, true>(); }

#include "wavefront.hpp"

// ---------------------------------------------------------------------------------------------
// device unit-test kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_test_trace(RenderParams P, bool useBvh, int n, int32_t const* pxs, int32_t const* pys,
                             int32_t const* ss, float* L3) {
  KArgs const k = kargs_base();
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  PathState st{};
  if (i < n) {
    ColdArgs const c = load_cold_args(k);
    path_begin(st, c.cam, c.sp, pxs[i], pys[i], halton_pixel_base(c.sp, pxs[i], pys[i]), uint32_t(ss[i]));
  }
  auto store = [&](f3 L, uint32_t) { L3[3 * i] = L.x, L3[3 * i + 1] = L.y, L3[3 * i + 2] = L.z; };
  uint32_t const gtid = blockIdx.x * blockDim.x + threadIdx.x;
  for (;;) {
    if (!__any(st.active || st.hasShadow)) break;
    bool const useEnv = kargs(k)->env.w > 0;
    if (kargs(k)->lightTree != nullptr) {
      if (useEnv) {
        if (useBvh)
          lane_step<true, false, true, false, false, true>(k, gtid, st, store);
        else
          lane_step<false, false, true, false, false, true>(k, gtid, st, store);
      } else if (useBvh) {
        lane_step<true, false, false, false, false, true>(k, gtid, st, store);
      } else {
        lane_step<false, false, false, false, false, true>(k, gtid, st, store);
      }
    } else if (kargs(k)->matTex != nullptr) {
      if (useEnv) {
        if (useBvh)
          lane_step<true, false, true, false, true>(k, gtid, st, store);
        else
          lane_step<false, false, true, false, true>(k, gtid, st, store);
      } else if (useBvh) {
        lane_step<true, false, false, false, true>(k, gtid, st, store);
      } else {
        lane_step<false, false, false, false, true>(k, gtid, st, store);
      }
    } else if (kargs(k)->areaCount > 0 && useEnv) {
      if (useBvh)
        lane_step<true, false, true, true>(k, gtid, st, store);
      else
        lane_step<false, false, true, true>(k, gtid, st, store);
    } else if (kargs(k)->areaCount > 0) {
      if (useBvh)
        lane_step<true, false, false, true>(k, gtid, st, store);
      else
        lane_step<false, false, false, true>(k, gtid, st, store);
    } else if (useEnv) {
      if (useBvh)
        lane_step<true, false, true>(k, gtid, st, store);
      else
        lane_step<false, false, true>(k, gtid, st, store);
    } else if (useBvh) {
      lane_step<true>(k, gtid, st, store);
    } else {
      lane_step<false>(k, gtid, st, store);
    }
  }
}

// A18 probes: env-map sampling (u2 -> wi, pdf, uv, Le by uv) and evaluation by direction (wi -> Le, pdf)
__global__ void k_test_envmap(EnvView env, int n, float const* u2, float const* wiIn, float* wi3, float* pdf, float* uv2,
                              float* Le3, int32_t* ok, float* LeDir3, float* pdfDir) {
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  EnvSampleDev const es = env_sample(env, f2{u2[2 * i], u2[2 * i + 1]});
  f3 const Le = env_eval_uv(env, es.uv);
  wi3[3 * i] = es.wi.x, wi3[3 * i + 1] = es.wi.y, wi3[3 * i + 2] = es.wi.z;
  pdf[i] = es.pdf, uv2[2 * i] = es.uv.x, uv2[2 * i + 1] = es.uv.y;
  Le3[3 * i] = Le.x, Le3[3 * i + 1] = Le.y, Le3[3 * i + 2] = Le.z;
  ok[i] = es.ok ? 1 : 0;
  float p = 0.f;
  f3 const Ld = env_eval_dir(env, mk3(wiIn[3 * i], wiIn[3 * i + 1], wiIn[3 * i + 2]), p);
  LeDir3[3 * i] = Ld.x, LeDir3[3 * i + 1] = Ld.y, LeDir3[3 * i + 2] = Ld.z;
  pdfDir[i] = p;
}

// single path with a per-bounce log {tri, pos3, beta3, L3 (before shading), depth, dim}
__global__ void k_test_trace_log(RenderParams P, int px, int py, int smp, float* rec12, int cap, int* nOut,
                                 float* L3) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  KArgs const k = kargs_base();
  SceneView const sc = load_scene(k);
  PathState st{};
  {
    ColdArgs const c = load_cold_args(k);
    path_begin(st, c.cam, c.sp, px, py, halton_pixel_base(c.sp, px, py), uint32_t(smp));
  }
  int n = 0;
  for (;;) {
    bool const doC = st.active, doS = st.hasShadow;
    int bestTri;
    float bu, bv;
    bool occluded;
    trace_pair_brute(k, st, doC, doS, bestTri, bu, bv, occluded);
    if (doS) {
      if (!occluded) st.L = st.L + get_C();
      st.hasShadow = false;
    }
    bool ended = true;
    if (doC) {
      if (n < cap) {
        float* r = rec12 + 12 * n++;
        f3 pos = mk3(0, 0, 0);
        if (bestTri >= 0) pos = hit_finish(sc.post[bestTri], bu, bv, ray_dir(st)).pos;
        r[0] = float(bestTri), r[1] = pos.x, r[2] = pos.y, r[3] = pos.z;
        r[4] = st.beta.x, r[5] = st.beta.y, r[6] = st.beta.z, r[7] = st.L.x, r[8] = st.L.y, r[9] = st.L.z;
        r[10] = float(st.depth), r[11] = float(st.rng.dim);
      }
      ended = path_shade(k, st, bestTri, bu, bv);
      if (ended) st.active = false;
    }
    if (ended && !st.hasShadow) break;
  }
  *nOut = n;
  L3[0] = st.L.x, L3[1] = st.L.y, L3[2] = st.L.z;
}

__global__ void k_test_tri(float const* xs, float const* ys, float const* zs, uint32_t n, f3 o, f3 d,
                           int32_t* hit, float* t, float* pos3, float* nrm3, float* err3) {
  uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f3 const p0 = mk3(xs[4 * i], ys[4 * i], zs[4 * i]);
  f3 const p1 = mk3(xs[4 * i + 1], ys[4 * i + 1], zs[4 * i + 1]);
  f3 const p2 = mk3(xs[4 * i + 2], ys[4 * i + 2], zs[4 * i + 2]);
  f3 const e0 = mk3(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
  f3 const e1 = mk3(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z);
  Ray const ray{o, d};
  MTResult const r = mt_test(p0, e0, e1, ray);
  hit[i] = r.valid ? 1 : 0;
  float const inf = kInf;
  t[i] = r.valid ? r.t : inf;
  f3 pos = mk3(0, 0, 0), nrm = mk3(0, 0, 0), err = mk3(0, 0, 0);
  if (r.valid) {
    TriPost P;
    P.p0x = p0.x, P.p0y = p0.y, P.p0z = p0.z, P.p1x = p1.x, P.p1y = p1.y, P.p1z = p1.z;
    P.p2x = p2.x, P.p2y = p2.y, P.p2z = p2.z;
    f3 const nn = normalize(cross(e1, e0));
    P.nx = nn.x, P.ny = nn.y, P.nz = nn.z;
    P.matId = 0;
    Hit const h = hit_finish(P, r.u, r.v, mk3(0, 0, 0));  // zero direction: normal not flipped
    pos = h.pos, nrm = h.normal, err = h.error;
  }
  pos3[3 * i] = pos.x, pos3[3 * i + 1] = pos.y, pos3[3 * i + 2] = pos.z;
  nrm3[3 * i] = nrm.x, nrm3[3 * i + 1] = nrm.y, nrm3[3 * i + 2] = nrm.z;
  err3[3 * i] = err.x, err3[3 * i + 1] = err.y, err3[3 * i + 2] = err.z;
}

__global__ void k_test_sampler(SamplerParams sp, int n, int32_t const* pxs, int32_t const* pys,
                               int32_t const* ss, int ndims, int32_t* hidx, float* pix2, float* dims) {
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  int32_t const h = halton_pixel_base(sp, pxs[i], pys[i]) + ss[i] * (sp.scale0 * sp.scale1);
  hidx[i] = h;
  f2 const p = pixel2d(sp, h);
  pix2[2 * i] = p.x, pix2[2 * i + 1] = p.y;
  Sampler r;
  r.start(uint32_t(h));
  for (int d = 0; d < ndims; ++d) dims[size_t(i) * ndims + d] = r.get1D();
}

__global__ void k_test_camera(CameraXf cam, SamplerParams sp, int n, int32_t const* pxs,
                              int32_t const* pys, int32_t const* ss, float* o3, float* d3) {
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  int32_t const h = halton_pixel_base(sp, pxs[i], pys[i]) + ss[i] * (sp.scale0 * sp.scale1);
  Ray const r = camera_ray(cam, sp, pxs[i], pys[i], h);
  o3[3 * i] = r.o.x, o3[3 * i + 1] = r.o.y, o3[3 * i + 2] = r.o.z;
  d3[3 * i] = r.d.x, d3[3 * i + 1] = r.d.y, d3[3 * i + 2] = r.d.z;
}

__global__ void k_test_bsdf(Rec32 rec, int n, float const* ns3, float const* wo3, float const* u2,
                            float const* uc, float const* wi3, float* prep12, float* samp10,
                            float* eval4) {
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  f3 const ns = mk3(ns3[3 * i], ns3[3 * i + 1], ns3[3 * i + 2]);
  f3 const wo = mk3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
  Bsdf const b = bsdf_prepare(rec, ns, wo);
  float* p = prep12 + 12 * size_t(i);
  p[0] = b.weight.x, p[1] = b.weight.y, p[2] = b.weight.z;
  bool const oren = b.type == BS_OREN, ggx = b.type == BS_GGX_DIEL || b.type == BS_GGX_COND;  // the two kinds share registers (Bsdf)
  p[3] = oren ? b.ms.x : 0.f, p[4] = oren ? b.ms.y : 0.f, p[5] = oren ? b.ms.z : 0.f;
  p[6] = ggx ? b.escale : 0.f, p[7] = float(b.type), p[8] = ggx ? b.ax : 0.f, p[9] = ggx ? b.ay : 0.f, p[10] = ggx ? b.phi0 : 0.f;
  p[11] = b.type == BS_GGX_DIEL ? b.eta : 0.f;
  BsdfSample const s = sample_bsdf(b, wo, ns, ns, mk2(u2[2 * i], u2[2 * i + 1]), uc[i]);
  float* o = samp10 + 10 * size_t(i);
  o[0] = s.wi.x, o[1] = s.wi.y, o[2] = s.wi.z, o[3] = s.f.x, o[4] = s.f.y, o[5] = s.f.z;
  o[6] = s.pdf, o[7] = s.eta, o[8] = s.delta ? 1.f : 0.f, o[9] = s.refract ? 1.f : 0.f;
  float pdf = 0.f;
  f3 const f = eval_bsdf(b, wo, mk3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), ns, ns, pdf) * b.weight;
  float* e = eval4 + 4 * size_t(i);
  e[0] = f.x, e[1] = f.y, e[2] = f.z, e[3] = pdf;
}

__global__ void k_test_light(Rec32 rec, int n, float const* pos3, float const* nrm3, float const* u2,
                             int32_t const* hadT, float* out14) {
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  LightSample const s = sample_light(rec, mk3(pos3[3 * i], pos3[3 * i + 1], pos3[3 * i + 2]),
                                     mk2(u2[2 * i], u2[2 * i + 1]), hadT[i] != 0,
                                     mk3(nrm3[3 * i], nrm3[3 * i + 1], nrm3[3 * i + 2]));
  f3 const Le = eval_light(rec, s);
  float* o = out14 + 14 * size_t(i);
  o[0] = s.pLight.x, o[1] = s.pLight.y, o[2] = s.pLight.z;
  o[3] = s.direction.x, o[4] = s.direction.y, o[5] = s.direction.z;
  o[6] = s.pdf, o[7] = float(s.delta), o[8] = s.distance, o[9] = s.factor;
  o[10] = Le.x, o[11] = Le.y, o[12] = Le.z, o[13] = s.valid() ? 1.f : 0.f;
}

__global__ void k_test_half(int n, float const* fin, uint16_t* hout, uint16_t const* hin, float* fout) {
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  if (fin && hout) hout[i] = uint16_t(f2h(fin[i]));
  if (hin && fout) fout[i] = h2f(hin[i]);
}

__global__ void k_test_closest(RenderParams P, bool useBvh, int n, float const* o3, float const* d3, int32_t* tri,
                               float* tOut) {
  KArgs const k = kargs_base();
  int const i = int(blockIdx.x * blockDim.x + threadIdx.x);
  bool const alive = i < n;
  PathState st{};
  if (alive)
    set_ray(st, mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]));
  st.active = alive;
  int best;
  float bu, bv, bt = kInf;
  bool occluded;
  if (useBvh)
    trace_pair_bvh(k, st, alive, false, blockIdx.x * blockDim.x + threadIdx.x, best, bu, bv, occluded);
  else
    trace_pair_brute(k, st, alive, false, best, bu, bv, occluded);
  if (alive && best >= 0) {  // t of the winning triangle (same arithmetic as the loops)
    TriS const T = load_tri(to_const_as(load_scene(k).tris), uint32_t(best));
    bt = mt_pair(T, st.rp).t.x;
  }
  if (alive) tri[i] = best, tOut[i] = bt;
}

}  // namespace

// =============================================================================================
// host side of the C ABI
// =============================================================================================
struct dmt_ctx {
  int device = 0;
  hipStream_t ownStream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  // scene
  TriIsect* d_tris = nullptr;
  TriPost* d_post = nullptr;
  Rec32* d_bsdfs = nullptr;
  Rec32* d_lights = nullptr;
  Rec32* d_inf = nullptr;
  uint32_t triCount = 0, bsdfCount = 0, lightCount = 0, infCount = 0;
  uint32_t maxMatId = 0;
  // BVH (built on demand for DMT_ACCEL_BVH)
  std::vector<float> h_xs, h_ys, h_zs;  // host copy of the soup (the builder's input)
  std::vector<uint32_t> h_mat;
  Bvh4Node* d_bvhNodes = nullptr;
  int shadeThresholdEnv = 0;  // DMT_BVH_SHADE_THRESHOLD from the environment, 0 = choose by tree size
  TriPair* d_trisBvh = nullptr;   // leaf storage of the BVH
  uint32_t* d_overflow = nullptr;
  size_t overflowThreads = 0;
  bool haveBvh = false;
  int bvhDepth = 0;
  uint32_t bvhNodeCount = 0, bvhPairCount = 0;
  int blocksPerCUBvh = 0;
  // light tree (light_tree.hpp): built from the uploaded lights when dmt_set_light_sampling asks for it
  int lightSampling = DMT_LIGHTS_UNIFORM;
  std::vector<uint8_t> h_lights;  // host copy of the packed light records
  LightTreeNode* d_lightTree = nullptr;
  LightTreeRefNode* d_lightTreeRef = nullptr;  // DMT_LIGHTS_TREE_REFERENCE
  uint32_t lightTreeNodes = 0;
  int lightTreeDepth = 0;
  bool lightTreeValid = false;
  bool lightTreeTooDeep = false;   // the last build exceeded the walk's depth guard: the uniform pick is used instead (dmt_last_error says so)
  bool lightsTreeable = false;     // every record of the light list is a point or spot light
  std::vector<std::pair<void const*, int>> occupancy;  // megakernel variant -> resident 256-thread blocks per CU
  // SURVEY 8f-1 image textures (one allocation each)
  uint32_t* d_texRgba = nullptr;
  int32_t* d_texDesc = nullptr;
  uint32_t* d_matTex = nullptr;
  float* d_triUv = nullptr;
  uint32_t texCount = 0, matTexCount = 0;
  bool hasBlend = false;  // some uploaded BSDF record is a BS_GGX_BLEND pair: the *_tex kernels carry that code
  size_t triUvCount = 0;
  // wavefront form of the BVH path (wavefront.hpp)
  int bvhStrategy = 0;             // 0 = automatic (by launch size), 1 = megakernel, 2 = wavefront
  size_t wfTargetPaths = size_t(1) << 22;  // path slots per pass
  float* d_wfState = nullptr;
  uint32_t* d_wfQueue = nullptr;   // two queues
  uint32_t* d_wfCounts = nullptr;  // counts + cursors
  size_t wfSlotsCap = 0, wfCountsCap = 0;
  int wfBlocksTrace = 0, wfBlocksShade = 0;
  float* d_env = nullptr;  // A18: one allocation holding the five tables and the image
  EnvView env{};           // env.w == 0: no env map
  uint32_t* d_areaOf = nullptr;   // per-triangle area-light index
  uint32_t* d_areaTri = nullptr;
  float* d_areaLe = nullptr;
  uint32_t areaCount = 0;
  std::vector<uint32_t> h_areaTri;  // kept to rebuild areaOf when triangles are re-uploaded
  std::vector<float> h_areaLe;
  bool haveTris = false, haveBsdfs = false, haveLights = false, haveCamera = false;
  // camera
  dmt_camera cam{};
  CameraXf xf{};
  SamplerParams sp{};
  // film
  float4* d_mean = nullptr;
  float4* d_m2 = nullptr;
  bool ownFilm = false;
  int filmW = 0, filmH = 0;
  uint32_t* d_counter = nullptr;   // work counter
  uint32_t* d_sched = nullptr;     // [waves * kSlabsPerWave] slab-busy marks, then [numChunks][numItems] hand-over words; zeroed per launch
  unsigned long long* d_schedDiag = nullptr;  // kSchedDiagWords counters over all launches (dmt_sched_diag)
  unsigned long long expectedFolds = 0;       // work items launched so far: what d_schedDiag[0] must read once the stream has drained
  float* d_stage = nullptr;        // staging slabs of finished samples, [wave][kSlabsPerWave][chunkSpp][64] float3
  size_t stageFloats = 0;
  size_t schedCap = 0;
  uint32_t chunkSpp = 0;           // samples per work item, 0 = automatic
  int subShift = -1;               // row bands per tile (log2); -1 = choose per launch
  int maxDepth = 32;
  int accel = DMT_ACCEL_BRUTE_FORCE;
  int rank = 0, world = 1;
  // launch geometry
  int cuCount = 0;
  int blocksPerCU = 0;
  // timing
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t eventsUsed = 0;
  double accumMs = 0.0;
  uint64_t accumLaunches = 0;
};

namespace {

std::string g_createError;

float h2f_host(uint16_t h) {  // exact fp16 -> fp32 (CC/private/encoding.cu:124-155)
  uint32_t const sgn = uint32_t(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, out;
  if (e == 0) {
    if (m == 0) {
      out = sgn;
    } else {
      e = 113;
      while (!(m & 0x400u)) m <<= 1, --e;
      out = sgn | (e << 23) | ((m & 0x3FFu) << 13);
    }
  } else if (e == 31) {
    out = sgn | 0x7F800000u | (m << 13);
  } else {
    out = sgn | ((e + 112) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &out, 4);
  return f;
}

#define HIP_TRY(ctx, call)                                                                 \
  do {                                                                                     \
    hipError_t const e__ = (call);                                                         \
    if (e__ != hipSuccess) {                                                               \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorName(e__) + " - " +              \
                   hipGetErrorString(e__);                                                 \
      return DMT_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

int fail(dmt_ctx* ctx, int code, char const* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

// ---- host math for the one-off camera / sampler setup (IEEE fp32, same expressions as the
// reference host code: CC/private/extra_math.cu:43-90, CC/private/common_math.cu:16-78,
// CC/private/rng.cu:21-46,182-208)
struct H3 {
  float x, y, z;
};
H3 hcross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
H3 hnormalize(H3 a) {
  float const inv = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  return {a.x * inv, a.y * inv, a.z * inv};
}
void worldFromCamera(float const dir[3], float const pos[3], float m[16]) {
  H3 const fwd = hnormalize({dir[0], dir[1], dir[2]});
  H3 const right = hnormalize(hcross(fwd, {0, 0, 1}));
  H3 const up = hcross(right, fwd);
  m[0] = right.x, m[4] = up.x, m[8] = fwd.x, m[12] = pos[0];
  m[1] = right.y, m[5] = up.y, m[9] = fwd.y, m[13] = pos[1];
  m[2] = right.z, m[6] = up.z, m[10] = fwd.z, m[14] = pos[2];
  m[3] = 0.f, m[7] = 0.f, m[11] = 0.f, m[15] = 1.f;
}
void cameraFromRaster(float focal_mm, float sensorH_mm, uint32_t xRes, uint32_t yRes, float m[16]) {
  float const sensorW_mm = sensorH_mm * float(xRes) / float(yRes);
  float const MM = 0.001f;
  float const focal = focal_mm * MM, sh = sensorH_mm * MM, sw = sensorW_mm * MM;
  float const psx = sw / float(xRes), psy = sh / float(yRes);
  float const tx = -0.5f * sw + 0.5f * psx;
  float const ty = 0.5f * sh - 0.5f * psy;
  for (int i = 0; i < 16; ++i) m[i] = 0.f;
  m[0] = psx, m[5] = -psy, m[10] = 1.f, m[12] = tx, m[13] = ty, m[14] = focal, m[15] = 1.f;
}
int64_t multInverse(int64_t a, int64_t n) {
  int64_t t = 0, nt = 1, r = n, nr = a;
  while (nr != 0) {
    int64_t const q = r / nr;
    int64_t tmp = t - q * nt;
    t = nt, nt = tmp;
    tmp = r - q * nr;
    r = nr, nr = tmp;
  }
  return t < 0 ? t + n : t;
}
SamplerParams computeSamplerParams(int width, int height) {
  SamplerParams p{};
  int const res[2] = {width, height};
  int32_t scale[2], ex[2];
  int const base[2] = {2, 3};
  for (int i = 0; i < 2; ++i) {
    scale[i] = 1, ex[i] = 0;
    int const lim = res[i] < 128 ? res[i] : 128;
    while (scale[i] < lim) scale[i] *= base[i], ++ex[i];
  }
  p.scale0 = scale[0], p.scale1 = scale[1], p.exp0 = ex[0], p.exp1 = ex[1];
  p.inv0 = int32_t(multInverse(scale[1], scale[0]));
  p.inv1 = int32_t(multInverse(scale[0], scale[1]));
  return p;
}

// the light tree applies to plain point / spot light lists; textured or emissive-triangle scenes keep the uniform pick
bool useLightTreeRef(dmt_ctx const* c) {
  return !c->lightTreeTooDeep && c->lightSampling == DMT_LIGHTS_TREE_REFERENCE && c->lightCount > 1 && c->lightsTreeable && c->areaCount == 0 && c->texCount == 0 && !c->hasBlend;
}
bool useLightTree(dmt_ctx const* c) {
  return !c->lightTreeTooDeep && c->lightSampling == DMT_LIGHTS_TREE && c->lightCount > 1 && c->lightsTreeable && c->areaCount == 0 && c->texCount == 0 && !c->hasBlend;
}
int ensureLightTree(dmt_ctx* ctx);

SceneView sceneView(dmt_ctx const* c) {
  SceneView s;
  s.tris = c->d_tris, s.post = c->d_post, s.bsdfs = c->d_bsdfs, s.lights = c->d_lights;
  s.infLights = c->d_inf;
  s.triCount = c->triCount, s.bsdfCount = c->bsdfCount, s.lightCount = c->lightCount;
  s.infLightCount = c->infCount;
  return s;
}

BvhView bvhView(dmt_ctx const* c, size_t threads) {
  BvhView b;
  b.nodes = c->d_bvhNodes, b.pairs = c->d_trisBvh, b.overflow = c->d_overflow;
  b.overflowStride = uint32_t(threads);
  return b;
}

// BVH megakernel: how many lanes of a wave must have finished their rays before the wave stops traversing and shades
// (megakernel_body_bvh, step C).  Traversing lanes idle while the wave shades and finished lanes idle while it traverses, so
// the best value follows the cost ratio of the two -- low where rays take hundreds of steps, high where the tree is shallow
// and shading dominates.  Measured on MI355X, Msamples/s by threshold (profiles/r03/shade_threshold_sweep.txt):
//   Cornell box, 6 nodes, depth 3             32: 1 923   48: 2 184   56: 2 266   60: 2 244   64: 2 053
//   sphere.fbx + veranda, 123 nodes, depth 5  32: 6 867   48: 7 416   56: 7 683   64: 7 800
//   tessellated sphere, 4 588 nodes, depth 10 32: 3 043   48: 3 381   52: 3 399   56: 3 365   64: 2 748
//   1 M random triangles, 300 k nodes         28: 490     32: 489     36: 485     40: 473     48: 452
//   16 M random triangles                     24: 435     28: 439     32: 437     36: 436
// The step between 16 k and 128 k nodes is interpolated, not measured.  DMT_BVH_SHADE_THRESHOLD in the environment overrides
// the choice (tuning runs).  Results do not depend on it.
int bvhShadeThreshold(dmt_ctx const* c) {
  if (c->shadeThresholdEnv > 0) return c->shadeThresholdEnv;
  uint32_t const n = c->bvhNodeCount;
  return n <= 1024u ? 56 : n <= 16384u ? 52 : n <= 131072u ? 40 : DMT_BVH_SHADE_THRESHOLD;
}
// scene / camera / limits part of the argument struct (what the path-tracing device code reads)
RenderParams baseParams(dmt_ctx const* c, size_t threads) {
  RenderParams P{};
  P.scene = sceneView(c);
  P.bvh = bvhView(c, threads);
  P.cam = c->xf;
  P.sp = c->sp;
  P.maxDepth = c->maxDepth;
  P.shadeThreshold = bvhShadeThreshold(c);
  P.env = c->env;
  P.areaOf = c->d_areaOf, P.areaTri = c->d_areaTri, P.areaLe = c->d_areaLe, P.areaCount = c->areaCount;
  if (c->texCount > 0) P.texRgba = c->d_texRgba, P.texDesc = c->d_texDesc, P.matTex = c->d_matTex, P.triUv = c->d_triUv;
  if (useLightTree(c) && c->lightTreeValid) P.lightTree = c->d_lightTree;
  if (useLightTreeRef(c) && c->lightTreeValid) P.lightTreeRef = c->d_lightTreeRef;
  return P;
}

// which megakernel a launch of this context runs: accelerator x optional light kinds x textures
typedef void (*MegakernelFn)(RenderParams);
MegakernelFn megakernelOf(dmt_ctx const* c) {
  bool const bvh = c->accel == DMT_ACCEL_BVH, env = c->env.w > 0, area = c->areaCount > 0, tex = c->texCount > 0;
  if (c->hasBlend) return bvh ? (env ? k_megakernel_bvh_env_blend : k_megakernel_bvh_blend) : (env ? k_megakernel_env_blend : k_megakernel_blend);
  if (tex) return bvh ? (env ? k_megakernel_bvh_env_tex : k_megakernel_bvh_tex) : (env ? k_megakernel_env_tex : k_megakernel_tex);
  if (useLightTreeRef(c)) return bvh ? (env ? k_megakernel_bvh_env_ltree2 : k_megakernel_bvh_ltree2) : (env ? k_megakernel_env_ltree2 : k_megakernel_ltree2);
  if (useLightTree(c)) return bvh ? (env ? k_megakernel_bvh_env_ltree : k_megakernel_bvh_ltree) : (env ? k_megakernel_env_ltree : k_megakernel_ltree);
  if (area && env) return bvh ? k_megakernel_bvh_env_area : k_megakernel_env_area;
  if (area) return bvh ? k_megakernel_bvh_area : k_megakernel_area;
  if (env) return bvh ? k_megakernel_bvh_env : k_megakernel_env;
  return bvh ? k_megakernel_bvh : k_megakernel;
}
int blocksPerCuOf(dmt_ctx* c) {
  void const* const fn = reinterpret_cast<void const*>(megakernelOf(c));
  for (auto const& e : c->occupancy)
    if (e.first == fn) return e.second;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, 0) != hipSuccess || n <= 0) n = 1;
  c->occupancy.emplace_back(fn, n);
  return n;
}

template <class T>
int devAlloc(dmt_ctx* ctx, T** p, size_t n) {
  if (*p) {
    (void)hipFree(*p);
    *p = nullptr;
  }
  if (n == 0) n = 1;
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return DMT_OK;
}

int ensureOverflow(dmt_ctx* ctx, size_t threads) {
  if (threads <= ctx->overflowThreads && ctx->d_overflow) return DMT_OK;
  if (ctx->d_overflow) (void)hipFree(ctx->d_overflow);
  ctx->d_overflow = nullptr, ctx->overflowThreads = 0;
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_overflow), threads * size_t(kBvhOverflowStack) * sizeof(uint32_t)));
  ctx->overflowThreads = threads;
  return DMT_OK;
}

// (re)build the 4-wide BVH of the uploaded soup and upload nodes + triangle pairs
int buildBvh(dmt_ctx* ctx) {
  uint32_t const n = ctx->triCount;
  bvh_build::Result r = bvh_build::build(ctx->h_xs.data(), ctx->h_ys.data(), ctx->h_zs.data(), n);
  // leaf storage: triangle pairs (bvh.hpp TriPair).  Edges are the reference's own subtractions
  // (CC/private/shapes.cu:10-11) in IEEE fp32.
  size_t const npairs = r.pairTris.size() / 2;
  if (npairs > 0x7FFFFFFFull || r.nodes.size() > 0x7FFFFFFFull) return fail(ctx, DMT_ERR_INVALID, "BVH: too many nodes / triangle pairs");
  // + 3 guard pairs (copies of the last one).  An EMPTY child slot holds an inverted quantised box and no reference of its
  // own; its slab test misses by itself except in one corner: a ray exactly parallel to an axis through a node that is flat
  // on the remaining axes (255 quantisation steps below half an ulp of the plane distance), where near == far.  The slot's
  // implicit reference is then leafRef + slot, i.e. a pair of the NEXT node -- or, for the last node, up to three
  // pairs past the array.  Testing some real triangle of the scene once more changes no result (the triangle test decides
  // hits, and a scene triangle is a scene triangle); reading past the array would, hence the guards.
  std::vector<TriPair> pairs(npairs ? npairs + 3 : 0);
  for (size_t p = 0; p < npairs; ++p)
    for (int half = 0; half < 2; ++half) {
      uint32_t const i = r.pairTris[2 * p + size_t(half)];
      float const* xs = &ctx->h_xs[4 * size_t(i)];
      float const* ys = &ctx->h_ys[4 * size_t(i)];
      float const* zs = &ctx->h_zs[4 * size_t(i)];
      TriPair& P = pairs[p];
      P.p0x[half] = xs[0], P.p0y[half] = ys[0], P.p0z[half] = zs[0];
      P.e0x[half] = xs[1] - xs[0], P.e0y[half] = ys[1] - ys[0], P.e0z[half] = zs[1] - zs[0];
      P.e1x[half] = xs[2] - xs[0], P.e1y[half] = ys[2] - ys[0], P.e1z[half] = zs[2] - zs[0];
      P.orig[half] = i;
    }
  for (size_t p = npairs; p < pairs.size(); ++p) pairs[p] = pairs[npairs - 1];
  int rc = devAlloc(ctx, &ctx->d_bvhNodes, r.nodes.size());
  if (rc) return rc;
  rc = devAlloc(ctx, &ctx->d_trisBvh, pairs.size());
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->d_bvhNodes, r.nodes.data(), r.nodes.size() * sizeof(Bvh4Node), hipMemcpyHostToDevice));
  if (!pairs.empty())
    HIP_TRY(ctx, hipMemcpy(ctx->d_trisBvh, pairs.data(), pairs.size() * sizeof(TriPair), hipMemcpyHostToDevice));
  ctx->bvhDepth = r.depth;
  ctx->bvhNodeCount = uint32_t(r.nodes.size());
  ctx->bvhPairCount = uint32_t(npairs);
  ctx->haveBvh = true;
  return DMT_OK;
}

// octahedral decode on the host (CC/private/encoding.cu:39-60), as pt_device.hpp's dir_from_octa
void octa_host(uint32_t octa, float out[3]) {
  float const mx = 65535.f;
  float const fx = float(octa & 0xFFFFu) / mx * 2.f - 1.f, fy = float((octa >> 16) & 0xFFFFu) / mx * 2.f - 1.f;
  float n[3] = {fx, fy, 1.f - fabsf(fx) - fabsf(fy)};
  if (n[2] < 0.f) {
    n[0] = (1.f - fabsf(fy)) * (std::signbit(fx) ? -1.f : 1.f);
    n[1] = (1.f - fabsf(fx)) * (std::signbit(fy) ? -1.f : 1.f);
  }
  float const inv = 1.f / sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  out[0] = n[0] * inv, out[1] = n[1] * inv, out[2] = n[2] * inv;
}
std::vector<LightTreeRefNode> buildLightTreeRef(uint8_t const* lights32, uint32_t count, int* depth, bool* ok) {
  std::vector<light_tree_ref::Item> items;
  *ok = true;
  for (uint32_t i = 0; i < count; ++i) {
    light_tree_ref::Item it{};
    if (!light_tree_ref::itemOf(lights32 + 32 * size_t(i), i, [](uint16_t h) { return h2f_host(h); }, [](uint32_t o, float* d) { octa_host(o, d); }, it)) {
      *ok = false;
      return {};
    }
    items.push_back(it);
  }
  return light_tree_ref::build(items, depth);
}
// (re)build the light tree from the host copy of the light records and upload it
int ensureLightTree(dmt_ctx* ctx) {
  if (ctx->lightTreeValid) return DMT_OK;
  if (ctx->lightSampling == DMT_LIGHTS_TREE_REFERENCE) {
    bool ok = true;
    std::vector<LightTreeRefNode> const nodes = buildLightTreeRef(ctx->h_lights.data(), ctx->lightCount, &ctx->lightTreeDepth, &ok);
    if (!ok) return fail(ctx, DMT_ERR_STATE, "light tree: only point and spot lights can be in the light list");
    if (ctx->lightTreeDepth > kLightTreeRefMaxDepth) {
      ctx->lightTreeTooDeep = true;
      return DMT_OK;
    }
    int const rcR = devAlloc(ctx, &ctx->d_lightTreeRef, nodes.size());
    if (rcR) return rcR;
    if (!nodes.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_lightTreeRef, nodes.data(), nodes.size() * sizeof(LightTreeRefNode), hipMemcpyHostToDevice));
    ctx->lightTreeNodes = uint32_t(nodes.size());
    ctx->lightTreeValid = true;
    return DMT_OK;
  }
  std::vector<light_tree::Item> items;
  for (uint32_t i = 0; i < ctx->lightCount; ++i) {
    light_tree::Item it{};
    if (!light_tree::itemOf(ctx->h_lights.data() + 32 * size_t(i), i, [](uint16_t h) { return h2f_host(h); }, it))
      return fail(ctx, DMT_ERR_STATE, "light tree: only point and spot lights can be in the light list");
    items.push_back(it);
  }
  std::vector<LightTreeNode> const nodes = light_tree::build(items, &ctx->lightTreeDepth);
  if (ctx->lightTreeDepth > 60) {  // lights at geometrically growing spacing: the walk's guard would cut paths short -> uniform pick
    ctx->lightTreeTooDeep = true;
    return DMT_OK;
  }
  int const rc = devAlloc(ctx, &ctx->d_lightTree, nodes.size());
  if (rc) return rc;
  if (!nodes.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_lightTree, nodes.data(), nodes.size() * sizeof(LightTreeNode), hipMemcpyHostToDevice));
  ctx->lightTreeNodes = uint32_t(nodes.size());
  ctx->lightTreeValid = true;
  return DMT_OK;
}

// scratch buffers for the test entry points
struct Scratch {
  dmt_ctx* ctx;
  std::vector<void*> ptrs;
  explicit Scratch(dmt_ctx* c) : ctx(c) {}
  ~Scratch() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <class T>
  T* up(T const* host, size_t n) {  // upload (or allocate when host == nullptr)
    void* d = nullptr;
    if (hipMalloc(&d, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    ptrs.push_back(d);
    if (host && n && hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return static_cast<T*>(d);
  }
};
#define SCRATCH_CHECK(ctx, p) \
  if (!(p)) return fail(ctx, DMT_ERR_HIP, "scratch allocation / upload failed")

// After the stream has drained: every work item of every past launch must have been folded into the film exactly once
// (item_complete / fold_chain count their folds).  Anything else means the hand-over protocol lost or duplicated a sample
// chunk and the film is not the ordered fold the contract promises.
int checkErrorFlag(dmt_ctx* ctx) {
  unsigned long long folds = 0;
  HIP_TRY(ctx, hipMemcpy(&folds, ctx->d_schedDiag, sizeof(folds), hipMemcpyDeviceToHost));
  if (folds != ctx->expectedFolds) {
    char msg[200];
    snprintf(msg, sizeof(msg), "in-launch ordering: %llu sample chunks were folded into the film, %llu were launched: the film is invalid",
             folds, ctx->expectedFolds);
    ctx->expectedFolds = folds;  // report once
    return fail(ctx, DMT_ERR_HIP, msg);
  }
  return DMT_OK;
}

int finishTest(dmt_ctx* ctx) {
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DMT_OK;
}

}  // namespace

static int rebuildAreaLights(dmt_ctx* ctx);

extern "C" {

int dmt_ctx_create(int device_ordinal, dmt_ctx** out) {
  if (!out) return DMT_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    g_createError = "no HIP device available (the HIP path has no CPU fallback)";
    return DMT_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= count) {
    g_createError = "device ordinal out of range";
    return DMT_ERR_INVALID;
  }
  dmt_ctx* ctx = new (std::nothrow) dmt_ctx();
  if (!ctx) return DMT_ERR_INVALID;
  ctx->device = device_ordinal;
  hipError_t e = hipSetDevice(device_ordinal);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking);
  hipDeviceProp_t prop{};
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_ordinal);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_counter), 2 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemset(ctx->d_counter, 0, 2 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->d_schedDiag), kSchedDiagWords * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(ctx->d_schedDiag, 0, kSchedDiagWords * sizeof(unsigned long long));
  int bpc = 0;
  if (e == hipSuccess)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, reinterpret_cast<void const*>(k_megakernel), 256, 0);
  int bpcBvh = 0;
  if (e == hipSuccess)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpcBvh, reinterpret_cast<void const*>(k_megakernel_bvh), 256, 0);
  if (e != hipSuccess) {
    g_createError = std::string("dmt_ctx_create: ") + hipGetErrorString(e);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
    return DMT_ERR_HIP;
  }
  ctx->stream = ctx->ownStream;
  ctx->cuCount = prop.multiProcessorCount;
  ctx->blocksPerCU = bpc > 0 ? bpc : 1;
  ctx->blocksPerCUBvh = bpcBvh > 0 ? bpcBvh : 1;
  if (char const* e3 = std::getenv("DMT_BVH_STRATEGY")) {  // experiments: 0 auto, 1 megakernel, 2 wavefront
    int const v = std::atoi(e3);
    ctx->bvhStrategy = v < 0 || v > 2 ? 0 : v;
  }
  if (char const* e5 = std::getenv("DMT_BVH_SHADE_THRESHOLD")) {  // tuning runs (bvhShadeThreshold)
    int const v = std::atoi(e5);
    ctx->shadeThresholdEnv = v < 1 ? 0 : (v > 64 ? 64 : v);
  }
  if (char const* e4 = std::getenv("DMT_WF_PATHS")) {
    long long const v = std::atoll(e4);
    if (v >= 4096) ctx->wfTargetPaths = size_t(v);
  }
  if (char const* e2 = std::getenv("DMT_SUB_SHIFT")) {  // scheduling experiments only: results do not depend on it
    int const v = std::atoi(e2);
    ctx->subShift = v < 0 ? -1 : (v > 2 ? 2 : v);
  }
  *out = ctx;
  return DMT_OK;
}

int dmt_ctx_destroy(dmt_ctx* ctx) {
  if (!ctx) return DMT_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& ev : ctx->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  (void)hipFree(ctx->d_tris);
  (void)hipFree(ctx->d_post);
  (void)hipFree(ctx->d_bsdfs);
  (void)hipFree(ctx->d_lights);
  (void)hipFree(ctx->d_inf);
  (void)hipFree(ctx->d_counter);
  (void)hipFree(ctx->d_sched);
  (void)hipFree(ctx->d_schedDiag);
  (void)hipFree(ctx->d_stage);
  (void)hipFree(ctx->d_env);
  (void)hipFree(ctx->d_areaOf);
  (void)hipFree(ctx->d_areaTri);
  (void)hipFree(ctx->d_areaLe);
  (void)hipFree(ctx->d_bvhNodes);
  (void)hipFree(ctx->d_trisBvh);
  (void)hipFree(ctx->d_overflow);
  (void)hipFree(ctx->d_lightTree);
  (void)hipFree(ctx->d_lightTreeRef);
  (void)hipFree(ctx->d_texRgba);
  (void)hipFree(ctx->d_texDesc);
  (void)hipFree(ctx->d_matTex);
  (void)hipFree(ctx->d_triUv);
  (void)hipFree(ctx->d_wfState);
  (void)hipFree(ctx->d_wfQueue);
  (void)hipFree(ctx->d_wfCounts);
  if (ctx->ownFilm) {
    (void)hipFree(ctx->d_mean);
    (void)hipFree(ctx->d_m2);
  }
  if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
  delete ctx;
  return DMT_OK;
}

const char* dmt_last_error(const dmt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int dmt_upload_triangles(dmt_ctx* ctx, const float* xs, const float* ys, const float* zs,
                         const uint32_t* mat_id, size_t count) {
  if (!ctx) return DMT_ERR_INVALID;
  if ((count && (!xs || !ys || !zs || !mat_id)) || count > 0x7FFFFFFFu)
    return fail(ctx, DMT_ERR_INVALID, "dmt_upload_triangles: null array or count out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<TriIsect> a(count);
  std::vector<TriPost> b(count);
  uint32_t maxMat = 0;
  for (size_t i = 0; i < count; ++i) {
    if (mat_id[i] > maxMat) maxMat = mat_id[i];
    H3 const p0{xs[4 * i], ys[4 * i], zs[4 * i]};
    H3 const p1{xs[4 * i + 1], ys[4 * i + 1], zs[4 * i + 1]};
    H3 const p2{xs[4 * i + 2], ys[4 * i + 2], zs[4 * i + 2]};
    H3 const e0{p1.x - p0.x, p1.y - p0.y, p1.z - p0.z};
    H3 const e1{p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
    H3 const n = hnormalize(hcross(e1, e0));
    TriIsect& t = a[i];
    t.p0x = p0.x, t.p0y = p0.y, t.p0z = p0.z;
    t.e0x = e0.x, t.e0y = e0.y, t.e0z = e0.z;
    t.e1x = e1.x, t.e1y = e1.y, t.e1z = e1.z;
    t.matId = mat_id[i], t.pad0 = 0, t.pad1 = 0;
    TriPost& q = b[i];
    q.p0x = p0.x, q.p0y = p0.y, q.p0z = p0.z;
    q.p1x = p1.x, q.p1y = p1.y, q.p1z = p1.z;
    q.p2x = p2.x, q.p2y = p2.y, q.p2z = p2.z;
    q.nx = n.x, q.ny = n.y, q.nz = n.z;
    q.matId = mat_id[i], q.pad0 = q.pad1 = q.pad2 = 0;
  }
  int rc = devAlloc(ctx, &ctx->d_tris, count);
  if (rc) return rc;
  rc = devAlloc(ctx, &ctx->d_post, count);
  if (rc) return rc;
  if (count) {
    HIP_TRY(ctx, hipMemcpy(ctx->d_tris, a.data(), count * sizeof(TriIsect), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_post, b.data(), count * sizeof(TriPost), hipMemcpyHostToDevice));
  }
  ctx->triCount = uint32_t(count);
  ctx->maxMatId = maxMat;
  ctx->haveTris = true;
  ctx->h_xs.assign(xs, xs + 4 * count), ctx->h_ys.assign(ys, ys + 4 * count), ctx->h_zs.assign(zs, zs + 4 * count);
  ctx->h_mat.assign(mat_id, mat_id + count);
  ctx->haveBvh = false;
  ctx->h_areaTri.clear(), ctx->h_areaLe.clear();  // emissive triangles are indices into the soup just replaced
  if (int const rcA = rebuildAreaLights(ctx)) return rcA;
  if (ctx->accel == DMT_ACCEL_BVH) return buildBvh(ctx);
  return DMT_OK;
}

int dmt_upload_bsdfs(dmt_ctx* ctx, const void* bsdf32, uint32_t count) {
  if (!ctx) return DMT_ERR_INVALID;
  if (count && !bsdf32) return fail(ctx, DMT_ERR_INVALID, "dmt_upload_bsdfs: null array");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = devAlloc(ctx, &ctx->d_bsdfs, count);
  if (rc) return rc;
  if (count) HIP_TRY(ctx, hipMemcpy(ctx->d_bsdfs, bsdf32, size_t(count) * 32, hipMemcpyHostToDevice));
  // fractional-metallic materials (BS_GGX_BLEND: this record + the conductor record after it) run on the *_tex kernels
  ctx->hasBlend = false;
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t w1;
    memcpy(&w1, static_cast<unsigned char const*>(bsdf32) + 32 * size_t(i) + 4, 4);
    if ((w1 >> 16) == BS_GGX_BLEND) {
      uint32_t w1next = 0;
      if (i + 1 < count) memcpy(&w1next, static_cast<unsigned char const*>(bsdf32) + 32 * size_t(i + 1) + 4, 4);
      if (i + 1 >= count || (w1next >> 16) != BS_GGX_COND)
        return fail(ctx, DMT_ERR_INVALID, "dmt_upload_bsdfs: a blend record (type 4) must be followed by its GGX conductor record");
      ctx->hasBlend = true;
    }
  }
  ctx->bsdfCount = count;
  ctx->haveBsdfs = true;
  return DMT_OK;
}

int dmt_upload_lights(dmt_ctx* ctx, const void* lights32, uint32_t count, const void* infinite32,
                      uint32_t infinite_count) {
  if (!ctx) return DMT_ERR_INVALID;
  if ((count && !lights32) || (infinite_count && !infinite32))
    return fail(ctx, DMT_ERR_INVALID, "dmt_upload_lights: null array");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = devAlloc(ctx, &ctx->d_lights, count);
  if (rc) return rc;
  rc = devAlloc(ctx, &ctx->d_inf, infinite_count);
  if (rc) return rc;
  if (count) HIP_TRY(ctx, hipMemcpy(ctx->d_lights, lights32, size_t(count) * 32, hipMemcpyHostToDevice));
  if (infinite_count)
    HIP_TRY(ctx, hipMemcpy(ctx->d_inf, infinite32, size_t(infinite_count) * 32, hipMemcpyHostToDevice));
  ctx->lightCount = count;
  ctx->infCount = infinite_count;
  ctx->haveLights = true;
  ctx->h_lights.assign(static_cast<uint8_t const*>(lights32), static_cast<uint8_t const*>(lights32) + size_t(count) * 32);
  ctx->lightTreeValid = false;
  ctx->lightTreeTooDeep = false;
  ctx->lightsTreeable = count > 0;
  for (uint32_t i = 0; i < count; ++i) {  // a directional light in the list (PBRT "distant") has no position: such lists keep the uniform pick
    uint16_t type;
    memcpy(&type, ctx->h_lights.data() + 32 * size_t(i) + 6, 2);
    if (type != 0 && type != 1) ctx->lightsTreeable = false;
  }
  return DMT_OK;
}

int dmt_set_camera(dmt_ctx* ctx, const dmt_camera* cam) {
  if (!ctx) return DMT_ERR_INVALID;
  if (!cam || cam->width <= 0 || cam->height <= 0 || cam->width > 65536 || cam->height > 65536)
    return fail(ctx, DMT_ERR_INVALID, "dmt_set_camera: bad camera / resolution");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->cam = *cam;
  float m[16];
  cameraFromRaster(cam->focal_length, cam->sensor_size, uint32_t(cam->width), uint32_t(cam->height), m);
  memcpy(ctx->xf.cfr, m, sizeof(m));
  worldFromCamera(cam->dir, cam->pos, m);
  memcpy(ctx->xf.rfc, m, sizeof(m));
  ctx->sp = computeSamplerParams(cam->width, cam->height);
  if (cam->width != ctx->filmW || cam->height != ctx->filmH) {
    if (ctx->ownFilm) {
      (void)hipFree(ctx->d_mean);
      (void)hipFree(ctx->d_m2);
    }
    ctx->d_mean = ctx->d_m2 = nullptr;
    ctx->ownFilm = false;
    size_t const bytes = size_t(cam->width) * size_t(cam->height) * sizeof(float4);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_mean), bytes));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_m2), bytes));
    ctx->ownFilm = true;
    ctx->filmW = cam->width, ctx->filmH = cam->height;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_mean, 0, bytes, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_m2, 0, bytes, ctx->stream));
  }
  ctx->haveCamera = true;
  return DMT_OK;
}

int dmt_set_limits(dmt_ctx* ctx, int max_depth) {
  if (!ctx) return DMT_ERR_INVALID;
  if (max_depth < 0) return fail(ctx, DMT_ERR_INVALID, "dmt_set_limits: max_depth < 0");
  ctx->maxDepth = max_depth;
  return DMT_OK;
}

int dmt_set_accel(dmt_ctx* ctx, int mode) {
  if (!ctx) return DMT_ERR_INVALID;
  if (mode != DMT_ACCEL_BRUTE_FORCE && mode != DMT_ACCEL_BVH) return fail(ctx, DMT_ERR_INVALID, "dmt_set_accel: unknown mode");
  ctx->accel = mode;
  if (mode == DMT_ACCEL_BVH && ctx->haveTris && !ctx->haveBvh) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return buildBvh(ctx);
  }
  return DMT_OK;
}

int dmt_set_light_sampling(dmt_ctx* ctx, int mode) {
  if (!ctx) return DMT_ERR_INVALID;
  if (mode != DMT_LIGHTS_UNIFORM && mode != DMT_LIGHTS_TREE && mode != DMT_LIGHTS_TREE_REFERENCE)
    return fail(ctx, DMT_ERR_INVALID, "dmt_set_light_sampling: unknown mode");
  if (mode != ctx->lightSampling) ctx->lightTreeValid = false, ctx->lightTreeTooDeep = false;  // the two trees are different structures
  ctx->lightSampling = mode;
  return DMT_OK;
}

// host only: the probability with which the light tree built from `lights32` picks each light at (p, n)
int dmt_light_tree_pmfs(const void* lights32, uint32_t count, const float* p3, const float* n3, float* pmf_out, int* node_count,
                        int* depth) {
  if ((count && !lights32) || !p3 || !n3 || !pmf_out) return DMT_ERR_INVALID;
  std::vector<light_tree::Item> items;
  for (uint32_t i = 0; i < count; ++i) {
    light_tree::Item it{};
    if (!light_tree::itemOf(static_cast<uint8_t const*>(lights32) + 32 * size_t(i), i, [](uint16_t h) { return h2f_host(h); }, it)) return DMT_ERR_INVALID;
    items.push_back(it);
  }
  int d = 0;
  std::vector<LightTreeNode> const nodes = light_tree::build(items, &d);
  light_tree::pmfs(nodes, p3, n3, pmf_out, count);
  if (node_count) *node_count = int(nodes.size());
  if (depth) *depth = d;
  return DMT_OK;
}

// host only: cut + selection of the reference-semantics tree at n shading points (light_tree_ref.hpp's ltr_select, the
// function the *_ltree2 kernels call)
int dmt_light_tree_ref_select(const void* lights32, uint32_t count, int n, const float* p3, const float* n3, const float* u, float start_pmf,
                              int32_t* indices4, float* pmfs4, int32_t* counts, int* node_count, int* depth) {
  if ((count && !lights32) || n < 0 || (n && (!p3 || !n3 || !u || !indices4 || !pmfs4 || !counts))) return DMT_ERR_INVALID;
  bool ok = true;
  int d = 0;
  std::vector<LightTreeRefNode> const nodes = buildLightTreeRef(static_cast<uint8_t const*>(lights32), count, &d, &ok);
  if (!ok || nodes.empty()) return DMT_ERR_INVALID;
  if (node_count) *node_count = int(nodes.size());
  if (depth) *depth = d;
  for (int i = 0; i < n; ++i) {
    LightTreeRefSelection const sel = ltr_select(nodes.data(), p3[3 * i], p3[3 * i + 1], p3[3 * i + 2], n3[3 * i], n3[3 * i + 1], n3[3 * i + 2], u[i], start_pmf);
    counts[i] = int32_t(sel.count);
    for (int k = 0; k < kLightTreeMaxSplitSize; ++k)
      indices4[4 * i + k] = k < int(sel.count) ? int32_t(sel.indices[k]) : -1, pmfs4[4 * i + k] = k < int(sel.count) ? sel.pmfs[k] : 0.f;
  }
  return DMT_OK;
}

int dmt_set_bvh_strategy(dmt_ctx* ctx, int strategy, uint64_t paths_per_pass) {
  if (!ctx) return DMT_ERR_INVALID;
  if (strategy < 0 || strategy > 2) return fail(ctx, DMT_ERR_INVALID, "dmt_set_bvh_strategy: 0 = automatic, 1 = megakernel, 2 = wavefront");
  ctx->bvhStrategy = strategy;
  if (paths_per_pass) ctx->wfTargetPaths = size_t(paths_per_pass);
  return DMT_OK;
}

int dmt_set_partition(dmt_ctx* ctx, int rank, int world) {
  if (!ctx) return DMT_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world) return fail(ctx, DMT_ERR_INVALID, "dmt_set_partition: bad rank/world");
  ctx->rank = rank, ctx->world = world;
  return DMT_OK;
}

int dmt_set_stream(dmt_ctx* ctx, void* hip_stream) {
  if (!ctx) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->ownStream;
  return DMT_OK;
}

int dmt_film_clear(dmt_ctx* ctx) {
  if (!ctx) return DMT_ERR_INVALID;
  if (!ctx->d_mean) return fail(ctx, DMT_ERR_STATE, "dmt_film_clear: no film (call dmt_set_camera first)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t const bytes = size_t(ctx->filmW) * size_t(ctx->filmH) * sizeof(float4);
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_mean, 0, bytes, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_m2, 0, bytes, ctx->stream));
  return DMT_OK;
}

int dmt_film_bind(dmt_ctx* ctx, void* d_mean, void* d_m2) {
  if (!ctx) return DMT_ERR_INVALID;
  if (!ctx->haveCamera) return fail(ctx, DMT_ERR_STATE, "dmt_film_bind: call dmt_set_camera first");
  if (!d_mean || !d_m2) return fail(ctx, DMT_ERR_INVALID, "dmt_film_bind: null buffer");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ownFilm) {
    (void)hipFree(ctx->d_mean);
    (void)hipFree(ctx->d_m2);
    ctx->ownFilm = false;
  }
  ctx->d_mean = static_cast<float4*>(d_mean);
  ctx->d_m2 = static_cast<float4*>(d_m2);
  return DMT_OK;
}

int dmt_film_device_ptrs(dmt_ctx* ctx, void** d_mean, void** d_m2) {
  if (!ctx || !d_mean || !d_m2) return DMT_ERR_INVALID;
  *d_mean = ctx->d_mean, *d_m2 = ctx->d_m2;
  return DMT_OK;
}

int dmt_download_film(dmt_ctx* ctx, float* mean4, float* m24) {
  if (!ctx) return DMT_ERR_INVALID;
  if (!ctx->d_mean) return fail(ctx, DMT_ERR_STATE, "dmt_download_film: no film");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t const bytes = size_t(ctx->filmW) * size_t(ctx->filmH) * sizeof(float4);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (int const rc = checkErrorFlag(ctx)) return rc;
  if (mean4) HIP_TRY(ctx, hipMemcpy(mean4, ctx->d_mean, bytes, hipMemcpyDeviceToHost));
  if (m24) HIP_TRY(ctx, hipMemcpy(m24, ctx->d_m2, bytes, hipMemcpyDeviceToHost));
  return DMT_OK;
}

// Wavefront form of a BVH launch (wavefront.hpp): passes over (owned tiles, sample range), each pass a fixed sequence of
// kernels on the context's stream.  `ownedTiles` = tiles this rank renders; P carries region / partition / film.
static int launchWavefront(dmt_ctx* ctx, RenderParams P, uint32_t ownedTiles, uint32_t sample_offset, uint32_t spp, bool useEnv,
                           bool useArea, uint64_t* stats, int nstats) {
  uint32_t const iters = uint32_t(ctx->maxDepth) + 2u;  // closest rays at depth 0..maxDepth, + one trailing shadow ray
  size_t const target = ctx->wfTargetPaths < 4096 ? 4096 : ctx->wfTargetPaths;
  uint32_t const tilesPerPass = uint32_t(std::min<size_t>(ownedTiles, std::max<size_t>(1, target / 64)));
  uint32_t n = uint32_t(std::max<size_t>(1, target / (size_t(tilesPerPass) * 64)));
  if (n > spp) n = spp;
  size_t const slotsMax = size_t(tilesPerPass) * 64 * n;
  if (slotsMax > ctx->wfSlotsCap) {
    (void)hipFree(ctx->d_wfState), (void)hipFree(ctx->d_wfQueue);
    ctx->d_wfState = nullptr, ctx->d_wfQueue = nullptr, ctx->wfSlotsCap = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_wfState), size_t(WF_PLANES) * slotsMax * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_wfQueue), 2 * slotsMax * sizeof(uint32_t)));
    ctx->wfSlotsCap = slotsMax;
  }
  size_t const nCounts = 2 * (size_t(iters) + 2);
  if (nCounts > ctx->wfCountsCap) {
    (void)hipFree(ctx->d_wfCounts);
    ctx->d_wfCounts = nullptr, ctx->wfCountsCap = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_wfCounts), nCounts * sizeof(uint32_t)));
    ctx->wfCountsCap = nCounts;
  }
  if (ctx->wfBlocksTrace == 0) {
    int a = 0, b = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<void const*>(k_wf_trace), 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<void const*>(k_wf_shade_env_area), 256, 0);
    ctx->wfBlocksTrace = a > 0 ? a : 1, ctx->wfBlocksShade = b > 0 ? b : 1;
  }
  uint32_t const traceBlocks = uint32_t(ctx->cuCount) * uint32_t(stats ? 4 : ctx->wfBlocksTrace);
  uint32_t const shadeBlocks = uint32_t(ctx->cuCount) * uint32_t(stats ? 2 : ctx->wfBlocksShade);
  int const rcO = ensureOverflow(ctx, size_t(traceBlocks) * 256);
  if (rcO) return rcO;
  P.bvh = bvhView(ctx, size_t(traceBlocks) * 256);
  unsigned long long* dstats = nullptr;
  if (stats) {
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dstats), 16 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(dstats, 0, 16 * sizeof(unsigned long long), ctx->stream));
    P.stats = dstats;
  }
  WfParams W{};
  W.state = ctx->d_wfState;
  W.queue[0] = ctx->d_wfQueue, W.queue[1] = ctx->d_wfQueue + slotsMax;
  W.counts = ctx->d_wfCounts, W.cursors = ctx->d_wfCounts + (iters + 2);
  for (uint32_t tile0 = 0; tile0 < ownedTiles; tile0 += tilesPerPass) {
    uint32_t const tiles = std::min(tilesPerPass, ownedTiles - tile0);
    for (uint32_t s = 0; s < spp; s += n) {
      W.tile0 = tile0, W.pixelSlots = tiles * 64u, W.s0 = sample_offset + s, W.n = std::min(n, spp - s);
      W.slots = W.pixelSlots * W.n;
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_wfCounts, 0, nCounts * sizeof(uint32_t), ctx->stream));
      uint32_t const genBlocks = std::min<uint32_t>((W.slots + 255u) / 256u, uint32_t(ctx->cuCount) * 8u);
      W.it = 0;
      hipLaunchKernelGGL(k_wf_generate, dim3(genBlocks), dim3(256), 0, ctx->stream, P, W);
      for (uint32_t it = 0; it < iters; ++it) {
        W.it = it;
        // a persistent grid no larger than the pass: small passes do not pay for 2 048 idle blocks per launch
        uint32_t const tb = std::min<uint32_t>(traceBlocks, (W.slots + 255u) / 256u);
        uint32_t const sb = std::min<uint32_t>(shadeBlocks, (W.slots + 255u) / 256u);
        if (stats) {
          hipLaunchKernelGGL(k_wf_trace_stats, dim3(tb), dim3(256), 0, ctx->stream, P, W);
          if (useEnv)
            hipLaunchKernelGGL(k_wf_shade_stats_env, dim3(sb), dim3(256), 0, ctx->stream, P, W);
          else
            hipLaunchKernelGGL(k_wf_shade_stats, dim3(sb), dim3(256), 0, ctx->stream, P, W);
          continue;
        }
        hipLaunchKernelGGL(k_wf_trace, dim3(tb), dim3(256), 0, ctx->stream, P, W);
        if (useEnv && useArea)
          hipLaunchKernelGGL(k_wf_shade_env_area, dim3(sb), dim3(256), 0, ctx->stream, P, W);
        else if (useArea)
          hipLaunchKernelGGL(k_wf_shade_area, dim3(sb), dim3(256), 0, ctx->stream, P, W);
        else if (useEnv)
          hipLaunchKernelGGL(k_wf_shade_env, dim3(sb), dim3(256), 0, ctx->stream, P, W);
        else
          hipLaunchKernelGGL(k_wf_shade, dim3(sb), dim3(256), 0, ctx->stream, P, W);
      }
      if (!stats) hipLaunchKernelGGL(k_wf_fold, dim3((W.pixelSlots + 255u) / 256u), dim3(256), 0, ctx->stream, P, W);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  if (stats) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    std::vector<unsigned long long> h(16, 0);
    if (e == hipSuccess) e = hipMemcpy(h.data(), dstats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(dstats);
    HIP_TRY(ctx, e);
    {  // samples = pixels of the owned tiles inside the region x spp (the kernels count rays, visits and bounces)
      uint64_t pixels = 0;
      for (uint32_t item = 0; item < ownedTiles; ++item) {
        uint32_t const j = uint32_t(P.rank) + item * uint32_t(P.world);
        int const px0 = (P.tx0 + int(j % uint32_t(P.rtx))) * 8, py0 = (P.ty0 + int(j / uint32_t(P.rtx))) * 8;
        int const w = std::min(px0 + 8, P.x1) - std::max(px0, P.x0), hgt = std::min(py0 + 8, P.y1) - std::max(py0, P.y0);
        if (w > 0 && hgt > 0) pixels += uint64_t(w) * uint64_t(hgt);
      }
      h[0] = pixels * spp;
    }
    for (int k = 0; k < nstats && k < 16; ++k) stats[k] = h[size_t(k)];
  }
  return DMT_OK;
}

static int renderImpl(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1, uint64_t* stats6, int nstats) {
  if (!ctx) return DMT_ERR_INVALID;
  if (stats6) memset(stats6, 0, size_t(nstats) * sizeof(uint64_t));
  if (!(ctx->haveTris && ctx->haveBsdfs && ctx->haveLights && ctx->haveCamera))
    return fail(ctx, DMT_ERR_STATE, "dmt_render: upload triangles, bsdfs, lights and set the camera first");
  // the Halton index sample * stride must stay inside int32 (CC/private/rng.cu:229)
  uint64_t const stride = uint64_t(ctx->sp.scale0) * uint64_t(ctx->sp.scale1);
  if ((uint64_t(sample_offset) + spp + 1) * stride > 0x7FFFFFFFull)
    return fail(ctx, DMT_ERR_INVALID, "dmt_render: sample index overflows the 32-bit Halton index");
  // every material index must address an uploaded BSDF (the kernel gathers bsdfs[matId])
  if (ctx->triCount > 0 && ctx->maxMatId >= ctx->bsdfCount)
    return fail(ctx, DMT_ERR_INVALID, "dmt_render: a triangle's material index is outside the BSDF array");
  if (x0 < 0) x0 = 0;
  if (y0 < 0) y0 = 0;
  if (x1 > ctx->filmW) x1 = ctx->filmW;
  if (y1 > ctx->filmH) y1 = ctx->filmH;
  if (spp == 0 || x1 <= x0 || y1 <= y0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the light tree is built before anything asks which kernel will run: a tree deeper than the walk's guard switches the
  // context back to the uniform pick (lightTreeTooDeep), which changes the kernel, its occupancy and the launch shape
  if ((useLightTree(ctx) || useLightTreeRef(ctx)) && !ctx->lightTreeValid) {
    if (int const rcT = ensureLightTree(ctx)) return rcT;
  }
  if (stats6 && (ctx->texCount > 0 || ctx->hasBlend || ctx->areaCount > 0 || useLightTree(ctx) || useLightTreeRef(ctx)))
    return fail(ctx, DMT_ERR_STATE, "dmt_render_stats / dmt_render_profile: the counting kernels exist for the plain and env-map BVH kernels only; "
                                    "with textures, blended materials, emissive triangles or a light tree they would describe a different kernel");

  RenderParams P{};
  P.scene = sceneView(ctx);
  P.cam = ctx->xf;
  P.sp = ctx->sp;
  P.mean = ctx->d_mean, P.m2 = ctx->d_m2, P.counter = ctx->d_counter;
  P.width = ctx->filmW, P.height = ctx->filmH;
  P.x0 = x0, P.y0 = y0, P.x1 = x1, P.y1 = y1;
  P.tx0 = x0 / 8, P.ty0 = y0 / 8;
  int const tx1 = (x1 + 7) / 8, ty1 = (y1 + 7) / 8;
  P.rtx = tx1 - P.tx0;
  uint32_t const tiles = uint32_t(P.rtx) * uint32_t(ty1 - P.ty0);
  P.rank = ctx->rank, P.world = ctx->world;
  P.numItems = tiles > uint32_t(ctx->rank) ? (tiles - uint32_t(ctx->rank) + uint32_t(ctx->world) - 1) / uint32_t(ctx->world) : 0;
  P.sampleOffset = sample_offset, P.spp = spp;
  P.maxDepth = ctx->maxDepth;
  P.shadeThreshold = bvhShadeThreshold(ctx);
  if (P.numItems == 0) return DMT_OK;
  uint32_t const ownedTiles = P.numItems;
  // BVH launches run as the megakernel unless the wavefront form (wavefront.hpp) is asked for: on the measured scenes
  // the megakernel is faster (1 M triangles: 489 vs 378 Msamples/s), so "automatic" means megakernel
  bool const wavefront = ctx->accel == DMT_ACCEL_BVH && ctx->bvhStrategy == 2 && ctx->texCount == 0 && !ctx->hasBlend && !useLightTree(ctx) && !useLightTreeRef(ctx);  // textures / blends / light tree: megakernels only
  {  // fewer owned tiles than ~4 per resident wave: schedule row bands of the tiles instead of whole tiles
    uint32_t const waves = uint32_t(ctx->cuCount) * uint32_t(blocksPerCuOf(ctx)) * 4u;
    P.subShift = ctx->subShift >= 0 ? uint32_t(ctx->subShift) : 0u;
    if (ctx->subShift < 0)
      while (P.subShift < 2u && (uint64_t(P.numItems) << P.subShift) < 4ull * waves) ++P.subShift;
    P.numItems <<= P.subShift;
  }
  // samples per work item: automatic = 16 per 64 pixels (1 024 path samples per item); bounded so the staging
  // area (768 B per sample index per resident wave and slot) stays small
  P.chunkSpp = ctx->chunkSpp ? ctx->chunkSpp : (16u << P.subShift);
  if (P.chunkSpp > spp) P.chunkSpp = spp;
  if (P.chunkSpp > kMaxChunkSpp) P.chunkSpp = kMaxChunkSpp;
  P.numChunks = (spp + P.chunkSpp - 1) / P.chunkSpp;
  if (uint64_t(P.numItems) * P.numChunks > 0x7FFFFFFFull)
    return fail(ctx, DMT_ERR_INVALID, "dmt_render: too many work items (tiles x sample chunks); raise dmt_set_chunk or split the pass");
  P.schedDiag = ctx->d_schedDiag;

  bool const useBvh = ctx->accel == DMT_ACCEL_BVH;
  uint32_t const wavesWanted = P.numItems * P.numChunks < P.numItems ? P.numItems : P.numItems * P.numChunks;
  uint32_t blocks = uint32_t(ctx->cuCount) * uint32_t(blocksPerCuOf(ctx));
  bool const useEnv = ctx->env.w > 0;
  bool const useArea = ctx->areaCount > 0;
  P.env = ctx->env;
  P.areaOf = ctx->d_areaOf, P.areaTri = ctx->d_areaTri, P.areaLe = ctx->d_areaLe, P.areaCount = ctx->areaCount;
  if (ctx->hasBlend && useArea) return fail(ctx, DMT_ERR_STATE, "dmt_render: fractional-metallic materials together with emissive triangles are not supported");
  if (ctx->texCount > 0) {
    if (useArea) return fail(ctx, DMT_ERR_STATE, "dmt_render: image textures together with emissive triangles are not supported");
    if (ctx->matTexCount != ctx->bsdfCount || ctx->triUvCount != ctx->triCount)
      return fail(ctx, DMT_ERR_STATE, "dmt_render: texture tables do not match the uploaded BSDFs / triangles (upload textures last)");
    P.texRgba = ctx->d_texRgba, P.texDesc = ctx->d_texDesc, P.matTex = ctx->d_matTex, P.triUv = ctx->d_triUv;
  }
  if (useLightTree(ctx)) {
    if (int const rcT = ensureLightTree(ctx)) return rcT;
    P.lightTree = ctx->d_lightTree;
  }
  if (useLightTreeRef(ctx)) {
    if (int const rcT = ensureLightTree(ctx)) return rcT;
    P.lightTreeRef = ctx->d_lightTreeRef;
  }
  uint32_t const blocksNeeded = (wavesWanted + 3) / 4;
  if (blocks > blocksNeeded) blocks = blocksNeeded;
  if (blocks == 0) blocks = 1;

  if (ctx->eventsUsed == ctx->events.size()) {
    hipEvent_t a, b;
    HIP_TRY(ctx, hipEventCreate(&a));
    HIP_TRY(ctx, hipEventCreate(&b));
    ctx->events.emplace_back(a, b);
  }
  auto& ev = ctx->events[ctx->eventsUsed];
  {  // staging slabs of finished samples, kSlabsPerWave per wave of the launch
    size_t const floats = size_t(blocks) * 4u * size_t(kSlabsPerWave) * size_t(P.chunkSpp) * 192u;
    if (floats > ctx->stageFloats) {
      if (ctx->d_stage) (void)hipFree(ctx->d_stage);
      ctx->d_stage = nullptr, ctx->stageFloats = 0;
      HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_stage), floats * sizeof(float)));
      ctx->stageFloats = floats;
    }
    P.stage = ctx->d_stage;
  }
  {  // slab-busy marks + one hand-over word per (chunk, tile item), all zero at launch
    size_t const busyWords = size_t(blocks) * 4u * size_t(kSlabsPerWave);
    size_t const linkWords = P.numChunks > 1 ? size_t(P.numItems) * size_t(P.numChunks) : 0;
    if (busyWords + linkWords > ctx->schedCap) {
      if (ctx->d_sched) (void)hipFree(ctx->d_sched);
      ctx->d_sched = nullptr, ctx->schedCap = 0;
      HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_sched), (busyWords + linkWords) * sizeof(uint32_t)));
      ctx->schedCap = busyWords + linkWords;
    }
    P.slabBusy = ctx->d_sched, P.link = ctx->d_sched + busyWords;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sched, 0, (busyWords + linkWords) * sizeof(uint32_t), ctx->stream));
  }
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_counter, 0, sizeof(uint32_t), ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ev.first, ctx->stream));
  uint64_t const launchFolds = uint64_t(P.numItems) * P.numChunks;  // every item is folded exactly once (checkErrorFlag)
  if (useBvh && wavefront) {
    if (!ctx->haveBvh) return fail(ctx, DMT_ERR_STATE, "dmt_render: BVH not built");
    int const rcW = launchWavefront(ctx, P, ownedTiles, sample_offset, spp, useEnv, useArea, stats6, nstats);
    if (rcW) return rcW;
    if (stats6) return DMT_OK;
  } else if (useBvh) {
    if (!ctx->haveBvh) return fail(ctx, DMT_ERR_STATE, "dmt_render: BVH not built");
    int const rcO = ensureOverflow(ctx, size_t(ctx->cuCount) * size_t(std::max(blocksPerCuOf(ctx), ctx->blocksPerCUBvh)) * 256);
    if (rcO) return rcO;
    P.bvh = bvhView(ctx, size_t(blocks) * 256);
    if (stats6) {
      unsigned long long* dstats = nullptr;
      HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dstats), 16 * sizeof(unsigned long long)));
      HIP_TRY(ctx, hipMemsetAsync(dstats, 0, 16 * sizeof(unsigned long long), ctx->stream));
      P.stats = dstats;
      if (useEnv)
        hipLaunchKernelGGL(k_megakernel_bvh_stats_env, dim3(blocks), dim3(256), 0, ctx->stream, P);
      else
        hipLaunchKernelGGL(k_megakernel_bvh_stats, dim3(blocks), dim3(256), 0, ctx->stream, P);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) ctx->expectedFolds += launchFolds;
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e == hipSuccess) e = hipMemcpy(stats6, dstats, size_t(nstats) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      (void)hipFree(dstats);
      HIP_TRY(ctx, e);
      return DMT_OK;
    }
    hipLaunchKernelGGL(megakernelOf(ctx), dim3(blocks), dim3(256), 0, ctx->stream, P);
  } else {
    hipLaunchKernelGGL(megakernelOf(ctx), dim3(blocks), dim3(256), 0, ctx->stream, P);
  }
  HIP_TRY(ctx, hipGetLastError());
  if (!(useBvh && wavefront)) ctx->expectedFolds += launchFolds;
  HIP_TRY(ctx, hipEventRecord(ev.second, ctx->stream));
  ++ctx->eventsUsed;
  return DMT_OK;
}

#if DMT_SECTION_TIMING
extern "C" int dmt_diag_section_cycles(unsigned long long* out16, int reset) {
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sect), 16 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_sect), z, sizeof(z));
  }
  return e == hipSuccess ? 0 : -1;
}
#endif

// Host-only: build the BVH of a soup and check its invariants (every triangle in exactly one leaf, every DECODED
// (quantised) child box encloses all vertices below it and lies inside its parent's decoded box up to one quantisation
// step, inner children first with consecutive indices, depth within the traversal-stack bound).
int dmt_bvh_validate(const float* xs, const float* ys, const float* zs, size_t count, int* node_count, int* depth,
                     int* max_leaf) {
  if ((count && (!xs || !ys || !zs)) || count > 0x0FFFFFFFu) return DMT_ERR_INVALID;
  bvh_build::Result const r = bvh_build::build(xs, ys, zs, uint32_t(count));
  if (node_count) *node_count = int(r.nodes.size());
  if (depth) *depth = r.depth;
  std::vector<uint8_t> seen(count, 0);
  std::vector<uint8_t> nodeSeen(r.nodes.size(), 0);
  int maxLeaf = 0;
  bool ok = r.depth <= kBvhMaxDepth && !r.nodes.empty();
  size_t const npairs = r.pairTris.size() / 2;
  struct Item {
    uint32_t node;
    float lo[3], hi[3];  // decoded box of the slot this node hangs in
  };
  std::vector<Item> stack;
  float const inf = std::numeric_limits<float>::infinity();
  stack.push_back({0u, {-inf, -inf, -inf}, {inf, inf, inf}});
  while (!stack.empty() && ok) {
    Item const it = stack.back();
    stack.pop_back();
    if (it.node >= r.nodes.size() || nodeSeen[it.node]++) { ok = false; break; }
    Bvh4Node const& n = r.nodes[it.node];
    int const inner = bvhNodeInner(n), cnt = bvhNodeCount(n);
    if (inner > cnt || cnt > 4 || (cnt == 0 && count > 0)) { ok = false; break; }
    for (int k = 0; k < cnt && ok; ++k) {
      Item c{};
      bvhChildBox(n, k, c.lo, c.hi);
      for (int a = 0; a < 3; ++a) {  // nested up to the parent's quantisation step (the child is re-quantised on a finer grid)
        float const step = bvhNodeScale(n, a);
        ok = ok && c.lo[a] <= c.hi[a] && c.lo[a] >= it.lo[a] - step && c.hi[a] <= it.hi[a] + step;
      }
      if (k < inner) {
        c.node = n.childBase + uint32_t(k);
        stack.push_back(c);
        continue;
      }
      size_t const pair = size_t(uint32_t(n.leafRef + uint32_t(k) - kBvhLeafFlag));  // the slot's reference without its flag
      if (pair >= npairs) { ok = false; break; }
      uint32_t const t0 = r.pairTris[2 * pair], t1 = r.pairTris[2 * pair + 1];
      maxLeaf = std::max(maxLeaf, t0 == t1 ? 1 : 2);
      for (int half = 0; half < (t0 == t1 ? 1 : 2) && ok; ++half) {  // a one-triangle leaf repeats its triangle
        uint32_t const t = half ? t1 : t0;
        if (t >= count || seen[t]++) { ok = false; break; }
        for (int v = 0; v < 3 && ok; ++v) {
          float const p[3] = {xs[4 * size_t(t) + v], ys[4 * size_t(t) + v], zs[4 * size_t(t) + v]};
          for (int a = 0; a < 3; ++a) ok = ok && p[a] >= c.lo[a] && p[a] <= c.hi[a];
        }
      }
    }
  }
  for (size_t i = 0; i < count && ok; ++i) ok = seen[i] == 1;
  for (size_t i = 0; i < r.nodes.size() && ok; ++i) ok = nodeSeen[i] == 1;
  if (max_leaf) *max_leaf = maxLeaf;
  return ok && maxLeaf <= kBvhMaxLeafTris ? DMT_OK : DMT_ERR_STATE;
}

static int renderImpl(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1, uint64_t* stats6, int nstats);

int dmt_render(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1) {
  return renderImpl(ctx, sample_offset, spp, x0, y0, x1, y1, nullptr, 0);
}

int dmt_render_stats(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1,
                     uint64_t* stats6) {
  if (!ctx || !stats6) return DMT_ERR_INVALID;
  if (ctx->accel != DMT_ACCEL_BVH) return fail(ctx, DMT_ERR_STATE, "dmt_render_stats: only the BVH path has device counters");
  return renderImpl(ctx, sample_offset, spp, x0, y0, x1, y1, stats6, 6);
}

int dmt_render_profile(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1,
                       uint64_t* stats16) {
  if (!ctx || !stats16) return DMT_ERR_INVALID;
  if (ctx->accel != DMT_ACCEL_BVH) return fail(ctx, DMT_ERR_STATE, "dmt_render_profile: only the BVH path has device counters");
  return renderImpl(ctx, sample_offset, spp, x0, y0, x1, y1, stats16, 16);
}

int dmt_sync(dmt_ctx* ctx) {
  if (!ctx) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return checkErrorFlag(ctx);
}

int dmt_sched_diag(dmt_ctx* ctx, uint64_t* out8, int reset) {
  if (!ctx || !out8) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long d[kSchedDiagWords] = {};
  HIP_TRY(ctx, hipMemcpy(d, ctx->d_schedDiag, sizeof(d), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i) out8[i] = i < kSchedDiagWords ? d[i] : 0;
  out8[6] = ctx->expectedFolds;
  out8[7] = uint64_t(kSlabsPerWave);
  if (reset) {
    HIP_TRY(ctx, hipMemset(ctx->d_schedDiag, 0, sizeof(d)));
    ctx->expectedFolds = 0;
  }
  return DMT_OK;
}

// (re)builds the per-triangle lookup from the host copy of the area-light list
static int rebuildAreaLights(dmt_ctx* ctx) {
  (void)hipFree(ctx->d_areaOf), (void)hipFree(ctx->d_areaTri), (void)hipFree(ctx->d_areaLe);
  ctx->d_areaOf = nullptr, ctx->d_areaTri = nullptr, ctx->d_areaLe = nullptr, ctx->areaCount = 0;
  uint32_t const n = uint32_t(ctx->h_areaTri.size());
  if (n == 0 || !ctx->haveTris) return DMT_OK;
  std::vector<uint32_t> of(ctx->triCount, 0xFFFFFFFFu);
  for (uint32_t k = 0; k < n; ++k) {
    if (ctx->h_areaTri[k] >= ctx->triCount) return fail(ctx, DMT_ERR_INVALID, "area light refers to a triangle outside the uploaded soup");
    of[ctx->h_areaTri[k]] = k;
  }
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_areaOf), of.size() * 4));
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_areaTri), size_t(n) * 4));
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_areaLe), size_t(n) * 12));
  HIP_TRY(ctx, hipMemcpy(ctx->d_areaOf, of.data(), of.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_areaTri, ctx->h_areaTri.data(), size_t(n) * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_areaLe, ctx->h_areaLe.data(), size_t(n) * 12, hipMemcpyHostToDevice));
  ctx->areaCount = n;
  return DMT_OK;
}

int dmt_upload_area_lights(dmt_ctx* ctx, const uint32_t* triangle_index, const float* radiance_rgb, uint32_t count) {
  if (!ctx || (count && (!triangle_index || !radiance_rgb))) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->h_areaTri.assign(triangle_index, triangle_index + count);
  ctx->h_areaLe.assign(radiance_rgb, radiance_rgb + 3 * size_t(count));
  return rebuildAreaLights(ctx);
}

int dmt_upload_textures(dmt_ctx* ctx, const uint8_t* rgba8, uint64_t texel_count, const int32_t* desc3, uint32_t texture_count,
                        const uint32_t* mat_tex4, uint32_t bsdf_count, const float* tri_uv6, uint64_t triangle_count) {
  if (!ctx) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(ctx->d_texRgba), (void)hipFree(ctx->d_texDesc), (void)hipFree(ctx->d_matTex), (void)hipFree(ctx->d_triUv);
  ctx->d_texRgba = nullptr, ctx->d_texDesc = nullptr, ctx->d_matTex = nullptr, ctx->d_triUv = nullptr;
  ctx->texCount = 0, ctx->matTexCount = 0, ctx->triUvCount = 0;
  if (texture_count == 0) return DMT_OK;  // cleared
  if (!rgba8 || !desc3 || !mat_tex4 || !tri_uv6 || texel_count == 0 || bsdf_count == 0 || triangle_count == 0)
    return fail(ctx, DMT_ERR_INVALID, "dmt_upload_textures: null array or zero count");
  for (uint32_t k = 0; k < texture_count; ++k) {  // descriptors must stay inside the texel array
    int64_t const first = desc3[3 * k], w = desc3[3 * k + 1], h = desc3[3 * k + 2];
    if (first < 0 || w <= 0 || h <= 0 || uint64_t(first) + uint64_t(w) * uint64_t(h) > texel_count)
      return fail(ctx, DMT_ERR_INVALID, "dmt_upload_textures: texture descriptor outside the texel array");
  }
  for (uint32_t b = 0; b < bsdf_count; ++b)
    for (int j = 0; j < 3; ++j) {
      uint32_t const t = mat_tex4[4 * size_t(b) + size_t(j)];
      if (t != 0xFFFFFFFFu && t >= texture_count) return fail(ctx, DMT_ERR_INVALID, "dmt_upload_textures: material refers to a texture that does not exist");
    }
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_texRgba), size_t(texel_count) * 4));
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_texDesc), size_t(texture_count) * 12));
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_matTex), size_t(bsdf_count) * 16));
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_triUv), size_t(triangle_count) * 24));
  HIP_TRY(ctx, hipMemcpy(ctx->d_texRgba, rgba8, size_t(texel_count) * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_texDesc, desc3, size_t(texture_count) * 12, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_matTex, mat_tex4, size_t(bsdf_count) * 16, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->d_triUv, tri_uv6, size_t(triangle_count) * 24, hipMemcpyHostToDevice));
  ctx->texCount = texture_count, ctx->matTexCount = bsdf_count, ctx->triUvCount = size_t(triangle_count);
  return DMT_OK;
}

int dmt_envmap_tables(const float* rgb, int width, int height, float* func, float* cdf, float* row_integral,
                      float* marginal_func, float* marginal_cdf, float* marginal_integral) {
  if (!rgb || width < 8 || height < 8 || (width & (width - 1)) || (height & (height - 1)) || !func || !cdf ||
      !row_integral || !marginal_func || !marginal_cdf || !marginal_integral)
    return DMT_ERR_INVALID;
  envmap::Tables const t = envmap::build(rgb, width, height);
  memcpy(func, t.func.data(), t.func.size() * 4), memcpy(cdf, t.cdf.data(), t.cdf.size() * 4);
  memcpy(row_integral, t.rowInt.data(), t.rowInt.size() * 4);
  memcpy(marginal_func, t.mFunc.data(), t.mFunc.size() * 4), memcpy(marginal_cdf, t.mCdf.data(), t.mCdf.size() * 4);
  *marginal_integral = t.mInt;
  return DMT_OK;
}

int dmt_clear_envmap(dmt_ctx* ctx) {
  if (!ctx) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_env) (void)hipFree(ctx->d_env);
  ctx->d_env = nullptr;
  ctx->env = EnvView{};
  return DMT_OK;
}

int dmt_upload_envmap(dmt_ctx* ctx, const float* rgb, int width, int height, const float* quat_xyzw, float scale) {
  if (!ctx || !rgb || !quat_xyzw) return DMT_ERR_INVALID;
  // the reference asserts powers of two and width == 2 * height (core-light.cpp:116); 8 = one AVX2 block of its CDF
  if (width < 8 || height < 8 || (width & (width - 1)) || (height & (height - 1)) || width != 2 * height)
    return fail(ctx, DMT_ERR_INVALID, "dmt_upload_envmap: resolution must be powers of two >= 8 with width == 2 * height");
  float const len = std::sqrt(quat_xyzw[0] * quat_xyzw[0] + quat_xyzw[1] * quat_xyzw[1] + quat_xyzw[2] * quat_xyzw[2] +
                              quat_xyzw[3] * quat_xyzw[3]);
  if (!(len > 0.f)) return fail(ctx, DMT_ERR_INVALID, "dmt_upload_envmap: zero quaternion");
  (void)scale;  // stored by the reference and never applied (core-light.cpp:115, :444-452)
  int rc = dmt_clear_envmap(ctx);
  if (rc) return rc;
  envmap::Tables const t = envmap::build(rgb, width, height);
  size_t const wh = size_t(width) * size_t(height), h = size_t(height);
  size_t const total = 2 * wh + 3 * h + 3 * wh;
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_env), total * sizeof(float)));
  float* p = ctx->d_env;
  EnvView e{};
  auto put = [&](float const* src, size_t n) -> float const* {
    float* dst = p;
    p += n;
    return hipMemcpy(dst, src, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? dst : nullptr;
  };
  e.func = put(t.func.data(), wh), e.cdf = put(t.cdf.data(), wh), e.rowInt = put(t.rowInt.data(), h);
  e.mFunc = put(t.mFunc.data(), h), e.mCdf = put(t.mCdf.data(), h), e.rgb = put(rgb, 3 * wh);
  if (!e.func || !e.cdf || !e.rowInt || !e.mFunc || !e.mCdf || !e.rgb) {
    (void)dmt_clear_envmap(ctx);
    return fail(ctx, DMT_ERR_HIP, "dmt_upload_envmap: copy failed");
  }
  e.mInt = t.mInt, e.w = width, e.h = height;
  e.qx = quat_xyzw[0] / len, e.qy = quat_xyzw[1] / len, e.qz = quat_xyzw[2] / len, e.qw = quat_xyzw[3] / len;
  ctx->env = e;
  return DMT_OK;
}

int dmt_test_envmap(dmt_ctx* ctx, int n, const float* u2, const float* wi_in3, float* wi3, float* pdf, float* uv2,
                    float* Le3, int32_t* ok, float* Le_dir3, float* pdf_dir) {
  if (!ctx || n <= 0 || !u2 || !wi_in3 || !wi3 || !pdf || !uv2 || !Le3 || !ok || !Le_dir3 || !pdf_dir) return DMT_ERR_INVALID;
  if (ctx->env.w <= 0) return fail(ctx, DMT_ERR_STATE, "dmt_test_envmap: no env map uploaded");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t const N = size_t(n);
  float* d = nullptr;  // u2(2) wiIn(3) wi(3) pdf(1) uv(2) Le(3) ok(1) LeDir(3) pdfDir(1) = 19 words per case
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d), N * 19 * sizeof(float)));
  float *du = d, *dwin = d + 2 * N, *dwi = d + 5 * N, *dpdf = d + 8 * N, *duv = d + 9 * N, *dLe = d + 11 * N;
  int32_t* dok = reinterpret_cast<int32_t*>(d + 14 * N);
  float *dLd = d + 15 * N, *dpd = d + 18 * N;
  hipError_t e = hipMemcpy(du, u2, N * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dwin, wi_in3, N * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_test_envmap, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->env, n, du, dwin, dwi, dpdf, duv,
                       dLe, dok, dLd, dpd);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(wi3, dwi, N * 12, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(pdf, dpdf, N * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(uv2, duv, N * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(Le3, dLe, N * 12, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(ok, dok, N * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(Le_dir3, dLd, N * 12, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(pdf_dir, dpd, N * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(ctx, e);
  return DMT_OK;
}

int dmt_set_chunk(dmt_ctx* ctx, uint32_t samples_per_item) {
  if (!ctx) return DMT_ERR_INVALID;
  ctx->chunkSpp = samples_per_item;
  return DMT_OK;
}

int dmt_kernel_time(dmt_ctx* ctx, double* total_ms, uint64_t* launches, int reset) {
  if (!ctx) return DMT_ERR_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (int const rc = checkErrorFlag(ctx)) return rc;  // a time measured on an invalid film is not reported
  for (size_t i = 0; i < ctx->eventsUsed; ++i) {
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
    ctx->accumMs += double(ms);
    ++ctx->accumLaunches;
  }
  ctx->eventsUsed = 0;
  if (total_ms) *total_ms = ctx->accumMs;
  if (launches) *launches = ctx->accumLaunches;
  if (reset) ctx->accumMs = 0.0, ctx->accumLaunches = 0;
  return DMT_OK;
}

int dmt_kernel_info(dmt_ctx* ctx, int* vgprs, int* sgprs, int* lds_bytes, int* blocks_per_cu, int* cu_count) {
  if (!ctx) return DMT_ERR_INVALID;
  hipFuncAttributes attr{};
  bool const useBvh = ctx->accel == DMT_ACCEL_BVH;  // facts of the kernel dmt_render would launch now
  HIP_TRY(ctx, hipFuncGetAttributes(&attr, useBvh ? reinterpret_cast<void const*>(k_megakernel_bvh)
                                                  : reinterpret_cast<void const*>(k_megakernel)));
  if (vgprs) *vgprs = attr.numRegs;
  if (sgprs) *sgprs = 0;
  if (lds_bytes) *lds_bytes = int(attr.sharedSizeBytes);
  if (blocks_per_cu) *blocks_per_cu = blocksPerCuOf(ctx);
  if (cu_count) *cu_count = ctx->cuCount;
  return DMT_OK;
}

// ---- device unit-test entry points --------------------------------------------------------------
int dmt_test_triangle_intersect(dmt_ctx* ctx, const float* xs, const float* ys, const float* zs,
                                size_t count, const float* o3, const float* d3, int32_t* hit, float* t,
                                float* pos3, float* nrm3, float* err3) {
  if (!ctx || !xs || !ys || !zs || !o3 || !d3 || !hit) return DMT_ERR_INVALID;
  if (count == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  float* dx = S.up(xs, 4 * count);
  float* dy = S.up(ys, 4 * count);
  float* dz = S.up(zs, 4 * count);
  int32_t* dh = S.up<int32_t>(nullptr, count);
  float* dt = S.up<float>(nullptr, count);
  float* dp = S.up<float>(nullptr, 3 * count);
  float* dn = S.up<float>(nullptr, 3 * count);
  float* de = S.up<float>(nullptr, 3 * count);
  SCRATCH_CHECK(ctx, dx && dy && dz && dh && dt && dp && dn && de);
  uint32_t const n = uint32_t(count);
  hipLaunchKernelGGL(k_test_tri, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dx, dy, dz, n,
                     f3{o3[0], o3[1], o3[2]}, f3{d3[0], d3[1], d3[2]}, dh, dt, dp, dn, de);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(hit, dh, count * 4, hipMemcpyDeviceToHost));
  if (t) HIP_TRY(ctx, hipMemcpy(t, dt, count * 4, hipMemcpyDeviceToHost));
  if (pos3) HIP_TRY(ctx, hipMemcpy(pos3, dp, count * 12, hipMemcpyDeviceToHost));
  if (nrm3) HIP_TRY(ctx, hipMemcpy(nrm3, dn, count * 12, hipMemcpyDeviceToHost));
  if (err3) HIP_TRY(ctx, hipMemcpy(err3, de, count * 12, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_sampler(dmt_ctx* ctx, int width, int height, int n, const int32_t* pxs, const int32_t* pys,
                     const int32_t* ss, int ndims, int32_t* halton_index, float* pixel2d_out, float* dims) {
  if (!ctx || n < 0 || ndims < 0 || !pxs || !pys || !ss || !halton_index || !pixel2d_out || !dims || width <= 0 || height <= 0)
    return DMT_ERR_INVALID;
  if (n == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  int32_t* dpx = S.up(pxs, size_t(n));
  int32_t* dpy = S.up(pys, size_t(n));
  int32_t* dss = S.up(ss, size_t(n));
  int32_t* dh = S.up<int32_t>(nullptr, size_t(n));
  float* dp2 = S.up<float>(nullptr, 2 * size_t(n));
  float* dd = S.up<float>(nullptr, size_t(n) * size_t(ndims));
  SCRATCH_CHECK(ctx, dpx && dpy && dss && dh && dp2 && dd);
  hipLaunchKernelGGL(k_test_sampler, dim3((n + 63) / 64), dim3(64), 0, ctx->stream,
                     computeSamplerParams(width, height), n, dpx, dpy, dss, ndims, dh, dp2, dd);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(halton_index, dh, size_t(n) * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(pixel2d_out, dp2, size_t(n) * 8, hipMemcpyDeviceToHost));
  if (ndims) HIP_TRY(ctx, hipMemcpy(dims, dd, size_t(n) * size_t(ndims) * 4, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_camera_rays(dmt_ctx* ctx, int n, const int32_t* pxs, const int32_t* pys, const int32_t* ss,
                         float* o3, float* d3) {
  if (!ctx || n < 0 || !pxs || !pys || !ss || !o3 || !d3) return DMT_ERR_INVALID;
  if (!ctx->haveCamera) return fail(ctx, DMT_ERR_STATE, "dmt_test_camera_rays: set the camera first");
  if (n == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  int32_t* dpx = S.up(pxs, size_t(n));
  int32_t* dpy = S.up(pys, size_t(n));
  int32_t* dss = S.up(ss, size_t(n));
  float* dO = S.up<float>(nullptr, 3 * size_t(n));
  float* dD = S.up<float>(nullptr, 3 * size_t(n));
  SCRATCH_CHECK(ctx, dpx && dpy && dss && dO && dD);
  hipLaunchKernelGGL(k_test_camera, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->xf, ctx->sp, n, dpx,
                     dpy, dss, dO, dD);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(o3, dO, size_t(n) * 12, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(d3, dD, size_t(n) * 12, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_bsdf(dmt_ctx* ctx, const void* bsdf32, int n, const float* ns3, const float* wo3,
                  const float* u2, const float* uc, const float* wi_eval3, float* prepared12,
                  float* sample10, float* eval4) {
  if (!ctx || n < 0 || !bsdf32 || !ns3 || !wo3 || !u2 || !uc || !wi_eval3 || !prepared12 || !sample10 || !eval4)
    return DMT_ERR_INVALID;
  if (n == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  float* dns = S.up(ns3, 3 * size_t(n));
  float* dwo = S.up(wo3, 3 * size_t(n));
  float* du2 = S.up(u2, 2 * size_t(n));
  float* duc = S.up(uc, size_t(n));
  float* dwi = S.up(wi_eval3, 3 * size_t(n));
  float* dp = S.up<float>(nullptr, 12 * size_t(n));
  float* ds = S.up<float>(nullptr, 10 * size_t(n));
  float* de = S.up<float>(nullptr, 4 * size_t(n));
  SCRATCH_CHECK(ctx, dns && dwo && du2 && duc && dwi && dp && ds && de);
  Rec32 rec;
  memcpy(&rec, bsdf32, 32);
  hipLaunchKernelGGL(k_test_bsdf, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, rec, n, dns, dwo, du2, duc,
                     dwi, dp, ds, de);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(prepared12, dp, size_t(n) * 48, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(sample10, ds, size_t(n) * 40, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(eval4, de, size_t(n) * 16, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_light(dmt_ctx* ctx, const void* light32, int n, const float* pos3, const float* nrm3,
                   const float* u2, const int32_t* had_transmission, float* out14) {
  if (!ctx || n < 0 || !light32 || !pos3 || !nrm3 || !u2 || !had_transmission || !out14) return DMT_ERR_INVALID;
  if (n == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  float* dp = S.up(pos3, 3 * size_t(n));
  float* dn = S.up(nrm3, 3 * size_t(n));
  float* du = S.up(u2, 2 * size_t(n));
  int32_t* dh = S.up(had_transmission, size_t(n));
  float* dout = S.up<float>(nullptr, 14 * size_t(n));
  SCRATCH_CHECK(ctx, dp && dn && du && dh && dout);
  Rec32 rec;
  memcpy(&rec, light32, 32);
  hipLaunchKernelGGL(k_test_light, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, rec, n, dp, dn, du, dh, dout);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(out14, dout, size_t(n) * 56, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_half(dmt_ctx* ctx, int n, const float* f_in, uint16_t* h_out, const uint16_t* h_in, float* f_out) {
  if (!ctx || n < 0) return DMT_ERR_INVALID;
  if (n == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  float* dfi = f_in ? S.up(f_in, size_t(n)) : nullptr;
  uint16_t* dho = h_out ? S.up<uint16_t>(nullptr, size_t(n)) : nullptr;
  uint16_t* dhi = h_in ? S.up(h_in, size_t(n)) : nullptr;
  float* dfo = f_out ? S.up<float>(nullptr, size_t(n)) : nullptr;
  if ((f_in && !dfi) || (h_out && !dho) || (h_in && !dhi) || (f_out && !dfo))
    return fail(ctx, DMT_ERR_HIP, "scratch allocation / upload failed");
  hipLaunchKernelGGL(k_test_half, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, dfi, dho, dhi, dfo);
  int rc = finishTest(ctx);
  if (rc) return rc;
  if (h_out && f_in) HIP_TRY(ctx, hipMemcpy(h_out, dho, size_t(n) * 2, hipMemcpyDeviceToHost));
  if (f_out && h_in) HIP_TRY(ctx, hipMemcpy(f_out, dfo, size_t(n) * 4, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_trace_samples(dmt_ctx* ctx, int n, const int32_t* pxs, const int32_t* pys, const int32_t* ss,
                           float* L3) {
  if (!ctx || n < 0 || !pxs || !pys || !ss || !L3) return DMT_ERR_INVALID;
  if (!(ctx->haveTris && ctx->haveBsdfs && ctx->haveLights && ctx->haveCamera))
    return fail(ctx, DMT_ERR_STATE, "dmt_test_trace_samples: scene/camera not set");
  if (ctx->triCount > 0 && ctx->maxMatId >= ctx->bsdfCount)
    return fail(ctx, DMT_ERR_INVALID, "dmt_test_trace_samples: material index outside the BSDF array");
  if (n == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  int32_t* dpx = S.up(pxs, size_t(n));
  int32_t* dpy = S.up(pys, size_t(n));
  int32_t* dss = S.up(ss, size_t(n));
  float* dL = S.up<float>(nullptr, 3 * size_t(n));
  SCRATCH_CHECK(ctx, dpx && dpy && dss && dL);
  bool const useBvh = ctx->accel == DMT_ACCEL_BVH;
  size_t const threads = size_t((n + 63) / 64) * 64;
  if (useBvh) {
    if (!ctx->haveBvh) return fail(ctx, DMT_ERR_STATE, "dmt_test_trace_samples: BVH not built");
    int const rcO = ensureOverflow(ctx, threads);
    if (rcO) return rcO;
  }
  hipLaunchKernelGGL(k_test_trace, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, baseParams(ctx, threads), useBvh, n,
                     dpx, dpy, dss, dL);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(L3, dL, size_t(n) * 12, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_trace_log(dmt_ctx* ctx, int px, int py, int s, float* rec12, int cap, int* n_out, float* L3) {
  if (!ctx || !rec12 || cap <= 0 || !n_out || !L3) return DMT_ERR_INVALID;
  if (!(ctx->haveTris && ctx->haveBsdfs && ctx->haveLights && ctx->haveCamera))
    return fail(ctx, DMT_ERR_STATE, "dmt_test_trace_log: scene/camera not set");
  if (ctx->triCount > 0 && ctx->maxMatId >= ctx->bsdfCount)
    return fail(ctx, DMT_ERR_INVALID, "dmt_test_trace_log: material index outside the BSDF array");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  float* dr = S.up<float>(nullptr, 12 * size_t(cap));
  int* dn = S.up<int>(nullptr, 1);
  float* dL = S.up<float>(nullptr, 3);
  SCRATCH_CHECK(ctx, dr && dn && dL);
  hipLaunchKernelGGL(k_test_trace_log, dim3(1), dim3(64), 0, ctx->stream, baseParams(ctx, 64), px, py, s, dr, cap, dn,
                     dL);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(rec12, dr, size_t(cap) * 48, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(n_out, dn, 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(L3, dL, 12, hipMemcpyDeviceToHost));
  return DMT_OK;
}

int dmt_test_closest_hit(dmt_ctx* ctx, int nrays, const float* o3, const float* d3, int32_t* tri_index, float* t) {
  if (!ctx || nrays < 0 || !o3 || !d3 || !tri_index || !t) return DMT_ERR_INVALID;
  if (!ctx->haveTris) return fail(ctx, DMT_ERR_STATE, "dmt_test_closest_hit: upload triangles first");
  if (nrays == 0) return DMT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  Scratch S(ctx);
  float* dO = S.up(o3, 3 * size_t(nrays));
  float* dD = S.up(d3, 3 * size_t(nrays));
  int32_t* di = S.up<int32_t>(nullptr, size_t(nrays));
  float* dt = S.up<float>(nullptr, size_t(nrays));
  SCRATCH_CHECK(ctx, dO && dD && di && dt);
  bool const useBvh = ctx->accel == DMT_ACCEL_BVH;
  size_t const threads = size_t((nrays + 63) / 64) * 64;
  if (useBvh) {
    if (!ctx->haveBvh) return fail(ctx, DMT_ERR_STATE, "dmt_test_closest_hit: BVH not built");
    int const rcO = ensureOverflow(ctx, threads);
    if (rcO) return rcO;
  }
  hipLaunchKernelGGL(k_test_closest, dim3((nrays + 63) / 64), dim3(64), 0, ctx->stream, baseParams(ctx, threads),
                     useBvh, nrays, dO, dD, di, dt);
  int rc = finishTest(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(tri_index, di, size_t(nrays) * 4, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(t, dt, size_t(nrays) * 4, hipMemcpyDeviceToHost));
  return DMT_OK;
}

}  // extern "C"
