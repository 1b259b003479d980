// light_tree.hpp -- light BVH for many-light scenes (second half): importance-driven choice of the NEE light
// instead of the uniform pick of the reference's megakernel (T/megakernel/megakernel.cu:170-173).
//
// What it follows: the reference's CPU-side light tree, src/core/public/core-light-tree-builder.h:17-110 and
// src/core/private/core-light-tree-builder.cpp (after Conty & Kulla, "Importance Sampling of Many Lights with Adaptive Tree
// Splitting", 2018): LightBounds (box + emitted flux), lbImportance (:98-146: flux x clamped cosine towards the cluster /
// clamped squared distance), a binary tree split by an area x flux cost (:246-263), selection by walking down the tree and
// choosing a child in proportion to its importance (selectLightsFromSplit :496-539), and the per-light probability as the
// product of those choices (lightSelectionPMF :541-557).  That code is experimental, disabled in the reference's build and
// has no test or golden output: PARITY UNPINNED; the uniform pick stays the default (parity mode) and this is opt-in
// (dmt_set_light_sampling).  Differences, all deliberate and needed for an unbiased estimator on THIS path:
//   * every light is omnidirectional for the tree.  The megakernel's spot light does not attenuate outside its cone
//     (spotLightAttenuation's smoothstep is 1 for every input, CC/public/cuda-core/light.cuh:77-81), so the
//     reference's cone test (importance 0 when cosTheta_p <= cosTheta_e) would starve directions the light does reach;
//     with one cone for all lights the orientation factor M_omega of the split cost is a constant and drops out;
//   * one light per bounce: the reference's adaptive split returns up to four lights (= four shadow rays); the megakernel
//     traces one shadow ray per bounce, so the cut is the root;
//   * the build sorts the node's lights along the longest axis and sweeps every split (the reference bins 32 positions and,
//     as written, accumulates the bins over ALL lights of the scene rather than the node's, :335-349); costs are compared
//     in double with a relative margin, so that the oracle's independent builder makes the same tree;
//   * bounding-sphere distance uses |centre - p|^2 (the reference writes dot(centre, p), :90).
// Shared by the host (builder, per-light probabilities for tests) and the device (selection): plain functions.
#pragma once

#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#if defined(__HIPCC__)
#define DMT_LT_HD __host__ __device__ inline
#else
#define DMT_LT_HD inline
#endif

namespace dmt {

struct LightTreeNode {  // 32 B
  float lo[3];
  float phi;     // emitted flux of the cluster (relative units)
  float hi[3];
  uint32_t ref;  // leaf: 0x80000000 | light index; inner: index of the left child (right = left + 1)
};
static_assert(sizeof(LightTreeNode) == 32, "light tree node");
constexpr uint32_t kLightLeaf = 0x80000000u;

// lbImportance (core-light-tree-builder.cpp:98-146) for an omnidirectional cluster
DMT_LT_HD float lt_importance(LightTreeNode const& nd, float px, float py, float pz, float nx, float ny, float nz) {
  float const cx = 0.5f * (nd.lo[0] + nd.hi[0]), cy = 0.5f * (nd.lo[1] + nd.hi[1]), cz = 0.5f * (nd.lo[2] + nd.hi[2]);
  float const dx = nd.hi[0] - nd.lo[0], dy = nd.hi[1] - nd.lo[1], dz = nd.hi[2] - nd.lo[2];
  float const halfDiag = 0.5f * sqrtf(dx * dx + dy * dy + dz * dz);
  float wx = px - cx, wy = py - cy, wz = pz - cz;
  float const d2 = wx * wx + wy * wy + wz * wz;
  // :112-113 (distance squared against a length, as written); the floor keeps a zero-radius light AT the shading point from
  // producing inf importance -> NaN probabilities (round-2 advisor)
  float const distSqr = fmaxf(fmaxf(d2, halfDiag), 1e-20f);
  // sine / cosine of the angle the cluster's bounding sphere subtends (:84-96)
  float sinB = 0.f, cosB = -1.f;  // inside the sphere: the cluster fills the hemisphere
  if (d2 >= halfDiag * halfDiag && d2 > 0.f) {
    float const s2 = (halfDiag * halfDiag) / d2;
    sinB = sqrtf(s2), cosB = sqrtf(fmaxf(0.f, 1.f - s2));
  }
  float cosI = 1.f;
  if (d2 > 0.f) {
    float const inv = 1.f / sqrtf(d2);
    cosI = fabsf((wx * nx + wy * ny + wz * nz) * inv);  // absDot(wi, n): transmission reaches lights behind the surface
  }
  float const sinI = sqrtf(fmaxf(0.f, 1.f - cosI * cosI));
  float const cosIB = cosI > cosB ? 1.f : cosI * cosB + sinI * sinB;  // cosSubClamped
  return fmaxf(nd.phi * cosIB / distSqr, 0.f);
}

// Walk from the root to one light, choosing a child in proportion to its importance (sampleDiscrete with remapped u).
// Returns the light index and its probability, or -1 when both children of some node have importance 0.
DMT_LT_HD int lt_select(LightTreeNode const* nodes, float px, float py, float pz, float nx, float ny, float nz, float u, float& pmf) {
  uint32_t at = 0;
  pmf = 1.f;
  for (int guard = 0; guard < 64; ++guard) {
    LightTreeNode const nd = nodes[at];
    if (nd.ref & kLightLeaf) return int(nd.ref & ~kLightLeaf);
    float const i0 = lt_importance(nodes[nd.ref], px, py, pz, nx, ny, nz);
    float const i1 = lt_importance(nodes[nd.ref + 1u], px, py, pz, nx, ny, nz);
    float const sum = i0 + i1;
    if (!(sum > 0.f)) return -1;
    float const p0 = i0 / sum;
    if (u < p0) {
      pmf *= p0;
      u = fminf(u / p0, 0.99999994f);
      at = nd.ref;
    } else {
      pmf *= 1.f - p0;
      u = fminf((u - p0) / (1.f - p0), 0.99999994f);
      at = nd.ref + 1u;
    }
  }
  return -1;
}

namespace light_tree {

struct Item {
  float pos[3], radius, phi;
  uint32_t index;
};

// flux of a packed light record (CC/public/cuda-core/light.cuh:10-49): luminance of its fp16 intensity x 4 pi; point and
// spot lights only (type 0 / 1).  `half` decodes an fp16 bit pattern.
template <class HalfToFloat>
inline bool itemOf(uint8_t const* rec32, uint32_t index, HalfToFloat half, Item& it) {
  uint16_t h[4];
  std::memcpy(h, rec32, 8);
  uint16_t const type = h[3];
  if (type != 0 && type != 1) return false;
  float const lum = 0.2126f * half(h[0]) + 0.7152f * half(h[1]) + 0.0722f * half(h[2]);
  std::memcpy(it.pos, rec32 + 8, 12);
  uint16_t r;
  std::memcpy(&r, rec32 + (type == 0 ? 20 : 28), 2);
  it.radius = std::fmax(half(r), 0.f);
  it.phi = 4.f * 3.14159265358979323846f * std::fmax(lum, 0.f);
  it.index = index;
  return true;
}

inline void boundsOf(std::vector<Item> const& items, size_t a, size_t b, float lo[3], float hi[3], float& phi) {
  phi = 0.f;
  for (int k = 0; k < 3; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;
  for (size_t i = a; i < b; ++i) {
    for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], items[i].pos[k] - items[i].radius), hi[k] = std::max(hi[k], items[i].pos[k] + items[i].radius);
    phi += items[i].phi;
  }
}
inline double areaOf(float const lo[3], float const hi[3]) {
  double const dx = double(hi[0]) - lo[0], dy = double(hi[1]) - lo[1], dz = double(hi[2]) - lo[2];
  return 2.0 * (dx * dy + dy * dz + dz * dx);
}

// nodes[0] = root; empty input -> empty vector.  depth (optional) = levels.
inline std::vector<LightTreeNode> build(std::vector<Item> items, int* depthOut = nullptr) {
  std::vector<LightTreeNode> nodes;
  if (depthOut) *depthOut = 0;
  if (items.empty()) return nodes;
  struct Work {
    uint32_t node;
    size_t a, b;
    int depth;
  };
  nodes.emplace_back();
  std::vector<Work> stack{{0u, 0, items.size(), 1}};
  while (!stack.empty()) {
    Work const w = stack.back();
    stack.pop_back();
    if (depthOut) *depthOut = std::max(*depthOut, w.depth);
    LightTreeNode nd{};
    boundsOf(items, w.a, w.b, nd.lo, nd.hi, nd.phi);
    if (w.b - w.a == 1) {
      nd.ref = kLightLeaf | items[w.a].index;
      nodes[w.node] = nd;
      continue;
    }
    int axis = 0;
    for (int k = 1; k < 3; ++k)
      if (nd.hi[k] - nd.lo[k] > nd.hi[axis] - nd.lo[axis]) axis = k;
    std::stable_sort(items.begin() + long(w.a), items.begin() + long(w.b), [axis](Item const& x, Item const& y) {
      return x.pos[axis] < y.pos[axis] || (x.pos[axis] == y.pos[axis] && x.index < y.index);
    });
    // sweep every split: cost = flux_L x area_L + flux_R x area_R (the parent's factors are common to all candidates)
    size_t const n = w.b - w.a;
    std::vector<double> rightCost(n, 0.0);
    {
      float lo[3], hi[3], phi;
      for (size_t k = n - 1; k >= 1; --k) {
        boundsOf(items, w.a + k, w.b, lo, hi, phi);  // O(n^2) per node: light counts are small (tens .. thousands)
        rightCost[k] = double(phi) * areaOf(lo, hi);
      }
    }
    size_t best = n / 2;
    double bestCost = INFINITY;
    for (size_t k = 1; k < n; ++k) {
      float lo[3], hi[3], phi;
      boundsOf(items, w.a, w.a + k, lo, hi, phi);
      double const cost = double(phi) * areaOf(lo, hi) + rightCost[k];
      if (cost < bestCost * (1.0 - 1e-9)) bestCost = cost, best = k;  // first clearly better split wins (deterministic across builders)
    }
    uint32_t const left = uint32_t(nodes.size());
    nodes.emplace_back(), nodes.emplace_back();
    nd.ref = left;
    nodes[w.node] = nd;
    stack.push_back({left + 1u, w.a + best, w.b, w.depth + 1});
    stack.push_back({left, w.a, w.a + best, w.depth + 1});
  }
  return nodes;
}

// probability of every light at (p, n): product of the child choices on its root path (lightSelectionPMF); lights the
// walk cannot reach get 0.  out has one entry per light INDEX (max index + 1 entries expected by the caller).
inline void pmfs(std::vector<LightTreeNode> const& nodes, float const p[3], float const n[3], float* out, size_t count) {
  for (size_t i = 0; i < count; ++i) out[i] = 0.f;
  if (nodes.empty()) return;
  struct W {
    uint32_t node;
    float pmf;
  };
  std::vector<W> stack{{0u, 1.f}};
  while (!stack.empty()) {
    W const w = stack.back();
    stack.pop_back();
    LightTreeNode const& nd = nodes[w.node];
    if (nd.ref & kLightLeaf) {
      uint32_t const li = nd.ref & ~kLightLeaf;
      if (li < count) out[li] = w.pmf;
      continue;
    }
    float const i0 = lt_importance(nodes[nd.ref], p[0], p[1], p[2], n[0], n[1], n[2]);
    float const i1 = lt_importance(nodes[nd.ref + 1u], p[0], p[1], p[2], n[0], n[1], n[2]);
    float const sum = i0 + i1;
    if (!(sum > 0.f)) continue;
    float const p0 = i0 / sum;
    stack.push_back({nd.ref, w.pmf * p0});
    stack.push_back({nd.ref + 1u, w.pmf * (1.f - p0)});
  }
}

}  // namespace light_tree
}  // namespace dmt
