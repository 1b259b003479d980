// bvh.hpp -- 4-wide BVH: node layout (shared host/device) and the host builder.
//
// Semantics follow the reference's only BVH, the CPU renderer's: binned-SAH splits on
// the longest centroid axis, small leaves, wide nodes obtained by collapsing binary splits
// (src/core/private/core-bvh-builder.cpp:58-223 builds an 8-ary tree the same way, 16-128 bins,
// leaves <= 7).  Layout and traversal are designed for gfx950 instead of AVX2: 128-byte nodes read
// as eight 16-byte per-lane loads, child boxes as SoA so a lane tests the four children with plain
// VALU min/max, leaves of <= 2 triangles stored as one interleaved pair.
//
// Correctness contract (tests/test_parity_gpu.py::test_bvh_*): traversal returns exactly the
// brute-force closest hit -- same triangle (lowest ORIGINAL index on equal t) and bit-identical
// (t,u,v), because the same Moeller-Trumbore routine runs on the same triangle record -- and the same
// any-hit answer.  Boxes are padded so that a ray accepted by the triangle test (which has its own
// 1e-7 barycentric slack and rounding) can never be culled by a box test.
#pragma once

#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace dmt {

constexpr uint32_t kBvhLeafFlag = 0x80000000u;
constexpr uint32_t kBvhEmpty = 0xFFFFFFFFu;
constexpr int kBvhMaxLeafTris = 2;   // one triangle pair per leaf: measured best (1 M random triangles: 405 vs 391 Msamples/s for 4)
constexpr int kBvhMaxDepth = 48;   // depth bound (binary levels, hence also 4-wide levels) enforced by the builder
#ifndef DMT_BVH_LDS_STACK
#define DMT_BVH_LDS_STACK 16  // tests build a variant with 2 so that every non-trivial traversal runs through the global overflow stack
#endif
constexpr int kBvhLdsStack = DMT_BVH_LDS_STACK;   // traversal stack entries kept in LDS per lane
constexpr int kBvhOverflowStack = 3 * kBvhMaxDepth - kBvhLdsStack;  // the rest, per lane, in global memory

// child reference: kBvhEmpty | inner node index | kBvhLeafFlag | (count-1) << 28 | first
// (builder: count triangles from slot `first`; device: count triangle PAIRS from pair `first`)
inline uint32_t bvhLeafRef(uint32_t first, uint32_t count) { return kBvhLeafFlag | ((count - 1u) << 28) | first; }

// Leaf storage on the device: two triangles interleaved component by component, so that the 16-byte loads
// deliver (first, second) register pairs ready for packed math; one 128-byte cache line per pair.  A leaf of
// 1-4 triangles is 1-2 consecutive pairs; an odd leaf repeats its last triangle (same original index, so the
// repeated test can never win the tie-break against itself).
struct TriPair {  // 128 B
  float p0x[2], p0y[2], p0z[2], e0x[2], e0y[2], e0z[2], e1x[2], e1y[2], e1z[2];
  uint32_t orig[2];  // ORIGINAL triangle indices (brute-force tie-break, shading)
  uint32_t pad[12];
};
static_assert(sizeof(TriPair) == 128, "pair size");

struct Bvh4Node {  // 128 B
  float minx[4], miny[4], minz[4];
  float maxx[4], maxy[4], maxz[4];
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "node size");


namespace bvh_build {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; ++a) lo[a] = std::numeric_limits<float>::infinity(), hi[a] = -std::numeric_limits<float>::infinity();
  }
  void grow(float const p[3]) {
    for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], p[a]), hi[a] = std::max(hi[a], p[a]);
  }
  void grow(Box const& b) {
    for (int a = 0; a < 3; ++a) lo[a] = std::min(lo[a], b.lo[a]), hi[a] = std::max(hi[a], b.hi[a]);
  }
  float area() const {
    float const dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.f)) return 0.f;
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
};

struct Node2 {  // binary build node
  Box box;
  int left = -1, right = -1;  // inner
  uint32_t first = 0, count = 0;  // leaf range in `order`
  bool leaf() const { return left < 0; }
};

struct Builder {
  std::vector<Box> triBox;
  std::vector<float> centroid;  // 3 per triangle
  std::vector<uint32_t> order;
  std::vector<Node2> nodes;

  static int ceilLog2(uint32_t v) {
    int l = 0;
    while ((1u << l) < v) ++l;
    return l;
  }

  // depthBudget: binary levels still allowed below this node (guarantees the 4-wide depth bound)
  int build(uint32_t first, uint32_t count, int depthBudget) {
    int const id = int(nodes.size());
    nodes.emplace_back();
    Box box, cbox;
    box.reset(), cbox.reset();
    for (uint32_t i = first; i < first + count; ++i) {
      box.grow(triBox[order[i]]);
      cbox.grow(&centroid[3 * order[i]]);
    }
    nodes[id].box = box;
    if (count <= uint32_t(kBvhMaxLeafTris)) {
      nodes[id].first = first, nodes[id].count = count;
      return id;
    }
    int axis = 0;
    float ext = cbox.hi[0] - cbox.lo[0];
    for (int a = 1; a < 3; ++a)
      if (cbox.hi[a] - cbox.lo[a] > ext) ext = cbox.hi[a] - cbox.lo[a], axis = a;
    uint32_t mid = first + count / 2;
    bool const mustBalance = ceilLog2((count + kBvhMaxLeafTris - 1) / kBvhMaxLeafTris) >= depthBudget;
    bool split = false;
    if (!mustBalance && ext > 0.f) {  // binned SAH, best of the three axes
      constexpr int B = 16;
      float bestCost = std::numeric_limits<float>::infinity();
      int bestSplit = -1, bestAxis = -1;
      float bestK = 0.f;
      for (int ax = 0; ax < 3; ++ax) {
        float const e = cbox.hi[ax] - cbox.lo[ax];
        if (!(e > 0.f)) continue;
        Box bb[B];
        uint32_t bc[B] = {};
        for (auto& b : bb) b.reset();
        float const k = float(B) * (1.f - 1e-6f) / e;
        for (uint32_t i = first; i < first + count; ++i) {
          int b = int((centroid[3 * order[i] + ax] - cbox.lo[ax]) * k);
          b = b < 0 ? 0 : (b >= B ? B - 1 : b);
          bb[b].grow(triBox[order[i]]);
          ++bc[b];
        }
        float rightArea[B];
        uint32_t rightCnt[B];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = B - 1; b > 0; --b) {
          acc.grow(bb[b]);
          c += bc[b];
          rightArea[b] = acc.area(), rightCnt[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < B - 1; ++b) {
          acc.grow(bb[b]);
          c += bc[b];
          if (c == 0 || rightCnt[b + 1] == 0) continue;
          float const cost = acc.area() * float(c) + rightArea[b + 1] * float(rightCnt[b + 1]);
          if (cost < bestCost) bestCost = cost, bestSplit = b, bestAxis = ax, bestK = k;
        }
      }
      if (bestSplit >= 0) {
        int const ax = bestAxis;
        float const lo = cbox.lo[ax];
        auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
          int b = int((centroid[3 * t + ax] - lo) * bestK);
          b = b < 0 ? 0 : (b >= B ? B - 1 : b);
          return b <= bestSplit;
        });
        mid = uint32_t(it - order.begin());
        split = mid > first && mid < first + count;
      }
    }
    if (!split) {  // median split on the same axis (degenerate centroids, or depth budget exhausted)
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint32_t a, uint32_t b) {
                         float const ca = centroid[3 * a + axis], cb = centroid[3 * b + axis];
                         return ca < cb || (ca == cb && a < b);
                       });
    }
    int const l = build(first, mid - first, depthBudget - 1);
    int const r = build(mid, first + count - mid, depthBudget - 1);
    nodes[id].left = l, nodes[id].right = r;
    return id;
  }
};

struct Result {
  std::vector<Bvh4Node> nodes;
  std::vector<uint32_t> slotToTri;  // triangle stored in slot s (leaf order)
  int depth = 0;
};

// xs/ys/zs: the reference's SoA (4 floats per triangle: c0, c1, c2, pad)
inline Result build(float const* xs, float const* ys, float const* zs, uint32_t n) {
  Result out;
  Builder b;
  b.triBox.resize(n), b.centroid.resize(3 * size_t(n)), b.order.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    Box bx;
    bx.reset();
    for (int v = 0; v < 3; ++v) {
      float const p[3] = {xs[4 * size_t(i) + v], ys[4 * size_t(i) + v], zs[4 * size_t(i) + v]};
      bx.grow(p);
    }
    // padding: far above the rounding of the triangle and slab tests (~1e-7 relative), far below
    // anything that costs traversal work
    for (int a = 0; a < 3; ++a) {
      float const m = std::max(std::fabs(bx.lo[a]), std::fabs(bx.hi[a]));
      float const pad = 1e-5f * (bx.hi[a] - bx.lo[a]) + 4e-6f * m + 1e-7f;
      bx.lo[a] -= pad, bx.hi[a] += pad;
    }
    b.triBox[i] = bx;
    for (int a = 0; a < 3; ++a) b.centroid[3 * size_t(i) + a] = 0.5f * (bx.lo[a] + bx.hi[a]);
    b.order[i] = i;
  }
  if (n == 0) {  // a root whose children are all empty
    Bvh4Node root;
    for (int k = 0; k < 4; ++k) {
      root.minx[k] = root.miny[k] = root.minz[k] = std::numeric_limits<float>::infinity();
      root.maxx[k] = root.maxy[k] = root.maxz[k] = -std::numeric_limits<float>::infinity();
      root.child[k] = kBvhEmpty, root.pad[k] = 0;
    }
    out.nodes.push_back(root);
    return out;
  }
  b.nodes.reserve(size_t(n));
  int const root2 = b.build(0, n, kBvhMaxDepth);
  out.slotToTri = b.order;

  // collapse binary splits into 4-wide nodes: repeatedly open the inner child of largest area
  struct Work {
    int node2;
    uint32_t node4;
    int depth;
  };
  std::vector<Work> stack;
  out.nodes.emplace_back();
  stack.push_back({root2, 0u, 1});
  while (!stack.empty()) {
    Work const w = stack.back();
    stack.pop_back();
    out.depth = std::max(out.depth, w.depth);
    int kids[4];
    int nk = 0;
    if (b.nodes[w.node2].leaf()) {
      kids[nk++] = w.node2;
    } else {
      kids[nk++] = b.nodes[w.node2].left;
      kids[nk++] = b.nodes[w.node2].right;
      while (nk < 4) {
        int pick = -1;
        float bestA = -1.f;
        for (int k = 0; k < nk; ++k)
          if (!b.nodes[kids[k]].leaf() && b.nodes[kids[k]].box.area() > bestA) bestA = b.nodes[kids[k]].box.area(), pick = k;
        if (pick < 0) break;
        int const open = kids[pick];
        kids[pick] = b.nodes[open].left;
        kids[nk++] = b.nodes[open].right;
      }
    }
    Bvh4Node nd;
    for (int k = 0; k < 4; ++k) {
      nd.pad[k] = 0;
      if (k >= nk) {
        nd.minx[k] = nd.miny[k] = nd.minz[k] = std::numeric_limits<float>::infinity();
        nd.maxx[k] = nd.maxy[k] = nd.maxz[k] = -std::numeric_limits<float>::infinity();
        nd.child[k] = kBvhEmpty;
        continue;
      }
      Node2 const& c = b.nodes[kids[k]];
      nd.minx[k] = c.box.lo[0], nd.miny[k] = c.box.lo[1], nd.minz[k] = c.box.lo[2];
      nd.maxx[k] = c.box.hi[0], nd.maxy[k] = c.box.hi[1], nd.maxz[k] = c.box.hi[2];
      if (c.leaf()) {
        nd.child[k] = bvhLeafRef(c.first, c.count);
      } else {
        uint32_t const id4 = uint32_t(out.nodes.size());
        out.nodes.emplace_back();
        nd.child[k] = id4;
        stack.push_back({kids[k], id4, w.depth + 1});
      }
    }
    out.nodes[w.node4] = nd;
  }
  return out;
}

}  // namespace bvh_build


}  // namespace dmt
