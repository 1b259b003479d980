// bvh.hpp -- 4-wide BVH: node / leaf layout (shared host/device) and the host builder.
//
// Semantics follow the reference's only BVH, the CPU renderer's: binned-SAH splits,
// small leaves, wide nodes obtained by collapsing binary splits (src/core/private/core-bvh-builder.cpp:58-223
// builds an 8-ary tree the same way, 16-128 bins, leaves <= 7).  The LAYOUT is designed for what bounds per-lane
// traversal on gfx950: the vector-memory front end.  tools/ubench/gather.hip (a dependent per-lane gather like a
// traversal's) measures, chip-wide, for records served by L2:
//     128-byte record, 7 x 16-byte loads per lane     88 G lane-steps/s   (round 1's node)
//     128-byte record, 5 loads                       121                  (round 1's triangle pair)
//      64-byte slot, 4 loads                         214
//      64-byte slot, 3 loads                         290                  (this node)
//      80-byte record, 5 loads                       178                  (this triangle pair)
// i.e. a step costs ~max(0.7 x loads, 64-byte sectors touched x 64 / 26) clocks of the CU's L1 path, whatever the
// occupancy; round 1's kernel ran at 75 % of the first two rates.  So:
//   * Bvh4Node = 48 bytes of payload in a 64-byte slot (one sector, three loads): the children's boxes are
//     quantised to 8 bits per plane relative to the node's own box (origin + power-of-two scale per axis) and child
//     references are implicit (inner children contiguous from childBase, leaves contiguous from the pair leafRef names).
//   * TriPair = 80 bytes (five loads): two triangles interleaved component by component for packed math.
//
// Correctness contract (tests/test_parity_gpu.py::test_bvh_*): traversal returns exactly the brute-force closest hit
// -- same triangle (lowest ORIGINAL index on equal t) and bit-identical (t,u,v), because the same Moeller-Trumbore
// routine runs on the same fp32 triangle record -- and the same any-hit answer.  Boxes only cull: triangle boxes are
// padded so that a ray the triangle test accepts can never be culled, and quantisation only ever GROWS a box (lo rounded
// down, hi rounded up, in exact arithmetic: origin and scale are floats, q * scale is exact).
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

// traversal-stack entry / current position: kBvhEmpty | inner node index | kBvhLeafFlag | pair index
inline uint32_t bvhLeafRef(uint32_t pair) { return kBvhLeafFlag | pair; }

// Leaf storage on the device: two triangles interleaved component by component, so that the 16-byte loads
// deliver (first, second) register pairs ready for packed math.  A leaf is ONE pair; a one-triangle leaf repeats its
// triangle (same original index, so the repeated test can never win the tie-break against itself).
struct TriPair {  // 80 B = five 16-byte loads
  float p0x[2], p0y[2], p0z[2], e0x[2], e0y[2], e0z[2], e1x[2], e1y[2], e1z[2];
  uint32_t orig[2];  // ORIGINAL triangle indices (brute-force tie-break, shading)
};
static_assert(sizeof(TriPair) == 80, "pair size");

// Inner node.  Child k (k < count) has the box  [origin + qlo_k * scale, origin + qhi_k * scale]  per axis, with
// scale_axis = 2^(exp_axis - 127) and the q's the k-th BYTES of the six plane words.  Children 0 .. inner-1 are the inner
// nodes childBase + k; children inner .. count-1 are the leaves (triangle pairs) with the REFERENCES leafRef + k, where
// leafRef = first pair - inner + kBvhLeafFlag (modulo 2^32): a slot's reference is one add from either base (bvh_device.hpp).
struct Bvh4Node {  // 64-byte slot, 48 bytes read (three 16-byte loads)
  float ox, oy, oz;       // quantisation origin = lower corner of the node's box
  uint32_t meta;          // byte 0-2: biased power-of-two exponent of the x / y / z scale; byte 3: inner | count << 4
  uint32_t childBase;     // first inner child
  uint32_t leafRef;       // flagged reference of leaf slot k, minus k:  first pair - inner + kBvhLeafFlag  (mod 2^32)
  uint32_t qlox, qhix;    // byte k: child k's quantised planes
  uint32_t qloy, qhiy, qloz, qhiz;
  uint32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 64, "node size");

inline int bvhNodeInner(Bvh4Node const& n) { return int((n.meta >> 24) & 0xFu); }
inline int bvhNodeCount(Bvh4Node const& n) { return int((n.meta >> 28) & 0xFu); }
inline float bvhNodeScale(Bvh4Node const& n, int axis) {
  uint32_t const bits = ((n.meta >> (8 * axis)) & 0xFFu) << 23;
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}
// decoded box of child k (host side: validation and tests; the device never forms the planes, see bvh_device.hpp)
inline void bvhChildBox(Bvh4Node const& n, int k, float lo[3], float hi[3]) {
  uint32_t const ql[3] = {n.qlox, n.qloy, n.qloz}, qh[3] = {n.qhix, n.qhiy, n.qhiz};
  float const o[3] = {n.ox, n.oy, n.oz};
  for (int a = 0; a < 3; ++a) {
    float const s = bvhNodeScale(n, a);
    lo[a] = o[a] + float((ql[a] >> (8 * k)) & 0xFFu) * s;
    hi[a] = o[a] + float((qh[a] >> (8 * k)) & 0xFFu) * s;
  }
}

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
  std::vector<uint32_t> pairTris;  // two ORIGINAL triangle indices per leaf pair (the second repeats the first in a one-triangle leaf)
  int depth = 0;
};

// Quantise the boxes of `nk` children (inner children first) into node `nd`.  Exact-arithmetic guarantee: the decoded
// box encloses the given one.
inline void encodeNode(Bvh4Node& nd, Box const* kid, int nk, int nInner) {
  std::memset(&nd, 0, sizeof(nd));
  Box all;
  all.reset();
  for (int k = 0; k < nk; ++k) all.grow(kid[k]);
  uint32_t ebytes[3] = {127, 127, 127};
  float org[3] = {0, 0, 0};
  double scale[3] = {1, 1, 1};
  for (int a = 0; a < 3 && nk > 0; ++a) {
    org[a] = all.lo[a];
    double const ext = double(all.hi[a]) - double(all.lo[a]);
    int e = -60;  // floor of the scale: the traversal's slope a = scale * (1 / d) must never flush to zero (bvh_device.hpp)
    if (ext > 0.0) {
      int fe;
      (void)std::frexp(ext / 255.0, &fe);  // ext / 255 = m * 2^fe, m in [0.5, 1)  ->  2^fe >= ext / 255
      e = fe;
    }
    e = std::min(std::max(e, -60), 127);
    while (e < 127 && std::ldexp(255.0, e) < ext) ++e;
    ebytes[a] = uint32_t(e + 127);
    scale[a] = std::ldexp(1.0, e);
  }
  nd.ox = org[0], nd.oy = org[1], nd.oz = org[2];
  nd.meta = ebytes[0] | (ebytes[1] << 8) | (ebytes[2] << 16) | (uint32_t(nInner) << 24) | (uint32_t(nk) << 28);
  uint32_t* const qlo[3] = {&nd.qlox, &nd.qloy, &nd.qloz};
  uint32_t* const qhi[3] = {&nd.qhix, &nd.qhiy, &nd.qhiz};
  for (int k = 0; k < 4; ++k)
    for (int a = 0; a < 3; ++a) {
      uint32_t l = 255, h = 0;  // empty slot: inverted box; NOT masked by count on the device -- see buildBvh's guard pairs
      if (k < nk) {
        double const fl = std::floor((double(kid[k].lo[a]) - double(org[a])) / scale[a]);
        double const fh = std::ceil((double(kid[k].hi[a]) - double(org[a])) / scale[a]);
        l = uint32_t(std::min(std::max(fl, 0.0), 255.0));
        h = uint32_t(std::min(std::max(fh, 0.0), 255.0));
      }
      *qlo[a] |= l << (8 * k), *qhi[a] |= h << (8 * k);
    }
}

// xs/ys/zs: the reference's SoA (4 floats per triangle: c0, c1, c2, pad)
inline Result build(float const* xs, float const* ys, float const* zs, uint32_t n) {
  static_assert(kBvhMaxLeafTris == 2, "a leaf is one triangle pair");
  Result out;
  Builder b;
  b.triBox.resize(n), b.centroid.resize(3 * size_t(n)), b.order.resize(n);
  // largest |coordinate| of the soup: the slab arithmetic of the traversal (bvh_device.hpp: t = q a + b with b = origin inv +
  // (-o inv), two separately rounded products) errs by a few ulp of the ORIGIN-SIDE magnitudes, ~1.2e-7 (|o| + |node origin|) in
  // space, whatever the triangle's own coordinates are -- a triangle near coordinate 0 seen from far away is the case the
  // per-triangle terms below do not cover (round-2 advisor).  Rays start on the scene's surfaces or at a camera; the padding
  // allows for origins up to 8x the scene's largest |coordinate| away from the axes' zero.
  float sceneMaxAbs = 0.f;
  for (size_t k = 0; k < size_t(n) * 4; ++k) {
    if ((k & 3) == 3) continue;  // the SoA's pad lane
    sceneMaxAbs = std::max(sceneMaxAbs, std::max(std::fabs(xs[k]), std::max(std::fabs(ys[k]), std::fabs(zs[k]))));
  }
  float const slabPad = 2.5e-7f * 9.f * sceneMaxAbs;
  for (uint32_t i = 0; i < n; ++i) {
    Box bx;
    bx.reset();
    for (int v = 0; v < 3; ++v) {
      float const p[3] = {xs[4 * size_t(i) + v], ys[4 * size_t(i) + v], zs[4 * size_t(i) + v]};
      bx.grow(p);
    }
    // padding: far above the rounding of the triangle test (~1e-7 relative to the triangle) and of the slab tests (slabPad),
    // far below anything that costs traversal work
    for (int a = 0; a < 3; ++a) {
      float const m = std::max(std::fabs(bx.lo[a]), std::fabs(bx.hi[a]));
      float const pad = 1e-5f * (bx.hi[a] - bx.lo[a]) + 4e-6f * m + 1e-7f + slabPad;
      bx.lo[a] -= pad, bx.hi[a] += pad;
    }
    b.triBox[i] = bx;
    for (int a = 0; a < 3; ++a) b.centroid[3 * size_t(i) + a] = 0.5f * (bx.lo[a] + bx.hi[a]);
    b.order[i] = i;
  }
  if (n == 0) {  // a root without children
    Bvh4Node root;
    encodeNode(root, nullptr, 0, 0);
    out.nodes.push_back(root);
    return out;
  }
  b.nodes.reserve(size_t(n));
  int const root2 = b.build(0, n, kBvhMaxDepth);

  // collapse binary splits into 4-wide nodes: repeatedly open the inner child of largest area.  A node's inner
  // children get CONSECUTIVE node indices (implicit child references), processed breadth-first so that index order
  // is also level order of each subtree's top.
  struct Wide {
    int kids[4];  // binary node ids, inner children first
    int nk = 0, nInner = 0;
    int depth = 1;
  };
  std::vector<Wide> wide;
  wide.emplace_back();
  std::vector<int> source;  // binary node each wide node came from
  source.push_back(root2);
  for (size_t w = 0; w < wide.size(); ++w) {
    int const node2 = source[w];
    int kids[4];
    int nk = 0;
    if (b.nodes[size_t(node2)].leaf()) {
      kids[nk++] = node2;
    } else {
      kids[nk++] = b.nodes[size_t(node2)].left;
      kids[nk++] = b.nodes[size_t(node2)].right;
      while (nk < 4) {
        int pick = -1;
        float bestA = -1.f;
        for (int k = 0; k < nk; ++k)
          if (!b.nodes[size_t(kids[k])].leaf() && b.nodes[size_t(kids[k])].box.area() > bestA) bestA = b.nodes[size_t(kids[k])].box.area(), pick = k;
        if (pick < 0) break;
        int const open = kids[pick];
        kids[pick] = b.nodes[size_t(open)].left;
        kids[nk++] = b.nodes[size_t(open)].right;
      }
    }
    Wide W = wide[w];
    W.nk = nk;
    W.nInner = 0;
    for (int k = 0; k < nk; ++k)
      if (!b.nodes[size_t(kids[k])].leaf()) W.kids[W.nInner++] = kids[k];
    int at = W.nInner;
    for (int k = 0; k < nk; ++k)
      if (b.nodes[size_t(kids[k])].leaf()) W.kids[at++] = kids[k];
    wide[w] = W;
    out.depth = std::max(out.depth, W.depth);
    for (int k = 0; k < W.nInner; ++k) {  // consecutive indices: size() .. size() + nInner - 1
      Wide c;
      c.depth = W.depth + 1;
      wide.push_back(c);
      source.push_back(W.kids[k]);
    }
  }
  // encode in index order; leaves of a node become consecutive pairs
  out.nodes.resize(wide.size());
  uint32_t nextChild = 1;
  for (size_t w = 0; w < wide.size(); ++w) {
    Wide const& W = wide[w];
    Box kb[4];
    for (int k = 0; k < W.nk; ++k) kb[k] = b.nodes[size_t(W.kids[k])].box;
    Bvh4Node nd;
    encodeNode(nd, kb, W.nk, W.nInner);
    nd.childBase = nextChild;
    nextChild += uint32_t(W.nInner);
    nd.leafRef = uint32_t(out.pairTris.size() / 2) - uint32_t(W.nInner) + kBvhLeafFlag;
    for (int k = W.nInner; k < W.nk; ++k) {
      Node2 const& c = b.nodes[size_t(W.kids[k])];
      uint32_t const t0 = b.order[c.first], t1 = b.order[c.first + (c.count > 1 ? 1 : 0)];
      out.pairTris.push_back(t0), out.pairTris.push_back(t1);
    }
    out.nodes[w] = nd;
  }
  return out;
}

}  // namespace bvh_build


}  // namespace dmt
