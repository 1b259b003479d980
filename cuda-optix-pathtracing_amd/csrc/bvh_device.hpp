// bvh_device.hpp -- per-lane traversal of the 4-wide BVH (bvh.hpp) on gfx950.
//
// One lane = one ray.  The traversal stack lives in LDS as [entry][thread] (16 entries, conflict
// free: consecutive lanes hit consecutive banks); deeper entries -- rare -- go to a per-lane column of
// a global overflow array.  A node is seven 16-byte per-lane loads (boxes as SoA + child refs), the
// four slab tests are plain VALU min/max, hit children are ordered by a 5-exchange network and the
// nearest is entered directly (no push/pop).  Leaves hold <= 4 triangles, contiguous in traversal
// order, tested with the SAME Moeller-Trumbore core as the brute-force loop, so the closest hit
// (t,u,v and triangle, lowest original index on ties) is bit-identical to brute force.
#pragma once

#include "bvh.hpp"
#include "pt_device.hpp"

namespace dmt {

struct BvhView {
  Bvh4Node const* __restrict__ nodes;
  TriIsect const* __restrict__ tris;  // slot order; pad0 = original triangle index
  uint32_t* __restrict__ overflow;    // [kBvhOverflowStack][overflowStride]
  uint32_t overflowStride;            // total threads of the launch
};

__shared__ uint32_t s_bvh_stack[kBvhLdsStack * kLdsThreads];

struct BvhStack {
  int sp;
  uint32_t* ovf;  // this lane's overflow column
  uint32_t stride;
  DMT_DEV void push(uint32_t ref) {
    if (sp < kBvhLdsStack)
      s_bvh_stack[sp * kLdsThreads + int(threadIdx.x)] = ref;
    else
      ovf[size_t(sp - kBvhLdsStack) * stride] = ref;
    ++sp;
  }
  DMT_DEV uint32_t pop() {
    if (sp == 0) return kBvhEmpty;
    --sp;
    // LDS read unconditional (clamped slot), global read only when needed: selecting between the two
    // POINTERS would make hipcc emit one flat_load through a generic pointer
    int const slot = sp < kBvhLdsStack ? sp : kBvhLdsStack - 1;
    uint32_t v = s_bvh_stack[slot * kLdsThreads + int(threadIdx.x)];
    if (sp >= kBvhLdsStack) v = ovf[size_t(sp - kBvhLdsStack) * stride];
    return v;
  }
};

struct SlabRay {
  f3 o, inv;
};
DMT_DEV SlabRay slab_ray(f3 o, f3 d) {
  SlabRay r;
  r.o = o;
  r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // +-inf for zero components: handled by min/max
  return r;
}
// entry distance of the box (+inf on a miss); conservative: boxes are padded by the builder and the
// exit distance is widened by 2 ulp-ish (pbrt's 1 + 2*gamma(3))
DMT_DEV float slab(SlabRay const& r, float lx, float ly, float lz, float hx, float hy, float hz, float tmax) {
  float const ax = (lx - r.o.x) * r.inv.x, bx = (hx - r.o.x) * r.inv.x;
  float const ay = (ly - r.o.y) * r.inv.y, by = (hy - r.o.y) * r.inv.y;
  float const az = (lz - r.o.z) * r.inv.z, bz = (hz - r.o.z) * r.inv.z;
  float const tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.f));
  float const tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax)) * 1.0000004f;
  return tn <= tf ? tn : kInf;
}
DMT_DEV void cswap(float& ka, uint32_t& ra, float& kb, uint32_t& rb) {
  bool const sw = kb < ka;
  float const k = sw ? kb : ka;
  uint32_t const r = sw ? rb : ra;
  kb = sw ? ka : kb, rb = sw ? ra : rb;
  ka = k, ra = r;
}
DMT_DEV TriS tri_from(TriIsect const& T) {
  TriS t;
  t.p0x = T.p0x, t.p0y = T.p0y, t.p0z = T.p0z, t.e0x = T.e0x, t.e0y = T.e0y, t.e0z = T.e0z;
  t.e1x = T.e1x, t.e1y = T.e1y, t.e1z = T.e1z;
  return t;
}

struct TraversalCounters {  // per-lane work counters (stats build of the kernel only)
  uint32_t nodes = 0, tris = 0;
};

// closest hit: bestTri = ORIGINAL triangle index or -1
template <bool STATS = false>
DMT_DEV void bvh_closest(BvhView const& bv, bool active, f3 o, f3 d, uint32_t gtid, int& bestTri, float& bt,
                         float& bu, float& bvv, TraversalCounters* tc = nullptr) {
  bt = kInf, bestTri = -1, bu = 0.f, bvv = 0.f;
  uint32_t bestOrig = 0xFFFFFFFFu;
  SlabRay const sr = slab_ray(o, d);
  BvhStack st{0, bv.overflow + gtid, bv.overflowStride};
  uint32_t cur = active ? 0u : kBvhEmpty;  // node 0 = root
  while (cur != kBvhEmpty) {
    if (!(cur & kBvhLeafFlag)) {
      Bvh4Node const& n = bv.nodes[cur];
      if constexpr (STATS) ++tc->nodes;
      float k0 = slab(sr, n.minx[0], n.miny[0], n.minz[0], n.maxx[0], n.maxy[0], n.maxz[0], bt);
      float k1 = slab(sr, n.minx[1], n.miny[1], n.minz[1], n.maxx[1], n.maxy[1], n.maxz[1], bt);
      float k2 = slab(sr, n.minx[2], n.miny[2], n.minz[2], n.maxx[2], n.maxy[2], n.maxz[2], bt);
      float k3 = slab(sr, n.minx[3], n.miny[3], n.minz[3], n.maxx[3], n.maxy[3], n.maxz[3], bt);
      uint32_t r0 = n.child[0], r1 = n.child[1], r2 = n.child[2], r3 = n.child[3];
      // an empty slot's (+inf,-inf) box is NOT a miss for the slab test (min/max swap it): mask by ref
      k0 = r0 == kBvhEmpty ? kInf : k0, k1 = r1 == kBvhEmpty ? kInf : k1;
      k2 = r2 == kBvhEmpty ? kInf : k2, k3 = r3 == kBvhEmpty ? kInf : k3;
      cswap(k0, r0, k1, r1);
      cswap(k2, r2, k3, r3);
      cswap(k0, r0, k2, r2);
      cswap(k1, r1, k3, r3);
      cswap(k1, r1, k2, r2);
      if (k3 < kInf) st.push(r3);  // far to near, nearest entered directly
      if (k2 < kInf) st.push(r2);
      if (k1 < kInf) st.push(r1);
      cur = k0 < kInf ? r0 : st.pop();
    } else {
      uint32_t const first = cur & 0x0FFFFFFFu;
      uint32_t const cnt = ((cur >> 28) & 7u) + 1u;
      if constexpr (STATS) tc->tris += cnt;
      for (uint32_t j = 0; j < cnt; ++j) {
        TriIsect const T = bv.tris[first + j];
        float det, t, u, v;
        mt_core<float>(tri_from(T), o.x, o.y, o.z, d.x, d.y, d.z, det, t, u, v);
        // brute force keeps the lowest index among equal t (strict < in index order)
        if (mt_valid(det, t, u, v) && (t < bt || (t == bt && T.pad0 < bestOrig))) {
          bt = t, bu = u, bvv = v, bestOrig = T.pad0, bestTri = int(T.pad0);
        }
      }
      cur = st.pop();
    }
  }
}

// any hit with t < tmax
template <bool STATS = false>
DMT_DEV bool bvh_any(BvhView const& bv, bool active, f3 o, f3 d, float tmax, uint32_t gtid,
                     TraversalCounters* tc = nullptr) {
  SlabRay const sr = slab_ray(o, d);
  BvhStack st{0, bv.overflow + gtid, bv.overflowStride};
  uint32_t cur = active ? 0u : kBvhEmpty;
  bool occluded = false;
  while (cur != kBvhEmpty) {
    if (!(cur & kBvhLeafFlag)) {
      Bvh4Node const& n = bv.nodes[cur];
      if constexpr (STATS) ++tc->nodes;
      float const k0 = slab(sr, n.minx[0], n.miny[0], n.minz[0], n.maxx[0], n.maxy[0], n.maxz[0], tmax);
      float const k1 = slab(sr, n.minx[1], n.miny[1], n.minz[1], n.maxx[1], n.maxy[1], n.maxz[1], tmax);
      float const k2 = slab(sr, n.minx[2], n.miny[2], n.minz[2], n.maxx[2], n.maxy[2], n.maxz[2], tmax);
      float const k3 = slab(sr, n.minx[3], n.miny[3], n.minz[3], n.maxx[3], n.maxy[3], n.maxz[3], tmax);
      uint32_t const r0 = n.child[0], r1 = n.child[1], r2 = n.child[2], r3 = n.child[3];
      uint32_t next = kBvhEmpty;  // empty slots are masked by ref (their box is not a slab miss)
      if (k3 < kInf && r3 != kBvhEmpty) next = r3;
      if (k2 < kInf && r2 != kBvhEmpty) {
        if (next != kBvhEmpty) st.push(next);
        next = r2;
      }
      if (k1 < kInf && r1 != kBvhEmpty) {
        if (next != kBvhEmpty) st.push(next);
        next = r1;
      }
      if (k0 < kInf && r0 != kBvhEmpty) {
        if (next != kBvhEmpty) st.push(next);
        next = r0;
      }
      cur = next != kBvhEmpty ? next : st.pop();
    } else {
      uint32_t const first = cur & 0x0FFFFFFFu;
      uint32_t const cnt = ((cur >> 28) & 7u) + 1u;
      if constexpr (STATS) tc->tris += cnt;
      for (uint32_t j = 0; j < cnt; ++j) {
        TriIsect const T = bv.tris[first + j];
        float det, t, u, v;
        mt_core<float>(tri_from(T), o.x, o.y, o.z, d.x, d.y, d.z, det, t, u, v);
        if (mt_valid(det, t, u, v) && t < tmax) occluded = true;
      }
      cur = occluded ? kBvhEmpty : st.pop();
    }
  }
  return occluded;
}

// ---------------------------------------------------------------------------------------------
// Resumable traversal: one step (one node, or one leaf) per call, so that the lanes of a wave can be at
// different points of different traversals and nobody waits for the wave's longest ray.  A lane first
// runs its closest-hit traversal, then (same stack) the any-hit traversal of its pending shadow ray.
// Both phases share one code path (children are always distance-sorted) to keep divergence low.
// ---------------------------------------------------------------------------------------------
enum : int { TR_IDLE = 0, TR_CLOSEST = 1, TR_SHADOW = 2, TR_DONE = 3 };
struct Traversal {
  int phase;
  bool doC, doS;     // what this round of the lane consists of
  uint32_t cur;
  BvhStack stack;
  f3 o, d;           // ray of the current phase
  SlabRay sr;
  float tmax;        // closest: best t so far; shadow: light distance
  int bestTri;       // ORIGINAL index
  uint32_t bestOrig;
  float bt, bu, bv;
  bool occluded;
};
DMT_DEV void trav_set_ray(Traversal& tv, f3 o, f3 d) {
  tv.o = o, tv.d = d;
  tv.sr = slab_ray(o, d);
  tv.cur = 0u;
  tv.stack.sp = 0;
}
// node step: cur is an inner node
template <bool STATS = false>
DMT_DEV void trav_node(BvhView const& bv, Traversal& tv, TraversalCounters* tc = nullptr) {
  float const tlimit = tv.phase == TR_CLOSEST ? tv.bt : tv.tmax;
  Bvh4Node const& n = bv.nodes[tv.cur];
  if constexpr (STATS) ++tc->nodes;
  float k0 = slab(tv.sr, n.minx[0], n.miny[0], n.minz[0], n.maxx[0], n.maxy[0], n.maxz[0], tlimit);
  float k1 = slab(tv.sr, n.minx[1], n.miny[1], n.minz[1], n.maxx[1], n.maxy[1], n.maxz[1], tlimit);
  float k2 = slab(tv.sr, n.minx[2], n.miny[2], n.minz[2], n.maxx[2], n.maxy[2], n.maxz[2], tlimit);
  float k3 = slab(tv.sr, n.minx[3], n.miny[3], n.minz[3], n.maxx[3], n.maxy[3], n.maxz[3], tlimit);
  uint32_t r0 = n.child[0], r1 = n.child[1], r2 = n.child[2], r3 = n.child[3];
  k0 = r0 == kBvhEmpty ? kInf : k0, k1 = r1 == kBvhEmpty ? kInf : k1;
  k2 = r2 == kBvhEmpty ? kInf : k2, k3 = r3 == kBvhEmpty ? kInf : k3;
  cswap(k0, r0, k1, r1);
  cswap(k2, r2, k3, r3);
  cswap(k0, r0, k2, r2);
  cswap(k1, r1, k3, r3);
  cswap(k1, r1, k2, r2);
  if (k3 < kInf) tv.stack.push(r3);
  if (k2 < kInf) tv.stack.push(r2);
  if (k1 < kInf) tv.stack.push(r1);
  tv.cur = k0 < kInf ? r0 : tv.stack.pop();
}
// leaf step: cur is a leaf reference
template <bool STATS = false>
DMT_DEV void trav_leaf(BvhView const& bv, Traversal& tv, TraversalCounters* tc = nullptr) {
  bool const closest = tv.phase == TR_CLOSEST;
  uint32_t const first = tv.cur & 0x0FFFFFFFu;
  uint32_t const cnt = ((tv.cur >> 28) & 7u) + 1u;
  if constexpr (STATS) tc->tris += cnt;
  for (uint32_t j = 0; j < cnt; ++j) {
    TriIsect const T = bv.tris[first + j];
    float det, t, u, v;
    mt_core<float>(tri_from(T), tv.o.x, tv.o.y, tv.o.z, tv.d.x, tv.d.y, tv.d.z, det, t, u, v);
    bool const valid = mt_valid(det, t, u, v);
    if (closest) {
      if (valid && (t < tv.bt || (t == tv.bt && T.pad0 < tv.bestOrig)))
        tv.bt = t, tv.bu = u, tv.bv = v, tv.bestOrig = T.pad0, tv.bestTri = int(T.pad0);
    } else if (valid && t < tv.tmax) {
      tv.occluded = true;
    }
  }
  tv.cur = (!closest && tv.occluded) ? kBvhEmpty : tv.stack.pop();
}

}  // namespace dmt
