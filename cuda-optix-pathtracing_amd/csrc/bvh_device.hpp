// bvh_device.hpp -- per-lane traversal of the 4-wide BVH (bvh.hpp) on gfx950.
//
// One lane = one ray.  The traversal stack lives in LDS as [entry][thread] (16 entries, conflict
// free: consecutive lanes hit consecutive banks); deeper entries -- rare -- go to a per-lane column of
// a global overflow array.  A node is seven 16-byte per-lane loads (boxes as SoA + child refs), the
// four slab tests are plain VALU min/max, hit children are ordered by a 5-exchange network and the
// nearest is entered directly (no push/pop).  Leaves hold <= 2 triangles (one pair), in traversal
// order, tested with the SAME Moeller-Trumbore core as the brute-force loop, so the closest hit
// (t,u,v and triangle, lowest original index on ties) is bit-identical to brute force.
#pragma once

#include "bvh.hpp"
#include "pt_device.hpp"

namespace dmt {

struct BvhView {
  Bvh4Node const* __restrict__ nodes;
  TriPair const* __restrict__ pairs;  // leaf storage, see bvh.hpp
  uint32_t* __restrict__ overflow;    // [kBvhOverflowStack][overflowStride]
  uint32_t overflowStride;            // total threads of the launch
};

__shared__ uint32_t s_bvh_stack[kBvhLdsStack * kLdsThreads];

struct BvhStack {
  int sp;
  uint32_t* ovf;  // this lane's overflow column
  uint32_t stride;
  uint32_t* ovfCount = nullptr;  // stats build only: counts the pushes that went to the global overflow area
  DMT_DEV void push(uint32_t ref) {
    if (sp < kBvhLdsStack) {
      s_bvh_stack[sp * kLdsThreads + int(threadIdx.x)] = ref;
    } else {
      ovf[size_t(sp - kBvhLdsStack) * stride] = ref;
      if (ovfCount) ++*ovfCount;
    }
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
// one ray against one triangle pair: validity and (t,u,v) per half, bit-identical to the brute-force test
struct PairHit {
  bool valid0, valid1;
  v2f t, u, v;
  uint32_t orig0, orig1;
};
DMT_DEV PairHit pair_test(TriPair const& P, f3 o, f3 d) {
  // five 16-byte loads; consecutive floats of the record are (first, second) pairs
  float4 const a = *reinterpret_cast<float4 const*>(&P.p0x[0]);  // p0x p0x p0y p0y
  float4 const b = *reinterpret_cast<float4 const*>(&P.p0z[0]);  // p0z p0z e0x e0x
  float4 const c = *reinterpret_cast<float4 const*>(&P.e0y[0]);  // e0y e0y e0z e0z
  float4 const e = *reinterpret_cast<float4 const*>(&P.e1x[0]);  // e1x e1x e1y e1y
  float4 const f = *reinterpret_cast<float4 const*>(&P.e1z[0]);  // e1z e1z orig orig
  PairHit h;
  v2f det;
  mt_core9_tri2(v2f{a.x, a.y}, v2f{a.z, a.w}, v2f{b.x, b.y}, v2f{b.z, b.w}, v2f{c.x, c.y}, v2f{c.z, c.w}, v2f{e.x, e.y},
                v2f{e.z, e.w}, v2f{f.x, f.y}, o.x, o.y, o.z, d.x, d.y, d.z, det, h.t, h.u, h.v);
  h.valid0 = mt_valid(det.x, h.t.x, h.u.x, h.v.x);
  h.valid1 = mt_valid(det.y, h.t.y, h.u.y, h.v.y);
  h.orig0 = __float_as_uint(f.z), h.orig1 = __float_as_uint(f.w);
  return h;
}

struct TraversalCounters {  // per-lane work counters (stats build of the kernel only)
  uint32_t nodes = 0, tris = 0;
  uint32_t deadNodes = 0;  // visited nodes none of whose children was entered or pushed
  uint32_t overflowPushes = 0;  // stack pushes that went to the global overflow area (entries >= kBvhLdsStack)
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
      for (uint32_t j = 0; j < cnt; ++j) {
        PairHit const h = pair_test(bv.pairs[first + j], o, d);
        if constexpr (STATS) tc->tris += h.orig0 != h.orig1 ? 2u : 1u;  // an odd leaf repeats its last triangle
        // brute force keeps the lowest index among equal t (strict < in index order)
        if (h.valid0 && (h.t.x < bt || (h.t.x == bt && h.orig0 < bestOrig)))
          bt = h.t.x, bu = h.u.x, bvv = h.v.x, bestOrig = h.orig0, bestTri = int(h.orig0);
        if (h.valid1 && (h.t.y < bt || (h.t.y == bt && h.orig1 < bestOrig)))
          bt = h.t.y, bu = h.u.y, bvv = h.v.y, bestOrig = h.orig1, bestTri = int(h.orig1);
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
      for (uint32_t j = 0; j < cnt; ++j) {
        PairHit const h = pair_test(bv.pairs[first + j], o, d);
        if constexpr (STATS) tc->tris += h.orig0 != h.orig1 ? 2u : 1u;  // an odd leaf repeats its last triangle
        if ((h.valid0 && h.t.x < tmax) || (h.valid1 && h.t.y < tmax)) occluded = true;
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
// Slab setup of the resumable traversal: t = plane * inv + oi (one FMA per plane) with the near / far plane of
// each axis picked by the sign of the direction, so that a child costs 6 FMA + max3 + min3 instead of
// 12 sub/mul + 6 min/max.  Both shortcuts only have to stay CONSERVATIVE: the FMA form errs by
// ~eps * (|plane| + |o|) * |inv|, i.e. ~1e-7 of the coordinates in space, 30x below the builder's box padding
// (4e-6 * |coordinate| + 1e-5 * extent); a zero direction component gives inf - inf = NaN on that axis, which
// max/min drop, i.e. the axis is ignored.  Hits are decided by the triangle test alone.
struct SlabRay2 {
  f3 inv, oi;
  uint32_t nearOff;  // byte offsets of the near planes inside a node, packed: x | y << 8 | z << 16
};
DMT_DEV SlabRay2 slab_ray2(f3 o, f3 d) {
  SlabRay2 r;
  r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.oi = mk3(-o.x * r.inv.x, -o.y * r.inv.y, -o.z * r.inv.z);
  r.nearOff = (d.x < 0.f ? 48u : 0u) | ((d.y < 0.f ? 64u : 16u) << 8) | ((d.z < 0.f ? 80u : 32u) << 16);
  return r;
}
enum : int { TR_IDLE = 0, TR_CLOSEST = 1, TR_SHADOW = 2, TR_DONE = 3 };
struct Traversal {
  int phase;
  bool doC, doS;     // what this round of the lane consists of
  uint32_t cur;
  BvhStack stack;
  f3 o, d;           // ray of the current phase
  SlabRay2 sr;
  float tmax;        // closest: best t so far; shadow: light distance
  int bestTri;       // ORIGINAL index
  uint32_t bestOrig;
  float bt, bu, bv;
  bool occluded;
};
DMT_DEV void trav_set_ray(Traversal& tv, f3 o, f3 d) {
  tv.o = o, tv.d = d;
  tv.sr = slab_ray2(o, d);
  tv.cur = 0u;
  tv.stack.sp = 0;
}
// entry distance of one child (+inf on a miss) from its near / far plane triples
DMT_DEV float slab2(float nx, float ny, float nz, float fx, float fy, float fz, SlabRay2 const& r, float tlimit,
                    uint32_t ref) {
  float const tn = fmaxf(fmaxf(fmaxf(fma_(nx, r.inv.x, r.oi.x), fma_(ny, r.inv.y, r.oi.y)), fma_(nz, r.inv.z, r.oi.z)), 0.f);
  float const tf =
      fminf(fminf(fminf(fma_(fx, r.inv.x, r.oi.x), fma_(fy, r.inv.y, r.oi.y)), fma_(fz, r.inv.z, r.oi.z)), tlimit) * 1.0000004f;
  // an empty slot's (+inf,-inf) box is not reliably a miss (NaNs are dropped): mask by ref
  return (tn <= tf && ref != kBvhEmpty) ? tn : kInf;
}
// node step: cur is an inner node
template <bool STATS = false>
DMT_DEV void trav_node(BvhView const& bv, Traversal& tv, TraversalCounters* tc = nullptr) {
  float const tlimit = tv.phase == TR_CLOSEST ? tv.bt : tv.tmax;
  char const* const nb = reinterpret_cast<char const*>(bv.nodes + tv.cur);
  if constexpr (STATS) ++tc->nodes;
  uint32_t const ox = tv.sr.nearOff & 0xFFu, oy = (tv.sr.nearOff >> 8) & 0xFFu, oz = (tv.sr.nearOff >> 16) & 0xFFu;
  float4 const nx = *reinterpret_cast<float4 const*>(nb + ox), fx = *reinterpret_cast<float4 const*>(nb + (48u - ox));
  float4 const ny = *reinterpret_cast<float4 const*>(nb + oy), fy = *reinterpret_cast<float4 const*>(nb + (80u - oy));
  float4 const nz = *reinterpret_cast<float4 const*>(nb + oz), fz = *reinterpret_cast<float4 const*>(nb + (112u - oz));
  uint4 const ch = *reinterpret_cast<uint4 const*>(nb + 96);
  uint32_t r0 = ch.x, r1 = ch.y, r2 = ch.z, r3 = ch.w;
  float k0 = slab2(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, tv.sr, tlimit, r0);
  float k1 = slab2(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, tv.sr, tlimit, r1);
  float k2 = slab2(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, tv.sr, tlimit, r2);
  float k3 = slab2(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, tv.sr, tlimit, r3);
  cswap(k0, r0, k1, r1);
  cswap(k2, r2, k3, r3);
  cswap(k0, r0, k2, r2);
  cswap(k1, r1, k3, r3);
  cswap(k1, r1, k2, r2);
  bool const p3 = k3 < kInf, p2 = k2 < kInf, p1 = k1 < kInf, p0 = k0 < kInf;
  if constexpr (STATS) tc->deadNodes += p0 ? 0u : 1u;
  if (!__any(tv.stack.sp > kBvhLdsStack - 3)) {
    // whole wave within the LDS part of the stack: branch-free.  All three candidates are written, far to
    // near, each at the slot the previous one left free if it was a miss; slots above the new top are dead.
    uint32_t* const base = s_bvh_stack + int(threadIdx.x);
    int const a3 = tv.stack.sp, a2 = a3 + (p3 ? 1 : 0), a1 = a2 + (p2 ? 1 : 0);
    base[a3 * kLdsThreads] = r3;
    base[a2 * kLdsThreads] = r2;
    base[a1 * kLdsThreads] = r1;
    int const top = a1 + (p1 ? 1 : 0);  // > 0 whenever something was pushed; p0 false implies nothing was
    uint32_t const popped = base[(top > 0 ? top - 1 : 0) * kLdsThreads];
    tv.cur = p0 ? r0 : (top > 0 ? popped : kBvhEmpty);
    tv.stack.sp = p0 ? top : (top > 0 ? top - 1 : 0);
  } else {
    if (p3) tv.stack.push(r3);
    if (p2) tv.stack.push(r2);
    if (p1) tv.stack.push(r1);
    tv.cur = p0 ? r0 : tv.stack.pop();
  }
}
// leaf step: cur is a leaf reference (1-2 triangle pairs)
template <bool STATS = false>
DMT_DEV void trav_leaf(BvhView const& bv, Traversal& tv, TraversalCounters* tc = nullptr) {
  bool const closest = tv.phase == TR_CLOSEST;
  uint32_t const first = tv.cur & 0x0FFFFFFFu;
  uint32_t const cnt = kBvhMaxLeafTris <= 2 ? 1u : ((tv.cur >> 28) & 7u) + 1u;  // builder's leaf size: one pair
  for (uint32_t j = 0; j < cnt; ++j) {
    PairHit const h = pair_test(bv.pairs[first + j], tv.o, tv.d);
    if constexpr (STATS) tc->tris += h.orig0 != h.orig1 ? 2u : 1u;  // an odd leaf repeats its last triangle
    if (closest) {
      if (h.valid0 && (h.t.x < tv.bt || (h.t.x == tv.bt && h.orig0 < tv.bestOrig)))
        tv.bt = h.t.x, tv.bu = h.u.x, tv.bv = h.v.x, tv.bestOrig = h.orig0, tv.bestTri = int(h.orig0);
      if (h.valid1 && (h.t.y < tv.bt || (h.t.y == tv.bt && h.orig1 < tv.bestOrig)))
        tv.bt = h.t.y, tv.bu = h.u.y, tv.bv = h.v.y, tv.bestOrig = h.orig1, tv.bestTri = int(h.orig1);
    } else if ((h.valid0 && h.t.x < tv.tmax) || (h.valid1 && h.t.y < tv.tmax)) {
      tv.occluded = true;
    }
  }
  tv.cur = (!closest && tv.occluded) ? kBvhEmpty : tv.stack.pop();
}

}  // namespace dmt
