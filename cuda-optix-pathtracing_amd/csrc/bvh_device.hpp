// bvh_device.hpp -- per-lane traversal of the 4-wide quantised BVH (bvh.hpp) on gfx950.
//
// One lane = one ray.  What bounds this on gfx950 is the CU's vector-memory front end (tools/ubench/gather.hip,
// bvh.hpp header): a traversal step costs ~0.7 clocks of the L1 path per 16-byte per-lane load and ~2.5 per 64-byte
// sector touched, at any occupancy.  So a node is ONE sector read with THREE loads (quantised child boxes, implicit
// child references) and a leaf is one 80-byte triangle pair (five loads).  The traversal stack lives in LDS as
// [entry][thread] (16 entries, conflict free: consecutive lanes hit consecutive banks); deeper entries -- rare -- go to
// a per-lane column of a global overflow array.  The four slab tests work on the quantised planes directly
// (t = q * (scale * inv) + (origin - o) * inv: one v_cvt_f32_ubyteN + one FMA per plane), hit children are ordered by
// a 5-exchange network and the nearest is entered directly (no push/pop).  Leaves are tested with the SAME
// Moeller-Trumbore core as the brute-force loop, so the closest hit (t,u,v and triangle, lowest original index on ties)
// is bit-identical to brute force.
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

// The stack pointer is all a lane keeps.  The overflow column (entries beyond the LDS part, rare) is addressed from the
// launch's view when it is needed: a lane's column is overflow + its global thread index, stride = threads of the launch.
// (Round 2 kept the column pointer and the stride in three VGPRs of every lane for the whole kernel.)
struct BvhStack {
  int sp;
  uint32_t* ovfCount = nullptr;  // stats build only: counts the pushes that went to the global overflow area
  DMT_DEV static uint32_t* column(BvhView const& bv) { return bv.overflow + (blockIdx.x * blockDim.x + threadIdx.x); }
  DMT_DEV void push(BvhView const& bv, uint32_t ref) {
    if (sp < kBvhLdsStack) {
      s_bvh_stack[sp * kLdsThreads + int(threadIdx.x)] = ref;
    } else {
      column(bv)[size_t(sp - kBvhLdsStack) * bv.overflowStride] = ref;
      if (ovfCount) ++*ovfCount;
    }
    ++sp;
  }
  DMT_DEV uint32_t pop(BvhView const& bv) {
    // Whole wave within the LDS part (the usual case): a plain ds_read.  The general form below reads LDS or the overflow
    // column per lane, and hipcc turns its two loads into ONE flat_load through a selected generic pointer, however it is
    // written -- an LDS access by way of the vector-memory path, waited for with vmcnt(0) AND lgkmcnt(0).  The wave-uniform
    // branch keeps that out of the common path.
    if (!__any(sp > kBvhLdsStack)) {
      if (sp == 0) return kBvhEmpty;
      --sp;
      return s_bvh_stack[sp * kLdsThreads + int(threadIdx.x)];
    }
    if (sp == 0) return kBvhEmpty;
    --sp;
    int const slot = sp < kBvhLdsStack ? sp : kBvhLdsStack - 1;
    uint32_t v = s_bvh_stack[slot * kLdsThreads + int(threadIdx.x)];
    if (sp >= kBvhLdsStack) v = column(bv)[size_t(sp - kBvhLdsStack) * bv.overflowStride];
    return v;
  }
};

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
DMT_DEV PairHit pair_test(TriPair const* P, f3 o, f3 d) {
  // five 16-byte loads; consecutive floats of the record are (first, second) pairs
  float4 const* const q = reinterpret_cast<float4 const*>(P);
  float4 const a = q[0];  // p0x p0x p0y p0y
  float4 const b = q[1];  // p0z p0z e0x e0x
  float4 const c = q[2];  // e0y e0y e0z e0z
  float4 const e = q[3];  // e1x e1x e1y e1y
  float4 const f = q[4];  // e1z e1z orig orig
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

// ---------------------------------------------------------------------------------------------
// Resumable traversal: one step (one node, or one leaf) per call, so that the lanes of a wave can be at
// different points of different traversals and nobody waits for the wave's longest ray.  A lane first
// runs its closest-hit traversal, then (same stack) the any-hit traversal of its pending shadow ray.
// Both phases share one code path (children are always distance-sorted) to keep divergence low.
// ---------------------------------------------------------------------------------------------
// Slab setup: a child's plane at quantised coordinate q is at  origin + q * scale, so its ray parameter is
//   t = (origin + q * scale - o) * inv = q * (scale * inv) + (origin * inv + oi),   oi = -o * inv,
// i.e. one FMA per plane after three multiplies and three FMAs per node.  The near / far plane of each axis is picked
// by the sign of the direction.  Everything here only has to stay CONSERVATIVE.  b adds two separately rounded products
// (origin * inv by the FMA, -o * inv before it), so t errs by a few ulp of |o * inv| + |origin * inv| -- in space ~1.2e-7 of
// the ray origin's and the node origin's COORDINATE MAGNITUDES, not of their difference and not of the triangle's own
// coordinates.  The builder therefore pads every triangle box by 2.25e-6 x the soup's largest |coordinate| on top of its
// per-triangle terms (bvh.hpp: slabPad; covers ray origins up to 8x that far out), which costs no instruction here;
// subtracting first ((origin - o) * inv) would cost three.  A zero direction component gives inf * 0 or inf - inf = NaN on
// that axis, which max/min drop, i.e. the axis is ignored.  Hits are decided by the triangle test alone.
struct SlabRay2 {
  f3 inv, oi;  // (which of a node's plane words is the near one is read off inv's sign in the step: as loop-carried flags
               //  the three cost six SGPRs and eighteen scalar instructions per loop iteration)
};
DMT_DEV SlabRay2 slab_ray2(f3 o, f3 d) {
  SlabRay2 r;
  r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  r.oi = mk3(-o.x * r.inv.x, -o.y * r.inv.y, -o.z * r.inv.z);
  return r;
}
enum : int { TR_IDLE = 0, TR_CLOSEST = 1, TR_SHADOW = 2, TR_DONE = 3 };
// The ray itself (origin, direction) is NOT part of the state: the caller holds it anyway and hands it to the leaf step
// (six registers per lane less in the megakernel, where every register of the traversal is live through shading).
struct Traversal {
  int phase;
  bool doC, doS;     // what this round of the lane consists of
  uint32_t cur;
  BvhStack stack;
  SlabRay2 sr;       // slab form of the ray of the current phase
  float tlim;        // closest: t of the best hit so far (kInf: none); shadow: the light's distance, NEGATIVE once the ray is occluded
                     // (no separate flag: a bool changed inside the traversal loop is a lane mask with scalar bookkeeping at every join)
  int bestTri;       // closest: ORIGINAL index of the best hit, -1 = none
  float bu, bv;
  DMT_DEV bool occluded() const { return tlim < 0.f; }  // of the shadow ray just traversed
};
DMT_DEV void trav_set_ray(Traversal& tv, f3 o, f3 d, float tlim) {
  tv.sr = slab_ray2(o, d);
  tv.tlim = tlim;
  tv.cur = 0u;
  tv.stack.sp = 0;
}
template <int K>
DMT_DEV float qbyte(uint32_t w) {  // byte K of w as a float: v_cvt_f32_ubyteK
  return float((w >> (8 * K)) & 0xFFu);
}
// Sort key of a child: the entry distance's bit pattern (t >= 0, so unsigned order == float order) with the child's
// slot number in the two lowest mantissa bits, or kMissKey | slot on a miss.  One v_min_u32 / v_max_u32 pair then is a
// compare-exchange of (distance, child) -- no separate payload to move -- and the reference is rebuilt from the slot
// number after sorting.  The 2 bits perturb the ORDER of children whose distances agree to 22 bits, nothing else.
constexpr uint32_t kMissKey = 0xFFFFFFFCu;
template <int K, int J>
DMT_DEV void slab_keys(uint32_t nqx, uint32_t nqy, uint32_t nqz, uint32_t fqx, uint32_t fqy, uint32_t fqz, f3 a, f3 b, float tlimit,
                       uint32_t& keyK, uint32_t& keyJ) {
  // children K and J side by side in packed registers: six v_pk_fma_f32 instead of twelve v_fma_f32
  v2f const tnx = fma_(v2f{qbyte<K>(nqx), qbyte<J>(nqx)}, a.x, v2f{b.x, b.x});
  v2f const tny = fma_(v2f{qbyte<K>(nqy), qbyte<J>(nqy)}, a.y, v2f{b.y, b.y});
  v2f const tnz = fma_(v2f{qbyte<K>(nqz), qbyte<J>(nqz)}, a.z, v2f{b.z, b.z});
  v2f const tfx = fma_(v2f{qbyte<K>(fqx), qbyte<J>(fqx)}, a.x, v2f{b.x, b.x});
  v2f const tfy = fma_(v2f{qbyte<K>(fqy), qbyte<J>(fqy)}, a.y, v2f{b.y, b.y});
  v2f const tfz = fma_(v2f{qbyte<K>(fqz), qbyte<J>(fqz)}, a.z, v2f{b.z, b.z});
  float const tnK = fmaxf(fmaxf(fmaxf(tnx.x, tny.x), tnz.x), 0.f), tnJ = fmaxf(fmaxf(fmaxf(tnx.y, tny.y), tnz.y), 0.f);
  float const tfK = fminf(fminf(fminf(tfx.x, tfy.x), tfz.x), tlimit), tfJ = fminf(fminf(fminf(tfx.y, tfy.y), tfz.y), tlimit);
  keyK = tnK <= tfK ? ((__float_as_uint(tnK) & ~3u) | uint32_t(K)) : (kMissKey | uint32_t(K));
  keyJ = tnJ <= tfJ ? ((__float_as_uint(tnJ) & ~3u) | uint32_t(J)) : (kMissKey | uint32_t(J));
}
DMT_DEV void kswap(uint32_t& a, uint32_t& b) {
  uint32_t const lo = a < b ? a : b, hi = a < b ? b : a;  // v_min_u32 / v_max_u32
  a = lo, b = hi;
}
// the 48 bytes of a node that a step reads
struct NodeWords {
  uint4 w0;  // ox oy oz meta
  uint4 w1;  // childBase leafRef qlox qhix
  uint4 w2;  // qloy qhiy qloz qhiz
};
DMT_DEV NodeWords node_fetch(BvhView const& bv, uint32_t ref) {
  uint4 const* const nb = reinterpret_cast<uint4 const*>(bv.nodes + ref);
  return {nb[0], nb[1], nb[2]};
}
// node step: cur is an inner node whose words are `nd` (fetched by the caller: the megakernel issues the loads of a lane's
// NEXT node at the end of the step that found it, see megakernel_body_bvh)
template <bool STATS = false>
DMT_DEV void trav_node(BvhView const& bv, Traversal& tv, NodeWords const& nd, TraversalCounters* tc = nullptr) {
  float const tlimit = tv.tlim;
  if constexpr (STATS) ++tc->nodes;
  uint4 const w0 = nd.w0, w1 = nd.w1, w2 = nd.w2;
  uint32_t const meta = w0.w;
  f3 const scale = mk3(__uint_as_float((meta << 23) & 0x7F800000u), __uint_as_float((meta << 15) & 0x7F800000u),
                       __uint_as_float((meta << 7) & 0x7F800000u));
  f3 const a = mk3(scale.x * tv.sr.inv.x, scale.y * tv.sr.inv.y, scale.z * tv.sr.inv.z);
  f3 const b = mk3(fma_(__uint_as_float(w0.x), tv.sr.inv.x, tv.sr.oi.x), fma_(__uint_as_float(w0.y), tv.sr.inv.y, tv.sr.oi.y),
                   fma_(__uint_as_float(w0.z), tv.sr.inv.z, tv.sr.oi.z));
  // 1/d < 0 exactly when d < 0, except d = -0 (1/d = -inf): that axis is dropped as NaN whichever word is called near
  bool const negx = tv.sr.inv.x < 0.f, negy = tv.sr.inv.y < 0.f, negz = tv.sr.inv.z < 0.f;
  uint32_t const nqx = negx ? w1.w : w1.z, fqx = negx ? w1.z : w1.w;
  uint32_t const nqy = negy ? w2.y : w2.x, fqy = negy ? w2.x : w2.y;
  uint32_t const nqz = negz ? w2.w : w2.z, fqz = negz ? w2.z : w2.w;
  // An empty slot holds the inverted box (lo 255, hi 0): on every axis with a finite slope its near plane lies a whole
  // 255 * |a| behind its far plane, so it misses by itself -- unless 255 * |a| vanishes against |b| in fp32 on every axis
  // that has a finite slope (axis-parallel ray, node flat on the other axes).  Then the slot "hits" and its implicit
  // reference names a pair of the next node, or one of the three guard pairs behind the array (buildBvh): a real
  // triangle is tested once more, no result changes, nothing is read out of bounds.  Cheaper than four compares per step.
  uint32_t k0, k1, k2, k3;
  slab_keys<0, 1>(nqx, nqy, nqz, fqx, fqy, fqz, a, b, tlimit, k0, k1);
  slab_keys<2, 3>(nqx, nqy, nqz, fqx, fqy, fqz, a, b, tlimit, k2, k3);
  kswap(k0, k1);
  kswap(k2, k3);
  kswap(k0, k2);
  kswap(k1, k3);
  kswap(k1, k2);
  // implicit references: slot s < inner is node childBase + s, else the flagged leaf reference leafRef + s (the builder stores
  // first pair - inner + kBvhLeafFlag; modulo 2^32 the sum is exact for every leaf slot)
  uint32_t const inner = (meta >> 24) & 0xFu;
  auto ref_of = [&](uint32_t key) {
    uint32_t const slot = key & 3u;
    return (slot < inner ? w1.x : w1.y) + slot;
  };
  uint32_t const r0 = ref_of(k0), r1 = ref_of(k1), r2 = ref_of(k2), r3 = ref_of(k3);
  bool const p3 = k3 < kMissKey, p2 = k2 < kMissKey, p1 = k1 < kMissKey, p0 = k0 < kMissKey;
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
    if (p3) tv.stack.push(bv, r3);
    if (p2) tv.stack.push(bv, r2);
    if (p1) tv.stack.push(bv, r1);
    tv.cur = p0 ? r0 : tv.stack.pop(bv);
  }
}
template <bool STATS = false>
DMT_DEV void trav_node(BvhView const& bv, Traversal& tv, TraversalCounters* tc = nullptr) {
  trav_node<STATS>(bv, tv, node_fetch(bv, tv.cur), tc);
}
// leaf test: `ref` is a leaf reference (one triangle pair) of the ray (o, d); updates the best hit / the occlusion flag only
template <bool STATS = false>
DMT_DEV void trav_leaf_ref(BvhView const& bv, Traversal& tv, f3 o, f3 d, uint32_t ref, TraversalCounters* tc = nullptr) {
  PairHit const h = pair_test(bv.pairs + (ref & ~kBvhLeafFlag), o, d);
  if constexpr (STATS) tc->tris += h.orig0 != h.orig1 ? 2u : 1u;  // a one-triangle leaf repeats its triangle
  if (tv.phase == TR_CLOSEST) {  // brute force keeps the lowest index among equal t (strict < in index order; -1 = none is the largest)
    if (h.valid0 && (h.t.x < tv.tlim || (h.t.x == tv.tlim && h.orig0 < uint32_t(tv.bestTri))))
      tv.tlim = h.t.x, tv.bu = h.u.x, tv.bv = h.v.x, tv.bestTri = int(h.orig0);
    if (h.valid1 && (h.t.y < tv.tlim || (h.t.y == tv.tlim && h.orig1 < uint32_t(tv.bestTri))))
      tv.tlim = h.t.y, tv.bu = h.u.y, tv.bv = h.v.y, tv.bestTri = int(h.orig1);
  } else if ((h.valid0 && h.t.x < tv.tlim) || (h.valid1 && h.t.y < tv.tlim)) {
    tv.tlim = -1.f;
  }
}
// leaf step of the synchronous traversal: cur is a leaf reference
template <bool STATS = false>
DMT_DEV void trav_leaf(BvhView const& bv, Traversal& tv, f3 o, f3 d, TraversalCounters* tc = nullptr) {
  trav_leaf_ref<STATS>(bv, tv, o, d, tv.cur, tc);
  tv.cur = tv.tlim < 0.f ? kBvhEmpty : tv.stack.pop(bv);  // (a closest-hit limit is never negative)
}
// ---- whole traversals of one ray per lane (test kernels, lane_step<BVH>): the same step functions in a loop ----
template <bool STATS>
DMT_DEV void trav_run(BvhView const& bv, Traversal& tv, f3 o, f3 d, TraversalCounters* tc) {
  for (;;) {
    bool const live = tv.cur != kBvhEmpty;
    if (!__any(live)) break;
    bool const onLeaf = live && (tv.cur & kBvhLeafFlag) != 0u;
    if (live && !onLeaf) trav_node<STATS>(bv, tv, tc);
    if (onLeaf) trav_leaf<STATS>(bv, tv, o, d, tc);
  }
}
// closest hit: bestTri = ORIGINAL triangle index or -1
template <bool STATS = false>
DMT_DEV void bvh_closest(BvhView const& bv, bool active, f3 o, f3 d, uint32_t gtid, int& bestTri, float& bt,
                         float& bu, float& bvv, TraversalCounters* tc = nullptr) {
  Traversal tv{};
  if constexpr (STATS) tv.stack.ovfCount = &tc->overflowPushes;
  tv.phase = TR_CLOSEST;
  tv.bestTri = -1, tv.bu = 0.f, tv.bv = 0.f;
  trav_set_ray(tv, o, d, kInf);
  if (!active) tv.cur = kBvhEmpty;
  trav_run<STATS>(bv, tv, o, d, tc);
  bestTri = tv.bestTri, bt = tv.tlim, bu = tv.bu, bvv = tv.bv;
}
// any hit with t < tmax
template <bool STATS = false>
DMT_DEV bool bvh_any(BvhView const& bv, bool active, f3 o, f3 d, float tmax, uint32_t gtid,
                     TraversalCounters* tc = nullptr) {
  Traversal tv{};
  if constexpr (STATS) tv.stack.ovfCount = &tc->overflowPushes;
  tv.phase = TR_SHADOW;
  tv.bestTri = -1;
  trav_set_ray(tv, o, d, tmax);
  if (!active) tv.cur = kBvhEmpty;
  trav_run<STATS>(bv, tv, o, d, tc);
  return tv.occluded();
}

}  // namespace dmt
