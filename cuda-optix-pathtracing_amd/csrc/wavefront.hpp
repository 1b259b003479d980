// wavefront.hpp -- device-side wavefront form of the BVH path (included by dmt_hip.hip inside its anonymous namespace,
// after the megakernel: it reuses path_shade, prepare_sample, the film arithmetic and the traversal steps).
//
// Why it exists.  The BVH megakernel holds 3 waves per SIMD (a lane carries its whole path: shading temporaries +
// sampler + traversal = 162 VGPRs, 51 KB LDS per block) and its throughput still rises from 2 to 3 resident blocks per
// CU (178 / 334 / 445 Msamples/s at 1 / 2 / 3 blocks on BASELINE config 4), which suggested splitting the launch by WORK
// KIND so that traversal alone can run at 8 waves per SIMD -- the split round 1's review asked for.  Built and measured:
// it is bit-identical to the megakernel and SLOWER (1 M triangles, 1024^2: 252 Msamples/s at 2^22 paths per pass, 378 at
// 2^27, megakernel 489).  The traversal kernel's rate does not depend on its occupancy (3, 4, 6, 8 waves per SIMD: the
// same 2.4 G rays/s within 1 %): with the compact node format traversal is bound by VALU issue (84 % busy) and the
// vector-memory address path (TA 45 % busy), not by latency, its lane utilisation is the megakernel's (node steps 61 %,
// leaf steps 31 %: the loss is node / leaf divergence inside a wave, which refilling idle lanes does not touch), and the
// split adds ~1.2 KB of path-state traffic per sample plus ten grid drains per pass while the megakernel overlaps
// shading with other waves' fetches for free.  It stays as strategy 2 of dmt_set_bvh_strategy: a tested alternative
// for scenes whose shading dominates, not the default.  All on the device, no host synchronisation:
//
//   k_wf_generate   one lane per path slot: sampler values + camera ray of (pixel, sample) -> path state, queue 0
//   repeat maxDepth + 2 times (enqueued back to back; an iteration whose queue is empty exits at once):
//     k_wf_trace    TRAVERSAL ONLY, persistent waves at 8 waves per SIMD (<= 64 VGPRs, 16 KB LDS per block for the
//                   stacks): lanes pull path indices from the iteration's queue with a wave-wide atomic (work
//                   stealing), run closest-hit then shadow ray through the 4-wide BVH with majority node / leaf
//                   stepping, write the hit record, and REFILL as soon as enough lanes of the wave are idle
//     k_wf_shade    one lane per queued path: resolve the shadow ray, shade the hit with the megakernel's own
//                   path_shade (identical arithmetic, so films stay bit-identical to the megakernel's and hence to
//                   brute force), append surviving paths to the next queue by wave ballot + prefix count + one atomic
//   k_wf_fold       one lane per pixel: the reference's Welford update over the pass's samples IN SAMPLE ORDER
//
// A pass covers a range of owned tiles x a range of n samples; slot = sample-in-pass x pixelSlots + (tile x 64 + pixel),
// so consecutive lanes are neighbouring pixels of one sample (coherent camera rays).  Path state lives in HBM as
// structure-of-arrays planes of 4 bytes x slots (144 B per path): the cost of the split, ~1.2 KB of coalesced traffic
// per sample against ~10 KB of gathered BVH bytes.
#pragma once

// ---- path state planes ------------------------------------------------------------------------------
enum : int {
  WF_OX = 0, WF_OY, WF_OZ, WF_DX, WF_DY, WF_DZ,                 // closest-hit ray of the next trace
  WF_SOX, WF_SOY, WF_SOZ, WF_SDX, WF_SDY, WF_SDZ, WF_SMAX,      // pending shadow ray
  WF_BX, WF_BY, WF_BZ,                                          // throughput beta
  WF_LX, WF_LY, WF_LZ,                                          // radiance so far
  WF_CX, WF_CY, WF_CZ,                                          // pending NEE contribution (added when unoccluded)
  WF_FLAGS,                                                     // bit 0 active, 1 hasShadow, 2 lastT, 3 lastSpecular; 8-15 depth; 16-19 sampler dimension
  WF_LASTPDF,
  WF_U0,                                                        // 8 sampler values
  WF_HIT_TRI = WF_U0 + 8, WF_HIT_U, WF_HIT_V, WF_OCCLUDED,
  WF_PLANES
};
constexpr uint32_t kWfActive = 1u, kWfShadow = 2u, kWfLastT = 4u, kWfLastSpecular = 8u;

struct WfParams {
  float* state;        // [WF_PLANES][slots]
  uint32_t* queue[2];  // path indices, ping-pong by iteration
  uint32_t* counts;    // [iterations + 2] queue sizes; counts[it + 1] is filled by k_wf_shade(it)
  uint32_t* cursors;   // [iterations + 2] work-stealing cursors of k_wf_trace
  uint32_t slots;      // pixelSlots * n
  uint32_t pixelSlots; // tiles of the pass * 64
  uint32_t tile0;      // first owned tile (item index) of the pass
  uint32_t s0, n;      // sample range of the pass
  uint32_t it;         // iteration (k_wf_trace / k_wf_shade)
};

DMT_DEV float* wf_plane(WfParams const& W, int plane) { return W.state + size_t(plane) * W.slots; }

// owned tile `item` of the launch -> pixel of `lane`, inside the render region?
DMT_DEV bool wf_pixel(KArgs Pk, uint32_t item, int lane, int& px, int& py) {
  TileArgs const T = load_tile_args(Pk);
  uint32_t const j = uint32_t(T.rank) + item * uint32_t(T.world);
  px = (T.tx0 + int(j % uint32_t(T.rtx))) * 8 + (lane & 7);
  py = (T.ty0 + int(j / uint32_t(T.rtx))) * 8 + (lane >> 3);
  return px >= T.x0 && px < T.x1 && py >= T.y0 && py < T.y1;
}

// append the lanes with `keep` to a queue: one atomic per wave, order within the wave preserved
DMT_DEV void wf_append(uint32_t* queue, uint32_t* count, int lane, bool keep, uint32_t value) {
  unsigned long long const m = __ballot(keep);
  uint32_t const n = uint32_t(__popcll(m));
  if (n == 0u) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(count, n);
  base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
  if (keep) queue[base + uint32_t(__popcll(m & ((1ull << lane) - 1ull)))] = value;
}

// ---- generate ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_wf_generate(RenderParams P, WfParams W) {
  KArgs const Pk = kargs_base();
  int const lane = int(threadIdx.x) & 63;
  uint32_t const stride = gridDim.x * 256u;
  for (uint32_t base = blockIdx.x * 256u + (threadIdx.x & ~63u); base < W.slots; base += stride) {
    uint32_t const slot = base + uint32_t(lane);  // slots is a multiple of 64
    uint32_t const j = slot / W.pixelSlots, pix = slot - j * W.pixelSlots;
    int px, py;
    bool const inside = wf_pixel(Pk, W.tile0 + (pix >> 6), int(pix & 63u), px, py);
    if (inside) {
      ColdArgs const cold = load_cold_args(Pk);
      prepare_sample(Pk, px, py, halton_pixel_base(cold.sp, px, py), W.s0 + j);
      float const* const prep = s_prep + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 8; ++k) wf_plane(W, WF_U0 + k)[slot] = prep[k * kLdsThreads];
#pragma unroll
      for (int k = 0; k < 6; ++k) wf_plane(W, WF_OX + k)[slot] = prep[(8 + k) * kLdsThreads];
      wf_plane(W, WF_BX)[slot] = 1.f, wf_plane(W, WF_BY)[slot] = 1.f, wf_plane(W, WF_BZ)[slot] = 1.f;
      wf_plane(W, WF_LASTPDF)[slot] = 0.f;
    }
    wf_plane(W, WF_LX)[slot] = 0.f, wf_plane(W, WF_LY)[slot] = 0.f, wf_plane(W, WF_LZ)[slot] = 0.f;
    reinterpret_cast<uint32_t*>(wf_plane(W, WF_FLAGS))[slot] = inside ? (kWfActive | (2u << 16)) : 0u;
    wf_append(W.queue[0], W.counts, lane, inside, slot);
  }
}

// ---- trace ------------------------------------------------------------------------------------------------
#ifndef DMT_WF_REFILL
#define DMT_WF_REFILL 16  // idle lanes of a wave that trigger a refill from the queue
#endif
#ifndef DMT_WF_CHUNK
#define DMT_WF_CHUNK 256  // queue entries a wave draws per atomic
#endif
#ifndef DMT_WF_TRACE_WAVES
#define DMT_WF_TRACE_WAVES 8
#endif
template <bool STATS>
DMT_DEV void wf_trace_body(WfParams const& W) {
  KArgs const Pk = kargs_base();
  int const lane = int(threadIdx.x) & 63;
  uint32_t const gtid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t const count = W.counts[W.it];
  if (count == 0u) return;
  uint32_t const* const queue = W.queue[W.it & 1u];
  uint32_t* const cursor = W.cursors + W.it;
  uint32_t const* const flagsPlane = reinterpret_cast<uint32_t const*>(wf_plane(W, WF_FLAGS));
  BvhView const bvh = load_bvh(Pk);
  LaneStats ls;
  Traversal tv{};
  tv.phase = TR_IDLE;
  tv.cur = kBvhEmpty;
  uint32_t path = 0;
  f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1);  // the ray this lane is tracing
  // The wave draws CHUNKS of the queue with one atomic each and hands their entries to its lanes as they fall idle: one
  // shared cursor word saturates near 90 dequeues per microsecond on this chip (MI355X_MICROARCH.md, price list
  // "dequeue"), which a per-refill atomic reaches at ~1.4 G rays/s.
  uint32_t chunkNext = 0, chunkEnd = 0;  // wave-uniform: undistributed entries of the current chunk
  bool exhausted = false;                // the cursor has passed the end of the queue
  auto start_shadow = [&]() {
    tv.phase = TR_SHADOW;
    ro = mk3(wf_plane(W, WF_SOX)[path], wf_plane(W, WF_SOY)[path], wf_plane(W, WF_SOZ)[path]);
    rd = mk3(wf_plane(W, WF_SDX)[path], wf_plane(W, WF_SDY)[path], wf_plane(W, WF_SDZ)[path]);
    trav_set_ray(tv, ro, rd, wf_plane(W, WF_SMAX)[path]);
    if constexpr (STATS) ++ls.shadow;
  };
  for (;;) {
    // Lanes whose ray has finished, and idle lanes, need SERVICE: record stores, the shadow ray's loads, a new path
    // from the queue.  All of that is vector-memory work that costs the wave the same whether one lane or sixty-four
    // take part (~40 clocks of the CU's address path per instruction), so it is batched: a service step runs when
    // DMT_WF_REFILL lanes are waiting for it, or when nobody can traverse.  (Servicing every finished ray at once, a
    // dozen memory instructions for one or two lanes on almost every iteration, cost 3x the traversal itself.)
    bool const traversing = tv.phase == TR_CLOSEST || tv.phase == TR_SHADOW;
    bool const finished = traversing && tv.cur == kBvhEmpty;
    bool const drained = exhausted && chunkNext == chunkEnd;
    bool const wantsPath = tv.phase == TR_IDLE && !drained;
    int const nService = __popcll(__ballot(finished || wantsPath));
    bool const canStep = traversing && !finished;
    bool const anyStep = __any(canStep);
    if (!anyStep && nService == 0) break;  // every lane idle and the queue drained
    if (nService >= DMT_WF_REFILL || !anyStep) {
      if (finished) {
        if (tv.phase == TR_CLOSEST) {
          reinterpret_cast<int32_t*>(wf_plane(W, WF_HIT_TRI))[path] = tv.bestTri;
          wf_plane(W, WF_HIT_U)[path] = tv.bu, wf_plane(W, WF_HIT_V)[path] = tv.bv;
          if (tv.doS) start_shadow();
          else tv.phase = TR_IDLE;
        } else {
          reinterpret_cast<uint32_t*>(wf_plane(W, WF_OCCLUDED))[path] = tv.occluded() ? 1u : 0u;
          tv.phase = TR_IDLE;
        }
      }
      bool const idle = tv.phase == TR_IDLE;
      unsigned long long const idleMask = __ballot(idle);
      int const nIdle = __popcll(idleMask);
      if (!drained && nIdle > 0) {
        if (chunkNext == chunkEnd) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor, uint32_t(DMT_WF_CHUNK));
          base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
          chunkNext = base < count ? base : count;
          chunkEnd = base + uint32_t(DMT_WF_CHUNK) < count ? base + uint32_t(DMT_WF_CHUNK) : count;
          if (base + uint32_t(DMT_WF_CHUNK) >= count) exhausted = true;
        }
        uint32_t const my = chunkNext + uint32_t(__popcll(idleMask & ((1ull << lane) - 1ull)));
        bool const take = idle && my < chunkEnd;
        chunkNext = chunkNext + uint32_t(nIdle) < chunkEnd ? chunkNext + uint32_t(nIdle) : chunkEnd;
        if (take) {
          path = queue[my];
          uint32_t const f = flagsPlane[path];
          tv.doC = (f & kWfActive) != 0u, tv.doS = (f & kWfShadow) != 0u;
          tv.bestTri = -1, tv.bu = 0.f, tv.bv = 0.f;
          if (tv.doC) {
            tv.phase = TR_CLOSEST;
            ro = mk3(wf_plane(W, WF_OX)[path], wf_plane(W, WF_OY)[path], wf_plane(W, WF_OZ)[path]);
            rd = mk3(wf_plane(W, WF_DX)[path], wf_plane(W, WF_DY)[path], wf_plane(W, WF_DZ)[path]);
            trav_set_ray(tv, ro, rd, kInf);
            if constexpr (STATS) ++ls.closest;
          } else if (tv.doS) {
            start_shadow();
          }
        }
      }
      continue;
    }
    // one node step or one leaf step, whichever serves more lanes per unit of cost (see megakernel_body_bvh)
    bool const onNode = canStep && !(tv.cur & kBvhLeafFlag);
    bool const onLeaf = canStep && (tv.cur & kBvhLeafFlag) != 0u;
    int const nNode = __popcll(__ballot(onNode)), nLeaf = __popcll(__ballot(onLeaf));
    if (nNode * DMT_BVH_NODE_WEIGHT >= nLeaf * DMT_BVH_LEAF_WEIGHT) {
      if constexpr (STATS) ++ls.itNode;
      if (onNode) trav_node<STATS>(bvh, tv, STATS ? &ls.tc : nullptr);
    } else {
      if constexpr (STATS) ++ls.itLeaf, ls.lanesLeaf += onLeaf ? 1u : 0u;
      if (onLeaf) trav_leaf<STATS>(bvh, tv, ro, rd, STATS ? &ls.tc : nullptr);
    }
  }
  flush_stats<STATS>(Pk, ls);
}
__global__ void __launch_bounds__(256, DMT_WF_TRACE_WAVES) k_wf_trace(RenderParams P, WfParams W) { wf_trace_body<false>(W); }
__global__ void __launch_bounds__(256, 4) k_wf_trace_stats(RenderParams P, WfParams W) { wf_trace_body<true>(W); }

// ---- shade ------------------------------------------------------------------------------------------------
template <bool ENV, bool AREA, bool STATS>
DMT_DEV void wf_shade_body(WfParams const& W) {
  KArgs const Pk = kargs_base();
  int const lane = int(threadIdx.x) & 63;
  uint32_t const count = W.counts[W.it];
  if (count == 0u) return;
  uint32_t const* const queue = W.queue[W.it & 1u];
  uint32_t* const flagsPlane = reinterpret_cast<uint32_t*>(wf_plane(W, WF_FLAGS));
  uint32_t const stride = gridDim.x * 256u;
  LaneStats ls;
  for (uint32_t base = blockIdx.x * 256u + (threadIdx.x & ~63u); base < count; base += stride) {
    uint32_t const i = base + uint32_t(lane);
    bool const valid = i < count;
    bool alive = false;
    uint32_t path = 0;
    if (valid) {
      path = queue[i];
      uint32_t const f = flagsPlane[path];
      bool const doC = (f & kWfActive) != 0u, doS = (f & kWfShadow) != 0u;
      PathState st{};
      st.L = mk3(wf_plane(W, WF_LX)[path], wf_plane(W, WF_LY)[path], wf_plane(W, WF_LZ)[path]);
      bool changedL = false;
      if (doS && reinterpret_cast<uint32_t const*>(wf_plane(W, WF_OCCLUDED))[path] == 0u) {
        // NEE of the previous bounce, added before anything of this bounce (megakernel.cu:219-240, lane_finish)
        st.L = st.L + mk3(wf_plane(W, WF_CX)[path], wf_plane(W, WF_CY)[path], wf_plane(W, WF_CZ)[path]);
        changedL = true;
      }
      uint32_t nf = f & ~(kWfActive | kWfShadow);
      if (doC) {
        float* const u = s_sampler_u + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k * kLdsThreads] = wf_plane(W, WF_U0 + k)[path];
        st.rng.dim = int((f >> 16) & 15u);
        st.depth = int((f >> 8) & 255u);
        st.lastT = (f & kWfLastT) != 0u;
        st.lastSpecular = (f & kWfLastSpecular) != 0u;
        st.lastPdf = wf_plane(W, WF_LASTPDF)[path];
        st.beta = mk3(wf_plane(W, WF_BX)[path], wf_plane(W, WF_BY)[path], wf_plane(W, WF_BZ)[path]);
        set_ray(st, mk3(wf_plane(W, WF_OX)[path], wf_plane(W, WF_OY)[path], wf_plane(W, WF_OZ)[path]),
                mk3(wf_plane(W, WF_DX)[path], wf_plane(W, WF_DY)[path], wf_plane(W, WF_DZ)[path]));
        st.active = true, st.hasShadow = false;
        int const tri = reinterpret_cast<int32_t const*>(wf_plane(W, WF_HIT_TRI))[path];
        if constexpr (STATS) ls.bounces += (tri >= 0 && st.depth < kargs(Pk)->maxDepth) ? 1u : 0u;
        bool const ended = path_shade<ENV, AREA>(Pk, st, tri, wf_plane(W, WF_HIT_U)[path], wf_plane(W, WF_HIT_V)[path]);
        changedL = true;
        if (!ended) {
          wf_plane(W, WF_OX)[path] = st.rp.ox.x, wf_plane(W, WF_OY)[path] = st.rp.oy.x, wf_plane(W, WF_OZ)[path] = st.rp.oz.x;
          wf_plane(W, WF_DX)[path] = st.rp.dx.x, wf_plane(W, WF_DY)[path] = st.rp.dy.x, wf_plane(W, WF_DZ)[path] = st.rp.dz.x;
          wf_plane(W, WF_BX)[path] = st.beta.x, wf_plane(W, WF_BY)[path] = st.beta.y, wf_plane(W, WF_BZ)[path] = st.beta.z;
          if constexpr (ENV || AREA) wf_plane(W, WF_LASTPDF)[path] = st.lastPdf;
        }
        if (st.hasShadow) {
          wf_plane(W, WF_SOX)[path] = st.rp.ox.y, wf_plane(W, WF_SOY)[path] = st.rp.oy.y, wf_plane(W, WF_SOZ)[path] = st.rp.oz.y;
          wf_plane(W, WF_SDX)[path] = st.rp.dx.y, wf_plane(W, WF_SDY)[path] = st.rp.dy.y, wf_plane(W, WF_SDZ)[path] = st.rp.dz.y;
          wf_plane(W, WF_SMAX)[path] = st.smax;
          f3 const C = get_C();
          wf_plane(W, WF_CX)[path] = C.x, wf_plane(W, WF_CY)[path] = C.y, wf_plane(W, WF_CZ)[path] = C.z;
        }
        nf = (ended ? 0u : kWfActive) | (st.hasShadow ? kWfShadow : 0u) | (st.lastT ? kWfLastT : 0u) |
             (st.lastSpecular ? kWfLastSpecular : 0u) | (uint32_t(st.depth) << 8) | (uint32_t(st.rng.dim) << 16);
      }
      if (changedL) wf_plane(W, WF_LX)[path] = st.L.x, wf_plane(W, WF_LY)[path] = st.L.y, wf_plane(W, WF_LZ)[path] = st.L.z;
      flagsPlane[path] = nf;
      alive = (nf & (kWfActive | kWfShadow)) != 0u;
    }
    wf_append(W.queue[(W.it + 1u) & 1u], W.counts + W.it + 1u, lane, alive, path);
  }
  flush_stats<STATS>(Pk, ls);
}
__global__ void __launch_bounds__(256, 4) k_wf_shade(RenderParams P, WfParams W) { wf_shade_body<false, false, false>(W); }
__global__ void __launch_bounds__(256, 4) k_wf_shade_env(RenderParams P, WfParams W) { wf_shade_body<true, false, false>(W); }
__global__ void __launch_bounds__(256, 4) k_wf_shade_area(RenderParams P, WfParams W) { wf_shade_body<false, true, false>(W); }
__global__ void __launch_bounds__(256, 4) k_wf_shade_env_area(RenderParams P, WfParams W) { wf_shade_body<true, true, false>(W); }
__global__ void __launch_bounds__(256, 2) k_wf_shade_stats(RenderParams P, WfParams W) { wf_shade_body<false, false, true>(W); }
__global__ void __launch_bounds__(256, 2) k_wf_shade_stats_env(RenderParams P, WfParams W) { wf_shade_body<true, false, true>(W); }

// ---- fold -------------------------------------------------------------------------------------------------
// One lane per pixel: running (mean, M2, N) of the film + the pass's n radiances in sample order
// (SMEMLayout::startSample / updateSample / endSample, T/megakernel/megakernel.cuh:45-85; same expressions as item_fold).
__global__ void __launch_bounds__(256) k_wf_fold(RenderParams P, WfParams W) {
  KArgs const Pk = kargs_base();
  TileArgs const T = load_tile_args(Pk);
  uint32_t const stride = gridDim.x * 256u;
  for (uint32_t pix = blockIdx.x * 256u + threadIdx.x; pix < W.pixelSlots; pix += stride) {
    int px, py;
    if (!wf_pixel(Pk, W.tile0 + (pix >> 6), int(pix & 63u), px, py)) continue;
    size_t const pidx = size_t(px) + size_t(py) * size_t(T.width);
    float4 const m = T.mean[pidx], v = T.m2[pidx];
    f3 mean = mk3(m.x, m.y, m.z), M2 = mk3(v.x, v.y, v.z);
    float N = v.w;
    float const *lx = wf_plane(W, WF_LX) + pix, *ly = wf_plane(W, WF_LY) + pix, *lz = wf_plane(W, WF_LZ) + pix;
    for (uint32_t k = 0; k < W.n; ++k) {
      welford_update(mean, M2, N, mk3(lx[size_t(k) * W.pixelSlots], ly[size_t(k) * W.pixelSlots], lz[size_t(k) * W.pixelSlots]));
    }
    T.mean[pidx] = make_float4(mean.x, mean.y, mean.z, 0.f);
    T.m2[pidx] = make_float4(M2.x, M2.y, M2.z, N);
  }
}
