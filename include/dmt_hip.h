/* dmt_hip.h -- C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * Drop-in boundary for the reference's kernel-launch boundary (paths relative to
 * /root/reference/examples/triangles/, T/ = that directory, CC/ = T/cuda-core/):
 *
 *   reference interface                                          replaced by
 *   ------------------------------------------------------------ ---------------------------
 *   cudaInitDevice/cudaSetDevice/cudaStreamCreate                dmt_ctx_create / _destroy
 *     (T/megakernel/main.cu:135-136,264-265)
 *   triSoupFromTriangles (CC/public/cuda-core/host_utils.cuh:163) dmt_upload_triangles
 *   deviceBSDF           (host_utils.cuh:166)                    dmt_upload_bsdfs
 *   deviceLights         (host_utils.cuh:167-169)                dmt_upload_lights
 *   deviceCamera + allocateDeviceConstantMemory                  dmt_set_camera
 *     (host_utils.cuh:170, T/megakernel/megakernel.cuh:18)
 *   DeviceOutputBuffer::allocate/free (CC/public/cuda-core/types.cuh:175-193)
 *                                                                dmt_set_camera / dmt_film_clear /
 *                                                                dmt_film_bind
 *   copyHaltonOwenToDeviceAlloc (host_utils.cuh:159)             (none: sampler state is per-lane
 *                                                                registers, nothing to upload)
 *   pathTraceMegakernel<<<...>>>(..., sampleOffset, ...)         dmt_render
 *     (T/megakernel/megakernel.cuh:99-112, launch main.cu:141-155)
 *   cudaStreamSynchronize (main.cu:169)                          dmt_sync
 *   cudaMemcpyAsync D2H of mean / M2 (main.cu:202-205)           dmt_download_film
 *   triangleIntersectKernel (T/tests/triangle_intersect.cu:146)  dmt_test_triangle_intersect
 *
 * Record layouts are byte-identical to the reference so its host packers interoperate:
 *   BSDF  32 B  CC/public/cuda-core/bsdf.cuh:18-73
 *   Light 32 B  CC/public/cuda-core/light.cuh:10-49
 *   DeviceCamera 44 B  CC/public/cuda-core/types.cuh:101-109  (= dmt_camera below)
 *   TriangleSoup: one float4 per axis per triangle {c0,c1,c2,pad} + uint32 matId
 *                 (types.cuh:119-129)
 *   film: two row-major float4 planes, mean.xyz (w = 0) and M2.xyz with the sample count N
 *         in .w  (T/megakernel/megakernel.cuh:81-85, megakernel.cu:92-93)
 *
 * Ownership: the context owns all device memory; host pointers are borrowed for the call.
 * Errors: every call returns DMT_OK (0) or a DMT_ERR_* code; the library never exits the
 * process (the reference's CUDA_CHECK does, types.cuh:20-29).  A context is bound to one device
 * and one stream and is not thread-safe; contexts are independent (one per GPU).
 */
#ifndef DMT_HIP_H
#define DMT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmt_ctx dmt_ctx;

typedef struct dmt_camera {
  float dir[3];
  float pos[3];
  int32_t width;
  int32_t height;
  int32_t spp; /* kept for layout parity; dmt_render's spp argument is what is rendered */
  float focal_length; /* mm */
  float sensor_size;  /* mm */
} dmt_camera;

enum {
  DMT_OK = 0,
  DMT_ERR_INVALID = 1,   /* bad argument */
  DMT_ERR_HIP = 2,       /* a HIP runtime call failed; see dmt_last_error */
  DMT_ERR_STATE = 3,     /* call sequence error (e.g. render before upload) */
  DMT_ERR_NO_DEVICE = 4, /* no usable GPU */
};

enum {
  DMT_ACCEL_BRUTE_FORCE = 0, /* the reference's loop over all triangles */
  DMT_ACCEL_BVH = 1,         /* same closest hit (lowest index on ties), BVH traversal */
};

/* ---- lifetime ---------------------------------------------------------------------------- */
int dmt_ctx_create(int device_ordinal, dmt_ctx** out);
int dmt_ctx_destroy(dmt_ctx* ctx);
/* last error text of the context; ctx == NULL returns the last dmt_ctx_create failure */
const char* dmt_last_error(const dmt_ctx* ctx);

/* ---- scene upload ------------------------------------------------------------------------ */
int dmt_upload_triangles(dmt_ctx* ctx, const float* xs, const float* ys, const float* zs,
                         const uint32_t* mat_id, size_t count);
int dmt_upload_bsdfs(dmt_ctx* ctx, const void* bsdf32, uint32_t count);
int dmt_upload_lights(dmt_ctx* ctx, const void* lights32, uint32_t count, const void* infinite32,
                      uint32_t infinite_count);
/* (re)computes camera transforms and sampler parameters; (re)allocates and zeroes the film when
 * the resolution changes */
int dmt_set_camera(dmt_ctx* ctx, const dmt_camera* cam);
/* depth cap of the bounce loop; the reference hard-codes 32 (megakernel.cu:154) */
int dmt_set_limits(dmt_ctx* ctx, int max_depth);
int dmt_set_accel(dmt_ctx* ctx, int mode);
/* SURVEY 8f-4 -- how next-event estimation picks its light from the uploaded light list.  DMT_LIGHTS_UNIFORM (default) is
 * the reference megakernel's uniform pick (T/megakernel/megakernel.cu:170-173): the parity mode.  DMT_LIGHTS_TREE builds a
 * light BVH over the point / spot lights (after src/core/public/core-light-tree-builder.h:17-110; differences and why in
 * csrc/light_tree.hpp) and picks in proportion to flux x cosine / distance^2: same expected image, less noise with many
 * lights.  Applies to light lists of point / spot lights (scenes with emissive triangles, image textures, or a
 * directional light in the list keep the uniform pick). */
#define DMT_LIGHTS_UNIFORM 0
#define DMT_LIGHTS_TREE 1
/* the reference's light tree with its OWN semantics (src/core/public/core-light-tree-builder.h:17-104, .cpp:5-539): LightBounds with
 * normal cone and emission falloff, lbImportance's orientation term, 32-bin summed-area-orientation splits, an adaptive cut of
 * up to LightTreeMaxSplitSize = 4 nodes per shading point = up to four lights and four shadow rays per bounce
 * (csrc/light_tree_ref.hpp lists what is kept as written and the four places that had to be corrected).  Same applicability
 * rule as DMT_LIGHTS_TREE.  Unpinned by the reference (experimental, disabled code without outputs). */
#define DMT_LIGHTS_TREE_REFERENCE 2
int dmt_set_light_sampling(dmt_ctx* ctx, int mode);
/* host only (no GPU): the cut + selection of DMT_LIGHTS_TREE_REFERENCE at n shading points p3 / n3 with one random number u each:
 * up to four (light index, pmf) pairs per point (indices4 = -1 beyond counts[i]); start_pmf = probability that the light list
 * (not the env map) was asked */
int dmt_light_tree_ref_select(const void* lights32, uint32_t count, int n, const float* p3, const float* n3, const float* u, float start_pmf,
                              int32_t* indices4, float* pmfs4, int32_t* counts, int* node_count, int* depth);
/* host only (no GPU): probability of each of the `count` packed lights at point p3 with normal n3 under the tree */
int dmt_light_tree_pmfs(const void* lights32, uint32_t count, const float* p3, const float* n3, float* pmf_out, int* node_count,
                        int* depth);
/* How DMT_ACCEL_BVH launches are executed (results are bit-identical either way): 0 = automatic = 1 = the megakernel
 * (fastest on every measured scene); 2 = the device-side wavefront -- generate / trace / shade / fold kernels over
 * path-state arrays in HBM, csrc/wavefront.hpp -- kept as a measured alternative.  paths_per_pass: path
 * slots of one wavefront pass (0 = keep; default 2^22, 144 bytes of state each; more is faster, up to ~2^27). */
int dmt_set_bvh_strategy(dmt_ctx* ctx, int strategy, uint64_t paths_per_pass);
/* tile partition for multi-GPU rendering: this context renders only the 8x8-pixel tiles whose
 * index (row-major over the tile grid) is congruent to `rank` modulo `world`.  Default 0 of 1. */
int dmt_set_partition(dmt_ctx* ctx, int rank, int world);
/* SURVEY 8f-3 -- emissive triangles (diffuse area lights; what scenes/cornell-box.pbrt's AreaLightSource needs).  The
 * reference has no implementation; semantics are pbrt-v4's: one-sided emission on the side of
 * normalize(cross(p1 - p0, p2 - p0)), uniform point sampling, uniform choice among [uploaded lights..., emissive
 * triangles...], power-heuristic MIS.  triangle_index refers to the triangles of the last dmt_upload_triangles (which
 * clears this list); count == 0 removes the lights.  With an env map set, the map still takes half of the NEE samples. */
int dmt_upload_area_lights(dmt_ctx* ctx, const uint32_t* triangle_index, const float* radiance_rgb, uint32_t count);
/* A18 -- environment-map light of the reference's CPU renderer (src/core/private/core-light.cpp:84-117,394-491;
 * PiecewiseConstant2D src/core/private/core-math.cu:385-675; MIS rules src/core/private/core-render.cpp:154-163,
 * 290-299,357-369), added to the megakernel path: rgb = height x width x 3 floats (equirectangular, powers of two,
 * width == 2 * height), quat_xyzw = lightFromRender.  While an env map is set, rays that leave the scene see the map
 * (MIS against NEE) instead of the constant environment records, and NEE samples the map with probability 1/2.
 * `scale` is accepted for signature parity; the reference stores it and never applies it. */
int dmt_upload_envmap(dmt_ctx* ctx, const float* rgb, int width, int height, const float* quat_xyzw, float scale);
int dmt_clear_envmap(dmt_ctx* ctx);
/* SURVEY 8f-1 -- image textures of JSON materials ("textures" + string-valued "diffuse" / "roughness" / "normal" of
 * src/core/private/core-parser.cpp:306-560; the reference's megakernel has none, semantics are its CPU renderer's,
 * src/core/private/core-material.cpp:20-56,180-240): bilinear at MIP level 0, mirror wrap, byte / 255; the sampled albedo
 * and roughness patch the material's packed record as the host packers would build it, a normal map turns the geometric
 * normal into the shading normal.  rgba8: texel_count RGBA8 texels, all textures back to back, row major;
 * desc3[texture] = {first texel, width, height}; mat_tex4[bsdf] = {diffuse, roughness, normal texture index or
 * 0xFFFFFFFF, anisotropy as float bits}; tri_uv6[triangle] = {u0, v0, u1, v1, u2, v2}.  Upload AFTER triangles and
 * BSDFs (the counts must match at render time); texture_count == 0 clears.  Not combinable with emissive triangles. */
int dmt_upload_textures(dmt_ctx* ctx, const uint8_t* rgba8, uint64_t texel_count, const int32_t* desc3, uint32_t texture_count,
                        const uint32_t* mat_tex4, uint32_t bsdf_count, const float* tri_uv6, uint64_t triangle_count);
/* host only (no GPU needed): the sampling tables dmt_upload_envmap builds; func/cdf: height*width floats each,
 * row_integral / marginal_func / marginal_cdf: height floats each */
int dmt_envmap_tables(const float* rgb, int width, int height, float* func, float* cdf, float* row_integral,
                      float* marginal_func, float* marginal_cdf, float* marginal_integral);
/* device probes of the env-map functions: per case u2 -> sampled wi3, pdf, uv2, Le3 (by uv), ok; and
 * wi_in3 -> Le_dir3, pdf_dir (evaluation by direction) */
int dmt_test_envmap(dmt_ctx* ctx, int n, const float* u2, const float* wi_in3, float* wi3, float* pdf, float* uv2,
                    float* Le3, int32_t* ok, float* Le_dir3, float* pdf_dir);
/* samples per work item inside one dmt_render pass (0 = automatic, the default: 1 024 path samples per item; at most 512).  Purely a
 * scheduling knob: a pixel's samples are folded in index order for any value, the film is bit-identical. */
int dmt_set_chunk(dmt_ctx* ctx, uint32_t samples_per_item);
/* borrow an external hipStream_t (e.g. the caller's); NULL restores the context's own stream */
int dmt_set_stream(dmt_ctx* ctx, void* hip_stream);

/* ---- film ---------------------------------------------------------------------------------- */
int dmt_film_clear(dmt_ctx* ctx);
/* use caller-owned device buffers (width*height float4 each) instead of the context's own */
int dmt_film_bind(dmt_ctx* ctx, void* d_mean, void* d_m2);
int dmt_film_device_ptrs(dmt_ctx* ctx, void** d_mean, void** d_m2);
/* synchronises the stream; DMT_ERR_HIP (and no copy) if a past launch did not fold every sample chunk exactly once (see dmt_sync) */
int dmt_download_film(dmt_ctx* ctx, float* mean4, float* m24);

/* ---- render -------------------------------------------------------------------------------- */
/* Enqueue one pass: samples [sample_offset, sample_offset + spp) of every owned pixel in
 * [x0,x1) x [y0,y1) are traced and folded into the film (Welford, in sample order).
 * Asynchronous on the context's stream. */
int dmt_render(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1);
/* Same pass through the counting build of the BVH kernel (synchronous, not for timing): stats6 =
 * {samples, closest-hit rays, shadow rays, BVH node visits, triangle tests, bounces}.  The film is
 * updated exactly as by dmt_render. */
int dmt_render_stats(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1,
                     uint64_t* stats6);
/* Diagnostic: dmt_render_stats plus the loop profile of the BVH kernel.  stats16 = the six counters above, then
 * wave-loop iterations x 64 (node steps, leaf steps, shading steps, outer iterations, sample preparations) and the
 * lanes that did work in leaf / shading / preparation steps; node visits that entered no child; traversal-stack pushes that
 * went to the global overflow area (entries beyond the LDS part of the stack). */
int dmt_render_profile(dmt_ctx* ctx, uint32_t sample_offset, uint32_t spp, int x0, int y0, int x1, int y1,
                       uint64_t* stats16);
/* waits for the stream; DMT_ERR_HIP if the launches so far did not fold every (tile, sample chunk) into the film exactly
 * once (the in-launch hand-over that makes the film schedule-independent counts its folds): the film is then invalid.
 * dmt_download_film and dmt_kernel_time report the same condition. */
int dmt_sync(dmt_ctx* ctx);
/* Diagnostics of the in-launch fold hand-over, accumulated over the launches since the last reset (synchronises):
 * out8 = {sample chunks folded, chunks handed over to the folder of their predecessor, chunks folded on behalf of another
 * wave, times a wave found all its staging slabs handed over, waves that left the launch early because of that,
 * longest such stall in 10 ns ticks, sample chunks launched (must equal [0]), staging slabs per wave}. */
int dmt_sched_diag(dmt_ctx* ctx, uint64_t* out8, int reset);
/* HIP-event time of the megakernel launches since the last reset (synchronises the stream):
 * total milliseconds and launch count. */
int dmt_kernel_time(dmt_ctx* ctx, double* total_ms, uint64_t* launches, int reset);
/* compile-time facts of the loaded code object: vgprs, sgprs, LDS bytes, max resident waves/CU
 * as reported by the runtime for the megakernel; CU count of the device */
int dmt_kernel_info(dmt_ctx* ctx, int* vgprs, int* sgprs, int* lds_bytes, int* blocks_per_cu,
                    int* cu_count);

/* host-only check of the BVH builder behind DMT_ACCEL_BVH (no GPU needed): builds the tree of a soup
 * and verifies that every triangle sits in exactly one leaf, child boxes nest and enclose their
 * vertices, leaves hold <= 4 triangles and the depth respects the traversal-stack bound */
int dmt_bvh_validate(const float* xs, const float* ys, const float* zs, size_t count, int* node_count,
                     int* depth, int* max_leaf);

/* ---- device unit-test entry points (GPU twins of the reference's T/tests kernels) ---------- */
int dmt_test_triangle_intersect(dmt_ctx* ctx, const float* xs, const float* ys, const float* zs,
                                size_t count, const float* o3, const float* d3, int32_t* hit,
                                float* t, float* pos3, float* nrm3, float* err3);
int dmt_test_sampler(dmt_ctx* ctx, int width, int height, int n, const int32_t* pxs,
                     const int32_t* pys, const int32_t* ss, int ndims, int32_t* halton_index,
                     float* pixel2d, float* dims);
int dmt_test_camera_rays(dmt_ctx* ctx, int n, const int32_t* pxs, const int32_t* pys,
                         const int32_t* ss, float* o3, float* d3);
int dmt_test_bsdf(dmt_ctx* ctx, const void* bsdf32, int n, const float* ns3, const float* wo3,
                  const float* u2, const float* uc, const float* wi_eval3, float* prepared12,
                  float* sample10, float* eval4);
int dmt_test_light(dmt_ctx* ctx, const void* light32, int n, const float* pos3, const float* nrm3,
                   const float* u2, const int32_t* had_transmission, float* out14);
int dmt_test_half(dmt_ctx* ctx, int n, const float* f_in, uint16_t* h_out, const uint16_t* h_in,
                  float* f_out);
/* radiance of individual (pixel, sample) paths of the uploaded scene */
int dmt_test_trace_samples(dmt_ctx* ctx, int n, const int32_t* pxs, const int32_t* pys,
                           const int32_t* ss, float* L3);
/* per-bounce log of one path: records of 12 floats {tri, pos3, beta3, L3 before shading, depth,
 * sampler dimension}; *n_out = records written (<= cap) */
int dmt_test_trace_log(dmt_ctx* ctx, int px, int py, int s, float* rec12, int cap, int* n_out,
                       float* L3);
/* closest hit (triangle index or -1, t) of rays against the uploaded scene, current accel mode */
int dmt_test_closest_hit(dmt_ctx* ctx, int nrays, const float* o3, const float* d3,
                         int32_t* tri_index, float* t);

#ifdef __cplusplus
}
#endif
#endif /* DMT_HIP_H */
