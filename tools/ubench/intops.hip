// Microbenchmark behind the sampler's integer code: issue cost of common VALU instructions on gfx950
// relative to a plain 32-bit add.  Each kernel runs 8 independent chains of one instruction per lane, 16 waves per CU resident,
// so the figure is issue throughput, not latency.  Build: hipcc -O3 --offload-arch=gfx950 -o intops intops.hip ; run: ./intops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define CHAIN8(INSTR)                                                                                     \
  _Pragma("unroll 1") for (int i = 0; i < iters; ++i) {                                                   \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                       \
      asm volatile(INSTR : "+v"(a0) : "v"(c)); asm volatile(INSTR : "+v"(a1) : "v"(c));                   \
      asm volatile(INSTR : "+v"(a2) : "v"(c)); asm volatile(INSTR : "+v"(a3) : "v"(c));                   \
      asm volatile(INSTR : "+v"(a4) : "v"(c)); asm volatile(INSTR : "+v"(a5) : "v"(c));                   \
      asm volatile(INSTR : "+v"(a6) : "v"(c)); asm volatile(INSTR : "+v"(a7) : "v"(c));                   \
    }                                                                                                     \
  }

#define KERNEL(NAME, INSTR)                                                                               \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters, uint32_t c) {                     \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    CHAIN8(INSTR)                                                                                         \
    out[blockIdx.x * 256u + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                         \
  }

KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %1")
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %1")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %1")
KERNEL(k_addf, "v_add_f32 %0, %0, %1")
KERNEL(k_mulf, "v_mul_f32 %0, %0, %1")
KERNEL(k_maxf, "v_max_f32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_add_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp_s_cnd, "v_cmp_lt_u32_e64 s[10:11], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL(k_fma3, "v_fma_f32 %0, %0, %1, v100")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL(k_subf, "v_sub_f32 %0, %0, %1")
KERNEL(k_min, "v_min_f32 %0, %0, %1")
KERNEL(k_lshr, "v_lshrrev_b32 %0, 1, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL(k_mac_mix, "v_add_f32 %0, %0, %1\n v_fma_f32 %0, %0, %1, %1")
// round 3: the instructions of the BVH node step (quantised-plane decode, slab min/max, sort keys) and their candidates
KERNEL(k_cvt_ub1, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %1")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %1")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]")
KERNEL(k_cvt_pk_fp8, "v_cvt_f32_fp8 %0, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")

#define KERNEL2(NAME, INSTR)                                                                              \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters, uint32_t c) {                     \
    typedef float v2 __attribute__((ext_vector_type(2)));                                                 \
    v2 a0 = {float(threadIdx.x), 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    v2 const cc = {float(c), 0.5f};                                                                       \
    _Pragma("unroll 1") for (int i = 0; i < iters; ++i) {                                                 \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                     \
        asm volatile(INSTR : "+v"(a0) : "v"(cc)); asm volatile(INSTR : "+v"(a1) : "v"(cc));               \
        asm volatile(INSTR : "+v"(a2) : "v"(cc)); asm volatile(INSTR : "+v"(a3) : "v"(cc));               \
        asm volatile(INSTR : "+v"(a4) : "v"(cc)); asm volatile(INSTR : "+v"(a5) : "v"(cc));               \
        asm volatile(INSTR : "+v"(a6) : "v"(cc)); asm volatile(INSTR : "+v"(a7) : "v"(cc));               \
      }                                                                                                   \
    }                                                                                                     \
    v2 const t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                   \
    out[blockIdx.x * 256u + threadIdx.x] = __float_as_uint(t.x + t.y);                                    \
  }
KERNEL2(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %1")
KERNEL2(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL2(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")

__global__ void __launch_bounds__(256) k_mad_u64(uint32_t* out, int iters, uint32_t c) {
  unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#define M(A) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(A) : "v"(c) : "vcc")
      M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
    }
  }
  out[blockIdx.x * 256u + threadIdx.x] = uint32_t(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}

template <class K>
static double run(K kernel, uint32_t* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 16, 3u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  int const cus = prop.multiProcessorCount, blocks = cus * 4, iters = 4096;
  uint32_t* out;
  CHECK(hipMalloc(&out, size_t(blocks) * 256 * 4));
  double const instr = double(blocks) * 4 /*waves*/ * iters * 64.0;  // wave instructions per kernel
  struct { char const* name; double ms; } rows[] = {
      {"v_add_u32", run(k_add, out, blocks, iters)},         {"v_add_u32_e64", run(k_add_e64, out, blocks, iters)},
      {"v_xor_b32", run(k_xor, out, blocks, iters)},         {"v_lshlrev_b32", run(k_lshl, out, blocks, iters)},
      {"v_mov_b32", run(k_mov, out, blocks, iters)},         {"v_add3_u32", run(k_add3, out, blocks, iters)},
      {"v_cndmask_b32", run(k_cndmask, out, blocks, iters)}, {"v_cmp_lt_f32", run(k_cmp, out, blocks, iters)},
      {"v_cndmask s[10:11]", run(k_cndmask_s, out, blocks, iters)}, {"cmp vcc + cndmask (2)", run(k_cmp_cnd, out, blocks, iters)},
      {"cmp s + cndmask (2)", run(k_cmp_s_cnd, out, blocks, iters)}, {"v_fma_f32 3 regs", run(k_fma3, out, blocks, iters)},
      {"v_and_b32", run(k_and, out, blocks, iters)}, {"v_sub_u32", run(k_sub, out, blocks, iters)},
      {"v_sub_f32", run(k_subf, out, blocks, iters)}, {"v_min_f32", run(k_min, out, blocks, iters)},
      {"v_lshrrev_b32", run(k_lshr, out, blocks, iters)}, {"v_bfe_u32", run(k_bfe, out, blocks, iters)},
      {"add_f32 + fma (2)", run(k_mac_mix, out, blocks, iters)},
      {"v_cvt_f32_ubyte1", run(k_cvt_ub1, out, blocks, iters)}, {"v_perm_b32", run(k_perm, out, blocks, iters)},
      {"v_min3_f32", run(k_min3, out, blocks, iters)}, {"v_max3_f32", run(k_max3, out, blocks, iters)},
      {"v_min_u32", run(k_min_u32, out, blocks, iters)}, {"v_and_or_b32", run(k_and_or, out, blocks, iters)},
      {"v_fma_mix_f32 (f16 src0)", run(k_fma_mix, out, blocks, iters)}, {"v_cvt_f32_fp8", run(k_cvt_pk_fp8, out, blocks, iters)},
      {"v_ldexp_f32", run(k_ldexp, out, blocks, iters)},
      {"v_add_f32", run(k_addf, out, blocks, iters)},        {"v_mul_f32", run(k_mulf, out, blocks, iters)},
      {"v_max_f32", run(k_maxf, out, blocks, iters)},        {"v_fmac_f32", run(k_fmac, out, blocks, iters)},
      {"v_fma_f32", run(k_fma, out, blocks, iters)},         {"v_pk_fma_f32", run(k_pk_fma, out, blocks, iters)},
      {"v_pk_add_f32", run(k_pk_add, out, blocks, iters)},   {"v_pk_mul_f32", run(k_pk_mul, out, blocks, iters)},
      {"v_cvt_f32_u32", run(k_cvt, out, blocks, iters)},     {"v_rcp_f32", run(k_rcp, out, blocks, iters)},
      {"v_lshl_add_u32", run(k_lshl_add, out, blocks, iters)}, {"v_xad_u32", run(k_xad, out, blocks, iters)},
      {"v_mul_u32_u24", run(k_mul_u24, out, blocks, iters)}, {"v_mad_u32_u24", run(k_mad_u24, out, blocks, iters)},
      {"v_mul_lo_u32", run(k_mul_lo, out, blocks, iters)},   {"v_mul_hi_u32", run(k_mul_hi, out, blocks, iters)},
      {"v_mad_u64_u32", run(k_mad_u64, out, blocks, iters)}};
  double const base = rows[0].ms;
  printf("%d CUs, %d blocks x 256 threads, %d x 64 instructions per wave\n", cus, blocks, iters);
  for (auto const& r : rows)
    printf("  %-16s %8.3f ms  %6.2f x v_add_u32   %7.2f cycles per wave instruction per SIMD at 2.4 GHz\n", r.name, r.ms, r.ms / base,
           r.ms * 1e-3 * 2.4e9 / (instr / (cus * 4.0)));
  return 0;
}
