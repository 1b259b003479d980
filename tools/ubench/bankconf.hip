// Microbenchmark: does the VGPR bank (register number mod 4) of the source operands change the issue cost of a VALU
// instruction on gfx950?  Every case is 8 independent chains written with FIXED physical registers, 4 waves per SIMD.
// chains: scalar v20..v27 (banks 0..3 twice), packed v[20:21] .. v[34:35] (even-aligned pairs: banks 0/1 or 2/3).
// Build: hipcc -O3 --offload-arch=gfx950 -o bankconf bankconf.hip ; run: ./bankconf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35", \
             "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","vcc","s10","s11","s12","s13"

// S8: one instruction per scalar chain; A = "rest of the operands"
#define S8(OP, A) OP " v20, v20, " A "\n" OP " v21, v21, " A "\n" OP " v22, v22, " A "\n" OP " v23, v23, " A "\n" \
                  OP " v24, v24, " A "\n" OP " v25, v25, " A "\n" OP " v26, v26, " A "\n" OP " v27, v27, " A "\n"
// same-bank chains: v20, v24, v28, v32 (bank 0) twice
#define S8B0(OP, A) OP " v20, v20, " A "\n" OP " v24, v24, " A "\n" OP " v28, v28, " A "\n" OP " v32, v32, " A "\n" \
                    OP " v20, v20, " A "\n" OP " v24, v24, " A "\n" OP " v28, v28, " A "\n" OP " v32, v32, " A "\n"
#define P8(OP, A) OP " v[20:21], v[20:21], " A "\n" OP " v[22:23], v[22:23], " A "\n" OP " v[24:25], v[24:25], " A "\n" \
                  OP " v[26:27], v[26:27], " A "\n" OP " v[28:29], v[28:29], " A "\n" OP " v[30:31], v[30:31], " A "\n" \
                  OP " v[32:33], v[32:33], " A "\n" OP " v[34:35], v[34:35], " A "\n"
// packed chains all on banks 0/1: v[20:21], v[24:25], v[28:29], v[32:33]
#define P8B0(OP, A) OP " v[20:21], v[20:21], " A "\n" OP " v[24:25], v[24:25], " A "\n" OP " v[28:29], v[28:29], " A "\n" \
                    OP " v[32:33], v[32:33], " A "\n" OP " v[20:21], v[20:21], " A "\n" OP " v[24:25], v[24:25], " A "\n" \
                    OP " v[28:29], v[28:29], " A "\n" OP " v[32:33], v[32:33], " A "\n"
#define C8(OP, A) OP " vcc, v20, " A "\n" OP " vcc, v21, " A "\n" OP " vcc, v22, " A "\n" OP " vcc, v23, " A "\n" \
                  OP " vcc, v24, " A "\n" OP " vcc, v25, " A "\n" OP " vcc, v26, " A "\n" OP " vcc, v27, " A "\n"

#define KERNEL(NAME, BODY)                                                                         \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, int iters) {                          \
    asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n" \
                 "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n" \
                 "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" \
                 "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n" \
                 "v_mov_b32 v40, 0.5\n v_mov_b32 v41, 0.5\n v_mov_b32 v42, 0.5\n v_mov_b32 v43, 0.5\n" \
                 "v_mov_b32 v44, 0.5\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n" \
                 "v_mov_b32 v48, 0.5\n v_mov_b32 v49, 0.5\n v_mov_b32 v50, 0.5\n v_mov_b32 v51, 0.5\n" \
                 "s_mov_b32 s10, 0.5\n s_mov_b32 s11, 0.5\n s_mov_b32 s12, 0.5\n s_mov_b32 s13, 0.5\n" ::: CLOB); \
    _Pragma("unroll 1") for (int i = 0; i < iters; ++i) {                                          \
      asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY ::: CLOB);                              \
    }                                                                                              \
    uint32_t r;                                                                                    \
    asm volatile("v_add_u32 %0, v20, v27\n v_add_u32 %0, %0, v34" : "=v"(r) :: CLOB);              \
    out[blockIdx.x * 256u + threadIdx.x] = r;                                                      \
  }

KERNEL(k_fma_rep, S8("v_fma_f32", "v40, v40"))
KERNEL(k_fma_same_bank, S8("v_fma_f32", "v40, v44"))
KERNEL(k_fma_diff_bank, S8("v_fma_f32", "v40, v41"))
KERNEL(k_fma_b0_c123, S8B0("v_fma_f32", "v41, v42"))
KERNEL(k_fma_b0_c000, S8B0("v_fma_f32", "v40, v44"))
KERNEL(k_fma_sgpr, S8("v_fma_f32", "s10, v41"))
KERNEL(k_fma_sgpr2, S8("v_fma_f32", "s10, s10"))
KERNEL(k_add_diff, S8("v_add_f32", "v41"))
KERNEL(k_add_b0_same, S8B0("v_add_f32", "v40"))
KERNEL(k_add_b0_diff, S8B0("v_add_f32", "v41"))
KERNEL(k_max_b0_same, S8B0("v_max_f32", "v40"))
KERNEL(k_max_b0_diff, S8B0("v_max_f32", "v41"))
KERNEL(k_pk_rep, P8("v_pk_fma_f32", "v[40:41], v[40:41]"))
KERNEL(k_pk_same_bank, P8("v_pk_fma_f32", "v[40:41], v[44:45]"))
KERNEL(k_pk_diff_bank, P8("v_pk_fma_f32", "v[40:41], v[42:43]"))
KERNEL(k_pk_b0_c23, P8B0("v_pk_fma_f32", "v[42:43], v[46:47]"))
KERNEL(k_pk_b0_c01_23, P8B0("v_pk_fma_f32", "v[40:41], v[42:43]"))
KERNEL(k_pk_sgpr, P8("v_pk_fma_f32", "s[10:11], v[42:43]"))
KERNEL(k_pk_sgpr_b0, P8B0("v_pk_fma_f32", "s[10:11], v[42:43]"))
KERNEL(k_pk_sgpr_sgpr, P8("v_pk_fma_f32", "s[10:11], s[10:11]"))
KERNEL(k_pk_mul_v, P8("v_pk_mul_f32", "v[42:43]"))
KERNEL(k_pk_mul_s, P8("v_pk_mul_f32", "s[10:11]"))
KERNEL(k_pk_mul_b0_v23, P8B0("v_pk_mul_f32", "v[42:43]"))
KERNEL(k_pk_add_b0_v23, P8B0("v_pk_add_f32", "v[42:43]"))
KERNEL(k_cmp_v, C8("v_cmp_lt_f32", "v41"))
KERNEL(k_cmp_s, C8("v_cmp_lt_f32", "s10"))
KERNEL(k_cmp_e64_s, "v_cmp_lt_f32_e64 s[12:13], v20, s10\n v_cmp_lt_f32_e64 s[12:13], v21, s10\n v_cmp_lt_f32_e64 s[12:13], v22, s10\n v_cmp_lt_f32_e64 s[12:13], v23, s10\n"
                    "v_cmp_lt_f32_e64 s[12:13], v24, s10\n v_cmp_lt_f32_e64 s[12:13], v25, s10\n v_cmp_lt_f32_e64 s[12:13], v26, s10\n v_cmp_lt_f32_e64 s[12:13], v27, s10\n")

template <class K>
static double run(K kernel, uint32_t* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 16);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
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
  double const instr = double(blocks) * 4 * iters * 64.0;
#define ROW(K, WHAT) { double const ms = run(K, out, blocks, iters); printf("  %-58s %7.3f ms  %5.2f cycles/instr\n", WHAT, ms, ms * 1e-3 * 2.4e9 / (instr / (cus * 4.0))); }
  printf("%d CUs x 16 waves, %d x 64 instructions per wave; cycles at 2.4 GHz per wave instruction per SIMD\n", cus, iters);
  ROW(k_fma_rep, "v_fma_f32 a, a, v40, v40 (register repeated)");
  ROW(k_fma_same_bank, "v_fma_f32 a, a, v40, v44 (sources in one bank)");
  ROW(k_fma_diff_bank, "v_fma_f32 a, a, v40, v41 (banks 0, 1; a in all banks)");
  ROW(k_fma_b0_c123, "v_fma_f32 a(bank 0), a, v41, v42 (three banks)");
  ROW(k_fma_b0_c000, "v_fma_f32 a(bank 0), a, v40, v44 (all bank 0)");
  ROW(k_fma_sgpr, "v_fma_f32 a, a, s10, v41");
  ROW(k_fma_sgpr2, "v_fma_f32 a, a, s10, s10");
  ROW(k_add_diff, "v_add_f32 a, a, v41");
  ROW(k_add_b0_same, "v_add_f32 a(bank 0), a, v40 (same bank)");
  ROW(k_add_b0_diff, "v_add_f32 a(bank 0), a, v41 (other bank)");
  ROW(k_max_b0_same, "v_max_f32 a(bank 0), a, v40 (same bank)");
  ROW(k_max_b0_diff, "v_max_f32 a(bank 0), a, v41 (other bank)");
  ROW(k_pk_rep, "v_pk_fma_f32 a, a, v[40:41], v[40:41] (repeated)");
  ROW(k_pk_same_bank, "v_pk_fma_f32 a, a, v[40:41], v[44:45] (same banks)");
  ROW(k_pk_diff_bank, "v_pk_fma_f32 a, a, v[40:41], v[42:43] (banks 01, 23)");
  ROW(k_pk_b0_c23, "v_pk_fma_f32 a(banks 01), a, v[42:43], v[46:47]");
  ROW(k_pk_b0_c01_23, "v_pk_fma_f32 a(banks 01), a, v[40:41], v[42:43]");
  ROW(k_pk_sgpr, "v_pk_fma_f32 a, a, s[10:11], v[42:43]");
  ROW(k_pk_sgpr_b0, "v_pk_fma_f32 a(banks 01), a, s[10:11], v[42:43]");
  ROW(k_pk_sgpr_sgpr, "v_pk_fma_f32 a, a, s[10:11], s[10:11]");
  ROW(k_pk_mul_v, "v_pk_mul_f32 a, a, v[42:43]");
  ROW(k_pk_mul_s, "v_pk_mul_f32 a, a, s[10:11]");
  ROW(k_pk_mul_b0_v23, "v_pk_mul_f32 a(banks 01), a, v[42:43]");
  ROW(k_pk_add_b0_v23, "v_pk_add_f32 a(banks 01), a, v[42:43]");
  ROW(k_cmp_v, "v_cmp_lt_f32 vcc, a, v41");
  ROW(k_cmp_s, "v_cmp_lt_f32 vcc, a, s10");
  ROW(k_cmp_e64_s, "v_cmp_lt_f32 s[12:13], a, s10");
  return 0;
}
