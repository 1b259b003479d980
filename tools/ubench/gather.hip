// Microbenchmark behind the BVH node / leaf layout decisions (DESIGN.md "C4 kernel"): what does a per-lane dependent
// gather of a small record cost on gfx950, as a function of record size (number of 16-byte loads per lane), table size
// (L2-resident vs Infinity-Cache-resident vs HBM) and occupancy?  One lane = one "ray": it walks a pseudo-random chain of
// records (next index = hash of loaded data, so the loads are truly dependent like a traversal), STEPS steps.
//   variant L<k>: k dwordx4 loads from one (16*k')-byte record per lane (k' = record size / 16)
//   variant C   : 128-byte records fetched cooperatively: 8 lanes x 16 B per record (coalesced), staged through LDS
// Build: hipcc -O3 --offload-arch=gfx950 -o gather gather.hip ; run: ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}

// REC = record bytes (power of two >= 16), LOADS = 16-byte loads issued per step (<= REC/16)
template <int REC, int LOADS, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_lane(uint4 const* __restrict__ table, uint32_t mask, int steps, uint32_t* out) {
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    uint4 const* rec = table + size_t(idx) * (REC / 16);
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
      uint4 const v = rec[k];
      h += v.x ^ v.y ^ v.z ^ v.w;
    }
    // ~40 VALU ops of "work" so that the loop is not pure latency (a slab test is ~100)
#pragma unroll
    for (int j = 0; j < 8; ++j) h = mix(h + j);
    acc += h;
    idx = h & mask;
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// cooperative: per step, the wave fetches its 64 records (128 B each) with 8 wave-level loads in which 8 consecutive
// lanes read the 8 pieces of one record (coalesced), through LDS, then each lane reads its own record back.
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_coop(uint4 const* __restrict__ table, uint32_t mask, int steps, uint32_t* out) {
  __shared__ uint32_t s_idx[256];
  __shared__ uint4 s_rec[256 * 9];  // 144-byte stride per lane: conflict-free b128 reads
  uint32_t const lane = threadIdx.x & 63u, wbase = threadIdx.x & ~63u;
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    s_idx[threadIdx.x] = idx;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t const owner = j * 8u + (lane >> 3);
      uint32_t const oi = s_idx[wbase + owner];
      v[j] = table[size_t(oi) * 8 + (lane & 7u)];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t const owner = j * 8u + (lane >> 3);
      s_rec[(wbase + owner) * 9 + (lane & 7u)] = v[j];
    }
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      uint4 const r = s_rec[threadIdx.x * 9 + k];
      h += r.x ^ r.y ^ r.z ^ r.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) h = mix(h + j);
    acc += h;
    idx = h & mask;
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

static uint32_t pow2mask(size_t n) { uint32_t p = 1; while (size_t(p) * 2 <= n) p *= 2; return p - 1; }

template <class K>
double run(K kernel, int blocks, uint4* table, uint32_t mask, int steps, uint32_t* out) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, mask, steps / 4, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, mask, steps, out);
  CHECK(hipEventRecord(b));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  size_t const maxBytes = size_t(1) << 31;  // 2 GiB
  uint4* table; uint32_t* out;
  CHECK(hipMalloc(&table, maxBytes));
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * 4));
  std::vector<uint32_t> h(maxBytes / 4);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
  CHECK(hipMemcpy(table, h.data(), maxBytes, hipMemcpyHostToDevice));
  int const steps = 2000;
  printf("%-10s %-8s %-6s %10s %12s %12s %12s\n", "variant", "tableMB", "waves", "ms", "Gsteps/s", "ns/step/wave", "TB/s(rec)");
  for (size_t tableBytes : {size_t(1) << 21, size_t(1) << 25, size_t(1) << 27, size_t(1) << 31}) {
    auto report = [&](char const* name, int rec, int waves, double ms) {
      double const lanes = 256.0 * 4 * waves * 64;   // resident lanes (grid = one wave set)
      double const gsteps = lanes * steps / (ms * 1e-3) / 1e9;
      printf("%-10s %-8.0f %-6d %10.2f %12.2f %12.1f %12.2f\n", name, tableBytes / 1048576.0, waves, ms, gsteps,
             ms * 1e6 / steps, gsteps * rec / 1e3);
    };
#define LANE(REC, LOADS, WAVES, NAME) report(NAME, REC, WAVES, run(k_lane<REC, LOADS, WAVES>, 256 * WAVES, table, pow2mask(tableBytes / REC), steps, out))
    LANE(128, 7, 3, "128B/7ld");
    LANE(128, 7, 8, "128B/7ld");
    LANE(128, 5, 3, "128B/5ld");
    LANE(64, 4, 3, "64B/4ld");
    LANE(64, 4, 4, "64B/4ld");
    LANE(64, 4, 8, "64B/4ld");
    LANE(64, 3, 8, "64B/3ld");
    LANE(32, 2, 4, "32B/2ld");
    LANE(32, 2, 8, "32B/2ld");
    LANE(16, 1, 8, "16B/1ld");
    LANE(48, 3, 3, "48B/3ld");
    LANE(48, 3, 4, "48B/3ld");
    LANE(48, 3, 8, "48B/3ld");
    LANE(80, 5, 3, "80B/5ld");
    LANE(80, 5, 4, "80B/5ld");
    LANE(96, 5, 4, "96B/5ld");
    report("coop128", 128, 3, run(k_coop<3>, 256 * 3, table, uint32_t(tableBytes / 128 - 1), steps, out));
    report("coop128", 128, 4, run(k_coop<4>, 256 * 4, table, uint32_t(tableBytes / 128 - 1), steps, out));
  }
  return 0;
}
