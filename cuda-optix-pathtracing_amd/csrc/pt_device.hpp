// pt_device.hpp -- device-side rendering math of the gfx950 path tracer.
//
// Written for CDNA4: one lane = one path, all per-path state in VGPRs, scene constants in SGPRs /
// the scalar cache, no shared mutable sampler state.  The arithmetic follows the reference's
// megakernel path (file:line citations relative to /root/reference/examples/triangles/;
// CC/ = cuda-core/) so that images agree with it to within float rounding:
//   * host-branch semantics wherever the reference has host/device branches (software fp16
//     round-half-up, table lookup with (size-1) scaling) -- see DESIGN.md "Numerics";
//   * documented quirks of the reference are kept (see each function).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmt_ggx_tables.inc"

#define DMT_DEV __device__ __forceinline__

namespace dmt {

// ---------------------------------------------------------------------------------------------
// vectors
// ---------------------------------------------------------------------------------------------
struct f2 {
  float x, y;
};
struct f3 {
  float x, y, z;
};
DMT_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DMT_DEV f2 mk2(float x, float y) { return f2{x, y}; }
DMT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
DMT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
DMT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
DMT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
DMT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
DMT_DEV f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
// Vector divisions go through v_rcp_f32 (1 ulp) and a multiply.  The compiler's own "fast" fp32 division (the
// build runs with -fno-hip-fp32-correctly-rounded-divide-sqrt) wraps the same v_rcp in a range
// rescue for |divisor| > 2^96 that costs four more instructions per component and never triggers here.
DMT_DEV f3 operator/(f3 a, float s) {
  float const inv = __builtin_amdgcn_rcpf(s);
  return mk3(a.x * inv, a.y * inv, a.z * inv);
}
DMT_DEV f3 operator/(f3 a, f3 b) {
  return mk3(a.x * __builtin_amdgcn_rcpf(b.x), a.y * __builtin_amdgcn_rcpf(b.y), a.z * __builtin_amdgcn_rcpf(b.z));
}
DMT_DEV f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
DMT_DEV f3 operator+(float s, f3 a) { return mk3(a.x + s, a.y + s, a.z + s); }
DMT_DEV f3 operator-(f3 a, float s) { return mk3(a.x - s, a.y - s, a.z - s); }
DMT_DEV f3 operator-(float s, f3 a) { return mk3(s - a.x, s - a.y, s - a.z); }
DMT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DMT_DEV float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
DMT_DEV f3 cross(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
DMT_DEV float rsqrt_ieee(float x) { return __builtin_amdgcn_rsqf(x); }  // the reference's rsqrtf; v_rsq_f32, 1 ulp
DMT_DEV f3 normalize(f3 a) {                                   // CC common_math.cuh:297-300
  float const inv = rsqrt_ieee(a.x * a.x + a.y * a.y + a.z * a.z);
  return mk3(a.x * inv, a.y * inv, a.z * inv);
}
DMT_DEV f3 abs3(f3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
DMT_DEV f3 sqrt3(f3 a) { return mk3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
DMT_DEV float sqr(float x) { return x * x; }
DMT_DEV float safe_sqrt(float x) { return sqrtf(fmaxf(x, 0.f)); }
DMT_DEV float max3(f3 v) { return fmaxf(v.x, fmaxf(v.y, v.z)); }
DMT_DEV bool is_zero(f3 v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; }
DMT_DEV bool near_zero_pos(f3 v, float tol) { return v.x < tol && v.y < tol && v.z < tol; }
DMT_DEV float luminance(f3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }
DMT_DEV float average(f3 a) { return (a.x + a.y + a.z) / 3.f; }
DMT_DEV float lerp1(float a, float b, float t) {
  float const omt = 1.f - t;
  return omt * a + t * b;
}
DMT_DEV f3 lerp3(f3 a, f3 b, float t) {
  float const omt = 1.f - t;
  return omt * a + t * b;
}
DMT_DEV float safe_acos(float v) { return acosf(fminf(fmaxf(v, -1.f), 1.f)); }
DMT_DEV float sin_from_cos(float c) { return safe_sqrt(1.f - sqr(c)); }
DMT_DEV float sin_sqr_to_one_minus_cos(float s) {  // common_math.cuh:439-443
  return s > 0.0004f ? 1.0f - safe_sqrt(1.0f - s) : 0.5f * s;
}
// common_math.cuh:453-465
DMT_DEV void gram_schmidt(f3 n, f3& a, f3& b) {
  if (fabsf(n.x - n.y) > 1e-3f || fabsf(n.x - n.z) > 1e-3f)
    a = mk3(n.z - n.y, n.x - n.z, n.y - n.x);
  else
    a = mk3(n.z - n.y, n.x + n.z, -n.y - n.x);
  a = normalize(a);
  b = cross(n, a);
}
DMT_DEV void orthonormal_tangent(f3 n, f3 t, f3& a, f3& b) {  // :466-472
  b = normalize(cross(n, t));
  a = cross(b, n);
}

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.318309886183790671538f;
constexpr float kInf = __builtin_huge_valf();

// ---------------------------------------------------------------------------------------------
// fp16 storage codec.  The reference quantises BSDF terms to fp16 every bounce
// (CC/private/bsdf.cu:424,948-950,963).  Its host branch rounds half UP, not to nearest even
// (CC/private/encoding.cu:93-115); that branch is the one the oracle follows, so it is
// reproduced bit for bit here with integer ops instead of v_cvt_f16_f32.
// ---------------------------------------------------------------------------------------------
DMT_DEV uint32_t f2h(float f) {
  uint32_t const u = __float_as_uint(f);
  uint32_t const sign = (u >> 16) & 0x8000u;
  int32_t exp = int32_t((u >> 23) & 0xFFu) - 112;
  uint32_t mant = u & 0x007FFFFFu;
  if (exp >= 31) return sign | 0x7C00u | (mant ? 0x200u : 0u);
  if (exp <= 0) {
    if (exp < -10) return sign;
    mant |= 0x00800000u;
    uint32_t const shift = uint32_t(14 - exp);
    return sign | (((mant >> shift) + ((mant >> (shift - 1)) & 1u)) & 0xFFFFu);
  }
  uint32_t rounded = mant + 0x00001000u;
  if (rounded & 0x00800000u) {
    rounded = 0;
    exp += 1;
    if (exp >= 31) return sign | 0x7C00u;
  }
  return sign | (uint32_t(exp) << 10) | (rounded >> 13);
}
DMT_DEV float h2f(uint32_t h) {  // exact; v_cvt_f32_f16 with fp16 denormals enabled
  _Float16 hv = __builtin_bit_cast(_Float16, (unsigned short)(h & 0xFFFFu));
  return float(hv);
}
// fp32 -> fp16 -> fp32.  For magnitudes in fp16's normal range [2^-14, 65520) the codec's round-half-up is
// "add half of the 13 dropped mantissa bits, truncate" on the fp32 encoding (a mantissa carry runs into the
// exponent by itself): 3 integer ops instead of ~55.  Everything else takes the full codec.
DMT_DEV float q16(float f) {
  uint32_t const u = __float_as_uint(f), a = u & 0x7FFFFFFFu;
  if (a >= 0x38800000u && a < 0x477FF000u) return __uint_as_float((u + 0x1000u) & 0xFFFFE000u);
  return h2f(f2h(f));
}
DMT_DEV f3 q16(f3 v) { return mk3(q16(v.x), q16(v.y), q16(v.z)); }

// octahedral decode                                             CC/private/encoding.cu:39-60
DMT_DEV float sign_pm1(float f) { return __builtin_signbit(f) ? -1.f : 1.f; }
DMT_DEV f3 dir_from_octa(uint32_t octa) {
  float const mx = 65535.f;
  float const fx = float(octa & 0xFFFFu) / mx * 2.f - 1.f;
  float const fy = float((octa >> 16) & 0xFFFFu) / mx * 2.f - 1.f;
  f3 n = mk3(fx, fy, 1.f - fabsf(fx) - fabsf(fy));
  float const nx = n.x, ny = n.y;
  // flip * a + !flip * b with flip in {0,1}: the dropped product is an exact +-0 addend
  if (n.z < 0.f) {
    n.x = 1.f * (1.f - fabsf(ny)) * sign_pm1(nx) + 0.f * nx;
    n.y = 1.f * (1.f - fabsf(nx)) * sign_pm1(ny) + 0.f * ny;
  } else {
    n.x = 0.f * (1.f - fabsf(ny)) * sign_pm1(nx) + 1.f * nx;
    n.y = 0.f * (1.f - fabsf(nx)) * sign_pm1(ny) + 1.f * ny;
  }
  return normalize(n);
}

// ---------------------------------------------------------------------------------------------
// scene records
// ---------------------------------------------------------------------------------------------
struct Rec32 {  // packed BSDF / Light record, 8 dwords
  uint32_t w[8];
};
DMT_DEV uint32_t lo16(uint32_t w) { return w & 0xFFFFu; }
DMT_DEV uint32_t hi16(uint32_t w) { return w >> 16; }

// hot-loop triangle record, built on the host at upload: p0, e0 = p1-p0, e1 = p2-p0 (same float
// subtractions the reference does per test, CC/private/shapes.cu:10-11)
struct TriIsect {  // 48 B, three 16-byte loads
  float p0x, p0y, p0z, e0x;
  float e0y, e0z, e1x, e1y;
  float e1z;
  uint32_t matId;
  uint32_t pad0, pad1;
};
// post-hit record: original vertices (error bound needs them) + unit geometric normal
// normalize(cross(e1,e0)) precomputed on the host with the same IEEE expression (shapes.cu:48)
struct TriPost {  // 64 B
  float p0x, p0y, p0z, p1x;
  float p1y, p1z, p2x, p2y;
  float p2z, nx, ny, nz;
  uint32_t matId, pad0, pad1, pad2;
};

struct CameraXf {  // the matrix entries the perspective path needs (column-major m[16])
  float cfr[16];   // cameraFromRaster
  float rfc[16];   // renderFromCamera
};

struct SamplerParams {  // CC types.cuh:93-97
  int32_t scale0, scale1;
  int32_t exp0, exp1;
  int32_t inv0, inv1;
};

// Pointer in the constant address space: a wave-uniform index then becomes an s_load (scalar cache,
// result in SGPRs) no matter what the kernel stores elsewhere; a divergent index still becomes an
// ordinary global_load.  The scene arrays are never written by a kernel.
#define DMT_CONST_AS __attribute__((address_space(4)))
template <class T>
__host__ __device__ inline T const DMT_CONST_AS* to_const_as(T const* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (T const DMT_CONST_AS*)p;
#pragma clang diagnostic pop
}

struct SceneView {
  TriIsect const* __restrict__ tris;
  TriPost const* __restrict__ post;
  Rec32 const* __restrict__ bsdfs;
  Rec32 const* __restrict__ lights;
  Rec32 const* __restrict__ infLights;
  uint32_t triCount;
  uint32_t bsdfCount;
  uint32_t lightCount;
  uint32_t infLightCount;
};

// ---------------------------------------------------------------------------------------------
// Halton-Owen sampler                                             CC/private/rng.cu:48-262
// Every number is a pure function of (pixel, sample, dimension); dimensions cycle through 2..9
// (rng.cu:236,246), so a path can only ever see 8 distinct values.  They are computed once per
// sample with compile-time bases (constant division -> mul_hi), and later draws are register
// selects.
// ---------------------------------------------------------------------------------------------
DMT_DEV uint32_t mix_bits32(uint32_t v) {  // rng.cu:61-68
  v ^= v >> 16;
  v *= 0x7feb352dU;
  v ^= v >> 15;
  v *= 0x846ca68bU;
  v ^= v >> 16;
  return v;
}
// x - q * BASE for the TRUE quotient q = x / BASE.  The result is a digit (< 32), so only its low bits are needed and one
// multiply-add supplies them whatever the compiler makes of the constant division: (q mod 2^24) * (2^24 - BASE) + x has the
// low 24 bits of x - q * BASE.  (Integer multiplies issue at the full VALU rate on gfx950 -- tools/ubench/intops.hip -- so
// the sampler's cost is its instruction COUNT; this form is one instruction where the compiler otherwise emits two or three.)
template <uint32_t BASE>
DMT_DEV uint32_t digit_of(uint32_t x, uint32_t q) {
  return ((q & 0xFFFFFFu) * (0x1000000u - BASE) + x) & 0xFFu;
}
// Permutation tables for the low digits of every dimension.  The Owen scramble of a digit hashes the digits BELOW it, so
// at digit position k the hash input takes BASE^k values: for the positions where that is small the table holds
// mix_bits32(seed ^ prefix) % BASE for each of them, and the permuted digit (digit + scramble) % BASE becomes an add and
// a conditional subtract instead of the two-multiply hash and a division.  Position 0 needs no table (its scramble is
// a constant of the dimension); positions 1 .. kOwenTablePositions[d] - 1 are tabulated: 12.4 KB for the eight
// dimensions, 28 of the ~60 digits of a sample.  Exact only while digit + scramble does not wrap at 2^32, which `exact`
// checks for every entry at compile time.
constexpr uint32_t mix_bits32_c(uint32_t v) {
  v ^= v >> 16;
  v *= 0x7feb352dU;
  v ^= v >> 15;
  v *= 0x846ca68bU;
  v ^= v >> 16;
  return v;
}
constexpr uint32_t kOwenBases[8] = {5, 7, 11, 13, 17, 19, 23, 29};  // dimensions 2..9 (rng.cu:175-178 via sampler_values)
constexpr int kOwenTablePositions[8] = {6, 5, 4, 4, 3, 3, 3, 3};    // digit positions below this come from the table
constexpr uint32_t owen_seed(int dim) { return mix_bits32_c(1u + (uint32_t(dim) << 4)); }
constexpr uint32_t owen_pow(uint32_t base, int k) {
  uint32_t p = 1;
  for (int i = 0; i < k; ++i) p *= base;
  return p;
}
constexpr uint32_t owen_dim_bytes(int dim) {  // BASE + BASE^2 + ... + BASE^(P-1)
  uint32_t n = 0;
  for (int k = 1; k < kOwenTablePositions[dim - 2]; ++k) n += owen_pow(kOwenBases[dim - 2], k);
  return n;
}
constexpr uint32_t owen_table_offset(int dim, int pos) {  // first entry of digit position `pos` (>= 1) of dimension `dim`
  uint32_t at = 0;
  for (int d = 2; d < dim; ++d) at += owen_dim_bytes(d);
  for (int k = 1; k < pos; ++k) at += owen_pow(kOwenBases[dim - 2], k);
  return at;
}
constexpr uint32_t kOwenTableBytes = owen_table_offset(10, 1);
struct OwenTables {
  uint8_t t[kOwenTableBytes];
  bool exact;
};
constexpr OwenTables make_owen_tables() {
  OwenTables T{};
  T.exact = true;
  uint32_t at = 0;
  for (int dim = 2; dim < 10; ++dim) {
    uint32_t const base = kOwenBases[dim - 2], seed = owen_seed(dim);
    if (mix_bits32_c(seed) > 0xFFFFFFFFu - (base - 1u)) T.exact = false;  // position 0 (a constant, see sample_dim)
    for (int pos = 1; pos < kOwenTablePositions[dim - 2]; ++pos) {
      uint32_t const n = owen_pow(base, pos);
      for (uint32_t prefix = 0; prefix < n; ++prefix) {
        uint32_t const sc = mix_bits32_c(seed ^ prefix);
        if (sc > 0xFFFFFFFFu - (base - 1u)) T.exact = false;
        T.t[at++] = uint8_t(sc % base);
      }
    }
  }
  if (at != kOwenTableBytes) T.exact = false;
  return T;
}
static_assert(kOwenTableBytes == 12655, "table size");
static_assert(make_owen_tables().exact, "a scramble within BASE of 2^32: that prefix needs the wrapping sum");
__constant__ OwenTables c_owen = make_owen_tables();

template <uint32_t BASE>
DMT_DEV uint32_t add_mod(uint32_t digit, uint32_t scrambleModBase) {  // (digit + s) % BASE for digit, s < BASE
  uint32_t const t = digit + scrambleModBase;
  uint32_t const w = t - BASE;  // wraps to a huge value when t < BASE
  return t < w ? t : w;
}
constexpr float owen_inv_pow(uint32_t base, int k) {  // invBase multiplied up k times in float, as the reference's loop does
  float const invBase = 1.0f / float(base);
  float p = invBase;
  for (int i = 1; i < k; ++i) p *= invBase;
  return p;
}
template <int DIM, uint32_t BASE>
DMT_DEV float sample_dim(uint32_t index) {  // owenScrambledRadicalInverse, rng.cu:137-178
  static_assert(kOwenBases[DIM - 2] == BASE, "dimension / base");
  constexpr uint32_t seed = owen_seed(DIM);
  constexpr int P = kOwenTablePositions[DIM - 2];
  float const invBase = 1.0f / float(BASE);  // correctly rounded at compile time (= __frcp_rn)
  float result = 0.0f;
  if (index > 0) {  // digit 0: empty prefix, the scramble is a constant
    uint32_t next = index / BASE;
    uint32_t revHash = digit_of<BASE>(index, next);
    result = float(add_mod<BASE>(revHash, mix_bits32_c(seed) % BASE)) * invBase;  // fmaf(p, invBase, 0) == p * invBase
    index = next;
#pragma unroll
    for (int pos = 1; pos < P; ++pos) {  // tabulated positions: prefix = the digits so far
      if (index == 0) break;
      next = index / BASE;
      uint32_t const digit = digit_of<BASE>(index, next);
      uint32_t const permuted = add_mod<BASE>(digit, c_owen.t[owen_table_offset(DIM, pos) + revHash]);
      result = __builtin_fmaf(float(permuted), owen_inv_pow(BASE, pos + 1), result);
      revHash = revHash * BASE + digit;
      index = next;
    }
    float invBasePow = owen_inv_pow(BASE, P + 1);
    while (index > 0) {  // the rest as the reference writes it
      next = index / BASE;
      uint32_t const digit = digit_of<BASE>(index, next);
      uint32_t const scramble = mix_bits32(seed ^ revHash);
      uint32_t const sum = digit + scramble;  // 32-bit wraparound, as the reference
      uint32_t const permuted = digit_of<BASE>(sum, sum / BASE);
      result = __builtin_fmaf(float(permuted), invBasePow, result);
      revHash = revHash * BASE + digit;
      invBasePow *= invBase;
      index = next;
    }
  }
  return fminf(result, 0.99999994f);
}
template <uint32_t BASE>
DMT_DEV float radical_inverse(uint32_t index) {  // rng.cu:70-94
  float const invBase = 1.0f / float(BASE);
  float result = 0.0f;
  float invBasePow = invBase;
  while (index > 0) {
    uint32_t const next = index / BASE;
    uint32_t const digit = index - next * BASE;
    result = __builtin_fmaf(float(digit), invBasePow, result);
    invBasePow *= invBase;
    index = next;
  }
  return fminf(result, 0.99999994f);
}

// Halton index of sample 0 of a pixel (rng.cu:216-228); add s * stride for sample s
DMT_DEV int32_t halton_pixel_base(SamplerParams const& p, int px, int py) {
  int const stride = p.scale0 * p.scale1;
  int const pmx = px % 128, pmy = py % 128;
  // inverseRadicalInverse(pm, base, nDigits), rng.cu:48-59
  int32_t inv0 = 0, t = pmx;
  for (int i = 0; i < p.exp0; ++i) {
    inv0 = inv0 * 2 + (t % 2);
    t /= 2;
  }
  int32_t inv1 = 0;
  t = pmy;
  for (int i = 0; i < p.exp1; ++i) {
    inv1 = inv1 * 3 + (t % 3);
    t /= 3;
  }
  int32_t idx = 0;
  idx += inv0 * (stride / p.scale0) * p.inv0;
  idx += inv1 * (stride / p.scale1) * p.inv1;
  idx %= stride;
  return idx;
}

// The 8 values of a sample live in LDS as [dimension][thread] (bank-conflict free: consecutive lanes
// hit consecutive banks), one 4-byte read per draw instead of 8 VGPRs + a select chain per draw.
// Every thread only ever touches its own column, so no barrier is needed.
constexpr int kLdsThreads = 256;  // threads per block of every kernel that traces paths
__shared__ float s_sampler_u[8 * kLdsThreads];

// the 8 values of Halton index `haltonIndex` into column u[k * kLdsThreads], k = 0..7
DMT_DEV void sampler_values(uint32_t haltonIndex, float* u) {
  u[0 * kLdsThreads] = sample_dim<2, 5>(haltonIndex);
  u[1 * kLdsThreads] = sample_dim<3, 7>(haltonIndex);
  u[2 * kLdsThreads] = sample_dim<4, 11>(haltonIndex);
  u[3 * kLdsThreads] = sample_dim<5, 13>(haltonIndex);
  u[4 * kLdsThreads] = sample_dim<6, 17>(haltonIndex);
  u[5 * kLdsThreads] = sample_dim<7, 19>(haltonIndex);
  u[6 * kLdsThreads] = sample_dim<8, 23>(haltonIndex);
  u[7 * kLdsThreads] = sample_dim<9, 29>(haltonIndex);
}

struct Sampler {
  int dim;

  DMT_DEV void start(uint32_t haltonIndex) {
    float* const u = s_sampler_u + threadIdx.x;
    u[0 * kLdsThreads] = sample_dim<2, 5>(haltonIndex);
    u[1 * kLdsThreads] = sample_dim<3, 7>(haltonIndex);
    u[2 * kLdsThreads] = sample_dim<4, 11>(haltonIndex);
    u[3 * kLdsThreads] = sample_dim<5, 13>(haltonIndex);
    u[4 * kLdsThreads] = sample_dim<6, 17>(haltonIndex);
    u[5 * kLdsThreads] = sample_dim<7, 19>(haltonIndex);
    u[6 * kLdsThreads] = sample_dim<8, 23>(haltonIndex);
    u[7 * kLdsThreads] = sample_dim<9, 29>(haltonIndex);
    dim = 2;
  }
  DMT_DEV float pick(int d) const {  // d in 2..9
    return s_sampler_u[(d - 2) * kLdsThreads + int(threadIdx.x)];
  }
  DMT_DEV float get1D() {  // rng.cu:233-240
    if (dim >= 10) dim = 2;
    int const d = dim++;
    return pick(d);
  }
  DMT_DEV f2 get2D() {  // rng.cu:242-252
    if (dim + 1 >= 10) dim = 2;
    int const d = dim;
    dim += 2;
    return mk2(pick(d), pick(d + 1));
  }
};
// getPixel2D, rng.cu:254-262.  Base 2: every partial sum of distinct powers of two with <= 24
// significant bits is exact, so the fmaf chain equals bit reversal.
DMT_DEV f2 pixel2d(SamplerParams const& p, int32_t haltonIndex) {
  uint32_t const a = uint32_t(haltonIndex >> p.exp0);
  float const rx = fminf(float(__brev(a)) * 2.3283064365386963e-10f, 0.99999994f);
  float const ry = radical_inverse<3>(uint32_t(haltonIndex / p.scale1));
  return mk2(rx, ry);
}

// ---------------------------------------------------------------------------------------------
// camera                                   CC/private/extra_math.cu:7-42, common_math.cu:80-102
// ---------------------------------------------------------------------------------------------
struct Ray {
  f3 o, d;
};
DMT_DEV f3 xf_point(float const* m, f3 p) {
  float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
  float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
  float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
  float const w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
  if (w != 1.0f && w != 0.0f) {
    float const iw = 1.0f / w;
    x *= iw, y *= iw, z *= iw;
  }
  return mk3(x, y, z);
}
DMT_DEV f3 xf_dir(float const* m, f3 v) {
  return mk3(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
             m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
DMT_DEV Ray camera_ray(CameraXf const& cam, SamplerParams const& sp, int px, int py,
                       int32_t haltonIndex) {
  f2 const r = pixel2d(sp, haltonIndex);
  // (getPixel2D - 0.5) + 0.5 + pixel, left to right (extra_math.cu:10-12)
  float const fx = ((r.x - 0.5f) + 0.5f) + float(px);
  float const fy = ((r.y - 0.5f) + 0.5f) + float(py);
  f3 const pCamera = xf_point(cam.cfr, mk3(fx, fy, 0.0f));
  Ray ray;
  ray.o = xf_point(cam.rfc, mk3(0.f, 0.f, 0.f));
  ray.d = normalize(xf_dir(cam.rfc, pCamera));
  return ray;
}

// ---------------------------------------------------------------------------------------------
// Moeller-Trumbore, two-sided                                    CC/private/shapes.cu:5-57
// Returns validity and (t,u,v); position / normal / error bound are rebuilt once for the
// winning triangle only (hit_finish) instead of per candidate.
// ---------------------------------------------------------------------------------------------
struct MTResult {
  bool valid;
  float t, u, v;
};
// Two rays per lane in packed registers: component .x belongs to the lane's closest-hit ray,
// .y to its pending shadow ray.  One v_pk_* instruction then advances both Moeller-Trumbore chains
// (this is how CDNA reaches its fp32 peak), with the triangle in SGPRs broadcast to both halves.
typedef float v2f __attribute__((ext_vector_type(2)));
struct RayPair {
  v2f ox, oy, oz, dx, dy, dz;
};
struct TriS {  // one triangle held in scalars
  float p0x, p0y, p0z, e0x, e0y, e0z, e1x, e1y, e1z;
};
struct MTPair {
  v2f det, t, u, v;
};
// Moeller-Trumbore with every multiply-add spelled out, so that the packed two-ray form (brute-force
// loop) and the scalar form (BVH leaves, test kernels) round identically: a triangle gives bit-equal
// (det,t,u,v) whichever way it is reached.  Operation order = CC/private/shapes.cu:10-33 with
// a*b+c -> fma(a,b,c) and the reciprocal by v_rcp_f32 (the reference's GPU build is -use_fast_math).
DMT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DMT_DEV v2f fma_(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
DMT_DEV v2f fma_(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, v2f{b, b}, c); }
DMT_DEV v2f fma_(float a, v2f b, v2f c) { return __builtin_elementwise_fma(v2f{a, a}, b, c); }
DMT_DEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
DMT_DEV v2f rcp_(v2f x) { return v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
// (the triangle is passed as nine scalars on purpose: handed over as a struct, LLVM keeps part of it in
// a stack slot and "shuffles" its fields through scratch memory with overlapping <2 x float> accesses)
template <class T>
DMT_DEV void mt_core9(float p0x, float p0y, float p0z, float e0x, float e0y, float e0z, float e1x, float e1y,
                      float e1z, T ox, T oy, T oz, T dx, T dy, T dz, T& det, T& tt, T& u, T& v) {
  T const cx = fma_(dy, e1z, -(dz * e1y));
  T const cy = fma_(dz, e1x, -(dx * e1z));
  T const cz = fma_(dx, e1y, -(dy * e1x));
  det = fma_(cz, e0z, fma_(cy, e0y, cx * e0x));
  T const inv = rcp_(det);
  T const ovx = ox - p0x, ovy = oy - p0y, ovz = oz - p0z;
  T const qx = fma_(ovy, e0z, -(ovz * e0y));
  T const qy = fma_(ovz, e0x, -(ovx * e0z));
  T const qz = fma_(ovx, e0y, -(ovy * e0x));
  u = inv * fma_(cz, ovz, fma_(cy, ovy, cx * ovx));
  v = inv * fma_(qz, dz, fma_(qy, dy, qx * dx));
  tt = inv * fma_(qz, e1z, fma_(qy, e1y, qx * e1x));
}
// One ray against TWO triangles held as packed pairs (.x / .y = first / second triangle): the same operations
// in the same order as mt_core9, element by element, so each half is bit-identical to the scalar test.
DMT_DEV void mt_core9_tri2(v2f p0x, v2f p0y, v2f p0z, v2f e0x, v2f e0y, v2f e0z, v2f e1x, v2f e1y, v2f e1z, float ox,
                           float oy, float oz, float dx, float dy, float dz, v2f& det, v2f& tt, v2f& u, v2f& v) {
  v2f const cx = fma_(dy, e1z, -(dz * e1y));
  v2f const cy = fma_(dz, e1x, -(dx * e1z));
  v2f const cz = fma_(dx, e1y, -(dy * e1x));
  det = fma_(cz, e0z, fma_(cy, e0y, cx * e0x));
  v2f const inv = rcp_(det);
  v2f const ovx = ox - p0x, ovy = oy - p0y, ovz = oz - p0z;
  v2f const qx = fma_(ovy, e0z, -(ovz * e0y));
  v2f const qy = fma_(ovz, e0x, -(ovx * e0z));
  v2f const qz = fma_(ovx, e0y, -(ovy * e0x));
  u = inv * fma_(cz, ovz, fma_(cy, ovy, cx * ovx));
  v = inv * fma_(qz, dz, fma_(qy, dy, qx * dx));
  tt = inv * fma_(qz, e1z, fma_(qy, e1y, qx * e1x));
}
template <class T>
DMT_DEV void mt_core(TriS const& t, T ox, T oy, T oz, T dx, T dy, T dz, T& det, T& tt, T& u, T& v) {
  mt_core9<T>(t.p0x, t.p0y, t.p0z, t.e0x, t.e0y, t.e0z, t.e1x, t.e1y, t.e1z, ox, oy, oz, dx, dy, dz, det, tt, u, v);
}
DMT_DEV MTPair mt_pair(TriS const& T, RayPair const& r) {
  MTPair m;
  mt_core<v2f>(T, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, m.det, m.t, m.u, m.v);
  return m;
}
DMT_DEV bool mt_valid(float det, float t, float u, float v) {  // shapes.cu:19,35-38
  float const tol = 1e-7f;
  return !(fabsf(det) < tol) && (u >= -tol && v >= -tol && (u + v) <= 1 + tol) && (t > 1e-4f);
}

DMT_DEV MTResult mt_test(f3 p0, f3 e0, f3 e1, Ray const& ray) {
  TriS T;
  T.p0x = p0.x, T.p0y = p0.y, T.p0z = p0.z, T.e0x = e0.x, T.e0y = e0.y, T.e0z = e0.z;
  T.e1x = e1.x, T.e1y = e1.y, T.e1z = e1.z;
  float det;
  MTResult r;
  mt_core<float>(T, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, det, r.t, r.u, r.v);
  r.valid = mt_valid(det, r.t, r.u, r.v);
  return r;
}
DMT_DEV MTResult mt_test(TriIsect const& T, Ray const& ray) {
  return mt_test(mk3(T.p0x, T.p0y, T.p0z), mk3(T.e0x, T.e0y, T.e0z), mk3(T.e1x, T.e1y, T.e1z), ray);
}

struct Hit {
  f3 pos, normal, error;
  uint32_t matId;
};
DMT_DEV float gamma7() {  // CC extra_math.cuh:21-29, gamma(7)
  float const f = 7.f * 1.1920928955078125e-07f * 0.5f;
  return f / (1 - f);
}
DMT_DEV Hit hit_finish(TriPost const& P, float u, float v, f3 rayDir) {
  f3 const p0 = mk3(P.p0x, P.p0y, P.p0z);
  f3 const p1 = mk3(P.p1x, P.p1y, P.p1z);
  f3 const p2 = mk3(P.p2x, P.p2y, P.p2z);
  f3 const e0 = mk3(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
  f3 const e1 = mk3(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z);
  Hit h;
  h.pos = p0 + u * e0 + v * e1;                                                // shapes.cu:47
  h.error = gamma7() * (abs3(u * p0) + abs3(v * p1) + abs3((1 - u - v) * p2));  // :53
  f3 n = mk3(P.nx, P.ny, P.nz);
  if (dot(rayDir, n) > 0) n = -n;  // T/megakernel/megakernel.cu:129-131
  h.normal = n;
  h.matId = P.matId;
  return h;
}
// nextafterf(x, up ? +inf : -inf) for finite x, as integer arithmetic on the encoding (the library routine handles
// NaNs, infinities and a general target: 19 instructions per call, six calls per bounce)
DMT_DEV float next_float(float x, bool up) {
  uint32_t const b = __float_as_uint(x);
  if ((b & 0x7FFFFFFFu) == 0u) return __uint_as_float(up ? 0x00000001u : 0x80000001u);
  bool const positive = (b >> 31) == 0u;
  return __uint_as_float(positive == up ? b + 1u : b - 1u);
}
// CC extra_math.cuh:36-59
DMT_DEV f3 offset_ray_origin(f3 p, f3 error, f3 ng, f3 w) {
  float const d = dot(abs3(ng), error);
  f3 offset = ng * d;
  if (dot(w, ng) < 0.f) offset = -offset;
  f3 po = p + offset;
  po.x = next_float(po.x, offset.x > 0);
  po.y = next_float(po.y, offset.y > 0);
  po.z = next_float(po.z, offset.z > 0);
  return po;
}

// sin and cos of a BOUNDED angle (|x| < ~1e3; every angle on this path is within [-2 pi, 2 pi]): three-constant
// Cody-Waite reduction to [-pi/4, pi/4] and the classic single-precision minimax polynomials, ~1 ulp, i.e. the
// same accuracy class as libm's sinf/cosf on both sides of the parity tests.  The library functions carry a
// Payne-Hanek path for huge arguments that is never taken here but was a THIRD of the megakernel's code
// (3 300 of 9 900 instructions for three call sites): the kernel no longer fit the instruction cache it shares
// with the neighbouring CU.
DMT_DEV void sincos_bounded(float x, float& sn, float& cs) {
  float const q = rintf(x * 0.636619772367581343f);  // x * 2/pi
  float r = fma_(q, -1.5703125f, x);
  r = fma_(q, -4.837512969970703125e-4f, r);
  r = fma_(q, -7.54978995489188216e-8f, r);
  float const r2 = r * r;
  float const sp = fma_(fma_(fma_(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
  float const cp = fma_(fma_(fma_(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2,
                        fma_(-0.5f, r2, 1.0f));
  int const n = int(q) & 3;
  float const a = (n & 1) ? cp : sp, b = (n & 1) ? sp : cp;
  sn = (n & 2) ? -a : a;
  cs = ((n + 1) & 2) ? -b : b;
}

// ---------------------------------------------------------------------------------------------
// sampling helpers                                               CC/private/sampling.cu
// ---------------------------------------------------------------------------------------------
DMT_DEV f2 sample_uniform_disk(f2 u) {  // :135-155 (the 3pi/4 branch is the reference's)
  float const a = 2.f * u.x - 1.f;
  float const b = 2.f * u.y - 1.f;
  if (a == 0.f && b == 0.f) return mk2(0.f, 0.f);
  float rho, phi;
  if (fabsf(a) > fabsf(b)) {
    rho = a;
    phi = (kPi / 4) * (b / a);
  } else {
    rho = b;
    phi = (3 * kPi / 4) * (a / b);
  }
  float sn, cs;
  sincos_bounded(phi, sn, cs);
  return mk2(rho * cs, rho * sn);
}
DMT_DEV f3 sample_cos_hemisphere(f3 n, f2 u, float& pdf) {  // :157-167
  f2 const r = sample_uniform_disk(u);
  float const cosTheta = safe_sqrt(1.f - dot(r, r));
  f3 T, B;
  gram_schmidt(n, T, B);
  pdf = cosTheta * kInvPi;
  return r.x * T + r.y * B + cosTheta * n;
}
DMT_DEV f3 sample_uniform_sphere(f2 rnd) {  // :123-133
  float const z = 1.0f - 2.0f * rnd.x;
  float const r = safe_sqrt(1.f - z * z);
  float const phi = 2 * kPi * rnd.y;
  float sn, cs;
  sincos_bounded(phi, sn, cs);
  return mk3(r * cs, r * sn, z);
}
// :86-121.  `xy *= s` in the reference assigns s to both components (common_math.cuh:349-353).
DMT_DEV f3 sample_uniform_cone(f3 N, float omc, f2 rnd, float& cosTheta, float& pdf, int& delta) {
  if (omc > 0) {
    f2 const xy = sample_uniform_disk(rnd);
    float const r2 = dot(xy, xy);
    cosTheta = 1.0f - r2 * omc;
    float const s = safe_sqrt(r2 * omc * (2.0f - r2 * omc));
    pdf = 0.5f / (kPi * fmaxf(omc, 1e-8f));
    f3 T, B;
    gram_schmidt(N, T, B);
    return s * T + s * B + cosTheta * N;
  }
  delta = 1;
  cosTheta = 1.0f;
  pdf = 1.0f;
  return N;
}
DMT_DEV bool ray_sphere(f3 o, f3 d, float tMin, float tMax, f3 C, float radius, f3& ip, float& it) {
  f3 const dv = C - o;  // :51-84
  float const r_sq = radius * radius;
  float const d_sq = dot(dv, dv);
  float const dcos = dot(dv, d);
  if (d_sq > r_sq && dcos < 0.0f) return false;
  f3 const perp = dv - dcos * d;
  float const dsin_sq = dot(perp, perp);
  if (dsin_sq > r_sq) return false;
  float const t = dcos - copysignf(sqrtf(r_sq - dsin_sq), d_sq - r_sq);
  if (t > tMin && t < tMax) {
    it = t;
    ip = o + d * t;
    return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// lights                                                          CC/private/light.cu
// record layout CC/public/cuda-core/light.cuh:10-49 (dword view):
//   w0 = I.x | I.y<<16, w1 = I.z | type<<16, w2..w4 = pos (point/spot)
//   point: w5.lo = radius;  spot: w5 = octa dir, w6 = cosTheta0 | cosThetaE<<16, w7.lo = radius
//   directional: w2 = octa dir, w3.lo = oneMinusCosAngle
// ---------------------------------------------------------------------------------------------
enum : uint32_t { LT_POINT = 0, LT_SPOT = 1, LT_ENV = 2, LT_DIR = 3 };
struct LightSample {
  f3 pLight, direction;
  float pdf;
  int delta;
  float distance;
  float factor;
  DMT_DEV bool valid() const {  // light.cuh:65-67
    return direction.x != 0 && direction.y != 0 && direction.z != 0 && pdf != 0;
  }
};
DMT_DEV uint32_t light_type(Rec32 const& L) { return hi16(L.w[1]); }
DMT_DEV f3 light_intensity(Rec32 const& L) {
  return mk3(h2f(lo16(L.w[0])), h2f(hi16(L.w[0])), h2f(lo16(L.w[1])));
}
DMT_DEV f3 light_pos(Rec32 const& L) {
  return mk3(__uint_as_float(L.w[2]), __uint_as_float(L.w[3]), __uint_as_float(L.w[4]));
}
DMT_DEV LightSample sample_point_light(Rec32 const& L, f3 position, f2 u, bool hadT, f3 normal) {
  LightSample s{};  // light.cu:13-80
  s.factor = 1.0f;
  float const radius = h2f(lo16(L.w[5]));
  float const radiusSqr = sqr(radius);
  f3 lightN = position - light_pos(L);
  float const distSqr = dot(lightN, lightN);
  float const dist = sqrtf(distSqr);
  lightN = lightN / dist;
  bool const effDelta = (radius / dist) < 1e-3f;
  float cosTheta = 0.f;
  if (distSqr > radiusSqr) {
    float const omc = sin_sqr_to_one_minus_cos(radiusSqr / distSqr);
    s.direction = sample_uniform_cone(-lightN, omc, u, cosTheta, s.pdf, s.delta);
    if (effDelta || s.delta) {
      s.pdf = 1;
      s.delta = 1;
    }
  } else {
    if (hadT) {
      s.direction = sample_uniform_sphere(u);
      s.pdf = 0.25f * kInvPi;
    } else {
      s.direction = sample_cos_hemisphere(normal, u, s.pdf);
    }
    cosTheta = -dot(s.direction, lightN);
  }
  s.distance = dist * cosTheta -
               copysignf(safe_sqrt(radiusSqr - distSqr + distSqr * cosTheta * cosTheta),
                         distSqr - radiusSqr);
  s.pLight = position + s.direction * s.distance;
  return s;
}
DMT_DEV LightSample sample_spot_light(Rec32 const& L, f3 position, f2 u, bool hadT, f3 normal) {
  LightSample s{};  // light.cu:110-205
  float const fltMax = 3.402823466e+38f;
  s.distance = fltMax;
  float const radius = h2f(lo16(L.w[7]));
  float const cosThetaE = h2f(hi16(L.w[6]));
  f3 const lpos = light_pos(L);
  float const radiusSqr = radius * radius;
  f3 lightN = position - lpos;
  float const distSqr = dot(lightN, lightN);
  float const dist = sqrtf(distSqr);
  lightN = lightN / dist;
  bool const effDelta = (radius / dist) < 1e-3f;
  bool outside = false;
  float cosTheta = 0.f;
  if (distSqr > radiusSqr) {
    float const omcSpread = 1.f - cosThetaE;
    float const omcHalf = sin_sqr_to_one_minus_cos(radiusSqr / distSqr);
    if (omcHalf < omcSpread) {
      s.direction = sample_uniform_cone(-lightN, omcHalf, u, cosTheta, s.pdf, s.delta);
    } else {
      f3 const spotDir = normalize(dir_from_octa(L.w[5]));  // decoded only where the spread cone is the narrower one
      s.direction = sample_uniform_cone(-spotDir, omcSpread, u, cosTheta, s.pdf, s.delta);
      if (!ray_sphere(position, s.direction, 0.f, fltMax, lpos, radius, s.pLight, s.distance)) {
        outside = true;
        s.pdf = 0;
      }
    }
  } else {
    if (hadT) {
      s.direction = sample_uniform_sphere(u);
      s.pdf = 0.25f * kInvPi;
    } else {
      s.direction = sample_cos_hemisphere(normal, u, s.pdf);
    }
    cosTheta = -dot(s.direction, lightN);
  }
  if (!outside) {
    // spotLightAttenuation -> smoothstep(a,b,x) clamps with fmaxf(fminf(.,0),1), i.e. t == 1 for
    // every input including NaN, so the factor is exactly 1 (light.cuh:77-81,
    // common_math.cuh:484-489); the local-frame transform feeding it is dead and not computed.
    s.factor = 1.f;
    if (s.distance == fltMax) {
      s.distance = dist * cosTheta *
                   copysignf(safe_sqrt(radiusSqr - distSqr + distSqr * cosTheta * cosTheta),
                             distSqr - radiusSqr);  // light.cu:176-179 (product, as written)
    }
    if (effDelta) {
      s.pdf = 1.f;
      s.delta = 1;
    }
    s.pLight = position + s.direction * s.distance;
    f3 const ng = normalize(s.pLight - lpos);
    s.pLight = ng * radius + lpos;
    f3 const newDir = s.pLight - position;
    float const distance = sqrtf(newDir.x * newDir.x + newDir.y * newDir.y + newDir.z * newDir.z);
    s.direction = newDir / distance;
    s.distance = distance;
  }
  return s;
}
DMT_DEV LightSample sample_light(Rec32 const& L, f3 position, f2 u, bool hadT, f3 normal) {
  LightSample s{};  // light.cu:211-253
  uint32_t const type = light_type(L);
  if (type == LT_SPOT) {
    s = sample_spot_light(L, position, u, hadT, normal);
  } else if (type == LT_POINT) {
    s = sample_point_light(L, position, u, hadT, normal);
  } else if (type == LT_ENV) {
    s.direction = sample_uniform_sphere(u);
    s.pdf = 0.25f * kInvPi;
    s.factor = 1.f;
    s.pLight = s.direction;
    s.distance = 3.402823466e+38f;
  } else if (type == LT_DIR) {
    float unused = 0.f;
    s.pLight = sample_uniform_cone(dir_from_octa(L.w[2]), h2f(lo16(L.w[3])), u, unused, s.pdf,
                                   s.delta);
    s.direction = -s.pLight;
    s.factor = 1.f;
    s.delta = 1;
    s.distance = 3.402823466e+38f;
  }
  return s;
}
DMT_DEV f3 eval_light(Rec32 const& L, LightSample const& ls) {  // light.cu:309-320
  f3 Le = light_intensity(L) * ls.factor;
  uint32_t const t = light_type(L);
  if (t == LT_POINT || t == LT_SPOT) Le = Le / (ls.distance * ls.distance);
  return Le;
}

// ---------------------------------------------------------------------------------------------
// BSDFs                      CC/public/cuda-core/bsdf.cuh, CC/private/bsdf.cu
// record layout bsdf.cuh:18-73 (dword view):
//   w0 = W.x | W.y<<16, w1 = W.z | type<<16
//   Oren-Nayar: w2.lo..w3.lo albedo (unused), w3.hi,w4.lo,w4.hi multiScatter, w5.lo roughness,
//               w5.hi = a, w6.lo = b
//   GGX: w2 = energyScale (f32), w3 = phi0 | alphax<<16, w4.lo = alphay
//        dielectric: w4.hi = eta, w5.lo,w5.hi,w6.lo reflectanceTint, w6.hi,w7.lo,w7.hi transTint
//        conductor:  w4.hi,w5.lo,w5.hi eta, w6.lo,w6.hi,w7.lo kappa
// The reference mutates the packed record per hit (prepareBSDF) and re-reads it; here the
// record is decoded once into registers and every fp16 store of the reference becomes q16().
// ---------------------------------------------------------------------------------------------
enum : uint32_t { BS_OREN = 0, BS_GGX_DIEL = 1, BS_GGX_COND = 2, BS_LAMBERT = 3,
                  // This build's own tag: a GGX dielectric record of a material that is "metallic" by a fraction; the NEXT record
                  // of the array is its GGX conductor, the first half of the weight field holds the fraction (fp16).  Blended as
                  // the reference's CPU renderer does (core-material.cpp:275-286, :383-394); *_tex kernels only (path_shade).
                  BS_GGX_BLEND = 4 };

__constant__ float c_ggxE[DMT_GGX_E_ROWS * DMT_GGX_E_COLS] = {DMT_GGX_E_TABLE_VALUES};
__constant__ float c_ggxEavg[DMT_GGX_EAVG_COUNT] = {DMT_GGX_EAVG_TABLE_VALUES};

DMT_DEV float table_read(float const* table, float x, int size) {  // CC extra_math.cu:95-109
  x = fminf(fmaxf(x, 0.f), 1.f) * float(size - 1);
  int const index = int(fminf(float(int(x)), float(size - 1)));
  int const nIndex = int(fminf(float(index + 1), float(size - 1)));
  float const t = x - float(index);
  float const d0 = table[index];
  if (t == 0.f) return d0;
  float const d1 = table[nIndex];
  return (1.f - t) * d0 + t * d1;
}
DMT_DEV float table_read_2d(float const* table, float x, float y, int sx, int sy) {  // :111-126
  y = fminf(fmaxf(y, 0.f), 1.f) * float(sy - 1);
  int const index = int(fminf(float(int(y)), float(sy - 1)));
  int const nIndex = int(fminf(float(index + 1), float(sy - 1)));
  float const t = y - float(index);
  float const d0 = table_read(table + sx * index, x, sx);
  if (t == 0.f) return d0;
  float const d1 = table_read(table + sx * nIndex, x, sx);
  return (1.f - t) * d0 + t * d1;
}

struct Bsdf {  // decoded + prepared, registers only
  uint32_t type;
  f3 weight;
  // A lane's material is ONE kind, so the Oren-Nayar terms and the GGX terms share five registers (both kernels sit at their
  // VGPR budget; every register not kept live across the shading step is one less reload from scratch inside it).
  // Oren-Nayar: a, b, ms.   GGX: escale, ax, ay, phi0, eta, iso, and {rt, tt} or {eta3, kappa3}
  union {
    struct {
      float a, b;
      f3 ms;
    };
    struct {
      float escale, ax, ay, phi0, eta;
    };
  };
  bool iso;
  f3 c0, c1;  // dielectric: reflectanceTint, transmittanceTint; conductor: eta, kappa
};

DMT_DEV float fresnel_dielectric(float cosI, float eta, float& cosT_out) {  // bsdf.cuh:175-202
  cosI = fmaxf(-1.f, fminf(1.f, cosI));
  if (!(cosI > 0.f)) {
    eta = 1.f / eta;
    cosI = fabsf(cosI);
  }
  float const sinI = safe_sqrt(fmaxf(0.f, 1.f - cosI * cosI));
  float const sinT = sinI / eta;
  if (sinT >= 1.f) return 1.f;
  float const cosT = safe_sqrt(fmaxf(0.f, 1.f - sinT * sinT));
  cosT_out = cosT;
  float const rParl = ((eta * cosI) - (cosT)) / ((eta * cosI) + (cosT));
  float const rPerp = ((cosI) - (eta * cosT)) / ((cosI) + (eta * cosT));
  return (rParl * rParl + rPerp * rPerp) * 0.5f;
}
DMT_DEV f3 fresnel_conductor(float cosI, f3 eta, f3 k) {  // bsdf.cuh:204-224
  cosI = fmaxf(-1.f, fminf(1.f, cosI));
  float const c2 = cosI * cosI;
  float const s2 = 1.f - c2;
  f3 const eta2 = eta * eta;
  f3 const k2 = k * k;
  f3 const t0 = eta2 - k2 - s2;
  f3 const a2b2 = sqrt3(t0 * t0 + 4.f * eta2 * k2);
  f3 const t1 = a2b2 + c2;
  f3 const a = sqrt3(0.5f * (a2b2 + t0));
  f3 const t2 = 2.f * cosI * a;
  f3 const Rs = (t1 - t2) / (t1 + t2);
  f3 const t3 = c2 * a2b2 + s2 * s2;
  f3 const t4 = t2 * s2;
  f3 const Rp = Rs * (t3 - t4) / (t3 + t4);
  return 0.5f * (Rp + Rs);
}
DMT_DEV void microfacet_fresnel(Bsdf const& b, float cos_HO, float& cos_HI, f3& R, f3& T) {
  if (b.type == BS_GGX_DIEL) {  // bsdf.cu:331-354
    float const F = fresnel_dielectric(cos_HO, b.eta, cos_HI);
    R = F * b.c0;
    T = (1.f - F) * b.c1;
  } else {
    R = fresnel_conductor(cos_HO, b.c0, b.c1);
    T = mk3(0, 0, 0);
  }
}
DMT_DEV float oren_nayar_G(float cosTheta) {  // bsdf.cu:751-763
  float const piOver2 = kPi / 2;
  float const twoThirds = 2.f / 3.f;
  float const piOver2m = piOver2 - twoThirds;
  if (cosTheta < 1e-6f) return piOver2m - cosTheta;
  float const sinTheta = sin_from_cos(cosTheta);
  float const theta = safe_acos(cosTheta);
  return sinTheta * (theta - twoThirds - sinTheta * cosTheta) +
         twoThirds * (sinTheta / cosTheta) * (1.f - sqr(sinTheta) * sinTheta);
}

// decode + prepareBSDF (bsdf.cu:909-1011) in one step
DMT_DEV Bsdf bsdf_prepare(Rec32 const& r, f3 ns, f3 wo) {
  Bsdf b{};
  b.type = hi16(r.w[1]);
  b.weight = mk3(h2f(lo16(r.w[0])), h2f(hi16(r.w[0])), h2f(lo16(r.w[1])));
  if (b.type == BS_OREN) {
    b.a = h2f(hi16(r.w[5]));
    b.b = h2f(lo16(r.w[6]));
    float const nl = fmaxf(0.f, dot(ns, wo));
    float const Ev = b.a * kPi + b.b * oren_nayar_G(nl);
    f3 ms = b.weight * (1.f - Ev);
    ms = mk3(fmaxf(ms.x, 0.f), fmaxf(ms.y, 0.f), fmaxf(ms.z, 0.f));
    b.ms = q16(ms);
  } else if (b.type == BS_GGX_DIEL || b.type == BS_GGX_COND) {
    b.phi0 = float(lo16(r.w[3])) / 65535 * 2.f * kPi;  // bsdf.cuh:52-55
    uint32_t const axq = hi16(r.w[3]), ayq = lo16(r.w[4]);
    b.iso = axq == ayq;
    b.ax = float(axq) / 65535;
    b.ay = float(ayq) / 65535;
    f3 Fss;
    if (b.type == BS_GGX_DIEL) {
      b.eta = h2f(hi16(r.w[4]));
      b.c0 = mk3(h2f(lo16(r.w[5])), h2f(hi16(r.w[5])), h2f(lo16(r.w[6])));
      b.c1 = mk3(h2f(hi16(r.w[6])), h2f(lo16(r.w[7])), h2f(hi16(r.w[7])));
      Fss = b.c1;
    } else {
      b.c0 = mk3(h2f(hi16(r.w[4])), h2f(lo16(r.w[5])), h2f(hi16(r.w[5])));
      b.c1 = mk3(h2f(lo16(r.w[6])), h2f(hi16(r.w[6])), h2f(lo16(r.w[7])));
    }
    {
      float const cos_HO = fabsf(dot(wo, ns));
      float unused = 0.f;
      f3 R, T;
      microfacet_fresnel(b, cos_HO, unused, R, T);
      b.weight = q16(R + T);  // setWeight -> fp16
    }
    if (b.type == BS_GGX_COND) {  // F82-tint, bsdf.cu:981-992
      f3 const F0 = fresnel_conductor(1.f, b.c0, b.c1);
      f3 const F82 = fresnel_conductor(1.f / 7.f, b.c0, b.c1);
      f3 const B = (lerp3(F0, mk3(1, 1, 1), 0.46266436f) - F82) * 17.651384f;
      Fss = lerp3(F0, mk3(1, 1, 1), 1.f / 21.f) - B * (1.f / 126.f);
    }
    float const alpha2 = b.ax * b.ay;
    float const cos_NO = fmaxf(0.f, dot(ns, wo));
    // energyPreservingGGXScale, bsdf.cu:407-426 (software table lookup = the host branch)
    float const E = table_read_2d(c_ggxE, alpha2, cos_NO, DMT_GGX_E_COLS, DMT_GGX_E_ROWS);
    float const Eavg = table_read(c_ggxEavg, alpha2, DMT_GGX_EAVG_COUNT);
    float const missing = (1.f - E) / E;
    b.escale = 1.f + missing;
    f3 const Fms = Fss * Eavg / (mk3(1, 1, 1) - Fss * (1.f - Eavg));
    b.weight = q16(b.weight * (1.f + Fms * missing) / b.escale);
  }
  return b;
}

struct BsdfSample {
  f3 wi, f;
  float pdf, eta;
  bool delta, refract;
  DMT_DEV bool valid() const { return wi.x != 0 && wi.y != 0 && wi.z != 0 && pdf != 0.f; }
};

DMT_DEV f3 oren_nayar_intensity(Bsdf const& b, f3 n, f3 v, f3 l) {  // bsdf.cu:765-788
  float const nl = fmaxf(dot(n, l), 0.f);
  if (b.b <= 0) {
    float const r = nl * kInvPi;
    return mk3(r, r, r);
  }
  float const nv = fmaxf(dot(n, v), 0.f);
  float t = dot(l, v) - nl * nv;
  if (t > 0.f) t /= fmaxf(nl, nv) + 1.175494351e-38f;
  float const single = b.a + b.b * t;
  float const El = b.a * kPi + b.b * oren_nayar_G(nl);
  f3 const multi = b.ms * (1.f - El);
  return nl * (single + multi);
}
DMT_DEV f3 tangent_from_phi(f3 ns, float phi0) {  // bsdf.cu:279-294
  f3 const ref = fabsf(ns.x) < 0.999f ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 1.0f, 0.0f);
  f3 const t = normalize(cross(ref, ns));
  f3 const bt = cross(ns, t);
  float s, c;
  sincos_bounded(phi0, s, c);
  return c * t + s * bt;
}
DMT_DEV f3 sample_ggx_vndf(f3 wo, f2 u, float ax, float ay) {  // bsdf.cu:303-329
  f3 const V = normalize(mk3(ax * wo.x, ay * wo.y, wo.z));
  f3 T1, T2;
  float const lensq = V.x * V.x + V.y * V.y;
  if (lensq > 1e-7f) {
    float const invLen = rsqrt_ieee(lensq);
    T1 = mk3(-V.y * invLen, V.x * invLen, 0.f);
    T2 = cross(V, T1);
  } else {
    T1 = mk3(1, 0, 0);
    T2 = mk3(0, 1, 0);
  }
  f2 t = sample_uniform_disk(u);
  t.y = lerp1(safe_sqrt(1.f - t.x * t.x), t.y, 0.5f * (1.f + V.z));
  f3 Nh = t.x * T1 + t.y * T2 + safe_sqrt(1.f - dot(t, t)) * V;
  Nh = normalize(mk3(ax * Nh.x, ay * Nh.y, fmaxf(0.f, Nh.z)));
  return Nh;
}
DMT_DEV float ggx_lambda_from(float x) { return 0.5f * (sqrtf(1.f + x) - 1.f); }
DMT_DEV float ggx_D(float alpha2, float cos_NH) {
  float const c2 = fminf(cos_NH * cos_NH, 1.f);
  float const omc2 = 1.f - c2;
  return alpha2 / (kPi * sqr(omc2 + alpha2 * c2));
}
DMT_DEV float ggx_lambda(float alpha2, float cos_N) {
  return ggx_lambda_from(alpha2 * fmaxf(0.f, 1.f / sqr(cos_N)));
}
DMT_DEV float ggx_aniso_D(float ax, float ay, f3 lH) {
  lH = lH / mk3(ax, ay, 1.f);
  return kInvPi / ((ax * ay) * sqr(dot(lH, lH)));
}
DMT_DEV float ggx_aniso_lambda(float ax, float ay, f3 V) {
  return ggx_lambda_from((sqr(ax * V.x) + sqr(ay * V.y)) / sqr(V.z));
}
constexpr float kThroughputEps = 1e-6f;

DMT_DEV BsdfSample sample_ggx(Bsdf const& b, f3 wo, f3 ns, f3 ng, f2 u, float uc) {
  BsdfSample s{};  // bsdf.cu:457-569
  float const cos_NO = dot(ns, wo);
  s.eta = 1.f;
  float invEta = 1.f;
  s.delta = fmaxf(b.ax, b.ay) < 1e-3f;
  f3 H = ns, lH = mk3(0, 0, 0), lO = mk3(0, 0, 0);
  if (!s.delta) {
    f3 X, Y;
    if (b.iso)
      gram_schmidt(ns, X, Y);
    else
      orthonormal_tangent(ns, tangent_from_phi(ns, b.phi0), X, Y);
    lO = mk3(dot(X, wo), dot(Y, wo), cos_NO);
    lH = sample_ggx_vndf(lO, u, b.ax, b.ay);
    H = lH.x * X + lH.y * Y + lH.z * ns;
  }
  float const cos_HO = dot(H, wo);
  float cos_HI = 0.f;
  f3 R, T;
  microfacet_fresnel(b, cos_HO, cos_HI, R, T);
  if (near_zero_pos(R, kThroughputEps) && near_zero_pos(T, kThroughputEps)) return s;
  float const pdfReflect = fminf(fmaxf(average(R) / average(R + T), 0.f), 1.f);
  s.refract = uc > pdfReflect;
  if (s.refract) invEta = 1.f / b.eta;
  // refractAngle (bsdf.cu:358-364) / mirror direction
  s.wi = s.refract ? (invEta * dot(H, wo) + cos_HI) * H - invEta * wo : 2.f * cos_HO * H - wo;
  if (dot(ng, s.wi) <= 0 && !s.refract) {
    s.pdf = 0;
    return s;
  }
  if (s.refract) {
    s.f = T;
    s.pdf = 1.f - pdfReflect;
    s.delta = s.delta || (fabsf(s.eta - 1.f) < 1e-4f);  // eta is 1 here: refraction is "delta"
  } else {
    s.pdf = pdfReflect;  // the reference leaves f = 0 on the reflection lobe (bsdf.cu:526-534)
  }
  if (s.delta) {
    s.pdf *= 1e6f;
    s.f = s.f * 1e6f;
  } else {
    float D, lamI, lamO;
    if (b.iso || s.refract) {
      float const alpha2 = b.ax * b.ay;
      D = ggx_D(alpha2, lH.z);
      lamI = ggx_lambda(alpha2, dot(ns, s.wi));
      lamO = ggx_lambda(alpha2, cos_NO);
    } else {
      f3 const lI = 2.f * cos_HO * lH - lO;
      D = ggx_aniso_D(b.ax, b.ay, lH);
      lamI = ggx_aniso_lambda(b.ax, b.ay, lI);
      lamO = ggx_aniso_lambda(b.ax, b.ay, lO);
    }
    float const common =
        D / cos_NO * (s.refract ? fabsf(cos_HO * cos_HI) / sqr(cos_HI + cos_HO * invEta) : 0.25f);
    s.pdf *= common / (1.f + lamO);
    s.f = s.f * (common / (1.f + lamO + lamI));
  }
  return s;
}
DMT_DEV f3 eval_ggx(Bsdf const& b, f3 wo, f3 wi, f3 ns, f3 ng, float& pdf) {  // bsdf.cu:571-667
  bool const conductor = b.type == BS_GGX_COND;
  bool const hasReflection = conductor ? true : luminance(b.c0) > kThroughputEps;
  bool const hasTransmission = conductor ? false : luminance(b.c1) > kThroughputEps;
  bool const isotropic = b.ax == b.ay;
  float const cos_NO = dot(ns, wo);
  float const cos_NI = dot(ns, wi);
  float const cos_NgI = dot(ng, wi);
  bool const isT = cos_NI < 0.f;
  float const ior = isT ? b.eta : 1.f;
  bool const effSpecular = fmaxf(b.ax, b.ay) < 1e-3f;
  if (cos_NO <= 0.f || (cos_NgI < 0) != isT || effSpecular || (!hasReflection && cos_NgI > 0.f) ||
      (!hasTransmission && cos_NgI < 0.f)) {
    pdf = 0.f;
    return mk3(0, 0, 0);
  }
  f3 H = isT ? (ior * wi + wo) : (wi + wo);
  float const invLen_H = rsqrt_ieee(dot(H, H));
  H = H * invLen_H;
  float const cos_HO = dot(H, wo);
  float unused = 0.f;
  f3 R, T;
  microfacet_fresnel(b, cos_HO, unused, R, T);
  if (near_zero_pos(R, kThroughputEps) && near_zero_pos(T, kThroughputEps)) {
    pdf = 0.f;
    return mk3(0, 0, 0);
  }
  float const cos_NH = dot(ns, H);
  float D, lamI, lamO;
  if (isotropic || isT) {
    float const alpha2 = b.ax * b.ay;
    D = ggx_D(alpha2, cos_NH);
    lamI = ggx_lambda(alpha2, cos_NI);
    lamO = ggx_lambda(alpha2, cos_NO);
  } else {
    f3 X, Y;
    orthonormal_tangent(ns, tangent_from_phi(ns, b.phi0), X, Y);
    f3 const lH = mk3(dot(X, H), dot(Y, H), dot(ns, H));
    f3 const lO = mk3(dot(X, wo), dot(Y, wo), cos_NO);
    f3 const lI = mk3(dot(X, wi), dot(Y, wi), cos_NI);
    D = ggx_aniso_D(b.ax, b.ay, lH);
    lamI = ggx_aniso_lambda(b.ax, b.ay, lI);
    lamO = ggx_aniso_lambda(b.ax, b.ay, lO);
  }
  float const common = D / cos_NO * (isT ? sqr(ior * invLen_H) * fabsf(cos_HO * dot(H, wi)) : 0.25f);
  float const pdfReflect = average(R) / average(R + T);
  float const lobePdf = isT ? 1.f - pdfReflect : pdfReflect;
  pdf = lobePdf * common / (1.f + lamO);
  return b.escale * (isT ? T : R) * common / (1.f + lamO + lamI);
}

DMT_DEV BsdfSample sample_bsdf(Bsdf const& b, f3 wo, f3 ns, f3 ng, f2 u, float uc) {
  BsdfSample s{};  // bsdf.cu:851-880
  if (dot(wo, ng) > 0.0f) {
    if (dot(ns, ng) < 0.0f) ns = -ns;  // faceForward
    if (b.type == BS_OREN || b.type == BS_LAMBERT) {
      s.eta = 1.f;
      s.wi = sample_cos_hemisphere(ns, u, s.pdf);
      if (dot(ng, s.wi) > 0.f) {
        s.f = b.type == BS_OREN ? oren_nayar_intensity(b, ns, wo, s.wi) : mk3(s.pdf, s.pdf, s.pdf);
      } else {
        s.pdf = 0;
      }
    } else {
      s = sample_ggx(b, wo, ns, ng, u, uc);
    }
  }
  return s;
}
DMT_DEV f3 eval_bsdf(Bsdf const& b, f3 wo, f3 wi, f3 ns, f3 ng, float& pdf) {  // bsdf.cu:882-907
  pdf = 0;
  if (b.type == BS_OREN) {
    float const cos_NI = dot(ns, wi);
    if (cos_NI > 0.f) {
      pdf = cos_NI * kInvPi;
      return oren_nayar_intensity(b, ns, wo, wi);
    }
    return mk3(0, 0, 0);
  }
  if (b.type == BS_LAMBERT) {
    float const cos_NI = fmaxf(dot(ns, wi), 0.f);
    pdf = cos_NI * kInvPi;
    return mk3(pdf, pdf, pdf);
  }
  return eval_ggx(b, wo, wi, ns, ng, pdf);
}

// min(static_cast<int>(u * count), count - 1) with CUDA's min(int, unsigned) -> unsigned
DMT_DEV uint32_t pick_index(float u, uint32_t count) {
  uint32_t const a = uint32_t(int(u * float(count)));
  uint32_t const b = count - 1u;
  return a < b ? a : b;
}

}  // namespace dmt
