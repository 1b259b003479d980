// envmap.hpp -- environment-map light: host table builder + device sampling / evaluation.
//
// Semantics are the reference CPU renderer's: an equirectangular RGB image importance-sampled through a
// piecewise-constant 2-D distribution (src/core/private/core-math.cu:385-675, PiecewiseConstant1D/2D), sampled and
// evaluated by src/core/private/core-light.cpp:84-117 (EnvLight), :394-442 (sample), :444-452 (eval by uv),
// :454-491 (eval by direction + pdf), and combined with BSDF sampling by the rules of
// src/core/private/core-render.cpp:154-163 (a path ray leaves the scene), :290-299 (env map chosen for NEE with
// probability 1/2), :357-369 (NEE estimator).  The reference's quirks are kept ("A18").
//
// Layout in HBM: five float tables -- conditional function |f| and its normalised inclusive CDF (h x w each, row
// major), per-row integrals, marginal |f| and CDF (h each) -- plus the RGB image (h x w x 3 floats).  A sample is
// two binary searches (log2 h + log2 w dependent 4-byte reads, L2 resident: 1k x 512 map = 6 MiB image + 4 MiB
// tables) and one texel fetch.
#pragma once

#include <stdint.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "pt_device.hpp"

namespace dmt {

struct EnvView {  // device pointers; w == 0 means "no env map"
  float const* func;     // h x w
  float const* cdf;      // h x w
  float const* rowInt;   // h
  float const* mFunc;    // h
  float const* mCdf;     // h
  float const* rgb;      // h x w x 3
  float mInt;
  int w, h;
  float qx, qy, qz, qw;  // lightFromRender, normalised
};

namespace envmap {

struct Tables {
  std::vector<float> func, cdf, rowInt, mFunc, mCdf;
  float mInt = 0.f;
};

// PiecewiseConstant1D constructor on [0,1] (core-math.cu:385-535): |f|, inclusive CDF summed in the AVX2 block
// order of the reference (in-lane prefix sums of 4, lane 0's total added to lane 1, running carry), normalised.
inline float build1d(float const* f, uint32_t n, float* absf, float* cdf) {
  for (uint32_t i = 0; i < n; ++i) absf[i] = std::fabs(f[i]);
  float const fac = 1.0f / float(n);
  float carry = 0.f;
  uint32_t const nb = n & ~7u;
  for (uint32_t b = 0; b < nb; b += 8) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = f[b + uint32_t(j)] * fac;
    for (int l = 0; l < 8; l += 4) {
      float const s1 = x[l + 1] + x[l], s2 = x[l + 2] + x[l + 1], s3 = x[l + 3] + x[l + 2];
      x[l + 1] = s1, x[l + 2] = s2 + x[l], x[l + 3] = s3 + s1;
    }
    float const lane0 = x[3];
    for (int j = 4; j < 8; ++j) x[j] = x[j] + lane0;
    for (int j = 0; j < 8; ++j) cdf[b + uint32_t(j)] = x[j] + carry;
    carry = cdf[b + 7];
  }
  for (uint32_t i = nb; i < n; ++i) cdf[i] = (i ? cdf[i - 1] : 0.f) + f[i];
  float const integral = cdf[n - 1];
  bool const zero = std::fabs(integral) <= std::numeric_limits<float>::epsilon();
  float const norm = 1.0f / (zero ? float(n) : integral);
  for (uint32_t i = 0; i < n; ++i) cdf[i] = zero ? float(i) * norm : cdf[i] * norm;
  return integral;
}

inline Tables build(float const* rgb, int w, int h) {
  Tables t;
  t.func.resize(size_t(w) * size_t(h)), t.cdf.resize(size_t(w) * size_t(h));
  t.rowInt.resize(size_t(h)), t.mFunc.resize(size_t(h)), t.mCdf.resize(size_t(h));
  std::vector<float> row(static_cast<size_t>(w), 0.f);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {  // RGB::avg(), core-light.cpp:93-98
      float const* p = rgb + 3 * (size_t(x) + size_t(y) * size_t(w));
      row[size_t(x)] = (p[0] + p[1] + p[2]) / 3.f;
    }
    t.rowInt[size_t(y)] = build1d(row.data(), uint32_t(w), &t.func[size_t(y) * size_t(w)], &t.cdf[size_t(y) * size_t(w)]);
  }
  t.mInt = build1d(t.rowInt.data(), uint32_t(h), t.mFunc.data(), t.mCdf.data());
  return t;
}

}  // namespace envmap

// ---- device side ------------------------------------------------------------------------------------------
DMT_DEV int env_find_interval(int sz, float const* cdf, float u) {  // core-math.cu:553-564
  int size = sz - 2, first = 1;
  while (size > 0) {
    int const half = size >> 1, middle = first + half;
    bool const pred = cdf[middle] <= u;
    first = pred ? middle + 1 : first;
    size = pred ? size - (half + 1) : half;
  }
  int const r = first - 1;
  return r < 0 ? 0 : (r > sz - 2 ? sz - 2 : r);
}
DMT_DEV float env_sample1d(float const* absf, float const* cdf, int n, float integral, float u, float& pdf, int& off) {
#pragma clang fp reciprocal(off)  // uv picks a texel downstream: keep these divisions as written (build uses -freciprocal-math)
  off = env_find_interval(n, cdf, u);
  float const c0 = cdf[off], c1 = cdf[off + 1];
  float du = u - c0;
  if (c1 - c0 > 0) du = du / (c1 - c0);
  pdf = integral > 0 ? absf[off] / integral : 0.f;
  return (float(off) + du) / float(n);
}
DMT_DEV f3 env_rotate(f3 v, float ax, float ay, float az, float aw, float bx, float by, float bz, float bw) {
  // (a * (v,0)) * b, Hamilton products in glm's operand order; returns the vector part
  float const tx = aw * v.x + ay * v.z - az * v.y, ty = aw * v.y + az * v.x - ax * v.z;
  float const tz = aw * v.z + ax * v.y - ay * v.x, tw = -ax * v.x - ay * v.y - az * v.z;
  return mk3(tw * bx + tx * bw + ty * bz - tz * by, tw * by + ty * bw + tz * bx - tx * bz,
             tw * bz + tz * bw + tx * by - ty * bx);
}
struct EnvSampleDev {
  f3 wi;
  float pdf;
  f2 uv;
  bool ok;
};
DMT_DEV EnvSampleDev env_sample(EnvView const& e, f2 u) {  // core-light.cpp:394-442
  EnvSampleDev r;
  r.ok = false, r.pdf = 0.f, r.wi = mk3(0, 0, 0);
  float p0, p1;
  int iu, iv;
  float const d1 = env_sample1d(e.mFunc, e.mCdf, e.h, e.mInt, u.y, p1, iv);
  float const d0 = env_sample1d(e.func + size_t(iv) * size_t(e.w), e.cdf + size_t(iv) * size_t(e.w), e.w, e.rowInt[iv], u.x, p0, iu);
  r.uv.x = d0, r.uv.y = d1;
  float const mapPdf = p0 * p1;
  if (mapPdf == 0.f) return r;
  float const phi = fminf(fmaxf(1.f - 2.f * kPi * d0, -kPi), kPi);
  float const theta = fminf(fmaxf(kPi * d1, 0.f), kPi);
  float sp, cp, st, ct;
  sincos_bounded(phi, sp, cp);
  sincos_bounded(theta, st, ct);
  r.wi = env_rotate(mk3(sp * ct, sp * st, cp), -e.qx, -e.qy, -e.qz, e.qw, e.qx, e.qy, e.qz, e.qw);
  r.pdf = mapPdf / (4 * kPi);
  r.ok = true;
  return r;
}
DMT_DEV f3 env_texel(EnvView const& e, int xi, int yi) {
  float const* p = e.rgb + 3 * (size_t(xi) + size_t(yi) * size_t(e.w));
  return mk3(p[0], p[1], p[2]);
}
DMT_DEV f3 env_eval_uv(EnvView const& e, f2 uv) {  // core-light.cpp:444-452
  int const xi = int(roundf(fminf(fmaxf(uv.x, 0.f), 1.f) * float(e.w - 1)));
  int const yi = int(roundf(fminf(fmaxf(uv.y, 0.f), 1.f) * float(e.h - 1)));
  return env_texel(e, xi, yi);
}
DMT_DEV f3 env_eval_dir(EnvView const& e, f3 wi, float& pdf) {  // core-light.cpp:454-491
  f3 const wl = env_rotate(wi, e.qx, e.qy, e.qz, e.qw, -e.qx, -e.qy, -e.qz, e.qw);
  float const theta = acosf(fminf(fmaxf(wl.z, -1.f), 1.f));
  float const phi = atan2f(wl.y, wl.x);
  float const uvx = 0.5f * (1.f + phi / kPi), uvy = 1.f - theta / kPi;
  int xi = int(uvx * float(e.w)), yi = int(uvy * float(e.h));
  xi = xi < 0 ? 0 : (xi > e.w - 1 ? e.w - 1 : xi);
  yi = yi < 0 ? 0 : (yi > e.h - 1 ? e.h - 1 : yi);
  pdf = e.func[size_t(yi) * size_t(e.w) + size_t(xi)] / e.mInt / (4.f * kPi);
  return env_texel(e, xi, yi);
}

}  // namespace dmt
