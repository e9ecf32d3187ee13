// trace_math.h -- the arithmetic shared by the HIP kernels and the host-side
// fetch counter: per-ray setup, the padded slab test and the closest-hit
// triangle test.  Compiled with -ffp-contract=off on both sides; the only
// fused operations are the explicit fmaf() calls in the slab test.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

constexpr float kFltMax = 3.402823466e+38f;
constexpr float kInvClamp = 1.2676506e30f;         // 2^100: stands in for 1/0
constexpr float kPadFactor = 7.62939453125e-06f;   // 2^-17 of (|origin|max + scene scale)
constexpr float kDetEps = 1e-6f;                   // (float)1e-6 = 0x358637BD, just BELOW the double 1e-6

struct RayCtx {
  float ox, oy, oz;      // origin
  float dx, dy, dz;      // direction (not renormalised)
  float ix, iy, iz;      // clamped reciprocal direction
  float nx, ny, nz;      // -(origin + sgn(d)*pad) * inv : near-plane constant
  float fx, fy, fz;      // -(origin - sgn(d)*pad) * inv : far-plane constant
  bool sx, sy, sz;       // direction component negative
};

RT_HD float SafeInv(float d)
{
  float inv = 1.0f / d;
  // |inv| above 2^100 (including the +-inf of d == 0) and NaN-free clamping.
  if (!(fabsf(inv) <= kInvClamp)) inv = copysignf(kInvClamp, d);
  return inv;
}

// Boxes are tested against planes pushed outward by `pad` = kPadFactor * (max|o| + scale).
// That margin covers the rounding of the slab arithmetic below and the rounding of the
// triangle test (whose accepted hits may lie a few ulps of the origin distance outside the
// exact triangle), so a box holding an accepted hit is never culled.
RT_HD void SetupRay(RayCtx &r, float ox, float oy, float oz, float dx, float dy, float dz, float scale)
{
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.dx = dx; r.dy = dy; r.dz = dz;
  r.ix = SafeInv(dx); r.iy = SafeInv(dy); r.iz = SafeInv(dz);
  r.sx = r.ix < 0.f; r.sy = r.iy < 0.f; r.sz = r.iz < 0.f;
  const float pad = kPadFactor * (fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + scale);
  const float px = r.sx ? -pad : pad, py = r.sy ? -pad : pad, pz = r.sz ? -pad : pad;
  r.nx = -(ox + px) * r.ix; r.ny = -(oy + py) * r.iy; r.nz = -(oz + pz) * r.iz;
  r.fx = -(ox - px) * r.ix; r.fy = -(oy - py) * r.iy; r.fz = -(oz - pz) * r.iz;
}

// Entry distance (clamped at 0) of the padded box [lo,hi]; the box is worth visiting
// iff the returned value <= min(exit distance, tbest) -- inclusive, so a box whose entry
// equals the best hit so far is still opened and exact ties are resolved by triangle index.
RT_HD bool SlabTest(const RayCtx &r, float lox, float loy, float loz, float hix, float hiy, float hiz, float tbest,
                    float &tenter)
{
  const float nearx = r.sx ? hix : lox, farx = r.sx ? lox : hix;
  const float neary = r.sy ? hiy : loy, fary = r.sy ? loy : hiy;
  const float nearz = r.sz ? hiz : loz, farz = r.sz ? loz : hiz;
  const float t0x = fmaf(nearx, r.ix, r.nx), t1x = fmaf(farx, r.ix, r.fx);
  const float t0y = fmaf(neary, r.iy, r.ny), t1y = fmaf(fary, r.iy, r.fy);
  const float t0z = fmaf(nearz, r.iz, r.nz), t1z = fmaf(farz, r.iz, r.fz);
  const float tmin = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), 0.f);
  const float tmax = fminf(fminf(fminf(t1x, t1y), t1z), tbest);
  tenter = tmin;
  return tmin <= tmax;
}

// ---- 16-bit quantized boxes -------------------------------------------------------------
// A child box is three dwords, one per axis, lo in the low half and hi in the high half, on a
// uniform 65536-step grid over the scene box: plane(q) = grid_lo + q * step, lo rounded down and
// hi rounded up on the host, so the decoded box always contains the float box.  A child costs
// 16 bytes (three dwords + reference) instead of 28, which halves the record fetches the texture
// path has to serve per traversal step.  The slab distances come straight from the grid indices:
//   t = (grid_lo + q*step - o') * inv = q * (step*inv) + (grid_lo - o') * inv = q*A + B
// with A and B per-ray constants.  The index never goes through an int-to-float conversion: the
// 16 bits are dropped into the mantissa of 2^23 (one byte permute picks the near or the far half
// of the axis word and supplies the exponent), which reads as the float F = 2^23 + q exactly, and
//   t = fma(F, A, B - 2^23*A).
// Rounding B - 2^23*A once costs at most half a grid cell of plane position on top of the
// roundings of A, B and the fma that the outward padding of o' already covers; the host rounds
// the boxes outward by one more cell for it.
struct QuantGrid {
  float lo[3];
  float step[3];
  float inv_limit[3];   // bound on |1/d| per axis: keeps 2^23 * step * |1/d| finite for grids coarser than 1 unit per cell
};

RT_HD void SetQuantGrid(QuantGrid &g, const float lo[3], const float step[3])
{
  for (int k = 0; k < 3; ++k) {
    g.lo[k] = lo[k];
    g.step[k] = step[k];
    g.inv_limit[k] = kInvClamp / fmaxf(step[k], 1.f);
  }
}

constexpr uint32_t kBiasBits = 0x4B000000u;   // 2^23 as a float
constexpr float kBias = 8388608.0f;
// Byte selectors for {axis word (bytes 4..7), kBiasBits (bytes 0..3)}: result = bias exponent | chosen half.
constexpr uint32_t kSelLoHalf = 0x03020504u, kSelHiHalf = 0x03020706u;

struct RayCtxQ {
  float ax, ay, az;        // step * inv
  float bnx, bny, bnz;     // (grid_lo - (o + sgn*pad)) * inv - 2^23 * a : near planes
  float bfx, bfy, bfz;     // (grid_lo - (o - sgn*pad)) * inv - 2^23 * a : far planes
  uint32_t nx, ny, nz;     // selector of the near plane's half of the axis word
  uint32_t fx, fy, fz;     // selector of the far plane's half
};

RT_HD void SetupRayQAxis(float o, float inv, bool neg, float pad, float glo, float step, float lim, float &a, float &bn,
                         float &bf, uint32_t &seln, uint32_t &self)
{
  inv = fminf(fmaxf(inv, -lim), lim);
  const float p = neg ? -pad : pad;
  a = step * inv;
  bn = fmaf(-kBias, a, (glo - (o + p)) * inv);
  bf = fmaf(-kBias, a, (glo - (o - p)) * inv);
  seln = neg ? kSelHiHalf : kSelLoHalf;
  self = neg ? kSelLoHalf : kSelHiHalf;
}

RT_HD void SetupRayQ(RayCtxQ &q, const RayCtx &r, const QuantGrid &g, float scale)
{
  const float pad = kPadFactor * (fmaxf(fmaxf(fabsf(r.ox), fabsf(r.oy)), fabsf(r.oz)) + scale);
  SetupRayQAxis(r.ox, r.ix, r.sx, pad, g.lo[0], g.step[0], g.inv_limit[0], q.ax, q.bnx, q.bfx, q.nx, q.fx);
  SetupRayQAxis(r.oy, r.iy, r.sy, pad, g.lo[1], g.step[1], g.inv_limit[1], q.ay, q.bny, q.bfy, q.ny, q.fy);
  SetupRayQAxis(r.oz, r.iz, r.sz, pad, g.lo[2], g.step[2], g.inv_limit[2], q.az, q.bnz, q.bfz, q.nz, q.fz);
}

// 2^23 + (the half of `word` that `sel` names), as a float.
RT_HD float BiasedPlane(uint32_t word, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(__builtin_amdgcn_perm(word, kBiasBits, sel));
#else
  const uint32_t bits = kBiasBits | (sel == kSelLoHalf ? (word & 0xFFFFu) : (word >> 16));
  float f;
  std::memcpy(&f, &bits, sizeof f);
  return f;
#endif
}

// Entry (clamped at 0) and exit (clamped at tbest) distances from already biased plane values.
RT_HD void SlabRangeBiased(const RayCtxQ &q, float nx, float ny, float nz, float fx, float fy, float fz, float tbest,
                           float &tmin, float &tmax)
{
  const float t0x = fmaf(nx, q.ax, q.bnx), t1x = fmaf(fx, q.ax, q.bfx);
  const float t0y = fmaf(ny, q.ay, q.bny), t1y = fmaf(fy, q.ay, q.bfy);
  const float t0z = fmaf(nz, q.az, q.bnz), t1z = fmaf(fz, q.az, q.bfz);
  tmin = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), 0.f);
  tmax = fminf(fminf(fminf(t1x, t1y), t1z), tbest);
}

// Same acceptance rule as SlabTest.
RT_HD bool SlabTestBiased(const RayCtxQ &q, float nx, float ny, float nz, float fx, float fy, float fz, float tbest,
                          float &tenter)
{
  float tmax;
  SlabRangeBiased(q, nx, ny, nz, fx, fy, fz, tbest, tenter, tmax);
  return tenter <= tmax;
}

// cx, cy, cz: the three packed axis words of one child.
RT_HD bool SlabTestQ(const RayCtxQ &q, uint32_t cx, uint32_t cy, uint32_t cz, float tbest, float &tenter)
{
  return SlabTestBiased(q, BiasedPlane(cx, q.nx), BiasedPlane(cy, q.ny), BiasedPlane(cz, q.nz), BiasedPlane(cx, q.fx),
                        BiasedPlane(cy, q.fy), BiasedPlane(cz, q.fz), tbest, tenter);
}

// Closest-hit triangle test on a pre-gathered record (a, e1 = b - a, e2 = c - a).
// Operation for operation the reference's core/ray_tri.cpp:20-51 (SUB/CROSS/DOT of
// core/raydata.h:104-117, DOT summed left to right, 1.f/det a correctly-rounded divide),
// evaluated without the early returns; the accept conditions are the same predicates in
// the same sense, so NaNs fall the same way.  NEAR_ZERO(det) and t > EPSILON compare
// against a DOUBLE 1e-6 in the reference; for float operands those are |det| <= 1e-6f and
// t > 1e-6f because (float)1e-6 < 1e-6 < nextafterf((float)1e-6, 1).
RT_HD bool TriTest(float ax, float ay, float az, float e1x, float e1y, float e1z, float e2x, float e2y, float e2z,
                   const RayCtx &r, float &tout, float &uout, float &vout)
{
  const float Px = r.dy * e2z - r.dz * e2y;
  const float Py = r.dz * e2x - r.dx * e2z;
  const float Pz = r.dx * e2y - r.dy * e2x;
  const float det = e1x * Px + e1y * Py + e1z * Pz;
  bool ok = !(fabsf(det) <= kDetEps);
  const float det_inv = 1.f / det;
  const float Tx = r.ox - ax, Ty = r.oy - ay, Tz = r.oz - az;
  const float u = (Tx * Px + Ty * Py + Tz * Pz) * det_inv;
  ok = ok && !(u < 0.f || u > 1.f);
  const float Qx = Ty * e1z - Tz * e1y;
  const float Qy = Tz * e1x - Tx * e1z;
  const float Qz = Tx * e1y - Ty * e1x;
  const float v = (r.dx * Qx + r.dy * Qy + r.dz * Qz) * det_inv;
  ok = ok && !(v < 0.f || v + u > 1.f);
  const float t = (e2x * Qx + e2y * Qy + e2z * Qz) * det_inv;
  ok = ok && (t > kDetEps);
  tout = t; uout = u; vout = v;
  return ok;
}

}  // namespace rt
