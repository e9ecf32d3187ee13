// raygen.cpp -- host ray generators.
//
// PrimaryRays restates the camera table of render_client/img_main.cpp:253-267
// (pinhole at (W/2, H/2, -5W) looking at the pixel grid in the z=0 plane, so
// tan(half horizontal fov) = 0.1; directions normalised by core/raymath.cpp:6-15).
// AoRays is a seeded, trig-free generator of incoherent secondary rays; it has
// no counterpart in the reference.
#include "host_scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

namespace rt {

void PrimaryRays(int W, int H, const float origin[3], rt_ray_isect *out)
{
  const rt_vec3 eye{W / 2.f, H / 2.f, static_cast<float>(-5 * W), 0.f};
  for (int j = 0; j < H; ++j) {
    for (int i = 0; i < W; ++i) {
      rt_vec3 to{static_cast<float>(i) - eye.x, static_cast<float>(j) - eye.y, 0.f - eye.z, 0.f};
      rt_vec3 d = Normalize(to);
      rt_ray_isect &r = out[static_cast<size_t>(i) + static_cast<size_t>(j) * W];
      r.origin = rt_float4{origin[0], origin[1], origin[2], 1.f};
      r.direction = rt_float4{d.x, d.y, d.z, 0.f};
      r.normal = rt_float4{0.f, 0.f, 1.f, 0.f};
      r.uv = rt_vec2{0.f, 0.f};
      r.triangle_id = 0xFFFFFFFFu;
      r.t_bits = 0;
    }
  }
}

void FrameCamera(rt_aabb const &root, int W, int H, float origin_out[3])
{
  const float cx = 0.5f * (root.a.x + root.b.x), cy = 0.5f * (root.a.y + root.b.y);
  const float ex = root.b.x - root.a.x, ey = root.b.y - root.a.y;
  const float tan_h = 0.1f, tan_v = 0.1f * static_cast<float>(H) / static_cast<float>(W);
  const float dist = 1.1f * std::max(ex / (2.f * tan_h), ey / (2.f * tan_v));
  origin_out[0] = cx;
  origin_out[1] = cy;
  origin_out[2] = root.a.z - dist;
}

namespace {

inline uint64_t SplitMix64(uint64_t &s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

inline float Unit(uint64_t bits) { return static_cast<float>(bits >> 40) * (1.0f / 16777216.0f); }  // [0,1)

void AoRange(rt_vec3 const *verts, rt_face const *faces, size_t ntris, float offset, uint64_t seed, size_t first,
             size_t count, rt_ray_isect *out)
{
  for (size_t n = 0; n < count; ++n) {
    const size_t k = first + n;
    uint64_t s = seed ^ (k * 0xD1342543DE82EF95ull);
    const rt_face &f = faces[k % ntris];
    const rt_vec3 &a = verts[f.s[0]], &b = verts[f.s[1]], &c = verts[f.s[2]];
    float u = Unit(SplitMix64(s)), v = Unit(SplitMix64(s));
    if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }
    const float e1[3] = {b.x - a.x, b.y - a.y, b.z - a.z}, e2[3] = {c.x - a.x, c.y - a.y, c.z - a.z};
    float nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    float len = std::sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    if (!(len > 0.f)) { nrm[0] = 0.f; nrm[1] = 0.f; nrm[2] = 1.f; len = 1.f; }
    for (float &x : nrm) x /= len;
    // Uniform point on the unit sphere (Marsaglia 1972: arithmetic and sqrt only).
    float x1, x2, ss;
    do {
      x1 = 2.f * Unit(SplitMix64(s)) - 1.f;
      x2 = 2.f * Unit(SplitMix64(s)) - 1.f;
      ss = x1 * x1 + x2 * x2;
    } while (ss >= 1.f || ss == 0.f);
    const float root = std::sqrt(1.f - ss);
    // normal + sphere point = cosine-weighted direction about the normal.
    float d[3] = {nrm[0] + 2.f * x1 * root, nrm[1] + 2.f * x2 * root, nrm[2] + (1.f - 2.f * ss)};
    float dl = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (!(dl > 1e-6f)) { d[0] = nrm[0]; d[1] = nrm[1]; d[2] = nrm[2]; dl = 1.f; }
    rt_ray_isect &r = out[n];
    r.origin = rt_float4{a.x + u * e1[0] + v * e2[0] + offset * nrm[0], a.y + u * e1[1] + v * e2[1] + offset * nrm[1],
                         a.z + u * e1[2] + v * e2[2] + offset * nrm[2], 1.f};
    r.direction = rt_float4{d[0] / dl, d[1] / dl, d[2] / dl, 0.f};
    r.normal = rt_float4{nrm[0], nrm[1], nrm[2], 0.f};
    r.uv = rt_vec2{0.f, 0.f};
    r.triangle_id = 0xFFFFFFFFu;
    r.t_bits = 0;
  }
}

}  // namespace

void AoRays(rt_vec3 const *verts, rt_face const *faces, size_t ntris, rt_aabb const &root, uint64_t seed, size_t first,
            size_t count, rt_ray_isect *out)
{
  if (ntris == 0 || count == 0) return;
  const float ext = std::max(root.b.x - root.a.x, std::max(root.b.y - root.a.y, root.b.z - root.a.z));
  const float offset = 1e-3f * ext;
  unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (count < 65536) nt = 1;
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; ++t) {
    size_t b = count * t / nt, e = count * (t + 1) / nt;
    pool.emplace_back(AoRange, verts, faces, ntris, offset, seed, first + b, e - b, out + b);
  }
  for (auto &th : pool) th.join();
}

namespace {

// core/Affine.h:41-48: length over all four components, the fourth component passed through undivided.
void Normalize4(const float in[4], float out[4])
{
  const float len = std::sqrt(in[0] * in[0] + in[1] * in[1] + in[2] * in[2] + in[3] * in[3]);
  out[0] = in[0] / len;
  out[1] = in[1] / len;
  out[2] = in[2] / len;
  out[3] = in[3];
}

// core/Affine.h:165-192.  `out` arrives zeroed (the caller's `float roty[4][4]{ 0 }`, img_main.cpp:629-630).  The
// second assignment to out[1][0] -- meant for out[1][3] -- zeroes the element the line three above it computed;
// it is restated, not repaired: the matrices the reference uploads have that zero in them.
void Rotate3(float out[4][4], const float vec[4], float angle)
{
  const float cos_a = static_cast<float>(std::cos(static_cast<double>(angle)));
  const float sin_a = static_cast<float>(std::sin(static_cast<double>(angle)));
  float v[4];
  Normalize4(vec, v);
  out[0][0] = cos_a + v[0] * v[0] * (1 - cos_a);
  out[0][1] = v[0] * v[1] * (1 - cos_a) - v[2] * sin_a;
  out[0][2] = v[0] * v[2] * (1 - cos_a) + v[1] * sin_a;
  out[0][3] = 0;
  out[1][0] = v[0] * v[1] * (1 - cos_a) + v[2] * sin_a;
  out[1][1] = cos_a + v[1] * v[1] * (1 - cos_a);
  out[1][2] = v[1] * v[2] * (1 - cos_a) - v[0] * sin_a;
  out[1][0] = 0;
  out[2][0] = v[0] * v[2] * (1 - cos_a) - v[1] * sin_a;
  out[2][1] = v[1] * v[2] * (1 - cos_a) + v[0] * sin_a;
  out[2][2] = cos_a + v[2] * v[2] * (1 - cos_a);
  out[2][3] = 0;
  out[3][0] = 0;
  out[3][1] = 0;
  out[3][2] = 0;
  out[3][3] = 1;
}

// core/Affine.h:194-211 for 4x4 operands: out[j][i] = sum_k a[k][i] * b[j][k], accumulated from zero in k order.
void Multiply4(const float a[4][4], const float b[4][4], float out[4][4])
{
  float tmp_m[4][4];
  for (unsigned i = 0; i < 4; ++i)
    for (unsigned j = 0; j < 4; ++j) {
      float tmp = 0;
      for (unsigned k = 0; k < 4; ++k) tmp += a[k][i] * b[j][k];
      tmp_m[j][i] = tmp;
    }
  std::memcpy(out, tmp_m, sizeof tmp_m);
}

}  // namespace

void CameraMatrix(float rot_x, float rot_y, const float trans_xyz[3], float out[16])
{
  const float axisy[4] = {0, 1, 0, 0}, axisx[4] = {1, 0, 0, 0};
  float roty[4][4] = {{0}}, rotx[4][4] = {{0}}, rotation[4][4], cam[4][4];
  Rotate3(roty, axisy, rot_y);
  Rotate3(rotx, axisx, rot_x);
  Multiply4(roty, rotx, rotation);
  const float trans[4][4] = {{1, 0, 0, trans_xyz[0]}, {0, 1, 0, trans_xyz[1]}, {0, 0, 1, trans_xyz[2]}, {0, 0, 0, 1}};
  Multiply4(trans, rotation, cam);
  std::memcpy(out, cam, sizeof cam);
}

}  // namespace rt
