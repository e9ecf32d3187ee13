/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" doorways onto the
 * reference's own functions so Python (ctypes) can call them.  It contains no
 * algorithm: every call forwards to code compiled from /root/reference/core
 * (see Makefile).  Built only where /root/reference exists; the product never
 * links it.
 */
#include "raydata.h"
#include "ray_tri.h"
#include "GeomUtil.h"
#include "CentroidBVH.h"
#include "MtlLib.h"
#include <cstring>
#include <cstddef>

static_assert(sizeof(vec3) == 16 && sizeof(Ray) == 32 && sizeof(RayIntersection) == 64, "layout");
static_assert(sizeof(CentroidBVHNode) == 80 && sizeof(Material) == 48 && sizeof(cl_uint8) == 32, "layout");

extern "C" {

int ref_intersect_ray_triangle(const vec3 *a, const vec3 *b, const vec3 *c, const Ray *r, vec2 *uv, float *t)
{ return intersect_ray_triangle(a, b, c, r, uv, t); }

int ref_intersect_ray_aabb(const AABB *bb, const Ray *r)
{ float t = 0; return intersect_ray_aabb(bb, r, &t); }

void ref_tri_batch(const vec3 *tri, const Ray *ray, size_t n, int *out_hit, float *out_t, vec2 *out_uv)
{
  for (size_t i = 0; i < n; ++i) {
    float t = -1.f; vec2 uv = { 0.f, 0.f };
    out_hit[i] = intersect_ray_triangle(&tri[3 * i], &tri[3 * i + 1], &tri[3 * i + 2], &ray[i], &uv, &t);
    out_t[i] = t; out_uv[i] = uv;
  }
}

void ref_aabb_batch(const AABB *bb, const Ray *ray, size_t n, int *out_hit)
{ for (size_t i = 0; i < n; ++i) { float t = 0; out_hit[i] = intersect_ray_aabb(&bb[i], &ray[i], &t); } }

void ref_normalize(const vec3 *in, vec3 *out) { *out = normalize(*in); }

/* Builds with the reference's CentroidBVHBuild and copies the node array out
 * (nodes may be NULL to query the count).  Returns the node count; stats =
 * {depth, tricount, leafcount}. */
size_t ref_bvh_build(const cl_uint8 *tris, const vec3 *verts, size_t ntris, size_t nverts,
                     CentroidBVHNode *nodes, size_t capacity, int *stats)
{
  CentroidBVH bvh;
  CentroidBVHInit(&bvh);
  CentroidBVHBuild(&bvh, tris, verts, ntris, nverts);
  size_t n = bvh.listSize;
  if (nodes && n <= capacity) std::memcpy(nodes, bvh.nodeList, n * sizeof(CentroidBVHNode));
  if (stats) { stats[0] = bvh.depth; stats[1] = bvh.tricount; stats[2] = bvh.leafcount; }
  CentroidBVHFree(&bvh);
  return n;
}

/* Index-order brute force (the loop shape of cpu_raycast/raycast.cpp:113-122) whose every
 * triangle test is the reference's intersect_ray_triangle.  hits[k] = {id, t, u, v}. */
void ref_closest_hit(const vec3 *verts, const cl_uint8 *faces, size_t ntris, const RayIntersection *rays, size_t n,
                     float *hits)
{
  for (size_t k = 0; k < n; ++k) {
    Ray ray;
    ray.origin.x = rays[k].origin.s[0]; ray.origin.y = rays[k].origin.s[1]; ray.origin.z = rays[k].origin.s[2]; ray.origin.w = 0;
    ray.dir.x = rays[k].direction.s[0]; ray.dir.y = rays[k].direction.s[1]; ray.dir.z = rays[k].direction.s[2]; ray.dir.w = 0;
    float closest = 3.402823466e+38f, bu = 0, bv = 0;
    unsigned best = 0xFFFFFFFFu;
    for (size_t mi = 0; mi < ntris; ++mi) {
      vec2 uv; float dist;
      if (intersect_ray_triangle(&verts[faces[mi].s[0]], &verts[faces[mi].s[1]], &verts[faces[mi].s[2]], &ray, &uv, &dist)) {
        if (dist < closest) { closest = dist; best = (unsigned)mi; bu = uv.u; bv = uv.v; }
      }
    }
    std::memcpy(&hits[4 * k], &best, 4);
    hits[4 * k + 1] = closest; hits[4 * k + 2] = bu; hits[4 * k + 3] = bv;
  }
}

void ref_min_size_aabb(AABB *bb) { MinSizeAABB(bb); }
void ref_centroid(const vec3 *tri3, vec3 *c) { Centroid(reinterpret_cast<Tri const *>(tri3), c); }
int ref_point_in_aabb(const vec3 *p, const AABB *bb) { return PointInAABB(p, bb) ? 1 : 0; }

}

// ---- core/Affine.h (header-only): the camera matrix as render_client/img_main.cpp:627-660 builds it ------------
#include "Affine.h"
extern "C" void ref_camera_matrix(float rot_x, float rot_y, const float trans_xyz[3], float out[16])
{
  float axisy[4]{ 0, 1, 0, 0 };
  float axisx[4]{ 1, 0, 0, 0 };
  float roty[4][4]{ 0 };
  float rotx[4][4]{ 0 };
  fmat4x4 rotation, cam_matrix;
  rotate3(roty, axisy, rot_y);
  rotate3(rotx, axisx, rot_x);
  multiply(roty, rotx, rotation);
  float trans[4][4] = {
    1, 0, 0, trans_xyz[0],
    0, 1, 0, trans_xyz[1],
    0, 0, 1, trans_xyz[2],
    0, 0, 0, 1
  };
  multiply(trans, rotation, cam_matrix);
  memcpy(out, cam_matrix, sizeof cam_matrix);
}

// ---- cpu_raycast/bmp.c (plain C, compiled where it lies): printImage into a file opened in binary mode ---------
typedef unsigned char ref_pixel[4];
extern "C" void printImage(ref_pixel *data, int width, int height, FILE *file);
extern "C" int ref_print_image(const unsigned char *bgra, int width, int height, const char *path)
{
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  printImage((ref_pixel *)bgra, width, height, f);
  fclose(f);
  return 0;
}
