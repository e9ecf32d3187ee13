/*
 * oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the reference's closest-hit arithmetic.  It is
 * the checker the GPU path is compared against and the CPU baseline the
 * benchmark times; nothing in the shipped library calls or links it.
 *
 * Parity pin: every function here is checked bit-for-bit against the
 * reference's own core/ray_tri.cpp and core/raymath.cpp compiled from
 * /root/reference into oracle/_ref (tests/test_oracle.py, random pairs
 * plus edge cases) and against the one known-answer test the reference holds
 * (cpu_raycast/raycast.cpp:12-19).
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native, no -ffast-math): every
 * float operation is a single correctly-rounded IEEE-754 binary32 operation
 * in source order, which is what the reference computes when built for x86-64.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_layout.h"

/* core/raydata.h:100 -- a DOUBLE literal; comparisons against it promote the
 * float operand to double. */
#define O_EPSILON (0.000001)
/* core/raydata.h:129 */
#define O_NEAR_ZERO(a) ((a) > -O_EPSILON && (a) < O_EPSILON)
/* core/raydata.h:104-117: SUB, DOT ((x*x + y*y) + z*z, left to right), CROSS */
#define O_SUB(a, b, o) do { (o).x = (a).x - (b).x; (o).y = (a).y - (b).y; (o).z = (a).z - (b).z; } while (0)
#define O_DOT(a, b) ((a).x * (b).x + (a).y * (b).y + (a).z * (b).z)
#define O_CROSS(a, b, o) do { \
  (o).x = (a).y * (b).z - (a).z * (b).y; \
  (o).y = (a).z * (b).x - (a).x * (b).z; \
  (o).z = (a).x * (b).y - (a).y * (b).x; } while (0)

/* core/ray_tri.cpp:6-53 -- Moller-Trumbore, two-sided, t > EPSILON. */
int oracle_intersect_ray_triangle(const rt_vec3 *a, const rt_vec3 *b, const rt_vec3 *c,
                                  const rt_ray *r, rt_vec2 *uvout, float *tout)
{
  rt_vec3 e1, e2, P, Q, T;
  rt_vec2 uv;
  float det, det_inv, t;

  O_SUB(*b, *a, e1);
  O_SUB(*c, *a, e2);
  O_CROSS(r->dir, e2, P);

  det = O_DOT(e1, P);
  if (O_NEAR_ZERO(det))
    return 0;

  det_inv = 1.f / det;

  O_SUB(r->origin, *a, T);

  uv.u = O_DOT(T, P) * det_inv;
  if (uv.u < 0.f || uv.u > 1.f)
    return 0;

  O_CROSS(T, e1, Q);

  uv.v = O_DOT(r->dir, Q) * det_inv;
  if (uv.v < 0.f || uv.v + uv.u > 1.f)
    return 0;

  t = O_DOT(e2, Q) * det_inv;
  if (t > O_EPSILON) {
    *tout = t;
    *uvout = uv;
    return 1;
  }
  return 0;
}

/* core/ray_tri.cpp:57-78 -- slab test with macro min/max (NaN-propagating
 * the way the ternaries fall); does not report tmin. */
#define O_MIN(a, b) ((a) < (b) ? (a) : (b))
#define O_MAX(a, b) ((a) > (b) ? (a) : (b))
int oracle_intersect_ray_aabb(const rt_aabb *aabb, const rt_ray *ray)
{
  const float *lo = &aabb->a.x, *hi = &aabb->b.x, *o = &ray->origin.x, *d = &ray->dir.x;
  float dir_inv[3] = { 1 / d[0], 1 / d[1], 1 / d[2] };
  float t1 = (lo[0] - o[0]) * dir_inv[0];
  float t2 = (hi[0] - o[0]) * dir_inv[0];
  float tmin = O_MIN(t1, t2);
  float tmax = O_MAX(t1, t2);
  for (int i = 1; i < 3; ++i) {
    t1 = (lo[i] - o[i]) * dir_inv[i];
    t2 = (hi[i] - o[i]) * dir_inv[i];
    tmin = O_MAX(tmin, O_MIN(t1, t2));
    tmax = O_MIN(tmax, O_MAX(t1, t2));
  }
  return tmax > O_MAX(tmin, 0);
}

/* core/raymath.cpp:6-15 -- float dot, double sqrt rounded to float, float divides. */
rt_vec3 oracle_normalize(rt_vec3 in)
{
  float lensq = O_DOT(in, in);
  float len = sqrt(lensq);
  rt_vec3 ret;
  ret.x = in.x / len;
  ret.y = in.y / len;
  ret.z = in.z / len;
  ret.w = 0.f;
  return ret;
}

/* render_client/img_main.cpp:253-267 generalised from IMG_X x IMG_Y to W x H.
 * The reference indexes [i + j*IMG_Y], identical to [i + j*W] for its square
 * image; W is used here so rows do not overlap when W != H.  `origin` replaces
 * the constant (0,0,-5,1). */
void oracle_primary_rays(int W, int H, const float origin[3], rt_ray_isect *out)
{
  rt_vec3 o = { W / 2.f, H / 2.f, (float)(-5 * W), 0.f };
  for (int i = 0; i < W; ++i) {
    for (int j = 0; j < H; ++j) {
      rt_vec3 end = { (float)i, (float)j, 0.f, 0.f }, d;
      O_SUB(end, o, d);
      d.w = 0.f;
      d = oracle_normalize(d);
      rt_ray_isect *r = &out[(size_t)i + (size_t)j * W];
      memset(r, 0, sizeof *r);
      r->direction.x = d.x; r->direction.y = d.y; r->direction.z = d.z; r->direction.w = 0.f;
      r->origin.x = origin[0]; r->origin.y = origin[1]; r->origin.z = origin[2]; r->origin.w = 1.f;
      r->normal.x = 0.f; r->normal.y = 0.f; r->normal.z = 1.f; r->normal.w = 0.f;
      r->triangle_id = 0xFFFFFFFFu;
    }
  }
}

/* Brute-force closest hit, the loop shape of cpu_raycast/raycast.cpp:113-122:
 * every triangle in index order, strict `dist < closest`, so the LOWEST index
 * wins an exact tie.  closest starts at FLT_MAX (the kernel's MAXFLOAT,
 * render_client/tracekernel.ck:278) rather than main2's 99999.
 *
 * Also records the second-smallest t over all OTHER triangles so callers can
 * count near-tie rays (|t2 - t1| <= eps * t1). */
typedef struct {
  const rt_vec3 *verts; const rt_face *faces; size_t ntris;
  const rt_ray_isect *rays; size_t begin, end;
  rt_hit16 *hits; float *second_t;
} bf_job;

static void *bf_worker(void *arg)
{
  bf_job *j = (bf_job *)arg;
  for (size_t k = j->begin; k < j->end; ++k) {
    rt_ray ray;
    ray.origin.x = j->rays[k].origin.x; ray.origin.y = j->rays[k].origin.y; ray.origin.z = j->rays[k].origin.z; ray.origin.w = 0.f;
    ray.dir.x = j->rays[k].direction.x; ray.dir.y = j->rays[k].direction.y; ray.dir.z = j->rays[k].direction.z; ray.dir.w = 0.f;
    float closest = 3.402823466e+38f, second = 3.402823466e+38f;
    rt_hit16 h = { RT_MISS_ID, closest, 0.f, 0.f };
    for (size_t mi = 0; mi < j->ntris; ++mi) {
      const uint32_t *s = j->faces[mi].s;
      rt_vec2 uv; float dist;
      if (oracle_intersect_ray_triangle(&j->verts[s[0]], &j->verts[s[1]], &j->verts[s[2]], &ray, &uv, &dist)) {
        if (dist < closest) {
          second = closest;
          closest = dist; h.triangle_id = (uint32_t)mi; h.t = dist; h.u = uv.u; h.v = uv.v;
        } else if (dist < second) {
          second = dist;
        }
      }
    }
    j->hits[k] = h;
    if (j->second_t) j->second_t[k] = second;
  }
  return NULL;
}

/* hits[k] = closest hit of rays[k] (origin.xyz, direction.xyz used as given;
 * no camera matrix).  second_t may be NULL.  nthreads <= 0 -> 1. */
int oracle_closest_hit(const rt_vec3 *verts, const rt_face *faces, size_t ntris,
                       const rt_ray_isect *rays, size_t nrays,
                       rt_hit16 *hits, float *second_t, int nthreads)
{
  if (nthreads <= 0) nthreads = 1;
  if ((size_t)nthreads > nrays) nthreads = nrays ? (int)nrays : 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  bf_job *jobs = (bf_job *)malloc(sizeof(bf_job) * nthreads);
  if (!th || !jobs) { free(th); free(jobs); return -1; }
  for (int i = 0; i < nthreads; ++i) {
    bf_job jb = { verts, faces, ntris, rays, nrays * i / nthreads, nrays * (i + 1) / nthreads, hits, second_t };
    jobs[i] = jb;
    if (i > 0) pthread_create(&th[i], NULL, bf_worker, &jobs[i]);
  }
  bf_worker(&jobs[0]);
  for (int i = 1; i < nthreads; ++i) pthread_join(th[i], NULL);
  free(th); free(jobs);
  return 0;
}

/* The same closest hit found through the reference's own tree (core/CentroidBVH.h:52-59 nodes, root at 0): a plain
 * stack walk, children tested with a slab test whose box is pushed outward by 2^-17 of (largest |origin| + largest
 * |box coordinate|) so that rounding can never cull a box that holds an accepted hit, boxes entered when their entry
 * distance is <= the best hit so far (inclusive), and on equal t the lower triangle index kept: the answer does not
 * depend on the order of the walk and equals the brute-force loop's, which tests/test_oracle.py asserts.  The
 * reference has no CPU traversal (its cpu_raycast is the brute-force loop); this is the "accelerated CPU" figure
 * beside it, timed by bench.py, and a second opinion on the kernels' answers that shares no traversal code with them. */
typedef struct {
  const rt_vec3 *verts; const rt_face *faces; const rt_bvh_node *nodes; size_t nnodes;
  const rt_ray_isect *rays; size_t begin, end;
  rt_hit16 *hits; int failed;
} bvh_job;

static int padded_slab(const rt_aabb *bb, const float o[3], const float inv[3], float pad, float closest, float *tenter)
{
  const float *lo = &bb->a.x, *hi = &bb->b.x;
  float tmin = 0.f, tmax = closest;
  for (int i = 0; i < 3; ++i) {
    float t1 = ((lo[i] - pad) - o[i]) * inv[i], t2 = ((hi[i] + pad) - o[i]) * inv[i];
    if (t1 != t1 || t2 != t2) continue;   /* 0 * inf: the origin lies on a padded plane of a flat direction: no constraint */
    float a = t1 < t2 ? t1 : t2, b = t1 < t2 ? t2 : t1;
    a -= fabsf(a) * 1e-6f; b += fabsf(b) * 1e-6f;   /* the two roundings of each product */
    if (a > tmin) tmin = a;
    if (b < tmax) tmax = b;
  }
  *tenter = tmin;
  return tmin <= tmax;
}

static void *bvh_worker(void *arg)
{
  bvh_job *j = (bvh_job *)arg;
  uint32_t stack[256];
  float scale = 0.f;
  const float *rb = &j->nodes[0].aabb.a.x;
  for (int i = 0; i < 8; ++i) if (i != 3 && i != 7 && fabsf(rb[i]) > scale) scale = fabsf(rb[i]);
  for (size_t k = j->begin; k < j->end; ++k) {
    rt_ray ray;
    ray.origin.x = j->rays[k].origin.x; ray.origin.y = j->rays[k].origin.y; ray.origin.z = j->rays[k].origin.z; ray.origin.w = 0.f;
    ray.dir.x = j->rays[k].direction.x; ray.dir.y = j->rays[k].direction.y; ray.dir.z = j->rays[k].direction.z; ray.dir.w = 0.f;
    const float o[3] = { ray.origin.x, ray.origin.y, ray.origin.z };
    const float inv[3] = { 1.f / ray.dir.x, 1.f / ray.dir.y, 1.f / ray.dir.z };
    float omax = fabsf(o[0]) > fabsf(o[1]) ? fabsf(o[0]) : fabsf(o[1]);
    if (fabsf(o[2]) > omax) omax = fabsf(o[2]);
    const float pad = 7.62939453125e-06f * (omax + scale);
    float closest = 3.402823466e+38f;
    rt_hit16 h = { RT_MISS_ID, closest, 0.f, 0.f };
    int sp = 0;
    float te;
    if (padded_slab(&j->nodes[0].aabb, o, inv, pad, closest, &te)) stack[sp++] = 0;
    while (sp) {
      const rt_bvh_node *n = &j->nodes[stack[--sp]];
      if (n->type == RT_BVH_LEAF) {
        for (uint32_t q = 0; q < n->node.leaf.count && q < RT_TRIS_PER_LEAF; ++q) {
          const uint32_t mi = n->node.leaf.tris[q];
          const uint32_t *s = j->faces[mi].s;
          rt_vec2 uv; float dist;
          if (oracle_intersect_ray_triangle(&j->verts[s[0]], &j->verts[s[1]], &j->verts[s[2]], &ray, &uv, &dist) &&
              (dist < closest || (dist == closest && mi < h.triangle_id))) {
            closest = dist; h.triangle_id = mi; h.t = dist; h.u = uv.u; h.v = uv.v;
          }
        }
        continue;
      }
      const uint32_t c0 = n->node.inner.children_index[0], c1 = n->node.inner.children_index[1];
      if (c0 >= j->nnodes || c1 >= j->nnodes || sp + 2 > 256) { j->failed = 1; break; }
      float t0, t1;
      const int h0 = padded_slab(&j->nodes[c0].aabb, o, inv, pad, closest, &t0);
      const int h1 = padded_slab(&j->nodes[c1].aabb, o, inv, pad, closest, &t1);
      if (h0 && h1) {   /* nearer child on top */
        if (t1 < t0) { stack[sp++] = c0; stack[sp++] = c1; } else { stack[sp++] = c1; stack[sp++] = c0; }
      } else if (h0) stack[sp++] = c0;
      else if (h1) stack[sp++] = c1;
    }
    j->hits[k] = h;
  }
  return NULL;
}

/* hits[k] = closest hit of rays[k] through the tree `nodes`; returns 0, or -1 if the tree is malformed or deeper than
 * the walk's 256-entry stack.  nthreads <= 0 -> 1. */
int oracle_closest_hit_bvh(const rt_vec3 *verts, const rt_face *faces, const rt_bvh_node *nodes, size_t nnodes,
                           const rt_ray_isect *rays, size_t nrays, rt_hit16 *hits, int nthreads)
{
  if (nnodes == 0) return -1;
  if (nthreads <= 0) nthreads = 1;
  if ((size_t)nthreads > nrays) nthreads = nrays ? (int)nrays : 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  bvh_job *jobs = (bvh_job *)malloc(sizeof(bvh_job) * nthreads);
  if (!th || !jobs) { free(th); free(jobs); return -1; }
  for (int i = 0; i < nthreads; ++i) {
    bvh_job jb = { verts, faces, nodes, nnodes, rays, nrays * i / nthreads, nrays * (i + 1) / nthreads, hits, 0 };
    jobs[i] = jb;
    if (i > 0) pthread_create(&th[i], NULL, bvh_worker, &jobs[i]);
  }
  bvh_worker(&jobs[0]);
  int failed = jobs[0].failed;
  for (int i = 1; i < nthreads; ++i) { pthread_join(th[i], NULL); failed |= jobs[i].failed; }
  free(th); free(jobs);
  return failed ? -1 : 0;
}

/* Batch form of the single test, for the bit-equality sweeps:
 * tri[i] = 3 x rt_vec3, ray[i]; out_hit[i], out_t[i], out_uv[i]. */
void oracle_tri_batch(const rt_vec3 *tri, const rt_ray *ray, size_t n, int32_t *out_hit, float *out_t, rt_vec2 *out_uv)
{
  for (size_t i = 0; i < n; ++i) {
    float t = -1.f; rt_vec2 uv = { 0.f, 0.f };
    out_hit[i] = oracle_intersect_ray_triangle(&tri[3 * i], &tri[3 * i + 1], &tri[3 * i + 2], &ray[i], &uv, &t);
    out_t[i] = t; out_uv[i] = uv;
  }
}

void oracle_aabb_batch(const rt_aabb *bb, const rt_ray *ray, size_t n, int32_t *out_hit)
{
  for (size_t i = 0; i < n; ++i) out_hit[i] = oracle_intersect_ray_aabb(&bb[i], &ray[i]);
}

/* Op-1 ray regeneration, render_client/tracekernel.ck:101-104,227-233, in
 * left-to-right unfused float arithmetic: dir = v - (2*dot4(v,n))*n with a
 * 4-component dot, origin += n * 0.00001f.  OpenCL leaves the rounding of its
 * dot() builtin to the implementation, so this ordering is the definition the
 * GPU path is held to, not a reference-pinned one. */
void oracle_reflect_ray(const rt_ray_isect *in, rt_ray *out)
{
  const rt_float4 v = in->direction, n = in->normal;
  float d = v.x * n.x + v.y * n.y + v.z * n.z + v.w * n.w;
  float s = 2.0f * d;
  out->dir.x = v.x - s * n.x; out->dir.y = v.y - s * n.y; out->dir.z = v.z - s * n.z; out->dir.w = 0.f;
  out->origin.x = in->origin.x + n.x * 0.00001f;
  out->origin.y = in->origin.y + n.y * 0.00001f;
  out->origin.z = in->origin.z + n.z * 0.00001f;
  out->origin.w = 0.f;
}

/* ---- the `trace` kernel's per-ray setup and hit epilogue, restated -----------------------------------------
 * render_client/tracekernel.ck:224-265 (setup) and :282-286, :373-413 (output record, pixel), in unfused
 * float arithmetic, every sum left to right as written there: mult_m_f4 (:182-190) is four-term dots of the
 * matrix rows with (x,y,z,w); reflect (:101-104) is v - 2*dot4(v,n)*n; the hit point is
 * (1-u-v)*V[s0] + u*V[s1] + v*V[s2] (:378) and the normal u*N[s4] + v*N[s5] + (1-u-v)*N[s6] (:376).
 * tracekernel.ck is OpenCL C and cannot run in this container, and OpenCL leaves the rounding and order of its
 * dot() builtin to the implementation, so no reference-derived vector pins these: PARITY UNPINNED for the
 * interpolated outputs and the colour.  What this restatement fixes is one definite evaluation order that the
 * HIP path is held to bit for bit. */
void oracle_trace_setup(const rt_ray_isect *in, const float cam[16], unsigned op, rt_ray *out)
{
  if (op == 1u) {
    oracle_reflect_ray(in, out);
    return;
  }
  const float o[4] = { in->origin.x, in->origin.y, in->origin.z, 1.f };
  const float d[4] = { in->direction.x, in->direction.y, in->direction.z, 0.f };
  float ro[3], rd[3];
  for (int r = 0; r < 3; ++r) {
    const float *m = cam + 4 * r;
    ro[r] = m[0] * o[0] + m[1] * o[1] + m[2] * o[2] + m[3] * o[3];
    rd[r] = m[0] * d[0] + m[1] * d[1] + m[2] * d[2] + m[3] * d[3];
  }
  out->origin.x = ro[0]; out->origin.y = ro[1]; out->origin.z = ro[2]; out->origin.w = 0.f;
  out->dir.x = rd[0]; out->dir.y = rd[1]; out->dir.z = rd[2]; out->dir.w = 0.f;
}

/* Output record and RGBA pixel of one ray.  `ray` is the ray that was traced (after setup), `hit` its closest
 * hit.  Miss: origin and normal := the input record's origin and direction, uv = 0, id = ~0 (:282-286), pixel
 * (0,0,0,0); `direction` is left unwritten by the reference and carries the input direction here.  Returns 0 when
 * the face names a normal index outside `normals` (the reference reads out of bounds there; the HIP path
 * substitutes a geometric normal, which this restatement does not define), else 1. */
int oracle_trace_epilogue(const rt_vec3 *verts, const rt_face *faces, const rt_float4 *normals, size_t nnormals,
                          const rt_material *mats, size_t nmats, const rt_ray_isect *in, const rt_ray *ray,
                          const rt_hit16 *hit, rt_ray_isect *out, float rgba[4])
{
  memset(out, 0, sizeof *out);
  out->origin = in->origin;
  out->direction = in->direction;
  out->normal = in->direction;
  out->uv.u = 0.f; out->uv.v = 0.f;
  out->triangle_id = RT_MISS_ID;
  memcpy(&out->t_bits, &hit->t, sizeof(float));
  rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.f;
  if (hit->triangle_id == RT_MISS_ID) return 1;
  const uint32_t *s = faces[hit->triangle_id].s;
  if (s[4] >= nnormals || s[5] >= nnormals || s[6] >= nnormals) return 0;
  const float u = hit->u, v = hit->v, w = 1.f - u - v;
  const rt_vec3 *v0 = &verts[s[0]], *v1 = &verts[s[1]], *v2 = &verts[s[2]];
  const rt_float4 *n0 = &normals[s[4]], *n1 = &normals[s[5]], *n2 = &normals[s[6]];
  out->origin.x = w * v0->x + u * v1->x + v * v2->x;
  out->origin.y = w * v0->y + u * v1->y + v * v2->y;
  out->origin.z = w * v0->z + u * v1->z + v * v2->z;
  out->origin.w = 1.f;
  const float nx = u * n0->x + v * n1->x + w * n2->x;
  const float ny = u * n0->y + v * n1->y + w * n2->y;
  const float nz = u * n0->z + v * n1->z + w * n2->z;
  out->normal.x = nx; out->normal.y = ny; out->normal.z = nz; out->normal.w = 0.f;
  out->direction.x = ray->dir.x; out->direction.y = ray->dir.y; out->direction.z = ray->dir.z; out->direction.w = 0.f;
  out->uv.u = u; out->uv.v = v;
  out->triangle_id = hit->triangle_id;
  if (nmats) {
    const rt_material *m = &mats[s[3] < nmats ? s[3] : 0u];
    const float k = fabsf(nx * ray->dir.x + ny * ray->dir.y + nz * ray->dir.z);
    rgba[0] = k * m->color.x; rgba[1] = k * m->color.y; rgba[2] = k * m->color.z; rgba[3] = m->Rf;
  }
  return 1;
}

/* Batch form: n records; rays_out[i] = the traced ray, rec_out[i], rgba_out[4*i..], defined[i] as returned above. */
void oracle_trace_batch(const rt_vec3 *verts, const rt_face *faces, const rt_float4 *normals, size_t nnormals,
                        const rt_material *mats, size_t nmats, const rt_ray_isect *in, size_t n, const float cam[16],
                        unsigned op, const rt_hit16 *hits, rt_ray_isect *rec_out, float *rgba_out, int32_t *defined)
{
  for (size_t i = 0; i < n; ++i) {
    rt_ray ray;
    oracle_trace_setup(&in[i], cam, op, &ray);
    defined[i] = oracle_trace_epilogue(verts, faces, normals, nnormals, mats, nmats, &in[i], &ray, &hits[i], &rec_out[i], rgba_out + 4 * i);
  }
}

/* rays_out[i] (as a 64-byte record: origin, direction) = the ray the kernel traces for record i. */
void oracle_trace_setup_batch(const rt_ray_isect *in, size_t n, const float cam[16], unsigned op, rt_ray_isect *rays_out)
{
  for (size_t i = 0; i < n; ++i) {
    rt_ray ray;
    oracle_trace_setup(&in[i], cam, op, &ray);
    memset(&rays_out[i], 0, sizeof rays_out[i]);
    rays_out[i].origin.x = ray.origin.x; rays_out[i].origin.y = ray.origin.y; rays_out[i].origin.z = ray.origin.z;
    rays_out[i].direction.x = ray.dir.x; rays_out[i].direction.y = ray.dir.y; rays_out[i].direction.z = ray.dir.z;
  }
}
