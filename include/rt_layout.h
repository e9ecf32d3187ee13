/*
 * rt_layout.h -- plain-C record layouts shared by the host library, the HIP
 * kernels, the oracle and every caller of the C ABI.
 *
 * These are byte-for-byte the layouts the reference shares between its host
 * code and its device kernel:
 *   rt_vec3            core/raydata.h:40-50   (xyz + pad, 16 B)
 *   rt_ray             core/raydata.h:56-60   (32 B)
 *   rt_vec2            core/raydata.h:62-66   (8 B)
 *   rt_ray_isect       core/raydata.h:70-78   (RayIntersection, 64 B)
 *   rt_aabb            core/raydata.h:87-94   (32 B)
 *   rt_face            core/ObjLoader.cpp:145,171-174,193-195 (cl_uint8, 32 B)
 *   rt_material        core/MtlLib.h:7-15     (48 B)
 *   rt_bvh_node        core/CentroidBVH.h:13-59 (CentroidBVHNode, 80 B)
 *
 * No OpenCL headers are needed: cl_float4 / cl_uint8 are restated as PODs of
 * the same size and alignment (16 B and 32 B).
 */
#ifndef RT_LAYOUT_H
#define RT_LAYOUT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#define RT_ALIGN(n) alignas(n)
extern "C" {
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#define RT_ALIGN(n) _Alignas(n)
#endif

typedef struct rt_vec3 { float x, y, z, w; } rt_vec3;           /* w is padding (loader writes 1 for vertices) */
typedef struct rt_float4 { RT_ALIGN(16) float x; float y, z, w; } rt_float4;
typedef struct rt_vec2 { float u, v; } rt_vec2;
typedef struct rt_ray { rt_vec3 origin; rt_vec3 dir; } rt_ray;

/* Ray-in AND hit-out record.  On output triangle_id is the hit triangle
 * (0xFFFFFFFF = miss) and t_bits carries the IEEE-754 bit pattern of the hit
 * distance t (0x7F7FFFFF = FLT_MAX on a miss); the reference leaves both
 * unwritten (render_client/tracekernel.ck:286, core/raydata.h:77). */
typedef struct rt_ray_isect {
  rt_float4 origin;       /* @0  */
  rt_float4 direction;    /* @16 */
  rt_float4 normal;       /* @32 */
  rt_vec2   uv;           /* @48 */
  uint32_t  triangle_id;  /* @56 */
  uint32_t  t_bits;       /* @60  (reference: pad word p1) */
} rt_ray_isect;

typedef struct rt_aabb { rt_vec3 a; rt_vec3 b; } rt_aabb;       /* a = min, b = max */

/* {v0, v1, v2, material, n0, n1, n2, 0} */
typedef struct rt_face { RT_ALIGN(32) uint32_t s[8]; } rt_face;

typedef struct rt_material {
  rt_float4 color;        /* cl_float3 == 16 B */
  rt_float4 uv;
  float     Si;           /* index of refraction */
  float     Rf;           /* "rough": used as the layer alpha */
  int32_t   texId;
  int32_t   padding;
} rt_material;

#define RT_TRIS_PER_LEAF 4
enum { RT_BVH_INNER = 0, RT_BVH_LEAF = 1 };

typedef struct rt_bvh_leaf  { uint32_t tris[RT_TRIS_PER_LEAF]; uint32_t count; uint32_t pad[3]; } rt_bvh_leaf;
typedef struct rt_bvh_inner { rt_float4 plane_norm; uint32_t children_index[2]; uint32_t pad[2]; } rt_bvh_inner;
typedef struct rt_bvh_node {
  union { rt_bvh_leaf leaf; rt_bvh_inner inner; } node;   /* @0, 32 B */
  rt_aabb  aabb;          /* @32 */
  uint32_t type;          /* @64 */
  uint32_t parent_index;  /* @68 */
  uint32_t padding[2];
} rt_bvh_node;

/* Compact per-ray result used by the framebuffer gather and the tests. */
typedef struct rt_hit16 { uint32_t triangle_id; float t, u, v; } rt_hit16;

#define RT_MISS_ID 0xFFFFFFFFu
#define RT_MISS_T_BITS 0x7F7FFFFFu

RT_STATIC_ASSERT(sizeof(rt_vec3) == 16, "vec3 is 16 B");
RT_STATIC_ASSERT(sizeof(rt_ray) == 32, "Ray is 32 B");
RT_STATIC_ASSERT(sizeof(rt_vec2) == 8, "vec2 is 8 B");
RT_STATIC_ASSERT(sizeof(rt_ray_isect) == 64, "RayIntersection is 64 B");
RT_STATIC_ASSERT(offsetof(rt_ray_isect, direction) == 16, "direction @16");
RT_STATIC_ASSERT(offsetof(rt_ray_isect, normal) == 32, "normal @32");
RT_STATIC_ASSERT(offsetof(rt_ray_isect, uv) == 48, "uv @48");
RT_STATIC_ASSERT(offsetof(rt_ray_isect, triangle_id) == 56, "triangle_id @56");
RT_STATIC_ASSERT(offsetof(rt_ray_isect, t_bits) == 60, "p1 @60");
RT_STATIC_ASSERT(sizeof(rt_aabb) == 32, "AABB is 32 B");
RT_STATIC_ASSERT(sizeof(rt_face) == 32, "cl_uint8 is 32 B");
RT_STATIC_ASSERT(sizeof(rt_material) == 48, "Material is 48 B");
RT_STATIC_ASSERT(sizeof(rt_bvh_node) == 80, "CentroidBVHNode is 80 B");
RT_STATIC_ASSERT(offsetof(rt_bvh_node, aabb) == 32, "aabb @32");
RT_STATIC_ASSERT(offsetof(rt_bvh_node, type) == 64, "type @64");
RT_STATIC_ASSERT(offsetof(rt_bvh_node, parent_index) == 68, "parent @68");
RT_STATIC_ASSERT(offsetof(rt_bvh_leaf, count) == 16, "leaf.count @16");
RT_STATIC_ASSERT(offsetof(rt_bvh_inner, children_index) == 16, "inner.children_index @16");
RT_STATIC_ASSERT(sizeof(rt_hit16) == 16, "compact hit is 16 B");

#ifdef __cplusplus
}
#endif
#endif /* RT_LAYOUT_H */
