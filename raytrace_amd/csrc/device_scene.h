// device_scene.h -- kernel argument block and launch entry points shared by the
// kernels (trace_kernels.hip) and the C ABI implementation (capi.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rt_layout.h"

namespace rt {

constexpr uint32_t kLeafBitDev = 0x80000000u;
constexpr int kModeIntersect = 0;  // rays used as given (ray_scene_intersection)
constexpr int kModeTrace = 1;      // full `trace` step: camera / reflection setup + pixel

struct KernelArgs {
  const float4 *nodes;        // two-wide: 4 float4 per record; four-wide: 8 float4 per record
  const float4 *tris;         // 3 float4 per record
  const rt_ray_isect *rays;
  rt_ray_isect *hits;         // nullable
  rt_hit16 *hits16;           // nullable
  const rt_vec3 *verts;
  const rt_face *faces;
  const rt_float4 *normals;
  const rt_material *mats;
  float *rgba;                // nullable, kModeTrace only
  uint32_t *counter;          // persistent-wave ray cursor (zeroed before every launch)
  unsigned long long *stats;  // instrumented build only: nodes, tris, leaves, rays hit
  uint32_t n;
  uint32_t nnormals;
  uint32_t nmats;
  uint32_t stack_entries;     // LDS stack entries per lane
  uint32_t op;                // kModeTrace: 0 = camera transform, 1 = reflect
  float scale;                // max |coordinate| of the scene box (slab padding)
  float cam[16];
};

hipError_t LaunchClosestHit(const KernelArgs &a, int width, int mode, bool persistent, bool count, uint32_t grid,
                            hipStream_t stream);
hipError_t LaunchTriBatch(const rt_vec3 *tri, const rt_ray *ray, uint32_t n, int32_t *hit, float *t, rt_vec2 *uv,
                          hipStream_t stream);

}  // namespace rt
