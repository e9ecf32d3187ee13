// device_scene.h -- kernel argument block and launch entry points shared by the
// kernels (trace_kernels.hip) and the C ABI implementation (capi.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rt_layout.h"
#include "trace_math.h"

namespace rt {

constexpr uint32_t kLeafBitDev = 0x80000000u;
constexpr int kModeIntersect = 0;  // rays used as given (ray_scene_intersection)
constexpr int kFbBuckets = 32;     // cost bins of the launch-to-launch feedback schedule
constexpr int kModeTrace = 1;      // full `trace` step: camera / reflection setup + pixel
constexpr int kModePrimary = 2;    // rays of the reference's primary table, generated in the kernel (no ray array)
constexpr uint32_t kDeepEntries = 56;   // traversal-stack entries per lane beyond the LDS part, in private memory
#ifndef RT_REFILL_WAVES_PER_SIMD
#define RT_REFILL_WAVES_PER_SIMD 6
#endif
#ifndef RT_REFILL_CHUNK
#define RT_REFILL_CHUNK 128
#endif
constexpr uint32_t kRefillWavesPerSimd = RT_REFILL_WAVES_PER_SIMD;   // resident waves per SIMD of the lane-refilling kernel (80 VGPRs)
constexpr uint32_t kRefillChunk = RT_REFILL_CHUNK;                   // consecutive rays a wave claims per atomic
constexpr uint32_t kMinLdsEntries = 8;  // the output stage exchanges 64 x 64-byte records through the LDS part

struct KernelArgs {
  const float4 *nodes;        // float boxes: 4 (two-wide) or 8 (four-wide) float4 per record;
                              // quantized boxes: one 16-byte child per float4 slot, 2 or 4 per record
  const float4 *tris;         // 3 float4 per record
  const rt_ray_isect *rays;
  rt_ray_isect *hits;         // nullable
  rt_hit16 *hits16;           // nullable
  const rt_vec3 *verts;
  const rt_face *faces;
  const rt_float4 *normals;
  const rt_material *mats;
  float *rgba;                // nullable, kModeTrace only
  const uint32_t *order;      // nullable, image mode: tile handled by the k-th wave-sized work item
  // Cost feedback between consecutive launches of the same shape (nullable): every wave stores the number of
  // traversal iterations its work unit needed; a small sort kernel turns that into the next launch's order.
  uint32_t *unit_cost;
  uint32_t *cursor;           // nullable: persistent launch, waves take dispatch-order positions from this counter
  unsigned long long *stats;  // instrumented build only: nodes, tris, leaves, rays hit
  unsigned long long *timeline;  // instrumented build only, nullable: per wave {start, end (100 MHz ticks), hw id, xcc id}
  uint32_t n;
  uint32_t ntasks;            // wave-sized work items: ceil(n/64), or the number of 8x8 tiles
  uint32_t split_units;       // the first this-many units of `order` are traced by four waves each, a quarter of the unit's rays per wave
  uint32_t coop;              // 1: the quarter waves of split units work four lanes per ray (four-wide quantized records; the tree's
                              // worst-case stack must fit 4 * stack_entries, the LDS columns of a quad)
  uint32_t compact_words;     // what `hits16` receives per ray: 4 = {id, t, u, v} (also 0), 2 = {id, t}, 1 = {id}
  uint32_t pixels8;           // kModeTrace, 1: `rgba` receives 8-bit BGRA pixels (4 bytes each) instead of RGBA32F
  uint32_t refill_min;        // RefillHitKernel: idle lanes of a wave that trigger handing out new rays
  uint32_t prio_slots;        // the first this-many dispatch slots run at raised issue priority (0 = none)
  uint32_t cost_radius;       // a unit's cost is also recorded for its neighbours within this many units (<= 3)
  uint32_t cost_share;        // ... scaled by this many 128ths
  uint32_t nnormals;
  uint32_t nmats;
  uint32_t stack_entries;     // traversal-stack entries per lane held in LDS (8 bytes each), >= kMinLdsEntries
  uint32_t op;                // kModeTrace: 0 = camera transform, 1 = reflect
  uint32_t img_w;             // > 0: rays are a row-major img_w x img_h image, walked in 8x8 pixel tiles per wave
  uint32_t img_h;
  uint32_t tiles_x;           // image: ceil(img_w / 8), and the multiplier that divides a tile index by it:
  uint32_t tiles_x_magic;     //   floor(2^32 / tiles_x) + 1 (mulhi gives the quotient or one more); unused when tiles_x == 1
  uint32_t waves_per_block;   // set by LaunchClosestHit: block / 64
  uint32_t leaf_pct;          // switch from node steps to leaf steps once this percentage of busy lanes hold a leaf
  float scale;                // max |coordinate| of the scene box (slab padding)
  float cam[16];              // kModeTrace: camera matrix; kModePrimary: cam[0..2] = ray origin
  QuantGrid grid;             // quantized layouts only
};

// done (nullable): event signalled by the kernel's own completion (no separate record packet behind the launch)
hipError_t LaunchClosestHit(const KernelArgs &a, int width, bool quant, int mode, bool count, uint32_t grid,
                            uint32_t block, hipStream_t stream, hipEvent_t done = nullptr);
// Fixed grid of waves that stream through a linear ray set, refilling idle lanes (see RefillHitKernel); a.cursor must be a
// zeroed word, the layout four-wide quantized.
hipError_t LaunchRefill(const KernelArgs &a, uint32_t grid, uint32_t block, hipStream_t stream);
// order[k] = k-th unit to start, costliest bin first, index order within a bin (three small launches on `stream`).
// unit_cost must hold OrderPaddedUnits(nunits) entries and is zeroed after being read; bins is OrderBinWords(nunits) words
// of work space.
uint32_t OrderPaddedUnits(uint32_t nunits);
uint32_t OrderBinWords(uint32_t nunits);
hipError_t LaunchBuildOrder(uint32_t *unit_cost, uint32_t nunits, uint32_t *bins, uint32_t *order, hipStream_t stream);
// dst[pixel] = src[record] for the listed tiles (see ScatterTilesKernel); record_bytes is a multiple of 4.
hipError_t LaunchScatterTiles(const void *src, uint32_t src_records, const uint32_t *tile_ids, const uint32_t *tile_first, uint32_t ntiles,
                              uint32_t tile, uint32_t img_w, uint32_t img_h, uint32_t record_bytes, void *dst, hipStream_t stream);
// bgra[i] = 8-bit BGRA of the RGBA32F pixel rgba[4i..4i+3] (see PackPixelsKernel).
hipError_t LaunchPackPixels(const float *rgba, uint32_t n, void *bgra, hipStream_t stream);
hipError_t LaunchTriBatch(const rt_vec3 *tri, const rt_ray *ray, uint32_t n, int32_t *hit, float *t, rt_vec2 *uv,
                          hipStream_t stream);

}  // namespace rt
