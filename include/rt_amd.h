/*
 * rt_amd.h -- C ABI of librtamd.so: the MI355X (gfx950) closest-hit ray/scene
 * intersection path and the host-side scene front end that feeds it.
 *
 * Conventions: every function returns 0 on success and a negative rt_status on
 * failure; rt_last_error() returns the message of the calling thread's most
 * recent failure.  Nothing aborts and no C++ exception crosses the boundary
 * (the reference prints OpenCL error codes and carries on,
 * core/clerrorcheck.cpp:11-18).  One rt_scene may be used by one host thread
 * at a time; distinct scenes are independent.  Record layouts: rt_layout.h.
 *
 * Each entry point names the reference interface it stands in for.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include "rt_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,     /* bad argument / malformed scene */
  RT_ERR_IO = -2,          /* file could not be read or written */
  RT_ERR_DEVICE = -3,      /* HIP runtime error or no usable GPU */
  RT_ERR_NOMEM = -4
} rt_status;

const char *rt_last_error(void);
/* Number of visible HIP devices; <= 0 with rt_last_error() set if the runtime is unusable. */
int rt_device_count(void);

/* ------------------------------------------------------------------ host: scene load
 * Stands in for LoadObj / LoadObjMaterials (core/ObjLoader.h:12-14) and MtlLib
 * (core/MtlLib.h:23-30).  The all-zero "SKY_DEFAULT" material is inserted first, as
 * render_client/img_main.cpp:240 does, so material 0 is the sky. */
#define RT_LOADER_REFERENCE 0u  /* the reference tokenizer: repeated spaces produce empty tokens */
#define RT_LOADER_TOLERANT  1u  /* empty tokens dropped; otherwise identical */
#define RT_LOADER_EXTENDED  2u  /* tolerant, and quads / n-gons become triangle fans (the reference drops them,
                                  core/ObjLoader.cpp:151-154) and negative indices count from the end */
typedef struct rt_mesh rt_mesh;
int rt_mesh_load_obj(const char *path, unsigned loader, rt_mesh **out);
/* Wraps caller arrays (copied).  normals/materials may be NULL with a zero count. */
int rt_mesh_from_arrays(const rt_vec3 *verts, size_t nverts, const rt_face *faces, size_t ntris,
                        const rt_float4 *normals, size_t nnormals, const rt_material *materials, size_t nmats,
                        rt_mesh **out);
size_t rt_mesh_nverts(const rt_mesh *m);
size_t rt_mesh_ntris(const rt_mesh *m);
size_t rt_mesh_nnormals(const rt_mesh *m);
size_t rt_mesh_nmaterials(const rt_mesh *m);
const rt_vec3 *rt_mesh_verts(const rt_mesh *m);
const rt_face *rt_mesh_faces(const rt_mesh *m);
const rt_float4 *rt_mesh_normals(const rt_mesh *m);
const rt_material *rt_mesh_materials(const rt_mesh *m);
/* `vt` records (core/ObjLoader.cpp:96-118: u, v, optional w in x, y, z) and, three per face, the 0-based `vt` index
 * of each corner -- the indices the reference parses no further (its lines :178-186 are commented out);
 * 0xFFFFFFFF where a corner names none.  rt_mesh_face_uvws is NULL for meshes made by rt_mesh_from_arrays. */
size_t rt_mesh_nuvws(const rt_mesh *m);
const rt_vec3 *rt_mesh_uvws(const rt_mesh *m);
const uint32_t *rt_mesh_face_uvws(const rt_mesh *m);
/* Index of a material name, -1 if absent (does not insert). */
int rt_mesh_material_index(const rt_mesh *m, const char *name);
/* Loader log lines ("Quads and ngons not supported, ignoring." ...), newline-joined. */
const char *rt_mesh_log(const rt_mesh *m);
void rt_mesh_destroy(rt_mesh *m);

/* ------------------------------------------------------------------ host: BVH build
 * Stands in for CentroidBVHInit / CentroidBVHBuild / CentroidBVHFree
 * (core/CentroidBVH.h:90-94).  Node array: root at 0, pre-order, 80-byte records. */
#define RT_BVH_REFERENCE 0u  /* centroid of triangles[i]      (core/CentroidBVH.cpp:65 as written) */
#define RT_BVH_FIXED     1u  /* centroid of triangles[set[i]] */
#define RT_BVH_SAH       2u  /* same node format, splits chosen by the binned surface-area heuristic:
                               identical hits, about a quarter fewer boxes and triangles tested per ray */
typedef struct rt_bvh rt_bvh;
int rt_bvh_build(const rt_face *faces, const rt_vec3 *verts, size_t ntris, size_t nverts, unsigned mode, rt_bvh **out);
const rt_bvh_node *rt_bvh_nodes(const rt_bvh *b);
size_t rt_bvh_nnodes(const rt_bvh *b);
int rt_bvh_depth(const rt_bvh *b);       /* CentroidBVH::depth     */
int rt_bvh_leafcount(const rt_bvh *b);   /* CentroidBVH::leafcount */
int rt_bvh_fallback_splits(const rt_bvh *b);
void rt_bvh_destroy(rt_bvh *b);

/* ------------------------------------------------------------------ host: rays
 * rt_primary_rays: the camera table of render_client/img_main.cpp:253-267 for a W x H
 * image (record i + j*W), every ray starting at `origin`.
 * rt_frame_camera: an origin on the -z side from which that pinhole frames `root`.
 * rt_ao_rays: `count` seeded incoherent rays numbered first..first+count-1 (no reference
 * counterpart; the divergence stress input). */
int rt_primary_rays(int W, int H, const float origin[3], rt_ray_isect *out);
/* The camera matrix of the reference's frame loop (render_client/img_main.cpp:627-660): rotate3 about y by rot_y
 * and about x by rot_x (core/Affine.h:165-192, its out[1][0] = 0 slip included), multiply (core/Affine.h:194-211),
 * then the translation; rot_x / rot_y are its cursor / 50 angles.  out[16] is what rt_trace_pass takes as cam. */
int rt_camera_matrix(float rot_x, float rot_y, const float trans[3], float out[16]);
int rt_frame_camera(const rt_aabb *root, int W, int H, float origin_out[3]);
int rt_ao_rays(const rt_mesh *m, const rt_aabb *root, uint64_t seed, size_t first, size_t count, rt_ray_isect *out);

/* ------------------------------------------------------------------ device: scene
 * Stands in for the clCreateBuffer(COPY_HOST_PTR) uploads and static clSetKernelArg
 * bindings of render_client/img_main.cpp:243-249,293,298-311.  Inputs are copied
 * (the caller keeps ownership); the BVH is validated and re-flattened.  `device` is a
 * HIP device ordinal. */
#define RT_DEVICE_NONE (-1) /* host-only scene: validation, rt_scene_info and rt_scene_count_fetches work;
                               every tracing entry point fails with RT_ERR_DEVICE */
#define RT_SCENE_WIDE2 0u   /* traverse the two-child records (32 bytes quantized, 64 bytes with float boxes)   */
#define RT_SCENE_WIDE4 4u   /* traverse the four-child records (64 bytes quantized, 128 bytes with float boxes): the faster layout */
#define RT_SCENE_FLATTEN_DEVICE 0x100u /* OR-ed in: build the traversal records from the uploaded 80-byte nodes with GPU kernels
                                          (the host then only validates and numbers the tree); the records are bit-identical */
typedef struct rt_scene rt_scene;
int rt_scene_create(const rt_vec3 *verts, size_t nverts, const rt_face *faces, size_t ntris,
                    const rt_float4 *normals, size_t nnormals, const rt_material *materials, size_t nmats,
                    const rt_bvh_node *bvh_nodes, size_t nnodes, int device, unsigned flags, rt_scene **out);
void rt_scene_destroy(rt_scene *s);
/* info[0..7] = {bvh2 nodes, bvh4 nodes, triangles in leaves, bvh2 depth, bvh4 depth,
 *               bytes per bvh2 node, bytes per bvh4 node, bytes per triangle record}
 * (node sizes are those of the layout currently selected by "quantized") */
int rt_scene_info(const rt_scene *s, uint64_t info[8]);
/* The re-flattened records as the scene holds them (host copies): which = 0 two-wide float records (64 B), 1 two-wide
 * quantized (2 x 16 B per node), 2 four-wide float (128 B), 3 four-wide quantized (4 x 16 B), 4 triangle records
 * (48 B, leaf order).  out == NULL only returns the size in *bytes.  For inspection and tests. */
int rt_scene_records(const rt_scene *s, int which, void *out, size_t capacity, size_t *bytes);
/* Tuning knobs (none of them changes a result, only the schedule or the record layout walked):
 *   "width"          2 or 4: which record layout the traversal walks.
 *   "quantized"      1: records hold 16-bit grid-quantized boxes, 16 bytes per child; 0: float boxes, 32 bytes
 *                    per child.
 *   "image_width"    W > 0 declares that ray arrays whose length is a multiple of W are row-major W-wide images,
 *                    as the reference's 2-D NDRange is (render_client/img_main.cpp:457-461); each wave then
 *                    takes an 8x8 pixel tile.  0 = unstructured: each wave takes 64 consecutive records.
 *   "centre_first"   image mode, while no cost-sorted order exists (the first two launches of a shape, or "feedback"
 *                    0): 1 dispatches the tiles nearest the image centre first, 0 in row order.
 *   "feedback"       1 (default): launches record the traversal cost of each wave-sized work unit; small sort kernels
 *                    turn that into a costliest-first dispatch order, which later launches of the same shape (ray
 *                    count, image width) on this scene use.  A schedule only: every ray is traced on every launch.
 *                    Any change of shape starts over.
 *   "feedback_period" 1..1024 (default 8, with "feedback"): every this-many launches one records its costs and a
 *                    sort follows it; the launches between follow the newest finished order unchanged (a sort costs
 *                    the stream that traces about 13 us at 1080p: two small launches).  1 = every launch.
 *   "side_stream"    0 (default): the sort kernels run on the launch's own stream, behind the launch that recorded the
 *                    costs; the next launch already follows the new order.  1: on a library-owned non-blocking stream of
 *                    default priority, beside the next launch (ordered by events; the order is used from the launch
 *                    after next on).  Measured slower for one stream of launches (every launch starts ~5 us later with
 *                    the second queue in play, more than the sort costs); offered for callers who time it otherwise.
 *   "feedback_radius" 0..3, "feedback_share" 0..128 (with "feedback"): a unit's measured cost also counts, scaled
 *                    by share/128, for the units within `radius` of it (tiles in both image directions), so
 *                    that the order still starts the right tiles first when the camera moved between the
 *                    launches.  Defaults 1 and 96; 0 and anything = exact per-unit costs, best for repeated rays.
 *   "refill"         0 (default) or 1..64, for ray sets without an image hint of at least 65 536 rays on the four-wide
 *                    quantized layout: a fixed grid of waves streams through the rays in index order, and whenever
 *                    this many lanes of a wave are idle they store their finished hits and take the wave's next
 *                    rays, while the other lanes carry on.  For incoherent rays (AO-style: +34 % with 64-byte
 *                    records, +50 % with 16-byte hits at 40); coherent rays lose their shared paths with it (a
 *                    1080p primary table without the image hint runs 2.4x slower), hence off unless asked for.
 *   "split_units"    0..65536 (default 128, with "feedback"): a launch ends with its longest wave, and a wave is as
 *                    long as the traversal steps its rays need between them; this many units at the head of the
 *                    costliest-first order are traced by four waves each (4x4 pixels or 16 records per wave) instead
 *                    of one.  Capped at the launch's unit count; 0 = never.
 *   "split_max_units" 0..16777216 (default 40960): launches of more work units than this are not split.  A launch is mostly
 *                    tail while it is small -- a 1080p frame (32 400 units) gains 3-14 %, a rank's share of a sharded frame up
 *                    to 40 % -- and from about 2560x1440 (57 600 units) on the extra waves cost more than the tail (-1..-4 %).
 *   "trace_split"    -1 (default), 0 or 1: whether the full trace step (rt_trace_pass*) splits its costliest units as well.
 *                    A small launch -- one pass of a 1080p frame, one rank's share of a sharded frame -- is mostly tail and
 *                    gains 3-40 % from it; a launch that fills the GPU many times over, or one of several frames in flight,
 *                    only pays for the extra waves (-5 %).  -1 splits by size like the intersection launches ("split_max_units"),
 *                    1 whatever the size, 0 never.
 *   "split_coop"     1 (default): those quarter waves work four lanes per ray -- one child box of the ray's node or one
 *                    triangle of its leaf per lane, the quad agreeing on the nearest child / closest hit -- on the
 *                    four-wide quantized layout, when the tree's worst-case stack fits the LDS columns of a quad.
 *                    0: one ray per lane, 16 lanes busy.
 *   "leaf_pct"       1..100: a wave switches from node steps to leaf steps once this share of its busy lanes
 *                    holds a leaf; 100 = descend until every lane does.
 *   "block_threads"  64 (default), 128 or 256.
 *   "waves_per_cu"   0..64: cap on resident waves per CU, held by sizing the traversal stack in LDS (default 0 =
 *                    whatever registers and the tree depth allow).
 *   "lds_stack"      8..32 (default 10): traversal-stack entries per lane kept in LDS (8 bytes each); deeper entries
 *                    spill to the lane's private memory.  Rays of a framed object rarely go deeper than 11.
 *   "persistent"     1: instead of one wave per work unit, "persistent_waves" (1..64, default 24) waves per CU
 *                    stay resident and take units from a global cursor, one wave-level atomic per unit, in
 *                    dispatch order.  0 (default): the hardware dispatcher hands units out (measured faster).
 *   "extra_stack"    diagnostic: extra LDS traversal-stack entries per lane (lowers occupancy).
 * Calls on one scene must not overlap in time from different host threads (the scene caches launch state);
 * the launches themselves may go to any streams. */
int rt_scene_set_param(rt_scene *s, const char *key, int value);

/* ------------------------------------------------------------------ device: the hot path
 * ray_scene_intersection: closest hit of rays[i] = (origin.xyz, direction.xyz as given)
 * against the scene; hits[i] is a RayIntersection filled as the `trace` kernel fills its
 * output record (render_client/tracekernel.ck:282-286 on a miss, :375-399 on a hit) plus
 * triangle_id and t_bits.  The factoring is intersect_ray_scene of
 * render_client/ray_scene_intersection.ckm:157-250; the launch it replaces is
 * clEnqueueNDRangeKernel + clFinish, render_client/img_main.cpp:457-480.  rays and hits
 * may each be host or device memory (device memory must belong to the scene's device)
 * and must not overlap.  Returns when the hits are complete. */
int ray_scene_intersection(const void *rays, size_t n, const rt_scene *scene, void *hits);

/* Same computation, device pointers only, enqueued on `stream` (a hipStream_t; NULL =
 * default stream) without waiting.  hits16 (optional, may be NULL) receives the compact
 * {triangle_id, t, u, v} record; hits (optional) the 64-byte record.  At least one output. */
int rt_intersect_async(const rt_scene *scene, const void *rays_dev, size_t n, void *hits_dev, void *hits16_dev,
                       void *stream);

/* rt_intersect_async with only a compact record per ray, of a chosen size: record_bytes 16 = {triangle_id, t, u, v},
 * 8 = {triangle_id, t}, 4 = {triangle_id}; out_dev holds n of them.  The smaller records are what a frame of hits costs
 * on the links between GPUs (rt_gather_scatter); the values are the same bits as the 16-byte record's.  No counterpart
 * in the reference, whose kernel always writes the whole RayIntersection (render_client/tracekernel.ck:373-399). */
int rt_intersect_compact(const rt_scene *scene, const void *rays_dev, size_t n, void *out_dev, size_t record_bytes,
                         void *stream);

/* rt_intersect_async for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  It computes the same
 * records as rt_intersect_async and always runs the cost-feedback schedule, but keeps the schedule's tables
 * (per-unit cost, dispatch order, sort counters) in a fixed device array of 2 MiB + 32 KiB inside the library
 * instead of hipMalloc'd buffers: after rt_scene_create the call performs no hipMalloc / hipFree and creates no
 * stream or event; the order kernels run on `stream` behind the trace kernel of every "feedback_period"-th call.
 * Limit: n <= 16 Mi rays (2^18 wave-sized work units) per call, larger n fails with RT_ERR_INVALID.  Every call
 * on one GPU, whatever the scene, shares that GPU's array: calls that overlap in time on one GPU race on the
 * schedule tables (the hits stay correct only if the calls are ordered, e.g. on one stream); the tables keep the
 * previous call's contents and are cleared on `stream` when the shape (ray count, image width) changes. */
int rt_intersect_static(const rt_scene *scene, const void *rays_dev, size_t n, void *hits_dev, void *hits16_dev,
                        void *stream);

/* The same closest-hit query for the reference's primary-ray table (render_client/img_main.cpp:253-267) without
 * the table: ray (i, j) of a width x height image is generated inside the kernel, bit-identical to what
 * rt_primary_rays writes, so no 64-byte ray record is read.  Outputs as rt_intersect_async, record index
 * i + j * width.  Enqueued on `stream` without waiting. */
int rt_trace_primary(const rt_scene *scene, int width, int height, const float origin[3], void *hits_dev,
                     void *hits16_dev, void *stream);

/* The full `trace` kernel step (render_client/tracekernel.ck:192-414): op 0 applies the
 * camera matrix cam[16] (row-major, rows dotted with (x,y,z,w)) to origin (w=1) and
 * direction (w=0); op 1 reflects direction about the record's normal and nudges the
 * origin by normal*1e-5.  Writes the output record and, if rgba is non-NULL, the RGBA32F
 * pixel |n.d|*material.color, alpha = material.Rf ((0,0,0,0) on a miss).  Device pointers,
 * enqueued on `stream`.
 * One deliberate difference from the reference in op 1: a record whose previous pass missed (triangle_id = ~0) is
 * not traced again -- it passes through unchanged with a (0,0,0,0) pixel.  The reference has no such test: it
 * reflects every record, and since a miss stores normal := direction (tracekernel.ck:282-283) and leaves
 * `direction` unwritten, it sends those rays straight back where they came from. */
int rt_trace_pass(const rt_scene *scene, const void *rays_in_dev, void *rays_out_dev, float *rgba_dev, size_t n,
                  const float cam[16], unsigned trace_operation, void *stream);

/* rt_trace_pass with the pixel written as 8-bit BGRA (4 bytes, the byte order of rt_pack_bgra8 and of the reference's
 * bitmap writer, cpu_raycast/bmp.c:49-53) instead of RGBA32F: the frame leaves the kernel in the form a framebuffer or
 * the links between GPUs carry, without a packing pass.  bgra_dev holds n pixels; same bits as rt_pack_bgra8 of
 * rt_trace_pass's pixels. */
int rt_trace_pass_bgra8(const rt_scene *scene, const void *rays_in_dev, void *rays_out_dev, void *bgra_dev, size_t n,
                        const float cam[16], unsigned trace_operation, void *stream);

/* Mean records fetched per ray by the traversal, counted on the host over rays
 * 0, stride, 2*stride, ... : counts[0..5] = {node records, triangle records, non-empty
 * leaves, rays that hit, rays counted, deepest traversal stack (entries) of any counted ray}.  per_ray (nullable, 2 x rays-counted uint32) receives each counted
 * ray's own {node records, triangle records}.  A measurement aid; computes no hits. */
int rt_scene_count_fetches(const rt_scene *s, const rt_ray_isect *rays_host, size_t n, size_t stride,
                           uint64_t counts[6], uint32_t *per_ray);
/* Planning aid: a host-side lockstep model of one 64-lane wave running the traversal (a phase lasts as
 * long as its slowest lane).  Samples `max_waves` waves (0 = all) over the ray stream and returns
 * out[0..15] = {waves, rays, wave node steps, useful lane node steps, wave triangle steps, useful lane
 * triangle steps, wave leaf phases, lanes in leaf phases, refill events, lanes refilled, steps of the longest
 * wave, node steps in which all lanes visit one node, loop iterations (each exposes one memory latency), iterations
 * of the longest wave, node steps whose lanes visit exactly two different nodes, ... exactly three}.  policy 0 = descend until every lane holds a leaf, then test leaves;
 * 1 = one node step and one leaf test per iteration; 2..100 = leave the node loop once that percentage of the busy
 * lanes holds a leaf (the kernel's "leaf_pct"); -k = fused iterations: the lanes at inner nodes take a node step while,
 * in the same iteration, the lanes at leaves test up to k triangles.  refill = 0
 * models one batch of 64 rays per wave; refill = k models a wave that takes new rays for its idle
 * lanes whenever k or more lanes have finished.  Computes no hits. */
int rt_scene_wave_model(const rt_scene *s, const rt_ray_isect *rays_host, size_t n, unsigned image_width, unsigned refill,
                        size_t max_waves, int policy, uint64_t out[16]);
/* The same counters taken on the device by an instrumented build of the kernel over all n
 * rays (device pointer): counts[0..3] as above, counts[4] = n, counts[5] = the deepest traversal stack
 * (entries) any ray reached. */
int rt_scene_count_fetches_device(const rt_scene *s, const void *rays_dev, size_t n, uint64_t counts[6]);

/* Traversal cost of every wave-sized work unit (8x8 pixel tile with the image hint, else 64 consecutive records) of
 * one launch over the given rays: costs_host[k] = loop iterations the wave that traced unit k needed -- the number the
 * cost-feedback schedule sorts by.  The multi-GPU tile partition balances ranks with it.  costs_host == NULL only
 * returns the unit count in *nunits.  The _primary form traces the reference's primary-ray table of a width x height
 * image from `origin` (units = 8x8 tiles, row-major) without needing the table in memory. */
int rt_scene_unit_costs(const rt_scene *s, const void *rays_dev, size_t n, uint32_t *costs_host, size_t capacity, size_t *nunits);
int rt_scene_unit_costs_primary(const rt_scene *s, int width, int height, const float origin[3], uint32_t *costs_host,
                                size_t capacity, size_t *nunits);

/* Framebuffer assembly for image-tile sharding (the reference's only hint of tiling is the commented-out four-offset
 * NDRange of render_client/img_main.cpp:464-478; it renders straight into one GL texture).  src holds records tile by
 * tile: tile i of the list is square number tile_ids[i] (row-major over the image cut into tile x tile squares, edge
 * squares smaller), its records row-major inside the square, starting at record tile_first[i].  Each record
 * (record_bytes, a multiple of 4: 16-byte hits, 64-byte records, 4-byte pixels) is copied to its pixel of the row-major width x height image dst.  src holds
 * src_records records; a list entry that names no square of the image, or whose records would run past the end of
 * src, is skipped.  All pointers are device memory of `device`; enqueued on `stream`. */
int rt_scatter_tiles(int device, const void *src_dev, size_t src_records, const uint32_t *tile_ids_dev,
                     const uint32_t *tile_first_dev, size_t ntiles, int tile, int width, int height, size_t record_bytes,
                     void *dst_dev, void *stream);

/* RGBA32F pixels (what rt_trace_pass writes) -> 8-bit BGRA in the byte order rt_write_bmp32 and the reference's bitmap
 * writer take (4-byte `pixel`s written as they are, cpu_raycast/bmp.h:12, bmp.c:49-53; a 32-bit BMP stores B, G, R, A):
 * channel = min(255, max(0, x * 255 + 0.5)) truncated, alpha 255.  A quarter of
 * the bytes when a framebuffer has to cross GPUs.  Device pointers of `device`, enqueued on `stream`. */
int rt_pack_bgra8(int device, const float *rgba_dev, size_t n, void *bgra_dev, void *stream);

/* 16-byte {id, t, u, v} hits -> 8-byte {id, t} records: what a framebuffer of hits needs on the wire between GPUs (the
 * barycentrics follow from the ray and t wherever the frame is shaded).  No counterpart in the reference.  Device pointers
 * of `device`, enqueued on `stream`. */
int rt_pack_hits8(int device, const void *hits16_dev, size_t n, void *hits8_dev, void *stream);

/* The reference's display pass on the device, so that a frame leaves the GPU once, as 8-bit pixels: layer 0 (the primary
 * pass's RGBA32F pixels) drawn opaque, layer 1 (the reflection pass's; NULL = none) blended over it with
 * glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_DST_ALPHA) (render_client/img_main.cpp:434,512-526):
 * out = layer1 * layer1.a + layer0 * (1 - layer0.a) per channel.  Writes the blended RGBA32F pixels (rgba_out_dev, nullable)
 * and/or their 8-bit BGRA form as rt_pack_bgra8 makes it (bgra_out_dev, nullable).  Enqueued on `stream`. */
int rt_composite_layers(int device, const float *layer0_rgba_dev, const float *layer1_rgba_dev, size_t n, float *rgba_out_dev,
                        void *bgra_out_dev, void *stream);

/* ------------------------------------------------------------------ one frame on several GPUs (one process per GPU)
 * The reference renders one image on one device; its only hint of tiling is the commented-out four-offset NDRange of
 * render_client/img_main.cpp:464-478.  Here the image is cut into tile x tile squares (edge squares smaller, numbered
 * row-major), every rank traces the squares dealt to it and the records are gathered into one rank's framebuffer.
 *
 * rt_shard_plan deals the squares to `nranks` ranks.  Costs, at most one of: unit_costs = the per-8x8-tile cost table of
 * rt_scene_unit_costs_primary for this image ((ceil(height/8) x ceil(width/8)), row-major; tile must then be a multiple of
 * 8); square_costs = one cost per square (tiles_x * tiles_y, row-major), e.g. what rt_shard_square_costs summed from every
 * rank's own last launch.  Squares go costliest first to the rank with the least cost so far, so the ranks finish together
 * whatever the silhouette looks like; the tables are deterministic, so every rank derives the same plan from the same
 * numbers.  With neither the squares are dealt round-robin.  rank_head_start (nullable, nranks values >= 0): the share of
 * the WHOLE frame's cost a rank counts as carrying before the deal -- for the gathering rank, the time of its gather and
 * scatter over the time one GPU needs for the frame -- so that it is dealt that much less.
 * Outputs: tile_order[tiles_x * tiles_y] = the squares rank after rank, each rank's in the order dealt (costliest first);
 * rank_first_tile[nranks + 1] = where each rank's squares start in tile_order; rank_first_record[nranks + 1] = where each
 * rank's records (pixels, square after square, row-major inside a square) start in the gather buffer. */
int rt_shard_plan(int width, int height, int tile, int nranks, const uint32_t *unit_costs, const double *square_costs,
                  const double *rank_head_start, uint32_t *tile_order, uint64_t *rank_first_tile, uint64_t *rank_first_record);
/* Re-dealing while the job runs (a camera that moves): every rank measures the work-unit costs of its OWN rays
 * (rt_scene_unit_costs over its nrecords rays, with the scene's "image_width" as it traces them) and this adds them up per
 * square: a unit counts for the listed square that holds its first record (tile_ids / tile_first: the rank's squares and
 * where each one's records start among the rank's, in record order).  square_costs[nsquares] is indexed by square number
 * and only added to: zero it, let every rank add its own, sum the array over the ranks (rt_comm_sum_f64) and hand it to
 * rt_shard_plan on every rank. */
int rt_shard_square_costs(const uint32_t *unit_costs, size_t nunits, int image_width, size_t nrecords, const uint32_t *tile_ids,
                          const uint32_t *tile_first, size_t ntiles, double *square_costs, size_t nsquares);
/* For a list of squares: ray_index[] (nullable) = the row-major pixel / ray index (i + j * width) of every record, square
 * after square; tile_first[i] (nullable) = first_record + position of square i's first record -- the table
 * rt_scatter_tiles takes. */
int rt_shard_ray_indices(int width, int height, int tile, const uint32_t *tile_ids, size_t ntiles, uint32_t first_record,
                         uint32_t *ray_index, uint32_t *tile_first);

/* Communicator of the job: a thin face of RCCL (ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy), loaded on first
 * use.  One rank makes the id, hands its RT_COMM_ID_BYTES bytes to the others by any means, every rank calls
 * rt_comm_create with its own device current; nranks = 1 is valid.  *comm_out is an ncclComm_t: a caller that already
 * holds one passes that to rt_gather_scatter instead. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(uint8_t id_out[RT_COMM_ID_BYTES]);
int rt_comm_create(const uint8_t id[RT_COMM_ID_BYTES], int nranks, int rank, int device, void **comm_out);
void rt_comm_destroy(void *comm);

/* values[i] := sum over the ranks of values[i] (host memory; ncclAllReduce on a stream of its own, waited for).  A collective:
 * every rank calls it with the same n.  nranks = 1 needs no communicator (the values are left alone; with one, the call
 * still goes through RCCL). */
int rt_comm_sum_f64(void *comm, int nranks, int device, double *values, size_t n);

/* Health check of a communicator, a collective (every rank calls it): each rank sends 4 KB that name it to the next rank
 * and receives the previous rank's in one grouped ncclSend/ncclRecv (one rank exchanges with itself), the words are
 * compared on the host, and {1, rank} is summed over the ranks with rt_comm_sum_f64.  Exercises every RCCL call the
 * exchange step uses; a job runs it once after rt_comm_create, before the first frame.  No counterpart in the reference
 * (which renders on one device). */
int rt_comm_check(void *comm, int nranks, int rank, int device);

/* The exchange step of a sharded frame, enqueued on `stream` of every rank (call it on all ranks, like a collective):
 * every rank but `root` sends its records (mine_dev, rank_first_record[rank + 1] - rank_first_record[rank] of them,
 * record_bytes each) to root; root receives rank r's into gather_dev + rank_first_record[r] * record_bytes -- one grouped
 * RCCL call, a send/receive pair per peer, so root's xGMI links all carry their share at once (point to point, no ring)
 * -- and then scatters the square-major gather buffer into the row-major framebuffer with rt_scatter_tiles (tile_ids_dev /
 * tile_first_dev: all ranks' squares as rt_shard_plan / rt_shard_ray_indices list them, root only).  Root's own records
 * may already sit in its slice of gather_dev (mine_dev pointing there: no copy).  nranks = 1 moves nothing between GPUs
 * and needs no communicator.  Replaces nothing in the reference (it has no multi-device path). */
int rt_gather_scatter(void *comm, int rank, int nranks, int root, int device, const void *mine_dev, void *gather_dev,
                      const uint64_t *rank_first_record, size_t record_bytes, const uint32_t *tile_ids_dev, const uint32_t *tile_first_dev,
                      size_t ntiles, int tile, int width, int height, void *framebuffer_dev, void *stream);

/* The calls above as one object per rank: the frame loop of render_client/img_main.cpp:447-498 (one enqueue on one
 * device per frame) for a job of one process per GPU.  rt_tiled_create takes the frame's ray table (width x height records,
 * host memory, copied: e.g. rt_primary_rays, the reference's camera table), deals the squares round-robin, uploads this
 * rank's rays (its squares one after the other) and sets the scene's "image_width" to the squares' width -- give the object
 * a scene of its own (`device` is that scene's device).  rt_tiled_render enqueues one frame on `stream` of every rank: the full trace step over this rank's
 * rays (rt_trace_pass_bgra8, operation 0 with `cam`) and rt_gather_scatter; rank 0's framebuffer_dev (width x height 8-bit
 * BGRA pixels, row-major; ignored on the other ranks) holds the frame when the stream gets there.  One frame at a time
 * per object (frames on different streams would share its buffers): for frames in flight make one object, with a scene of
 * its own, per slot.  rt_tiled_redeal is a collective that returns when done: every rank measures the traversal cost of
 * its own squares under `cam`, the costs are summed over the ranks and the squares dealt again by cost
 * (rt_shard_square_costs, rt_comm_sum_f64, rt_shard_plan; root_head_start as in rt_shard_plan, for rank 0); it waits for
 * the device first and rebuilds the buffers; *squares_moved (nullable) = squares that changed rank.  What follows a
 * moving camera: call it every so many frames.  rt_tiled_info: info[0..3] = {this rank's rays, its squares, squares of
 * the frame, the image width its launches declare}.  nranks = 1 needs no communicator. */
typedef struct rt_tiled rt_tiled;
int rt_tiled_create(rt_scene *scene, void *comm, int rank, int nranks, int width, int height, int tile,
                    const rt_ray_isect *frame_rays_host, int device, double root_head_start, rt_tiled **out);
int rt_tiled_render(rt_tiled *t, const float cam[16], void *framebuffer_dev, void *stream);
int rt_tiled_redeal(rt_tiled *t, const float cam[16], size_t *squares_moved);
int rt_tiled_info(const rt_tiled *t, uint64_t info[4]);
void rt_tiled_destroy(rt_tiled *t);

/* Diagnostic: runs the instrumented kernel once over n device-resident rays and returns, per wave of
 * 64 rays (or per 8x8 tile with the image hint), records[4*w..4*w+3] = {start, end in 100 MHz ticks,
 * HW_ID register, XCC_ID register}.  capacity is in uint64 elements. */
int rt_scene_wave_timeline(const rt_scene *s, const void *rays_dev, size_t n, uint64_t *records, size_t capacity);

/* Test hook: the dispatch order the cost-feedback schedule builds from a table of per-unit costs (host arrays): units by
 * cost bin (cost / 8, capped at 31), costliest bin first, index order inside a bin.  costs_after (nullable) receives the
 * cost table as the build leaves it (all zero). */
int rt_debug_build_order(int device, const uint32_t *unit_costs, size_t nunits, uint32_t *order_out, uint32_t *costs_after);

/* Test hook: runs the kernel's triangle test on n independent (triangle, ray) pairs
 * (host arrays: tri = 3 rt_vec3 per pair) and returns hit flag, t and (u,v) per pair. */
int rt_debug_tri_batch(int device, const rt_vec3 *tri, const rt_ray *ray, size_t n, int32_t *hit, float *t, rt_vec2 *uv);

/* ------------------------------------------------------------------ host: framebuffer out
 * 32-bit BMP with the byte layout of cpu_raycast/bmp.c:6-56 (58-byte header, bottom-up
 * rows), written in binary mode; PFM for lossless float output. */
int rt_write_bmp32(const char *path, const uint8_t *bgra, int width, int height);
int rt_write_pfm(const char *path, const float *rgba, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
