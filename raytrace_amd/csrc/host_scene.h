// host_scene.h -- host-side scene front end: OBJ/MTL load, material library,
// centroid-split BVH build, ray generators, BVH re-flattening.
//
// Mirrors the reference's host interfaces for this path:
//   LoadObj / LoadObjMaterials      core/ObjLoader.h:12-14
//   MtlLib                          core/MtlLib.h:23-30
//   CentroidBVHInit/Build/Free      core/CentroidBVH.h:90-94
//   InitAABB/GrowAABB/MinSizeAABB   core/GeomUtil.h:5-44
//   normalize                       core/raydata.h:52, core/raymath.cpp:6-15
#pragma once

#include <cstddef>
#include <cstdint>
#include <istream>
#include <map>
#include <string>
#include <vector>

#include "../../include/rt_layout.h"

namespace rt {

// ---- material library (core/MtlLib.h:23-30, core/MtlLib.cpp:3-12) ----------
struct MtlLib {
  std::map<std::string, int> indices;
  std::vector<rt_material> materials;
  void InsertMaterial(std::string const &name, rt_material const &mtl);
  int MaterialIndex(std::string const &name);  // unknown names map to 0, as map::operator[] does
};

// ---- OBJ / MTL loader (core/ObjLoader.cpp:37-224, :244-354) ----------------
enum LoaderMode : unsigned {
  kLoaderReference = 0,  // one token per delimiter: repeated spaces yield empty tokens (core/ObjLoader.cpp:12-34)
  kLoaderTolerant = 1,   // empty tokens from repeated / trailing spaces are dropped; all else identical
  kLoaderExtended = 2,   // tolerant + quads / n-gons fanned into triangles + negative (relative) indices
};

struct Mesh {
  std::vector<rt_vec3> verts;
  std::vector<rt_float4> normals;
  std::vector<rt_face> faces;
  std::vector<rt_vec3> uvws;          // `vt` records as the reference parses them (core/ObjLoader.cpp:96-118): u, v, optional w
  std::vector<uint32_t> face_uvws;    // 3 per face: 0-based `vt` index of each corner ("v/vt/vn"), 0xFFFFFFFF where absent --
                                      // the indices the reference leaves commented out (core/ObjLoader.cpp:178-186)
  MtlLib mtllib;
  std::vector<std::string> log;  // the messages the reference printf()s
};

void SplitString(std::string const &str, char token, std::vector<std::string> &out, bool keep_empty);
void LoadObj(std::istream &file, Mesh &mesh, std::string const &folder, LoaderMode mode);
void LoadObjMaterials(std::istream &file, MtlLib &matlib, LoaderMode mode, std::vector<std::string> *log);
// Inserts the all-zero "SKY_DEFAULT" material first (render_client/img_main.cpp:240) then loads.
bool LoadObjFile(std::string const &path, Mesh &mesh, LoaderMode mode);

// ---- geometry helpers (core/GeomUtil.cpp) ----------------------------------
void InitAABB(rt_vec3 const *v, rt_aabb *dest);
void GrowAABB(rt_vec3 const *test, rt_aabb *dest);
void MinSizeAABB(rt_aabb *dest);
void Centroid(rt_vec3 const &a, rt_vec3 const &b, rt_vec3 const &c, rt_vec3 *out);
bool PointInAABB(rt_vec3 const *p, rt_aabb const *bb, float e = 0.000001f);
rt_vec3 Normalize(rt_vec3 in);

// ---- BVH build (core/CentroidBVH.cpp:226-253, :44-223) ---------------------
enum BvhMode : unsigned {
  kBvhReference = 0,  // centroids taken from triangles[i] (core/CentroidBVH.cpp:65, as written)
  kBvhFixed = 1,      // centroids taken from triangles[set[i]] (the intended partition)
};

struct CentroidBVH {
  std::vector<rt_bvh_node> nodes;  // root = nodes[0], pre-order
  int depth = 0;
  int tricount = 0;
  int leafcount = 0;
  int fallback_splits = 0;  // index-median splits taken where the reference would loop forever
};

void CentroidBVHBuild(CentroidBVH *bvh, rt_face const *triangles, rt_vec3 const *verts, size_t tricount,
                      size_t vertcount, BvhMode mode);
// Same output format, topology from the binned surface-area heuristic (sah_bvh.cpp).
void SahBVHBuild(CentroidBVH *bvh, rt_face const *triangles, rt_vec3 const *verts, size_t tricount, size_t vertcount);

// ---- camera matrix ----------------------------------------------------------
// The matrix the reference's frame loop uploads each frame (render_client/img_main.cpp:627-660): rotations about y
// and x built by rotate3 (core/Affine.h:165-192, including its out[1][0] = 0 slip), combined by multiply
// (core/Affine.h:194-211), then the translation.  rot_x / rot_y are the angles it derives from the cursor
// (cursor / 50); out is the 16 floats handed to the kernel, whose rows are dotted with (x,y,z,w).
void CameraMatrix(float rot_x, float rot_y, const float trans[3], float out[16]);

// ---- ray generators ---------------------------------------------------------
// Primary-ray table of render_client/img_main.cpp:253-267 for a W x H image;
// record index = i + j*W (the reference's i + j*IMG_Y, equal for its square image).
void PrimaryRays(int W, int H, const float origin[3], rt_ray_isect *out);
// Ray origin that frames `root` with the reference pinhole (tan half-hfov 0.1) from -z with 10 % margin.
void FrameCamera(rt_aabb const &root, int W, int H, float origin_out[3]);
// Seeded incoherent rays: point on triangle (k mod ntris), cosine-weighted direction about its geometric normal.
void AoRays(rt_vec3 const *verts, rt_face const *faces, size_t ntris, rt_aabb const &root, uint64_t seed,
            size_t first, size_t count, rt_ray_isect *out);

// ---- re-flattened GPU layouts ------------------------------------------------
// One 48-byte record per triangle, stored in leaf order.
struct GpuTri {
  float ax, ay, az; uint32_t id;   // vertex a, original triangle index
  float e1x, e1y, e1z; uint32_t pad0;  // b - a   (the subtraction core/ray_tri.cpp:20 performs)
  float e2x, e2y, e2z; uint32_t pad1;  // c - a   (core/ray_tri.cpp:21)
};
static_assert(sizeof(GpuTri) == 48, "triangle record is 48 B");

// Child reference: inner -> node index; leaf -> kLeafBit | first_tri << 3 | count (count 0 = empty).
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kEmptyRef = kLeafBit;  // leaf with zero triangles

// BVH2 node, 64 B = half a 128-byte line: both children's boxes and references.
struct GpuNode2 {
  float lo0[3], hi0[3], lo1[3], hi1[3];
  uint32_t child[2];
  uint32_t pad[2];
};
static_assert(sizeof(GpuNode2) == 64, "BVH2 node is 64 B");

// BVH4 node, 128 B = one cache line: four children's boxes (SoA per plane) and references.
struct GpuNode4 {
  float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(GpuNode4) == 128, "BVH4 node is 128 B");

// One child of a quantized node: three axis words (lo | hi << 16 on the scene grid) + reference.
struct GpuChildQ {
  uint32_t x, y, z, ref;
};
static_assert(sizeof(GpuChildQ) == 16, "quantized child is 16 B");

struct FlatScene {
  std::vector<GpuNode2> nodes2;
  std::vector<GpuNode4> nodes4;
  std::vector<GpuChildQ> nodesq2;  // 2 children per record (32 B), same indexing as nodes2
  std::vector<GpuChildQ> nodesq4;  // 4 children per record (64 B), same indexing as nodes4
  float grid_lo[3] = {0, 0, 0};    // quantization grid: plane(q) = grid_lo + q * grid_step
  float grid_step[3] = {1, 1, 1};
  std::vector<GpuTri> tris;
  int depth2 = 0;       // max number of inner nodes on a root-to-leaf path (BVH2)
  int depth4 = 0;       // same for the collapsed BVH4
  int stack2 = 0;       // traversal stack entries sufficient for BVH2
  int stack4 = 0;       // traversal stack entries sufficient for BVH4
  rt_aabb root_box{};
  float scale = 0.f;    // max |coordinate| of the root box; feeds the per-ray slab padding
};

// Result of the sequential part of the re-flattening: the numbering of inner nodes and triangle records.
struct FlatIndex {
  std::vector<uint32_t> order;          // caller-array indices of the inner nodes, in the new (depth-first) order
  std::vector<uint32_t> inner_id;       // per caller node: its two-wide record index (inner nodes)
  std::vector<uint32_t> leaf_ref;       // per caller node: kLeafBit | first record << 3 | count (leaves)
  std::vector<uint32_t> tri_of_record;  // per triangle record: original triangle index
  int depth2 = 0;
  float grid_lo[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
};
std::string FlattenIndexPass(rt_bvh_node const *nodes, size_t nnodes, rt_vec3 const *verts, size_t nverts,
                             rt_face const *faces, size_t ntris, FlatIndex *ix);
void FlattenFinish(rt_bvh_node const *nodes, FlatIndex const &ix, int depth4, FlatScene *out);

// Validates `nodes` (indices in range, tree-shaped, leaves <= 4 triangles, root inner) and builds both layouts.
// Returns an empty string on success, else the reason.
std::string Flatten(rt_bvh_node const *nodes, size_t nnodes, rt_vec3 const *verts, size_t nverts,
                    rt_face const *faces, size_t ntris, FlatScene *out);

// Host traversal of the flattened layouts, used ONLY to count records fetched per ray
// (the algorithmic-bytes figure); it returns counters, never hits.
struct FetchCounts { uint64_t nodes = 0, tris = 0, leaves = 0, rays_hit = 0, max_stack = 0; };
// per_ray (optional): per_ray[2*i], per_ray[2*i+1] = node and triangle records fetched by the i-th counted ray.
FetchCounts CountFetches(FlatScene const &fs, int width, bool quant, rt_ray_isect const *rays, size_t nrays,
                         size_t stride, uint32_t *per_ray = nullptr);

// Lockstep wave model (see flatten.cpp): issued wave-level steps vs useful lane slots.
struct WaveModel {
  uint64_t waves = 0, rays = 0;
  uint64_t wave_node_steps = 0, lane_node_steps = 0;
  uint64_t wave_tri_steps = 0, lane_tri_steps = 0;
  uint64_t wave_leaf_phases = 0, lane_leaf_phases = 0;
  uint64_t refills = 0, refill_lanes = 0;
  uint64_t max_wave_steps = 0;  // node + triangle steps of the longest wave
  uint64_t uniform_node_steps = 0;  // node steps in which every participating lane visits the same node
  uint64_t two_node_steps = 0, three_node_steps = 0;  // node steps whose lanes visit exactly two / three different nodes
  uint64_t iterations = 0;          // loop iterations (each exposes one memory latency), all waves
  uint64_t max_wave_iterations = 0; // ... of the wave that needs most
};
WaveModel SimulateWaves(FlatScene const &fs, int width, bool quant, rt_ray_isect const *rays, size_t nrays,
                        uint32_t img_w, uint32_t refill, size_t max_waves, int policy);

// ---- framebuffer output (cpu_raycast/bmp.c:6-56 byte layout) ---------------
bool WriteBmp32(const char *path, const uint8_t *bgra, int width, int height);
bool WritePfm(const char *path, const float *rgba, int width, int height);

}  // namespace rt
