// flatten_device.hip -- the record-building half of the BVH re-flattening on the GPU.
//
// The caller's 80-byte CentroidBVHNode array (core/CentroidBVH.h:13-59), vertices and faces are already on the
// device (the scene uploads them for the hit epilogue anyway).  The host does the one sequential step, the
// validating depth-first walk that numbers inner nodes and triangle records (FlattenIndexPass); everything that is
// work per node happens here, with the arithmetic of flatten_math.h that the host path compiles too:
//   * triangle records: one thread per record gathers three vertices and subtracts the edges;
//   * two-wide records: one thread per inner node, float boxes + 16-bit grid boxes (double-precision rounding loops);
//   * four-wide records: level by level from the root, because whether a two-wide node becomes a four-wide node
//     depends on what its four-wide parent absorbed.  Per level: one thread per node chooses its <= 4 children
//     (and re-orders the records of the leaves it opens), a single-workgroup scan numbers the next level breadth
//     first, a third kernel writes the records.  Twelve levels for the dragon.
#include <hip/hip_runtime.h>

#include <string>

#include "flatten_math.h"
#include "flatten_device.h"

namespace rt {

namespace {

__global__ void TriRecordsKernel(const uint32_t *tri_of_record, uint32_t n, const rt_face *faces, const rt_vec3 *verts, GpuTri *tris)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tris[i] = MakeTri(faces, verts, tri_of_record[i]);
}

struct Grid3 {
  float lo[3], step[3];
};

__global__ void Nodes2Kernel(const uint32_t *order, uint32_t n, const rt_bvh_node *nodes, const uint32_t *inner_id, const uint32_t *leaf_ref,
                             Grid3 grid, GpuNode2 *nodes2, GpuChildQ *nodesq2)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t node = order[i];
  const uint32_t c0 = nodes[node].node.inner.children_index[0], c1 = nodes[node].node.inner.children_index[1];
  const uint32_t r0 = nodes[c0].type == RT_BVH_INNER ? inner_id[c0] : leaf_ref[c0];
  const uint32_t r1 = nodes[c1].type == RT_BVH_INNER ? inner_id[c1] : leaf_ref[c1];
  GpuNode2 g;
  GpuChildQ q[2];
  MakeNode2(nodes, node, r0, r1, grid.lo, grid.step, &g, q);
  nodes2[i] = g;
  nodesq2[2 * i] = q[0];
  nodesq2[2 * i + 1] = q[1];
}

// frontier[r] = two-wide inner node that four-wide node base + r is rooted at.
__global__ void ChooseKidsKernel(const uint32_t *frontier, uint32_t nf, uint32_t base, const rt_bvh_node *nodes, const uint32_t *leaf_ref,
                                 GpuTri *tris, const rt_face *faces, const rt_vec3 *verts, Kid *kids, uint32_t *nkids, uint32_t *ninner)
{
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nf) return;
  Kid set[4];
  const int n = ChooseKids4(nodes, leaf_ref, tris, faces, verts, frontier[r], set);
  uint32_t inner = 0;
  for (int k = 0; k < n; ++k) {
    kids[static_cast<size_t>(base + r) * 4u + k] = set[k];
    inner += set[k].inner;
  }
  nkids[base + r] = static_cast<uint32_t>(n);
  ninner[r] = inner;
}

// Exclusive scan of v[0..n) in place by one workgroup; total[0] = the sum.
__global__ void __launch_bounds__(1024) ScanKernel(uint32_t *v, uint32_t n, uint32_t *total)
{
  __shared__ uint32_t part[1024];
  const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u, lo = t * per, hi = min(n, lo + per);
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += v[i];
  part[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < 1024u; off <<= 1) {
    const uint32_t add = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;   // exclusive prefix of this thread's chunk
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t x = v[i];
    v[i] = run;
    run += x;
  }
  if (t == 1023u) total[0] = part[1023];
}

__global__ void EmitNodes4Kernel(uint32_t nf, uint32_t base, const Kid *kids, const uint32_t *nkids, const uint32_t *offset, Grid3 grid,
                                 GpuNode4 *nodes4, GpuChildQ *nodesq4, uint32_t *next_frontier)
{
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nf) return;
  Kid set[4];
  const int n = static_cast<int>(nkids[base + r]);
  uint32_t child_ref[4] = {0, 0, 0, 0};
  uint32_t rank = offset[r];
  for (int k = 0; k < n; ++k) {
    set[k] = kids[static_cast<size_t>(base + r) * 4u + k];
    if (set[k].inner) {
      child_ref[k] = base + nf + rank;      // the next level is numbered right behind this one, in frontier order
      next_frontier[rank] = set[k].node;
      ++rank;
    } else {
      child_ref[k] = kLeafBit | (set[k].first << 3) | set[k].cnt;
    }
  }
  GpuNode4 g;
  GpuChildQ q[4];
  MakeNode4(set, n, child_ref, grid.lo, grid.step, &g, q);
  nodes4[base + r] = g;
  for (int k = 0; k < 4; ++k) nodesq4[static_cast<size_t>(base + r) * 4u + k] = q[k];
}

template <typename T>
hipError_t Put(T **dst, const void *src, size_t bytes)
{
  *dst = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), bytes ? bytes : 16);
  if (e == hipSuccess && bytes && src) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  return e;
}

}  // namespace

hipError_t FlattenOnDevice(FlatIndex const &ix, const rt_bvh_node *d_nodes, size_t nnodes, const rt_vec3 *d_verts, const rt_face *d_faces,
                           DeviceFlat *out)
{
  *out = DeviceFlat();
  const uint32_t n2 = static_cast<uint32_t>(ix.order.size()), ntri = static_cast<uint32_t>(ix.tri_of_record.size());
  uint32_t *d_order = nullptr, *d_inner_id = nullptr, *d_leaf_ref = nullptr, *d_tri_of = nullptr;
  uint32_t *d_front[2] = {nullptr, nullptr}, *d_nkids = nullptr, *d_ninner = nullptr, *d_total = nullptr;
  Kid *d_kids = nullptr;
  Grid3 grid;
  for (int k = 0; k < 3; ++k) { grid.lo[k] = ix.grid_lo[k]; grid.step[k] = ix.grid_step[k]; }
  hipError_t e = Put(&d_order, ix.order.data(), n2 * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_inner_id, ix.inner_id.data(), nnodes * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_leaf_ref, ix.leaf_ref.data(), nnodes * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_tri_of, ix.tri_of_record.data(), ntri * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&out->tris, nullptr, ntri * sizeof(GpuTri));
  if (e == hipSuccess) e = Put(&out->nodes2, nullptr, n2 * sizeof(GpuNode2));
  if (e == hipSuccess) e = Put(&out->nodesq2, nullptr, n2 * 2 * sizeof(GpuChildQ));
  // every four-wide node is rooted at a distinct two-wide inner node: n2 bounds their number
  if (e == hipSuccess) e = Put(&out->nodes4, nullptr, n2 * sizeof(GpuNode4));
  if (e == hipSuccess) e = Put(&out->nodesq4, nullptr, static_cast<size_t>(n2) * 4 * sizeof(GpuChildQ));
  if (e == hipSuccess) e = Put(&d_kids, nullptr, static_cast<size_t>(n2) * 4 * sizeof(Kid));
  if (e == hipSuccess) e = Put(&d_nkids, nullptr, n2 * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_ninner, nullptr, n2 * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_front[0], nullptr, n2 * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_front[1], nullptr, n2 * sizeof(uint32_t));
  if (e == hipSuccess) e = Put(&d_total, nullptr, sizeof(uint32_t));
  if (e == hipSuccess) {
    (void)hipGetLastError();
    if (ntri) hipLaunchKernelGGL(TriRecordsKernel, dim3((ntri + 255u) / 256u), dim3(256), 0, nullptr, d_tri_of, ntri, d_faces, d_verts, out->tris);
    hipLaunchKernelGGL(Nodes2Kernel, dim3((n2 + 255u) / 256u), dim3(256), 0, nullptr, d_order, n2, d_nodes, d_inner_id, d_leaf_ref, grid,
                       out->nodes2, out->nodesq2);
    e = hipGetLastError();
  }
  uint32_t base = 0, nf = 1, root = 0;
  int depth4 = 0, cur = 0;
  if (e == hipSuccess) e = hipMemcpy(d_front[0], &root, sizeof root, hipMemcpyHostToDevice);
  while (e == hipSuccess && nf > 0) {
    if (base + nf > n2) { e = hipErrorInvalidValue; break; }   // cannot happen for a tree; never write past the arrays
    ++depth4;
    const dim3 g((nf + 63u) / 64u), b(64);
    hipLaunchKernelGGL(ChooseKidsKernel, g, b, 0, nullptr, d_front[cur], nf, base, d_nodes, d_leaf_ref, out->tris, d_faces, d_verts, d_kids,
                       d_nkids, d_ninner);
    hipLaunchKernelGGL(ScanKernel, dim3(1), dim3(1024), 0, nullptr, d_ninner, nf, d_total);
    hipLaunchKernelGGL(EmitNodes4Kernel, g, b, 0, nullptr, nf, base, d_kids, d_nkids, d_ninner, grid, out->nodes4, out->nodesq4,
                       d_front[cur ^ 1]);
    e = hipGetLastError();
    uint32_t next = 0;
    if (e == hipSuccess) e = hipMemcpy(&next, d_total, sizeof next, hipMemcpyDeviceToHost);   // also the level's sync point
    base += nf;
    nf = next;
    cur ^= 1;
  }
  out->n2 = n2;
  out->n4 = base;
  out->ntris = ntri;
  out->depth4 = depth4;
  (void)hipFree(d_order); (void)hipFree(d_inner_id); (void)hipFree(d_leaf_ref); (void)hipFree(d_tri_of);
  (void)hipFree(d_front[0]); (void)hipFree(d_front[1]); (void)hipFree(d_nkids); (void)hipFree(d_ninner); (void)hipFree(d_total);
  (void)hipFree(d_kids);
  if (e != hipSuccess) {
    (void)hipFree(out->tris); (void)hipFree(out->nodes2); (void)hipFree(out->nodesq2); (void)hipFree(out->nodes4); (void)hipFree(out->nodesq4);
    *out = DeviceFlat();
  }
  return e;
}

}  // namespace rt
