// flatten_math.h -- the record-building arithmetic of the BVH re-flattening, written once for the host
// (flatten.cpp) and the device (flatten_device.hip): box quantization, the pre-gathered triangle record, and the
// choice of the <= 4 children of one four-wide node.  Both sides compile this source with contraction off, so the
// device builds the records the host builds, bit for bit.
#pragma once

#include <cmath>
#include <cstdint>

#include "host_scene.h"
#include "trace_math.h"

namespace rt {

// Grid index of the plane at or below (down) / at or above (up) coordinate v, with two cells of slack.
RT_HD uint32_t QuantDown(float v, float lo, float step)
{
  double q = floor((static_cast<double>(v) - lo) / step);
  while (q > 0 && lo + q * static_cast<double>(step) > v) q -= 1;
  q -= 2;  // one cell for the grid arithmetic, one for the traversal's biased plane constant
  return static_cast<uint32_t>(fmin(65535.0, fmax(0.0, q)));
}
RT_HD uint32_t QuantUp(float v, float lo, float step)
{
  double q = ceil((static_cast<double>(v) - lo) / step);
  while (q < 65535 && lo + q * static_cast<double>(step) < v) q += 1;
  q += 2;
  return static_cast<uint32_t>(fmin(65535.0, fmax(0.0, q)));
}

RT_HD GpuChildQ QuantChild(rt_aabb const &b, uint32_t ref, const float *lo, const float *step)
{
  GpuChildQ c;
  c.x = QuantDown(b.a.x, lo[0], step[0]) | (QuantUp(b.b.x, lo[0], step[0]) << 16);
  c.y = QuantDown(b.a.y, lo[1], step[1]) | (QuantUp(b.b.y, lo[1], step[1]) << 16);
  c.z = QuantDown(b.a.z, lo[2], step[2]) | (QuantUp(b.b.z, lo[2], step[2]) << 16);
  c.ref = ref;
  return c;
}

RT_HD float Area(rt_aabb const &b)
{
  float x = b.b.x - b.a.x, y = b.b.y - b.a.y, z = b.b.z - b.a.z;
  return x * y + y * z + z * x;
}

// The 48-byte record of triangle t: vertex a and the two edges the reference's test subtracts first
// (core/ray_tri.cpp:20-21), plus the original index.
RT_HD GpuTri MakeTri(rt_face const *faces, rt_vec3 const *verts, uint32_t t)
{
  rt_face const &f = faces[t];
  rt_vec3 const &a = verts[f.s[0]], &b = verts[f.s[1]], &c = verts[f.s[2]];
  GpuTri g;
  g.ax = a.x; g.ay = a.y; g.az = a.z; g.id = t;
  g.e1x = b.x - a.x; g.e1y = b.y - a.y; g.e1z = b.z - a.z; g.pad0 = 0;
  g.e2x = c.x - a.x; g.e2y = c.y - a.y; g.e2z = c.z - a.z; g.pad1 = 0;
  return g;
}

// One child of a four-wide node while it is being chosen.
struct Kid {
  uint32_t inner;       // 1: a two-wide inner node (index `node` of the caller's array); 0: a run of triangle records
  uint32_t node;
  uint32_t first, cnt;
  rt_aabb box;
};

RT_HD Kid KidOf(rt_bvh_node const *nodes, uint32_t const *leaf_ref, uint32_t node)
{
  Kid k;
  k.inner = nodes[node].type == RT_BVH_INNER ? 1u : 0u;
  k.node = node;
  k.box = nodes[node].aabb;
  k.first = k.inner ? 0u : (leaf_ref[node] & ~kLeafBit) >> 3;
  k.cnt = k.inner ? 0u : leaf_ref[node] & 7u;
  return k;
}

// Grows *b (or starts it, when init) by the three vertices of the triangle behind record `rec`.
RT_HD void GrowByRecord(GpuTri const *tris, rt_face const *faces, rt_vec3 const *verts, uint32_t rec, rt_aabb *b, bool init)
{
  rt_face const &f = faces[tris[rec].id];
  for (int c = 0; c < 3; ++c) {
    rt_vec3 const &v = verts[f.s[c]];
    if (init && c == 0) { b->a = v; b->b = v; continue; }
    b->a.x = fminf(b->a.x, v.x); b->a.y = fminf(b->a.y, v.y); b->a.z = fminf(b->a.z, v.z);
    b->b.x = fmaxf(b->b.x, v.x); b->b.y = fmaxf(b->b.y, v.y); b->b.z = fmaxf(b->b.z, v.z);
  }
}

// The children of the four-wide node rooted at two-wide inner node r.  The set {child0, child1} is widened by
// replacing its largest inner member with that member's two children until it holds four entries or only leaves.
// Slots still free after that (a quarter of all slots on the benchmark meshes) would be tested by every visiting
// ray for nothing, so they are given to the node's leaves: a leaf of 2..4 triangles is split into two runs with their
// own, tighter boxes (the two-way partition of its triangles with the least summed area x count; the leaf's records
// in `tris` are reordered to make both runs contiguous).  Same triangles under the same node, same node count, fewer
// triangle tests per ray.  Returns the number of children written to set[0..3].
RT_HD int ChooseKids4(rt_bvh_node const *nodes, uint32_t const *leaf_ref, GpuTri *tris, rt_face const *faces, rt_vec3 const *verts,
                      uint32_t r, Kid set[4])
{
  set[0] = KidOf(nodes, leaf_ref, nodes[r].node.inner.children_index[0]);
  set[1] = KidOf(nodes, leaf_ref, nodes[r].node.inner.children_index[1]);
  int n = 2;
  while (n < 4) {
    int pick = -1;
    float best = -1.f;
    for (int k = 0; k < n; ++k)
      if (set[k].inner) {
        float a = Area(set[k].box);
        if (a > best) { best = a; pick = k; }
      }
    if (pick < 0) break;
    const uint32_t p = set[pick].node;
    set[pick] = KidOf(nodes, leaf_ref, nodes[p].node.inner.children_index[0]);
    set[n++] = KidOf(nodes, leaf_ref, nodes[p].node.inner.children_index[1]);
  }
  while (n < 4) {
    int pick = -1;
    float best = -1.f;
    for (int k = 0; k < n; ++k)
      if (!set[k].inner && set[k].cnt >= 2) {
        float a = Area(set[k].box) * static_cast<float>(set[k].cnt);
        if (a > best) { best = a; pick = k; }
      }
    if (pick < 0) break;
    const Kid leaf = set[pick];
    // every two-way partition of the <= 4 records: bit j of `mask` set = record j goes to the second run
    uint32_t best_mask = 0;
    float best_cost = kFltMax;
    for (uint32_t mask = 1; mask + 1 < (1u << leaf.cnt); ++mask) {
      if (mask & 1u) continue;  // record 0 stays in the first run: each partition is seen once
      rt_aabb ba = leaf.box, bb = leaf.box;   // both runs are non-empty for every mask tried: overwritten below
      uint32_t na = 0, nb = 0;
      for (uint32_t j = 0; j < leaf.cnt; ++j) {
        if (mask >> j & 1u) GrowByRecord(tris, faces, verts, leaf.first + j, &bb, nb++ == 0);
        else GrowByRecord(tris, faces, verts, leaf.first + j, &ba, na++ == 0);
      }
      const float cost = Area(ba) * static_cast<float>(na) + Area(bb) * static_cast<float>(nb);
      if (cost < best_cost) { best_cost = cost; best_mask = mask; }
    }
    GpuTri tmp[4];
    uint32_t na = 0;
    for (uint32_t j = 0; j < leaf.cnt; ++j)
      if (!(best_mask >> j & 1u)) tmp[na++] = tris[leaf.first + j];
    uint32_t w = na;
    for (uint32_t j = 0; j < leaf.cnt; ++j)
      if (best_mask >> j & 1u) tmp[w++] = tris[leaf.first + j];
    for (uint32_t j = 0; j < leaf.cnt; ++j) tris[leaf.first + j] = tmp[j];
    Kid a = leaf, b = leaf;
    a.inner = b.inner = 0u;
    a.node = b.node = leaf.node;
    a.first = leaf.first; a.cnt = na;
    b.first = leaf.first + na; b.cnt = leaf.cnt - na;
    for (uint32_t j = 0; j < a.cnt; ++j) GrowByRecord(tris, faces, verts, a.first + j, &a.box, j == 0);
    for (uint32_t j = 0; j < b.cnt; ++j) GrowByRecord(tris, faces, verts, b.first + j, &b.box, j == 0);
    set[pick] = a;
    set[n++] = b;
  }
  return n;
}

// Record of the four-wide node with the given children; child_ref[k] = reference of set[k].
RT_HD void MakeNode4(const Kid set[4], int n, const uint32_t child_ref[4], const float *grid_lo, const float *grid_step, GpuNode4 *g,
                     GpuChildQ q[4])
{
  for (int k = 0; k < 4; ++k) {
    if (k < n) {
      rt_aabb const &b = set[k].box;
      g->lox[k] = b.a.x; g->loy[k] = b.a.y; g->loz[k] = b.a.z;
      g->hix[k] = b.b.x; g->hiy[k] = b.b.y; g->hiz[k] = b.b.z;
      g->child[k] = child_ref[k];
      q[k] = QuantChild(b, child_ref[k], grid_lo, grid_step);
    } else {  // unused slot: inverted box never passes the slab test
      g->lox[k] = g->loy[k] = g->loz[k] = kFltMax;
      g->hix[k] = g->hiy[k] = g->hiz[k] = -kFltMax;
      g->child[k] = kEmptyRef;
      q[k] = GpuChildQ{0x0000FFFFu, 0x0000FFFFu, 0x0000FFFFu, kEmptyRef};
    }
    g->pad[k] = 0;
  }
}

// Record of the two-wide node of inner node `node`; ref0/ref1 = references of its children.
RT_HD void MakeNode2(rt_bvh_node const *nodes, uint32_t node, uint32_t ref0, uint32_t ref1, const float *grid_lo, const float *grid_step,
                     GpuNode2 *g, GpuChildQ q[2])
{
  const uint32_t c0 = nodes[node].node.inner.children_index[0], c1 = nodes[node].node.inner.children_index[1];
  rt_aabb const &b0 = nodes[c0].aabb, &b1 = nodes[c1].aabb;
  g->lo0[0] = b0.a.x; g->lo0[1] = b0.a.y; g->lo0[2] = b0.a.z; g->hi0[0] = b0.b.x; g->hi0[1] = b0.b.y; g->hi0[2] = b0.b.z;
  g->lo1[0] = b1.a.x; g->lo1[1] = b1.a.y; g->lo1[2] = b1.a.z; g->hi1[0] = b1.b.x; g->hi1[1] = b1.b.y; g->hi1[2] = b1.b.z;
  g->child[0] = ref0; g->child[1] = ref1;
  g->pad[0] = g->pad[1] = 0;
  q[0] = QuantChild(b0, ref0, grid_lo, grid_step);
  q[1] = QuantChild(b1, ref1, grid_lo, grid_step);
}

}  // namespace rt
