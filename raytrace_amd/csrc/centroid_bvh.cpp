// centroid_bvh.cpp -- the reference's centroid-split BVH2 builder and its AABB helpers.
//
// Tree shape, node numbering and every float are those of core/CentroidBVH.cpp:44-253
// with core/GeomUtil.cpp:21-101: split point = running mean of the set's centroids,
// partition by centroid[axis] < split, axis cycles x->y->z and advances past axes that
// leave one side empty, <= 4 triangles per leaf, children pushed in pre-order, every
// node stores its OWN box grown from its member triangles' vertices.
//
// Two deliberate differences, neither reachable on inputs the reference handles:
//  * when no axis separates the set (the reference loops forever, :79-136) the set is
//    cut in half by position;
//  * a scene of <= 4 triangles (the reference prints "bounding volume broke..." and
//    leaves the root's children unset, :46-49) gets a root with one real leaf and one
//    empty leaf.
// Padding words and unused leaf slots, which the reference leaves uninitialised
// (:177,:201), are zero here.
#include "host_scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rt {

// ---- core/GeomUtil.cpp ---------------------------------------------------------
void InitAABB(rt_vec3 const *v, rt_aabb *dest)
{
  dest->a = *v;
  dest->b = *v;
}

void GrowAABB(rt_vec3 const *p, rt_aabb *dest)
{
  dest->a.x = std::min(p->x, dest->a.x);
  dest->a.y = std::min(p->y, dest->a.y);
  dest->a.z = std::min(p->z, dest->a.z);
  dest->b.x = std::max(p->x, dest->b.x);
  dest->b.y = std::max(p->y, dest->b.y);
  dest->b.z = std::max(p->z, dest->b.z);
}

// Extents below 0.01 get 0.01 added to the max side, compared and added in double
// (core/GeomUtil.cpp:27-41: the literals are doubles).
void MinSizeAABB(rt_aabb *dest)
{
  if (dest->b.x - dest->a.x < 0.01) dest->b.x += 0.01;
  if (dest->b.y - dest->a.y < 0.01) dest->b.y += 0.01;
  if (dest->b.z - dest->a.z < 0.01) dest->b.z += 0.01;
}

void Centroid(rt_vec3 const &a, rt_vec3 const &b, rt_vec3 const &c, rt_vec3 *out)
{
  out->x = (a.x + b.x + c.x) / 3.f;
  out->y = (a.y + b.y + c.y) / 3.f;
  out->z = (a.z + b.z + c.z) / 3.f;
}

bool PointInAABB(rt_vec3 const *p, rt_aabb const *bb, float e)
{
  bool a = p->x - bb->a.x > -e && p->x - bb->b.x < e;
  bool b = p->y - bb->a.y > -e && p->y - bb->b.y < e;
  bool c = p->z - bb->a.z > -e && p->z - bb->b.z < e;
  return a && b && c;
}

// core/raymath.cpp:6-15
rt_vec3 Normalize(rt_vec3 in)
{
  float lensq = in.x * in.x + in.y * in.y + in.z * in.z;
  float len = std::sqrt(lensq);
  rt_vec3 r;
  r.x = in.x / len;
  r.y = in.y / len;
  r.z = in.z / len;
  r.w = 0.f;
  return r;
}

// ---- core/CentroidBVH.cpp ------------------------------------------------------
namespace {

struct Builder {
  CentroidBVH *bvh;
  rt_face const *tris;
  rt_vec3 const *verts;
  BvhMode mode;
  std::vector<rt_bvh_node> &nodes;

  rt_aabb BoxOf(std::vector<size_t> const &set) const
  {
    rt_aabb bb;
    InitAABB(&verts[tris[set[0]].s[0]], &bb);
    for (size_t t : set) {
      GrowAABB(&verts[tris[t].s[0]], &bb);
      GrowAABB(&verts[tris[t].s[1]], &bb);
      GrowAABB(&verts[tris[t].s[2]], &bb);
    }
    MinSizeAABB(&bb);
    return bb;
  }

  size_t PushChild(size_t parent, int slot, std::vector<size_t> const &set, rt_aabb const &box)
  {
    size_t idx = nodes.size();
    rt_bvh_node n;
    std::memset(&n, 0, sizeof n);
    n.aabb = box;
    n.parent_index = static_cast<uint32_t>(parent);
    nodes[parent].node.inner.children_index[slot] = static_cast<uint32_t>(idx);
    if (set.size() <= RT_TRIS_PER_LEAF) {
      n.type = RT_BVH_LEAF;
      bvh->leafcount++;
      n.node.leaf.count = static_cast<uint32_t>(set.size());
      for (size_t i = 0; i < set.size(); ++i) n.node.leaf.tris[i] = static_cast<uint32_t>(set[i]);
    } else {
      n.type = RT_BVH_INNER;
    }
    nodes.push_back(n);
    return idx;
  }

  void Recurse(std::vector<size_t> const &set, size_t parent, unsigned axis, int depth)
  {
    const size_t n = set.size();
    bvh->depth = std::max(bvh->depth, depth);

    // Centroids and their running mean (core/CentroidBVH.cpp:57-73).  In reference
    // mode the i-th centroid comes from triangle i, not from set[i] (:65).
    std::vector<rt_vec3> cen(n);
    rt_vec3 split{0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
      rt_face const &f = tris[mode == kBvhReference ? i : set[i]];
      Centroid(verts[f.s[0]], verts[f.s[1]], verts[f.s[2]], &cen[i]);
      float keep = i / (i + 1.f), add = 1.f / (i + 1.f);
      split.x = split.x * keep + cen[i].x * add;
      split.y = split.y * keep + cen[i].y * add;
      split.z = split.z * keep + cen[i].z * add;
    }

    std::vector<size_t> lo, hi;
    unsigned axis_count = axis;
    int tries = 0;
    while (lo.empty() || hi.empty()) {
      lo.clear();
      hi.clear();
      if (tries++ == 3) {  // no axis separates this set
        lo.assign(set.begin(), set.begin() + n / 2);
        hi.assign(set.begin() + n / 2, set.end());
        bvh->fallback_splits++;
        break;
      }
      const unsigned k = axis_count % 3;
      const float s = (&split.x)[k];
      for (size_t i = 0; i < n; ++i) ((&cen[i].x)[k] < s ? lo : hi).push_back(set[i]);
      rt_float4 pn{0, 0, 0, 0};
      (&pn.x)[k] = -1.f;
      nodes[parent].node.inner.plane_norm = pn;
      axis_count++;
    }

    rt_aabb lo_box = BoxOf(lo), hi_box = BoxOf(hi);

    size_t c0 = PushChild(parent, 0, lo, lo_box);
    if (lo.size() > RT_TRIS_PER_LEAF) Recurse(lo, c0, axis_count, depth + 1);
    size_t c1 = PushChild(parent, 1, hi, hi_box);
    if (hi.size() > RT_TRIS_PER_LEAF) Recurse(hi, c1, axis_count, depth + 1);
  }
};

}  // namespace

void CentroidBVHBuild(CentroidBVH *bvh, rt_face const *triangles, rt_vec3 const *verts, size_t tricount,
                      size_t vertcount, BvhMode mode)
{
  bvh->nodes.clear();
  bvh->depth = bvh->tricount = bvh->leafcount = bvh->fallback_splits = 0;
  if (tricount == 0) return;
  bvh->tricount = static_cast<int>(tricount);

  // Root box spans ALL vertices, referenced or not (core/CentroidBVH.cpp:238-240).
  rt_bvh_node root;
  std::memset(&root, 0, sizeof root);
  root.type = RT_BVH_INNER;
  InitAABB(verts, &root.aabb);
  for (size_t i = 0; i < vertcount; ++i) GrowAABB(verts + i, &root.aabb);
  MinSizeAABB(&root.aabb);
  root.parent_index = 0xFFFFFFFFu;
  bvh->nodes.push_back(root);

  std::vector<size_t> all(tricount);
  for (size_t i = 0; i < tricount; ++i) all[i] = i;
  Builder b{bvh, triangles, verts, mode, bvh->nodes};
  if (tricount <= RT_TRIS_PER_LEAF) {
    rt_float4 pn{-1.f, 0, 0, 0};
    bvh->nodes[0].node.inner.plane_norm = pn;
    const rt_aabb box = b.BoxOf(all);
    b.PushChild(0, 0, all, box);
    b.PushChild(0, 1, std::vector<size_t>(), box);
    bvh->leafcount--;  // the empty leaf is not a real leaf
    return;
  }
  b.Recurse(all, 0, 0, 0);
}

}  // namespace rt
