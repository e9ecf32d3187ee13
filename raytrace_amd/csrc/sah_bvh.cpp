// sah_bvh.cpp -- a surface-area-heuristic builder that emits the reference's node format.
//
// The reference builds its tree by splitting at the running mean of the triangle centroids
// (core/CentroidBVH.cpp:44-223).  This builder produces the SAME array of 80-byte
// CentroidBVHNode records (core/CentroidBVH.h:13-59: leaves of at most four triangles, inner
// nodes with plane_norm = -e_axis and children_index[0] on the low side, parent links), so
// everything that consumes such an array -- the reference's own traversal as much as the
// re-flattening in flatten.cpp -- takes it unchanged; only the topology differs.  Splits are
// chosen by the binned surface-area heuristic: 16 bins per axis over the centroid bounds, cost
// = 1 + (area_left * n_left + area_right * n_right) / area_parent, and a set of four or fewer
// triangles is split further only when that is cheaper than testing them all.
// Closest hits do not depend on the tree, so results are identical; rays just open fewer boxes.
#include <algorithm>
#include <cstring>
#include <vector>

#include "host_scene.h"

namespace rt {

namespace {

constexpr int kBins = 16;
constexpr double kNodeCost = 1.0, kTriCost = 1.0;

struct Box3 {
  float lo[3], hi[3];
  void Reset() { for (int k = 0; k < 3; ++k) { lo[k] = 3.402823466e+38f; hi[k] = -3.402823466e+38f; } }
  void Grow(const Box3 &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
  double HalfArea() const
  {
    const double x = std::max(0.0, double(hi[0]) - lo[0]), y = std::max(0.0, double(hi[1]) - lo[1]), z = std::max(0.0, double(hi[2]) - lo[2]);
    return x * y + y * z + z * x;
  }
};

struct Split {
  double cost = 0;
  int axis = -1;      // -1: no plane separates the centroids
  int bin = 0;        // triangles whose bin <= `bin` go left
  double lo = 0, scale = 0;
};

struct SahBuilder {
  CentroidBVH *bvh;
  rt_face const *tris;
  rt_vec3 const *verts;
  std::vector<rt_bvh_node> &nodes;
  std::vector<Box3> tri_box;
  std::vector<uint32_t> order;   // triangle indices, partitioned in place

  int BinOf(uint32_t t, const Split &s) const
  {
    const double c = 0.5 * (double(tri_box[t].lo[s.axis]) + tri_box[t].hi[s.axis]);
    return std::min(kBins - 1, std::max(0, static_cast<int>((c - s.lo) * s.scale)));
  }

  Box3 Bounds(size_t begin, size_t end) const
  {
    Box3 b;
    b.Reset();
    for (size_t i = begin; i < end; ++i) b.Grow(tri_box[order[i]]);
    return b;
  }

  Split BestSplit(size_t begin, size_t end, const Box3 &box) const
  {
    Split best;
    const double parent_area = std::max(box.HalfArea(), 1e-300);
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = begin; i < end; ++i)
      for (int k = 0; k < 3; ++k) {
        const double c = 0.5 * (double(tri_box[order[i]].lo[k]) + tri_box[order[i]].hi[k]);
        clo[k] = std::min(clo[k], c);
        chi[k] = std::max(chi[k], c);
      }
    for (int axis = 0; axis < 3; ++axis) {
      if (!(chi[axis] > clo[axis])) continue;
      Split s;
      s.axis = axis;
      s.lo = clo[axis];
      s.scale = kBins / (chi[axis] - clo[axis]);
      Box3 bin_box[kBins];
      size_t bin_cnt[kBins] = {0};
      for (auto &b : bin_box) b.Reset();
      for (size_t i = begin; i < end; ++i) {
        const int b = BinOf(order[i], s);
        bin_box[b].Grow(tri_box[order[i]]);
        bin_cnt[b]++;
      }
      double right_area[kBins];
      size_t right_cnt[kBins];
      Box3 acc;
      acc.Reset();
      size_t cnt = 0;
      for (int b = kBins - 1; b >= 1; --b) {
        acc.Grow(bin_box[b]);
        cnt += bin_cnt[b];
        right_area[b] = acc.HalfArea();
        right_cnt[b] = cnt;
      }
      acc.Reset();
      cnt = 0;
      for (int b = 0; b + 1 < kBins; ++b) {
        acc.Grow(bin_box[b]);
        cnt += bin_cnt[b];
        if (cnt == 0 || right_cnt[b + 1] == 0) continue;
        const double cost = kNodeCost + kTriCost * (acc.HalfArea() * cnt + right_area[b + 1] * right_cnt[b + 1]) / parent_area;
        if (best.axis < 0 || cost < best.cost) {
          best = s;
          best.cost = cost;
          best.bin = b;
        }
      }
    }
    return best;
  }

  static rt_aabb ToAabb(const Box3 &b)
  {
    rt_aabb a;
    std::memset(&a, 0, sizeof a);
    a.a.x = b.lo[0]; a.a.y = b.lo[1]; a.a.z = b.lo[2];
    a.b.x = b.hi[0]; a.b.y = b.hi[1]; a.b.z = b.hi[2];
    // an exactly flat box gets the reference's minimum thickness (core/GeomUtil.cpp:36-44): its slab test
    // needs exit > entry; thin but non-flat boxes are left tight
    if (a.b.x == a.a.x) a.b.x += 0.01f;
    if (a.b.y == a.a.y) a.b.y += 0.01f;
    if (a.b.z == a.a.z) a.b.z += 0.01f;
    return a;
  }

  // Creates the node for order[begin, end) as child `slot` of `parent` and, if it is inner, its subtree.
  void MakeChild(size_t parent, int slot, size_t begin, size_t end, int depth)
  {
    const size_t n = end - begin, idx = nodes.size();
    const Box3 box = Bounds(begin, end);
    rt_bvh_node node;
    std::memset(&node, 0, sizeof node);
    node.aabb = ToAabb(box);
    node.parent_index = static_cast<uint32_t>(parent);
    nodes[parent].node.inner.children_index[slot] = static_cast<uint32_t>(idx);
    Split s;
    bool leaf = n <= RT_TRIS_PER_LEAF;
    if (n >= 2) {
      s = BestSplit(begin, end, box);
      if (leaf && s.axis >= 0 && s.cost < kTriCost * static_cast<double>(n)) leaf = false;
    }
    if (leaf) {
      node.type = RT_BVH_LEAF;
      node.node.leaf.count = static_cast<uint32_t>(n);
      std::sort(order.begin() + begin, order.begin() + end);   // leaf triangles in index order
      for (size_t i = 0; i < n; ++i) node.node.leaf.tris[i] = order[begin + i];
      nodes.push_back(node);
      bvh->leafcount++;
      return;
    }
    node.type = RT_BVH_INNER;
    nodes.push_back(node);
    SplitInner(idx, begin, end, s, depth + 1);
  }

  void SplitInner(size_t self, size_t begin, size_t end, const Split &s, int depth)
  {
    bvh->depth = std::max(bvh->depth, depth);
    size_t mid;
    rt_float4 pn{0, 0, 0, 0};
    if (s.axis < 0) {  // coincident centroids: halve by index
      mid = begin + (end - begin) / 2;
      pn.x = -1.f;
      bvh->fallback_splits++;
    } else {
      mid = std::stable_partition(order.begin() + begin, order.begin() + end,
                                  [&](uint32_t t) { return BinOf(t, s) <= s.bin; }) - order.begin();
      (&pn.x)[s.axis] = -1.f;
    }
    nodes[self].node.inner.plane_norm = pn;
    MakeChild(self, 0, begin, mid, depth);
    MakeChild(self, 1, mid, end, depth);
  }
};

}  // namespace

void SahBVHBuild(CentroidBVH *bvh, rt_face const *triangles, rt_vec3 const *verts, size_t tricount, size_t vertcount)
{
  if (tricount <= RT_TRIS_PER_LEAF) {  // root + one leaf + the empty leaf, exactly as the reference lays it out
    CentroidBVHBuild(bvh, triangles, verts, tricount, vertcount, kBvhFixed);
    return;
  }
  bvh->nodes.clear();
  bvh->depth = bvh->leafcount = bvh->fallback_splits = 0;
  bvh->tricount = static_cast<int>(tricount);

  rt_bvh_node root;
  std::memset(&root, 0, sizeof root);
  root.type = RT_BVH_INNER;
  InitAABB(verts, &root.aabb);   // the root spans all vertices, referenced or not (core/CentroidBVH.cpp:238-240)
  for (size_t i = 0; i < vertcount; ++i) GrowAABB(verts + i, &root.aabb);
  MinSizeAABB(&root.aabb);
  root.parent_index = 0xFFFFFFFFu;
  bvh->nodes.reserve(tricount);
  bvh->nodes.push_back(root);

  SahBuilder b{bvh, triangles, verts, bvh->nodes, {}, {}};
  b.tri_box.resize(tricount);
  b.order.resize(tricount);
  for (size_t t = 0; t < tricount; ++t) {
    b.order[t] = static_cast<uint32_t>(t);
    Box3 &bx = b.tri_box[t];
    bx.Reset();
    for (int c = 0; c < 3; ++c) {
      rt_vec3 const &v = verts[triangles[t].s[c]];
      const float p[3] = {v.x, v.y, v.z};
      for (int k = 0; k < 3; ++k) { bx.lo[k] = std::min(bx.lo[k], p[k]); bx.hi[k] = std::max(bx.hi[k], p[k]); }
    }
  }
  const Box3 all = b.Bounds(0, tricount);
  b.SplitInner(0, 0, tricount, b.BestSplit(0, tricount, all), 0);
}

}  // namespace rt
