// flatten.cpp -- turns the reference's 80-byte parent-linked AoS node array
// (core/CentroidBVH.h:52-59) into the two device layouts and counts, on the host, how
// many records a ray fetches from them.
//
// The reference layout stores each node's OWN box, so visiting a child costs a second
// dependent line; its traversal walks parent links (render_client/tracekernel.ck:115-138).
// Here a node record carries its CHILDREN's boxes and references: one fetch decides both
// (BVH2, 64 B) or all four (BVH4, 128 B = one cache line) descents, leaves are folded into
// the parent's reference, and triangles are pre-gathered in leaf order so a leaf is one
// contiguous run of 48-byte records (the reference needs a 32-byte index record plus three
// dependent 16-byte vertex fetches per triangle, tracekernel.ck:159-162).
#include "host_scene.h"
#include "trace_math.h"
#include "flatten_math.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

namespace rt {

// Validation and numbering: one sequential walk of the caller's tree.  Everything after it (records, quantization,
// the four-wide collapse) is per-node work that the host (Flatten) or the device (FlattenOnDevice) can do.
std::string FlattenIndexPass(rt_bvh_node const *nodes, size_t nnodes, rt_vec3 const *verts, size_t nverts,
                             rt_face const *faces, size_t ntris, FlatIndex *ix)
{
  *ix = FlatIndex();
  if (!nodes || nnodes < 3) return "bvh needs a root with two children";
  if (nodes[0].type != RT_BVH_INNER) return "bvh root must be an inner node";
  if (ntris >= (1u << 28)) return "too many triangles for the 28-bit leaf reference";

  // ---- validate + number inner nodes and triangles in depth-first pre-order -------
  ix->inner_id.assign(nnodes, 0xFFFFFFFFu);
  ix->leaf_ref.assign(nnodes, 0);
  std::vector<uint8_t> seen(nnodes, 0);
  struct Item { uint32_t node; int depth; };
  std::vector<Item> todo{{0u, 1}};
  seen[0] = 1;
  int max_depth = 0;
  std::vector<uint32_t> path;   // path[d - 1] = the node at depth d on the way to the node being visited
  // A triangle outside the box of its leaf or of any ancestor would be culled for some rays (and, outside the root
  // box, clamped off the quantization grid): the traversal's promise that the tree never changes a result needs
  // every box on the path to hold it.  Same tolerance as the reference's own check (core/GeomUtil.cpp:94-101).
  auto holds = [](rt_aabb const &b, rt_vec3 const &p) {
    const float e = 0.000001f;
    return p.x >= b.a.x - e && p.x <= b.b.x + e && p.y >= b.a.y - e && p.y <= b.b.y + e && p.z >= b.a.z - e && p.z <= b.b.z + e;
  };
  while (!todo.empty()) {
    Item it = todo.back();
    todo.pop_back();
    rt_bvh_node const &n = nodes[it.node];
    path.resize(static_cast<size_t>(it.depth));
    path[it.depth - 1] = it.node;
    if (n.type == RT_BVH_INNER) {
      ix->inner_id[it.node] = static_cast<uint32_t>(ix->order.size());
      ix->order.push_back(it.node);
      max_depth = std::max(max_depth, it.depth);
      for (int k = 1; k >= 0; --k) {  // child 0 is processed first
        uint32_t c = n.node.inner.children_index[k];
        if (c >= nnodes) return "child index out of range";
        if (seen[c]) return "bvh is not a tree (node reached twice)";
        seen[c] = 1;
        todo.push_back({c, it.depth + 1});
      }
    } else if (n.type == RT_BVH_LEAF) {
      uint32_t cnt = n.node.leaf.count;
      if (cnt > RT_TRIS_PER_LEAF) return "leaf holds more than 4 triangles";
      uint32_t first = static_cast<uint32_t>(ix->tri_of_record.size());
      for (uint32_t j = 0; j < cnt; ++j) {
        uint32_t t = n.node.leaf.tris[j];
        if (t >= ntris) return "leaf triangle index out of range";
        rt_face const &f = faces[t];
        if (f.s[0] >= nverts || f.s[1] >= nverts || f.s[2] >= nverts) return "face vertex index out of range";
        rt_vec3 const &a = verts[f.s[0]], &b = verts[f.s[1]], &c = verts[f.s[2]];
        for (uint32_t anc : path)
          if (!holds(nodes[anc].aabb, a) || !holds(nodes[anc].aabb, b) || !holds(nodes[anc].aabb, c))
            return "box of node " + std::to_string(anc) + " does not contain triangle " + std::to_string(t) + " of the leaf below it";
        ix->tri_of_record.push_back(t);
      }
      ix->leaf_ref[it.node] = kLeafBit | (first << 3) | cnt;
    } else {
      return "unknown node type";
    }
  }
  ix->depth2 = max_depth;

  // ---- quantization grid over the root box (which spans every vertex) --------------------
  const float lo[3] = {nodes[0].aabb.a.x, nodes[0].aabb.a.y, nodes[0].aabb.a.z};
  const float hi[3] = {nodes[0].aabb.b.x, nodes[0].aabb.b.y, nodes[0].aabb.b.z};
  for (int k = 0; k < 3; ++k) {
    ix->grid_lo[k] = lo[k];
    float st = (hi[k] - lo[k]) / 65535.f;
    if (!(st > 0.f)) st = 1e-6f;
    // the top plane must reach the box: widen the step until lo + 65535*step >= hi in exact arithmetic
    while (static_cast<double>(lo[k]) + 65535.0 * st < hi[k]) st = std::nextafter(st, 3.0e38f);
    ix->grid_step[k] = st;
  }
  return std::string();
}

// What does not depend on who builds the records.
void FlattenFinish(rt_bvh_node const *nodes, FlatIndex const &ix, int depth4, FlatScene *out)
{
  for (int k = 0; k < 3; ++k) { out->grid_lo[k] = ix.grid_lo[k]; out->grid_step[k] = ix.grid_step[k]; }
  out->depth2 = ix.depth2;
  out->stack2 = ix.depth2 + 1;
  out->depth4 = depth4;
  out->stack4 = 3 * depth4 + 1;
  out->root_box = nodes[0].aabb;
  float s = 0.f;
  for (float v : {nodes[0].aabb.a.x, nodes[0].aabb.a.y, nodes[0].aabb.a.z, nodes[0].aabb.b.x, nodes[0].aabb.b.y,
                  nodes[0].aabb.b.z})
    s = std::max(s, std::fabs(v));
  out->scale = s;
}

std::string Flatten(rt_bvh_node const *nodes, size_t nnodes, rt_vec3 const *verts, size_t nverts,
                    rt_face const *faces, size_t ntris, FlatScene *out)
{
  *out = FlatScene();
  FlatIndex ix;
  std::string why = FlattenIndexPass(nodes, nnodes, verts, nverts, faces, ntris, &ix);
  if (!why.empty()) return why;

  // ---- triangle records, leaf order ----------------------------------------------------
  out->tris.resize(ix.tri_of_record.size());
  for (size_t i = 0; i < out->tris.size(); ++i) out->tris[i] = MakeTri(faces, verts, ix.tri_of_record[i]);

  auto ref_of = [&](uint32_t node) { return nodes[node].type == RT_BVH_INNER ? ix.inner_id[node] : ix.leaf_ref[node]; };

  // ---- BVH2 records ---------------------------------------------------------------
  out->nodes2.resize(ix.order.size());
  out->nodesq2.resize(ix.order.size() * 2);
  for (size_t i = 0; i < ix.order.size(); ++i) {
    rt_bvh_node const &n = nodes[ix.order[i]];
    MakeNode2(nodes, ix.order[i], ref_of(n.node.inner.children_index[0]), ref_of(n.node.inner.children_index[1]), ix.grid_lo,
              ix.grid_step, &out->nodes2[i], &out->nodesq2[2 * i]);
  }

  // ---- BVH4: one four-wide node per chosen two-wide inner node, numbered breadth first (keeps the top levels
  // dense); ChooseKids4 (flatten_math.h) picks each node's children ---------------------
  std::vector<uint32_t> id4(nnodes, 0xFFFFFFFFu);
  std::vector<std::array<Kid, 4>> kids;
  std::vector<int> nk;
  id4[0] = 0;
  std::vector<uint32_t> roots{0u};
  std::vector<int> depths{1};
  int max4 = 0;
  for (size_t qi = 0; qi < roots.size(); ++qi) {
    max4 = std::max(max4, depths[qi]);
    std::array<Kid, 4> set{};
    const int n = ChooseKids4(nodes, ix.leaf_ref.data(), out->tris.data(), faces, verts, roots[qi], set.data());
    for (int k = 0; k < n; ++k)
      if (set[k].inner) {
        id4[set[k].node] = static_cast<uint32_t>(roots.size());
        roots.push_back(set[k].node);
        depths.push_back(depths[qi] + 1);
      }
    kids.push_back(set);
    nk.push_back(n);
  }
  out->nodes4.resize(roots.size());
  out->nodesq4.resize(roots.size() * 4);
  for (size_t i = 0; i < roots.size(); ++i) {
    uint32_t child_ref[4] = {0, 0, 0, 0};
    for (int k = 0; k < nk[i]; ++k)
      child_ref[k] = kids[i][k].inner ? id4[kids[i][k].node] : (kLeafBit | (kids[i][k].first << 3) | kids[i][k].cnt);
    MakeNode4(kids[i].data(), nk[i], child_ref, ix.grid_lo, ix.grid_step, &out->nodes4[i], &out->nodesq4[4 * i]);
  }
  FlattenFinish(nodes, ix, max4, out);
  return std::string();
}

namespace {

struct StackEntry { uint32_t ref; float t; };

// The device's Pop(): entries whose box starts beyond the best hit are dropped unfetched.
inline uint32_t PopHost(std::vector<StackEntry> &stack, float tbest)
{
  while (!stack.empty()) {
    const StackEntry e = stack.back();
    stack.pop_back();
    if (e.t <= tbest) return e.ref;
  }
  return 0xFFFFFFFFu;
}

// The device's Descend2()/Descend4(): continue with the nearest hit child (lowest slot on equal distance), push the
// others, highest slot deepest.  hit_ref/hit_t/hit_slot list the hit children in slot order.
inline uint32_t DescendHost(std::vector<StackEntry> &stack, const uint32_t *hit_ref, const float *hit_t, int nh, float tbest)
{
  if (nh == 0) return PopHost(stack, tbest);
  int m = 0;
  for (int i = 1; i < nh; ++i)
    if (hit_t[i] < hit_t[m]) m = i;
  for (int i = nh - 1; i >= 0; --i)
    if (i != m) stack.push_back({hit_ref[i], hit_t[i]});
  return hit_ref[m];
}

// Slab tests of one node of the selected layout; fills the hit children in slot order.
inline int TestNodeHost(FlatScene const &fs, int width, bool quant, const RayCtx &r, const RayCtxQ &rq, uint32_t cur, float tbest,
                        uint32_t *hit_ref, float *hit_t)
{
  int nh = 0;
  if (width == 8) {
    // An eight-wide node that does not exist as a record: the two-wide node `cur` with its descendants three levels down
    // (leaves stop the descent earlier), boxes from the quantized two-wide records.  What a node built by collapsing three
    // levels of the binary tree would hold; for the wave model only.
    GpuChildQ const *cand[8];
    int nc = 0;
    uint32_t level[4] = {cur, 0, 0, 0};
    int nlevel = 1;
    for (int depth = 0; depth < 3; ++depth) {
      uint32_t next[8];
      int nnext = 0;
      for (int i = 0; i < nlevel; ++i)
        for (int k = 0; k < 2; ++k) {
          GpuChildQ const *c = fs.nodesq2.data() + static_cast<size_t>(level[i]) * 2 + k;
          if ((c->ref & kLeafBit) || depth == 2) cand[nc++] = c;
          else next[nnext++] = c->ref;
        }
      nlevel = nnext;
      for (int i = 0; i < nnext; ++i) level[i] = next[i];
    }
    for (int i = 0; i < nc; ++i) {
      float te;
      if (SlabTestQ(rq, cand[i]->x, cand[i]->y, cand[i]->z, tbest, te)) { hit_ref[nh] = cand[i]->ref; hit_t[nh++] = te; }
    }
  } else if (quant) {
    const int w = width == 4 ? 4 : 2;
    GpuChildQ const *c = (w == 4 ? fs.nodesq4.data() : fs.nodesq2.data()) + static_cast<size_t>(cur) * w;
    for (int i = 0; i < w; ++i) {
      float te;
      if (SlabTestQ(rq, c[i].x, c[i].y, c[i].z, tbest, te)) { hit_ref[nh] = c[i].ref; hit_t[nh++] = te; }
    }
  } else if (width == 4) {
    GpuNode4 const &n = fs.nodes4[cur];
    for (int c = 0; c < 4; ++c) {
      float te;
      if (SlabTest(r, n.lox[c], n.loy[c], n.loz[c], n.hix[c], n.hiy[c], n.hiz[c], tbest, te)) { hit_ref[nh] = n.child[c]; hit_t[nh++] = te; }
    }
  } else {
    GpuNode2 const &n = fs.nodes2[cur];
    float te;
    if (SlabTest(r, n.lo0[0], n.lo0[1], n.lo0[2], n.hi0[0], n.hi0[1], n.hi0[2], tbest, te)) { hit_ref[nh] = n.child[0]; hit_t[nh++] = te; }
    if (SlabTest(r, n.lo1[0], n.lo1[1], n.lo1[2], n.hi1[0], n.hi1[1], n.hi1[2], tbest, te)) { hit_ref[nh] = n.child[1]; hit_t[nh++] = te; }
  }
  return nh;
}

}  // namespace

// Host mirror of the device traversal (same order, same arithmetic) that only counts.
FetchCounts CountFetches(FlatScene const &fs, int width, bool quant, rt_ray_isect const *rays, size_t nrays,
                         size_t stride, uint32_t *per_ray)
{
  QuantGrid grid;
  SetQuantGrid(grid, fs.grid_lo, fs.grid_step);
  FetchCounts fc;
  if (stride == 0) stride = 1;
  std::vector<StackEntry> stack;
  for (size_t k = 0; k < nrays; k += stride) {
    RayCtx r;
    SetupRay(r, rays[k].origin.x, rays[k].origin.y, rays[k].origin.z, rays[k].direction.x, rays[k].direction.y,
             rays[k].direction.z, fs.scale);
    RayCtxQ rq;
    SetupRayQ(rq, r, grid, fs.scale);
    float tbest = kFltMax;
    uint32_t best = 0xFFFFFFFFu;
    stack.clear();
    uint32_t cur = 0;
    const uint64_t nodes_before = fc.nodes, tris_before = fc.tris;
    while (cur != 0xFFFFFFFFu) {
      if (!(cur & kLeafBit)) {
        fc.nodes++;
        uint32_t hit_ref[4];
        float hit_t[4];
        const int nh = TestNodeHost(fs, width, quant, r, rq, cur, tbest, hit_ref, hit_t);
        cur = DescendHost(stack, hit_ref, hit_t, nh, tbest);
        fc.max_stack = std::max<uint64_t>(fc.max_stack, stack.size());
      } else {
        const uint32_t first = (cur & ~kLeafBit) >> 3, cnt = cur & 7u;
        if (cnt) fc.leaves++;
        for (uint32_t j = 0; j < cnt; ++j) {
          GpuTri const &g = fs.tris[first + j];
          fc.tris++;
          float t, u, v;
          if (TriTest(g.ax, g.ay, g.az, g.e1x, g.e1y, g.e1z, g.e2x, g.e2y, g.e2z, r, t, u, v) &&
              (t < tbest || (t == tbest && g.id < best))) {
            tbest = t;
            best = g.id;
          }
        }
        cur = PopHost(stack, tbest);
      }
    }
    if (best != 0xFFFFFFFFu) fc.rays_hit++;
    if (per_ray) {
      per_ray[2 * (k / stride)] = static_cast<uint32_t>(fc.nodes - nodes_before);
      per_ray[2 * (k / stride) + 1] = static_cast<uint32_t>(fc.tris - tris_before);
    }
  }
  return fc;
}

}  // namespace rt

// ---------------------------------------------------------------------------------------
// Lockstep model of one wave running the traversal kernel: 64 lanes share a program counter,
// so a phase lasts as long as its slowest lane.  It reports how many wave-level node steps and
// triangle steps are issued and how many of their lane slots do useful work -- the quantity the
// VALU-utilisation counter measures -- for the static 64-rays-per-wave assignment and for waves
// that refill idle lanes from the ray stream once `refill` or more lanes have finished.
// A planning aid; it computes no hits that anything consumes.
namespace rt {

namespace {

struct LaneSim {
  RayCtx r;
  RayCtxQ rq;
  float tbest = kFltMax;
  uint32_t best = 0xFFFFFFFFu;
  uint32_t cur = 0xFFFFFFFFu;  // kDone
  std::vector<StackEntry> stack;
  bool busy = false;
};

}  // namespace

WaveModel SimulateWaves(FlatScene const &fs, int width, bool quant, rt_ray_isect const *rays, size_t nrays,
                        uint32_t img_w, uint32_t refill, size_t max_waves, int policy)
{
  WaveModel m;
  QuantGrid grid;
  SetQuantGrid(grid, fs.grid_lo, fs.grid_step);
  const uint32_t kDoneRef = 0xFFFFFFFFu;
  const uint32_t img_h = img_w ? static_cast<uint32_t>(nrays / img_w) : 0;
  const uint32_t tiles_x = img_w ? (img_w + 7) / 8 : 0;
  const size_t ntasks = img_w ? static_cast<size_t>(tiles_x) * ((img_h + 7) / 8) : (nrays + 63) / 64;
  // ray order seen by the stream: task-major, lane-minor; invalid slots skipped
  auto ray_of = [&](size_t k) -> long long {
    const size_t task = k / 64, lane = k % 64;
    if (!img_w) return k < nrays ? static_cast<long long>(k) : -1;
    const uint32_t x = static_cast<uint32_t>(task % tiles_x) * 8 + (lane & 7), y = static_cast<uint32_t>(task / tiles_x) * 8 + (lane >> 3);
    return (x < img_w && y < img_h) ? static_cast<long long>(y) * img_w + x : -1;
  };
  const size_t stream_len = ntasks * 64;
  // waves are sampled evenly over the stream; with refill each simulated wave owns a contiguous span
  const size_t nwaves = std::min(max_waves ? max_waves : ntasks, ntasks);
  const size_t span = (refill && refill < 1000u) ? 64 * 8 : 64;  // a refilling wave works through 8 tiles' worth of rays
  std::vector<LaneSim> lanes(64);
  for (size_t w = 0; w < nwaves; ++w) {
    size_t pos = (ntasks * w / nwaves) * 64, end = std::min(stream_len, pos + span);
    for (auto &l : lanes) { l.busy = false; l.cur = kDoneRef; }
    auto assign = [&](LaneSim &l) -> bool {
      while (pos < end) {
        long long ri = ray_of(pos++);
        if (ri < 0) continue;
        rt_ray_isect const &ry = rays[ri];
        SetupRay(l.r, ry.origin.x, ry.origin.y, ry.origin.z, ry.direction.x, ry.direction.y, ry.direction.z, fs.scale);
        SetupRayQ(l.rq, l.r, grid, fs.scale);
        l.tbest = kFltMax; l.best = 0xFFFFFFFFu; l.cur = 0; l.stack.clear(); l.busy = true;
        m.rays++;
        return true;
      }
      return false;
    };
    for (auto &l : lanes) assign(l);
    m.waves++;
    const uint64_t wave_steps_before = m.wave_node_steps + m.wave_tri_steps;
    uint64_t wave_iterations = 0;   // loop iterations of this wave = memory latencies it exposes
    for (;;) {
      // retire / refill
      uint32_t idle = 0, active = 0;
      for (auto &l : lanes) { if (l.cur == kDoneRef) idle++; else active++; }
      if (refill && refill < 1000u && idle >= refill && pos < end) {
        uint32_t got = 0;
        for (auto &l : lanes) if (l.cur == kDoneRef && assign(l)) got++;
        if (got) { m.refills++; m.refill_lanes += got; }
        active += got;
      }
      // optional model of intra-wave work donation: idle lanes take the bottom stack entry of the busiest lane
      if (refill >= 1000u && active > 0 && idle >= refill - 1000u) {
        uint32_t events = 0;
        for (auto &h : lanes) {
          if (h.cur != kDoneRef) continue;
          LaneSim *donor = nullptr;
          for (auto &d : lanes) if (d.cur != kDoneRef && d.stack.size() >= 2 && (!donor || d.stack.size() > donor->stack.size())) donor = &d;
          if (!donor) break;
          h.r = donor->r; h.rq = donor->rq; h.tbest = donor->tbest; h.best = donor->best;
          h.cur = donor->stack.front().ref;
          donor->stack.erase(donor->stack.begin());
          h.stack.clear();
          events++;
          active++;
        }
        if (events) { m.refills++; m.refill_lanes += events; }
      }
      if (active == 0) {
        m.max_wave_steps = std::max(m.max_wave_steps, m.wave_node_steps + m.wave_tri_steps - wave_steps_before);
        m.max_wave_iterations = std::max(m.max_wave_iterations, wave_iterations);
        break;
      }
      if (policy < 0) {
        // fused iteration: the lanes holding an inner reference take one node step and, in the same iteration, the
        // lanes parked at a leaf test up to -policy of its triangles (both kinds of record are fetched together, so
        // the iteration exposes one memory latency); a lane does one kind of work per iteration
        const uint32_t per_iter = static_cast<uint32_t>(-policy);
        uint32_t n_inner = 0, n_leaf = 0, maxk = 0;
        for (auto &l : lanes) { if (!(l.cur & kLeafBit)) n_inner++; else if (l.cur != kDoneRef) n_leaf++; }
        m.iterations++;
        wave_iterations++;
        for (auto &l : lanes) {
          if (l.cur == kDoneRef) continue;
          if (!(l.cur & kLeafBit)) {
            uint32_t hit_ref[8]; float hit_t[8];
            const int nh = TestNodeHost(fs, width, quant, l.r, l.rq, l.cur, l.tbest, hit_ref, hit_t);
            l.cur = DescendHost(l.stack, hit_ref, hit_t, nh, l.tbest);
            continue;
          }
          const uint32_t first = (l.cur & ~kLeafBit) >> 3, cnt = l.cur & 7u, now = std::min(cnt, per_iter);
          maxk = std::max(maxk, now);
          m.lane_tri_steps += now;
          for (uint32_t j = 0; j < now; ++j) {
            GpuTri const &g = fs.tris[first + j];
            float t, u, v;
            if (TriTest(g.ax, g.ay, g.az, g.e1x, g.e1y, g.e1z, g.e2x, g.e2y, g.e2z, l.r, t, u, v) &&
                (t < l.tbest || (t == l.tbest && g.id < l.best))) { l.tbest = t; l.best = g.id; }
          }
          if (cnt > now) l.cur = kLeafBit | ((first + now) << 3) | (cnt - now);
          else l.cur = PopHost(l.stack, l.tbest);
        }
        if (n_inner) { m.wave_node_steps++; m.lane_node_steps += n_inner; }
        if (n_leaf) { m.wave_leaf_phases++; m.lane_leaf_phases += n_leaf; }
        m.wave_tri_steps += maxk;
        continue;
      }
      // node phase: until no lane holds an inner reference
      for (int node_iter = 0;; ++node_iter) {
        if (policy == 1 && node_iter == 1) break;  // if-if: one node step, then the leaf part, every iteration
        uint32_t n_inner = 0, n_leaf = 0;
        for (auto &l : lanes) { if (!(l.cur & kLeafBit)) n_inner++; else if (l.cur != kDoneRef) n_leaf++; }
        if (!n_inner) break;
        // policy >= 2: leave the node loop once `policy` percent of the busy lanes are parked at a leaf
        if (policy >= 2 && n_leaf * 100u >= static_cast<uint32_t>(policy) * (n_inner + n_leaf)) break;
        m.wave_node_steps++;
        m.lane_node_steps += n_inner;
        m.iterations++;
        wave_iterations++;
        {
          uint32_t first_ref = kDoneRef;
          bool same = true;
          for (auto &l : lanes)
            if (!(l.cur & kLeafBit)) {
              if (first_ref == kDoneRef) first_ref = l.cur;
              else if (l.cur != first_ref) same = false;
            }
          if (same) m.uniform_node_steps++;
          else {
            // how many different nodes the stepping lanes hold (2 and 3 counted: candidates for a per-node scalar fetch)
            uint32_t seen_refs[4];
            int distinct = 0;
            for (auto &l : lanes) {
              if (l.cur & kLeafBit) continue;
              bool known = false;
              for (int k = 0; k < distinct && k < 4; ++k) known = known || seen_refs[k] == l.cur;
              if (!known) { if (distinct < 4) seen_refs[distinct] = l.cur; ++distinct; if (distinct > 3) break; }
            }
            if (distinct == 2) m.two_node_steps++;
            else if (distinct == 3) m.three_node_steps++;
          }
        }
        for (auto &l : lanes) {
          if (l.cur & kLeafBit) continue;
          uint32_t hit_ref[8]; float hit_t[8];
          const int nh = TestNodeHost(fs, width, quant, l.r, l.rq, l.cur, l.tbest, hit_ref, hit_t);
          l.cur = DescendHost(l.stack, hit_ref, hit_t, nh, l.tbest);
        }
      }
      // leaf phase: the wave loops max(count) times
      uint32_t maxcnt = 0, nleaf = 0;
      for (auto &l : lanes) if (l.cur != kDoneRef && (l.cur & kLeafBit)) { maxcnt = std::max(maxcnt, l.cur & 7u); nleaf++; }
      if (nleaf) { m.wave_leaf_phases++; m.lane_leaf_phases += nleaf; }
      if (nleaf) {   // the kernel fetches a leaf's records two at a time: one exposed latency per pair
        m.iterations += (maxcnt + 1u) / 2u;
        wave_iterations += (maxcnt + 1u) / 2u;
      }
      m.wave_tri_steps += maxcnt;
      for (auto &l : lanes) {
        if (l.cur == kDoneRef || !(l.cur & kLeafBit)) continue;
        const uint32_t first = (l.cur & ~kLeafBit) >> 3, cnt = l.cur & 7u;
        m.lane_tri_steps += cnt;
        for (uint32_t j = 0; j < cnt; ++j) {
          GpuTri const &g = fs.tris[first + j];
          float t, u, v;
          if (TriTest(g.ax, g.ay, g.az, g.e1x, g.e1y, g.e1z, g.e2x, g.e2y, g.e2z, l.r, t, u, v) &&
              (t < l.tbest || (t == l.tbest && g.id < l.best))) { l.tbest = t; l.best = g.id; }
        }
        l.cur = PopHost(l.stack, l.tbest);
      }
    }
  }
  return m;
}

}  // namespace rt
