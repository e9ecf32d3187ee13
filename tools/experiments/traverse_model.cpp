// Host-side what-if model of traversal variants: counts node/triangle records per ray, stack depths and the
// lockstep cost of 8x8-tile waves for several node widths, orderings and stack-entry formats.
// A planning aid; computes no hits that anything consumes.
// Build: g++ -std=c++17 -O2 -ffp-contract=off -Iinclude -Iraytrace_amd/csrc tools/experiments/traverse_model.cpp build/{obj_loader,centroid_bvh,sah_bvh,raygen,flatten,framebuffer}.o -o build/traverse_model
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_scene.h"
#include "trace_math.h"

using namespace rt;

struct Variant {
  const char *name;
  int width;        // 2 or 4
  bool popcull;     // stack entries carry the entry distance; a popped entry beyond tbest is dropped unfetched
  int order;        // 0 = full sort by entry distance, 1 = nearest first, rest in slot order, 2 = slot order
  int leaf_pct;
  bool single_tri = false;  // a leaf step tests ONE triangle per parked lane
};

struct Lane {
  RayCtx r; RayCtxQ rq;
  float tbest; uint32_t best; uint32_t cur;
  std::vector<std::pair<uint32_t, float>> stk;
  uint32_t tri_j;
};

int main(int argc, char **argv)
{
  const char *model = argc > 1 ? argv[1] : "assets/models/dragon.obj";
  const int W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
  const int tile_stride = argc > 4 ? atoi(argv[4]) : 4;   // model every k-th tile
  const char *bvhmode = argc > 5 ? argv[5] : "fixed";
  Mesh mesh;
  if (!LoadObjFile(model, mesh, kLoaderTolerant)) return 1;
  CentroidBVH bvh;
  if (!strcmp(bvhmode, "sah")) SahBVHBuild(&bvh, mesh.faces.data(), mesh.verts.data(), mesh.faces.size(), mesh.verts.size());
  else CentroidBVHBuild(&bvh, mesh.faces.data(), mesh.verts.data(), mesh.faces.size(), mesh.verts.size(), kBvhFixed);
  FlatScene fs;
  std::string why = Flatten(bvh.nodes.data(), bvh.nodes.size(), mesh.verts.data(), mesh.verts.size(), mesh.faces.data(), mesh.faces.size(), &fs);
  if (!why.empty()) { printf("flatten: %s\n", why.c_str()); return 1; }
  float origin[3];
  FrameCamera(bvh.nodes[0].aabb, W, H, origin);
  std::vector<rt_ray_isect> rays(size_t(W) * H);
  PrimaryRays(W, H, origin, rays.data());
  QuantGrid grid; SetQuantGrid(grid, fs.grid_lo, fs.grid_step);
  printf("%s %dx%d bvh=%s nodes2=%zu nodes4=%zu tris=%zu depth2=%d depth4=%d\n", model, W, H, bvhmode, fs.nodes2.size(), fs.nodes4.size(), fs.tris.size(), fs.depth2, fs.depth4);

  const Variant variants[] = {
    {"w2 sort            ", 2, false, 0, 50},
    {"w2 sort popcull    ", 2, true, 0, 50},
    {"w4 sort            ", 4, false, 0, 50},
    {"w4 sort popcull    ", 4, true, 0, 50},
    {"w4 nearest popcull ", 4, true, 1, 50},
    {"w4 nearest         ", 4, false, 1, 50},
    {"w4 sort popcull l30", 4, true, 0, 30},
    {"w4 sort popcull l70", 4, true, 0, 70},
    {"w2 sort single-tri ", 2, false, 0, 50, true},
    {"w4 near pc single  ", 4, true, 1, 50, true},
    {"w4 near pc single70", 4, true, 1, 70, true},
    {"w4 near pc single30", 4, true, 1, 30, true},
  };
  const uint32_t kDoneRef = 0xFFFFFFFFu;
  const int tx = (W + 7) / 8, ty = (H + 7) / 8;
  for (auto const &v : variants) {
    uint64_t rays_n = 0, nodes = 0, tris = 0, waves = 0, wave_node = 0, wave_tri = 0, lane_node = 0, lane_tri = 0, max_wave = 0, culled = 0;
    uint64_t depth_hist[64] = {0};   // stack depth after each push
    uint64_t pushes = 0; size_t max_depth = 0;
    std::vector<uint64_t> wave_cost;
    std::vector<Lane> lanes(64);
    for (int t = 0; t < tx * ty; t += tile_stride) {
      const int x0 = (t % tx) * 8, y0 = (t / tx) * 8;
      int nl = 0;
      for (int l = 0; l < 64; ++l) {
        const int x = x0 + (l & 7), y = y0 + (l >> 3);
        Lane &L = lanes[l];
        L.cur = kDoneRef; L.stk.clear();
        if (x >= W || y >= H) continue;
        rt_ray_isect const &ry = rays[size_t(y) * W + x];
        SetupRay(L.r, ry.origin.x, ry.origin.y, ry.origin.z, ry.direction.x, ry.direction.y, ry.direction.z, fs.scale);
        SetupRayQ(L.rq, L.r, grid, fs.scale);
        L.tbest = kFltMax; L.best = kDoneRef; L.cur = 0; L.tri_j = 0;
        nl++; rays_n++;
      }
      waves++;
      uint64_t wn = 0, wt = 0;
      for (;;) {
        uint32_t ni = 0, nleaf = 0;
        for (auto &L : lanes) { if (L.cur == kDoneRef) continue; if (L.cur & kLeafBit) nleaf++; else ni++; }
        if (ni + nleaf == 0) break;
        if (ni != 0 && nleaf * 100u < uint32_t(v.leaf_pct) * (ni + nleaf)) {
          wn++; lane_node += ni;
          for (auto &L : lanes) {
            if (L.cur == kDoneRef || (L.cur & kLeafBit)) continue;
            nodes++;
            uint32_t href[4]; float ht[4]; int nh = 0;
            GpuChildQ const *c = (v.width == 4 ? fs.nodesq4.data() : fs.nodesq2.data()) + size_t(L.cur) * v.width;
            for (int i = 0; i < v.width; ++i) { float te; if (SlabTestQ(L.rq, c[i].x, c[i].y, c[i].z, L.tbest, te)) { href[nh] = c[i].ref; ht[nh++] = te; } }
            if (v.order == 0) {
              for (int i = 1; i < nh; ++i) for (int j = i; j > 0 && ht[j] < ht[j - 1]; --j) { std::swap(ht[j], ht[j - 1]); std::swap(href[j], href[j - 1]); }
            } else if (v.order == 1 && nh > 1) {
              int m = 0; for (int i = 1; i < nh; ++i) if (ht[i] < ht[m]) m = i;
              std::swap(ht[0], ht[m]); std::swap(href[0], href[m]);
            }
            for (int i = nh - 1; i >= 1; --i) { L.stk.push_back({href[i], ht[i]}); pushes++; depth_hist[std::min<size_t>(63, L.stk.size())]++; max_depth = std::max(max_depth, L.stk.size()); }
            if (nh) L.cur = href[0];
            else {
              L.cur = kDoneRef;
              while (!L.stk.empty()) { auto e = L.stk.back(); L.stk.pop_back(); if (v.popcull && e.second > L.tbest) { culled++; continue; } L.cur = e.first; break; }
            }
          }
        } else {
          uint32_t maxcnt = 0;
          for (auto &L : lanes) if (L.cur != kDoneRef && (L.cur & kLeafBit)) maxcnt = std::max(maxcnt, L.cur & 7u);
          wt += v.single_tri ? 1 : maxcnt;
          for (auto &L : lanes) {
            if (L.cur == kDoneRef || !(L.cur & kLeafBit)) continue;
            const uint32_t first = (L.cur & ~kLeafBit) >> 3, cnt = L.cur & 7u;
            if (v.single_tri) {
              if (L.tri_j < cnt) {
                lane_tri++; tris++;
                GpuTri const &g = fs.tris[first + L.tri_j];
                float tt, u, vv;
                if (TriTest(g.ax, g.ay, g.az, g.e1x, g.e1y, g.e1z, g.e2x, g.e2y, g.e2z, L.r, tt, u, vv) && (tt < L.tbest || (tt == L.tbest && g.id < L.best))) { L.tbest = tt; L.best = g.id; }
                L.tri_j++;
              }
              if (L.tri_j < cnt) continue;
              L.tri_j = 0;
              L.cur = kDoneRef;
              while (!L.stk.empty()) { auto e = L.stk.back(); L.stk.pop_back(); if (v.popcull && e.second > L.tbest) { culled++; continue; } L.cur = e.first; break; }
              continue;
            }
            lane_tri += cnt; tris += cnt;
            for (uint32_t j = 0; j < cnt; ++j) {
              GpuTri const &g = fs.tris[first + j];
              float tt, u, vv;
              if (TriTest(g.ax, g.ay, g.az, g.e1x, g.e1y, g.e1z, g.e2x, g.e2y, g.e2z, L.r, tt, u, vv) && (tt < L.tbest || (tt == L.tbest && g.id < L.best))) { L.tbest = tt; L.best = g.id; }
            }
            L.cur = kDoneRef;
            while (!L.stk.empty()) { auto e = L.stk.back(); L.stk.pop_back(); if (v.popcull && e.second > L.tbest) { culled++; continue; } L.cur = e.first; break; }
          }
        }
      }
      wave_node += wn; wave_tri += wt;
      max_wave = std::max(max_wave, wn + wt);
      wave_cost.push_back(wn + wt);
    }
    uint64_t over8 = 0, over12 = 0, over16 = 0;
    for (int d = 0; d < 64; ++d) { if (d > 8) over8 += depth_hist[d]; if (d > 12) over12 += depth_hist[d]; if (d > 16) over16 += depth_hist[d]; }
    std::sort(wave_cost.begin(), wave_cost.end());
    printf("%s nodes/ray %.2f tris/ray %.2f | per wave: node steps %.1f tri steps %.1f (lane util node %.2f tri %.2f) longest %llu p99 %llu | culled pops/ray %.2f | max depth %zu, pushes >8: %.4f >12: %.5f >16: %.6f\n",
           v.name, double(nodes) / rays_n, double(tris) / rays_n, double(wave_node) / waves, double(wave_tri) / waves,
           double(lane_node) / (wave_node * 64.0), double(lane_tri) / (wave_tri * 64.0), (unsigned long long)max_wave,
           (unsigned long long)wave_cost[wave_cost.size() * 99 / 100], double(culled) / rays_n, max_depth,
           double(over8) / pushes, double(over12) / pushes, double(over16) / pushes);
  }
  return 0;
}
