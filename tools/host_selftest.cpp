// host_selftest -- drives the host-only half of the library (OBJ/MTL load, the three BVH builders,
// re-flattening, fetch counting, ray generators, image writers) so that it can be run under
// AddressSanitizer / UndefinedBehaviorSanitizer (`make sanitize-host`).  Touches no GPU.
//   host_selftest MODELS_DIR
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../raytrace_amd/csrc/host_scene.h"

int main(int argc, char **argv)
{
  const std::string dir = argc > 1 ? argv[1] : "assets/models";
  const char *models[] = {"cube.obj", "cube_mirror_test.obj", "teapot.obj", "tri.obj", "std_square.obj", "sphere.obj",
                          "wheel.obj", "bunny.obj"};
  int failures = 0;
  for (const char *name : models) {
    for (unsigned loader = 0; loader < 2; ++loader) {
      rt::Mesh mesh;
      if (!rt::LoadObjFile(dir + "/" + name, mesh, static_cast<rt::LoaderMode>(loader))) {
        std::printf("%s: cannot open\n", name);
        failures++;
        continue;
      }
      if (mesh.faces.empty()) continue;   // the reference loader reads nothing usable from some files
      for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0 && mesh.faces.size() > 5000) continue;   // the tree as written is huge on big models
        rt::CentroidBVH bvh;
        if (mode == 2) rt::SahBVHBuild(&bvh, mesh.faces.data(), mesh.verts.data(), mesh.faces.size(), mesh.verts.size());
        else rt::CentroidBVHBuild(&bvh, mesh.faces.data(), mesh.verts.data(), mesh.faces.size(), mesh.verts.size(),
                                  static_cast<rt::BvhMode>(mode));
        rt::FlatScene flat;
        const std::string why = rt::Flatten(bvh.nodes.data(), bvh.nodes.size(), mesh.verts.data(), mesh.verts.size(),
                                            mesh.faces.data(), mesh.faces.size(), &flat);
        if (!why.empty()) {
          std::printf("%s loader %u bvh %d: flatten refused: %s\n", name, loader, mode, why.c_str());
          failures++;
          continue;
        }
        const int W = 96, H = 64;
        std::vector<rt_ray_isect> rays(static_cast<size_t>(W) * H);
        float origin[3];
        rt::FrameCamera(bvh.nodes[0].aabb, W, H, origin);
        rt::PrimaryRays(W, H, origin, rays.data());
        uint64_t hit[4];
        int k = 0;
        for (int width : {2, 4})
          for (bool quant : {false, true}) hit[k++] = rt::CountFetches(flat, width, quant, rays.data(), rays.size(), 1).rays_hit;
        if (hit[0] != hit[1] || hit[0] != hit[2] || hit[0] != hit[3]) {
          std::printf("%s loader %u bvh %d: layouts disagree on the number of rays that hit\n", name, loader, mode);
          failures++;
        }
        rt::SimulateWaves(flat, 2, true, rays.data(), rays.size(), W, 0, 16, 50);
        std::vector<rt_ray_isect> ao(1000);
        rt::AoRays(mesh.verts.data(), mesh.faces.data(), mesh.faces.size(), bvh.nodes[0].aabb, 7, 0, ao.size(), ao.data());
      }
    }
  }
  std::vector<uint8_t> px(16 * 8 * 4, 200);
  std::vector<float> fpx(16 * 8 * 4, 0.5f);
  if (!rt::WriteBmp32("/tmp/host_selftest.bmp", px.data(), 16, 8) || !rt::WritePfm("/tmp/host_selftest.pfm", fpx.data(), 16, 8)) failures++;
  std::printf("host_selftest: %d failure(s)\n", failures);
  return failures ? 1 : 0;
}
