// rt_headless -- headless stand-in for the reference's windowed client
// (render_client/img_main.cpp): loads a model, builds the BVH, creates the device scene, runs
// the two passes of one frame (primary rays, then one reflection bounce, img_main.cpp:447-498),
// composites the two RGBA32F layers the way gl_update blends its two textures
// (img_main.cpp:434,512-526: layer 0 opaque, layer 1 with SRC_ALPHA / ONE_MINUS_DST_ALPHA) and
// writes a BMP instead of presenting a window.
//
//   rt_headless [--model PATH] [--width W] [--height H] [--loader reference|tolerant]
//               [--bvh reference|fixed|sah] [--wide 2|4] [--bounces N] [--device D] [--out FILE.bmp]
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_amd.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    if ((call) != 0) {                                                     \
      std::fprintf(stderr, "%s failed: %s\n", #call, rt_last_error());     \
      return 1;                                                            \
    }                                                                      \
  } while (0)
#define HIPCHECK(call)                                                     \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));      \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(int argc, char **argv)
{
  std::string model = "assets/models/cube_mirror_test.obj", out = "frame.bmp";
  int W = 560, H = 560, wide = 2, bounces = 1, device = 0;
  unsigned loader = RT_LOADER_TOLERANT, bvhmode = RT_BVH_FIXED;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--model") model = next();
    else if (a == "--width") W = std::atoi(next());
    else if (a == "--height") H = std::atoi(next());
    else if (a == "--loader") loader = std::string(next()) == "reference" ? RT_LOADER_REFERENCE : RT_LOADER_TOLERANT;
    else if (a == "--bvh") {
      const std::string m = next();
      bvhmode = m == "reference" ? RT_BVH_REFERENCE : (m == "sah" ? RT_BVH_SAH : RT_BVH_FIXED);
    }
    else if (a == "--wide") wide = std::atoi(next());
    else if (a == "--bounces") bounces = std::atoi(next());
    else if (a == "--device") device = std::atoi(next());
    else if (a == "--out") out = next();
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (W <= 0 || H <= 0 || bounces < 0 || (wide != 2 && wide != 4)) { std::fprintf(stderr, "bad arguments\n"); return 2; }

  rt_mesh *mesh = nullptr;
  CHECK(rt_mesh_load_obj(model.c_str(), loader, &mesh));
  std::printf("model loaded: %zu vertices, %zu triangles, %zu normals, %zu materials\n", rt_mesh_nverts(mesh),
              rt_mesh_ntris(mesh), rt_mesh_nnormals(mesh), rt_mesh_nmaterials(mesh));
  rt_bvh *bvh = nullptr;
  CHECK(rt_bvh_build(rt_mesh_faces(mesh), rt_mesh_verts(mesh), rt_mesh_ntris(mesh), rt_mesh_nverts(mesh), bvhmode, &bvh));
  // the statistics the reference prints (img_main.cpp:287-290)
  std::printf("Triangles: %zu\nMax depth: %d\nNode count: %zx\nAvg tris per leaf: %f\n", rt_mesh_ntris(mesh),
              rt_bvh_depth(bvh), rt_bvh_nnodes(bvh), static_cast<float>(rt_mesh_ntris(mesh)) / rt_bvh_leafcount(bvh));
  rt_scene *scene = nullptr;
  CHECK(rt_scene_create(rt_mesh_verts(mesh), rt_mesh_nverts(mesh), rt_mesh_faces(mesh), rt_mesh_ntris(mesh),
                        rt_mesh_normals(mesh), rt_mesh_nnormals(mesh), rt_mesh_materials(mesh), rt_mesh_nmaterials(mesh),
                        rt_bvh_nodes(bvh), rt_bvh_nnodes(bvh), device, wide == 4 ? RT_SCENE_WIDE4 : RT_SCENE_WIDE2, &scene));

  const size_t n = static_cast<size_t>(W) * H;
  std::vector<rt_ray_isect> rays(n);
  float origin[3];
  CHECK(rt_frame_camera(&rt_bvh_nodes(bvh)[0].aabb, W, H, origin));
  CHECK(rt_primary_rays(W, H, origin, rays.data()));

  HIPCHECK(hipSetDevice(device));
  void *d_cam = nullptr, *d_a = nullptr, *d_b = nullptr;
  float *d_rgba = nullptr;
  HIPCHECK(hipMalloc(&d_cam, n * sizeof(rt_ray_isect)));
  HIPCHECK(hipMalloc(&d_a, n * sizeof(rt_ray_isect)));
  HIPCHECK(hipMalloc(&d_b, n * sizeof(rt_ray_isect)));
  HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_rgba), n * 4 * sizeof(float)));
  HIPCHECK(hipMemcpy(d_cam, rays.data(), n * sizeof(rt_ray_isect), hipMemcpyHostToDevice));

  const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<float> frame(n * 4, 0.f), layer(n * 4);
  auto t0 = std::chrono::steady_clock::now();
  void *src = d_cam, *dst = d_a;
  for (int pass = 0; pass <= bounces; ++pass) {
    CHECK(rt_trace_pass(scene, src, dst, d_rgba, n, identity, pass == 0 ? 0u : 1u, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(layer.data(), d_rgba, n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      float *f = &frame[4 * i];
      const float *l = &layer[4 * i];
      if (pass == 0) {
        std::memcpy(f, l, 4 * sizeof(float));
      } else {  // glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_DST_ALPHA)
        const float sa = l[3], da = f[3];
        for (int c = 0; c < 4; ++c) f[c] = l[c] * sa + f[c] * (1.f - da);
      }
    }
    src = dst;
    dst = (dst == d_a) ? d_b : d_a;
  }
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%d pass(es) over %zu rays in %.3f ms (including layer download)\n", bounces + 1, n, ms);

  std::vector<uint8_t> bgra(n * 4);
  for (size_t i = 0; i < n; ++i) {
    auto q = [](float x) { return static_cast<uint8_t>(std::min(255.f, std::max(0.f, x * 255.f + 0.5f))); };
    bgra[4 * i + 0] = q(frame[4 * i + 2]);
    bgra[4 * i + 1] = q(frame[4 * i + 1]);
    bgra[4 * i + 2] = q(frame[4 * i + 0]);
    bgra[4 * i + 3] = 255;
  }
  CHECK(rt_write_bmp32(out.c_str(), bgra.data(), W, H));
  std::printf("wrote %s\n", out.c_str());

  (void)hipFree(d_cam); (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_rgba);
  rt_scene_destroy(scene);
  rt_bvh_destroy(bvh);
  rt_mesh_destroy(mesh);
  return 0;
}
