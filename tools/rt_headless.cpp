// rt_headless -- headless stand-in for the reference's windowed client
// (render_client/img_main.cpp): loads a model, builds the BVH, creates the device scene, runs
// the two passes of one frame (primary rays, then one reflection bounce, img_main.cpp:447-498),
// composites the two RGBA32F layers on the device the way gl_update blends its two textures
// (rt_composite_layers; img_main.cpp:434,512-526: layer 0 opaque, layer 1 with SRC_ALPHA / ONE_MINUS_DST_ALPHA),
// downloads the 8-bit frame and writes a BMP instead of presenting a window.
//
//   rt_headless [--model PATH] [--width W] [--height H] [--loader reference|tolerant]
//               [--bvh reference|fixed|sah] [--wide 2|4] [--bounces N] [--device D] [--out FILE.bmp] [--tiled T]
// --tiled T (with --bounces 0): the frame goes through the sharded-frame object instead (rt_tiled_create / _redeal / _render
// with one rank: the image cut into T x T squares, traced square by square, scattered into the framebuffer on the device) --
// the call sequence of a job of one process per GPU, and the same bytes in the BMP.
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
  int W = 560, H = 560, wide = 2, bounces = 1, device = 0, tiled = 0;
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
    else if (a == "--tiled") tiled = std::atoi(next());
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (W <= 0 || H <= 0 || bounces < 0 || (wide != 2 && wide != 4) || tiled < 0 || (tiled > 0 && bounces != 0)) { std::fprintf(stderr, "bad arguments\n"); return 2; }

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
  void *d_cam = nullptr, *d_a = nullptr, *d_b = nullptr, *d_bgra = nullptr;
  float *d_frame = nullptr, *d_layer = nullptr;
  HIPCHECK(hipMalloc(&d_cam, n * sizeof(rt_ray_isect)));
  HIPCHECK(hipMalloc(&d_a, n * sizeof(rt_ray_isect)));
  HIPCHECK(hipMalloc(&d_b, n * sizeof(rt_ray_isect)));
  HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_frame), n * 4 * sizeof(float)));
  HIPCHECK(hipMalloc(reinterpret_cast<void **>(&d_layer), n * 4 * sizeof(float)));
  HIPCHECK(hipMalloc(&d_bgra, n * 4));
  HIPCHECK(hipMemcpy(d_cam, rays.data(), n * sizeof(rt_ray_isect), hipMemcpyHostToDevice));

  // The frame stays on the device: pass 0 writes the opaque layer, every reflection pass writes its layer and
  // rt_composite_layers blends it over the frame (in place); the last call also writes the 8-bit pixels, the only thing
  // that is downloaded.
  const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  auto t0 = std::chrono::steady_clock::now();
  void *src = d_cam, *dst = d_a;
  if (tiled > 0) {
    rt_tiled *frame = nullptr;
    size_t moved = 0;
    CHECK(rt_tiled_create(scene, /*comm*/nullptr, /*rank*/0, /*nranks*/1, W, H, tiled, rays.data(), device, 0.0, &frame));
    CHECK(rt_tiled_redeal(frame, identity, &moved));      // with one rank every square stays; the call is what a moving camera repeats
    CHECK(rt_tiled_render(frame, identity, d_bgra, nullptr));
    HIPCHECK(hipDeviceSynchronize());
    rt_tiled_destroy(frame);
  }
  for (int pass = 0; pass <= bounces && tiled == 0; ++pass) {
    CHECK(rt_trace_pass(scene, src, dst, pass == 0 ? d_frame : d_layer, n, identity, pass == 0 ? 0u : 1u, nullptr));
    const bool last = pass == bounces;
    if (pass > 0) CHECK(rt_composite_layers(device, d_frame, d_layer, n, d_frame, last ? d_bgra : nullptr, nullptr));
    else if (last) CHECK(rt_composite_layers(device, d_frame, nullptr, n, nullptr, d_bgra, nullptr));
    src = dst;
    dst = (dst == d_a) ? d_b : d_a;
  }
  std::vector<uint8_t> bgra(n * 4);
  HIPCHECK(hipMemcpy(bgra.data(), d_bgra, n * 4, hipMemcpyDeviceToHost));   // waits for the passes (null stream)
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%d pass(es) over %zu rays in %.3f ms (including the download of the 8-bit frame)\n", bounces + 1, n, ms);

  CHECK(rt_write_bmp32(out.c_str(), bgra.data(), W, H));
  std::printf("wrote %s\n", out.c_str());

  (void)hipFree(d_cam); (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_frame); (void)hipFree(d_layer); (void)hipFree(d_bgra);
  rt_scene_destroy(scene);
  rt_bvh_destroy(bvh);
  rt_mesh_destroy(mesh);
  return 0;
}
