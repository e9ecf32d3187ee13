// multi_gpu.hip -- one frame rendered by several GPUs, one process per GPU: which image squares a rank traces
// (rt_shard_plan, rt_shard_ray_indices), the one exchange step of the path (rt_gather_scatter: every rank's records to
// the root rank over RCCL, point to point, then the device scatter into the row-major framebuffer) and the small kernels
// around it (8-byte hits for the wire, the two-layer composite of the reference's display pass).
//
// The reference renders one image on one device; its only hint of tiling is a commented-out split of the NDRange into
// four offsets (render_client/img_main.cpp:464-478).  Nothing here is a translation of reference code.
//
// RCCL is loaded on first use (dlopen), so a one-GPU user of the library needs no communication library at all.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "device_scene.h"

int RtFail(int code, const std::string &msg);   // capi.hip: sets rt_last_error()

namespace {

// ---- RCCL, bound at run time (names and signatures of rccl.h; the handful the exchange needs) ------------------------
constexpr int kNcclUniqueIdBytes = 128;
struct NcclUniqueId { char internal[kNcclUniqueIdBytes]; };
typedef void *NcclComm;
constexpr int kNcclUint8 = 1;   // ncclUint8
constexpr int kNcclFloat64 = 8; // ncclFloat64
constexpr int kNcclSum = 0;     // ncclSum

struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
};

Rccl *LoadRccl()
{
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      r.why = std::string("cannot load librccl.so: ") + dlerror();
      return;
    }
    auto sym = [&](const char *n) {
      void *p = dlsym(r.lib, n);
      if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &r;
}

int FailNccl(Rccl *r, int e, const char *what)
{
  return RtFail(RT_ERR_DEVICE, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error") + " (" + std::to_string(e) + ")");
}

struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int device)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope()
  {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// 16-byte {id, t, u, v} hits -> 8-byte {id, t}: what has to cross the GPUs for a framebuffer of hits (the barycentrics can be
// recomputed from the ray and t wherever the frame is shaded).
__global__ void __launch_bounds__(256) PackHits8Kernel(const uint4 *hits16, uint32_t n, uint2 *hits8)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 h = hits16[i];
  hits8[i] = make_uint2(h.x, h.y);
}

// The reference's display pass: layer 0 (the primary pass's pixels) drawn opaque, layer 1 (the reflection pass's) blended
// over it with glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_DST_ALPHA) (render_client/img_main.cpp:434,512-526):
// out = layer1 * layer1.a + layer0 * (1 - layer0.a), per channel, then 8-bit BGRA as rt_pack_bgra8 makes it.
__global__ void __launch_bounds__(256) CompositeKernel(const float4 *layer0, const float4 *layer1, uint32_t n, float4 *rgba_out, uint32_t *bgra_out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 a = layer0[i];
  float4 c = a;
  if (layer1) {
    const float4 b = layer1[i];
    const float sa = b.w, da = 1.0f - a.w;
    c = make_float4(b.x * sa + a.x * da, b.y * sa + a.y * da, b.z * sa + a.z * da, b.w * sa + a.w * da);
  }
  if (rgba_out) rgba_out[i] = c;
  if (bgra_out) {
    const uint32_t r = static_cast<uint32_t>(fminf(255.f, fmaxf(0.f, c.x * 255.f + 0.5f)));
    const uint32_t g = static_cast<uint32_t>(fminf(255.f, fmaxf(0.f, c.y * 255.f + 0.5f)));
    const uint32_t bl = static_cast<uint32_t>(fminf(255.f, fmaxf(0.f, c.z * 255.f + 0.5f)));
    bgra_out[i] = bl | (g << 8) | (r << 16) | 0xFF000000u;
  }
}


}  // namespace

// One rank's part of a sharded frame, kept between frames: the deal, this rank's rays (its squares one after the other) and
// the buffers of the exchange.
struct rt_tiled {
  rt_scene *scene = nullptr;
  void *comm = nullptr;
  int rank = 0, nranks = 1, device = 0, width = 0, height = 0, tile = 0, hint = 0;
  double head = 0.0;                                   // the gathering rank's head start in a deal by cost
  std::vector<rt_ray_isect> frame;                     // the whole frame's ray table (host copy; a re-deal picks from it)
  std::vector<uint32_t> tile_order, my_first;          // the deal; where each of this rank's squares starts among its records
  std::vector<uint64_t> first_tile, first_record;
  std::vector<rt_ray_isect> rays;                      // host copy of this rank's rays
  rt_ray_isect *d_rays = nullptr;
  uint32_t *d_pixels = nullptr, *d_ids = nullptr, *d_first = nullptr;   // records (root: the gather buffer); root: all squares' tables
  size_t n_local = 0;
};

namespace {

void TiledFreeDevice(rt_tiled *t)
{
  if (t->d_rays) (void)hipFree(t->d_rays);
  if (t->d_pixels) (void)hipFree(t->d_pixels);
  if (t->d_ids) (void)hipFree(t->d_ids);
  if (t->d_first) (void)hipFree(t->d_first);
  t->d_rays = nullptr;
  t->d_pixels = t->d_ids = t->d_first = nullptr;
}

template <class T>
int TiledUpload(T **dst, const T *src, size_t count, size_t alloc_count)
{
  hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), std::max<size_t>(1, alloc_count) * sizeof(T));
  if (e == hipSuccess && src && count) e = hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) return RtFail(e == hipErrorOutOfMemory ? RT_ERR_NOMEM : RT_ERR_DEVICE, std::string("sharded frame buffers: ") + hipGetErrorString(e));
  return RT_OK;
}

// Device buffers and tables for the deal the object holds (everything in flight must have finished).
int TiledBuild(rt_tiled *t)
{
  DeviceScope scope(t->device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  TiledFreeDevice(t);
  const uint32_t *my_ids = t->tile_order.data() + t->first_tile[t->rank];
  const size_t nmy = t->first_tile[t->rank + 1] - t->first_tile[t->rank], ntiles = t->tile_order.size();
  t->n_local = t->first_record[t->rank + 1] - t->first_record[t->rank];
  std::vector<uint32_t> idx(t->n_local);
  t->my_first.assign(nmy, 0u);
  int rc = rt_shard_ray_indices(t->width, t->height, t->tile, my_ids, nmy, 0u, idx.data(), t->my_first.data());
  if (rc != RT_OK) return rc;
  t->rays.resize(t->n_local);
  for (size_t k = 0; k < t->n_local; ++k) t->rays[k] = t->frame[idx[k]];
  rc = TiledUpload(&t->d_rays, t->rays.data(), t->n_local, t->n_local);
  if (rc != RT_OK) return rc;
  if (t->rank == 0) {
    std::vector<uint32_t> all_first(ntiles);
    rc = rt_shard_ray_indices(t->width, t->height, t->tile, t->tile_order.data(), ntiles, 0u, nullptr, all_first.data());
    if (rc == RT_OK) rc = TiledUpload(&t->d_ids, t->tile_order.data(), ntiles, ntiles);
    if (rc == RT_OK) rc = TiledUpload(&t->d_first, all_first.data(), ntiles, ntiles);
    if (rc == RT_OK) rc = TiledUpload<uint32_t>(&t->d_pixels, nullptr, 0, t->first_record[t->nranks]);
  } else {
    rc = TiledUpload<uint32_t>(&t->d_pixels, nullptr, 0, t->n_local);
  }
  if (rc != RT_OK) return rc;
  // a rank's rays are its squares stacked: an image one square wide, so that a wave still takes an 8x8 block of pixels
  t->hint = t->width % t->tile == 0 ? t->tile : 0;
  return rt_scene_set_param(t->scene, "image_width", t->hint);
}

uint32_t *TiledMine(const rt_tiled *t) { return t->rank == 0 ? t->d_pixels + t->first_record[0] : t->d_pixels; }

}  // namespace

extern "C" {

int rt_shard_plan(int width, int height, int tile, int nranks, const uint32_t *unit_costs, const double *square_costs,
                  const double *rank_head_start, uint32_t *tile_order, uint64_t *rank_first_tile, uint64_t *rank_first_record)
{
  if (width <= 0 || height <= 0 || tile <= 0 || nranks <= 0 || !tile_order || !rank_first_tile || !rank_first_record)
    return RtFail(RT_ERR_INVALID, "bad shard plan arguments");
  if (unit_costs && square_costs) return RtFail(RT_ERR_INVALID, "give the costs per work unit or per square, not both");
  if (unit_costs && tile % 8 != 0) return RtFail(RT_ERR_INVALID, "with a cost table the squares must be made of whole 8x8 work units");
  const uint32_t tx = (width + tile - 1) / tile, ty = (height + tile - 1) / tile, ntiles = tx * ty;
  auto pixels = [&](uint32_t k) {
    const uint32_t x0 = (k % tx) * tile, y0 = (k / tx) * tile;
    return static_cast<uint64_t>(std::min<uint32_t>(tile, width - x0)) * std::min<uint32_t>(tile, height - y0);
  };
  std::vector<std::vector<uint32_t>> lists(nranks);
  if (!unit_costs && !square_costs) {
    for (uint32_t k = 0; k < ntiles; ++k) lists[k % nranks].push_back(k);   // round-robin
  } else {
    // squares dealt costliest first, each to the rank with the least cost so far (ties: the lower rank); every square also
    // carries a floor cost: its pixels are generated and stored whatever they hit
    std::vector<double> cost(ntiles, 0.0);
    if (unit_costs) {   // the kernel's per-8x8-tile costs summed over each square
      const uint32_t ux = (width + 7) / 8, uy = (height + 7) / 8, per = tile / 8;
      for (uint32_t j = 0; j < uy; ++j)
        for (uint32_t i = 0; i < ux; ++i) cost[(j / per) * tx + i / per] += unit_costs[j * ux + i];
    } else {
      for (uint32_t k = 0; k < ntiles; ++k) {
        if (!(square_costs[k] >= 0.0)) return RtFail(RT_ERR_INVALID, "square costs must be non-negative numbers");
        cost[k] = square_costs[k];
      }
    }
    double total = 0.0;
    for (uint32_t k = 0; k < ntiles; ++k) total += cost[k] + 1.0;
    std::vector<uint32_t> order(ntiles);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<double> load(nranks, 0.0);
    if (rank_head_start)
      for (int r = 0; r < nranks; ++r) {
        if (!(rank_head_start[r] >= 0.0)) return RtFail(RT_ERR_INVALID, "head starts must be non-negative numbers");
        load[r] = rank_head_start[r] * total;
      }
    for (uint32_t k : order) {
      const int r = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());
      lists[r].push_back(k);
      load[r] += cost[k] + 1.0;
    }
  }
  uint64_t t = 0, rec = 0;
  for (int r = 0; r < nranks; ++r) {
    rank_first_tile[r] = t;
    rank_first_record[r] = rec;
    for (uint32_t k : lists[r]) {
      tile_order[t++] = k;
      rec += pixels(k);
    }
  }
  rank_first_tile[nranks] = t;
  rank_first_record[nranks] = rec;
  return RT_OK;
}

int rt_shard_square_costs(const uint32_t *unit_costs, size_t nunits, int image_width, size_t nrecords, const uint32_t *tile_ids,
                          const uint32_t *tile_first, size_t ntiles, double *square_costs, size_t nsquares)
{
  if (!unit_costs || !tile_ids || !tile_first || !square_costs || image_width < 0) return RtFail(RT_ERR_INVALID, "bad square cost arguments");
  // the work units of a launch over nrecords rays, as the library cuts them: 8x8 blocks of an image_width-wide image, or
  // runs of 64 records
  const bool image = image_width > 0 && nrecords % static_cast<size_t>(image_width) == 0;
  const size_t ux = image ? (static_cast<size_t>(image_width) + 7) / 8 : 0;
  const size_t expect = image ? ux * ((nrecords / image_width + 7) / 8) : (nrecords + 63) / 64;
  if (nunits != expect) return RtFail(RT_ERR_INVALID, "the cost table has " + std::to_string(nunits) + " work units, the ray set " + std::to_string(expect));
  for (size_t i = 0; i < ntiles; ++i) {
    if (tile_ids[i] >= nsquares) return RtFail(RT_ERR_INVALID, "square number out of range");
    if (i && tile_first[i] < tile_first[i - 1]) return RtFail(RT_ERR_INVALID, "squares must be listed in record order");
  }
  for (size_t u = 0; u < nunits; ++u) {
    // a unit counts for the square that holds its first record
    const size_t rec = image ? (u / ux) * 8 * static_cast<size_t>(image_width) + (u % ux) * 8 : u * 64;
    const uint32_t *hi = std::upper_bound(tile_first, tile_first + ntiles, static_cast<uint32_t>(std::min<size_t>(rec, 0xFFFFFFFFu)));
    if (hi == tile_first) continue;   // before the first listed square
    square_costs[tile_ids[hi - tile_first - 1]] += unit_costs[u];
  }
  return RT_OK;
}

int rt_shard_ray_indices(int width, int height, int tile, const uint32_t *tile_ids, size_t ntiles, uint32_t first_record,
                         uint32_t *ray_index, uint32_t *tile_first)
{
  if (width <= 0 || height <= 0 || tile <= 0 || (!tile_ids && ntiles)) return RtFail(RT_ERR_INVALID, "bad shard arguments");
  const uint32_t tx = (width + tile - 1) / tile, ty = (height + tile - 1) / tile;
  uint64_t pos = 0;
  for (size_t i = 0; i < ntiles; ++i) {
    const uint32_t k = tile_ids[i];
    if (k >= tx * ty) return RtFail(RT_ERR_INVALID, "square number out of range");
    const uint32_t x0 = (k % tx) * tile, y0 = (k / tx) * tile;
    const uint32_t w = std::min<uint32_t>(tile, width - x0), h = std::min<uint32_t>(tile, height - y0);
    if (tile_first) tile_first[i] = first_record + static_cast<uint32_t>(pos);
    if (ray_index)
      for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) ray_index[pos + static_cast<uint64_t>(y) * w + x] = (y0 + y) * static_cast<uint32_t>(width) + x0 + x;
    pos += static_cast<uint64_t>(w) * h;
  }
  return RT_OK;
}

int rt_pack_hits8(int device, const void *hits16_dev, size_t n, void *hits8_dev, void *stream)
{
  if (!hits16_dev || !hits8_dev) return RtFail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  if (n > 0xFFFFFFFFull) return RtFail(RT_ERR_INVALID, "more records than one launch addresses");
  DeviceScope scope(device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  (void)hipGetLastError();
  hipLaunchKernelGGL(PackHits8Kernel, dim3((static_cast<uint32_t>(n) + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4 *>(hits16_dev), static_cast<uint32_t>(n), static_cast<uint2 *>(hits8_dev));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("pack launch failed: ") + hipGetErrorString(e));
  return RT_OK;
}

int rt_composite_layers(int device, const float *layer0_rgba_dev, const float *layer1_rgba_dev, size_t n, float *rgba_out_dev,
                        void *bgra_out_dev, void *stream)
{
  if (!layer0_rgba_dev || (!rgba_out_dev && !bgra_out_dev)) return RtFail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  if (n > 0xFFFFFFFFull) return RtFail(RT_ERR_INVALID, "more pixels than one launch addresses");
  DeviceScope scope(device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  (void)hipGetLastError();
  hipLaunchKernelGGL(CompositeKernel, dim3((static_cast<uint32_t>(n) + 255u) / 256u), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(layer0_rgba_dev), reinterpret_cast<const float4 *>(layer1_rgba_dev),
                     static_cast<uint32_t>(n), reinterpret_cast<float4 *>(rgba_out_dev), static_cast<uint32_t *>(bgra_out_dev));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("composite launch failed: ") + hipGetErrorString(e));
  return RT_OK;
}

// ---- communicator: a thin face of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy -------------------------------
int rt_comm_unique_id(uint8_t id_out[RT_COMM_ID_BYTES])
{
  if (!id_out) return RtFail(RT_ERR_INVALID, "null argument");
  static_assert(RT_COMM_ID_BYTES == kNcclUniqueIdBytes, "id size");
  Rccl *r = LoadRccl();
  if (!r->why.empty()) return RtFail(RT_ERR_DEVICE, r->why);
  NcclUniqueId id;
  int e = r->GetUniqueId(&id);
  if (e != 0) return FailNccl(r, e, "ncclGetUniqueId");
  std::memcpy(id_out, id.internal, kNcclUniqueIdBytes);
  return RT_OK;
}

int rt_comm_create(const uint8_t id[RT_COMM_ID_BYTES], int nranks, int rank, int device, void **comm_out)
{
  if (!id || !comm_out || nranks <= 0 || rank < 0 || rank >= nranks) return RtFail(RT_ERR_INVALID, "bad communicator arguments");
  Rccl *r = LoadRccl();
  if (!r->why.empty()) return RtFail(RT_ERR_DEVICE, r->why);
  DeviceScope scope(device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  NcclUniqueId uid;
  std::memcpy(uid.internal, id, kNcclUniqueIdBytes);
  NcclComm comm = nullptr;
  int e = r->CommInitRank(&comm, nranks, uid, rank);
  if (e != 0) return FailNccl(r, e, "ncclCommInitRank");
  *comm_out = comm;
  return RT_OK;
}

void rt_comm_destroy(void *comm)
{
  Rccl *r = LoadRccl();
  if (comm && r->CommDestroy) (void)r->CommDestroy(comm);
}

int rt_comm_sum_f64(void *comm, int nranks, int device, double *values, size_t n)
{
  if (nranks <= 0 || (n && !values)) return RtFail(RT_ERR_INVALID, "bad arguments");
  if (n == 0 || (nranks == 1 && !comm)) return RT_OK;   // one rank WITH a communicator still goes through RCCL (rt_comm_check relies on it)
  if (!comm) return RtFail(RT_ERR_INVALID, "a communicator is needed for more than one rank");
  Rccl *r = LoadRccl();
  if (!r->why.empty()) return RtFail(RT_ERR_DEVICE, r->why);
  DeviceScope scope(device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  double *d = nullptr;
  hipError_t he = hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(double));
  if (he != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(he));
  hipStream_t q = nullptr;
  int e = 0;
  he = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipMemcpyAsync(d, values, n * sizeof(double), hipMemcpyHostToDevice, q);
  if (he == hipSuccess) e = r->AllReduce(d, d, n, kNcclFloat64, kNcclSum, comm, q);
  if (he == hipSuccess && e == 0) he = hipMemcpyAsync(values, d, n * sizeof(double), hipMemcpyDeviceToHost, q);
  if (he == hipSuccess && e == 0) he = hipStreamSynchronize(q);
  if (q) (void)hipStreamDestroy(q);
  (void)hipFree(d);
  if (e != 0) return FailNccl(r, e, "ncclAllReduce");
  if (he != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("sum over ranks: ") + hipGetErrorString(he));
  return RT_OK;
}

int rt_comm_check(void *comm, int nranks, int rank, int device)
{
  if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return RtFail(RT_ERR_INVALID, "bad communicator arguments");
  Rccl *r = LoadRccl();
  if (!r->why.empty()) return RtFail(RT_ERR_DEVICE, r->why);
  DeviceScope scope(device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  // Every rank sends kWords words that name it to the next rank and receives the previous rank's (one grouped call, the
  // same calls rt_gather_scatter makes; one rank exchanges with itself), then the ranks sum {1, rank} as doubles.
  constexpr size_t kWords = 1024;
  const int next = (rank + 1) % nranks, prev = (rank + nranks - 1) % nranks;
  std::vector<uint32_t> out(kWords), in(kWords, 0u);
  for (size_t i = 0; i < kWords; ++i) out[i] = 0x9E3779B9u * static_cast<uint32_t>(rank + 1) + static_cast<uint32_t>(i);
  uint32_t *d = nullptr;
  hipStream_t q = nullptr;
  hipError_t he = hipMalloc(reinterpret_cast<void **>(&d), 2 * kWords * sizeof(uint32_t));
  if (he != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(he));
  int e = 0, e2 = 0;
  he = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipMemcpyAsync(d, out.data(), kWords * sizeof(uint32_t), hipMemcpyHostToDevice, q);
  if (he == hipSuccess) he = hipMemsetAsync(d + kWords, 0, kWords * sizeof(uint32_t), q);
  if (he == hipSuccess) {
    e = r->GroupStart();
    if (e == 0) e = r->Send(d, kWords * sizeof(uint32_t), kNcclUint8, next, comm, q);
    if (e == 0) e = r->Recv(d + kWords, kWords * sizeof(uint32_t), kNcclUint8, prev, comm, q);
    e2 = r->GroupEnd();
  }
  if (he == hipSuccess && e == 0 && e2 == 0) he = hipMemcpyAsync(in.data(), d + kWords, kWords * sizeof(uint32_t), hipMemcpyDeviceToHost, q);
  if (he == hipSuccess && e == 0 && e2 == 0) he = hipStreamSynchronize(q);
  if (q) (void)hipStreamDestroy(q);
  (void)hipFree(d);
  if (e != 0) return FailNccl(r, e, "ncclSend/ncclRecv");
  if (e2 != 0) return FailNccl(r, e2, "ncclGroupEnd");
  if (he != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("communicator check: ") + hipGetErrorString(he));
  for (size_t i = 0; i < kWords; ++i)
    if (in[i] != 0x9E3779B9u * static_cast<uint32_t>(prev + 1) + static_cast<uint32_t>(i))
      return RtFail(RT_ERR_DEVICE, "communicator check: the words received from rank " + std::to_string(prev) + " are not the ones it sent");
  double sums[2] = {1.0, static_cast<double>(rank)};
  {
    int rc = rt_comm_sum_f64(comm, nranks, device, sums, 2);
    if (rc != RT_OK) return rc;
  }
  if (sums[0] != static_cast<double>(nranks) || sums[1] != 0.5 * nranks * (nranks - 1.0))
    return RtFail(RT_ERR_DEVICE, "communicator check: the sum over the ranks is wrong");
  return RT_OK;
}

int rt_gather_scatter(void *comm, int rank, int nranks, int root, int device, const void *mine_dev, void *gather_dev,
                      const uint64_t *rank_first_record, size_t record_bytes, const uint32_t *tile_ids_dev, const uint32_t *tile_first_dev,
                      size_t ntiles, int tile, int width, int height, void *framebuffer_dev, void *stream)
{
  if (nranks <= 0 || rank < 0 || rank >= nranks || root < 0 || root >= nranks || !rank_first_record)
    return RtFail(RT_ERR_INVALID, "bad rank arguments");
  if (record_bytes == 0 || record_bytes % 4u != 0) return RtFail(RT_ERR_INVALID, "record size must be a multiple of 4 bytes");
  if (nranks > 1 && !comm) return RtFail(RT_ERR_INVALID, "a communicator is needed for more than one rank");
  const uint64_t mine = rank_first_record[rank + 1] - rank_first_record[rank];
  if (mine && !mine_dev) return RtFail(RT_ERR_INVALID, "null record buffer");
  if (rank == root && (!gather_dev || (ntiles && (!tile_ids_dev || !tile_first_dev || !framebuffer_dev))))
    return RtFail(RT_ERR_INVALID, "the root rank needs the gather buffer, the square list and the framebuffer");
  DeviceScope scope(device);
  if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
  hipStream_t q = static_cast<hipStream_t>(stream);
  if (nranks > 1) {
    // One grouped call: the root posts a receive per peer and every peer one send, so all of the root's links carry their
    // share at once (point to point, no ring: xGMI is a full mesh of links, and a ring would pass every byte over several).
    Rccl *r = LoadRccl();
    if (!r->why.empty()) return RtFail(RT_ERR_DEVICE, r->why);
    int e = r->GroupStart();
    if (e != 0) return FailNccl(r, e, "ncclGroupStart");
    if (rank == root) {
      for (int peer = 0; peer < nranks && e == 0; ++peer) {
        const uint64_t cnt = rank_first_record[peer + 1] - rank_first_record[peer];
        if (peer == root || cnt == 0) continue;
        e = r->Recv(static_cast<char *>(gather_dev) + rank_first_record[peer] * record_bytes, cnt * record_bytes, kNcclUint8, peer, comm, q);
      }
    } else if (mine) {
      e = r->Send(mine_dev, mine * record_bytes, kNcclUint8, root, comm, q);
    }
    const int e2 = r->GroupEnd();
    if (e != 0) return FailNccl(r, e, "ncclSend/ncclRecv");
    if (e2 != 0) return FailNccl(r, e2, "ncclGroupEnd");
  }
  if (rank != root) return RT_OK;
  // the root's own records: traced straight into its slice of the gather buffer, or copied there now
  char *own = static_cast<char *>(gather_dev) + rank_first_record[root] * record_bytes;
  if (mine && mine_dev != own) {
    hipError_t he = hipMemcpyAsync(own, mine_dev, mine * record_bytes, hipMemcpyDeviceToDevice, q);
    if (he != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("hipMemcpyAsync: ") + hipGetErrorString(he));
  }
  if (ntiles == 0) return RT_OK;
  return rt_scatter_tiles(device, gather_dev, rank_first_record[nranks], tile_ids_dev, tile_first_dev, ntiles, tile, width, height, record_bytes,
                          framebuffer_dev, stream);
}

int rt_tiled_create(rt_scene *scene, void *comm, int rank, int nranks, int width, int height, int tile, const rt_ray_isect *frame_rays_host,
                    int device, double root_head_start, rt_tiled **out)
{
  if (!scene || !frame_rays_host || !out || width <= 0 || height <= 0 || tile <= 0 || nranks <= 0 || rank < 0 || rank >= nranks)
    return RtFail(RT_ERR_INVALID, "bad sharded frame arguments");
  if (nranks > 1 && !comm) return RtFail(RT_ERR_INVALID, "a communicator is needed for more than one rank");
  if (!(root_head_start >= 0.0)) return RtFail(RT_ERR_INVALID, "the head start must be a non-negative number");
  if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) > 0x7FFFFFFFull) return RtFail(RT_ERR_INVALID, "image too large");
  rt_tiled *t = new (std::nothrow) rt_tiled;
  if (!t) return RtFail(RT_ERR_NOMEM, "out of memory");
  t->scene = scene; t->comm = comm; t->rank = rank; t->nranks = nranks; t->device = device;
  t->width = width; t->height = height; t->tile = tile; t->head = root_head_start;
  const size_t tx = (width + tile - 1) / tile, ty = (height + tile - 1) / tile;
  int rc = RT_OK;
  try {
    t->frame.assign(frame_rays_host, frame_rays_host + static_cast<size_t>(width) * height);
    t->tile_order.resize(tx * ty);
    t->first_tile.resize(nranks + 1);
    t->first_record.resize(nranks + 1);
  } catch (const std::bad_alloc &) {
    rc = RtFail(RT_ERR_NOMEM, "out of memory");
  }
  // the first deal is round-robin; rt_tiled_redeal deals by measured cost
  if (rc == RT_OK) rc = rt_shard_plan(width, height, tile, nranks, nullptr, nullptr, nullptr, t->tile_order.data(), t->first_tile.data(), t->first_record.data());
  if (rc == RT_OK) rc = TiledBuild(t);
  if (rc != RT_OK) {
    rt_tiled_destroy(t);
    return rc;
  }
  *out = t;
  return RT_OK;
}

void rt_tiled_destroy(rt_tiled *t)
{
  if (!t) return;
  {
    DeviceScope scope(t->device);
    if (scope.ok) TiledFreeDevice(t);
  }
  delete t;
}

int rt_tiled_info(const rt_tiled *t, uint64_t info[4])
{
  if (!t || !info) return RtFail(RT_ERR_INVALID, "null argument");
  info[0] = t->n_local;
  info[1] = t->first_tile[t->rank + 1] - t->first_tile[t->rank];
  info[2] = t->tile_order.size();
  info[3] = static_cast<uint64_t>(t->hint);
  return RT_OK;
}

int rt_tiled_render(rt_tiled *t, const float cam[16], void *framebuffer_dev, void *stream)
{
  if (!t || !cam) return RtFail(RT_ERR_INVALID, "null argument");
  if (t->rank == 0 && !framebuffer_dev) return RtFail(RT_ERR_INVALID, "the gathering rank needs a framebuffer");
  uint32_t *mine = TiledMine(t);
  int rc = rt_trace_pass_bgra8(t->scene, t->d_rays, nullptr, mine, t->n_local, cam, 0u, stream);
  if (rc != RT_OK) return rc;
  const bool root = t->rank == 0;
  return rt_gather_scatter(t->comm, t->rank, t->nranks, 0, t->device, mine, root ? t->d_pixels : nullptr, t->first_record.data(), 4u,
                           root ? t->d_ids : nullptr, root ? t->d_first : nullptr, root ? t->tile_order.size() : 0u, t->tile, t->width, t->height,
                           framebuffer_dev, stream);
}

int rt_tiled_redeal(rt_tiled *t, const float cam[16], size_t *squares_moved)
{
  if (!t || !cam) return RtFail(RT_ERR_INVALID, "null argument");
  const size_t ntiles = t->tile_order.size(), nmy = t->first_tile[t->rank + 1] - t->first_tile[t->rank];
  std::vector<double> sq(ntiles, 0.0);
  {
    DeviceScope scope(t->device);
    if (!scope.ok) return RtFail(RT_ERR_DEVICE, "hipSetDevice failed");
    hipError_t he = hipDeviceSynchronize();   // frames in flight use the buffers this call replaces
    if (he != hipSuccess) return RtFail(RT_ERR_DEVICE, std::string("hipDeviceSynchronize: ") + hipGetErrorString(he));
    if (t->n_local) {
      // this rank's rays as the trace step sees them (camera rows dotted with (origin, 1) and (direction, 0)); the costs only
      // rank the squares, so the host's rounding of these sums need not be the kernel's
      std::vector<rt_ray_isect> seen = t->rays;
      for (rt_ray_isect &r : seen) {
        const float o[4] = {r.origin.x, r.origin.y, r.origin.z, 1.f}, d[4] = {r.direction.x, r.direction.y, r.direction.z, 0.f};
        float oo[3], dd[3];
        for (int k = 0; k < 3; ++k) {
          oo[k] = cam[4 * k] * o[0] + cam[4 * k + 1] * o[1] + cam[4 * k + 2] * o[2] + cam[4 * k + 3] * o[3];
          dd[k] = cam[4 * k] * d[0] + cam[4 * k + 1] * d[1] + cam[4 * k + 2] * d[2] + cam[4 * k + 3] * d[3];
        }
        r.origin.x = oo[0]; r.origin.y = oo[1]; r.origin.z = oo[2];
        r.direction.x = dd[0]; r.direction.y = dd[1]; r.direction.z = dd[2];
      }
      rt_ray_isect *d_seen = nullptr;
      int rc = TiledUpload(&d_seen, seen.data(), seen.size(), seen.size());
      size_t nunits = 0;
      if (rc == RT_OK) rc = rt_scene_unit_costs(t->scene, d_seen, t->n_local, nullptr, 0, &nunits);
      std::vector<uint32_t> costs(nunits);
      if (rc == RT_OK) rc = rt_scene_unit_costs(t->scene, d_seen, t->n_local, costs.data(), costs.size(), &nunits);
      if (d_seen) (void)hipFree(d_seen);
      if (rc == RT_OK)
        rc = rt_shard_square_costs(costs.data(), nunits, t->hint, t->n_local, t->tile_order.data() + t->first_tile[t->rank], t->my_first.data(), nmy,
                                   sq.data(), ntiles);
      if (rc != RT_OK) return rc;
    }
  }
  int rc = rt_comm_sum_f64(t->comm, t->nranks, t->device, sq.data(), ntiles);
  if (rc != RT_OK) return rc;
  std::vector<int> owner_before(ntiles, 0);
  for (int r = 0; r < t->nranks; ++r)
    for (uint64_t i = t->first_tile[r]; i < t->first_tile[r + 1]; ++i) owner_before[t->tile_order[i]] = r;
  std::vector<double> head(t->nranks, 0.0);
  head[0] = t->nranks > 1 ? t->head : 0.0;
  rc = rt_shard_plan(t->width, t->height, t->tile, t->nranks, nullptr, sq.data(), head.data(), t->tile_order.data(), t->first_tile.data(),
                     t->first_record.data());
  if (rc != RT_OK) return rc;
  size_t moved = 0;
  for (int r = 0; r < t->nranks; ++r)
    for (uint64_t i = t->first_tile[r]; i < t->first_tile[r + 1]; ++i) moved += owner_before[t->tile_order[i]] != r;
  if (squares_moved) *squares_moved = moved;
  return TiledBuild(t);
}

}  // extern "C"
