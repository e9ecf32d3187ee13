// capi.hip -- implementation of the C ABI declared in include/rt_amd.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <algorithm>
#include <atomic>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_amd.h"
#include "device_scene.h"
#include "host_scene.h"
#include "flatten_device.h"

namespace {

thread_local std::string g_error;

int Fail(rt_status code, const std::string &msg)
{
  g_error = msg;
  return code;
}

int FailHip(hipError_t e, const char *what)
{
  return Fail(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

// shared with multi_gpu.hip
int RtFail(int code, const std::string &msg) { return Fail(static_cast<rt_status>(code), msg); }

namespace {

#define RT_HIP(call)                                   \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return FailHip(e_, #call);   \
  } while (0)

// Makes `device` current for the scope and restores the caller's device afterwards.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard()
  {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

template <typename T>
int Upload(T **dst, const void *src, size_t bytes)
{
  *dst = nullptr;
  if (bytes == 0) bytes = 16;  // keep every pointer valid
  void *p = nullptr;
  RT_HIP(hipMalloc(&p, bytes));
  *dst = static_cast<T *>(p);
  if (src) RT_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
  else RT_HIP(hipMemset(p, 0, bytes));
  return RT_OK;
}

constexpr uint32_t kFeedbackMaxRays = 1u << 30;  // single-launch calls only
constexpr int kStatShards = 64;          // instrumented build: counter shards, 16 uint64 (one line) each
constexpr uint32_t kCursorSlots = 64, kCursorStride = 32;   // persistent launches: cursor words, one line apart

// ---- fixed device memory of the allocation-free entry point (rt_intersect_static) ----------------------
// The cost-feedback schedule needs two tables per launch shape: the per-unit cost the trace kernel records
// and the dispatch order the sort kernel derives from it.  rt_intersect_async keeps them in hipMalloc'd
// buffers owned by the scene; callers that must not allocate get them from this array instead, one copy
// per GPU the library runs on, held until the process ends: 2 MiB + 32 KiB = 2^18 cost words + 2^18 order words + the sort's
// 2^13 bin counters, enough for 2^18 wave-sized work units = 16 Mi rays per call.
constexpr uint32_t kStaticMaxUnits = 1u << 18;
constexpr size_t kStaticMaxRays = static_cast<size_t>(kStaticMaxUnits) * 64u;
constexpr uint32_t kStaticBinWords = kStaticMaxUnits / 1024u * static_cast<uint32_t>(rt::kFbBuckets);
static __device__ __align__(256) uint32_t scratch[2u * kStaticMaxUnits + kStaticBinWords];

// In host code the name `scratch` is only a handle; the array's address on a given GPU comes from
// hipGetSymbolAddress with that GPU current, once per device, and is what every launch and memset uses.
constexpr int kMaxDevices = 64;
struct StaticState {
  uint32_t *ptr = nullptr;      // device address of `scratch` on this GPU
  uint32_t ntasks = 0, n = 0, img_w = 0;   // shape the tables currently describe
  uint64_t scene_id = 0;        // ... and the scene whose costs they hold (0 = none)
  bool order_valid = false;     // the order half holds a schedule for that shape and scene
  uint32_t since_sort = 0;      // calls since the last one that recorded costs and was followed by a sort
  hipStream_t last_stream = nullptr;   // stream of the previous call
  bool have_last = false;
  hipEvent_t ev_left = nullptr; // recorded on the stream being left when calls change stream (created with the address lookup)
};
StaticState g_static[kMaxDevices];
std::mutex g_static_mutex;   // guards the table state: the lookup, and every rt_intersect_static call from its first read of the
                             // state to its last enqueue
std::atomic<uint64_t> g_next_scene_id{1};

// Caller holds the device current and g_static_mutex.
int StaticScratch(int device, StaticState **out)
{
  if (device < 0 || device >= kMaxDevices) return Fail(RT_ERR_DEVICE, "device ordinal beyond the allocation-free entry point's table");
  StaticState &st = g_static[device];
  if (!st.ptr) {
    void *p = nullptr;
    RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    // the event that orders calls on different streams is made here, where allocating is allowed (rt_scene_create)
    if (!st.ev_left) RT_HIP(hipEventCreateWithFlags(&st.ev_left, hipEventDisableTiming));
    st.ptr = static_cast<uint32_t *>(p);
  }
  *out = &st;
  return RT_OK;
}

}  // namespace

struct rt_mesh {
  rt::Mesh mesh;
  std::string log_joined;
};

struct rt_bvh {
  rt::CentroidBVH bvh;
};

struct rt_scene {
  uint64_t id = g_next_scene_id.fetch_add(1);   // never reused: tells scenes apart where a pointer could come back
  int device = 0;
  int num_cu = 256;
  int width = 2;
  int quant = 1;                 // traverse the 16-bit quantized records (16 B per child) instead of float boxes
  int block_threads = 64;        // workgroup size: 64, 128 or 256 (one wave per workgroup measured equal or up to 4 % faster)
  int centre_first = 1;          // image mode: dispatch tiles nearest the image centre first
  mutable uint32_t *d_order = nullptr;   // cached dispatch order for (order_w, order_h)
  mutable uint32_t order_w = 0, order_h = 0;
  int feedback = 1;              // reuse the previous launch's per-unit cost to start expensive units first
  mutable uint32_t *d_fb_cost[2] = {nullptr, nullptr}, *d_fb_order[2] = {nullptr, nullptr}, *d_fb_bins = nullptr;
  mutable uint32_t fb_ntasks = 0, fb_n = 0, fb_img_w = 0;  // shape the feedback buffers describe
  mutable bool fb_sorted[2] = {false, false};               // d_fb_order[k] holds (or will hold, after ev_sort[k]) an order
  mutable bool fb_pending[2] = {false, false};              // ... whose sort has not yet been seen to finish
  mutable hipStream_t fb_sort_on[2] = {nullptr, nullptr};   // ... and the stream that sort was enqueued on
  mutable uint32_t fb_launch = 0;
  mutable uint32_t fb_since_sort = 0;                       // launches since the last one that recorded costs
  mutable int fb_cur = 0, fb_newest = 0;                    // set the launches record into; set of the sort enqueued last
  int model_width = 0;                                      // rt_scene_wave_model only: 8 = price an eight-wide collapse of the tree
  int side_stream = 0;                                      // 0: the order-building kernel runs on the launch's own stream, behind the trace kernel; 1: on a side stream
  int feedback_period = 8;                                  // every this-many launches one records its costs and a sort follows it
  mutable hipStream_t side = nullptr;                       // the order-building kernel runs here, beside the next trace
  mutable hipEvent_t ev_main[2] = {nullptr, nullptr}, ev_sort[2] = {nullptr, nullptr};
  // Launches of one scene may go to several streams.  A sort overwrites an order table that launches on OTHER streams than
  // the one it follows may still be reading, so whenever the feedback launches change stream, an event recorded on the
  // stream being left marks everything enqueued there so far; the next sort waits for all of those (none in one-stream use).
  mutable hipStream_t fb_last_stream = nullptr;
  mutable bool fb_have_last = false;
  mutable std::vector<hipEvent_t> fb_left;      // pool of events, the first fb_left_used of them recorded since the last sort
  mutable size_t fb_left_used = 0;
  int waves_per_cu = 0;          // optional resident-wave cap per CU, held through the LDS stack size (0 = none)
  int refill = 0;                // > 0: linear ray sets with compact hits go to the lane-refilling kernel; idle lanes that trigger a refill
  int split_max_units = 40960;   // launches of more work units than this are not split: they fill the chip many times over and only pay for the extra waves
  int trace_split = -1;          // the full trace step: -1 splits like the intersection launches (by size), 1 whatever the size, 0 never
  int split_coop = 1;            // the quarter waves of split units work four lanes per ray (one child box / one triangle per lane)
  int split_units = 128;         // with a cost-sorted order: this many units, the costliest, are traced by four waves each
  int prio_permille = 0;         // with a cost-sorted order: this share (1/1000) of the leading dispatch slots runs at raised priority
  int lds_stack = 10;            // traversal-stack entries per lane kept in LDS; deeper entries spill to private memory
  int persistent = 0;            // 1: a fixed grid of waves pulls work units from a global cursor (one atomic per unit)
  int persistent_waves = 24;     // ... this many waves per CU
  uint32_t *d_cursor = nullptr;  // kCursorSlots cursors, one 128-byte line each, used round-robin by consecutive launches
  mutable uint32_t next_cursor = 0;
  int feedback_share = 96;       // ... scaled by this many 128ths
  int feedback_radius = 1;       // a unit's measured cost also counts for its neighbours this many units away
  int leaf_pct = 50;             // node/leaf step scheduling threshold (percent of busy lanes parked at a leaf)
  int extra_stack = 0;           // diagnostic: extra LDS stack entries per lane (lowers occupancy)
  int image_width = 0;           // hint: ray arrays are row-major images of this width (0 = unstructured)
  rt::FlatScene flat;          // host copy (fetch counting, info)
  size_t nverts = 0, ntris = 0, nnormals = 0, nmats = 0;
  float4 *d_nodes2 = nullptr, *d_nodes4 = nullptr, *d_nodesq2 = nullptr, *d_nodesq4 = nullptr, *d_tris = nullptr;
  rt_vec3 *d_verts = nullptr;
  rt_face *d_faces = nullptr;
  rt_float4 *d_normals = nullptr;
  rt_material *d_mats = nullptr;
  unsigned long long *d_stats = nullptr;
  // staging for host-pointer calls
  mutable void *d_stage_in = nullptr, *d_stage_out = nullptr;
  mutable size_t stage_in_bytes = 0, stage_out_bytes = 0;
  // host-pointer calls move rays and hits in chunks through these pinned buffers (two per direction) on streams
  // of their own, so the upload of chunk k+1, the trace of chunk k and the download of chunk k-1 overlap
  mutable void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};
  mutable hipStream_t st_h2d = nullptr, st_trace = nullptr, st_d2h = nullptr;
  mutable hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_trace[2] = {nullptr, nullptr}, ev_d2h[2] = {nullptr, nullptr};
  mutable hipEvent_t ev_entry = nullptr;   // marks what the caller's null stream held when a mixed host/device call came in
};

namespace {

void FillArgs(const rt_scene *s, rt::KernelArgs &a)
{
  std::memset(&a, 0, sizeof a);
  if (s->quant) a.nodes = s->width == 4 ? s->d_nodesq4 : s->d_nodesq2;
  else a.nodes = s->width == 4 ? s->d_nodes4 : s->d_nodes2;
  rt::SetQuantGrid(a.grid, s->flat.grid_lo, s->flat.grid_step);
  a.tris = s->d_tris;
  a.verts = s->d_verts;
  a.faces = s->d_faces;
  a.normals = s->d_normals;
  a.mats = s->d_mats;
  a.nnormals = static_cast<uint32_t>(s->nnormals);
  a.nmats = static_cast<uint32_t>(s->nmats);
  // the LDS part of the stack; the output stage also exchanges the wave's 64 x 64-byte records through it
  a.stack_entries = std::max(rt::kMinLdsEntries, static_cast<uint32_t>(s->lds_stack + s->extra_stack));
  a.scale = s->flat.scale;
  a.leaf_pct = static_cast<uint32_t>(s->leaf_pct);
  a.stats = s->d_stats;
}

// Dispatch order for a W x H image walked in 8x8 tiles: 16x16-pixel groups of four tiles (one workgroup's
// worth) sorted by their distance from the image centre, nearest first.  For a framed object those are the
// tiles whose rays traverse deepest; starting them first keeps the few long-running waves from being the
// last ones standing while the rest of the chip idles.  Purely a schedule: every tile is traced exactly once.
int EnsureOrder(const rt_scene *s, uint32_t w, uint32_t h)
{
  if (s->d_order && s->order_w == w && s->order_h == h) return RT_OK;
  const uint32_t tx = (w + 7u) / 8u, ty = (h + 7u) / 8u, gx = (tx + 1u) / 2u, gy = (ty + 1u) / 2u;
  std::vector<std::pair<float, uint32_t>> groups;
  groups.reserve(static_cast<size_t>(gx) * gy);
  for (uint32_t j = 0; j < gy; ++j)
    for (uint32_t i = 0; i < gx; ++i) {
      const float dx = ((i + 0.5f) * 16.f - 0.5f * w) / (0.5f * w), dy = ((j + 0.5f) * 16.f - 0.5f * h) / (0.5f * h);
      groups.emplace_back(dx * dx + dy * dy, j * gx + i);
    }
  std::stable_sort(groups.begin(), groups.end());
  std::vector<uint32_t> order;
  order.reserve(static_cast<size_t>(tx) * ty);
  for (auto const &g : groups) {
    const uint32_t i = g.second % gx, j = g.second / gx;
    for (uint32_t k = 0; k < 4; ++k) {
      const uint32_t x = 2 * i + (k & 1u), y = 2 * j + (k >> 1);
      if (x < tx && y < ty) order.push_back(y * tx + x);
    }
  }
  if (s->d_order) (void)hipFree(s->d_order);
  s->d_order = nullptr;
  s->order_w = s->order_h = 0;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_order), order.size() * sizeof(uint32_t)));
  RT_HIP(hipMemcpy(s->d_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  s->order_w = w;
  s->order_h = h;
  return RT_OK;
}

// Drops every piece of launch-to-launch feedback state (after waiting for a sort that may still be running).
int ResetFeedback(const rt_scene *s)
{
  if (s->side) RT_HIP(hipStreamSynchronize(s->side));
  for (int k = 0; k < 2; ++k)   // sorts that ran on a caller's stream
    if (s->fb_pending[k] && s->ev_sort[k]) RT_HIP(hipEventSynchronize(s->ev_sort[k]));
  for (int k = 0; k < 2; ++k) {
    if (s->d_fb_cost[k]) (void)hipFree(s->d_fb_cost[k]);
    if (s->d_fb_order[k]) (void)hipFree(s->d_fb_order[k]);
    s->d_fb_cost[k] = s->d_fb_order[k] = nullptr;
    s->fb_sorted[k] = s->fb_pending[k] = false;
  }
  s->fb_since_sort = 0;
  s->fb_cur = s->fb_newest = 0;
  if (s->d_fb_bins) (void)hipFree(s->d_fb_bins);
  s->d_fb_bins = nullptr;
  s->fb_launch = 0;
  s->fb_left_used = 0;
  s->fb_have_last = false;
  s->fb_ntasks = s->fb_n = s->fb_img_w = 0;
  return RT_OK;
}

// The traversal stack must cover the tree's worst case (LDS part + private part), and the LDS part of one
// workgroup must fit a CU.  May raise a.stack_entries.
int CheckStack(const rt_scene *s, rt::KernelArgs &a, uint32_t block)
{
  const uint32_t need = static_cast<uint32_t>(s->width == 4 ? s->flat.stack4 : s->flat.stack2);
  // a tree deeper than the configured LDS part plus the private part can hold (a caller's degenerate tree, long chains of
  // fallback splits): grow the LDS part to fit, as far as a CU's LDS allows
  if (need > a.stack_entries + rt::kDeepEntries) a.stack_entries = need - rt::kDeepEntries;
  if (static_cast<size_t>(a.stack_entries) * 8u * block > 160u * 1024u)
    return Fail(RT_ERR_INVALID, "traversal stack of " + std::to_string(a.stack_entries) + " LDS entries x " + std::to_string(block) +
                                    " threads exceeds a CU's 160 KB of LDS: the tree needs " + std::to_string(need) + " stack entries per ray, " +
                                    std::to_string(rt::kDeepEntries) + " of which live in private memory (a shallower tree, a smaller "
                                    "\"block_threads\", or lower \"waves_per_cu\" / \"lds_stack\" / \"extra_stack\" settings fit)");
  return RT_OK;
}

// The quarter waves of split units may work four lanes per ray: four-wide quantized records, and the tree's worst-case stack
// within the LDS columns of a quad (nothing of that mode spills to private memory).
bool CoopFits(const rt_scene *s, const rt::KernelArgs &a)
{
  return s->split_coop && s->width == 4 && s->quant && static_cast<uint32_t>(s->flat.stack4) <= 4u * a.stack_entries;
}

// Work units of a launch over a.n rays: 8x8 pixel tiles when the rays are declared an image, else runs of 64.
void SetShape(const rt_scene *s, rt::KernelArgs &a, int mode)
{
  const uint32_t width_hint = mode == rt::kModePrimary ? a.img_w : static_cast<uint32_t>(s->image_width);
  a.img_w = a.img_h = 0;
  a.ntasks = (a.n + 63u) / 64u;
  if (width_hint > 0 && a.n % width_hint == 0) {
    a.img_w = width_hint;
    a.img_h = a.n / a.img_w;
    a.ntasks = ((a.img_w + 7u) / 8u) * ((a.img_h + 7u) / 8u);
  }
}

// `plain` launches use no dispatch-order table and leave the feedback state alone (the chunked host-pointer path:
// its launches change shape from chunk to chunk, which would rebuild the tables every time).
int Enqueue(const rt_scene *s, rt::KernelArgs &a, int mode, bool count, hipStream_t stream, bool plain = false)
{
  SetShape(s, a, mode);
  if (a.img_w && s->centre_first && !plain) {
    int rc = EnsureOrder(s, a.img_w, a.img_h);
    if (rc != RT_OK) return rc;
    a.order = s->d_order;
  }
  const uint32_t block = static_cast<uint32_t>(s->block_threads), wpb = block / 64u;
  if (s->waves_per_cu > 0) {
    // Sizing the LDS stack so that exactly this many waves' workgroups fit holds the cap whatever the register
    // allocation of the kernel variant is.  Measured: 24 gains 3 % on 1080p frames and loses 9-14 % on 4K frames
    // and incoherent rays, so it is off unless asked for.
    const uint32_t groups = std::max(1u, static_cast<uint32_t>(s->waves_per_cu) / wpb);
    const uint32_t min_bytes = 160u * 1024u / (groups + 1u) + 1u, entry_bytes = 8u * block;
    a.stack_entries = std::max(a.stack_entries, (min_bytes + entry_bytes - 1u) / entry_bytes);
  }
  if (s->refill > 0 && !count && mode == rt::kModeIntersect && a.img_w == 0 && s->width == 4 && s->quant &&
      a.n >= 65536u) {
    // incoherent rays: a fixed grid of waves (as many as fit the chip at once) streams through the rays, every wave refilling
    // its idle lanes from chunks it claims on a cursor word that this launch's stream zeroes first
    int rc = CheckStack(s, a, block);
    if (rc != RT_OK) return rc;
    a.cursor = s->d_cursor + (s->next_cursor++ % kCursorSlots) * kCursorStride;
    a.refill_min = static_cast<uint32_t>(s->refill);
    RT_HIP(hipMemsetAsync(a.cursor, 0, sizeof(uint32_t), stream));
    const uint32_t waves = static_cast<uint32_t>(s->num_cu) * 4u * rt::kRefillWavesPerSimd;
    hipError_t e_ = rt::LaunchRefill(a, std::min((a.ntasks + wpb - 1u) / wpb, (waves + wpb - 1u) / wpb), block, stream);
    if (e_ != hipSuccess) return Fail(RT_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(e_));
    return RT_OK;
  }
  uint32_t grid = (a.ntasks + wpb - 1u) / wpb;   // one wave per work unit (more below if the costliest units are split)
  if (s->persistent && !count) {
    // a fixed number of resident waves; each takes units from a cursor word that this launch's stream zeroes first
    grid = std::min(grid, static_cast<uint32_t>(s->num_cu) * std::max(1u, static_cast<uint32_t>(s->persistent_waves) / wpb));
    a.cursor = s->d_cursor + (s->next_cursor++ % kCursorSlots) * kCursorStride;
    RT_HIP(hipMemsetAsync(a.cursor, 0, sizeof(uint32_t), stream));
  }
  int fb_slot = -1;
  bool sort_now = false;   // this launch ends a period of the cost feedback: it records its costs and a sort follows it
  if (s->feedback == 1 && !count && !plain && a.n < kFeedbackMaxRays) {
    if (s->fb_ntasks != a.ntasks || s->fb_n != a.n || s->fb_img_w != a.img_w) {  // new shape: start over
      int rc = ResetFeedback(s);
      if (rc != RT_OK) return rc;
      const size_t padded = rt::OrderPaddedUnits(a.ntasks);
      RT_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_fb_bins), rt::OrderBinWords(a.ntasks) * sizeof(uint32_t)));   // sorts run one at a time
      for (int k = 0; k < 2; ++k) {
        RT_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_fb_cost[k]), padded * sizeof(uint32_t)));
        RT_HIP(hipMemset(s->d_fb_cost[k], 0, padded * sizeof(uint32_t)));
        RT_HIP(hipMalloc(reinterpret_cast<void **>(&s->d_fb_order[k]), static_cast<size_t>(a.ntasks) * sizeof(uint32_t)));
      }
      if (!s->ev_sort[0]) {
        for (int k = 0; k < 2; ++k) {
          RT_HIP(hipEventCreateWithFlags(&s->ev_main[k], hipEventDisableTiming));
          RT_HIP(hipEventCreateWithFlags(&s->ev_sort[k], hipEventDisableTiming));
        }
      }
      if (!s->side && s->side_stream) {
        // The side stream is NON-BLOCKING and of DEFAULT priority.  A stream created with default flags synchronises
        // implicitly with the legacy null stream, so a null-stream caller's trace launch would wait for everything the side
        // stream holds while the side stream's head waits for that launch's stop event.  A stream of raised priority is not
        // offered either: its head is a wait for a signal that only a lower-priority queue's kernel can send, and a
        // null-stream caller was seen to stop for good with it (non-blocking flag set) in one run of two.
        RT_HIP(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        unsigned side_flags = 0;
        RT_HIP(hipStreamGetFlags(s->side, &side_flags));
        if (!(side_flags & hipStreamNonBlocking)) return Fail(RT_ERR_DEVICE, "internal: the order-building stream is not non-blocking");
      }
      s->fb_ntasks = a.ntasks; s->fb_n = a.n; s->fb_img_w = a.img_w;
    }
    // Every `feedback_period`-th launch records its per-unit costs into set `fb_cur`; a sort of that set is enqueued on the
    // side stream behind it and the next recording goes to the other set (the first two periods are one launch long, so
    // that an order exists from the third launch on).  A launch follows the newest order whose sort has finished -- with
    // a period of one that is the order from two launches back, sorted beside the launch before this one.  Each sort
    // costs the main stream about 6 us (the completion signal the side stream waits for, and the sort's waves among the
    // next launch's), which the period spreads over its launches.
    fb_slot = s->fb_cur;
    sort_now = s->fb_since_sort + 1u >= static_cast<uint32_t>(s->feedback_period) || s->fb_launch < 2u;
    for (int k = 0; k < 2; ++k)
      if (s->fb_pending[k] && hipEventQuery(s->ev_sort[k]) == hipSuccess) s->fb_pending[k] = false;
    (void)hipGetLastError();   // hipErrorNotReady from a query is not an error of this call
    // a sort enqueued on the stream this launch goes to is finished, as far as this launch can tell, by stream order
    auto pending = [&](int k) { return s->fb_pending[k] && !(s->side_stream == 0 && s->fb_sort_on[k] == stream); };
    bool waited = false;
    if (sort_now && pending(fb_slot)) {
      // the sort that reads and zeroes this set's costs is still running: no launch may record into it yet
      RT_HIP(hipStreamWaitEvent(stream, s->ev_sort[fb_slot], 0));
      waited = true;
    }
    int use = -1;
    for (int k : {s->fb_newest, 1 - s->fb_newest})
      if (use < 0 && s->fb_sorted[k] && (!pending(k) || (waited && k == fb_slot))) use = k;
    if (use < 0 && s->fb_sorted[s->fb_newest]) {
      use = s->fb_newest;
      RT_HIP(hipStreamWaitEvent(stream, s->ev_sort[use], 0));
    }
    if (use >= 0) a.order = s->d_fb_order[use];
    // The full trace step splits only where it pays: a small launch (a 1080p pass, a rank's share of a frame) is mostly tail
    // and gains 3-40 %; a launch that fills the chip many times over, or one of several frames in flight, only pays for the
    // extra waves (-5 %), so callers who pipeline frames switch it off ("trace_split").
    // Measured by launch size (dragon, one launch at a time): 32 400 units (1080p) +3..14 %, a rank's share of a sharded frame
    // up to +40 %; 57 600 units (2560x1440) and more -1..-4 %.
    const bool small = a.ntasks <= static_cast<uint32_t>(s->split_max_units);
    const bool may_split = mode != rt::kModeTrace ? small : (s->trace_split == 1 || (s->trace_split < 0 && small));
    if (a.order && !a.cursor && may_split) {
      // The launch cannot end before its longest wave does, and a wave's length is the number of traversal steps its 64 rays
      // need between them.  The costliest units are therefore cut into four (16 rays per wave, 4x4 pixels or 16 records):
      // four shorter waves instead of one long one, for a few percent more waves in total.
      a.split_units = std::min(static_cast<uint32_t>(s->split_units), a.ntasks);
      grid = (a.ntasks + 3u * a.split_units + wpb - 1u) / wpb;
      a.coop = CoopFits(s, a) ? 1u : 0u;
    }
    if (a.order) a.prio_slots = static_cast<uint32_t>(static_cast<uint64_t>(a.ntasks) * static_cast<uint32_t>(s->prio_permille) / 1000u);
    // only the launch that ends a period records its costs: the others save the atomics, and an order from one camera
    // position served the launches after it better than the maximum over a period's positions did (+2 %)
    a.unit_cost = sort_now ? s->d_fb_cost[fb_slot] : nullptr;
    a.cost_radius = static_cast<uint32_t>(s->feedback_radius);
    a.cost_share = static_cast<uint32_t>(s->feedback_share);
  }
  if (s->feedback == 1 && count && !plain && s->fb_sorted[s->fb_newest] && s->fb_ntasks == a.ntasks && s->fb_n == a.n && s->fb_img_w == a.img_w) {
    // instrumented launch: follow the newest cost-sorted order, read-only
    const int newest = s->fb_newest;
    if (!s->fb_sorted[newest]) return Fail(RT_ERR_INVALID, "internal: no cost-sorted order yet");
    RT_HIP(hipStreamWaitEvent(stream, s->ev_sort[newest], 0));
    a.order = s->d_fb_order[newest];
    a.split_units = a.ntasks <= static_cast<uint32_t>(s->split_max_units) ? std::min(static_cast<uint32_t>(s->split_units), a.ntasks) : 0u;   // ... split like the launches it stands for
    grid = (a.ntasks + 3u * a.split_units + wpb - 1u) / wpb;
    a.coop = CoopFits(s, a) ? 1u : 0u;
  }
  {
    int rc = CheckStack(s, a, block);
    if (rc != RT_OK) return rc;
  }
  {
    hipError_t e_ = rt::LaunchClosestHit(a, s->width, s->quant != 0, mode, count, grid, block, stream,
                                         sort_now && s->side_stream ? s->ev_main[fb_slot] : nullptr);
    if (e_ != hipSuccess)
      return Fail(RT_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(e_) + " (grid " + std::to_string(grid) +
                                     ", block " + std::to_string(block) + ", stack entries " + std::to_string(a.stack_entries) + ")");
  }
  if (fb_slot >= 0) {
    if (s->fb_have_last && s->fb_last_stream != stream) {
      if (s->fb_left_used == s->fb_left.size()) {
        hipEvent_t e = nullptr;
        RT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->fb_left.push_back(e);
      }
      RT_HIP(hipEventRecord(s->fb_left[s->fb_left_used++], s->fb_last_stream));
    }
    s->fb_last_stream = stream;
    s->fb_have_last = true;
  }
  if (sort_now) {
    // The sort runs behind this launch and behind whatever earlier launches of the period other streams still hold: on the
    // side stream (ev_main[fb_slot] is signalled by the trace kernel itself, its launch carried the event), or, with
    // "side_stream" 0, on the launch's own stream.  The second form costs that stream the sort's few microseconds but keeps
    // the process to one stream per frame in flight: streams share the GPU's few hardware queues, and a side stream whose head
    // waits for one frame's launch holds up every other stream that was mapped to the same queue.
    hipStream_t sort_on = s->side_stream ? s->side : stream;
    if (s->side_stream) RT_HIP(hipStreamWaitEvent(sort_on, s->ev_main[fb_slot], 0));
    for (size_t k = 0; k < s->fb_left_used; ++k) RT_HIP(hipStreamWaitEvent(sort_on, s->fb_left[k], 0));
    s->fb_left_used = 0;
    // sorts share their work space: one that may still be running on another stream comes first
    const int other = 1 - fb_slot;
    if (s->fb_pending[other] && s->fb_sort_on[other] != sort_on) RT_HIP(hipStreamWaitEvent(sort_on, s->ev_sort[other], 0));
    RT_HIP(rt::LaunchBuildOrder(s->d_fb_cost[fb_slot], a.ntasks, s->d_fb_bins, s->d_fb_order[fb_slot], sort_on));
    RT_HIP(hipEventRecord(s->ev_sort[fb_slot], sort_on));
    s->fb_sorted[fb_slot] = s->fb_pending[fb_slot] = true;
    s->fb_sort_on[fb_slot] = sort_on;
    s->fb_newest = fb_slot;
    s->fb_cur = 1 - fb_slot;
    s->fb_since_sort = 0;
  } else if (fb_slot >= 0) {
    s->fb_since_sort++;
  }
  if (fb_slot >= 0) s->fb_launch++;
  return RT_OK;
}

constexpr size_t kChunk = size_t(1) << 30;

bool IsDevicePointer(const void *p)
{
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory reports an error; clear it
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int EnsureStage(void **buf, size_t *have, size_t want)
{
  if (*have >= want) return RT_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *have = 0;
  RT_HIP(hipMalloc(buf, want));
  *have = want;
  return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }

int rt_device_count(void)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    FailHip(e, "hipGetDeviceCount");
    return 0;
  }
  return n;
}

// ---------------------------------------------------------------- mesh
int rt_mesh_load_obj(const char *path, unsigned loader, rt_mesh **out)
{
  if (!path || !out) return Fail(RT_ERR_INVALID, "null argument");
  if (loader > RT_LOADER_EXTENDED) return Fail(RT_ERR_INVALID, "unknown loader mode");
  rt_mesh *m = new (std::nothrow) rt_mesh;
  if (!m) return Fail(RT_ERR_NOMEM, "out of memory");
  try {
    if (!rt::LoadObjFile(path, m->mesh, static_cast<rt::LoaderMode>(loader))) {
      delete m;
      return Fail(RT_ERR_IO, std::string("cannot open ") + path);
    }
  } catch (const std::exception &ex) {
    delete m;
    return Fail(RT_ERR_INVALID, std::string("obj load failed: ") + ex.what());
  }
  *out = m;
  return RT_OK;
}

int rt_mesh_from_arrays(const rt_vec3 *verts, size_t nverts, const rt_face *faces, size_t ntris,
                        const rt_float4 *normals, size_t nnormals, const rt_material *materials, size_t nmats,
                        rt_mesh **out)
{
  if (!out || (!verts && nverts) || (!faces && ntris) || (!normals && nnormals) || (!materials && nmats))
    return Fail(RT_ERR_INVALID, "null argument");
  rt_mesh *m = new (std::nothrow) rt_mesh;
  if (!m) return Fail(RT_ERR_NOMEM, "out of memory");
  m->mesh.verts.assign(verts, verts + nverts);
  m->mesh.faces.assign(faces, faces + ntris);
  if (nnormals) m->mesh.normals.assign(normals, normals + nnormals);
  if (nmats) m->mesh.mtllib.materials.assign(materials, materials + nmats);
  *out = m;
  return RT_OK;
}

size_t rt_mesh_nverts(const rt_mesh *m) { return m ? m->mesh.verts.size() : 0; }
size_t rt_mesh_ntris(const rt_mesh *m) { return m ? m->mesh.faces.size() : 0; }
size_t rt_mesh_nnormals(const rt_mesh *m) { return m ? m->mesh.normals.size() : 0; }
size_t rt_mesh_nmaterials(const rt_mesh *m) { return m ? m->mesh.mtllib.materials.size() : 0; }
const rt_vec3 *rt_mesh_verts(const rt_mesh *m) { return m ? m->mesh.verts.data() : nullptr; }
const rt_face *rt_mesh_faces(const rt_mesh *m) { return m ? m->mesh.faces.data() : nullptr; }
const rt_float4 *rt_mesh_normals(const rt_mesh *m) { return m ? m->mesh.normals.data() : nullptr; }
const rt_material *rt_mesh_materials(const rt_mesh *m) { return m ? m->mesh.mtllib.materials.data() : nullptr; }
size_t rt_mesh_nuvws(const rt_mesh *m) { return m ? m->mesh.uvws.size() : 0; }
const rt_vec3 *rt_mesh_uvws(const rt_mesh *m) { return m ? m->mesh.uvws.data() : nullptr; }
const uint32_t *rt_mesh_face_uvws(const rt_mesh *m) { return m && !m->mesh.face_uvws.empty() ? m->mesh.face_uvws.data() : nullptr; }

int rt_mesh_material_index(const rt_mesh *m, const char *name)
{
  if (!m || !name) return -1;
  auto it = m->mesh.mtllib.indices.find(name);
  return it == m->mesh.mtllib.indices.end() ? -1 : it->second;
}

const char *rt_mesh_log(const rt_mesh *m)
{
  if (!m) return "";
  rt_mesh *mm = const_cast<rt_mesh *>(m);
  mm->log_joined.clear();
  for (auto const &l : m->mesh.log) {
    mm->log_joined += l;
    mm->log_joined += '\n';
  }
  return mm->log_joined.c_str();
}

void rt_mesh_destroy(rt_mesh *m) { delete m; }

// ---------------------------------------------------------------- bvh
int rt_bvh_build(const rt_face *faces, const rt_vec3 *verts, size_t ntris, size_t nverts, unsigned mode, rt_bvh **out)
{
  if (!out || !faces || !verts) return Fail(RT_ERR_INVALID, "null argument");
  if (mode > RT_BVH_SAH) return Fail(RT_ERR_INVALID, "unknown bvh mode");
  if (ntris == 0 || nverts == 0) return Fail(RT_ERR_INVALID, "bvh build needs at least one triangle");
  for (size_t i = 0; i < ntris; ++i)
    if (faces[i].s[0] >= nverts || faces[i].s[1] >= nverts || faces[i].s[2] >= nverts)
      return Fail(RT_ERR_INVALID, "face vertex index out of range");
  rt_bvh *b = new (std::nothrow) rt_bvh;
  if (!b) return Fail(RT_ERR_NOMEM, "out of memory");
  try {
    if (mode == RT_BVH_SAH) rt::SahBVHBuild(&b->bvh, faces, verts, ntris, nverts);
    else rt::CentroidBVHBuild(&b->bvh, faces, verts, ntris, nverts, static_cast<rt::BvhMode>(mode));
  } catch (const std::exception &ex) {
    delete b;
    return Fail(RT_ERR_NOMEM, std::string("bvh build failed: ") + ex.what());
  }
  *out = b;
  return RT_OK;
}

const rt_bvh_node *rt_bvh_nodes(const rt_bvh *b) { return b ? b->bvh.nodes.data() : nullptr; }
size_t rt_bvh_nnodes(const rt_bvh *b) { return b ? b->bvh.nodes.size() : 0; }
int rt_bvh_depth(const rt_bvh *b) { return b ? b->bvh.depth : 0; }
int rt_bvh_leafcount(const rt_bvh *b) { return b ? b->bvh.leafcount : 0; }
int rt_bvh_fallback_splits(const rt_bvh *b) { return b ? b->bvh.fallback_splits : 0; }
void rt_bvh_destroy(rt_bvh *b) { delete b; }

// ---------------------------------------------------------------- rays
int rt_primary_rays(int W, int H, const float origin[3], rt_ray_isect *out)
{
  if (W <= 0 || H <= 0 || !origin || !out) return Fail(RT_ERR_INVALID, "bad primary-ray arguments");
  rt::PrimaryRays(W, H, origin, out);
  return RT_OK;
}

int rt_camera_matrix(float rot_x, float rot_y, const float trans[3], float out[16])
{
  if (!trans || !out) return Fail(RT_ERR_INVALID, "null argument");
  rt::CameraMatrix(rot_x, rot_y, trans, out);
  return RT_OK;
}

int rt_frame_camera(const rt_aabb *root, int W, int H, float origin_out[3])
{
  if (!root || !origin_out || W <= 0 || H <= 0) return Fail(RT_ERR_INVALID, "bad camera arguments");
  rt::FrameCamera(*root, W, H, origin_out);
  return RT_OK;
}

int rt_ao_rays(const rt_mesh *m, const rt_aabb *root, uint64_t seed, size_t first, size_t count, rt_ray_isect *out)
{
  if (!m || !root || !out) return Fail(RT_ERR_INVALID, "null argument");
  if (m->mesh.faces.empty()) return Fail(RT_ERR_INVALID, "mesh has no triangles");
  for (auto const &f : m->mesh.faces)
    if (f.s[0] >= m->mesh.verts.size() || f.s[1] >= m->mesh.verts.size() || f.s[2] >= m->mesh.verts.size())
      return Fail(RT_ERR_INVALID, "face vertex index out of range");
  rt::AoRays(m->mesh.verts.data(), m->mesh.faces.data(), m->mesh.faces.size(), *root, seed, first, count, out);
  return RT_OK;
}

// ---------------------------------------------------------------- scene
int rt_scene_create(const rt_vec3 *verts, size_t nverts, const rt_face *faces, size_t ntris,
                    const rt_float4 *normals, size_t nnormals, const rt_material *materials, size_t nmats,
                    const rt_bvh_node *bvh_nodes, size_t nnodes, int device, unsigned flags, rt_scene **out)
{
  if (!out || !verts || !faces || !bvh_nodes) return Fail(RT_ERR_INVALID, "null argument");
  if ((!normals && nnormals) || (!materials && nmats)) return Fail(RT_ERR_INVALID, "null argument");
  const bool on_device = (flags & RT_SCENE_FLATTEN_DEVICE) != 0;
  flags &= ~RT_SCENE_FLATTEN_DEVICE;
  if (flags != RT_SCENE_WIDE2 && flags != RT_SCENE_WIDE4) return Fail(RT_ERR_INVALID, "unknown scene flags");
  if (on_device && device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "RT_SCENE_FLATTEN_DEVICE needs a device");
  if (device != RT_DEVICE_NONE) {
    int ndev = 0;
    RT_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return Fail(RT_ERR_DEVICE, "no such HIP device");
  }

  rt_scene *s = new (std::nothrow) rt_scene;
  if (!s) return Fail(RT_ERR_NOMEM, "out of memory");
  std::string why;
  rt::FlatIndex ix;
  try {
    // on the device: the host only validates and numbers; the records are built by flatten_device.hip further down
    if (on_device) why = rt::FlattenIndexPass(bvh_nodes, nnodes, verts, nverts, faces, ntris, &ix);
    else why = rt::Flatten(bvh_nodes, nnodes, verts, nverts, faces, ntris, &s->flat);
  } catch (const std::exception &ex) {
    why = ex.what();
  }
  if (!why.empty()) {
    delete s;
    return Fail(RT_ERR_INVALID, "bad scene: " + why);
  }
  s->device = device;
  s->width = flags == RT_SCENE_WIDE4 ? 4 : 2;
  s->nverts = nverts; s->ntris = ntris; s->nnormals = nnormals; s->nmats = nmats;

  if (device == RT_DEVICE_NONE) {  // host-only: layouts and fetch counting, nothing traces
    *out = s;
    return RT_OK;
  }
  DeviceGuard guard(device);
  int rc = guard.ok ? RT_OK : Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if (rc == RT_OK && hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cu = prop.multiProcessorCount;
  if (rc == RT_OK) {   // look the fixed tables of rt_intersect_static up now, so that its calls have nothing left to set up
    StaticState *st = nullptr;
    std::lock_guard<std::mutex> lock(g_static_mutex);
    rc = StaticScratch(device, &st);
  }
  if (rc == RT_OK) rc = Upload(&s->d_verts, verts, nverts * sizeof(rt_vec3));
  if (rc == RT_OK) rc = Upload(&s->d_faces, faces, ntris * sizeof(rt_face));
  if (rc == RT_OK && !on_device) {
    rc = Upload(&s->d_nodes2, s->flat.nodes2.data(), s->flat.nodes2.size() * sizeof(rt::GpuNode2));
    if (rc == RT_OK) rc = Upload(&s->d_nodes4, s->flat.nodes4.data(), s->flat.nodes4.size() * sizeof(rt::GpuNode4));
    if (rc == RT_OK) rc = Upload(&s->d_nodesq2, s->flat.nodesq2.data(), s->flat.nodesq2.size() * sizeof(rt::GpuChildQ));
    if (rc == RT_OK) rc = Upload(&s->d_nodesq4, s->flat.nodesq4.data(), s->flat.nodesq4.size() * sizeof(rt::GpuChildQ));
    if (rc == RT_OK) rc = Upload(&s->d_tris, s->flat.tris.data(), s->flat.tris.size() * sizeof(rt::GpuTri));
  }
  if (rc == RT_OK && on_device) {
    // the caller's 80-byte nodes go up as they are; the kernels of flatten_device.hip turn them into the records
    rt_bvh_node *d_caller_nodes = nullptr;
    rc = Upload(&d_caller_nodes, bvh_nodes, nnodes * sizeof(rt_bvh_node));
    rt::DeviceFlat df;
    if (rc == RT_OK) {
      hipError_t e = rt::FlattenOnDevice(ix, d_caller_nodes, nnodes, s->d_verts, s->d_faces, &df);
      if (e != hipSuccess) rc = FailHip(e, "device flatten");
    }
    (void)hipFree(d_caller_nodes);
    if (rc == RT_OK) {
      s->d_nodes2 = reinterpret_cast<float4 *>(df.nodes2); s->d_nodesq2 = reinterpret_cast<float4 *>(df.nodesq2);
      s->d_nodes4 = reinterpret_cast<float4 *>(df.nodes4); s->d_nodesq4 = reinterpret_cast<float4 *>(df.nodesq4);
      s->d_tris = reinterpret_cast<float4 *>(df.tris);
      // host copies of the device-built records: the fetch counter, the wave model and rt_scene_records read them
      s->flat.nodes2.resize(df.n2); s->flat.nodesq2.resize(df.n2 * 2);
      s->flat.nodes4.resize(df.n4); s->flat.nodesq4.resize(df.n4 * 4);
      s->flat.tris.resize(df.ntris);
      hipError_t e = hipMemcpy(s->flat.nodes2.data(), df.nodes2, df.n2 * sizeof(rt::GpuNode2), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(s->flat.nodesq2.data(), df.nodesq2, df.n2 * 2 * sizeof(rt::GpuChildQ), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(s->flat.nodes4.data(), df.nodes4, df.n4 * sizeof(rt::GpuNode4), hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(s->flat.nodesq4.data(), df.nodesq4, df.n4 * 4 * sizeof(rt::GpuChildQ), hipMemcpyDeviceToHost);
      if (e == hipSuccess && df.ntris) e = hipMemcpy(s->flat.tris.data(), df.tris, df.ntris * sizeof(rt::GpuTri), hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc = FailHip(e, "device flatten read-back");
      rt::FlattenFinish(bvh_nodes, ix, df.depth4, &s->flat);
    }
  }
  if (rc == RT_OK) rc = Upload(&s->d_normals, normals, nnormals * sizeof(rt_float4));
  if (rc == RT_OK) rc = Upload(&s->d_mats, materials, nmats * sizeof(rt_material));
  if (rc == RT_OK) rc = Upload(&s->d_stats, nullptr, kStatShards * 16 * sizeof(unsigned long long));
  if (rc == RT_OK) rc = Upload(&s->d_cursor, nullptr, kCursorSlots * kCursorStride * sizeof(uint32_t));
  if (rc != RT_OK) {
    std::string keep = g_error;
    rt_scene_destroy(s);
    g_error = keep;
    return rc;
  }
  *out = s;
  return RT_OK;
}

void rt_scene_destroy(rt_scene *s)
{
  if (!s) return;
  if (s->device == RT_DEVICE_NONE) {
    delete s;
    return;
  }
  DeviceGuard guard(s->device);
  (void)hipFree(s->d_nodes2); (void)hipFree(s->d_nodes4); (void)hipFree(s->d_nodesq2); (void)hipFree(s->d_nodesq4); (void)hipFree(s->d_tris);
  (void)hipFree(s->d_verts); (void)hipFree(s->d_faces); (void)hipFree(s->d_normals); (void)hipFree(s->d_mats);
  (void)hipFree(s->d_stats); (void)hipFree(s->d_cursor); (void)hipFree(s->d_order); (void)ResetFeedback(s);
  if (s->side) (void)hipStreamDestroy(s->side);
  for (int k = 0; k < 2; ++k) {
    if (s->ev_main[k]) (void)hipEventDestroy(s->ev_main[k]);
    if (s->ev_sort[k]) (void)hipEventDestroy(s->ev_sort[k]);
  }
  for (hipEvent_t e : s->fb_left) (void)hipEventDestroy(e);
  (void)hipFree(s->d_stage_in); (void)hipFree(s->d_stage_out);
  for (int k = 0; k < 2; ++k) {
    if (s->pin_in[k]) (void)hipHostFree(s->pin_in[k]);
    if (s->pin_out[k]) (void)hipHostFree(s->pin_out[k]);
    if (s->ev_h2d[k]) { (void)hipEventDestroy(s->ev_h2d[k]); (void)hipEventDestroy(s->ev_trace[k]); (void)hipEventDestroy(s->ev_d2h[k]); }
  }
  if (s->ev_entry) (void)hipEventDestroy(s->ev_entry);
  if (s->st_h2d) { (void)hipStreamDestroy(s->st_h2d); (void)hipStreamDestroy(s->st_trace); (void)hipStreamDestroy(s->st_d2h); }
  delete s;
}

int rt_scene_info(const rt_scene *s, uint64_t info[8])
{
  if (!s || !info) return Fail(RT_ERR_INVALID, "null argument");
  info[0] = s->flat.nodes2.size(); info[1] = s->flat.nodes4.size(); info[2] = s->flat.tris.size();
  info[3] = static_cast<uint64_t>(s->flat.depth2); info[4] = static_cast<uint64_t>(s->flat.depth4);
  info[5] = s->quant ? 2 * sizeof(rt::GpuChildQ) : sizeof(rt::GpuNode2);
  info[6] = s->quant ? 4 * sizeof(rt::GpuChildQ) : sizeof(rt::GpuNode4);
  info[7] = sizeof(rt::GpuTri);
  return RT_OK;
}

int rt_scene_records(const rt_scene *s, int which, void *out, size_t capacity, size_t *bytes)
{
  if (!s || !bytes) return Fail(RT_ERR_INVALID, "null argument");
  const void *src = nullptr;
  size_t n = 0;
  switch (which) {
    case 0: src = s->flat.nodes2.data(); n = s->flat.nodes2.size() * sizeof(rt::GpuNode2); break;
    case 1: src = s->flat.nodesq2.data(); n = s->flat.nodesq2.size() * sizeof(rt::GpuChildQ); break;
    case 2: src = s->flat.nodes4.data(); n = s->flat.nodes4.size() * sizeof(rt::GpuNode4); break;
    case 3: src = s->flat.nodesq4.data(); n = s->flat.nodesq4.size() * sizeof(rt::GpuChildQ); break;
    case 4: src = s->flat.tris.data(); n = s->flat.tris.size() * sizeof(rt::GpuTri); break;
    default: return Fail(RT_ERR_INVALID, "unknown record array");
  }
  *bytes = n;
  if (!out) return RT_OK;
  if (capacity < n) return Fail(RT_ERR_INVALID, "record buffer too small");
  if (n) std::memcpy(out, src, n);
  return RT_OK;
}

int rt_scene_set_param(rt_scene *s, const char *key, int value)
{
  if (!s || !key) return Fail(RT_ERR_INVALID, "null argument");
  const std::string k(key);
  if (k == "width") {
    if (value != 2 && value != 4) return Fail(RT_ERR_INVALID, "width must be 2 or 4");
    s->width = value;
  } else if (k == "feedback") {
    if (value < 0 || value > 1) return Fail(RT_ERR_INVALID, "feedback must be 0 or 1");
    s->feedback = value;
    if (s->device != RT_DEVICE_NONE) {
      DeviceGuard guard(s->device);
      int rc = ResetFeedback(s);
      if (rc != RT_OK) return rc;
    }
  } else if (k == "waves_per_cu") {
    if (value < 0 || value > 64) return Fail(RT_ERR_INVALID, "waves_per_cu must be in 0..64");
    s->waves_per_cu = value;
  } else if (k == "feedback_share") {
    if (value < 0 || value > 128) return Fail(RT_ERR_INVALID, "feedback_share must be in 0..128");
    s->feedback_share = value;
  } else if (k == "feedback_radius") {
    if (value < 0 || value > 3) return Fail(RT_ERR_INVALID, "feedback_radius must be in 0..3");
    s->feedback_radius = value;
  } else if (k == "centre_first") {
    s->centre_first = value != 0;
  } else if (k == "leaf_pct") {
    if (value < 1 || value > 100) return Fail(RT_ERR_INVALID, "leaf_pct must be in 1..100");
    s->leaf_pct = value;
  } else if (k == "feedback_period") {
    if (value < 1 || value > 1024) return Fail(RT_ERR_INVALID, "feedback_period must be in 1..1024");
    s->feedback_period = value;
  } else if (k == "model_width") {
    if (value != 0 && value != 8) return Fail(RT_ERR_INVALID, "model_width must be 0 (as traced) or 8");
    s->model_width = value;
  } else if (k == "side_stream") {
    if (value < 0 || value > 1) return Fail(RT_ERR_INVALID, "side_stream must be 0 or 1");
    if (value != s->side_stream) {
      DeviceGuard guard(s->device);
      int rc = ResetFeedback(s);
      if (rc != RT_OK) return rc;
      if (s->side) (void)hipStreamDestroy(s->side);
      s->side = nullptr;
    }
    s->side_stream = value;
  } else if (k == "refill") {
    if (value < 0 || value > 64) return Fail(RT_ERR_INVALID, "refill must be in 0..64");
    s->refill = value;
  } else if (k == "split_units") {
    if (value < 0 || value > 65536) return Fail(RT_ERR_INVALID, "split_units must be in 0..65536");
    s->split_units = value;
  } else if (k == "split_max_units") {
    if (value < 0 || value > (1 << 24)) return Fail(RT_ERR_INVALID, "split_max_units must be in 0..16777216");
    s->split_max_units = value;
  } else if (k == "trace_split") {
    if (value < -1 || value > 1) return Fail(RT_ERR_INVALID, "trace_split must be -1, 0 or 1");
    s->trace_split = value;
  } else if (k == "split_coop") {
    if (value < 0 || value > 1) return Fail(RT_ERR_INVALID, "split_coop must be 0 or 1");
    s->split_coop = value;
  } else if (k == "prio_permille") {
    if (value < 0 || value > 1000) return Fail(RT_ERR_INVALID, "prio_permille must be in 0..1000");
    s->prio_permille = value;
  } else if (k == "lds_stack") {
    if (value < static_cast<int>(rt::kMinLdsEntries) || value > 32) return Fail(RT_ERR_INVALID, "lds_stack must be in 8..32");
    s->lds_stack = value;
  } else if (k == "persistent") {
    if (value < 0 || value > 1) return Fail(RT_ERR_INVALID, "persistent must be 0 or 1");
    s->persistent = value;
  } else if (k == "persistent_waves") {
    if (value < 1 || value > 64) return Fail(RT_ERR_INVALID, "persistent_waves must be in 1..64");
    s->persistent_waves = value;
  } else if (k == "extra_stack") {
    if (value < 0 || value > 128) return Fail(RT_ERR_INVALID, "extra_stack out of range");
    s->extra_stack = value;
  } else if (k == "quantized") {
    s->quant = value != 0;
  } else if (k == "block_threads") {
    if (value != 64 && value != 128 && value != 256) return Fail(RT_ERR_INVALID, "block_threads must be 64, 128 or 256");
    s->block_threads = value;
  } else if (k == "image_width") {
    if (value < 0) return Fail(RT_ERR_INVALID, "image_width must be >= 0");
    s->image_width = value;
  } else {
    return Fail(RT_ERR_INVALID, "unknown parameter " + k);
  }
  return RT_OK;
}

// ---------------------------------------------------------------- hot path
int rt_intersect_async(const rt_scene *s, const void *rays_dev, size_t n, void *hits_dev, void *hits16_dev,
                       void *stream)
{
  if (!s) return Fail(RT_ERR_INVALID, "null scene");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (n == 0) return RT_OK;
  if (!rays_dev || (!hits_dev && !hits16_dev)) return Fail(RT_ERR_INVALID, "null ray or hit buffer");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  for (size_t off = 0; off < n; off += kChunk) {
    rt::KernelArgs a;
    FillArgs(s, a);
    a.rays = static_cast<const rt_ray_isect *>(rays_dev) + off;
    a.hits = hits_dev ? static_cast<rt_ray_isect *>(hits_dev) + off : nullptr;
    a.hits16 = hits16_dev ? static_cast<rt_hit16 *>(hits16_dev) + off : nullptr;
    a.n = static_cast<uint32_t>(std::min(kChunk, n - off));
    int rc = Enqueue(s, a, rt::kModeIntersect, false, static_cast<hipStream_t>(stream));
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

int rt_intersect_compact(const rt_scene *s, const void *rays_dev, size_t n, void *out_dev, size_t record_bytes, void *stream)
{
  if (!s) return Fail(RT_ERR_INVALID, "null scene");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (record_bytes != 4u && record_bytes != 8u && record_bytes != 16u)
    return Fail(RT_ERR_INVALID, "compact records are 4 ({id}), 8 ({id, t}) or 16 ({id, t, u, v}) bytes");
  if (n == 0) return RT_OK;
  if (!rays_dev || !out_dev) return Fail(RT_ERR_INVALID, "null ray or hit buffer");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  for (size_t off = 0; off < n; off += kChunk) {
    rt::KernelArgs a;
    FillArgs(s, a);
    a.rays = static_cast<const rt_ray_isect *>(rays_dev) + off;
    a.hits16 = reinterpret_cast<rt_hit16 *>(static_cast<char *>(out_dev) + off * record_bytes);
    a.compact_words = static_cast<uint32_t>(record_bytes / 4u);
    a.n = static_cast<uint32_t>(std::min(kChunk, n - off));
    int rc = Enqueue(s, a, rt::kModeIntersect, false, static_cast<hipStream_t>(stream));
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

int rt_intersect_static(const rt_scene *s, const void *rays_dev, size_t n, void *hits_dev, void *hits16_dev, void *stream)
{
  if (!s) return Fail(RT_ERR_INVALID, "null scene");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (n == 0) return RT_OK;
  if (!rays_dev || (!hits_dev && !hits16_dev)) return Fail(RT_ERR_INVALID, "null ray or hit buffer");
  if (n > kStaticMaxRays)
    return Fail(RT_ERR_INVALID, "rt_intersect_static takes at most " + std::to_string(kStaticMaxRays) + " rays per call (" +
                                    std::to_string(n) + " given); split the call or use rt_intersect_async");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  // One call at a time per process touches the table state and enqueues against it: two threads, two scenes or two streams
  // can no longer interleave a half-updated state with a launch (the enqueues are short; the GPU work is not waited for).
  std::lock_guard<std::mutex> lock(g_static_mutex);
  StaticState *st = nullptr;
  int rc = StaticScratch(s->device, &st);
  if (rc != RT_OK) return rc;
  hipStream_t q = static_cast<hipStream_t>(stream);
  rt::KernelArgs a;
  FillArgs(s, a);
  a.rays = static_cast<const rt_ray_isect *>(rays_dev);
  a.hits = static_cast<rt_ray_isect *>(hits_dev);
  a.hits16 = static_cast<rt_hit16 *>(hits16_dev);
  a.n = static_cast<uint32_t>(n);
  SetShape(s, a, rt::kModeIntersect);
  if (a.ntasks > kStaticMaxUnits)   // an image hint can round tiles up past n / 64
    return Fail(RT_ERR_INVALID, "rt_intersect_static: " + std::to_string(a.ntasks) + " work units exceed the fixed tables (" +
                                    std::to_string(kStaticMaxUnits) + ")");
  rc = CheckStack(s, a, 64u);
  if (rc != RT_OK) return rc;
  // Calls on another stream than the one before: what the stream being left still has enqueued (a trace that reads the
  // order half, a sort that rewrites both halves) must be done before this stream touches the tables.
  if (st->have_last && st->last_stream != q) {
    RT_HIP(hipEventRecord(st->ev_left, st->last_stream));
    RT_HIP(hipStreamWaitEvent(q, st->ev_left, 0));
  }
  st->last_stream = q;
  st->have_last = true;
  uint32_t *cost = st->ptr, *order = st->ptr + kStaticMaxUnits, *bins = st->ptr + 2u * kStaticMaxUnits;
  const size_t padded = rt::OrderPaddedUnits(a.ntasks);   // <= kStaticMaxUnits, and its bin counters <= kStaticBinWords
  if (st->ntasks != a.ntasks || st->n != a.n || st->img_w != a.img_w || st->scene_id != s->id) {
    // the array keeps the previous call's contents: a new shape, or another scene, starts from zero costs and no order
    RT_HIP(hipMemsetAsync(cost, 0, padded * sizeof(uint32_t), q));
    st->ntasks = a.ntasks; st->n = a.n; st->img_w = a.img_w; st->scene_id = s->id;
    st->order_valid = false;
    st->since_sort = 0;
  }
  // as in rt_intersect_async: only every feedback_period-th call (and the first of a shape) records costs and re-sorts
  const bool sort_now = !st->order_valid || st->since_sort + 1u >= static_cast<uint32_t>(s->feedback_period);
  a.order = st->order_valid ? order : nullptr;
  a.unit_cost = sort_now ? cost : nullptr;
  a.cost_radius = static_cast<uint32_t>(s->feedback_radius);
  a.cost_share = static_cast<uint32_t>(s->feedback_share);
  hipError_t e = rt::LaunchClosestHit(a, s->width, s->quant != 0, rt::kModeIntersect, false, a.ntasks, 64u, q);
  if (e != hipSuccess) return FailHip(e, "kernel launch failed");
  if (!sort_now) {
    st->since_sort++;
    return RT_OK;
  }
  st->since_sort = 0;
  // same stream, after the trace: turns this call's costs into the next calls' order and zeroes the cost half
  e = rt::LaunchBuildOrder(cost, a.ntasks, bins, order, q);
  if (e != hipSuccess) {
    st->order_valid = false;
    st->ntasks = 0;   // the cost half may be dirty: start over next time
    return FailHip(e, "order kernel launch failed");
  }
  st->order_valid = true;
  return RT_OK;
}

int rt_trace_primary(const rt_scene *s, int width, int height, const float origin[3], void *hits_dev, void *hits16_dev,
                     void *stream)
{
  if (!s) return Fail(RT_ERR_INVALID, "null scene");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (width <= 0 || height <= 0 || !origin) return Fail(RT_ERR_INVALID, "bad primary-ray arguments");
  if (static_cast<uint64_t>(width) * static_cast<uint64_t>(height) >= kFeedbackMaxRays)
    return Fail(RT_ERR_INVALID, "image too large for one launch");
  if (!hits_dev && !hits16_dev) return Fail(RT_ERR_INVALID, "null hit buffer");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  rt::KernelArgs a;
  FillArgs(s, a);
  a.hits = static_cast<rt_ray_isect *>(hits_dev);
  a.hits16 = static_cast<rt_hit16 *>(hits16_dev);
  a.n = static_cast<uint32_t>(width) * static_cast<uint32_t>(height);
  a.img_w = static_cast<uint32_t>(width);   // the generator needs pixel coordinates: always traced as an image
  a.cam[0] = origin[0]; a.cam[1] = origin[1]; a.cam[2] = origin[2];
  return Enqueue(s, a, rt::kModePrimary, false, static_cast<hipStream_t>(stream));
}

// Host-pointer form: rays and/or hits live in ordinary (pageable) host memory.  Measured on the MI355X host
// (tools/ubench/host_xfer.cpp, one 1080p table = 132.7 MB): hipMemcpy from pageable memory 4.8 ms up / 7.8 ms down,
// from pinned memory 2.3 ms each way, one core's memcpy into pinned memory 4.4 ms.  So the records go through
// pinned staging in chunks: the calling thread copies chunk k+1 into a pinned buffer while the DMA engines upload
// chunk k and download chunk k-1 and the kernel traces in between, and a second thread copies finished chunks out
// to the caller's buffer.  Pinning the caller's buffer in place was refused by the runtime for malloc'd memory.
namespace {

constexpr size_t kHostChunkRays = size_t(1) << 18;   // 16 MiB of records per direction per chunk

// A thread that copies the upper half of a block while its owner copies the lower half: one core moves ~30 GB/s into
// or out of pinned memory, two keep up with the 57 GB/s the DMA engines sustain.  Lives for one call.
class CopyMate {
 public:
  CopyMate() : thread_([this] { Run(); }) {}
  ~CopyMate()
  {
    quit_.store(true, std::memory_order_release);
    thread_.join();
  }
  void Copy(void *dst, const void *src, size_t bytes)
  {
    const size_t half = (bytes / 2) & ~size_t(63);
    dst_ = static_cast<char *>(dst) + half;
    src_ = static_cast<const char *>(src) + half;
    bytes_ = bytes - half;
    posted_.store(posted_.load(std::memory_order_relaxed) + 1, std::memory_order_release);
    std::memcpy(dst, src, half);
    while (done_.load(std::memory_order_acquire) != posted_.load(std::memory_order_relaxed)) std::this_thread::yield();
  }

 private:
  void Run()
  {
    size_t seen = 0;
    for (;;) {
      while (posted_.load(std::memory_order_acquire) == seen) {
        if (quit_.load(std::memory_order_acquire)) return;
        std::this_thread::yield();
      }
      ++seen;
      std::memcpy(dst_, src_, bytes_);
      done_.store(seen, std::memory_order_release);
    }
  }
  char *dst_ = nullptr;
  const char *src_ = nullptr;
  size_t bytes_ = 0;
  std::atomic<size_t> posted_{0}, done_{0};
  std::atomic<bool> quit_{false};
  std::thread thread_;
};

int EnsureHostPipeline(const rt_scene *s)
{
  if (s->st_h2d) return RT_OK;
  for (int k = 0; k < 2; ++k) {
    RT_HIP(hipHostMalloc(&s->pin_in[k], kHostChunkRays * sizeof(rt_ray_isect), hipHostMallocDefault));
    RT_HIP(hipHostMalloc(&s->pin_out[k], kHostChunkRays * sizeof(rt_ray_isect), hipHostMallocDefault));
    RT_HIP(hipEventCreateWithFlags(&s->ev_h2d[k], hipEventDisableTiming));
    RT_HIP(hipEventCreateWithFlags(&s->ev_trace[k], hipEventDisableTiming));
    RT_HIP(hipEventCreateWithFlags(&s->ev_d2h[k], hipEventDisableTiming));
  }
  RT_HIP(hipEventCreateWithFlags(&s->ev_entry, hipEventDisableTiming));
  RT_HIP(hipStreamCreateWithFlags(&s->st_trace, hipStreamNonBlocking));
  RT_HIP(hipStreamCreateWithFlags(&s->st_d2h, hipStreamNonBlocking));
  RT_HIP(hipStreamCreateWithFlags(&s->st_h2d, hipStreamNonBlocking));   // last: its presence marks the set complete
  return RT_OK;
}

}  // namespace

int ray_scene_intersection(const void *rays, size_t n, const rt_scene *scene, void *hits)
{
  if (!scene) return Fail(RT_ERR_INVALID, "null scene");
  if (scene->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (n == 0) return RT_OK;
  if (!rays || !hits) return Fail(RT_ERR_INVALID, "null ray or hit buffer");
  if (n > (SIZE_MAX / sizeof(rt_ray_isect))) return Fail(RT_ERR_INVALID, "ray count overflows");
  DeviceGuard guard(scene->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  const size_t bytes = n * sizeof(rt_ray_isect);
  const bool in_dev = IsDevicePointer(rays), out_dev = IsDevicePointer(hits);
  if (in_dev && out_dev) {
    int rc = rt_intersect_async(scene, rays, n, hits, nullptr, nullptr);
    if (rc != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(nullptr));
    return RT_OK;
  }
  const rt_scene *s = scene;
  int rc = EnsureHostPipeline(s);
  if (rc != RT_OK) return rc;
  const rt_ray_isect *d_in = static_cast<const rt_ray_isect *>(rays);
  rt_ray_isect *d_out = static_cast<rt_ray_isect *>(hits);
  if (!in_dev) {
    rc = EnsureStage(&s->d_stage_in, &s->stage_in_bytes, bytes);
    if (rc != RT_OK) return rc;
    d_in = static_cast<const rt_ray_isect *>(s->d_stage_in);
  }
  if (!out_dev) {
    rc = EnsureStage(&s->d_stage_out, &s->stage_out_bytes, bytes);
    if (rc != RT_OK) return rc;
    d_out = static_cast<rt_ray_isect *>(s->d_stage_out);
  }
  // whole image rows per chunk when the rays are declared an image, so that every chunk is still walked in 8x8 tiles
  size_t chunk = kHostChunkRays;
  if (s->image_width > 0 && n % static_cast<size_t>(s->image_width) == 0 && static_cast<size_t>(s->image_width) * 8u <= kHostChunkRays)
    chunk = kHostChunkRays / (static_cast<size_t>(s->image_width) * 8u) * (static_cast<size_t>(s->image_width) * 8u);
  const size_t nchunks = (n + chunk - 1) / chunk;

  // The copy-out thread: waits for chunk c's download, copies it to the caller, frees the pinned slot.
  std::atomic<size_t> enqueued{0}, copied_out{0};
  std::atomic<bool> failed{false};
  std::thread out_thread;
  if (!out_dev)
    out_thread = std::thread([&] {
      (void)hipSetDevice(s->device);
      CopyMate mate;
      for (size_t c = 0; c < nchunks; ++c) {
        while (enqueued.load(std::memory_order_acquire) <= c) {
          if (failed.load(std::memory_order_acquire)) return;
          std::this_thread::yield();
        }
        const size_t off = c * chunk, cnt = std::min(chunk, n - off);
        if (hipEventSynchronize(s->ev_d2h[c & 1]) != hipSuccess) { failed.store(true, std::memory_order_release); return; }
        mate.Copy(static_cast<rt_ray_isect *>(hits) + off, s->pin_out[c & 1], cnt * sizeof(rt_ray_isect));
        copied_out.store(c + 1, std::memory_order_release);
      }
    });
  auto finish = [&](int code) {
    if (code != RT_OK) failed.store(true, std::memory_order_release);
    if (out_thread.joinable()) out_thread.join();
    return code;
  };
#define RT_PIPE(call)                                             \
  do {                                                            \
    hipError_t e_ = (call);                                       \
    if (e_ != hipSuccess) return finish(FailHip(e_, #call));      \
  } while (0)

  if (in_dev || out_dev) {
    // A device-side buffer belongs to the caller's in-order null stream (the reference's one in-order queue,
    // render_client/img_main.cpp:313): rays that a kernel or copy enqueued there is still producing, hits that earlier work
    // there still reads.  The pipeline's streams do not synchronise with the null stream by themselves, so they wait for
    // everything it held when the call came in; the call returns only after its own streams have drained.
    RT_PIPE(hipEventRecord(s->ev_entry, nullptr));
    RT_PIPE(hipStreamWaitEvent(s->st_h2d, s->ev_entry, 0));
    RT_PIPE(hipStreamWaitEvent(s->st_trace, s->ev_entry, 0));
  }
  CopyMate in_mate;
  for (size_t c = 0; c < nchunks; ++c) {
    const int slot = static_cast<int>(c & 1);
    const size_t off = c * chunk, cnt = std::min(chunk, n - off), cbytes = cnt * sizeof(rt_ray_isect);
    if (!in_dev) {
      if (c >= 2) RT_PIPE(hipEventSynchronize(s->ev_h2d[slot]));   // the upload that last used this pinned slot is done
      in_mate.Copy(s->pin_in[slot], static_cast<const rt_ray_isect *>(rays) + off, cbytes);
      RT_PIPE(hipMemcpyAsync(const_cast<rt_ray_isect *>(d_in) + off, s->pin_in[slot], cbytes, hipMemcpyHostToDevice, s->st_h2d));
      RT_PIPE(hipEventRecord(s->ev_h2d[slot], s->st_h2d));
      RT_PIPE(hipStreamWaitEvent(s->st_trace, s->ev_h2d[slot], 0));
    }
    {
      rt::KernelArgs a;
      FillArgs(s, a);
      a.rays = d_in + off;
      a.hits = d_out + off;
      a.n = static_cast<uint32_t>(cnt);
      rc = Enqueue(s, a, rt::kModeIntersect, false, s->st_trace, true);
      if (rc != RT_OK) return finish(rc);
    }
    if (!out_dev) {
      RT_PIPE(hipEventRecord(s->ev_trace[slot], s->st_trace));
      // the pinned output slot is free once the copy-out thread has drained chunk c-2
      while (c >= 2 && copied_out.load(std::memory_order_acquire) < c - 1) {
        if (failed.load(std::memory_order_acquire)) return finish(Fail(RT_ERR_DEVICE, "download of a chunk failed"));
        std::this_thread::yield();
      }
      RT_PIPE(hipStreamWaitEvent(s->st_d2h, s->ev_trace[slot], 0));
      RT_PIPE(hipMemcpyAsync(s->pin_out[slot], d_out + off, cbytes, hipMemcpyDeviceToHost, s->st_d2h));
      RT_PIPE(hipEventRecord(s->ev_d2h[slot], s->st_d2h));
      enqueued.store(c + 1, std::memory_order_release);
    }
  }
#undef RT_PIPE
  if (out_thread.joinable()) out_thread.join();
  if (failed.load()) return Fail(RT_ERR_DEVICE, "download of a chunk failed");
  hipError_t e = hipStreamSynchronize(s->st_trace);
  if (e == hipSuccess) e = hipStreamSynchronize(s->st_h2d);
  if (e != hipSuccess) return FailHip(e, "host-pointer pipeline");
  return RT_OK;
}

static int TracePass(const rt_scene *s, const void *rays_in_dev, void *rays_out_dev, void *pixels_dev, bool pixels8, size_t n,
                     const float cam[16], unsigned trace_operation, void *stream);

int rt_trace_pass(const rt_scene *s, const void *rays_in_dev, void *rays_out_dev, float *rgba_dev, size_t n,
                  const float cam[16], unsigned trace_operation, void *stream)
{
  return TracePass(s, rays_in_dev, rays_out_dev, rgba_dev, false, n, cam, trace_operation, stream);
}

int rt_trace_pass_bgra8(const rt_scene *s, const void *rays_in_dev, void *rays_out_dev, void *bgra_dev, size_t n,
                        const float cam[16], unsigned trace_operation, void *stream)
{
  return TracePass(s, rays_in_dev, rays_out_dev, bgra_dev, true, n, cam, trace_operation, stream);
}

static int TracePass(const rt_scene *s, const void *rays_in_dev, void *rays_out_dev, void *pixels_dev, bool pixels8, size_t n,
                     const float cam[16], unsigned trace_operation, void *stream)
{
  float *rgba_dev = static_cast<float *>(pixels_dev);
  if (!s) return Fail(RT_ERR_INVALID, "null scene");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (trace_operation > 1) return Fail(RT_ERR_INVALID, "trace_operation must be 0 or 1");
  if (n == 0) return RT_OK;
  if (!rays_in_dev || (!rays_out_dev && !rgba_dev)) return Fail(RT_ERR_INVALID, "null ray or output buffer");
  if (trace_operation == 0 && !cam) return Fail(RT_ERR_INVALID, "camera matrix required for operation 0");
  if (rays_in_dev == rays_out_dev) return Fail(RT_ERR_INVALID, "input and output records must not alias");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  for (size_t off = 0; off < n; off += kChunk) {
    rt::KernelArgs a;
    FillArgs(s, a);
    a.rays = static_cast<const rt_ray_isect *>(rays_in_dev) + off;
    a.hits = rays_out_dev ? static_cast<rt_ray_isect *>(rays_out_dev) + off : nullptr;
    a.rgba = rgba_dev ? rgba_dev + off * (pixels8 ? 1 : 4) : nullptr;
    a.pixels8 = pixels8 ? 1u : 0u;
    a.n = static_cast<uint32_t>(std::min(kChunk, n - off));
    a.op = trace_operation;
    if (cam) std::memcpy(a.cam, cam, sizeof a.cam);
    int rc = Enqueue(s, a, rt::kModeTrace, false, static_cast<hipStream_t>(stream));
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
}

int rt_scene_count_fetches(const rt_scene *s, const rt_ray_isect *rays_host, size_t n, size_t stride,
                           uint64_t counts[6], uint32_t *per_ray)
{
  if (!s || !rays_host || !counts) return Fail(RT_ERR_INVALID, "null argument");
  if (stride == 0) stride = 1;
  rt::FetchCounts fc = rt::CountFetches(s->flat, s->width, s->quant != 0, rays_host, n, stride, per_ray);
  counts[0] = fc.nodes; counts[1] = fc.tris; counts[2] = fc.leaves; counts[3] = fc.rays_hit;
  counts[4] = (n + stride - 1) / stride;
  counts[5] = fc.max_stack;
  return RT_OK;
}

int rt_scene_wave_model(const rt_scene *s, const rt_ray_isect *rays_host, size_t n, unsigned image_width, unsigned refill,
                        size_t max_waves, int policy, uint64_t out[16])
{
  if (!s || !rays_host || !out) return Fail(RT_ERR_INVALID, "null argument");
  if (image_width && n % image_width) return Fail(RT_ERR_INVALID, "ray count is not a multiple of image_width");
  rt::WaveModel m = rt::SimulateWaves(s->flat, s->model_width == 8 ? 8 : s->width, s->quant != 0, rays_host, n, image_width, refill, max_waves, policy);
  out[0] = m.waves; out[1] = m.rays; out[2] = m.wave_node_steps; out[3] = m.lane_node_steps; out[4] = m.wave_tri_steps;
  out[5] = m.lane_tri_steps; out[6] = m.wave_leaf_phases; out[7] = m.lane_leaf_phases; out[8] = m.refills; out[9] = m.refill_lanes; out[10] = m.max_wave_steps; out[11] = m.uniform_node_steps;
  out[12] = m.iterations; out[13] = m.max_wave_iterations; out[14] = m.two_node_steps; out[15] = m.three_node_steps;
  return RT_OK;
}

int rt_scene_count_fetches_device(const rt_scene *s, const void *rays_dev, size_t n, uint64_t counts[6])
{
  if (!s || !rays_dev || !counts) return Fail(RT_ERR_INVALID, "null argument");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (n == 0 || n > kChunk) return Fail(RT_ERR_INVALID, "ray count out of range for the instrumented kernel");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  RT_HIP(hipMemset(s->d_stats, 0, kStatShards * 16 * sizeof(unsigned long long)));
  int rc = EnsureStage(&s->d_stage_out, &s->stage_out_bytes, n * sizeof(rt_hit16));
  if (rc != RT_OK) return rc;
  rt::KernelArgs a;
  FillArgs(s, a);
  a.rays = static_cast<const rt_ray_isect *>(rays_dev);
  a.hits16 = static_cast<rt_hit16 *>(s->d_stage_out);
  a.n = static_cast<uint32_t>(n);
  rc = Enqueue(s, a, rt::kModeIntersect, true, nullptr);
  if (rc != RT_OK) return rc;
  unsigned long long h[kStatShards * 16];
  RT_HIP(hipMemcpy(h, s->d_stats, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; ++i) {
    counts[i] = 0;
    for (int sh = 0; sh < kStatShards; ++sh) counts[i] += h[sh * 16 + i];
  }
  counts[4] = n;
  counts[5] = 0;
  for (int sh = 0; sh < kStatShards; ++sh) counts[5] = std::max<uint64_t>(counts[5], h[sh * 16 + 4]);
  return RT_OK;
}

int rt_scene_wave_timeline(const rt_scene *s, const void *rays_dev, size_t n, uint64_t *records, size_t capacity)
{
  if (!s || !rays_dev || !records) return Fail(RT_ERR_INVALID, "null argument");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (n == 0 || n > kChunk) return Fail(RT_ERR_INVALID, "ray count out of range for the instrumented kernel");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  int rc = EnsureStage(&s->d_stage_out, &s->stage_out_bytes, n * sizeof(rt_hit16));
  if (rc != RT_OK) return rc;
  rt::KernelArgs a;
  FillArgs(s, a);
  a.rays = static_cast<const rt_ray_isect *>(rays_dev);
  a.hits16 = static_cast<rt_hit16 *>(s->d_stage_out);
  a.n = static_cast<uint32_t>(n);
  // task count depends on the image hint; compute it the way Enqueue does
  size_t ntasks = (n + 63) / 64;
  if (s->image_width > 0 && n % static_cast<size_t>(s->image_width) == 0)
    ntasks = ((static_cast<size_t>(s->image_width) + 7) / 8) * ((n / s->image_width + 7) / 8);
  ntasks *= 4;  // room for the slots of a launch that traces every unit as four quarter waves
  if (capacity < ntasks * 4) return Fail(RT_ERR_INVALID, "timeline buffer too small");
  unsigned long long *d_tl = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_tl), ntasks * 4 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d_tl, 0, ntasks * 4 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(s->d_stats, 0, kStatShards * 16 * sizeof(unsigned long long));
  a.timeline = d_tl;
  if (e == hipSuccess) rc = Enqueue(s, a, rt::kModeIntersect, true, nullptr);
  if (e == hipSuccess && rc == RT_OK) e = hipMemcpy(records, d_tl, ntasks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d_tl);
  if (e != hipSuccess) return FailHip(e, "wave timeline");
  return rc;
}

// One launch that records every work unit's traversal cost (exact, no neighbourhood spreading) and returns the table.
static int UnitCosts(const rt_scene *s, rt::KernelArgs &a, int mode, uint32_t *costs_host, size_t capacity, size_t *nunits)
{
  SetShape(s, a, mode);
  if (nunits) *nunits = a.ntasks;
  if (!costs_host) return RT_OK;   // size query
  if (capacity < a.ntasks) return Fail(RT_ERR_INVALID, "cost buffer too small: " + std::to_string(a.ntasks) + " work units");
  int rc = CheckStack(s, a, 64u);
  if (rc != RT_OK) return rc;
  uint32_t *d_cost = nullptr;
  RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_cost), static_cast<size_t>(a.ntasks) * sizeof(uint32_t)));
  hipError_t e = hipMemset(d_cost, 0, static_cast<size_t>(a.ntasks) * sizeof(uint32_t));
  rc = EnsureStage(&s->d_stage_out, &s->stage_out_bytes, static_cast<size_t>(a.n) * sizeof(rt_hit16));
  if (e == hipSuccess && rc == RT_OK) {
    a.hits16 = static_cast<rt_hit16 *>(s->d_stage_out);
    a.unit_cost = d_cost;
    a.cost_radius = 0;
    a.cost_share = 0;
    e = rt::LaunchClosestHit(a, s->width, s->quant != 0, mode, false, a.ntasks, 64u, nullptr);
    if (e == hipSuccess) e = hipMemcpy(costs_host, d_cost, static_cast<size_t>(a.ntasks) * sizeof(uint32_t), hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_cost);
  if (rc != RT_OK) return rc;
  if (e != hipSuccess) return FailHip(e, "unit cost launch");
  return RT_OK;
}

int rt_scene_unit_costs(const rt_scene *s, const void *rays_dev, size_t n, uint32_t *costs_host, size_t capacity, size_t *nunits)
{
  if (!s || !rays_dev) return Fail(RT_ERR_INVALID, "null argument");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (n == 0 || n > kChunk) return Fail(RT_ERR_INVALID, "ray count out of range for one launch");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  rt::KernelArgs a;
  FillArgs(s, a);
  a.rays = static_cast<const rt_ray_isect *>(rays_dev);
  a.n = static_cast<uint32_t>(n);
  return UnitCosts(s, a, rt::kModeIntersect, costs_host, capacity, nunits);
}

int rt_scene_unit_costs_primary(const rt_scene *s, int width, int height, const float origin[3], uint32_t *costs_host, size_t capacity,
                                size_t *nunits)
{
  if (!s || !origin) return Fail(RT_ERR_INVALID, "null argument");
  if (s->device == RT_DEVICE_NONE) return Fail(RT_ERR_DEVICE, "scene was created without a device");
  if (width <= 0 || height <= 0 || static_cast<uint64_t>(width) * static_cast<uint64_t>(height) >= kFeedbackMaxRays)
    return Fail(RT_ERR_INVALID, "bad image size");
  DeviceGuard guard(s->device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  rt::KernelArgs a;
  FillArgs(s, a);
  a.n = static_cast<uint32_t>(width) * static_cast<uint32_t>(height);
  a.img_w = static_cast<uint32_t>(width);
  a.cam[0] = origin[0]; a.cam[1] = origin[1]; a.cam[2] = origin[2];
  return UnitCosts(s, a, rt::kModePrimary, costs_host, capacity, nunits);
}

int rt_scatter_tiles(int device, const void *src_dev, size_t src_records, const uint32_t *tile_ids_dev, const uint32_t *tile_first_dev,
                     size_t ntiles, int tile, int width, int height, size_t record_bytes, void *dst_dev, void *stream)
{
  if (!src_dev || !tile_ids_dev || !tile_first_dev || !dst_dev) return Fail(RT_ERR_INVALID, "null argument");
  if (tile <= 0 || width <= 0 || height <= 0) return Fail(RT_ERR_INVALID, "bad tile or image size");
  if (record_bytes == 0 || record_bytes % 4u != 0 || record_bytes > 1024u) return Fail(RT_ERR_INVALID, "record size must be a multiple of 4 bytes, at most 1024");
  if (ntiles == 0) return RT_OK;
  const uint64_t tiles_total = static_cast<uint64_t>((width + tile - 1) / tile) * static_cast<uint64_t>((height + tile - 1) / tile);
  if (ntiles > tiles_total) return Fail(RT_ERR_INVALID, "more tiles listed than the image has");
  if (src_records > 0xFFFFFFFFull) return Fail(RT_ERR_INVALID, "source holds more records than one launch addresses");
  DeviceGuard guard(device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  RT_HIP(rt::LaunchScatterTiles(src_dev, static_cast<uint32_t>(src_records), tile_ids_dev, tile_first_dev, static_cast<uint32_t>(ntiles), static_cast<uint32_t>(tile),
                                static_cast<uint32_t>(width), static_cast<uint32_t>(height), static_cast<uint32_t>(record_bytes), dst_dev,
                                static_cast<hipStream_t>(stream)));
  return RT_OK;
}

int rt_pack_bgra8(int device, const float *rgba_dev, size_t n, void *bgra_dev, void *stream)
{
  if (!rgba_dev || !bgra_dev) return Fail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  if (n > 0xFFFFFFFFull) return Fail(RT_ERR_INVALID, "more pixels than one launch addresses");
  DeviceGuard guard(device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  RT_HIP(rt::LaunchPackPixels(rgba_dev, static_cast<uint32_t>(n), bgra_dev, static_cast<hipStream_t>(stream)));
  return RT_OK;
}

int rt_debug_build_order(int device, const uint32_t *unit_costs, size_t nunits, uint32_t *order_out, uint32_t *costs_after)
{
  if (!unit_costs || !order_out || nunits == 0 || nunits > (size_t(1) << 24)) return Fail(RT_ERR_INVALID, "bad order arguments");
  DeviceGuard guard(device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  const uint32_t n = static_cast<uint32_t>(nunits);
  const size_t padded = rt::OrderPaddedUnits(n);
  uint32_t *d_cost = nullptr, *d_bins = nullptr, *d_order = nullptr;
  int rc = Upload(&d_cost, nullptr, padded * sizeof(uint32_t));
  if (rc == RT_OK) rc = Upload(&d_bins, nullptr, rt::OrderBinWords(n) * sizeof(uint32_t));
  if (rc == RT_OK) rc = Upload(&d_order, nullptr, nunits * sizeof(uint32_t));
  if (rc == RT_OK) {
    hipError_t e = hipMemcpy(d_cost, unit_costs, nunits * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = rt::LaunchBuildOrder(d_cost, n, d_bins, d_order, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(order_out, d_order, nunits * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && costs_after) e = hipMemcpy(costs_after, d_cost, nunits * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = FailHip(e, "order build");
  }
  (void)hipFree(d_cost); (void)hipFree(d_bins); (void)hipFree(d_order);
  return rc;
}

int rt_debug_tri_batch(int device, const rt_vec3 *tri, const rt_ray *ray, size_t n, int32_t *hit, float *t, rt_vec2 *uv)
{
  if (!tri || !ray || !hit || !t || !uv) return Fail(RT_ERR_INVALID, "null argument");
  if (n == 0) return RT_OK;
  if (n > (size_t(1) << 28)) return Fail(RT_ERR_INVALID, "batch too large");
  DeviceGuard guard(device);
  if (!guard.ok) return Fail(RT_ERR_DEVICE, "hipSetDevice failed");
  rt_vec3 *d_tri = nullptr; rt_ray *d_ray = nullptr; int32_t *d_hit = nullptr; float *d_t = nullptr; rt_vec2 *d_uv = nullptr;
  int rc = Upload(&d_tri, tri, n * 3 * sizeof(rt_vec3));
  if (rc == RT_OK) rc = Upload(&d_ray, ray, n * sizeof(rt_ray));
  if (rc == RT_OK) rc = Upload(&d_hit, nullptr, n * sizeof(int32_t));
  if (rc == RT_OK) rc = Upload(&d_t, nullptr, n * sizeof(float));
  if (rc == RT_OK) rc = Upload(&d_uv, nullptr, n * sizeof(rt_vec2));
  if (rc == RT_OK) {
    hipError_t e = rt::LaunchTriBatch(d_tri, d_ray, static_cast<uint32_t>(n), d_hit, d_t, d_uv, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hit, d_hit, n * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(t, d_t, n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(uv, d_uv, n * sizeof(rt_vec2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = FailHip(e, "tri batch");
  }
  (void)hipFree(d_tri); (void)hipFree(d_ray); (void)hipFree(d_hit); (void)hipFree(d_t); (void)hipFree(d_uv);
  return rc;
}

// ---------------------------------------------------------------- framebuffer out
int rt_write_bmp32(const char *path, const uint8_t *bgra, int width, int height)
{
  if (!path || !bgra || width <= 0 || height <= 0) return Fail(RT_ERR_INVALID, "bad image arguments");
  if (!rt::WriteBmp32(path, bgra, width, height)) return Fail(RT_ERR_IO, std::string("cannot write ") + path);
  return RT_OK;
}

int rt_write_pfm(const char *path, const float *rgba, int width, int height)
{
  if (!path || !rgba || width <= 0 || height <= 0) return Fail(RT_ERR_INVALID, "bad image arguments");
  if (!rt::WritePfm(path, rgba, width, height)) return Fail(RT_ERR_IO, std::string("cannot write ") + path);
  return RT_OK;
}

}  // extern "C"
