// trace_kernels.hip -- gfx950 closest-hit kernels (wave64, FP32 VALU, no MFMA: there is
// no dense contraction on this path).
//
// What the kernel computes per ray is what one work-item of the reference's `trace`
// kernel computes (render_client/tracekernel.ck:192-414): optional ray setup (:224-265),
// closest hit over the BVH (:271-370), output record and pixel (:373-413).  How it gets
// there is different:
//   * nodes hold their children's boxes (64 B two-wide or 128 B four-wide records), so
//     one coalescable record fetch replaces the reference's two or three dependent
//     80-byte node reads per step (near_child_impl/parent_impl/sibling_impl, :115-138);
//   * an explicit per-lane stack in LDS, laid out [entry][lane] so a wave's pushes and
//     pops never bank-conflict, replaces the parent-link state machine (:291-370);
//   * leaves are runs of 48-byte pre-gathered triangle records instead of index ->
//     3 x vertex gathers (:159-162);
//   * every wave owns one work unit (an 8x8 pixel tile, or 64 consecutive records) picked through a
//     dispatch-order table, so the order in which units start can follow their measured cost;
//   * reciprocal directions are computed once per ray, not once per box (:73);
//   * the closest hit is order-independent: on equal t the lower triangle index wins,
//     which is what an index-order brute-force loop with a strict `<` returns
//     (cpu_raycast/raycast.cpp:113-122).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>

#include "../../include/rt_layout.h"
#include "device_scene.h"
#include "trace_math.h"

namespace rt {

namespace {

constexpr uint32_t kDone = 0xFFFFFFFFu;

__device__ __forceinline__ float4 LoadF4(const float4 *p) { return *p; }

// Ray records are read once and hit records written once per launch, while node and triangle records are read
// over and over: the streams are marked non-temporal so that they do not push the scene out of the 4 MB L2 of
// an XCD (scenes of a few MB sit right at that size).  Not in the full trace step: there a pass reads the records the
// pass before it wrote (the bounce loop ping-pongs two buffers), and a 1080p frame of them fits the 256 MB
// Infinity Cache if it is allowed to stay there.
#ifndef RT_STREAM_NT
#define RT_STREAM_NT 1
#endif
typedef float NativeF4 __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ float4 LoadStreamF4(const float4 *p)
{
  if (RT_STREAM_NT && STREAM) {
    const NativeF4 v = __builtin_nontemporal_load(reinterpret_cast<const NativeF4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
template <bool STREAM>
__device__ __forceinline__ void StoreStreamF4(float4 *p, const float4 v)
{
  if (RT_STREAM_NT && STREAM) {
    const NativeF4 n = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(n, reinterpret_cast<NativeF4 *>(p));
    return;
  }
  *p = v;
}

// The compact hit of record idx: {id, t, u, v}, or only its first two words or its first word ("compact" argument).
template <bool STREAM>
__device__ __forceinline__ void StoreCompactHit(const KernelArgs &p, uint32_t idx, uint32_t id, float t, float u, float v)
{
  if (p.compact_words == 2u) {
    typedef uint32_t NativeU2 __attribute__((ext_vector_type(2)));
    const NativeU2 w = {id, __float_as_uint(t)};
    NativeU2 *dst = reinterpret_cast<NativeU2 *>(p.hits16) + idx;
    if (RT_STREAM_NT && STREAM) __builtin_nontemporal_store(w, dst);
    else *dst = w;
  } else if (p.compact_words == 1u) {
    uint32_t *dst = reinterpret_cast<uint32_t *>(p.hits16) + idx;
    if (RT_STREAM_NT && STREAM) __builtin_nontemporal_store(id, dst);
    else *dst = id;
  } else {
    StoreStreamF4<STREAM>(reinterpret_cast<float4 *>(p.hits16 + idx), make_float4(__uint_as_float(id), t, u, v));
  }
}

// 8-bit BGRA of an RGBA32F pixel: each channel min(255, max(0, x * 255 + 0.5)) truncated, alpha 255 (the byte order a
// 32-bit bitmap stores and the reference's writer passes through, cpu_raycast/bmp.h:12, bmp.c:49-53).
__device__ __forceinline__ uint32_t PackBgra8(const float4 c)
{
  const uint32_t r = static_cast<uint32_t>(fminf(255.f, fmaxf(0.f, c.x * 255.f + 0.5f)));
  const uint32_t g = static_cast<uint32_t>(fminf(255.f, fmaxf(0.f, c.y * 255.f + 0.5f)));
  const uint32_t b = static_cast<uint32_t>(fminf(255.f, fmaxf(0.f, c.z * 255.f + 0.5f)));
  return b | (g << 8) | (r << 16) | 0xFF000000u;
}

struct Best {
  float t, u, v;
  uint32_t id;
};


// The records of a leaf are fetched two at a time, before either is tested, so a leaf of up to two triangles exposes
// one memory latency and a full leaf two, instead of one per triangle (a leaf's records are consecutive).
template <bool COUNT>
__device__ __forceinline__ void LeafTest(const KernelArgs &p, const RayCtx &r, uint32_t ref, Best &b, uint32_t &ntri)
{
  const uint32_t first = (ref & 0x7FFFFFFFu) >> 3, cnt = ref & 7u;
  const float4 *tp = p.tris + static_cast<size_t>(first) * 3u;
  auto test = [&](const float4 &a, const float4 &e1, const float4 &e2) {
    float t, u, v;
    const bool ok = TriTest(a.x, a.y, a.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, r, t, u, v);
    const uint32_t id = __float_as_uint(a.w);
    if (ok && (t < b.t || (t == b.t && id < b.id))) {
      b.t = t; b.u = u; b.v = v; b.id = id;
    }
    if (COUNT) ntri++;
  };
  for (uint32_t j = 0; j < cnt; j += 2u, tp += 6) {
    const float4 a0 = LoadF4(tp), b0 = LoadF4(tp + 1), c0 = LoadF4(tp + 2);
    float4 a1, b1, c1;
    const bool two = j + 1u < cnt;
    if (two) { a1 = LoadF4(tp + 3); b1 = LoadF4(tp + 4); c1 = LoadF4(tp + 5); }
    test(a0, b0, c0);
    if (two) test(a1, b1, c1);
  }
}

// ---- per-lane traversal stack -----------------------------------------------------------------------------
// Entries are {child reference, entry distance}.  The first `lds_n` entries of a lane live in LDS, laid out
// [entry][reference | distance][lane] (each half is 64 consecutive dwords, so neither a push nor a pop ever
// bank-conflicts, and both halves go out in one two-address LDS instruction from two unrelated registers); deeper
// entries spill to the lane's private memory.  Primary rays on the benchmark scenes never go deeper than
// 11 entries (four-wide) or 9 (two-wide), so the LDS part is short -- what lets many waves share a CU's
// LDS -- and the spill path only has to be correct, not fast.  The host guarantees
// lds_n + kDeepEntries >= the worst case of the tree (3 * depth + 1 for four-wide nodes).
// The LDS column is addressed through an LDS-qualified pointer and the private part by plain array indexing, so no
// access ever goes through a generic (flat) pointer that would have to be told apart at run time.
// The stack pointer IS the LDS address of the lane's next free entry (its column in row 0 plus 512 bytes per entry held), so a
// push or a pop costs no address arithmetic and the column's base need not stay in a register through the traversal loop:
// whether the stack is empty, and whether an entry lives in LDS or in the private part, are comparisons with two wave-uniform
// addresses (a lane's column offset, 4 * lane, is smaller than one entry's 512 bytes).
typedef __attribute__((address_space(3))) uint32_t LdsWord;
constexpr uint32_t kEntryBytes = 512u;   // one entry of all 64 lanes: 64 references + 64 distances
struct Stack {
  uint32_t top;      // LDS byte address of this lane's next free entry
  uint32_t first;    // wave-uniform: LDS byte address of the wave's part
  uint32_t limit;    // wave-uniform: first + kEntryBytes * (entries per lane held in LDS)
};

__device__ __forceinline__ Stack MakeStack(uint2 *wave_lds, uint32_t lane, uint32_t lds_entries)
{
  const uint32_t first = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsWord *)reinterpret_cast<uint32_t *>(wave_lds)));
  return Stack{first + 4u * lane, first, first + kEntryBytes * lds_entries};
}
__device__ __forceinline__ uint32_t StackDepth(const Stack &st) { return (st.top - st.first) / kEntryBytes; }

__device__ __forceinline__ void Push(Stack &st, uint2 *deep, uint32_t ref, float t)
{
  if (st.top < st.limit) {
    LdsWord *slot = reinterpret_cast<LdsWord *>(st.top);
    slot[0] = ref; slot[64] = __float_as_uint(t);
  } else {
    deep[(st.top - st.limit) / kEntryBytes] = make_uint2(ref, __float_as_uint(t));
  }
  st.top += kEntryBytes;
}

// Next reference worth visiting: entries whose box starts beyond the best hit found since they were pushed are
// dropped without fetching their node (equal distances are kept: exact ties resolve by triangle index).
__device__ __forceinline__ uint32_t Pop(Stack &st, const uint2 *deep, float tbest)
{
  while (st.top >= st.first + kEntryBytes) {
    st.top -= kEntryBytes;
    uint32_t ref, tbits;
    if (st.top < st.limit) {
      const LdsWord *slot = reinterpret_cast<const LdsWord *>(st.top);
      ref = slot[0]; tbits = slot[64];
    } else {
      const uint2 e = deep[(st.top - st.limit) / kEntryBytes];
      ref = e.x; tbits = e.y;
    }
    if (__uint_as_float(tbits) <= tbest) return ref;
  }
  return kDone;
}

// Two children: continue with the nearer hit, push the farther.
__device__ __forceinline__ uint32_t Descend2(Stack &st, uint2 *deep, bool h0, bool h1, float t0, float t1, uint32_t r0,
                                             uint32_t r1, float tbest)
{
  if (h0 && h1) {
    const bool swap = t1 < t0;
    Push(st, deep, swap ? r0 : r1, swap ? t0 : t1);
    return swap ? r1 : r0;
  }
  if (h0) return r0;
  if (h1) return r1;
  return Pop(st, deep, tbest);
}

// Four children: continue with the nearest hit; the other hits are pushed in slot order, highest slot deepest.
// A full sort by entry distance buys 2 % fewer node fetches than this (the entry distance kept with each pushed
// reference culls most of what a wrong order would cost) and takes three times the instructions.
__device__ __forceinline__ uint32_t Descend4(Stack &st, uint2 *deep, bool h0, bool h1, bool h2, bool h3, float t0,
                                             float t1, float t2, float t3, uint32_t r0, uint32_t r1, uint32_t r2,
                                             uint32_t r3, float tbest)
{
  const float k0 = h0 ? t0 : kFltMax, k1 = h1 ? t1 : kFltMax, k2 = h2 ? t2 : kFltMax, k3 = h3 ? t3 : kFltMax;
  const bool s01 = k1 < k0, s23 = k3 < k2;
  const float k01 = s01 ? k1 : k0, k23 = s23 ? k3 : k2;
  const uint32_t r01 = s01 ? r1 : r0, r23 = s23 ? r3 : r2;
  const bool s = k23 < k01;
  const uint32_t next = s ? r23 : r01;
  if (!(h0 || h1 || h2 || h3)) return Pop(st, deep, tbest);
  // references are distinct within a node, so comparing against `next` identifies the slot taken
  if (h3 && r3 != next) Push(st, deep, r3, t3);
  if (h2 && r2 != next) Push(st, deep, r2, t2);
  if (h1 && r1 != next) Push(st, deep, r1, t1);
  if (h0 && r0 != next) Push(st, deep, r0, t0);
  return next;
}

// ---- cooperative quarter waves: four lanes per ray ---------------------------------------------------------------
// A launch ends with its longest wave, and the longest waves are those of the few tiles whose rays graze the model: a
// single ray of them fetches ~90 records one after the other, whatever its 63 neighbours do.  The SPLIT kernels already give
// such a tile to four waves of 16 rays; those waves leave 48 lanes idle.  Here the idle lanes work: lane 4r + c holds ray r
// (of 16) and takes CHILD c of the ray's current four-wide node (one 16-byte load and one slab test per lane instead of four)
// or TRIANGLE c of its leaf (one record and one triangle test instead of up to four); the quad then agrees on the nearest
// child / the closest hit with quad-permute DPP operations, which cost one vector instruction each.  The four lanes of a
// quad hold identical copies of the ray, the best hit, the current reference and the stack pointer at every step, so
// control flow diverges between quads only and the DPP operations never read a disabled lane.
// The stack of ray r lives in the LDS columns of its four lanes (entry e in row e / 4, column 4r + e % 4): four times the
// depth of a lane's LDS part, and nothing spills to private memory (the host launches this mode only if the tree fits).
template <int CTRL>
__device__ __forceinline__ uint32_t QuadMov(uint32_t x)
{
  return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(x), CTRL, 0xF, 0xF, true));
}
constexpr int kQuadSwap1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int kQuadSwap2 = 0x4E;   // quad_perm:[2,3,0,1]
__device__ __forceinline__ uint32_t QuadMinU32(uint32_t x)
{
  x = min(x, QuadMov<kQuadSwap1>(x));
  return min(x, QuadMov<kQuadSwap2>(x));
}
__device__ __forceinline__ uint32_t QuadOr(uint32_t x)
{
  x |= QuadMov<kQuadSwap1>(x);
  return x | QuadMov<kQuadSwap2>(x);
}

constexpr uint32_t kInfBits = 0x7F800000u;

// Entry e of this lane's ray: row e / 4 of the [entry][reference | distance][lane] layout, column (quad base) + e % 4.
__device__ __forceinline__ LdsWord *CoopSlot(LdsWord *quad_col, uint32_t e) { return quad_col + (e >> 2) * 128u + (e & 3u); }

__device__ __forceinline__ uint32_t CoopPop(LdsWord *quad_col, uint32_t &sp, float tbest)
{
  while (sp) {
    sp--;
    const LdsWord *slot = CoopSlot(quad_col, sp);
    const uint32_t ref = slot[0], tbits = slot[64];
    if (__uint_as_float(tbits) <= tbest) return ref;
  }
  return kDone;
}

// Traverses the rays of a quarter unit, four lanes per ray.  `cidx` = lane & 3; every lane of a quad enters with the same ray.
// Returns the wave's iteration count (all lanes the same value).
template <bool COUNT>
__device__ __forceinline__ uint32_t CoopTraverse(const KernelArgs &p, const RayCtx &r, const RayCtxQ &rq, LdsWord *quad_col, uint32_t cidx,
                                                 bool live, Best &best, uint32_t &nnode, uint32_t &ntri, uint32_t &nleaf, uint32_t &maxsp)
{
  uint32_t cur = live ? 0u : kDone, sp = 0u, iters = 0u;
  for (;;) {
    iters++;
    const bool inner = !(cur & kLeafBitDev);
    const bool parked = !inner && cur != kDone;
    const unsigned long long mi = __ballot(inner), ml = __ballot(parked);
    if ((mi | ml) == 0ull) break;
    const uint32_t ni = __popcll(mi), nl = __popcll(ml);
    if (ni != 0u && nl * 100u < p.leaf_pct * (ni + nl)) {
      if (inner) {
        if (COUNT && cidx == 0u) nnode++;
        const uint4 c = reinterpret_cast<const uint4 *>(p.nodes)[static_cast<size_t>(cur) * 4u + cidx];
        float t;
        const bool h = SlabTestQ(rq, c.x, c.y, c.z, best.t, t);
        // nearest hit child of the quad: entry distances are >= 0, so their bit patterns order like the floats; the two lowest
        // mantissa bits are replaced by the slot number, which makes the keys of a quad distinct (lowest slot wins a tie, as in
        // Descend4; a 3-ulp change of the ordering key changes no result, and the distance pushed below is the exact one)
        const uint32_t key = h ? ((__float_as_uint(t) & ~3u) | cidx) : 0xFFFFFFFFu;
        const uint32_t kmin = QuadMinU32(key);
        const bool take = key == kmin && h;
        const uint32_t next = QuadOr(take ? c.w : 0u);
        // the other hit children go on the ray's stack, each lane writing its own entry at its rank among the quad's pushes
        const bool push = h && !take;
        const unsigned long long pm = __ballot(push);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0u));
        const uint32_t base = QuadMov<0x00>(below);                               // pushes in lanes below the quad
        const uint32_t total = QuadMov<0xFF>(below + (push ? 1u : 0u)) - base;    // pushes of the quad
        if (push) {
          LdsWord *slot = CoopSlot(quad_col, sp + (below - base));
          slot[0] = c.w; slot[64] = __float_as_uint(t);
        }
        sp += total;
        if (COUNT) maxsp = max(maxsp, sp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // a pop below may read what another lane of the quad just wrote
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cur = kmin != 0xFFFFFFFFu ? next : CoopPop(quad_col, sp, best.t);
      }
    } else if (parked) {
      if (COUNT && cidx == 0u) nleaf += (cur & 7u) ? 1u : 0u;
      const uint32_t first = (cur & 0x7FFFFFFFu) >> 3, cnt = cur & 7u;
      for (uint32_t j = 0; j < cnt; j += 4u) {   // leaves hold at most four triangles: one round
        const bool has = j + cidx < cnt;
        float t = 0.f, u = 0.f, v = 0.f;
        uint32_t id = RT_MISS_ID;
        bool better = false;
        if (has) {
          const float4 *tp = p.tris + static_cast<size_t>(first + j + cidx) * 3u;
          const float4 a = LoadF4(tp), e1 = LoadF4(tp + 1), e2 = LoadF4(tp + 2);
          const bool ok = TriTest(a.x, a.y, a.z, e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, r, t, u, v);
          id = __float_as_uint(a.w);
          better = ok && (t < best.t || (t == best.t && id < best.id));
          if (COUNT) ntri++;
        }
        // accepted hits have t > 1e-6: positive floats, ordered like their bit patterns
        const uint32_t tmin = QuadMinU32(better ? __float_as_uint(t) : kInfBits);
        const bool cand = better && __float_as_uint(t) == tmin;
        const uint32_t idmin = QuadMinU32(cand ? id : RT_MISS_ID);     // equal distances: the lower triangle index
        const bool win = cand && id == idmin;
        const uint32_t wu = QuadOr(win ? __float_as_uint(u) : 0u), wv = QuadOr(win ? __float_as_uint(v) : 0u);
        if (tmin != kInfBits) {
          best.t = __uint_as_float(tmin); best.id = idmin; best.u = __uint_as_float(wu); best.v = __uint_as_float(wv);
        }
      }
      cur = CoopPop(quad_col, sp, best.t);
    }
  }
  return iters;
}

#ifndef RT_SCALAR_NODE
#define RT_SCALAR_NODE 1
#endif
// Node step of a whole wave at ONE four-wide quantized node (every stepping lane holds the same reference, as at the
// top of the tree): the 64-byte record is fetched once through the scalar cache into SGPRs instead of 64 times through
// the vector-memory path, which is the busiest unit of this kernel.
typedef uint32_t ScalarNode __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t NodeStepUniformQ4(const KernelArgs &p, const RayCtxQ &q, uint32_t node, float tbest, Stack &st,
                                                      uint2 *deep)
{
  // `node` is wave-uniform by contract; the intrinsic makes that a fact for the compiler, whatever it merged around the call
  const uint4 *np = reinterpret_cast<const uint4 *>(p.nodes) + static_cast<size_t>(__builtin_amdgcn_readfirstlane(node)) * 4u;
  ScalarNode w;
  asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(np) : "memory");
  float t0, t1, t2, t3;
  const bool h0 = SlabTestQ(q, w[0], w[1], w[2], tbest, t0);
  const bool h1 = SlabTestQ(q, w[4], w[5], w[6], tbest, t1);
  const bool h2 = SlabTestQ(q, w[8], w[9], w[10], tbest, t2);
  const bool h3 = SlabTestQ(q, w[12], w[13], w[14], tbest, t3);
  return Descend4(st, deep, h0, h1, h2, h3, t0, t1, t2, t3, w[3], w[7], w[11], w[15], tbest);
}

// One step at a node of the selected layout: returns the next reference to visit (kDone if the stack ran dry).
template <int WIDTH, bool QUANT>
__device__ __forceinline__ uint32_t NodeStep(const KernelArgs &p, const RayCtx &r, const RayCtxQ &q, uint32_t cur,
                                             float tbest, Stack &st, uint2 *deep)
{
  if (QUANT) {
    // quantized records: WIDTH children of 16 bytes each {x, y, z axis words, reference}
    const uint4 *np = reinterpret_cast<const uint4 *>(p.nodes) + static_cast<size_t>(cur) * WIDTH;
    if (WIDTH == 2) {
      const uint4 c0 = np[0], c1 = np[1];
      float t0, t1;
      const bool h0 = SlabTestQ(q, c0.x, c0.y, c0.z, tbest, t0);
      const bool h1 = SlabTestQ(q, c1.x, c1.y, c1.z, tbest, t1);
      return Descend2(st, deep, h0, h1, t0, t1, c0.w, c1.w, tbest);
    }
#if RT_EXP_DUP_LOADS
    // sensitivity experiment: the node's four 16-byte loads issued twice (the extra set first, so it returns first
    // and adds no latency of its own, only work for the vector-memory path)
    typedef uint32_t NativeU4 __attribute__((ext_vector_type(4)));
    NativeU4 d0, d1, d2, d3;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(np) : "memory");
#endif
    uint4 c0 = np[0], c1 = np[1], c2 = np[2], c3 = np[3];
    // the child references travel with the boxes in four 16-byte loads: without this the compiler splits them off
    // into four dependent 4-byte loads issued after the box tests, a second memory latency in every step
    asm volatile("" : "+v"(c0.w), "+v"(c1.w), "+v"(c2.w), "+v"(c3.w));
    float t0, t1, t2, t3;
#if RT_EXP_DUP_VALU
    {   // sensitivity experiment: the vector work of four more slab tests (24 permutes, 24 fused multiply-adds, 16 min/max)
      float x0 = tbest, x1 = q.ax, x2 = q.ay, x3 = q.az;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        asm volatile("v_perm_b32 %0, %0, %1, %2\n\tv_perm_b32 %1, %1, %2, %3\n\tv_perm_b32 %2, %2, %3, %0\n\tv_perm_b32 %3, %3, %0, %1\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1\n\t"
                     "v_max_f32 %0, %0, %1\n\tv_min3_f32 %1, %1, %2, %3\n\tv_max3_f32 %2, %2, %3, %0"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
      asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    }
#endif
    const bool h0 = SlabTestQ(q, c0.x, c0.y, c0.z, tbest, t0);
    const bool h1 = SlabTestQ(q, c1.x, c1.y, c1.z, tbest, t1);
    const bool h2 = SlabTestQ(q, c2.x, c2.y, c2.z, tbest, t2);
    const bool h3 = SlabTestQ(q, c3.x, c3.y, c3.z, tbest, t3);
#if RT_EXP_DUP_LOADS
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3) : "memory");
#endif
    return Descend4(st, deep, h0, h1, h2, h3, t0, t1, t2, t3, c0.w, c1.w, c2.w, c3.w, tbest);
  }
  if (WIDTH == 2) {
    const float4 *np = p.nodes + static_cast<size_t>(cur) * 4u;
    const float4 q0 = LoadF4(np), q1 = LoadF4(np + 1), q2 = LoadF4(np + 2);
    const uint4 q3 = *reinterpret_cast<const uint4 *>(np + 3);
    float t0, t1;
    const bool h0 = SlabTest(r, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tbest, t0);
    const bool h1 = SlabTest(r, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tbest, t1);
    return Descend2(st, deep, h0, h1, t0, t1, q3.x, q3.y, tbest);
  }
  const float4 *np = p.nodes + static_cast<size_t>(cur) * 8u;
  const float4 lox = LoadF4(np), loy = LoadF4(np + 1), loz = LoadF4(np + 2);
  const float4 hix = LoadF4(np + 3), hiy = LoadF4(np + 4), hiz = LoadF4(np + 5);
  const uint4 ch = *reinterpret_cast<const uint4 *>(np + 6);
  float t0, t1, t2, t3;
  const bool h0 = SlabTest(r, lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, tbest, t0);
  const bool h1 = SlabTest(r, lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, tbest, t1);
  const bool h2 = SlabTest(r, lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, tbest, t2);
  const bool h3 = SlabTest(r, lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, tbest, t3);
  return Descend4(st, deep, h0, h1, h2, h3, t0, t1, t2, t3, ch.x, ch.y, ch.z, ch.w, tbest);
}

__device__ __forceinline__ float Dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Where a wave's share of a work unit sits: the first record of a linear run, or the pixel origin of a square of pixels.
// A whole unit is 64 consecutive records, or an 8x8 pixel tile (the reference launches 8x8 work-groups too,
// img_main.cpp:457-461), which keeps the 64 rays of a wave on neighbouring paths through the tree.  SPLIT kernels also know
// quarter units, 16 consecutive records or a 4x4 pixel block, held by lanes 0..15 of the wave.
struct UnitPos {
  uint32_t first;    // linear: index of lane 0's record
  uint32_t x0, y0;   // image: pixel of lane 0
  uint32_t sub;      // 0 = whole unit, 1 = quarter unit (always 0 outside SPLIT kernels)
};

// Pixel of `lane`, row-major over the 8x8 or 4x4 square.
template <bool SPLIT>
__device__ __forceinline__ void PixelOfLane(const UnitPos &u, uint32_t lane, uint32_t &x, uint32_t &y)
{
  x = u.x0 + (lane & (SPLIT ? 7u >> u.sub : 7u));
  y = u.y0 + (lane >> (SPLIT ? 3u - u.sub : 3u));
}

// Ray record handled by `lane` of the wave whose share sits at `u`; false if the lane falls off the share or the ray set.
template <bool SPLIT>
__device__ __forceinline__ bool RayOfLane(const KernelArgs &p, const UnitPos &u, uint32_t lane, uint32_t &idx)
{
  const bool mine = !SPLIT || lane < (64u >> (2u * u.sub));
  if (p.img_w) {
    uint32_t x, y;
    PixelOfLane<SPLIT>(u, lane, x, y);
    idx = y * p.img_w + x;
    return mine && x < p.img_w && y < p.img_h;
  }
  idx = u.first + lane;
  return mine && idx < p.n;
}

#ifndef RT_MIN_WAVES_PER_SIMD
#define RT_MIN_WAVES_PER_SIMD 7
#endif
// Resident waves per SIMD the register allocation is held to: 7 (72 VGPRs) for the intersection kernels; the full trace
// step carries the camera / reflection setup and the pixel and needs 80 VGPRs to stay out of private memory, so it runs at 6
// (at 7 it spills inside the loop and loses a quarter of its speed).
constexpr int MinWavesPerSimd(int mode) { return mode == kModeTrace ? (RT_MIN_WAVES_PER_SIMD < 6 ? RT_MIN_WAVES_PER_SIMD : 6) : RT_MIN_WAVES_PER_SIMD; }

// SPLIT: the launch traces the first p.split_units units of its order with four waves each (a quarter of the unit per wave).
template <int WIDTH, bool QUANT, int MODE, bool COUNT, bool SPLIT>
__global__ void __launch_bounds__(256, MinWavesPerSimd(MODE)) ClosestHitKernel(KernelArgs p)
{
  extern __shared__ uint2 lds_stack[];
  // the wave's index is the same in all of its lanes: saying so keeps what depends on it (the dispatch slot, the order lookup,
  // the unit's position) in scalar registers and scalar loads, off the vector memory path the rays are about to need
  // (the LDS stack's base stays a per-lane vector value, as every access to it wants it)
  const uint32_t lane = threadIdx.x & 63u, wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint2 *const wave_lds = lds_stack + wave_s * p.stack_entries * 64u;   // stack_entries >= 8: 4 KB for the record exchange
  uint2 deep[kDeepEntries];   // private memory; touched only by rays whose stack outgrows the LDS part

  // A wave handles one work unit per round.  Static launch: one round, the unit's position in the dispatch order
  // is the wave's index in the grid.  Persistent launch (p.cursor != null): a fixed number of waves loop, each
  // round taking the next position from a global cursor with one atomic per wave, until the cursor runs past
  // the last unit -- an exit every wave reaches, whatever the order table holds.
  for (;;) {
    uint32_t slot;
    if (p.cursor) {
      uint32_t got = 0;
      if (lane == 0) got = atomicAdd(p.cursor, 1u);
      slot = __builtin_amdgcn_readfirstlane(got);
    } else {
      slot = blockIdx.x * p.waves_per_block + wave_s;
    }
    if (slot >= p.ntasks + (SPLIT ? 3u * p.split_units : 0u)) return;
    unsigned long long t_start = 0;
    if (COUNT && p.timeline) t_start = __builtin_amdgcn_s_memrealtime();
    // Which work unit this wave takes: the slot-th entry of the dispatch order.  The order comes from an earlier
    // launch's measured costs (costliest first, so the few long-running waves start first instead of being the
    // last ones alive) or, failing that, runs from the image centre out.
    // The first split_units entries of the order are taken by four consecutive slots each, a quarter of the unit per wave.
    const bool quarter = SPLIT && slot < 4u * p.split_units;
    const uint32_t entry = !SPLIT ? slot : (quarter ? slot >> 2 : slot - 3u * p.split_units);
    uint32_t unit = entry;
    if (p.order) {
      // one scalar load (an earlier kernel wrote the table): a vector load here would put a vector-memory round trip, the
      // slowest kind on a busy CU, in front of the wave's ray loads
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(unit) : "s"(p.order + entry) : "memory");
    }
    if (unit >= p.ntasks) unit = entry;            // a stale order table may cost time, never an out-of-range access
    // The launch ends with its longest-lived wave.  Under a costliest-first order those are the first slots: they
    // get the SIMD's issue priority, so that their long dependent chains are not stretched by waiting for the
    // instruction slots of the short waves that come and go beside them.
    if (slot < p.prio_slots) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
    UnitPos pos{unit * 64u + (quarter ? (slot & 3u) * 16u : 0u), 0u, 0u, quarter ? 1u : 0u};
    if (p.img_w) {   // the tile's origin, or that of its 4x4 block (slot & 1, slot >> 1 & 1)
      uint32_t ty = unit;   // unit / tiles_x without a divide: the multiplier's quotient is exact or one too large
      if (p.tiles_x != 1u) {
        ty = __umulhi(unit, p.tiles_x_magic);
        if (ty * p.tiles_x > unit) ty--;
      }
      const uint32_t tiles_x = p.tiles_x;
      pos.x0 = (unit - ty * tiles_x) * 8u + (quarter ? (slot & 1u) * 4u : 0u);
      pos.y0 = ty * 8u + (quarter ? (slot & 2u) * 2u : 0u);
    }
    // cooperative quarter wave (four-wide quantized records only): lane 4r + c works for ray r of the quarter's 16
    const bool coop = SPLIT && QUANT && WIDTH == 4 && quarter && p.coop != 0u;
    const uint32_t rlane = coop ? lane >> 2 : lane;
    uint32_t idx;
    const bool valid = RayOfLane<SPLIT>(p, pos, rlane, idx);
    float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0, rec2 = rec0, rec3 = rec0;   // this lane's output record
    uint32_t wave_cost = 0;
    uint32_t c_node = 0, c_tri = 0, c_leaf = 0, c_hit = 0, c_deep = 0;  // instrumented build: this lane's counts
    if (valid) {
      const float4 *rp = reinterpret_cast<const float4 *>(p.rays + idx);
      float4 in_o, in_d;
      if (MODE == kModePrimary) {
        // the reference's camera table, computed instead of read (render_client/img_main.cpp:253-267,
        // core/raymath.cpp:6-15): eye = (W/2, H/2, -5W), direction = normalize((i, j, 0) - eye), the same float
        // operations in the same order as the host generator, so the rays are bit-identical to its table
        const float ex = p.img_w / 2.f, ey = p.img_h / 2.f, ez = static_cast<float>(-5 * static_cast<int>(p.img_w));
        uint32_t px, py;
        PixelOfLane<SPLIT>(pos, rlane, px, py);
        const float tx = static_cast<float>(px) - ex, ty = static_cast<float>(py) - ey, tz = 0.f - ez;
        const float len = sqrtf(tx * tx + ty * ty + tz * tz);
        in_o = make_float4(p.cam[0], p.cam[1], p.cam[2], 1.f);
        in_d = make_float4(tx / len, ty / len, tz / len, 0.f);
      } else {
        in_o = LoadStreamF4<MODE != kModeTrace>(rp);
        in_d = LoadStreamF4<MODE != kModeTrace>(rp + 1);
      }
      float ox = in_o.x, oy = in_o.y, oz = in_o.z, dx = in_d.x, dy = in_d.y, dz = in_d.z;
      bool live = true;
      if (MODE == kModeTrace) {
        if (p.op == 1u) {
          // reflect about the stored normal and step off the surface (tracekernel.ck:101-104,227-233)
          const float4 nn = LoadStreamF4<MODE != kModeTrace>(rp + 2);
          const float s = 2.0f * Dot4(in_d, nn);
          dx = in_d.x - s * nn.x; dy = in_d.y - s * nn.y; dz = in_d.z - s * nn.z;
          ox = in_o.x + nn.x * 0.00001f; oy = in_o.y + nn.y * 0.00001f; oz = in_o.z + nn.z * 0.00001f;
          // a ray that missed in the previous pass stays terminated
          live = p.rays[idx].triangle_id != RT_MISS_ID;
        } else {
          // camera matrix rows dotted with (origin,1) and (direction,0) (tracekernel.ck:182-190,251-265)
          const float4 o4 = make_float4(ox, oy, oz, 1.f), d4 = make_float4(dx, dy, dz, 0.f);
          const float4 m0 = make_float4(p.cam[0], p.cam[1], p.cam[2], p.cam[3]);
          const float4 m1 = make_float4(p.cam[4], p.cam[5], p.cam[6], p.cam[7]);
          const float4 m2 = make_float4(p.cam[8], p.cam[9], p.cam[10], p.cam[11]);
          ox = Dot4(m0, o4); oy = Dot4(m1, o4); oz = Dot4(m2, o4);
          dx = Dot4(m0, d4); dy = Dot4(m1, d4); dz = Dot4(m2, d4);
        }
      }

      RayCtx r;
      SetupRay(r, ox, oy, oz, dx, dy, dz, p.scale);
      RayCtxQ rq;
      if (QUANT) SetupRayQ(rq, r, p.grid, p.scale);
      Best best{kFltMax, 0.f, 0.f, RT_MISS_ID};
      Stack st = MakeStack(wave_lds, lane, p.stack_entries);
      uint32_t cur = live ? 0u : kDone;
      uint32_t nnode = 0, ntri = 0, nleaf = 0, maxsp = 0;

      // One wave-uniform decision per iteration: take a node step (lanes holding an inner reference) or a
      // leaf step (lanes parked at a leaf).  Descending until EVERY lane is parked makes the wave wait for its
      // slowest lane in every phase; testing leaves as soon as they appear wastes the costlier triangle test
      // on a few lanes.  Switching to the leaf step once `leaf_pct` percent of the busy lanes are parked sits
      // between the two and shortens both the average and the longest wave.
      uint32_t iters = 0;
      if (SPLIT && QUANT && WIDTH == 4 && coop) {
        LdsWord *quad_col = (LdsWord *)(reinterpret_cast<uint32_t *>(wave_lds) + (lane & ~3u));
        iters = CoopTraverse<COUNT>(p, r, rq, quad_col, lane & 3u, live, best, nnode, ntri, nleaf, maxsp);
      } else
      for (;;) {
        iters++;
        const bool inner = !(cur & kLeafBitDev);
        const bool parked = !inner && cur != kDone;
        const unsigned long long mi = __ballot(inner), ml = __ballot(parked);
        if ((mi | ml) == 0ull) break;
        const uint32_t ni = __popcll(mi), nl = __popcll(ml);
        if (ni != 0u && nl * 100u < p.leaf_pct * (ni + nl)) {
          if (inner) {
            if (COUNT) nnode++;
            bool stepped = false;
            if (RT_SCALAR_NODE && QUANT && WIDTH == 4) {
              const uint32_t node = __builtin_amdgcn_readfirstlane(cur);   // of the first stepping lane
              const unsigned long long others = __ballot(cur != node);
              if (others == 0ull) {                                       // ... and of all of them
                cur = NodeStepUniformQ4(p, rq, node, best.t, st, deep);
                stepped = true;
              }
            }
            if (!stepped) cur = NodeStep<WIDTH, QUANT>(p, r, rq, cur, best.t, st, deep);
            if (COUNT) maxsp = max(maxsp, StackDepth(st));
          }
        } else if (parked) {
          if (COUNT) nleaf += (cur & 7u) ? 1u : 0u;
          LeafTest<COUNT>(p, r, cur, best, ntri);
          cur = Pop(st, deep, best.t);
        }
      }

      wave_cost = iters;
      if (COUNT) {
        c_node += nnode; c_tri += ntri; c_leaf += nleaf;
        c_hit += best.id != RT_MISS_ID && (!coop || (lane & 3u) == 0u) ? 1u : 0u;   // a cooperative quad holds one ray
        c_deep = maxsp;
      }

      // ---- output record (tracekernel.ck:282-286 miss defaults, :373-399 hit) ----------
      const bool hit = best.id != RT_MISS_ID;
      // of a cooperative quad, lane 0 writes the ray's outputs (whether the wave is one is worked out again rather than kept
      // in a register across the traversal loop)
      bool writer = true;
      if (SPLIT && QUANT && WIDTH == 4) {
        uint32_t w = threadIdx.x >> 6;
        asm volatile("" : "+v"(w));
        const bool q = blockIdx.x * p.waves_per_block + __builtin_amdgcn_readfirstlane(w) < 4u * p.split_units;
        writer = !(q && p.coop != 0u) || (threadIdx.x & 3u) == 0u;
      }
      if (p.hits16 && writer) {
        StoreCompactHit<MODE != kModeTrace>(p, idx, best.id, best.t, hit ? best.u : 0.f, hit ? best.v : 0.f);
      }
      float4 color = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((p.hits || (MODE == kModeTrace && p.rgba)) && writer) {
        float4 out_o = in_o, out_n = in_d, out_d = in_d;
        float out_u = 0.f, out_v = 0.f;
        if (MODE == kModeTrace && !live) {
          out_n = LoadStreamF4<MODE != kModeTrace>(rp + 2);  // terminated record passes through unchanged
        }
        if (hit) {
          const rt_face f = p.faces[best.id];
          const float u = best.u, v = best.v, w = 1.f - u - v;
          const float4 v0 = *reinterpret_cast<const float4 *>(p.verts + f.s[0]);
          const float4 v1 = *reinterpret_cast<const float4 *>(p.verts + f.s[1]);
          const float4 v2 = *reinterpret_cast<const float4 *>(p.verts + f.s[2]);
          out_o = make_float4(w * v0.x + u * v1.x + v * v2.x, w * v0.y + u * v1.y + v * v2.y,
                              w * v0.z + u * v1.z + v * v2.z, 1.f);
          float nx, ny, nz;
          if (f.s[4] < p.nnormals && f.s[5] < p.nnormals && f.s[6] < p.nnormals) {
            // u*N[n0] + v*N[n1] + (1-u-v)*N[n2]: the reference's pairing (tracekernel.ck:376)
            const float4 *np4 = reinterpret_cast<const float4 *>(p.normals);
            const float4 n0 = np4[f.s[4]], n1 = np4[f.s[5]], n2 = np4[f.s[6]];
            nx = u * n0.x + v * n1.x + w * n2.x;
            ny = u * n0.y + v * n1.y + w * n2.y;
            nz = u * n0.z + v * n1.z + w * n2.z;
          } else {
            // no vertex normals in the scene: unit geometric normal
            const float e1x = v1.x - v0.x, e1y = v1.y - v0.y, e1z = v1.z - v0.z;
            const float e2x = v2.x - v0.x, e2y = v2.y - v0.y, e2z = v2.z - v0.z;
            nx = e1y * e2z - e1z * e2y; ny = e1z * e2x - e1x * e2z; nz = e1x * e2y - e1y * e2x;
            const float len2 = nx * nx + ny * ny + nz * nz;
            if (len2 > 0.f) {   // one hardware reciprocal square root (1 ulp) instead of a square root and three divides
              const float inv_len = __builtin_amdgcn_rsqf(len2);
              nx *= inv_len; ny *= inv_len; nz *= inv_len;
            }
          }
          out_n = make_float4(nx, ny, nz, 0.f);
          out_d = make_float4(dx, dy, dz, 0.f);
          out_u = u; out_v = v;
          if (MODE == kModeTrace) {
            const uint32_t mi = f.s[3] < p.nmats ? f.s[3] : 0u;
            if (p.nmats) {
              const rt_material m = p.mats[mi];
              const float k = fabsf(nx * dx + ny * dy + nz * dz);
              color = make_float4(k * m.color.x, k * m.color.y, k * m.color.z, m.Rf);
            }
          }
        }
        rec0 = out_o; rec1 = out_d; rec2 = out_n;
        rec3 = make_float4(out_u, out_v, __uint_as_float(best.id), best.t);
      }
      if (MODE == kModeTrace && p.rgba && writer) {
        if (p.pixels8) reinterpret_cast<uint32_t *>(p.rgba)[idx] = PackBgra8(color);
        else StoreStreamF4<MODE != kModeTrace>(reinterpret_cast<float4 *>(p.rgba) + idx, color);
      }
    }
    // After the traversal loop a SPLIT kernel works out again whether the wave holds a quarter, instead of carrying the answer
    // through the loop in registers the loop is short of.
    bool quarter_out = quarter;
    if (SPLIT) {
      uint32_t w = threadIdx.x >> 6;
      asm volatile("" : "+v"(w));
      quarter_out = blockIdx.x * p.waves_per_block + __builtin_amdgcn_readfirstlane(w) < 4u * p.split_units;   // never with a cursor
      pos.sub = quarter_out ? 1u : 0u;
    }
    const bool coop_out = SPLIT && QUANT && WIDTH == 4 && quarter_out && p.coop != 0u;
    if (p.hits) {
      // 64-byte records, one per lane: written straight from the lanes, every store instruction would scatter 64
      // 16-byte pieces over 32 cache lines.  The wave's traversal stack is idle by now, so the records are
      // exchanged through it and each of the four stores covers 16 whole records (1 KB runs in linear mode,
      // 512-byte rows of an 8x8 tile in image mode).
      // [record][quarter ^ swizzle(record)]: the swizzle spreads the 64-byte-strided writes over all banks
      float4 *tile = reinterpret_cast<float4 *>(wave_lds);
      const uint32_t wl = coop_out ? lane >> 2 : lane;   // which record of the unit this lane holds
      const uint32_t wsw = (wl >> 2) & 3u;
      if (!coop_out || (lane & 3u) == 0u) {
        tile[wl * 4u + (0u ^ wsw)] = rec0; tile[wl * 4u + (1u ^ wsw)] = rec1;
        tile[wl * 4u + (2u ^ wsw)] = rec2; tile[wl * 4u + (3u ^ wsw)] = rec3;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t src = k * 16u + (lane >> 2), quarter = lane & 3u;
        const float4 v = tile[src * 4u + (quarter ^ ((src >> 2) & 3u))];
        uint32_t src_idx;
        if (RayOfLane<SPLIT>(p, pos, src, src_idx)) StoreStreamF4<MODE != kModeTrace>(reinterpret_cast<float4 *>(p.hits + src_idx) + quarter, v);
      }
      if (p.cursor) {   // the next round's pushes must not overtake this round's reads of the exchange tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (p.unit_cost) {
      // The cost is recorded for the unit and, scaled down by cost_share/128, for its neighbourhood (maximum
      // wins; the buffer is zero before the launch): between two frames of a moving camera the expensive places
      // of the image shift by a tile or two, and the order built from these numbers has to start THOSE tiles
      // early as well, while ordinary tiles keep their own number.  One atomic instruction per wave: lane l
      // handles neighbour l of the (2R+1)^2 window.
      // A quarter wave adds its count to the unit's own cell: the sum of the four is at least what the whole unit would
      // take as one wave, so a unit that was split keeps its rank and stays split while it stays expensive.
      const uint32_t cost = __builtin_amdgcn_readfirstlane(wave_cost);   // lane 0 owns a ray if any lane does
      const uint32_t side = 2u * p.cost_radius + 1u;
      if (p.img_w) {
        const uint32_t tiles_x = (p.img_w + 7u) >> 3, tiles_y = (p.img_h + 7u) >> 3;
        const uint32_t nx = (pos.x0 >> 3) + lane % side - p.cost_radius, ny = (pos.y0 >> 3) + lane / side - p.cost_radius;
        const bool self = nx == (pos.x0 >> 3) && ny == (pos.y0 >> 3);
        if (lane < side * side && nx < tiles_x && ny < tiles_y) {
          if (self && quarter_out) atomicAdd(p.unit_cost + ny * tiles_x + nx, cost);
          else atomicMax(p.unit_cost + ny * tiles_x + nx, self ? cost : cost * p.cost_share / 128u);
        }
      } else {
        const uint32_t nu = unit + lane - p.cost_radius;
        if (lane < side && nu < p.ntasks) {
          if (nu == unit && quarter_out) atomicAdd(p.unit_cost + nu, cost);
          else atomicMax(p.unit_cost + nu, nu == unit ? cost : cost * p.cost_share / 128u);
        }
      }
    }
    if (COUNT) {
      // one atomic per counter per wave, so the instrumented launch keeps the real launch's timing shape
      for (int off = 32; off > 0; off >>= 1) {
        c_node += __shfl_down(c_node, off); c_tri += __shfl_down(c_tri, off);
        c_leaf += __shfl_down(c_leaf, off); c_hit += __shfl_down(c_hit, off);
        c_deep = max(c_deep, __shfl_down(c_deep, off));
      }
      if (lane == 0) {
        // 64 shards, one 128-byte line each: atomics on one line serialise at a few tens per microsecond
        unsigned long long *sh = p.stats + (slot & 63u) * 16u;
        atomicAdd(sh + 0, static_cast<unsigned long long>(c_node));
        atomicAdd(sh + 1, static_cast<unsigned long long>(c_tri));
        atomicAdd(sh + 2, static_cast<unsigned long long>(c_leaf));
        atomicAdd(sh + 3, static_cast<unsigned long long>(c_hit));
        atomicMax(sh + 4, static_cast<unsigned long long>(c_deep));   // deepest stack any ray of the launch reached
      }
    }
    if (COUNT && p.timeline) {
      // diagnostic only: when and where this wave ran (the values feed nothing else)
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
      uint32_t hw_id, xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      if (lane == 0) {
        unsigned long long *rec = p.timeline + static_cast<size_t>(slot) * 4u;
        rec[0] = t_start; rec[1] = t_end; rec[2] = hw_id; rec[3] = xcc_id;
      }
    }
    if (!p.cursor) return;
  }
}

// ---- incoherent rays: a wave's lanes take new rays as their old ones finish ------------------------------------
// With one batch of 64 rays per wave, a wave of incoherent rays is as long as its longest ray and its other lanes idle
// for the rest of it (AO-style rays: a quarter of the lanes busy per vector instruction).  Here a fixed grid of waves
// streams through the ray set instead: whenever `refill_min` lanes of a wave are idle, the finished ones store their
// compact hit and the idle ones take the next rays of the wave's current chunk (kRefillChunk consecutive records, claimed
// with one wave-level atomic on a global cursor), set up and start at the root while the other lanes carry on where they
// are.  Linear ray sets only (no image tiles).  Every wave leaves once the cursor has passed the last ray and
// its lanes have finished -- an exit every wave reaches, whatever the rays are.

// The 64-byte hit record of a ray (tracekernel.ck:282-286 miss defaults, :373-399 hit) as four 16-byte quarters.
__device__ __forceinline__ void HitRecord(const KernelArgs &p, const Best &best, const float4 in_o, const float4 in_d, float4 &rec0, float4 &rec1,
                                          float4 &rec2, float4 &rec3)
{
  float4 out_o = in_o, out_n = in_d, out_d = in_d;
  float out_u = 0.f, out_v = 0.f;
  if (best.id != RT_MISS_ID) {
    const rt_face f = p.faces[best.id];
    const float u = best.u, v = best.v, w = 1.f - u - v;
    const float4 v0 = *reinterpret_cast<const float4 *>(p.verts + f.s[0]);
    const float4 v1 = *reinterpret_cast<const float4 *>(p.verts + f.s[1]);
    const float4 v2 = *reinterpret_cast<const float4 *>(p.verts + f.s[2]);
    out_o = make_float4(w * v0.x + u * v1.x + v * v2.x, w * v0.y + u * v1.y + v * v2.y, w * v0.z + u * v1.z + v * v2.z, 1.f);
    float nx, ny, nz;
    if (f.s[4] < p.nnormals && f.s[5] < p.nnormals && f.s[6] < p.nnormals) {
      // u*N[n0] + v*N[n1] + (1-u-v)*N[n2]: the reference's pairing (tracekernel.ck:376)
      const float4 *np4 = reinterpret_cast<const float4 *>(p.normals);
      const float4 n0 = np4[f.s[4]], n1 = np4[f.s[5]], n2 = np4[f.s[6]];
      nx = u * n0.x + v * n1.x + w * n2.x;
      ny = u * n0.y + v * n1.y + w * n2.y;
      nz = u * n0.z + v * n1.z + w * n2.z;
    } else {
      // no vertex normals in the scene: unit geometric normal (one hardware reciprocal square root)
      const float e1x = v1.x - v0.x, e1y = v1.y - v0.y, e1z = v1.z - v0.z;
      const float e2x = v2.x - v0.x, e2y = v2.y - v0.y, e2z = v2.z - v0.z;
      nx = e1y * e2z - e1z * e2y; ny = e1z * e2x - e1x * e2z; nz = e1x * e2y - e1y * e2x;
      const float len2 = nx * nx + ny * ny + nz * nz;
      if (len2 > 0.f) {
        const float inv_len = __builtin_amdgcn_rsqf(len2);
        nx *= inv_len; ny *= inv_len; nz *= inv_len;
      }
    }
    out_n = make_float4(nx, ny, nz, 0.f);
    out_d = make_float4(in_d.x, in_d.y, in_d.z, 0.f);
    out_u = u; out_v = v;
  }
  rec0 = out_o; rec1 = out_d; rec2 = out_n;
  rec3 = make_float4(out_u, out_v, __uint_as_float(best.id), best.t);
}

// Finished rays leave a refilling wave at unrelated times and hold unrelated records, so there is no tile to exchange as in the
// batch kernel; written straight from the lanes, the four 16-byte quarters of a record leave in four store instructions and
// reach memory as partial lines (1.8x the record bytes on the HBM side).  Instead the retiring lanes park their records in
// their own stack columns (a finished ray's stack is empty: rows 0..7 of its column take the 16 dwords), a small table maps
// the rank of a retiring lane to its lane number and ray index, and the wave then writes sixteen WHOLE records per store
// instruction: lane 4s + q fetches quarter q of the s-th retiring record.
constexpr uint32_t kRetireTableBytes = 512;   // per wave, behind its stack part: 64 lane numbers + 64 ray indices

__device__ __forceinline__ void RetireRecords(const KernelArgs &p, uint32_t stack_first, uint32_t lane, bool retiring, uint32_t idx,
                                              const float4 &rec0, const float4 &rec1, const float4 &rec2, const float4 &rec3)
{
  const unsigned long long rm = __ballot(retiring);
  const uint32_t nret = __popcll(rm);
  if (nret == 0u) return;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(rm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(rm), 0u));
  LdsWord *const col = reinterpret_cast<LdsWord *>(stack_first + 4u * lane);              // row k: col[128 k] and col[128 k + 64]
  LdsWord *const table = reinterpret_cast<LdsWord *>(stack_first + kEntryBytes * p.stack_entries);
  if (retiring) {
    const float4 rec[4] = {rec0, rec1, rec2, rec3};
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) {
      col[(2u * q) * 128u] = __float_as_uint(rec[q].x); col[(2u * q) * 128u + 64u] = __float_as_uint(rec[q].y);
      col[(2u * q + 1u) * 128u] = __float_as_uint(rec[q].z); col[(2u * q + 1u) * 128u + 64u] = __float_as_uint(rec[q].w);
    }
    table[rank] = lane; table[64u + rank] = idx;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint32_t q = lane & 3u;
  for (uint32_t s = lane >> 2; s < nret; s += 16u) {   // sixteen records per round
    const uint32_t from = table[s], ray = table[64u + s];
    const LdsWord *src = reinterpret_cast<const LdsWord *>(stack_first + 4u * from) + (2u * q) * 128u;
    const float4 v = make_float4(__uint_as_float(src[0]), __uint_as_float(src[64]), __uint_as_float(src[128]), __uint_as_float(src[192]));
    StoreStreamF4<true>(reinterpret_cast<float4 *>(p.hits + ray) + q, v);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the columns go back to being stacks
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(256, kRefillWavesPerSimd) RefillHitKernel(KernelArgs p)
{
  extern __shared__ uint2 lds_stack[];
  const uint32_t lane = threadIdx.x & 63u, wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint2 *const wave_lds = lds_stack + wave_s * (p.stack_entries * 64u + kRetireTableBytes / 8u);   // stack part, then the retire table
  uint2 deep[kDeepEntries];
  RayCtx r;
  RayCtxQ rq;
  SetupRay(r, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, p.scale);   // defined values for lanes that never get a ray
  SetupRayQ(rq, r, p.grid, p.scale);
  Best best{kFltMax, 0.f, 0.f, RT_MISS_ID};
  Stack st = MakeStack(wave_lds, lane, p.stack_entries);
  constexpr uint32_t kNoRay = 0xFFFFFFFFu;
  uint32_t cur = kDone, idx = kNoRay;
  float ray_ow = 0.f, ray_dw = 0.f;   // fourth components of the ray record: a missed ray's record passes them through
  uint32_t c_next = 0, c_end = 0;   // the wave's chunk: records [c_next, c_end) are still to be handed out
  bool exhausted = false;           // the global cursor has passed the last ray
  for (;;) {
    const unsigned long long idle = __ballot(cur == kDone);
    const uint32_t nidle = __popcll(idle);
    if (nidle == 64u || (!exhausted && nidle >= p.refill_min)) {
      // claim, then start loading the new rays, then write the finished ones out (their lookups run under the ray loads),
      // then set the new rays up
      uint32_t take = 0;
      if (!exhausted) {
        if (c_next == c_end) {
          uint32_t got = 0;
          if (lane == 0) got = atomicAdd(p.cursor, kRefillChunk);
          c_next = __builtin_amdgcn_readfirstlane(got);
          if (c_next >= p.n) {
            exhausted = true;
            c_next = c_end = 0;
          } else {
            c_end = min(c_next + kRefillChunk, p.n);
          }
        }
        take = min(nidle, c_end - c_next);
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
      const bool getting = cur == kDone && rank < take;
      const uint32_t new_idx = c_next + rank;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f), d = o;
      if (getting) {
        const float4 *rp = reinterpret_cast<const float4 *>(p.rays + new_idx);
        o = LoadStreamF4<true>(rp);
        d = LoadStreamF4<true>(rp + 1);
      }
      const bool retiring = cur == kDone && idx != kNoRay;
      float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0, rec2 = rec0, rec3 = rec0;
      if (retiring) {
        const bool hit = best.id != RT_MISS_ID;
        if (p.hits16) StoreCompactHit<true>(p, idx, best.id, best.t, hit ? best.u : 0.f, hit ? best.v : 0.f);
        if (p.hits) HitRecord(p, best, make_float4(r.ox, r.oy, r.oz, ray_ow), make_float4(r.dx, r.dy, r.dz, ray_dw), rec0, rec1, rec2, rec3);
      }
      if (p.hits) RetireRecords(p, st.first, lane, retiring, idx, rec0, rec1, rec2, rec3);
      if (retiring) idx = kNoRay;
      if (getting) {
        idx = new_idx;
        SetupRay(r, o.x, o.y, o.z, d.x, d.y, d.z, p.scale);
        SetupRayQ(rq, r, p.grid, p.scale);
        ray_ow = o.w; ray_dw = d.w;
        best = Best{kFltMax, 0.f, 0.f, RT_MISS_ID};
        st.top = st.first + 4u * lane;
        cur = 0u;
      }
      c_next += take;
      if (exhausted && __ballot(cur != kDone) == 0ull) return;
    }
    const bool inner = !(cur & kLeafBitDev);
    const bool parked = !inner && cur != kDone;
    const unsigned long long mi = __ballot(inner), ml = __ballot(parked);
    const uint32_t ni = __popcll(mi), nl = __popcll(ml);
    if (ni != 0u && nl * 100u < p.leaf_pct * (ni + nl)) {
      if (inner) cur = NodeStep<4, true>(p, r, rq, cur, best.t, st, deep);
    } else if (parked) {
      uint32_t unused = 0;
      LeafTest<false>(p, r, cur, best, unused);
      cur = Pop(st, deep, best.t);
    }
  }
}

// Counting sort of work units by cost bin (cost / 8, capped), costliest bin first, ascending unit index inside a bin so
// neighbouring units stay neighbours.  Small launches of ONE-WAVE workgroups: the sort may run on a side stream beside a
// trace launch that refills every wave slot the moment it frees, where a workgroup of several waves, which needs that many
// slots of one CU free at the same time, would wait for most of that launch.  Every wave owns a chunk of kOrderChunk
// consecutive keys (16 per lane, read as four 16-byte loads);
//   1. OrderCountKernel    bins[bin][chunk] = keys of the chunk in that bin        (bin 0 = the costliest)
//   2. OrderScanKernel     exclusive prefix sum over bins[][] in bin-major order  = where each (bin, chunk) run starts
//   3. OrderScatterKernel  ranks the chunk's keys inside each bin in index order and writes the order; zeroes the keys
// By default the sort runs on the trace launches' own stream, between two of them, where each launch costs its start-up
// latency (6 + 10 + 7 us at 1080p for microseconds of work): up to kOrderDirectChunks chunks step 2 is folded into step 3
// (every workgroup sums the rows of the raw count table itself), two launches instead of three.
// unit_cost must be readable and writable up to OrderPaddedUnits(n) entries and is left all zero.
constexpr uint32_t kOrderKeys = 16, kOrderChunk = 64u * kOrderKeys;
constexpr uint32_t kOrderDirectChunks = 512;   // up to this many chunks (512 Ki units) the scan launch is folded into the scatter

__device__ __forceinline__ uint32_t CostBin(uint32_t cost)
{
  return static_cast<uint32_t>(kFbBuckets - 1) - min(cost >> 3, static_cast<uint32_t>(kFbBuckets - 1));
}

__device__ __forceinline__ void LoadOrderKeys(const uint32_t *unit_cost, uint32_t first, uint32_t (&v)[kOrderKeys])
{
  const uint4 *k4 = reinterpret_cast<const uint4 *>(unit_cost + first);
#pragma unroll
  for (uint32_t j = 0; j < kOrderKeys / 4u; ++j) {
    const uint4 k = k4[j];
    v[4u * j] = k.x; v[4u * j + 1u] = k.y; v[4u * j + 2u] = k.z; v[4u * j + 3u] = k.w;
  }
}

// Inclusive prefix sum over the 64 lanes of the wave.
__device__ __forceinline__ uint32_t WaveScan(uint32_t x, uint32_t lane)
{
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off);
    if (lane >= static_cast<uint32_t>(off)) x += y;
  }
  return x;
}

__global__ void __launch_bounds__(64) OrderCountKernel(const uint32_t *unit_cost, uint32_t n, uint32_t nchunks, uint32_t *bins)
{
  __shared__ uint32_t count[kFbBuckets];
  const uint32_t t = threadIdx.x, first = blockIdx.x * kOrderChunk + t * kOrderKeys;
  if (t < static_cast<uint32_t>(kFbBuckets)) count[t] = 0;
  __syncthreads();
  uint32_t v[kOrderKeys];
  LoadOrderKeys(unit_cost, first, v);
#pragma unroll
  for (uint32_t e = 0; e < kOrderKeys; ++e)
    if (first + e < n) atomicAdd(&count[CostBin(v[e])], 1u);
  __syncthreads();
  if (t < static_cast<uint32_t>(kFbBuckets)) bins[t * nchunks + blockIdx.x] = count[t];
}

// One wave walks the table in tiles of 2048 entries staged in LDS with coalesced accesses (8 KB: a workgroup that asks for
// more LDS than the trace launch beside it leaves free on a CU waits for that launch to drain); inside a tile lane t sums
// a contiguous run, the runs' totals are scanned across the lanes, and the tile's total carries into the next.
constexpr uint32_t kOrderScanTile = 2048;

__global__ void __launch_bounds__(64) OrderScanKernel(uint32_t *bins, uint32_t total)
{
  __shared__ uint32_t staged[kOrderScanTile];
  const uint32_t t = threadIdx.x, per = kOrderScanTile / 64u;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < total; base += kOrderScanTile) {
    const uint32_t count = min(kOrderScanTile, total - base);
    for (uint32_t i = t; i < count; i += 64u) staged[i] = bins[base + i];
    __syncthreads();
    const uint32_t lo = min(t * per, count), hi = min(lo + per, count);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += staged[i];
    const uint32_t incl = WaveScan(sum, t);
    uint32_t run = carry + incl - sum;
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t c = staged[i];
      staged[i] = run;
      run += c;
    }
    carry += __shfl(incl, 63);
    __syncthreads();
    for (uint32_t i = t; i < count; i += 64u) bins[base + i] = staged[i];
    __syncthreads();
  }
}

// SCANNED: bins holds the scanned table (OrderScanKernel ran).  Otherwise it holds the raw counts and every workgroup works
// out its own starts from them -- lane b sums row b (all chunks: the bin's total; the chunks before this one: the offset
// inside the bin) and the totals are scanned across the lanes: redundant between workgroups, a few KB of L2 reads each, and
// one launch fewer between two trace launches.
template <bool SCANNED>
__global__ void __launch_bounds__(64) OrderScatterKernel(uint32_t *unit_cost, uint32_t n, uint32_t nchunks, const uint32_t *bins,
                                                        uint32_t *order)
{
  __shared__ uint32_t table[kFbBuckets * 64];   // [bin][lane]: counts, then next free position
  const uint32_t t = threadIdx.x, first = blockIdx.x * kOrderChunk + t * kOrderKeys;
  // where the chunk's run of bin t starts in the order (kFbBuckets <= 64)
  uint32_t start = 0u;
  if (SCANNED) {
    if (t < static_cast<uint32_t>(kFbBuckets)) start = bins[t * nchunks + blockIdx.x];
  } else {
    uint32_t total = 0u, before = 0u;
    if (t < static_cast<uint32_t>(kFbBuckets)) {
      const uint32_t *row = bins + t * nchunks;
      for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t v = row[c];
        total += v;
        before += c < blockIdx.x ? v : 0u;
      }
    }
    start = WaveScan(total, t) - total + before;   // lanes beyond the bins hold zeros
  }
  uint32_t v[kOrderKeys];
  LoadOrderKeys(unit_cost, first, v);
  {
    uint4 *k4 = reinterpret_cast<uint4 *>(unit_cost + first);   // the next launch that records into this buffer takes maxima
#pragma unroll
    for (uint32_t j = 0; j < kOrderKeys / 4u; ++j) k4[j] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int b = 0; b < kFbBuckets; ++b) table[b * 64 + t] = 0;
  // every lane only touches its own column of the table
#pragma unroll
  for (uint32_t e = 0; e < kOrderKeys; ++e)
    if (first + e < n) table[CostBin(v[e]) * 64u + t] += 1u;
#pragma unroll
  for (int b = 0; b < kFbBuckets; ++b) {   // per bin: exclusive scan over the lanes, offset by the run's start
    const uint32_t c = table[b * 64 + t];
    table[b * 64 + t] = __shfl(start, b) + WaveScan(c, t) - c;
  }
#pragma unroll
  for (uint32_t e = 0; e < kOrderKeys; ++e) {
    if (first + e < n) {
      uint32_t *cell = &table[CostBin(v[e]) * 64u + t];
      const uint32_t pos = *cell;
      *cell = pos + 1u;
      order[pos] = first + e;
    }
  }
}

// Tile-major records -> row-major framebuffer.  Tile i of the list covers the pixel block of tile_ids[i] in a W x H image
// cut into `tile`-sized squares (edge tiles are smaller); its records, row-major inside the block, start at record
// tile_first[i] of `src`.  gridDim.x = tiles, gridDim.y = workgroups that share a tile (large squares).  Consecutive threads
// move consecutive pieces of a tile row, 16 bytes each when the rows on both sides are 16-byte aligned (else 4 bytes), so
// both sides are read and written in runs of a tile row.
template <typename PIECE>
__device__ __forceinline__ void MoveTileRows(const char *from, char *to, uint32_t row_bytes, uint32_t h, size_t pitch_bytes)
{
  const uint32_t row_pieces = row_bytes / static_cast<uint32_t>(sizeof(PIECE)), total = row_pieces * h;
  for (uint32_t k = threadIdx.x + blockIdx.y * blockDim.x; k < total; k += blockDim.x * gridDim.y) {
    const uint32_t row = k / row_pieces, col = k - row * row_pieces;
    reinterpret_cast<PIECE *>(to + row * pitch_bytes)[col] = reinterpret_cast<const PIECE *>(from)[k];
  }
}

__global__ void __launch_bounds__(256) ScatterTilesKernel(const char *src, uint32_t src_records, const uint32_t *tile_ids,
                                                          const uint32_t *tile_first, uint32_t tile, uint32_t img_w, uint32_t img_h,
                                                          uint32_t record_bytes, char *dst)
{
  const uint32_t tiles_x = (img_w + tile - 1u) / tile, tiles_y = (img_h + tile - 1u) / tile;
  const uint32_t id = tile_ids[blockIdx.x];
  if (id >= tiles_x * tiles_y) return;   // a list entry that names no square of this image moves nothing
  const uint32_t ty = id / tiles_x, tx = id - ty * tiles_x;
  const uint32_t x0 = tx * tile, y0 = ty * tile;
  const uint32_t w = min(tile, img_w - x0), h = min(tile, img_h - y0);
  const uint32_t first = tile_first[blockIdx.x];
  if (static_cast<uint64_t>(first) + static_cast<uint64_t>(w) * h > src_records) return;   // nor one that runs off src
  const size_t from_off = static_cast<size_t>(first) * record_bytes, pitch = static_cast<size_t>(img_w) * record_bytes;
  const size_t to_off = (static_cast<size_t>(y0) * img_w + x0) * record_bytes;
  const uint32_t row_bytes = w * record_bytes;
  const bool wide = ((from_off | pitch | to_off | row_bytes | reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15u) == 0u;
  if (wide) MoveTileRows<uint4>(src + from_off, dst + to_off, row_bytes, h, pitch);
  else MoveTileRows<uint32_t>(src + from_off, dst + to_off, row_bytes, h, pitch);
}

// RGBA32F pixels -> 8-bit BGRA, the byte order a 32-bit bitmap stores and the reference's writer passes through
// (cpu_raycast/bmp.h:12, bmp.c:49-53): each channel
// min(255, max(0, x * 255 + 0.5)) truncated, alpha 255.
__global__ void __launch_bounds__(256) PackPixelsKernel(const float4 *rgba, uint32_t n, uint32_t *bgra)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bgra[i] = PackBgra8(rgba[i]);
}

__global__ void TriBatchKernel(const rt_vec3 *tri, const rt_ray *ray, uint32_t n, int32_t *hit, float *t, rt_vec2 *uv)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const rt_vec3 a = tri[3 * i], b = tri[3 * i + 1], c = tri[3 * i + 2];
  RayCtx r;
  SetupRay(r, ray[i].origin.x, ray[i].origin.y, ray[i].origin.z, ray[i].dir.x, ray[i].dir.y, ray[i].dir.z, 0.f);
  float tt, u, v;
  const bool ok = TriTest(a.x, a.y, a.z, b.x - a.x, b.y - a.y, b.z - a.z, c.x - a.x, c.y - a.y, c.z - a.z, r, tt, u, v);
  hit[i] = ok ? 1 : 0;
  t[i] = tt;
  uv[i].u = u;
  uv[i].v = v;
}

// `done` (nullable): an event that the kernel's own completion signals.  Recording an event behind the launch instead puts
// a separate packet into the queue, and the next launch of the stream then starts ~6-12 us after this one ends.
template <int WIDTH, bool QUANT, int MODE, bool COUNT, bool SPLIT>
hipError_t LaunchOne(const KernelArgs &a, uint32_t grid, uint32_t block, size_t lds, hipStream_t stream, hipEvent_t done)
{
  (void)hipGetLastError();  // the launch status below must be this launch's, not an earlier call's leftover
  if (lds > 64u * 1024u) {  // beyond the default dynamic-LDS limit: ask for what gfx950's 160 KB allow
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ClosestHitKernel<WIDTH, QUANT, MODE, COUNT, SPLIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  if (done)
    hipExtLaunchKernelGGL((ClosestHitKernel<WIDTH, QUANT, MODE, COUNT, SPLIT>), dim3(grid), dim3(block), static_cast<uint32_t>(lds),
                          stream, nullptr, done, 0u, a);
  else
    hipLaunchKernelGGL((ClosestHitKernel<WIDTH, QUANT, MODE, COUNT, SPLIT>), dim3(grid), dim3(block), lds, stream, a);
  return hipGetLastError();
}

template <int WIDTH, bool QUANT>
hipError_t LaunchLayout(const KernelArgs &a, int mode, bool count, uint32_t grid, uint32_t block, size_t lds,
                        hipStream_t stream, hipEvent_t done)
{
  if (count) {
    if (a.split_units) return LaunchOne<WIDTH, QUANT, kModeIntersect, true, true>(a, grid, block, lds, stream, done);
    return LaunchOne<WIDTH, QUANT, kModeIntersect, true, false>(a, grid, block, lds, stream, done);
  }
  if (mode == kModeTrace && a.split_units) return LaunchOne<WIDTH, QUANT, kModeTrace, false, true>(a, grid, block, lds, stream, done);
  if (mode == kModeTrace) return LaunchOne<WIDTH, QUANT, kModeTrace, false, false>(a, grid, block, lds, stream, done);
  if (a.split_units) {
    if (mode == kModePrimary) return LaunchOne<WIDTH, QUANT, kModePrimary, false, true>(a, grid, block, lds, stream, done);
    return LaunchOne<WIDTH, QUANT, kModeIntersect, false, true>(a, grid, block, lds, stream, done);
  }
  if (mode == kModePrimary) return LaunchOne<WIDTH, QUANT, kModePrimary, false, false>(a, grid, block, lds, stream, done);
  return LaunchOne<WIDTH, QUANT, kModeIntersect, false, false>(a, grid, block, lds, stream, done);
}

}  // namespace

hipError_t LaunchClosestHit(const KernelArgs &args, int width, bool quant, int mode, bool count, uint32_t grid,
                            uint32_t block, hipStream_t stream, hipEvent_t done)
{
  KernelArgs a = args;
  a.waves_per_block = block / 64u;
  a.tiles_x = (a.img_w + 7u) / 8u;
  a.tiles_x_magic = a.tiles_x > 1u ? static_cast<uint32_t>((1ull << 32) / a.tiles_x) + 1u : 0u;
  const size_t lds = static_cast<size_t>(a.stack_entries) * 8u * block;  // LDS entries x 8 B x lanes of the workgroup
  if (width == 4)
    return quant ? LaunchLayout<4, true>(a, mode, count, grid, block, lds, stream, done)
                 : LaunchLayout<4, false>(a, mode, count, grid, block, lds, stream, done);
  return quant ? LaunchLayout<2, true>(a, mode, count, grid, block, lds, stream, done)
               : LaunchLayout<2, false>(a, mode, count, grid, block, lds, stream, done);
}

hipError_t LaunchRefill(const KernelArgs &a, uint32_t grid, uint32_t block, hipStream_t stream)
{
  (void)hipGetLastError();
  const size_t lds = (static_cast<size_t>(a.stack_entries) * 512u + kRetireTableBytes) * (block / 64u);   // per wave: stack part + retire table
  if (lds > 64u * 1024u) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&RefillHitKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(RefillHitKernel, dim3(grid), dim3(block), lds, stream, a);
  return hipGetLastError();
}

uint32_t OrderPaddedUnits(uint32_t nunits) { return (nunits + kOrderChunk - 1u) / kOrderChunk * kOrderChunk; }
uint32_t OrderBinWords(uint32_t nunits) { return (nunits + kOrderChunk - 1u) / kOrderChunk * static_cast<uint32_t>(kFbBuckets); }

hipError_t LaunchBuildOrder(uint32_t *unit_cost, uint32_t nunits, uint32_t *bins, uint32_t *order, hipStream_t stream)
{
  (void)hipGetLastError();
  const uint32_t nchunks = (nunits + kOrderChunk - 1u) / kOrderChunk;
  if (nchunks == 0) return hipSuccess;
  hipLaunchKernelGGL(OrderCountKernel, dim3(nchunks), dim3(64), 0, stream, unit_cost, nunits, nchunks, bins);
  if (nchunks <= kOrderDirectChunks) {   // two launches: the scatter works its starts out from the raw counts
    hipLaunchKernelGGL(OrderScatterKernel<false>, dim3(nchunks), dim3(64), 0, stream, unit_cost, nunits, nchunks, bins, order);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(OrderScanKernel, dim3(1), dim3(64), 0, stream, bins, nchunks * static_cast<uint32_t>(kFbBuckets));
  hipLaunchKernelGGL(OrderScatterKernel<true>, dim3(nchunks), dim3(64), 0, stream, unit_cost, nunits, nchunks, bins, order);
  return hipGetLastError();
}

hipError_t LaunchScatterTiles(const void *src, uint32_t src_records, const uint32_t *tile_ids, const uint32_t *tile_first, uint32_t ntiles,
                              uint32_t tile, uint32_t img_w, uint32_t img_h, uint32_t record_bytes, void *dst, hipStream_t stream)
{
  (void)hipGetLastError();
  // workgroups per square: about 16 KB of records each
  const uint64_t tile_bytes = static_cast<uint64_t>(tile) * tile * record_bytes;
  const uint32_t parts = static_cast<uint32_t>(std::min<uint64_t>(64u, std::max<uint64_t>(1u, tile_bytes / 16384u)));
  hipLaunchKernelGGL(ScatterTilesKernel, dim3(ntiles, parts), dim3(256), 0, stream, static_cast<const char *>(src), src_records, tile_ids,
                     tile_first, tile, img_w, img_h, record_bytes, static_cast<char *>(dst));
  return hipGetLastError();
}

hipError_t LaunchPackPixels(const float *rgba, uint32_t n, void *bgra, hipStream_t stream)
{
  (void)hipGetLastError();
  hipLaunchKernelGGL(PackPixelsKernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, reinterpret_cast<const float4 *>(rgba), n,
                     static_cast<uint32_t *>(bgra));
  return hipGetLastError();
}

hipError_t LaunchTriBatch(const rt_vec3 *tri, const rt_ray *ray, uint32_t n, int32_t *hit, float *t, rt_vec2 *uv,
                          hipStream_t stream)
{
  hipLaunchKernelGGL(TriBatchKernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, tri, ray, n, hit, t, uv);
  return hipGetLastError();
}

}  // namespace rt
