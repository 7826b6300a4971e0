/*
 * flx_walk2.hip — the bounce walks of the wavefront pipeline, second generation (round 3).
 *
 * What the first walk kernel (flx_wavefront.hip: k_wf_walk_pre) is short of is not arithmetic: its counters say a wave spends half
 * of its life parked on s_waitcnt — the 48-byte entry fetch of its slowest lane, every trip — with only four waves per SIMD to
 * cover for each other, because a lane carries ~126 registers and 120 bytes of LDS (its ray pre-transformed into every object
 * space).  This kernel is built around that:
 *
 *   registers   a lane holds its walk and nothing else.  The rays of the path (next ray, shadow ray), its colours, flags and
 *               bounce stay in the path record and are re-read at the rare points that need them (a shadow walk ending, the
 *               fold); the pre-transformed rays live in a per-thread slice of HBM-backed scratch, 32 bytes per object space,
 *               and their reciprocal directions are recomputed on arrival (nine instructions, a few times per walk).
 *               -> 5 or 6 waves per SIMD instead of 4 (FLX_W2_WAVES).
 *   fetches     the skip list in a PACKED copy (flx_api.hip: build_packed): an AABB is 32 bytes (min, max, two links), a
 *               triangle 48 (a, b - a, c - a, link, id); a link names its target's 16-byte slot, kind AND object space, so a
 *               walk knows before the fetch what it will fetch (two loads for a box: 4/5 of all visits), whether its ray changes
 *               space (that load goes out with the entry's, not after it) and that a terminator needs no fetch at all.
 *   LDS         with the rays gone it all goes to the top of the tree: ~2 400 boxes per workgroup instead of 761 entries.
 *
 * Per ray nothing changes: the entries visited, their order, every arithmetic operation (flx_device.h) and the visit counts are
 * those of k_wf_walk_pre, of the per-pixel kernels and of the CPU oracle; frames and work counters are bit-identical.
 */
#include <cstdio>
#include "flx_kernels.h"
#include "flx_kernel_util.h"
#include "flx_wavefront_common.h"

namespace flx {

#ifndef FLX_W2_WAVES
#define FLX_W2_WAVES 6                      /* waves per SIMD the register allocation must allow */
#endif
#ifndef FLX_W2_THREADS
#define FLX_W2_THREADS (FLX_W2_WAVES * 128) /* two workgroups per CU: each holds half of the CU's waves and half of its LDS */
#endif
#ifndef FLX_W2_INNER
#define FLX_W2_INNER 8                      /* trips between two looks at the scheduler state */
#endif
#ifndef FLX_W2_BATCH
#define FLX_W2_BATCH 24                     /* parked lanes that trigger a fold + refill */
#endif
#ifndef FLX_W2_DRAWS_PER_WAVE
#define FLX_W2_DRAWS_PER_WAVE 16
#endif
#ifndef FLX_W2_ITEMS_PER_LANE
#define FLX_W2_ITEMS_PER_LANE 4
#endif

/* lane states */
enum { Q_EMPTY = 0, Q_WALKING = 1, Q_DONE = 2, Q_SWITCH = 3, Q_SETUP = 4 };
/* lane flag word: the record's RF_* bits in the low byte, then */
constexpr int QF_CULL = 0x100;              /* the lane walks a shadow ray (shadowTest's rules) */
constexpr int QF_SHADOWED = 0x200;          /* its shadow walk found an occluder */
constexpr int QF_FAST = 0x400;              /* the current ray is inside the exact-reciprocal box test's range */
constexpr int QF_BOUNCE_SHIFT = 16;         /* the path's bounce */

struct Walk2 {
  f3 o, d, inv;                             /* the ray in the current object space, RN(1 / d) */
  float minLen;
  f3 suv; int tri;                          /* closest hit so far */
  uint32_t link;                            /* the entry held in s0..s2 (packed link: slot | kind | object space) */
  float4 s0, s1, s2;
};

typedef float w2_v4f __attribute__((ext_vector_type(4)));

/* the box test of flx_device.h on this kernel's state (rayCuboidFast: interval test, exact quotients where a lane is unsure) */
FLX_DEV bool w2Box(const Walk2 &w, bool fast) {
  WalkState t;
  t.tR.origin = w.o; t.tR.dir = w.d; t.inv = w.inv; t.fastDiv = fast;
  return rayCuboidFast(w.minLen, t, F3(w.s0.x, w.s0.y, w.s0.z), F3(w.s0.w, w.s1.x, w.s1.y));
}

/* Arrive at the entry `link` names: count the visit, fetch it, change object space if the link says so.  Returns true when the walk
 * ends here (terminator — a visit without a fetch, fragment:208 — or the loop bound of fragment:184: neither).
 * pack: the packed copy; its first ldsSlots slots are also in `lds`.  rays: this thread's pre-transformed rays, 2 float4 per space. */
template <bool COUNT>
FLX_DEV bool w2Arrive(const DeviceScene &sc, const float4 *lds, uint32_t ldsSlots, const float4 *rays, Walk2 &w, int &fl, uint32_t link, WorkCounters &cnt) {
  w.link = link;
  if (link == PK_NOFETCH) return true;
  if (COUNT) { if (fl & QF_CULL) cnt.shadow_visits++; else cnt.closest_visits++; }
  const uint32_t kind = pkKind(link);
  if (kind == 0u) return true;
  const uint32_t slot = link & PK_SLOT;
  const float4 *src = (slot < ldsSlots) ? lds + slot : sc.pack + slot;       /* one generic pointer: flat loads from LDS or from the global copy */
  w.s0 = src[0]; w.s1 = src[1];
  if (kind == 2u) w.s2 = src[2];
  if (FLX_UNLIKELY(link & PK_XF)) {                                          /* the entry stands in another object space than the one before it */
    const float4 *r = rays + 2u * pkT(link);
    const float4 a = r[0], b = r[1];
    w.o = F3(a.x, a.y, a.z); w.d = F3(a.w, b.x, b.y);
    bool fast;
    reciprocalOfDir(sc, w.d, w.o, w.inv, fast);
    fl = fast ? (fl | QF_FAST) : (fl & ~QF_FAST);
  }
  return false;
}

/* One entry for a walking lane.  Returns true when the lane's current walk ended. */
template <bool COUNT>
FLX_DEV bool w2Trip(const DeviceScene &sc, const float4 *lds, uint32_t ldsSlots, const float4 *rays, Walk2 &w, int &fl, WorkCounters &cnt) {
  uint32_t next;
  if (pkKind(w.link) == 1u) {
    const bool hit = w2Box(w, (fl & QF_FAST) != 0);
    next = (uint32_t)__float_as_int(hit ? w.s1.z : w.s1.w);
  } else {
    f3 suv;
    const bool cull = (fl & QF_CULL) != 0;
    Ray r; r.origin = w.o; r.dir = w.d;
    const bool hit = moellerTrumboreAny(F3(w.s0.x, w.s0.y, w.s0.z), F3(w.s0.w, w.s1.x, w.s1.y), F3(w.s1.z, w.s1.w, w.s2.x), r, w.minLen, cull, suv);
    next = (uint32_t)__float_as_int(w.s2.y);
    if (hit) {
      if (cull) { fl |= QF_SHADOWED; return true; }
      if (suv.x != 0.0f) {                                                   /* fragment:217 */
        w.suv = suv; w.tri = __float_as_int(w.s2.z); w.minLen = suv.x;
      }
    }
  }
  return w2Arrive<COUNT>(sc, lds, ldsSlots, rays, w, fl, next, cnt);
}

/* A walk's ray in every object space (fragment:197-202 / :257-262), into this thread's scratch; xf = the inverse transforms in LDS. */
FLX_DEV void w2SetupRays(uint32_t nTransforms, const float4 *xf, float4 *rays, const Ray &src, bool shadowMode) {
  for (uint32_t t = 0; t < nTransforms; t++) {
    const float4 c0 = xf[4 * t], c1 = xf[4 * t + 1], c2 = xf[4 * t + 2], sh = xf[4 * t + 3];
    M3 rotationII; rotationII.c0 = F3(c0.x, c0.y, c0.z); rotationII.c1 = F3(c1.x, c1.y, c1.z); rotationII.c2 = F3(c2.x, c2.y, c2.z);
    const f3 o = mul(rotationII, src.origin + F3(sh.x, sh.y, sh.z));
    f3 d = mul(rotationII, src.dir);
    if (flx_ballot(shadowMode) != 0ull) {
      const f3 dn = normalize(d);
      if (shadowMode) d = dn;                                               /* fragment:261 normalises, fragment:201 does not */
    }
    rays[2 * t] = make_float4(o.x, o.y, o.z, d.x);
    rays[2 * t + 1] = make_float4(d.y, d.z, 0.0f, 0.0f);
  }
}

template <bool COUNT, bool FIRST>
__global__ __launch_bounds__(FLX_W2_THREADS, FLX_W2_WAVES / 2 > 0 ? FLX_W2_WAVES : 1) void k_walk2(DeviceScene sc, DeviceFrame fr, WavefrontBuffers wb, int b, uint32_t total_items,
                                                                                              uint32_t ldsSlots, uint32_t nTransforms, float4 *rayScratch) {
  const uint32_t n = FIRST ? total_items : wb.counts[b];
  if (n == 0u) return;
  const bool compact0 = FIRST && wb.rec0 != nullptr;
  extern __shared__ float4 ldsAll[];
  float4 *ldsTop = ldsAll;
  float4 *ldsXf = ldsAll + ldsSlots;
  for (uint32_t t = threadIdx.x; t < ldsSlots; t += FLX_W2_THREADS) ldsTop[t] = sc.pack[t];
  for (uint32_t t = threadIdx.x; t < nTransforms * 4u; t += FLX_W2_THREADS) {
    const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
    ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
  }
  __syncthreads();
  const uint32_t waveId = blockIdx.x * (FLX_W2_THREADS / 64u) + (threadIdx.x >> 6);
  if (waveId * (64u * FLX_W2_ITEMS_PER_LANE) >= n && waveId != 0u) return;          /* more waves than work */
  const uint32_t *__restrict__ listIn = wb.live[b & 1];
  uint32_t *__restrict__ listOut = wb.live[(b + 1) & 1];
  uint32_t *__restrict__ queue = wb.walkQueue + b;
  uint32_t *__restrict__ outAlloc = wb.counts + (b + 1);
  const uint32_t lane = threadIdx.x & 63u;
  float4 *myRays = rayScratch + ((size_t)blockIdx.x * FLX_W2_THREADS + threadIdx.x) * nTransforms * 2u;
  const uint32_t nWaves = gridDim.x * (FLX_W2_THREADS / 64u);
  uint32_t lastBase = 0;
  uint32_t inChunk = n / (nWaves * FLX_W2_DRAWS_PER_WAVE);
  inChunk = inChunk < 64u ? 64u : (inChunk > WF_IN_CHUNK ? WF_IN_CHUNK : inChunk);
  WorkCounters cnt = {};

  int st = Q_EMPTY;
  int fl = 0;
  uint32_t pathId = 0;
  float base = 0.0f;
  Walk2 w;
  w.o = w.d = w.inv = w.suv = F3(0.0f, 0.0f, 0.0f); w.minLen = 0.0f; w.tri = -1; w.link = PK_NOFETCH;
  w.s0 = w.s1 = w.s2 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool itemsLeft = true;
  uint32_t outBase = 0, outUsed = WF_OUT_CHUNK;
  bool outValid = false;

  /* the record pieces of path `id`: compact at bounce 0 of a frame whose shade0 wrote them, else the 128-byte line */
  auto pixPart = [&](uint32_t id) -> const float4 * {
    uint32_t tile0, s0;
    item_tile(fr, id, tile0, s0);
    return wb.pix0 + (((size_t)tile0 << 6) | (id & 63u)) * 3;
  };

  for (;;) {
    const unsigned long long walking = flx_ballot(st == Q_WALKING);
    const unsigned long long workMask = flx_ballot(st == Q_DONE || st == Q_SWITCH);
    const bool canRefill = itemsLeft || chunkNext != chunkEnd;
    const uint32_t parked = 64u - (uint32_t)__popcll(walking);
    if (walking == 0ull || (parked >= (uint32_t)FLX_W2_BATCH && (workMask != 0ull || canRefill))) {
      /* ---- a shadow walk ended: the path's closest-hit walk is next, unless the loop guard ends the path after this bounce ---- */
      Ray src; src.origin = F3(0.f, 0.f, 0.f); src.dir = src.origin;
      float len = POW32;
      if (st == Q_SWITCH) {
        if (fl & RF_NO_CLOSEST) st = Q_DONE;
        else {
          if (compact0) {
            const float4 p0 = pixPart(pathId)[0], a = wb.rec0[(size_t)pathId * 3];
            src.origin = F3(p0.x, p0.y, p0.z); src.dir = F3(a.x, a.y, a.z);
          } else {
            const float4 *rec = wb.rec + (size_t)pathId * 8;
            const float4 q0 = rec[0], q1 = rec[1];
            src.origin = F3(q0.x, q0.y, q0.z); src.dir = F3(q1.x, q1.y, q1.z);
          }
          fl &= ~QF_CULL;
          st = Q_SETUP;
        }
      }
      /* ---- fold the finished lanes: fragment:445-460, 580, 593-598 and the guard of :475; survivors go to the next round's list ---- */
      if (flx_ballot(st == Q_DONE) != 0ull) {
        bool append = false;
        if (st == Q_DONE) {
          float4 *rec = wb.rec + (size_t)pathId * 8;
          float4 q4, q5, q6, q7;
          const float4 *pp = nullptr;
          if (compact0) {
            pp = pixPart(pathId);
            q4 = wb.rec0[(size_t)pathId * 3 + 2]; q7 = pp[2];
            q5 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); q6 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
          } else {
            q4 = rec[4]; q5 = rec[5]; q6 = rec[6]; q7 = rec[7];
          }
          const bool shadowed = (fl & RF_SHADOWED_NO_WALK) || ((fl & RF_NEED_SHADOW) && (fl & QF_SHADOWED));
          const f3 localColor = shadowed ? F3(base, base, base) : F3(q4.x, q4.y, q4.z);
          const f3 importancy = F3(q6.x, q6.y, q6.z), originalColor = F3(q7.x, q7.y, q7.z);
          const f3 finalColor = F3(q5.x, q5.y, q5.z) + localColor * importancy;
          const int pathBounce = fl >> QF_BOUNCE_SHIFT;
          bool cont = w.tri != -1;
          if (cont) cont = (pathBounce + 1) < fr.max_reflections && length(importancy * originalColor) >= fr.min_importancy * SQRT3;
          if (cont) {
            if (compact0) {                                   /* the path goes on: now it gets its full record (what shade0 would have written) */
              const float4 a = wb.rec0[(size_t)pathId * 3], bq = wb.rec0[(size_t)pathId * 3 + 1], p0 = pp[0];
              rec[0] = make_float4(p0.x, p0.y, p0.z, a.w);
              rec[1] = make_float4(a.x, a.y, a.z, bq.w);
              rec[3] = make_float4(bq.x, bq.y, bq.z, __int_as_float(0));
              rec[6] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
              rec[7] = make_float4(q7.x, q7.y, q7.z, 0.0f);
            }
            rec[5] = make_float4(finalColor.x, finalColor.y, finalColor.z, 0.0f);
            rec[2] = make_float4(w.suv.x, w.suv.y, w.suv.z, __int_as_float(w.tri));
            append = true;
          } else {
            finalize_path(fr, wb, pathId, finalColor, importancy, originalColor);
          }
          st = Q_EMPTY;
        }
        const unsigned long long am = flx_ballot(append);
        if (am != 0ull) {
          const uint32_t cntA = (uint32_t)__popcll(am);
          const uint32_t r = lane_rank(am);
          const uint32_t room = WF_OUT_CHUNK - outUsed;
          const uint32_t seg1 = cntA < room ? cntA : room;
          if (append && r < seg1) listOut[outBase + outUsed + r] = pathId;
          outUsed += seg1;
          if (cntA > seg1) {
            uint32_t nb = 0;
            if (lane == 0) nb = atomicAdd(outAlloc, WF_OUT_CHUNK);
            nb = __builtin_amdgcn_readfirstlane(nb);
            outBase = nb; outValid = true;
            if (append && r >= seg1) listOut[outBase + (r - seg1)] = pathId;
            outUsed = cntA - seg1;
          }
        }
      }
      /* ---- refill the free lanes from the walk queue ---- */
      for (;;) {
        const unsigned long long idle = flx_ballot(st == Q_EMPTY);
        if (idle == 0ull) break;
        if (chunkNext == chunkEnd) {
          if (!itemsLeft) break;
          uint32_t base0 = 0;
          uint32_t want = (n - lastBase) / (nWaves * 2u);                 /* guided self-scheduling: draws shrink as the queue empties */
          want = want < 64u ? 64u : (want > inChunk ? inChunk : want);
          if (lane == 0) base0 = atomicAdd(queue, want);
          base0 = __builtin_amdgcn_readfirstlane(base0);
          if (base0 >= n) { itemsLeft = false; break; }
          lastBase = base0;
          chunkNext = base0;
          chunkEnd = (base0 + want < n) ? base0 + want : n;
        }
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        const uint32_t avail = chunkEnd - chunkNext;
        const uint32_t take = nIdle < avail ? nIdle : avail;
        const uint32_t r = lane_rank(idle);
        if (st == Q_EMPTY && r < take) {
          const uint32_t j = chunkNext + r;
          const uint32_t id = FIRST ? wb.item_base + j : listIn[j];
          if (id != WF_INVALID) {
            float4 q0, q1, q2, q3;
            if (compact0) {
              const float4 *pp = pixPart(id);
              const float4 a = wb.rec0[(size_t)id * 3], bq = wb.rec0[(size_t)id * 3 + 1];
              const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2];
              q0 = make_float4(p0.x, p0.y, p0.z, a.w);
              q1 = make_float4(a.x, a.y, a.z, bq.w);
              q2 = make_float4(p1.x, p1.y, p1.z, p2.w);
              q3 = make_float4(bq.x, bq.y, bq.z, __int_as_float(0));
            } else {
              const float4 *rec = wb.rec + (size_t)id * 8;
              q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3];
            }
            const int rf = __float_as_int(q0.w);
            if (!(rf & RF_DEAD)) {
              pathId = id; base = q2.w;
              fl = (rf & 0xff) | (__float_as_int(q3.w) << QF_BOUNCE_SHIFT);
              w.suv = F3(0.0f, 0.0f, 0.0f); w.tri = -1;
              const bool shadowFirst = (rf & RF_NEED_SHADOW) != 0;
              if (COUNT) { if (shadowFirst) cnt.shadow_walks++; if (!(rf & RF_NO_CLOSEST)) cnt.closest_walks++; }
              if (!shadowFirst && (rf & RF_NO_CLOSEST)) {
                st = Q_DONE;                                             /* nothing to walk: the next pass folds it */
              } else if (shadowFirst) {
                fl |= QF_CULL;
                src.origin = F3(q2.x, q2.y, q2.z); src.dir = F3(q3.x, q3.y, q3.z); len = q1.w;
                st = Q_SETUP;
              } else {
                src.origin = F3(q0.x, q0.y, q0.z); src.dir = F3(q1.x, q1.y, q1.z);
                st = Q_SETUP;
              }
            }
          }
        }
        chunkNext += take;
      }
      /* ---- set up the walks: the ray in every object space, then the root entry ---- */
      if (flx_ballot(st == Q_SETUP) != 0ull) {
        if (st == Q_SETUP) {
          const bool shadowMode = (fl & QF_CULL) != 0;
          w2SetupRays(nTransforms, ldsXf, myRays, src, shadowMode);
          w.o = src.origin; w.d = src.dir; w.minLen = len;                  /* the untransformed ray (cachedTI = 0, fragment:174-175) */
          bool fast;
          reciprocalOfDir(sc, w.d, w.o, w.inv, fast);
          fl = fast ? (fl | QF_FAST) : (fl & ~QF_FAST);
          st = Q_WALKING;
          if (w2Arrive<COUNT>(sc, ldsTop, ldsSlots, myRays, w, fl, sc.pack_root, cnt)) st = shadowMode ? Q_SWITCH : Q_DONE;
        }
      }
      if (flx_ballot(st == Q_WALKING) == 0ull) {
        if (itemsLeft || chunkNext != chunkEnd || flx_ballot(st != Q_EMPTY) != 0ull) continue;
        break;
      }
    }
    /* ---- FLX_W2_INNER entries for every walking lane ---- */
#pragma unroll 1
    for (int it = 0; it < FLX_W2_INNER; it++) {
      if (st == Q_WALKING) {
        if (w2Trip<COUNT>(sc, ldsTop, ldsSlots, myRays, w, fl, cnt)) st = (fl & QF_CULL) ? Q_SWITCH : Q_DONE;
      }
    }
  }
  if (outValid) {
    for (uint32_t t = outUsed + lane; t < WF_OUT_CHUNK; t += 64u) listOut[outBase + t] = WF_INVALID;
  }
  if (COUNT && b == 0 && (cnt.closest_visits | cnt.shadow_visits) != 0u) atomicAdd(wb.counters + 23, (unsigned long long)cnt.closest_visits + cnt.shadow_visits);
  flush_counters<COUNT>(cnt, wb.counters);
}

/* one round's walk kernel; returns false when this scene / frame is not for it (the caller launches k_wf_walk_pre) */
bool launch_walk2(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t compute_units, bool count, int r, uint32_t total,
                  float4 *rayScratch, hipStream_t stream) {
  if (!sc.pack || sc.n_transforms > 8u) return false;
  const uint32_t T = sc.n_transforms;
  const uint32_t perCu = 2u;
  const uint32_t ldsBytesMax = (160u * 1024u) / perCu - 1024u;
  uint32_t ldsSlots = (ldsBytesMax - T * 64u) / 16u;
  if (ldsSlots > sc.pack_hot_slots) ldsSlots = sc.pack_hot_slots;
  const uint32_t ldsBytes = ldsSlots * 16u + T * 64u;
  static bool attrSet = false;
  if (!attrSet) {
    (void)hipFuncSetAttribute((const void *)k_walk2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_walk2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_walk2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_walk2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attrSet = true;
  }
  const uint32_t blocks = compute_units * perCu;
  if (r == 0) {
    if (count) hipLaunchKernelGGL((k_walk2<true, true>), dim3(blocks), dim3(FLX_W2_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsSlots, T, rayScratch);
    else hipLaunchKernelGGL((k_walk2<false, true>), dim3(blocks), dim3(FLX_W2_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsSlots, T, rayScratch);
  } else {
    if (count) hipLaunchKernelGGL((k_walk2<true, false>), dim3(blocks), dim3(FLX_W2_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsSlots, T, rayScratch);
    else hipLaunchKernelGGL((k_walk2<false, false>), dim3(blocks), dim3(FLX_W2_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsSlots, T, rayScratch);
  }
  return true;
}

size_t walk2_scratch_float4(uint32_t compute_units, uint32_t n_transforms) {
  return (size_t)compute_units * 2u * FLX_W2_THREADS * n_transforms * 2u;
}

}  // namespace flx
