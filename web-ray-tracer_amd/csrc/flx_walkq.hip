/*
 * flx_walkq.hip — the bounce walks of the wavefront pipeline under a workgroup-wide test scheduler.
 *
 * A skip-list walk alternates between two very different tests: ray/box (fragment:161-167) on ~75 % of the entries it
 * visits and ray/triangle (fragment:123-158) on the rest.  With one walk per lane (k_wf_walk_pre) nearly every trip
 * of a wave has lanes at both kinds of entry, so the wave runs BOTH tests on every trip with part of its lanes masked
 * (~45 % SIMD efficiency; tests/analysis/walk_sim.py models 3.3-3.9 issue slots per entry against 1.8 with perfect grouping).
 *
 * Here a walk is not tied to a lane.  The state of the walks in flight lives in LDS (96 bytes per walk: next link,
 * minLen, closest hit so far, the ray in world space and in the current object space with 1/dir), and every walk
 * sits in one of four queues by what it needs next:
 *
 *     Q_BOX    test the box its link names            Q_XFORM  move the ray into the object space of the next entry
 *     Q_TRI    test the triangle its link names       Q_NEW    walk over: fold / switch shadow -> closest / refill
 *
 * The 16 waves of the workgroup each take up to 64 walks from the fullest queue, do that ONE thing for all of them,
 * and hand every walk on to the queue its next link selects (links carry the kind of entry they name and whether
 * the transform changes, flx_device.h LINK_*).  Every test therefore runs on (nearly) full waves.  Per walk the
 * entries visited, their order and every arithmetic operation are those of rayTracer / shadowTest; only which wave
 * does which step differs, so the frame is bit-identical to the other pipelines and the work counters are equal.
 *
 * Queues are rings in LDS: a producer reserves cells with one atomic add on the tail and then writes the walk ids,
 * a consumer claims [head, head + n) with a compare-and-swap and takes each id with an atomic exchange that leaves
 * the cell empty (it waits out the few cycles in which a reserved cell is not yet written).  A walk is in exactly
 * one cell or in the hands of exactly one wave, so a ring of capacity >= number of walks never overflows.
 * Every wait is bounded (FLX_WQ_MAX_TRIPS): a kernel that loses a walk raises the error word and ends.
 */
#include <atomic>
#include <cstdio>
#include "flx_wavefront_common.h"

namespace flx {

#ifndef FLX_WQ_THREADS
#define FLX_WQ_THREADS 1024
#endif
#ifndef FLX_WQ_MAX_TRIPS
#define FLX_WQ_MAX_TRIPS (1u << 22)
#endif
#ifndef FLX_WQ_LDS_TOTAL
#define FLX_WQ_LDS_TOTAL (156 * 1024)
#endif

constexpr uint32_t WQ_CAP = 2048u;             /* ring capacity: a power of two >= walks per workgroup */
constexpr uint32_t WQ_NONE = 0xffffffffu;      /* empty ring cell */
constexpr uint32_t WQ_SLOT_F4 = 6u;            /* float4 per walk slot */
enum { Q_BOX = 0, Q_TRI = 1, Q_XFORM = 2, Q_NEW = 3, Q_COUNT = 4 };
/* slot flags; the path record's RF_* bits sit above SF_RF_SHIFT */
constexpr uint32_t SF_CLOSEST = 1u, SF_FAST = 2u, SF_SHADOWED = 4u, SF_START = 8u, SF_ENDED = 16u, SF_EMPTY = 32u;
constexpr int SF_RF_SHIFT = 8;

/* slot: f4[0] = next link, minLen, path id, flags | f4[1] = closest hit s,u,v, entry | f4[2] = world dir.xyz, -
 *       f4[3] = origin.xyz, dir.x | f4[4] = dir.yz, inv.xy | f4[5] = inv.z, world origin.xyz   (origin/dir/inv: current object space) */

struct WqCtl {
  uint32_t head[Q_COUNT];
  uint32_t tail[Q_COUNT];
  unsigned long long chunk;       /* the workgroup's share of the walk queue: end << 32 | next */
  uint32_t lock, dry, retired, lastBase;
  uint32_t pad[18];
};
static_assert(sizeof(WqCtl) == 128, "WqCtl is 8 float4");

FLX_DEV uint32_t ldsLoad(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
FLX_DEV void ldsStore(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
FLX_DEV void wqRelease() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
FLX_DEV void wqAcquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
FLX_DEV uint32_t fbits(float f) { return (uint32_t)__float_as_int(f); }
FLX_DEV float bitsf(uint32_t u) { return __int_as_float((int)u); }

/* Hand the walks of the lanes with `pred` to queue q. */
FLX_DEV void wqPush(WqCtl *ctl, uint32_t *rings, int q, bool pred, uint32_t sid, uint32_t lane) {
  const unsigned long long m = __ballot(pred);
  if (m == 0ull) return;
  uint32_t base = 0;
  if (lane == 0) base = __hip_atomic_fetch_add(&ctl->tail[q], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  base = __builtin_amdgcn_readfirstlane(base);
  if (pred) __hip_atomic_store(rings + q * WQ_CAP + ((base + lane_rank(m)) & (WQ_CAP - 1u)), sid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* Lane 0: take up to `want` path-list positions from the workgroup's chunk, fetching a new chunk from the global walk
 * queue when it is used up.  got == 0 means the queue is dry. */
FLX_DEV void wqDraw(WqCtl *ctl, uint32_t *queue, uint32_t n, uint32_t nGroups, uint32_t want, uint32_t &base, uint32_t &got) {
  base = 0; got = 0;
  for (uint32_t spin = 0; spin < (1u << 20); spin++) {
    unsigned long long c = __hip_atomic_load(&ctl->chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t next = (uint32_t)c, end = (uint32_t)(c >> 32);
    if (next < end) {
      const uint32_t k = want < end - next ? want : end - next;
      unsigned long long nc = ((unsigned long long)end << 32) | (unsigned long long)(next + k);
      if (__hip_atomic_compare_exchange_strong(&ctl->chunk, &c, nc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { base = next; got = k; return; }
      continue;
    }
    if (ldsLoad(&ctl->dry)) return;
    uint32_t expect = 0u;
    if (__hip_atomic_compare_exchange_strong(&ctl->lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
      c = __hip_atomic_load(&ctl->chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((uint32_t)c >= (uint32_t)(c >> 32) && !ldsLoad(&ctl->dry)) {
        /* guided self-scheduling: 1/(2 x workgroups) of what was left at this workgroup's last draw, within [256, 4096] */
        const uint32_t last = ldsLoad(&ctl->lastBase);
        uint32_t g = (n > last ? n - last : 0u) / (nGroups * 2u);
        g = g < 256u ? 256u : (g > 4096u ? 4096u : g);
        const uint32_t b0 = atomicAdd(queue, g);
        if (b0 >= n) ldsStore(&ctl->dry, 1u);
        else {
          const uint32_t e = b0 + g < n ? b0 + g : n;
          ldsStore(&ctl->lastBase, b0);
          __hip_atomic_store(&ctl->chunk, ((unsigned long long)e << 32) | (unsigned long long)b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      __hip_atomic_store(&ctl->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      __builtin_amdgcn_s_sleep(2);
    }
  }
}

/* After a test: store the walk's slot head and queue the walk by its next link. */
template <bool COUNT>
FLX_DEV void wqRoute(WqCtl *ctl, uint32_t *rings, float4 *slots, bool act, uint32_t sid, uint32_t link, float minLen, uint32_t pathId, uint32_t flags,
                     bool ended, WorkCounters &cnt, uint32_t lane) {
  const uint32_t kind = linkKind(link);
  if (COUNT && act && !ended && kind == 0u) { if (flags & SF_CLOSEST) cnt.closest_visits++; else cnt.shadow_visits++; }   /* the fetch of the terminator */
  const bool end = ended || kind == 0u || kind == 3u;
  if (act) {
    if (end) flags |= SF_ENDED;
    slots[sid * WQ_SLOT_F4] = make_float4(bitsf(link), minLen, bitsf(pathId), bitsf(flags));
  }
  wqRelease();
  const bool go = act && !end, xf = (link & LINK_XFORM) != 0u;
  wqPush(ctl, rings, Q_BOX, go && !xf && kind == 1u, sid, lane);
  wqPush(ctl, rings, Q_TRI, go && !xf && kind == 2u, sid, lane);
  wqPush(ctl, rings, Q_XFORM, go && xf, sid, lane);
  wqPush(ctl, rings, Q_NEW, act && end, sid, lane);
}

template <bool COUNT, bool FIRST>
__global__ __launch_bounds__(FLX_WQ_THREADS) void k_wf_walk_queue(DeviceScene sc, DeviceFrame fr, WavefrontBuffers wb, int b, uint32_t total_items, uint32_t nSlots) {
  extern __shared__ float4 ldsAll[];
  WqCtl *ctl = (WqCtl *)ldsAll;
  uint32_t *rings = (uint32_t *)(ldsAll + 8);
  float4 *slots = ldsAll + 8 + (Q_COUNT * WQ_CAP) / 4u;
  const uint32_t lane = threadIdx.x & 63u;
  /* every walk slot starts empty and waiting for a path in Q_NEW */
  for (uint32_t t = threadIdx.x; t < Q_COUNT * WQ_CAP; t += FLX_WQ_THREADS) rings[t] = (t >= Q_NEW * WQ_CAP && t - Q_NEW * WQ_CAP < nSlots) ? t - Q_NEW * WQ_CAP : WQ_NONE;
  for (uint32_t t = threadIdx.x; t < nSlots; t += FLX_WQ_THREADS) slots[t * WQ_SLOT_F4] = make_float4(bitsf(WALK_END), 0.0f, 0.0f, bitsf(SF_EMPTY));
  if (threadIdx.x < sizeof(WqCtl) / 4u) ((uint32_t *)ctl)[threadIdx.x] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) ctl->tail[Q_NEW] = nSlots;
  __syncthreads();

  const uint32_t n = FIRST ? total_items : wb.counts[b];
  const uint32_t *__restrict__ listIn = wb.live[b & 1];
  uint32_t *__restrict__ listOut = wb.live[(b + 1) & 1];
  uint32_t *__restrict__ queue = wb.walkQueue + b;
  uint32_t *__restrict__ outAlloc = wb.counts + (b + 1);
  uint32_t *errWord = wb.walkQueue + (WF_MAX_ROUNDS + 1);
  WorkCounters cnt = {};
  uint32_t outBase = 0, outUsed = WF_OUT_CHUNK;
  bool outValid = false;
  /* scheduler statistics of counted frames (tools/diag_queue.py): per queue batches / walks / cycles, idle trips and cycles */
  unsigned long long dOps[Q_COUNT] = {0, 0, 0, 0}, dLanes[Q_COUNT] = {0, 0, 0, 0}, dCyc[Q_COUNT] = {0, 0, 0, 0}, dIdle = 0, dIdleCyc = 0, dClaimCyc = 0;
  const long long tStart = COUNT ? clock64() : 0;

  for (uint32_t trips = 0;; trips++) {
    const long long tTrip = COUNT ? clock64() : 0;
    if (trips > FLX_WQ_MAX_TRIPS) { if (lane == 0) atomicExch(errWord, 1u + (uint32_t)b); break; }
    /* ---- take up to 64 walks from the fullest queue ------------------------------------------------------ */
    uint32_t cq = Q_COUNT, ch = 0, cn = 0;
    if (lane == 0) {
      for (int attempt = 0; attempt < 4 && cq == Q_COUNT; attempt++) {
        uint32_t best = Q_COUNT, bestC = 0, bestH = 0;
#pragma unroll
        for (int q = 0; q < Q_COUNT; q++) {
          const uint32_t h = ldsLoad(&ctl->head[q]), t = ldsLoad(&ctl->tail[q]);
          const uint32_t c = t - h;
          if ((int)c > 0 && c > bestC) { best = (uint32_t)q; bestC = c; bestH = h; }
        }
        if (best == Q_COUNT) break;
        const uint32_t take = bestC < 64u ? bestC : 64u;
        uint32_t expect = bestH;
        if (__hip_atomic_compare_exchange_strong(&ctl->head[best], &expect, bestH + take, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
          cq = best; ch = bestH; cn = take;
        }
      }
    }
    cq = __builtin_amdgcn_readfirstlane(cq);
    ch = __builtin_amdgcn_readfirstlane(ch);
    cn = __builtin_amdgcn_readfirstlane(cn);
    if (cq == Q_COUNT) {
      uint32_t r = 0;
      if (lane == 0) r = ldsLoad(&ctl->retired);
      r = __builtin_amdgcn_readfirstlane(r);
      if (r >= nSlots) break;                               /* every slot found the walk queue dry: the bounce is done */
      __builtin_amdgcn_s_sleep(8);
      if (COUNT) { dIdle++; dIdleCyc += (unsigned long long)(clock64() - tTrip); }
      continue;
    }
    const bool act = lane < cn;
    uint32_t sid = 0;
    bool lost = false;
    if (act) {
      uint32_t *cell = rings + cq * WQ_CAP + ((ch + lane) & (WQ_CAP - 1u));
      uint32_t spins = 0;
      do {
        sid = __hip_atomic_exchange(cell, WQ_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } while (sid == WQ_NONE && ++spins < (1u << 20));
      lost = sid == WQ_NONE;
    }
    if (__ballot(lost) != 0ull) { if (lane == 0) atomicExch(errWord, 0x100u + (uint32_t)b); break; }
    wqAcquire();
    float4 *slot = slots + (act ? sid : 0u) * WQ_SLOT_F4;
    const long long tOp = COUNT ? clock64() : 0;
    if (COUNT) dClaimCyc += (unsigned long long)(tOp - tTrip);

    if (cq == Q_BOX) {
      /* ---- ray / box: fragment:210-213 (:264-267 in shadowTest) --------------------------------------- */
      uint32_t link = WALK_END, pathId = 0, flags = 0;
      float minLen = 0.0f;
      if (act) {
        const float4 r0 = slot[0], s3 = slot[3], s4 = slot[4];
        const float invz = slot[5].x;
        link = fbits(r0.x); minLen = r0.y; pathId = fbits(r0.z); flags = fbits(r0.w);
        const size_t i = (size_t)linkIndex(link) * 3u;
        const float4 e0 = sc.walk[i], e1 = sc.walk[i + 1], e2 = sc.walk[i + 2];
        if (COUNT) { if (flags & SF_CLOSEST) cnt.closest_visits++; else cnt.shadow_visits++; }
        WalkState w;
        w.tR.origin = F3(s3.x, s3.y, s3.z); w.tR.dir = F3(s3.w, s4.x, s4.y); w.inv = F3(s4.z, s4.w, invz); w.fastDiv = (flags & SF_FAST) != 0u;
        const bool hit = rayCuboidRecip(minLen, w, F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y));
        link = hit ? fbits(e2.x) : fbits(e2.y);
      }
      wqRoute<COUNT>(ctl, rings, slots, act, sid, link, minLen, pathId, flags, false, cnt, lane);
    } else if (cq == Q_TRI) {
      /* ---- ray / triangle: fragment:214-222 (:268-271) ------------------------------------------------- */
      uint32_t link = WALK_END, pathId = 0, flags = 0;
      float minLen = 0.0f;
      bool ended = false;
      if (act) {
        const float4 r0 = slot[0], s3 = slot[3], s4 = slot[4];
        link = fbits(r0.x); minLen = r0.y; pathId = fbits(r0.z); flags = fbits(r0.w);
        const size_t i = (size_t)linkIndex(link) * 3u;
        const float4 e0 = sc.walk[i], e1 = sc.walk[i + 1], e2 = sc.walk[i + 2];
        if (COUNT) { if (flags & SF_CLOSEST) cnt.closest_visits++; else cnt.shadow_visits++; }
        Ray tR; tR.origin = F3(s3.x, s3.y, s3.z); tR.dir = F3(s3.w, s4.x, s4.y);
        const bool cull = (flags & SF_CLOSEST) == 0u;
        f3 suv;
        const bool hit = moellerTrumboreAny(F3(e0.x, e0.y, e0.z), F3(e0.w, e1.x, e1.y), F3(e1.z, e1.w, e2.x), tR, minLen, cull, suv);
        if (hit) {
          if (cull) { flags |= SF_SHADOWED; ended = true; }
          else if (suv.x != 0.0f) {                           /* fragment:217 */
            minLen = suv.x;
            slot[1] = make_float4(suv.x, suv.y, suv.z, e2.w);
          }
        }
        link = fbits(e2.y);
      }
      wqRoute<COUNT>(ctl, rings, slots, act, sid, link, minLen, pathId, flags, ended, cnt, lane);
    } else if (cq == Q_XFORM) {
      /* ---- the ray in the object space of the entry the link names: fragment:194-203 (:254-263) ------------ */
      uint32_t link = WALK_END, pathId = 0, flags = 0;
      float minLen = 0.0f;
      bool need = false;
      if (act) {
        const float4 r0 = slot[0];
        link = fbits(r0.x); minLen = r0.y; pathId = fbits(r0.z); flags = fbits(r0.w);
        const uint32_t kind = linkKind(link);
        need = kind == 1u || kind == 2u;
      }
      const bool shadowLane = need && (flags & SF_CLOSEST) == 0u;
      const bool anyShadow = __ballot(shadowLane) != 0ull;
      if (need) {
        const int meta = ((const int *)sc.walk)[(size_t)linkIndex(link) * 12u + 10u];
        const int t = meta >> 2;
        const float4 s2 = slot[2], s5 = slot[5];
        const f3 wo = F3(s5.y, s5.z, s5.w), wd = F3(s2.x, s2.y, s2.z);
        f3 o, d;
        if ((flags & SF_START) && t == 0) { o = wo; d = wd; }                  /* a walk starts with the ray as it is (fragment:174) */
        else {
          const int iI = 2 * t + 1;
          const M3 rotationII = rotation_at(sc, iI);
          o = mul(rotationII, wo + shift_at(sc, iI));
          d = mul(rotationII, wd);
          if (anyShadow) { const f3 dn = normalize(d); if (shadowLane) d = dn; }      /* fragment:261 normalises, :201 does not */
        }
        f3 inv; bool fast;
        reciprocalOfDir(sc, d, o, inv, fast);
        flags = (flags & ~(SF_FAST | SF_START)) | (fast ? SF_FAST : 0u);
        link &= ~LINK_XFORM;
        slot[3] = make_float4(o.x, o.y, o.z, d.x);
        slot[4] = make_float4(d.y, d.z, inv.x, inv.y);
        slot[5] = make_float4(inv.z, wo.x, wo.y, wo.z);
      } else if (act) {
        flags &= ~SF_START;
        link &= ~LINK_XFORM;
      }
      wqRoute<COUNT>(ctl, rings, slots, act, sid, link, minLen, pathId, flags, false, cnt, lane);
    } else {
      /* ---- walks that ended, and empty slots ----------------------------------------------------------------- */
      uint32_t pathId = 0, flags = act ? 0u : 0u;
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (act) { r0 = slot[0]; pathId = fbits(r0.z); flags = fbits(r0.w); }
      /* shadow walk over and the path ends after this bounce: no closest-hit walk (nextBounceRuns), fold it now with "no hit" */
      if (act && (flags & SF_ENDED) && !(flags & SF_CLOSEST) && ((int)(flags >> SF_RF_SHIFT) & RF_NO_CLOSEST)) flags |= SF_CLOSEST;
      /* closest-hit walk over: fragment:445-460, 580, 593-598 and the guard of :475 */
      const bool foldMe = act && (flags & SF_ENDED) && (flags & SF_CLOSEST);
      if (__ballot(foldMe) != 0ull) {
        bool append = false;
        if (foldMe) {
          const float4 r1 = slot[1];
          float4 *rec = wb.rec + (size_t)pathId * 8;
          const float4 q2 = rec[2], q3 = rec[3], q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
          const int pathBounce = __float_as_int(q3.w);
          const int rf = (int)(flags >> SF_RF_SHIFT);
          const float base = q2.w;
          const bool shadowed = (rf & RF_SHADOWED_NO_WALK) || ((rf & RF_NEED_SHADOW) && (flags & SF_SHADOWED));
          const f3 localColor = shadowed ? F3(base, base, base) : F3(q4.x, q4.y, q4.z);
          const f3 importancy = F3(q6.x, q6.y, q6.z), originalColor = F3(q7.x, q7.y, q7.z);
          const f3 finalColor = F3(q5.x, q5.y, q5.z) + localColor * importancy;
          const int tri = __float_as_int(r1.w);
          bool cont = tri != -1;
          if (cont) cont = (pathBounce + 1) < fr.max_reflections && length(importancy * originalColor) >= fr.min_importancy * SQRT3;
          if (cont) {
            rec[5] = make_float4(finalColor.x, finalColor.y, finalColor.z, 0.0f);
            rec[2] = make_float4(r1.x, r1.y, r1.z, r1.w);
            append = true;
          } else {
            finalize_path(fr, wb, pathId, finalColor, importancy, originalColor);
          }
          flags = SF_EMPTY;
        }
        const unsigned long long am = __ballot(append);
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
      bool start = false, fresh = false;
      f3 wo = F3(0.f, 0.f, 0.f), wd = wo;
      float minLen = 0.0f;
      /* shadow walk over: the path's closest-hit walk follows (the two are independent, fragment:447-449 / :593) */
      if (act && (flags & SF_ENDED) && !(flags & SF_CLOSEST)) {
        const float4 *rec = wb.rec + (size_t)pathId * 8;
        const float4 q0 = rec[0], q1 = rec[1];
        wo = F3(q0.x, q0.y, q0.z); wd = F3(q1.x, q1.y, q1.z); minLen = POW32;
        flags = (flags & ~(SF_ENDED | SF_FAST)) | SF_CLOSEST | SF_START;
        start = true;
      }
      /* empty slots take the next paths of the walk queue */
      for (;;) {
        const bool idleLane = act && (flags & SF_EMPTY) != 0u;
        const unsigned long long idle = __ballot(idleLane);
        if (idle == 0ull) break;
        uint32_t base = 0, got = 0;
        if (lane == 0) wqDraw(ctl, queue, n, gridDim.x, (uint32_t)__popcll(idle), base, got);
        base = __builtin_amdgcn_readfirstlane(base);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got == 0u) break;
        const uint32_t r = lane_rank(idle);
        if (idleLane && r < got) {
          const uint32_t j = base + r;
          const uint32_t id = FIRST ? wb.item_base + j : listIn[j];
          if (id != WF_INVALID) {
            const float4 *rec = wb.rec + (size_t)id * 8;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];     /* one cache line */
            const int fl = __float_as_int(q0.w);
            if (!(fl & RF_DEAD) && !(fl & RF_NEED_SHADOW) && (fl & RF_NO_CLOSEST)) {
              /* nothing to walk (no shadow ray, and the path ends after this bounce): folded here, the slot stays free */
              const float4 q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
              const float base = q2.w;
              const f3 localColor = (fl & RF_SHADOWED_NO_WALK) ? F3(base, base, base) : F3(q4.x, q4.y, q4.z);
              const f3 importancy = F3(q6.x, q6.y, q6.z), originalColor = F3(q7.x, q7.y, q7.z);
              finalize_path(fr, wb, id, F3(q5.x, q5.y, q5.z) + localColor * importancy, importancy, originalColor);
            } else if (!(fl & RF_DEAD)) {
              pathId = id;
              const bool needShadow = (fl & RF_NEED_SHADOW) != 0;
              if (needShadow) { wo = F3(q2.x, q2.y, q2.z); wd = F3(q3.x, q3.y, q3.z); minLen = q1.w; }
              else { wo = F3(q0.x, q0.y, q0.z); wd = F3(q1.x, q1.y, q1.z); minLen = POW32; }
              flags = SF_START | (needShadow ? 0u : SF_CLOSEST) | ((uint32_t)fl << SF_RF_SHIFT);
              if (COUNT) { if (needShadow) cnt.shadow_walks++; if (!(fl & RF_NO_CLOSEST)) cnt.closest_walks++; }
              start = true; fresh = true;
            }
          }
        }
      }
      /* slots still empty found the queue dry: they leave the pool */
      const unsigned long long gone = __ballot(act && (flags & SF_EMPTY) != 0u);
      if (gone != 0ull && lane == 0) __hip_atomic_fetch_add(&ctl->retired, (uint32_t)__popcll(gone), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (start) {
        slot[0] = make_float4(bitsf(sc.walk_root | LINK_XFORM), minLen, bitsf(pathId), bitsf(flags));
        if (fresh) slot[1] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
        slot[2] = make_float4(wd.x, wd.y, wd.z, 0.0f);
        slot[5] = make_float4(0.0f, wo.x, wo.y, wo.z);
      }
      wqRelease();
      wqPush(ctl, rings, Q_XFORM, start, sid, lane);
    }
    if (COUNT) {
#pragma unroll
      for (int q = 0; q < Q_COUNT; q++) if (cq == (uint32_t)q) { dOps[q]++; dLanes[q] += cn; dCyc[q] += (unsigned long long)(clock64() - tOp); }
    }
  }
  if (COUNT && lane == 0 && b == 0) {
    for (int q = 0; q < Q_COUNT; q++) { atomicAdd(wb.counters + 8 + q, dOps[q]); atomicAdd(wb.counters + 12 + q, dLanes[q]); atomicAdd(wb.counters + 16 + q, dCyc[q]); }
    atomicAdd(wb.counters + 20, dIdle); atomicAdd(wb.counters + 21, dIdleCyc); atomicAdd(wb.counters + 22, (unsigned long long)(clock64() - tStart));
    atomicAdd(wb.counters + 23, 1ull); atomicAdd(wb.counters + 24, dClaimCyc);
  }
  if (outValid) {
    for (uint32_t t = outUsed + lane; t < WF_OUT_CHUNK; t += 64u) listOut[outBase + t] = WF_INVALID;
  }
  flush_counters<COUNT>(cnt, wb.counters);
}

/* Walk slots a workgroup holds: what fits beside the rings, at most the ring capacity, whole waves. */
static uint32_t walk_queue_slots() {
  uint32_t s = ((uint32_t)FLX_WQ_LDS_TOTAL - (uint32_t)sizeof(WqCtl) - Q_COUNT * WQ_CAP * 4u) / (WQ_SLOT_F4 * 16u);
  if (s > WQ_CAP) s = WQ_CAP;
  return s & ~63u;
}

void launch_walk_queue(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t compute_units, bool count, int b,
                       uint32_t total, hipStream_t stream) {
  const uint32_t nSlots = walk_queue_slots();
  const uint32_t ldsBytes = (uint32_t)sizeof(WqCtl) + Q_COUNT * WQ_CAP * 4u + nSlots * WQ_SLOT_F4 * 16u;
  static std::atomic<uint64_t> attrDone{0};                  /* (per device: see launch_wavefront) */
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if ((attrDone.fetch_or(bit) & bit) == 0ull) {
    (void)hipFuncSetAttribute((const void *)k_wf_walk_queue<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_wf_walk_queue<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_wf_walk_queue<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_wf_walk_queue<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const dim3 grid(compute_units), block(FLX_WQ_THREADS);
  if (b == 0) {
    if (count) hipLaunchKernelGGL((k_wf_walk_queue<true, true>), grid, block, ldsBytes, stream, sc, fr, wb, b, total, nSlots);
    else hipLaunchKernelGGL((k_wf_walk_queue<false, true>), grid, block, ldsBytes, stream, sc, fr, wb, b, total, nSlots);
  } else {
    if (count) hipLaunchKernelGGL((k_wf_walk_queue<true, false>), grid, block, ldsBytes, stream, sc, fr, wb, b, total, nSlots);
    else hipLaunchKernelGGL((k_wf_walk_queue<false, false>), grid, block, ldsBytes, stream, sc, fr, wb, b, total, nSlots);
  }
}

}  // namespace flx
