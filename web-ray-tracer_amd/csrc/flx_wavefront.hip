/*
 * flx_wavefront.hip — wavefront organisation of the path-trace pass for gfx950 (pipeline 3).
 *
 * The reference runs lightTrace's bounce loop (fragment:475-596) inside one fragment invocation.
 * On a wave64 machine that loop has two very different halves: ~2 k instructions of shading that
 * every lane executes alike, and two skip-list walks whose length varies from a handful of entries
 * (ray leaves the scene) to hundreds (ray grazes the dragon).  Run per lane in lock step, the walk
 * half sat at ~30 % lane utilisation (rocprofv3 SQ_THREAD_CYCLES_VALU, profiles/r01_*).  So the loop
 * is cut at that seam and each half gets the launch shape that suits it:
 *
 *   k_wf_shade(b)  one lane per live path: fragment:476-589 (surface fetch, material, RNG, Fresnel
 *                  choice, light pick, next direction).  Dense list in, every lane busy.
 *   k_wf_walk(b)   persistent waves: each lane walks its path's shadow ray, then its closest-hit ray
 *                  (walkStep, one 48-byte entry per trip).  Lanes whose walks have ended park; once a
 *                  quarter of the wave is parked the wave folds their results (fragment:445-460,580,
 *                  593-598 and the loop guard :475), compacts the survivors into the next bounce's
 *                  live list and refills the free lanes from the queue — wave-level ray compaction
 *                  and restart, so the wave never idles on its longest ray.
 *
 * Path state lives in HBM between the kernels as one 128-byte record per (pixel, sample) path (8
 * float4, written and read whole: one cache line per lane).  Every path does exactly the arithmetic
 * of the sequential shader, in the same order; only the interleaving between paths differs, and each
 * path writes its radiance to its own slot, summed in sample order by k_resolve — so the frame is
 * bit-identical to the per-pixel kernel and to the CPU oracle.
 *
 * Two layouts of the loop over the bounces (launch_wavefront's `organisation`):
 *   rounds        k_wf_shade0, then per bounce k_wf_walk_pre and k_wf_shade: a kernel boundary — and a tail — per bounce;
 *   frame kernel  k_wf_shade0, then k_wf_frame: ONE persistent launch in which the walk waves and the shade waves of a workgroup
 *                 hand paths to each other through rings until every path it drew has ended (round 3; the default up to 64 M paths
 *                 per pass: dragon 1080p 7.87 -> 6.83 ms per frame, a rank's eighth of the frame 2.94 -> 1.67 ms).
 * In front of either: k_primary + k_wf_shade0, or both as one kernel (k_wf_front); or, frame kernel only, nothing — its shade waves trace
 * the primary rays and shade bounce 0 themselves (k_wf_frame<COUNT, true>: dragon 1080p 6.83 -> 6.40 ms).
 */
#include <atomic>
#include <mutex>
#include <cstdio>
#include "flx_kernels.h"
#include "flx_kernel_util.h"
#include "flx_wavefront_common.h"
#include "flx_frame_common.h"

namespace flx {


#ifndef FLX_WF_WALK_THREADS
#define FLX_WF_WALK_THREADS 1024
#endif
#ifndef FLX_WF_DRAWS_PER_WAVE
#define FLX_WF_DRAWS_PER_WAVE 16
#endif
#ifndef FLX_WF_PRETRANSFORM
#define FLX_WF_PRETRANSFORM 1
#endif
#ifndef FLX_WF_LDS_TOTAL
#define FLX_WF_LDS_TOTAL (156 * 1024)         /* LDS a walk workgroup may use (of 160 KB per CU) */
#endif
#ifndef FLX_WF_INNER
#define FLX_WF_INNER 8
#endif
#ifndef FLX_WF_ITEMS_PER_LANE
#define FLX_WF_ITEMS_PER_LANE 4
#endif
#ifndef FLX_WF_CONSOLIDATE
#define FLX_WF_CONSOLIDATE 1                /* k_wf_walk_pre: merge the walks of thinning waves once the queue is dry */
#endif
#ifndef FLX_WF_SPREAD_DIV
#define FLX_WF_SPREAD_DIV 4                 /* tail: deal the walks out evenly once they fill less than 1/DIV of the workgroup's lanes */
#endif
#ifndef FLX_WF_TAIL_TRIPS
#define FLX_WF_TAIL_TRIPS 8                 /* trips of the scheduler loop between two consolidation rounds */
#endif
/* LDS words of the tail consolidation.  The per-round tallies rotate over THREE slots (TC_SLOT + 4 x (round % 3): walks, waves,
 * most walks in a wave, waves with walks): round r adds to slot r % 3 before its barrier and reads it after; after that barrier
 * every wave clears slot (r + 2) % 3 — the one round r - 1 used, which every wave has finished reading before it arrived at
 * round r's barrier, and which nobody adds to before round r + 1's barrier.  (With two slots a fast wave could add to the
 * next round's words before a slow wave had cleared them.) */
enum { TC_LIVE = 0, TC_TAIL = 1, TC_POOL = 2, TC_TAKE = 3, TC_MASK = 4, TC_SLOT = 8, TC_WORDS = 32 };
#ifndef FLX_WF_UNROLL
#define FLX_WF_UNROLL 8                 /* the FLX_WF_INNER trips of the stepping loop unrolled: no loop counter, compare and branch per trip (round 5: dragon 1080p 6.25 -> 6.14 ms, 4K 23.4 -> 23.0;
                                         * 2: 6.20, 4: 6.16; with six trips per check 6.12 but the frame server's eighth 0.94 -> 0.96: profiles/r05_trip_instructions.txt) */
#endif
#ifndef FLX_WF_WAVES_PER_EU
#define FLX_WF_WAVES_PER_EU 4               /* occupancy the register allocation of k_wf_walk_pre must allow */
#endif
#ifndef FLX_WF_SHADE_WAVES
#define FLX_WF_SHADE_WAVES 3                /* waves per SIMD the register allocation of the shade kernels must allow */
#endif
#ifndef FLX_WF_FRONT_WAVES
#define FLX_WF_FRONT_WAVES 3                /* waves per SIMD the register allocation of k_wf_front must allow */
#endif
#ifndef FLX_TAIL_DIAG_ROUND
#define FLX_TAIL_DIAG_ROUND 0
#endif
#ifndef FLX_DIAG_PAD_SALU
#define FLX_DIAG_PAD_SALU 0
#endif
#ifndef FLX_DIAG_PAD_VALU
#define FLX_DIAG_PAD_VALU 0
#endif
#ifndef FLX_EXPERIMENTS
#define FLX_EXPERIMENTS 0                   /* Makefile: EXPERIMENTS=1 adds the queue scheduler, the cooperative finisher and walk suspension */
#endif
#ifndef FLX_FRAME_EARLY_REFILL
#define FLX_FRAME_EARLY_REFILL 1
#endif
#ifndef FLX_WF_FETCH_VBASE
#define FLX_WF_FETCH_VBASE 1
#endif
#ifndef FLX_WF_BATCH
#define FLX_WF_BATCH 24                     /* parked lanes that trigger a fold + refill */
#endif

template <bool COUNT>
__global__ __launch_bounds__(256, FLX_WF_SHADE_WAVES) void k_wf_shade0(FrameArgs /* read through kernel_frame_args() */, uint32_t total_items) {
  const FrameArgsP ab = kernel_frame_args();
  FLX_ARGS_OF(ab);
  const uint32_t S = (uint32_t)fr.samples;
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = t & 63u;
  const uint32_t tileLocal = t >> 6;
  if (tileLocal * S * 64u >= total_items) return;
  const uint32_t tile = wb.item_base / (S * 64u) + tileLocal;
  WorkCounters cnt = {};
  uint32_t px, k;
  tile8_pixel(fr, tile, lane, px, k);
  float4 h = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
  if (px < fr.width && k < fr.rows) h = wb.hits[(size_t)k * fr.width + px];
  (void)shade0_tile<COUNT>(ab, tile, lane, h, cnt);
  flush_counters<COUNT>(cnt, wb.counters);
}

/* k_primary and k_wf_shade0 in one launch: a wave traces the primary rays of its screen tile and shades bounce 0 for it straight away (the hits go through
 * registers; they are still stored for k_resolve), so that the slowest primary ray of the frame holds up its own tile's shading only. */
template <bool COUNT>
__global__ __launch_bounds__(256, FLX_WF_FRONT_WAVES) void k_wf_front(FrameArgs /* read through kernel_frame_args() */, uint32_t total_items) {
  const FrameArgsP ab = kernel_frame_args();
  FLX_ARGS_OF(ab);
  const uint32_t S = (uint32_t)fr.samples;
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = t & 63u;
  const uint32_t tileLocal = t >> 6;
  if (tileLocal * S * 64u >= total_items) return;
  const uint32_t tile = wb.item_base / (S * 64u) + tileLocal;
  WorkCounters cnt = {};
  const float4 h = primary_tile<COUNT>(ab, tile, lane, cnt);
  (void)shade0_tile<COUNT>(ab, tile, lane, h, cnt);
  flush_counters<COUNT>(cnt, wb.counters);
}

/* Later rounds: one lane per live path. */
template <bool COUNT>
__global__ __launch_bounds__(256, FLX_WF_SHADE_WAVES) void k_wf_shade(FrameArgs /* read through kernel_frame_args() */, int b) {
  const FrameArgsP ab = kernel_frame_args();
  FLX_ARGS_OF(ab);
  const uint32_t n = wb.counts[b];
  const uint32_t *__restrict__ listIn = wb.live[b & 1];
  WorkCounters cnt = {};
  for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
    const uint32_t pathId = listIn[j];
    if (pathId == WF_INVALID) continue;
    shade_path<COUNT>(ab, pathId, cnt);
  }
  flush_counters<COUNT>(cnt, wb.counters);
}

enum { L_EMPTY = 0, L_WALKING = 1, L_DONE = 2 };

template <bool COUNT>
__global__ __launch_bounds__(FLX_WF_WALK_THREADS) void k_wf_walk(DeviceScene sc, DeviceFrame fr, WavefrontBuffers wb, int b, uint32_t total_items,
                                                                 uint32_t ldsCount) {
  /* The shallow levels of the tree — the front of the threaded array — are crossed by every ray
   * (2 730 entries take 73 % of all entry fetches of the dragon frame, tests/analysis/visit_histogram.py), so
   * each workgroup keeps them in LDS: ds_read_b128 instead of three divergent 16-byte global loads. */
  extern __shared__ float4 ldsEntries[];
  for (uint32_t t = threadIdx.x; t < ldsCount * 3u; t += FLX_WF_WALK_THREADS) ldsEntries[t] = sc.walk[t];
  __syncthreads();
  const uint32_t n = (b == 0) ? total_items : wb.counts[b];
  const uint32_t waveId = blockIdx.x * (FLX_WF_WALK_THREADS / 64u) + (threadIdx.x >> 6);
  if (waveId * (64u * FLX_WF_ITEMS_PER_LANE) >= n && waveId != 0u) return;   /* more waves than work: leave (a wave wants several items per lane, or its tail dominates) */
  const uint32_t *__restrict__ listIn = wb.live[b & 1];
  uint32_t *__restrict__ listOut = wb.live[(b + 1) & 1];
  uint32_t *__restrict__ queue = wb.walkQueue + b;
  uint32_t *__restrict__ outAlloc = wb.counts + (b + 1);
  const uint32_t lane = threadIdx.x & 63u;
  WorkCounters cnt = {};
  uint32_t diagIters = 0, diagBatches = 0;     /* COUNT builds only: scheduler statistics */
  long long tFold = 0, tRefill = 0, tInner = 0, tStart = COUNT ? clock64() : 0;

  int st = L_EMPTY;
  uint32_t pathId = 0;
  int flags = 0;
  int pathBounce = 0;                      /* the bounce the lane's path is at (its record carries it) */
  float base = 0.0f;
  Ray nextRay; nextRay.origin = F3(0.f, 0.f, 0.f); nextRay.dir = nextRay.origin;
  WalkState w;
  walkClearResults(w);
  w.mode = 2;
  WalkEntry cur;
  cur.e0 = cur.e1 = cur.e2 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t chunkNext = 0, chunkEnd = 0;      /* wave-uniform: ids still to hand out */
  bool itemsLeft = true;
  uint32_t outBase = 0, outUsed = WF_OUT_CHUNK;   /* wave-uniform: reserved live-list slots; none yet */
  bool outValid = false;

  for (;;) {
    const unsigned long long walking = flx_ballot(st == L_WALKING);
    const unsigned long long doneMask = flx_ballot(st == L_DONE);
    const bool canRefill = itemsLeft || chunkNext != chunkEnd;
    const uint32_t parked = 64u - (uint32_t)__popcll(walking);
    if (walking == 0ull || (parked >= (uint32_t)FLX_WF_BATCH && (doneMask != 0ull || canRefill))) {
      if (COUNT) diagBatches++;
      long long t0 = COUNT ? clock64() : 0;
      /* ---- fold the finished lanes: fragment:445-460, 580, 593-598 and the guard of :475 ------------ */
      if (doneMask != 0ull) {
        bool append = false;
        if (st == L_DONE) {
          float4 *rec = wb.rec + (size_t)pathId * 8;
          const float4 q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
          const bool shadowed = (flags & RF_SHADOWED_NO_WALK) || ((flags & RF_NEED_SHADOW) && w.shadowed);
          const f3 localColor = shadowed ? F3(base, base, base) : F3(q4.x, q4.y, q4.z);
          const f3 importancy = F3(q6.x, q6.y, q6.z), originalColor = F3(q7.x, q7.y, q7.z);
          const f3 finalColor = F3(q5.x, q5.y, q5.z) + localColor * importancy;
          bool cont = w.tri != -1;
          if (cont) cont = (pathBounce + 1) < fr.max_reflections && length(importancy * originalColor) >= fr.min_importancy * SQRT3;
          if (cont) {
            rec[5] = make_float4(finalColor.x, finalColor.y, finalColor.z, 0.0f);
            rec[2] = make_float4(w.suv.x, w.suv.y, w.suv.z, __int_as_float(w.tri));
            append = true;
          } else {
            finalize_path(fr, wb, pathId, finalColor, importancy, originalColor);
          }
          st = L_EMPTY;
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
      long long t1 = COUNT ? clock64() : 0;
      if (COUNT) tFold += t1 - t0;
      /* ---- refill the free lanes from the walk queue ------------------------------------------------ */
      for (;;) {
        const unsigned long long idle = flx_ballot(st == L_EMPTY);
        if (idle == 0ull) break;
        if (chunkNext == chunkEnd) {
          if (!itemsLeft) break;
          uint32_t base0 = 0;
          if (lane == 0) base0 = atomicAdd(queue, WF_IN_CHUNK);
          base0 = __builtin_amdgcn_readfirstlane(base0);
          if (base0 >= n) { itemsLeft = false; break; }
          chunkNext = base0;
          chunkEnd = (base0 + WF_IN_CHUNK < n) ? base0 + WF_IN_CHUNK : n;
        }
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        const uint32_t avail = chunkEnd - chunkNext;
        const uint32_t take = nIdle < avail ? nIdle : avail;
        const uint32_t r = lane_rank(idle);
        if (st == L_EMPTY && r < take) {
          const uint32_t j = chunkNext + r;
          const uint32_t id = (b == 0) ? wb.item_base + j : listIn[j];
          if (id != WF_INVALID) {
            const float4 *rec = wb.rec + (size_t)id * 8;
            const float4 q0 = rec[0];
            const int fl = __float_as_int(q0.w);
            if (!(fl & RF_DEAD)) {
              const float4 q1 = rec[1], q2 = rec[2], q3 = rec[3];
              pathId = id; flags = fl; base = q2.w; pathBounce = __float_as_int(q3.w);
              nextRay.origin = F3(q0.x, q0.y, q0.z);
              nextRay.dir = F3(q1.x, q1.y, q1.z);
              walkClearResults(w);
              const bool noClosest = (fl & RF_NO_CLOSEST) != 0;
              if (COUNT && !noClosest) cnt.closest_walks++;
              if (!(fl & RF_NEED_SHADOW) && noClosest) {
                w.mode = 2; st = L_DONE;                       /* nothing to walk: the fold finalises the path */
              } else {
                if (fl & RF_NEED_SHADOW) {
                  Ray sr; sr.origin = F3(q2.x, q2.y, q2.z); sr.dir = F3(q3.x, q3.y, q3.z);
                  walkStartT(sc, w, 0, sr, q1.w);
                  if (COUNT) cnt.shadow_walks++;
                } else {
                  walkStartT(sc, w, 1, nextRay, POW32);
                }
                st = L_WALKING;
                bool ended = walkFetchT<COUNT>(sc, ldsEntries, ldsCount, w, cur, cnt);
                if (ended && w.mode == 0 && !noClosest) { walkStartT(sc, w, 1, nextRay, POW32); ended = walkFetchT<COUNT>(sc, ldsEntries, ldsCount, w, cur, cnt); }
                if (ended) { w.mode = 2; st = L_DONE; }
              }
            }
          }
        }
        chunkNext += take;
      }
      if (COUNT) tRefill += clock64() - t1;
      if (flx_ballot(st == L_WALKING) == 0ull) {
        if (itemsLeft || chunkNext != chunkEnd || flx_ballot(st == L_DONE) != 0ull) continue;      /* (lanes that had nothing to walk wait for the fold) */
        break;
      }
    }
    long long t2 = COUNT ? clock64() : 0;
    /* ---- FLX_WF_INNER entries for every walking lane, without looking at the scheduler state in between
     * (the fold / refill decision above costs ballots and scalar work; amortise it) ------------------- */
#pragma unroll 1
    for (int it = 0; it < FLX_WF_INNER; it++) {
      if (COUNT) diagIters++;
      if (st == L_WALKING) {
        bool ended = false;
        if (walkIsBoxT(cur)) walkBoxT(w, cur); else ended = walkTriT(w, cur);
        if (!ended) ended = walkFetchT<COUNT>(sc, ldsEntries, ldsCount, w, cur, cnt);
        if (ended && w.mode == 0 && !(flags & RF_NO_CLOSEST)) {      /* shadow walk over: start the closest-hit walk at entry 0 */
          walkStartT(sc, w, 1, nextRay, POW32);
          ended = walkFetchT<COUNT>(sc, ldsEntries, ldsCount, w, cur, cnt);
        }
        if (ended) { w.mode = 2; st = L_DONE; }
      }
    }
    if (COUNT) tInner += clock64() - t2;
  }
  /* pad the unused tail of the reserved live-list chunk so the next bounce can skip it */
  if (outValid) {
    for (uint32_t t = outUsed + lane; t < WF_OUT_CHUNK; t += 64u) listOut[outBase + t] = WF_INVALID;
  }
  flush_counters<COUNT>(cnt, wb.counters);
  if (COUNT && lane == 0) {
    atomicAdd(wb.counters + 8 + 2 * (b < 4 ? b : 3), (unsigned long long)diagIters); atomicAdd(wb.counters + 9 + 2 * (b < 4 ? b : 3), (unsigned long long)diagBatches);
    if (b == 0) {   /* in-kernel stamps (diagnostic builds only): shader-clock cycles per phase summed over waves */
      atomicAdd(wb.counters + 16, (unsigned long long)tFold); atomicAdd(wb.counters + 17, (unsigned long long)tRefill);
      atomicAdd(wb.counters + 18, (unsigned long long)tInner); atomicAdd(wb.counters + 19, (unsigned long long)(clock64() - tStart)); atomicAdd(wb.counters + 20, 1ull);
    }
  }
}

/* ---- walk kernel, pre-transformed rays (the default when the scene's transforms fit in LDS) ---------- */

/* FIRST = bounce 0 (identity live list); a template parameter so that the dominant launch of a frame is a kernel symbol of
 * its own in profiler summaries (bench.py's roofline names it). */
template <bool COUNT, bool FIRST>
__global__ __launch_bounds__(FLX_WF_WALK_THREADS, FLX_WF_WAVES_PER_EU) void k_wf_walk_pre(DeviceScene sc, DeviceFrame fr, WavefrontBuffers wb, int b, uint32_t total_items,
                                                                     uint32_t ldsCount, uint32_t nTransforms, uint32_t suspendMax, uint32_t resumePrev) {
  /* b is the ROUND of the bounce loop.  Without suspension round b walks bounce b of every path.  With it (suspendMax > 0) a
   * workgroup that has found the queue dry and is down to suspendMax walks writes them to the straggler list and ends, and
   * the walk kernel of the next round takes them up first: the kernel no longer waits for its longest walk, the paths held
   * up run one round behind the others (every path's record carries its own bounce index) — or, with the cooperative
   * finisher (flx_walkcoop.hip, resumePrev == 0), are completed right after this kernel, a wave per walk. */
#if !FLX_EXPERIMENTS
  suspendMax = 0u; resumePrev = 0u;        /* the shipped library does not suspend walks (Makefile: EXPERIMENTS): constants, so that code folds away */
#endif
  const uint32_t nStrag = (resumePrev && b > 0) ? wb.stragCount[b - 1] : 0u;     /* with the cooperative finisher nothing is carried over */
  const uint32_t nList = FIRST ? total_items : wb.counts[b];
  const bool compact0 = FIRST && wb.rec0 != nullptr;         /* bounce 0 reads the compact records (flx_kernels.h) */
  const uint32_t n = nStrag + nList;                       /* queue positions: stragglers first, then the live list */
  if (n == 0u) return;
  /* LDS: [tree top: ldsCount entries x 48 B][nTransforms x (inverse rotation columns, inverse shift) float4 quadruples]
   *      [tailCtl: 16 words][per thread: nTransforms x 40 B (origin, dir, 1/dir, fast flag)] */
  extern __shared__ float4 ldsAll[];
  float4 *ldsEntries = ldsAll;
  float4 *ldsXf = ldsAll + (size_t)ldsCount * 3u;
  uint32_t *tailCtl = (uint32_t *)(ldsXf + (size_t)nTransforms * 4u);       /* TC_WORDS words, see TC_* */
  float2 *raysBase = (float2 *)(tailCtl + TC_WORDS);
  float2 *myRays = raysBase + (size_t)threadIdx.x * nTransforms * 5u;      /* follows a walk when it moves to another lane (tail consolidation) */
  for (uint32_t t = threadIdx.x; t < ldsCount * 3u; t += FLX_WF_WALK_THREADS) ldsEntries[t] = sc.walk[t];
  for (uint32_t t = threadIdx.x; t < nTransforms * 4u; t += FLX_WF_WALK_THREADS) {
    const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
    ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
  }
  const uint32_t waveId = blockIdx.x * (FLX_WF_WALK_THREADS / 64u) + (threadIdx.x >> 6);
  const bool surplus = waveId * (64u * FLX_WF_ITEMS_PER_LANE) >= n && waveId != 0u;       /* more waves than work */
  if (threadIdx.x < (uint32_t)TC_WORDS) tailCtl[threadIdx.x] = 0u;
  __syncthreads();
  if (!surplus && (threadIdx.x & 63u) == 0u) atomicAdd(&tailCtl[TC_LIVE], 1u);
  __syncthreads();
  if (surplus) return;
  const uint32_t *__restrict__ listIn = wb.live[b & 1];
  uint32_t *__restrict__ listOut = wb.live[(b + 1) & 1];
  uint32_t *__restrict__ queue = wb.walkQueue + b;
  uint32_t *__restrict__ outAlloc = wb.counts + (b + 1);
  const float4 *__restrict__ stragIn = wb.strag[(b + 1) & 1];
  float4 *__restrict__ stragOut = wb.strag[b & 1];
  const uint32_t lane = threadIdx.x & 63u;
  /* ids a wave draws per atomic: large while there is plenty (one atomic per 256 paths), small when the
   * whole queue is only a few draws per wave — the last draws decide how long the kernel's tail is */
  const uint32_t nWaves = gridDim.x * (FLX_WF_WALK_THREADS / 64u);
  uint32_t lastBase = 0;
  uint32_t inChunk = n / (nWaves * FLX_WF_DRAWS_PER_WAVE);
  inChunk = inChunk < 64u ? 64u : (inChunk > WF_IN_CHUNK ? WF_IN_CHUNK : inChunk);
  WorkCounters cnt = {};
  uint32_t diagIters = 0, diagBatches = 0;
  long long tFold = 0, tRefill = 0, tInner = 0, tLoad = 0, tTail = 0, tStart = COUNT ? clock64() : 0;

  int st = P_EMPTY;
  uint32_t pathId = 0;
  int flags = 0;
  int pathBounce = 0;                      /* the bounce the lane's path is at (its record carries it) */
  float base = 0.0f;
  Ray nextRay; nextRay.origin = F3(0.f, 0.f, 0.f); nextRay.dir = nextRay.origin;
  Ray shadowRay = nextRay;
  float shadowLen = 0.0f;
  WalkState w;
  walkClearResults(w);
  w.src = nextRay; w.tR = nextRay; w.minLen = 0.0f; w.i = 0; w.cachedTI = 0;
  w.mode = 2;
  WalkEntry cur;
  cur.e0 = cur.e1 = cur.e2 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool itemsLeft = true;
  uint32_t outBase = 0, outUsed = WF_OUT_CHUNK;
  bool outValid = false;
  bool tailMode = false, tailSynced = false, suspendNow = false;
  uint32_t tailSeen = 13u;                 /* COUNT builds: the tail profile's next threshold (2^(tailSeen - 1)); 13 = queue-dry not yet reported */
  uint32_t tailTrips = 0, tailRound = 0;
  int resumeSt = P_EMPTY;
  float4 *pool = wb.tailPool + (size_t)blockIdx.x * FLX_WF_WALK_THREADS * 8u;

  /* ---- fold the finished lanes (st == P_DONE): fragment:445-460, 580, 593-598 and the guard of :475; survivors go to the next
   * round's live list ---------------------------------------------------------------------------------------------- */
  auto foldDone = [&]() {
      if (flx_ballot(st == P_DONE) != 0ull) {
        bool append = false;
        if (st == P_DONE) {
          float4 *rec = wb.rec + (size_t)pathId * 8;
          float4 q4, q5, q6, q7;
          const float4 *pp = nullptr;
          if (compact0) {                                     /* bounce 0, compact records: lit colour per sample, albedo per pixel */
            uint32_t tile0, s0;
            item_tile(fr, pathId, tile0, s0);
            pp = wb.pix0 + (((size_t)tile0 << 6) | (pathId & 63u)) * 3;
            q4 = wb.rec0[(size_t)pathId * 3 + 2]; q7 = pp[2];
            q5 = make_float4(0.0f, 0.0f, 0.0f, 0.0f); q6 = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
          } else {
            q4 = rec[4]; q5 = rec[5]; q6 = rec[6]; q7 = rec[7];
          }
          const bool shadowed = (flags & RF_SHADOWED_NO_WALK) || ((flags & RF_NEED_SHADOW) && w.shadowed);
          const f3 localColor = shadowed ? F3(base, base, base) : F3(q4.x, q4.y, q4.z);
          const f3 importancy = F3(q6.x, q6.y, q6.z), originalColor = F3(q7.x, q7.y, q7.z);
          const f3 finalColor = F3(q5.x, q5.y, q5.z) + localColor * importancy;
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
          st = P_EMPTY;
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
  };

  for (;;) {
#if FLX_WF_CONSOLIDATE
    const long long tTail0 = COUNT ? clock64() : 0;
    /* ---- tail consolidation --------------------------------------------------------------------------------
     * Once the walk queue is dry a wave only loses lanes, and a SIMD that hosts four quarter-full waves spends four
     * times the issue slots of one full wave on the same walks: the end of the kernel — set by its longest walk — runs
     * at a fraction of the machine's speed.  So when every live wave of the workgroup has found the queue dry, the
     * waves meet every FLX_WF_TAIL_TRIPS trips: if the walks in flight fit in fewer waves, all of them are written to a
     * scratch pool (128 bytes of registers each; the pre-transformed rays stay where they are in LDS and the walk
     * keeps pointing at them), the waves take them back 64 at a time, and the waves left without a walk exit.  A walk
     * only changes lane: its state, entry order and arithmetic are untouched. */
    if (!tailMode && !(itemsLeft || chunkNext != chunkEnd)) {
      tailMode = true;
      if (lane == 0) { atomicAdd(&tailCtl[TC_TAIL], 1u); atomicOr(&tailCtl[TC_MASK], 1u << (threadIdx.x >> 6)); }
    }
    if (tailMode) {
      if (!tailSynced) {
        uint32_t a = 0, l = 1;
        if (lane == 0) { a = __hip_atomic_load(&tailCtl[TC_TAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); l = __hip_atomic_load(&tailCtl[TC_LIVE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        a = __builtin_amdgcn_readfirstlane(a); l = __builtin_amdgcn_readfirstlane(l);
        if (a == l) { tailSynced = true; tailTrips = FLX_WF_TAIL_TRIPS; }
      }
      if (tailSynced && ++tailTrips >= (uint32_t)FLX_WF_TAIL_TRIPS) {
        tailTrips = 0;
        const uint32_t slot = TC_SLOT + 4u * (tailRound % 3u), clr = TC_SLOT + 4u * ((tailRound + 2u) % 3u);
        tailRound++;
        const unsigned long long mine = flx_ballot(st != P_EMPTY);
        const uint32_t myCount = (uint32_t)__popcll(mine);
        if (lane == 0) {
          atomicAdd(&tailCtl[slot], myCount); atomicAdd(&tailCtl[slot + 1u], 1u);
          atomicMax(&tailCtl[slot + 2u], myCount); if (myCount) atomicAdd(&tailCtl[slot + 3u], 1u);
        }
        __syncthreads();
        uint32_t total = 0, waves = 0, most = 0, busy = 0;
        if (lane == 0) {
          total = __hip_atomic_load(&tailCtl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          waves = __hip_atomic_load(&tailCtl[slot + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          most = __hip_atomic_load(&tailCtl[slot + 2u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          busy = __hip_atomic_load(&tailCtl[slot + 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          for (uint32_t k = 0; k < 4u; k++) __hip_atomic_store(&tailCtl[clr + k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      /* the tallies of round + 2 */
        }
        total = __builtin_amdgcn_readfirstlane(total); waves = __builtin_amdgcn_readfirstlane(waves);
        most = __builtin_amdgcn_readfirstlane(most); busy = __builtin_amdgcn_readfirstlane(busy);
        if (COUNT && b == FLX_TAIL_DIAG_ROUND) {            /* tail profile (flx_get_tail_diag): when did this workgroup's walks in flight first number <= 2^k? */
          uint32_t lm = 0;
          if (lane == 0) lm = __hip_atomic_load(&tailCtl[TC_MASK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          lm = __builtin_amdgcn_readfirstlane(lm);
          if (lane == 0 && lm != 0u && (lm & (0u - lm)) == (1u << (threadIdx.x >> 6))) {       /* the lowest live wave reports for the workgroup */
            const unsigned long long now = (unsigned long long)(clock64() - tStart);
            if (tailSeen == 13u) { atomicAdd(wb.counters + 40 + 36, now); atomicAdd(wb.counters + 40 + 37, 1ull); atomicMax(wb.counters + 40 + 38, now); tailSeen = 12u; }
            while (tailSeen > 0u && total <= (1u << (tailSeen - 1u))) {
              tailSeen--;
              atomicAdd(wb.counters + 40 + 3 * tailSeen, now); atomicAdd(wb.counters + 40 + 3 * tailSeen + 1, 1ull); atomicMax(wb.counters + 40 + 3 * tailSeen + 2, now);
            }
          }
        }
        /* Plenty of walks: pack them 64 to a wave (fewer waves issue the same tests).  Few walks (under a quarter of the
         * lanes): what counts is how fast each walk goes, and a lane goes 2-2.5x faster in a wave with one or two walks
         * than in a full one whose lanes are at boxes and triangles at once (tools/diag_lone.py: 787 cycles per entry for
         * one lane, 2 030 for 64): deal them out evenly over the waves, which all stay for that. */
        /* dealt out = one wave per SIMD gets the walks: a SIMD issues one wave instruction per four cycles whatever the wave's
         * lane count, so four thin waves on a SIMD each run at a quarter of the speed of one; the workgroup's waves sit
         * on the SIMDs round robin, the lowest live wave of each residue class is that SIMD's taker */
        const uint32_t myWave = threadIdx.x >> 6;
        uint32_t liveMask = 0;
        if (lane == 0) liveMask = __hip_atomic_load(&tailCtl[TC_MASK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        liveMask = __builtin_amdgcn_readfirstlane(liveMask);
        uint32_t takers = 0, myRank = 0xffffffffu;
        for (uint32_t c = 0; c < 4u; c++) {
          const uint32_t cls = liveMask & (0x1111u << c);
          if (cls) { if (((cls & (0u - cls)) >> myWave) == 1u) myRank = takers; takers++; }
        }
        const bool spread = total * (uint32_t)FLX_WF_SPREAD_DIV < 64u * waves && takers > 0u && total <= 64u * takers;
        const uint32_t share = spread ? (total + takers - 1u) / takers : 64u;
        const bool uneven = spread ? (most > share || busy > takers) : busy > (total + 63u) / 64u;
        auto exportWalks = [&]() {                               /* every walk of the wave -> the workgroup's pool (128 B of registers each) */
          uint32_t pos0 = 0;
          if (lane == 0 && myCount) pos0 = atomicAdd(&tailCtl[TC_POOL], myCount);
          pos0 = __builtin_amdgcn_readfirstlane(pos0);
          if (st != P_EMPTY) {
            float4 *r = pool + (size_t)(pos0 + lane_rank(mine)) * 8u;
            const int packed = st | (w.mode << 4) | ((w.fastDiv ? 1 : 0) << 8) | ((w.shadowed ? 1 : 0) << 9) | (pathBounce << 16);
            r[0] = make_float4(__int_as_float((int)pathId), __int_as_float(flags), base, __int_as_float(packed));
            r[1] = make_float4(nextRay.origin.x, nextRay.origin.y, nextRay.origin.z, nextRay.dir.x);
            r[2] = make_float4(nextRay.dir.y, nextRay.dir.z, w.minLen, __int_as_float(w.i));
            r[3] = make_float4(w.tR.origin.x, w.tR.origin.y, w.tR.origin.z, w.tR.dir.x);
            r[4] = make_float4(w.tR.dir.y, w.tR.dir.z, w.inv.x, w.inv.y);
            r[5] = make_float4(w.inv.z, w.suv.x, w.suv.y, w.suv.z);
            r[6] = make_float4(__int_as_float(w.cachedTI), __int_as_float(w.tri), __int_as_float(w.hitTI), __int_as_float((int)(myRays - raysBase)));
            st = P_EMPTY;
          }
        };
        auto importWalks = [&](uint32_t at, uint32_t share) {      /* pool[at .. at + share) -> lanes 0 .. share - 1 */
          if (at != 0xffffffffu && lane < share && at + lane < total) {
            const float4 *r = pool + (size_t)(at + lane) * 8u;
            const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6];
            pathId = (uint32_t)__float_as_int(r0.x); flags = __float_as_int(r0.y); base = r0.z;
            const int packed = __float_as_int(r0.w);
            st = packed & 15; w.mode = (packed >> 4) & 15; w.fastDiv = ((packed >> 8) & 1) != 0; w.shadowed = ((packed >> 9) & 1) != 0; pathBounce = (packed >> 16) & 0xffff;
            nextRay.origin = F3(r1.x, r1.y, r1.z); nextRay.dir = F3(r1.w, r2.x, r2.y);
            w.minLen = r2.z; w.i = __float_as_int(r2.w);
            w.tR.origin = F3(r3.x, r3.y, r3.z); w.tR.dir = F3(r3.w, r4.x, r4.y); w.inv = F3(r4.z, r4.w, r5.x);
            w.suv = F3(r5.y, r5.z, r5.w);
            w.cachedTI = __float_as_int(r6.x); w.tri = __float_as_int(r6.y); w.hitTI = __float_as_int(r6.z);
            myRays = raysBase + __float_as_int(r6.w);
            if (st == P_WALKING) walkLoadEntry(sc, ldsEntries, ldsCount, (uint32_t)w.i, cur);       /* the entry the walk was about to test */
          }
        };
        if (total <= suspendMax) {
          suspendNow = true;                                   /* every wave of the workgroup reads the same total */
        } else if (waves > 1u && uneven) {
          exportWalks();
          __syncthreads();
          /* import: 64 walks per claim when packing; the SIMDs' takers a share each when dealing out */
          uint32_t at = 0xffffffffu;
          if (spread) { if (myRank != 0xffffffffu) at = myRank * share; }
          else {
            if (lane == 0) at = atomicAdd(&tailCtl[TC_TAKE], share);
            at = __builtin_amdgcn_readfirstlane(at);
          }
          importWalks(at, share);
          __syncthreads();
          if (lane == 0) {
            __hip_atomic_store(&tailCtl[TC_POOL], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&tailCtl[TC_TAKE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    if (COUNT) tTail += clock64() - tTail0;
#endif
    const unsigned long long walking = flx_ballot(st == P_WALKING);
    const unsigned long long workMask = flx_ballot(st == P_DONE || st == P_SWITCH);
    const bool canRefill = itemsLeft || chunkNext != chunkEnd;
    const uint32_t parked = 64u - (uint32_t)__popcll(walking);
    if (suspendNow || walking == 0ull || (parked >= (uint32_t)FLX_WF_BATCH && (workMask != 0ull || canRefill))) {
      if (COUNT) diagBatches++;
      long long t0 = COUNT ? clock64() : 0;
      /* ---- fold the finished lanes: fragment:445-460, 580, 593-598 and the guard of :475 (foldDone, above the loop) ---- */
      foldDone();
      long long t1 = COUNT ? clock64() : 0;
      if (COUNT) tFold += t1 - t0;
      if (suspendNow) {
        /* ---- suspend: the walks still in flight go to the straggler list (the lane's registers; rays and flags are
         * re-read from the path record when the walk is taken up again) --------------------------------------- */
        const bool keep = st == P_WALKING || st == P_SWITCH;
        const unsigned long long km = flx_ballot(keep);
        if (COUNT && lane == 0) {
          atomicAdd(wb.counters + 37, 1ull); atomicAdd(wb.counters + 38, (unsigned long long)__popcll(km));
          atomicMax(wb.counters + 39, (unsigned long long)(clock64() - tStart)); atomicAdd(wb.counters + 36, (unsigned long long)(clock64() - tStart));
        }
        if (km != 0ull) {
          uint32_t pos0 = 0;
          if (lane == 0) pos0 = atomicAdd(wb.stragCount + b, (uint32_t)__popcll(km));
          pos0 = __builtin_amdgcn_readfirstlane(pos0);
          if (keep) {
            float4 *r = stragOut + (size_t)(pos0 + lane_rank(km)) * WF_STRAG_F4;
            const int packed = st | (w.mode << 4) | ((w.fastDiv ? 1 : 0) << 8) | ((w.shadowed ? 1 : 0) << 9);
            r[0] = make_float4(__int_as_float((int)pathId), __int_as_float(packed), w.minLen, __int_as_float(w.i));
            r[1] = make_float4(w.tR.origin.x, w.tR.origin.y, w.tR.origin.z, w.tR.dir.x);
            r[2] = make_float4(w.tR.dir.y, w.tR.dir.z, w.inv.x, w.inv.y);
            r[3] = make_float4(w.inv.z, w.suv.x, w.suv.y, w.suv.z);
            r[4] = make_float4(__int_as_float(w.cachedTI), __int_as_float(w.tri), __int_as_float(w.hitTI), 0.0f);
          }
        }
        break;
      }
      /* ---- refill the free lanes from the walk queue ------------------------------------------------ */
      for (;;) {
        const unsigned long long idle = flx_ballot(st == P_EMPTY);
        if (idle == 0ull) break;
        if (chunkNext == chunkEnd) {
          if (!itemsLeft) break;
          uint32_t base0 = 0;
          /* guided self-scheduling: draw 1/(2 x waves) of what was left at the last draw, within [64, WF_IN_CHUNK] */
          uint32_t want = (n - lastBase) / (nWaves * 2u);
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
        if (st == P_EMPTY && r < take) {
          const uint32_t j = chunkNext + r;
          const bool resume = j < nStrag;                      /* a walk the previous round's kernel suspended */
          float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0, s4 = s0;
          if (resume) { const float4 *sr = stragIn + (size_t)j * WF_STRAG_F4; s0 = sr[0]; s1 = sr[1]; s2 = sr[2]; s3 = sr[3]; s4 = sr[4]; }
          const uint32_t id = resume ? (uint32_t)__float_as_int(s0.x) : (FIRST ? wb.item_base + (j - nStrag) : listIn[j - nStrag]);
          if (id != WF_INVALID) {
            const float4 *rec = wb.rec + (size_t)id * 8;
            float4 q0, q1, q2, q3;
            if (compact0) {                                   /* bounce 0, compact records: the record shade0 would have written, reassembled */
              /* five loads in flight (a dead path's pixel part was never written: whatever is there is not used) */
              uint32_t tile0, s0;
              item_tile(fr, id, tile0, s0);
              const float4 *pp = wb.pix0 + (((size_t)tile0 << 6) | (id & 63u)) * 3;
              const float4 a = wb.rec0[(size_t)id * 3], bq = wb.rec0[(size_t)id * 3 + 1];
              const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2];
              q0 = make_float4(p0.x, p0.y, p0.z, a.w);
              q1 = make_float4(a.x, a.y, a.z, bq.w);
              q2 = make_float4(p1.x, p1.y, p1.z, p2.w);
              q3 = make_float4(bq.x, bq.y, bq.z, __int_as_float(0));
            } else {
              q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3];     /* one cache line, four loads in flight */
            }
            const int fl = __float_as_int(q0.w);
            if (!(fl & RF_DEAD)) {
              pathId = id; flags = fl; base = q2.w; pathBounce = __float_as_int(q3.w);
              nextRay.origin = F3(q0.x, q0.y, q0.z);
              nextRay.dir = F3(q1.x, q1.y, q1.z);
              shadowRay.origin = F3(q2.x, q2.y, q2.z);
              shadowRay.dir = F3(q3.x, q3.y, q3.z);
              shadowLen = q1.w;
              if (resume) {
                const int packed = __float_as_int(s0.y);
                resumeSt = packed & 15; w.mode = (packed >> 4) & 15; w.fastDiv = ((packed >> 8) & 1) != 0; w.shadowed = ((packed >> 9) & 1) != 0;
                w.minLen = s0.z; w.i = __float_as_int(s0.w);
                w.tR.origin = F3(s1.x, s1.y, s1.z); w.tR.dir = F3(s1.w, s2.x, s2.y); w.inv = F3(s2.z, s2.w, s3.x);
                w.suv = F3(s3.y, s3.z, s3.w);
                w.cachedTI = __float_as_int(s4.x); w.tri = __float_as_int(s4.y); w.hitTI = __float_as_int(s4.z);
                st = (resumeSt == P_SWITCH) ? P_SWITCH : P_RESUME;      /* a shadow walk that had ended goes straight to the closest-hit set-up */
              } else {
                walkClearResults(w);
                w.mode = (fl & RF_NEED_SHADOW) ? 0 : 1;
                if (COUNT) { if (w.mode == 0) cnt.shadow_walks++; if (!(fl & RF_NO_CLOSEST)) cnt.closest_walks++; }
                st = (w.mode == 1 && (fl & RF_NO_CLOSEST)) ? P_DONE : P_SETUP;      /* nothing to walk: straight to the fold */
              }
            }
          }
        }
        chunkNext += take;
      }
      /* ---- set up walks: fresh lanes (shadow or closest) and lanes whose shadow walk just ended -------- */
      if (COUNT) { const long long tl = clock64(); tLoad += tl - t1; }
      if (st == P_SWITCH) {
        if (flags & RF_NO_CLOSEST) st = P_DONE;              /* the path ends after this bounce: no closest-hit walk (nextBounceRuns) */
        else { w.mode = 1; st = P_SETUP; }
      }
      if (flx_ballot(st == P_SETUP || st == P_RESUME) != 0ull) {
        if (st == P_SETUP || st == P_RESUME) {
          const bool shadowMode = w.mode == 0;
          const Ray src = shadowMode ? shadowRay : nextRay;
          walkSetupRays(sc, nTransforms, ldsXf, myRays, src, shadowMode);
          if (st == P_SETUP) {
            w.tR = src; w.cachedTI = 0; w.minLen = shadowMode ? shadowLen : POW32; w.i = (int)sc.walk_root;
            reciprocalOfDir(sc, src.dir, src.origin, w.inv, w.fastDiv);      /* the untransformed ray (cachedTI = 0, fragment:174-175) */
            st = P_WALKING;
            if (walkFetchP<COUNT>(sc, ldsEntries, ldsCount, myRays, w, cur, cnt)) st = shadowMode ? P_SWITCH : P_DONE;
          } else {
            /* a suspended walk: its registers came back from the straggler list, the rays in LDS are recomputed (the same
             * arithmetic on the same record), the entry it was about to test is fetched again (not a new visit) */
            st = P_WALKING;
            walkLoadEntry(sc, ldsEntries, ldsCount, (uint32_t)w.i, cur);
          }
        }
      }
      if (COUNT) tRefill += clock64() - t1;
      if (flx_ballot(st == P_WALKING) == 0ull) {
        if (itemsLeft || chunkNext != chunkEnd || flx_ballot(st == P_SWITCH || st == P_DONE) != 0ull) continue;
#if FLX_WF_CONSOLIDATE
        if (tailSynced) continue;          /* walks may be dealt this way at the next round; the round that counts none ends every wave */
#endif
        break;
      }
    }
    long long t2 = COUNT ? clock64() : 0;
    /* ---- FLX_WF_INNER entries for every walking lane ------------------------------------------------- */
#pragma unroll FLX_WF_UNROLL
    for (int it = 0; it < FLX_WF_INNER; it++) {
      if (COUNT) diagIters++;
      if (st == P_WALKING) {
        bool ended = false;
        if (walkIsBoxT(cur)) walkBoxP(w, cur); else ended = walkTriT(w, cur);
        if (!ended) ended = walkFetchP<COUNT>(sc, ldsEntries, ldsCount, myRays, w, cur, cnt);
        if (ended) st = (w.mode == 0) ? P_SWITCH : P_DONE;
      }
#if FLX_DIAG_PAD_SALU        /* diagnostic builds: what do N more scalar / vector instructions per trip cost? (profiles/r02_issue_sensitivity.txt) */
      { uint32_t a = 1, b = 2, c = 3, d = 4;
        for (int k = 0; k < FLX_DIAG_PAD_SALU / 4; k++) asm volatile("s_mov_b32 %0, %1\n s_mov_b32 %1, %2\n s_mov_b32 %2, %3\n s_mov_b32 %3, %0" : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc"); }      /* (s_mov leaves SCC alone; an s_add here without the clobber corrupts the loop's compare and the kernel never ends) */
#endif
#if FLX_DIAG_PAD_VALU
      { uint32_t a = lane, b = lane, c = lane, d = lane;
        for (int k = 0; k < FLX_DIAG_PAD_VALU / 4; k++) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
#endif
    }
    if (COUNT) tInner += clock64() - t2;
  }
#if FLX_WF_CONSOLIDATE
  if (lane == 0) { atomicSub(&tailCtl[TC_LIVE], 1u); if (tailMode) { atomicSub(&tailCtl[TC_TAIL], 1u); atomicAnd(&tailCtl[TC_MASK], ~(1u << (threadIdx.x >> 6))); } }
#endif
  if (outValid) {
    for (uint32_t t = outUsed + lane; t < WF_OUT_CHUNK; t += 64u) listOut[outBase + t] = WF_INVALID;
  }
  /* counted frames: the entries this kernel's walks visited in round 0 (bench.py: algorithmic bytes of the bounce-0 launch) */
  if (COUNT && b == 0 && (cnt.closest_visits | cnt.shadow_visits) != 0u) atomicAdd(wb.counters + 23, (unsigned long long)cnt.closest_visits + cnt.shadow_visits);
  flush_counters<COUNT>(cnt, wb.counters);
  if (COUNT && lane == 0) {
    atomicAdd(wb.counters + 8 + 2 * (b < 4 ? b : 3), (unsigned long long)diagIters); atomicAdd(wb.counters + 9 + 2 * (b < 4 ? b : 3), (unsigned long long)diagBatches);
    const unsigned long long life = (unsigned long long)(clock64() - tStart);
    if (b == 0) {
      atomicAdd(wb.counters + 16, (unsigned long long)tFold); atomicAdd(wb.counters + 17, (unsigned long long)tRefill);
      atomicAdd(wb.counters + 18, (unsigned long long)tInner); atomicAdd(wb.counters + 19, life); atomicAdd(wb.counters + 20, 1ull);
      atomicAdd(wb.counters + 21, (unsigned long long)tLoad); atomicAdd(wb.counters + 22, (unsigned long long)tTail);
    }
    if (b < 4) {   /* tail statistics per bounce: sum / count / max of wave lifetimes */
      atomicAdd(wb.counters + 24 + 3 * b, life); atomicAdd(wb.counters + 25 + 3 * b, 1ull); atomicMax(wb.counters + 26 + 3 * b, life);
    }
  }
}

/* ---- the frame kernel: every bounce of every path in ONE persistent launch (round 3) ---------------------------------------------
 * k_wf_shade / k_wf_walk_pre run the bounce loop as rounds, one kernel pair per bounce, and every walk kernel ends in a tail set by its
 * longest walk: rounds 1 - 3 of the dragon frame spend 25 / 39 / 65 % of their kernels after their queue has run dry, at 0.79 / 0.66 /
 * 0.58 lane utilisation, and a rank's eighth of the frame pays the same four tails for an eighth of the work
 * (profiles/r02_ab_walk_kernel.txt).  The rounds are not a dependency of the algorithm — a path's bounce b + 1 needs ITS bounce b, not
 * everybody's (the shader's loop, fragment:475-596, runs per pixel) — so here the barrier between rounds is gone: a workgroup keeps
 * the paths it has drawn until they end, and its waves are specialised:
 *
 *   walk waves   k_wf_walk_pre's stepping loop unchanged.  A lane whose closest-hit walk found a surface and whose path goes on pushes
 *                the path to the workgroup's shade queue (a ring of path ids in LDS) instead of a global list for the next round; free
 *                lanes are refilled from the workgroup's walk queue first (paths of any bounce that came back from shading), then from
 *                the frame's item queue (fresh bounce-0 paths, whose first shading k_wf_shade0 did for all pixels beforehand).
 *   shade waves  (FLX_FRAME_SHADERS of the 16) pop 64 paths, run one bounce's shading (shade_path: what k_wf_shade runs) densely,
 *                push them to the walk queue.  While the frame's item queue has work they wait for full batches; once it is dry they
 *                take what is there, so that the last paths are not held up.
 *
 * Hand-over between the waves of a workgroup needs no device-wide fence: the record is written with plain stores, a workgroup-scope
 * release / acquire pair around the LDS ring makes it visible to the other waves of the same CU (they share its L1), and nothing
 * crosses workgroups until the kernel ends.  The workgroup counts its live paths in LDS (optimistically at every draw, corrected when
 * an item turns out dead or a path ends); a wave leaves when the item queue is dry and that count is zero.  A draw that would take
 * the count past FQ_ALIVE_MAX is not made, so the rings cannot overflow; and a wave does not draw while FQ_LIMIT paths or more wait
 * for shading (the shade waves are behind: walking more new paths would only lengthen their queue).
 * Per path nothing changes — the same records, the same arithmetic in the same order, its radiance in its own slot — so frames and
 * work counters are bit-identical to the rounds (tests/test_parity_gpu.py: both organisations against the oracle).
 *
 * FRONT = true: the front of the frame is inside the launch as well (profiles/r03_ab_front.txt).  The shade waves — three of them then — also
 * make the fresh paths, one 8 x 8 screen tile at a time (makeTile: primary_tile = what a wave of k_primary does, shade0_tile = the body of
 * k_wf_shade0), and hand them to the walk waves as (tile, sample) units of 64 paths through a third ring; the frame's queue counts screen
 * tiles; k_primary and k_wf_shade0 are not launched.  The fresh paths of a tile stay with the workgroup that made them, so a frame needs
 * enough tiles per workgroup to balance (flx_api.hip: automatic from 32 on); FRONT = false is the kernel described above, register for
 * register. */
#ifndef FLX_FRAME_WALK_PRIO
#define FLX_FRAME_WALK_PRIO 0               /* ... and of the walk waves (shade waves above the walk waves: dragon 1080p 6.37 -> 6.82 ms; profiles/r04_ab_frame_kernel.txt) */
#endif
#ifndef FLX_FRAME_SHADE_PRIO
#define FLX_FRAME_SHADE_PRIO 0              /* issue priority of the shade waves (s_setprio 0 .. 3; the walk waves run at 0) */
#endif
#ifndef FLX_FRAME_SHADERS
#define FLX_FRAME_SHADERS 2                 /* shade waves of a frame-kernel workgroup (dragon 1080p: 1 -> 7.27, 2 -> 6.87, 3 -> 7.16 ms per frame) */
#endif
#ifndef FLX_FRAME_SHADERS_FRONT
#define FLX_FRAME_SHADERS_FRONT 3           /* ... when they also make the fresh paths (WavefrontBuffers::front) */
#endif
#ifndef FLX_FRAME_PROLOGUE_WAVES
#define FLX_FRAME_PROLOGUE_WAVES 2          /* walk waves that make a first tile before their loop (FRONT) */
#endif
#ifndef FLX_FRAME_GROUPS_PER_CU
#define FLX_FRAME_GROUPS_PER_CU 1           /* frame-kernel workgroups per CU (A/B builds with smaller workgroups) */
#endif
#ifndef FLX_FRAME_AUTO_MAX_ITEMS
#define FLX_FRAME_AUTO_MAX_ITEMS (64u << 20)
#endif
#ifndef FLX_FRAME_AUTO_MAX_ITEMS_FRONT
#define FLX_FRAME_AUTO_MAX_ITEMS_FRONT (128u << 20)
#endif
/* (the kernel's body: STAMP — the walk lanes stamp what a path cost, the measure of the adaptive tile order — is its own instantiation, because the stamp's few instructions
 * re-roll the register allocation of the whole kernel: whole frames run the unstamped code, k_wf_frame<COUNT, true>, the commit-before's to the instruction) */
template <bool COUNT, bool FRONT, bool STAMP>
__device__ __forceinline__ void wf_frame_body(uint32_t total_items, uint32_t ldsCount, uint32_t nTransforms, uint32_t shadeWaves, uint32_t readyUnits) {
  const uint32_t n = total_items;
  if (n == 0u) return;
  const FrameArgsP argBase = kernel_frame_args();
#define FLX_FRAME_ARGS() FLX_ARGS_OF(argBase)
  const uint32_t WALK_WAVES = FLX_WF_WALK_THREADS / 64u - shadeWaves;
  uint32_t *frameRings; uint32_t samples; bool compactRecs;
  { FLX_FRAME_ARGS(); frameRings = wb.frameRings; samples = (uint32_t)fr.samples; compactRecs = wb.rec0 != nullptr; }                /* compactRecs: bounce 0 comes with compact records (flx_kernels.h) */
  /* LDS: [tree top][inverse transforms][control words][per walk thread: nTransforms x 40 B of rays]; the two rings of path ids are this
   * workgroup's slice of wb.frameRings (a few lanes touch them per fold / refill, not per trip: they need no LDS) */
  extern __shared__ float4 ldsAll[];
  float4 *ldsEntries = ldsAll;
  float4 *ldsXf = ldsAll + (size_t)ldsCount * 3u;
  uint32_t *ctl = (uint32_t *)(ldsXf + (size_t)nTransforms * 4u);
  uint32_t *shadeRing = frameRings + (size_t)blockIdx.x * WF_FRAME_RINGS * FQ_SIZE, *walkRing = shadeRing + FQ_SIZE, *readyRing = walkRing + FQ_SIZE;
  /* The front of the frame inside the launch (wb.front): the shade waves also make the fresh paths — primary ray and bounce-0 shading of one 8 x 8 screen
   * tile at a time (what k_primary and k_wf_shade0 do in front of the launch otherwise) — and hand them to the walk waves as (tile, sample) units of 64
   * through a third ring; the frame's queue then counts screen tiles. */
  constexpr bool front = FRONT;                /* (a kernel of its own: the front's code costs the other one registers) */
  const uint32_t perTile = samples * 64u;
  float2 *raysBase = (float2 *)(ctl + FC_WORDS);
  {
    FLX_FRAME_ARGS();
    for (uint32_t t = threadIdx.x; t < ldsCount * 3u; t += FLX_WF_WALK_THREADS) ldsEntries[t] = sc.walk[t];
    for (uint32_t t = threadIdx.x; t < nTransforms * 4u; t += FLX_WF_WALK_THREADS) {
      const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
      ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
    }
  }
  if (threadIdx.x < (uint32_t)FC_WORDS) ctl[threadIdx.x] = 0u;
  /* (the rings are WF_INVALID everywhere when the launch begins: allocated so, and a launch clears every slot it pops — flx_api.hip re-initialises them after a device error) */
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  WorkCounters cnt = {};

  /* The front of the frame, one 8 x 8 screen tile: primary rays (the wave walks together), bounce-0 shading, and the tile's (tile, sample) units into
   * the ready ring.  0: not now (the walk waves have enough ready, or the workgroup holds as many paths as its rings take), 1: a tile made, 2: the
   * frame has no more tiles.  Every item of a tile counts as alive from before it is drawn until a walk wave has met it (the dead pixels' items are
   * taken off the count there); a tile none of whose pixels goes on is not handed over at all. */
  const uint32_t nTiles = n / perTile;
  auto makeTile = [&]() -> uint32_t {
    if (!FRONT) return 2u;
    FLX_FRAME_ARGS();
    uint32_t take = 0, tile = 0;
    if (lane == 0 && fq_load(&ctl[FC_RQ + 2]) < readyUnits) {
      const uint32_t before = atomicAdd(&ctl[FC_ALIVE], perTile);
      if (before + perTile > FQ_ALIVE_MAX) atomicSub(&ctl[FC_ALIVE], perTile);
      else { tile = atomicAdd(wb.walkQueue, 1u); take = 1; }
    }
    take = __builtin_amdgcn_readfirstlane(take);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (take == 0u) return 0u;
    if (tile >= nTiles) {
      if (lane == 0) atomicSub(&ctl[FC_ALIVE], perTile);
      return 2u;
    }
    if (wb.tileOrder) tile = __builtin_amdgcn_readfirstlane(wb.tileOrder[tile]);      /* (set only for a launch that owns the whole frame: item_base 0) */
    tile += wb.item_base / perTile;
    const float4 h = primary_tile<COUNT>(argBase, tile, lane, cnt);
    const bool runs = shade0_tile<COUNT>(argBase, tile, lane, h, cnt);
    if (flx_ballot(runs) == 0ull) {
      if (lane == 0) atomicSub(&ctl[FC_ALIVE], perTile);
    } else {
      for (uint32_t s0 = 0; s0 < samples; s0 += 64u)
        fq_push(readyRing, ctl + FC_RQ, s0 + lane < samples, tile * samples + s0 + lane, lane);
    }
    return 1u;
  };

  if (wave >= WALK_WAVES) {
    /* ================================ shade wave ================================ */
    uint32_t idle = 0;
    bool frontDone = !front;                                 /* this wave has found the frame's tile queue dry */
    uint32_t inject; { FLX_FRAME_ARGS(); inject = wb.inject; } asm volatile("" : "+s"(inject));      /* (read once: a scalar load per batch — and its wait — measures 0.3 % of the frame) */
    if (FLX_FRAME_SHADE_PRIO) __builtin_amdgcn_s_setprio(FLX_FRAME_SHADE_PRIO);
    const long long tStartShade = COUNT ? clock64() : 0;
    uint32_t thinMarks = 0;                                  /* COUNT: which of the "paths alive <= 256 / 64 / 16" moments this wave has stamped (flx_get_tail_diag 36 .. 39) */
    for (;;) {
      FLX_FRAME_ARGS();
      const bool dry = fq_load(&ctl[FC_DRY]) != 0u;
      if (COUNT && dry && wave == WALK_WAVES && lane == 0 && thinMarks != 7u) {      /* counted frames: when did the workgroup thin out after its queue ran dry? */
        const uint32_t alive = fq_load(&ctl[FC_ALIVE]);
        const unsigned long long now = (unsigned long long)(clock64() - tStartShade);
        for (uint32_t k = 0; k < 3u; k++)
          if ((thinMarks & (1u << k)) == 0u && alive <= (256u >> (2u * k))) { thinMarks |= 1u << k; atomicAdd(wb.counters + 76 + k, now); }
        if (thinMarks == 7u) atomicAdd(wb.counters + 79, 1ull);
      }
      uint32_t id = WF_INVALID;
      const uint32_t got = fq_pop(shadeRing, ctl + FC_SQ, ~0ull, 64u, dry ? 1u : 64u, lane, id);
      if (got == 0u && !frontDone) {
        /* no full batch to shade: make fresh paths */
        const uint32_t made = makeTile();
        if (made == 2u) {                                     /* the frame has no more tiles: the last shade wave to find that says so */
          if (lane == 0 && atomicAdd(&ctl[FC_FRONT_DONE], 1u) + 1u == shadeWaves) {
            atomicExch(&ctl[FC_DRY], 1u);
            if (COUNT) {                                        /* frame-kernel profile (flx_get_tail_diag 20..), as where the walk waves find the item queue dry */
              const unsigned long long now = (unsigned long long)(clock64() - tStartShade);
              atomicAdd(wb.counters + 60, now); atomicMax(wb.counters + 61, now); atomicAdd(wb.counters + 62, 1ull);
              atomicAdd(wb.counters + 63, (unsigned long long)fq_load(&ctl[FC_ALIVE]));
            }
          }
          frontDone = true;
        }
        if (made != 0u) { idle = 0; continue; }
      }
      if (got == 0u) {
        if (dry && fq_load(&ctl[FC_ALIVE]) == 0u) break;
        if (++idle > (wb.watchdog ? wb.watchdog : FQ_WATCHDOG)) {   /* never in a healthy frame: the host is told (WavefrontBuffers::error); counted builds also leave the control words behind (flx_get_tail_diag) */
          if (lane == 0 && wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_SHADE_WATCHDOG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (COUNT && lane == 0) { for (uint32_t k = 0; k < 8u; k++) atomicMax(wb.counters + 40 + k, (unsigned long long)fq_load(&ctl[k]) + 1ull); atomicAdd(wb.counters + 48, 1ull); }
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      idle = 0;
      const bool mine = lane < got && id != WF_INVALID;
      if (flx_ballot(lane < got && id == WF_INVALID) != 0ull && lane == 0 && wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      /* (a slot that never filled: fq_pop gave up on it) */
      if (inject & WF_INJECT_NO_SHADING) continue;            /* fault injection: the batch is dropped */
      if (mine) shade_path<COUNT>(argBase, id, cnt);
      fq_push(walkRing, ctl + FC_WQ, mine, id, lane);
    }
    { FLX_FRAME_ARGS(); flush_counters<COUNT>(cnt, wb.counters); }
    return;
  }

  /* ================================ walk wave ================================ */
  /* with the front in the kernel nothing is there to walk when it starts: the first walk waves make a tile each as well, before any walk state is
   * live that the tile's code would have to spill around (dragon 1080p by the number of such waves: 0 -> 6.62, 1 -> 6.44, 2 -> 6.42, 3 -> 6.45, all 13 -> 6.55 ms:
   * the more tiles are bound to the workgroup at once, the worse the frame's balance) */
  const long long tStart = COUNT ? clock64() : 0;
  if (front && wave < (uint32_t)FLX_FRAME_PROLOGUE_WAVES) while (makeTile() == 1u) {}
  if (FLX_FRAME_WALK_PRIO) __builtin_amdgcn_s_setprio(FLX_FRAME_WALK_PRIO);
  float2 *myRays = raysBase + (size_t)threadIdx.x * nTransforms * 5u;
  const float4 *walkG; { FLX_FRAME_ARGS(); walkG = pinnedWalkCopy(sc); }      /* the global copy of the tree, for the entries beyond the LDS top */
#if FLX_WF_FETCH_VBASE
  /* both homes of an entry as flat addresses in VECTOR registers: the fetch picks one with two selects; from scalar registers the pick costs four moves more (a VALU
   * instruction reads one scalar operand) and the LDS pointer's conversion to a flat one three scalar instructions, every trip */
  const float4 *ldsEntriesV = ldsEntries, *walkGV = walkG;
  asm volatile("" : "+v"(ldsEntriesV), "+v"(walkGV));
#define FLX_FETCH_G walkGV
#define FLX_FETCH_L ldsEntriesV
#else
#define FLX_FETCH_G walkG
#define FLX_FETCH_L ldsEntries
#endif
  const uint32_t nWaves = gridDim.x * WALK_WAVES;
  uint32_t lastBase = 0;
  uint32_t inChunk = n / (nWaves * FLX_WF_DRAWS_PER_WAVE);
  inChunk = inChunk < 64u ? 64u : (inChunk > WF_IN_CHUNK ? WF_IN_CHUNK : inChunk);

  long long tBlock = 0, tFoldT = 0, tRefillT = 0, tSetupT = 0, tTrips = 0, tAcqT = 0, tFoldOnlyT = 0; unsigned long long nBlocks = 0, nOuter = 0;      /* COUNT builds: where a walk wave's time goes (flx_get_tail_diag 27..33) */
  WalkLane L;                                                  /* the lane's path and its walks (flx_frame_common.h: one body for every persistent kernel) */
  walkLaneInit(L);
  uint32_t chunkNext = 0, chunkEnd = 0;
  bool itemsLeft = true;
  uint32_t idleSpins = 0;

  /* front in its own kernel: fresh paths come from the frame's item queue (guided self-scheduling: draws shrink as the queue empties).  false: none now */
  auto drawChunk = [&]() -> bool {
    FLX_FRAME_ARGS();
    uint32_t *__restrict__ queue = wb.walkQueue;
    if (!itemsLeft) return false;
    if (fq_load(&ctl[FC_SQ + 2]) >= FQ_LIMIT) return false;  /* the shade waves are behind: no new paths for now */
    uint32_t want = (n - lastBase) / (nWaves * 2u);    /* guided self-scheduling: draws shrink as the queue empties */
    want = want < 64u ? 64u : (want > inChunk ? inChunk : want);
    /* The workgroup's live paths are counted BEFORE they are drawn (the count never runs behind) and never exceed FQ_ALIVE_MAX: a
     * path is in one place at a time — a lane, a ring, a shade wave's batch — so neither ring can hold more than that, and a slot
     * is read and cleared before its position comes round again. */
    uint32_t base0 = 0, room = 1;
    if (lane == 0) {
      const uint32_t before = atomicAdd(&ctl[FC_ALIVE], want);
      if (before + want > FQ_ALIVE_MAX) { atomicSub(&ctl[FC_ALIVE], want); room = 0; }
      else base0 = atomicAdd(queue, want);
    }
    room = __builtin_amdgcn_readfirstlane(room);
    if (room == 0u) return false;                              /* as many live paths as the rings hold: no new ones until some end */
    base0 = __builtin_amdgcn_readfirstlane(base0);
    if (base0 >= n) {
      if (lane == 0) {
        atomicSub(&ctl[FC_ALIVE], want);
        const uint32_t was = atomicExch(&ctl[FC_DRY], 1u);
        if (COUNT && was == 0u) {                       /* frame-kernel profile (flx_get_tail_diag 20..): when did this workgroup find the item queue dry, with how many paths alive */
          const unsigned long long now = (unsigned long long)(clock64() - tStart);
          atomicAdd(wb.counters + 60, now); atomicMax(wb.counters + 61, now); atomicAdd(wb.counters + 62, 1ull);
          atomicAdd(wb.counters + 63, (unsigned long long)fq_load(&ctl[FC_ALIVE]));
        }
      }
      itemsLeft = false;
      return false;
    }
    const uint32_t have = (base0 + want < n) ? want : n - base0;
    if (have < want && lane == 0) atomicSub(&ctl[FC_ALIVE], want - have);
    lastBase = base0;
    chunkNext = base0; chunkEnd = base0 + have;
      return true;
  };
  /* the path item at position q of the frame's queue (flx_debug_set_tile_order: the q-th tile of the queue is tile order[q]) */
  auto itemOfQueue = [&](uint32_t q) -> uint32_t {
    FLX_FRAME_ARGS();
    uint32_t id = wb.item_base + q;
    if (wb.tileOrder) { uint32_t tile_, s_; item_tile(fr, id, tile_, s_); id = ((wb.tileOrder[tile_] * (uint32_t)fr.samples + s_) << 6) | (id & 63u); }
    return id;
  };
  for (;;) {
    const unsigned long long walking = flx_ballot(L.st == P_WALKING);
    if (walking != 0ull) idleSpins = 0;
    const unsigned long long workMask = flx_ballot(L.st == P_DONE || L.st == P_SWITCH);
    const uint32_t parked = 64u - (uint32_t)__popcll(walking);
    const bool mayRefill = (front ? fq_load(&ctl[FC_RQ + 2]) != 0u : itemsLeft) || chunkNext != chunkEnd || fq_load(&ctl[FC_WQ + 2]) != 0u;
    if (COUNT) nOuter++;
    const long long tB0 = COUNT ? clock64() : 0;
    if (walking == 0ull || (parked >= (uint32_t)FLX_WF_BATCH && (workMask != 0ull || mayRefill))) {
      if (COUNT) nBlocks++;
      /* ---- with the front inside the launch: the free lanes' NEXT paths first — ids (LDS only), then their records' loads, issued here and not waited for: the fold of the
       * lanes' old paths below has its own loads, and one wait then covers both (a walk wave spent 29 % of its time in this block, 6 800 cycles a time, most of it three
       * memory round trips one after the other: tools/frame_wave_time.py) ---- */
      /* 10 ns ticks: a path's cost is the time it spends in walk lanes (the adaptive tile order's measure; flx_api.hip).  Only where the front ran in its own kernel — thin
       * frames, whose items all workgroups draw 64 paths at a time: there the order is worth 4 % (a rank's eighth 1.47 -> 1.41 ms); with the front inside, the stamp alone
       * cost 2 % of the dragon frame (its wait is one more synchronisation of the block) and the order's gain did not cover the sort */
      const uint32_t stamp = STAMP ? ((uint32_t)__builtin_amdgcn_s_memrealtime() & 0xffffffu) : 0u;
      uint32_t newId = WF_INVALID;
      bool newFresh = false;
      WalkRecord newRec;
      newRec.q0 = newRec.q1 = newRec.q2 = newRec.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (FLX_FRAME_EARLY_REFILL) {
        const bool want = L.st == P_EMPTY || L.st == P_DONE;      /* (a lane at P_DONE is free once it is folded, whatever becomes of its path) */
        for (;;) {
          const unsigned long long idle = flx_ballot(want && newId == WF_INVALID);
          if (idle == 0ull) break;
          FLX_FRAME_ARGS();
          const uint32_t nIdle = (uint32_t)__popcll(idle);
          uint32_t id = WF_INVALID;
          const uint32_t back = fq_pop(walkRing, ctl + FC_WQ, idle, nIdle, 1u, lane, id);
          if (back != 0u) { if (((idle >> lane) & 1ull) != 0ull && id != WF_INVALID) newId = id; if (flx_ballot(((idle >> lane) & 1ull) != 0ull && id != WF_INVALID) == 0ull) break; continue; }
          if (chunkNext == chunkEnd && front) {
            if (fq_load(&ctl[FC_SQ + 2]) >= FQ_LIMIT) break;
            uint32_t unit = WF_INVALID;
            if (fq_pop(readyRing, ctl + FC_RQ, 1ull, 1u, 1u, lane, unit) == 0u) break;
            unit = __builtin_amdgcn_readfirstlane(unit);
            if (unit == WF_INVALID) {
              if (lane == 0) { atomicSub(&ctl[FC_ALIVE], 64u); if (wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
              break;
            }
            chunkNext = unit << 6; chunkEnd = chunkNext + 64u;
          } else if (chunkNext == chunkEnd) {
            if (!drawChunk()) break;
          }
          const uint32_t avail = chunkEnd - chunkNext;
          const uint32_t take = nIdle < avail ? nIdle : avail;
          const uint32_t r = lane_rank(idle);
          if (((idle >> lane) & 1ull) != 0ull && r < take) { newId = front ? chunkNext + r : itemOfQueue(chunkNext + r); newFresh = true; }
          chunkNext += take;
        }
        if (newId != WF_INVALID) { FLX_FRAME_ARGS(); walkLaneFetchRecord(fr, wb, newId, newFresh && compactRecs, newRec); }
      }
      const long long tBa = COUNT ? clock64() : 0; if (COUNT) tAcqT += tBa - tB0;
      /* ---- fold the finished lanes (FLX_WALK_LANE_FOLD); a path that goes on is handed to the shade waves ---- */
      if (flx_ballot(L.st == P_DONE) != 0ull) {
        FLX_FRAME_ARGS();
        bool toShade = false, ended = false;
        if (COUNT && wb.tileCost && L.st == P_DONE) {              /* flx_debug_tile_cost: what this bounce of the lane's path visited, to its screen tile */
          uint32_t tile_, s_; item_tile(fr, L.pathId, tile_, s_);
          atomicAdd(wb.tileCost + tile_, (unsigned long long)(cnt.closest_visits + cnt.shadow_visits - L.v0));
        }
        if (L.st == P_DONE) FLX_WALK_LANE_FOLD(false, fr, wb, compactRecs, L, nullptr, toShade, ended, STAMP, stamp);
        const long long tBf = COUNT ? clock64() : 0; if (COUNT) tFoldOnlyT += tBf - tBa;
        fq_push(shadeRing, ctl + FC_SQ, toShade, L.pathId, lane);
        const uint32_t nEnded = (uint32_t)__popcll(flx_ballot(ended));
        if (nEnded != 0u && lane == 0) atomicSub(&ctl[FC_ALIVE], nEnded);
      }
      const long long tB1 = COUNT ? clock64() : 0; if (COUNT) tFoldT += tB1 - tB0;
      if (FLX_FRAME_EARLY_REFILL) {
        /* ... the lanes' next paths (their records arrived while the old ones were folded) */
        bool dead = false;
        if (newId != WF_INVALID) dead = walkLaneInstall<COUNT>(newId, newRec, L, cnt, stamp);
        const uint32_t nDead = (uint32_t)__popcll(flx_ballot(dead));
        if (nDead != 0u && lane == 0) atomicSub(&ctl[FC_ALIVE], nDead);
      } else
      /* ---- refill the free lanes: paths that came back from shading first, then fresh ones from the frame's item queue ---- */
      for (;;) {
        const unsigned long long idle = flx_ballot(L.st == P_EMPTY);
        if (idle == 0ull) break;
        FLX_FRAME_ARGS();
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        uint32_t id = WF_INVALID;
        bool fresh = false;                                    /* a bounce-0 item: compact record, may be dead */
        const uint32_t back = fq_pop(walkRing, ctl + FC_WQ, idle, nIdle, 1u, lane, id);
        if (back == 0u) {
          if (chunkNext == chunkEnd && front) {
            /* fresh paths come from this workgroup's shade waves: one (tile, sample) unit of 64 at a time (they are counted alive already) */
            if (fq_load(&ctl[FC_SQ + 2]) >= FQ_LIMIT) break;
            uint32_t unit = WF_INVALID;
            if (fq_pop(readyRing, ctl + FC_RQ, 1ull, 1u, 1u, lane, unit) == 0u) break;
            unit = __builtin_amdgcn_readfirstlane(unit);
            if (unit == WF_INVALID) {                          /* the slot never filled (fq_pop's own watchdog): no unit, its 64 paths are written off, the host is told */
              if (lane == 0) { atomicSub(&ctl[FC_ALIVE], 64u); if (wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
              break;
            }
            chunkNext = unit << 6; chunkEnd = chunkNext + 64u;
          } else if (chunkNext == chunkEnd) {
            if (!drawChunk()) break;
          }
          const uint32_t avail = chunkEnd - chunkNext;
          const uint32_t take = nIdle < avail ? nIdle : avail;
          const uint32_t r = lane_rank(idle);
          if (L.st == P_EMPTY && r < take) { id = front ? chunkNext + r : itemOfQueue(chunkNext + r); fresh = true; }
          chunkNext += take;
        }
        bool dead = false;
        if (id != WF_INVALID) dead = walkLaneLoad<COUNT>(fr, wb, id, fresh && compactRecs, L, cnt, stamp);
        const uint32_t nDead = (uint32_t)__popcll(flx_ballot(dead));
        if (nDead != 0u && lane == 0) atomicSub(&ctl[FC_ALIVE], nDead);
      }
      const long long tB2 = COUNT ? clock64() : 0; if (COUNT) tRefillT += tB2 - tB1;
      /* ---- set up walks: fresh lanes (shadow or closest) and lanes whose shadow walk just ended ---- */
      walkLaneSwitch(L);
      if (flx_ballot(L.st == P_SETUP) != 0ull) {
        FLX_FRAME_ARGS();
        if (L.st == P_SETUP) walkLaneSetup<COUNT>(sc, nTransforms, ldsXf, myRays, FLX_FETCH_G, FLX_FETCH_L, ldsCount, L, cnt);
      }
      if (COUNT) { const long long tB3 = clock64(); tSetupT += tB3 - tB2; tBlock += tB3 - tB0; }
      if (flx_ballot(L.st == P_WALKING) == 0ull) {
        if (flx_ballot(L.st != P_EMPTY) != 0ull) continue;      /* lanes that had nothing to walk wait for the fold */
        /* nothing in this wave: done when the item queue is dry and no path of the workgroup is alive; else wait for the shade waves */
        if (front) {
          if (fq_load(&ctl[FC_DRY]) != 0u && chunkNext == chunkEnd && fq_load(&ctl[FC_ALIVE]) == 0u) break;
        } else if (!itemsLeft && chunkNext == chunkEnd && fq_load(&ctl[FC_ALIVE]) == 0u) break;
        if (fq_load(&ctl[FC_WQ + 2]) == 0u && (front ? (fq_load(&ctl[FC_RQ + 2]) == 0u || fq_load(&ctl[FC_SQ + 2]) >= FQ_LIMIT)
                                                     : !(itemsLeft && fq_load(&ctl[FC_SQ + 2]) < FQ_LIMIT && fq_load(&ctl[FC_ALIVE]) + 256u <= FQ_ALIVE_MAX))) {
          FLX_FRAME_ARGS();
          if (++idleSpins > (wb.watchdog ? wb.watchdog : FQ_WATCHDOG)) {
            if (lane == 0 && wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_WALK_WATCHDOG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (COUNT && lane == 0) { for (uint32_t k = 0; k < 8u; k++) atomicMax(wb.counters + 50 + k, (unsigned long long)fq_load(&ctl[k]) + 1ull); atomicAdd(wb.counters + 58, 1ull); }
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        continue;
      }
    }
    /* ---- FLX_WF_INNER entries for every walking lane (the one scene word the fetch needs — the global copy's address — is in registers: walkG) ---- */
    const long long tT0 = COUNT ? clock64() : 0;
    {
#pragma unroll FLX_WF_UNROLL
      for (int it = 0; it < FLX_WF_INNER; it++) FLX_WALK_LANE_STEP(COUNT, FLX_FETCH_G, FLX_FETCH_L, ldsCount, myRays, L, cnt);
    }
    if (COUNT) tTrips += clock64() - tT0;
  }
  FLX_FRAME_ARGS();
  if (COUNT && (cnt.closest_visits | cnt.shadow_visits) != 0u) atomicAdd(wb.counters + 23, (unsigned long long)cnt.closest_visits + cnt.shadow_visits);
  if (COUNT && lane == 0) {                                     /* wave lifetimes: sum / max / count (flx_get_tail_diag 24..26) */
    const unsigned long long life = (unsigned long long)(clock64() - tStart);
    atomicAdd(wb.counters + 64, life); atomicMax(wb.counters + 65, life); atomicAdd(wb.counters + 66, 1ull);
    atomicAdd(wb.counters + 67, (unsigned long long)tBlock); atomicAdd(wb.counters + 68, (unsigned long long)tFoldT); atomicAdd(wb.counters + 69, (unsigned long long)tRefillT); atomicAdd(wb.counters + 70, (unsigned long long)tSetupT);
    atomicAdd(wb.counters + 71, (unsigned long long)tTrips); atomicAdd(wb.counters + 72, nBlocks); atomicAdd(wb.counters + 73, nOuter);
    atomicAdd(wb.counters + 74, (unsigned long long)tAcqT); atomicAdd(wb.counters + 75, (unsigned long long)tFoldOnlyT);
  }
  flush_counters<COUNT>(cnt, wb.counters);
#undef FLX_FETCH_G
#undef FLX_FETCH_L
#undef FLX_FRAME_ARGS
}
template <bool COUNT, bool FRONT>
__global__ __launch_bounds__(FLX_WF_WALK_THREADS, FLX_WF_WAVES_PER_EU) void k_wf_frame(FrameArgs /* read through kernel_frame_args() */, uint32_t total_items,
                                                                                    uint32_t ldsCount, uint32_t nTransforms, uint32_t shadeWaves, uint32_t readyUnits) {
  wf_frame_body<COUNT, FRONT, !FRONT>(total_items, ldsCount, nTransforms, shadeWaves, readyUnits);      /* (front in its own kernel: thin frames, always stamped) */
}
/* the frame kernel with its front inside AND the cost stamps: frames of fewer than FLX_ADAPTIVE_FRONT_MAX_TILES_PER_CU screen tiles per workgroup (a rank's quarter) */
template <bool COUNT>
__global__ __launch_bounds__(FLX_WF_WALK_THREADS, FLX_WF_WAVES_PER_EU) void k_wf_frame_stamped(FrameArgs /* read through kernel_frame_args() */, uint32_t total_items,
                                                                                            uint32_t ldsCount, uint32_t nTransforms, uint32_t shadeWaves, uint32_t readyUnits) {
  wf_frame_body<COUNT, true, true>(total_items, ldsCount, nTransforms, shadeWaves, readyUnits);
}

/* ---- the frame kernel with TWO WALK JOBS PER LANE (round 5; profiles/r05_two_walks.txt) ------------------------------------------------------------
 * What bounds k_wf_frame is instruction issue with half of the lanes masked (VALU issue 0.48 priced as FMAs, 0.68 by class; lane utilisation 0.54): nearly every
 * trip of a walk wave has lanes at boxes AND lanes at triangles, so both test bodies run, each for part of the lanes, and the ~25 % of the lanes that stand at a
 * triangle pay for a 97-instruction body.  Making the lanes wait for company lost (profiles/r01_ab_deferred_triangles.txt: a lane that waits lengthens its chain).
 * Here a lane holds TWO independent jobs (each what a lane of k_wf_frame holds: a path's shadow walk, then its closest-hit walk), and a trip has two phases:
 *   box phase       every lane with a job at a box advances ONE such job (the one it did not advance last, when both are);
 *   triangle phase  only when at least FLX_FRAME2_TRI_MIN lanes have a job at a triangle (or no lane is at a box, or they have waited FLX_FRAME2_TRI_WAIT trips):
 *                   every such lane advances one — a lane that waits at a triangle with one job goes on at boxes with the other.
 * Links carry nothing new: a job's next entry is fetched right after its step, as before; the operands of a phase are picked from the two jobs with selects.
 * Per path nothing changes (entries, order, arithmetic, counters).  Half the waves: 512-thread workgroups at two waves per SIMD and 256 VGPRs — the LDS rays of a
 * workgroup stay what they are (2 jobs x 384 walk threads x T x 40 B), and the front's 164 - 167 live values fit (no spills).
 * tests/analysis/walk_sim2.py replays the oracle's traces through this policy: -14 % instructions per visit over the dragon frame's bounces.
 * MEASURED AND LOST (profiles/r05_two_walks.txt): bit-identical, 14.07 ms against 6.40 for the dragon's 1080p frame.  The compiler's trip is 872 instructions (553 vector,
 * 225 scalar: selects of the operands, masks around four arms) where k_wf_frame's is 401 for 1.35 steps per lane instead of 1, and two waves per SIMD do not hide the
 * fetches.  Not in the shipped library: `make EXPERIMENTS=1` carries it (flx_debug_set_walk_jobs(ctx, 2)), tests run it under the `experiments` marker. */
#if FLX_EXPERIMENTS

#ifndef FLX_FRAME2_THREADS
#define FLX_FRAME2_THREADS 512
#endif
#ifndef FLX_FRAME2_WAVES_PER_EU
#define FLX_FRAME2_WAVES_PER_EU 2
#endif
#ifndef FLX_FRAME2_SHADERS
#define FLX_FRAME2_SHADERS 2                /* shade waves of the eight */
#endif
#ifndef FLX_FRAME2_TRI_MIN
#define FLX_FRAME2_TRI_MIN 20
#endif
#ifndef FLX_FRAME2_TRI_WAIT
#define FLX_FRAME2_TRI_WAIT 4
#endif
#ifndef FLX_FRAME2_BATCH
#define FLX_FRAME2_BATCH 48                 /* parked jobs (of a wave's 128) that trigger a fold + refill */
#endif
#ifndef FLX_FRAME2_PROLOGUE_WAVES
#define FLX_FRAME2_PROLOGUE_WAVES 1
#endif
template <typename T> FLX_DEV T selv(bool c, T a, T b) { return c ? a : b; }      /* operands by VALUE: a select of two loaded values, never of two addresses (that would put the jobs in scratch memory) */
struct WalkJob : WalkLane { float2 *rays; };      /* a lane's job: what a lane of k_wf_frame holds (flx_frame_common.h) and where its pre-transformed rays live */
template <bool COUNT>
__global__ __launch_bounds__(FLX_FRAME2_THREADS, FLX_FRAME2_WAVES_PER_EU) void k_wf_frame2(FrameArgs /* read through argBase */, uint32_t total_items,
                                                                                       uint32_t ldsCount, uint32_t nTransforms, uint32_t shadeWaves, uint32_t readyUnits) {
  const uint32_t n = total_items;
  if (n == 0u) return;
  const FrameArgsP argBase = kernel_frame_args();
#define FLX_FRAME_ARGS() FLX_ARGS_OF(argBase)
  const uint32_t WALK_WAVES = FLX_FRAME2_THREADS / 64u - shadeWaves;
  uint32_t *frameRings; uint32_t samples; bool compactRecs;
  { FLX_FRAME_ARGS(); frameRings = wb.frameRings; samples = (uint32_t)fr.samples; compactRecs = wb.rec0 != nullptr; }
  /* LDS: [tree top][inverse transforms][control words][per walk thread: 2 jobs x nTransforms x 40 B of rays] */
  extern __shared__ float4 ldsAll[];
  float4 *ldsEntries = ldsAll;
  float4 *ldsXf = ldsAll + (size_t)ldsCount * 3u;
  uint32_t *ctl = (uint32_t *)(ldsXf + (size_t)nTransforms * 4u);
  uint32_t *shadeRing = frameRings + (size_t)blockIdx.x * WF_FRAME_RINGS * FQ_SIZE, *walkRing = shadeRing + FQ_SIZE, *readyRing = walkRing + FQ_SIZE;
  const uint32_t perTile = samples * 64u;
  float2 *raysBase = (float2 *)(ctl + FC_WORDS);
  {
    FLX_FRAME_ARGS();
    for (uint32_t t = threadIdx.x; t < ldsCount * 3u; t += FLX_FRAME2_THREADS) ldsEntries[t] = sc.walk[t];
    for (uint32_t t = threadIdx.x; t < nTransforms * 4u; t += FLX_FRAME2_THREADS) {
      const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
      ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
    }
  }
  if (threadIdx.x < (uint32_t)FC_WORDS) ctl[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  WorkCounters cnt = {};

  const uint32_t nTiles = n / perTile;
  auto makeTile = [&]() -> uint32_t {                          /* k_wf_frame's makeTile, word for word */
    FLX_FRAME_ARGS();
    uint32_t take = 0, tile = 0;
    if (lane == 0 && fq_load(&ctl[FC_RQ + 2]) < readyUnits) {
      const uint32_t before = atomicAdd(&ctl[FC_ALIVE], perTile);
      if (before + perTile > FQ_ALIVE_MAX) atomicSub(&ctl[FC_ALIVE], perTile);
      else { tile = atomicAdd(wb.walkQueue, 1u); take = 1; }
    }
    take = __builtin_amdgcn_readfirstlane(take);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (take == 0u) return 0u;
    if (tile >= nTiles) {
      if (lane == 0) atomicSub(&ctl[FC_ALIVE], perTile);
      return 2u;
    }
    tile += wb.item_base / perTile;
    const float4 h = primary_tile<COUNT>(argBase, tile, lane, cnt);
    const bool runs = shade0_tile<COUNT>(argBase, tile, lane, h, cnt);
    if (flx_ballot(runs) == 0ull) {
      if (lane == 0) atomicSub(&ctl[FC_ALIVE], perTile);
    } else {
      for (uint32_t s0 = 0; s0 < samples; s0 += 64u)
        fq_push(readyRing, ctl + FC_RQ, s0 + lane < samples, tile * samples + s0 + lane, lane);
    }
    return 1u;
  };

  if (wave >= WALK_WAVES) {
    /* ================================ shade wave (as in k_wf_frame<COUNT, true>) ================================ */
    uint32_t idle = 0;
    bool frontDone = false;
    uint32_t inject; { FLX_FRAME_ARGS(); inject = wb.inject; } asm volatile("" : "+s"(inject));
    const long long tStartShade = COUNT ? clock64() : 0;
    for (;;) {
      FLX_FRAME_ARGS();
      const bool dry = fq_load(&ctl[FC_DRY]) != 0u;
      uint32_t id = WF_INVALID;
      const uint32_t got = fq_pop(shadeRing, ctl + FC_SQ, ~0ull, 64u, dry ? 1u : 64u, lane, id);
      if (got == 0u && !frontDone) {
        const uint32_t made = makeTile();
        if (made == 2u) {
          if (lane == 0 && atomicAdd(&ctl[FC_FRONT_DONE], 1u) + 1u == shadeWaves) {
            atomicExch(&ctl[FC_DRY], 1u);
            if (COUNT) {
              const unsigned long long now = (unsigned long long)(clock64() - tStartShade);
              atomicAdd(wb.counters + 60, now); atomicMax(wb.counters + 61, now); atomicAdd(wb.counters + 62, 1ull);
              atomicAdd(wb.counters + 63, (unsigned long long)fq_load(&ctl[FC_ALIVE]));
            }
          }
          frontDone = true;
        }
        if (made != 0u) { idle = 0; continue; }
      }
      if (got == 0u) {
        if (dry && fq_load(&ctl[FC_ALIVE]) == 0u) break;
        if (++idle > (wb.watchdog ? wb.watchdog : FQ_WATCHDOG)) {
          if (lane == 0 && wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_SHADE_WATCHDOG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (COUNT && lane == 0) { for (uint32_t k = 0; k < 8u; k++) atomicMax(wb.counters + 40 + k, (unsigned long long)fq_load(&ctl[k]) + 1ull); atomicAdd(wb.counters + 48, 1ull); }
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      idle = 0;
      const bool mine = lane < got && id != WF_INVALID;
      if (flx_ballot(lane < got && id == WF_INVALID) != 0ull && lane == 0 && wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (inject & WF_INJECT_NO_SHADING) continue;
      if (mine) shade_path<COUNT>(argBase, id, cnt);
      fq_push(walkRing, ctl + FC_WQ, mine, id, lane);
    }
    { FLX_FRAME_ARGS(); flush_counters<COUNT>(cnt, wb.counters); }
    return;
  }

  /* ================================ walk wave: two jobs per lane ================================ */
  const long long tStart = COUNT ? clock64() : 0;
  if (wave < (uint32_t)FLX_FRAME2_PROLOGUE_WAVES) while (makeTile() == 1u) {}
  const float4 *walkG; { FLX_FRAME_ARGS(); walkG = pinnedWalkCopy(sc); }
  WalkJob j0, j1;
  walkLaneInit(j0); walkLaneInit(j1);
  j0.rays = raysBase + (size_t)threadIdx.x * 2u * nTransforms * 5u;      /* (walk waves are the first waves of the workgroup) */
  j1.rays = j0.rays + (size_t)nTransforms * 5u;
  uint32_t chunkNext = 0, chunkEnd = 0;
  uint32_t idleSpins = 0;
  uint32_t last = 0;                                           /* per lane: the job advanced last (the other one's entry has had the longer time to arrive) */
  uint32_t triWait = 0;                                        /* wave-uniform: trips the lanes at triangles have waited */
  unsigned long long diagBoxTrips = 0, diagBoxLanes = 0, diagTriTrips = 0, diagTriLanes = 0;      /* COUNT builds */

  /* ---- fold a job's finished path (FLX_WALK_LANE_FOLD); a path that goes on is handed to the shade waves ---- */
  auto foldJob = [&](WalkJob &J) {
    if (flx_ballot(J.st == P_DONE) == 0ull) return;
    FLX_FRAME_ARGS();
    bool toShade = false, ended = false;
    if (J.st == P_DONE) FLX_WALK_LANE_FOLD(false, fr, wb, compactRecs, J, nullptr, toShade, ended, false, 0u);
    fq_push(shadeRing, ctl + FC_SQ, toShade, J.pathId, lane);
    const uint32_t nEnded = (uint32_t)__popcll(flx_ballot(ended));
    if (nEnded != 0u && lane == 0) atomicSub(&ctl[FC_ALIVE], nEnded);
  };
  /* ---- refill a job's free lanes: paths that came back from shading first, then fresh (tile, sample) units of this workgroup's shade waves; false: nothing more to be had now ---- */
  auto refillJob = [&](WalkJob &J) -> bool {
    for (;;) {
      const unsigned long long idle = flx_ballot(J.st == P_EMPTY);
      if (idle == 0ull) return true;
      FLX_FRAME_ARGS();
      const uint32_t nIdle = (uint32_t)__popcll(idle);
      uint32_t id = WF_INVALID;
      bool fresh = false;
      const uint32_t back = fq_pop(walkRing, ctl + FC_WQ, idle, nIdle, 1u, lane, id);
      if (back == 0u) {
        if (chunkNext == chunkEnd) {
          if (fq_load(&ctl[FC_SQ + 2]) >= FQ_LIMIT) return false;
          uint32_t unit = WF_INVALID;
          if (fq_pop(readyRing, ctl + FC_RQ, 1ull, 1u, 1u, lane, unit) == 0u) return false;
          unit = __builtin_amdgcn_readfirstlane(unit);
          if (unit == WF_INVALID) {
            if (lane == 0) { atomicSub(&ctl[FC_ALIVE], 64u); if (wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            return false;
          }
          chunkNext = unit << 6; chunkEnd = chunkNext + 64u;
        }
        const uint32_t avail = chunkEnd - chunkNext;
        const uint32_t take = nIdle < avail ? nIdle : avail;
        const uint32_t r = lane_rank(idle);
        if (J.st == P_EMPTY && r < take) { id = chunkNext + r; fresh = true; }
        chunkNext += take;
      }
      bool dead = false;
      if (id != WF_INVALID) dead = walkLaneLoad<COUNT>(fr, wb, id, fresh && compactRecs, J, cnt);
      const uint32_t nDead = (uint32_t)__popcll(flx_ballot(dead));
      if (nDead != 0u && lane == 0) atomicSub(&ctl[FC_ALIVE], nDead);
    }
  };
  /* ---- set up a job's walks: fresh ones (shadow or closest) and the closest-hit walk of a job whose shadow walk just ended ---- */
  auto setupJob = [&](WalkJob &J) {
    walkLaneSwitch(J);
    if (flx_ballot(J.st == P_SETUP) != 0ull) {
      FLX_FRAME_ARGS();
      if (J.st == P_SETUP) walkLaneSetup<COUNT>(sc, nTransforms, ldsXf, J.rays, walkG, ldsEntries, ldsCount, J, cnt);
    }
  };
  /* After a phase's test: the entry the stepped job's link names — ONE fetch for the lanes of both jobs (the address, the visit count, the object-space check and the
   * terminator test are common; only the three loads are issued per job, each into its job's own registers).  p1: the lane stepped job 1; ended: the test ended the walk. */
  auto stepEnd = [&](bool p1, bool ended) {
    const uint32_t link = (uint32_t)selv(p1, j1.w.i, j0.w.i);
    if (link == WALK_END) ended = true;
    if (!ended) {
      const uint32_t i = linkIndex(link);
      const float4 *src = (i < ldsCount) ? ldsEntries + 3u * i : walkG + 3 * (size_t)i;
      /* (the two arms must not look alike to the optimiser: it would merge them into ONE set of loads stored through a pointer to either job, and both jobs would live in scratch memory) */
      if (p1) { asm volatile("; job 1"); j1.cur.e0 = src[0]; j1.cur.e1 = src[1]; j1.cur.e2 = src[2]; asm volatile("; job 1 loaded"); }
      else { asm volatile("; job 0"); j0.cur.e0 = src[0]; j0.cur.e1 = src[1]; j0.cur.e2 = src[2]; asm volatile("; job 0 loaded"); }
      const int mode = selv(p1, j1.w.mode, j0.w.mode);
      if (COUNT) { if (mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
      const int meta = __float_as_int(selv(p1, j1.cur.e2.z, j0.cur.e2.z));
      const int tI = (meta >> 2) << 1;
      if (FLX_UNLIKELY(tI != selv(p1, j1.w.cachedTI, j0.w.cachedTI))) {
        if (p1) { asm volatile("; job 1"); j1.w.cachedTI = tI; walkLoadRay(j1.rays, tI >> 1, j1.w); asm volatile("; job 1 ray"); }
        else { asm volatile("; job 0"); j0.w.cachedTI = tI; walkLoadRay(j0.rays, tI >> 1, j0.w); asm volatile("; job 0 ray"); }
      }
      ended = (meta & 3) == 0;
    }
    if (ended) {
      const int endSt = selv(p1, j1.w.mode, j0.w.mode) == 0 ? P_SWITCH : P_DONE;
      j1.st = selv(p1, endSt, j1.st);
      j0.st = selv(p1, j0.st, endSt);
    }
  };

  for (;;) {
    const unsigned long long wk0 = flx_ballot(j0.st == P_WALKING), wk1 = flx_ballot(j1.st == P_WALKING);
    if ((wk0 | wk1) != 0ull) idleSpins = 0;
    const unsigned long long workMask = flx_ballot(j0.st == P_DONE || j0.st == P_SWITCH || j1.st == P_DONE || j1.st == P_SWITCH);
    const uint32_t parked = 128u - (uint32_t)__popcll(wk0) - (uint32_t)__popcll(wk1);
    const bool mayRefill = fq_load(&ctl[FC_RQ + 2]) != 0u || chunkNext != chunkEnd || fq_load(&ctl[FC_WQ + 2]) != 0u;
    if ((wk0 | wk1) == 0ull || (parked >= (uint32_t)FLX_FRAME2_BATCH && (workMask != 0ull || mayRefill))) {
      foldJob(j0);
      foldJob(j1);
      if (refillJob(j0)) (void)refillJob(j1);
      setupJob(j0);
      setupJob(j1);
      if (flx_ballot(j0.st == P_WALKING || j1.st == P_WALKING) == 0ull) {
        if (flx_ballot(j0.st != P_EMPTY || j1.st != P_EMPTY) != 0ull) continue;      /* jobs that had nothing to walk wait for the fold */
        if (fq_load(&ctl[FC_DRY]) != 0u && chunkNext == chunkEnd && fq_load(&ctl[FC_ALIVE]) == 0u) break;
        if (fq_load(&ctl[FC_WQ + 2]) == 0u && (fq_load(&ctl[FC_RQ + 2]) == 0u || fq_load(&ctl[FC_SQ + 2]) >= FQ_LIMIT)) {
          FLX_FRAME_ARGS();
          if (++idleSpins > (wb.watchdog ? wb.watchdog : FQ_WATCHDOG)) {
            if (lane == 0 && wb.error) __hip_atomic_fetch_or(wb.error, WF_ERR_WALK_WATCHDOG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (COUNT && lane == 0) { for (uint32_t k = 0; k < 8u; k++) atomicMax(wb.counters + 50 + k, (unsigned long long)fq_load(&ctl[k]) + 1ull); atomicAdd(wb.counters + 58, 1ull); }
            break;
          }
          __builtin_amdgcn_s_sleep(8);
        }
        continue;
      }
    }
    /* ---- FLX_WF_INNER trips: a box phase, and a triangle phase when enough lanes want one ---- */
#pragma unroll 1
    for (int it = 0; it < FLX_WF_INNER; it++) {
      {
        const bool b0 = j0.st == P_WALKING && walkIsBoxT(j0.cur), b1 = j1.st == P_WALKING && walkIsBoxT(j1.cur);
        if (COUNT) { const unsigned long long m = flx_ballot(b0 | b1); if (m) { diagBoxTrips++; diagBoxLanes += (unsigned long long)__popcll(m); } }
        if (b0 | b1) {
          const bool p1 = b1 & (!b0 | (last == 0u));          /* job 1 unless only job 0 stands at a box, or both do and job 1 went last */
          WalkState ws;
          ws.tR.origin = sel3(p1, j1.w.tR.origin, j0.w.tR.origin);
          ws.inv = sel3(p1, j1.w.inv, j0.w.inv);
          ws.fastDiv = selv(p1, j1.w.fastDiv, j0.w.fastDiv);
          const float l = selv(p1, j1.w.minLen, j0.w.minLen);
          const f3 lo = sel3(p1, F3(j1.cur.e0.x, j1.cur.e0.y, j1.cur.e0.z), F3(j0.cur.e0.x, j0.cur.e0.y, j0.cur.e0.z));
          const f3 hi = sel3(p1, F3(j1.cur.e0.w, j1.cur.e1.x, j1.cur.e1.y), F3(j0.cur.e0.w, j0.cur.e1.x, j0.cur.e1.y));
          bool sure;
          bool hit = rayCuboidInterval(l, ws, lo, hi, sure);
          if (FLX_UNLIKELY(flx_ballot(!sure) != 0ull)) {       /* rare: the exact quotients, from the job's own state */
            if (!sure) {
              if (p1) { asm volatile("; job 1"); hit = rayCuboidRecip(l, j1.w, lo, hi); asm volatile("; job 1 exact"); }
              else { asm volatile("; job 0"); hit = rayCuboidRecip(l, j0.w, lo, hi); asm volatile("; job 0 exact"); }
            }
          }
          const float lx = selv(p1, j1.cur.e2.x, j0.cur.e2.x), ly = selv(p1, j1.cur.e2.y, j0.cur.e2.y);
          const int nxt = __float_as_int(hit ? lx : ly);
          j1.w.i = selv(p1, nxt, j1.w.i);
          j0.w.i = selv(p1, j0.w.i, nxt);
          stepEnd(p1, false);
          last = p1 ? 1u : 0u;
        }
      }
      {
        const bool w0 = j0.st == P_WALKING, w1 = j1.st == P_WALKING;
        const bool k0 = walkIsBoxT(j0.cur), k1 = walkIsBoxT(j1.cur);
        const bool t0 = w0 & !k0, t1 = w1 & !k1;
        const unsigned long long tm = flx_ballot(t0 | t1);
        if (tm != 0ull) {
          const unsigned long long bm = flx_ballot((w0 & k0) | (w1 & k1));
          if ((uint32_t)__popcll(tm) >= (uint32_t)FLX_FRAME2_TRI_MIN || bm == 0ull || ++triWait >= (uint32_t)FLX_FRAME2_TRI_WAIT) {
            triWait = 0;
            if (COUNT) { diagTriTrips++; diagTriLanes += (unsigned long long)__popcll(tm); }
            if (t0 | t1) {
              const bool p1 = t1 & (!t0 | (last == 0u));
              const f3 a = sel3(p1, F3(j1.cur.e0.x, j1.cur.e0.y, j1.cur.e0.z), F3(j0.cur.e0.x, j0.cur.e0.y, j0.cur.e0.z));
              const f3 edge1 = sel3(p1, F3(j1.cur.e0.w, j1.cur.e1.x, j1.cur.e1.y), F3(j0.cur.e0.w, j0.cur.e1.x, j0.cur.e1.y));
              const f3 edge2 = sel3(p1, F3(j1.cur.e1.z, j1.cur.e1.w, j1.cur.e2.x), F3(j0.cur.e1.z, j0.cur.e1.w, j0.cur.e2.x));
              Ray ray;
              ray.origin = sel3(p1, j1.w.tR.origin, j0.w.tR.origin);
              ray.dir = sel3(p1, j1.w.tR.dir, j0.w.tR.dir);
              const float l = selv(p1, j1.w.minLen, j0.w.minLen);
              const bool cull = selv(p1, j1.w.mode, j0.w.mode) == 0;
              f3 suv;
              const bool hit = moellerTrumboreAny(a, edge1, edge2, ray, l, cull, suv);
              bool ended;
              if (p1) { asm volatile("; job 1"); ended = walkTriApply(j1.w, j1.cur, hit, suv); asm volatile("; job 1 applied"); }
              else { asm volatile("; job 0"); ended = walkTriApply(j0.w, j0.cur, hit, suv); asm volatile("; job 0 applied"); }
              stepEnd(p1, ended);
              last = p1 ? 1u : 0u;
            }
          }
        }
      }
    }
  }
  FLX_FRAME_ARGS();
  if (COUNT && (cnt.closest_visits | cnt.shadow_visits) != 0u) atomicAdd(wb.counters + 23, (unsigned long long)cnt.closest_visits + cnt.shadow_visits);
  if (COUNT && lane == 0) {
    const unsigned long long life = (unsigned long long)(clock64() - tStart);
    atomicAdd(wb.counters + 64, life); atomicMax(wb.counters + 65, life); atomicAdd(wb.counters + 66, 1ull);
    atomicAdd(wb.counters + 67, diagBoxTrips); atomicAdd(wb.counters + 68, diagBoxLanes); atomicAdd(wb.counters + 69, diagTriTrips); atomicAdd(wb.counters + 70, diagTriLanes);      /* flx_get_tail_diag 27..30 */
  }
  flush_counters<COUNT>(cnt, wb.counters);
#undef FLX_FRAME_ARGS
}

/* Can the two-job frame kernel take this frame? */
static bool frame2_kernel_fits(const DeviceScene &sc, uint32_t &ldsCount, uint32_t &ldsBytes) {
  const uint32_t T = sc.n_transforms;
  const uint32_t walkThreads = FLX_FRAME2_THREADS - 64u * (uint32_t)FLX_FRAME2_SHADERS;
  const uint32_t fixed = walkThreads * 2u * T * 40u + T * 64u + FC_WORDS * 4u;
  if (!FLX_WF_PRETRANSFORM || fixed + 4096u > (uint32_t)FLX_WF_LDS_TOTAL) return false;
  ldsCount = ((uint32_t)FLX_WF_LDS_TOTAL - fixed) / 48u;
  if (ldsCount > sc.walk_hot) ldsCount = sc.walk_hot;
  ldsBytes = ldsCount * 48u + fixed;
  return true;
}

#endif /* FLX_EXPERIMENTS: k_wf_frame2 */

/* Can the frame kernel take this frame?  Its LDS holds the rays of its walk threads, the staged transforms, the two rings and
 * at least a little of the tree's top. */
static bool frame_kernel_fits(const DeviceScene &sc, bool withFront, uint32_t &ldsCount, uint32_t &ldsBytes) {
  const uint32_t T = sc.n_transforms;
  const uint32_t walkThreads = FLX_WF_WALK_THREADS - 64u * (withFront ? (uint32_t)FLX_FRAME_SHADERS_FRONT : (uint32_t)FLX_FRAME_SHADERS);      /* (the shade waves keep no rays) */
  const uint32_t fixed = walkThreads * T * 40u + T * 64u + FC_WORDS * 4u;
  if (!FLX_WF_PRETRANSFORM || fixed + 4096u > (uint32_t)FLX_WF_LDS_TOTAL) return false;
  ldsCount = ((uint32_t)FLX_WF_LDS_TOTAL - fixed) / 48u;
  if (ldsCount > sc.walk_hot) ldsCount = sc.walk_hot;
  ldsBytes = ldsCount * 48u + fixed;
  return true;
}

/* A kernel's dynamic-LDS limit (hipFuncSetAttribute) belongs to the device it is set on: once per device and kernel family, whichever context launches there
 * first (a group of contexts on several GPUs lives in one process). */
/* (A second host thread that launches on the same device — another context, a twin driven from another thread — waits until the attributes are set; a
 * limit that cannot be raised is remembered and the launch refused, so that the caller falls back or reports it.) */
template <typename SetAttributes>
static bool dynamic_lds_ready(int family, SetAttributes set) {
  static std::once_flag once[3][64];
  static bool ok[3][64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::call_once(once[family][dev], [&]() { ok[family][dev] = set(); });
  return ok[family][dev];
}
static bool set_lds_limit(const void *kernel) { return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }

/* Will launch_wavefront run these items as ONE frame kernel? */
static bool frame_kernel_wanted(const DeviceScene &sc, const DeviceFrame &fr, uint32_t item_count, int walk_scheduler, uint32_t suspend_max, int organisation,
                                bool withFront, uint32_t &ldsCountF, uint32_t &ldsBytesF) {
#if !FLX_EXPERIMENTS
  walk_scheduler = 0; suspend_max = 0u;                     /* (flx_set_walk_scheduler refuses anything else in this build) */
#endif
  /* automatic: the frame kernel while the pass is small enough for the rounds' tails to matter — measured crossover at ~66 M paths
   * (a 4K frame at 8 spp, or four 1080p frames per pass: tools/organisation_time.py, profiles/r03_organisation_crossover.txt), with the front of
   * the frame inside the launch somewhere between 66 M (3.7 % ahead) and 265 M (2.7 % behind); beyond it the rounds' sixteen walk waves per CU beat
   * fourteen walk + two shade waves */
  const bool wanted = organisation == 2 || (organisation == 0 && item_count <= (withFront ? (uint32_t)FLX_FRAME_AUTO_MAX_ITEMS_FRONT : (uint32_t)FLX_FRAME_AUTO_MAX_ITEMS));
  return wanted && walk_scheduler == 0 && suspend_max == 0u && fr.max_reflections >= 1 && frame_kernel_fits(sc, withFront, ldsCountF, ldsBytesF);
}
/* May the frame kernel also take the front of the frame (primary rays, bounce-0 shading: WavefrontBuffers::front)?  Then launch_primary is not needed. */
bool wavefront_front_in_kernel(const DeviceScene &sc, const DeviceFrame &fr, uint32_t item_count, int walk_scheduler, uint32_t suspend_max, int organisation) {
  uint32_t a = 0, b = 0;
  return frame_kernel_wanted(sc, fr, item_count, walk_scheduler, suspend_max, organisation, true, a, b) && (uint32_t)fr.samples * 64u * 2u <= FQ_ALIVE_MAX;
}

int launch_wavefront(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wbIn, uint32_t compute_units, bool count,
                     int walk_scheduler, uint32_t suspend_max, int organisation, hipEvent_t walk0_begin, hipEvent_t walk0_end, hipStream_t stream) {
  WavefrontBuffers wb = wbIn;
  const bool fused = wb.front == 2u;                        /* no hits yet: primary rays and bounce-0 shading in one launch (k_wf_front) in front of whatever walks */
  if (fused) wb.front = 0u;
#if !FLX_EXPERIMENTS
  walk_scheduler = 0; suspend_max = 0u;                     /* (flx_set_walk_scheduler refuses anything else in this build) */
#endif
  /* ---- one persistent launch for the whole bounce loop (k_wf_frame), where it fits ---- */
  {
    uint32_t ldsCountF = 0, ldsBytesF = 0;
    if (wb.frameRings != nullptr && frame_kernel_wanted(sc, fr, wb.item_count, walk_scheduler, suspend_max, organisation, wb.front != 0u, ldsCountF, ldsBytesF) &&
        dynamic_lds_ready(0, []() {
          return (int)set_lds_limit((const void *)k_wf_frame<true, false>) & (int)set_lds_limit((const void *)k_wf_frame<false, false>) &
                 (int)set_lds_limit((const void *)k_wf_frame<true, true>) & (int)set_lds_limit((const void *)k_wf_frame<false, true>) &
                 (int)set_lds_limit((const void *)k_wf_frame_stamped<true>) & (int)set_lds_limit((const void *)k_wf_frame_stamped<false>);
        })) {
      const uint32_t total = wb.item_count;
      const uint32_t pixels = total / (uint32_t)(fr.samples > 0 ? fr.samples : 1);
      const uint32_t shadeBlocks = (pixels + 255u) / 256u;
      FrameArgs fa; fa.sc = sc; fa.fr = fr; fa.wb = wb;
      if (wb.front) { /* the frame kernel shades bounce 0 itself */ }
      else if (fused) { if (count) hipLaunchKernelGGL(k_wf_front<true>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total); else hipLaunchKernelGGL(k_wf_front<false>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total); }
      else if (count) hipLaunchKernelGGL(k_wf_shade0<true>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total);
      else hipLaunchKernelGGL(k_wf_shade0<false>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total);
      if (walk0_begin) (void)hipEventRecord(walk0_begin, stream);
      const uint32_t shadeWaves = wb.front ? (uint32_t)FLX_FRAME_SHADERS_FRONT : (uint32_t)FLX_FRAME_SHADERS;
      const dim3 grid(compute_units * (uint32_t)FLX_FRAME_GROUPS_PER_CU), block(FLX_WF_WALK_THREADS);
      /* front inside: (tile, sample) units of 64 fresh paths a workgroup keeps ready for its walk waves before its shade waves stop making more.  What is
       * ready is bound to the workgroup, so the fewer tiles a workgroup gets the less it may hoard: FLX_FRAME_READY_UNITS from 48 tiles per workgroup on, below
       * that half a unit per tile it can expect but FLX_FRAME_READY_UNITS / 4 at the least (whole 1080p frame, 127 tiles per workgroup: 16 -> 6.61, 32 -> 6.43, 64 -> 6.49 ms; a quarter of
       * it, 31 tiles: 8 -> 2.40, 16 -> 2.33, 32 -> 2.57 ms; an eighth: 8 -> 1.76, 16 -> 1.83, 32 -> 2.0 ms) */
      const uint32_t tilesPerGroup = (total / ((uint32_t)fr.samples * 64u)) / compute_units;
#ifndef FLX_FRAME_READY_DIV
#define FLX_FRAME_READY_DIV 2u
#endif
      uint32_t readyUnits = tilesPerGroup >= 48u ? (uint32_t)FLX_FRAME_READY_UNITS : tilesPerGroup / FLX_FRAME_READY_DIV;
      readyUnits = readyUnits < (uint32_t)FLX_FRAME_READY_UNITS / 4u ? (uint32_t)FLX_FRAME_READY_UNITS / 4u : (readyUnits > (uint32_t)FLX_FRAME_READY_UNITS ? (uint32_t)FLX_FRAME_READY_UNITS : readyUnits);
#if FLX_EXPERIMENTS
      /* two walk jobs per lane (k_wf_frame2: 512-thread workgroups) where the front of the frame is inside the launch */
      uint32_t ldsCount2 = 0, ldsBytes2 = 0;
      if (wb.front && (wb.walkJobs & 0xffu) == 2u && frame2_kernel_fits(sc, ldsCount2, ldsBytes2) &&
          dynamic_lds_ready(2, []() { return (int)set_lds_limit((const void *)k_wf_frame2<true>) & (int)set_lds_limit((const void *)k_wf_frame2<false>); })) {
        const dim3 block2(FLX_FRAME2_THREADS);
        if (count) hipLaunchKernelGGL((k_wf_frame2<true>), grid, block2, ldsBytes2, stream, fa, total, ldsCount2, sc.n_transforms, (uint32_t)FLX_FRAME2_SHADERS, readyUnits);
        else hipLaunchKernelGGL((k_wf_frame2<false>), grid, block2, ldsBytes2, stream, fa, total, ldsCount2, sc.n_transforms, (uint32_t)FLX_FRAME2_SHADERS, readyUnits);
        if (walk0_end) (void)hipEventRecord(walk0_end, stream);
        return 3;
      }
#endif
      if (wb.front && (wb.walkJobs & WF_STAMP_COSTS)) {
        if (count) hipLaunchKernelGGL((k_wf_frame_stamped<true>), grid, block, ldsBytesF, stream, fa, total, ldsCountF, sc.n_transforms, shadeWaves, readyUnits);
        else hipLaunchKernelGGL((k_wf_frame_stamped<false>), grid, block, ldsBytesF, stream, fa, total, ldsCountF, sc.n_transforms, shadeWaves, readyUnits);
      } else if (wb.front) {
        if (count) hipLaunchKernelGGL((k_wf_frame<true, true>), grid, block, ldsBytesF, stream, fa, total, ldsCountF, sc.n_transforms, shadeWaves, readyUnits);
        else hipLaunchKernelGGL((k_wf_frame<false, true>), grid, block, ldsBytesF, stream, fa, total, ldsCountF, sc.n_transforms, shadeWaves, readyUnits);
      } else {
        if (count) hipLaunchKernelGGL((k_wf_frame<true, false>), grid, block, ldsBytesF, stream, fa, total, ldsCountF, sc.n_transforms, shadeWaves, readyUnits);
        else hipLaunchKernelGGL((k_wf_frame<false, false>), grid, block, ldsBytesF, stream, fa, total, ldsCountF, sc.n_transforms, shadeWaves, readyUnits);
      }
      if (walk0_end) (void)hipEventRecord(walk0_end, stream);
      return wb.front ? 3 : 2;
    }
  }
  if (wb.front) return -1;                              /* the caller left the primary rays to a frame kernel that does not run (wavefront_front_in_kernel said it would) */
  const uint32_t total = wb.item_count;                 /* items of this group (all of the frame when there is one group) */
  const uint32_t maxBlocks = compute_units * 8u;
  /* walk kernel: one big workgroup per CU.  LDS first holds every thread's pre-transformed rays
   * (n_transforms x 48 B each) when they fit, the rest goes to the tree top. */
  const uint32_t T = sc.n_transforms;
  const uint32_t rayBytes = FLX_WF_WALK_THREADS * T * 40u + T * 64u + 128u;      /* per-thread rays + the staged inverse transforms + tailCtl */
  const bool pre = FLX_WF_PRETRANSFORM && rayBytes <= 148u * 1024u && rayBytes <= (uint32_t)FLX_WF_LDS_TOTAL;
  const uint32_t ldsBudget = (uint32_t)FLX_WF_LDS_TOTAL - (pre ? rayBytes : 0u);
  uint32_t ldsCount = ldsBudget / 48u;
  if (ldsCount > sc.walk_hot) ldsCount = sc.walk_hot;
  const uint32_t ldsBytes = ldsCount * 48u + (pre ? rayBytes : 0u);
  uint32_t perCu = (160u * 1024u) / (ldsBytes > 20480u ? ldsBytes : 20480u);
  if (perCu > 8u) perCu = 8u;
  const uint32_t walkBlocks = compute_units * perCu;
  if (!dynamic_lds_ready(1, []() {
        int ok = 1;
        for (const void *k : { (const void *)k_wf_walk<true>, (const void *)k_wf_walk<false>, (const void *)k_wf_walk_pre<true, true>, (const void *)k_wf_walk_pre<false, true>,
                               (const void *)k_wf_walk_pre<true, false>, (const void *)k_wf_walk_pre<false, false> }) ok &= (int)set_lds_limit(k);
        return ok != 0;
      })) return -2;                                    /* the walk kernels cannot have their LDS on this device */
  const int bounces = fr.max_reflections > 0 ? fr.max_reflections : 1;   /* 0 bounces: shade(0) only finalises */
  /* Suspension needs the kernel that can take a walk up again (k_wf_walk_pre), at least two bounces to gain anything, and
   * as many extra rounds as a path can be held up: one per regular round.  The extra rounds find their lists empty
   * almost always and return at once. */
  const bool finisher = (walk_scheduler & 2) != 0;          /* suspended walks go to k_wf_walk_coop instead of the next round */
  const bool lanes = (walk_scheduler & 1) == 0;             /* one walk per lane (not the queue scheduler) */
  const bool suspend = suspend_max > 0u && pre && lanes && (bounces >= 2 || finisher) && FLX_WF_CONSOLIDATE;
  const int rounds = (suspend && !finisher) ? 2 * bounces : bounces;
  /* compact bounce-0 records: only the default walk kernel reads them (and a suspended walk re-reads its full record) */
  if (!(pre && lanes && !suspend)) { wb.rec0 = nullptr; wb.pix0 = nullptr; }
  FrameArgs fa; fa.sc = sc; fa.fr = fr; fa.wb = wb;                /* (after the choice of the records' layout above) */
  for (int r = 0; r < rounds; r++) {
    if (r == 0) {
      const uint32_t pixels = total / (uint32_t)(fr.samples > 0 ? fr.samples : 1);        /* 64 per screen tile */
      const uint32_t shadeBlocks = (pixels + 255u) / 256u;
      if (fused) { if (count) hipLaunchKernelGGL(k_wf_front<true>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total); else hipLaunchKernelGGL(k_wf_front<false>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total); }
      else if (count) hipLaunchKernelGGL(k_wf_shade0<true>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total);
      else hipLaunchKernelGGL(k_wf_shade0<false>, dim3(shadeBlocks), dim3(256), 0, stream, fa, total);
    } else {
      const uint32_t shadeBlocks = maxBlocks * 2u;
      if (count) hipLaunchKernelGGL(k_wf_shade<true>, dim3(shadeBlocks), dim3(256), 0, stream, fa, r);
      else hipLaunchKernelGGL(k_wf_shade<false>, dim3(shadeBlocks), dim3(256), 0, stream, fa, r);
    }
    if (r == 0 && walk0_begin) (void)hipEventRecord(walk0_begin, stream);
    const uint32_t smax = (suspend && r < bounces) ? suspend_max : 0u;
    if (!lanes) {
#if FLX_EXPERIMENTS
      launch_walk_queue(sc, fr, wb, compute_units, count, r, total, stream);
#endif
    } else if (pre) {
      if (r == 0) {
        if (count) hipLaunchKernelGGL((k_wf_walk_pre<true, true>), dim3(walkBlocks), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsCount, T, smax, finisher ? 0u : 1u);
        else hipLaunchKernelGGL((k_wf_walk_pre<false, true>), dim3(walkBlocks), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsCount, T, smax, finisher ? 0u : 1u);
      } else {
        if (count) hipLaunchKernelGGL((k_wf_walk_pre<true, false>), dim3(walkBlocks), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsCount, T, smax, finisher ? 0u : 1u);
        else hipLaunchKernelGGL((k_wf_walk_pre<false, false>), dim3(walkBlocks), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsCount, T, smax, finisher ? 0u : 1u);
      }
    } else {
      if (count) hipLaunchKernelGGL(k_wf_walk<true>, dim3(walkBlocks), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsCount);
      else hipLaunchKernelGGL(k_wf_walk<false>, dim3(walkBlocks), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, sc, fr, wb, r, total, ldsCount);
    }
    if (r == 0 && walk0_end) (void)hipEventRecord(walk0_end, stream);
#if FLX_EXPERIMENTS
    if (suspend && finisher) launch_walk_coop(sc, fr, wb, compute_units, count, r, stream);
#endif
  }
  return 1;
}

size_t wavefront_live_capacity(const DeviceFrame &fr, uint32_t compute_units) {
  return (size_t)path_item_count(fr) + (size_t)WF_OUT_CHUNK * compute_units * 8u * (FLX_WF_WALK_THREADS / 64u) + 1024u;
}

}  // namespace flx
