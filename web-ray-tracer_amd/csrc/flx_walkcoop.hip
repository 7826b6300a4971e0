/*
 * flx_walkcoop.hip — a wave walks ONE ray: the finisher of the walks the lane-per-walk kernel suspends.
 *
 * A walk is a chain: the test of one entry decides which entry comes next, and one lane gets through an entry in about
 * 2 000 cycles whatever else the machine does (tools/diag_lone.py).  The longest walks of a bounce (~1 000 entries for a
 * ray that grazes the dragon) therefore set a floor of ~0.8 ms under every walk kernel, and on a tile-sharded frame that
 * floor does not shrink with the number of GPUs.  k_wf_walk_pre hands the last walks of every workgroup over
 * (wb.strag); here each of them gets a whole wave:
 *
 *   the 64 lanes test the 64 entries [i, i + 64) of the skip list AT ONCE (original order: a hit box is followed by its
 *   first child, a leaf's triangles are consecutive, so the next few entries the walk visits are usually in that window;
 *   4-5 of them for long walks, profiles/r01_walk_policy_model.txt), then a short scalar scan follows the links through
 *   the 64 results exactly as the sequential loop would — visited entry by visited entry, skipping on a box miss —
 *   and stops where a result would be stale: after a triangle hit that shortens minLen, after an entry that changes
 *   the cached transform, or when the next entry lies outside the window.
 *
 * Only entries the sequential walk visits are ever committed or counted, each with the ray, the minLen and the
 * arithmetic it would have seen (fragment:184-224 / :240-277), so hits, colours and work counters are unchanged; the
 * tests of entries the walk turns out not to visit are thrown away.
 */
#include <cstdio>
#include "flx_wavefront_common.h"

namespace flx {

#ifndef FLX_COOP_THREADS
#define FLX_COOP_THREADS 256
#endif

template <bool COUNT>
__global__ __launch_bounds__(FLX_COOP_THREADS) void k_wf_walk_coop(DeviceScene sc, DeviceFrame fr, WavefrontBuffers wb, int b) {
  const uint32_t nStrag = wb.stragCount[b];
  if (nStrag == 0u) return;
  const uint32_t lane = threadIdx.x & 63u;
  const float4 *__restrict__ stragIn = wb.strag[b & 1];
  uint32_t *__restrict__ listOut = wb.live[(b + 1) & 1];
  uint32_t *__restrict__ outAlloc = wb.counts + (b + 1);
  uint32_t *__restrict__ cursor = wb.coopQueue + b;
  const int nEntries = (int)sc.n_entries;
  WorkCounters cnt = {};
  uint32_t outBase = 0, outUsed = WF_OUT_CHUNK;
  bool outValid = false;

  for (;;) {
    uint32_t j = 0;
    if (lane == 0) j = atomicAdd(cursor, 1u);
    j = __builtin_amdgcn_readfirstlane(j);
    if (j >= nStrag) break;
    /* ---- take the walk up: its registers from the straggler record, rays and flags from the path record ------- */
    const float4 *sr = stragIn + (size_t)j * WF_STRAG_F4;
    const float4 s0 = sr[0], s1 = sr[1], s2 = sr[2], s3 = sr[3], s4 = sr[4];
    const uint32_t pathId = (uint32_t)__float_as_int(s0.x);
    float4 *rec = wb.rec + (size_t)pathId * 8;
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
    const int flags = __float_as_int(q0.w);
    const float base = q2.w;
    const int pathBounce = __float_as_int(q3.w);
    Ray nextRay, shadowRay;
    nextRay.origin = F3(q0.x, q0.y, q0.z); nextRay.dir = F3(q1.x, q1.y, q1.z);
    shadowRay.origin = F3(q2.x, q2.y, q2.z); shadowRay.dir = F3(q3.x, q3.y, q3.z);
    const int packed = __float_as_int(s0.y);
    int mode = (packed >> 4) & 15;                         /* 0 shadowTest, 1 rayTracer */
    bool shadowed = ((packed >> 9) & 1) != 0;
    bool startClosest = (packed & 15) == P_SWITCH;   /* the shadow walk had ended when the walk was suspended */
    /* the walk's loop variables, uniform across the wave */
    float minLen = s0.z;
    Ray cur; cur.origin = F3(s1.x, s1.y, s1.z); cur.dir = F3(s1.w, s2.x, s2.y);
    f3 curInv = F3(s2.z, s2.w, s3.x);
    bool curFast = ((packed >> 8) & 1) != 0;
    f3 suv = F3(s3.y, s3.z, s3.w);
    int cachedTI = __float_as_int(s4.x), tri = __float_as_int(s4.y), hitTI = __float_as_int(s4.z);
    /* the entry the lane kernel had fetched (and counted) but not yet tested, in original order */
    int i = 0;
    bool firstCounted = false;
    if (!startClosest) {
      const uint32_t link = (uint32_t)__float_as_int(s0.w);
      i = ((const int *)sc.walk)[(size_t)linkIndex(link) * 12u + 11u];
      firstCounted = true;
    }

    for (;;) {                                              /* shadow walk, then closest-hit walk */
      if (startClosest && (flags & RF_NO_CLOSEST)) break;   /* the path ends after this bounce: no closest-hit walk (nextBounceRuns) */
      if (startClosest) {
        mode = 1; cur = nextRay; cachedTI = 0; minLen = POW32; i = 0; firstCounted = false;
        reciprocalOfDir(sc, cur.dir, cur.origin, curInv, curFast);
        startClosest = false;
      }
      const Ray src = (mode == 0) ? shadowRay : nextRay;
      bool walkEnded = false;
      while (!walkEnded) {
        /* ---- 64 entries at once ----------------------------------------------------------------------------- */
        const int e = i + (int)lane;
        const bool valid = e < nEntries;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        if (valid) { g0 = sc.geometry[3 * (size_t)e]; g1 = sc.geometry[3 * (size_t)e + 1]; g2 = sc.geometry[3 * (size_t)e + 2]; }
        const int type = valid ? (int)g2.z : -1;
        const int tI = (int)g2.y << 1;
        /* the ray in this entry's object space: the cached one, or fragment:197-202 / :257-262 */
        WalkState w;
        w.tR = cur; w.inv = curInv; w.fastDiv = curFast;
        if (valid && type != 0 && tI != cachedTI) {
          const int iI = tI + 1;
          const M3 rotationII = rotation_at(sc, iI);
          w.tR.origin = mul(rotationII, src.origin + shift_at(sc, iI));
          const f3 d = mul(rotationII, src.dir);
          w.tR.dir = (mode == 0) ? normalize(d) : d;
          reciprocalOfDir(sc, w.tR.dir, w.tR.origin, w.inv, w.fastDiv);
        }
        int step = 1;                                        /* entries to the next one the walk visits after this one */
        bool upd = false;
        f3 hs = F3(0.f, 0.f, 0.f);
        if (type == 1) {
          if (!rayCuboidRecip(minLen, w, F3(g0.x, g0.y, g0.z), F3(g0.w, g1.x, g1.y))) step += (int)g1.z;
        } else if (type == 2) {
          const f3 a = F3(g0.x, g0.y, g0.z), bb = F3(g0.w, g1.x, g1.y), c = F3(g1.z, g1.w, g2.x);
          const bool hit = moellerTrumboreAny(a, bb - a, c - a, w.tR, minLen, mode == 0, hs);
          upd = hit && (mode == 0 || hs.x != 0.0f);        /* fragment:271 / :217 */
        }
        /* ---- follow the links through the results, as the sequential loop would --------------------------------- */
        uint32_t p = 0;
        bool moved = false;
        for (;;) {
          if (p >= 64u) { i += (int)p; break; }            /* the next entry is outside this window */
          const int t = laneI(type, p);
          if (t < 0) { walkEnded = true; break; }          /* loop bound (fragment:184): no fetch */
          if (COUNT && !(firstCounted && !moved && p == 0u)) { if (mode == 0) cnt.shadow_visits++; else cnt.closest_visits++; }
          moved = true;
          if (t == 0) { walkEnded = true; break; }         /* terminator (fragment:208) */
          const int tIp = laneI(tI, p);
          bool changed = false;
          if (tIp != cachedTI) {
            cachedTI = tIp;
            cur.origin = F3(laneF(w.tR.origin.x, p), laneF(w.tR.origin.y, p), laneF(w.tR.origin.z, p));
            cur.dir = F3(laneF(w.tR.dir.x, p), laneF(w.tR.dir.y, p), laneF(w.tR.dir.z, p));
            curInv = F3(laneF(w.inv.x, p), laneF(w.inv.y, p), laneF(w.inv.z, p));
            curFast = laneI(w.fastDiv ? 1 : 0, p) != 0;
            changed = true;
          }
          if (t == 2 && laneI(upd ? 1 : 0, p) != 0) {
            if (mode == 0) { shadowed = true; walkEnded = true; break; }
            suv = F3(laneF(hs.x, p), laneF(hs.y, p), laneF(hs.z, p));
            minLen = suv.x; tri = i + (int)p; hitTI = tIp;
            i += (int)p + 1;                                /* later results were computed with the old minLen */
            break;
          }
          const uint32_t nx = (uint32_t)laneI(step, p);
          if (changed) { i += (int)(p + nx); break; }      /* later results of the old transform's entries may be stale */
          p += nx;
        }
        firstCounted = false;
      }
      if (mode == 0) { startClosest = true; continue; }
      break;
    }
    /* ---- fold: fragment:445-460, 580, 593-598 and the guard of :475 (every lane computes the same values) ------ */
    (void)hitTI;
    const float4 q4 = rec[4], q5 = rec[5], q6 = rec[6], q7 = rec[7];
    const bool isShadowed = (flags & RF_SHADOWED_NO_WALK) || ((flags & RF_NEED_SHADOW) && shadowed);
    const f3 localColor = isShadowed ? F3(base, base, base) : F3(q4.x, q4.y, q4.z);
    const f3 importancy = F3(q6.x, q6.y, q6.z), originalColor = F3(q7.x, q7.y, q7.z);
    const f3 finalColor = F3(q5.x, q5.y, q5.z) + localColor * importancy;
    bool cont = tri != -1;
    if (cont) cont = (pathBounce + 1) < fr.max_reflections && length(importancy * originalColor) >= fr.min_importancy * SQRT3;
    if (cont) {
      if (outUsed == WF_OUT_CHUNK) {
        uint32_t nb = 0;
        if (lane == 0) nb = atomicAdd(outAlloc, WF_OUT_CHUNK);
        outBase = __builtin_amdgcn_readfirstlane(nb); outUsed = 0; outValid = true;
      }
      if (lane == 0) {
        rec[5] = make_float4(finalColor.x, finalColor.y, finalColor.z, 0.0f);
        rec[2] = make_float4(suv.x, suv.y, suv.z, __int_as_float(tri));
        listOut[outBase + outUsed] = pathId;
      }
      outUsed++;
    } else if (lane == 0) {
      finalize_path(fr, wb, pathId, finalColor, importancy, originalColor);
    }
  }
  if (outValid) {
    for (uint32_t t = outUsed + lane; t < WF_OUT_CHUNK; t += 64u) listOut[outBase + t] = WF_INVALID;
  }
  if (COUNT && lane != 0) { cnt.shadow_visits = 0; cnt.closest_visits = 0; }      /* the wave's tallies are uniform: count them once */
  flush_counters<COUNT>(cnt, wb.counters);
}

void launch_walk_coop(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t compute_units, bool count, int b,
                      hipStream_t stream) {
  const dim3 grid(compute_units * 4u), block(FLX_COOP_THREADS);
  if (count) hipLaunchKernelGGL(k_wf_walk_coop<true>, grid, block, 0, stream, sc, fr, wb, b);
  else hipLaunchKernelGGL(k_wf_walk_coop<false>, grid, block, 0, stream, sc, fr, wb, b);
}

}  // namespace flx
