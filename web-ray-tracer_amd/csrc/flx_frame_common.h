/* flx_frame_common.h — what the shade kernels, the frame kernel (flx_wavefront.hip) and the chained frame kernel (flx_chain.hip) share: the
 * kernel arguments read from the kernarg segment, the front of a frame for one screen tile, one path's shading, the rings of a workgroup. */
#pragma once
#include "flx_kernels.h"
#include "flx_kernel_util.h"
#include "flx_wavefront_common.h"

namespace flx {

#ifndef FLX_WF_WALK_THREADS
#define FLX_WF_WALK_THREADS 1024
#endif
#ifndef FLX_WF_WAVES_PER_EU
#define FLX_WF_WAVES_PER_EU 4               /* occupancy the register allocation of the walk kernels must allow */
#endif
#ifndef FLX_WF_LDS_TOTAL
#define FLX_WF_LDS_TOTAL (156 * 1024)         /* LDS a walk workgroup may use (of 160 KB per CU) */
#endif
#ifndef FLX_WF_INNER
#define FLX_WF_INNER 8
#endif
#ifndef FLX_WF_UNROLL
#define FLX_WF_UNROLL 8                 /* the FLX_WF_INNER trips of the stepping loop unrolled: no loop counter, compare and branch per trip (round 5: dragon 1080p 6.25 -> 6.14 ms, 4K 23.4 -> 23.0;
                                         * 2: 6.20, 4: 6.16; with six trips per check 6.12 but the frame server's eighth 0.94 -> 0.96: profiles/r05_trip_instructions.txt) */
#endif
#ifndef FLX_WF_BATCH
#define FLX_WF_BATCH 24                     /* parked lanes that trigger a fold + refill */
#endif
#ifndef FLX_WF_DRAWS_PER_WAVE
#define FLX_WF_DRAWS_PER_WAVE 16
#endif
#ifndef FLX_FRAME_SHADERS_FRONT
#define FLX_FRAME_SHADERS_FRONT 3           /* shade waves of a frame-kernel workgroup when they also make the fresh paths (WavefrontBuffers::front) */
#endif
#ifndef FLX_FRAME_PROLOGUE_WAVES
#define FLX_FRAME_PROLOGUE_WAVES 2          /* walk waves that make a first tile before their loop (FRONT) */
#endif
#ifndef FLX_FRAME_READY_UNITS
#define FLX_FRAME_READY_UNITS 32            /* front in the kernel: the most (tile, sample) units of 64 fresh paths a workgroup keeps ready (launch_wavefront: readyUnits) */
#endif

/* The arguments of the shade kernels and of the frame kernel (2.7 KB: the scene's pointers, the frame with its views, the work buffers) stay in the
 * kernarg segment and are read where they are used, through a pointer the compiler cannot see through (an empty asm; the loads stay scalar loads):
 * as by-value parameters every field was loaded at the kernel's entry and kept — in the frame kernel 262 scalar registers spilled into vector-register
 * lanes around the stepping loop and 92 spilled vector registers (profiles/r04_resources.txt). */
typedef const __attribute__((address_space(4))) FrameArgs *FrameArgsP;
FLX_DEV const FrameArgs &frame_args(FrameArgsP p) { asm volatile("" : "+s"(p)); return *(const FrameArgs *)p; }
FLX_DEV FrameArgsP kernel_frame_args() { return (FrameArgsP)__builtin_amdgcn_kernarg_segment_ptr(); }      /* the kernel's FIRST parameter is the FrameArgs: offset 0 */
#define FLX_ARGS_OF(ab) const FrameArgs &A_ = frame_args(ab); const DeviceScene &sc = A_.sc; const DeviceFrame &fr = A_.fr; const WavefrontBuffers &wb = A_.wb; (void)sc; (void)fr; (void)wb

/* Bounce 0: one lane per PIXEL.  All samples of a pixel share the primary hit (fragment:606-613), so everything the shading
 * knows before it draws a random number — triangle and attribute fetch, normals, the acos / tan of the normal deviation,
 * material (shadeSurface) — is computed once and the per-sample rest (shadeSample) runs `samples` times.  Every path gets the
 * record the per-path kernel would have written, bit for bit. */
/* (the lane's pixel of screen tile `tile`; h = its primary hit: suv + triangle id as bits, -1 for none; returns whether the pixel's paths run) */
/* VER (the frame server with a scene that moves, flx_server.hip): DeviceScene::rotation / shift / lights hold one version of those arrays per (workgroup, frame
 * slot) — version blockIdx.x x frames + the frame's slot — and a path reads its frame's. */
template <bool VER>
FLX_DEV uint32_t scene_version_of(const DeviceFrame &fr, uint32_t frameIdx) { return VER ? blockIdx.x * fr.frames + frameIdx : 0u; }

template <bool COUNT, bool LV = false, bool VER = false>
FLX_DEV bool shade0_tile(FrameArgsP ab, uint32_t tile, uint32_t lane, float4 h, WorkCounters &cnt, const FrameView *lv = nullptr) {
  uint32_t S, frameIdx, lightBase = 0;
  int tri;
  bool alive, compact;
  f3 camera;
  SurfaceCtx sf;
  float ndcX = 0.0f, ndcY = 0.0f;
  Hit hit;
  {
  FLX_ARGS_OF(ab);
  S = (uint32_t)fr.samples;
  uint32_t px, k;
  tile8_pixel(fr, tile, lane, px, k);
  const bool inFrame = px < fr.width && k < fr.rows;
  frameIdx = inFrame ? frame_index(fr, k) : 0u;
  camera = view_camera(view_at<LV>(fr, lv, frameIdx));
  tri = inFrame ? __float_as_int(h.w) : -1;
  /* loop guard of fragment:475 before the first bounce (importancy and originalColor are 1) */
  alive = tri != -1 && fr.max_reflections > 0 && length(F3(1.0f, 1.0f, 1.0f) * F3(1.0f, 1.0f, 1.0f)) >= fr.min_importancy * SQRT3;
  hit.suv = F3(h.x, h.y, h.z); hit.triangleId = tri; hit.transformId = 0;
  compact = wb.rec0 != nullptr;
  if (alive) {
    hit.transformId = (int)sc.geometry[3 * tri + 2].y << 1;
    if (VER) { const uint32_t ver = scene_version_of<VER>(fr, frameIdx); hit.transformId += (int)(ver * 2u * sc.n_transforms); lightBase = ver * 6u * sc.n_lights; }
    const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
    float viewDepthPerS;
    Ray pr;
    pr.origin = camera;
    pr.dir = primary_dir_v(fr, view_at<LV>(fr, lv, frameIdx), px, py_gl, ndcX, ndcY, viewDepthPerS);
    const WorkCounters before = cnt;
    shadeSurface<COUNT>(sc, fr, hit, pr, camera, sf, cnt);
    if (COUNT) {                                            /* the per-path kernel counts these once per path */
      cnt.shades = before.shades + (cnt.shades - before.shades) * S;
      cnt.atlas_texels = before.atlas_texels + (cnt.atlas_texels - before.atlas_texels) * S;
    }
  }
  }
  for (uint32_t s = 0; s < S; s++) {
    FLX_ARGS_OF(ab);                                          /* (read again per sample: nothing of the arguments stays in registers across the loop) */
    const uint32_t pathId = ((tile * S + s) << 6) | lane;
    float4 *rec = wb.rec + (size_t)pathId * 8;
    if (!alive) {
      if (tri != -1) finalize_path<LV>(fr, wb, pathId, F3(0.0f, 0.0f, 0.0f), F3(1.0f, 1.0f, 1.0f), F3(1.0f, 1.0f, 1.0f), lv);
      if (compact) wb.rec0[(size_t)pathId * 3] = make_float4(0.f, 0.f, 0.f, __int_as_float(RF_DEAD));
      else rec[0] = make_float4(0.f, 0.f, 0.f, __int_as_float(RF_DEAD));
      continue;
    }
    PathState p;
    PixelState ps;
    ps.firstRayLength = 1.0f; ps.glassFilter = 0.0f; ps.originalRMEx = 0.0f; ps.originalTPOx = 0.0f;
    ps.renderId.x = ps.renderId.y = ps.renderId.z = ps.renderId.w = 0.0f;
    ps.renderOriginalId = ps.renderId;
    ps.ndc_x = ndcX; ps.ndc_y = ndcY;
    ps.seed = view_at<LV>(fr, lv, frameIdx).random_seed;
    ps.lightBase = lightBase;
    ps.originalColor = F3(1.0f, 1.0f, 1.0f);
    p.hit = hit;
    p.lastHitPoint = camera;
    p.dontFilter = true;
    p.importancyFactor = F3(1.0f, 1.0f, 1.0f);
    const float cosSampleN = flx_cos((float)s);
    ShadeOut so;
    shadeSample(sc, fr, sf, ps, p, camera, cosSampleN, 0, so);
    const int flags = (p.dontFilter ? RF_DONT_FILTER : 0) | (so.needShadow ? RF_NEED_SHADOW : 0) | (so.shadowedNoWalk ? RF_SHADOWED_NO_WALK : 0) |
                      (nextBounceRuns(fr, 0, p.importancyFactor, ps.originalColor) ? 0 : RF_NO_CLOSEST);
    if (compact) {
      /* the per-pixel part is the same for every sample (sf is; importancy stays (1,1,1) while dontFilter holds, as it does
       * on entry to bounce 0): written once */
      if (s == 0u) {
        float4 *pp = wb.pix0 + (((size_t)tile << 6) | lane) * 3;      /* [screen tile][lane]: what a path id gives without a division */
        pp[0] = make_float4(p.ray.origin.x, p.ray.origin.y, p.ray.origin.z, 0.0f);
        pp[1] = make_float4(so.shadowRay.origin.x, so.shadowRay.origin.y, so.shadowRay.origin.z, 0.0f);
        pp[2] = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, so.baseLuminance.x);
      }
      float4 *r0 = wb.rec0 + (size_t)pathId * 3;
      r0[0] = make_float4(p.ray.dir.x, p.ray.dir.y, p.ray.dir.z, __int_as_float(flags));
      r0[1] = make_float4(so.shadowRay.dir.x, so.shadowRay.dir.y, so.shadowRay.dir.z, so.shadowLen);
      r0[2] = make_float4(so.litColor.x, so.litColor.y, so.litColor.z, 0.0f);
      continue;
    }
    rec[0] = make_float4(p.ray.origin.x, p.ray.origin.y, p.ray.origin.z, __int_as_float(flags));
    rec[1] = make_float4(p.ray.dir.x, p.ray.dir.y, p.ray.dir.z, so.shadowLen);
    rec[2] = make_float4(so.shadowRay.origin.x, so.shadowRay.origin.y, so.shadowRay.origin.z, so.baseLuminance.x);
    rec[3] = make_float4(so.shadowRay.dir.x, so.shadowRay.dir.y, so.shadowRay.dir.z, __int_as_float(0));
    rec[4] = make_float4(so.litColor.x, so.litColor.y, so.litColor.z, 0.0f);
    rec[5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    rec[6] = make_float4(p.importancyFactor.x, p.importancyFactor.y, p.importancyFactor.z, 0.0f);
    rec[7] = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, 0.0f);
  }
  return alive;
}

/* The primary ray of the lane's pixel of screen tile `tile` (what k_primary does for it, flx_kernels.hip): the wave walks the forward-ordered copy
 * together.  -> suv + triangle id as bits (-1: no hit, or no pixel), also stored for k_resolve. */
template <bool COUNT, bool LV = false, bool VER = false>
FLX_DEV float4 primary_tile(FrameArgsP ab, uint32_t tile, uint32_t lane, WorkCounters &cnt, const FrameView *lv = nullptr) {
  FLX_ARGS_OF(ab);
  float4 *__restrict__ hits = const_cast<float4 *>(wb.hits);
  uint32_t px, k;
  tile8_pixel(fr, tile, lane, px, k);
  const bool inImage = px < fr.width && k < fr.rows;
  float nx, ny, viewDepthPerS = 0.0f;
  Ray pr; pr.origin = F3(0.0f, 0.0f, 0.0f); pr.dir = F3(0.0f, 0.0f, 1.0f);
  int xfBase = 0;
  if (inImage) {
    const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
    const uint32_t frameIdx = frame_index(fr, k);
    if (VER) xfBase = (int)(scene_version_of<VER>(fr, frameIdx) * 2u * sc.n_transforms);      /* (per lane, from the lane's own row: vector loads — the scalar cache is not coherent with the stores that filled the version) */
    const FrameView &v = view_at<LV>(fr, lv, frameIdx);
    pr.dir = primary_dir_v(fr, v, px, py_gl, nx, ny, viewDepthPerS);
    pr.origin = view_camera(v);
  }
  const uint32_t visitsBefore = cnt.primary_visits;
  const Hit hp = primaryWalkF(sc, inImage, pr, viewDepthPerS, cnt.primary_visits, xfBase);
  if (COUNT && wb.tileCost && wb.tileCostPrimary) {              /* flx_debug_tile_cost with room for two sums per tile: the primary rays' visits in the second half */
    const uint32_t nTiles = ((fr.width + 7u) >> 3) * ((fr.rows + 7u) >> 3);
    atomicAdd(wb.tileCost + nTiles + tile, (unsigned long long)(cnt.primary_visits - visitsBefore));
  }
  float4 h = make_float4(hp.suv.x, hp.suv.y, hp.suv.z, __int_as_float(inImage ? hp.triangleId : -1));
  if (inImage) {
    if (COUNT && hp.triangleId != -1) cnt.primary_hits++;
    hits[(size_t)k * fr.width + px] = h;
  }
  return h;
}

/* One path's shading for its next bounce (fragment:476-589): the record the last walk left -> the record the next walk reads. */
template <bool COUNT, bool LV = false, bool VER = false>
FLX_DEV void shade_path(FrameArgsP ab, uint32_t pathId, WorkCounters &cnt, const FrameView *lv = nullptr) {
  FLX_ARGS_OF(ab);
  float4 *rec = wb.rec + (size_t)pathId * 8;
  uint32_t px, k, s;
  item_pixel(fr, pathId, px, k, s);
  const uint32_t frameIdx = frame_index(fr, k);
  const f3 camera = view_camera(view_at<LV>(fr, lv, frameIdx));
  PathState p;
  PixelState ps;
  ps.firstRayLength = 1.0f; ps.glassFilter = 0.0f; ps.originalRMEx = 0.0f; ps.originalTPOx = 0.0f;
  ps.renderId.x = ps.renderId.y = ps.renderId.z = ps.renderId.w = 0.0f;
  ps.renderOriginalId = ps.renderId;
  const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q6 = rec[6], q7 = rec[7];
  const int pb = __float_as_int(q3.w) + 1;            /* the bounce this path is at (its record carries the last one shaded) */
  p.ray.origin = F3(q0.x, q0.y, q0.z);
  p.lastHitPoint = p.ray.origin;                      /* fragment:595 */
  p.ray.dir = F3(q1.x, q1.y, q1.z);
  p.hit.suv = F3(q2.x, q2.y, q2.z);
  p.hit.triangleId = __float_as_int(q2.w);
  p.hit.transformId = (int)sc.geometry[3 * p.hit.triangleId + 2].y << 1;
  if (VER) { const uint32_t ver = scene_version_of<VER>(fr, frameIdx); p.hit.transformId += (int)(ver * 2u * sc.n_transforms); ps.lightBase = ver * 6u * sc.n_lights; }
  p.dontFilter = (__float_as_int(q0.w) & RF_DONT_FILTER) != 0;
  p.importancyFactor = F3(q6.x, q6.y, q6.z);
  ps.originalColor = F3(q7.x, q7.y, q7.z);
  const uint32_t py_gl = fr.height - 1u - image_row(fr, k);
  float viewDepthPerS;
  (void)primary_dir_v(fr, view_at<LV>(fr, lv, frameIdx), px, py_gl, ps.ndc_x, ps.ndc_y, viewDepthPerS);
  ps.seed = view_at<LV>(fr, lv, frameIdx).random_seed;
  const float cosSampleN = flx_cos((float)s);
  ShadeOut so;
  bounceShade<COUNT>(sc, fr, ps, p, camera, cosSampleN, pb, so, cnt);
  const int flags = (p.dontFilter ? RF_DONT_FILTER : 0) | (so.needShadow ? RF_NEED_SHADOW : 0) | (so.shadowedNoWalk ? RF_SHADOWED_NO_WALK : 0) |
                    (nextBounceRuns(fr, pb, p.importancyFactor, ps.originalColor) ? 0 : RF_NO_CLOSEST);
  rec[0] = make_float4(p.ray.origin.x, p.ray.origin.y, p.ray.origin.z, __int_as_float(flags));
  rec[1] = make_float4(p.ray.dir.x, p.ray.dir.y, p.ray.dir.z, so.shadowLen);
  rec[2] = make_float4(so.shadowRay.origin.x, so.shadowRay.origin.y, so.shadowRay.origin.z, so.baseLuminance.x);
  rec[3] = make_float4(so.shadowRay.dir.x, so.shadowRay.dir.y, so.shadowRay.dir.z, __int_as_float(pb));
  rec[4] = make_float4(so.litColor.x, so.litColor.y, so.litColor.z, 0.0f);
  rec[6] = make_float4(p.importancyFactor.x, p.importancyFactor.y, p.importancyFactor.z, 0.0f);
  rec[7] = make_float4(ps.originalColor.x, ps.originalColor.y, ps.originalColor.z, 0.0f);
}

/* ---- a walk lane of the persistent frame kernels ------------------------------------------------------------------------------------------------------
 * What a lane of a walk wave holds and does is the same in every persistent kernel (k_wf_frame, the frame server's k_wf_server, the experiments' k_wf_frame2): a path —
 * its shadow walk, then its closest-hit walk — whose record it loads, walks and folds.  The kernels differ in where path ids come from and go to (one set of rings, rings
 * per frame slot, a mailbox); that stays with them.  One body here, so that a change to the walk is a change to all of them. */
struct WalkLane {
  int st;                                  /* P_EMPTY .. P_SETUP */
  uint32_t pathId; int flags; int pathBounce; float base;
  Ray nextRay, shadowRay; float shadowLen;
  WalkState w; WalkEntry cur;
  uint32_t v0;                             /* counted builds: the lane's visit count when its path came in (flx_debug_tile_cost) */
};
FLX_DEV void walkLaneInit(WalkLane &L) {
  L.v0 = 0u;
  L.st = P_EMPTY; L.pathId = 0; L.flags = 0; L.pathBounce = 0; L.base = 0.0f;
  L.nextRay.origin = F3(0.f, 0.f, 0.f); L.nextRay.dir = L.nextRay.origin;
  L.shadowRay = L.nextRay; L.shadowLen = 0.0f;
  walkClearResults(L.w);
  L.w.src = L.nextRay; L.w.tR = L.nextRay; L.w.minLen = 0.0f; L.w.i = 0; L.w.cachedTI = 0;
  L.w.mode = 2;
  L.cur.e0 = L.cur.e1 = L.cur.e2 = make_float4(0.f, 0.f, 0.f, 0.f);
}
/* the per-pixel part of a bounce-0 path's compact record */
FLX_DEV const float4 *pix_part(const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t id) {
  uint32_t tile0, s0;
  item_tile(fr, id, tile0, s0);
  return wb.pix0 + (((size_t)tile0 << 6) | (id & 63u)) * 3;
}
/* Fold a lane whose walks are done (st == P_DONE): fragment:445-460, 580, 593-598 and the guard of :475.  A path that goes on gets its record completed (toShade_: the
 * caller hands it to the shade waves); one that ends has its radiance stored (ended_: the caller takes it off its count).  The lane is free afterwards.
 * A macro for the reason FLX_WALK_LANE_STEP is one: as an inlined function taking the lane by reference the same statements schedule differently in the frame kernel's
 * fold (loads hoisted over a mask change, eight more instructions) and the dragon frame measures 0.65 % slower (6.39 -> 6.44 ms, same lease, three runs each). */
#define FLX_WALK_LANE_FOLD(LV_, fr_, wb_, compactRecs_, L_, lv_, toShade_, ended_, costed_, stamp_)                              \
  do {                                                                                                                           \
    float4 *rec_ = (wb_).rec + (size_t)(L_).pathId * 8;                                                                          \
    const bool compact_ = (compactRecs_) && (L_).pathBounce == 0;                                                                \
    float4 q4_, q5_, q6_, q7_;                                                                                                   \
    const float4 *pp_ = nullptr;                                                                                                 \
    if (compact_) {                                                                                                              \
      pp_ = pix_part(fr_, wb_, (L_).pathId);                                                                                     \
      q4_ = (wb_).rec0[(size_t)(L_).pathId * 3 + 2]; q7_ = pp_[2];                                                               \
      q5_ = make_float4(0.0f, 0.0f, 0.0f, 0.0f); q6_ = make_float4(1.0f, 1.0f, 1.0f, 0.0f);                                      \
    } else {                                                                                                                     \
      q4_ = rec_[4]; q5_ = rec_[5]; q6_ = rec_[6]; q7_ = rec_[7];                                                                \
    }                                                                                                                            \
    const bool shadowed_ = ((L_).flags & RF_SHADOWED_NO_WALK) || (((L_).flags & RF_NEED_SHADOW) && (L_).w.shadowed);            \
    const f3 localColor_ = shadowed_ ? F3((L_).base, (L_).base, (L_).base) : F3(q4_.x, q4_.y, q4_.z);                            \
    const f3 importancy_ = F3(q6_.x, q6_.y, q6_.z), originalColor_ = F3(q7_.x, q7_.y, q7_.z);                                    \
    const f3 finalColor_ = F3(q5_.x, q5_.y, q5_.z) + localColor_ * importancy_;                                                  \
    /* what the path has cost so far (time in the walk lanes, 10 ns ticks; q5.w, then the w of its radiance slot: the adaptive tile order's measure) */ \
    const float cost_ = (costed_) ? q5_.w + (float)(((stamp_) - ((uint32_t)(L_).flags >> 8)) & 0xffffffu) : 1.0f;     /* (costed_: a compile-time constant) */ \
    bool cont_ = (L_).w.tri != -1;                                                                                               \
    if (cont_) cont_ = ((L_).pathBounce + 1) < (fr_).max_reflections && length(importancy_ * originalColor_) >= (fr_).min_importancy * SQRT3;      \
    if (cont_) {                                                                                                                 \
      if (compact_) {                                   /* the path goes on: now it gets its full record (what shade0 would have written) */      \
        const float4 a_ = (wb_).rec0[(size_t)(L_).pathId * 3], bq_ = (wb_).rec0[(size_t)(L_).pathId * 3 + 1], p0_ = pp_[0];      \
        rec_[0] = make_float4(p0_.x, p0_.y, p0_.z, a_.w);                                                                        \
        rec_[1] = make_float4(a_.x, a_.y, a_.z, bq_.w);                                                                          \
        rec_[3] = make_float4(bq_.x, bq_.y, bq_.z, __int_as_float(0));                                                           \
        rec_[6] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);                                                                           \
        rec_[7] = make_float4(q7_.x, q7_.y, q7_.z, 0.0f);                                                                        \
      }                                                                                                                          \
      rec_[5] = make_float4(finalColor_.x, finalColor_.y, finalColor_.z, (costed_) ? cost_ : 0.0f);                                                 \
      rec_[2] = make_float4((L_).w.suv.x, (L_).w.suv.y, (L_).w.suv.z, __int_as_float((L_).w.tri));                               \
      toShade_ = true;                                                                                                           \
    } else {                                                                                                                     \
      finalize_path<LV_>(fr_, wb_, (L_).pathId, finalColor_, importancy_, originalColor_, lv_, cost_);                           \
      ended_ = true;                                                                                                             \
    }                                                                                                                            \
    (L_).st = P_EMPTY;                                                                                                           \
  } while (0)

/* Take path `id` into a free lane: its record (the compact bounce-0 form where `compactFresh`).  true: the item is dead (a pixel without a path), the lane stays free.
 * In two halves — the loads, and what they mean for the lane — so that a kernel can have the loads in flight while it does something else (k_wf_frame folds the lane's old path). */
struct WalkRecord { float4 q0, q1, q2, q3; };
FLX_DEV void walkLaneFetchRecord(const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t id, bool compactFresh, WalkRecord &R) {
  const float4 *rec = wb.rec + (size_t)id * 8;
  if (compactFresh) {
    const float4 *pp = pix_part(fr, wb, id);
    const float4 a = wb.rec0[(size_t)id * 3], bq = wb.rec0[(size_t)id * 3 + 1];
    const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2];
    R.q0 = make_float4(p0.x, p0.y, p0.z, a.w);
    R.q1 = make_float4(a.x, a.y, a.z, bq.w);
    R.q2 = make_float4(p1.x, p1.y, p1.z, p2.w);
    R.q3 = make_float4(bq.x, bq.y, bq.z, __int_as_float(0));
  } else {
    R.q0 = rec[0]; R.q1 = rec[1]; R.q2 = rec[2]; R.q3 = rec[3];
  }
}
template <bool COUNT>
FLX_DEV bool walkLaneInstall(uint32_t id, const WalkRecord &R, WalkLane &L, WorkCounters &cnt, uint32_t stamp = 0u /* 24 bits: when the path came in (the fold takes the difference) */) {
  const int fl = __float_as_int(R.q0.w);
  if (fl & RF_DEAD) return true;
  L.pathId = id; L.flags = fl | (int)(stamp << 8); L.base = R.q2.w; L.pathBounce = __float_as_int(R.q3.w);
  L.nextRay.origin = F3(R.q0.x, R.q0.y, R.q0.z);
  L.nextRay.dir = F3(R.q1.x, R.q1.y, R.q1.z);
  L.shadowRay.origin = F3(R.q2.x, R.q2.y, R.q2.z);
  L.shadowRay.dir = F3(R.q3.x, R.q3.y, R.q3.z);
  L.shadowLen = R.q1.w;
  walkClearResults(L.w);
  L.w.mode = (fl & RF_NEED_SHADOW) ? 0 : 1;
  if (COUNT) { if (L.w.mode == 0) cnt.shadow_walks++; if (!(fl & RF_NO_CLOSEST)) cnt.closest_walks++; L.v0 = cnt.closest_visits + cnt.shadow_visits; }
  L.st = (L.w.mode == 1 && (fl & RF_NO_CLOSEST)) ? P_DONE : P_SETUP;      /* nothing to walk: straight to the fold */
  return false;
}
template <bool COUNT>
FLX_DEV bool walkLaneLoad(const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t id, bool compactFresh, WalkLane &L, WorkCounters &cnt, uint32_t stamp = 0u) {
  WalkRecord R;
  walkLaneFetchRecord(fr, wb, id, compactFresh, R);
  return walkLaneInstall<COUNT>(id, R, L, cnt, stamp);
}
/* Set up walks: fresh lanes (shadow or closest) and lanes whose shadow walk just ended (P_SWITCH).  xf: the staged inverse transforms the lane's path reads. */
template <bool COUNT>
FLX_DEV void walkLaneSetup(const DeviceScene &sc, uint32_t nTransforms, const float4 *xf, float2 *myRays, const float4 *walkG, const float4 *ldsEntries, uint32_t ldsCount,
                           WalkLane &L, WorkCounters &cnt) {
  const bool shadowMode = L.w.mode == 0;
  const Ray src = shadowMode ? L.shadowRay : L.nextRay;
  walkSetupRays(sc, nTransforms, xf, myRays, src, shadowMode);
  L.w.tR = src; L.w.cachedTI = 0; L.w.minLen = shadowMode ? L.shadowLen : POW32; L.w.i = (int)sc.walk_root;
  reciprocalOfDir(sc, src.dir, src.origin, L.w.inv, L.w.fastDiv);
  L.st = P_WALKING;
  if (walkFetchG<COUNT>(walkG, ldsEntries, ldsCount, myRays, L.w, L.cur, cnt)) L.st = shadowMode ? P_SWITCH : P_DONE;
}
FLX_DEV void walkLaneSwitch(WalkLane &L) {                  /* the shadow walk is over: the closest-hit walk next, unless the path ends after this bounce (nextBounceRuns) */
  if (L.st == P_SWITCH) {
    if (L.flags & RF_NO_CLOSEST) L.st = P_DONE;
    else { L.w.mode = 1; L.st = P_SETUP; }
  }
}
/* One entry for a walking lane: the test its entry asks for, then the entry its link names.  (A macro, not a function: as an inlined function the same statements cost
 * the frame kernel's stepping loop six more instructions per trip — two register copies and four mask operations — and the dragon frame 2.4 %: 6.40 -> 6.56 ms.) */
#define FLX_WALK_LANE_STEP(COUNT_, walkG_, ldsEntries_, ldsCount_, myRays_, L_, cnt_)                                              \
  do {                                                                                                                           \
    if ((L_).st == P_WALKING) {                                                                                                  \
      bool ended_ = false;                                                                                                       \
      if (FLX_WF_LINK_ISBOX ? walkIsBoxL((L_).w) : walkIsBoxT((L_).cur)) walkBoxP((L_).w, (L_).cur); else ended_ = walkTriT((L_).w, (L_).cur);                             \
      if (!ended_) ended_ = walkFetchG<COUNT_>(walkG_, ldsEntries_, ldsCount_, myRays_, (L_).w, (L_).cur, cnt_);                 \
      if (ended_) (L_).st = ((L_).w.mode == 0) ? P_SWITCH : P_DONE;                                                              \
    }                                                                                                                            \
  } while (0)

constexpr uint32_t FQ_SIZE = WF_FRAME_RING;   /* ids per ring (the rings live in HBM-backed memory private to the workgroup, their counts in LDS) */
#ifndef FLX_FQ_LIMIT
#define FLX_FQ_LIMIT 4096
#endif
constexpr uint32_t FQ_LIMIT = FLX_FQ_LIMIT; /* paths waiting for shading beyond which the walk waves stop drawing new ones (a policy: the shade waves are behind) */
constexpr uint32_t FQ_ALIVE_MAX = FQ_SIZE - 256u;     /* live paths of a workgroup, enforced at every draw (the rings' capacity) */
#ifndef FLX_FQ_WATCHDOG_LOG2
#define FLX_FQ_WATCHDOG_LOG2 24
#endif
constexpr uint32_t FQ_WATCHDOG = 1u << FLX_FQ_WATCHDOG_LOG2;  /* polls (~500 cycles each) after which a wave that waits gives up: a seconds-long guard against a hung GPU, never reached by a frame */
enum { FC_ALIVE = 0, FC_DRY = 1, FC_SQ = 2 /* tail, head, avail */, FC_WQ = 5 /* tail, head, avail */, FC_RQ = 8 /* tail, head, avail */, FC_FRONT_DONE = 11, FC_WORDS = 16 };
#ifndef FLX_FRAME_READY_UNITS
#define FLX_FRAME_READY_UNITS 32            /* front in the kernel: the most (tile, sample) units of 64 fresh paths a workgroup keeps ready (launch_wavefront: readyUnits) */
#endif

FLX_DEV uint32_t fq_load(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
/* the lanes of `mine` append their `id` to the ring (ctl: tail, head, avail) */
FLX_DEV void fq_push(uint32_t *ring, uint32_t *ctl, bool mine, uint32_t id, uint32_t lane) {
  const unsigned long long m = flx_ballot(mine);
  if (m == 0ull) return;
  const uint32_t c = (uint32_t)__popcll(m);
  uint32_t pos0 = 0;
  if (lane == 0) pos0 = atomicAdd(&ctl[0], c);
  pos0 = __builtin_amdgcn_readfirstlane(pos0);
  if (mine) __hip_atomic_store(&ring[(pos0 + lane_rank(m)) & (FQ_SIZE - 1u)], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          /* the path's record (plain global stores) and the slot, before the count */
  if (lane == 0) atomicAdd(&ctl[2], c);
}
/* up to `want` ids (none unless at least `atLeast` are there) for the lanes of `takers`, lowest lanes first; returns how many (uniform) */
FLX_DEV uint32_t fq_pop(uint32_t *ring, uint32_t *ctl, unsigned long long takers, uint32_t want, uint32_t atLeast, uint32_t lane, uint32_t &id) {
  uint32_t n = 0, pos0 = 0;
  if (lane == 0) {
    uint32_t a = fq_load(&ctl[2]);
    while (a != 0u && a >= atLeast) {
      const uint32_t take = a < want ? a : want;
      const uint32_t seen = atomicCAS(&ctl[2], a, a - take);
      if (seen == a) { n = take; pos0 = atomicAdd(&ctl[1], take); break; }
      a = seen;
    }
  }
  n = __builtin_amdgcn_readfirstlane(n);
  pos0 = __builtin_amdgcn_readfirstlane(pos0);
  if (n == 0u) return 0u;
  const uint32_t r = lane_rank(takers);
  if (((takers >> lane) & 1ull) != 0ull && r < n) {
    uint32_t *slot = &ring[(pos0 + r) & (FQ_SIZE - 1u)];
    uint32_t v, spins = 0;
    do { v = fq_load(slot); } while (v == WF_INVALID && ++spins < FQ_WATCHDOG);      /* (pushes are counted in the order they finish, not in slot order: the slot says when it is filled) */
    __hip_atomic_store(slot, WF_INVALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    id = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return n;
}


}  // namespace flx
