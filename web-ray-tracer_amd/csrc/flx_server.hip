/*
 * flx_server.hip — the frame server (round 4): ONE persistent launch renders the frames of the loop as flx_frame_begin posts them.
 *
 * Why.  A frame kernel launch (flx_wavefront.hip: k_wf_frame) ends in a drain: once its tile queue is dry a workgroup finishes the paths it holds at the pace
 * of their chains walk -> shade -> walk.  For a rank's eighth of the 1080p dragon frame that is half of the launch (1.65 ms where the work is 0.8: 3.9x at 8
 * GPUs).  Two launches cannot share a CU, so a second lane only fills the CUs the first one has left (1.29 ms per frame); and a chain of launches that work
 * ahead on each other's frames (flx_chain.hip) pays per launch — stop, hand-over lists, relaunch, ramp-up — what the overlap gains (1.36 - 1.47 ms).  The
 * chains themselves are long (1.3 - 2 ms from a frame's first tile under load): the machine needs two to three frames in flight AT ALL TIMES, with nothing in
 * between.  The reference's loop renders frame after frame from one context without waiting for the GPU (modules/pathtracerWGL2.js:254-303); the shader's
 * loop is per pixel (shaders/pathtracer_fragment.glsl:475-596): nothing in the algorithm ties a launch to a frame.
 *
 * How.  The loop's frames in flight (depth 2 or 3) are slots of one stacked workspace (the layout of a batch of frames: path ids, screen tiles and packed
 * rows of slot s + 1 follow those of slot s).  The launch's workgroups are k_wf_frame's — walk waves, shade waves that also make the fresh paths, rings of
 * path ids per slot — and every workgroup works on the frames of all slots at once, older frames first.  A workgroup that holds nothing of its oldest frame
 * any more and finds that frame's tile queue dry ROTATES: the slot becomes the place of the frame `depth` later, whose view the host will post.  It then
 * publishes its part (an agent-scope release: the radiance its paths stored must be in memory) and adds to the slot's counter; the workgroup whose add is the
 * last resets the slot's tile queue and tells the HOST (a word in pinned memory) that the frame is complete.  flx_frame_end waits for that word, finds the
 * frame resolved (every workgroup resolves the screen tiles it made when it is through with the frame), and the next flx_frame_begin posts the next frame's view into the slot (plain stores into pinned memory; a few
 * waves relay posts into device memory for the others).  The launch ends when the host says after which frame (the loop ran empty, or something else needs
 * the device); a workgroup with nothing to do for milliseconds gives up with an error.
 *
 * Per path nothing changes — the same records, arithmetic and order as in k_wf_frame, the radiance in the path's own slot — so the frames are bit-identical to
 * single renders (tests/test_server_gpu.py).
 */
#include <cstddef>
#include <cstdio>
#include <mutex>
#include "flx_server.h"
#include "flx_frame_common.h"

namespace flx {

#ifndef FLX_SERVER_RELAY_GROUPS
#define FLX_SERVER_RELAY_GROUPS 16           /* workgroups whose shade waves read the host's mailbox and pass it on in device memory */
#endif
#ifndef FLX_SERVER_SHADERS
#define FLX_SERVER_SHADERS FLX_FRAME_SHADERS_FRONT      /* shade waves of a server workgroup (they also make the fresh paths) */
#endif
#ifndef FLX_SERVER_SHADERS_DEPTH2_MAX_TILES
#define FLX_SERVER_SHADERS_DEPTH2_MAX_TILES 24         /* ... of at most this many screen tiles per workgroup and frame */
#endif
#ifndef FLX_SERVER_SHADERS_DEPTH2
#define FLX_SERVER_SHADERS_DEPTH2 2                    /* ... of a launch with two frame slots: a rank's eighth with two frames in flight 1.23 -> 1.18 ms (with three slots two shade waves lose: 0.975 -> 1.11;
                                                        * profiles/r04_paths_occupancy.txt) */
#endif
#ifndef FLX_SERVER_SHADE_PRIO
#define FLX_SERVER_SHADE_PRIO 0              /* issue priority of the shade waves (s_setprio 0 .. 3) */
#endif
#ifndef FLX_SERVER_PRIO
#define FLX_SERVER_PRIO 1                    /* waves that hold paths of the oldest frame run at a raised priority once its tile queue is dry */
#endif
constexpr uint32_t SV_S = SV_MAX_DEPTH;
enum { RK_SHADE = 0, RK_WALK = 1, RK_READY = 2 };
/* control words of a workgroup (LDS); everything per slot */
enum { SC_ALIVE = 0, SC_RING = SC_ALIVE + SV_S /* [kind][slot] x (tail, head, avail) */, SC_TILEDRY = SC_RING + 9 * SV_S, SC_SAVAIL = SC_TILEDRY + SV_S /* the view of the slot's frame is in LDS */,
       SC_SEQ = SC_SAVAIL + SV_S /* sequence number of the frame this workgroup has in the slot */, SC_SLOTP = SC_SEQ + SV_S /* the slot of its oldest frame */, SC_ROTLOCK, SC_EXIT, SC_STOPPED,
       SC_STOPAFTER, SC_LASTWORK, SC_NTILES /* [slot] screen tiles this workgroup made of the slot's frame */, SC_WORDS = 64 };
static_assert(SC_NTILES + SV_S <= SC_WORDS, "control words");
constexpr uint32_t SC_VIEW_WORDS = 64;

typedef const __attribute__((address_space(4))) ServerKernelArgs *ServerKernelArgsP;
FLX_DEV const ServerArgs &server_args(FrameArgsP p) {
  ServerKernelArgsP q = (ServerKernelArgsP)p;
  asm volatile("" : "+s"(q));
  return *(const ServerArgs *)&q->sa;
}
#define FLX_SERVER_ARGS() FLX_ARGS_OF(ab); const ServerArgs &sa = server_args(ab); (void)sa

/* VER: the scene MOVES — the lights and the transforms travel with the frame (ServerMail::blob: the host posts them with the view).  The launch keeps one version
 * of those arrays per (workgroup, slot): DeviceScene::rotation / shift / lights point at the launch's own version buffer, written by the wave that brings the slot's
 * view into the workgroup and read by that workgroup alone (plain loads: one CU, one vector L1; the inverse transforms of the walk waves go to LDS per slot). */
template <bool VER>
__global__ __launch_bounds__(FLX_WF_WALK_THREADS, FLX_WF_WAVES_PER_EU) void k_wf_server(ServerKernelArgs /* read through kernel_frame_args() */, uint32_t ldsCount, uint32_t nTransforms,
                                                                                     uint32_t shadeWaves, uint32_t readyUnits) {
  const FrameArgsP ab = kernel_frame_args();
  constexpr uint32_t WAVES = FLX_WF_WALK_THREADS / 64u;
  const uint32_t WALK_WAVES = WAVES - shadeWaves;
  uint32_t samples, depth, itemsPerSlot, tilesPerSlot;
  { FLX_SERVER_ARGS(); samples = (uint32_t)fr.samples; depth = sa.depth; itemsPerSlot = sa.itemsPerSlot; tilesPerSlot = sa.tilesPerSlot; }
  /* LDS: [tree top][inverse transforms (VER: per slot)][control words][the slots' views][per walk thread: nTransforms x 40 B of rays] */
  extern __shared__ float4 ldsAll[];
  float4 *ldsEntries = ldsAll;
  float4 *ldsXf = ldsAll + (size_t)ldsCount * 3u;
  uint32_t *ctl = (uint32_t *)(ldsXf + (size_t)nTransforms * 4u * (VER ? depth : 1u));
  FrameView *lv = (FrameView *)(ctl + SC_WORDS);
  float2 *raysBase = (float2 *)(ctl + SC_WORDS + SC_VIEW_WORDS);
  uint32_t *rings;
  {
    FLX_SERVER_ARGS();
    rings = wb.frameRings + (size_t)blockIdx.x * SV_RINGS * FQ_SIZE;
    for (uint32_t t = threadIdx.x; t < ldsCount * 3u; t += FLX_WF_WALK_THREADS) ldsEntries[t] = sc.walk[t];
    if (!VER)
    for (uint32_t t = threadIdx.x; t < nTransforms * 4u; t += FLX_WF_WALK_THREADS) {
      const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
      ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
    }
    if (threadIdx.x < (uint32_t)SC_WORDS) {
      uint32_t v = 0u;
      const uint32_t w = threadIdx.x;
      if (w >= (uint32_t)SC_SEQ && w < (uint32_t)SC_SEQ + SV_S) { const uint32_t slot = w - SC_SEQ; v = sa.seq0 + (slot >= sa.slot0 ? slot - sa.slot0 : slot + depth - sa.slot0); }
      if (w == (uint32_t)SC_SLOTP) v = sa.slot0;
      if (w == (uint32_t)SC_LASTWORK) v = (uint32_t)wall_clock64();
      ctl[w] = v;
    }
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t perTile = samples * 64u;
  WorkCounters cnt = {};

  auto ring = [&](uint32_t kind, uint32_t slot) -> uint32_t * { return rings + (size_t)(kind * SV_S + slot) * FQ_SIZE; };
  auto rctl = [&](uint32_t kind, uint32_t slot) -> uint32_t * { return ctl + SC_RING + 3u * (kind * SV_S + slot); };
  auto slotOf = [&](uint32_t id) -> uint32_t { return (id >= itemsPerSlot ? 1u : 0u) + (id >= 2u * itemsPerSlot ? 1u : 0u); };      /* ids are stacked by slot */
  auto slotAt = [&](uint32_t P, uint32_t r) -> uint32_t { const uint32_t s = P + r; return s >= depth ? s - depth : s; };          /* the slot of the r-th oldest frame */
  auto aliveAll = [&]() -> uint32_t { return fq_load(&ctl[SC_ALIVE]) + fq_load(&ctl[SC_ALIVE + 1]) + fq_load(&ctl[SC_ALIVE + 2]); };
  auto statAdd = [&](int word, unsigned long long v) { FLX_SERVER_ARGS(); if (sa.stats && lane == 0) atomicAdd(sa.stats + word, v); };
  auto worked = [&]() { if (lane == 0) __hip_atomic_store(&ctl[SC_LASTWORK], (uint32_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto giveUp = [&](uint32_t code) {
    FLX_SERVER_ARGS();
    if (sa.stats && fq_load(&ctl[SC_EXIT]) == 0u) {                     /* the first workgroups that give up leave their control words behind (flx_get_server_stats) */
      unsigned long long at = SV_DUMP_MAX;
      if (lane == 0) at = atomicAdd(sa.stats + SVS_DUMPS, 1ull);
      at = __shfl(at, 0, 64);
      if (at < (unsigned long long)SV_DUMP_MAX) {
        unsigned long long *d = sa.stats + SV_STAT_WORDS + at * SV_DUMP_WORDS;
        d[2 + lane] = ctl[lane];
        if (lane == 0) { d[0] = blockIdx.x | ((unsigned long long)__hip_atomic_load(&sa.mail->posted[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) << 32);
                         d[1] = wave | ((unsigned long long)__hip_atomic_load(&sa.mail->posted[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) << 32); d[66] = __hip_atomic_load(&sa.relay->posted[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); d[67] = __hip_atomic_load(&sa.relay->posted[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                         d[68] = __hip_atomic_load(&sa.relay->posted[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); d[69] = __hip_atomic_load(&sa.slots[0].tileNext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                         d[70] = __hip_atomic_load(&sa.slots[1].tileNext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); d[71] = __hip_atomic_load(&sa.slots[2].tileNext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
    }
    if (lane == 0) { __hip_atomic_fetch_or(sa.error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&ctl[SC_EXIT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  };
  if (wave == 0u && blockIdx.x == 0u) { FLX_SERVER_ARGS(); if (sa.stats && lane == 0) sa.stats[SVS_START] = (unsigned long long)wall_clock64(); }

  /* Is the view of the frame this workgroup has in `slot` here?  The host posts (view, then sequence number) into pinned HOST memory; a few waves of the grid
   * read that and pass it on in device memory, where the others look (agent-scope loads and stores: past the L1, valid across XCDs). */
  auto sAvail = [&](uint32_t slot, bool poll) -> bool {
    if (fq_load(&ctl[SC_SAVAIL + slot]) == 1u) return true;
    if (!poll) return false;
    FLX_SERVER_ARGS();
    const uint32_t want = fq_load(&ctl[SC_SEQ + slot]);
    if (blockIdx.x < (uint32_t)FLX_SERVER_RELAY_GROUPS) {
      uint32_t seq = 0, stop = 0;
      if (lane == 0) { seq = __hip_atomic_load(&sa.mail->posted[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); stop = __hip_atomic_load(&sa.mail->stopAfter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
      if (lane == 0 && stop != 0u) __hip_atomic_store(&sa.relay->stopAfter, stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_readfirstlane(seq) == want) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                    /* (system scope: the host wrote the view and the blob before the number) */
        if (lane < 19u) {
          const uint32_t v = __hip_atomic_load((const uint32_t *)&sa.mail->view[slot] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store((uint32_t *)&sa.relay->view[slot] + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (VER)
          for (uint32_t t = lane; t < sa.blobWords; t += 64u)
            __hip_atomic_store(&sa.relay->blob[slot][t], __hip_atomic_load(&sa.mail->blob[slot][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        /* seqlock: a wave that was held up in the middle of the copy (its siblings relayed the frame, the frame completed, the host posted the slot again) has copied a mix
         * of two posts — it says nothing then, and never moves the relay's number back */
        uint32_t again = 0;
        if (lane == 0) again = __hip_atomic_load(&sa.mail->posted[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool still = __builtin_amdgcn_readfirstlane(again) == want && fq_load(&ctl[SC_SEQ + slot]) == want;
        if (still && lane == 0) __hip_atomic_store(&sa.relay->posted[slot], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    uint32_t seq = 0, stop = 0;
    if (lane == 0) { seq = __hip_atomic_load(&sa.relay->posted[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); stop = __hip_atomic_load(&sa.relay->stopAfter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (lane == 0 && stop != 0u) __hip_atomic_store(&ctl[SC_STOPAFTER], stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    seq = __builtin_amdgcn_readfirstlane(seq);
    if (seq != want) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                     /* the relay's view and blob were stored before its number */
    if (fq_load(&ctl[SC_SAVAIL + slot]) == 1u) return true;                 /* a sibling wave got here first: the view is in place, nothing to copy twice */
    if (lane < 19u) ((uint32_t *)&lv[slot])[lane] = __hip_atomic_load((const uint32_t *)&sa.relay->view[slot] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (VER) {
      /* the frame's lights and transforms: into this workgroup's version of the slot (global memory it alone reads and writes) and, the inverse transforms, into LDS.
       * Blob: [rotation: 2 x 3 float4 per transform][shift: 2 float4 per transform][lights: 6 floats each] */
      const uint32_t nRot = nTransforms * 24u, nSh = nTransforms * 8u, ver = blockIdx.x * depth + slot;
      uint32_t *rotW = (uint32_t *)const_cast<float4 *>(sc.rotation) + (size_t)ver * nRot;
      uint32_t *shW = (uint32_t *)const_cast<float4 *>(sc.shift) + (size_t)ver * nSh;
      uint32_t *ltW = (uint32_t *)const_cast<float *>(sc.lights) + (size_t)ver * 6u * sc.n_lights;
      uint32_t *xfW = (uint32_t *)(ldsXf + (size_t)slot * nTransforms * 4u);
      for (uint32_t t = lane; t < sa.blobWords; t += 64u) {
        const uint32_t v = __hip_atomic_load(&sa.relay->blob[slot][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < nRot) {
          rotW[t] = v;
          const uint32_t q = t >> 2, i = q / 3u, k = q - 3u * i;
          if (i & 1u) xfW[((i >> 1) * 4u + k) * 4u + (t & 3u)] = v;
        } else if (t < nRot + nSh) {
          const uint32_t u = t - nRot, i = u >> 2;
          shW[u] = v;
          if (i & 1u) xfW[((i >> 1) * 4u + 3u) * 4u + (u & 3u)] = v;
        } else {
          ltW[t - nRot - nSh] = v;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (fq_load(&ctl[SC_SEQ + slot]) != want) return false;                /* (the slot went on to its next frame while this wave was copying: that frame's view is not this one) */
    if (lane == 0) __hip_atomic_store(&ctl[SC_SAVAIL + slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      /* (the slot's frame cannot change while its view is not here: a rotation needs it) */
    return true;
  };

  /* The front of a frame for one 8 x 8 screen tile of `slot` (k_wf_frame's makeTile), r = the age of its frame here: 0 not now, 1 a tile made, 2 no more tiles.
   * The tile's paths are counted alive BEFORE anything else: the workgroup cannot rotate the slot away under a tile in the making (the checks after the
   * count see a rotation that began before it). */
  auto makeTile = [&](uint32_t slot, uint32_t r) -> uint32_t {
    if (fq_load(&ctl[SC_TILEDRY + slot]) != 0u) return 2u;
    FLX_SERVER_ARGS();
    uint32_t take = 0, tile = 0;
    if (lane == 0 && fq_load(&ctl[SC_NTILES + slot]) >= sa.tileListCap) {
      /* the list of the tiles this workgroup has to resolve is full (a frame of mostly empty tiles, and this workgroup found them): the others make the rest;
       * all it still needs of the queue is to see it dry */
      if (fq_load(&ctl[SC_SAVAIL + slot]) == 1u && __hip_atomic_load(&sa.slots[slot].tileNext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= tilesPerSlot)
        __hip_atomic_store(&ctl[SC_TILEDRY + slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else
    if (lane == 0 && fq_load(&rctl(RK_READY, slot)[2]) < readyUnits) {
      const uint32_t mySeq = fq_load(&ctl[SC_SEQ + slot]);
      atomicAdd(&ctl[SC_ALIVE + slot], perTile);
      const bool still = fq_load(&ctl[SC_SEQ + slot]) == mySeq && fq_load(&ctl[SC_SAVAIL + slot]) == 1u && fq_load(&ctl[SC_TILEDRY + slot]) == 0u && fq_load(&ctl[SC_SEQ + slot]) == mySeq;
      if (!still || aliveAll() > FQ_ALIVE_MAX - r * (uint32_t)FLX_SERVER_RESERVE) atomicSub(&ctl[SC_ALIVE + slot], perTile);
      else { tile = __hip_atomic_fetch_add(&sa.slots[slot].tileNext, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); take = 1; }
    }
    take = __builtin_amdgcn_readfirstlane(take);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (take == 0u) return 0u;
    if (tile >= tilesPerSlot) {
      if (lane == 0) { __hip_atomic_store(&ctl[SC_TILEDRY + slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicSub(&ctl[SC_ALIVE + slot], perTile); }
      return 2u;
    }
    tile += slot * tilesPerSlot;                                        /* stacked: the tiles of slot s + 1 follow those of slot s */
    if (lane == 0) {                                                    /* this workgroup resolves the tile's pixels when it is through with the frame (tryRotate) */
      const uint32_t at = atomicAdd(&ctl[SC_NTILES + slot], 1u);
      if (at < sa.tileListCap) sa.tileLists[((size_t)blockIdx.x * SV_S + slot) * sa.tileListCap + at] = tile;
      else __hip_atomic_fetch_or(sa.error, WF_ERR_LIST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float4 h = primary_tile<false, true, VER>(ab, tile, lane, cnt, lv);
    const bool runs = shade0_tile<false, true, VER>(ab, tile, lane, h, cnt, lv);
    if (flx_ballot(runs) == 0ull) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            /* (what the tile's dead pixels stored, before the count) */
      if (lane == 0) atomicSub(&ctl[SC_ALIVE + slot], perTile);
    } else {
      for (uint32_t s0 = 0; s0 < samples; s0 += 64u)
        fq_push(ring(RK_READY, slot), rctl(RK_READY, slot), s0 + lane < samples, tile * samples + s0 + lane, lane);
    }
    statAdd(SVS_TILES, 1ull);
    return 1u;
  };

  /* The workgroup's oldest frame: nothing of it held, its tile queue dry?  Then the slot moves on to the frame `depth` later (one lane of one wave does
   * it), this workgroup's part of the frame is published, and the workgroup that publishes last tells the host. */
  auto tryRotate = [&]() {
    const uint32_t P = fq_load(&ctl[SC_SLOTP]);
    if (!(fq_load(&ctl[SC_SAVAIL + P]) == 1u && fq_load(&ctl[SC_TILEDRY + P]) != 0u && fq_load(&ctl[SC_ALIVE + P]) == 0u)) return;
    uint32_t mine = 0;
    if (lane == 0) mine = atomicCAS(&ctl[SC_ROTLOCK], 0u, 1u) == 0u ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(mine) == 0u) return;
    uint32_t ok = 0;
    if (lane == 0) ok = (fq_load(&ctl[SC_SLOTP]) == P && fq_load(&ctl[SC_SAVAIL + P]) == 1u && fq_load(&ctl[SC_TILEDRY + P]) != 0u && fq_load(&ctl[SC_ALIVE + P]) == 0u) ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(ok) != 0u) {
      FLX_SERVER_ARGS();
      /* The frame's pixels of THIS workgroup — the screen tiles it made: their paths never left it — are resolved here (fragment:608-632, flx_kernel_util.h:
       * resolve_pixel, what k_resolve does for a whole frame): no kernel beside the launch, and the frame is in its output buffer when the host hears of it. */
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const uint32_t nT = fq_load(&ctl[SC_NTILES + P]) < sa.tileListCap ? fq_load(&ctl[SC_NTILES + P]) : sa.tileListCap;
      const uint32_t *list = sa.tileLists + ((size_t)blockIdx.x * SV_S + P) * sa.tileListCap;
      const size_t stride = (size_t)fr.rows * fr.width;                 /* (stacked planes) */
      for (uint32_t t = 0; t < nT; t++) {
        uint32_t px, k;
        tile8_pixel(fr, list[t], lane, px, k);
        if (px < fr.width && k < fr.rows) {
          const size_t o = (size_t)k * fr.width + px;
          const float4 c = resolve_pixel(fr, wb.hits, wb.sampleRadiance, wb.lastOriginal, o, stride);
          uint32_t row = k - P * fr.frame_rows;
          if (sa.outStripRows) { const uint32_t strip = row / sa.outStripRows; row = strip * sa.outStripStep + (row - strip * sa.outStripRows); }
          if (sa.out8) ((uint32_t *)sa.out[P])[(size_t)row * fr.width + px] = pack_rgba8(c.x, c.y, c.z, c.w);
          else sa.out[P][(size_t)row * fr.width + px] = c;
        }
      }
      /* this workgroup's part of the frame is in its XCD's L2 at the latest: written back (to the memory of whoever owns the frame), then counted */
      if (sa.outSystem) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); else
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const uint32_t seqOld = fq_load(&ctl[SC_SEQ + P]);
        __hip_atomic_store(&ctl[SC_NTILES + P], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[SC_SEQ + P], seqOld + depth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[SC_SAVAIL + P], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&ctl[SC_TILEDRY + P], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[SC_SLOTP], P + 1u >= depth ? 0u : P + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t before = __hip_atomic_fetch_add(&sa.slots[P].groupsDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == gridDim.x) {
          /* the frame is complete: the slot's queue back to its start for the frame the host will post next, then the word the host waits for */
          __hip_atomic_store(&sa.slots[P].tileNext, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&sa.slots[P].groupsDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(&sa.mail->done[P], seqOld, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          if (sa.stats) atomicAdd(sa.stats + SVS_FRAMES, 1ull);
        }
        if (sa.stats) atomicAdd(sa.stats + SVS_ROTATIONS, 1ull);
      }
    }
    if (lane == 0) __hip_atomic_store(&ctl[SC_ROTLOCK], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  /* Through with the last frame the host will post, or nothing to do for milliseconds?  Then the workgroup ends. */
  auto checkExit = [&]() {
    const uint32_t stop = fq_load(&ctl[SC_STOPAFTER]);
    if (stop != 0u && aliveAll() == 0u) {
      /* (under the rotation's lock: the slot of the oldest frame and that slot's sequence number are two words, and a rotation changes both) */
      uint32_t out = 0;
      if (lane == 0 && atomicCAS(&ctl[SC_ROTLOCK], 0u, 1u) == 0u) {
        const uint32_t P = fq_load(&ctl[SC_SLOTP]);
        out = ((int32_t)(fq_load(&ctl[SC_SEQ + P]) - stop) > 0 && aliveAll() == 0u) ? 1u : 0u;
        if (out) __hip_atomic_store(&ctl[SC_EXIT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[SC_ROTLOCK], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (__builtin_amdgcn_readfirstlane(out) != 0u) return;
    }
    FLX_SERVER_ARGS();
    const uint32_t last = fq_load(&ctl[SC_LASTWORK]);
    if ((int32_t)((uint32_t)wall_clock64() - last) > (int32_t)sa.idleExit) giveUp(WF_ERR_SERVER_IDLE);      /* (nothing moved for seconds, whatever is held; signed: another wave may have stamped after this one read the clock) */
  };

  auto leave = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t last = 0;
    if (lane == 0) last = atomicAdd(&ctl[SC_STOPPED], 1u) + 1u == WAVES ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
    FLX_SERVER_ARGS();
    for (uint32_t k = 0; k < 9u; k++)                                   /* (a launch leaves its rings empty: every slot it popped was cleared) */
      if (fq_load(&ctl[SC_RING + 3u * k]) != fq_load(&ctl[SC_RING + 3u * k + 1u]) && lane == 0) __hip_atomic_fetch_or(sa.error, WF_ERR_LEFTOVER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (sa.stats && lane == 0) atomicMax(sa.stats + SVS_END, (unsigned long long)wall_clock64());
  };

  if (wave >= WALK_WAVES) {
    /* ================================ shade wave ================================ */
    uint32_t idle = 0;
    if (FLX_SERVER_SHADE_PRIO) __builtin_amdgcn_s_setprio(FLX_SERVER_SHADE_PRIO);
    const long long tShade0 = wall_clock64();
    long long tTile = 0, tBatch = 0;
    for (;;) {
      if (fq_load(&ctl[SC_EXIT]) != 0u) break;
      tryRotate();
      const uint32_t P = fq_load(&ctl[SC_SLOTP]);
      bool did = false;
      /* ---- a batch of paths to shade, from every frame's ring, the oldest frame's first: when 64 wait in all, or one of the oldest frame once its fresh
       * paths are all out (its last chains set its end), or anything at all once this wave has found nothing else to do.  (Batches are not per frame: a
       * workgroup seldom has 64 paths of ONE frame waiting, and a batch costs its wave the same time whatever its size.) ---- */
      uint32_t nAvail = 0, waiting = 0;
      for (uint32_t r = 0; r < depth; r++) {
        const uint32_t slot = slotAt(P, r);
        if (!sAvail(slot, (idle & 7u) == 0u)) break;      /* (a frame that is not posted yet: nor are the ones after it) */
        nAvail = r + 1u;
        waiting += fq_load(&rctl(RK_SHADE, slot)[2]);
      }
      if (nAvail != 0u && waiting != 0u &&
          (waiting >= 64u || idle != 0u || (fq_load(&ctl[SC_TILEDRY + P]) != 0u && fq_load(&rctl(RK_SHADE, P)[2]) != 0u))) {
        uint32_t id = WF_INVALID, got = 0;
        for (uint32_t r = 0; r < nAvail && got < 64u; r++) {
          const uint32_t slot = slotAt(P, r);
          const unsigned long long takers = got == 0u ? ~0ull : (~0ull << got);
          got += fq_pop(ring(RK_SHADE, slot), rctl(RK_SHADE, slot), takers, 64u - got, 1u, lane, id);
        }
        if (got != 0u) {
          const long long tb0 = wall_clock64();
          const bool mine = lane < got && id != WF_INVALID;
          statAdd(SVS_BATCHES, 1ull); statAdd(SVS_BATCH_LANES, got);
          if (mine) shade_path<false, true, VER>(ab, id, cnt, lv);
          const uint32_t slotMine = slotOf(id);
          for (uint32_t sl = 0; sl < depth; sl++) fq_push(ring(RK_WALK, sl), rctl(RK_WALK, sl), mine && slotMine == sl, id, lane);
          tBatch += wall_clock64() - tb0;
          did = true;
        }
      }
      /* ---- else the front of a frame for one screen tile, the oldest frame that still has tiles first ---- */
      for (uint32_t r = 0; r < nAvail && !did; r++) {
        const long long tt0 = wall_clock64();
        if (makeTile(slotAt(P, r), r) == 1u) { tTile += wall_clock64() - tt0; did = true; }
      }
      if (did) { idle = 0; worked(); continue; }
      checkExit();
      idle++;
      __builtin_amdgcn_s_sleep(8);
    }
    statAdd(SVS_SHADE_TILE_T, (unsigned long long)tTile); statAdd(SVS_SHADE_BATCH_T, (unsigned long long)tBatch); statAdd(SVS_SHADE_TOTAL_T, (unsigned long long)(wall_clock64() - tShade0));
    leave();
    return;
  }

  /* ================================ walk wave ================================ */
  float2 *myRays = raysBase + (size_t)threadIdx.x * nTransforms * 5u;
  const float4 *walkG; { FLX_ARGS_OF(ab); walkG = pinnedWalkCopy(sc); }      /* the global copy of the tree, for the entries beyond the LDS top: in registers for the stepping loop */
  /* ... both homes of an entry as flat addresses in VECTOR registers (flx_wavefront.hip: FLX_WF_FETCH_VBASE) */
  const float4 *ldsEntriesV = ldsEntries, *walkGV = walkG;
  asm volatile("" : "+v"(ldsEntriesV), "+v"(walkGV));
  WalkLane L;                                                  /* the lane's path and its walks (flx_frame_common.h: the body k_wf_frame's walk waves run) */
  walkLaneInit(L);
  uint32_t chunkNext = 0, chunkEnd = 0;          /* wave-uniform: the rest of a (tile, sample) unit of fresh paths */
  uint32_t statTrips = 0, statLaneTrips = 0;

  for (;;) {
    const unsigned long long walking = flx_ballot(L.st == P_WALKING);
    const unsigned long long workMask = flx_ballot(L.st == P_DONE || L.st == P_SWITCH);
    const uint32_t parked = 64u - (uint32_t)__popcll(walking);
    bool mayRefill = chunkNext != chunkEnd;
    for (uint32_t s = 0; s < depth && !mayRefill; s++) mayRefill = fq_load(&rctl(RK_WALK, s)[2]) != 0u || fq_load(&rctl(RK_READY, s)[2]) != 0u;
    statTrips += (uint32_t)FLX_WF_INNER; statLaneTrips += (uint32_t)__popcll(walking) * (uint32_t)FLX_WF_INNER;
    if (walking == 0ull || (parked >= (uint32_t)FLX_WF_BATCH && (workMask != 0ull || mayRefill))) {
      /* ---- the free lanes' NEXT paths first — ids (LDS only), then their records' loads, issued here and not waited for: the fold of the lanes' old paths below has its
       * own loads, and one wait covers both round trips (flx_wavefront.hip: FLX_FRAME_EARLY_REFILL).  Older frames before younger ones; per frame the paths that came back
       * from shading, then fresh ones. ---- */
      uint32_t newId = WF_INVALID;
      bool newFresh = false;
      WalkRecord newRec;
      newRec.q0 = newRec.q1 = newRec.q2 = newRec.q3 = make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t P = fq_load(&ctl[SC_SLOTP]);
      {
        const bool want = L.st == P_EMPTY || L.st == P_DONE;      /* (a lane at P_DONE is free once it is folded, whatever becomes of its path) */
        for (;;) {
          const unsigned long long idle = flx_ballot(want && newId == WF_INVALID);
          if (idle == 0ull) break;
          const bool mineIdle = ((idle >> lane) & 1ull) != 0ull;
          const uint32_t nIdle = (uint32_t)__popcll(idle);
          const uint32_t rk = lane_rank(idle);
          bool got = false, any = false;
          for (uint32_t r = 0; r < depth; r++) {
            const uint32_t slot = slotAt(P, r);
            if (fq_load(&ctl[SC_SAVAIL + slot]) != 1u) break;
            uint32_t id = WF_INVALID;
            if (fq_pop(ring(RK_WALK, slot), rctl(RK_WALK, slot), idle, nIdle, 1u, lane, id) != 0u) {
              if (mineIdle && id != WF_INVALID) newId = id;
              any = flx_ballot(mineIdle && id != WF_INVALID) != 0ull;
              got = true;
              break;
            }
            if (chunkNext == chunkEnd && fq_load(&rctl(RK_SHADE, slot)[2]) < FQ_LIMIT) {      /* (the shade waves are not behind with this frame) */
              uint32_t unit = WF_INVALID;
              if (fq_pop(ring(RK_READY, slot), rctl(RK_READY, slot), 1ull, 1u, 1u, lane, unit) != 0u) {
                unit = __builtin_amdgcn_readfirstlane(unit);
                if (unit != WF_INVALID) { chunkNext = unit << 6; chunkEnd = chunkNext + 64u; }
                else if (lane == 0) { FLX_SERVER_ARGS(); atomicSub(&ctl[SC_ALIVE + slot], 64u); __hip_atomic_fetch_or(sa.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
              }
            }
            if (chunkNext != chunkEnd && slotOf(chunkNext) == slot) {
              const uint32_t avail = chunkEnd - chunkNext;
              const uint32_t take = nIdle < avail ? nIdle : avail;
              if (mineIdle && rk < take) { newId = chunkNext + rk; newFresh = true; }
              chunkNext += take;
              got = true; any = true;
              break;
            }
          }
          if (!got || !any) break;
        }
        if (newId != WF_INVALID) { FLX_ARGS_OF(ab); walkLaneFetchRecord(fr, wb, newId, newFresh, newRec); }
      }
      /* ---- fold the finished lanes (FLX_WALK_LANE_FOLD); a path that goes on is handed to the shade waves of its frame's slot ---- */
      if (flx_ballot(L.st == P_DONE) != 0ull) {
        FLX_ARGS_OF(ab);
        const bool compactRecs = wb.rec0 != nullptr;
        bool toShade = false, ended = false;
        const uint32_t slot = slotOf(L.pathId);
        if (L.st == P_DONE) FLX_WALK_LANE_FOLD(true, fr, wb, compactRecs, L, lv, toShade, ended, false, 0u);
        if (flx_ballot(ended) != 0ull) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      /* the radiance stored, before the paths leave the count (tryRotate publishes what is counted off) */
        for (uint32_t s = 0; s < depth; s++) {
          fq_push(ring(RK_SHADE, s), rctl(RK_SHADE, s), toShade && slot == s, L.pathId, lane);
          const uint32_t e = (uint32_t)__popcll(flx_ballot(ended && slot == s));
          if (e != 0u && lane == 0) atomicSub(&ctl[SC_ALIVE + s], e);
        }
        worked();
      }
      /* ... the lanes' next paths: their records arrived while the old ones were folded */
      {
        bool dead = false;
        if (newId != WF_INVALID) dead = walkLaneInstall<false>(newId, newRec, L, cnt);
        if (flx_ballot(dead) != 0ull) {
          const uint32_t dslot = slotOf(newId);
          for (uint32_t sl = 0; sl < depth; sl++) {
            const uint32_t nDead = (uint32_t)__popcll(flx_ballot(dead && dslot == sl));
            if (nDead != 0u && lane == 0) atomicSub(&ctl[SC_ALIVE + sl], nDead);      /* (a dead item's finalisation was stored by the wave that made the tile, before the tile was handed over) */
          }
        }
      }
      if (FLX_SERVER_PRIO) {
        const bool tail = fq_load(&ctl[SC_TILEDRY + P]) != 0u;
        if (tail && flx_ballot(L.st != P_EMPTY && slotOf(L.pathId) == P) != 0ull) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
      }
      /* ---- set up walks: fresh lanes (shadow or closest) and lanes whose shadow walk just ended ---- */
      walkLaneSwitch(L);
      if (flx_ballot(L.st == P_SETUP) != 0ull) {
        FLX_ARGS_OF(ab);
        if (L.st == P_SETUP) walkLaneSetup<false>(sc, nTransforms, VER ? ldsXf + (size_t)slotOf(L.pathId) * nTransforms * 4u : ldsXf, myRays, walkGV, ldsEntriesV, ldsCount, L, cnt);
      }
      if (flx_ballot(L.st == P_WALKING) == 0ull) {
        if (flx_ballot(L.st != P_EMPTY) != 0ull) continue;      /* lanes that had nothing to walk wait for the fold */
        /* nothing in this wave: wait for the shade waves (or for the next frame), or end with the workgroup */
        if (fq_load(&ctl[SC_EXIT]) != 0u) break;
        tryRotate();
        if (!mayRefill) __builtin_amdgcn_s_sleep(8);
        continue;
      }
    }
    /* ---- FLX_WF_INNER entries for every walking lane (the few scene words the fetch needs are read before the loop) ---- */
    {
#pragma unroll FLX_WF_UNROLL
      for (int it = 0; it < FLX_WF_INNER; it++) FLX_WALK_LANE_STEP(false, walkGV, ldsEntriesV, ldsCount, myRays, L, cnt);
    }
  }
  statAdd(SVS_WALK_TRIPS, statTrips); statAdd(SVS_WALK_LANE_TRIPS, statLaneTrips);
  leave();
}

bool server_kernel_fits(const DeviceScene &sc, uint32_t &ldsCount, uint32_t &ldsBytes, uint32_t xfSlots, uint32_t shadeWaves) {
  const uint32_t T = sc.n_transforms;
  if (shadeWaves == 0u) shadeWaves = (uint32_t)FLX_SERVER_SHADERS;
  const uint32_t walkThreads = FLX_WF_WALK_THREADS - 64u * shadeWaves;
  const uint32_t fixed = walkThreads * T * 40u + xfSlots * T * 64u + (SC_WORDS + SC_VIEW_WORDS) * 4u;
  if (fixed + 4096u > (uint32_t)FLX_WF_LDS_TOTAL) return false;
  ldsCount = ((uint32_t)FLX_WF_LDS_TOTAL - fixed) / 48u;
  if (ldsCount > sc.walk_hot) ldsCount = sc.walk_hot;
  ldsBytes = ldsCount * 48u + fixed;
  return true;
}

size_t server_rings_per_group() { return (size_t)SV_RINGS * FQ_SIZE; }

int launch_server(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, const ServerArgs &sa, uint32_t compute_units, hipStream_t stream) {
  uint32_t ldsCount = 0, ldsBytes = 0;
  const bool ver = sa.blobWords != 0u;                         /* the scene moves: its lights and transforms come with every frame */
  if (sa.depth < 2u || sa.depth > SV_MAX_DEPTH) return -1;
  /* (two slots of THIN frames: a whole frame through two slots wants its three shade waves, 6.49 against 7.23 ms) */
  const uint32_t shadeWaves = sa.depth == 2u && sa.tilesPerSlot / compute_units <= (uint32_t)FLX_SERVER_SHADERS_DEPTH2_MAX_TILES ? (uint32_t)FLX_SERVER_SHADERS_DEPTH2 : (uint32_t)FLX_SERVER_SHADERS;
  if (!server_kernel_fits(sc, ldsCount, ldsBytes, ver ? sa.depth : 1u, shadeWaves)) return -1;
  if (ver && (sa.blobWords > SV_BLOB_WORDS || sa.blobWords != server_blob_words(sc.n_transforms, sc.n_lights))) return -1;
  static std::once_flag once[64];
  static bool ok[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::call_once(once[dev], [&]() { ok[dev] = hipFuncSetAttribute((const void *)k_wf_server<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                                              hipFuncSetAttribute((const void *)k_wf_server<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; });
  if (!ok[dev]) return -1;
  ServerKernelArgs ka;
  ka.fa.sc = sc; ka.fa.fr = fr; ka.fa.wb = wb; ka.sa = sa;
  const uint32_t tilesPerGroup = sa.tilesPerSlot / compute_units;
  uint32_t readyUnits = tilesPerGroup >= 48u ? (uint32_t)FLX_FRAME_READY_UNITS : tilesPerGroup / 2u;
  readyUnits = readyUnits < (uint32_t)FLX_FRAME_READY_UNITS / 4u ? (uint32_t)FLX_FRAME_READY_UNITS / 4u : (readyUnits > (uint32_t)FLX_FRAME_READY_UNITS ? (uint32_t)FLX_FRAME_READY_UNITS : readyUnits);
  if (ver) hipLaunchKernelGGL(k_wf_server<true>, dim3(compute_units), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, ka, ldsCount, sc.n_transforms, shadeWaves, readyUnits);
  else hipLaunchKernelGGL(k_wf_server<false>, dim3(compute_units), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, ka, ldsCount, sc.n_transforms, shadeWaves, readyUnits);
  return 0;
}

}  // namespace flx
