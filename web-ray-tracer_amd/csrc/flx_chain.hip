/*
 * flx_chain.hip — the chained frame kernel (round 4): consecutive frames of the frame loop overlap INSIDE the persistent launch.
 *
 * The frame kernel (flx_wavefront.hip: k_wf_frame) renders one frame per launch, and every launch ends in a drain: once the frame's tile queue is dry a
 * workgroup finishes the ~1 500 paths it holds alone, at the pace of their chains walk -> shade -> walk (0.4 - 0.5 ms).  For a whole 1080p frame that is 7 %
 * of the launch; for a rank's eighth of it — what each of 8 GPUs renders — it is a third (1.65 ms per frame where the bulk is 0.8: 3.9x at 8 GPUs,
 * profiles/r03_share_scaling.txt).  Two launches cannot share a CU (one 1 024-thread workgroup takes its registers and its LDS), so the next frame cannot
 * fill the lanes the drain leaves idle — unless the SAME launch works on it.  The reference's loop renders frame after frame from one context
 * (modules/pathtracerWGL2.js:254-303, the shader's loop per pixel: shaders/pathtracer_fragment.glsl:475-596); nothing in the algorithm ties a launch to a frame.
 *
 * So the frame loop keeps its frames in flight — two or three: the chain's DEPTH — as frame slots of one stacked workspace (the layout of a batch of frames:
 * DeviceFrame::frames = depth; path ids, screen tiles and packed rows of slot s + 1 follow those of slot s), and the kernel of frame k — K(k) — has a role for each:
 *
 *   P   (role 0) the slot of frame k: K(k) must complete it.  Its sources: the resume lists K(k - 1) left (below) and the rest of its tile queue.
 *   S1  (role 1) the slot of frame k + 1, S2 (role 2, depth 3) that of frame k + 2: work for lanes that the older frames cannot fill.  Their cameras need
 *       not be known when K(k) is launched; flx_frame_begin of a later frame POSTS its view into a mailbox (pinned host memory the host writes with plain
 *       stores; tools/micro/mailbox.hip; a few waves relay it into device memory) and the workgroups pick it up when they first run out of older work.
 *       Every source of an older frame is tried before any of a younger one, and role r may not take the last r x FLX_CHAIN_RESERVE places of a workgroup's rings.
 *
 * Why three: a frame's paths are chains (walk, shade, walk, ...); the longest of a 1080p dragon share take 1.3 - 2 ms from the frame's first tile under load,
 * the share's work is 0.8 ms of the machine.  With two slots a launch ends up waiting for its own frame's last chains with nothing else to do (measured: a
 * frame every 1.45 ms, no better than two separate lanes); with three the machine always has a younger frame's bulk to work on.
 *
 * K(k) ends when P is complete on every workgroup (a counter in device memory that the workgroups add to when they hold nothing of P any more, polled by
 * one wave of each such workgroup).  What a workgroup then holds of S is NOT finished — that would be the drain again — but handed to K(k + 1), for which that
 * slot is P: the ids in its rings go to three global resume lists (paths to shade, paths to walk, fresh (tile, sample) units); a walk in flight is abandoned
 * and its path listed for a walk from the start (its record is untouched until the walk is folded: ~3 % of a frame's visits are walked twice; counted frames
 * therefore never run chained).  The kernel boundary between K(k) and K(k + 1) makes all of it visible: no device-wide fence inside the launch.
 * Every kernel is a finite launch in stream order; a frame nobody follows simply drains as before (S never becomes available), and K(k + 1) launched into an
 * already finished slot finds its queue dry and goes straight to the frame after.
 *
 * Per path nothing changes — the same records, arithmetic and order as in k_wf_frame, the radiance in the path's own slot — so the frames are bit-identical to
 * single renders (tests/test_chain_gpu.py).
 */
#include <cstdio>
#include "flx_chain.h"
#include "flx_frame_common.h"

namespace flx {

#ifndef FLX_CHAIN_PRIO
#define FLX_CHAIN_PRIO 1                    /* waves that hold paths of P run at a raised priority once P's fresh sources are dry */
#endif
#ifndef FLX_CHAIN_RELAY_GROUPS
#define FLX_CHAIN_RELAY_GROUPS 16            /* workgroups whose shade waves read the host's mailbox and pass the post on in device memory */
#endif
constexpr uint32_t CH_R = CH_MAX_DEPTH;
enum { CL_WALK = 0, CL_SHADE = 1, CL_READY = 2, CL_SUSP = 3, CL_ALL = 15 };
enum { P_RESUMED = 6 };                       /* a lane state beside flx_wavefront_common.h's: a suspended walk taken up again, its rays not yet in LDS */
/* control words of a workgroup (LDS) */
enum { CC_ALIVE = 0 /* [role] live paths */, CC_RING = 3 /* [kind: shade, walk, ready][role] x (tail, head, avail) */, CC_TILEDRY = CC_RING + 9 * CH_R /* [role] */,
       CC_SAVAIL = CC_TILEDRY + CH_R /* [role]: 0 its view is not posted yet, 1 it is in LDS, 2 it never comes */, CC_LISTDRY = CC_SAVAIL + CH_R /* [role]: bits CL_* */,
       CC_LCOUNT = CC_LISTDRY + CH_R /* [role][list] entries of the resume lists read here */, CC_STOP = CC_LCOUNT + CH_R * CH_LISTS, CC_STOPPED, CC_PDONE, CC_PDONE_T, CC_WORDS = 64 };
static_assert(CC_PDONE_T < CC_WORDS, "control words");
#ifndef FLX_CHAIN_POST_WAIT_US
#define FLX_CHAIN_POST_WAIT_US 150          /* how long a launch that is through with its own frame waits for a younger frame's view before it ends */
#endif
#ifndef FLX_CHAIN_QUOTA
#define FLX_CHAIN_QUOTA 1                   /* a launch makes a frame's worth of younger tiles before it ends (0: it ends when its own frame is complete) */
#endif
constexpr uint32_t CC_VIEW_WORDS = 64;        /* three FrameViews (19 floats each) behind the control words */
enum { RK_SHADE = 0, RK_WALK = 1, RK_READY = 2 };

typedef const __attribute__((address_space(4))) ChainKernelArgs *ChainKernelArgsP;
FLX_DEV const ChainArgs &chain_args(FrameArgsP p) {
  ChainKernelArgsP q = (ChainKernelArgsP)p;
  asm volatile("" : "+s"(q));
  return *(const ChainArgs *)&q->ca;
}
#define FLX_CHAIN_ARGS() FLX_ARGS_OF(ab); const ChainArgs &ca = chain_args(ab); (void)ca

__global__ __launch_bounds__(FLX_WF_WALK_THREADS, FLX_WF_WAVES_PER_EU) void k_wf_chain(ChainKernelArgs /* read through kernel_frame_args() */, uint32_t ldsCount, uint32_t nTransforms,
                                                                                    uint32_t shadeWaves, uint32_t readyUnits) {
  const FrameArgsP ab = kernel_frame_args();
  constexpr uint32_t WAVES = FLX_WF_WALK_THREADS / 64u;
  const uint32_t WALK_WAVES = WAVES - shadeWaves;
  /* LDS: [tree top][inverse transforms][control words][the slots' views][per walk thread: nTransforms x 40 B of rays] */
  extern __shared__ float4 ldsAll[];
  float4 *ldsEntries = ldsAll;
  float4 *ldsXf = ldsAll + (size_t)ldsCount * 3u;
  uint32_t *ctl = (uint32_t *)(ldsXf + (size_t)nTransforms * 4u);
  FrameView *lv = (FrameView *)(ctl + CC_WORDS);
  float2 *raysBase = (float2 *)(ctl + CC_WORDS + CC_VIEW_WORDS);
  uint32_t samples, slotP, depth, itemsPerSlot, tilesPerSlot, rdSet;
  uint32_t *rings;
  {
    FLX_CHAIN_ARGS();
    samples = (uint32_t)fr.samples; slotP = ca.slotP; depth = ca.depth; itemsPerSlot = ca.itemsPerSlot; tilesPerSlot = ca.tilesPerSlot;
    rdSet = (ca.seqP & 1u) ^ 1u;                               /* the list set the kernel before this one wrote */
    rings = wb.frameRings + (size_t)blockIdx.x * CH_RINGS * FQ_SIZE;
    for (uint32_t t = threadIdx.x; t < ldsCount * 3u; t += FLX_WF_WALK_THREADS) ldsEntries[t] = sc.walk[t];
    for (uint32_t t = threadIdx.x; t < nTransforms * 4u; t += FLX_WF_WALK_THREADS) {
      const uint32_t tr = t >> 2, k = t & 3u, iI = 2u * tr + 1u;
      ldsXf[t] = k < 3u ? sc.rotation[3u * iI + k] : sc.shift[iI];
    }
    if (threadIdx.x < (uint32_t)CC_WORDS) {
      uint32_t v = 0u;
      const uint32_t w = threadIdx.x;
      /* roles 0 .. depth - 2 were worked ahead on by the kernel before: they have resume lists; the youngest is new */
      if (w >= (uint32_t)CC_LCOUNT && w < (uint32_t)CC_LCOUNT + CH_R * CH_LISTS) {
        const uint32_t r = (w - CC_LCOUNT) / CH_LISTS, l = (w - CC_LCOUNT) % CH_LISTS;
        if (r + 1u < depth) { uint32_t sl = slotP + r; if (sl >= depth) sl -= depth; v = ca.slots[sl].count[rdSet][l]; }
      }
      if (w >= (uint32_t)CC_LISTDRY && w < (uint32_t)CC_LISTDRY + CH_R) {
        const uint32_t r = w - CC_LISTDRY;
        v = CL_ALL;
        if (r + 1u < depth) { uint32_t sl = slotP + r; if (sl >= depth) sl -= depth; v = 0u; for (uint32_t l = 0; l < CH_LISTS; l++) if (ca.slots[sl].count[rdSet][l] == 0u) v |= 1u << l; }
      }
      if (w >= (uint32_t)CC_SAVAIL && w < (uint32_t)CC_SAVAIL + CH_R) { const uint32_t r = w - CC_SAVAIL; v = r == 0u ? 1u : (r <= ca.ahead && r < depth ? 0u : 2u); }
      ctl[w] = v;
    }
    if (threadIdx.x >= 64u && threadIdx.x < 64u + 19u) ((float *)&lv[slotP])[threadIdx.x - 64u] = ((const float *)&fr.view[slotP])[threadIdx.x - 64u];
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t perTile = samples * 64u;
  WorkCounters cnt = {};

  auto ring = [&](uint32_t kind, uint32_t r) -> uint32_t * { return rings + (size_t)(kind * CH_R + r) * FQ_SIZE; };
  auto rctl = [&](uint32_t kind, uint32_t r) -> uint32_t * { return ctl + CC_RING + 3u * (kind * CH_R + r); };
  auto slotOfRole = [&](uint32_t r) -> uint32_t { uint32_t sl = slotP + r; return sl >= depth ? sl - depth : sl; };
  auto roleOf = [&](uint32_t id) -> uint32_t {                            /* 0: a path of P, 1, 2: of the frames after it (ids are stacked by slot) */
    const uint32_t x = id & ~CH_FRESH;
    const uint32_t sl = (x >= itemsPerSlot ? 1u : 0u) + (x >= 2u * itemsPerSlot ? 1u : 0u);
    return sl >= slotP ? sl - slotP : sl + depth - slotP;
  };
  auto aliveAll = [&]() -> uint32_t { return fq_load(&ctl[CC_ALIVE]) + fq_load(&ctl[CC_ALIVE + 1]) + fq_load(&ctl[CC_ALIVE + 2]); };
  auto giveUp = [&](uint32_t code) {                                    /* a watchdog: the frame is wrong, and the host is told (flx_frame_end: FLX_ERR_DEVICE) */
    FLX_CHAIN_ARGS();
    if (lane == 0) { __hip_atomic_fetch_or(ca.error, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&ctl[CC_STOP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  };
  auto statAdd = [&](int word, unsigned long long v) { FLX_CHAIN_ARGS(); if (ca.stats && lane == 0) atomicAdd(ca.stats + word, v); };
  auto statMin = [&](int word) { FLX_CHAIN_ARGS(); if (ca.stats && lane == 0) atomicMin(ca.stats + word, (unsigned long long)wall_clock64()); };
  auto statMax = [&](int word) { FLX_CHAIN_ARGS(); if (ca.stats && lane == 0) atomicMax(ca.stats + word, (unsigned long long)wall_clock64()); };
  if (wave == 0u) statMin(CS_START_MIN);
  auto listDry = [&](uint32_t r) -> bool { return (fq_load(&ctl[CC_LISTDRY + r]) & (uint32_t)CL_ALL) == (uint32_t)CL_ALL; };
  auto listBase = [&](const ChainArgs &ca, uint32_t slot, uint32_t set, uint32_t list) -> uint32_t * { return ca.lists + (size_t)((slot * 2u + set) * 3u + list) * ca.listCap; };
  auto suspBase = [&](const ChainArgs &ca, uint32_t slot, uint32_t set) -> float4 * { return ca.susp + (size_t)(slot * 2u + set) * ca.suspCap * CH_SUSP_F4; };

  /* May this workgroup work on role r's frame (r >= 1)?  Its view must have been posted (the sequence number says for which frame).  The host posts into
   * pinned HOST memory; only a few waves of the grid read that (hundreds of waves polling across PCIe take milliseconds per read): they relay the post into
   * device memory, where everybody else looks (agent-scope loads and stores: past the L1, valid across XCDs). */
  auto sAvail = [&](uint32_t r, bool poll) -> bool {
    const uint32_t a = fq_load(&ctl[CC_SAVAIL + r]);
    if (a == 1u) return true;
    if (a == 2u || !poll) return false;
    FLX_CHAIN_ARGS();
    const uint32_t slot = slotOfRole(r), want = ca.seqP + r;
    if (blockIdx.x < (uint32_t)FLX_CHAIN_RELAY_GROUPS) {
      uint32_t seq = 0;
      if (lane == 0) seq = __hip_atomic_load(&ca.mail->posted[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (__builtin_amdgcn_readfirstlane(seq) == want) {
        if (lane < 19u) {                                                 /* (the view was written before the number) */
          const uint32_t v = __hip_atomic_load((const uint32_t *)&ca.mail->view[slot] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store((uint32_t *)&ca.relay->view[slot] + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ca.relay->posted[slot], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    uint32_t seq = 0;
    if (lane == 0) seq = __hip_atomic_load(&ca.relay->posted[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seq = __builtin_amdgcn_readfirstlane(seq);
    if (seq != want) return false;
    if (lane < 19u) ((uint32_t *)&lv[slot])[lane] = __hip_atomic_load((const uint32_t *)&ca.relay->view[slot] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&ctl[CC_SAVAIL + r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (r == 1u) { statMin(CS_SAVAIL_MIN); statMax(CS_SAVAIL_MAX); } else statMin(CS_SAVAIL2_MIN);
    return true;
  };

  /* `want` entries of one of role r's resume lists (each standing for `each` live paths): -> how many (uniform), starting at `first`.  Counted alive BEFORE they are taken. */
  auto pullList = [&](uint32_t r, uint32_t list, uint32_t want, uint32_t each, uint32_t &first) -> uint32_t {
    if ((fq_load(&ctl[CC_LISTDRY + r]) >> list) & 1u) return 0u;
    uint32_t n = 0, t = 0;
    if (lane == 0) {
      FLX_CHAIN_ARGS();
      const uint32_t have = fq_load(&ctl[CC_LCOUNT + r * CH_LISTS + list]);
      const uint32_t add = want * each;
      const uint32_t before = atomicAdd(&ctl[CC_ALIVE + r], add);
      if (aliveAll() > FQ_ALIVE_MAX - r * (uint32_t)FLX_CHAIN_RESERVE && before != 0u) atomicSub(&ctl[CC_ALIVE + r], add);      /* (no room; a role that holds nothing may always take one draw) */
      else {
        t = atomicAdd(&ca.slots[slotOfRole(r)].taken[rdSet][list], want);
        n = t >= have ? 0u : (have - t < want ? have - t : want);
        if (n < want) atomicSub(&ctl[CC_ALIVE + r], (want - n) * each);
        if (t + want >= have) atomicOr(&ctl[CC_LISTDRY + r], 1u << list);
        if (ca.stats && r == 0u && n != 0u) atomicAdd(ca.stats + CS_P_PULL + list, (unsigned long long)n);
      }
    }
    n = __builtin_amdgcn_readfirstlane(n);
    first = __builtin_amdgcn_readfirstlane(t);
    return n;
  };

  /* The front of a frame for one 8 x 8 screen tile of role r's slot (k_wf_frame's makeTile): 0 not now, 1 a tile made, 2 that slot has no more tiles. */
  auto makeTile = [&](uint32_t r) -> uint32_t {
    if (fq_load(&ctl[CC_TILEDRY + r]) != 0u) return 2u;
    FLX_CHAIN_ARGS();
    const uint32_t slot = slotOfRole(r);
    uint32_t take = 0, tile = 0;
    if (lane == 0 && fq_load(&rctl(RK_READY, r)[2]) < readyUnits) {
      atomicAdd(&ctl[CC_ALIVE + r], perTile);
      if (aliveAll() > FQ_ALIVE_MAX - r * (uint32_t)FLX_CHAIN_RESERVE) atomicSub(&ctl[CC_ALIVE + r], perTile);
      else { tile = atomicAdd(&ca.slots[slot].tileNext, 1u); take = 1; }
    }
    take = __builtin_amdgcn_readfirstlane(take);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (take == 0u) return 0u;
    if (tile >= tilesPerSlot) {
      if (lane == 0) { atomicSub(&ctl[CC_ALIVE + r], perTile); __hip_atomic_store(&ctl[CC_TILEDRY + r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (r == 1u) { statMin(CS_SDRY_MIN); statMax(CS_SDRY_MAX); }
      return 2u;
    }
    if (ca.order[slot] != nullptr) tile = ca.order[slot][tile];
    tile += slot * tilesPerSlot;                                        /* stacked: the tiles of slot s + 1 follow those of slot s */
    const float4 h = primary_tile<false, true>(ab, tile, lane, cnt, lv);
    const bool runs = shade0_tile<false, true>(ab, tile, lane, h, cnt, lv);
    if (flx_ballot(runs) == 0ull) {
      if (lane == 0) atomicSub(&ctl[CC_ALIVE + r], perTile);
    } else {
      for (uint32_t s0 = 0; s0 < samples; s0 += 64u)
        fq_push(ring(RK_READY, r), rctl(RK_READY, r), s0 + lane < samples, tile * samples + s0 + lane, lane);
    }
    statAdd(r == 0u ? CS_TILES_P : (r == 1u ? CS_TILES_S : CS_TILES_S2), 1ull);
    if (lane == 0) __hip_atomic_fetch_add(&ca.slots[slotP].tilesMade, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 1u;
  };

  /* Does this workgroup hold nothing of P any more (then it says so, once)?  And when every workgroup has said so: has the launch done its share of the work
   * ahead?  The share: as many screen tiles of the younger frames as one frame has — every launch then moves the chain on by one frame's worth of fresh paths
   * (a launch whose own frame the launches before it had completed would otherwise end at once, having prepared nothing, and a later one would find its frame
   * untouched and pay that frame's whole latency) — or nothing more to make: the younger frames' tile queues dry, or their views not posted
   * FLX_CHAIN_POST_WAIT_US after this workgroup was through with P. */
  auto checkDone = [&](bool pollGlobal) {
    if (fq_load(&ctl[CC_PDONE]) == 0u) {
      if (!(fq_load(&ctl[CC_TILEDRY]) != 0u && listDry(0u) && fq_load(&ctl[CC_ALIVE]) == 0u)) return;
      FLX_CHAIN_ARGS();
      if (lane == 0 && atomicExch(&ctl[CC_PDONE], 1u) == 0u) {
        __hip_atomic_fetch_add(&ca.slots[slotP].groupsDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ctl[CC_PDONE_T], (uint32_t)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ca.stats) {
          const unsigned long long now = (unsigned long long)wall_clock64();
          atomicMin(ca.stats + CS_PDONE_MIN, now); atomicMax(ca.stats + CS_PDONE_MAX, now);
          const unsigned long long us = (now - ca.stats[CS_START_MIN]) / 100ull;      /* (the first workgroup's start: close enough) */
          const int bin = us < 100ull ? 0 : (us < 200ull ? 1 : (us >= 2000ull ? 11 : (int)(us / 200ull) + 1));
          atomicAdd(ca.stats + CS_PDONE_HIST + bin, 1ull);
        }
      }
      return;
    }
    if (!pollGlobal || fq_load(&ctl[CC_STOP]) != 0u) return;
    FLX_CHAIN_ARGS();
    uint32_t stopNow = 0;
    if (lane == 0 && __hip_atomic_load(&ca.slots[slotP].groupsDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) {
      stopNow = 1;
      if (FLX_CHAIN_QUOTA && __hip_atomic_load(&ca.slots[slotP].tilesMade, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tilesPerSlot) {
        const bool waited = (uint32_t)wall_clock64() - fq_load(&ctl[CC_PDONE_T]) > (uint32_t)FLX_CHAIN_POST_WAIT_US * 100u;
        for (uint32_t r = 1; r < depth; r++) {
          const uint32_t a = fq_load(&ctl[CC_SAVAIL + r]);
          if (a == 2u) break;
          if (a == 0u) { if (!waited) stopNow = 0; break; }              /* its view may still come */
          if (__hip_atomic_load(&ca.slots[slotOfRole(r)].tileNext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tilesPerSlot) { stopNow = 0; break; }      /* tiles left to make */
        }
      }
    }
    if (__builtin_amdgcn_readfirstlane(stopNow) != 0u) { if (lane == 0) __hip_atomic_store(&ctl[CC_STOP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); statMin(CS_STOP_MIN); }
  };
  /* nothing left for this workgroup, now or later: P done here, and of every younger frame nothing held and nothing more to come (or it never comes) */
  auto groupFinished = [&]() -> bool {
    if (fq_load(&ctl[CC_PDONE]) == 0u) return false;
    for (uint32_t r = 1; r < depth; r++) {
      const uint32_t a = fq_load(&ctl[CC_SAVAIL + r]);
      if (a == 2u) break;                                               /* (nor do the frames after it) */
      if (a == 0u || fq_load(&ctl[CC_ALIVE + r]) != 0u || fq_load(&ctl[CC_TILEDRY + r]) == 0u || !listDry(r)) return false;
    }
    return true;
  };

  /* A wave leaves.  The last one hands what the rings still hold — paths and units of the younger frames — to the next kernel's resume lists. */
  auto leave = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t last = 0;
    if (lane == 0) last = atomicAdd(&ctl[CC_STOPPED], 1u) + 1u == WAVES ? 1u : 0u;
    if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    statMax(CS_END_MAX); statMin(CS_END_MIN);
    statAdd(fq_load(&ctl[CC_STOP]) != 0u ? CS_EXIT_STOP : CS_EXIT_FINISHED, 1ull);
    FLX_CHAIN_ARGS();
    const uint32_t wrSet = ca.seqP & 1u;
    for (uint32_t r = 0; r < depth; r++) {
      const uint32_t slot = slotOfRole(r);
      for (uint32_t kind = 0; kind < 3u; kind++) {
        uint32_t *c = rctl(kind, r);
        const uint32_t tail = fq_load(&c[0]), head = fq_load(&c[1]);
        const uint32_t n = tail - head;
        if (n == 0u) continue;
        if (r == 0u && lane == 0) __hip_atomic_fetch_or(ca.error, CH_ERR_LEFTOVER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      /* (P was complete: nothing of it can be here) */
        const uint32_t list = kind == RK_SHADE ? (uint32_t)CL_SHADE : (kind == RK_WALK ? (uint32_t)CL_WALK : (uint32_t)CL_READY);
        uint32_t pos = 0;
        if (lane == 0) pos = __hip_atomic_fetch_add(&ca.slots[slot].count[wrSet][list], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos = __builtin_amdgcn_readfirstlane(pos);
        statAdd(CS_DUMPED, n);
        uint32_t *dst = listBase(ca, slot, wrSet, list);
        uint32_t *src = ring(kind, r);
        for (uint32_t i = lane; i < n; i += 64u) {
          uint32_t *s = &src[(head + i) & (FQ_SIZE - 1u)];
          const uint32_t v = fq_load(s);
          __hip_atomic_store(s, WF_INVALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (pos + i < ca.listCap) dst[pos + i] = v; else if (i == lane) __hip_atomic_fetch_or(ca.error, CH_ERR_LIST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  };

  if (wave >= WALK_WAVES) {
    /* ================================ shade wave ================================ */
    uint32_t idle = 0;
    const long long tShade0 = wall_clock64();
    long long tTile = 0, tBatch = 0;
    for (;;) {
      if (fq_load(&ctl[CC_STOP]) != 0u) break;
      checkDone(true);
      bool did = false;
      for (uint32_t r = 0; r < depth; r++) {
        if (r != 0u && !sAvail(r, (idle & 15u) == 0u)) break;      /* (the mailbox is asked at most every 16th idle round; a frame that is not there yet: nor are the ones after it) */
        const bool dry = fq_load(&ctl[CC_TILEDRY + r]) != 0u && listDry(r);
        uint32_t id = WF_INVALID;
        uint32_t got = fq_pop(ring(RK_SHADE, r), rctl(RK_SHADE, r), ~0ull, 64u, dry ? 1u : 64u, lane, id);
        if (got == 0u && r + 1u < depth) {
          uint32_t first = 0;
          got = pullList(r, CL_SHADE, 64u, 1u, first);
          if (got != 0u) statAdd(CS_PULL_SHADE, got);
          if (got != 0u && lane < got) { FLX_CHAIN_ARGS(); id = listBase(ca, slotOfRole(r), rdSet, CL_SHADE)[first + lane]; }
        }
        if (got != 0u) {
          const long long tb0 = wall_clock64();
          const bool mine = lane < got && id != WF_INVALID;
          statAdd(r == 0u ? CS_BATCHES_P : (r == 1u ? CS_BATCHES_S : CS_BATCHES_S2), 1ull); statAdd(r == 0u ? CS_BATCH_LANES_P : (r == 1u ? CS_BATCH_LANES_S : CS_BATCH_LANES_S2), got);
          if (FLX_CHAIN_PRIO && r == 0u) __builtin_amdgcn_s_setprio(3);
          if (mine) shade_path<false, true>(ab, id, cnt, lv);
          if (FLX_CHAIN_PRIO && r == 0u) __builtin_amdgcn_s_setprio(0);
          {
            FLX_CHAIN_ARGS();
            const uint32_t slot = slotOfRole(r);
            if (mine && ca.cost[slot] != nullptr) {
              uint32_t t0, s0;
              item_tile(fr, id, t0, s0);
              atomicAdd(&ca.cost[slot][t0 - slot * tilesPerSlot], 1u);
            }
          }
          fq_push(ring(RK_WALK, r), rctl(RK_WALK, r), mine, id, lane);
          tBatch += wall_clock64() - tb0;
          did = true;
          break;
        }
        const long long tt0 = wall_clock64();
        if (makeTile(r) == 1u) { tTile += wall_clock64() - tt0; did = true; break; }
      }
      if (did) { idle = 0; continue; }
      if (groupFinished()) break;
      if (++idle > FQ_WATCHDOG) { giveUp(CH_ERR_SHADE_WATCHDOG); break; }
      __builtin_amdgcn_s_sleep(8);
    }
    statAdd(CS_SHADE_TILE_T, (unsigned long long)tTile); statAdd(CS_SHADE_BATCH_T, (unsigned long long)tBatch); statAdd(CS_SHADE_TOTAL_T, (unsigned long long)(wall_clock64() - tShade0));
    leave();
    return;
  }

  /* ================================ walk wave ================================ */
  if (wave < (uint32_t)FLX_FRAME_PROLOGUE_WAVES) while (makeTile(0u) == 1u) {}
  float2 *myRays = raysBase + (size_t)threadIdx.x * nTransforms * 5u;
  const float4 *walkG; { FLX_ARGS_OF(ab); walkG = pinnedWalkCopy(sc); }      /* the global copy of the tree, for the entries beyond the LDS top: in registers for the stepping loop */
  int st = P_EMPTY;
  uint32_t pathId = 0;
  int flags = 0;
  int pathBounce = 0;
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
  uint32_t chunkNext = 0, chunkEnd = 0;          /* wave-uniform: the rest of a (tile, sample) unit of fresh paths */
  uint32_t idleSpins = 0;
  bool stopped = false;
  uint32_t statTrips = 0, statLaneTrips = 0;

  auto pixPart = [&](const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t id) -> const float4 * {
    uint32_t tile0, s0;
    item_tile(fr, id, tile0, s0);
    return wb.pix0 + (((size_t)tile0 << 6) | (id & 63u)) * 3;
  };

  for (;;) {
    const bool stop = fq_load(&ctl[CC_STOP]) != 0u;
    const unsigned long long walking = flx_ballot(st == P_WALKING);
    if (walking != 0ull) idleSpins = 0;
    const unsigned long long workMask = flx_ballot(st == P_DONE || st == P_SWITCH);
    const uint32_t parked = 64u - (uint32_t)__popcll(walking);
    bool mayRefill = chunkNext != chunkEnd;
    for (uint32_t r = 0; r < depth && !mayRefill; r++)
      mayRefill = fq_load(&rctl(RK_WALK, r)[2]) != 0u || fq_load(&rctl(RK_READY, r)[2]) != 0u || (fq_load(&ctl[CC_LISTDRY + r]) & 13u) != 13u;
    statTrips += (uint32_t)FLX_WF_INNER; statLaneTrips += (uint32_t)__popcll(walking) * (uint32_t)FLX_WF_INNER;      /* (of the trips before this pass) */
    if (stop || walking == 0ull || (parked >= (uint32_t)FLX_WF_BATCH && (workMask != 0ull || mayRefill))) {
      /* ---- fold the finished lanes: fragment:445-460, 580, 593-598 and the guard of :475; a path that goes on is handed to the shade waves ---- */
      if (flx_ballot(st == P_DONE) != 0ull) {
        FLX_ARGS_OF(ab);
        const bool compactRecs = wb.rec0 != nullptr;
        bool toShade = false, ended = false;
        const uint32_t role = roleOf(pathId);
        if (st == P_DONE) {
          float4 *rec = wb.rec + (size_t)pathId * 8;
          const bool compact = compactRecs && pathBounce == 0;
          float4 q4, q5, q6, q7;
          const float4 *pp = nullptr;
          if (compact) {
            pp = pixPart(fr, wb, pathId);
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
            if (compact) {                                    /* the path goes on: now it gets its full record (what shade0 would have written) */
              const float4 a = wb.rec0[(size_t)pathId * 3], bq = wb.rec0[(size_t)pathId * 3 + 1], p0 = pp[0];
              rec[0] = make_float4(p0.x, p0.y, p0.z, a.w);
              rec[1] = make_float4(a.x, a.y, a.z, bq.w);
              rec[3] = make_float4(bq.x, bq.y, bq.z, __int_as_float(0));
              rec[6] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
              rec[7] = make_float4(q7.x, q7.y, q7.z, 0.0f);
            }
            rec[5] = make_float4(finalColor.x, finalColor.y, finalColor.z, 0.0f);
            rec[2] = make_float4(w.suv.x, w.suv.y, w.suv.z, __int_as_float(w.tri));
            toShade = true;
          } else {
            finalize_path<true>(fr, wb, pathId, finalColor, importancy, originalColor, lv);
            ended = true;
          }
          st = P_EMPTY;
        }
        {
          FLX_CHAIN_ARGS();
          if (ca.stats && (toShade || ended) && role == 0u) {
            const int b = pathBounce < 3 ? pathBounce : 3;
            atomicAdd(ca.stats + CS_P_FOLD_BOUNCE + b, 1ull);
            if ((unsigned long long)wall_clock64() - ca.stats[CS_START_MIN] > 50000ull) atomicAdd(ca.stats + CS_P_FOLD_LATE + b, 1ull);
          }
        }
        for (uint32_t r = 0; r < depth; r++) {
          fq_push(ring(RK_SHADE, r), rctl(RK_SHADE, r), toShade && role == r, pathId, lane);
          const uint32_t e = (uint32_t)__popcll(flx_ballot(ended && role == r));
          if (e != 0u && lane == 0) atomicSub(&ctl[CC_ALIVE + r], e);
        }
      }
      if (stop) { stopped = true; break; }
      /* ---- refill the free lanes: everything of an older frame before anything of a younger one; per frame the paths that came back from shading (the
       * workgroup's ring, then the resume lists), then fresh ones ---- */
      for (;;) {
        const unsigned long long idle = flx_ballot(st == P_EMPTY);
        if (idle == 0ull) break;
        FLX_CHAIN_ARGS();
        const uint32_t nIdle = (uint32_t)__popcll(idle);
        const uint32_t rk = lane_rank(idle);
        uint32_t id = WF_INVALID;
        bool fresh = false;                                    /* a bounce-0 item: compact record, may be dead */
        bool got = false;
        int resumed = 0;                                       /* the state a suspended walk was in (P_WALKING, P_SWITCH), or 0 */
        uint32_t role = 0;
        for (uint32_t r = 0; r < depth; r++) {
          if (r != 0u && !sAvail(r, false)) break;
          role = r;
          if (fq_pop(ring(RK_WALK, r), rctl(RK_WALK, r), idle, nIdle, 1u, lane, id) != 0u) { got = true; break; }
          if (r + 1u < depth) {
            uint32_t first = 0;
            const uint32_t n = pullList(r, CL_WALK, nIdle, 1u, first);
            if (n != 0u) {
              if (ca.stats && lane == 0) atomicAdd(ca.stats + CS_PULL_WALK, (unsigned long long)n);
              if (st == P_EMPTY && rk < n) { const uint32_t v = listBase(ca, slotOfRole(r), rdSet, CL_WALK)[first + rk]; fresh = (v & CH_FRESH) != 0u; id = v & ~CH_FRESH; }
              got = true;
              break;
            }
            const uint32_t ns = pullList(r, CL_SUSP, nIdle, 1u, first);
            if (ns != 0u) {                                  /* walks the kernel before suspended in flight: the lane's walk state comes back as it was */
              if (ca.stats && lane == 0) atomicAdd(ca.stats + CS_PULL_SUSP, (unsigned long long)ns);
              if (st == P_EMPTY && rk < ns) {
                const float4 *sr = suspBase(ca, slotOfRole(r), rdSet) + (size_t)(first + rk) * CH_SUSP_F4;
                const float4 s0 = sr[0], s1 = sr[1], s2 = sr[2], s3 = sr[3], s4 = sr[4];
                id = (uint32_t)__float_as_int(s0.x);
                const int packed = __float_as_int(s0.y);
                resumed = packed & 15; fresh = ((packed >> 10) & 1) != 0;
                w.mode = (packed >> 4) & 15; w.fastDiv = ((packed >> 8) & 1) != 0; w.shadowed = ((packed >> 9) & 1) != 0;
                w.minLen = s0.z; w.i = __float_as_int(s0.w);
                w.tR.origin = F3(s1.x, s1.y, s1.z); w.tR.dir = F3(s1.w, s2.x, s2.y); w.inv = F3(s2.z, s2.w, s3.x);
                w.suv = F3(s3.y, s3.z, s3.w);
                w.cachedTI = __float_as_int(s4.x); w.tri = __float_as_int(s4.y); w.hitTI = __float_as_int(s4.z);
              }
              got = true;
              break;
            }
          }
          if (chunkNext == chunkEnd && fq_load(&rctl(RK_SHADE, r)[2]) < FQ_LIMIT) {      /* (the shade waves are not behind with this frame) */
            uint32_t unit = WF_INVALID;
            if (fq_pop(ring(RK_READY, r), rctl(RK_READY, r), 1ull, 1u, 1u, lane, unit) != 0u) {
              unit = __builtin_amdgcn_readfirstlane(unit);
              if (unit != WF_INVALID) { chunkNext = unit << 6; chunkEnd = chunkNext + 64u; }
              else if (lane == 0) { atomicSub(&ctl[CC_ALIVE + r], 64u); __hip_atomic_fetch_or(ca.error, WF_ERR_RING_SLOT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }      /* (a slot that never filled: fq_pop's own watchdog) */
            } else if (r + 1u < depth) {
              uint32_t first = 0;
              if (pullList(r, CL_READY, 1u, 64u, first) != 0u) {
                unit = listBase(ca, slotOfRole(r), rdSet, CL_READY)[first]; chunkNext = unit << 6; chunkEnd = chunkNext + 64u;
                if (ca.stats && lane == 0) atomicAdd(ca.stats + CS_PULL_READY, 1ull);
              }
            }
          }
          if (chunkNext != chunkEnd && roleOf(chunkNext) == r) {
            const uint32_t avail = chunkEnd - chunkNext;
            const uint32_t take = nIdle < avail ? nIdle : avail;
            if (st == P_EMPTY && rk < take) { id = chunkNext + rk; fresh = true; }
            if (ca.stats && r == 0u && lane == 0) atomicMax(ca.stats + CS_P_LAST_FRESH, (unsigned long long)wall_clock64());
            chunkNext += take;
            got = true;
            break;
          }
        }
        if (!got) break;
        bool dead = false;
        if (id != WF_INVALID) {
          const float4 *rec = wb.rec + (size_t)id * 8;
          float4 q0, q1, q2, q3;
          if (fresh) {
            const float4 *pp = pixPart(fr, wb, id);
            const float4 a = wb.rec0[(size_t)id * 3], bq = wb.rec0[(size_t)id * 3 + 1];
            const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2];
            q0 = make_float4(p0.x, p0.y, p0.z, a.w);
            q1 = make_float4(a.x, a.y, a.z, bq.w);
            q2 = make_float4(p1.x, p1.y, p1.z, p2.w);
            q3 = make_float4(bq.x, bq.y, bq.z, __int_as_float(0));
          } else {
            q0 = rec[0]; q1 = rec[1]; q2 = rec[2]; q3 = rec[3];
          }
          const int fl = __float_as_int(q0.w);
          if (fl & RF_DEAD) {
            dead = true;
          } else {
            pathId = id; flags = fl; base = q2.w; pathBounce = __float_as_int(q3.w);
            nextRay.origin = F3(q0.x, q0.y, q0.z);
            nextRay.dir = F3(q1.x, q1.y, q1.z);
            shadowRay.origin = F3(q2.x, q2.y, q2.z);
            shadowRay.dir = F3(q3.x, q3.y, q3.z);
            shadowLen = q1.w;
            if (resumed != 0) {
              st = resumed == P_SWITCH ? P_SWITCH : P_RESUMED;      /* (a shadow walk that had ended goes straight to the closest-hit set-up) */
            } else {
              walkClearResults(w);
              w.mode = (fl & RF_NEED_SHADOW) ? 0 : 1;
              st = (w.mode == 1 && (fl & RF_NO_CLOSEST)) ? P_DONE : P_SETUP;      /* nothing to walk: straight to the fold */
            }
          }
        }
        const uint32_t nDead = (uint32_t)__popcll(flx_ballot(dead));
        if (nDead != 0u && lane == 0) atomicSub(&ctl[CC_ALIVE + role], nDead);
      }
      if (FLX_CHAIN_PRIO) {
        const bool tail = fq_load(&ctl[CC_TILEDRY]) != 0u && listDry(0u);
        if (tail && flx_ballot(st != P_EMPTY && roleOf(pathId) == 0u) != 0ull) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
      }
      /* ---- set up walks: fresh lanes (shadow or closest) and lanes whose shadow walk just ended ---- */
      if (st == P_SWITCH) {
        if (flags & RF_NO_CLOSEST) st = P_DONE;
        else { w.mode = 1; st = P_SETUP; }
      }
      if (flx_ballot(st == P_SETUP || st == P_RESUMED) != 0ull) {
        FLX_ARGS_OF(ab);
        if (st == P_SETUP || st == P_RESUMED) {
          const bool shadowMode = w.mode == 0;
          const Ray src = shadowMode ? shadowRay : nextRay;
          walkSetupRays(sc, nTransforms, ldsXf, myRays, src, shadowMode);
          if (st == P_SETUP) {
            w.tR = src; w.cachedTI = 0; w.minLen = shadowMode ? shadowLen : POW32; w.i = (int)sc.walk_root;
            reciprocalOfDir(sc, src.dir, src.origin, w.inv, w.fastDiv);
            st = P_WALKING;
            if (walkFetchG<false>(walkG, ldsEntries, ldsCount, myRays, w, cur, cnt)) st = shadowMode ? P_SWITCH : P_DONE;
          } else {
            /* a suspended walk: its registers came back from the list, the rays in LDS are recomputed (the same arithmetic on the same record), the entry it
             * was about to test is fetched again */
            st = P_WALKING;
            walkLoadEntry(sc, ldsEntries, ldsCount, (uint32_t)w.i, cur);
          }
        }
      }
      if (flx_ballot(st == P_WALKING) == 0ull) {
        if (flx_ballot(st != P_EMPTY) != 0ull) continue;      /* lanes that had nothing to walk wait for the fold */
        /* nothing in this wave: gone when nothing can come any more, else wait for the shade waves (or for a later frame's view, or for the stop) */
        checkDone(false);
        if (groupFinished()) break;
        if (!mayRefill) {
          if (++idleSpins > FQ_WATCHDOG) { giveUp(CH_ERR_WALK_WATCHDOG); break; }
          __builtin_amdgcn_s_sleep(8);
        }
        continue;
      }
    }
    /* ---- FLX_WF_INNER entries for every walking lane (the few scene words the fetch needs are read before the loop) ---- */
    {
#pragma unroll FLX_WF_UNROLL
      for (int it = 0; it < FLX_WF_INNER; it++) {
        if (st == P_WALKING) {
          bool ended = false;
          if (walkIsBoxT(cur)) walkBoxP(w, cur); else ended = walkTriT(w, cur);
          if (!ended) ended = walkFetchG<false>(walkG, ldsEntries, ldsCount, myRays, w, cur, cnt);
          if (ended) st = (w.mode == 0) ? P_SWITCH : P_DONE;
        }
      }
    }
  }
  statAdd(CS_WALK_TRIPS, statTrips); statAdd(CS_WALK_LANE_TRIPS, statLaneTrips);
  if (stopped) {
    /* ---- the launch ends: what this wave still walks (all of it of younger frames: P is complete everywhere) goes to the next kernel: walks in flight
     * with their state, walks not begun and the rest of a fresh unit as paths to walk from their records ---- */
    FLX_CHAIN_ARGS();
    const uint32_t wrSet = ca.seqP & 1u;
    const uint32_t role = roleOf(pathId);
    const bool compact0 = wb.rec0 != nullptr && pathBounce == 0;
    for (uint32_t r = 1; r < depth; r++) {
      const uint32_t slot = slotOfRole(r);
      const bool susp = (st == P_WALKING || st == P_SWITCH) && role == r;               /* a walk in flight (or a shadow walk just over): its state goes along */
      const unsigned long long sm = flx_ballot(susp);
      if (sm != 0ull) {
        const uint32_t nsu = (uint32_t)__popcll(sm);
        if (ca.stats && lane == 0) atomicAdd(ca.stats + CS_SUSPENDED, (unsigned long long)nsu);
        uint32_t pos = 0;
        if (lane == 0) pos = __hip_atomic_fetch_add(&ca.slots[slot].count[wrSet][CL_SUSP], nsu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (pos + nsu > ca.suspCap) { if (lane == 0) __hip_atomic_fetch_or(ca.error, CH_ERR_LIST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else if (susp) {
          float4 *q = suspBase(ca, slot, wrSet) + (size_t)(pos + lane_rank(sm)) * CH_SUSP_F4;
          const int packed = st | (w.mode << 4) | ((w.fastDiv ? 1 : 0) << 8) | ((w.shadowed ? 1 : 0) << 9) | (compact0 ? 1 << 10 : 0);
          q[0] = make_float4(__int_as_float((int)pathId), __int_as_float(packed), w.minLen, __int_as_float(w.i));
          q[1] = make_float4(w.tR.origin.x, w.tR.origin.y, w.tR.origin.z, w.tR.dir.x);
          q[2] = make_float4(w.tR.dir.y, w.tR.dir.z, w.inv.x, w.inv.y);
          q[3] = make_float4(w.inv.z, w.suv.x, w.suv.y, w.suv.z);
          q[4] = make_float4(__int_as_float(w.cachedTI), __int_as_float(w.tri), __int_as_float(w.hitTI), 0.0f);
        }
      }
      const bool keep = (st == P_SETUP || st == P_RESUMED) && role == r;                 /* not begun: the next kernel walks it from its record */
      const unsigned long long km = flx_ballot(keep);
      const uint32_t nk = (uint32_t)__popcll(km), nChunk = (chunkNext != chunkEnd && roleOf(chunkNext) == r) ? chunkEnd - chunkNext : 0u;
      if (nk + nChunk != 0u) {
        if (ca.stats && lane == 0) { atomicAdd(ca.stats + CS_ABANDONED, (unsigned long long)nk); atomicAdd(ca.stats + CS_LEFT_CHUNK, (unsigned long long)nChunk); }
        uint32_t pos = 0;
        if (lane == 0) pos = __hip_atomic_fetch_add(&ca.slots[slot].count[wrSet][CL_WALK], nk + nChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (pos + nk + nChunk > ca.listCap) { if (lane == 0) __hip_atomic_fetch_or(ca.error, CH_ERR_LIST, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else {
          uint32_t *dst = listBase(ca, slot, wrSet, CL_WALK);
          if (keep) dst[pos + lane_rank(km)] = pathId | (compact0 ? CH_FRESH : 0u);
          for (uint32_t i = lane; i < nChunk; i += 64u) dst[pos + nk + i] = (chunkNext + i) | CH_FRESH;
        }
      }
    }
  }
  leave();
}

/* Does the chained kernel take this scene?  (LDS: as the frame kernel with its front inside, plus the views.) */
bool chain_kernel_fits(const DeviceScene &sc, uint32_t &ldsCount, uint32_t &ldsBytes) {
  const uint32_t T = sc.n_transforms;
  const uint32_t walkThreads = FLX_WF_WALK_THREADS - 64u * (uint32_t)FLX_FRAME_SHADERS_FRONT;
  const uint32_t fixed = walkThreads * T * 40u + T * 64u + (CC_WORDS + CC_VIEW_WORDS) * 4u;
  if (fixed + 4096u > (uint32_t)FLX_WF_LDS_TOTAL) return false;
  ldsCount = ((uint32_t)FLX_WF_LDS_TOTAL - fixed) / 48u;
  if (ldsCount > sc.walk_hot) ldsCount = sc.walk_hot;
  ldsBytes = ldsCount * 48u + fixed;
  return true;
}

/* (the mailbox needs no reset: sequence numbers are never reused) */
__global__ void k_chain_reset(ChainSlot *slots, uint32_t depth, uint32_t whole, uint32_t set) {
  const uint32_t slot = threadIdx.x >> 5, word = threadIdx.x & 31u;
  if (slot >= depth) return;
  uint32_t *p = (uint32_t *)&slots[slot];
  const uint32_t cnt0 = (uint32_t)(offsetof(ChainSlot, count) / 4u) + set * CH_LISTS, tak0 = (uint32_t)(offsetof(ChainSlot, taken) / 4u) + set * CH_LISTS;
  if (((whole >> slot) & 1u) || (word >= cnt0 && word < cnt0 + CH_LISTS) || (word >= tak0 && word < tak0 + CH_LISTS)) p[word] = 0u;
}
void launch_chain_reset(ChainSlot *slots, uint32_t depth, uint32_t whole, uint32_t set, hipStream_t stream) {
  hipLaunchKernelGGL(k_chain_reset, dim3(1), dim3(32 * CH_MAX_DEPTH), 0, stream, slots, depth, whole, set);
}

size_t chain_rings_per_group() { return (size_t)CH_RINGS * FQ_SIZE; }

int launch_chain(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, const ChainArgs &ca, uint32_t compute_units, hipStream_t stream) {
  uint32_t ldsCount = 0, ldsBytes = 0;
  if (!chain_kernel_fits(sc, ldsCount, ldsBytes)) return -1;
  if (ca.depth < 2u || ca.depth > CH_MAX_DEPTH) return -1;
  static std::once_flag once[64];
  static bool ok[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::call_once(once[dev], [&]() { ok[dev] = hipFuncSetAttribute((const void *)k_wf_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; });
  if (!ok[dev]) return -1;
  ChainKernelArgs ka;
  ka.fa.sc = sc; ka.fa.fr = fr; ka.fa.wb = wb; ka.ca = ca;
  const uint32_t tilesPerGroup = ca.tilesPerSlot / compute_units;
  uint32_t readyUnits = tilesPerGroup >= 48u ? (uint32_t)FLX_FRAME_READY_UNITS : tilesPerGroup / 2u;
  readyUnits = readyUnits < (uint32_t)FLX_FRAME_READY_UNITS / 4u ? (uint32_t)FLX_FRAME_READY_UNITS / 4u : (readyUnits > (uint32_t)FLX_FRAME_READY_UNITS ? (uint32_t)FLX_FRAME_READY_UNITS : readyUnits);
  hipLaunchKernelGGL(k_wf_chain, dim3(compute_units), dim3(FLX_WF_WALK_THREADS), ldsBytes, stream, ka, ldsCount, sc.n_transforms, (uint32_t)FLX_FRAME_SHADERS_FRONT, readyUnits);
  return 0;
}

}  // namespace flx
