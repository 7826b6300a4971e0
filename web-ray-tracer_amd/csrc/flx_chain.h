/* flx_chain.h — the chained frame kernel's device-side state and launcher (flx_chain.hip), shared with the frame loop (flx_api.hip). */
#ifndef FLX_CHAIN_H
#define FLX_CHAIN_H

#include <mutex>
#include "flx_kernels.h"

namespace flx {

constexpr uint32_t CH_FRESH = 0x80000000u;     /* walk resume list: the path is a bounce-0 item with a compact record */
#ifndef FLX_CHAIN_RESERVE
#define FLX_CHAIN_RESERVE 4096            /* places of a workgroup's rings that only paths of its own frame (P) may take */
#endif
#define FLX_CHAIN_RESERVE_HOST FLX_CHAIN_RESERVE
constexpr uint32_t CH_SUSP_F4 = 5;             /* float4 of state a walk suspended in flight takes to the next kernel */
constexpr uint32_t CH_RINGS = 6;               /* rings of a chained workgroup: (to shade, to walk, fresh units) x (its own frame, the next one) */
/* bits of the context's device error word (flx_kernels.h: WF_ERR_*) */
constexpr uint32_t CH_ERR_SHADE_WATCHDOG = WF_ERR_SHADE_WATCHDOG, CH_ERR_WALK_WATCHDOG = WF_ERR_WALK_WATCHDOG, CH_ERR_LIST = WF_ERR_LIST, CH_ERR_LEFTOVER = WF_ERR_LEFTOVER;

/* One of the two frame slots of a chain, in device memory; zeroed in stream order when the slot is recycled for a new frame. */
struct ChainSlot {
  uint32_t tileNext;                           /* cursor of the slot's queue of screen tiles */
  uint32_t groupsDone;                         /* workgroups of the kernel that must complete this slot which hold nothing of it any more */
  uint32_t error;
  uint32_t pad0;
  uint32_t walkCount, walkTaken;               /* resume lists: entries written by the kernel that worked ahead on this slot, taken by the kernel that completes it */
  uint32_t shadeCount, shadeTaken;
  uint32_t readyCount, readyTaken;
  uint32_t suspCount, suspTaken;               /* walks suspended in flight: CH_SUSP_F4 float4 of walk state each */
  uint32_t pad1[4];
};
/* What flx_frame_begin of the NEXT frame posts while the kernel runs (pinned host memory, plain stores): the view, then the frame's sequence number. */
struct ChainMail {
  uint32_t posted[2];
  uint32_t pad[2];
  FrameView view[2];
};
struct ChainArgs {
  ChainSlot *slots;                            /* [2] */
  const ChainMail *mail;                       /* pinned host memory */
  ChainMail *relay;                            /* device memory: the post as the kernel's relaying waves pass it on */
  uint32_t *walkList[2], *shadeList[2], *readyList[2];
  float4 *suspList[2];                         /* CH_SUSP_F4 float4 per suspended walk */
  uint32_t suspCap;                            /* walks per list */
  uint32_t listCap;                            /* entries per list */
  uint32_t slotP;                              /* the slot this kernel completes */
  uint32_t seqS;                               /* the sequence number a post for the other slot must carry (0: the kernel never works ahead) */
  uint32_t tilesPerSlot, itemsPerSlot;
  const uint32_t *order[2];                    /* or nullptr: the order in which a slot's screen tiles are drawn (a permutation of 0 .. tilesPerSlot - 1) */
  uint32_t *cost[2];                           /* or nullptr: per screen tile of a slot, the shadings its paths took after bounce 0 (what orders a later frame) */
  uint32_t *error;                             /* the context's device error word (pinned host memory): CH_ERR_* bits */
  unsigned long long *stats;                   /* or nullptr: CH_STAT_WORDS words of this launch's diagnostics (flx_get_chain_stats) */
};
/* diagnostics of one launch (times: wall_clock64 ticks, 100 MHz) */
enum { CS_START_MIN = 0, CS_END_MAX, CS_SAVAIL_MIN, CS_STOP_MIN, CS_PDONE_MIN, CS_PDONE_MAX, CS_TILES_P, CS_TILES_S, CS_ABANDONED, CS_LEFT_CHUNK, CS_DUMPED, CS_PULL_WALK, CS_PULL_SHADE,
       CS_PULL_READY, CS_BATCHES_P, CS_BATCHES_S, CS_BATCH_LANES_P, CS_BATCH_LANES_S, CS_EXIT_FINISHED, CS_EXIT_STOP, CS_END_MIN, CS_SEQ, CS_SAVAIL_MAX, CS_SUSPENDED, CS_PULL_SUSP, CS_SDRY_MIN, CS_SDRY_MAX, CS_SHADE_TILE_T, CS_SHADE_BATCH_T, CS_SHADE_TOTAL_T, CS_WALK_LANE_TRIPS, CS_WALK_TRIPS, CS_PDONE_HIST /* [12]: workgroups through with P by 100, 200, 400, 600, .. 2000 us, later */, CH_STAT_WORDS = 48 };
constexpr int CH_STAT_LAUNCHES = 64;
struct ChainKernelArgs { FrameArgs fa; ChainArgs ca; };

bool chain_kernel_fits(const DeviceScene &sc, uint32_t &ldsCount, uint32_t &ldsBytes);
void launch_chain_reset(ChainSlot *slots, uint32_t slot_mask, hipStream_t stream);
size_t chain_rings_per_group();                /* uint32_t words of WavefrontBuffers::frameRings a chained workgroup uses */
/* fr: the two slots stacked (frames = 2, view[ca.slotP] filled in); wb over the stacked workspace, front = 1, item_base = 0.  0, or -1 if the kernel does not fit */
int launch_chain(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, const ChainArgs &ca, uint32_t compute_units, hipStream_t stream);

}  // namespace flx
#endif
