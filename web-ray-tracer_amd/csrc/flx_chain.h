/* flx_chain.h — the chained frame kernel's device-side state and launcher (flx_chain.hip), shared with the frame loop (flx_api.hip). */
#ifndef FLX_CHAIN_H
#define FLX_CHAIN_H

#include <mutex>
#include "flx_kernels.h"

namespace flx {

constexpr uint32_t CH_MAX_DEPTH = 3;           /* most frame slots of a chain = most frames in flight */
constexpr uint32_t CH_FRESH = 0x80000000u;     /* walk resume list: the path is a bounce-0 item with a compact record */
constexpr uint32_t CH_SUSP_F4 = 5;             /* float4 of state a walk suspended in flight takes to the next kernel */
#ifndef FLX_CHAIN_RESERVE
#define FLX_CHAIN_RESERVE 2048            /* places of a workgroup's rings that only paths of an older frame may take: role r draws up to FQ_ALIVE_MAX - r x this */
#endif
constexpr uint32_t CH_RINGS = 3 * CH_MAX_DEPTH;   /* rings of a chained workgroup: (to shade, to walk, fresh units) x (its own frame, the frames after it) */
constexpr uint32_t CH_LISTS = 4;               /* resume lists of a slot: paths to walk, paths to shade, fresh units, walks suspended in flight */
/* bits of the context's device error word (flx_kernels.h: WF_ERR_*) */
constexpr uint32_t CH_ERR_SHADE_WATCHDOG = WF_ERR_SHADE_WATCHDOG, CH_ERR_WALK_WATCHDOG = WF_ERR_WALK_WATCHDOG, CH_ERR_LIST = WF_ERR_LIST, CH_ERR_LEFTOVER = WF_ERR_LEFTOVER;

/* One frame slot of a chain, in device memory.  The resume lists exist twice: the kernel of sequence number m reads the set (m - 1) & 1 — what the kernel
 * before it left — and writes the set m & 1. */
struct ChainSlot {
  uint32_t tileNext;                           /* cursor of the slot's queue of screen tiles */
  uint32_t groupsDone;                         /* workgroups of the kernel that must complete this slot which hold nothing of it any more */
  uint32_t tilesMade;                          /* screen tiles (of any frame) the kernel that completes this slot has made: its share of the work ahead */
  uint32_t pad0;
  uint32_t count[2][CH_LISTS];                 /* entries written */
  uint32_t taken[2][CH_LISTS];                 /* entries handed out */
  uint32_t pad1[12];
};
static_assert(sizeof(ChainSlot) == 128, "k_chain_reset clears a slot as 32 words");
/* What flx_frame_begin of a LATER frame posts while the kernel runs (pinned host memory, plain stores): the view, then the frame's sequence number. */
struct ChainMail {
  uint32_t posted[4];
  FrameView view[CH_MAX_DEPTH];
};
struct ChainArgs {
  ChainSlot *slots;                            /* [depth] */
  const ChainMail *mail;                       /* pinned host memory */
  ChainMail *relay;                            /* device memory: the post as the kernel's relaying waves pass it on */
  uint32_t *lists;                             /* [slot][set][walk, shade, ready] x listCap entries */
  float4 *susp;                                /* [slot][set] x suspCap x CH_SUSP_F4 */
  uint32_t listCap, suspCap;
  uint32_t depth;                              /* slots of the chain (2 or 3) */
  uint32_t ahead;                              /* roles beyond its own frame this kernel may work on (0: none: the frame begins no chain) */
  uint32_t slotP;                              /* the slot this kernel completes; role r works on slot (slotP + r) % depth */
  uint32_t seqP;                               /* its frame's sequence number; a post for role r must carry seqP + r */
  uint32_t tilesPerSlot, itemsPerSlot;
  const uint32_t *order[CH_MAX_DEPTH];         /* or nullptr: the order in which a slot's screen tiles are drawn (a permutation of 0 .. tilesPerSlot - 1) */
  uint32_t *cost[CH_MAX_DEPTH];                /* or nullptr: per screen tile of a slot, the shadings its paths took after bounce 0 */
  uint32_t *error;                             /* the context's device error word (pinned host memory): CH_ERR_* bits */
  unsigned long long *stats;                   /* or nullptr: CH_STAT_WORDS words of this launch's diagnostics (flx_get_chain_stats) */
};
/* diagnostics of one launch (times: wall_clock64 ticks, 100 MHz) */
enum { CS_START_MIN = 0, CS_END_MAX, CS_SAVAIL_MIN, CS_STOP_MIN, CS_PDONE_MIN, CS_PDONE_MAX, CS_TILES_P, CS_TILES_S, CS_ABANDONED, CS_LEFT_CHUNK, CS_DUMPED, CS_PULL_WALK, CS_PULL_SHADE,
       CS_PULL_READY, CS_BATCHES_P, CS_BATCHES_S, CS_BATCH_LANES_P, CS_BATCH_LANES_S, CS_EXIT_FINISHED, CS_EXIT_STOP, CS_END_MIN, CS_SEQ, CS_SAVAIL_MAX, CS_SUSPENDED, CS_PULL_SUSP,
       CS_SDRY_MIN, CS_SDRY_MAX, CS_SHADE_TILE_T, CS_SHADE_BATCH_T, CS_SHADE_TOTAL_T, CS_WALK_LANE_TRIPS, CS_WALK_TRIPS,
       CS_PDONE_HIST /* [12]: workgroups through with P by 100, 200, 400, 600, .. 2000 us, later */, CS_TILES_S2 = 44, CS_BATCHES_S2, CS_BATCH_LANES_S2, CS_SAVAIL2_MIN,
       CS_P_PULL /* [4] entries of P's own resume lists taken: walk, shade, ready, susp */ = 48, CS_P_FOLD_BOUNCE /* [4] folds of P's paths by the bounce they were at */ = 52,
       CS_P_LAST_FRESH /* time the last fresh path of P began its first walk */ = 56, CS_P_FOLD_LATE /* [4] the same folds after 500 us */ = 57, CH_STAT_WORDS = 64 };
constexpr int CH_STAT_LAUNCHES = 64;
struct ChainKernelArgs { FrameArgs fa; ChainArgs ca; };

bool chain_kernel_fits(const DeviceScene &sc, uint32_t &ldsCount, uint32_t &ldsBytes);
/* a chain's slots back to "nothing yet": everything of the slots in `whole` (bit per slot), and of every slot the list set `set` the next kernel writes */
void launch_chain_reset(ChainSlot *slots, uint32_t depth, uint32_t whole, uint32_t set, hipStream_t stream);
size_t chain_rings_per_group();                /* uint32_t words of WavefrontBuffers::frameRings a chained workgroup uses */
/* fr: the slots stacked (frames = depth, view[ca.slotP] filled in); wb over the stacked workspace, front = 1, item_base = 0.  0, or -1 if the kernel does not fit */
int launch_chain(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, const ChainArgs &ca, uint32_t compute_units, hipStream_t stream);

}  // namespace flx
#endif
