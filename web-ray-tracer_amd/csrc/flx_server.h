/* flx_server.h — the frame server: ONE persistent launch that renders the frames of the loop as they are posted (flx_server.hip), and the words it
 * shares with the frame loop (flx_api.hip). */
#ifndef FLX_SERVER_H
#define FLX_SERVER_H

#include "flx_kernels.h"

namespace flx {

constexpr uint32_t SV_MAX_DEPTH = 3;           /* most frame slots = most frames in flight */
#ifndef FLX_SERVER_RESERVE
#define FLX_SERVER_RESERVE 2048           /* places of a workgroup's rings that only paths of an older frame may take: the r-th oldest frame draws up to FQ_ALIVE_MAX - r x this */
#endif
constexpr uint32_t SV_RINGS = 3 * SV_MAX_DEPTH;   /* rings of a workgroup: (to shade, to walk, fresh units) x slot */
constexpr uint32_t SV_BLOB_WORDS = 1024;       /* most words of a frame's lights and transforms that travel with it (ServerMail::blob): 32 per transform + 6 per light */
inline uint32_t server_blob_words(uint32_t n_transforms, uint32_t n_lights) { return n_transforms * 32u + n_lights * 6u; }

/* Device memory: what the workgroups of the launch share about a slot. */
struct ServerSlot {
  uint32_t tileNext;                           /* cursor of the queue of screen tiles of the frame in the slot */
  uint32_t groupsDone;                         /* workgroups that are through with that frame */
  uint32_t pad[14];
};
/* Pinned host memory.  Host -> device: the frames' views and sequence numbers (plain stores, the view before the number), and the number after which the
 * launch ends.  Device -> host: the frame in a slot is complete. */
struct ServerMail {
  uint32_t posted[4];                          /* [slot] sequence number of the frame whose view is in view[slot] */
  uint32_t stopAfter;                          /* the launch ends when every workgroup is through with this frame (0: go on) */
  uint32_t pad[3];
  uint32_t done[4];                            /* [slot] device -> host: sequence number of the last frame completed in the slot */
  FrameView view[SV_MAX_DEPTH];
  uint32_t blob[SV_MAX_DEPTH][SV_BLOB_WORDS];  /* [slot] a scene that moves: the frame's transforms and lights (written with the view, before the number): rotation (2 x 3 float4 per
                                                * transform), shift (2 float4 per transform), lights (6 floats each) */
};
struct ServerArgs {
  ServerSlot *slots;                           /* [depth], zeroed before the launch */
  ServerMail *mail;                            /* pinned host memory */
  ServerMail *relay;                           /* device memory: posts and the stop word as the relaying waves pass them on */
  uint32_t depth;                              /* slots (2 or 3) */
  uint32_t slot0, seq0;                        /* the first frame: the slot it is in, its sequence number; frame seq0 + i is in slot (slot0 + i) % depth */
  uint32_t tilesPerSlot, itemsPerSlot;
  float4 *out[SV_MAX_DEPTH];                   /* per slot: the resolved frame (float4[frame_rows][width]); a workgroup resolves the screen tiles it made when it is through with the frame */
  uint32_t outStripRows, outStripStep;         /* 0, or: the frame is a rank's row strips and out[] the WHOLE image from the rank's first strip on — row r of the frame is row
                                                * (r / outStripRows) x outStripStep + r % outStripRows there (strips of outStripRows rows, the rank's next one outStripStep rows on) */
  uint32_t outSystem;                          /* out[] is not this GPU's memory (a peer's, or the host's): a workgroup's part is released at system scope */
  uint32_t out8;                               /* out[] holds the canvas' RGBA8 (uint32 per pixel: flx_present's bytes, pack_rgba8) instead of float4: the tiles are quantised as they are resolved */
  uint32_t *tileLists;                         /* [workgroup][slot] x tileListCap: the screen tiles the workgroup made of the frame in the slot */
  uint32_t tileListCap;
  uint32_t idleExit;                           /* 100 MHz ticks without anything to do after which a workgroup gives up (an error: the host always says when to stop) */
  uint32_t blobWords;                          /* 0, or: the scene moves — words of ServerMail::blob that hold a frame's transforms and lights; DeviceScene::rotation / shift / lights
                                                * of the launch then are its version buffers ([workgroup x depth + slot] versions of each array) */
  uint32_t *error;                             /* the context's device error word (pinned host memory) */
  unsigned long long *stats;                   /* or nullptr: SV_STAT_WORDS diagnostics of the launch */
};
enum { SVS_START = 0, SVS_END, SVS_FRAMES, SVS_TILES, SVS_BATCHES, SVS_BATCH_LANES, SVS_ROTATIONS, SVS_WALK_LANE_TRIPS, SVS_WALK_TRIPS, SVS_SHADE_TILE_T, SVS_SHADE_BATCH_T, SVS_SHADE_TOTAL_T,
       SVS_POST_WAIT_T, SVS_DUMPS = 15 /* workgroups that gave up and left their control words behind */, SV_STAT_WORDS = 16,
       SV_DUMP_MAX = 4, SV_DUMP_WORDS = 72 /* blockIdx, wave, 64 control words, ... */, SV_STAT_TOTAL = SV_STAT_WORDS + SV_DUMP_MAX * SV_DUMP_WORDS };
struct ServerKernelArgs { FrameArgs fa; ServerArgs sa; };

bool server_kernel_fits(const DeviceScene &sc, uint32_t &ldsCount, uint32_t &ldsBytes, uint32_t xfSlots = 1u /* depth for a scene that moves */, uint32_t shadeWaves = 0u /* 0: the default of three slots */);
size_t server_rings_per_group();
/* fr: the slots stacked (frames = depth); its views are NOT used (they come through the mailbox).  0, or -1 if the kernel does not fit */
int launch_server(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, const ServerArgs &sa, uint32_t compute_units, hipStream_t stream);

}  // namespace flx
#endif
