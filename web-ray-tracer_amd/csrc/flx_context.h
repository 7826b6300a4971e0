/* flx_context.h — the context behind the C ABI and the internal helpers flx_api.hip shares with flx_group.hip (the RCCL
 * gather across contexts).  Private to the library. */
#ifndef FLX_CONTEXT_H
#define FLX_CONTEXT_H

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "flexlight_hip.h"
#include "flexlight_hip_debug.h"
#include "flx_kernels.h"
#include "flx_chain.h"
#include "flx_server.h"

struct flx_share;                                 /* flx_share.hip: this context's part in the ranks' shared frames */
typedef struct ncclComm *flx_nccl_comm;          /* = ncclComm_t (rccl.h), kept out of this header */

#ifndef FLX_WF_GROUPS
#define FLX_WF_GROUPS 1      /* measured on MI355X: 2-4 concurrent chains are slower than one (profiles/r01_ab_stream_groups.txt) */
#endif
constexpr int WF_MAX_GROUPS = 4;
#ifndef FLX_FRAME_CHAIN_DEFAULT
#ifndef FLX_SAMPLE_PARALLEL_DEFAULT
#define FLX_SAMPLE_PARALLEL_DEFAULT 1
#endif
#ifndef FLX_WALK_JOBS_DEFAULT
#define FLX_WALK_JOBS_DEFAULT 1
#endif
#define FLX_FRAME_CHAIN_DEFAULT 2      /* flx_set_frame_chain's default: the frame server where a frame is a rank's thin share */
#endif
constexpr int FLX_COUNTER_SLOTS = 80;          /* 8 work counters + 32 scheduler diagnostics (flx_get_diag) + 40 tail profile (flx_get_tail_diag) */
constexpr uint32_t WF_STRAG_MAX = 512;         /* most walks a walk workgroup can suspend */

extern thread_local std::string g_create_error;

struct flx_context {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;
  hipDeviceProp_t prop;
  /* resident scene */
  float4 *d_geometry = nullptr, *d_attributes = nullptr, *d_rotation = nullptr, *d_shift = nullptr;
  float4 *d_walk = nullptr;                      /* threaded hot-first copy of the skip list */
  uint32_t walk_entries = 0, walk_hot = 0, walk_root = 0, walk_fast_boxes = 0;
  float4 *d_fwd = nullptr;                       /* the live entries in the reference's order (every successor further on): primary walk, lockstep walk */
  uint32_t fwd_entries = 0, fwd_root = 0, lock_boxes = 0;
  int last_organisation = 0;                     /* flx_last_organisation: what launch_wavefront ran for the last frame (0: another pipeline) */
  int frame_front = 1;                           /* flx_set_frame_front: the frame kernel traces the primary rays and shades bounce 0 itself (0 two kernels in front, 1 automatic, 2 inside wherever the frame kernel runs, 3 one kernel in front) */
  uint32_t *d_frame_rings = nullptr;             /* k_wf_frame: per chain and workgroup three rings of WF_FRAME_RING path ids */
  int frame_rings_chains = 0;                    /* chains it has slices for */
  int wf_organisation = 0;                       /* wavefront pipeline: 0 automatic, 1 rounds, 2 frame kernel (flx_set_wavefront_organisation) */
  bool lock_ok = false;                          /* the scene is small and in one object space: its bounce walks may go in lockstep */
  bool lock_use = true;                          /* flx_set_lockstep */
  bool gb_float_wanted = false;                  /* flx_render was given `gbuffers`: the filter frame keeps its float G-buffers */
  int walk_scheduler = 0;
  int sample_parallel = FLX_SAMPLE_PARALLEL_DEFAULT;      /* flx_debug_set_sample_parallel: k_trace_samples instead of k_trace_pixels where the frame allows it */
  /* Adaptive tile order (flx_debug_set_adaptive_order; on by default): k_resolve sums what every screen tile's paths cost (time in walk lanes), k_tile_order makes the
   * next frame's draw order of it — the lightest tiles last, so that the launch does not end in the chains of a heavy tile's paths.  Frames do not depend on the order. */
  int adaptive_order = 1;
  float *d_tile_time = nullptr; uint32_t *d_auto_order = nullptr; uint32_t tile_time_cap = 0;
  uint32_t auto_order_tiles = 0, auto_order_width = 0, auto_order_rows = 0; int auto_order_mode = -1;      /* the frame shape d_auto_order is for (0 tiles: none yet) */
  uint32_t *d_tile_order = nullptr; uint32_t tile_order_n = 0;      /* flx_debug_set_tile_order: the frame kernel's draw order over the frame's screen tiles */
  unsigned long long *d_tile_cost = nullptr; uint32_t tile_cost_n = 0;      /* flx_debug_tile_cost: counted frames' visits per screen tile */
  uint32_t walk_jobs = FLX_WALK_JOBS_DEFAULT;    /* flx_debug_set_walk_jobs: walk jobs per lane of the frame kernel's walk waves */
  int32_t *d_ids = nullptr;
  float *d_lights = nullptr;
  uchar4 *d_atlas[3] = { nullptr, nullptr, nullptr };
  uint32_t atlas_w[3] = { 0, 0, 0 }, atlas_h[3] = { 0, 0, 0 };
  uint32_t n_entries = 0, n_ids = 0, n_transforms = 0, n_lights = 0;
  uint32_t max_transform = 0;                   /* largest transform number an entry names */
  bool have_scene = false, have_transforms = false, have_lights = false;
  /* DeviceScene::angle_tan: per triangle, from the geometry / attribute arrays and this context's transforms; made again (on this context's stream, in front of
   * the frame that needs it) when any of them changed: angle_key = the versions it was made from */
  float4 *d_angle_tan = nullptr;
  size_t angle_capacity = 0;
  uint64_t angle_key = ~0ull;
  uint32_t geometry_version = 0;                 /* counts uploads of the geometry / attribute arrays (the second lane copies the primary's: mirror_scene) */
  uint32_t transforms_version = 0;               /* counts uploads of this context's transforms */
  int angle_table = 1;                           /* flx_debug: 0 = the shading computes the values itself */
  /* frame workspace */
  float4 *d_out = nullptr;
  size_t out_capacity = 0;                       /* pixels */
  float4 *d_gb[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  size_t gb_capacity = 0;
  uint32_t *d_planes[13] = {};                   /* the filter chain's RGBA8 render targets */
  uint32_t *d_qbatch = nullptr;                  /* batches of filter frames: the five render targets of every frame, 5 x pixels */
  size_t qbatch_capacity = 0;                    /* pixels */
  size_t planes_capacity = 0;
  /* temporal history: rings of RGBA8 planes (colour, colour ip, location id, original id), newest at ring_head */
  uint32_t *d_ring[4][16] = {};
  int ring_n = 0, ring_head = 0;
  uint32_t ring_w = 0, ring_h = 0;
  /* v2 pipeline workspace: primary hits, per-(sample,pixel) radiance, last sample's originalColor, item queue */
  float4 *d_hits = nullptr, *d_samples = nullptr, *d_last = nullptr;
  size_t hits_capacity = 0, samples_capacity = 0, last_capacity = 0;
  uint32_t *d_queue = nullptr;
  /* pipeline 3 (wavefront) workspace */
  float4 *d_rec = nullptr;
  float4 *d_tail_pool = nullptr;                 /* per walk workgroup: WF_TAIL_POOL_F4 float4 */
  uint32_t *d_aa[10] = {};                       /* RGBA8 planes of the anti-aliasing passes: [0..8] the TAA ring, [9] FXAA's input */
  size_t aa_capacity = 0;
  uint32_t aa_w = 0, aa_h = 0;
  int taa_head = 0, taa_filled = 0;
  float4 *d_aa_io[2] = {};                       /* staging for the host-pointer variants */
  size_t aa_io_capacity = 0;
  float4 *d_rec0 = nullptr, *d_pix0 = nullptr;   /* compact bounce-0 records: 3 float4 per path, 3 float4 per pixel */
  size_t rec0_capacity = 0, pix0_capacity = 0;
  float4 *d_strag = nullptr;                     /* per chain 2 x (walk workgroups x strag_walks) suspended walks; allocated only while suspension is on */
  size_t strag_capacity = 0;                     /* float4 units */
  uint32_t walk_suspend = 0;                     /* walks a walk workgroup may leave to the next round (0 = off) */
  size_t rec_capacity = 0;                       /* float4 units */
  uint32_t *d_live[2] = { nullptr, nullptr };
  size_t live_capacity = 0;
  uint32_t *d_wfcounts = nullptr;                /* per chain: counts, walkQueue, stragCount, [WF_MAX_ROUNDS + 2] each */
  int pipeline = 0;                              /* 0 auto, 1 per-pixel megakernel, 2 persistent paths, 3 wavefront */
  int last_pipeline = 0;                         /* what the last frame ran */
  int wf_groups = FLX_WF_GROUPS;                 /* wavefront pipeline: independent item groups on separate streams (tails of one overlap the other) */
  hipStream_t aux_stream[3] = { nullptr, nullptr, nullptr };
  hipEvent_t ev_fork = nullptr, ev_join[3] = { nullptr, nullptr, nullptr };
  /* device error word: pinned, device-mapped; a frame kernel's watchdog that trips sets WF_ERR_* bits in it and the host returns FLX_ERR_DEVICE where it next waits */
  uint32_t *h_dev_error = nullptr, *d_dev_error = nullptr;
  uint32_t inject_watchdog = 0, inject_flags = 0; /* flx_debug_inject_fault */
  unsigned long long *d_counters = nullptr;
  bool counters_enabled = false;
  flx_counters last_counters = {};
  hipEvent_t ev_frame0 = nullptr, ev_frame1 = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
  bool timed = false;
  /* several GPUs (flx_group.hip): this context's RCCL communicator and the buffers of the gather */
  flx_nccl_comm comm = nullptr;
  flx_nccl_comm comm_twin = nullptr;             /* a second communicator over the same ranks (ncclCommSplit) for the frame loop's second lane */
  int last_gather_root = -1;                     /* how the last gathered frame was exchanged: -1 all-gather, else the receiving rank */
  int comm_rank = 0, comm_size = 1;
  bool comm_owned = false;                       /* made by flx_comm_init_rank (else by a group's ncclCommInitAll) */
  float4 *d_send = nullptr, *d_recv = nullptr;   /* this rank's packed strips; every rank's */
  size_t send_capacity = 0, recv_capacity = 0;   /* float4 units */
  float4 *d_send8 = nullptr;                     /* this rank's strips as RGBA8 texels (flx_render_gathered_rgba8_device) */
  size_t send8_capacity = 0;
  float4 *d_frames = nullptr;                    /* group mode: the gathered frames in image order */
  size_t frames_capacity = 0;
  float4 *d_gplanes = nullptr;                   /* filter frames: the five gathered render targets in image order */
  size_t gplanes_capacity = 0;
  /* the frame loop (flx_frame_begin / flx_frame_end): two slots of device output + pinned host memory, a copy stream */
  /* (three slots where flx_set_frame_lanes(3) lets three chained frames be in flight, two otherwise) */
  float4 *d_slot[3] = { nullptr, nullptr, nullptr };
  size_t slot_capacity[3] = { 0, 0, 0 };         /* pixels */
  uint32_t *d_slot8[3] = { nullptr, nullptr, nullptr };
  size_t slot8_capacity[3] = { 0, 0, 0 };
  void *h_slot[3] = { nullptr, nullptr, nullptr };
  size_t h_slot_capacity[3] = { 0, 0, 0 };       /* bytes */
  size_t slot_bytes[3] = { 0, 0, 0 };
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_slot_start[3] = {}, ev_slot_traced[3] = {}, ev_slot_done[3] = {};
  uint64_t frames_begun = 0, frames_ended = 0;
  bool slot_host[3] = { true, true, true };      /* the slot's frame is copied to pinned host memory (else it stays in d_slot) */
  /* two lanes: a twin context (own stream + workspace, shared static scene arrays) takes every other frame of the loop */
  flx_context *twin = nullptr;
  bool is_twin = false;
  int frame_lanes = 2;
  uint64_t lane_next = 0;
  struct { flx_context *lane; int slot; } fifo[3] = {};
  int fifo_n = 0;
  std::vector<float> h_lights, h_rotation, h_shift;      /* host copies of what changes per frame, for the twin's own buffers */
  uint64_t dyn_version = 0, twin_dyn_version = 0;
  /* the chained frame loop (flx_chain.hip): consecutive frames of flx_frame_begin / _end overlap inside the persistent launch */
  int frame_chain = FLX_FRAME_CHAIN_DEFAULT;                           /* flx_set_frame_chain: 0 never, 1 where the frame loop's second lane would be used and the frame kernel takes the frame */
  flx::ChainSlot *d_chain_slots = nullptr;       /* [2] */
  flx::ChainMail *h_chain_mail = nullptr;        /* pinned host memory: the next frame's view is posted (plain stores) while the kernel runs */
  flx::ChainMail *d_chain_mail = nullptr;        /* its device address */
  flx::ChainMail *d_chain_relay = nullptr;       /* device memory: the post as the kernel's relaying waves pass it on to the other workgroups */
  uint32_t *d_chain_lists = nullptr;             /* resume lists: [slot: 3][set: 2][walk, shade, ready] x chain_list_cap entries */
  size_t chain_list_cap = 0;
  uint32_t *d_chain_order = nullptr;             /* flx_set_chain_order: the order of a slot's screen tiles, or nullptr */
  size_t chain_order_n = 0;
  uint32_t *d_chain_cost = nullptr;              /* flx_set_chain_cost: 2 x chain_cost_n per-tile counts */
  size_t chain_cost_n = 0;
  float4 *d_chain_susp = nullptr;                /* walks suspended in flight: [slot: 3][set: 2] x chain_susp_cap x CH_SUSP_F4 float4 */
  size_t chain_susp_cap = 0;
  uint32_t *d_chain_rings = nullptr;             /* per workgroup CH_RINGS rings; all slots WF_INVALID between launches */

  unsigned long long *d_chain_stats = nullptr;   /* flx_set_chain_stats: CH_STAT_LAUNCHES x CH_STAT_WORDS diagnostics, by sequence number */
  uint64_t chain_seq = 0;                        /* sequence number of the last chained frame begun (0: no chain stands) */
  uint32_t chain_counter = 0;                    /* sequence numbers handed out (never 0) */
  uint32_t chain_slot = 0;                       /* the slot of that frame */
  uint32_t chain_depth = 0;                      /* slots of the chain that stands (2 or 3) */
  flx_frame_params chain_params = {};            /* its shape: a frame continues the chain only with the same one */
  uint64_t chain_scene_version = 0;              /* ... and the same scene */
  uint64_t scene_version = 0;                    /* bumped by every upload */
  uint64_t begin_scene_version = 0;              /* ... as the last flx_frame_begin found it (0: no frame begun yet) */
  int last_chained = 0;                          /* flx_last_chained: 0 the last frame of the loop was not chained, 1 it began a chain, 2 it continued one */
  /* the frame server (flx_server.hip): one persistent launch renders the loop's frames as they are posted (flx_set_frame_chain mode 2) */
  flx::ServerSlot *d_sv_slots = nullptr;
  flx::ServerMail *h_sv_mail = nullptr, *d_sv_mail = nullptr, *d_sv_relay = nullptr;      /* pinned host memory (and its device address); device memory */
  uint32_t *d_sv_rings = nullptr;
  float4 *d_sv_out = nullptr;                    /* the launch's resolved frames: [slot] x sv_out_pixels */
  size_t sv_out_capacity = 0, sv_out_pixels = 0;
  uint32_t *d_sv_tiles = nullptr;                /* [workgroup][slot] x sv_tile_cap: the screen tiles a workgroup made of a frame */
  size_t sv_tile_cap = 0;
  float slot_latency_ms[3] = { -1.f, -1.f, -1.f }; /* server frames: post .. complete on the host's clock (flx_frame_end's gpu_ms); < 0: timed by events */
  const void *slot_dev_ptr[3] = { nullptr, nullptr, nullptr };      /* where the frame of an output slot really is in device memory when that is not d_slot[k] (server frames) */
  unsigned long long *d_sv_stats = nullptr;
  hipStream_t sv_stream = nullptr;
  bool sv_running = false;
  uint32_t sv_depth = 0, sv_next_seq = 0, sv_next_slot = 0, sv_counter = 0;
  flx_frame_params sv_params = {};               /* the shape of the frames the running launch takes */
  uint64_t sv_scene_version = 0;
  /* A scene that MOVES (changed lights / transforms of the same counts, frame after frame): the launch takes those arrays with every frame (ServerMail::blob) and
   * goes on over their uploads. */
  uint64_t structure_version = 0;                /* bumped by every upload but those (scene_version counts them all) */
  uint64_t sv_structure_version = 0;
  bool sv_ver = false;                           /* the running launch takes the lights and transforms per frame */
  uint32_t dyn_device_stale = 0;                 /* bit 0 / 1: d_rotation + d_shift / d_lights are behind h_rotation .. (uploads the launch went on over): dyn_flush */
  bool sv_want_ver = false;                      /* the scene has moved since it was uploaded: the next launch does */
  int sv_moving = 1;                             /* flx_set_server_moving_scenes: 0 = never (every changed upload ends the launch, as before round 4) */
  uint32_t *d_sv_versions = nullptr;             /* [3 arrays][workgroup x depth + slot]: the launch's versions of rotation / shift / lights */
  bool sv_out8 = false;                          /* the running launch resolves RGBA8 (ServerArgs::out8) */
  bool sv_target8 = false;                       /* flx_frame_target_set8: the target images are uint32 RGBA8 per pixel */
  float4 *sv_target[3] = { nullptr, nullptr, nullptr };      /* flx_frame_target_set: whole images the launch resolves this context's strips into (a peer GPU's memory, pinned host memory, ..) */
  uint64_t sv_target_posted = 0;                 /* frames posted since flx_frame_target_set: frame g goes to image g % n, whatever launch takes it */
  uint32_t sv_target_slots = 0;                  /* 0: none — the launch's own d_sv_out */
  flx_share *share = nullptr;
  uint32_t sv_groups = 0;                        /* flx_debug_set_server_groups: workgroups of the launch (0: one per CU) — two launches beside each other on one GPU, to rehearse a device group */
  struct { bool valid; uint32_t seq, slot; int format; flx::DeviceFrame fr; std::chrono::steady_clock::time_point posted; } sv_pending[3] = {};      /* per output slot: the server frame that will land there */
  /* uploads: capacity of every persistent scene buffer (keyed by the address of its pointer), pinned staging ring */
  std::map<void **, size_t> upload_capacity;
  uint8_t *stage = nullptr;
  hipEvent_t stage_done[8] = {};
  bool stage_used[8] = {};
  int stage_next = 0;
};

#define FLX_HIP(ctx, expr)                                                                    \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
      return FLX_ERR_DEVICE;                                                                  \
    }                                                                                         \
  } while (0)

flx_status flx_fail(flx_context *ctx, flx_status code, const char *msg);


/* flx_group.hip: this context's strips traced, exchanged over its communicator (root < 0: all-gather; else only `root` receives) and
 * put in image order, all enqueued on its stream */
flx_status flx_gather_enqueue(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, int root, void *d_frames, bool rgba8 = false);

/* flx_api.hip */
flx_status flx_make_frame(flx_context *ctx, const flx_frame_params *p, flx::DeviceScene &sc, flx::DeviceFrame &fr);
flx_status flx_make_batch(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, flx::DeviceScene &sc, flx::DeviceFrame &fr);
flx_status flx_run_frame(flx_context *ctx, const flx::DeviceScene &sc, const flx::DeviceFrame &fr, float4 *d_out, const flx::GBufferPtrs &gb);
flx_status flx_ensure_pixels(flx_context *ctx, float4 **buf, size_t *cap, size_t pixels);
int flx_server_takes_moving_scene(const flx_context *ctx);                       /* the scene has moved and its lights and transforms fit a post: the server's launches take them per frame */
int flx_server_continues(flx_context *ctx, const flx_frame_params *params);      /* the running launch of the frame server takes this frame as it is */
flx_status flx_server_prepare(flx_context *ctx, const flx_frame_params *params); /* the launch ends; everything a launch for frames like this needs is allocated */
flx_status flx_server_stop(flx_context *ctx);      /* the frame server's launch ends (after the frames posted to it), the frames in flight are resolved into their output slots */
flx_status flx_check_device_error(flx_context *ctx);      /* FLX_ERR_DEVICE (and the word cleared) if a frame kernel's watchdog has tripped since the last check */
/* flx_filter_planes_device; stamp_start = false leaves the frame's start event alone (the trace of the same frame recorded it) */
flx_status flx_filter_planes_enqueue(flx_context *ctx, const flx_frame_params *params, const void *d_planes, void *d_out_rgba, bool stamp_start);

#endif
