/* flx_kernels.h — host-visible launchers of the HIP kernels (flx_kernels.hip). */
#ifndef FLX_KERNELS_H
#define FLX_KERNELS_H

#include "flx_device.h"

#ifndef FLX_PATHS_ANGLE_TABLE
#define FLX_PATHS_ANGLE_TABLE 1        /* k_paths reads the shading's per-triangle table too (DeviceScene::angle_tan): at the seven waves per SIMD it ran at first the table's dependent load cost it 1 %,
                                        * at four it gains 5 % — theater 9.29 -> 8.80 ms (profiles/r04_paths_occupancy.txt) */
#endif
namespace flx {

struct GBufferPtrs {
  float4 *color, *color_ip, *original_color, *id, *original_id, *location_id;
  /* the same five planes as the RGBA8 render targets the filter chain reads (stored as k_quantize would store them), or null */
  uint32_t *q_color = nullptr, *q_color_ip = nullptr, *q_original_color = nullptr, *q_id = nullptr, *q_original_id = nullptr;
};

/* counters: 8 x u64 in flx_counters order, or nullptr (no counting code is compiled in). */
void launch_trace_pixels(const DeviceScene &sc, const DeviceFrame &fr, float4 *out, const GBufferPtrs &gb,
                         unsigned long long *counters, hipStream_t stream, int sample_parallel = 0);
/* v2 pipeline: primary hits (float4 s,u,v,triangleId-as-bits per pixel) -> persistent path kernel -> resolve. */
uint32_t path_item_count(const DeviceFrame &fr);
uint64_t path_item_count64(const DeviceFrame &fr);
void launch_primary(const DeviceScene &sc, const DeviceFrame &fr, float4 *hits, unsigned long long *counters, hipStream_t stream);
void launch_paths(const DeviceScene &sc, const DeviceFrame &fr, const float4 *hits, float4 *sampleRadiance, float4 *lastOriginal,
                  uint32_t *queue, uint32_t blocks, unsigned long long *counters, hipStream_t stream);
/* sampleStride: float4 between the planes of two samples (0: the frame's own pixel count; the chained frame loop resolves one slot of a stacked workspace) */
/* tileTime (or nullptr): += what the paths of every 8 x 8 screen tile cost, for launch_tile_order */
void launch_resolve(const DeviceFrame &fr, const float4 *hits, const float4 *sampleRadiance, const float4 *lastOriginal, float4 *out,
                    hipStream_t stream, size_t sampleStride = 0, float *tileTime = nullptr);
/* the frame kernel's draw order over n screen tiles from their cost in the last frame (cleared for the next); mode 0: the lightest tenth last,
 * 1: sixteen classes, heaviest first; screen order inside a class */
void launch_tile_order(float *tileTime, uint32_t *order, uint32_t n, int mode, hipStream_t stream);
/* pipeline 3 (flx_wavefront.hip): per bounce a dense shade kernel and a persistent walk kernel; path state in HBM. */
constexpr int WF_MAX_BOUNCES = 250;
struct WavefrontBuffers {
  float4 *rec;                  /* 8 x float4 (128 B) per path item */
  uint32_t *live[2];            /* live path lists, alternating per bounce */
  uint32_t *counts;             /* [WF_MAX_ROUNDS + 2] slots used in the live list of round r */
  uint32_t *walkQueue;          /* [WF_MAX_ROUNDS + 2] per-round refill cursor of the walk kernel */
  uint32_t item_base, item_count; /* the path items [item_base, item_base + item_count) this group of launches owns */
  const float4 *hits;
  float4 *sampleRadiance, *lastOriginal;
  unsigned long long *counters; /* or nullptr */
  float4 *tailPool;             /* WF_TAIL_POOL_F4 float4 per walk workgroup: scratch of the tail consolidation */
  float4 *strag[2];             /* walks suspended by the walk kernel of round r (slot r & 1), WF_STRAG_F4 float4 each */
  uint32_t *stragCount;         /* [WF_MAX_ROUNDS + 2] walks suspended in round r */
  uint32_t *coopQueue;          /* [WF_MAX_ROUNDS + 2] cursor of the cooperative finisher over round r's suspended walks */
  /* Compact records of bounce 0 (or nullptr: full records).  The samples of a pixel share the primary hit, so what their
   * first shading yields splits into a part per pixel — next origin, shadow origin, albedo, base luminance: pix0, 3 float4,
   * indexed [screen tile][lane] —
   * and a part per sample — next direction + flags, shadow direction + length, lit colour: rec0, 3 float4; importancy is
   * (1,1,1) and the running colour 0 there.  48 B per path instead of a 128-byte line: shade0 writes, and the bounce-0 walk
   * kernel reads, 0.9 GB per 1080p x 8 frame instead of 2.1; the walk kernel's fold writes a full record for the paths
   * that go on (a fifth of them). */
  float4 *rec0, *pix0;
  uint32_t *frameRings;         /* the frame kernel's rings of path ids: WF_FRAME_RINGS x WF_FRAME_RING per walk workgroup (k_wf_frame), or nullptr */
  uint32_t front;               /* frame kernel: 1 = it also traces the primary rays and shades bounce 0 (hits need not be there, item_base must be 0) */
  uint32_t *error;              /* the context's device error word (pinned host memory, WF_ERR_* bits), or nullptr: a watchdog that trips says so here (flx_status FLX_ERR_DEVICE at the next point the host waits) */
  uint32_t watchdog;            /* frame kernels: polls after which a wave that waits gives up (0: FQ_WATCHDOG, seconds); fault injection sets it low */
  uint32_t inject;              /* fault injection (flx_debug_inject_fault): WF_INJECT_* */
  uint32_t walkJobs;            /* frame kernel with its front inside: low byte 2 = two walk jobs per lane (k_wf_frame2), else one (k_wf_frame); | WF_STAMP_COSTS: the variant whose walk lanes stamp what a path cost (adaptive tile order) */
  uint32_t tileCostPrimary;     /* tileCost (below) has a second half for the primary rays' visits per tile */
  const uint32_t *tileOrder;    /* frame kernel with its front inside: the screen tile the q-th draw from the frame's tile queue makes (a permutation of the frame's tiles), or nullptr: tile q */
  unsigned long long *tileCost; /* counted frames: entries visited by the paths of every screen tile (flx_debug_tile_cost), or nullptr */
};
/* the arguments of the shade kernels and the frame kernels, read from the kernarg segment where they are used (flx_frame_common.h) */
struct FrameArgs { DeviceScene sc; DeviceFrame fr; WavefrontBuffers wb; };
constexpr uint32_t WF_FRAME_RING = 16384;
constexpr uint32_t WF_STAMP_COSTS = 0x100u;   /* WavefrontBuffers::walkJobs */
/* device error word: who gave up */
constexpr uint32_t WF_ERR_SHADE_WATCHDOG = 1u, WF_ERR_WALK_WATCHDOG = 2u, WF_ERR_LIST = 4u, WF_ERR_LEFTOVER = 8u, WF_ERR_RING_SLOT = 16u, WF_ERR_SERVER_IDLE = 32u, WF_ERR_SERVER_TIMEOUT = 64u;
constexpr uint32_t WF_INJECT_NO_SHADING = 1u;      /* the shade waves of a frame kernel drop what they pop: the paths never come back and the walk waves' watchdog must trip */
constexpr uint32_t WF_FRAME_RINGS = 3;       /* to shade, to walk, fresh (tile, sample) units */
constexpr size_t WF_TAIL_POOL_F4 = 1024 * 8;
constexpr uint32_t WF_STRAG_F4 = 5;
constexpr int WF_MAX_ROUNDS = 2 * WF_MAX_BOUNCES;    /* regular rounds + the rounds that drain suspended walks */
size_t wavefront_live_capacity(const DeviceFrame &fr, uint32_t compute_units);
/* walk_scheduler: bit 0: 0 = one walk per lane (k_wf_walk_pre / k_wf_walk), 1 = workgroup-wide test queues (flx_walkq.hip);
 * bit 1: suspended walks are finished by k_wf_walk_coop (a wave per walk) instead of being carried to the next round */
/* suspend_max: walks a walk workgroup may hand over to the next round instead of finishing them (0 = never) */
/* organisation: 0 automatic — the whole bounce loop in ONE persistent launch (k_wf_frame: walk waves and shade waves of a workgroup
 * hand paths to each other through LDS rings, no barrier between bounces) where the scene's transforms leave room in LDS, else
 * rounds; 1 = rounds (one k_wf_shade + k_wf_walk_pre pair per bounce); 2 = the frame kernel (rounds if it does not fit). */
bool wavefront_front_in_kernel(const DeviceScene &sc, const DeviceFrame &fr, uint32_t item_count, int walk_scheduler, uint32_t suspend_max, int organisation);
/* -> what ran: 1 rounds, 2 the frame kernel, 3 the frame kernel with the front of the frame inside it; -1: wb.front set but the frame kernel cannot run; -2: the walk
 * kernels' dynamic LDS limit could not be raised on this device */
int launch_wavefront(const DeviceScene &sc, const DeviceFrame &fr, const WavefrontBuffers &wb, uint32_t compute_units, bool count,
                     int walk_scheduler, uint32_t suspend_max, int organisation, hipEvent_t walk0_begin, hipEvent_t walk0_end, hipStream_t stream);
/* denoise chain (flx_filter.hip): 13 RGBA8 planes = the reference's RenderTexture[0..3], IpRenderTexture[0..3],
 * OriginalRenderTexture[0..1], IdRenderTexture[0..1], OriginalIdRenderTexture (pathtracerWGL2.js:224-252). */
struct FilterPlanes { uint32_t *R[4], *Ip[4], *O[2], *Id[2], *OId; };
/* anti-aliasing post passes over RGBA8 planes (modules/fxaa.js, modules/taa.js); taa planes newest first, null = zero texture */
void launch_fxaa(const uint32_t *plane, float4 *out, int W, int H, hipStream_t stream);
void launch_taa(const uint32_t *const planes[9], float4 *out, int W, int H, hipStream_t stream);
/* float4 plane -> RGBA8 plane (a render-target store) */
void launch_quantize(const float4 *src, uint32_t *dst, size_t n, hipStream_t stream);
void launch_angle_tan(const DeviceScene &sc, float4 *out, hipStream_t stream);      /* DeviceScene::angle_tan for the scene as it stands */
/* the chain over planes whose slot 0 (R[0], Ip[0], O[0], Id[0], OId) holds the frame */
void launch_filter_chain(const FilterPlanes &pl, float4 *out, int W, int H, int hdr, hipStream_t stream);
/* temporal accumulation (pathtracerWGL2.js:571-662) over rings of n RGBA8 planes, slot 0 = newest:
 * without filter -> canvas float4 `out`; with filter -> dColor / dIp (RenderTexture[0] / IpRenderTexture[0]). */
struct TemporalRings { const uint32_t *c[16], *ip[16], *id[16], *oid[16]; int n; };
void launch_temporal(const TemporalRings &rings, int W, int H, int hdr, int use_filter, uint32_t *dColor, uint32_t *dIp, float4 *out,
                     hipStream_t stream);
void launch_debug_math(int fn, const float *a, const float *b, float *out, uint32_t n, hipStream_t stream);
void launch_debug_intersect(int fn, const float *in, float *out, uint32_t n, hipStream_t stream);
bool launch_debug_walk(int variant, const DeviceScene &sc, const float *in, float *out, uint32_t n, hipStream_t stream);      /* false: the scene does not allow that variant */

}  // namespace flx
#endif
