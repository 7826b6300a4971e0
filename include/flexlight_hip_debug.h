/*
 * flexlight_hip_debug.h — instrumentation, A/B knobs and test hooks of libflexlight_hip.so.
 *
 * NOT the drop-in boundary (that is include/flexlight_hip.h: what a FlexLight host binds).  What is here exists for the parity tests
 * (device-side evaluators of single routines, work counters compared with the oracle's), for A/B measurements (which kernel organisation runs a frame:
 * every choice renders the same frame, bit for bit), for fault injection and for the launch diagnostics tools/ prints.  Same library, same ABI rules
 * (extern "C", plain pointers and sizes); a host never needs any of it.  tests/, tools/ and bench.py bind these through flexlight_hip/capi.py.
 */
#ifndef FLEXLIGHT_HIP_DEBUG_H
#define FLEXLIGHT_HIP_DEBUG_H

#include "flexlight_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Counters of the last flx_render_device frame (collected only when enabled; costs atomics). */
flx_status flx_set_counters_enabled(flx_context *ctx, int enabled);
flx_status flx_get_counters(flx_context *ctx, flx_counters *out);

/* The last frame begun in the loop: 0 its own launches, 1 it began a chain of launches, 2 it continued one, 3 it went to the frame server. */
flx_status flx_last_chained(flx_context *ctx, int *chained);

/* A scene that MOVES in the frame server.  The reference refills its transform UBO and its light texture before every frame (modules/pathtracerWGL2.js:258-262,
 * 361-365) and an example like examples/dragon.js turns an object every tick.  Once flx_transforms_upload / flx_lights_upload has brought CHANGED contents of the
 * same counts, the server's next launch takes those arrays WITH every frame — they are posted with the frame's view, a version per frame in flight, and a later
 * upload no longer ends the launch (round 4 before this: every changed upload ended it and the frames went to the lanes).  Scenes whose 32 words per transform + 6
 * per light exceed 1024 keep the old behaviour, and so does flx_set_server_moving_scenes(ctx, 0).  Frames are bit-identical either way.
 * flx_server_moving: 1 while a launch of that kind is running. */
flx_status flx_set_server_moving_scenes(flx_context *ctx, int on);
int flx_server_moving(const flx_context *ctx);

/* Would flx_frame_begin hand this frame to the frame server (under flx_set_frame_chain(ctx, 3): whatever its size)?  1 / 0. */
int flx_frame_server_takes(flx_context *ctx, const flx_frame_params *params);

/* Device faults reach the status code.  The frame kernels' wait loops have watchdogs (seconds); a wave that gives up — or finds a ring slot that never
 * fills — sets a bit in the context's device error word (pinned host memory), and the next call in which the host waits for frames (flx_render,
 * flx_render_batch, flx_frame_end, flx_sync) returns FLX_ERR_DEVICE with the bits in flx_last_error and clears the word: the frame is incomplete.  A healthy
 * frame never gets there; this hook forces it for tests: the next frames' kernels give up after `watchdog_polls` polls (0: the built-in limit) and, with
 * FLX_INJECT_NO_SHADING, their shade waves drop every batch they pop, so that the walk waves wait for paths that never come back. */
#define FLX_INJECT_NO_SHADING 1u
flx_status flx_debug_inject_fault(flx_context *ctx, uint32_t watchdog_polls, uint32_t flags);
/* The shading keeps a per-triangle table of what fragment:500-512 derives from a triangle and its transform alone (three acos and three tan per shade otherwise),
 * made again at scene / transform uploads.  0: every shade computes the values itself — the same floats; for A/B runs. */
flx_status flx_debug_set_angle_table(flx_context *ctx, int on);
/* The per-pixel kernel (filter / temporal frames, small scenes) with the samples of a pixel side by side: k_trace_samples — a workgroup per 8 x 8 screen tile, a wave per
 * sample, the cross-sample globals of the shader (fragment:83-89) replayed in the shader's order afterwards — for frames of 2, 4 or 8 samples and at most 4 bounces; 0: the
 * sample-sequential k_trace_pixels always.  Frames, G-buffers and work counters are identical; for A/B runs (profiles/r05_sample_parallel.txt). */
flx_status flx_debug_set_sample_parallel(flx_context *ctx, int on);
/* Walk jobs per lane of the frame kernel's walk waves: 1 = k_wf_frame (1 024-thread workgroups, a path's walks per lane), 2 = k_wf_frame2 (512-thread workgroups, two
 * independent jobs per lane, a box phase and a triangle phase per trip; only where the front of the frame is inside the launch), 0 = the library's default.  Frames and
 * work counters are identical; for A/B runs (profiles/r05_two_walks.txt). */
flx_status flx_debug_set_walk_jobs(flx_context *ctx, int jobs);
/* the order in which the frame kernel draws a frame's 8 x 8 screen tiles: order[q] = the tile of the q-th draw (a permutation of the frame's n tiles; n = 0: tile q); frames do not depend on it */
flx_status flx_debug_set_tile_order(flx_context *ctx, const uint32_t *order, uint32_t n);
/* the adaptive tile order — the draw order made from what every tile cost in the last frame of the same shape (the lightest tiles last) — on (1, the default) or off (0: screen order) */
flx_status flx_debug_set_adaptive_order(flx_context *ctx, int on);
/* the sort behind it on its own (tests): the order k_tile_order makes of n per-tile costs — mode 1: sixteen classes of equal size, heaviest first, screen order inside a class (what the library uses);
 * 0: two classes, the lightest tenth last */
flx_status flx_debug_tile_order_of(flx_context *ctx, const float *cost, uint32_t n, int mode, uint32_t *order);
/* counted frames sum the entries visited by the paths of every screen tile: n > 0 turns that on (and zeroes the sums) for frames of up to n tiles, out copies the sums out first, n = 0 turns it off */
flx_status flx_debug_tile_cost(flx_context *ctx, unsigned long long *out, uint32_t n);
/* Rehearsal of a device group on ONE GPU: the frame server's launch takes `groups` CUs only (0: all), so that the launches of several contexts run beside
 * each other. */
flx_status flx_debug_set_server_groups(flx_context *ctx, uint32_t groups);
/* diagnostics of the frame server's last launch (csrc/flx_server.h: SVS_*): start, end, frames completed, tiles, batches, rotations, ... */
flx_status flx_get_server_stats(flx_context *ctx, uint64_t *out /* [16] */);
/* the control words of up to four workgroups of the frame server that gave up (72 words each: workgroup, wave, its 64 LDS control words, the relayed posts, the slots' tile cursors) */
flx_status flx_get_server_dump(flx_context *ctx, uint64_t *out /* [4 * 72] */);
/* ---- only in `make EXPERIMENTS=1`'s libflexlight_hip_experiments.so (flx_has_experiments): the chained launches of flx_set_frame_chain(ctx, 1), csrc/flx_chain.hip ---- */
#ifdef FLX_EXPERIMENTS
/* Diagnostics of the chained launches of mode 1 (tools/chain_stats.py): 64 launches (by sequence number mod 64) x 64 words — when the launch started and ended,
 * when the next frame's view was seen, when its own frame was complete, tiles made for either frame, paths handed to the next kernel, walks abandoned. */
flx_status flx_set_chain_stats(flx_context *ctx, int on);
flx_status flx_get_chain_stats(flx_context *ctx, uint64_t *out /* [64 * 64] */);
/* Experiments with the order in which a chained frame's 8 x 8 screen tiles are drawn (tools/chain_order.py): an explicit permutation of the frame's tiles
 * (n = 0: the row-major default), and per-tile counts of the shadings its paths took after bounce 0 (n tiles per slot; 2 x n words out). */
flx_status flx_set_chain_order(flx_context *ctx, const uint32_t *order, uint32_t n);
flx_status flx_set_chain_cost(flx_context *ctx, uint32_t n);
flx_status flx_get_chain_cost(flx_context *ctx, uint32_t *out);
#endif /* FLX_EXPERIMENTS */

/* ranks of the context's communicator as RCCL reports them (ncclCommCount); 0: the context belongs to none */
int flx_comm_count(const flx_context *ctx);

/* Kernel organisation of the path-trace pass.  0 = automatic: the sample-sequential per-pixel kernel when use_filter /
 * is_temporal need the cross-sample G-buffer state; for scenes of up to 128 entries the per-pixel kernel too, or — from 32 bounce
 * iterations per pixel (4 with four or more lights) in frames of at least 2^20 paths — the persistent path kernel; the wavefront
 * pipeline for everything larger; 1 = per-pixel kernel, 2 = persistent path kernel, 3 = wavefront pipeline.
 * Results are identical; the explicit values are for A/B timing and tests.  flx_last_pipeline: what the last frame ran. */
flx_status flx_set_pipeline(flx_context *ctx, int pipeline);
flx_status flx_last_pipeline(flx_context *ctx, int *pipeline);
/* Scenes of at most 128 entries that all stand in transform 0 (cornell, cornell.obj, the theater) are walked by the wave in
 * lockstep over the entries in the reference's order instead of lane by lane over the threaded copy (per-pixel and persistent
 * path kernels; same entries per ray, same arithmetic, same counters).  on = 0 switches that off, for A/B timing and tests. */
flx_status flx_set_lockstep(flx_context *ctx, int on);
/* Wavefront pipeline: run the bounce loop as 1..4 independent chains of screen-tile ranges on separate HIP
 * streams (default 1: more chains measured slower, profiles/r01_ab_stream_groups.txt), so that the tail of one chain's persistent walk kernel overlaps the other's work. */
flx_status flx_set_wavefront_groups(flx_context *ctx, int groups);
/* Wavefront pipeline: how the bounce loop is laid out on the GPU.  0 (default): automatic — ONE persistent launch for all bounces
 * (k_wf_frame: the walk waves and shade waves of a workgroup pass paths to each other through LDS rings; no kernel boundary, hence no
 * tail, between the bounces) where the scene's object spaces leave room in LDS, rounds otherwise; 1: rounds, a shade + walk kernel
 * pair per bounce (the round-1/2 organisation); 2: the frame kernel where it fits.  Frames and work counters are identical. */
flx_status flx_set_wavefront_organisation(flx_context *ctx, int organisation);
/* Wavefront pipeline: who traces the primary rays and shades bounce 0.  0: two kernels in front of the bounce loop (k_primary, k_wf_shade0); 3: ONE kernel in
 * front (k_wf_front: a wave traces the primary rays of its 8 x 8 screen tile and shades it straight away); 2: the frame kernel itself wherever it runs — its shade waves
 * make the fresh paths one screen tile at a time while its walk waves walk the earlier ones — and one kernel in front elsewhere; 1 (default): automatic — the frame
 * kernel itself where a workgroup gets at least 32 screen tiles (a whole 1080p frame: yes, a rank's eighth of one: no), else one kernel in front up to 128 M paths
 * a pass, two beyond.  Frames and work counters are identical. */
flx_status flx_set_frame_front(flx_context *ctx, int mode);
/* What the wavefront pipeline ran for the last frame: 1 rounds, 2 the frame kernel, 3 the frame kernel with the front of the frame inside it;
 * 0 when another pipeline rendered it (flx_last_pipeline). */
flx_status flx_last_organisation(flx_context *ctx, int *organisation);
/* Wavefront pipeline: how the bounce walks are scheduled.  Every mode walks every ray through the same entries with the
 * same arithmetic (frames and work counters are identical); they differ in speed and exist for A/B measurements
 * (profiles/r01_ab_tail_schedulers.txt).
 *   scheduler      FLX_WALK_LANES (default): one walk per lane, lanes refilled as walks end (k_wf_walk_pre)
 *                  FLX_WALK_QUEUES: walk states in LDS, waves take 64 walks that need the same test (flx_walkq.hip)
 *                  FLX_WALK_LANES_FINISHER: as FLX_WALK_LANES, suspended walks are finished a wave per walk (flx_walkcoop.hip)
 *   suspend_walks  FLX_WALK_LANES*: a walk workgroup that has found the queue dry and is down to this many walks hands
 *                  them over (to the next round's walk kernel, or to the finisher) instead of finishing them; 0 = never */
#define FLX_WALK_LANES 0
#define FLX_WALK_QUEUES 1
#define FLX_WALK_LANES_FINISHER 2
flx_status flx_set_walk_scheduler(flx_context *ctx, int scheduler, uint32_t suspend_walks);
/* 1 when the library carries the experimental schedulers above (`make EXPERIMENTS=1`: libflexlight_hip_experiments.so); the
 * shipped library returns 0 and its flx_set_walk_scheduler accepts (FLX_WALK_LANES, 0) only. */
int flx_has_experiments(void);

/* ---- diagnostics --------------------------------------------------------------------------------- */
/* Evaluate one of include/flx_math.h's routines on the GPU for n inputs (b may be NULL for unary
 * functions); used by tests to prove CPU/GPU bit equality.  fn: 0 sin 1 cos 2 tan 3 acos 4 atan2
 * 5 exp 6 pow 7 tanh 8 floor 9 sqrt 10 div. */
flx_status flx_debug_math(flx_context *ctx, int fn, const float *a, const float *b, float *out, uint32_t n);
/* Evaluate one of the intersection routines on the GPU, AS THE KERNELS CALL IT, for n rows; used by tests to hold the device code against literal
 * answers computed from the shader text (tests/golden/intersect_kat.json.gz).  fn 0: moellerTrumbore (fragment:123-140) through the walk kernels'
 * routine over stored edges (exact 1/det from v_rcp_f32, branch-free acceptance), 1: moellerTrumboreCull (:143-158) through the same routine, 2: rayCuboid
 * (:161-167) through the walk kernels' box test (interval test, exact quotients by reciprocal, IEEE division where their preconditions fail); 3, 4, 5: the
 * same three as the per-pixel kernel calls them.  Rows: triangles 16 floats (a, b, c, origin, direction, l), boxes 13 (l, origin, direction, min, max);
 * out: 3 floats per row for fn 0 and 3 ((s, u, v) of a hit, zeros otherwise), else one float 0 / 1. */
flx_status flx_debug_intersect(flx_context *ctx, int fn, const float *in, float *out, uint32_t n);
/* Walk n rays through the uploaded scene on the GPU, AS THE KERNELS DO, one ray per lane: rayTracer (fragment:172-227) and shadowTest (:230-279) of each ray;
 * used by tests to hold the device walks against literal answers computed from the shader text (tests/golden/walk_kat.json.gz).  variant 0: the wavefront
 * pipeline's lane walk over the threaded, hot-first copy with the rays pre-transformed into every object space; 1: the per-pixel / persistent kernels' lane
 * walk; 2: their wave-wide lockstep walk (scenes of at most 128 entries in one object space).  rays: 7 floats each (origin, direction, shadowTest's l);
 * out: 8 floats each: s, u, v, 2 x transform number and entry index of the closest hit (zeros and -1 for none), entries that walk fetched, shadowTest's
 * answer (0 / 1), entries the shadow walk fetched. */
flx_status flx_debug_walk(flx_context *ctx, int variant, const float *rays, float *out, uint32_t n);
/* Scheduler statistics of the last counted frame (wavefront pipeline): for bounce b = 0..3 (3 = all
 * later ones) out[2b] = wave-iterations of the walk kernel, out[2b+1] = fold/refill batches. */
flx_status flx_get_diag(flx_context *ctx, uint64_t out[32]);   /* out[8..12]: bounce-0 walk kernel stamps: fold, refill, step cycles, wave lifetime, waves; out[16+3b..]: per bounce sum / count / max of wave lifetimes */
/* Tail profile of the last counted frame (wavefront pipeline, walk kernel of the round given to the build by FLX_TAIL_DIAG_ROUND, default 0):
 * for k = 0 .. 11 out[3k], out[3k+1], out[3k+2] = sum / count / max over the walk workgroups of the shader cycles since the workgroup's start
 * at which its walks in flight first numbered <= 2^k; out[36..38] the same for the moment the workgroup found the walk queue dry. */
flx_status flx_get_tail_diag(flx_context *ctx, uint64_t out[40]);

#ifdef __cplusplus
}
#endif
#endif /* FLEXLIGHT_HIP_DEBUG_H */
