/*
 * flx_api.hip — the C ABI of include/flexlight_hip.h: context, uploads, one-frame driver.
 *
 * Host-side counterpart of what modules/pathtracerWGL2.js does around its draw call: keep the
 * scene arrays resident on the GPU (here: plain device buffers in HBM, uploaded once and reused
 * every frame; the reference re-uploads lights and transforms per frame, pathtracerWGL2.js:258-262,
 * 361-365), derive the per-frame constants, launch, hand the radiance back.  No CPU fallback.
 */
#include <hip/hip_runtime.h>

#include <chrono>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "flx_context.h"

using namespace flx;

#ifndef FLX_EXPERIMENTS
#define FLX_EXPERIMENTS 0                   /* Makefile: EXPERIMENTS=1 */
#endif
#ifndef FLX_FRONT_FUSED_MAX_ITEMS
#define FLX_FRONT_FUSED_MAX_ITEMS (128u << 20)
#endif
#ifndef FLX_FRONT_MIN_TILES_PER_CU
#define FLX_FRONT_MIN_TILES_PER_CU 24     /* (a quarter of a 1080p frame, 31.6 tiles per workgroup: 2.02 ms inside against 2.19 in front; an eighth, 15.9: 1.46 against 1.45; a sixteenth 1.27 against 1.21 — tools/front_rule.py) */
#endif
#ifndef FLX_ADAPTIVE_FRONT_MAX_TILES_PER_CU
#define FLX_ADAPTIVE_FRONT_MAX_TILES_PER_CU 48      /* front inside the frame kernel: frames below this get the stamped kernel and the adaptive tile order */
#endif
#ifndef FLX_COMM_RESERVED_CUS
#define FLX_COMM_RESERVED_CUS 8u            /* CUs a context with two gathering lanes leaves free of persistent walk workgroups */
#endif
#ifndef FLX_WF_ORGANISATION_DEFAULT
#define FLX_WF_ORGANISATION_DEFAULT 0       /* A/B builds: force rounds (1) or the frame kernel (2) whatever the context says */
#endif

thread_local std::string g_create_error;

#ifndef FLX_PATHS_LOCK_MIN_BOXES
#define FLX_PATHS_LOCK_MIN_BOXES 0          /* k_paths walks in lockstep from this many boxes on (flx_run_frame): always, since the kernel runs at 4 waves per SIMD (flx_kernels.hip: FLX_PATHS_WAVES) */
#endif
flx_status flx_fail(flx_context *ctx, flx_status code, const char *msg) {
  if (ctx) ctx->err = msg;
  return code;
}
static flx_status fail(flx_context *ctx, flx_status code, const char *msg) { return flx_fail(ctx, code, msg); }

extern "C" const char *flx_version(void) { return "flexlight-hip 0.1 (gfx950)"; }

extern "C" const char *flx_last_error(const flx_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" flx_status flx_context_create(int device, flx_context **out) {
  if (!out) { g_create_error = "flx_context_create: out is NULL"; return FLX_ERR_INVALID; }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("flx_context_create: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU path";
    return FLX_ERR_NO_GPU;
  }
  if (device < 0 || device >= n) { g_create_error = "flx_context_create: device index out of range"; return FLX_ERR_INVALID; }
  flx_context *ctx = new flx_context();
  ctx->device = device;
#if FLX_EXPERIMENTS
  if (const char *wj = getenv("FLX_WALK_JOBS")) { if (wj[0] == '1' || wj[0] == '2') ctx->walk_jobs = (uint32_t)(wj[0] - '0'); }      /* A/B runs of whole test suites and bench.py (flx_debug_set_walk_jobs) */
#endif
  auto bail = [&](const char *what, hipError_t err) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(err);
    delete ctx;
    return FLX_ERR_DEVICE;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
  if ((e = hipGetDeviceProperties(&ctx->prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
  if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  ctx->stream = ctx->own_stream;
  if ((e = hipEventCreate(&ctx->ev_frame0)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_frame1)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_k0)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_k1)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipHostMalloc((void **)&ctx->h_dev_error, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail("hipHostMalloc", e);
  ctx->h_dev_error[0] = 0u;
  if ((e = hipHostGetDevicePointer((void **)&ctx->d_dev_error, ctx->h_dev_error, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", e);
  if ((e = hipMalloc(&ctx->d_counters, FLX_COUNTER_SLOTS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(ctx->d_counters, 0, FLX_COUNTER_SLOTS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
  if ((e = hipMalloc(&ctx->d_queue, sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&ctx->d_wfcounts, WF_MAX_GROUPS * 4 * (WF_MAX_ROUNDS + 2) * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
  for (int i = 0; i < 3; i++) {
    if ((e = hipStreamCreateWithFlags(&ctx->aux_stream[i], hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  }
  if ((e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  *out = ctx;
  return FLX_OK;
}

extern "C" void flx_context_destroy(flx_context *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->share) (void)flx_share_leave(ctx);
  if (ctx->sv_running && ctx->h_sv_mail) { __atomic_store_n(&ctx->h_sv_mail->stopAfter, ctx->sv_next_seq - 1u, __ATOMIC_RELEASE); ctx->sv_running = false; (void)hipStreamSynchronize(ctx->sv_stream); }
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->twin) { flx_context_destroy(ctx->twin); ctx->twin = nullptr; }
  if (ctx->is_twin) {                      /* the static scene arrays belong to the primary context */
    ctx->d_geometry = ctx->d_attributes = nullptr; ctx->d_ids = nullptr; ctx->d_walk = nullptr; ctx->d_fwd = nullptr;
    ctx->d_atlas[0] = ctx->d_atlas[1] = ctx->d_atlas[2] = nullptr;
  }
  (void)flx_comm_destroy(ctx);
  void *bufs[] = { ctx->d_geometry, ctx->d_attributes, ctx->d_rotation, ctx->d_shift, ctx->d_ids, ctx->d_lights,
                   ctx->d_atlas[0], ctx->d_atlas[1], ctx->d_atlas[2], ctx->d_out, ctx->d_gb[0], ctx->d_gb[1], ctx->d_gb[2],
                   ctx->d_gb[3], ctx->d_gb[4], ctx->d_gb[5], ctx->d_counters, ctx->d_hits, ctx->d_samples, ctx->d_last, ctx->d_queue,
                   ctx->d_send, ctx->d_send8, ctx->d_recv, ctx->d_frames, ctx->d_gplanes, ctx->d_angle_tan, ctx->d_rec, ctx->d_rec0, ctx->d_pix0, ctx->d_tail_pool, ctx->d_strag, ctx->d_live[0], ctx->d_live[1], ctx->d_wfcounts, ctx->d_walk, ctx->d_fwd, ctx->d_frame_rings, ctx->d_qbatch, ctx->d_tile_order, ctx->d_tile_cost, ctx->d_tile_time, ctx->d_auto_order,
                   ctx->d_planes[0], ctx->d_planes[1], ctx->d_planes[2], ctx->d_planes[3], ctx->d_planes[4], ctx->d_planes[5], ctx->d_planes[6],
                   ctx->d_planes[7], ctx->d_planes[8], ctx->d_planes[9], ctx->d_planes[10], ctx->d_planes[11], ctx->d_planes[12] };
  for (void *b : bufs) if (b) (void)hipFree(b);
  for (auto &ring : ctx->d_ring) for (uint32_t *pl : ring) if (pl) (void)hipFree(pl);
  for (uint32_t *pl : ctx->d_aa) if (pl) (void)hipFree(pl);
  for (float4 *b : ctx->d_aa_io) if (b) (void)hipFree(b);
  for (hipEvent_t ev : { ctx->ev_frame0, ctx->ev_frame1, ctx->ev_k0, ctx->ev_k1 }) if (ev) (void)hipEventDestroy(ev);
  for (int i = 0; i < 3; i++) { if (ctx->aux_stream[i]) (void)hipStreamDestroy(ctx->aux_stream[i]); if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  for (int i = 0; i < 3; i++) {
    if (ctx->d_slot[i]) (void)hipFree(ctx->d_slot[i]);
    if (ctx->d_slot8[i]) (void)hipFree(ctx->d_slot8[i]);
    if (ctx->h_slot[i]) (void)hipHostFree(ctx->h_slot[i]);
    for (hipEvent_t ev : { ctx->ev_slot_start[i], ctx->ev_slot_traced[i], ctx->ev_slot_done[i] }) if (ev) (void)hipEventDestroy(ev);
  }
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  for (hipEvent_t ev : ctx->stage_done) if (ev) (void)hipEventDestroy(ev);
  if (ctx->stage) (void)hipHostFree(ctx->stage);
  if (ctx->h_chain_mail) (void)hipHostFree(ctx->h_chain_mail);
  if (ctx->h_sv_mail) (void)hipHostFree(ctx->h_sv_mail);
  for (void *b : { (void *)ctx->d_sv_slots, (void *)ctx->d_sv_relay, (void *)ctx->d_sv_rings, (void *)ctx->d_sv_stats, (void *)ctx->d_sv_out, (void *)ctx->d_sv_tiles, (void *)ctx->d_sv_versions }) if (b) (void)hipFree(b);
  if (ctx->sv_stream) (void)hipStreamDestroy(ctx->sv_stream);
  for (void *b : { (void *)ctx->d_chain_slots, (void *)ctx->d_chain_relay, (void *)ctx->d_chain_lists, (void *)ctx->d_chain_rings, (void *)ctx->d_chain_stats, (void *)ctx->d_chain_susp, (void *)ctx->d_chain_order, (void *)ctx->d_chain_cost }) if (b) (void)hipFree(b);
  if (ctx->h_dev_error) (void)hipHostFree(ctx->h_dev_error);

  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

/* Scene arrays live in device buffers that persist across uploads: an upload of the same or a smaller size reuses the
 * buffer (the reference refills its transform UBO and light texture every frame, pathtracerWGL2.js:258-262, 361-365 — a
 * per-frame hipFree + hipMalloc would put two device synchronisations into every frame).  Small arrays (transforms, lights:
 * a few hundred bytes per frame) go through a ring of pinned staging slots and are copied in stream order, without waiting
 * for the frames already enqueued; large ones are copied from the caller's memory and waited for.  Either way the caller's
 * buffer is not retained. */
constexpr size_t STAGE_SLOT_BYTES = 64 * 1024;
constexpr int STAGE_SLOTS = 8;

/* The static scene arrays (geometry, attributes, ids, the threaded and forward-ordered copies, atlases) are SHARED with the
 * frame loop's second lane (mirror_scene): its frames read them on another stream.  An upload of one of them therefore first
 * waits for the twin's frames in flight (they must not see the array change under them, nor new metadata over old contents)
 * and ends with the copy complete, so that the twin's next frame — enqueued on its own stream, which does not order itself after
 * the primary's — finds the new contents.  Scene and atlas uploads are per scene, not per frame; what changes per frame
 * (lights, transforms) lives in per-lane buffers and stays asynchronous. */
static flx_status shared_upload_begin(flx_context *ctx) {
  if (ctx->twin) FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream));
  return FLX_OK;
}
static flx_status shared_upload_end(flx_context *ctx) {
  if (ctx->twin) FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FLX_OK;
}

template <typename T>
static flx_status upload(flx_context *ctx, T **dst, const void *src, size_t bytes) {
  { flx_status ss = flx_server_stop(ctx); if (ss) return ss; }      /* (a running frame server reads the scene) */
  ctx->structure_version++;              /* (the uploads a launch for a scene that moves goes on over do not come through here: flx_transforms_upload) */
  ctx->scene_version++;                  /* (a chain of frames does not go on over a changed scene: flx_chain.hip) */
  size_t &cap = ctx->upload_capacity[(void **)dst];
  if (bytes == 0) {                      /* "none": the kernels test the pointer */
    if (*dst && ctx->twin) FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream));
    if (*dst) { FLX_HIP(ctx, hipFree(*dst)); *dst = nullptr; }
    cap = 0;
    return FLX_OK;
  }
  if (bytes > cap || !*dst) {
    if (*dst) {
      FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->twin) FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream));      /* (the frame loop's second lane may be reading a shared array) */
      FLX_HIP(ctx, hipFree(*dst)); *dst = nullptr;
    }
    cap = 0;
    FLX_HIP(ctx, hipMalloc(dst, bytes));
    cap = bytes;
  }
  if (bytes <= STAGE_SLOT_BYTES) {
    if (!ctx->stage) {
      FLX_HIP(ctx, hipHostMalloc((void **)&ctx->stage, STAGE_SLOT_BYTES * STAGE_SLOTS, hipHostMallocDefault));
      for (auto &ev : ctx->stage_done) FLX_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const int k = ctx->stage_next;
    ctx->stage_next = (k + 1) % STAGE_SLOTS;
    if (ctx->stage_used[k]) FLX_HIP(ctx, hipEventSynchronize(ctx->stage_done[k]));      /* the copy that last read this slot (eight uploads ago) */
    memcpy(ctx->stage + (size_t)k * STAGE_SLOT_BYTES, src, bytes);
    FLX_HIP(ctx, hipMemcpyAsync(*dst, ctx->stage + (size_t)k * STAGE_SLOT_BYTES, bytes, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipEventRecord(ctx->stage_done[k], ctx->stream));
    ctx->stage_used[k] = true;
    return FLX_OK;
  }
  if (ctx->twin) FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream));      /* a scene array the second lane may still be reading */
  FLX_HIP(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));          /* the caller's buffer is not retained */
  return FLX_OK;
}

/* Threaded, hot-first copy of the skip list (DeviceScene::walk).  The reference's array is the DFS
 * pre-order of the AABB tree with a skip count per node (scene.js:224-282); a walk only ever moves to
 * "the next entry" or "the next entry after the subtree".  Writing those two successors into every
 * entry makes the storage order free, so the shallow levels — which every ray crosses — go to the
 * front where the walk kernel keeps them in LDS.  Entry contents, the sequence of entries a given
 * ray visits and therefore every result are unchanged. */
#ifndef FLX_AB_HOT_ORDER
#define FLX_AB_HOT_ORDER 0
#endif
static void build_threaded(const float *geometry, uint32_t n, std::vector<float> &out, uint32_t &n_out, uint32_t &n_hot, uint32_t &root) {
  const uint32_t HOT_MAX = 4096;                 /* upper bound of entries worth ordering by depth */
  /* live entries: everything a walk can reach = all entries before the first terminator that is reached;
   * keep every non-terminator entry plus ONE shared terminator. */
  std::vector<uint32_t> depth(n, 0);
  {
    std::vector<uint32_t> stack;                 /* last index of the enclosing subtrees */
    for (uint32_t i = 0; i < n; i++) {
      while (!stack.empty() && i > stack.back()) stack.pop_back();
      depth[i] = (uint32_t)stack.size();
      const float *e = geometry + (size_t)i * 12;
      if (e[10] == 1.0f) stack.push_back(i + (uint32_t)e[6]);
    }
  }
  std::vector<uint32_t> order;                   /* original indices of non-terminator entries, hot first */
  order.reserve(n);
  for (uint32_t i = 0; i < n; i++) if (geometry[(size_t)i * 12 + 10] != 0.0f) order.push_back(i);
  /* shallowest HOT_MAX entries first (stable: by depth, then original index), the rest in original order */
  std::vector<uint32_t> byDepth(order);
  std::stable_sort(byDepth.begin(), byDepth.end(), [&](uint32_t a, uint32_t b) { return depth[a] < depth[b]; });
#if FLX_AB_HOT_ORDER      /* A/B builds only: the hot-first order from a file of original indices (tools: the oracle's visit histogram), to bound what a better choice of the LDS top can bring */
  if (const char *f = getenv("FLX_HOT_ORDER")) {
    if (FILE *fh = fopen(f, "rb")) {
      std::vector<uint32_t> given(byDepth.size());
      const size_t got = fread(given.data(), 4, given.size(), fh);
      fclose(fh);
      if (got == given.size()) byDepth = given;
    }
  }
#endif
  const uint32_t hot = (uint32_t)std::min<size_t>(HOT_MAX, byDepth.size());
  std::vector<char> isHot(n, 0);
  for (uint32_t k = 0; k < hot; k++) isHot[byDepth[k]] = 1;
  std::vector<uint32_t> newIndex(n, WALK_END);
  uint32_t next = 0;
  const uint32_t terminator = next++;            /* threaded index 0: the shared terminator (every full walk ends on it) */
  for (uint32_t k = 0; k < hot; k++) newIndex[byDepth[k]] = next++;
  for (uint32_t i : order) if (!isHot[i]) newIndex[i] = next++;
  n_out = next;
  n_hot = hot + 1;
  out.assign((size_t)n_out * 12, 0.0f);
  auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  /* link to original successor j from an entry with transform number `fromT` (flx_device.h: LINK_*) */
  auto succ = [&](uint64_t j, uint32_t fromT) -> uint32_t {
    if (j >= n) return WALK_END;                 /* loop bound reached: no fetch (fragment:184) */
    const float *e = geometry + (size_t)j * 12;
    if (e[10] == 0.0f) return terminator;        /* kind 0; a terminator's transform is never used */
    const uint32_t kind = e[10] == 1.0f ? 1u : 2u;
    return newIndex[j] | kind << LINK_KIND_SHIFT | ((uint32_t)e[9] != fromT ? LINK_XFORM : 0u);
  };
  root = succ(0, 0);                             /* a walk starts with the untransformed ray (cachedTI = 0, fragment:174) */
  for (uint32_t i : order) {
    const float *e = geometry + (size_t)i * 12;
    float *o = out.data() + (size_t)newIndex[i] * 12;
    const uint32_t type = e[10] == 1.0f ? 1u : 2u;
    const uint32_t meta = type | ((uint32_t)e[9] << 2);
    if (type == 1u) {
      for (int k = 0; k < 6; k++) o[k] = e[k];
      o[8] = bits(succ((uint64_t)i + 1, (uint32_t)e[9]));
      o[9] = bits(succ((uint64_t)i + 1 + (uint64_t)e[6], (uint32_t)e[9]));
      o[10] = bits(meta);
      o[11] = bits(i);
    } else {
      /* vertex a, then the two edges b - a and c - a of fragment:124-125 (the same single-precision subtractions the
       * shader does per visit, done once here) */
      for (int k = 0; k < 3; k++) { o[k] = e[k]; o[3 + k] = e[3 + k] - e[k]; o[6 + k] = e[6 + k] - e[k]; }
      o[9] = bits(succ((uint64_t)i + 1, (uint32_t)e[9]));
      o[10] = bits(meta);
      o[11] = bits(i);
    }
  }
  /* terminator entry stays all zero (meta type 0) */
}

/* The lockstep walk's copy (flx_device.h: walkLockPass): the live entries in the reference's own order — every successor of
 * an entry lies further on — in the threaded layout, with plain indices as links and the shared terminator last.  Built for
 * scenes of at most FLX_LOCK_MAX entries whose entries all stand in transform 0, i.e. are tested with the untransformed ray
 * (fragment:174-175). */
static void build_lockstep(const float *geometry, uint32_t n, std::vector<float> &out, uint32_t &n_out, uint32_t &root, uint32_t &boxes) {
  std::vector<uint32_t> order;
  boxes = 0;
  for (uint32_t i = 0; i < n; i++) if (geometry[(size_t)i * 12 + 10] == 1.0f) boxes++;
  for (uint32_t i = 0; i < n; i++) if (geometry[(size_t)i * 12 + 10] != 0.0f) order.push_back(i);
  const uint32_t terminator = (uint32_t)order.size();
  std::vector<uint32_t> newIndex(n, terminator);
  for (uint32_t k = 0; k < terminator; k++) newIndex[order[k]] = k;
  n_out = terminator + 1u;
  out.assign((size_t)n_out * 12, 0.0f);
  auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  auto succ = [&](uint64_t j) -> uint32_t { return j >= n ? WALK_END : newIndex[j]; };      /* (a terminator's newIndex is the shared one) */
  root = succ(0);
  for (uint32_t i : order) {
    const float *e = geometry + (size_t)i * 12;
    float *o = out.data() + (size_t)newIndex[i] * 12;
    if (e[10] == 1.0f) {
      for (int k = 0; k < 6; k++) o[k] = e[k];
      o[8] = bits(succ((uint64_t)i + 1));
      o[9] = bits(succ((uint64_t)i + 1 + (uint64_t)e[6]));
      o[10] = bits(1u | ((uint32_t)e[9] << 2));
    } else {
      for (int k = 0; k < 3; k++) { o[k] = e[k]; o[3 + k] = e[3 + k] - e[k]; o[6 + k] = e[6 + k] - e[k]; }      /* as build_threaded */
      o[9] = bits(succ((uint64_t)i + 1));
      o[10] = bits(2u | ((uint32_t)e[9] << 2));
    }
    o[11] = bits(i);
  }
}

extern "C" flx_status flx_scene_upload(flx_context *ctx, const float *geometry, const float *attributes, uint32_t n_entries_padded,
                                       const int32_t *ids, uint32_t n_ids) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!geometry || !attributes || n_entries_padded == 0) return fail(ctx, FLX_ERR_INVALID, "flx_scene_upload: empty scene");
  if (n_ids && !ids) return fail(ctx, FLX_ERR_INVALID, "flx_scene_upload: ids is NULL");
  if (n_entries_padded > LINK_INDEX) return fail(ctx, FLX_ERR_INVALID, "flx_scene_upload: more than 2^28 - 1 entries");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  ctx->sv_want_ver = false;                 /* (another scene: it has not moved yet) */
  /* Validate the skip list on the host: a skip that leaves the array would make the walk read out of
   * bounds on the GPU (the shader's texelFetch would be robust-access clamped; we refuse instead). */
  uint32_t max_transform = 0;
  for (uint32_t i = 0; i < n_entries_padded; i++) {
    const float *e = geometry + (size_t)i * 12;
    if (e[10] != 0.0f) {
      if (!(e[9] >= 0.0f && e[9] < 1048576.0f)) return fail(ctx, FLX_ERR_INVALID, "flx_scene_upload: transform number out of range");
      if ((uint32_t)e[9] > max_transform) max_transform = (uint32_t)e[9];
    }
    if (e[10] == 1.0f) {
      float skip = e[6];
      if (!(skip >= 0.0f) || (double)i + (double)skip >= (double)n_entries_padded)
        return fail(ctx, FLX_ERR_INVALID, "flx_scene_upload: AABB skip count leaves the entry array");
    } else if (e[10] != 0.0f && e[10] != 2.0f) {
      return fail(ctx, FLX_ERR_INVALID, "flx_scene_upload: entry type is not 0, 1 or 2");
    }
  }
  flx_status s;
  if ((s = shared_upload_begin(ctx))) return s;
  ctx->have_scene = false;
  ctx->geometry_version++;
  if ((s = upload(ctx, &ctx->d_geometry, geometry, (size_t)n_entries_padded * 48))) return s;
  if ((s = upload(ctx, &ctx->d_attributes, attributes, (size_t)n_entries_padded * 112))) return s;
  if ((s = upload(ctx, &ctx->d_ids, ids, (size_t)n_ids * 4))) return s;
  {
    std::vector<float> threaded;
    build_threaded(geometry, n_entries_padded, threaded, ctx->walk_entries, ctx->walk_hot, ctx->walk_root);
    if ((s = upload(ctx, &ctx->d_walk, threaded.data(), threaded.size() * sizeof(float)))) return s;
    /* precondition of the walk kernel's fast box test (flx_device.h: rayCuboidR): bounded, finite AABBs */
    bool bounded = true;
    for (uint32_t i = 0; i < n_entries_padded && bounded; i++) {
      const float *e = geometry + (size_t)i * 12;
      if (e[10] == 1.0f)
        for (int k = 0; k < 6; k++) if (!(std::fabs(e[k]) <= 5.764607523034235e17f)) bounded = false;
    }
    ctx->walk_fast_boxes = bounded ? 1u : 0u;
    /* the forward-ordered copy: the primary rays' walk steps through it wave by wave, and — small scenes in one object space — the
     * bounce walks of the per-pixel and persistent path kernels do */
    {
      std::vector<float> fwd;
      build_lockstep(geometry, n_entries_padded, fwd, ctx->fwd_entries, ctx->fwd_root, ctx->lock_boxes);
      if ((s = upload(ctx, &ctx->d_fwd, fwd.data(), fwd.size() * sizeof(float)))) return s;
      ctx->lock_ok = max_transform == 0 && ctx->fwd_entries <= FLX_LOCK_MAX;
    }
  }
  ctx->n_entries = n_entries_padded;
  ctx->n_ids = n_ids;
  ctx->max_transform = max_transform;
  if ((s = shared_upload_end(ctx))) return s;
  ctx->have_scene = true;
  return FLX_OK;
}

extern "C" flx_status flx_transforms_upload(flx_context *ctx, const float *rotation, const float *shift, uint32_t n_transforms) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!rotation || !shift || n_transforms == 0) return fail(ctx, FLX_ERR_INVALID, "flx_transforms_upload: need at least the identity transform");
  /* The reference refills its transform UBO and its light texture every frame (pathtracerWGL2.js:258-262, 361-365), changed or not.  An upload of what the
   * device holds already is nothing: no copy — and above all no end of a running frame server (upload() stops it: it reads the scene), whose frames in
   * flight would otherwise be completed one by one under a host that re-sends a static scene's arrays per frame. */
  if (!ctx->is_twin && ctx->have_transforms && ctx->n_transforms == n_transforms && ctx->d_rotation && ctx->d_shift &&
      ctx->h_rotation.size() == (size_t)n_transforms * 24 && ctx->h_shift.size() == (size_t)n_transforms * 8 &&
      memcmp(ctx->h_rotation.data(), rotation, (size_t)n_transforms * 96) == 0 && memcmp(ctx->h_shift.data(), shift, (size_t)n_transforms * 32) == 0) return FLX_OK;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_status s;
  /* the same transforms, moved: the scene moves (the frame server's next launch takes them per frame; one that does already goes on) */
  const bool moved = !ctx->is_twin && ctx->have_transforms && ctx->n_transforms == n_transforms && ctx->d_rotation && ctx->d_shift;
  if (moved && ctx->sv_moving) ctx->sv_want_ver = true;
  if (moved && ctx->sv_running && ctx->sv_ver) {
    /* the running launch takes the transforms with every frame (server_post: from the host's copy) and does not read the device's arrays: they follow when the
     * launch has ended (dyn_flush) — a copy on this context's stream now would wait for that end */
    ctx->transforms_version++; ctx->scene_version++; ctx->dyn_version++;
    ctx->h_rotation.assign(rotation, rotation + (size_t)n_transforms * 24);
    ctx->h_shift.assign(shift, shift + (size_t)n_transforms * 8);
    ctx->dyn_device_stale |= 1u;
    return FLX_OK;
  }
  ctx->have_transforms = false;             /* (until both arrays are in: a failed upload must not pass for the arrays it replaced) */
  ctx->transforms_version++;
  if ((s = upload(ctx, &ctx->d_rotation, rotation, (size_t)n_transforms * 96))) return s;
  if ((s = upload(ctx, &ctx->d_shift, shift, (size_t)n_transforms * 32))) return s;
  ctx->dyn_device_stale &= ~1u;
  ctx->n_transforms = n_transforms;
  ctx->have_transforms = true;
  if (!ctx->is_twin) {                      /* kept for the frame loop's second lane (flx_frame_begin) */
    ctx->h_rotation.assign(rotation, rotation + (size_t)n_transforms * 24);
    ctx->h_shift.assign(shift, shift + (size_t)n_transforms * 8);
    ctx->dyn_version++;
  }
  return FLX_OK;
}

extern "C" flx_status flx_lights_upload(flx_context *ctx, const float *lights, uint32_t n_lights) {
  if (!ctx) return FLX_ERR_INVALID;
  if (n_lights && !lights) return fail(ctx, FLX_ERR_INVALID, "flx_lights_upload: lights is NULL");
  if (!ctx->is_twin && ctx->have_lights && ctx->n_lights == n_lights && ctx->h_lights.size() == (size_t)n_lights * 6 &&
      (n_lights == 0 || memcmp(ctx->h_lights.data(), lights, (size_t)n_lights * 24) == 0)) return FLX_OK;      /* (as flx_transforms_upload: the same lights again) */
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_status s;
  const bool moved = !ctx->is_twin && ctx->have_lights && ctx->n_lights == n_lights && n_lights != 0 && ctx->d_lights;      /* (as flx_transforms_upload) */
  if (moved && ctx->sv_moving) ctx->sv_want_ver = true;
  if (moved && ctx->sv_running && ctx->sv_ver) {
    ctx->scene_version++; ctx->dyn_version++;
    ctx->h_lights.assign(lights, lights + (size_t)n_lights * 6);
    ctx->dyn_device_stale |= 2u;
    return FLX_OK;
  }
  ctx->have_lights = false;
  if ((s = upload(ctx, &ctx->d_lights, lights, (size_t)n_lights * 24))) return s;
  ctx->dyn_device_stale &= ~2u;
  ctx->n_lights = n_lights;
  if (!ctx->is_twin) { ctx->h_lights.assign(lights, lights + (size_t)n_lights * 6); ctx->dyn_version++; ctx->have_lights = true; }
  return FLX_OK;
}

extern "C" flx_status flx_atlas_upload(flx_context *ctx, int which, const uint8_t *rgba, uint32_t width, uint32_t height) {
  if (!ctx) return FLX_ERR_INVALID;
  if (which < 0 || which > 2) return fail(ctx, FLX_ERR_INVALID, "flx_atlas_upload: which must be 0, 1 or 2");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  size_t bytes = rgba ? (size_t)width * height * 4 : 0;
  flx_status s;
  if ((s = shared_upload_begin(ctx))) return s;
  if ((s = upload(ctx, &ctx->d_atlas[which], rgba, bytes))) return s;
  ctx->atlas_w[which] = bytes ? width : 0;
  ctx->atlas_h[which] = bytes ? height : 0;
  return shared_upload_end(ctx);
}

extern "C" flx_status flx_scene_upload_view(flx_context *ctx, const flx_scene_view *v) {
  if (!ctx || !v) return FLX_ERR_INVALID;
  flx_status s;
  if ((s = flx_scene_upload(ctx, v->geometry, v->attributes, v->n_entries_padded, v->ids, v->n_ids))) return s;
  if ((s = flx_transforms_upload(ctx, v->rotation, v->shift, v->n_transforms))) return s;
  if ((s = flx_lights_upload(ctx, v->lights, v->n_lights))) return s;
  for (int i = 0; i < 3; i++)
    if ((s = flx_atlas_upload(ctx, i, v->atlas[i], v->atlas_w[i], v->atlas_h[i]))) return s;
  return FLX_OK;
}

/* ---- tile policy ------------------------------------------------------------------------------------ */
static void tile_normalise(const flx_frame_params *p, uint32_t &tr, uint32_t &ti, uint32_t &tc) {
  tr = p->tile_rows; ti = p->tile_index; tc = p->tile_count;
  if (tr == 0 || tc <= 1) { tr = p->height ? p->height : 1; ti = 0; tc = 1; }
}
extern "C" uint32_t flx_tile_row_count(const flx_frame_params *p) {
  if (!p) return 0;
  uint32_t tr, ti, tc;
  tile_normalise(p, tr, ti, tc);
  if (ti >= tc) return 0;
  uint32_t strips = (p->height + tr - 1) / tr, n = 0;
  for (uint32_t s = ti; s < strips; s += tc) {
    uint32_t y0 = s * tr, y1 = y0 + tr;
    if (y1 > p->height) y1 = p->height;
    n += y1 - y0;
  }
  return n;
}
extern "C" uint32_t flx_tile_row_at(const flx_frame_params *p, uint32_t k) {
  uint32_t tr, ti, tc;
  tile_normalise(p, tr, ti, tc);
  uint32_t strip = k / tr;
  return (strip * tc + ti) * tr + (k - strip * tr);
}

static void fill_view(const flx_frame_params *p, FrameView &v) {
  memcpy(v.camera, p->camera, sizeof v.camera);
  flx_invert3x3(p->view_matrix, v.inv_view);
  v.view_row2[0] = p->view_matrix[6]; v.view_row2[1] = p->view_matrix[7]; v.view_row2[2] = p->view_matrix[8];
  memcpy(v.ambient, p->ambient, sizeof v.ambient);
  v.random_seed = p->random_seed;
}

flx_status flx_make_frame(flx_context *ctx, const flx_frame_params *p, DeviceScene &sc, DeviceFrame &fr) {
  if (!p) return fail(ctx, FLX_ERR_INVALID, "frame params are NULL");
  if (!ctx->have_scene || !ctx->have_transforms) return fail(ctx, FLX_ERR_NO_SCENE, "render before flx_scene_upload / flx_transforms_upload");
  if (p->width == 0 || p->height == 0 || p->samples < 1 || p->max_reflections < 0 || p->texture_width < 1)
    return fail(ctx, FLX_ERR_INVALID, "frame params: width/height/samples/texture_width must be positive");
  if (p->tile_count > 1 && p->tile_index >= p->tile_count) return fail(ctx, FLX_ERR_INVALID, "frame params: tile_index >= tile_count");
  sc.geometry = ctx->d_geometry; sc.attributes = ctx->d_attributes;
  sc.rotation = ctx->d_rotation; sc.shift = ctx->d_shift; sc.lights = ctx->d_lights;
  for (int i = 0; i < 3; i++) { sc.atlas[i] = ctx->d_atlas[i]; sc.atlas_w[i] = ctx->atlas_w[i]; sc.atlas_h[i] = ctx->atlas_h[i]; }
  sc.n_entries = ctx->n_entries; sc.n_lights = ctx->n_lights; sc.n_transforms = ctx->n_transforms;
  sc.walk = ctx->d_walk; sc.walk_entries = ctx->walk_entries; sc.walk_hot = ctx->walk_hot; sc.walk_root = ctx->walk_root; sc.walk_fast_boxes = ctx->walk_fast_boxes;
  sc.fwd = ctx->d_fwd; sc.fwd_entries = ctx->fwd_entries; sc.fwd_root = ctx->fwd_root;
  sc.lock = ctx->d_fwd; sc.lock_entries = (ctx->lock_ok && ctx->lock_use) ? ctx->fwd_entries : 0u; sc.lock_root = ctx->fwd_root;
  sc.angle_tan = nullptr;                  /* (the per-pixel kernel's table: flx_run_frame makes it where that kernel is launched) */
  uint32_t tr, ti, tc;
  tile_normalise(p, tr, ti, tc);
  fr.width = p->width; fr.height = p->height;
  fr.rows = flx_tile_row_count(p);
  fr.frame_rows = fr.rows; fr.frames = 1;
  fr.tile_rows = tr; fr.tile_index = ti; fr.tile_count = tc;
  memset(fr.view, 0, sizeof fr.view);
  fill_view(p, fr.view[0]);
  fr.samples = p->samples; fr.max_reflections = p->max_reflections;
  fr.samples_shift = -1;
  if ((p->samples & (p->samples - 1)) == 0) { fr.samples_shift = 0; while ((1 << fr.samples_shift) < p->samples) fr.samples_shift++; }
  fr.min_importancy = p->min_importancy;
  fr.use_filter = p->use_filter; fr.is_temporal = p->is_temporal;
  fr.texture_width = (float)p->texture_width;
  /* An entry names a transform; the shader indexes the UBO unchecked, we refuse an index past the upload. */
  if (ctx->max_transform >= ctx->n_transforms) return fail(ctx, FLX_ERR_INVALID, "scene names a transform that was not uploaded");
  return FLX_OK;
}


/* Did a frame kernel's watchdog trip (WavefrontBuffers::error, ChainArgs::error)?  Asked wherever the host has just waited for frames: the frame that was
 * being rendered — and whatever a chained kernel had worked ahead on — is incomplete.  Never reached by a healthy frame; tests force it (flx_debug_inject_fault). */
/* The frame server's launch ended by itself because the HOST did nothing for seconds (an application that paused with frames in flight) — and every frame it had been
 * given is complete in its image: nothing is lost, the next frame starts another launch.  (Anything else in the error word, or a frame that was posted and not
 * completed, is an error.) */
static bool server_idled_with_nothing_owed(const flx_context *ctx) {
  if (!ctx->h_dev_error || !ctx->h_sv_mail) return false;
  if (__atomic_load_n(&ctx->h_dev_error[0], __ATOMIC_ACQUIRE) != WF_ERR_SERVER_IDLE) return false;
  if (ctx->twin && ctx->twin->h_dev_error && __atomic_load_n(&ctx->twin->h_dev_error[0], __ATOMIC_ACQUIRE) != 0u) return false;
  for (uint32_t k = 0; k < SV_MAX_DEPTH; k++)
    if (__atomic_load_n(&ctx->h_sv_mail->posted[k], __ATOMIC_ACQUIRE) != __atomic_load_n(&ctx->h_sv_mail->done[k], __ATOMIC_ACQUIRE)) return false;
  return true;
}

flx_status flx_check_device_error(flx_context *ctx) {
  flx_context *owner = ctx;
  if (!owner->h_dev_error) return FLX_OK;
  uint32_t bits = __atomic_load_n(&owner->h_dev_error[0], __ATOMIC_ACQUIRE);
  if (ctx->twin && ctx->twin->h_dev_error) { bits |= __atomic_load_n(&ctx->twin->h_dev_error[0], __ATOMIC_ACQUIRE); }
  if (bits == 0u) return FLX_OK;
  if (server_idled_with_nothing_owed(ctx)) {
    ctx->sv_running = false;                                 /* (the frames still pending are complete: server_take hands them out) */
    if (ctx->sv_stream) (void)hipStreamSynchronize(ctx->sv_stream);
    ctx->h_dev_error[0] = 0u;
    return FLX_OK;
  }
  if (ctx->sv_stream) {                                      /* a frame server that is still up: it ends, whatever it holds */
    if (ctx->h_sv_mail) __atomic_store_n(&ctx->h_sv_mail->stopAfter, 1u, __ATOMIC_RELEASE);
    ctx->sv_running = false;
    for (auto &pf : ctx->sv_pending) pf.valid = false;
    (void)hipStreamSynchronize(ctx->sv_stream);
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->twin) { (void)hipStreamSynchronize(ctx->twin->stream); ctx->twin->h_dev_error[0] = 0u; }
  ctx->h_dev_error[0] = 0u;
  ctx->chain_seq = 0;
  /* the kernels' rings may hold ids nobody popped: back to "empty" for the next launch */
#if FLX_EXPERIMENTS
  if (ctx->d_chain_rings) (void)hipMemsetAsync(ctx->d_chain_rings, 0xff, (size_t)ctx->prop.multiProcessorCount * chain_rings_per_group() * sizeof(uint32_t), ctx->stream);
#endif
  if (ctx->d_sv_rings) (void)hipMemsetAsync(ctx->d_sv_rings, 0xff, (size_t)ctx->prop.multiProcessorCount * server_rings_per_group() * sizeof(uint32_t), ctx->stream);
  for (flx_context *c : { ctx, ctx->twin })
    if (c && c->d_frame_rings) (void)hipMemsetAsync(c->d_frame_rings, 0xff, (size_t)c->frame_rings_chains * c->prop.multiProcessorCount * WF_FRAME_RINGS * WF_FRAME_RING * sizeof(uint32_t), c->stream);
  char msg[360];
  snprintf(msg, sizeof msg, "device error in a frame kernel (bits 0x%x:%s%s%s%s%s%s%s): the frame is incomplete", bits, (bits & WF_ERR_SHADE_WATCHDOG) ? " shade-wave watchdog" : "",
           (bits & WF_ERR_WALK_WATCHDOG) ? " walk-wave watchdog" : "", (bits & WF_ERR_LIST) ? " resume list overflow" : "", (bits & WF_ERR_LEFTOVER) ? " paths left behind" : "",
           (bits & WF_ERR_RING_SLOT) ? " ring slot never filled" : "", (bits & WF_ERR_SERVER_IDLE) ? " frame server: nothing to do for seconds" : "",
           (bits & WF_ERR_SERVER_TIMEOUT) ? " frame server: no answer within 5 s" : "");
  std::string full = msg;
  if (ctx->h_sv_mail && (bits & (WF_ERR_SERVER_IDLE | WF_ERR_SERVER_TIMEOUT))) {
    char more[240];
    unsigned long long st[SV_STAT_WORDS] = {};
    if (ctx->d_sv_stats) (void)hipMemcpy(st, ctx->d_sv_stats, sizeof st, hipMemcpyDeviceToHost);
    snprintf(more, sizeof more, " [frame server: posted %u %u %u, done %u %u %u, stop after %u, next %u; launch completed %llu frames, %llu rotations, %llu tiles]", ctx->h_sv_mail->posted[0],
             ctx->h_sv_mail->posted[1], ctx->h_sv_mail->posted[2], ctx->h_sv_mail->done[0], ctx->h_sv_mail->done[1], ctx->h_sv_mail->done[2], ctx->h_sv_mail->stopAfter, ctx->sv_next_seq,
             st[SVS_FRAMES], st[SVS_ROTATIONS], st[SVS_TILES]);
    full += more;
  }
  return flx_fail(ctx, FLX_ERR_DEVICE, full.c_str());
}

flx_status flx_ensure_pixels(flx_context *ctx, float4 **buf, size_t *cap, size_t pixels) {
  if (*cap >= pixels && *buf) return FLX_OK;
  if (*buf) { FLX_HIP(ctx, hipFree(*buf)); *buf = nullptr; *cap = 0; }
  FLX_HIP(ctx, hipMalloc(buf, pixels * sizeof(float4)));
  *cap = pixels;
  return FLX_OK;
}

/* The buffers pipelines 2 and 3 need for this frame (or batch of frames): sized before anything is freed or allocated, so that a batch that is too
 * large is refused with a message that says what to do and the context keeps the buffers it has.  -> the chains of the bounce loop (pipeline 3). */
static flx_status ensure_workspace(flx_context *ctx, const DeviceFrame &fr, int pipeline, bool counted, int &wf_chains) {
  const size_t P = (size_t)fr.rows * fr.width;
  const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
  wf_chains = 1;
  if (pipeline != 1) {
    flx_status s;
    if (path_item_count64(fr) + 1000000ull >= 4294967296ull) return fail(ctx, FLX_ERR_INVALID, "frame (or batch of frames) too large: more than 2^32 path items");
    /* Does the workspace fit?  Asked before anything is freed or allocated, so that a batch that is too large is refused with
     * a message that says what to do, and the context keeps the buffers it has. */
    {
      const size_t items = (size_t)path_item_count64(fr);
      struct { size_t need, have; } w[] = {
        { P, ctx->hits_capacity }, { P, ctx->last_capacity }, { P * (size_t)fr.samples, ctx->samples_capacity },
        { pipeline == 3 ? items * 8 : 0, ctx->rec_capacity }, { pipeline == 3 ? items * 3 : 0, ctx->rec0_capacity },
        { pipeline == 3 ? items / (size_t)fr.samples * 3 : 0, ctx->pix0_capacity } };
      size_t grow = 0, freed = 0;
      for (auto &b : w) if (b.need > b.have) { grow += b.need * sizeof(float4); freed += b.have * sizeof(float4); }
      if (pipeline == 3) {
        const size_t live = wavefront_live_capacity(fr, cus) * (size_t)(ctx->wf_groups < 1 ? 1 : (ctx->wf_groups > WF_MAX_GROUPS ? WF_MAX_GROUPS : ctx->wf_groups));      /* what the allocation below asks for */
        if (live > ctx->live_capacity) { grow += 2 * live * sizeof(uint32_t); freed += 2 * ctx->live_capacity * sizeof(uint32_t); }
      }
      size_t memFree = 0, memTotal = 0;
      if (grow && hipMemGetInfo(&memFree, &memTotal) == hipSuccess && grow > memFree + freed) {
        char msg[256];
        snprintf(msg, sizeof msg, "the path records of this frame / batch need %.1f GB of device memory, %.1f GB are free (of %.1f GB): render fewer frames per batch or a smaller frame",
                 grow / 1e9, (memFree + freed) / 1e9, memTotal / 1e9);
        return fail(ctx, FLX_ERR_DEVICE, msg);
      }
    }
    if ((s = flx_ensure_pixels(ctx, &ctx->d_hits, &ctx->hits_capacity, P))) return s;
    if ((s = flx_ensure_pixels(ctx, &ctx->d_last, &ctx->last_capacity, P))) return s;
    if ((s = flx_ensure_pixels(ctx, &ctx->d_samples, &ctx->samples_capacity, P * (size_t)fr.samples))) return s;
  }
  if (pipeline == 3) {
    flx_status s;
    if ((s = flx_ensure_pixels(ctx, &ctx->d_rec, &ctx->rec_capacity, (size_t)path_item_count(fr) * 8))) return s;
    if ((s = flx_ensure_pixels(ctx, &ctx->d_rec0, &ctx->rec0_capacity, (size_t)path_item_count(fr) * 3))) return s;
    if ((s = flx_ensure_pixels(ctx, &ctx->d_pix0, &ctx->pix0_capacity, (size_t)path_item_count(fr) / (size_t)fr.samples * 3))) return s;      /* 64 per screen tile */
    /* the live lists are cut into one slice per chain of the bounce loop (flx_set_wavefront_groups; counted frames run one
     * chain), each able to hold the whole frame */
    wf_chains = ctx->wf_groups < 1 ? 1 : (ctx->wf_groups > WF_MAX_GROUPS ? WF_MAX_GROUPS : ctx->wf_groups);
    {
      const uint32_t tiles = path_item_count(fr) / ((uint32_t)fr.samples * 64u);
      if ((uint32_t)wf_chains > tiles || counted) wf_chains = 1;
    }
    const size_t need = wavefront_live_capacity(fr, cus) * (size_t)wf_chains;
    if (ctx->live_capacity < need) {
      ctx->live_capacity = 0;               /* a failed allocation below must not leave the old size standing over freed lists */
      for (int i = 0; i < 2; i++) {
        if (ctx->d_live[i]) { FLX_HIP(ctx, hipFree(ctx->d_live[i])); ctx->d_live[i] = nullptr; }
        FLX_HIP(ctx, hipMalloc(&ctx->d_live[i], need * sizeof(uint32_t)));
      }
      ctx->live_capacity = need;
    }
    /* suspended walks (flx_set_walk_scheduler): room for every possible walk workgroup of every chain, only while suspension is on */
    const size_t needStrag = ctx->walk_suspend ? (size_t)wf_chains * 2 * cus * 8u * ctx->walk_suspend * WF_STRAG_F4 : 0;
    if (ctx->strag_capacity < needStrag) {
      ctx->strag_capacity = 0;
      if (ctx->d_strag) { FLX_HIP(ctx, hipFree(ctx->d_strag)); ctx->d_strag = nullptr; }
      FLX_HIP(ctx, hipMalloc(&ctx->d_strag, needStrag * sizeof(float4)));
      ctx->strag_capacity = needStrag;
    }
  }
  return FLX_OK;
}

/* sc.angle_tan: the context's per-triangle table, made again first when the geometry, the attributes or this context's transforms were uploaded since it was made */
static flx_status angle_table(flx_context *ctx, DeviceScene &scT) {
  scT.angle_tan = nullptr;
  if (ctx->angle_table && ctx->n_entries != 0u) {
    const uint64_t key = ((uint64_t)ctx->geometry_version << 32) | ctx->transforms_version;
    if (key != ctx->angle_key || !ctx->d_angle_tan) {
      flx_status es = flx_ensure_pixels(ctx, &ctx->d_angle_tan, &ctx->angle_capacity, ctx->n_entries);
      if (es) return es;
      launch_angle_tan(scT, ctx->d_angle_tan, ctx->stream);
      FLX_HIP(ctx, hipGetLastError());
      ctx->angle_key = key;
    }
    scT.angle_tan = ctx->d_angle_tan;
  }
  return FLX_OK;
}

flx_status flx_run_frame(flx_context *ctx, const DeviceScene &sc, const DeviceFrame &fr, float4 *d_out, const GBufferPtrs &gb) {
  { flx_status ss = flx_server_stop(ctx); if (ss) return ss; }      /* (the frame server renders into the same workspace) */
  ctx->chain_seq = 0;                    /* this frame's kernels use the workspace a chain of frames keeps its state in: the chain ends here */
  unsigned long long *cnt = ctx->counters_enabled ? ctx->d_counters : nullptr;
  /* The G-buffer accumulators of the filter path carry state from sample to sample (fragment:83-89),
   * so filter frames use the sample-sequential kernel; everything else runs the wavefront pipeline. */
  int pipeline = ctx->pipeline;
  /* automatic: tiny scenes (a Cornell box, the theater: a few dozen entries) spend the wavefront pipeline's time on its 128-byte
   * path records, not on walks — the persistent path kernel, which keeps a path in registers from bounce to bounce, is faster
   * there (tools/pipeline_crossover.py: 48 entries 1.96 vs 2.41 ms, 329 entries 4.03 vs 2.85 ms) */
  if (pipeline == 0) {
    if (fr.use_filter || fr.is_temporal) pipeline = 1;
    else if (ctx->walk_entries <= 128u) {
      /* Small scenes (profiles/r02_ab_lockstep.txt, 6.): the per-pixel kernel — primary hit, surface and samples in one thread —
       * is the fastest while a pixel's thread is short (cornell 256 x 256 1 spp 1 bounce 0.057 ms against 0.144 wavefront and 0.220
       * persistent; cornell.obj 1080p 4 spp 3 bounces 0.65 against 1.09); the persistent path kernel, which refills the lanes of
       * dead paths, overtakes it at 32 bounce iterations per pixel (cornell.obj 8 x 6: 1.49 against 1.66 ms), with many lights
       * to shade per bounce from 4 on (theater 4 spp 3 bounces: 3.08 against 4.19 ms).  (A frame of fewer than 2^20 paths does
       * not fill the persistent grid.) */
      const uint64_t work = (uint64_t)fr.samples * (uint64_t)(fr.max_reflections > 0 ? fr.max_reflections : 1);
      const bool persistent = path_item_count64(fr) >= (1u << 20) && (work >= 32u || (ctx->n_lights >= 4u && work >= 4u));
      pipeline = persistent ? 2 : 1;
    } else pipeline = 3;
  }
  if (pipeline == 3 && fr.max_reflections > WF_MAX_BOUNCES) pipeline = 2;
  ctx->last_pipeline = pipeline;
  ctx->last_organisation = 0;
  if (pipeline != 1 && (fr.use_filter || fr.is_temporal)) return fail(ctx, FLX_ERR_INVALID, "pipelines 2 and 3 do not produce the G-buffers of filter / temporal frames");
  const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
  int wf_chains = 1;
  { flx_status s = ensure_workspace(ctx, fr, pipeline, cnt != nullptr, wf_chains); if (s) return s; }
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame0, ctx->stream));
  if (cnt) FLX_HIP(ctx, hipMemsetAsync(cnt, 0, FLX_COUNTER_SLOTS * sizeof(unsigned long long), ctx->stream));
  if (pipeline == 1) {
    /* the per-triangle table of the shading (DeviceScene::angle_tan; read by this kernel only): made again, on this context's stream and so in front of the
     * launch, when the geometry, the attributes or this context's transforms were uploaded since it was made */
    DeviceScene scT = sc;
    { flx_status es = angle_table(ctx, scT); if (es) return es; }
    FLX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    launch_trace_pixels(scT, fr, d_out, gb, cnt, ctx->stream, ctx->sample_parallel);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  } else if (pipeline == 2) {
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_queue, 0, sizeof(uint32_t), ctx->stream));
    launch_primary(sc, fr, ctx->d_hits, cnt, ctx->stream);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    /* persistent grid: enough workgroups to fill every CU at the kernel's occupancy; surplus ones find the queue dry */
    /* The lockstep walk pays where a wave's lanes stand at boxes and at triangles in the same trip of the lane walk.  A nearly flat
     * tree (the theater: 3 boxes over 20 triangles) has few such trips, and at seven waves per SIMD — which hide the lane walk's
     * fetches — the lane walk measured 2 % faster there (round 2: 10.70 vs 10.91 ms).  At the kernel's four waves per SIMD (late round 4) the
     * lockstep walk wins everywhere: theater 9.13 against 9.54 ms (profiles/r04_paths_occupancy.txt). */
    DeviceScene scPaths = sc;
    if (FLX_PATHS_ANGLE_TABLE) { flx_status es = angle_table(ctx, scPaths); if (es) return es; }
    if (ctx->lock_boxes < (uint32_t)FLX_PATHS_LOCK_MIN_BOXES) scPaths.lock_entries = 0u;
    launch_paths(scPaths, fr, ctx->d_hits, ctx->d_samples, ctx->d_last, ctx->d_queue, cus * 8u, cnt, ctx->stream);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    launch_resolve(fr, ctx->d_hits, ctx->d_samples, ctx->d_last, d_out, ctx->stream);
    FLX_HIP(ctx, hipGetLastError());
  } else {
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_wfcounts, 0, WF_MAX_GROUPS * 4 * (WF_MAX_ROUNDS + 2) * sizeof(uint32_t), ctx->stream));
    /* scratch of the walk kernel's tail consolidation and suspension: one slice per chain and possible walk workgroup */
    if (!ctx->d_tail_pool) FLX_HIP(ctx, hipMalloc(&ctx->d_tail_pool, (size_t)WF_MAX_GROUPS * cus * 8u * WF_TAIL_POOL_F4 * sizeof(float4)));
    /* the frame kernel's rings: one slice per chain that can run (48 MB each at 256 CUs), WF_INVALID everywhere — a launch leaves them so */
    if (ctx->frame_rings_chains < wf_chains) {
      if (ctx->d_frame_rings) { FLX_HIP(ctx, hipStreamSynchronize(ctx->stream)); FLX_HIP(ctx, hipFree(ctx->d_frame_rings)); ctx->d_frame_rings = nullptr; ctx->frame_rings_chains = 0; }
      const size_t words = (size_t)wf_chains * cus * WF_FRAME_RINGS * WF_FRAME_RING;
      FLX_HIP(ctx, hipMalloc(&ctx->d_frame_rings, words * sizeof(uint32_t)));
      FLX_HIP(ctx, hipMemsetAsync(ctx->d_frame_rings, 0xff, words * sizeof(uint32_t), ctx->stream));
      ctx->frame_rings_chains = wf_chains;
    }
    /* Who traces the primary rays and shades bounce 0 (flx_set_frame_front; one chain, all of the frame in it — several chains keep the two kernels):
     * the frame kernel itself from FLX_FRONT_MIN_TILES_PER_CU screen tiles per workgroup on (the fresh paths of a tile stay with the workgroup that made them,
     * and a workgroup with a dozen tiles, a rank's eighth of a 1080p frame, may have drawn the dragon or the sky: tools/front_time.py, profiles/r03_ab_front.txt);
     * below that ONE kernel in front (k_wf_front: a wave traces its tile and shades it straight away), which beats k_primary + k_wf_shade0 up to ~128 M paths a pass. */
    const int organisationNow = FLX_WF_ORGANISATION_DEFAULT ? FLX_WF_ORGANISATION_DEFAULT : ctx->wf_organisation;
    const uint32_t frameTiles = path_item_count(fr) / ((uint32_t)fr.samples * 64u);
    const int frontMode = wf_chains == 1 ? ctx->frame_front : 0;
    const bool front = (frontMode == 1 || frontMode == 2) && ctx->d_rec0 != nullptr &&
                       (frontMode == 2 || frameTiles >= (uint32_t)FLX_FRONT_MIN_TILES_PER_CU * cus) &&
                       wavefront_front_in_kernel(sc, fr, path_item_count(fr), ctx->walk_scheduler, ctx->walk_suspend, organisationNow);
    const bool fusedFront = !front && (frontMode == 2 || frontMode == 3 || (frontMode == 1 && path_item_count(fr) <= (uint32_t)FLX_FRONT_FUSED_MAX_ITEMS));
    if (!front && !fusedFront) launch_primary(sc, fr, ctx->d_hits, cnt, ctx->stream);
    FLX_HIP(ctx, hipGetLastError());
    /* The bounce loop runs as `groups` independent chains (contiguous ranges of screen tiles), group 0 on the
     * context's stream and the others on auxiliary streams: every persistent walk kernel ends in a tail set by
     * its longest walk (~0.3 ms), and with two chains that tail overlaps the other chain's next kernel. */
    const uint32_t total = path_item_count(fr);
    const uint32_t perTile = (uint32_t)fr.samples * 64u;
    const uint32_t tiles = total / perTile;
    const int groups = wf_chains;                      /* (counted frames: one chain, so the scheduler statistics describe whole kernels) */
    /* adaptive tile order: a single frame in one chain whose front runs in its own kernel, or inside the frame kernel where a workgroup gets fewer than
     * FLX_ADAPTIVE_FRONT_MAX_TILES_PER_CU screen tiles (a rank's quarter: 2.05 -> 1.78 ms heaviest-first; a half and a whole frame want another policy and gain 2 - 3 %:
     * profiles/r05_tile_order.txt) — k_resolve measures, k_tile_order sorts (below); the order is used, and the cost stamped, by the frame kernel only: `measured` */
    const bool adaptive = ctx->adaptive_order && (!front || tiles < (uint32_t)FLX_ADAPTIVE_FRONT_MAX_TILES_PER_CU * cus) && groups == 1 && fr.frames <= 1u && !ctx->d_tile_order && cnt == nullptr;
    const int orderMode = 1;                                  /* sixteen classes, heaviest first (k_tile_order) */
    if (adaptive && ctx->tile_time_cap < tiles) {
      FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      for (void *b : { (void *)ctx->d_tile_time, (void *)ctx->d_auto_order }) if (b) (void)hipFree(b);
      ctx->d_tile_time = nullptr; ctx->d_auto_order = nullptr; ctx->tile_time_cap = 0; ctx->auto_order_tiles = 0;
      FLX_HIP(ctx, hipMalloc(&ctx->d_tile_time, (size_t)tiles * sizeof(float)));
      FLX_HIP(ctx, hipMalloc(&ctx->d_auto_order, (size_t)tiles * sizeof(uint32_t)));
      ctx->tile_time_cap = tiles;
    }
    if (adaptive && !(ctx->auto_order_tiles == tiles && ctx->auto_order_width == fr.width && ctx->auto_order_rows == fr.rows && ctx->auto_order_mode == orderMode)) {
      ctx->auto_order_tiles = 0;                              /* another shape: this frame in screen order, measured from zero */
      FLX_HIP(ctx, hipMemsetAsync(ctx->d_tile_time, 0, (size_t)tiles * sizeof(float), ctx->stream));
    }
    const size_t listSlice = ctx->live_capacity / (size_t)groups;
    if (groups > 1) {
      FLX_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
      for (int g = 1; g < groups; g++) FLX_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream[g - 1], ctx->ev_fork, 0));
    }
    for (int g = 0; g < groups; g++) {
      const uint32_t t0 = (uint32_t)((uint64_t)tiles * g / groups), t1 = (uint32_t)((uint64_t)tiles * (g + 1) / groups);
      WavefrontBuffers wb;
      wb.rec = ctx->d_rec; wb.rec0 = ctx->d_rec0; wb.pix0 = ctx->d_pix0;
      wb.tailPool = ctx->d_tail_pool + (size_t)g * cus * 8u * WF_TAIL_POOL_F4;
      wb.frameRings = ctx->d_frame_rings + (size_t)g * cus * WF_FRAME_RINGS * WF_FRAME_RING;
      wb.front = front ? 1u : (fusedFront ? 2u : 0u);
      wb.error = ctx->d_dev_error; wb.watchdog = ctx->inject_watchdog; wb.inject = ctx->inject_flags; wb.walkJobs = ctx->walk_jobs | ((adaptive && front) ? WF_STAMP_COSTS : 0u);
      wb.tileOrder = (groups == 1 && ctx->d_tile_order && ctx->tile_order_n == tiles) ? ctx->d_tile_order : nullptr;
      if (!wb.tileOrder && adaptive && ctx->auto_order_tiles == tiles && ctx->auto_order_width == fr.width && ctx->auto_order_rows == fr.rows && ctx->auto_order_mode == orderMode)
        wb.tileOrder = ctx->d_auto_order;                     /* made by the last frame of this shape */
      wb.tileCost = (cnt && ctx->d_tile_cost && ctx->tile_cost_n >= tiles) ? ctx->d_tile_cost : nullptr;
      wb.tileCostPrimary = (wb.tileCost && ctx->tile_cost_n >= 2u * tiles) ? 1u : 0u;
      wb.live[0] = ctx->d_live[0] + listSlice * g; wb.live[1] = ctx->d_live[1] + listSlice * g;
      wb.counts = ctx->d_wfcounts + (size_t)g * 4 * (WF_MAX_ROUNDS + 2); wb.walkQueue = wb.counts + (WF_MAX_ROUNDS + 2); wb.stragCount = wb.walkQueue + (WF_MAX_ROUNDS + 2);
      wb.coopQueue = wb.stragCount + (WF_MAX_ROUNDS + 2);
      for (int k = 0; k < 2; k++) wb.strag[k] = ctx->d_strag ? ctx->d_strag + ((size_t)g * 2 + k) * cus * 8u * ctx->walk_suspend * WF_STRAG_F4 : nullptr;
      wb.item_base = t0 * perTile; wb.item_count = (t1 - t0) * perTile;
      wb.hits = ctx->d_hits; wb.sampleRadiance = ctx->d_samples; wb.lastOriginal = ctx->d_last; wb.counters = cnt;
      hipStream_t st = g == 0 ? ctx->stream : ctx->aux_stream[g - 1];
      /* Two lanes gathering over a communicator (flx_frame_begin_gathered): the persistent walk workgroups of one lane's frame hold every
       * CU until they end, and the other lane's RCCL kernel — a handful of workgroups that carry the finished frame's strips — would wait
       * behind them for a whole frame.  Such a context leaves a few CUs to the exchange. */
      const uint32_t cusWalk = (ctx->comm && (ctx->twin || ctx->is_twin) && cus > 4u * FLX_COMM_RESERVED_CUS) ? cus - FLX_COMM_RESERVED_CUS : cus;
      const int ran = launch_wavefront(sc, fr, wb, cusWalk, cnt != nullptr, ctx->walk_scheduler, ctx->walk_suspend, organisationNow,
                                       g == 0 ? ctx->ev_k0 : nullptr, g == 0 ? ctx->ev_k1 : nullptr, st);
      FLX_HIP(ctx, hipGetLastError());
      if (ran == -2) return fail(ctx, FLX_ERR_DEVICE, "the walk kernels need 156 KB of dynamic LDS and hipFuncSetAttribute refused it on this device");
      if (ran < 0) return fail(ctx, FLX_ERR_DEVICE, "internal: the frame kernel was to trace the primary rays but does not take this frame");
      ctx->last_organisation = ran;
      if (g > 0) {
        FLX_HIP(ctx, hipEventRecord(ctx->ev_join[g - 1], st));
        FLX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[g - 1], 0));
      }
    }
    const bool measured = adaptive && ctx->last_organisation >= 2;      /* (the frame kernel ran: its walk lanes stamp what a path cost) */
    launch_resolve(fr, ctx->d_hits, ctx->d_samples, ctx->d_last, d_out, ctx->stream, 0, measured ? ctx->d_tile_time : nullptr);
    FLX_HIP(ctx, hipGetLastError());
    if (measured) {
      launch_tile_order(ctx->d_tile_time, ctx->d_auto_order, tiles, orderMode, ctx->stream);
      FLX_HIP(ctx, hipGetLastError());
      ctx->auto_order_tiles = tiles; ctx->auto_order_width = fr.width; ctx->auto_order_rows = fr.rows; ctx->auto_order_mode = orderMode;
    } else if (adaptive) ctx->auto_order_tiles = 0;
  }
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
  ctx->timed = true;
  return FLX_OK;
}

/* Frames with a post pass (temporal accumulation and / or the denoise chain): trace with the per-pixel kernel
 * into float G-buffers, store them to RGBA8 planes like the reference's render targets, run the passes. */
static flx_status run_post_frame(flx_context *ctx, const DeviceScene &sc, const DeviceFrame &fr, const flx_frame_params *p, float4 *d_out) {
  if (fr.rows != fr.height)
    return fail(ctx, FLX_ERR_INVALID, "temporal / filter frames cannot be tiled: they read neighbouring pixels and history (render the whole frame on one context)");
  const size_t pixels = (size_t)fr.rows * fr.width;
  const bool temporal = fr.is_temporal == 1, filter = fr.use_filter == 1;
  flx_status s;
  if (ctx->gb_capacity < pixels) {
    ctx->gb_capacity = 0;                 /* a failed allocation below must not leave the old size standing */
    for (int i = 0; i < 6; i++) {
      size_t cap = 0;
      if (ctx->d_gb[i]) { FLX_HIP(ctx, hipFree(ctx->d_gb[i])); ctx->d_gb[i] = nullptr; }
      if ((s = flx_ensure_pixels(ctx, &ctx->d_gb[i], &cap, pixels))) return s;
    }
    ctx->gb_capacity = pixels;
  }
  if (ctx->planes_capacity < pixels) {
    ctx->planes_capacity = 0;                 /* a failed allocation below must not leave the old size standing */
    for (int i = 0; i < 13; i++) {
      if (ctx->d_planes[i]) { FLX_HIP(ctx, hipFree(ctx->d_planes[i])); ctx->d_planes[i] = nullptr; }
      FLX_HIP(ctx, hipMalloc(&ctx->d_planes[i], pixels * sizeof(uint32_t)));
    }
    ctx->planes_capacity = pixels;
  }
  FilterPlanes pl;
  for (int i = 0; i < 4; i++) { pl.R[i] = ctx->d_planes[i]; pl.Ip[i] = ctx->d_planes[4 + i]; }
  pl.O[0] = ctx->d_planes[8]; pl.O[1] = ctx->d_planes[9]; pl.Id[0] = ctx->d_planes[10]; pl.Id[1] = ctx->d_planes[11]; pl.OId = ctx->d_planes[12];
  int N = 1;
  if (temporal) {
    N = p->temporal_samples <= 0 ? 4 : (p->temporal_samples > 16 ? 16 : p->temporal_samples);
    if (ctx->ring_n != N || ctx->ring_w != fr.width || ctx->ring_h != fr.height) {      /* new size: fresh (zero) history, like a resize */
      ctx->ring_n = 0;                     /* (re)allocation in progress: a failure below leaves no size to match */
      for (auto &ring : ctx->d_ring) for (uint32_t *&plane : ring) if (plane) { FLX_HIP(ctx, hipFree(plane)); plane = nullptr; }
      for (int r = 0; r < 4; r++) for (int i = 0; i < N; i++) {
        FLX_HIP(ctx, hipMalloc(&ctx->d_ring[r][i], pixels * sizeof(uint32_t)));
        FLX_HIP(ctx, hipMemsetAsync(ctx->d_ring[r][i], 0, pixels * sizeof(uint32_t), ctx->stream));
      }
      ctx->ring_n = N; ctx->ring_head = 0; ctx->ring_w = fr.width; ctx->ring_h = fr.height;
    }
  }
  GBufferPtrs gb = { ctx->d_gb[0], ctx->d_gb[1], ctx->d_gb[2], ctx->d_gb[3], ctx->d_gb[4], ctx->d_gb[5] };
  if (filter && !temporal && !ctx->gb_float_wanted) {
    /* nobody reads the float G-buffers of this frame (flx_render was not given `gbuffers`): the trace kernel stores the five render
     * targets as RGBA8 itself — the values k_quantize would store — instead of 80 bytes per pixel written, read back and
     * quantised by five more launches */
    GBufferPtrs q = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    q.q_color = pl.R[0]; q.q_color_ip = pl.Ip[0]; q.q_original_color = pl.O[0]; q.q_id = pl.Id[0]; q.q_original_id = pl.OId;
    if ((s = flx_run_frame(ctx, sc, fr, nullptr, q))) return s;
    launch_filter_chain(pl, d_out, (int)fr.width, (int)fr.height, p->hdr, ctx->stream);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
    return FLX_OK;
  }
  if ((s = flx_run_frame(ctx, sc, fr, nullptr, gb))) return s;
  launch_quantize(gb.color, pl.R[0], pixels, ctx->stream);
  launch_quantize(gb.color_ip, pl.Ip[0], pixels, ctx->stream);
  if (temporal) {
    ctx->ring_head = (ctx->ring_head + N - 1) % N;                  /* TempTexture.unshift(TempTexture.pop()) */
    TemporalRings rings;
    rings.n = N;
    for (int k = 0; k < 16; k++) {
      const int slot = (ctx->ring_head + k) % N;
      rings.c[k] = k < N ? ctx->d_ring[0][slot] : nullptr; rings.ip[k] = k < N ? ctx->d_ring[1][slot] : nullptr;
      rings.id[k] = k < N ? ctx->d_ring[2][slot] : nullptr; rings.oid[k] = k < N ? ctx->d_ring[3][slot] : nullptr;
    }
    const int h0 = ctx->ring_head;
    FLX_HIP(ctx, hipMemcpyAsync(ctx->d_ring[0][h0], pl.R[0], pixels * 4, hipMemcpyDeviceToDevice, ctx->stream));
    FLX_HIP(ctx, hipMemcpyAsync(ctx->d_ring[1][h0], pl.Ip[0], pixels * 4, hipMemcpyDeviceToDevice, ctx->stream));
    launch_quantize(gb.location_id, ctx->d_ring[2][h0], pixels, ctx->stream);
    launch_quantize(gb.original_id, ctx->d_ring[3][h0], pixels, ctx->stream);
    launch_temporal(rings, (int)fr.width, (int)fr.height, p->hdr, filter ? 1 : 0, pl.R[0], pl.Ip[0], d_out, ctx->stream);
  }
  if (filter) {
    launch_quantize(gb.original_color, pl.O[0], pixels, ctx->stream);
    launch_quantize(gb.id, pl.Id[0], pixels, ctx->stream);
    launch_quantize(gb.original_id, pl.OId, pixels, ctx->stream);
    launch_filter_chain(pl, d_out, (int)fr.width, (int)fr.height, p->hdr, ctx->stream);
  }
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
  return FLX_OK;
}

extern "C" flx_status flx_temporal_reset(flx_context *ctx) {
  if (!ctx) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &ring : ctx->d_ring) for (uint32_t *&plane : ring) if (plane) { FLX_HIP(ctx, hipFree(plane)); plane = nullptr; }
  ctx->ring_n = 0; ctx->ring_head = 0; ctx->ring_w = ctx->ring_h = 0;
  return FLX_OK;
}

extern "C" flx_status flx_set_wavefront_groups(flx_context *ctx, int groups) {
  if (!ctx) return FLX_ERR_INVALID;
  if (groups < 1 || groups > WF_MAX_GROUPS) return fail(ctx, FLX_ERR_INVALID, "flx_set_wavefront_groups: 1..4");
  ctx->wf_groups = groups;
  return FLX_OK;
}

/* ---- anti-aliasing post passes ---- */
static flx_status aa_prepare(flx_context *ctx, uint32_t w, uint32_t h) {
  if (w == 0 || h == 0) return fail(ctx, FLX_ERR_INVALID, "anti-aliasing pass: empty frame");
  const size_t pixels = (size_t)w * h;
  if (ctx->aa_capacity < pixels) {
    ctx->aa_capacity = 0;
    for (auto &pl : ctx->d_aa) { if (pl) { FLX_HIP(ctx, hipFree(pl)); pl = nullptr; } FLX_HIP(ctx, hipMalloc(&pl, pixels * sizeof(uint32_t))); }
    ctx->aa_capacity = pixels;
    ctx->aa_w = 0;
  }
  if (ctx->aa_w != w || ctx->aa_h != h) {            /* new size: the ring starts from zero textures, like buildTexture() */
    ctx->aa_w = w; ctx->aa_h = h; ctx->taa_head = 0; ctx->taa_filled = 0;
  }
  return FLX_OK;
}

extern "C" flx_status flx_fxaa_device(flx_context *ctx, uint32_t width, uint32_t height, const void *d_in_rgba, void *d_out_rgba) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_in_rgba || !d_out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_fxaa_device: NULL pointer");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_status s = aa_prepare(ctx, width, height);
  if (s) return s;
  launch_quantize((const float4 *)d_in_rgba, ctx->d_aa[9], (size_t)width * height, ctx->stream);      /* the texture the renderer drew into */
  launch_fxaa(ctx->d_aa[9], (float4 *)d_out_rgba, (int)width, (int)height, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  return FLX_OK;
}

extern "C" flx_status flx_taa_device(flx_context *ctx, uint32_t width, uint32_t height, const void *d_in_rgba, void *d_out_rgba) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_in_rgba || !d_out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_taa_device: NULL pointer");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_status s = aa_prepare(ctx, width, height);
  if (s) return s;
  /* textures.unshift(textureIn); textureIn = textures.pop() (taa.js:100-102): the oldest plane takes the new frame */
  ctx->taa_head = (ctx->taa_head + 8) % 9;
  if (ctx->taa_filled < 9) ctx->taa_filled++;
  launch_quantize((const float4 *)d_in_rgba, ctx->d_aa[ctx->taa_head], (size_t)width * height, ctx->stream);
  const uint32_t *planes[9];
  for (int k = 0; k < 9; k++) planes[k] = k < ctx->taa_filled ? ctx->d_aa[(ctx->taa_head + k) % 9] : nullptr;
  launch_taa(planes, (float4 *)d_out_rgba, (int)width, (int)height, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  return FLX_OK;
}

extern "C" flx_status flx_taa_reset(flx_context *ctx) {
  if (!ctx) return FLX_ERR_INVALID;
  ctx->taa_head = 0; ctx->taa_filled = 0;
  return FLX_OK;
}

/* host-pointer variants: in / out are width * height * 4 floats */
static flx_status aa_host(flx_context *ctx, int which, uint32_t width, uint32_t height, const float *in_rgba, float *out_rgba) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!in_rgba || !out_rgba) return fail(ctx, FLX_ERR_INVALID, "anti-aliasing pass: NULL pointer");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pixels = (size_t)width * height;
  if (pixels == 0) return fail(ctx, FLX_ERR_INVALID, "anti-aliasing pass: empty frame");
  if (ctx->aa_io_capacity < pixels) {
    ctx->aa_io_capacity = 0;
    for (auto &b : ctx->d_aa_io) { if (b) { FLX_HIP(ctx, hipFree(b)); b = nullptr; } FLX_HIP(ctx, hipMalloc(&b, pixels * sizeof(float4))); }
    ctx->aa_io_capacity = pixels;
  }
  FLX_HIP(ctx, hipMemcpyAsync(ctx->d_aa_io[0], in_rgba, pixels * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  flx_status s = which == 0 ? flx_fxaa_device(ctx, width, height, ctx->d_aa_io[0], ctx->d_aa_io[1]) : flx_taa_device(ctx, width, height, ctx->d_aa_io[0], ctx->d_aa_io[1]);
  if (s) return s;
  FLX_HIP(ctx, hipMemcpyAsync(out_rgba, ctx->d_aa_io[1], pixels * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FLX_OK;
}
extern "C" flx_status flx_fxaa(flx_context *ctx, uint32_t width, uint32_t height, const float *in_rgba, float *out_rgba) { return aa_host(ctx, 0, width, height, in_rgba, out_rgba); }
extern "C" flx_status flx_taa(flx_context *ctx, uint32_t width, uint32_t height, const float *in_rgba, float *out_rgba) { return aa_host(ctx, 1, width, height, in_rgba, out_rgba); }

extern "C" flx_status flx_present_device(flx_context *ctx, uint32_t width, uint32_t height, const void *d_in_rgba, void *d_out_rgba8) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_in_rgba || !d_out_rgba8) return fail(ctx, FLX_ERR_INVALID, "flx_present_device: NULL pointer");
  if (width == 0 || height == 0) return fail(ctx, FLX_ERR_INVALID, "flx_present_device: empty frame");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  launch_quantize((const float4 *)d_in_rgba, (uint32_t *)d_out_rgba8, (size_t)width * height, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  return FLX_OK;
}
extern "C" flx_status flx_present(flx_context *ctx, uint32_t width, uint32_t height, const float *in_rgba, uint8_t *out_rgba8) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!in_rgba || !out_rgba8) return fail(ctx, FLX_ERR_INVALID, "flx_present: NULL pointer");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pixels = (size_t)width * height;
  if (pixels == 0) return fail(ctx, FLX_ERR_INVALID, "flx_present: empty frame");
  if (ctx->aa_io_capacity < pixels) {
    ctx->aa_io_capacity = 0;
    for (auto &b : ctx->d_aa_io) { if (b) { FLX_HIP(ctx, hipFree(b)); b = nullptr; } FLX_HIP(ctx, hipMalloc(&b, pixels * sizeof(float4))); }
    ctx->aa_io_capacity = pixels;
  }
  FLX_HIP(ctx, hipMemcpyAsync(ctx->d_aa_io[0], in_rgba, pixels * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
  flx_status s = flx_present_device(ctx, width, height, ctx->d_aa_io[0], ctx->d_aa_io[1]);
  if (s) return s;
  FLX_HIP(ctx, hipMemcpyAsync(out_rgba8, ctx->d_aa_io[1], pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FLX_OK;
}

extern "C" flx_status flx_set_walk_scheduler(flx_context *ctx, int scheduler, uint32_t suspend_walks) {
  if (!ctx) return FLX_ERR_INVALID;
  if (scheduler < FLX_WALK_LANES || scheduler > FLX_WALK_LANES_FINISHER) return fail(ctx, FLX_ERR_INVALID, "flx_set_walk_scheduler: scheduler 0 lanes, 1 queues, 2 lanes + finisher");
  if (suspend_walks > WF_STRAG_MAX) return fail(ctx, FLX_ERR_INVALID, "flx_set_walk_scheduler: suspend_walks 0..512");
  if (scheduler == FLX_WALK_QUEUES && suspend_walks != 0u) return fail(ctx, FLX_ERR_INVALID, "flx_set_walk_scheduler: the queue scheduler does not suspend walks");
#if !FLX_EXPERIMENTS
  if (scheduler != FLX_WALK_LANES || suspend_walks != 0u)
    return fail(ctx, FLX_ERR_INVALID, "flx_set_walk_scheduler: this library was built without the experimental walk schedulers (make EXPERIMENTS=1 builds libflexlight_hip_experiments.so)");
#endif
  ctx->walk_scheduler = scheduler;
  ctx->walk_suspend = suspend_walks;
  return FLX_OK;
}

extern "C" int flx_has_experiments(void) { return FLX_EXPERIMENTS; }

extern "C" flx_status flx_last_organisation(flx_context *ctx, int *organisation) {
  if (!ctx || !organisation) return FLX_ERR_INVALID;
  *organisation = ctx->last_organisation;
  return FLX_OK;
}

extern "C" flx_status flx_last_pipeline(flx_context *ctx, int *pipeline) {
  if (!ctx || !pipeline) return FLX_ERR_INVALID;
  *pipeline = ctx->last_pipeline;
  return FLX_OK;
}

extern "C" flx_status flx_set_pipeline(flx_context *ctx, int pipeline) {
  if (!ctx) return FLX_ERR_INVALID;
  if (pipeline < 0 || pipeline > 3) return fail(ctx, FLX_ERR_INVALID, "flx_set_pipeline: 0 auto, 1 per-pixel, 2 persistent paths, 3 wavefront");
  ctx->pipeline = pipeline;
  return FLX_OK;
}

extern "C" flx_status flx_set_wavefront_organisation(flx_context *ctx, int organisation) {
  if (!ctx) return FLX_ERR_INVALID;
  if (organisation < 0 || organisation > 2) return fail(ctx, FLX_ERR_INVALID, "flx_set_wavefront_organisation: 0 automatic, 1 rounds (a kernel pair per bounce), 2 frame kernel (one persistent launch)");
  ctx->wf_organisation = organisation;
  return FLX_OK;
}

extern "C" flx_status flx_set_frame_front(flx_context *ctx, int mode) {
  if (!ctx) return FLX_ERR_INVALID;
  if (mode < 0 || mode > 3) return fail(ctx, FLX_ERR_INVALID, "flx_set_frame_front: 0 k_primary + k_wf_shade0, 1 automatic, 2 inside the frame kernel wherever it runs, 3 one kernel in front");
  ctx->frame_front = mode;
  if (ctx->twin) ctx->twin->frame_front = ctx->frame_front;
  return FLX_OK;
}

extern "C" flx_status flx_set_lockstep(flx_context *ctx, int on) {
  if (!ctx) return FLX_ERR_INVALID;
  ctx->lock_use = on != 0;
  if (ctx->twin) ctx->twin->lock_use = ctx->lock_use;
  return FLX_OK;
}

/* A context whose tile policy gives it no strip (tile_count greater than the number of strips) has nothing to render: the
 * device entry points return FLX_OK like flx_render does, with the frame events recorded so that flx_last_frame_ms works. */
static flx_status empty_share(flx_context *ctx) {
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame0, ctx->stream));
  if (ctx->counters_enabled) FLX_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, FLX_COUNTER_SLOTS * sizeof(unsigned long long), ctx->stream));      /* no work: every counter 0 */
  FLX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
  FLX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
  ctx->timed = true;
  return FLX_OK;
}

extern "C" flx_status flx_render_device(flx_context *ctx, const flx_frame_params *params, void *d_out_rgba) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_render_device: output pointer is NULL");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_frame(ctx, params, sc, fr);
  if (s) return s;
  if ((size_t)fr.rows * fr.width == 0) return empty_share(ctx);
  if (params->use_filter || params->is_temporal) return run_post_frame(ctx, sc, fr, params, (float4 *)d_out_rgba);
  GBufferPtrs gb = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  return flx_run_frame(ctx, sc, fr, (float4 *)d_out_rgba, gb);
}

/* A batch of frames in ONE pass of the pipeline: the frames are stacked in the packed-row dimension, so every kernel of the
 * pass — primary, shade, walk, resolve — runs once over n times the paths.  A walk kernel lasts as long as its longest walk
 * (DESIGN.md §4); over a batch that tail is paid once per n frames. */
static_assert(FLX_MAX_BATCH == FLX_MAX_BATCH_FRAMES, "flx_device.h and flexlight_hip.h disagree on the batch limit");
flx_status flx_make_batch(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, DeviceScene &sc, DeviceFrame &fr) {
  if (!params || n_frames < 1 || n_frames > FLX_MAX_BATCH) return fail(ctx, FLX_ERR_INVALID, "flx_render_batch: 1 .. 32 frames per batch");
  flx_status s = flx_make_frame(ctx, params, sc, fr);
  if (s) return s;
  if (params->is_temporal)
    return fail(ctx, FLX_ERR_INVALID, "flx_render_batch: temporal frames depend on the frames before and cannot be batched");
  if (params->use_filter && flx_tile_row_count(params) != params->height)
    return fail(ctx, FLX_ERR_INVALID, "flx_render_batch: filter frames of a batch are whole frames (the chain reads neighbouring rows; strips go through flx_render_planes_device)");
  for (uint32_t i = 1; i < n_frames; i++) {
    const flx_frame_params &a = params[0], &b = params[i];
    if (a.width != b.width || a.height != b.height || a.samples != b.samples || a.max_reflections != b.max_reflections ||
        a.min_importancy != b.min_importancy || a.use_filter != b.use_filter || a.is_temporal != b.is_temporal || a.hdr != b.hdr ||
        a.texture_width != b.texture_width || a.tile_rows != b.tile_rows || a.tile_index != b.tile_index || a.tile_count != b.tile_count)
      return fail(ctx, FLX_ERR_INVALID, "flx_render_batch: the frames of a batch may differ in camera, view_matrix, ambient and random_seed only");
    fill_view(&b, fr.view[i]);
  }
  fr.frames = n_frames;
  fr.rows = fr.frame_rows * n_frames;
  if ((double)fr.rows * fr.width >= 4294967296.0) return fail(ctx, FLX_ERR_INVALID, "flx_render_batch: batch too large");
  return FLX_OK;
}

static flx_status ensure_post_buffers(flx_context *ctx, size_t pixels, bool gbuffers, bool planes);

/* A batch of filter frames: ONE trace pass over the stacked frames into the RGBA8 render targets (the per-pixel kernel: the G-buffer
 * accumulators carry state from sample to sample), then per frame the denoise chain — which starts from the reference's frame-0
 * texture state every time (launch_filter_chain), so the frames of a batch do not depend on each other.  Each frame equals its
 * own flx_render bit for bit. */
static flx_status run_filter_batch(flx_context *ctx, const DeviceScene &sc, const DeviceFrame &fr, const flx_frame_params *params, float4 *d_out) {
  const size_t per = (size_t)fr.frame_rows * fr.width, pixels = per * fr.frames;
  flx_status s;
  /* the five RGBA8 render targets of every frame of the batch: 20 bytes per pixel (not the float G-buffers' 96), refused with
   * advice when they do not fit */
  if (ctx->qbatch_capacity < pixels) {
    const size_t need = 5 * pixels * sizeof(uint32_t);
    size_t memFree = 0, memTotal = 0;
    if (hipMemGetInfo(&memFree, &memTotal) == hipSuccess && need > memFree + 5 * ctx->qbatch_capacity * sizeof(uint32_t)) {
      char msg[200];
      snprintf(msg, sizeof msg, "the render targets of this batch of filter frames need %.1f GB of device memory, %.1f GB are free: render fewer frames per batch", need / 1e9, memFree / 1e9);
      return fail(ctx, FLX_ERR_DEVICE, msg);
    }
    ctx->qbatch_capacity = 0;
    if (ctx->d_qbatch) { FLX_HIP(ctx, hipFree(ctx->d_qbatch)); ctx->d_qbatch = nullptr; }
    FLX_HIP(ctx, hipMalloc(&ctx->d_qbatch, need));
    ctx->qbatch_capacity = pixels;
  }
  if ((s = ensure_post_buffers(ctx, per, false, true))) return s;
  FilterPlanes pl;
  for (int i = 0; i < 4; i++) { pl.R[i] = ctx->d_planes[i]; pl.Ip[i] = ctx->d_planes[4 + i]; }
  pl.O[0] = ctx->d_planes[8]; pl.O[1] = ctx->d_planes[9]; pl.Id[0] = ctx->d_planes[10]; pl.Id[1] = ctx->d_planes[11]; pl.OId = ctx->d_planes[12];
  /* the trace kernel stores the five render targets of every frame as RGBA8 itself (run_post_frame), stacked frame after frame
   * in planes of their own; each frame's chain starts from its own slices */
  GBufferPtrs q = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  q.q_color = ctx->d_qbatch; q.q_color_ip = ctx->d_qbatch + pixels; q.q_original_color = ctx->d_qbatch + 2 * pixels;
  q.q_id = ctx->d_qbatch + 3 * pixels; q.q_original_id = ctx->d_qbatch + 4 * pixels;
  if ((s = flx_run_frame(ctx, sc, fr, nullptr, q))) return s;
  for (uint32_t f = 0; f < fr.frames; f++) {
    const size_t o = (size_t)f * per;
    FilterPlanes plf = pl;
    plf.R[0] = q.q_color + o; plf.Ip[0] = q.q_color_ip + o; plf.O[0] = q.q_original_color + o; plf.Id[0] = q.q_id + o; plf.OId = q.q_original_id + o;
    launch_filter_chain(plf, d_out + o, (int)fr.width, (int)fr.height, params->hdr, ctx->stream);
  }
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
  return FLX_OK;
}

extern "C" flx_status flx_render_batch_device(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, void *d_out_rgba) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_render_batch_device: output pointer is NULL");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_batch(ctx, params, n_frames, sc, fr);
  if (s) return s;
  if ((size_t)fr.rows * fr.width == 0) return empty_share(ctx);
  if (params->use_filter) return run_filter_batch(ctx, sc, fr, params, (float4 *)d_out_rgba);
  GBufferPtrs gb = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  return flx_run_frame(ctx, sc, fr, (float4 *)d_out_rgba, gb);
}

extern "C" flx_status flx_render_batch(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, float *out_rgba, flx_counters *counters) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_render_batch: out_rgba is NULL");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_batch(ctx, params, n_frames, sc, fr);
  if (s) return s;
  const size_t pixels = (size_t)fr.rows * fr.width;
  if (pixels == 0) return FLX_OK;
  if ((s = flx_ensure_pixels(ctx, &ctx->d_out, &ctx->out_capacity, pixels))) return s;
  const bool saved = ctx->counters_enabled;
  if (counters) ctx->counters_enabled = true;
  if (params->use_filter) {
    s = run_filter_batch(ctx, sc, fr, params, ctx->d_out);
  } else {
    GBufferPtrs gb = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    s = flx_run_frame(ctx, sc, fr, ctx->d_out, gb);
  }
  ctx->counters_enabled = saved;
  if (s) return s;
  FLX_HIP(ctx, hipMemcpyAsync(out_rgba, ctx->d_out, pixels * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  unsigned long long host_cnt[8];
  if (counters) FLX_HIP(ctx, hipMemcpyAsync(host_cnt, ctx->d_counters, sizeof host_cnt, hipMemcpyDeviceToHost, ctx->stream));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (counters) {
    memcpy(counters, host_cnt, sizeof host_cnt);
    ctx->last_counters = *counters;
  }
  return flx_check_device_error(ctx);
}

/* ---- filter frames across GPUs: the trace is per pixel and shards by row strips, the denoise chain is not ---- */
static flx_status ensure_post_buffers(flx_context *ctx, size_t pixels, bool gbuffers, bool planes) {
  flx_status s;
  if (gbuffers && ctx->gb_capacity < pixels) {
    ctx->gb_capacity = 0;                 /* a failed allocation below must not leave the old size standing */
    for (int i = 0; i < 6; i++) {
      size_t cap = 0;
      if (ctx->d_gb[i]) { FLX_HIP(ctx, hipFree(ctx->d_gb[i])); ctx->d_gb[i] = nullptr; }
      if ((s = flx_ensure_pixels(ctx, &ctx->d_gb[i], &cap, pixels))) return s;
    }
    ctx->gb_capacity = pixels;
  }
  if (planes && ctx->planes_capacity < pixels) {
    ctx->planes_capacity = 0;                 /* a failed allocation below must not leave the old size standing */
    for (int i = 0; i < 13; i++) {
      if (ctx->d_planes[i]) { FLX_HIP(ctx, hipFree(ctx->d_planes[i])); ctx->d_planes[i] = nullptr; }
      FLX_HIP(ctx, hipMalloc(&ctx->d_planes[i], pixels * sizeof(uint32_t)));
    }
    ctx->planes_capacity = pixels;
  }
  return FLX_OK;
}

extern "C" flx_status flx_render_planes_device(flx_context *ctx, const flx_frame_params *params, void *d_planes) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_planes) return fail(ctx, FLX_ERR_INVALID, "flx_render_planes_device: output pointer is NULL");
  if (!params || params->use_filter != 1 || params->is_temporal != 0)
    return fail(ctx, FLX_ERR_INVALID, "flx_render_planes_device: needs use_filter = 1 and is_temporal = 0 (history is per context)");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_frame(ctx, params, sc, fr);
  if (s) return s;
  const size_t pixels = (size_t)fr.rows * fr.width;
  if (pixels == 0) return empty_share(ctx);
  /* the reference's five render targets, stored as it stores them (RGBA8) by the trace kernel itself, strips packed like the
   * radiance of a tiled frame */
  uint32_t *out = (uint32_t *)d_planes;
  GBufferPtrs q = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  q.q_color = out; q.q_color_ip = out + pixels; q.q_original_color = out + 2 * pixels; q.q_id = out + 3 * pixels; q.q_original_id = out + 4 * pixels;
  if ((s = flx_run_frame(ctx, sc, fr, nullptr, q))) return s;
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
  return FLX_OK;
}

extern "C" flx_status flx_filter_planes_device(flx_context *ctx, const flx_frame_params *params, const void *d_planes, void *d_out_rgba) {
  return flx_filter_planes_enqueue(ctx, params, d_planes, d_out_rgba, true);
}

flx_status flx_filter_planes_enqueue(flx_context *ctx, const flx_frame_params *params, const void *d_planes, void *d_out_rgba, bool stamp_start) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!d_planes || !d_out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_filter_planes_device: NULL pointer");
  if (!params || params->width == 0 || params->height == 0) return fail(ctx, FLX_ERR_INVALID, "flx_filter_planes_device: empty frame");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  const size_t pixels = (size_t)params->width * params->height;
  flx_status s;
  if ((s = ensure_post_buffers(ctx, pixels, false, true))) return s;
  FilterPlanes pl;
  for (int i = 0; i < 4; i++) { pl.R[i] = ctx->d_planes[i]; pl.Ip[i] = ctx->d_planes[4 + i]; }
  pl.O[0] = ctx->d_planes[8]; pl.O[1] = ctx->d_planes[9]; pl.Id[0] = ctx->d_planes[10]; pl.Id[1] = ctx->d_planes[11]; pl.OId = ctx->d_planes[12];
  const uint32_t *in = (const uint32_t *)d_planes;
  uint32_t *dst[5] = { pl.R[0], pl.Ip[0], pl.O[0], pl.Id[0], pl.OId };
  if (stamp_start) FLX_HIP(ctx, hipEventRecord(ctx->ev_frame0, ctx->stream));
  for (int k = 0; k < 5; k++) FLX_HIP(ctx, hipMemcpyAsync(dst[k], in + (size_t)k * pixels, pixels * 4, hipMemcpyDeviceToDevice, ctx->stream));
  launch_filter_chain(pl, (float4 *)d_out_rgba, (int)params->width, (int)params->height, params->hdr, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
  ctx->timed = true;
  return FLX_OK;
}

extern "C" flx_status flx_render(flx_context *ctx, const flx_frame_params *params, float *out_rgba, const flx_gbuffers *gbuffers,
                                 flx_counters *counters) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!out_rgba) return fail(ctx, FLX_ERR_INVALID, "flx_render: out_rgba is NULL");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_frame(ctx, params, sc, fr);
  if (s) return s;
  if (gbuffers && !params->use_filter && !params->is_temporal) return fail(ctx, FLX_ERR_INVALID, "flx_render: G-buffers are only produced with use_filter = 1 or is_temporal = 1");
  const size_t pixels = (size_t)fr.rows * fr.width;
  if (pixels == 0) return FLX_OK;
  if ((s = flx_ensure_pixels(ctx, &ctx->d_out, &ctx->out_capacity, pixels))) return s;
  const bool saved = ctx->counters_enabled;
  if (counters) ctx->counters_enabled = true;
  if (params->use_filter || params->is_temporal) {
    ctx->gb_float_wanted = gbuffers != nullptr;
    s = run_post_frame(ctx, sc, fr, params, ctx->d_out);
    ctx->gb_float_wanted = false;
  } else {
    GBufferPtrs gb = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    s = flx_run_frame(ctx, sc, fr, ctx->d_out, gb);
  }
  ctx->counters_enabled = saved;
  if (s) return s;
  FLX_HIP(ctx, hipMemcpyAsync(out_rgba, ctx->d_out, pixels * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  if (gbuffers) {
    float *dst[6] = { gbuffers->color, gbuffers->color_ip, gbuffers->original_color, gbuffers->id, gbuffers->original_id,
                      gbuffers->location_id };
    for (int i = 0; i < 6; i++)
      if (dst[i]) FLX_HIP(ctx, hipMemcpyAsync(dst[i], ctx->d_gb[i], pixels * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  }
  unsigned long long host_cnt[8];
  if (counters) FLX_HIP(ctx, hipMemcpyAsync(host_cnt, ctx->d_counters, sizeof host_cnt, hipMemcpyDeviceToHost, ctx->stream));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (counters) {
    memcpy(counters, host_cnt, sizeof host_cnt);
    ctx->last_counters = *counters;
  }
  return flx_check_device_error(ctx);
}

/* ---- the frame loop: begin / end with two frames in flight (include/flexlight_hip.h) -------------------------------------
 * Two frames in flight overlap on the GPU when they run on two streams with a workspace each: the kernels of a frame are a
 * dependent chain and every one of them ends in a tail that leaves most CUs idle; frame k + 1's kernels fill those
 * CUs.  So the context keeps a TWIN — a second context on the same device with its own stream and workspace that shares the
 * static scene arrays — and frames alternate between the two lanes.  What changes per frame (lights, transforms) is kept as a
 * host copy and re-sent to the twin's own small buffers on its stream, so neither lane ever overwrites what the other is
 * reading.  Temporal frames keep their history in one context and stay on the primary lane. */
static void mirror_scene(flx_context *ctx) {
  flx_context *t = ctx->twin;
  if (!t) return;
  t->d_geometry = ctx->d_geometry; t->d_attributes = ctx->d_attributes; t->d_ids = ctx->d_ids; t->d_walk = ctx->d_walk;
  t->d_fwd = ctx->d_fwd; t->fwd_entries = ctx->fwd_entries; t->fwd_root = ctx->fwd_root; t->lock_ok = ctx->lock_ok; t->lock_use = ctx->lock_use; t->lock_boxes = ctx->lock_boxes;
  t->walk_entries = ctx->walk_entries; t->walk_hot = ctx->walk_hot; t->walk_root = ctx->walk_root; t->walk_fast_boxes = ctx->walk_fast_boxes;
  for (int i = 0; i < 3; i++) { t->d_atlas[i] = ctx->d_atlas[i]; t->atlas_w[i] = ctx->atlas_w[i]; t->atlas_h[i] = ctx->atlas_h[i]; }
  t->n_entries = ctx->n_entries; t->n_ids = ctx->n_ids; t->max_transform = ctx->max_transform; t->have_scene = ctx->have_scene;
  t->geometry_version = ctx->geometry_version; t->angle_table = ctx->angle_table;      /* (the lane's own angle table follows: flx_make_frame) */
}


/* ---- the chained frame loop (flx_chain.hip) -------------------------------------------------------------------------------
 * Two frames in flight share ONE stacked workspace (the layout of a batch of two) and one stream: the kernel of frame k completes the slot of frame k and
 * works ahead on the slot of frame k + 1 — whose view flx_frame_begin of that frame posts while the kernel runs — and hands what it holds of it to the
 * kernel of frame k + 1 when frame k is complete.  The drain of a launch, a third of a rank's share of a 1080p frame, disappears under the next frame's bulk. */
#ifndef FLX_SERVER_MAX_TILES_PER_CU
#define FLX_SERVER_MAX_TILES_PER_CU 64
#endif
#ifndef FLX_CHAIN_MIN_LANES
#define FLX_CHAIN_MIN_LANES 2
#endif
static bool chain_same_shape(const flx_frame_params &a, const flx_frame_params &b) {
  return a.width == b.width && a.height == b.height && a.samples == b.samples && a.max_reflections == b.max_reflections && a.min_importancy == b.min_importancy &&
         a.use_filter == b.use_filter && a.is_temporal == b.is_temporal && a.hdr == b.hdr && a.texture_width == b.texture_width &&
         a.tile_rows == b.tile_rows && a.tile_index == b.tile_index && a.tile_count == b.tile_count;
}
/* May this frame of the loop run chained?  (The frame kernel with the front of the frame inside it must take it, uncounted; tiles must not straddle the slots.) */
static bool chain_wanted(flx_context *ctx, const flx_frame_params *p, const DeviceScene &sc, const DeviceFrame &fr) {
  if (!ctx->frame_chain || ctx->frame_lanes < FLX_CHAIN_MIN_LANES || ctx->is_twin) return false;
  if (p->use_filter || p->is_temporal || ctx->counters_enabled) return false;
  if (!(ctx->pipeline == 3 || (ctx->pipeline == 0 && ctx->walk_entries > 128u))) return false;
  if (fr.max_reflections < 1 || fr.max_reflections > WF_MAX_BOUNCES) return false;
  const int organisation = FLX_WF_ORGANISATION_DEFAULT ? FLX_WF_ORGANISATION_DEFAULT : ctx->wf_organisation;
  if (organisation == 1 || ctx->walk_scheduler != 0 || ctx->walk_suspend != 0u || ctx->wf_groups > 1 || !(ctx->frame_front == 1 || ctx->frame_front == 2)) return false;
  if (fr.frame_rows == 0u || (fr.frame_rows & 7u) != 0u) return false;
  if (path_item_count64(fr) * (uint64_t)(ctx->frame_lanes == 3 ? 3 : 2) >= (1ull << 31)) return false;
  if ((uint32_t)fr.samples * 64u * 2u + 2u * (uint32_t)FLX_CHAIN_RESERVE > (uint32_t)WF_FRAME_RING - 256u) return false;
  uint32_t a = 0, b = 0;
#if FLX_EXPERIMENTS
  if (ctx->frame_chain == 1) return chain_kernel_fits(sc, a, b);
#endif
  return server_kernel_fits(sc, a, b, SV_MAX_DEPTH);          /* (with room for a moving scene's transforms per slot) */
}

#if FLX_EXPERIMENTS      /* the chain of launches (mode 1) was measured and lost to the frame server: only `make EXPERIMENTS=1` carries it (Makefile) */
static flx_status chain_resources(flx_context *ctx, size_t itemsPerSlot) {
  const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
  if (!ctx->d_chain_slots) {
    FLX_HIP(ctx, hipMalloc(&ctx->d_chain_slots, CH_MAX_DEPTH * sizeof(ChainSlot)));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_chain_slots, 0, CH_MAX_DEPTH * sizeof(ChainSlot), ctx->stream));
    /* the mailbox: pinned host memory the host writes with plain stores while the kernel runs and the kernel reads with system-scope loads
     * (tools/micro/mailbox.hip).  Not a copy on a stream: a small hipMemcpyAsync is a kernel of its own, and the persistent launch leaves it no CU. */
    FLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_chain_mail, sizeof(ChainMail), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_chain_mail, 0, sizeof(ChainMail));
    FLX_HIP(ctx, hipHostGetDevicePointer((void **)&ctx->d_chain_mail, ctx->h_chain_mail, 0));
    FLX_HIP(ctx, hipMalloc(&ctx->d_chain_relay, sizeof(ChainMail)));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_chain_relay, 0, sizeof(ChainMail), ctx->stream));
    const size_t ringWords = (size_t)cus * chain_rings_per_group();
    FLX_HIP(ctx, hipMalloc(&ctx->d_chain_rings, ringWords * sizeof(uint32_t)));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_chain_rings, 0xff, ringWords * sizeof(uint32_t), ctx->stream));      /* WF_INVALID everywhere; a kernel leaves them so */
    ctx->chain_susp_cap = (size_t)cus * 1024u;                                  /* every lane of every workgroup may hold a walk when the launch stops */
    FLX_HIP(ctx, hipMalloc(&ctx->d_chain_susp, (size_t)CH_MAX_DEPTH * 2 * ctx->chain_susp_cap * CH_SUSP_F4 * sizeof(float4)));
  }
  /* a resume list holds what the workgroups had in their rings and lanes when a launch stopped: never more than the slot's paths, nor than the rings take */
  size_t cap = (size_t)cus * WF_FRAME_RING;
  if (cap > itemsPerSlot) cap = itemsPerSlot;
  if (ctx->chain_list_cap < cap) {
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->chain_list_cap = 0; ctx->chain_seq = 0;
    if (ctx->d_chain_lists) { FLX_HIP(ctx, hipFree(ctx->d_chain_lists)); ctx->d_chain_lists = nullptr; }
    FLX_HIP(ctx, hipMalloc(&ctx->d_chain_lists, (size_t)CH_MAX_DEPTH * 2 * 3 * cap * sizeof(uint32_t)));
    ctx->chain_list_cap = cap;
  }
  return FLX_OK;
}

/* One frame of the loop, chained: its view posted to the kernels before it (when it continues a chain), the slot it will leave behind reset for the frame
 * `depth` later, its own kernel and the resolve of its slot — everything on the context's stream but the post. */
static flx_status chain_run_frame(flx_context *ctx, const flx_frame_params *params, const DeviceScene &sc, const DeviceFrame &frOne, float4 *d_out) {
  { flx_status ss = flx_server_stop(ctx); if (ss) return ss; }
  const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
  const uint32_t depth = ctx->frame_lanes == 3 ? 3u : 2u;
  DeviceFrame fr = frOne;                                   /* the slots stacked like a batch of frames */
  fr.frames = depth; fr.rows = depth * frOne.frame_rows;
  const size_t itemsPerSlot = (size_t)path_item_count64(frOne);
  flx_status s;
  if ((s = chain_resources(ctx, itemsPerSlot))) return s;
  int chains = 1;
  if ((s = ensure_workspace(ctx, fr, 3, false, chains))) return s;
  const bool continuing = ctx->chain_seq != 0 && ctx->chain_depth == depth && chain_same_shape(ctx->chain_params, *params) && ctx->chain_scene_version == ctx->scene_version;
  /* sequence numbers: consecutive within a chain (the kernel of frame q expects the posts q + 1, q + 2), never reused, never 0 */
  if (!continuing) ctx->chain_counter += 8u;
  uint32_t seq = ++ctx->chain_counter;
  if (seq < 16u) { ctx->chain_counter = 16u; seq = 16u; }
  const uint32_t slotP = continuing ? (ctx->chain_slot + 1u) % depth : 0u;
  for (uint32_t i = 0; i < depth; i++) memset(&fr.view[i], 0, sizeof(FrameView));
  fr.view[slotP] = frOne.view[0];
  if (continuing) {
    /* post: the view, then the number that says whose view it is (a kernel takes the view of exactly the frame it was told to expect, so nothing has to be
     * reset).  The kernels before this one — the one running and, at depth 3, the one queued behind it — may work ahead on this frame from now on. */
    memcpy((void *)&ctx->h_chain_mail->view[slotP], &frOne.view[0], sizeof(FrameView));
    __atomic_store_n(&ctx->h_chain_mail->posted[slotP], seq, __ATOMIC_RELEASE);
  }
  /* The slot of the frame before this one is free for the frame `depth - 1` after this one once that frame's kernel and resolve — earlier in this stream
   * — are through: its cursors back to zero; and of every slot the list set this kernel writes (a frame that begins a chain resets everything). */
  const uint32_t recycled = (slotP + depth - 1u) % depth;
  launch_chain_reset(ctx->d_chain_slots, depth, continuing ? 1u << recycled : 7u, seq & 1u, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  WavefrontBuffers wb = {};
  wb.rec = ctx->d_rec; wb.rec0 = ctx->d_rec0; wb.pix0 = ctx->d_pix0;
  wb.frameRings = ctx->d_chain_rings; wb.front = 1u;
  wb.error = ctx->d_dev_error; wb.watchdog = 0u; wb.inject = 0u;
  wb.item_base = 0u; wb.item_count = (uint32_t)(depth * itemsPerSlot);
  wb.hits = ctx->d_hits; wb.sampleRadiance = ctx->d_samples; wb.lastOriginal = ctx->d_last; wb.counters = nullptr;
  ChainArgs ca = {};
  ca.slots = ctx->d_chain_slots; ca.mail = ctx->d_chain_mail; ca.relay = ctx->d_chain_relay;
  ca.lists = ctx->d_chain_lists; ca.listCap = (uint32_t)ctx->chain_list_cap;
  ca.susp = ctx->d_chain_susp; ca.suspCap = (uint32_t)ctx->chain_susp_cap;
  ca.depth = depth; ca.ahead = depth - 1u;
  ca.slotP = slotP;
  ca.seqP = seq;
  ca.tilesPerSlot = (uint32_t)(itemsPerSlot / ((size_t)fr.samples * 64u));
  ca.itemsPerSlot = (uint32_t)itemsPerSlot;
  for (uint32_t i = 0; i < depth; i++) {
    ca.order[i] = (ctx->d_chain_order && ctx->chain_order_n == ca.tilesPerSlot) ? ctx->d_chain_order : nullptr;
    ca.cost[i] = (ctx->d_chain_cost && ctx->chain_cost_n == ca.tilesPerSlot) ? ctx->d_chain_cost + (size_t)i * ctx->chain_cost_n : nullptr;
  }
  ca.error = ctx->d_dev_error;
  ca.stats = nullptr;
  if (ctx->d_chain_stats) {
    ca.stats = ctx->d_chain_stats + (size_t)(seq % CH_STAT_LAUNCHES) * CH_STAT_WORDS;
    unsigned long long init[CH_STAT_WORDS] = {};
    init[CS_START_MIN] = init[CS_SAVAIL_MIN] = init[CS_STOP_MIN] = init[CS_PDONE_MIN] = init[CS_END_MIN] = init[CS_SDRY_MIN] = init[CS_SAVAIL2_MIN] = ~0ull; init[CS_SEQ] = seq;
    FLX_HIP(ctx, hipMemcpyAsync(ca.stats, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));      /* (pageable: the copy is staged before the call returns) */
  }
  const uint32_t cusWalk = (ctx->comm && cus > 4u * FLX_COMM_RESERVED_CUS) ? cus - FLX_COMM_RESERVED_CUS : cus;      /* (a gathering rank leaves a few CUs to the exchange of the frame before) */
  if (launch_chain(sc, fr, wb, ca, cusWalk, ctx->stream) != 0) return fail(ctx, FLX_ERR_DEVICE, "internal: the chained frame kernel does not take this scene");
  FLX_HIP(ctx, hipGetLastError());
  const size_t P1 = (size_t)frOne.rows * frOne.width;
  launch_resolve(frOne, ctx->d_hits + (size_t)slotP * P1, ctx->d_samples + (size_t)slotP * P1, ctx->d_last + (size_t)slotP * P1, d_out, ctx->stream, (size_t)depth * P1);
  FLX_HIP(ctx, hipGetLastError());
  ctx->timed = false;                                        /* (flx_last_frame_ms: the chained loop's frames are timed by flx_frame_end) */
  ctx->last_pipeline = 3; ctx->last_organisation = 4;
  ctx->last_chained = continuing ? 2 : 1;
  ctx->chain_seq = seq; ctx->chain_slot = slotP; ctx->chain_depth = depth; ctx->chain_params = *params; ctx->chain_scene_version = ctx->scene_version;
  return FLX_OK;
}
#endif /* FLX_EXPERIMENTS */


/* ---- the frame server (flx_server.hip) -------------------------------------------------------------------------------------
 * flx_set_frame_chain(ctx, 2): the frames of the loop are rendered by ONE persistent launch that takes them as they are posted.  flx_frame_begin posts the
 * frame's view into pinned memory (starting the launch if none runs); flx_frame_end waits for the launch's word that the frame is complete, resolves its slot
 * on the context's stream — a few CUs the launch leaves free — and goes on as for any frame.  The launch is told to end when the loop runs empty or anything
 * else needs the device or the workspace (server_stop). */
#ifndef FLX_SERVER_RESERVED_CUS
#define FLX_SERVER_RESERVED_CUS 0u          /* CUs the server launch leaves free (nothing needs them: the frames are resolved inside the launch and copied out by the DMA engines;
                                             * a kernel beside the launch would not get them anyway — its workgroups are dealt to shader engines that may have no free CU) */
#endif
#ifndef FLX_SERVER_TILE_LIST_FACTOR
#define FLX_SERVER_TILE_LIST_FACTOR 4       /* a workgroup's list of tiles to resolve: this many times its even share .. */
#define FLX_SERVER_TILE_LIST_MIN 64         /* .. this many at the least; a workgroup whose list is full leaves the rest of the frame's tiles to the others */
#endif
static flx_status server_take(flx_context *ctx, int k);
/* the device's transforms and lights follow the host's copies (uploads that a launch for a scene that moves went on over): before anything else reads them */
static flx_status dyn_flush(flx_context *ctx) {
  const uint32_t stale = ctx->dyn_device_stale;
  if (!stale || ctx->sv_running) return FLX_OK;
  FLX_HIP(ctx, hipSetDevice(ctx->device));      /* (a group ends its contexts' launches one after the other from one thread) */
  ctx->dyn_device_stale = 0u;
  flx_status s;
  const uint64_t sv = ctx->scene_version, dv = ctx->dyn_version;
  const uint32_t tv = ctx->transforms_version;
  if (stale & 1u) {
    const std::vector<float> r = ctx->h_rotation, sh = ctx->h_shift;
    if ((s = upload(ctx, &ctx->d_rotation, r.data(), r.size() * sizeof(float)))) return s;
    if ((s = upload(ctx, &ctx->d_shift, sh.data(), sh.size() * sizeof(float)))) return s;
  }
  if (stale & 2u) {
    const std::vector<float> l = ctx->h_lights;
    if ((s = upload(ctx, &ctx->d_lights, l.data(), l.size() * sizeof(float)))) return s;
  }
  ctx->scene_version = sv; ctx->dyn_version = dv; ctx->transforms_version = tv;      /* (the same contents the versions were counted for) */
  return FLX_OK;
}
flx_status flx_server_stop(flx_context *ctx) {
  if (!ctx->sv_running && !(ctx->sv_pending[0].valid || ctx->sv_pending[1].valid || ctx->sv_pending[2].valid)) return dyn_flush(ctx);
  if (ctx->sv_running) {
    /* every frame posted is completed before the launch ends */
    __atomic_store_n(&ctx->h_sv_mail->stopAfter, ctx->sv_next_seq - 1u, __ATOMIC_RELEASE);
    ctx->sv_running = false;
    FLX_HIP(ctx, hipStreamSynchronize(ctx->sv_stream));
  }
  /* the frames still in flight are resolved into their output slots now: whoever asked for the stop may overwrite the workspace */
  for (int i = 0; i < ctx->fifo_n; i++) {
    if (ctx->fifo[i].lane != ctx) continue;
    const int k = ctx->fifo[i].slot;
    if (ctx->sv_pending[k].valid) { flx_status s = server_take(ctx, k); if (s) return s; }
  }
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return dyn_flush(ctx);
}
static flx_status server_stop(flx_context *ctx) { return flx_server_stop(ctx); }

/* everything a launch for frames of this shape needs in memory (hipMalloc / hipFree wait for the device: never while a launch is running there) */
static flx_status server_allocate(flx_context *ctx, const DeviceFrame &frOne, uint32_t depth) {
  const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
  flx_status s;
  DeviceFrame fr = frOne;
  fr.frames = depth; fr.rows = depth * frOne.frame_rows;
  const size_t itemsPerSlot = (size_t)path_item_count64(frOne);
  if (!ctx->d_sv_slots) {
    /* The launch's stream must not share a hardware queue with the streams that work beside it (the runtime maps streams onto a few queues, and a queue
     * is served in order: a copy behind the persistent launch in the same queue would wait for its end).  Streams of another priority have queues of
     * their own: the launch goes to the lowest. */
    int prLow = 0, prHigh = 0;
    FLX_HIP(ctx, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    FLX_HIP(ctx, hipStreamCreateWithPriority(&ctx->sv_stream, hipStreamNonBlocking, prLow));
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_slots, SV_MAX_DEPTH * sizeof(ServerSlot)));
    FLX_HIP(ctx, hipHostMalloc((void **)&ctx->h_sv_mail, sizeof(ServerMail), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_sv_mail, 0, sizeof(ServerMail));
    FLX_HIP(ctx, hipHostGetDevicePointer((void **)&ctx->d_sv_mail, ctx->h_sv_mail, 0));
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_relay, sizeof(ServerMail)));
    const size_t ringWords = (size_t)cus * server_rings_per_group();
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_rings, ringWords * sizeof(uint32_t)));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_sv_rings, 0xff, ringWords * sizeof(uint32_t), ctx->stream));      /* WF_INVALID everywhere; a launch leaves them so */
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_stats, SV_STAT_TOTAL * sizeof(unsigned long long)));
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_versions, (size_t)cus * SV_MAX_DEPTH * SV_BLOB_WORDS * sizeof(uint32_t)));      /* (3 MB at 256 CUs) */
  }
  int chains = 1;
  if ((s = ensure_workspace(ctx, fr, 3, false, chains))) return s;
  const size_t P1 = (size_t)frOne.rows * frOne.width;
  if (!ctx->sv_target_slots && ctx->sv_out_capacity < SV_MAX_DEPTH * P1) {      /* (with a frame target the launch resolves into the caller's images) */
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sv_out_capacity = 0;
    if (ctx->d_sv_out) { FLX_HIP(ctx, hipFree(ctx->d_sv_out)); ctx->d_sv_out = nullptr; }
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_out, SV_MAX_DEPTH * P1 * sizeof(float4)));
    ctx->sv_out_capacity = SV_MAX_DEPTH * P1;
  }
  /* a workgroup's list of the screen tiles it made of a frame: a few times its even share (its tiles come to it as it asks for them) */
  const size_t tilesPerSlot0 = itemsPerSlot / ((size_t)fr.samples * 64u);
  const size_t groupsMin = ctx->sv_groups && ctx->sv_groups < cus ? ctx->sv_groups : (cus > 16 ? cus - 8 : cus);
  size_t tcap = FLX_SERVER_TILE_LIST_FACTOR * (tilesPerSlot0 / groupsMin + 1);
  if (tcap < FLX_SERVER_TILE_LIST_MIN) tcap = FLX_SERVER_TILE_LIST_MIN;
  if (ctx->sv_tile_cap < tcap) {
    ctx->sv_tile_cap = 0;
    if (ctx->d_sv_tiles) { FLX_HIP(ctx, hipFree(ctx->d_sv_tiles)); ctx->d_sv_tiles = nullptr; }
    FLX_HIP(ctx, hipMalloc(&ctx->d_sv_tiles, (size_t)cus * SV_MAX_DEPTH * tcap * sizeof(uint32_t)));
    ctx->sv_tile_cap = tcap;
  }
  if (!ctx->copy_stream) {
    FLX_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 3; i++) {
      FLX_HIP(ctx, hipEventCreate(&ctx->ev_slot_start[i]));
      FLX_HIP(ctx, hipEventCreate(&ctx->ev_slot_traced[i]));
      FLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slot_done[i], hipEventDisableTiming));
    }
  }
  return FLX_OK;
}

/* the lights and transforms of this scene can travel with the frames of a launch */
static bool server_versions_fit(const flx_context *ctx) {
  return ctx->sv_moving && ctx->have_transforms && ctx->n_transforms != 0 && server_blob_words(ctx->n_transforms, ctx->n_lights) <= SV_BLOB_WORDS &&
         ctx->h_rotation.size() == (size_t)ctx->n_transforms * 24 && ctx->h_shift.size() == (size_t)ctx->n_transforms * 8 && ctx->h_lights.size() == (size_t)ctx->n_lights * 6;
}
static bool server_continues(flx_context *ctx, const flx_frame_params *params, bool out8) {      /* the running launch takes this frame as it is */
  const uint32_t depth = ctx->frame_lanes == 3 ? 3u : 2u;
  /* (a launch that takes the lights and transforms per frame goes on over their uploads; one that read them at its start does not) */
  const bool same_scene = ctx->sv_ver ? ctx->sv_structure_version == ctx->structure_version : ctx->sv_scene_version == ctx->scene_version;
  return ctx->sv_running && ctx->sv_depth == depth && chain_same_shape(ctx->sv_params, *params) && same_scene && ctx->sv_out8 == out8;
}
/* For a device group (flx_group_frame_begin): would flx_frame_begin of this frame have to end or start a launch?  And the memory a launch needs, made while
 * NO launch of the group runs — where contexts share a device, an allocation in one waits for the launch of the other. */
int flx_server_takes_moving_scene(const flx_context *ctx) { return ctx->sv_want_ver && server_versions_fit(ctx) ? 1 : 0; }
int flx_server_continues(flx_context *ctx, const flx_frame_params *params) { return server_continues(ctx, params, ctx->sv_target_slots != 0u && ctx->sv_target8) ? 1 : 0; }
flx_status flx_server_prepare(flx_context *ctx, const flx_frame_params *params) {
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_frame(ctx, params, sc, fr);
  if (s) return s;
  if ((s = flx_server_stop(ctx))) return s;
  if (ctx->sv_stream) FLX_HIP(ctx, hipStreamSynchronize(ctx->sv_stream));
  if ((s = server_allocate(ctx, fr, ctx->frame_lanes == 3 ? 3u : 2u))) return s;
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return FLX_OK;
}

static flx_status server_post(flx_context *ctx, const flx_frame_params *params, const DeviceScene &sc, const DeviceFrame &frOne, uint32_t *seqOut, uint32_t *slotOut, bool out8) {
  const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
  const uint32_t depth = ctx->frame_lanes == 3 ? 3u : 2u;
  flx_status s;
  if (ctx->sv_running && !server_continues(ctx, params, out8))
    if ((s = server_stop(ctx))) return s;
  if (ctx->sv_running && server_idled_with_nothing_owed(ctx)) {      /* the launch ended by itself while the host was away (nothing owed): another begins */
    ctx->sv_running = false;
    FLX_HIP(ctx, hipStreamSynchronize(ctx->sv_stream));
    __atomic_store_n(&ctx->h_dev_error[0], 0u, __ATOMIC_RELEASE);
  }
  if (!ctx->sv_running) {
    if ((s = dyn_flush(ctx))) return s;
    if (ctx->sv_stream) FLX_HIP(ctx, hipStreamSynchronize(ctx->sv_stream));      /* a launch that was told to end reads the mailbox until it has */
    if (server_idled_with_nothing_owed(ctx)) __atomic_store_n(&ctx->h_dev_error[0], 0u, __ATOMIC_RELEASE);
    /* frames of the launch before that nobody has taken yet (it ended by itself while the host was away): taken now, before the mailbox becomes the new launch's */
    for (int i = 0; i < ctx->fifo_n; i++) {
      if (ctx->fifo[i].lane != ctx) continue;
      const int kp = ctx->fifo[i].slot;
      if (ctx->sv_pending[kp].valid && (s = server_take(ctx, kp))) return s;
    }
    DeviceFrame fr = frOne;                                 /* the slots stacked like a batch of frames; the views come through the mailbox */
    fr.frames = depth; fr.rows = depth * frOne.frame_rows;
    for (uint32_t i = 0; i < depth; i++) memset(&fr.view[i], 0, sizeof(FrameView));
    const size_t itemsPerSlot = (size_t)path_item_count64(frOne);
    if ((s = server_allocate(ctx, frOne, depth))) return s;
    const size_t P1 = (size_t)frOne.rows * frOne.width;
    ctx->sv_out_pixels = P1;
    ctx->chain_seq = 0;                                      /* (the workspace a chain of launches keeps its state in is the server's now) */
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));         /* whatever used the workspace before, and the allocations above */
    ctx->sv_counter += 16u;                                  /* sequence numbers: consecutive within a launch, never reused across launches, never 0 */
    if (ctx->sv_counter < 16u) ctx->sv_counter = 16u;
    const uint32_t seq0 = ctx->sv_counter;
    memset((void *)ctx->h_sv_mail, 0, sizeof(ServerMail));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_sv_slots, 0, SV_MAX_DEPTH * sizeof(ServerSlot), ctx->sv_stream));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_sv_relay, 0, sizeof(ServerMail), ctx->sv_stream));
    FLX_HIP(ctx, hipMemsetAsync(ctx->d_sv_stats, 0, SV_STAT_TOTAL * sizeof(unsigned long long), ctx->sv_stream));
    WavefrontBuffers wb = {};
    wb.rec = ctx->d_rec; wb.rec0 = ctx->d_rec0; wb.pix0 = ctx->d_pix0;
    wb.frameRings = ctx->d_sv_rings; wb.front = 1u;
    wb.error = ctx->d_dev_error; wb.watchdog = 0u; wb.inject = 0u;
    wb.item_base = 0u; wb.item_count = (uint32_t)(depth * itemsPerSlot);
    wb.hits = ctx->d_hits; wb.sampleRadiance = ctx->d_samples; wb.lastOriginal = ctx->d_last; wb.counters = nullptr;
    ServerArgs sa = {};
    sa.slots = ctx->d_sv_slots; sa.mail = ctx->d_sv_mail; sa.relay = ctx->d_sv_relay;
    /* frames of the launch before may still be in its images (taken, not yet handed out): this launch goes on with the slot after them */
    /* with a frame target the image IS the slot, and the caller was promised "image i takes the frames begun i-th, (i + n)-th, ..": a launch that starts on an empty
     * loop (frames begun and ended one at a time end the launch every frame) goes on counting where the last one stopped — flx_share's ranks reuse the image of
     * frame g - n for frame g and would otherwise all write image 0 while the root still reads it */
    const uint32_t slot0 = ctx->sv_target_slots ? (uint32_t)(ctx->sv_target_posted % depth) : ((ctx->fifo_n && ctx->sv_depth == depth) ? ctx->sv_next_slot % depth : 0u);
    sa.depth = depth; sa.slot0 = slot0; sa.seq0 = seq0;
    sa.tilesPerSlot = (uint32_t)(itemsPerSlot / ((size_t)fr.samples * 64u));
    sa.itemsPerSlot = (uint32_t)itemsPerSlot;
    for (uint32_t i = 0; i < depth; i++) sa.out[i] = ctx->d_sv_out + (size_t)i * P1;
    if (ctx->sv_target_slots) {
      /* the frames are this context's row strips of images somebody else owns: resolved straight into them, each row where the image has it */
      const size_t first = (size_t)params->tile_index * params->tile_rows * params->width;      /* pixels in front of this context's first strip */
      for (uint32_t i = 0; i < depth; i++) sa.out[i] = out8 ? (float4 *)((uint32_t *)ctx->sv_target[i] + first) : ctx->sv_target[i] + first;
      sa.outStripRows = params->tile_rows; sa.outStripStep = params->tile_rows * params->tile_count; sa.outSystem = 1u;
    }
    sa.out8 = out8 ? 1u : 0u;
    sa.tileLists = ctx->d_sv_tiles; sa.tileListCap = (uint32_t)ctx->sv_tile_cap;
    sa.idleExit = 200000000u;                                /* 2 s at 100 MHz: a safety net, the host always says when to stop */
    sa.error = ctx->d_dev_error;
    sa.stats = ctx->d_sv_stats;
    uint32_t cusWalk = cus > 4u * FLX_SERVER_RESERVED_CUS ? cus - FLX_SERVER_RESERVED_CUS : cus;
    if (ctx->sv_groups && ctx->sv_groups < cusWalk) cusWalk = ctx->sv_groups;
    /* a scene that moves: the launch reads the lights and transforms of every frame from versions of its own, filled from what is posted with the frame */
    const bool ver = ctx->sv_want_ver && server_versions_fit(ctx);
    DeviceScene scL = sc;
    if (ver) {
      const size_t versions = (size_t)cusWalk * depth;
      sa.blobWords = server_blob_words(ctx->n_transforms, ctx->n_lights);
      scL.rotation = (const float4 *)ctx->d_sv_versions;
      scL.shift = (const float4 *)(ctx->d_sv_versions + versions * ctx->n_transforms * 24u);
      scL.lights = (const float *)(ctx->d_sv_versions + versions * ctx->n_transforms * 32u);
    }
    if (launch_server(scL, fr, wb, sa, cusWalk, ctx->sv_stream) != 0) return fail(ctx, FLX_ERR_DEVICE, "internal: the frame server does not take this scene");
    FLX_HIP(ctx, hipGetLastError());
    ctx->sv_running = true; ctx->sv_depth = depth; ctx->sv_next_seq = seq0; ctx->sv_next_slot = slot0;
    ctx->sv_params = *params; ctx->sv_scene_version = ctx->scene_version; ctx->sv_out8 = out8;
    ctx->sv_ver = ver; ctx->sv_structure_version = ctx->structure_version;
  }
  const uint32_t seq = ctx->sv_next_seq++, slot = ctx->sv_next_slot;
  ctx->sv_next_slot = (slot + 1u) % depth;
  if (ctx->sv_target_slots) ctx->sv_target_posted++;
  /* post: the view, then the number that says whose view it is.  (The frame that was in this slot was taken by flx_frame_end `depth` frames ago.) */
  memcpy((void *)&ctx->h_sv_mail->view[slot], &frOne.view[0], sizeof(FrameView));
  if (ctx->sv_ver) {                                         /* ... and the frame's transforms and lights (as the host holds them: what the last uploads said) */
    uint32_t *b = (uint32_t *)ctx->h_sv_mail->blob[slot];
    memcpy(b, ctx->h_rotation.data(), (size_t)ctx->n_transforms * 96);
    memcpy(b + (size_t)ctx->n_transforms * 24, ctx->h_shift.data(), (size_t)ctx->n_transforms * 32);
    if (ctx->n_lights) memcpy(b + (size_t)ctx->n_transforms * 32, ctx->h_lights.data(), (size_t)ctx->n_lights * 24);
  }
  __atomic_store_n(&ctx->h_sv_mail->posted[slot], seq, __ATOMIC_RELEASE);
  *seqOut = seq; *slotOut = slot;
  ctx->last_pipeline = 3; ctx->last_organisation = 5; ctx->last_chained = 3;
  return FLX_OK;
}

/* flx_frame_end of a server frame: wait for the launch's word, resolve the frame's slot into the output slot */
static flx_status server_take(flx_context *ctx, int k) {
  auto &pf = ctx->sv_pending[k];
  pf.valid = false;
  ctx->slot_latency_ms[k] = -1.f;
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (__atomic_load_n(&ctx->h_sv_mail->done[pf.slot], __ATOMIC_ACQUIRE) != pf.seq) {
    if (__atomic_load_n(&ctx->h_dev_error[0], __ATOMIC_ACQUIRE) != 0u) break;
    if ((++spins & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
      __atomic_fetch_or(&ctx->h_dev_error[0], WF_ERR_SERVER_TIMEOUT, __ATOMIC_RELEASE);
      break;
    }
    /* back off: a pause per poll while the word is about to come, then the core goes to whoever else wants it (eight ranks' hosts share a box's cores; the
     * Node host waits on a worker thread, napi/flexlight_napi.cc frameEndAsync) */
    if (spins < 4096u) __builtin_ia32_pause(); else sched_yield();
  }
  if (server_idled_with_nothing_owed(ctx)) {
    /* the launch ended by itself while the host was away, this frame and every other it had been given complete: forgiven HERE, while the mailbox still is that
     * launch's (a later launch's posts must not be held against it) */
    ctx->sv_running = false;
    FLX_HIP(ctx, hipStreamSynchronize(ctx->sv_stream));
    __atomic_store_n(&ctx->h_dev_error[0], 0u, __ATOMIC_RELEASE);
  }
  if (__atomic_load_n(&ctx->h_dev_error[0], __ATOMIC_ACQUIRE) != 0u) {
    /* the launch gave up (or never answered): tell it to end, then report */
    __atomic_store_n(&ctx->h_sv_mail->stopAfter, 1u, __ATOMIC_RELEASE);
    ctx->sv_running = false;
    return FLX_OK;                                           /* (flx_frame_end's own check reports the error word) */
  }
  /* the frame is in the launch's output buffer of its slot (every workgroup resolved the screen tiles it made): nothing to launch beside the server */
  ctx->slot_latency_ms[k] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - pf.posted).count();      /* post .. the launch's word, on the host's clock */
  const float4 *src = ctx->sv_target_slots ? ctx->sv_target[pf.slot] : ctx->d_sv_out + (size_t)pf.slot * ctx->sv_out_pixels;
  ctx->slot_dev_ptr[k] = src;
  FLX_HIP(ctx, hipEventRecord(ctx->ev_slot_start[k], ctx->copy_stream));
  FLX_HIP(ctx, hipEventRecord(ctx->ev_slot_traced[k], ctx->copy_stream));
  if (pf.format != FLX_FRAME_DEVICE) {
    if (ctx->slot_bytes[k]) FLX_HIP(ctx, hipMemcpyAsync(ctx->h_slot[k], src, ctx->slot_bytes[k], hipMemcpyDeviceToHost, ctx->copy_stream));
    FLX_HIP(ctx, hipEventRecord(ctx->ev_slot_done[k], ctx->copy_stream));
  }
  return FLX_OK;
}

constexpr int NOT_GATHERED = -2;      /* frame_begin_on's `gather`: this context's own frame; -1: gathered on every rank; >= 0: on that rank */
static flx_status frame_begin_on(flx_context *ctx, const flx_frame_params *params, int format, int gather, int *slot, int chained = 0 /* 1: a chain of launches, 2: the frame server */) {
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  DeviceScene sc; DeviceFrame fr;
  flx_status s = flx_make_frame(ctx, params, sc, fr);
  if (s) return s;
  if (!chained) { ctx->chain_seq = 0; ctx->last_chained = 0; }      /* (a frame of another kind ends the chain: its kernels share the workspace) */
  const bool gathered = gather != NOT_GATHERED;
  const bool receiver = !gathered || gather < 0 || gather == ctx->comm_rank;
  if (gathered) { fr.rows = receiver ? params->height : 0u; }      /* the slot holds the WHOLE frame on a rank that receives it, nothing elsewhere */
  const int k = (int)(ctx->frames_begun % (uint64_t)(ctx->frame_lanes == 3 ? 3 : 2));      /* (the depth of the loop does not change while frames are in flight) */
  if (!ctx->copy_stream) {
    FLX_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 3; i++) {
      FLX_HIP(ctx, hipEventCreate(&ctx->ev_slot_start[i]));
      FLX_HIP(ctx, hipEventCreate(&ctx->ev_slot_traced[i]));
      FLX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slot_done[i], hipEventDisableTiming));
    }
  }
  const size_t pixels = (size_t)fr.rows * fr.width;
  const size_t bytes = pixels * (format == FLX_FRAME_RGBA8 ? sizeof(uint32_t) : sizeof(float4));
  /* (hipMalloc / hipFree / hipHostMalloc wait for the device: with the frame server's launch running they would wait for its end — which waits for this
   * frame.  A slot that has to grow ends the launch first; the next frame starts another.) */
  const bool served = chained == 2 && !gathered && pixels;      /* (a frame of the server is handed out from the launch's own images, or the caller's: it needs no device slot) */
  if (ctx->sv_running && ((!served && (ctx->slot_capacity[k] < (pixels ? pixels : 1) || !ctx->d_slot[k])) || (format != FLX_FRAME_DEVICE && (ctx->h_slot_capacity[k] < bytes || !ctx->h_slot[k]))))
    if ((s = flx_server_stop(ctx))) return s;
  if (!served && (s = flx_ensure_pixels(ctx, &ctx->d_slot[k], &ctx->slot_capacity[k], pixels ? pixels : 1))) return s;
  if (format == FLX_FRAME_RGBA8 && !served && ctx->slot8_capacity[k] < pixels) {      /* (a frame of the server is quantised where it is resolved) */
    ctx->slot8_capacity[k] = 0;
    if (ctx->d_slot8[k]) { FLX_HIP(ctx, hipFree(ctx->d_slot8[k])); ctx->d_slot8[k] = nullptr; }
    FLX_HIP(ctx, hipMalloc(&ctx->d_slot8[k], (pixels ? pixels : 1) * sizeof(uint32_t)));
    ctx->slot8_capacity[k] = pixels;
  }
  if (format != FLX_FRAME_DEVICE && (ctx->h_slot_capacity[k] < bytes || !ctx->h_slot[k])) {
    ctx->h_slot_capacity[k] = 0;
    if (ctx->h_slot[k]) { FLX_HIP(ctx, hipHostFree(ctx->h_slot[k])); ctx->h_slot[k] = nullptr; }
    FLX_HIP(ctx, hipHostMalloc(&ctx->h_slot[k], bytes ? bytes : 16, hipHostMallocDefault));
    ctx->h_slot_capacity[k] = bytes;
  }
  if (served) {
    /* the frame server: the frame is posted to the running launch; flx_frame_end takes it (server_take) */
    uint32_t seq = 0, sslot = 0;
    if ((s = server_post(ctx, params, sc, fr, &seq, &sslot, ctx->sv_target_slots ? ctx->sv_target8 : format == FLX_FRAME_RGBA8))) return s;
    ctx->sv_pending[k].valid = true; ctx->sv_pending[k].seq = seq; ctx->sv_pending[k].slot = sslot; ctx->sv_pending[k].format = format; ctx->sv_pending[k].fr = fr;
    ctx->sv_pending[k].posted = std::chrono::steady_clock::now();
    ctx->slot_host[k] = format != FLX_FRAME_DEVICE;
    ctx->slot_bytes[k] = bytes;
    ctx->frames_begun++;
    *slot = k;
    return FLX_OK;
  }
  ctx->slot_dev_ptr[k] = nullptr; ctx->slot_latency_ms[k] = -1.f;
  FLX_HIP(ctx, hipEventRecord(ctx->ev_slot_start[k], ctx->stream));
  if (gathered) {
    /* this rank's strips, the exchange over the lane's communicator and the reassembly, all on the lane's stream */
    if ((s = flx_gather_enqueue(ctx, params, 1, gather, receiver ? ctx->d_slot[k] : nullptr))) return s;
    if (pixels && format == FLX_FRAME_RGBA8) { launch_quantize(ctx->d_slot[k], ctx->d_slot8[k], pixels, ctx->stream); FLX_HIP(ctx, hipGetLastError()); }
  } else if (pixels) {
    if (params->use_filter || params->is_temporal) {
      s = run_post_frame(ctx, sc, fr, params, ctx->d_slot[k]);
#if FLX_EXPERIMENTS
    } else if (chained == 1) {
      s = chain_run_frame(ctx, params, sc, fr, ctx->d_slot[k]);
#endif
    } else {
      GBufferPtrs gb = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
      s = flx_run_frame(ctx, sc, fr, ctx->d_slot[k], gb);
    }
    if (s) return s;
    if (format == FLX_FRAME_RGBA8) { launch_quantize(ctx->d_slot[k], ctx->d_slot8[k], pixels, ctx->stream); FLX_HIP(ctx, hipGetLastError()); }
  }
  FLX_HIP(ctx, hipEventRecord(ctx->ev_slot_traced[k], ctx->stream));
  if (format == FLX_FRAME_DEVICE) {
    ctx->slot_host[k] = false;                   /* the pixels stay in device memory: nothing to copy, the frame is done when it is traced */
  } else {
    /* the copy to the host runs beside the next frame's kernels */
    FLX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_slot_traced[k], 0));
    if (bytes) FLX_HIP(ctx, hipMemcpyAsync(ctx->h_slot[k], format == FLX_FRAME_RGBA8 ? (const void *)ctx->d_slot8[k] : (const void *)ctx->d_slot[k], bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    FLX_HIP(ctx, hipEventRecord(ctx->ev_slot_done[k], ctx->copy_stream));
    ctx->slot_host[k] = true;
  }
  ctx->slot_bytes[k] = bytes;
  ctx->frames_begun++;
  *slot = k;
  return FLX_OK;
}

extern "C" int flx_frames_in_flight(const flx_context *ctx) { return ctx ? ctx->fifo_n : 0; }

extern "C" flx_status flx_debug_inject_fault(flx_context *ctx, uint32_t watchdog_polls, uint32_t flags) {
  if (!ctx) return FLX_ERR_INVALID;
  if (flags & ~WF_INJECT_NO_SHADING) return fail(ctx, FLX_ERR_INVALID, "flx_debug_inject_fault: unknown flag");
  ctx->inject_watchdog = watchdog_polls; ctx->inject_flags = flags;
  if (ctx->twin) { ctx->twin->inject_watchdog = watchdog_polls; ctx->twin->inject_flags = flags; }
  return FLX_OK;
}
extern "C" flx_status flx_set_frame_chain(flx_context *ctx, int mode) {
  if (!ctx) return FLX_ERR_INVALID;
  if (mode < 0 || mode > 3) return fail(ctx, FLX_ERR_INVALID, "flx_set_frame_chain: 0 (every frame its own launches), 1 (a chain of launches that work ahead on each other's frames), 2 (the frame server — one persistent launch takes the loop's frames as they are posted — for frames of fewer than 64 screen tiles per CU) or 3 (the frame server for every frame it can take)");
#if !FLX_EXPERIMENTS
  if (mode == 1) return fail(ctx, FLX_ERR_INVALID, "flx_set_frame_chain: the chain of launches (mode 1) is not in the shipped library (make EXPERIMENTS=1; it lost to the frame server, modes 2 / 3)");
#endif
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_set_frame_chain: frames are in flight");
  ctx->frame_chain = mode; ctx->chain_seq = 0;
  return FLX_OK;
}
#if FLX_EXPERIMENTS
extern "C" flx_status flx_set_chain_stats(flx_context *ctx, int on) {
  if (!ctx) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (on && !ctx->d_chain_stats) {
    FLX_HIP(ctx, hipMalloc(&ctx->d_chain_stats, (size_t)CH_STAT_LAUNCHES * CH_STAT_WORDS * sizeof(unsigned long long)));
    FLX_HIP(ctx, hipMemset(ctx->d_chain_stats, 0, (size_t)CH_STAT_LAUNCHES * CH_STAT_WORDS * sizeof(unsigned long long)));
  } else if (!on && ctx->d_chain_stats) { FLX_HIP(ctx, hipFree(ctx->d_chain_stats)); ctx->d_chain_stats = nullptr; }
  return FLX_OK;
}
extern "C" flx_status flx_get_chain_stats(flx_context *ctx, uint64_t *out) {
  if (!ctx || !out) return FLX_ERR_INVALID;
  if (!ctx->d_chain_stats) return fail(ctx, FLX_ERR_INVALID, "flx_get_chain_stats: flx_set_chain_stats(ctx, 1) first");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FLX_HIP(ctx, hipMemcpy(out, ctx->d_chain_stats, (size_t)CH_STAT_LAUNCHES * CH_STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return FLX_OK;
}
/* diagnostics / experiments: an explicit order of the screen tiles of a chained frame (n = tiles of the frame, or 0: none), and per-tile shading counts */
extern "C" flx_status flx_set_chain_order(flx_context *ctx, const uint32_t *order, uint32_t n) {
  if (!ctx) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->chain_seq = 0;
  if (ctx->d_chain_order) { FLX_HIP(ctx, hipFree(ctx->d_chain_order)); ctx->d_chain_order = nullptr; ctx->chain_order_n = 0; }
  if (!order || n == 0u) return FLX_OK;
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t i = 0; i < n; i++) { if (order[i] >= n || seen[order[i]]) return fail(ctx, FLX_ERR_INVALID, "flx_set_chain_order: not a permutation"); seen[order[i]] = 1; }
  FLX_HIP(ctx, hipMalloc(&ctx->d_chain_order, (size_t)n * sizeof(uint32_t)));
  FLX_HIP(ctx, hipMemcpy(ctx->d_chain_order, order, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  ctx->chain_order_n = n;
  return FLX_OK;
}
extern "C" flx_status flx_set_chain_cost(flx_context *ctx, uint32_t n) {
  if (!ctx) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->chain_seq = 0;
  if (ctx->d_chain_cost) { FLX_HIP(ctx, hipFree(ctx->d_chain_cost)); ctx->d_chain_cost = nullptr; ctx->chain_cost_n = 0; }
  if (n == 0u) return FLX_OK;
  FLX_HIP(ctx, hipMalloc(&ctx->d_chain_cost, 2 * (size_t)n * sizeof(uint32_t)));
  FLX_HIP(ctx, hipMemset(ctx->d_chain_cost, 0, 2 * (size_t)n * sizeof(uint32_t)));
  ctx->chain_cost_n = n;
  return FLX_OK;
}
extern "C" flx_status flx_get_chain_cost(flx_context *ctx, uint32_t *out /* [2 * n] */) {
  if (!ctx || !out || !ctx->d_chain_cost) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FLX_HIP(ctx, hipMemcpy(out, ctx->d_chain_cost, 2 * ctx->chain_cost_n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return FLX_OK;
}
#endif /* FLX_EXPERIMENTS */
extern "C" flx_status flx_get_server_dump(flx_context *ctx, uint64_t *out /* [4 * 72] */) {
  if (!ctx || !out || !ctx->d_sv_stats) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipMemcpy(out, ctx->d_sv_stats + SV_STAT_WORDS, (size_t)SV_DUMP_MAX * SV_DUMP_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return FLX_OK;
}
extern "C" flx_status flx_get_server_stats(flx_context *ctx, uint64_t *out /* [16] */) {
  if (!ctx || !out) return FLX_ERR_INVALID;
  if (!ctx->d_sv_stats) return fail(ctx, FLX_ERR_INVALID, "flx_get_server_stats: no frame server has run");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipMemcpy(out, ctx->d_sv_stats, SV_STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (ctx->h_sv_mail) { out[13] = ((uint64_t)ctx->h_sv_mail->done[0] << 32) | ctx->h_sv_mail->done[1]; out[14] = ((uint64_t)ctx->h_sv_mail->posted[0] << 32) | ctx->h_sv_mail->posted[1]; out[15] = ((uint64_t)ctx->sv_next_seq << 32) | ctx->h_sv_mail->stopAfter; }
  return FLX_OK;
}
extern "C" flx_status flx_last_chained(flx_context *ctx, int *chained) {
  if (!ctx || !chained) return FLX_ERR_INVALID;
  *chained = ctx->last_chained;
  return FLX_OK;
}

/* Would flx_frame_begin hand this frame to the frame server (with flx_set_frame_chain(ctx, 3): whatever its size)? */
extern "C" int flx_frame_server_takes(flx_context *ctx, const flx_frame_params *params) {
  if (!ctx || !params) return 0;
  DeviceScene scT; DeviceFrame frT;
  if (flx_make_frame(ctx, params, scT, frT) != FLX_OK || frT.rows == 0u) return 0;
  return chain_wanted(ctx, params, scT, frT) && ctx->frame_chain >= 2 ? 1 : 0;
}

/* The frames of the loop are resolved by the server's launch straight into images the caller owns: d_images[i] (i < n_images = the loop's frames in flight;
 * float4[height][width], addresses this context's GPU can write: a peer's memory, pinned host memory, its own) takes the frames begun i-th, (i + n)-th, ..
 * With params.tile_count > 1 the context writes its row strips where the image has them, so the contexts of a device group complete ONE image between them
 * without any exchange.  n_images = 0: the launch's own buffers again. */
extern "C" flx_status flx_frame_target_set(flx_context *ctx, void *const *d_images, uint32_t n_images) {
  if (!ctx) return FLX_ERR_INVALID;
  if (n_images > 3u || n_images == 1u || (n_images && !d_images)) return fail(ctx, FLX_ERR_INVALID, "flx_frame_target_set: 0, 2 or 3 images");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_status s = flx_server_stop(ctx);      /* (frames in flight are completed where they were begun: flx_frame_end still hands them out from there) */
  if (s) return s;
  if (ctx->sv_stream) FLX_HIP(ctx, hipStreamSynchronize(ctx->sv_stream));
  for (uint32_t i = 0; i < 3u; i++) ctx->sv_target[i] = i < n_images ? (float4 *)d_images[i] : nullptr;
  for (uint32_t i = 0; i < n_images; i++) if (!ctx->sv_target[i]) { ctx->sv_target_slots = 0; return fail(ctx, FLX_ERR_INVALID, "flx_frame_target_set: an image is NULL"); }
  ctx->sv_target_slots = n_images;
  ctx->sv_target_posted = 0;
  ctx->sv_target8 = false;
  return FLX_OK;
}
/* ... images of the canvas' RGBA8 (uint32 per pixel): the launch quantises its tiles as it resolves them */
extern "C" flx_status flx_frame_target_set8(flx_context *ctx, void *const *d_images, uint32_t n_images) {
  const flx_status s = flx_frame_target_set(ctx, d_images, n_images);
  if (!s) ctx->sv_target8 = n_images != 0u;
  return s;
}
/* the image (index into flx_frame_target_set's) the frame begun last will be in */
extern "C" int flx_frame_target_index(const flx_context *ctx) {
  if (!ctx || !ctx->sv_target_slots || !ctx->sv_running) return -1;
  return (int)((ctx->sv_next_slot + ctx->sv_depth - 1u) % ctx->sv_depth);
}
/* the per-pixel kernel with a pixel's samples side by side (k_trace_samples) where the frame allows it (2, 4 or 8 samples, at most 4 bounces): 1 (default) / 0; for A/B runs */
extern "C" flx_status flx_debug_set_sample_parallel(flx_context *ctx, int on) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_sample_parallel: frames are in flight");
  ctx->sample_parallel = on ? 1 : 0;
  return FLX_OK;
}
/* walk jobs per lane of the frame kernel's walk waves (1: k_wf_frame, 2: k_wf_frame2 where the front of the frame is inside the launch); for A/B runs */
extern "C" flx_status flx_debug_set_walk_jobs(flx_context *ctx, int jobs) {
  if (!ctx) return FLX_ERR_INVALID;
  if (jobs < 0 || jobs > 2) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_walk_jobs: 0 (the default), 1 or 2");
#if !FLX_EXPERIMENTS
  if (jobs == 2) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_walk_jobs: two jobs per lane (k_wf_frame2) measured slower and is not in the shipped library (make EXPERIMENTS=1)");
#endif
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_walk_jobs: frames are in flight");
  ctx->walk_jobs = jobs ? (uint32_t)jobs : (uint32_t)FLX_WALK_JOBS_DEFAULT;
  return FLX_OK;
}
/* The order in which the frame kernel's workgroups draw the frame's 8 x 8 screen tiles (k_wf_frame with its front inside): order[q] = the tile the q-th draw makes, a permutation of
 * 0 .. n - 1 (n = the frame's tiles), or n = 0: tile q (the default).  Frames do not depend on it (every path writes its own slot); a launch's drain does. */
extern "C" flx_status flx_debug_set_tile_order(flx_context *ctx, const uint32_t *order, uint32_t n) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_tile_order: frames are in flight");
  flx_status s = flx_server_stop(ctx);
  if (s) return s;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->d_tile_order) { (void)hipFree(ctx->d_tile_order); ctx->d_tile_order = nullptr; }
  ctx->tile_order_n = 0;
  if (n == 0u) return FLX_OK;
  if (!order) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_tile_order: no order");
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t q = 0; q < n; q++) {
    if (order[q] >= n || seen[order[q]]) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_tile_order: not a permutation of the frame's tiles");
    seen[order[q]] = 1;
  }
  FLX_HIP(ctx, hipMalloc(&ctx->d_tile_order, (size_t)n * sizeof(uint32_t)));
  FLX_HIP(ctx, hipMemcpy(ctx->d_tile_order, order, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  ctx->tile_order_n = n;
  return FLX_OK;
}
/* The adaptive tile order (on by default): the frame kernel draws a frame's screen tiles in an order made from what the tiles cost in the last frame of the same shape
 * (flx_context.h).  0 turns it off (screen order), for A/B runs and for tests that want launches to be reproducible in their scheduling. */
extern "C" flx_status flx_debug_set_adaptive_order(flx_context *ctx, int on) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_adaptive_order: frames are in flight");
  ctx->adaptive_order = on ? 1 : 0;
  ctx->auto_order_tiles = 0;
  return FLX_OK;
}
/* k_tile_order on its own (tests): the draw order it makes of n per-tile costs; mode as launch_tile_order's */
extern "C" flx_status flx_debug_tile_order_of(flx_context *ctx, const float *cost, uint32_t n, int mode, uint32_t *order) {
  if (!ctx || !cost || !order || n == 0u) return FLX_ERR_INVALID;
  if (mode < 0 || mode > 1) return fail(ctx, FLX_ERR_INVALID, "flx_debug_tile_order_of: mode 0 (the lightest tenth last) or 1 (sixteen classes, heaviest first)");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  float *d_c = nullptr; uint32_t *d_o = nullptr;
  FLX_HIP(ctx, hipMalloc(&d_c, (size_t)n * sizeof(float)));
  if (hipMalloc(&d_o, (size_t)n * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(d_c); return fail(ctx, FLX_ERR_DEVICE, "flx_debug_tile_order_of: hipMalloc"); }
  hipError_t e = hipMemcpy(d_c, cost, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_o, 0xff, (size_t)n * sizeof(uint32_t));
  if (e == hipSuccess) { launch_tile_order(d_c, d_o, n, mode, ctx->stream); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(order, d_o, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_c); (void)hipFree(d_o);
  if (e != hipSuccess) return fail(ctx, FLX_ERR_DEVICE, hipGetErrorString(e));
  return FLX_OK;
}
/* Counted frames add, per 8 x 8 screen tile, the entries its paths' walks visited (bounce loop only): n > 0 turns that on for frames of up to n tiles and zeroes the sums,
 * out != nullptr copies them out first (as many as were asked for when it was turned on); n = 0 turns it off. */
extern "C" flx_status flx_debug_tile_cost(flx_context *ctx, unsigned long long *out, uint32_t n) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_tile_cost: frames are in flight");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (out && ctx->d_tile_cost) FLX_HIP(ctx, hipMemcpy(out, ctx->d_tile_cost, (size_t)(n && n < ctx->tile_cost_n ? n : ctx->tile_cost_n) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (n != ctx->tile_cost_n) {
    if (ctx->d_tile_cost) { (void)hipFree(ctx->d_tile_cost); ctx->d_tile_cost = nullptr; }
    ctx->tile_cost_n = 0;
    if (n) { FLX_HIP(ctx, hipMalloc(&ctx->d_tile_cost, (size_t)n * sizeof(unsigned long long))); ctx->tile_cost_n = n; }
  }
  if (ctx->d_tile_cost) FLX_HIP(ctx, hipMemset(ctx->d_tile_cost, 0, (size_t)ctx->tile_cost_n * sizeof(unsigned long long)));
  return FLX_OK;
}
/* the shading's per-triangle table (DeviceScene::angle_tan) off: every shade computes the values itself, as before round 4 — for A/B runs and the test that both agree */
extern "C" flx_status flx_debug_set_angle_table(flx_context *ctx, int on) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_angle_table: frames are in flight");
  flx_status s = flx_server_stop(ctx);
  if (s) return s;
  ctx->angle_table = on ? 1 : 0;
  ctx->scene_version++;                    /* (a chain of frames does not go on over it) */
  return FLX_OK;
}

/* rehearsal of a device group on ONE GPU: the server's launch takes only `groups` CUs, so that the launches of several contexts run beside each other */
extern "C" flx_status flx_debug_set_server_groups(flx_context *ctx, uint32_t groups) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_debug_set_server_groups: frames are in flight");
  flx_status s = flx_server_stop(ctx);
  if (s) return s;
  ctx->sv_groups = groups;
  return FLX_OK;
}

extern "C" flx_status flx_set_server_moving_scenes(flx_context *ctx, int on) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_set_server_moving_scenes: frames are in flight");
  flx_status s = flx_server_stop(ctx);
  if (s) return s;
  ctx->sv_moving = on ? 1 : 0;
  if (!on) ctx->sv_want_ver = false;
  return FLX_OK;
}
extern "C" int flx_server_moving(const flx_context *ctx) { return ctx && ctx->sv_running && ctx->sv_ver ? 1 : 0; }

extern "C" flx_status flx_set_frame_lanes(flx_context *ctx, int lanes) {
  if (!ctx) return FLX_ERR_INVALID;
  if (lanes < 1 || lanes > 3) return fail(ctx, FLX_ERR_INVALID, "flx_set_frame_lanes: 1 (frames one after the other), 2 (two frames overlap on the GPU) or 3 (three frames in flight where the loop is chained: flx_set_frame_chain; as 2 elsewhere)");
  if (ctx->fifo_n) return fail(ctx, FLX_ERR_INVALID, "flx_set_frame_lanes: frames are in flight");
  ctx->frame_lanes = lanes;
  return FLX_OK;
}

static flx_status frame_begin(flx_context *ctx, const flx_frame_params *params, int format, int gather) {
  if (format != FLX_FRAME_FLOAT && format != FLX_FRAME_RGBA8 && format != FLX_FRAME_DEVICE) return fail(ctx, FLX_ERR_INVALID, "flx_frame_begin: format is FLX_FRAME_FLOAT, FLX_FRAME_RGBA8 or FLX_FRAME_DEVICE");
  if (ctx->fifo_n >= (ctx->frame_lanes == 3 ? 3 : 2)) return fail(ctx, FLX_ERR_INVALID, ctx->frame_lanes == 3 ? "flx_frame_begin: three frames are in flight already (flx_set_frame_lanes), take one with flx_frame_end first" : "flx_frame_begin: two frames are in flight already, take one with flx_frame_end first");
  if (!params) return fail(ctx, FLX_ERR_INVALID, "frame params are NULL");
  if (gather != NOT_GATHERED) {
    if (!ctx->comm) return fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_gathered: the context belongs to no communicator (flx_comm_init_rank)");
    if (params->is_temporal) return fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_gathered: temporal frames keep their history in one context and are not sharded");
    if (ctx->frame_lanes >= 2 && !ctx->comm_twin) return fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_gathered: the second lane has no communicator (contexts of a flx_group render through flx_group_render)");
  }
  flx_context *lane = ctx;
  int chained = 0;
  if (gather == NOT_GATHERED) {
    DeviceScene scT; DeviceFrame frT;
    if (flx_make_frame(ctx, params, scT, frT) == FLX_OK && frT.rows != 0u && chain_wanted(ctx, params, scT, frT)) {
      chained = ctx->frame_chain;
      /* The frame server pays where a frame is short against its own chains — a rank's share of a frame: below FLX_SERVER_MAX_TILES_PER_CU screen tiles per
       * workgroup (a rank's eighth of a 1080p frame has 16, a whole 1080p frame 127: 6.46 ms per frame through the server against 6.32 on two lanes,
       * profiles/r04_server.txt); mode 3 takes every frame it can. */
      if (chained == 2 && !ctx->sv_target_slots && path_item_count64(frT) / ((uint64_t)frT.samples * 64u) >= (uint64_t)FLX_SERVER_MAX_TILES_PER_CU * (uint64_t)ctx->prop.multiProcessorCount) chained = 0;
      /* ... and a scene that moves — lights or transforms that change from frame to frame, examples/dragon.js turns its monkey every tick — goes through a launch
       * that takes those arrays with every frame (flx_server.hip: VER; a rank's eighth of the dragon frame with the monkey turning: 2.51 ms per frame when every
       * upload ended the launch, 1.42 on two lanes with their own copies of the arrays: tools/dynamic_scene_time.py).  Where they do not fit a post
       * (SV_BLOB_WORDS), a frame that follows an upload goes to the lanes. */
      if (chained == 2 && !ctx->sv_target_slots && ctx->begin_scene_version != 0 && ctx->begin_scene_version != ctx->scene_version &&
          !(ctx->sv_want_ver && server_versions_fit(ctx))) chained = 0;
      if (chained == 3) chained = 2;
    }
  }
  ctx->begin_scene_version = ctx->scene_version;
  if (ctx->sv_target_slots && (chained != 2 || format != FLX_FRAME_DEVICE || ctx->sv_target_slots != (ctx->frame_lanes == 3 ? 3u : 2u)))
    return fail(ctx, FLX_ERR_INVALID, "flx_frame_begin: a frame target is set (flx_frame_target_set) — the frame must be one the frame server takes (flx_frame_server_takes), FLX_FRAME_DEVICE, and the target must have as many images as the loop has frames in flight");
  if (chained == 0) ctx->last_chained = 0;
  if (chained != 2) { flx_status ss = flx_server_stop(ctx); if (ss) return ss; }      /* (a frame of another kind: the server's launch ends, its frames are resolved) */
  if (chained) {
    /* both frames in flight live in the primary context: make sure nothing of the second lane is (a frame of another kind just before) */
    if (ctx->twin && ctx->fifo_n && ctx->fifo[ctx->fifo_n - 1].lane != ctx) FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream));
  } else if (ctx->frame_lanes >= 2 && !params->is_temporal && (ctx->lane_next & 1u)) {
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->twin) {
      flx_status s = flx_context_create(ctx->device, &ctx->twin);
      if (s) return fail(ctx, s, flx_last_error(nullptr));
      ctx->twin->is_twin = true;
      ctx->twin_dyn_version = 0;
    }
    flx_context *t = ctx->twin;
    t->comm = ctx->comm_twin; t->comm_rank = ctx->comm_rank; t->comm_size = ctx->comm_size; t->comm_owned = false;      /* (the primary owns and destroys both) */
    mirror_scene(ctx);
    t->pipeline = ctx->pipeline; t->wf_groups = ctx->wf_groups; t->wf_organisation = ctx->wf_organisation; t->frame_front = ctx->frame_front; t->walk_scheduler = ctx->walk_scheduler; t->walk_suspend = ctx->walk_suspend;
    if (ctx->twin_dyn_version != ctx->dyn_version) {          /* lights / transforms changed since the twin's last frame: its own copies, on its stream */
      flx_status s;
      if (ctx->have_transforms && (s = flx_transforms_upload(t, ctx->h_rotation.data(), ctx->h_shift.data(), ctx->n_transforms))) return fail(ctx, s, flx_last_error(t));
      if ((s = flx_lights_upload(t, ctx->h_lights.data(), ctx->n_lights))) return fail(ctx, s, flx_last_error(t));
      ctx->twin_dyn_version = ctx->dyn_version;
    }
    lane = t;
  }
  int slot = 0;
  if (!chained && lane != ctx && ctx->chain_seq) { ctx->chain_seq = 0; ctx->last_chained = 0; }
  flx_status s = frame_begin_on(lane, params, format, gather, &slot, chained);
  if (s) { if (lane != ctx) ctx->err = lane->err; return s; }
  ctx->fifo[ctx->fifo_n].lane = lane; ctx->fifo[ctx->fifo_n].slot = slot; ctx->fifo_n++;
  if (!params->is_temporal) ctx->lane_next++;
  return FLX_OK;
}

extern "C" flx_status flx_frame_begin(flx_context *ctx, const flx_frame_params *params, int format) {
  if (!ctx) return FLX_ERR_INVALID;
  return frame_begin(ctx, params, format, NOT_GATHERED);
}

/* The frame loop over a communicator: every rank begins the same frame (its own tile_index of tile_count = the communicator's
 * size); frames alternate between the two lanes like flx_frame_begin's, each lane gathering over its own communicator, so the
 * kernels of frame k + 1 fill the CUs that the tails of frame k's kernels leave idle ON EVERY RANK.  root < 0: every rank's
 * flx_frame_end hands out the whole frame; root >= 0: that rank's does, the others get 0 bytes. */
extern "C" flx_status flx_frame_begin_gathered(flx_context *ctx, const flx_frame_params *params, int format, int root) {
  if (!ctx) return FLX_ERR_INVALID;
  return frame_begin(ctx, params, format, root < 0 ? -1 : root);
}

extern "C" flx_status flx_frame_end(flx_context *ctx, const void **pixels, size_t *bytes, float *gpu_ms) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->fifo_n == 0) return fail(ctx, FLX_ERR_INVALID, "flx_frame_end: no frame in flight");
  flx_context *lane = ctx->fifo[0].lane;
  const int k = ctx->fifo[0].slot;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  if (lane == ctx && ctx->sv_pending[k].valid) { flx_status ss = server_take(ctx, k); if (ss) return ss; }
  if (ctx->fifo_n == 1 && ctx->sv_running) {                  /* the loop runs empty: the launch is told to end (the next flx_frame_begin starts another) */
    __atomic_store_n(&ctx->h_sv_mail->stopAfter, ctx->sv_next_seq - 1u, __ATOMIC_RELEASE);
    ctx->sv_running = false;
  }
  if (!(__atomic_load_n(&ctx->h_dev_error[0], __ATOMIC_ACQUIRE) != 0u && lane == ctx)) {
    FLX_HIP(ctx, hipEventSynchronize(lane->slot_host[k] ? lane->ev_slot_done[k] : lane->ev_slot_traced[k]));
  }
  ctx->fifo[0] = ctx->fifo[1]; ctx->fifo[1] = ctx->fifo[2]; ctx->fifo_n--;
  lane->frames_ended++;
  { flx_status es = flx_check_device_error(ctx); if (es) return es; }
  if (gpu_ms) {
    float ms = lane->slot_latency_ms[k];                      /* a frame of the frame server: there are no events inside its launch — the time from its post to the launch's word */
    if (ms < 0.f) FLX_HIP(ctx, hipEventElapsedTime(&ms, lane->ev_slot_start[k], lane->ev_slot_traced[k]));
    *gpu_ms = ms;
  }
  if (pixels) *pixels = lane->slot_host[k] ? lane->h_slot[k] : (lane->slot_dev_ptr[k] ? lane->slot_dev_ptr[k] : (const void *)lane->d_slot[k]);
  if (bytes) *bytes = lane->slot_bytes[k];
  return FLX_OK;
}

extern "C" flx_status flx_frame_host_slots(flx_context *ctx, const void *slots[4], int *last_begun) {
  if (!ctx || !slots) return FLX_ERR_INVALID;
  for (int i = 0; i < 2; i++) { slots[i] = ctx->h_slot[i]; slots[2 + i] = ctx->twin ? ctx->twin->h_slot[i] : nullptr; }
  if (last_begun) {
    *last_begun = -1;
    if (ctx->fifo_n > 0) {
      const auto &f = ctx->fifo[ctx->fifo_n - 1];
      if (f.lane->slot_host[f.slot]) *last_begun = (f.lane == ctx ? 0 : 2) + f.slot;
    }
  }
  return FLX_OK;
}

extern "C" flx_status flx_sync(flx_context *ctx) {
  if (!ctx) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  { flx_status ss = flx_server_stop(ctx); if (ss) return ss; }
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->twin) FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream));
  return flx_check_device_error(ctx);
}

extern "C" flx_status flx_set_stream(flx_context *ctx, void *hip_stream) {
  if (!ctx) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  if (hip_stream) {                      /* kernels of this context launched on another GPU's stream would fail one by one, later and less clearly */
    hipDevice_t dev = 0;
    if (hipStreamGetDevice((hipStream_t)hip_stream, &dev) == hipSuccess && (int)dev != ctx->device) {
      char msg[160];
      snprintf(msg, sizeof msg, "flx_set_stream: the stream belongs to device %d, the context to device %d", (int)dev, ctx->device);
      return fail(ctx, FLX_ERR_INVALID, msg);
    }
  }
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  ctx->timed = false;
  return FLX_OK;
}

extern "C" flx_status flx_set_counters_enabled(flx_context *ctx, int enabled) {
  if (!ctx) return FLX_ERR_INVALID;
  ctx->counters_enabled = enabled != 0;
  return FLX_OK;
}

extern "C" flx_status flx_get_counters(flx_context *ctx, flx_counters *out) {
  if (!ctx || !out) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long host_cnt[8];
  FLX_HIP(ctx, hipMemcpy(host_cnt, ctx->d_counters, sizeof host_cnt, hipMemcpyDeviceToHost));
  memcpy(out, host_cnt, sizeof host_cnt);
  return FLX_OK;
}

/* tail profile of the last counted frame's bounce-0 walk kernel (diagnostics): for k = 0 .. 11, out[3k .. 3k+2] = sum / count / max over
 * the walk workgroups of the cycles (since the workgroup's start) at which its walks in flight first numbered <= 2^k; out[36..38] the
 * same for the moment the workgroup found the queue dry; out[39] = the longest workgroup lifetime */
extern "C" flx_status flx_get_tail_diag(flx_context *ctx, uint64_t out[40]) {
  if (!ctx || !out) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FLX_HIP(ctx, hipMemcpy(out, ctx->d_counters + 40, 40 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return FLX_OK;
}

extern "C" flx_status flx_get_diag(flx_context *ctx, uint64_t out[32]) {
  if (!ctx || !out) return FLX_ERR_INVALID;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FLX_HIP(ctx, hipMemcpy(out, ctx->d_counters + 8, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return FLX_OK;
}

extern "C" flx_status flx_last_frame_ms(flx_context *ctx, float *frame_ms, float *trace_kernel_ms) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!ctx->timed) return fail(ctx, FLX_ERR_INVALID, "flx_last_frame_ms: no frame rendered yet");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  FLX_HIP(ctx, hipEventSynchronize(ctx->ev_frame1));
  float a = 0.f, b = 0.f;
  FLX_HIP(ctx, hipEventElapsedTime(&a, ctx->ev_frame0, ctx->ev_frame1));
  FLX_HIP(ctx, hipEventElapsedTime(&b, ctx->ev_k0, ctx->ev_k1));
  if (frame_ms) *frame_ms = a;
  if (trace_kernel_ms) *trace_kernel_ms = b;
  return FLX_OK;
}

extern "C" flx_status flx_debug_math(flx_context *ctx, int fn, const float *a, const float *b, float *out, uint32_t n) {
  if (!ctx || !a || !out) return FLX_ERR_INVALID;
  if (n == 0) return FLX_OK;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  float *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
  FLX_HIP(ctx, hipMalloc(&d_a, (size_t)n * 4));
  FLX_HIP(ctx, hipMalloc(&d_o, (size_t)n * 4));
  FLX_HIP(ctx, hipMemcpy(d_a, a, (size_t)n * 4, hipMemcpyHostToDevice));
  if (b) {
    FLX_HIP(ctx, hipMalloc(&d_b, (size_t)n * 4));
    FLX_HIP(ctx, hipMemcpy(d_b, b, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  launch_debug_math(fn, d_a, d_b, d_o, n, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FLX_HIP(ctx, hipMemcpy(out, d_o, (size_t)n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_a); (void)hipFree(d_o); if (d_b) (void)hipFree(d_b);
  return FLX_OK;
}

struct DeviceScratch {                 /* a device buffer of a debug hook */
  void *p = nullptr;
  ~DeviceScratch() { if (p) (void)hipFree(p); }
};

extern "C" flx_status flx_debug_intersect(flx_context *ctx, int fn, const float *in, float *out, uint32_t n) {
  if (!ctx || !in || !out) return FLX_ERR_INVALID;
  if (fn < 0 || fn > 5) return fail(ctx, FLX_ERR_INVALID, "flx_debug_intersect: fn 0 .. 5");
  if (n == 0) return FLX_OK;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nin = (size_t)n * ((fn == 2 || fn == 5) ? 13u : 16u), nout = (size_t)n * ((fn == 0 || fn == 3) ? 3u : 1u);
  DeviceScratch d_in, d_out;                               /* freed on every way out */
  FLX_HIP(ctx, hipMalloc(&d_in.p, nin * 4));
  FLX_HIP(ctx, hipMalloc(&d_out.p, nout * 4));
  FLX_HIP(ctx, hipMemcpy(d_in.p, in, nin * 4, hipMemcpyHostToDevice));
  launch_debug_intersect(fn, (const float *)d_in.p, (float *)d_out.p, n, ctx->stream);
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  FLX_HIP(ctx, hipMemcpy(out, d_out.p, nout * 4, hipMemcpyDeviceToHost));
  return FLX_OK;
}

extern "C" flx_status flx_debug_walk(flx_context *ctx, int variant, const float *rays, float *out, uint32_t n) {
  if (!ctx || !rays || !out) return FLX_ERR_INVALID;
  if (variant < 0 || variant > 2) return fail(ctx, FLX_ERR_INVALID, "flx_debug_walk: variant 0 .. 2");
  if (n == 0) return FLX_OK;
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_frame_params p;
  memset(&p, 0, sizeof p);
  p.width = p.height = 8; p.samples = 1; p.max_reflections = 1; p.texture_width = 1;
  DeviceScene sc; DeviceFrame fr;
  flx_status st = flx_make_frame(ctx, &p, sc, fr);
  if (st != FLX_OK) return st;
  DeviceScratch d_in, d_out;                               /* freed on every way out */
  FLX_HIP(ctx, hipMalloc(&d_in.p, (size_t)n * 7 * 4));
  FLX_HIP(ctx, hipMalloc(&d_out.p, (size_t)n * 8 * 4));
  FLX_HIP(ctx, hipMemcpy(d_in.p, rays, (size_t)n * 7 * 4, hipMemcpyHostToDevice));
  const bool ok = launch_debug_walk(variant, sc, (const float *)d_in.p, (float *)d_out.p, n, ctx->stream);
  if (ok) {
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FLX_HIP(ctx, hipMemcpy(out, d_out.p, (size_t)n * 8 * 4, hipMemcpyDeviceToHost));
  }
  return ok ? FLX_OK : fail(ctx, FLX_ERR_INVALID, "flx_debug_walk: this scene does not have that walk (variant 2 needs the lockstep copy: at most 128 entries in one object space)");
}

extern "C" flx_status flx_device_info(flx_context *ctx, char *name, uint32_t name_len, uint32_t *compute_units) {
  if (!ctx) return FLX_ERR_INVALID;
  if (name && name_len) { snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName); }
  if (compute_units) *compute_units = (uint32_t)ctx->prop.multiProcessorCount;
  return FLX_OK;
}
