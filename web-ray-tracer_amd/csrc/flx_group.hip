/*
 * flx_group.hip — one frame on several GPUs: row strips dealt round robin to the contexts, scene replicated,
 * ONE exchange step per frame or batch of frames — ncclAllGather (RCCL, over xGMI) of the packed strips on the contexts'
 * streams — and a kernel that puts the gathered rows in image order.  The reference has nothing like it (one WebGL2 context,
 * modules/pathtracerWGL2.js:60-68); the boundary it sits behind is still the renderer object of flexlight.js:106-129.
 *
 * Two ways to hold the communicator, the same gather code behind both:
 *   flx_comm_init_rank   one process per GPU (bench.py under torch.distributed.run): ncclCommInitRank from an id rank 0 made
 *   flx_group_create     one process, N contexts (the JavaScript host — Node is one process): ncclCommInitAll
 * A group whose device list names one GPU more than once (the rehearsal on a one-GPU box: RCCL refuses two ranks on a device)
 * exchanges the strips with device-to-device copies instead; everything else — tile policy, packing, reassembly — is the same.
 */
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>

#include "flx_context.h"

using namespace flx;

#define FLX_NCCL(ctx, expr)                                                                   \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      (ctx)->err = std::string(#expr) + ": " + ncclGetErrorString(r_);                         \
      return FLX_ERR_DEVICE;                                                                  \
    }                                                                                         \
  } while (0)

namespace {

/* rows of the frame the tile policy gives rank r of n */
__host__ __device__ inline uint32_t rows_of_rank(uint32_t height, uint32_t tile_rows, uint32_t n, uint32_t r) {
  const uint32_t strips = (height + tile_rows - 1u) / tile_rows;
  if (r >= strips) return 0u;
  const uint32_t mine = (strips - r + n - 1u) / n;               /* strips s = r, r + n, ... below `strips` */
  uint32_t rows = mine * tile_rows;
  if ((strips - 1u) % n == r) rows -= strips * tile_rows - height;      /* the last strip may be cut by the frame's edge */
  return rows;
}
inline uint32_t rows_padded(uint32_t height, uint32_t tile_rows, uint32_t n) {
  const uint32_t strips = (height + tile_rows - 1u) / tile_rows;
  return ((strips + n - 1u) / n) * tile_rows;
}

/* gathered = [rank][slot words]; rank r's slot starts with its strips packed tight: float4[frames][rows_r][width].
 * out = float4[frames][height][width].  One thread per float4. */
__global__ __launch_bounds__(256) void k_reassemble(const float4 *__restrict__ gathered, float4 *__restrict__ out, uint32_t width, uint32_t height,
                                                    uint32_t frames, uint32_t tile_rows, uint32_t n, size_t slot) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  const size_t total = (size_t)frames * height * width;
  if (i >= total) return;
  const uint32_t x = (uint32_t)(i % width);
  const size_t row = i / width;
  const uint32_t y = (uint32_t)(row % height), f = (uint32_t)(row / height);
  const uint32_t strip = y / tile_rows, r = strip % n;
  const uint32_t k = (strip / n) * tile_rows + (y - strip * tile_rows);
  const uint32_t rows_r = rows_of_rank(height, tile_rows, n, r);
  out[i] = gathered[(size_t)r * slot + ((size_t)f * rows_r + k) * width + x];
}

/* the same for the five RGBA8 render targets of a filter frame: rank r's slot = uint32[5][rows_r][width] */
__global__ __launch_bounds__(256) void k_reassemble_planes(const uint32_t *__restrict__ gathered, uint32_t *__restrict__ out, uint32_t width, uint32_t height,
                                                           uint32_t tile_rows, uint32_t n, size_t slot) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  const size_t plane = (size_t)height * width;
  if (i >= 5u * plane) return;
  const uint32_t x = (uint32_t)(i % width);
  const size_t row = i / width;
  const uint32_t y = (uint32_t)(row % height), p = (uint32_t)(row / height);
  const uint32_t strip = y / tile_rows, r = strip % n;
  const uint32_t k = (strip / n) * tile_rows + (y - strip * tile_rows);
  const uint32_t rows_r = rows_of_rank(height, tile_rows, n, r);
  out[i] = gathered[(size_t)r * slot + ((size_t)p * rows_r + k) * width + x];
}

/* the same for frames that travel as the canvas' RGBA8 (a quarter of the bytes): rank r's slot = uint32[frames][rows_r][width], out = uint32[frames][height][width] */
__global__ __launch_bounds__(256) void k_reassemble8(const uint32_t *__restrict__ gathered, uint32_t *__restrict__ out, uint32_t width, uint32_t height,
                                                     uint32_t frames, uint32_t tile_rows, uint32_t n, size_t slot) {
  const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  const size_t total = (size_t)frames * height * width;
  if (i >= total) return;
  const uint32_t x = (uint32_t)(i % width);
  const size_t row = i / width;
  const uint32_t y = (uint32_t)(row % height), f = (uint32_t)(row / height);
  const uint32_t strip = y / tile_rows, r = strip % n;
  const uint32_t k = (strip / n) * tile_rows + (y - strip * tile_rows);
  const uint32_t rows_r = rows_of_rank(height, tile_rows, n, r);
  out[i] = gathered[(size_t)r * slot + ((size_t)f * rows_r + k) * width + x];
}

struct Share {                   /* what one context contributes to a gathered frame / batch */
  uint32_t width, height, tile_rows, n, frames;
  size_t slot;                   /* float4 (radiance) or uint32 x 4 (planes, counted in float4 units too) per rank in the exchange */
  bool planes;
  bool rgba8 = false;            /* the presenter wants the canvas' RGBA8: the strips are quantised where they were traced and travel as uint32 texels (slot of them per rank) */
};

flx_status check_params(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, int rank, int size, Share &sh) {
  if (!params || n_frames < 1u || n_frames > FLX_MAX_BATCH_FRAMES) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: 1 .. 32 frames");
  if (params->is_temporal) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: temporal frames keep their history in one context and are not sharded");
  if (params->use_filter && n_frames != 1u) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: filter frames are rendered one by one");
  if (params->tile_rows == 0u) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: tile_rows must be positive");
  for (uint32_t i = 0; i < n_frames; i++)
    if (params[i].tile_index != (uint32_t)rank || params[i].tile_count != (uint32_t)size)
      return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: tile_index / tile_count must be this context's rank / the communicator's size");
  sh.width = params->width; sh.height = params->height; sh.tile_rows = params->tile_rows; sh.n = (uint32_t)size; sh.frames = n_frames;
  sh.planes = params->use_filter != 0;
  const size_t rmax = rows_padded(sh.height, sh.tile_rows, sh.n);
  sh.slot = sh.planes ? (5u * rmax * sh.width + 3u) / 4u : (size_t)n_frames * rmax * sh.width;
  return FLX_OK;
}

flx_status ensure_f4(flx_context *ctx, float4 **buf, size_t *cap, size_t n) { return flx_ensure_pixels(ctx, buf, cap, n ? n : 1); }

/* step 1: this context's strips -> ctx->d_send (enqueued, no host sync) */
flx_status trace_share(flx_context *ctx, const flx_frame_params *params, const Share &sh) {
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_status s;
  if ((s = ensure_f4(ctx, &ctx->d_send, &ctx->send_capacity, sh.slot))) return s;
  if ((s = ensure_f4(ctx, &ctx->d_recv, &ctx->recv_capacity, sh.slot * sh.n))) return s;
  if (sh.planes) return flx_render_planes_device(ctx, params, ctx->d_send);
  if ((s = flx_render_batch_device(ctx, params, sh.frames, ctx->d_send))) return s;
  if (sh.rgba8) {
    /* floor(clamp(x) * 255 + 0.5) per channel — flx_present's store, texel by texel: the same bytes whether a strip is quantised here or the frame after
     * the exchange.  (The padding rows of a rank with a strip less hold whatever was there: nobody reads them.) */
    if ((s = ensure_f4(ctx, &ctx->d_send8, &ctx->send8_capacity, (sh.slot + 3u) / 4u))) return s;
    launch_quantize(ctx->d_send, (uint32_t *)ctx->d_send8, sh.slot, ctx->stream);
    FLX_HIP(ctx, hipGetLastError());
  }
  return FLX_OK;
}
/* what a context sends (and how many 4-byte words per rank): its float4 strips, the five planes of a filter frame, or its RGBA8 strips */
inline const void *share_send(const flx_context *ctx, const Share &sh) { return sh.rgba8 ? (const void *)ctx->d_send8 : (const void *)ctx->d_send; }
inline size_t share_words(const Share &sh) { return sh.rgba8 ? sh.slot : sh.slot * 4u; }
inline void *share_recv(flx_context *ctx, const Share &sh, int r) { return (char *)ctx->d_recv + (size_t)r * share_words(sh) * 4u; }

/* step 3: gathered strips -> frames in image order (+ the denoise chain for filter frames), on the context's stream */
flx_status finish_share(flx_context *ctx, const flx_frame_params *params, const Share &sh, void *d_frames) {
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  if (sh.planes) {
    flx_status s;
    const size_t words = 5u * (size_t)sh.height * sh.width;
    if ((s = ensure_f4(ctx, &ctx->d_gplanes, &ctx->gplanes_capacity, (words + 3u) / 4u))) return s;
    hipLaunchKernelGGL(k_reassemble_planes, dim3((uint32_t)((words + 255u) / 256u)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_recv,
                       (uint32_t *)ctx->d_gplanes, sh.width, sh.height, sh.tile_rows, sh.n, sh.slot * 4u);
    FLX_HIP(ctx, hipGetLastError());
    flx_frame_params whole = *params;
    whole.tile_rows = whole.tile_index = whole.tile_count = 0;
    return flx_filter_planes_enqueue(ctx, &whole, ctx->d_gplanes, d_frames, false);      /* the frame began with the trace: that stamp stays */
  }
  const size_t total = (size_t)sh.frames * sh.height * sh.width;
  if (sh.rgba8)
    hipLaunchKernelGGL(k_reassemble8, dim3((uint32_t)((total + 255u) / 256u)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_recv, (uint32_t *)d_frames, sh.width,
                       sh.height, sh.frames, sh.tile_rows, sh.n, sh.slot);
  else
  hipLaunchKernelGGL(k_reassemble, dim3((uint32_t)((total + 255u) / 256u)), dim3(256), 0, ctx->stream, ctx->d_recv, (float4 *)d_frames, sh.width, sh.height,
                     sh.frames, sh.tile_rows, sh.n, sh.slot);
  FLX_HIP(ctx, hipGetLastError());
  FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));         /* frame time = first kernel .. last byte of the gathered frame */
  return FLX_OK;
}

}  // namespace

/* ---- one process per GPU ------------------------------------------------------------------------------------------------ */
static_assert(FLX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "flexlight_hip.h and rccl.h disagree on the size of a communicator id");

extern "C" flx_status flx_comm_unique_id(uint8_t *id) {
  if (!id) return FLX_ERR_INVALID;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) { g_create_error = "ncclGetUniqueId failed"; return FLX_ERR_DEVICE; }
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return FLX_OK;
}

extern "C" flx_status flx_comm_init_rank(flx_context *ctx, const uint8_t *id, int n_ranks, int rank) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return flx_fail(ctx, FLX_ERR_INVALID, "flx_comm_init_rank: need an id and 0 <= rank < n_ranks");
  if (ctx->comm) return flx_fail(ctx, FLX_ERR_INVALID, "flx_comm_init_rank: the context already belongs to a communicator");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  FLX_NCCL(ctx, ncclCommInitRank(&comm, n_ranks, u, rank));
  ctx->comm = (flx_nccl_comm)comm; ctx->comm_rank = rank; ctx->comm_size = n_ranks; ctx->comm_owned = true;
  /* a second communicator over the same ranks for the frame loop's second lane (flx_frame_begin_gathered): two frames in flight
   * gather on two streams, and collectives of ONE communicator must not be in flight on two streams at once.  ncclCommSplit is
   * collective: every rank makes it here, in the same order. */
  ncclComm_t second = nullptr;
  FLX_NCCL(ctx, ncclCommSplit(comm, 0, rank, &second, nullptr));
  ctx->comm_twin = (flx_nccl_comm)second;
  return FLX_OK;
}

extern "C" int flx_comm_count(const flx_context *ctx) {
  if (!ctx || !ctx->comm) return 0;
  int n = 0;
  return ncclCommCount((ncclComm_t)ctx->comm, &n) == ncclSuccess ? n : -1;
}

extern "C" flx_status flx_comm_destroy(flx_context *ctx) {
  if (!ctx) return FLX_ERR_INVALID;
  if (ctx->comm && ctx->comm_owned) {
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->twin) { FLX_HIP(ctx, hipStreamSynchronize(ctx->twin->stream)); ctx->twin->comm = nullptr; }
    if (ctx->comm_twin) FLX_NCCL(ctx, ncclCommDestroy((ncclComm_t)ctx->comm_twin));
    FLX_NCCL(ctx, ncclCommDestroy((ncclComm_t)ctx->comm));
  }
  ctx->comm = nullptr; ctx->comm_twin = nullptr; ctx->comm_rank = 0; ctx->comm_size = 1; ctx->comm_owned = false;
  return FLX_OK;
}

/* trace, exchange, reassembly — all enqueued on the context's stream, nothing waits on the host.  root < 0: ncclAllGather, every rank
 * ends up with the frames.  root >= 0: only that rank receives (one ncclSend per rank and n ncclRecv on the root inside one
 * ncclGroupStart / End: 1/n of the all-gather's bytes on every link but the root's), reassembles and — filter frames — runs the
 * denoise chain; the other ranks are done when their strips are sent. */
flx_status flx_gather_enqueue(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, int root, void *d_frames, bool rgba8) {
  if (!ctx->comm) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: the context belongs to no communicator (flx_comm_init_rank)");
  if (root >= ctx->comm_size) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: root is not a rank of the communicator");
  const bool receiver = root < 0 || root == ctx->comm_rank;
  if (receiver && !d_frames) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: output pointer is NULL on a rank that receives the frames");
  Share sh;
  flx_status s = check_params(ctx, params, n_frames, ctx->comm_rank, ctx->comm_size, sh);
  if (s) return s;
  if (rgba8 && sh.planes) return flx_fail(ctx, FLX_ERR_INVALID, "gathered render: filter frames are gathered as their five render targets, not as RGBA8 (present the float frame)");
  sh.rgba8 = rgba8;
  if ((s = trace_share(ctx, params, sh))) return s;
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  const size_t words = share_words(sh);          /* 4-byte words per rank: ncclFloat is only the element size here */
  if (root < 0) {
    FLX_NCCL(ctx, ncclAllGather(share_send(ctx, sh), ctx->d_recv, words, ncclFloat, comm, ctx->stream));
  } else {
    FLX_NCCL(ctx, ncclGroupStart());
    ncclResult_t rc = ncclSend(share_send(ctx, sh), words, ncclFloat, root, comm, ctx->stream);
    if (receiver)
      for (int r = 0; r < ctx->comm_size && rc == ncclSuccess; r++) rc = ncclRecv(share_recv(ctx, sh, r), words, ncclFloat, r, comm, ctx->stream);
    const ncclResult_t rc2 = ncclGroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess) { ctx->err = std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(rc != ncclSuccess ? rc : rc2); return FLX_ERR_DEVICE; }
  }
  ctx->last_gather_root = root;
  if (!receiver) {
    FLX_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
    return FLX_OK;
  }
  return finish_share(ctx, params, sh, d_frames);
}

extern "C" flx_status flx_render_gathered_device(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, void *d_frames) {
  if (!ctx) return FLX_ERR_INVALID;
  return flx_gather_enqueue(ctx, params, n_frames, -1, d_frames, false);
}

extern "C" flx_status flx_render_gathered_root_device(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, int root, void *d_frames) {
  if (!ctx) return FLX_ERR_INVALID;
  if (root < 0) return flx_fail(ctx, FLX_ERR_INVALID, "flx_render_gathered_root_device: root must be a rank (flx_render_gathered_device gathers on every rank)");
  return flx_gather_enqueue(ctx, params, n_frames, root, d_frames, false);
}

/* The frames as the canvas' RGBA8 (flx_present's bytes): every rank quantises its strips where it traced them, and a quarter of the bytes travels —
 * 8.3 MB of a 1080p frame in all instead of 33 MB, 33 instead of 133 MB at 4K.  root < 0: all-gather; root >= 0: that rank alone receives. */
extern "C" flx_status flx_render_gathered_rgba8_device(flx_context *ctx, const flx_frame_params *params, uint32_t n_frames, int root, void *d_frames_rgba8) {
  if (!ctx) return FLX_ERR_INVALID;
  return flx_gather_enqueue(ctx, params, n_frames, root, d_frames_rgba8, true);
}

/* ---- one process, several GPUs ------------------------------------------------------------------------------------------- */
struct flx_group {
  int gather_root = 0;                  /* 0 (default): only context 0, the one that hands the frame out, receives the strips; -1: every context (all-gather) */
  std::vector<flx_context *> ctx;
  std::vector<ncclComm_t> comms;        /* empty: the contexts share a device, strips are exchanged by copies */
  std::vector<hipEvent_t> traced;       /* per context: its strips are in d_send */
  std::vector<hipEvent_t> gathered;     /* per context: every copy INTO its d_recv has been enqueued and this marks their end */
  std::string err;
  /* the frame loop (flx_group_frame_begin / _end): every context's frame server resolves its strips straight into ONE image — in context 0's memory
   * (FLX_FRAME_DEVICE) or in pinned host memory (FLX_FRAME_FLOAT) — so a frame needs no exchange, no reassembly and no copy */
  int lanes = 3;
  bool peer_ok = true;                  /* every context's GPU can write context 0's memory */
  struct Target { void *base = nullptr; bool host = false; size_t pixels = 0; uint32_t n = 0; size_t texel = sizeof(float4); /* bytes per pixel: float4, or 4 for the canvas' RGBA8 */ };
  Target target;                        /* target.n images of target.pixels float4 */
  std::vector<Target> retired;          /* targets of an earlier frame shape that frames in flight still live in */
  flx_frame_params shape = {};          /* what the target was made for */
  std::vector<uint64_t> seen_version;   /* per context: its scene_version at the group's last flx_group_frame_begin */
  std::vector<hipStream_t> strip_copy;  /* per context: the stream its strips of a frame of the lanes (InFlight::kind 2) are copied into the frame's image on */
  int shape_format = -1;
  struct InFlight { int kind; /* 1: through the servers, 0: rendered synchronously (frames the server does not take), 2: on the contexts' two lanes (a scene that moves): the strips are
                               * copied into the image when the frame is taken */ const void *pixels; size_t bytes; float ms; flx_frame_params params; uint32_t tile_rows; };
  InFlight fifo[3] = {};
  int fifo_n = 0;
  float4 *h_sync[3] = { nullptr, nullptr, nullptr };      /* pinned: frames rendered synchronously (as many as may be in flight) */
  size_t h_sync_pixels[3] = { 0, 0, 0 };
  uint32_t h_sync_next = 0;
};

static thread_local std::string g_group_error;

extern "C" const char *flx_group_last_error(const flx_group *g) { return g ? g->err.c_str() : g_group_error.c_str(); }

static void group_free_target(flx_group *g, flx_group::Target &t) {
  if (!t.base) return;
  if (t.host) (void)hipHostFree(t.base);
  else { (void)hipSetDevice(g->ctx[0]->device); (void)hipFree(t.base); }
  t = flx_group::Target();
}

extern "C" void flx_group_destroy(flx_group *g) {
  if (!g) return;
  for (flx_context *c : g->ctx) if (c) { (void)hipSetDevice(c->device); (void)flx_frame_target_set(c, nullptr, 0); }      /* (ends the servers' launches) */
  if (!g->ctx.empty() && g->ctx[0]) {
    group_free_target(g, g->target);
    for (auto &t : g->retired) group_free_target(g, t);
    for (float4 *h : g->h_sync) if (h) (void)hipHostFree(h);
  }
  for (size_t r = 0; r < g->ctx.size(); r++) {
    if (!g->ctx[r]) continue;
    (void)hipSetDevice(g->ctx[r]->device);
    if (r < g->strip_copy.size() && g->strip_copy[r]) { (void)hipStreamSynchronize(g->strip_copy[r]); (void)hipStreamDestroy(g->strip_copy[r]); }
    (void)hipStreamSynchronize(g->ctx[r]->stream);
    if (r < g->comms.size() && g->comms[r]) (void)ncclCommDestroy(g->comms[r]);
    g->ctx[r]->comm = nullptr;
    if (r < g->traced.size() && g->traced[r]) (void)hipEventDestroy(g->traced[r]);
    if (r < g->gathered.size() && g->gathered[r]) (void)hipEventDestroy(g->gathered[r]);
    flx_context_destroy(g->ctx[r]);
  }
  delete g;
}

extern "C" flx_status flx_group_create(int n, const int *devices, flx_group **out) {
  if (!out) { g_group_error = "flx_group_create: out is NULL"; return FLX_ERR_INVALID; }
  *out = nullptr;
  if (n < 1 || n > 64 || !devices) { g_group_error = "flx_group_create: 1 .. 64 devices"; return FLX_ERR_INVALID; }
  flx_group *g = new flx_group();
  bool distinct = true;
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (devices[i] == devices[j]) distinct = false;
  for (int r = 0; r < n; r++) {
    flx_context *c = nullptr;
    flx_status s = flx_context_create(devices[r], &c);
    if (s) { g_group_error = std::string("flx_group_create: ") + flx_last_error(nullptr); flx_group_destroy(g); return s; }
    c->comm_rank = r; c->comm_size = n;
    g->ctx.push_back(c);
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
      g_group_error = "flx_group_create: hipEventCreate failed"; flx_group_destroy(g); return FLX_ERR_DEVICE;
    }
    g->traced.push_back(a); g->gathered.push_back(b);
  }
  if (distinct && n > 1) {
    g->comms.assign((size_t)n, nullptr);
    ncclResult_t rc = ncclCommInitAll(g->comms.data(), n, devices);
    if (rc != ncclSuccess) {
      g_group_error = std::string("flx_group_create: ncclCommInitAll: ") + ncclGetErrorString(rc);
      g->comms.clear(); flx_group_destroy(g); return FLX_ERR_DEVICE;
    }
    for (int r = 0; r < n; r++) g->ctx[r]->comm = (flx_nccl_comm)g->comms[r];
    /* the frame loop's contexts write their strips into context 0's memory: a peer mapping where the devices have one (else that loop hands out host frames only) */
    for (int r = 1; r < n; r++) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[r], devices[0]) == hipSuccess && can) {
        (void)hipSetDevice(devices[r]);
        const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
        (void)hipGetLastError();
      }
      if (!can) g->peer_ok = false;
    }
  }
  /* A device named more than once (the rehearsal of a group on a one-GPU box): the frame loop's servers are persistent launches, and two of them want the
   * whole GPU each — the second would not start before the first ends, which it never does while the host waits for both.  Their launches take an equal
   * part of the CUs each (what flx_debug_set_server_groups sets by hand), so that they run beside each other as they do on two GPUs. */
  if (!distinct) {
    for (int r = 0; r < n; r++) {
      uint32_t sharing = 0;
      for (int q = 0; q < n; q++) if (devices[q] == devices[r]) sharing++;
      const uint32_t cus = (uint32_t)g->ctx[r]->prop.multiProcessorCount;
      if (sharing > 1u) g->ctx[r]->sv_groups = cus / sharing ? cus / sharing : 1u;
    }
  }
  *out = g;
  return FLX_OK;
}

extern "C" int flx_group_size(const flx_group *g) { return g ? (int)g->ctx.size() : 0; }
extern "C" flx_context *flx_group_context(flx_group *g, int rank) { return (g && rank >= 0 && rank < (int)g->ctx.size()) ? g->ctx[rank] : nullptr; }
extern "C" int flx_group_uses_rccl(const flx_group *g) { return g && !g->comms.empty(); }
extern "C" flx_status flx_group_set_gather(flx_group *g, int to_root) {
  if (!g) return FLX_ERR_INVALID;
  g->gather_root = to_root ? 0 : -1;
  return FLX_OK;
}

#define FLX_GROUP_EACH(g, call)                                                           \
  do {                                                                                    \
    if (!(g)) return FLX_ERR_INVALID;                                                     \
    for (flx_context *c : (g)->ctx) {                                                     \
      flx_status s_ = (call);                                                             \
      if (s_) { (g)->err = flx_last_error(c); return s_; }                                \
    }                                                                                     \
    return FLX_OK;                                                                        \
  } while (0)

extern "C" flx_status flx_group_scene_upload(flx_group *g, const float *geometry, const float *attributes, uint32_t n_entries_padded, const int32_t *ids, uint32_t n_ids) {
  FLX_GROUP_EACH(g, flx_scene_upload(c, geometry, attributes, n_entries_padded, ids, n_ids));
}
extern "C" flx_status flx_group_transforms_upload(flx_group *g, const float *rotation, const float *shift, uint32_t n_transforms) {
  FLX_GROUP_EACH(g, flx_transforms_upload(c, rotation, shift, n_transforms));
}
extern "C" flx_status flx_group_lights_upload(flx_group *g, const float *lights, uint32_t n_lights) { FLX_GROUP_EACH(g, flx_lights_upload(c, lights, n_lights)); }
extern "C" flx_status flx_group_atlas_upload(flx_group *g, int which, const uint8_t *rgba, uint32_t width, uint32_t height) {
  FLX_GROUP_EACH(g, flx_atlas_upload(c, which, rgba, width, height));
}
extern "C" flx_status flx_group_scene_upload_view(flx_group *g, const flx_scene_view *scene) { FLX_GROUP_EACH(g, flx_scene_upload_view(c, scene)); }

/* n_frames frames (a batch; 1 = one frame) of a camera path on all contexts of the group; the frames arrive on the host. */
static flx_status group_render(flx_group *g, const flx_frame_params *params, uint32_t n_frames, uint32_t tile_rows, void *out_rgba, flx_counters *counters, bool rgba8) {
  if (!g) return FLX_ERR_INVALID;
  auto gfail = [&](flx_context *c, flx_status s) { g->err = c ? flx_last_error(c) : "flx_group_render: bad arguments"; return s; };
  if (!params || !out_rgba || n_frames < 1u || n_frames > FLX_MAX_BATCH_FRAMES || tile_rows == 0u) return gfail(nullptr, FLX_ERR_INVALID);
  const int n = (int)g->ctx.size();
  std::vector<std::vector<flx_frame_params>> p((size_t)n, std::vector<flx_frame_params>(params, params + n_frames));
  std::vector<Share> sh((size_t)n);
  flx_status s;
  for (int r = 0; r < n; r++) {
    for (auto &q : p[r]) { q.tile_rows = tile_rows; q.tile_index = (uint32_t)r; q.tile_count = (uint32_t)n; }
    if ((s = check_params(g->ctx[r], p[r].data(), n_frames, r, n, sh[r]))) return gfail(g->ctx[r], s);
    if (rgba8 && sh[r].planes) { g->err = "flx_group_render_rgba8: filter frames are gathered as their five render targets (present the float frame)"; return FLX_ERR_INVALID; }
    sh[r].rgba8 = rgba8;
  }
  /* 1. every context traces its strips (enqueued on its own stream; the GPUs run concurrently) */
  for (int r = 0; r < n; r++) {
    flx_context *c = g->ctx[r];
    if (counters) (void)flx_set_counters_enabled(c, 1);
    s = trace_share(c, p[r].data(), sh[r]);
    if (counters) (void)flx_set_counters_enabled(c, 0);
    if (s) return gfail(c, s);
    if (hipEventRecord(g->traced[r], c->stream) != hipSuccess) return gfail(nullptr, FLX_ERR_DEVICE);
  }
  /* 2. the exchange: one all-gather over RCCL, or (contexts on one device) plain copies ordered by events */
  const bool toRoot = g->gather_root == 0;      /* only context 0 hands the frame out: the others need not receive anything */
  if (!g->comms.empty()) {
    ncclResult_t rc = ncclGroupStart();
    for (int r = 0; r < n && rc == ncclSuccess; r++) {
      flx_context *c = g->ctx[r];
      (void)hipSetDevice(c->device);
      if (!toRoot) { rc = ncclAllGather(share_send(c, sh[r]), c->d_recv, share_words(sh[r]), ncclFloat, g->comms[r], c->stream); continue; }
      rc = ncclSend(share_send(c, sh[r]), share_words(sh[r]), ncclFloat, 0, g->comms[r], c->stream);
      if (r == 0)
        for (int q = 0; q < n && rc == ncclSuccess; q++) rc = ncclRecv(share_recv(c, sh[0], q), share_words(sh[0]), ncclFloat, q, g->comms[0], c->stream);
    }
    ncclResult_t rc2 = ncclGroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess) { g->err = std::string(toRoot ? "ncclSend / ncclRecv: " : "ncclAllGather: ") + ncclGetErrorString(rc != ncclSuccess ? rc : rc2); return FLX_ERR_DEVICE; }
  } else {
    for (int r = 0; r < (toRoot ? 1 : n); r++) {
      flx_context *c = g->ctx[r];
      (void)hipSetDevice(c->device);
      for (int q = 0; q < n; q++) {
        if (hipStreamWaitEvent(c->stream, g->traced[q], 0) != hipSuccess) return gfail(nullptr, FLX_ERR_DEVICE);
        if (hipMemcpyAsync(share_recv(c, sh[r], q), share_send(g->ctx[q], sh[q]), share_words(sh[r]) * 4u, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
          return gfail(nullptr, FLX_ERR_DEVICE);
      }
      if (hipEventRecord(g->gathered[r], c->stream) != hipSuccess) return gfail(nullptr, FLX_ERR_DEVICE);
    }
    /* a context's d_send may be overwritten by its next frame only after every receiving peer has copied it */
    for (int r = 0; r < n; r++) for (int q = 0; q < (toRoot ? 1 : n); q++) if (q != r && hipStreamWaitEvent(g->ctx[r]->stream, g->gathered[q], 0) != hipSuccess) return gfail(nullptr, FLX_ERR_DEVICE);
  }
  /* 3. context 0 puts the rows in image order (and runs the denoise chain of a filter frame); the frames go to the host */
  flx_context *c0 = g->ctx[0];
  const size_t pixels = (size_t)n_frames * params->height * params->width;
  if ((s = ensure_f4(c0, &c0->d_frames, &c0->frames_capacity, pixels))) return gfail(c0, s);
  if ((s = finish_share(c0, p[0].data(), sh[0], c0->d_frames))) return gfail(c0, s);
  (void)hipSetDevice(c0->device);
  if (hipMemcpyAsync(out_rgba, c0->d_frames, pixels * (rgba8 ? sizeof(uint32_t) : sizeof(float4)), hipMemcpyDeviceToHost, c0->stream) != hipSuccess) return gfail(nullptr, FLX_ERR_DEVICE);
  for (int r = 0; r < n; r++) if ((s = flx_sync(g->ctx[r]))) return gfail(g->ctx[r], s);
  if (counters) {
    memset(counters, 0, sizeof *counters);
    uint64_t *acc = (uint64_t *)counters;
    for (int r = 0; r < n; r++) {
      flx_counters one;
      if ((s = flx_get_counters(g->ctx[r], &one))) return gfail(g->ctx[r], s);
      const uint64_t *v = (const uint64_t *)&one;
      for (size_t k = 0; k < sizeof one / sizeof(uint64_t); k++) acc[k] += v[k];
    }
  }
  return FLX_OK;
}

extern "C" flx_status flx_group_render(flx_group *g, const flx_frame_params *params, uint32_t n_frames, uint32_t tile_rows, float *out_rgba, flx_counters *counters) {
  return group_render(g, params, n_frames, tile_rows, out_rgba, counters, false);
}
/* the frames as the canvas' RGBA8: every context quantises its strips, a quarter of the bytes is exchanged and copied out (flx_render_gathered_rgba8_device) */
extern "C" flx_status flx_group_render_rgba8(flx_group *g, const flx_frame_params *params, uint32_t n_frames, uint32_t tile_rows, uint8_t *out_rgba8, flx_counters *counters) {
  return group_render(g, params, n_frames, tile_rows, out_rgba8, counters, true);
}

/* ---- the group's frame loop ------------------------------------------------------------------------------------------------
 * The reference's loop never waits for the GPU (pathtracerWGL2.js:254-303: requestAnimationFrame, a frame per callback).  Here: flx_group_frame_begin posts
 * the frame to every context's frame server (flx_server.hip) and returns; each server renders its context's row strips and resolves them straight into the
 * ONE image the group owns — pinned host memory every GPU writes over its own PCIe link (FLX_FRAME_FLOAT), or context 0's device memory through the peer
 * mapping (FLX_FRAME_DEVICE).  flx_group_frame_end waits for every server's word and hands the image out: no exchange, no reassembly kernel, no copy, and no
 * host synchronisation with any stream.  Frames the server does not take (filter / temporal frames, scenes of <= 128 entries, strips that are no multiple of
 * 8 rows) are rendered synchronously by flx_group_render at flx_group_frame_begin and handed out by the matching _end. */
extern "C" flx_status flx_group_set_frame_lanes(flx_group *g, int lanes) {
  if (!g) return FLX_ERR_INVALID;
  if (lanes < 2 || lanes > 3) { g->err = "flx_group_set_frame_lanes: 2 or 3 frames in flight"; return FLX_ERR_INVALID; }
  if (g->fifo_n) { g->err = "flx_group_set_frame_lanes: frames are in flight"; return FLX_ERR_INVALID; }
  g->lanes = lanes;
  g->shape_format = -1;                 /* (the target is made again, with as many images) */
  return FLX_OK;
}
extern "C" int flx_group_frames_in_flight(const flx_group *g) { return g ? g->fifo_n : 0; }

static bool group_same_shape(const flx_frame_params &a, const flx_frame_params &b) {
  return a.width == b.width && a.height == b.height && a.samples == b.samples && a.tile_rows == b.tile_rows;
}

/* the target for frames of this shape and format exists and every context resolves into it */
static flx_status group_target(flx_group *g, const flx_frame_params *p, uint32_t tile_rows, int format) {
  const bool host = format != FLX_FRAME_DEVICE;
  const bool rgba8 = format == FLX_FRAME_RGBA8;
  const size_t texel = rgba8 ? sizeof(uint32_t) : sizeof(float4);
  const size_t pixels = (size_t)p->width * p->height;
  flx_frame_params want = *p; want.tile_rows = tile_rows;
  if (g->target.base && g->shape_format == format && group_same_shape(g->shape, want) && g->target.n == (uint32_t)g->lanes) return FLX_OK;
  const int n = (int)g->ctx.size();
  flx_status s;
  /* the servers end (frames in flight are completed in the old target, which lives until they are taken) */
  for (int r = 0; r < n; r++) if ((s = flx_frame_target_set(g->ctx[r], nullptr, 0))) { g->err = flx_last_error(g->ctx[r]); return s; }
  if (g->target.base) { if (g->fifo_n) g->retired.push_back(g->target); else group_free_target(g, g->target); g->target = flx_group::Target(); }
  flx_group::Target t;
  t.host = host; t.pixels = pixels; t.n = (uint32_t)g->lanes; t.texel = texel;
  (void)hipSetDevice(g->ctx[0]->device);
  const size_t bytes = (size_t)t.n * pixels * texel;
  const hipError_t e = host ? hipHostMalloc(&t.base, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) : hipMalloc(&t.base, bytes);
  if (e != hipSuccess) { g->err = std::string("flx_group_frame_begin: the frames' images: ") + hipGetErrorString(e); return FLX_ERR_DEVICE; }
  g->target = t; g->shape = want; g->shape_format = format;
  for (int r = 0; r < n; r++) {
    void *img[3] = { nullptr, nullptr, nullptr };
    for (uint32_t i = 0; i < t.n; i++) {
      void *at = (char *)t.base + (size_t)i * pixels * texel;
      if (host) {
        (void)hipSetDevice(g->ctx[r]->device);
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, at, 0) != hipSuccess) { g->err = "flx_group_frame_begin: the pinned frame has no address on a device of the group"; return FLX_ERR_DEVICE; }
        at = dp;
      }
      img[i] = at;
    }
    if ((s = rgba8 ? flx_frame_target_set8(g->ctx[r], img, t.n) : flx_frame_target_set(g->ctx[r], img, t.n))) { g->err = flx_last_error(g->ctx[r]); return s; }
  }
  return FLX_OK;
}

extern "C" flx_status flx_group_frame_begin(flx_group *g, const flx_frame_params *params, uint32_t tile_rows, int format) {
  if (!g) return FLX_ERR_INVALID;
  if (!params || tile_rows == 0u) { g->err = "flx_group_frame_begin: params and tile_rows"; return FLX_ERR_INVALID; }
  if (format != FLX_FRAME_FLOAT && format != FLX_FRAME_DEVICE && format != FLX_FRAME_RGBA8) { g->err = "flx_group_frame_begin: format is FLX_FRAME_FLOAT (the frame in pinned host memory), FLX_FRAME_RGBA8 (the canvas' bytes there) or FLX_FRAME_DEVICE (in context 0's device memory)"; return FLX_ERR_INVALID; }
  if (g->fifo_n >= g->lanes) { g->err = "flx_group_frame_begin: as many frames are in flight as the loop has lanes (flx_group_set_frame_lanes), take one with flx_group_frame_end first"; return FLX_ERR_INVALID; }
  const int n = (int)g->ctx.size();
  if (g->fifo_n == 0) for (auto &t : g->retired) group_free_target(g, t);
  if (g->fifo_n == 0) g->retired.clear();
  std::vector<flx_frame_params> p((size_t)n, *params);
  bool server = format != FLX_FRAME_DEVICE || g->peer_ok, moved = false;
  flx_status s;
  for (int r = 0; r < n; r++) {
    p[r].tile_rows = tile_rows; p[r].tile_index = (uint32_t)r; p[r].tile_count = (uint32_t)n;
    flx_context *c = g->ctx[r];
    if (c->frame_lanes != g->lanes && c->fifo_n == 0) c->frame_lanes = g->lanes;
    c->frame_chain = 3;                    /* (every frame the server can take, whatever its size: the target says where it goes; read by flx_frame_begin only) */
    if (!flx_frame_server_takes(c, &p[r])) server = false;
    /* a scene that changed since the frame before: lights and transforms that move travel with the frames of the servers' launches (flx_server.hip: VER); where they
     * do not fit a post, or something else was uploaded, the launches would end and start again around the frame (2.4 ms per frame on a rank's eighth of the dragon
     * frame against 1.4 on two lanes, tools/dynamic_scene_time.py): such a frame does not go to the servers */
    if (g->seen_version.size() != (size_t)n) g->seen_version.assign((size_t)n, 0);
    if (g->seen_version[(size_t)r] != 0 && g->seen_version[(size_t)r] != c->scene_version && format != FLX_FRAME_DEVICE && !flx_server_takes_moving_scene(c)) moved = true;      /* (a frame for context 0's memory stays with the servers) */
    g->seen_version[(size_t)r] = c->scene_version;
  }
  /* ... as floats on the contexts' two lanes (every lane keeps its own copy of the lights and transforms; nothing waits for a GPU here), its strips copied into the frame's
   * image when the frame is taken; the canvas' bytes through flx_group_render_rgba8 */
  if (moved) server = false;
  /* (... and so does every other float frame the servers do not take — a scene of <= 128 entries, strips that are no multiple of 8 rows —; filter and temporal frames need
   * the whole frame in one context: flx_group_render) */
  const bool lanes = !server && format == FLX_FRAME_FLOAT && !params->use_filter && !params->is_temporal;
  auto &slot = g->fifo[g->fifo_n];
  if (!server && format == FLX_FRAME_DEVICE) { g->err = "flx_group_frame_begin: FLX_FRAME_DEVICE takes only frames the frame server takes (flx_frame_server_takes) on GPUs that can write context 0's memory; FLX_FRAME_FLOAT takes every frame"; return FLX_ERR_INVALID; }
  if (!server) {
    /* a frame of another kind: the frames in flight stay where they are, this one is rendered now */
    const size_t pixels = (size_t)params->width * params->height;
    for (int r = 0; r < n; r++) if ((s = flx_frame_target_set(g->ctx[r], nullptr, 0))) { g->err = flx_last_error(g->ctx[r]); return s; }      /* (the servers end; their frames are complete) */
    if (g->target.base) { if (g->fifo_n) g->retired.push_back(g->target); else group_free_target(g, g->target); g->target = flx_group::Target(); g->shape_format = -1; }
    const uint32_t b = g->h_sync_next; g->h_sync_next = (b + 1u) % 3u;
    if (g->h_sync_pixels[b] < pixels) {
      if (g->h_sync[b]) (void)hipHostFree(g->h_sync[b]);
      g->h_sync[b] = nullptr; g->h_sync_pixels[b] = 0;
      if (hipHostMalloc((void **)&g->h_sync[b], pixels * sizeof(float4), hipHostMallocDefault) != hipSuccess) { g->err = "flx_group_frame_begin: pinned memory for the frame"; return FLX_ERR_DEVICE; }
      g->h_sync_pixels[b] = pixels;
    }
    if (lanes) {
      if (g->strip_copy.size() != (size_t)n) {
        g->strip_copy.assign((size_t)n, nullptr);
        for (int r = 0; r < n; r++) { (void)hipSetDevice(g->ctx[r]->device); if (hipStreamCreateWithFlags(&g->strip_copy[(size_t)r], hipStreamNonBlocking) != hipSuccess) { g->err = "flx_group_frame_begin: a copy stream"; return FLX_ERR_DEVICE; } }
      }
      for (int r = 0; r < n; r++) {
        flx_context *c = g->ctx[r];
        c->frame_chain = 0;                /* its own launches, alternating between the context's two lanes */
        if ((s = flx_frame_begin(c, &p[r], FLX_FRAME_DEVICE))) { g->err = flx_last_error(c); return s; }
      }
      slot.kind = 2; slot.bytes = pixels * sizeof(float4); slot.pixels = (const void *)g->h_sync[b]; slot.ms = 0.f; slot.params = *params; slot.tile_rows = tile_rows;
      g->fifo_n++;
      return FLX_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (format == FLX_FRAME_RGBA8) {
      /* the canvas' bytes of a frame the servers do not take: a filter / temporal frame is rendered as floats and quantised by context 0 (flx_present's store);
       * anything else is gathered as RGBA8 (flx_group_render_rgba8) */
      if (params->use_filter || params->is_temporal) {
        std::vector<float> tmp(pixels * 4u);
        if ((s = flx_group_render(g, params, 1, tile_rows, tmp.data(), nullptr))) return s;
        if ((s = flx_present(g->ctx[0], params->width, params->height, tmp.data(), (uint8_t *)g->h_sync[b]))) { g->err = flx_last_error(g->ctx[0]); return s; }
      } else if ((s = flx_group_render_rgba8(g, params, 1, tile_rows, (uint8_t *)g->h_sync[b], nullptr))) return s;
    } else
    if ((s = flx_group_render(g, params, 1, tile_rows, (float *)g->h_sync[b], nullptr))) return s;
    slot.kind = 0; slot.bytes = pixels * (format == FLX_FRAME_RGBA8 ? sizeof(uint32_t) : sizeof(float4));
    slot.pixels = (const void *)g->h_sync[b];
    slot.ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g->fifo_n++;
    return FLX_OK;
  }
  /* frames of the lanes still in flight: their launches complete before a server's does — a persistent launch that got the CUs first would hold them, waiting for the host,
   * while the host waits for those frames (where the contexts share one GPU: for one another's) */
  for (int i = 0; i < g->fifo_n; i++) if (g->fifo[i].kind == 2) {
    for (int r = 0; r < n; r++) {
      flx_context *c = g->ctx[r];
      (void)hipSetDevice(c->device);
      if (hipStreamSynchronize(c->stream) != hipSuccess || (c->twin && hipStreamSynchronize(c->twin->stream) != hipSuccess)) { g->err = "flx_group_frame_begin: waiting for the frames of the lanes"; return FLX_ERR_DEVICE; }
    }
    break;
  }
  if ((s = group_target(g, params, tile_rows, format))) return s;
  /* A launch that has to end or start allocates (and hipMalloc / hipFree wait for the device): where contexts share a device that must not happen while
   * another context's launch runs — it would wait for a launch that waits for the host.  So if ANY context cannot simply post, all launches end first (their
   * frames complete), every context gets its memory, and then the frame is posted everywhere. */
  bool simply = true;
  for (int r = 0; r < n; r++) if (!flx_server_continues(g->ctx[r], &p[r])) simply = false;
  if (!simply) {
    for (int r = 0; r < n; r++) if ((s = flx_server_stop(g->ctx[r]))) { g->err = flx_last_error(g->ctx[r]); return s; }
    for (int r = 0; r < n; r++) if ((s = flx_server_prepare(g->ctx[r], &p[r]))) { g->err = flx_last_error(g->ctx[r]); return s; }
  }
  int image = -1;
  for (int r = 0; r < n; r++) {
    flx_context *c = g->ctx[r];
    if ((s = flx_frame_begin(c, &p[r], FLX_FRAME_DEVICE))) { g->err = flx_last_error(c); return s; }
    const int at = flx_frame_target_index(c);
    if (r == 0) image = at;
    else if (at != image) { g->err = "flx_group_frame_begin: internal: the contexts' servers disagree about the frame's image"; return FLX_ERR_DEVICE; }
  }
  if (image < 0) { g->err = "flx_group_frame_begin: internal: no image"; return FLX_ERR_DEVICE; }
  slot.kind = 1; slot.bytes = g->target.pixels * g->target.texel;
  slot.pixels = (const char *)g->target.base + (size_t)image * slot.bytes;
  slot.ms = 0.f;
  g->fifo_n++;
  return FLX_OK;
}

extern "C" flx_status flx_group_frame_end(flx_group *g, const void **pixels, size_t *bytes, float *ms) {
  if (!g) return FLX_ERR_INVALID;
  if (g->fifo_n == 0) { g->err = "flx_group_frame_end: no frame in flight"; return FLX_ERR_INVALID; }
  flx_group::InFlight f = g->fifo[0];
  g->fifo[0] = g->fifo[1]; g->fifo[1] = g->fifo[2]; g->fifo_n--;
  flx_status first = FLX_OK;
  if (f.kind == 1) {
    float worst = 0.f;
    for (flx_context *c : g->ctx) {     /* every context's part of the image is complete (and written back to where the image lives) when its server says so */
      float one = 0.f;
      const flx_status s = flx_frame_end(c, nullptr, nullptr, &one);
      if (s && !first) { first = s; g->err = flx_last_error(c); }
      if (one > worst) worst = one;
    }
    f.ms = worst;
  }
  if (f.kind == 2) {
    /* every context's frame of its lanes is complete in ITS device memory (packed strips); each strip goes where the image has it — one copy per strip, all contexts'
     * at once over their own PCIe links — and the frame is handed out when the last byte is there */
    float worst = 0.f;
    const int n = (int)g->ctx.size();
    const uint32_t W = f.params.width, H = f.params.height, tr = f.tile_rows;
    for (int r = 0; r < n; r++) {
      flx_context *c = g->ctx[r];
      const void *dptr = nullptr; size_t got = 0; float one = 0.f;
      const flx_status s = flx_frame_end(c, &dptr, &got, &one);
      if (s) { if (!first) { first = s; g->err = flx_last_error(c); } continue; }
      if (one > worst) worst = one;
      (void)hipSetDevice(c->device);
      uint32_t packed = 0;
      for (uint32_t strip = (uint32_t)r; (size_t)strip * tr < H && !first; strip += (uint32_t)n) {
        const uint32_t row0 = strip * tr, rows = row0 + tr <= H ? tr : H - row0;
        if (hipMemcpyAsync((char *)const_cast<void *>(f.pixels) + (size_t)row0 * W * sizeof(float4), (const char *)dptr + (size_t)packed * W * sizeof(float4),
                           (size_t)rows * W * sizeof(float4), hipMemcpyDeviceToHost, g->strip_copy[(size_t)r]) != hipSuccess) { first = FLX_ERR_DEVICE; g->err = "flx_group_frame_end: copying a strip into the frame's image"; }
        packed += rows;
      }
    }
    for (int r = 0; r < n; r++) { (void)hipSetDevice(g->ctx[r]->device); if (hipStreamSynchronize(g->strip_copy[(size_t)r]) != hipSuccess && !first) { first = FLX_ERR_DEVICE; g->err = "flx_group_frame_end: the strips' copies"; } }
    f.ms = worst;
  }
  if (first) return first;
  if (pixels) *pixels = f.pixels;
  if (bytes) *bytes = f.bytes;
  if (ms) *ms = f.ms;
  return FLX_OK;
}
