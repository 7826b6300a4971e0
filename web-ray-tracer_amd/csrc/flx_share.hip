/* flx_share.hip — one process per GPU: the ranks' frame servers complete ONE image in the root rank's device memory.
 *
 * A rank's frame server (flx_server.hip) resolves the row strips it rendered straight into an image of the caller's (flx_frame_target_set).  Here the image
 * lives on the root's GPU: the root allocates it, exports it (hipIpcGetMemHandle), the other ranks map it (hipIpcOpenMemHandle: stores go over xGMI) and every
 * rank's launch writes its strips where the image has them.  A frame therefore needs no collective, no reassembly kernel and no copy — and no kernel beside
 * the servers' launches, which is what a collective would be (csrc/flx_group.hip's RCCL gather runs between the launches of one frame at a time).
 *
 * What the ranks tell each other goes through one page of POSIX shared memory (plain host loads and stores, release / acquire):
 *   done[r]   frames whose strips rank r has completed (its flx_frame_end_shared stores it)            -> the root waits for all of them before it hands a frame out
 *   released  frames the root is through with (set where the root begins its next frame)              -> no rank posts frame g before the image of g - n is free
 *   error[r]  rank r failed: everybody's next call fails instead of waiting
 * The reference presents every frame from its one context (modules/pathtracerWGL2.js:254-303, 552-553); the root is that context here. */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>

#include "flx_context.h"

using namespace flx;

struct SharePage {
  uint32_t magic, n_ranks, n_images, pad;
  uint64_t released;
  uint64_t done[64];
  uint32_t error[64];
};
constexpr uint32_t SHARE_MAGIC = 0x46585348u;      /* "FXSH" */

struct flx_share {
  SharePage *page = nullptr;
  char name[48] = {};
  bool root = false;
  int rank = 0, n_ranks = 1;
  uint32_t width = 0, height = 0, n_images = 0;
  float4 *images = nullptr;            /* the root's allocation, or this rank's mapping of it */
  uint64_t begun = 0, ended = 0;
  /* frames the frame server does not take (a scene of <= 128 entries, strips that are no multiple of 8 rows): rendered on the context's two lanes and copied, strip by
   * strip, into the image when the frame is taken — the same image on every rank: frame g of such a run goes to image g % n_images */
  bool lanes_mode = false;             /* the frames in flight are of that kind (the kinds do not mix in flight) */
  hipStream_t copy = nullptr;
  struct Pending { uint32_t image; flx_frame_params params; } fifo[3] = {};
  int fifo_n = 0;
};

struct ShareHandle {                   /* FLX_SHARE_HANDLE_BYTES */
  hipIpcMemHandle_t mem;
  uint32_t width, height, n_images, n_ranks;
  char name[48];
};
static_assert(sizeof(ShareHandle) == FLX_SHARE_HANDLE_BYTES, "flexlight_hip.h: FLX_SHARE_HANDLE_BYTES");

static flx_status share_fail(flx_context *ctx, flx_status s, const char *msg) { ctx->err = msg; return s; }

static flx_status share_attach(flx_context *ctx, flx_share *sh) {
  void *img[3] = { nullptr, nullptr, nullptr };
  for (uint32_t i = 0; i < sh->n_images; i++) img[i] = sh->images + (size_t)i * sh->width * sh->height;
  flx_status s = flx_set_frame_lanes(ctx, (int)sh->n_images);
  if (!s) s = flx_set_frame_chain(ctx, 3);
  if (!s) s = flx_frame_target_set(ctx, img, sh->n_images);
  if (!s) ctx->sv_target_posted = sh->begun;      /* frame g of the share goes to image g % n on every rank, whichever launch takes it and whatever ran on the lanes before */
  return s;
}

extern "C" flx_status flx_share_leave(flx_context *ctx) {
  if (!ctx) return FLX_ERR_INVALID;
  flx_share *sh = ctx->share;
  if (!sh) return FLX_OK;
  (void)hipSetDevice(ctx->device);
  while (ctx->fifo_n) (void)flx_frame_end(ctx, nullptr, nullptr, nullptr);
  (void)flx_frame_target_set(ctx, nullptr, 0);
  if (sh->copy) { (void)hipStreamSynchronize(sh->copy); (void)hipStreamDestroy(sh->copy); sh->copy = nullptr; }
  if (sh->page) {
    if (sh->root) __atomic_store_n(&sh->page->released, ~0ull >> 1, __ATOMIC_RELEASE);      /* nobody waits for a root that has left */
    else if (sh->rank >= 0) __atomic_store_n(&sh->page->error[sh->rank], 2u, __ATOMIC_RELEASE);      /* left: the root's next call fails instead of handing out images without this rank's strips */
    munmap(sh->page, sizeof(SharePage));
  }
  if (sh->images) { if (sh->root) (void)hipFree(sh->images); else (void)hipIpcCloseMemHandle(sh->images); }
  if (sh->root && sh->name[0]) shm_unlink(sh->name);
  delete sh;
  ctx->share = nullptr;
  return FLX_OK;
}

extern "C" flx_status flx_share_create(flx_context *ctx, uint32_t width, uint32_t height, uint32_t n_images, int n_ranks, int rank, uint8_t *handle_out) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!handle_out || width == 0u || height == 0u || (n_images != 2u && n_images != 3u) || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
    return share_fail(ctx, FLX_ERR_INVALID, "flx_share_create: a handle, width, height, 2 or 3 images, 1 .. 64 ranks, 0 <= rank < n_ranks");
  if (ctx->share) return share_fail(ctx, FLX_ERR_INVALID, "flx_share_create: the context shares frames already (flx_share_leave)");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_share *sh = new flx_share();
  sh->root = true; sh->rank = rank; sh->n_ranks = n_ranks; sh->width = width; sh->height = height; sh->n_images = n_images;
  ctx->share = sh;
  const size_t bytes = (size_t)n_images * width * height * sizeof(float4);
  if (hipMalloc(&sh->images, bytes) != hipSuccess) { flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_DEVICE, "flx_share_create: hipMalloc of the images"); }
  ShareHandle h;
  memset(&h, 0, sizeof h);
  if (hipIpcGetMemHandle(&h.mem, sh->images) != hipSuccess) { flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_DEVICE, "flx_share_create: hipIpcGetMemHandle"); }
  static std::atomic<unsigned> counter{0};
  snprintf(sh->name, sizeof sh->name, "/flx_share_%ld_%u", (long)getpid(), counter.fetch_add(1u));
  const int fd = shm_open(sh->name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(SharePage)) != 0) { if (fd >= 0) close(fd); flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_DEVICE, "flx_share_create: shm_open of the ranks' page"); }
  void *m = mmap(nullptr, sizeof(SharePage), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_DEVICE, "flx_share_create: mmap of the ranks' page"); }
  sh->page = (SharePage *)m;
  memset(sh->page, 0, sizeof(SharePage));
  sh->page->n_ranks = (uint32_t)n_ranks; sh->page->n_images = n_images;
  __atomic_store_n(&sh->page->magic, SHARE_MAGIC, __ATOMIC_RELEASE);
  h.width = width; h.height = height; h.n_images = n_images; h.n_ranks = (uint32_t)n_ranks;
  memcpy(h.name, sh->name, sizeof h.name);
  memcpy(handle_out, &h, sizeof h);
  const flx_status s = share_attach(ctx, sh);
  if (s) { const std::string keep = ctx->err; flx_share_leave(ctx); ctx->err = keep; }
  return s;
}

extern "C" flx_status flx_share_join(flx_context *ctx, const uint8_t *handle, int rank) {
  if (!ctx) return FLX_ERR_INVALID;
  if (!handle) return share_fail(ctx, FLX_ERR_INVALID, "flx_share_join: handle is NULL");
  if (ctx->share) return share_fail(ctx, FLX_ERR_INVALID, "flx_share_join: the context shares frames already (flx_share_leave)");
  ShareHandle h;
  memcpy(&h, handle, sizeof h);
  h.name[sizeof h.name - 1] = 0;
  if (h.n_ranks < 1u || h.n_ranks > 64u || rank < 0 || rank >= (int)h.n_ranks || (h.n_images != 2u && h.n_images != 3u) || h.name[0] != '/' || h.width == 0u || h.height == 0u) return share_fail(ctx, FLX_ERR_INVALID, "flx_share_join: not a handle of flx_share_create, or rank out of range");
  FLX_HIP(ctx, hipSetDevice(ctx->device));
  flx_share *sh = new flx_share();
  sh->root = false; sh->rank = rank; sh->n_ranks = (int)h.n_ranks; sh->width = h.width; sh->height = h.height; sh->n_images = h.n_images;
  ctx->share = sh;
  const int fd = shm_open(h.name, O_RDWR, 0600);
  void *m = fd >= 0 ? mmap(nullptr, sizeof(SharePage), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
  if (fd >= 0) close(fd);
  if (m == MAP_FAILED) { flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_DEVICE, "flx_share_join: the ranks' page (shm_open / mmap): is the root on this host?"); }
  sh->page = (SharePage *)m;
  if (__atomic_load_n(&sh->page->magic, __ATOMIC_ACQUIRE) != SHARE_MAGIC) { sh->rank = -1; flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_DEVICE, "flx_share_join: the ranks' page is not initialised"); }
  if (sh->page->n_ranks != h.n_ranks || sh->page->n_images != h.n_images) { sh->rank = -1; flx_share_leave(ctx); return share_fail(ctx, FLX_ERR_INVALID, "flx_share_join: the handle's ranks / images are not the page's"); }
  void *p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h.mem, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    __atomic_store_n(&sh->page->error[rank], 1u, __ATOMIC_RELEASE);
    flx_share_leave(ctx);
    ctx->err = std::string("flx_share_join: hipIpcOpenMemHandle: ") + hipGetErrorString(e);
    return FLX_ERR_DEVICE;
  }
  sh->images = (float4 *)p;
  const flx_status s = share_attach(ctx, sh);
  if (s) { const std::string keep = ctx->err; flx_share_leave(ctx); ctx->err = keep; }
  return s;
}

/* error[r]: 1 = rank r failed; 2 = rank r has left in good order (flx_share_leave): the frames it completed stay completed, no later frame will ever have its strips */
static bool share_broken(const flx_share *sh) {
  for (int r = 0; r < sh->n_ranks; r++) if (__atomic_load_n(&sh->page->error[r], __ATOMIC_ACQUIRE) == 1u) return true;
  return false;
}
static bool share_gone(const flx_share *sh, uint64_t want) {      /* a rank has left before it completed frame number `want` (counted from 1) */
  for (int r = 0; r < sh->n_ranks; r++)
    if (__atomic_load_n(&sh->page->error[r], __ATOMIC_ACQUIRE) == 2u && __atomic_load_n(&sh->page->done[r], __ATOMIC_ACQUIRE) < want) return true;
  return false;
}
template <class F> static bool share_wait(const flx_share *sh, F ready) {      /* false: a rank failed, or nothing happened for 5 s */
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t spins = 0;; spins++) {
    if (ready()) return true;
    if ((spins & 255u) == 255u) {
      if (share_broken(sh) || share_gone(sh, sh->ended)) return false;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) return false;
    }
    __builtin_ia32_pause();
  }
}

extern "C" flx_status flx_frame_begin_shared(flx_context *ctx, const flx_frame_params *params) {
  if (!ctx) return FLX_ERR_INVALID;
  flx_share *sh = ctx->share;
  if (!sh) return share_fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_shared: flx_share_create / flx_share_join first");
  if (!params || params->width != sh->width || params->height != sh->height || (int)params->tile_count != sh->n_ranks || (int)params->tile_index != sh->rank)
    return share_fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_shared: the frame must have the images' size, tile_count = the ranks and tile_index = this rank");
  if (share_broken(sh)) return share_fail(ctx, FLX_ERR_DEVICE, "flx_frame_begin_shared: a rank of the share has failed");
  if (sh->root && share_gone(sh, sh->begun + 1u)) return share_fail(ctx, FLX_ERR_DEVICE, "flx_frame_begin_shared: a rank of the share has left: this frame would never have its strips");
  if (params->use_filter || params->is_temporal) return share_fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_shared: filter and temporal frames need the whole frame in one context (flx_render_gathered_root_device)");
  /* the frame server where it takes the frame; else the context's two lanes and a copy of the strips into the image (the same choice on every rank: it depends on the
   * scene and on the frame's shape only) */
  bool lanes = false;
  for (int r = 0; r < sh->n_ranks && !lanes; r++) {          /* (EVERY rank's share: a frame whose last strip is cut by the frame's edge is one the server does not take on the rank that owns that strip) */
    flx_frame_params q = *params;
    q.tile_index = (uint32_t)r;
    if (!flx_frame_server_takes(ctx, &q)) lanes = true;
  }
  if (lanes != sh->lanes_mode) {
    if (ctx->fifo_n) return share_fail(ctx, FLX_ERR_INVALID, "flx_frame_begin_shared: frames the frame server takes and frames it does not take cannot be in flight together: take the frames in flight first");
    flx_status ts = FLX_OK;
    if (lanes) {
      ts = flx_frame_target_set(ctx, nullptr, 0);                    /* (the lanes' frames stay in this context's memory until they are copied) */
      if (!ts) ts = flx_set_frame_chain(ctx, 0);
      if (!ts && !sh->copy && hipStreamCreateWithFlags(&sh->copy, hipStreamNonBlocking) != hipSuccess) ts = share_fail(ctx, FLX_ERR_DEVICE, "flx_frame_begin_shared: a copy stream");
    } else ts = share_attach(ctx, sh);
    if (ts) { __atomic_store_n(&sh->page->error[sh->rank], 1u, __ATOMIC_RELEASE); return ts; }
    sh->lanes_mode = lanes;
  }
  if (sh->root) {
    /* the root is through with every frame it has been handed (the caller begins the next one): their images are free */
    __atomic_store_n(&sh->page->released, sh->ended, __ATOMIC_RELEASE);
  } else {
    /* frame g goes into the image of frame g - n: not before the root is through with that one (and so never before the root has begun frame g itself) */
    const uint64_t need = sh->begun + 1u > sh->n_images ? sh->begun + 1u - sh->n_images : 0u;
    if (!share_wait(sh, [&] { return __atomic_load_n(&sh->page->released, __ATOMIC_ACQUIRE) >= need; })) {
      __atomic_store_n(&sh->page->error[sh->rank], 1u, __ATOMIC_RELEASE);
      return share_fail(ctx, FLX_ERR_DEVICE, "flx_frame_begin_shared: the root does not take the frames (no release within 5 s, or a rank failed)");
    }
  }
  const flx_status s = flx_frame_begin(ctx, params, FLX_FRAME_DEVICE);
  if (s) { __atomic_store_n(&sh->page->error[sh->rank], 1u, __ATOMIC_RELEASE); return s; }
  if (sh->fifo_n < 3) { sh->fifo[sh->fifo_n].image = (uint32_t)(sh->begun % sh->n_images); sh->fifo[sh->fifo_n].params = *params; sh->fifo_n++; }
  sh->begun++;
  return FLX_OK;
}

extern "C" flx_status flx_frame_end_shared(flx_context *ctx, const void **image, size_t *bytes, float *ms) {
  if (!ctx) return FLX_ERR_INVALID;
  flx_share *sh = ctx->share;
  if (!sh) return share_fail(ctx, FLX_ERR_INVALID, "flx_frame_end_shared: flx_share_create / flx_share_join first");
  const void *p = nullptr;
  const flx_status s = flx_frame_end(ctx, &p, nullptr, ms);      /* this rank's strips are in the image (written back at system scope by the launch) */
  if (s) { __atomic_store_n(&sh->page->error[sh->rank], 1u, __ATOMIC_RELEASE); return s; }
  flx_share::Pending pf = sh->fifo[0];
  if (sh->fifo_n) { sh->fifo[0] = sh->fifo[1]; sh->fifo[1] = sh->fifo[2]; sh->fifo_n--; }
  if (sh->lanes_mode) {
    /* a frame of the lanes: this rank's packed strips (its own device memory) go where the image has them — the root's memory, through the mapping: over xGMI */
    float4 *image = sh->images + (size_t)pf.image * sh->width * sh->height;
    /* runs of consecutive image rows among this rank's packed rows (flx_tile_row_at: the tile policy itself says where a packed row lives) */
    const uint32_t mine = flx_tile_row_count(&pf.params);
    hipError_t e = hipSuccess;
    for (uint32_t k = 0; k < mine && e == hipSuccess;) {
      const uint32_t row0 = flx_tile_row_at(&pf.params, k);
      uint32_t run = 1;
      while (k + run < mine && flx_tile_row_at(&pf.params, k + run) == row0 + run) run++;
      e = hipMemcpyAsync(image + (size_t)row0 * sh->width, (const float4 *)p + (size_t)k * sh->width, (size_t)run * sh->width * sizeof(float4), hipMemcpyDeviceToDevice, sh->copy);
      k += run;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(sh->copy);
    if (e != hipSuccess) { __atomic_store_n(&sh->page->error[sh->rank], 1u, __ATOMIC_RELEASE); ctx->err = std::string("flx_frame_end_shared: copying the strips into the image: ") + hipGetErrorString(e); return FLX_ERR_DEVICE; }
    p = image;
  }
  sh->ended++;
  __atomic_store_n(&sh->page->done[sh->rank], sh->ended, __ATOMIC_RELEASE);
  if (image) *image = nullptr;
  if (bytes) *bytes = 0;
  if (sh->root) {
    const uint64_t want = sh->ended;
    if (!share_wait(sh, [&] { for (int r = 0; r < sh->n_ranks; r++) if (__atomic_load_n(&sh->page->done[r], __ATOMIC_ACQUIRE) < want) return false; return true; })) {
      __atomic_store_n(&sh->page->error[sh->rank], 1u, __ATOMIC_RELEASE);
      return share_fail(ctx, FLX_ERR_DEVICE, "flx_frame_end_shared: a rank did not complete its strips (nothing within 5 s, or it failed)");
    }
    if (image) *image = p;
    if (bytes) *bytes = (size_t)sh->width * sh->height * sizeof(float4);
  }
  return FLX_OK;
}
