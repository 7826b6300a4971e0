/*
 * flx_oracle_filter.c — CPU oracle of the denoise chain.  TEST INFRASTRUCTURE ONLY (flx_oracle.h).
 *
 * Follows shaders/pathtracer_first_filter.glsl:18-123, pathtracer_second_filter.glsl:17-79,
 * pathtracer_final_filter.glsl:13-71 and the pass schedule of modules/pathtracerWGL2.js:462-550
 * with firstPasses = secondPasses = 3 (forced at :818-819).  Every render target is RGBA8
 * (pathtracerWGL2.js:790-799): a write stores floor(clamp(x,0,1)*255 + 0.5), a read returns k/255.
 *
 * The schedule as the driver really runs it, frozen for frame 0 (all targets zero):
 *   pass  program  reads  R/Ip[n]  O[nO]  Id[nId]   writes R/Ip[np]   third output
 *    0    first          0        0      0                 1          Id[1]
 *    1    first          1        0      1                 0          Id[0]
 *    2    first          0        0      0                 1          Id[1]
 *    3    first          1        0      1                 2          dropped (IdRenderTexture[2] does not exist)
 *    4    second         2        1*     1                 3          dropped      (* O[1] not yet written: zeros)
 *    5    second         3        0      1                 2          O[1]
 *   final                2        1      1                 canvas
 * Behaviour the GLSL leaves open, pinned here: texelFetch outside the image returns 0; the first
 * filter's unwritten renderColorIp (read at :123) starts at 0; float->int casts truncate.
 * Image rows are stored top-down; the shaders' texel coordinates count from the bottom
 * (gl_FragCoord), so y is flipped when indexing — tap order and the 4-neighbour vote depend on it.
 */
#include "flx_oracle.h"
#include "flx_math.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define INV_256 0.00390625f

typedef struct { float x, y, z, w; } v4;
typedef struct { const uint8_t *p; int W, H; } Tex;      /* RGBA8, rows top-down */

static inline v4 V4(float x, float y, float z, float w) { v4 r = { x, y, z, w }; return r; }
static inline v4 add4(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 scale4(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline int eq3(v4 a, v4 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline int eq4(v4 a, v4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

/* texelFetch(tex, ivec2(x, y_gl), 0) with zero outside (and for a NULL = never-written texture) */
static inline v4 fetch(Tex t, int x, int y_gl) {
  if (!t.p || x < 0 || y_gl < 0 || x >= t.W || y_gl >= t.H) return V4(0.0f, 0.0f, 0.0f, 0.0f);
  const uint8_t *q = t.p + ((size_t)(t.H - 1 - y_gl) * t.W + x) * 4;
  return V4((float)q[0] / 255.0f, (float)q[1] / 255.0f, (float)q[2] / 255.0f, (float)q[3] / 255.0f);
}
static inline uint8_t quant(float x) {
  if (!(x > 0.0f)) return 0;               /* NaN and negatives clamp to 0 */
  if (x >= 1.0f) return 255;
  return (uint8_t)(x * 255.0f + 0.5f);
}
static inline void store(uint8_t *dst, int W, int H, int x, int y_gl, v4 v) {
  if (!dst) return;
  uint8_t *q = dst + ((size_t)(H - 1 - y_gl) * W + x) * 4;
  q[0] = quant(v.x); q[1] = quant(v.y); q[2] = quant(v.z); q[3] = quant(v.w);
}

static const float STENCIL3_37[37][2] = {
                              {-3, -1}, {-3, 0}, {-3, 1},
                    {-2, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-2, 2},
  {-1, -3}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {-1, 2}, {-1, 3},
  { 0, -3}, { 0, -2}, { 0, -1}, { 0, 0}, { 0, 1}, { 0, 2}, { 0, 3},
  { 1, -3}, { 1, -2}, { 1, -1}, { 1, 0}, { 1, 1}, { 1, 2}, { 1, 3},
                    { 2, -2}, { 2, -1}, { 2, 0}, { 2, 1}, { 2, 2},
                              { 3, -1}, { 3, 0}, { 3, 1}
};
static const int STENCIL1[4][2] = { {-1, 0}, {0, -1}, {0, 1}, {1, 0} };

/* pathtracer_first_filter.glsl:18-123 for texel (x, y_gl) */
static void first_filter(Tex tColor, Tex tIp, Tex tOColor, Tex tId, Tex tOId, int x, int y, v4 *oColor, v4 *oIp, v4 *oId) {
  v4 centerColor = fetch(tColor, x, y);
  v4 centerColorIp = fetch(tIp, x, y);
  v4 centerOColor = fetch(tOColor, x, y);
  v4 centerId = fetch(tId, x, y);
  int centerIdw = (int)(centerId.w * 255.0f);
  int centerLightNum = centerIdw / 2;
  int centerShadow = centerIdw % 2;
  v4 renderId = centerId;
  v4 renderColorIp = V4(0.0f, 0.0f, 0.0f, 0.0f);
  v4 centerOId = fetch(tOId, x, y);
  v4 color = V4(0.0f, 0.0f, 0.0f, 0.0f);
  float count = 0.0f;
  if (centerOId.w != 0.0f && centerColorIp.w != 0.0f) {
    v4 id = centerId;
    v4 ids[4], oIds[4];
    float ipws[4];
    for (int i = 0; i < 4; i++) {
      ids[i] = fetch(tId, x + STENCIL1[i][0], y + STENCIL1[i][1]);
      oIds[i] = fetch(tOId, x + STENCIL1[i][0], y + STENCIL1[i][1]);
      ipws[i] = fetch(tIp, x + STENCIL1[i][0], y + STENCIL1[i][1]).w;
    }
    int vote[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
      if (ipws[i] == 0.0f) {
        vote[i] = 1;
        if (eq3(ids[i], id) && eq4(oIds[i], centerOId)) vote[i]++;
        for (int j = i + 1; j < 4; j++) if (eq3(ids[i], ids[j]) && eq4(oIds[i], oIds[j])) vote[i]++;
      }
    }
    int maxVote = vote[0];
    int idNumber = 0;
    for (int i = 1; i < 4; i++) {
      if (vote[i] >= maxVote) { maxVote = vote[i]; idNumber = i; }
    }
    renderId = ids[idNumber];
    renderColorIp.w = flx_max(1.0f - flx_sign((float)maxVote), 0.0f);
  }
  if (centerOColor.w == 0.0f) {
    color = centerColor;
    count = 1.0f;
  } else {
    for (int i = 0; i < 37; i++) {
      float k = 1.0f + centerOColor.w;
      int cx = x + (int)(STENCIL3_37[i][0] * k * k * 3.5f);
      int cy = y + (int)(STENCIL3_37[i][1] * k * k * 3.5f);
      v4 id = fetch(tId, cx, cy);
      v4 originalId = fetch(tOId, cx, cy);
      int idW = (int)(id.w * 255.0f);
      int lightNum = idW / 2;
      int shadow = idW % 2;
      v4 nextColor = fetch(tColor, cx, cy);
      v4 nextColorIp = fetch(tIp, cx, cy);
      if (eq3(centerId, id) && eq4(centerOId, originalId) && (centerLightNum != lightNum || centerShadow == shadow)) {
        color = add4(color, add4(nextColor, scale4(nextColorIp, 256.0f)));
        count += 1.0f;
      }
    }
  }
  float invCount = 1.0f / count;
  float sg = flx_sign(centerColor.w);
  float cx_ = color.x * invCount, cy_ = color.y * invCount, cz_ = color.z * invCount;
  *oColor = V4(sg * flx_mod(cx_, 1.0f), sg * flx_mod(cy_, 1.0f), sg * flx_mod(cz_, 1.0f), sg * centerColor.w);
  *oIp = V4(sg * (flx_floor(cx_) * INV_256), sg * (flx_floor(cy_) * INV_256), sg * (flx_floor(cz_) * INV_256), sg * renderColorIp.w);
  *oId = renderId;
}

static const float STENCIL3_36[36][2] = {
                              {-3, -1}, {-3, 0}, {-3, 1},
                    {-2, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-2, 2},
  {-1, -3}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {-1, 2}, {-1, 3},
  { 0, -3}, { 0, -2}, { 0, -1},          { 0, 1}, { 0, 2}, { 0, 3},
  { 1, -3}, { 1, -2}, { 1, -1}, { 1, 0}, { 1, 1}, { 1, 2}, { 1, 3},
                    { 2, -2}, { 2, -1}, { 2, 0}, { 2, 1}, { 2, 2},
                              { 3, -1}, { 3, 0}, { 3, 1}
};

/* pathtracer_second_filter.glsl:17-79 */
static void second_filter(Tex tColor, Tex tIp, Tex tOColor, Tex tId, Tex tOId, int x, int y, v4 *oColor_, v4 *oIp, v4 *oOrig) {
  v4 centerColor = fetch(tColor, x, y);
  v4 centerColorIp = fetch(tIp, x, y);
  v4 centerOColor = fetch(tOColor, x, y);
  v4 centerId = fetch(tId, x, y);
  v4 centerOId = fetch(tOId, x, y);
  v4 color = add4(centerColor, scale4(V4(centerColorIp.x, centerColorIp.y, centerColorIp.z, 0.0f), 256.0f));
  v4 oColor = centerOColor;
  float ipw = centerColorIp.w;
  float count = 1.0f;
  float oCount = 1.0f;
  float scale = 1.0f + 2.0f * flx_tanh(centerOColor.w + centerOId.w * 4.0f);
  for (int i = 0; i < 36; i++) {
    int cx = x + (int)(STENCIL3_36[i][0] * scale);
    int cy = y + (int)(STENCIL3_36[i][1] * scale);
    v4 id = fetch(tId, cx, cy);
    v4 nextOId = fetch(tOId, cx, cy);
    v4 nextColor = fetch(tColor, cx, cy);
    v4 nextColorIp = fetch(tIp, cx, cy);
    v4 nextOColor = fetch(tOColor, cx, cy);
    if (eq3(centerOId, nextOId)) {
      if (flx_min(centerOId.w, nextOId.w) > 0.1f && (eq4(id, centerId) || flx_max(nextColorIp.w, centerColorIp.w) >= 0.1f)) {
        color = add4(color, add4(nextColor, scale4(V4(nextColorIp.x, nextColorIp.y, nextColorIp.z, 0.0f), 256.0f)));
        count += 1.0f;
        ipw += nextColorIp.w;
        oColor = add4(oColor, nextOColor);
        oCount += 1.0f;
      } else if (eq3(id, centerId)) {
        color = add4(color, add4(nextColor, scale4(V4(nextColorIp.x, nextColorIp.y, nextColorIp.z, 0.0f), 256.0f)));
        count += 1.0f;
      }
    }
  }
  float invCount = 1.0f / count;
  float w = centerColor.w;
  float cx_ = color.x * invCount, cy_ = color.y * invCount, cz_ = color.z * invCount;
  *oColor_ = V4(w * flx_mod(cx_, 1.0f), w * flx_mod(cy_, 1.0f), w * flx_mod(cz_, 1.0f), w * (color.w * invCount));
  *oIp = V4(w * (flx_floor(cx_) * INV_256), w * (flx_floor(cy_) * INV_256), w * (flx_floor(cz_) * INV_256), w * ipw);
  *oOrig = V4((w * oColor.x) / oCount, (w * oColor.y) / oCount, (w * oColor.z) / oCount, (w * oColor.w) / oCount);
}

/* pathtracer_final_filter.glsl:13-71; returns the canvas colour before its RGBA8 store */
static v4 final_filter(Tex tColor, Tex tIp, Tex tOColor, Tex tId, Tex tOId, int x, int y, int hdr) {
  v4 centerColor = fetch(tColor, x, y);
  v4 centerColorIp = fetch(tIp, x, y);
  v4 centerOColor = fetch(tOColor, x, y);
  v4 centerId = fetch(tId, x, y);
  v4 centerOId = fetch(tOId, x, y);
  v4 color = V4(0.0f, 0.0f, 0.0f, 0.0f);
  v4 oColor = V4(0.0f, 0.0f, 0.0f, 0.0f);
  float count = 0.0f, oCount = 0.0f;
  float scale = 0.7f + 2.0f * flx_tanh(centerOColor.w + centerOId.w * 4.0f);
  for (int i = 0; i < 37; i++) {
    int cx = x + (int)(STENCIL3_37[i][0] * scale);
    int cy = y + (int)(STENCIL3_37[i][1] * scale);
    v4 id = fetch(tId, cx, cy);
    v4 nextOId = fetch(tOId, cx, cy);
    v4 nextColor = fetch(tColor, cx, cy);
    v4 nextColorIp = fetch(tIp, cx, cy);
    v4 nextOColor = fetch(tOColor, cx, cy);
    int blurTranslucent = flx_max(nextColorIp.w, centerColorIp.w) != 0.0f && flx_min(centerOId.w, nextOId.w) > 0.0f;
    if (blurTranslucent && eq3(centerOId, nextOId)) {
      oColor = add4(oColor, nextOColor);
      oCount += 1.0f;
    }
    if ((blurTranslucent || eq3(centerId, id)) && eq3(centerOId, nextOId)) {
      color = add4(color, add4(nextColor, scale4(nextColorIp, 255.0f)));
      count += 1.0f;
    }
  }
  if (centerColor.w > 0.0f) {
    float f[3] = { color.x / count, color.y / count, color.z / count };
    float m[3];
    if (oCount == 0.0f) { m[0] = centerOColor.x; m[1] = centerOColor.y; m[2] = centerOColor.z; }
    else { m[0] = oColor.x / oCount; m[1] = oColor.y / oCount; m[2] = oColor.z / oCount; }
    for (int c = 0; c < 3; c++) {
      f[c] = f[c] * m[c];
      if (hdr == 1) {
        f[c] = f[c] / (f[c] + 1.0f);
        const float gamma = 0.8f;
        f[c] = flx_pow(4.0f * f[c], 1.0f / gamma) / 4.0f * 1.3f;
      }
    }
    return V4(f[0], f[1], f[2], 1.0f);
  }
  return V4(0.0f, 0.0f, 0.0f, 0.0f);
}

static uint8_t *quantise_plane(const float *src, int W, int H) {
  uint8_t *dst = (uint8_t *)malloc((size_t)W * H * 4);
  if (!dst) return NULL;
  for (size_t i = 0; i < (size_t)W * H * 4; i++) dst[i] = quant(src[i]);
  return dst;
}

/* The chain on RGBA8 planes (slot 0 of every ring given, rows top-down); planes are consumed (freed). */
static int filter_chain_u8(int W, int H, int hdr, uint8_t *R0, uint8_t *Ip0, uint8_t *O0, uint8_t *Id0, uint8_t *OId, float *out_rgba) {
  const size_t bytes = (size_t)W * H * 4;
  uint8_t *R[4] = { R0, (uint8_t *)calloc(bytes, 1), (uint8_t *)calloc(bytes, 1), (uint8_t *)calloc(bytes, 1) };
  uint8_t *Ip[4] = { Ip0, (uint8_t *)calloc(bytes, 1), (uint8_t *)calloc(bytes, 1), (uint8_t *)calloc(bytes, 1) };
  uint8_t *O[2] = { O0, (uint8_t *)calloc(bytes, 1) };
  uint8_t *Id[2] = { Id0, (uint8_t *)calloc(bytes, 1) };
  int ok = OId != NULL;
  for (int i = 0; i < 4; i++) ok = ok && R[i] && Ip[i];
  for (int i = 0; i < 2; i++) ok = ok && O[i] && Id[i];
  if (ok) {
    /* modules/pathtracerWGL2.js:462-512 with firstPasses = secondPasses = 3 */
    int n = 0, nId = 0, nOriginal = 0;
    for (int i = 0; i < 6; i++) {
      int np = (i % 2) ^ 1;
      int npOriginal = ((i - 3) % 2) ^ 1;
      if (3 <= i) np += 2;
      uint8_t *third = NULL;                          /* attachment 2 */
      if (3 <= i - 2) third = O[npOriginal];
      else if (np < 2) third = Id[np];                /* IdRenderTexture has two elements; [2],[3] do not exist */
      Tex tColor = { R[n], W, H }, tIp = { Ip[n], W, H }, tOColor = { O[nOriginal], W, H }, tId = { Id[nId], W, H }, tOId = { OId, W, H };
      const int first = n < 2;                        /* PostProgram[0..1] first filter, [2..3] second */
      uint8_t *dR = R[np], *dIp = Ip[np];
#pragma omp parallel for schedule(dynamic, 4)
      for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
          v4 a, b, c;
          if (first) first_filter(tColor, tIp, tOColor, tId, tOId, x, y, &a, &b, &c);
          else second_filter(tColor, tIp, tOColor, tId, tOId, x, y, &a, &b, &c);
          store(dR, W, H, x, y, a);
          store(dIp, W, H, x, y, b);
          store(third, W, H, x, y, c);
        }
      }
      n = np;
      if (3 <= i) nOriginal = npOriginal; else nId = np;
    }
    Tex tColor = { R[2], W, H }, tIp = { Ip[2], W, H }, tOColor = { O[1], W, H }, tId = { Id[1], W, H }, tOId = { OId, W, H };
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++) {
      for (int x = 0; x < W; x++) {
        v4 c = final_filter(tColor, tIp, tOColor, tId, tOId, x, y, hdr);
        float *o = out_rgba + ((size_t)(H - 1 - y) * W + x) * 4;
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
      }
    }
  }
  for (int i = 0; i < 4; i++) { free(R[i]); free(Ip[i]); }
  for (int i = 0; i < 2; i++) { free(O[i]); free(Id[i]); }
  free(OId);
  return ok ? FLX_OK : FLX_ERR_INVALID;
}

int flx_oracle_filter(const flx_frame_params *params, const flx_gbuffers *gb, float *out_rgba, int threads) {
  if (!params || !gb || !out_rgba || !gb->color || !gb->color_ip || !gb->original_color || !gb->id || !gb->original_id) return FLX_ERR_INVALID;
  const int W = (int)params->width, H = (int)params->height;
  if (W <= 0 || H <= 0) return FLX_ERR_INVALID;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  return filter_chain_u8(W, H, params->hdr, quantise_plane(gb->color, W, H), quantise_plane(gb->color_ip, W, H),
                         quantise_plane(gb->original_color, W, H), quantise_plane(gb->id, W, H), quantise_plane(gb->original_id, W, H), out_rgba);
}

/* ---- temporal accumulation (modules/pathtracerWGL2.js:389-402, 419-460, generated GLSL :571-662) -------------
 * History rings of N = temporalSamples RGBA8 planes: colour, colour integer part, location id, original id;
 * slot 0 is the frame just traced.  For texel t: the slots 1..N-1 whose location id equals slot 0's add their
 * colour, those whose original id equals slot 0's add their glass filter.  The generated shader walks the slots in
 * groups of four with vec4(0) standing in for slots >= N, so a texel whose own id is all zero (an uncovered pixel)
 * also "matches" those stand-ins: counted, as the shader does. */
typedef struct { uint8_t **c, **ip, **id, **oid; int n; } History;

static void temporal_pass(const History *h, int W, int H, int hdr, int use_filter, uint8_t *dColor, uint8_t *dIp, float *canvas) {
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++) {
    for (int x = 0; x < W; x++) {
      Tex c0 = { h->c[0], W, H }, ip0 = { h->ip[0], W, H }, id0 = { h->id[0], W, H }, oid0 = { h->oid[0], W, H };
      v4 id = fetch(id0, x, y), originalId = fetch(oid0, x, y);
      float counter = 1.0f, glassCounter = 1.0f;
      v4 cc = fetch(c0, x, y), ci = fetch(ip0, x, y);
      float centerW = cc.w;
      float color[3] = { cc.x + ci.x * 256.0f, cc.y + ci.y * 256.0f, cc.z + ci.z * 256.0f };
      float glassFilter = ci.w;
      for (int i = 1; i < h->n; i += 4) {
        v4 cs[4], ips[4], ids[4], oids[4];
        for (int j = 0; j < 4; j++) {
          int k = i + j;
          if (k < h->n) {
            Tex tc = { h->c[k], W, H }, tip = { h->ip[k], W, H }, tid = { h->id[k], W, H }, toid = { h->oid[k], W, H };
            cs[j] = fetch(tc, x, y); ips[j] = fetch(tip, x, y); ids[j] = fetch(tid, x, y); oids[j] = fetch(toid, x, y);
          } else {
            cs[j] = ips[j] = ids[j] = oids[j] = V4(0.0f, 0.0f, 0.0f, 0.0f);
          }
        }
        for (int j = 0; j < 4; j++) if (eq4(ids[j], id)) {
          color[0] += cs[j].x + ips[j].x * 256.0f; color[1] += cs[j].y + ips[j].y * 256.0f; color[2] += cs[j].z + ips[j].z * 256.0f;
          counter += 1.0f;
        }
        for (int j = 0; j < 4; j++) if (eq4(oids[j], originalId)) {
          glassFilter += ips[j].w;
          glassCounter += 1.0f;
        }
      }
      for (int k = 0; k < 3; k++) color[k] /= counter;
      glassFilter /= glassCounter;
      if (use_filter) {
        store(dColor, W, H, x, y, V4(flx_mod(color[0], 1.0f), flx_mod(color[1], 1.0f), flx_mod(color[2], 1.0f), centerW));
        store(dIp, W, H, x, y, V4(flx_floor(color[0]) / 256.0f, flx_floor(color[1]) / 256.0f, flx_floor(color[2]) / 256.0f, glassFilter));
      } else {
        if (hdr == 1) {
          for (int k = 0; k < 3; k++) {
            color[k] = color[k] / (color[k] + 1.0f);
            const float gamma = 0.8f;
            color[k] = flx_pow(4.0f * color[k], 1.0f / gamma) / 4.0f * 1.3f;
          }
        }
        float *o = canvas + ((size_t)(H - 1 - y) * W + x) * 4;
        o[0] = color[0]; o[1] = color[1]; o[2] = color[2]; o[3] = centerW;
      }
    }
  }
}

int flx_oracle_render_sequence_impl(const flx_scene_view *scene, const flx_frame_params *params, int n_frames, float *out_all,
                                    const flx_gbuffers *gb_last, flx_counters *counters_last, int threads) {
  if (!scene || !params || n_frames < 1 || !out_all) return FLX_ERR_INVALID;
  const int W = (int)params->width, H = (int)params->height;
  const size_t px = (size_t)W * H, bytes = px * 4;
  const int temporal = params->is_temporal == 1;
  int N = params->temporal_samples <= 0 ? 4 : (params->temporal_samples > 16 ? 16 : params->temporal_samples);
  if (!temporal) N = 1;
  uint8_t *ring[4][16];
  for (int r = 0; r < 4; r++) for (int i = 0; i < 16; i++) ring[r][i] = i < N ? (uint8_t *)calloc(bytes, 1) : NULL;
  float *g[6];
  for (int i = 0; i < 6; i++) g[i] = (float *)calloc(px * 4, sizeof(float));
  flx_gbuffers gb = { g[0], g[1], g[2], g[3], g[4], g[5] };
  int rc = FLX_OK;
  for (int f = 0; f < n_frames && rc == FLX_OK; f++) {
    flx_frame_params p = *params;
    if (temporal) p.random_seed = (float)(f % N);
    float *canvas = out_all + (size_t)f * px * 4;
    rc = flx_oracle_trace(scene, &p, NULL, &gb, f == n_frames - 1 ? counters_last : NULL, threads);
    if (rc != FLX_OK) break;
    uint8_t *R0 = quantise_plane(g[0], W, H), *Ip0 = quantise_plane(g[1], W, H);
    if (temporal) {
      /* rotate the rings (TempTexture.unshift(TempTexture.pop())) and store the new frame in slot 0 */
      for (int r = 0; r < 4; r++) { uint8_t *last = ring[r][N - 1]; for (int i = N - 1; i > 0; i--) ring[r][i] = ring[r][i - 1]; ring[r][0] = last; }
      memcpy(ring[0][0], R0, bytes); memcpy(ring[1][0], Ip0, bytes);
      uint8_t *q = quantise_plane(g[5], W, H); memcpy(ring[2][0], q, bytes); free(q);
      q = quantise_plane(g[4], W, H); memcpy(ring[3][0], q, bytes); free(q);
      History h = { ring[0], ring[1], ring[2], ring[3], N };
      temporal_pass(&h, W, H, p.hdr, p.use_filter == 1, R0, Ip0, canvas);      /* with filter: R0 / Ip0 become RenderTexture[0] / IpRenderTexture[0] */
    }
    if (p.use_filter == 1) {
      rc = filter_chain_u8(W, H, p.hdr, R0, Ip0, quantise_plane(g[2], W, H), quantise_plane(g[3], W, H), quantise_plane(g[4], W, H), canvas);
    } else {
      free(R0); free(Ip0);
    }
  }
  if (gb_last) {
    float *dst[6] = { gb_last->color, gb_last->color_ip, gb_last->original_color, gb_last->id, gb_last->original_id, gb_last->location_id };
    for (int i = 0; i < 6; i++) if (dst[i]) memcpy(dst[i], g[i], px * 4 * sizeof(float));
  }
  for (int i = 0; i < 6; i++) free(g[i]);
  for (int r = 0; r < 4; r++) for (int i = 0; i < 16; i++) free(ring[r][i]);
  return rc;
}

int flx_oracle_render_sequence(const flx_scene_view *scene, const flx_frame_params *params, int n_frames, float *out_all, int threads) {
  return flx_oracle_render_sequence_impl(scene, params, n_frames, out_all, NULL, NULL, threads);
}

/* ---- anti-aliasing post passes: modules/fxaa.js:7-137, modules/taa.js:11-59 ---------------------------------
 * Both read the RGBA8 texture the renderer drew into (the frame, stored floor(clamp(x) * 255 + 0.5)) and write the canvas;
 * the value kept here is the float the shader outputs.  Texels outside the image read as zero. */
static inline float fxaa_luma(v4 c) { return (c.y * (0.587f / 0.299f) + c.x) * c.w; }                 /* fxaa.js:27-29 */
static inline v4 mix4(v4 a, v4 b, float t) { return V4(flx_mix(a.x, b.x, t), flx_mix(a.y, b.y, t), flx_mix(a.z, b.z, t), flx_mix(a.w, b.w, t)); }

static void fxaa_texel(Tex t, int px, int py, float *out) {
#define FETCH(dx, dy) fetch(t, px + (dx), py + (dy))
#define LUMA(dx, dy) fxaa_luma(FETCH(dx, dy))
  const v4 original = FETCH(0, 0);
  float luma[3][3];                                         /* luma[j][i] = tex_luma(i - 1, j - 1): fxaa.js:75-79 */
  for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) luma[j][i] = LUMA(i - 1, j - 1);
  const float edge_vert = flx_abs((0.25f * luma[0][0]) + (-0.5f * luma[0][1]) + (0.25f * luma[0][2])) +
                          flx_abs((0.50f * luma[1][0]) + (-1.0f * luma[1][1]) + (0.50f * luma[1][2])) +
                          flx_abs((0.25f * luma[2][0]) + (-0.5f * luma[2][1]) + (0.25f * luma[2][2]));
  const float edge_horz = flx_abs((0.25f * luma[0][0]) + (-0.5f * luma[1][0]) + (0.25f * luma[2][0])) +
                          flx_abs((0.50f * luma[0][1]) + (-1.0f * luma[1][1]) + (0.50f * luma[2][1])) +
                          flx_abs((0.25f * luma[0][2]) + (-0.5f * luma[1][2]) + (0.25f * luma[2][2]));
  const int horz_span = edge_horz >= edge_vert;
  const int sx = horz_span ? 1 : 0, sy = horz_span ? 0 : 1;
  /* fxaa_contrast / fxaa_is_low_contrast (fxaa.js:37-50) at (0, 0) */
  {
    const float c = LUMA(0, 0), n = LUMA(0, -1), w = LUMA(-1, 0), s = LUMA(0, 1), e = LUMA(1, 0);
    const float lo = flx_min(c, flx_min(flx_min(n, w), flx_min(s, e))), hi = flx_max(c, flx_max(flx_max(n, w), flx_max(s, e)));
    const float range = hi - lo;
    if (range < flx_max(1.0f / 32.0f, hi * 1.0f / 2.0f)) { out[0] = original.x; out[1] = original.y; out[2] = original.z; out[3] = original.w; return; }
  }
  int nx = -sx, ny = -sy, qx = sx, qy = sy;
  v4 color = original;
  float pixel_count = 1.0f;
  int done_n = 0, done_p = 0;
  const float luma_mcn = flx_max(flx_max(flx_abs(luma[0][1] - luma[1][1]), flx_abs(luma[1][2] - luma[1][1])),
                                 flx_max(flx_abs(luma[2][1] - luma[1][1]), flx_abs(luma[1][0] - luma[1][1])));
  const float gradient = flx_abs(luma_mcn - luma[1][1]);
  for (int i = 0; i < 6; i++) {
    int x, y;
    if (!done_n) { x = nx; y = ny; }
    else if (!done_p) { x = qx; y = qy; }
    else break;
    /* blur_3x3 (fxaa.js:52-58): rows y-1, y, y+1, each x-1, x, x+1, summed left to right, times 1/9 */
    v4 sum = FETCH(x - 1, y - 1);
    sum = add4(sum, FETCH(x, y - 1)); sum = add4(sum, FETCH(x + 1, y - 1));
    sum = add4(sum, FETCH(x - 1, y)); sum = add4(sum, FETCH(x, y)); sum = add4(sum, FETCH(x + 1, y));
    sum = add4(sum, FETCH(x - 1, y + 1)); sum = add4(sum, FETCH(x, y + 1)); sum = add4(sum, FETCH(x + 1, y + 1));
    const v4 blur = scale4(sum, 1.0f / 9.0f);
    const int done = flx_abs(fxaa_luma(blur) - luma_mcn) >= gradient;
    /* fxaa_sub_pixel_aliasing (fxaa.js:60-70) at (x, y) */
    const float c = LUMA(x, y), n = LUMA(x, y - 1), w = LUMA(x - 1, y), s = LUMA(x, y + 1), e = LUMA(x + 1, y);
    const float luma_l = 0.25f * (((n + w) + e) + s);
    const float range_l = flx_abs(luma_l - c);
    const float lo = flx_min(c, flx_min(flx_min(n, w), flx_min(s, e))), hi = flx_max(c, flx_max(flx_max(n, w), flx_max(s, e)));
    const float range = hi - lo;
    float blend = flx_max(0.0f, (range_l / range) - 0.0f) * 1.0f;
    blend = flx_min(7.0f / 8.0f, blend);
    color = add4(color, mix4(FETCH(x, y), blur, blend));
    pixel_count += 1.0f;
    if (!done_n) { done_n = done; nx -= sx; ny -= sy; }
    else { done_p = done; qx += sx; qy += sy; }
  }
  out[0] = color.x / pixel_count; out[1] = color.y / pixel_count; out[2] = color.z / pixel_count; out[3] = color.w / pixel_count;
#undef FETCH
#undef LUMA
}

/* in / out: float RGBA frames, rows top-down */
int flx_oracle_fxaa(const float *in_rgba, uint32_t width, uint32_t height, float *out_rgba) {
  if (!in_rgba || !out_rgba || width == 0 || height == 0) return FLX_ERR_INVALID;
  const int W = (int)width, H = (int)height;
  uint8_t *q = quantise_plane(in_rgba, W, H);
  if (!q) return FLX_ERR_DEVICE;
  Tex t = { q, W, H };
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) fxaa_texel(t, x, y, out_rgba + ((size_t)(H - 1 - y) * W + x) * 4);
  free(q);
  return FLX_OK;
}

/* TAA over `n_frames` float frames (rows top-down), newest first — the state of the reference's ring after that many
 * renderFrame() calls; missing history (n_frames < 9) reads as the zero-initialised textures (taa.js:99-117). */
int flx_oracle_taa(const float *const *frames_newest_first, int n_frames, uint32_t width, uint32_t height, float *out_rgba) {
  if (!frames_newest_first || n_frames < 1 || !out_rgba || width == 0 || height == 0) return FLX_ERR_INVALID;
  const int W = (int)width, H = (int)height;
  uint8_t *q[9];
  for (int i = 0; i < 9; i++) q[i] = (i < n_frames && frames_newest_first[i]) ? quantise_plane(frames_newest_first[i], W, H) : NULL;
  Tex t[9];
  for (int i = 0; i < 9; i++) { t[i].p = q[i]; t[i].W = W; t[i].H = H; }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      v4 lo = V4(1.0f, 1.0f, 1.0f, 1.0f), hi = V4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {       /* length(vec2(i-1, j-1)) <= sqrt 2: nothing is skipped (taa.js:43) */
        const v4 p = fetch(t[0], x + (i - 1), y + (j - 1));
        lo = V4(flx_min(lo.x, p.x), flx_min(lo.y, p.y), flx_min(lo.z, p.z), flx_min(lo.w, p.w));
        hi = V4(flx_max(hi.x, p.x), flx_max(hi.y, p.y), flx_max(hi.z, p.z), flx_max(hi.w, p.w));
      }
      v4 o = fetch(t[0], x, y);
      for (int k = 1; k < 9; k++) {
        const v4 c = fetch(t[k], x, y);
        o = add4(o, V4(flx_min(flx_max(c.x, lo.x), hi.x), flx_min(flx_max(c.y, lo.y), hi.y), flx_min(flx_max(c.z, lo.z), hi.z), flx_min(flx_max(c.w, lo.w), hi.w)));
      }
      float *dst = out_rgba + ((size_t)(H - 1 - y) * W + x) * 4;
      dst[0] = o.x / 9.0f; dst[1] = o.y / 9.0f; dst[2] = o.z / 9.0f; dst[3] = o.w / 9.0f;
    }
  for (int i = 0; i < 9; i++) free(q[i]);
  return FLX_OK;
}

/* 8-bit present: the canvas' RGBA8 drawing buffer (pathtracerWGL2.js:552-553 draws the last pass into framebuffer null) */
int flx_oracle_present(const float *in_rgba, uint32_t width, uint32_t height, uint8_t *out_rgba8) {
  if (!in_rgba || !out_rgba8 || width == 0 || height == 0) return 1;
  for (size_t i = 0; i < (size_t)width * height * 4; i++) out_rgba8[i] = quant(in_rgba[i]);
  return 0;
}
