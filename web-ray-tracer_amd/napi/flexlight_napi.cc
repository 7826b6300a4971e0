/*
 * flexlight_napi.cc — thin N-API shim over the C ABI of libflexlight_hip.so (include/flexlight_hip.h).
 *
 * One JS function per C entry point, typed arrays in, typed arrays out, no logic: the JavaScript
 * renderer (js/pathtracerHIP.js) is to this addon what modules/pathtracerWGL2.js is to the WebGL2
 * context.  A non-zero flx_status becomes a JS exception carrying flx_last_error().  Memory: every
 * typed array stays owned by JS; the library copies in during the call (napi_get_typedarray_info
 * gives the backing store, nothing is retained).  Built with plain g++ against /usr/include/node
 * (no node-gyp, no network): napi/Makefile.
 */
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "flexlight_hip.h"

#define NAPI_OK(env, call)                                                     \
  do {                                                                         \
    if ((call) != napi_ok) { napi_throw_error((env), nullptr, "N-API call failed: " #call); return nullptr; } \
  } while (0)

static napi_value fail(napi_env env, flx_context *ctx, const char *what, flx_status rc) {
  std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + flx_last_error(ctx);
  napi_throw_error(env, nullptr, msg.c_str());
  return nullptr;
}

static bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
    napi_throw_type_error(env, nullptr, "wrong number of arguments");
    return false;
  }
  return true;
}

/* What a context handle points to: the context and the ArrayBuffers handed out over its pinned frame slots (frameEnd).  Those
 * buffers are views of memory the LIBRARY owns — freed by halt(), re-allocated when a larger frame needs a bigger slot, re-used
 * by the frame after the next — so the box keeps a weak reference to each and DETACHES it (length 0, no pointer) the moment its
 * memory stops being that frame's: a JavaScript holder of an old frame then reads an empty array, never freed memory. */
struct CtxBox {
  flx_context *ctx = nullptr;              /* first member: a handle also reads as flx_context ** */
  struct View { const void *ptr; napi_ref ref; };
  std::vector<View> views;
  bool busy = false;                       /* a frameEndAsync is waiting for the GPU on a worker thread: the context is that thread's until its promise settles */
};
static void detach_view(napi_env env, CtxBox::View &v) {
  napi_value buf = nullptr;
  if (napi_get_reference_value(env, v.ref, &buf) == napi_ok && buf) (void)napi_detach_arraybuffer(env, buf);      /* (collected already: nothing to do) */
  napi_delete_reference(env, v.ref);
}
/* detach the views for which keep(ptr) is false */
template <typename Keep> static void detach_views(napi_env env, CtxBox *box, Keep keep) {
  size_t n = 0;
  for (auto &v : box->views) { if (keep(v.ptr)) box->views[n++] = v; else detach_view(env, v); }
  box->views.resize(n);
}
static CtxBox *get_box(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "expected a context handle"); return nullptr; }
  CtxBox *box = static_cast<CtxBox *>(p);
  if (!box->ctx) { napi_throw_error(env, nullptr, "context was halted"); return nullptr; }
  if (box->busy) { napi_throw_error(env, nullptr, "a frameEndAsync of this context is pending: await it first"); return nullptr; }
  return box;
}
static flx_context *get_ctx(napi_env env, napi_value v) {
  CtxBox *box = get_box(env, v);
  return box ? box->ctx : nullptr;
}

/* typed array -> pointer + element count; null / undefined -> nullptr, 0 */
static bool typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len) {
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_null || t == napi_undefined) { *data = nullptr; *len = 0; return true; }
  bool is = false;
  napi_is_typedarray(env, v, &is);
  napi_typedarray_type type;
  if (!is || napi_get_typedarray_info(env, v, &type, len, data, nullptr, nullptr) != napi_ok ||
      (type != want && !(want == napi_uint8_array && type == napi_uint8_clamped_array))) {
    napi_throw_type_error(env, nullptr, "expected a typed array of the right element type");
    return false;
  }
  return true;
}

static bool num(napi_env env, napi_value obj, const char *key, double *out, bool required = true) {
  napi_value v;
  bool has = false;
  napi_has_named_property(env, obj, key, &has);
  if (!has) {
    if (required) { napi_throw_type_error(env, nullptr, (std::string("frame params: missing ") + key).c_str()); return false; }
    return true;
  }
  napi_get_named_property(env, obj, key, &v);
  napi_valuetype t;
  napi_typeof(env, v, &t);
  if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); *out = b ? 1.0 : 0.0; return true; }
  if (napi_get_value_double(env, v, out) != napi_ok) { napi_throw_type_error(env, nullptr, (std::string("frame params: ") + key + " is not a number").c_str()); return false; }
  return true;
}

static bool floats(napi_env env, napi_value obj, const char *key, float *dst, size_t n) {
  napi_value v;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return false;
  for (size_t i = 0; i < n; i++) {
    napi_value e; double d;
    if (napi_get_element(env, v, (uint32_t)i, &e) != napi_ok || napi_get_value_double(env, e, &d) != napi_ok) {
      napi_throw_type_error(env, nullptr, (std::string("frame params: ") + key + " needs " + std::to_string(n) + " numbers").c_str());
      return false;
    }
    dst[i] = (float)d;
  }
  return true;
}

static void finalize_ctx(napi_env env, void *data, void *) {
  CtxBox *box = static_cast<CtxBox *>(data);
  for (auto &v : box->views) napi_delete_reference(env, v.ref);      /* (finalizers may not touch JS values; whoever still holds a view keeps the handle alive through its frame object) */
  if (box->ctx) flx_context_destroy(box->ctx);
  delete box;
}

/* createContext(device) -> handle */
static napi_value CreateContext(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  int32_t device = 0;
  NAPI_OK(env, napi_get_value_int32(env, argv[0], &device));
  flx_context *ctx = nullptr;
  flx_status rc = flx_context_create(device, &ctx);
  if (rc != FLX_OK) return fail(env, nullptr, "flx_context_create", rc);
  napi_value ext;
  CtxBox *box = new CtxBox();
  box->ctx = ctx;
  NAPI_OK(env, napi_create_external(env, box, finalize_ctx, nullptr, &ext));
  return ext;
}

/* destroyContext(handle): halt() */
static napi_value DestroyContext(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  void *p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    CtxBox *box = static_cast<CtxBox *>(p);
    if (box->busy) { napi_throw_error(env, nullptr, "destroyContext: a frameEndAsync is pending: await it first"); return nullptr; }
    detach_views(env, box, [](const void *) { return false; });      /* the pinned slots are about to be freed */
    if (box->ctx) { flx_context_destroy(box->ctx); box->ctx = nullptr; }
  }
  return nullptr;
}

/* uploadScene(handle, geometry Float32Array, attributes Float32Array, ids Int32Array) */
static napi_value UploadScene(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  void *g, *a, *ids; size_t ng, na, nids;
  if (!typed(env, argv[1], napi_float32_array, &g, &ng) || !typed(env, argv[2], napi_float32_array, &a, &na) ||
      !typed(env, argv[3], napi_int32_array, &ids, &nids)) return nullptr;
  if (ng % 12 != 0 || na / 28 != ng / 12 || na % 28 != 0) { napi_throw_range_error(env, nullptr, "geometry needs 12 and attributes 28 floats per entry"); return nullptr; }
  flx_status rc = flx_scene_upload(ctx, (const float *)g, (const float *)a, (uint32_t)(ng / 12), (const int32_t *)ids, (uint32_t)nids);
  if (rc != FLX_OK) return fail(env, ctx, "flx_scene_upload", rc);
  return nullptr;
}

/* uploadTransforms(handle, rotation Float32Array(24 T), shift Float32Array(8 T)) */
static napi_value UploadTransforms(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  void *r, *s; size_t nr, ns;
  if (!typed(env, argv[1], napi_float32_array, &r, &nr) || !typed(env, argv[2], napi_float32_array, &s, &ns)) return nullptr;
  if (ns % 8 != 0 || nr != ns * 3) { napi_throw_range_error(env, nullptr, "rotation needs 24 and shift 8 floats per transform"); return nullptr; }
  flx_status rc = flx_transforms_upload(ctx, (const float *)r, (const float *)s, (uint32_t)(ns / 8));
  if (rc != FLX_OK) return fail(env, ctx, "flx_transforms_upload", rc);
  return nullptr;
}

/* uploadLights(handle, lights Float32Array(6 L)) */
static napi_value UploadLights(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  void *l; size_t nl;
  if (!typed(env, argv[1], napi_float32_array, &l, &nl)) return nullptr;
  if (nl % 6 != 0) { napi_throw_range_error(env, nullptr, "lights needs 6 floats per light"); return nullptr; }
  flx_status rc = flx_lights_upload(ctx, (const float *)l, (uint32_t)(nl / 6));
  if (rc != FLX_OK) return fail(env, ctx, "flx_lights_upload", rc);
  return nullptr;
}

/* uploadAtlas(handle, which, rgba Uint8Array | null, width, height) */
static napi_value UploadAtlas(napi_env env, napi_callback_info info) {
  napi_value argv[5];
  if (!get_args(env, info, 5, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  int32_t which; uint32_t w, h;
  NAPI_OK(env, napi_get_value_int32(env, argv[1], &which));
  void *px; size_t n;
  if (!typed(env, argv[2], napi_uint8_array, &px, &n)) return nullptr;
  NAPI_OK(env, napi_get_value_uint32(env, argv[3], &w));
  NAPI_OK(env, napi_get_value_uint32(env, argv[4], &h));
  if (px && n != (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "atlas needs width*height*4 bytes"); return nullptr; }
  flx_status rc = flx_atlas_upload(ctx, which, (const uint8_t *)px, w, h);
  if (rc != FLX_OK) return fail(env, ctx, "flx_atlas_upload", rc);
  return nullptr;
}

static bool read_params(napi_env env, napi_value o, flx_frame_params *p) {
  memset(p, 0, sizeof *p);
  double d = 0;
  if (!num(env, o, "width", &d)) return false; p->width = (uint32_t)d;
  if (!num(env, o, "height", &d)) return false; p->height = (uint32_t)d;
  if (!floats(env, o, "camera", p->camera, 3) || !floats(env, o, "viewMatrix", p->view_matrix, 9) || !floats(env, o, "ambient", p->ambient, 3)) return false;
  if (!num(env, o, "samples", &d)) return false; p->samples = (int32_t)d;
  if (!num(env, o, "maxReflections", &d)) return false; p->max_reflections = (int32_t)d;
  if (!num(env, o, "minImportancy", &d)) return false; p->min_importancy = (float)d;
  d = 0; if (!num(env, o, "useFilter", &d, false)) return false; p->use_filter = (int32_t)d;
  d = 0; if (!num(env, o, "isTemporal", &d, false)) return false; p->is_temporal = (int32_t)d;
  d = 1; if (!num(env, o, "hdr", &d, false)) return false; p->hdr = (int32_t)d;
  d = 0; if (!num(env, o, "randomSeed", &d, false)) return false; p->random_seed = (float)d;
  if (!num(env, o, "textureWidth", &d)) return false; p->texture_width = (int32_t)d;
  d = 0; if (!num(env, o, "tileRows", &d, false)) return false; p->tile_rows = (uint32_t)d;
  d = 0; if (!num(env, o, "tileIndex", &d, false)) return false; p->tile_index = (uint32_t)d;
  d = 0; if (!num(env, o, "tileCount", &d, false)) return false; p->tile_count = (uint32_t)d;
  d = 4; if (!num(env, o, "temporalSamples", &d, false)) return false; p->temporal_samples = (int32_t)d;
  return true;
}

/* tileRowCount(params) -> rows this context renders */
static napi_value TileRowCount(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_frame_params p;
  if (!read_params(env, argv[0], &p)) return nullptr;
  napi_value r;
  NAPI_OK(env, napi_create_uint32(env, flx_tile_row_count(&p), &r));
  return r;
}

/* render(handle, params, out Float32Array(rows*width*4), wantCounters) -> { frameMs, traceMs, counters? } */
static napi_value Render(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  flx_frame_params p;
  if (!read_params(env, argv[1], &p)) return nullptr;
  void *out; size_t n;
  if (!typed(env, argv[2], napi_float32_array, &out, &n)) return nullptr;
  if (!out || n != (size_t)flx_tile_row_count(&p) * p.width * 4) { napi_throw_range_error(env, nullptr, "out needs rows*width*4 floats"); return nullptr; }
  bool want = false;
  napi_get_value_bool(env, argv[3], &want);
  flx_counters c;
  flx_status rc = flx_render(ctx, &p, (float *)out, nullptr, want ? &c : nullptr);
  if (rc != FLX_OK) return fail(env, ctx, "flx_render", rc);
  float frame_ms = 0.f, trace_ms = 0.f;
  flx_last_frame_ms(ctx, &frame_ms, &trace_ms);
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_double(env, frame_ms, &v); napi_set_named_property(env, res, "frameMs", v);
  napi_create_double(env, trace_ms, &v); napi_set_named_property(env, res, "traceMs", v);
  if (want) {
    napi_value co;
    napi_create_object(env, &co);
    const char *names[8] = { "primaryVisits", "closestVisits", "shadowVisits", "closestWalks", "shadowWalks", "shades", "primaryHits", "atlasTexels" };
    const uint64_t vals[8] = { c.primary_visits, c.closest_visits, c.shadow_visits, c.closest_walks, c.shadow_walks, c.shades, c.primary_hits, c.atlas_texels };
    for (int i = 0; i < 8; i++) { napi_create_double(env, (double)vals[i], &v); napi_set_named_property(env, co, names[i], v); }
    napi_set_named_property(env, res, "counters", co);
  }
  return res;
}

/* renderBatch(handle, [params, ...], out Float32Array, wantCounters) -> { frameMs, counters? }: flx_render_batch, the frames
 * one after the other in `out` */
static napi_value RenderBatch(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  uint32_t count = 0;
  bool isArray = false;
  napi_is_array(env, argv[1], &isArray);
  if (!isArray || napi_get_array_length(env, argv[1], &count) != napi_ok || count < 1 || count > FLX_MAX_BATCH_FRAMES) {
    napi_throw_range_error(env, nullptr, "renderBatch: an array of 1 .. 32 frame parameter objects");
    return nullptr;
  }
  flx_frame_params p[FLX_MAX_BATCH_FRAMES];
  for (uint32_t i = 0; i < count; i++) {
    napi_value e;
    NAPI_OK(env, napi_get_element(env, argv[1], i, &e));
    if (!read_params(env, e, &p[i])) return nullptr;
  }
  void *out; size_t n;
  if (!typed(env, argv[2], napi_float32_array, &out, &n)) return nullptr;
  if (!out || n != (size_t)count * flx_tile_row_count(&p[0]) * p[0].width * 4) { napi_throw_range_error(env, nullptr, "out needs frames*rows*width*4 floats"); return nullptr; }
  bool want = false;
  napi_get_value_bool(env, argv[3], &want);
  flx_counters c;
  flx_status rc = flx_render_batch(ctx, p, count, (float *)out, want ? &c : nullptr);
  if (rc != FLX_OK) return fail(env, ctx, "flx_render_batch", rc);
  float frame_ms = 0.f, trace_ms = 0.f;
  flx_last_frame_ms(ctx, &frame_ms, &trace_ms);
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_double(env, frame_ms, &v); napi_set_named_property(env, res, "frameMs", v);
  if (want) {
    napi_value co;
    napi_create_object(env, &co);
    const char *names[8] = { "primaryVisits", "closestVisits", "shadowVisits", "closestWalks", "shadowWalks", "shades", "primaryHits", "atlasTexels" };
    const uint64_t vals[8] = { c.primary_visits, c.closest_visits, c.shadow_visits, c.closest_walks, c.shadow_walks, c.shades, c.primary_hits, c.atlas_texels };
    for (int i = 0; i < 8; i++) { napi_create_double(env, (double)vals[i], &v); napi_set_named_property(env, co, names[i], v); }
    napi_set_named_property(env, res, "counters", co);
  }
  return res;
}

/* temporalReset(handle): forget the temporal history */
static napi_value TemporalReset(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  flx_status rc = flx_temporal_reset(ctx);
  if (rc != FLX_OK) return fail(env, ctx, "flx_temporal_reset", rc);
  return nullptr;
}

/* deviceInfo(handle) -> { name, computeUnits } */
static napi_value DeviceInfo(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  char name[256]; uint32_t cus = 0;
  flx_device_info(ctx, name, sizeof name, &cus);
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, res, "name", v);
  napi_create_uint32(env, cus, &v); napi_set_named_property(env, res, "computeUnits", v);
  return res;
}

/* ---- native scene import (flx_mesh_*): meshImport(objText, mtlText | null) -> handle; the Object3D operations; counts; flatten -- */
static void mesh_finalize(napi_env, void *data, void *) {
  flx_mesh **box = static_cast<flx_mesh **>(data);
  if (*box) flx_mesh_destroy(*box);
  delete box;
}
static flx_mesh *get_mesh(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p || !*static_cast<flx_mesh **>(p)) { napi_throw_type_error(env, nullptr, "expected a mesh handle"); return nullptr; }
  return *static_cast<flx_mesh **>(p);
}
static bool get_text(napi_env env, napi_value v, std::string &out, bool &present) {
  napi_valuetype t;
  napi_typeof(env, v, &t);
  present = !(t == napi_null || t == napi_undefined);
  if (!present) return true;
  size_t len = 0;
  if (napi_get_value_string_utf8(env, v, nullptr, 0, &len) != napi_ok) { napi_throw_type_error(env, nullptr, "expected a string"); return false; }
  out.resize(len + 1);
  napi_get_value_string_utf8(env, v, &out[0], len + 1, &len);
  out.resize(len);
  return true;
}
static napi_value MeshImport(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  std::string obj, mtl; bool hasObj = false, hasMtl = false;
  if (!get_text(env, argv[0], obj, hasObj) || !get_text(env, argv[1], mtl, hasMtl)) return nullptr;
  if (!hasObj) { napi_throw_type_error(env, nullptr, "meshImport: OBJ text expected"); return nullptr; }
  flx_mesh *m = nullptr;
  if (flx_mesh_import_obj(obj.data(), obj.size(), hasMtl ? mtl.data() : nullptr, mtl.size(), &m) != FLX_OK) { napi_throw_error(env, nullptr, "flx_mesh_import_obj failed"); return nullptr; }
  flx_mesh **box = new flx_mesh *(m);
  napi_value ext;
  NAPI_OK(env, napi_create_external(env, box, mesh_finalize, nullptr, &ext));
  return ext;
}
static napi_value MeshCounts(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_uint32(env, flx_mesh_entry_count(m), &v); napi_set_named_property(env, res, "entries", v);
  napi_create_uint32(env, flx_mesh_triangle_count(m), &v); napi_set_named_property(env, res, "triangles", v);
  return res;
}
static napi_value MeshSetTransform(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  uint32_t n = 0;
  NAPI_OK(env, napi_get_value_uint32(env, argv[1], &n));
  flx_mesh_set_transform(m, n);
  return nullptr;
}
static napi_value MeshMove(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  double d[3];
  for (int i = 0; i < 3; i++) NAPI_OK(env, napi_get_value_double(env, argv[1 + i], &d[i]));
  flx_mesh_move(m, d[0], d[1], d[2]);
  return nullptr;
}
static napi_value MeshScale(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  double s = 1;
  NAPI_OK(env, napi_get_value_double(env, argv[1], &s));
  flx_mesh_scale(m, s);
  return nullptr;
}
/* meshSetMaterial(handle, field, Float64Array values) */
static napi_value MeshSetMaterial(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  int32_t field = 0;
  NAPI_OK(env, napi_get_value_int32(env, argv[1], &field));
  void *vals = nullptr; size_t n = 0;
  if (!typed(env, argv[2], napi_float64_array, &vals, &n)) return nullptr;
  if (n < 3) { napi_throw_type_error(env, nullptr, "meshSetMaterial: Float64Array(3) expected"); return nullptr; }
  if (flx_mesh_set_material(m, field, static_cast<const double *>(vals)) != FLX_OK) { napi_throw_error(env, nullptr, "flx_mesh_set_material: unknown field"); return nullptr; }
  return nullptr;
}
static napi_value MeshBounding(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  double box[6];
  flx_mesh_bounding(m, box);
  napi_value res;
  NAPI_OK(env, napi_create_array_with_length(env, 6, &res));
  for (uint32_t i = 0; i < 6; i++) { napi_value v; napi_create_double(env, box[i], &v); napi_set_element(env, res, i, v); }
  return res;
}
/* meshFlatten(handle, Float32Array geometry[12 * entries], Float32Array attributes[28 * entries], Int32Array ids[triangles]) -> [min xyz, max xyz] */
static napi_value MeshFlatten(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_mesh *m = get_mesh(env, argv[0]);
  if (!m) return nullptr;
  void *g = nullptr, *a = nullptr, *ids = nullptr; size_t ng = 0, na = 0, ni = 0;
  if (!typed(env, argv[1], napi_float32_array, &g, &ng) || !typed(env, argv[2], napi_float32_array, &a, &na) || !typed(env, argv[3], napi_int32_array, &ids, &ni)) return nullptr;
  const size_t e = flx_mesh_entry_count(m), t = flx_mesh_triangle_count(m);
  if (ng < 12 * e || na < 28 * e || ni < t) { napi_throw_range_error(env, nullptr, "meshFlatten: arrays too small"); return nullptr; }
  float box[6] = { 0, 0, 0, 0, 0, 0 };
  if (flx_mesh_flatten(m, static_cast<float *>(g), static_cast<float *>(a), static_cast<int32_t *>(ids), box) != FLX_OK) { napi_throw_error(env, nullptr, "flx_mesh_flatten failed"); return nullptr; }
  napi_value res;
  NAPI_OK(env, napi_create_array_with_length(env, 6, &res));
  for (uint32_t i = 0; i < 6; i++) { napi_value v; napi_create_double(env, box[i], &v); napi_set_element(env, res, i, v); }
  return res;
}

/* fxaa / taa (handle, width, height, Float32Array in, Float32Array out); taaReset(handle) */
static napi_value post_pass(napi_env env, napi_callback_info info, int which) {
  napi_value argv[5];
  if (!get_args(env, info, 5, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  uint32_t w = 0, h = 0;
  NAPI_OK(env, napi_get_value_uint32(env, argv[1], &w));
  NAPI_OK(env, napi_get_value_uint32(env, argv[2], &h));
  void *in = nullptr, *out = nullptr; size_t ni = 0, no = 0;
  if (!typed(env, argv[3], napi_float32_array, &in, &ni) || !typed(env, argv[4], napi_float32_array, &out, &no)) return nullptr;
  if (ni < (size_t)w * h * 4 || no < (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "anti-aliasing pass: arrays smaller than width * height * 4"); return nullptr; }
  flx_status rc = which == 0 ? flx_fxaa(ctx, w, h, static_cast<const float *>(in), static_cast<float *>(out))
                             : flx_taa(ctx, w, h, static_cast<const float *>(in), static_cast<float *>(out));
  if (rc != FLX_OK) return fail(env, ctx, which == 0 ? "flx_fxaa" : "flx_taa", rc);
  return nullptr;
}
/* present(handle, width, height, in Float32Array, out Uint8Array | Uint8ClampedArray): the canvas' RGBA8 drawing buffer */
static napi_value Present(napi_env env, napi_callback_info info) {
  napi_value argv[5];
  if (!get_args(env, info, 5, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  uint32_t w = 0, h = 0;
  napi_get_value_uint32(env, argv[1], &w); napi_get_value_uint32(env, argv[2], &h);
  void *in; size_t n_in;
  if (!typed(env, argv[3], napi_float32_array, &in, &n_in)) return nullptr;
  napi_typedarray_type type; size_t n_out; void *out; napi_value buf; size_t off;
  if (napi_get_typedarray_info(env, argv[4], &type, &n_out, &out, &buf, &off) != napi_ok || (type != napi_uint8_array && type != napi_uint8_clamped_array)) {
    napi_throw_type_error(env, nullptr, "present: out must be a Uint8Array or Uint8ClampedArray");
    return nullptr;
  }
  if (!in || !out || n_in != (size_t)w * h * 4 || n_out != n_in) { napi_throw_range_error(env, nullptr, "present: in and out need width*height*4 elements"); return nullptr; }
  flx_status rc = flx_present(ctx, w, h, (const float *)in, (uint8_t *)out);
  if (rc != FLX_OK) return fail(env, ctx, "flx_present", rc);
  return nullptr;
}
static napi_value Fxaa(napi_env env, napi_callback_info info) { return post_pass(env, info, 0); }
static napi_value Taa(napi_env env, napi_callback_info info) { return post_pass(env, info, 1); }
static napi_value TaaReset(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  flx_taa_reset(ctx);
  return nullptr;
}

/* packTransforms(Float64Array matrices[9 T], Float64Array positions[3 T], Float32Array rotation[24 T], Float32Array shift[8 T]) */
static napi_value PackTransforms(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  void *m = nullptr, *p = nullptr, *r = nullptr, *s = nullptr; size_t nm = 0, np = 0, nr = 0, ns = 0;
  if (!typed(env, argv[0], napi_float64_array, &m, &nm) || !typed(env, argv[1], napi_float64_array, &p, &np) ||
      !typed(env, argv[2], napi_float32_array, &r, &nr) || !typed(env, argv[3], napi_float32_array, &s, &ns)) return nullptr;
  const size_t T = nm / 9;
  if (nm != 9 * T || np < 3 * T || nr < 24 * T || ns < 8 * T) { napi_throw_range_error(env, nullptr, "packTransforms: array sizes do not agree"); return nullptr; }
  if (flx_transforms_pack((uint32_t)T, static_cast<const double *>(m), static_cast<const double *>(p), static_cast<float *>(r), static_cast<float *>(s)) != FLX_OK) {
    napi_throw_error(env, nullptr, "flx_transforms_pack failed"); return nullptr;
  }
  return nullptr;
}


/* ---- the frame loop: frameBegin(handle, params, rgba8) / frameEnd(handle) -> { pixels, gpuMs } ---------------------------
 * `pixels` is a typed array over the context's pinned host buffer (no copy).  It belongs to that frame until the library re-uses
 * or frees the buffer — the frameBegin that will copy a newer frame into it (with two lanes: the fourth after the one that made
 * it; with one lane the second), a frameBegin that re-allocates it for a larger frame, or halt() — at which point the addon
 * DETACHES it: `pixels.length` becomes 0 instead of the array showing another frame or freed memory. */
static napi_value FrameBegin(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  CtxBox *box = get_box(env, argv[0]);
  if (!box) return nullptr;
  flx_context *ctx = box->ctx;
  flx_frame_params p;
  if (!read_params(env, argv[1], &p)) return nullptr;
  bool rgba8 = false;
  napi_get_value_bool(env, argv[2], &rgba8);
  flx_status rc = flx_frame_begin(ctx, &p, rgba8 ? FLX_FRAME_RGBA8 : FLX_FRAME_FLOAT);
  /* a larger frame (canvas or renderQuality changed) made the library re-allocate a pinned slot: the views over the old one go
   * (no JavaScript has run since the free — this is one native call) */
  const void *slots[4] = { nullptr, nullptr, nullptr, nullptr };
  int begun = -1;
  (void)flx_frame_host_slots(ctx, slots, &begun);
  /* ... and the slot the frame just begun will be copied into stops being the older frame's that was handed out over it */
  const void *reused = (rc == FLX_OK && begun >= 0) ? slots[begun] : nullptr;
  detach_views(env, box, [&](const void *q) { return q != reused && (q == slots[0] || q == slots[1] || q == slots[2] || q == slots[3]); });
  if (rc != FLX_OK) return fail(env, ctx, "flx_frame_begin", rc);
  return nullptr;
}
static void no_free(napi_env, void *, void *) {}
/* { pixels, gpuMs } over the pinned slot (or the group's image) the frame is in */
static napi_value frame_result(napi_env env, std::vector<CtxBox::View> &views, const void *pixels, size_t bytes, float ms, bool rgba8) {
  napi_value res, buf, arr, v;
  NAPI_OK(env, napi_create_object(env, &res));
  NAPI_OK(env, napi_create_external_arraybuffer(env, const_cast<void *>(pixels), bytes, no_free, nullptr, &buf));
  {
    CtxBox::View view = { pixels, nullptr };
    NAPI_OK(env, napi_create_reference(env, buf, 0, &view.ref));      /* weak: the view lives as long as JavaScript holds it */
    views.push_back(view);
  }
  if (rgba8) NAPI_OK(env, napi_create_typedarray(env, napi_uint8_clamped_array, bytes, buf, 0, &arr));
  else NAPI_OK(env, napi_create_typedarray(env, napi_float32_array, bytes / 4, buf, 0, &arr));
  napi_set_named_property(env, res, "pixels", arr);
  napi_create_double(env, ms, &v); napi_set_named_property(env, res, "gpuMs", v);
  return res;
}
static napi_value FrameEnd(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  CtxBox *box = get_box(env, argv[0]);
  if (!box) return nullptr;
  flx_context *ctx = box->ctx;
  bool rgba8 = false;
  napi_get_value_bool(env, argv[1], &rgba8);
  const void *pixels = nullptr; size_t bytes = 0; float ms = 0.f;
  flx_status rc = flx_frame_end(ctx, &pixels, &bytes, &ms);
  if (rc != FLX_OK) return fail(env, ctx, "flx_frame_end", rc);
  /* the slot now holds THIS frame: an older frame's view of the same memory (two frames back on this lane) is detached */
  detach_views(env, box, [&](const void *q) { return q != pixels; });
  return frame_result(env, box->views, pixels, bytes, ms, rgba8);
}

static napi_value FramesInFlight(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_context *ctx = get_ctx(env, argv[0]);
  if (!ctx) return nullptr;
  napi_value v;
  napi_create_int32(env, flx_frames_in_flight(ctx), &v);
  return v;
}

/* ---- several GPUs in this process (flx_group_*): the same calls with a group handle -------------------------------------- */
/* What a group handle points to: the group and the ArrayBuffers handed out over its frame images (groupFrameEnd), detached when their memory stops being
 * that frame's (CtxBox above does the same for one context). */
struct GroupBox {
  flx_group *g = nullptr;                  /* first member: a handle also reads as flx_group ** */
  std::vector<CtxBox::View> views;
  bool busy = false;                       /* a groupFrameEndAsync is pending */
};
static void detach_group_views(napi_env env, GroupBox *box) {
  for (auto &v : box->views) detach_view(env, v);
  box->views.clear();
}
static void finalize_group(napi_env env, void *data, void *) {
  GroupBox *box = static_cast<GroupBox *>(data);
  for (auto &v : box->views) napi_delete_reference(env, v.ref);
  if (box->g) flx_group_destroy(box->g);
  delete box;
}
static GroupBox *get_group_box(napi_env env, napi_value v) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, nullptr, "expected a group handle"); return nullptr; }
  GroupBox *box = static_cast<GroupBox *>(p);
  if (!box->g) { napi_throw_error(env, nullptr, "group was halted"); return nullptr; }
  if (box->busy) { napi_throw_error(env, nullptr, "a groupFrameEndAsync of this group is pending: await it first"); return nullptr; }
  return box;
}
static flx_group *get_group(napi_env env, napi_value v) {
  GroupBox *box = get_group_box(env, v);
  return box ? box->g : nullptr;
}
static napi_value gfail(napi_env env, flx_group *g, const char *what, flx_status rc) {
  std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + flx_group_last_error(g);
  napi_throw_error(env, nullptr, msg.c_str());
  return nullptr;
}
/* createGroup([device, ...]) -> handle */
static napi_value CreateGroup(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  uint32_t n = 0;
  bool isArray = false;
  napi_is_array(env, argv[0], &isArray);
  if (!isArray || napi_get_array_length(env, argv[0], &n) != napi_ok || n < 1 || n > 64) { napi_throw_range_error(env, nullptr, "createGroup: an array of 1 .. 64 device numbers"); return nullptr; }
  int devices[64];
  for (uint32_t i = 0; i < n; i++) { napi_value e; int32_t d = 0; napi_get_element(env, argv[0], i, &e); NAPI_OK(env, napi_get_value_int32(env, e, &d)); devices[i] = d; }
  flx_group *g = nullptr;
  flx_status rc = flx_group_create((int)n, devices, &g);
  if (rc != FLX_OK) return gfail(env, nullptr, "flx_group_create", rc);
  napi_value ext;
  GroupBox *box = new GroupBox();
  box->g = g;
  NAPI_OK(env, napi_create_external(env, box, finalize_group, nullptr, &ext));
  return ext;
}
static napi_value DestroyGroup(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  void *p = nullptr;
  if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) {
    GroupBox *box = static_cast<GroupBox *>(p);
    if (box->busy) { napi_throw_error(env, nullptr, "destroyGroup: a groupFrameEndAsync is pending: await it first"); return nullptr; }
    detach_group_views(env, box);          /* the images are freed with the group: whoever still holds a frame reads an empty array */
    if (box->g) { flx_group_destroy(box->g); box->g = nullptr; }
  }
  return nullptr;
}
static napi_value GroupInfo(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_group *g = get_group(env, argv[0]);
  if (!g) return nullptr;
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_int32(env, flx_group_size(g), &v); napi_set_named_property(env, res, "size", v);
  napi_get_boolean(env, flx_group_uses_rccl(g) != 0, &v); napi_set_named_property(env, res, "rccl", v);
  return res;
}
static napi_value GroupUploadScene(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_group *grp = get_group(env, argv[0]);
  if (!grp) return nullptr;
  void *g, *a, *ids; size_t ng, na, nids;
  if (!typed(env, argv[1], napi_float32_array, &g, &ng) || !typed(env, argv[2], napi_float32_array, &a, &na) ||
      !typed(env, argv[3], napi_int32_array, &ids, &nids)) return nullptr;
  if (ng % 12 != 0 || na / 28 != ng / 12 || na % 28 != 0) { napi_throw_range_error(env, nullptr, "geometry needs 12 and attributes 28 floats per entry"); return nullptr; }
  flx_status rc = flx_group_scene_upload(grp, (const float *)g, (const float *)a, (uint32_t)(ng / 12), (const int32_t *)ids, (uint32_t)nids);
  if (rc != FLX_OK) return gfail(env, grp, "flx_group_scene_upload", rc);
  return nullptr;
}
static napi_value GroupUploadTransforms(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  flx_group *grp = get_group(env, argv[0]);
  if (!grp) return nullptr;
  void *r, *s; size_t nr, ns;
  if (!typed(env, argv[1], napi_float32_array, &r, &nr) || !typed(env, argv[2], napi_float32_array, &s, &ns)) return nullptr;
  if (ns % 8 != 0 || nr != ns * 3) { napi_throw_range_error(env, nullptr, "rotation needs 24 and shift 8 floats per transform"); return nullptr; }
  flx_status rc = flx_group_transforms_upload(grp, (const float *)r, (const float *)s, (uint32_t)(ns / 8));
  if (rc != FLX_OK) return gfail(env, grp, "flx_group_transforms_upload", rc);
  return nullptr;
}
static napi_value GroupUploadLights(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  flx_group *grp = get_group(env, argv[0]);
  if (!grp) return nullptr;
  void *l; size_t nl;
  if (!typed(env, argv[1], napi_float32_array, &l, &nl)) return nullptr;
  if (nl % 6 != 0) { napi_throw_range_error(env, nullptr, "lights needs 6 floats per light"); return nullptr; }
  flx_status rc = flx_group_lights_upload(grp, (const float *)l, (uint32_t)(nl / 6));
  if (rc != FLX_OK) return gfail(env, grp, "flx_group_lights_upload", rc);
  return nullptr;
}
static napi_value GroupUploadAtlas(napi_env env, napi_callback_info info) {
  napi_value argv[5];
  if (!get_args(env, info, 5, argv)) return nullptr;
  flx_group *grp = get_group(env, argv[0]);
  if (!grp) return nullptr;
  int32_t which; uint32_t w, h;
  NAPI_OK(env, napi_get_value_int32(env, argv[1], &which));
  void *px; size_t n;
  if (!typed(env, argv[2], napi_uint8_array, &px, &n)) return nullptr;
  NAPI_OK(env, napi_get_value_uint32(env, argv[3], &w));
  NAPI_OK(env, napi_get_value_uint32(env, argv[4], &h));
  if (px && n != (size_t)w * h * 4) { napi_throw_range_error(env, nullptr, "atlas needs width*height*4 bytes"); return nullptr; }
  flx_status rc = flx_group_atlas_upload(grp, which, (const uint8_t *)px, w, h);
  if (rc != FLX_OK) return gfail(env, grp, "flx_group_atlas_upload", rc);
  return nullptr;
}
/* groupRender(handle, [params, ...], tileRows, out Float32Array(frames*height*width*4), wantCounters) -> { frameMs, counters? } */
static napi_value GroupRender(napi_env env, napi_callback_info info) {
  napi_value argv[5];
  if (!get_args(env, info, 5, argv)) return nullptr;
  flx_group *grp = get_group(env, argv[0]);
  if (!grp) return nullptr;
  uint32_t count = 0;
  bool isArray = false;
  napi_is_array(env, argv[1], &isArray);
  if (!isArray || napi_get_array_length(env, argv[1], &count) != napi_ok || count < 1 || count > FLX_MAX_BATCH_FRAMES) {
    napi_throw_range_error(env, nullptr, "groupRender: an array of 1 .. 32 frame parameter objects");
    return nullptr;
  }
  flx_frame_params p[FLX_MAX_BATCH_FRAMES];
  for (uint32_t i = 0; i < count; i++) {
    napi_value e;
    NAPI_OK(env, napi_get_element(env, argv[1], i, &e));
    if (!read_params(env, e, &p[i])) return nullptr;
  }
  uint32_t tileRows = 8;
  NAPI_OK(env, napi_get_value_uint32(env, argv[2], &tileRows));
  void *out; size_t n;
  if (!typed(env, argv[3], napi_float32_array, &out, &n)) return nullptr;
  if (!out || n != (size_t)count * p[0].height * p[0].width * 4) { napi_throw_range_error(env, nullptr, "out needs frames*height*width*4 floats"); return nullptr; }
  bool want = false;
  napi_get_value_bool(env, argv[4], &want);
  flx_counters c;
  flx_status rc = flx_group_render(grp, p, count, tileRows, (float *)out, want ? &c : nullptr);
  if (rc != FLX_OK) return gfail(env, grp, "flx_group_render", rc);
  float frame_ms = 0.f, trace_ms = 0.f;
  flx_last_frame_ms(flx_group_context(grp, 0), &frame_ms, &trace_ms);
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_double(env, frame_ms, &v); napi_set_named_property(env, res, "frameMs", v);
  napi_create_double(env, trace_ms, &v); napi_set_named_property(env, res, "traceMs", v);
  if (want) {
    napi_value co;
    napi_create_object(env, &co);
    const char *names[8] = { "primaryVisits", "closestVisits", "shadowVisits", "closestWalks", "shadowWalks", "shades", "primaryHits", "atlasTexels" };
    const uint64_t vals[8] = { c.primary_visits, c.closest_visits, c.shadow_visits, c.closest_walks, c.shadow_walks, c.shades, c.primary_hits, c.atlas_texels };
    for (int i = 0; i < 8; i++) { napi_create_double(env, (double)vals[i], &v); napi_set_named_property(env, co, names[i], v); }
    napi_set_named_property(env, res, "counters", co);
  }
  return res;
}

/* groupRenderRgba8(handle, params, tileRows, out Uint8ClampedArray(height*width*4)) -> { frameMs }: one frame as the canvas' RGBA8, the strips quantised on
 * their GPUs before the exchange (flx_group_render_rgba8: a quarter of the bytes gathered and copied out) */
static napi_value GroupRenderRgba8(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  flx_group *grp = get_group(env, argv[0]);
  if (!grp) return nullptr;
  flx_frame_params p;
  if (!read_params(env, argv[1], &p)) return nullptr;
  uint32_t tileRows = 8;
  NAPI_OK(env, napi_get_value_uint32(env, argv[2], &tileRows));
  void *out; size_t n;
  if (!typed(env, argv[3], napi_uint8_array, &out, &n)) return nullptr;
  if (!out || n != (size_t)p.height * p.width * 4) { napi_throw_range_error(env, nullptr, "out needs height*width*4 bytes"); return nullptr; }
  flx_status rc = flx_group_render_rgba8(grp, &p, 1, tileRows, (uint8_t *)out, nullptr);
  if (rc != FLX_OK) return gfail(env, grp, "flx_group_render_rgba8", rc);
  float frame_ms = 0.f, trace_ms = 0.f;
  flx_last_frame_ms(flx_group_context(grp, 0), &frame_ms, &trace_ms);
  napi_value res, v;
  NAPI_OK(env, napi_create_object(env, &res));
  napi_create_double(env, frame_ms, &v); napi_set_named_property(env, res, "frameMs", v);
  return res;
}

/* ---- the group's frame loop: groupFrameBegin(handle, params, tileRows, rgba8) / groupFrameEnd(handle, rgba8) -> { pixels, gpuMs } ------
 * flx_group_frame_begin / _end: every GPU's frame server resolves its strips straight into ONE image in pinned host memory; `pixels` is a Float32Array
 * (or, rgba8, a Uint8ClampedArray of the canvas' bytes: the servers quantise as they resolve) over that image (no copy), the frame's until the NEXT groupFrameBegin — which may be the frame that re-uses the image — detaches it (and
 * groupSetFrameLanes, destroyGroup). */
static napi_value GroupFrameBegin(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  GroupBox *box = get_group_box(env, argv[0]);
  if (!box) return nullptr;
  flx_frame_params p;
  if (!read_params(env, argv[1], &p)) return nullptr;
  uint32_t tileRows = 8;
  NAPI_OK(env, napi_get_value_uint32(env, argv[2], &tileRows));
  bool rgba8 = false;
  napi_get_value_bool(env, argv[3], &rgba8);
  detach_group_views(env, box);
  flx_status rc = flx_group_frame_begin(box->g, &p, tileRows, rgba8 ? FLX_FRAME_RGBA8 : FLX_FRAME_FLOAT);
  if (rc != FLX_OK) return gfail(env, box->g, "flx_group_frame_begin", rc);
  return nullptr;
}
static napi_value GroupFrameEnd(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  GroupBox *box = get_group_box(env, argv[0]);
  if (!box) return nullptr;
  bool rgba8 = false;
  napi_get_value_bool(env, argv[1], &rgba8);
  const void *pixels = nullptr; size_t bytes = 0; float ms = 0.f;
  flx_status rc = flx_group_frame_end(box->g, &pixels, &bytes, &ms);
  if (rc != FLX_OK) return gfail(env, box->g, "flx_group_frame_end", rc);
  return frame_result(env, box->views, pixels, bytes, ms, rgba8);
}
/* ---- frameEndAsync(handle, rgba8) / groupFrameEndAsync(handle, rgba8) -> Promise<{ pixels, gpuMs }> ------------------------------------------
 * The reference's loop hands the frame to the browser and returns to the event loop (modules/pathtracerWGL2.js:300-302); flx_frame_end WAITS for the GPU.
 * Here the wait happens on a libuv worker thread (napi_async_work) and the promise settles on the main thread, which meanwhile runs timers, I/O and the
 * application's own code.  While the promise is pending the context (group) belongs to the worker: every other call on it throws. */
struct EndWork {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref handle = nullptr;               /* keeps the context / group handle alive */
  CtxBox *box = nullptr;
  GroupBox *gbox = nullptr;
  bool rgba8 = false;
  flx_status rc = FLX_OK;
  const void *pixels = nullptr; size_t bytes = 0; float ms = 0.f;
  std::string err;
};
static void end_execute(napi_env, void *data) {      /* worker thread: no N-API calls here */
  EndWork *w = static_cast<EndWork *>(data);
  if (w->box) {
    w->rc = flx_frame_end(w->box->ctx, &w->pixels, &w->bytes, &w->ms);
    if (w->rc != FLX_OK) w->err = std::string("flx_frame_end failed (") + std::to_string(w->rc) + "): " + flx_last_error(w->box->ctx);
  } else {
    w->rc = flx_group_frame_end(w->gbox->g, &w->pixels, &w->bytes, &w->ms);
    if (w->rc != FLX_OK) w->err = std::string("flx_group_frame_end failed (") + std::to_string(w->rc) + "): " + flx_group_last_error(w->gbox->g);
  }
}
static void end_complete(napi_env env, napi_status status, void *data) {      /* main thread */
  EndWork *w = static_cast<EndWork *>(data);
  if (w->box) w->box->busy = false; else w->gbox->busy = false;
  napi_value out = nullptr;
  if (status == napi_ok && w->rc == FLX_OK) {
    if (w->box) detach_views(env, w->box, [&](const void *q) { return q != w->pixels; });
    out = frame_result(env, w->box ? w->box->views : w->gbox->views, w->pixels, w->bytes, w->ms, w->rgba8);
  }
  if (out) napi_resolve_deferred(env, w->deferred, out);
  else {
    bool pending = false;
    napi_is_exception_pending(env, &pending);
    napi_value ex = nullptr;
    if (pending) napi_get_and_clear_last_exception(env, &ex);
    if (!ex) {
      napi_value msg;
      napi_create_string_utf8(env, w->err.empty() ? "frameEndAsync: cancelled" : w->err.c_str(), NAPI_AUTO_LENGTH, &msg);
      napi_create_error(env, nullptr, msg, &ex);
    }
    napi_reject_deferred(env, w->deferred, ex);
  }
  napi_delete_reference(env, w->handle);
  napi_delete_async_work(env, w->work);
  delete w;
}
static napi_value end_async(napi_env env, napi_callback_info info, bool group) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  CtxBox *box = group ? nullptr : get_box(env, argv[0]);
  GroupBox *gbox = group ? get_group_box(env, argv[0]) : nullptr;
  if (!box && !gbox) return nullptr;
  EndWork *w = new EndWork();
  w->box = box; w->gbox = gbox;
  napi_get_value_bool(env, argv[1], &w->rgba8);
  napi_value promise, name;
  if (napi_create_promise(env, &w->deferred, &promise) != napi_ok || napi_create_reference(env, argv[0], 1, &w->handle) != napi_ok ||
      napi_create_string_utf8(env, group ? "flx_group_frame_end" : "flx_frame_end", NAPI_AUTO_LENGTH, &name) != napi_ok ||
      napi_create_async_work(env, nullptr, name, end_execute, end_complete, w, &w->work) != napi_ok) {
    delete w;
    napi_throw_error(env, nullptr, "frameEndAsync: could not create the work item");
    return nullptr;
  }
  if (box) box->busy = true; else gbox->busy = true;
  if (napi_queue_async_work(env, w->work) != napi_ok) {
    if (box) box->busy = false; else gbox->busy = false;
    napi_delete_reference(env, w->handle); napi_delete_async_work(env, w->work); delete w;
    napi_throw_error(env, nullptr, "frameEndAsync: could not queue the work item");
    return nullptr;
  }
  return promise;
}
static napi_value FrameEndAsync(napi_env env, napi_callback_info info) { return end_async(env, info, false); }
static napi_value GroupFrameEndAsync(napi_env env, napi_callback_info info) { return end_async(env, info, true); }
static napi_value GroupFramesInFlight(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  flx_group *g = get_group(env, argv[0]);
  if (!g) return nullptr;
  napi_value v;
  napi_create_int32(env, flx_group_frames_in_flight(g), &v);
  return v;
}
static napi_value GroupSetFrameLanes(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  GroupBox *box = get_group_box(env, argv[0]);
  if (!box) return nullptr;
  int32_t lanes = 3;
  NAPI_OK(env, napi_get_value_int32(env, argv[1], &lanes));
  flx_status rc = flx_group_set_frame_lanes(box->g, lanes);
  if (rc != FLX_OK) return gfail(env, box->g, "flx_group_set_frame_lanes", rc);
  detach_group_views(env, box);            /* (the images are made again for the new depth) */
  return nullptr;
}

static napi_value Version(napi_env env, napi_callback_info) {
  napi_value v;
  napi_create_string_utf8(env, flx_version(), NAPI_AUTO_LENGTH, &v);
  return v;
}

static napi_value Init(napi_env env, napi_value exports) {
  const struct { const char *name; napi_callback fn; } fns[] = {
    { "createContext", CreateContext }, { "destroyContext", DestroyContext }, { "uploadScene", UploadScene },
    { "uploadTransforms", UploadTransforms }, { "uploadLights", UploadLights }, { "uploadAtlas", UploadAtlas },
    { "tileRowCount", TileRowCount }, { "render", Render }, { "renderBatch", RenderBatch }, { "temporalReset", TemporalReset }, { "deviceInfo", DeviceInfo }, { "version", Version },
    { "meshImport", MeshImport }, { "meshCounts", MeshCounts }, { "meshSetTransform", MeshSetTransform }, { "meshMove", MeshMove },
    { "meshScale", MeshScale }, { "meshSetMaterial", MeshSetMaterial }, { "meshFlatten", MeshFlatten }, { "meshBounding", MeshBounding }, { "packTransforms", PackTransforms },
    { "present", Present }, { "fxaa", Fxaa }, { "taa", Taa }, { "taaReset", TaaReset },
    { "frameBegin", FrameBegin }, { "frameEnd", FrameEnd }, { "frameEndAsync", FrameEndAsync }, { "groupFrameEndAsync", GroupFrameEndAsync }, { "framesInFlight", FramesInFlight },
    { "createGroup", CreateGroup }, { "destroyGroup", DestroyGroup }, { "groupInfo", GroupInfo }, { "groupUploadScene", GroupUploadScene },
    { "groupUploadTransforms", GroupUploadTransforms }, { "groupUploadLights", GroupUploadLights }, { "groupUploadAtlas", GroupUploadAtlas },
    { "groupRender", GroupRender }, { "groupRenderRgba8", GroupRenderRgba8 }, { "groupFrameBegin", GroupFrameBegin }, { "groupFrameEnd", GroupFrameEnd }, { "groupFramesInFlight", GroupFramesInFlight },
    { "groupSetFrameLanes", GroupSetFrameLanes },
  };
  for (const auto &f : fns) {
    napi_value fn;
    if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn) != napi_ok) return nullptr;
    napi_set_named_property(env, exports, f.name, fn);
  }
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
