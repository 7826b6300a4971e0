'use strict';
/*
 * PathTracerHIP — the FlexLight renderer object whose frames are traced by libflexlight_hip.so on an
 * MI355X instead of by WebGL2.  It has the shape FlexLight expects from a renderer
 * (reference modules/pathtracerWGL2.js:25-78,143,167,191; SURVEY.md §8b): type, public config /
 * camera / scene, fps, fpsLimit, canvas getter, async render(), halt(), async updateScene(),
 * async updatePrimaryLightSources() — plus renderFrame(), a synchronous single frame for headless use.
 * Conventions kept from the reference: config, camera and scene are re-read every frame; lights,
 * transforms and (when their list changed) atlases are re-derived every frame (pathtracerWGL2.js:
 * 258-262, 361-365); errors of the frame loop go to console.error, renderFrame() throws.
 * The "canvas" is any object with width and height; if it has onFrame(frame) the loop calls it.
 */
const path = require('path');
const { Transform } = require('./scene.js');
const sceneFile = require('./sceneFile.js');

let addon = null;
function native () {
  if (!addon) {
    const file = path.join(__dirname, '..', 'napi', 'flexlight_napi.node');
    try {
      addon = require(file);
    } catch (e) {
      throw new Error('flexlight_napi.node is not built or cannot load libflexlight_hip.so (' + e.message +
        '). Build with `python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback.');
    }
  }
  return addon;
}

class PathTracerHIP {
  constructor (canvas, scene, camera, config, options) {
    this.type = 'pathtracer';
    this.config = config;
    this.camera = camera;
    this.scene = scene;
    this.fps = 0;
    this.fpsLimit = Infinity;
    this._canvas = canvas;
    this._device = (options && options.device) || 0;
    this._tile = (options && options.tile) || null;      // {rows, index, count}: this context's strips of the frame
    /* several GPUs in this process (SURVEY.md 8e; not in the reference, which has one WebGL2 context): devices: N = GPUs 0 .. N - 1,
     * or a list of device numbers (a number may repeat: a rehearsal on a one-GPU box).  The frame is cut into strips of tileRows
     * rows dealt round robin to the GPUs and gathered inside the library (flx_group_render: RCCL all-gather + reassembly). */
    const dv = options && options.devices;
    this._devices = Array.isArray(dv) ? dv.slice() : (dv > 1 ? Array.from({ length: dv }, (_, i) => i) : null);
    this._tileRows = (options && options.tileRows) || 8;
    this.present8 = !!(options && options.present8);     // the frame loop hands out the canvas' RGBA8 instead of float radiance
    this.groupLanes = 3;                                  // frames in flight of a group's frame loop (2 or 3)
    this._group = null;
    this._ctx = null;
    this._halt = true;
    this._atlasLists = [null, null, null];
    this._haveScene = false;
    this._temporalFrame = 0;                             // pathtracerWGL2.js:291,562
    this.lastFrame = null;
  }

  get canvas () { return this._canvas; }

  _context () {
    if (this._devices) throw new Error('this renderer drives a group of GPUs: per-context calls are not available');
    if (!this._ctx) this._ctx = native().createContext(this._device);
    return this._ctx;
  }

  /* the GPU(s) behind the renderer: one context or a group of them, the same four uploads and one render call */
  _gpu () {
    const n = native();
    if (this._devices) {
      if (!this._group) this._group = n.createGroup(this._devices);
      const g = this._group;
      return {
        uploadScene: (a, b, c) => n.groupUploadScene(g, a, b, c), uploadTransforms: (a, b) => n.groupUploadTransforms(g, a, b),
        uploadLights: a => n.groupUploadLights(g, a), uploadAtlas: (w, px, x, y) => n.groupUploadAtlas(g, w, px, x, y),
        render: (p, out, counters) => n.groupRender(g, [p], this._tileRows, out, counters),
        renderBatch: (ps, out, counters) => n.groupRender(g, ps, this._tileRows, out, counters),
        rows: p => p.height
      };
    }
    const c = this._context();
    return {
      uploadScene: (a, b, d) => n.uploadScene(c, a, b, d), uploadTransforms: (a, b) => n.uploadTransforms(c, a, b),
      uploadLights: a => n.uploadLights(c, a), uploadAtlas: (w, px, x, y) => n.uploadAtlas(c, w, px, x, y),
      render: (p, out, counters) => n.render(c, p, out, counters), renderBatch: (ps, out, counters) => n.renderBatch(c, ps, out, counters),
      rows: p => n.tileRowCount(p)
    };
  }

  get gpuInfo () {                                         // { size, rccl } of a group, { size: 1 } otherwise
    return this._devices ? native().groupInfo((this._gpu(), this._group)) : { size: 1, rccl: false };
  }

  halt () {                                               // pathtracerWGL2.js:70-77
    this._halt = true;
    if (this._pendingEnd) { this._releaseWhenIdle = true; return; }      // a frame is being waited for on a worker thread (render()'s frameEndAsync): released when it settles
    this._release();
  }

  _release () {
    if (this._ctx) {
      try { native().destroyContext(this._ctx); } catch (e) { console.warn('Unable to release the GPU context', e.message); }
      this._ctx = null;
    }
    if (this._group) {
      try { native().destroyGroup(this._group); } catch (e) { console.warn('Unable to release the GPU group', e.message); }
      this._group = null;
    }
    this._haveScene = false;
    this._atlasLists = [null, null, null];
    this._inFlight = 0;
    this._lanesSet = undefined;
  }

  async updateScene () {                                  // pathtracerWGL2.js:167-189
    const built = await this.scene.generateArraysFromGraph();
    this._gpu().uploadScene(built.geometryBuffer, built.sceneBuffer, built.idBuffer);
    this._haveScene = true;
  }

  async updatePrimaryLightSources () {                    // pathtracerWGL2.js:143-165
    this._gpu().uploadLights(sceneFile.buildLightArray(this.scene));
  }

  _updateAtlases () {                                     // pathtracerWGL2.js:106-140: rebuild only when the list object or its members changed
    if (this.scene.prebuiltAtlases) {                    // a replayed scene (sceneFile.sceneFromFlxs): the atlases as they were built
      if (this._atlasLists[0] !== this.scene.prebuiltAtlases) {
        this.scene.prebuiltAtlases.forEach((a, which) => this._gpu().uploadAtlas(which, a.data, a.width, a.height));
        this._atlasLists = [this.scene.prebuiltAtlases, null, null];
      }
      return;
    }
    const lists = [this.scene.textures, this.scene.pbrTextures, this.scene.translucencyTextures];
    lists.forEach((list, which) => {
      const old = this._atlasLists[which];
      if (old && old.length === list.length && list.every((e, i) => e === old[i])) return;
      this._atlasLists[which] = list.slice();
      if (list.length === 0) { this._gpu().uploadAtlas(which, null, 0, 0); return; }
      const atlas = sceneFile.buildAtlas(list, this.scene.standardTextureSizes);
      this._gpu().uploadAtlas(which, atlas.data, atlas.width, atlas.height);
    });
  }

  frameParams (jitter) {                                  // pathtracerWGL2.js:307-347
    /* config.renderQuality scales the resolution the frame is traced at (pathtracerWGL2.js:264-267, 810-811: the canvas'
     * drawing buffer is clientWidth x renderQuality); the frame handed back has that size */
    const q = this.config.renderQuality > 0 ? this.config.renderQuality : 1;
    const w = Math.max(1, Math.round(this._canvas.width * q)), h = Math.max(1, Math.round(this._canvas.height * q));
    const cam = jitter ? Object.assign(Object.create(this.camera), { fx: this.camera.fx + jitter.x, fy: this.camera.fy + jitter.y }) : this.camera;
    const p = {
      width: w, height: h,
      camera: [this.camera.x, this.camera.y, this.camera.z],
      viewMatrix: Array.from(sceneFile.buildViewMatrix(cam, w, h)),                 // view rotation and TAA jitter (pathtracerWGL2.js:310-318)
      samples: this.config.samplesPerRay,
      maxReflections: this.config.maxReflections,
      minImportancy: this.config.minImportancy,
      useFilter: this.config.filter ? 1 : 0,
      isTemporal: this.config.temporal ? 1 : 0,
      temporalSamples: this.config.temporalSamples,
      hdr: this.config.hdr ? 1 : 0,
      ambient: [this.scene.ambientLight[0], this.scene.ambientLight[1], this.scene.ambientLight[2]],
      randomSeed: this.config.temporal ? this._temporalFrame : 0,          // pathtracerWGL2.js:347
      textureWidth: Math.floor(2048 / this.scene.standardTextureSizes[0])
    };
    if (this._tile) { p.tileRows = this._tile.rows; p.tileIndex = this._tile.index; p.tileCount = this._tile.count; }
    return p;
  }

  /* config.antialiasing (pathtracerWGL2.js:268-286): 'fxaa' | 'taa' | anything else = none.  TAA turns the camera by a sub-pixel
   * offset every frame (taa.js:120-127); the nine offsets sum to zero and are drawn once per renderer (taa.js:130-149) from
   * this.random, Math.random unless the application supplies its own. */
  _antialiasing () {
    const v = typeof this.config.antialiasing === 'string' ? this.config.antialiasing.toLowerCase() : undefined;
    const mode = (v === 'fxaa' || v === 'taa') ? v : undefined;
    if (mode !== this._aaMode) {
      this._aaMode = mode;
      this._taaNum = 0;
      if (mode === 'taa') { this._taaVecs = taaVectors(9, this.random || Math.random); if (this._ctx) native().taaReset(this._ctx); }
    }
    return mode;
  }

  _jitter () {                                            // taa.js:120-127
    this._taaNum = (this._taaNum + 1) % 9;
    const scale = 0.3 / Math.min(this._canvas.width, this._canvas.height);
    return { x: this._taaVecs[this._taaNum][0] * scale, y: this._taaVecs[this._taaNum][1] * scale };
  }

  /* One frame, synchronously.  Returns {width, height, rows, radiance: Float32Array(rows*width*4), frameMs, traceMs, counters?}. */
  /* scene arrays of this frame: scene once, then what the reference re-derives every frame (pathtracerWGL2.js:258-262, 361-365) */
  _uploadFrameState () {
    const gpu = this._gpu();
    if (!this._haveScene) {
      const built = this.scene.generateArraysFromGraph();
      gpu.uploadScene(built.geometryBuffer, built.sceneBuffer, built.idBuffer);
      this._haveScene = true;
    }
    this._updateAtlases();
    gpu.uploadLights(sceneFile.buildLightArray(this.scene));
    const tr = Transform.buildWGL2Arrays();
    gpu.uploadTransforms(tr[0], tr[1]);
    return gpu;
  }

  renderFrame (options) {
    const gpu = this._uploadFrameState();
    const aa = this._antialiasing();
    const jitter = aa === 'taa' ? this._jitter() : { x: 0, y: 0 };
    const p = this.frameParams(jitter);
    const rows = gpu.rows(p);
    /* options.reuse: write into the array of the frame before (a frame loop that consumes each frame before asking for the
     * next saves a 33 MB allocation per 1080p frame); by default every frame gets its own */
    const n = rows * p.width * 4;
    let radiance = (options && options.reuse && this._out && this._out.length === n) ? this._out : new Float32Array(n);
    this._out = radiance;
    const info = gpu.render(p, radiance, !!(options && options.counters));
    if (aa && rows === p.height) {                        // the pass reads neighbouring texels: whole frames only (pathtracerWGL2.js:552-553)
      if (this._devices) throw new Error('antialiasing passes run on one context: use a single device');
      const out = new Float32Array(radiance.length);
      if (aa === 'fxaa') native().fxaa(this._context(), p.width, p.height, radiance, out);
      else native().taa(this._context(), p.width, p.height, radiance, out);
      radiance = out;
    }
    this._temporalFrame = (this._temporalFrame + 1) % Math.max(1, this.config.temporalSamples);     // pathtracerWGL2.js:291
    this.lastFrame = Object.assign({ width: p.width, height: p.height, rows, radiance }, info);
    return this.lastFrame;
  }

  /* The RGBA8 the reference's canvas would hold for a frame of renderFrame() (whole frames): { width, height, data: Uint8ClampedArray },
   * the shape of an ImageData. */
  presentFrame (frame) {
    const f = frame || this.lastFrame;
    if (!f || f.rows !== f.height) throw new Error('presentFrame: a whole frame of renderFrame() is needed');
    const data = new Uint8ClampedArray(f.width * f.height * 4);
    native().present(this._context(), f.width, f.height, f.radiance, data);
    return { width: f.width, height: f.height, data };
  }

  /* Several frames of a camera path in ONE pass of the GPU pipeline (flx_render_batch; not in the reference, which renders frame
   * after frame): `cameras` is an array of up to 32 camera states { x, y, z, fx, fy } (missing fields default to this.camera's;
   * fov comes from this.camera).  Frames without temporal accumulation and anti-aliasing only — those depend on the frames
   * before (filter frames are fine: their chain starts from the same state every frame).  Returns { width, height, rows, frames: [Float32Array(rows*width*4), ...], frameMs, counters? }; every frame equals
   * the renderFrame() of its camera. */
  renderBatch (cameras, options) {
    if (this.config.temporal || this._antialiasing()) throw new Error('renderBatch: temporal and antialiasing frames depend on the frames before');
    const gpu = this._uploadFrameState();
    const saved = this.camera;
    const params = cameras.map(c => {
      this.camera = Object.assign(Object.create(saved), c);
      try { return this.frameParams(); } finally { this.camera = saved; }
    });
    const rows = gpu.rows(params[0]);
    const per = rows * params[0].width * 4;
    const all = new Float32Array(per * params.length);
    const info = gpu.renderBatch(params, all, !!(options && options.counters));
    const frames = params.map((_, i) => all.subarray(i * per, (i + 1) * per));
    return Object.assign({ width: params[0].width, height: params[0].height, rows, frames }, info);
  }

  /* The frame loop (pathtracerWGL2.js:191-831).  Like the reference's, it does not wait for the GPU inside a frame: frame
   * k + 1 is prepared and enqueued (flx_frame_begin) while frame k is traced and copied to pinned host memory, then frame k is
   * taken (flx_frame_end) and handed to canvas.onFrame — `pixels` is a view of that pinned memory (Float32Array, or the canvas'
   * RGBA8 as a Uint8ClampedArray with this.present8), valid until the frame after the next is begun.  A group of GPUs (`devices`) runs the
   * same loop through flx_group_frame_begin / _end with up to this.groupLanes (3) frames in flight, `pixels` valid until the next frame is begun.
   * Anti-aliasing passes and tiles take the synchronous renderFrame() per cycle instead.  `fps` as in pathtracerWGL2.js:293-298;
   * `gpuMs` = GPU time of the last frame taken. */
  async render () {
    if (!this._halt) return;                              // already running (the WebGPU renderer guards the same way)
    this._halt = false;
    await this.updateScene();
    let frames = 0, windowStart = Date.now();
    this._inFlight = 0;
    const pending = [];                                    // sizes of the frames begun and not yet taken
    const deliver = frame => {
      this.lastFrame = frame;
      if (typeof this._canvas.onFrame === 'function') this._canvas.onFrame(frame);
      frames++;
      const now = Date.now();
      if (now - windowStart >= 500) {                     // pathtracerWGL2.js:293-298
        this.fps = (1000 * frames / (now - windowStart)).toFixed(0);
        frames = 0; windowStart = now;
      }
    };
    /* flx_frame_end waits for the GPU: on a worker thread (frameEndAsync: napi_async_work), so that this thread is back in the event loop for most of every frame,
     * as the reference's is (pathtracerWGL2.js:300-302).  this.blockingFrameEnd = true: the synchronous call, for measurements. */
    const take = async () => {
      const q = pending.shift();
      let r;
      if (this.blockingFrameEnd) r = q.group ? native().groupFrameEnd(this._group, q.rgba8) : native().frameEnd(this._ctx, q.rgba8);
      else {
        this._pendingEnd = q.group ? native().groupFrameEndAsync(this._group, q.rgba8) : native().frameEndAsync(this._ctx, q.rgba8);
        try { r = await this._pendingEnd; } finally { this._pendingEnd = null; }
      }
      this._inFlight--;
      if (this._releaseWhenIdle) {                          // halt() came while the frame was being waited for: its memory goes now
        this._releaseWhenIdle = false;
        this._release();
        return;
      }
      this.gpuMs = r.gpuMs;
      deliver({ width: q.width, height: q.height, rows: q.rows, radiance: q.rgba8 ? undefined : r.pixels, rgba8: q.rgba8 ? r.pixels : undefined, pixels: r.pixels, frameMs: r.gpuMs });
    };
    const cycle = async () => {
      if (this._halt) return;
      try {
        const aa = this._antialiasing();
        /* a group of GPUs (flx_group_frame_begin / _end): every GPU's frame server resolves its strips straight into one image in pinned host memory, up to
         * three frames in flight, nothing waits for a GPU inside a frame; `pixels` is a view of that image, the frame's until the next frame is begun */
        const grouped = !!this._devices && !this._tile && !aa;      // (present8: the servers quantise their tiles as they resolve them — the canvas' bytes, a quarter of what every GPU writes)
        const pipelined = !this._devices && !this._tile && !aa;
        if (grouped) {
          this._uploadFrameState();
          const p = this.frameParams();
          if (this._lanesSet !== this.groupLanes) {         // (the library's default is 3)
            while (this._inFlight > 0 && !this._halt) await take();
            if (this._halt) return;
            native().groupSetFrameLanes(this._group, this.groupLanes);
            this._lanesSet = this.groupLanes;
          }
          if (this._inFlight === this.groupLanes) await take();
          if (this._halt) return;                           // (the application halted the renderer from its onFrame)
          native().groupFrameBegin(this._group, p, this._tileRows, this.present8);
          this._inFlight++;
          pending.push({ width: p.width, height: p.height, rows: p.height, rgba8: this.present8, group: true });
          this._temporalFrame = (this._temporalFrame + 1) % Math.max(1, this.config.temporalSamples);
        } else if (pipelined) {
          this._uploadFrameState();
          const p = this.frameParams();
          native().frameBegin(this._context(), p, this.present8);
          this._inFlight++;
          pending.push({ width: p.width, height: p.height, rows: p.height, rgba8: this.present8 });
          this._temporalFrame = (this._temporalFrame + 1) % Math.max(1, this.config.temporalSamples);
          if (this._inFlight === 2) await take();
        } else {
          while (this._inFlight > 0 && !this._halt) await take();
          if (this._halt) return;
          deliver(this.renderFrame({ reuse: true }));
        }
      } catch (e) {
        console.error(e);
        this._halt = true;
        return;
      }
      if (this._halt) return;
      if (this.fpsLimit === Infinity) setImmediate(cycle);
      else setTimeout(cycle, 1000 / this.fpsLimit);
    };
    setImmediate(cycle);
  }
}

/* taa.js:130-149: n two-dimensional offsets that add up to zero */
function taaVectors (n, random) {
  const vecs = new Array(n).fill(0).map(() => new Array(2));
  vecs[0] = [0, 1];
  vecs[1] = [1, 0];
  const combined = [1, 1];
  for (let i = 2; i < n; i++) {
    for (let j = 0; j < 2; j++) {
      const lo = Math.max(-Math.min(i + 1, n - 1 - i), combined[j] - 1);
      const hi = Math.min(Math.min(i + 1, n - 1 - i), combined[j] + 1);
      vecs[i][j] = 0.5 * ((hi + lo) + (hi - lo) * Math.sign(random() - 0.5) * Math.pow(random() * 0.5, 1 / 2)) - combined[j];
      combined[j] += vecs[i][j];
    }
  }
  return vecs;
}

module.exports = { PathTracerHIP, taaVectors };
