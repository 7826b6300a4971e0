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
    this._ctx = null;
    this._halt = true;
    this._atlasLists = [null, null, null];
    this._haveScene = false;
    this._temporalFrame = 0;                             // pathtracerWGL2.js:291,562
    this.lastFrame = null;
  }

  get canvas () { return this._canvas; }

  _context () {
    if (!this._ctx) this._ctx = native().createContext(this._device);
    return this._ctx;
  }

  halt () {                                               // pathtracerWGL2.js:70-77
    this._halt = true;
    if (this._ctx) {
      try { native().destroyContext(this._ctx); } catch (e) { console.warn('Unable to release the GPU context', e.message); }
      this._ctx = null;
      this._haveScene = false;
      this._atlasLists = [null, null, null];
    }
  }

  async updateScene () {                                  // pathtracerWGL2.js:167-189
    const built = await this.scene.generateArraysFromGraph();
    native().uploadScene(this._context(), built.geometryBuffer, built.sceneBuffer, built.idBuffer);
    this._haveScene = true;
  }

  async updatePrimaryLightSources () {                    // pathtracerWGL2.js:143-165
    native().uploadLights(this._context(), sceneFile.buildLightArray(this.scene));
  }

  _updateAtlases () {                                     // pathtracerWGL2.js:106-140: rebuild only when the list object or its members changed
    const lists = [this.scene.textures, this.scene.pbrTextures, this.scene.translucencyTextures];
    lists.forEach((list, which) => {
      const old = this._atlasLists[which];
      if (old && old.length === list.length && list.every((e, i) => e === old[i])) return;
      this._atlasLists[which] = list.slice();
      if (list.length === 0) { native().uploadAtlas(this._context(), which, null, 0, 0); return; }
      const atlas = sceneFile.buildAtlas(list, this.scene.standardTextureSizes);
      native().uploadAtlas(this._context(), which, atlas.data, atlas.width, atlas.height);
    });
  }

  frameParams (jitter) {                                  // pathtracerWGL2.js:307-347
    const w = this._canvas.width, h = this._canvas.height;
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
  renderFrame (options) {
    const ctx = this._context();
    if (!this._haveScene) {
      const built = this.scene.generateArraysFromGraph();
      native().uploadScene(ctx, built.geometryBuffer, built.sceneBuffer, built.idBuffer);
      this._haveScene = true;
    }
    this._updateAtlases();
    native().uploadLights(ctx, sceneFile.buildLightArray(this.scene));
    const tr = Transform.buildWGL2Arrays();
    native().uploadTransforms(ctx, tr[0], tr[1]);
    const aa = this._antialiasing();
    const jitter = aa === 'taa' ? this._jitter() : { x: 0, y: 0 };
    const p = this.frameParams(jitter);
    const rows = native().tileRowCount(p);
    let radiance = new Float32Array(rows * p.width * 4);
    const info = native().render(ctx, p, radiance, !!(options && options.counters));
    if (aa && rows === p.height) {                        // the pass reads neighbouring texels: whole frames only (pathtracerWGL2.js:552-553)
      const out = new Float32Array(radiance.length);
      if (aa === 'fxaa') native().fxaa(ctx, p.width, p.height, radiance, out);
      else native().taa(ctx, p.width, p.height, radiance, out);
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
   * fov comes from this.camera).  Frames without filter, temporal accumulation and anti-aliasing only — those depend on the frame
   * before.  Returns { width, height, rows, frames: [Float32Array(rows*width*4), ...], frameMs, counters? }; every frame equals
   * the renderFrame() of its camera. */
  renderBatch (cameras, options) {
    if (this.config.filter || this.config.temporal || this._antialiasing()) throw new Error('renderBatch: filter, temporal and antialiasing frames depend on the frame before');
    const ctx = this._context();
    if (!this._haveScene) {
      const built = this.scene.generateArraysFromGraph();
      native().uploadScene(ctx, built.geometryBuffer, built.sceneBuffer, built.idBuffer);
      this._haveScene = true;
    }
    this._updateAtlases();
    native().uploadLights(ctx, sceneFile.buildLightArray(this.scene));
    const tr = Transform.buildWGL2Arrays();
    native().uploadTransforms(ctx, tr[0], tr[1]);
    const saved = this.camera;
    const params = cameras.map(c => {
      this.camera = Object.assign(Object.create(saved), c);
      try { return this.frameParams(); } finally { this.camera = saved; }
    });
    const rows = native().tileRowCount(params[0]);
    const per = rows * params[0].width * 4;
    const all = new Float32Array(per * params.length);
    const info = native().renderBatch(ctx, params, all, !!(options && options.counters));
    const frames = params.map((_, i) => all.subarray(i * per, (i + 1) * per));
    return Object.assign({ width: params[0].width, height: params[0].height, rows, frames }, info);
  }

  async render () {                                       // pathtracerWGL2.js:191-831: start the frame loop
    if (!this._halt) return;                              // already running (the WebGPU renderer guards the same way)
    this._halt = false;
    await this.updateScene();
    let frames = 0, windowStart = Date.now();
    const cycle = () => {
      if (this._halt) return;
      try {
        const frame = this.renderFrame();
        if (typeof this._canvas.onFrame === 'function') this._canvas.onFrame(frame);
      } catch (e) {
        console.error(e);
        this._halt = true;
        return;
      }
      frames++;
      const now = Date.now();
      if (now - windowStart >= 500) {                     // pathtracerWGL2.js:293-298
        this.fps = (1000 * frames / (now - windowStart)).toFixed(0);
        frames = 0; windowStart = now;
      }
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
