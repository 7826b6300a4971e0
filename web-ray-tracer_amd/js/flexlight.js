'use strict';
/*
 * FlexLight — engine facade for the HIP back-end, same surface as the reference's flexlight.js:13-144:
 *   const engine = new FlexLight(canvas);  engine.scene.queue.push(...);
 *   engine.renderer = 'pathtracer';        engine.renderer.render();
 * `canvas` is any {width, height} object (headless); api is 'hip'.  Only the path tracer exists here:
 * asking for the rasterizer is reported like an unknown renderer in the reference (console.error, the
 * current renderer stays).  io / ui are browser input handling and are accepted but ignored.
 */
const { Camera } = require('./camera.js');
const { Config } = require('./config.js');
const { Scene, Transform, Primitive, Triangle, Plane, Object3D, Cuboid, Bounding } = require('./scene.js');
const { PathTracerHIP } = require('./pathtracerHIP.js');

class FlexLight {
  constructor (canvas, options) {
    this._options = options || {};
    this._api = 'hip';
    this._canvas = canvas;
    this._camera = new Camera();
    this._config = new Config();
    this._scene = new Scene(this._options);
    this._idRenderer = 'pathtracer';
    this._renderer = new PathTracerHIP(canvas, this._scene, this._camera, this._config, this._options);
    this._io = undefined;
  }

  get canvas () { return this._canvas; }
  get api () { return this._api; }
  get camera () { return this._camera; }
  get config () { return this._config; }
  get scene () { return this._scene; }
  get renderer () { return this._renderer; }
  get io () { return this._io; }

  set canvas (canvas) {
    if (canvas === this._canvas) return;
    this._canvas = canvas;
    this.renderer = this._idRenderer;
  }

  set api (api) {
    if (api === this._api) return;
    if (api !== 'hip') { console.error('Only the hip api is available in this build, not', api); return; }
    this._api = api;
  }

  set config (config) { this._config = config; this._renderer.config = config; }
  set camera (camera) { this._camera = camera; this._renderer.camera = camera; }
  set scene (scene) { this._scene = scene; this._renderer.scene = scene; }

  set renderer (name) {
    if (name !== 'pathtracer') {
      console.error('Renderer option', name, 'on api', this._api, 'doesn\'t exist.');
      return;
    }
    this._idRenderer = name;
    const wasRunning = this._renderer && !this._renderer._halt;
    if (this._renderer) this._renderer.halt();
    this._renderer = new PathTracerHIP(this._canvas, this._scene, this._camera, this._config, this._options);
    if (wasRunning) this._renderer.render();
  }

  set io (name) { this._io = name; }
}

module.exports = { FlexLight, PathTracerHIP, Camera, Config, Scene, Transform, Primitive, Triangle, Plane, Object3D, Cuboid, Bounding };
