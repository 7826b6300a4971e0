'use strict';
/*
 * The JavaScript frame loop on a replayed scene: FlexLight facade -> PathTracerHIP.render() -> N-API -> libflexlight_hip.so, on a
 * box without the OBJ assets (the GPU box): the flattened scene comes from a .flxs fixture, the Transform objects are re-created
 * here so that the animation of examples/dragon.js:97-110 (the monkey turns to face the camera, every tick) can run.
 *   node tools/js_loop.js tests/golden/ref_dragon.flxs.gz [--frames N] [--move 1] [--present8 1] [--width W --height H --spp S --bounces B]
 *                         [--devices 0,1,..  (a group of GPUs in this process; a number may repeat: rehearsal on one GPU)  --lanes 2|3]
 *                         [--dump PREFIX --dump-frames K]    (the first K frames as PREFIX<k>.f32 + the camera / transforms used, for the parity test)
 *                         [--blocking 1]   (flx_frame_end on the main thread, as before round 5, instead of frameEndAsync on a worker thread)
 * Prints one JSON line: frames, seconds, fps (wall clock over the loop), the renderer's own fps counter, median GPU ms per frame, and how late a 1 ms
 * timer of the application fires while the loop runs (eventLoopLagMs: the main thread is in the event loop while a worker waits for the GPU).
 */
const fs = require('fs');
const path = require('path');
const ROOT = path.resolve(__dirname, '..');
const { FlexLight, Transform } = require(path.join(ROOT, 'web-ray-tracer_amd', 'js', 'flexlight.js'));
const sceneFile = require(path.join(ROOT, 'web-ray-tracer_amd', 'js', 'sceneFile.js'));

const args = process.argv.slice(2);
const opt = (flag, d) => { const i = args.indexOf(flag); return i >= 0 ? args[i + 1] : d; };
const replay = sceneFile.sceneFromFlxs(path.resolve(args[0]));
const meta = replay.meta;
const canvas = { width: Number(opt('--width', meta.frame.width)), height: Number(opt('--height', meta.frame.height)) };
const frames = Number(opt('--frames', 120));
const moveMode = Number(opt('--move', 0));
const move = moveMode !== 0;                              // 1: the camera moves and the monkey turns to face it; 2: only the camera moves (the scene's arrays stay as they are)
const dumpPrefix = opt('--dump', null), dumpFrames = Number(opt('--dump-frames', 3));

Transform.reset();
const devices = opt('--devices', null);
const engine = new FlexLight(canvas, devices ? { devices: devices.split(',').map(Number) } : {});
engine.scene = replay;
Object.assign(engine.camera, meta.camera);
engine.config.samplesPerRay = Number(opt('--spp', meta.frame.samplesPerRay));
engine.config.maxReflections = Number(opt('--bounces', meta.frame.maxReflections));
engine.config.filter = !!meta.frame.filter;
let faceCamera = () => {};
if (meta.name === 'dragon') {                              // the two transforms of examples/dragon.js:27-29, 52-54
  const dragonTransform = new Transform(); dragonTransform.move(15, 0, 15); dragonTransform.scale(0.5);
  const monkeTransform = new Transform(); monkeTransform.move(5, 1, 12); monkeTransform.scale(2);
  faceCamera = () => {                                      // examples/dragon.js:105-109
    const cam = engine.camera, p = monkeTransform.position;
    const d = [cam.x - p[0], cam.y - p[1], cam.z - p[2]];
    const r = Math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const theta = Math.sign(d[2]) * Math.acos(d[0] / Math.sqrt(d[0] * d[0] + d[2] * d[2])) - Math.PI * 0.5;
    const psi = Math.acos(d[1] / r) - Math.PI * 0.5;
    monkeTransform.rotateSpherical(theta, psi);
  };
  faceCamera();
} else {
  for (let t = 1; t < meta.transforms; t++) new Transform();   // (identity stand-ins: replayed scenes other than the dragon have one transform)
}
engine.renderer = 'pathtracer';
engine.renderer.scene = replay;
engine.renderer.present8 = Number(opt('--present8', 0)) === 1;
if (devices) engine.renderer.groupLanes = Number(opt('--lanes', 3));
engine.renderer.blockingFrameEnd = Number(opt('--blocking', 0)) === 1;
/* the application's own timer: how late does it fire?  (setInterval(…, 1) asks for 1 ms; what comes on top is the time the main thread was not in the event loop) */
const lag = [];
let lastTick = 0;
const lagTimer = setInterval(() => {
  const now = Number(process.hrtime.bigint()) / 1e6;
  if (lastTick && got > 0) lag.push(Math.max(0, now - lastTick - 1));
  lastTick = now;
}, 1);

const gpuMs = [];
const log = [];
let begun = 0, got = 0, t0 = 0;
const tick = () => {                                        // the application's per-frame update: a camera move + the monkey following it
  if (move) {
    engine.camera.x += 0.05; engine.camera.y += 0.02; engine.camera.z -= 0.03;
    engine.camera.fx += 0.004; engine.camera.fy -= 0.002;
    if (moveMode === 1) faceCamera();
  }
  if (dumpPrefix && begun < dumpFrames) {
    const tr = Transform.buildWGL2Arrays();
    const fp = engine.renderer.frameParams();
    log.push({ camera: fp.camera, viewMatrix: fp.viewMatrix, rotation: Array.from(tr[0]), shift: Array.from(tr[1]) });
  }
  begun++;
};
/* the renderer reads camera / transforms when it begins a frame: run the tick right before each begin */
const origUpload = engine.renderer._uploadFrameState.bind(engine.renderer);
engine.renderer._uploadFrameState = () => { tick(); return origUpload(); };
canvas.onFrame = f => {
  if (got === 0) t0 = Date.now();
  if (dumpPrefix && got < dumpFrames) fs.writeFileSync(dumpPrefix + got + '.f32', Buffer.from(f.pixels.buffer, f.pixels.byteOffset, f.pixels.byteLength));
  gpuMs.push(f.frameMs);
  got++;
  if (got === frames + 1) {                                  // frame 0 starts the clock
    const seconds = (Date.now() - t0) / 1000;
    const rendererFps = engine.renderer.fps;
    engine.renderer.halt();
    clearInterval(lagTimer);
    lag.sort((a, b) => a - b);
    const lagStats = lag.length ? { samples: lag.length, mean: lag.reduce((a, b) => a + b, 0) / lag.length, median: lag[lag.length >> 1], p99: lag[Math.floor(lag.length * 0.99)], max: lag[lag.length - 1] } : null;
    gpuMs.sort((a, b) => a - b);
    if (dumpPrefix) fs.writeFileSync(dumpPrefix + 'log.json', JSON.stringify(log));
    console.log(JSON.stringify({ scene: meta.name, width: canvas.width, height: canvas.height, spp: engine.config.samplesPerRay, bounces: engine.config.maxReflections,
      frames, seconds, fps: frames / seconds, rendererFps: Number(rendererFps), gpuMsMedian: gpuMs[gpuMs.length >> 1], present8: engine.renderer.present8, moving: move, devices: devices ? devices.split(',').map(Number) : null, lanes: devices ? engine.renderer.groupLanes : 2,
      frameEnd: engine.renderer.blockingFrameEnd ? 'blocking' : 'async (worker thread)', eventLoopLagMs: lagStats }));
  }
};
engine.renderer.render().catch(e => { console.error(e); process.exit(1); });
