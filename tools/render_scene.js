'use strict';
/*
 * Render one BASELINE scene through the whole JavaScript path — FlexLight facade, scene graph, host
 * flattening, N-API addon, libflexlight_hip.so — and write the float32 radiance to a file.
 *   node tools/render_scene.js <scene> --out frame.f32 [--width W --height H --spp S --bounces B --filter 0|1 --aa fxaa|taa --frames N --batch N --present FILE --assets DIR --devices N | a,b,...]
 *   --devices: the frame split over several GPUs in this process (flx_group_*): N = GPUs 0 .. N - 1, or a list (a number may repeat)
 */
const fs = require('fs');
const os = require('os');
const path = require('path');
const childProcess = require('child_process');
const ROOT = path.resolve(__dirname, '..');
const { FlexLight, Transform } = require(path.join(ROOT, 'web-ray-tracer_amd', 'js', 'flexlight.js'));
const scenes = require(path.join(ROOT, 'web-ray-tracer_amd', 'js', 'scenes', 'index.js'));

const args = process.argv.slice(2);
const name = args[0];
const opt = (flag, d) => { const i = args.indexOf(flag); return i >= 0 ? args[i + 1] : d; };
const assets = opt('--assets', process.env.FLX_REFERENCE || '/root/reference');

function loadImage (rel) {
  const tmp = path.join(os.tmpdir(), 'flx-img-' + process.pid + '.rgba');
  const py = 'import sys; from PIL import Image; im = Image.open(sys.argv[1]).convert("RGBA"); open(sys.argv[2], "wb").write(im.tobytes()); print(im.width, im.height)';
  const dims = childProcess.execFileSync('python3', ['-c', py, path.join(assets, rel), tmp]).toString().trim().split(' ').map(Number);
  const data = new Uint8Array(fs.readFileSync(tmp));
  fs.unlinkSync(tmp);
  return { width: dims[0], height: dims[1], data };
}

(async () => {
  const frame = scenes[name].frame;
  const canvas = { width: Number(opt('--width', frame.width)), height: Number(opt('--height', frame.height)) };
  Transform.reset();
  const log = console.log; console.log = () => {};
  const dv = opt('--devices', null);
  const devices = dv === null ? undefined : (dv.includes(',') ? dv.split(',').map(Number) : Number(dv));
  const engine = new FlexLight(canvas, { assetRoot: assets, devices });
  engine.loadImage = async rel => loadImage(rel);
  await scenes[name](engine);
  console.log = log;
  engine.config.samplesPerRay = Number(opt('--spp', frame.samplesPerRay));
  engine.config.maxReflections = Number(opt('--bounces', frame.maxReflections));
  engine.config.filter = Number(opt('--filter', frame.filter ? 1 : 0)) === 1;
  engine.config.antialiasing = opt('--aa', undefined);               // 'fxaa' | 'taa'
  engine.renderer = 'pathtracer';
  await engine.renderer.updateScene();
  const batch = Number(opt('--batch', 0));                            // N frames of a camera move through renderBatch: all N are written
  if (batch > 0) {
    const cam = engine.camera;
    const cameras = Array.from({ length: batch }, (_, i) => ({ x: cam.x + 0.3 * i, y: cam.y + 0.1 * i, fx: cam.fx + 0.05 * i, fy: cam.fy - 0.02 * i }));
    const b = engine.renderer.renderBatch(cameras, { counters: true });
    let same = true;                                                  // every frame of the batch against the renderFrame() of its camera
    cameras.forEach((c, i) => {
      Object.assign(cam, c);
      const single = engine.renderer.renderFrame().radiance;
      same = same && single.length === b.frames[i].length && single.every((v, k) => Object.is(v, b.frames[i][k]));
    });
    fs.writeFileSync(opt('--out', 'frame.f32'), Buffer.concat(b.frames.map(f => Buffer.from(f.buffer, f.byteOffset, f.byteLength))));
    console.log(JSON.stringify({ width: b.width, height: b.height, rows: b.rows, frames: batch, frameMs: b.frameMs, counters: b.counters, batchEqualsFrames: same }));
    engine.renderer.halt();
    return;
  }
  const frames = Number(opt('--frames', 1));                          // the last of `frames` frames is written (TAA keeps history)
  let f;
  for (let k = 0; k < frames; k++) f = engine.renderer.renderFrame({ counters: true });
  fs.writeFileSync(opt('--out', 'frame.f32'), Buffer.from(f.radiance.buffer));
  if (opt('--present', null)) fs.writeFileSync(opt('--present', null), Buffer.from(engine.renderer.presentFrame(f).data.buffer));      // the canvas' RGBA8
  console.log(JSON.stringify({ width: f.width, height: f.height, rows: f.rows, frameMs: f.frameMs, counters: f.counters, gpus: engine.renderer.gpuInfo }));
  engine.renderer.halt();
})().catch(e => { console.error(e); process.exit(1); });
