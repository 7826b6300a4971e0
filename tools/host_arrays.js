'use strict';
/*
 * Build one of the BASELINE scenes with THIS repo's JavaScript host layer (web-ray-tracer_amd/js) and
 * print the sha256 of every array it would hand to the GPU, as JSON — tests/test_js_host.py compares
 * them with tests/golden/ref_<scene>.json, the hashes of what the reference's own scene.js emits.
 *   node tools/host_arrays.js <scene> [--assets DIR] [--save FILE.flxs.gz] [--native]   (--native: OBJ import / BVH / flatten through flx_mesh_*)
 * Asset files (OBJ/MTL/JPEG) are read from --assets (default: $FLX_REFERENCE or /root/reference).
 */
const fs = require('fs');
const os = require('os');
const path = require('path');
const crypto = require('crypto');
const childProcess = require('child_process');

const ROOT = path.resolve(__dirname, '..');
const JS = path.join(ROOT, 'web-ray-tracer_amd', 'js');
const { Scene, Transform } = require(path.join(JS, 'scene.js'));
const { Camera } = require(path.join(JS, 'camera.js'));
const scenes = require(path.join(JS, 'scenes', 'index.js'));
const sceneFile = require(path.join(JS, 'sceneFile.js'));

const args = process.argv.slice(2);
const name = args[0];
const opt = flag => { const i = args.indexOf(flag); return i >= 0 ? args[i + 1] : undefined; };
const assets = opt('--assets') || process.env.FLX_REFERENCE || '/root/reference';
const sha256 = a => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex');

function loadImage (rel) {           // decode with PIL: Node 12 has no JPEG decoder
  const tmp = path.join(os.tmpdir(), 'flx-img-' + process.pid + '.rgba');
  const py = 'import sys; from PIL import Image; im = Image.open(sys.argv[1]).convert("RGBA"); open(sys.argv[2], "wb").write(im.tobytes()); print(im.width, im.height)';
  const dims = childProcess.execFileSync('python3', ['-c', py, path.join(assets, rel), tmp]).toString().trim().split(' ').map(Number);
  const data = new Uint8Array(fs.readFileSync(tmp));
  fs.unlinkSync(tmp);
  return { width: dims[0], height: dims[1], data };
}

(async () => {
  if (!scenes[name]) { console.error('unknown scene ' + name + '; one of ' + Object.keys(scenes).join(', ')); process.exit(2); }
  Transform.reset();
  const log = console.log; console.log = () => {}; const warn = console.warn; console.warn = () => {};
  const engine = { scene: new Scene({ assetRoot: assets }), camera: new Camera(), loadImage: async rel => loadImage(rel), nativeImport: args.includes('--native') };
  const t0 = Date.now();
  await scenes[name](engine);
  const built = engine.scene.generateArraysFromGraph();
  const buildMs = Date.now() - t0;
  const transforms = engine.nativeImport ? Transform.buildWGL2ArraysNative(Scene.loadAddon()) : Transform.buildWGL2Arrays();
  console.log = log; console.warn = warn;
  const s = sceneFile.assemble(name, engine, built, transforms, scenes[name].frame, { producer: 'web-ray-tracer_amd/js host via tools/host_arrays.js' });
  const out = { name, textureLength: built.textureLength, bufferLength: built.bufferLength, entriesPadded: s.meta.entriesPadded, transforms: s.meta.transforms, buildMs, native: engine.nativeImport, sha256: {} };
  ['geometry', 'attributes', 'ids', 'rotation', 'shift', 'lights', 'atlasAlbedo', 'atlasPbr', 'atlasTpo', 'viewMatrix'].forEach(k => { out.sha256[k] = sha256(s.arrays[k]); });
  const save = opt('--save');
  if (save) require(path.join(JS, 'flxs.js')).write(save, s.meta, s.arrays);
  console.log(JSON.stringify(out));
})().catch(e => { console.error(e); process.exit(1); });
