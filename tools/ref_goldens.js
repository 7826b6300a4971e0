'use strict';
/*
 * Golden host arrays from the REFERENCE's own modules/scene.js (SURVEY.md §8c, Appendix C).
 *
 *   node tools/ref_goldens.js            # all scenes -> tests/golden/ref_*.flxs[.gz] + ref_dragon.json
 *   node tools/ref_goldens.js <scene>    # one scene (child-process mode; Transform numbering is static state)
 *
 * Runs only in the authoring container (needs /root/reference).  Nothing of the reference is copied
 * into this repository: math.js / arrays.js / scene.js are read where they lie, written to a fresh
 * temp directory with two mechanical accommodations for Node 12 — (i) the seven `a ?? b`
 * expressions (scene.js:258,406-411) become `(a != null ? a : b)`, (ii) import specifiers get the
 * .mjs suffix — and imported from there; the temp directory is deleted afterwards.  `window`,
 * `fetch`, `document` and `Image` are shimmed (math.js:3, scene.js:23-38,427,480).
 *
 * What is stored: the arrays the reference's generateArraysFromGraph() / Transform.buildWGL2Arrays()
 * return for the four BASELINE scenes, i.e. the inputs of the hot path exactly as the reference's
 * host would upload them.  The dragon scene (11.8 MB) is stored in full once, gzip'd, because the
 * GPU box has neither the reference nor its OBJ assets and bench.py needs that scene.
 */
const fs = require('fs');
const os = require('os');
const path = require('path');
const crypto = require('crypto');
const childProcess = require('child_process');

const REF = process.env.FLX_REFERENCE || '/root/reference';
const ROOT = path.resolve(__dirname, '..');
const OUT = path.join(ROOT, 'tests', 'golden');
const scenes = require(path.join(ROOT, 'web-ray-tracer_amd', 'js', 'scenes', 'index.js'));
const sceneFile = require(path.join(ROOT, 'web-ray-tracer_amd', 'js', 'sceneFile.js'));

const sha256 = typed => crypto.createHash('sha256').update(Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength)).digest('hex');

function stageReferenceModules () {
  const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'flx-ref-'));
  ['math.js', 'arrays.js', 'scene.js'].forEach(f => {
    let src = fs.readFileSync(path.join(REF, 'modules', f), 'utf8');
    src = src.replace(/([A-Za-z_][A-Za-z_.]*) \?\? ([^;]+);/g, '($1 != null ? $1 : $2);');
    src = src.replace(/from '\.\/(\w+)\.js'/g, "from './$1.mjs'");
    fs.writeFileSync(path.join(dir, f.replace(/\.js$/, '.mjs')), src);
  });
  return dir;
}

function installShims () {
  global.window = { Math };
  const EXTRA = process.env.FLX_EXTRA_ASSETS || path.join(ROOT, 'build', 'assets');      // generated assets (the synthetic 100k dragon) first
  global.fetch = async p => ({ text: async () => fs.readFileSync(fs.existsSync(path.join(EXTRA, p)) ? path.join(EXTRA, p) : path.join(REF, p), 'utf8') });
  // Just enough canvas/Image for Scene.textureFromRGB (scene.js:22-39): the "image" keeps the RGBA bytes.
  global.document = {
    createElement: () => {
      const canvas = { width: 0, height: 0, pixels: null };
      canvas.getContext = () => ({
        imageSmoothingEnabled: true,
        createImageData: (w, h) => ({ width: w, height: h, data: new Uint8ClampedArray(w * h * 4) }),
        putImageData: img => { canvas.pixels = img; }
      });
      canvas.toDataURL = () => ({ width: canvas.pixels.width, height: canvas.pixels.height, data: new Uint8Array(canvas.pixels.data) });
      return canvas;
    }
  };
  global.Image = class {
    set src (v) { if (v && v.data) { this.width = v.width; this.height = v.height; this.data = v.data; } }
  };
}

// Decode an image asset to RGBA with PIL (no JPEG decoder in Node 12).
function loadImage (rel) {
  const tmp = path.join(os.tmpdir(), 'flx-img-' + process.pid + '.rgba');
  const py = 'import sys; from PIL import Image; im = Image.open(sys.argv[1]).convert("RGBA"); ' +
    'open(sys.argv[2], "wb").write(im.tobytes()); print(im.width, im.height)';
  const dims = childProcess.execFileSync('python3', ['-c', py, path.join(REF, rel), tmp]).toString().trim().split(' ').map(Number);
  const data = new Uint8Array(fs.readFileSync(tmp));
  fs.unlinkSync(tmp);
  return { width: dims[0], height: dims[1], data };
}

async function runScene (name) {
  const dir = stageReferenceModules();
  try {
    installShims();
    const log = console.log;
    console.log = () => {};            // the reference chats while parsing
    const mod = await import(path.join(dir, 'scene.mjs'));
    const engine = {
      scene: new mod.Scene(),
      camera: { x: 0, y: 0, z: 0, fx: 0, fy: 0, fov: 1 / Math.PI },      // modules/camera.js:5-10
      loadImage: async rel => loadImage(rel)
    };
    const t0 = Date.now();
    await scenes[name](engine);
    const built = engine.scene.generateArraysFromGraph();
    const transforms = mod.Transform.buildWGL2Arrays();
    console.log = log;
    const frame = scenes[name].frame;
    const file = path.join(OUT, 'ref_' + name + '.flxs.gz');
    const extra = scenes[name].extraAssets ? { synthetic: 'objects/dragon_100k.obj is generated (tools/make_dragon_100k.py: dragon_lp.obj, every triangle split 1 -> 4); the arrays are what the reference\'s scene.js emits for it' } : {};
    const s = sceneFile.save(file, name, engine, built, transforms, frame, Object.assign({ producer: 'reference modules/scene.js via tools/ref_goldens.js' }, extra));
    const summary = {
      name,
      textureLength: built.textureLength,
      bufferLength: built.bufferLength,
      entriesPadded: s.meta.entriesPadded,
      transforms: s.meta.transforms,
      aabbNodes: 0, triangles: 0,
      sha256: {},
      head: Array.from(built.geometryBuffer.slice(0, 12 * 8)),
      buildSeconds: (Date.now() - t0) / 1000
    };
    for (let i = 0; i < built.textureLength; i++) {
      const type = built.geometryBuffer[i * 12 + 10];
      if (type === 1) summary.aabbNodes++; else if (type === 2) summary.triangles++;
    }
    ['geometry', 'attributes', 'ids', 'rotation', 'shift', 'lights'].forEach(k => { summary.sha256[k] = sha256(s.arrays[k]); });
    fs.writeFileSync(path.join(OUT, 'ref_' + name + '.json'), JSON.stringify(summary, null, 1) + '\n');
    log(name + ': entries ' + built.textureLength + ' (' + summary.aabbNodes + ' AABB + ' + summary.triangles + ' tris), transforms ' +
      s.meta.transforms + ', ' + summary.buildSeconds + ' s -> ' + path.relative(ROOT, file));
  } finally {
    fs.readdirSync(dir).forEach(f => fs.unlinkSync(path.join(dir, f)));
    fs.rmdirSync(dir);
  }
}

if (process.argv[2]) {
  runScene(process.argv[2]).catch(e => { console.error(e); process.exit(1); });
} else {
  if (!fs.existsSync(REF)) { console.error('reference not mounted at ' + REF); process.exit(1); }
  fs.mkdirSync(OUT, { recursive: true });
  Object.keys(scenes).forEach(name => {
    childProcess.execFileSync(process.execPath, ['--max-old-space-size=6000', __filename, name], { stdio: 'inherit' });
  });
}
