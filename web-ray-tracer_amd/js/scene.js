'use strict';
/*
 * Scene graph + host flattening for the FlexLight HIP back-end (Node, CommonJS, Node-12 syntax).
 *
 * This is the JavaScript side of the drop-in: the same public surface as the reference's
 * modules/scene.js (Scene, Transform, Primitive/Triangle/Plane, Object3D/Bounding/Cuboid; SURVEY.md
 * Appendix A) written from scratch, with one hard requirement — the arrays it hands to the renderer
 * (generateArraysFromGraph, Transform.buildWGL2Arrays) must equal the reference's bit for bit, because
 * they ARE the input of the GPU hot path (SURVEY.md §8a D1–D4).  tests/test_js_host.py checks that
 * against tests/golden/ref_*.json (hashes of the arrays the reference's own scene.js emits).
 * Where the numerical recipe matters the reference line is cited; the structure is this repo's own:
 * primitives keep their state in one typed record and rebuild their two flat buffers on demand,
 * node types share one walker, file access goes through a pluggable `readText` (fs by default).
 */
const fs = require('fs');
const path = require('path');
const la = require('./linalg.js');

const LEAF_MAX = 4;                 // scene.js:6  BVH_MAX_LEAVES_PER_NODE
const NODE_BIAS = 0.00152587890625; // scene.js:159,907  (100 * 2^-16)
const ENTRY = 12, ATTR = 28, ROW = 256;

const isGroup = item => Array.isArray(item) || item.indexable === true;

/* ---- Transform ------------------------------------------------------------------------------------ */
class Transform {
  constructor () {
    this._rotation = la.identity(3);
    this._position = [0, 0, 0];
    this._scale = 1;
    let n = 0;
    while (Transform.used[n]) n++;                       // lowest free number, never released (scene.js:600-605)
    Transform.used[n] = true;
    this.number = n;
    Transform.count = Math.max(Transform.count, n + 1);
    Transform.transformList[n] = this;
  }

  get matrix () { return la.scaleMat(this._rotation, this._scale); }   // scene.js:545-549 (scale * rotation, entries not snapped)
  get position () { return this._position; }
  move (x, y, z) { this._position = [x, y, z]; }
  scale (s) { this._scale = s; }

  rotateAxis (n, theta) {                                 // scene.js:559-569 (Rodrigues)
    const s = Math.sin(theta), c = Math.cos(theta), k = 1 - c;
    this._rotation = [
      [n[0] * n[0] * k + c, n[0] * n[1] * k - n[2] * s, n[0] * n[2] * k + n[1] * s],
      [n[0] * n[1] * k + n[2] * s, n[1] * n[1] * k + c, n[1] * n[2] * k - n[0] * s],
      [n[0] * n[2] * k - n[1] * s, n[1] * n[2] * k + n[0] * s, n[2] * n[2] * k + c]
    ];
  }

  rotateSpherical (theta, psi) {                          // scene.js:571-584
    const sT = Math.sin(theta), cT = Math.cos(theta), sP = Math.sin(psi), cP = Math.cos(psi);
    this._rotation = [[cT, 0, sT], [-sT * sP, cP, cT * sP], [-sT * cP, -sP, cT * cP]];
  }

  /* Per transform: forward matrix rows, inverse rows (3 x vec4 each, std140) and position / -position
   * (vec4 each).  JS rows land in GLSL columns (scene.js:510-517). */
  static buildWGL2Arrays () {
    const rotation = new Float32Array(24 * Transform.count);
    const shift = new Float32Array(8 * Transform.count);
    for (let t = 0; t < Transform.count; t++) {
      const tr = Transform.transformList[t];
      const m = tr.matrix;
      const inv = la.pseudoInverse(m);
      for (let r = 0; r < 3; r++) {
        rotation.set(m[r], t * 24 + 4 * r);
        rotation.set(inv[r], t * 24 + 12 + 4 * r);
      }
      shift.set(tr.position, t * 8);
      shift.set(la.scaleVec(tr.position, -1), t * 8 + 4);
    }
    return [rotation, shift];
  }

  /* buildWGL2Arrays in native code (flx_transforms_pack through the N-API addon; SURVEY 8f N3): the same arrays */
  static buildWGL2ArraysNative (addon) {
    const T = Transform.count;
    const matrices = new Float64Array(9 * T), positions = new Float64Array(3 * T);
    for (let t = 0; t < T; t++) {
      const tr = Transform.transformList[t];
      const m = tr.matrix;
      for (let r = 0; r < 3; r++) matrices.set(m[r], t * 9 + r * 3);
      positions.set(tr.position, t * 3);
    }
    const rotation = new Float32Array(24 * T), shift = new Float32Array(8 * T);
    addon.packTransforms(matrices, positions, rotation, shift);
    return [rotation, shift];
  }

  /* Forget every transform but the identity at number 0 (a fresh page load in the browser). */
  static reset () {
    Transform.used = [];
    Transform.count = 0;
    Transform.transformList = [];
    new Transform();                                      // eslint-disable-line no-new
  }
}
Transform.used = [];
Transform.count = 0;
Transform.transformList = [];
new Transform();                                          // number 0 = identity (scene.js:590-593)

/* ---- Primitives: 1 (Triangle) or 2 (Plane) triangles, each a 12-float geometry + 28-float attribute entry -- */
class Primitive {
  constructor (length, vertices, normal, uvs) {
    this.indexable = false;
    this.length = length;
    this._vertices = new Float32Array(vertices);
    this._normal = new Float32Array(normal);
    this._normals = new Float32Array(length * 9);
    for (let i = 0; i < length * 3; i++) this._normals.set(normal, i * 3);
    this._uvs = new Float32Array(uvs);
    this._transform = undefined;
    this._textureNums = new Float32Array([-1, -1, -1]);
    this._albedo = new Float32Array([1, 1, 1]);
    this._rme = new Float32Array([1, 0, 0]);
    this._tpo = new Float32Array([0, 0, 1]);
    this.geometryBuffer = new Float32Array(length * ENTRY);
    this.sceneBuffer = new Float32Array(length * ATTR);
    this._flatten();
  }

  /* entry layouts: SURVEY.md §8a D1 / D2 (scene.js:628-643) */
  _flatten () {
    for (let t = 0; t < this.length; t++) {
      const g = t * ENTRY, a = t * ATTR;
      this.geometryBuffer.set(this._vertices.subarray(t * 9, t * 9 + 9), g);
      this.geometryBuffer[g + 9] = this.transformNum;
      this.geometryBuffer[g + 10] = 2;
      this.sceneBuffer.set(this._normals.subarray(t * 9, t * 9 + 9), a);
      this.sceneBuffer.set(this._uvs.subarray(t * 6, t * 6 + 6), a + 9);
      this.sceneBuffer.set(this._textureNums, a + 15);
      this.sceneBuffer.set(this._albedo, a + 18);
      this.sceneBuffer.set(this._rme, a + 21);
      this.sceneBuffer.set(this._tpo, a + 24);
    }
  }

  get transformNum () { return this._transform === undefined ? 0 : this._transform.number; }
  get transform () { return this._transform; }
  set transform (t) { this._transform = t; this._flatten(); }

  /* The getters hand out the live typed arrays: importObj writes uvs / normals into them in place and
   * relies on a LATER setter to re-flatten (scene.js:381-400) — kept, it decides what reaches the GPU. */
  get vertices () { return this._vertices; }
  set vertices (v) { this._vertices = new Float32Array(v); this._flatten(); }
  get normals () { return this._normals; }
  set normals (ns) { this._normals = new Float32Array(ns); this._normal = new Float32Array(Array.prototype.slice.call(ns, 0, 3)); this._flatten(); }
  get normal () { return this._normal; }
  set normal (n) {
    this._normals = new Float32Array(this.length * 9);
    for (let i = 0; i < this.length * 3; i++) this._normals.set(n, i * 3);
    this._normal = new Float32Array(n);
    this._flatten();
  }
  get uvs () { return this._uvs; }
  set uvs (uv) { this._uvs = new Float32Array(uv); this._flatten(); }
  get textureNums () { return this._textureNums; }
  set textureNums (tn) { this._textureNums = tn; this._flatten(); }
  get color () { return this._albedo; }
  set color (c) { this._albedo = new Float32Array(Array.prototype.map.call(c, v => v / 255)); this._flatten(); }   // 0..255 in, /255 stored
  get albedo () { return this._albedo; }
  set albedo (a) { this.color = a; }
  get roughness () { return this._rme[0]; }
  set roughness (r) { this._rme[0] = r; this._flatten(); }
  get metallicity () { return this._rme[1]; }
  set metallicity (m) { this._rme[1] = m; this._flatten(); }
  get emissiveness () { return this._rme[2]; }
  set emissiveness (e) { this._rme[2] = e; this._flatten(); }
  get translucency () { return this._tpo[0]; }
  set translucency (t) { this._tpo[0] = t; this._flatten(); }
  get ior () { return this._tpo[2]; }
  set ior (o) { this._tpo[2] = o; this._flatten(); }
}

const flat2 = rows => rows.reduce((p, r) => p.concat(r), []);

class Plane extends Primitive {                            // scene.js:747-751
  constructor (c0, c1, c2, c3) {
    super(2, flat2([c0, c1, c2, c2, c3, c0]), la.unit(la.cross(la.subVec(c0, c2), la.subVec(c0, c1))), [0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0]);
  }
}

class Triangle extends Primitive {                         // scene.js:753-757
  constructor (a, b, c) {
    super(1, flat2([a, b, c]), la.unit(la.cross(la.subVec(a, c), la.subVec(a, b))), [0, 0, 0, 1, 1, 1]);
  }
}

/* ---- Groups: array-like nodes whose setters fan out to their children ------------------------------- */
class Object3D {
  constructor (length) {
    this.relativePosition = [0, 0, 0];
    this.length = length;
    this.indexable = true;
    this._transform = undefined;
  }

  get transformNum () { return this._transform ? this._transform.number : 0; }
  get transform () { return this._transform; }
  set transform (t) { this._transform = t; this._each(c => { c.transform = t; }); }
  _each (fn) { for (let i = 0; i < this.length; i++) fn(this[i]); }

  move (x, y, z) {                                         // scene.js:811-829: adds to the vertices
    this.relativePosition = [x, y, z];
    const d = [x, y, z];
    this._each(c => {
      if (c.indexable) c.move(x, y, z);
      else c.vertices = c.vertices.map((v, i) => v + d[i % 3]);
    });
  }

  scale (s) {                                              // scene.js:831-839: about relativePosition
    const o = this.relativePosition;
    this._each(c => {
      if (c.indexable) c.scale(s);
      else c.vertices = c.vertices.map((v, i) => (v - o[i % 3]) * s + o[i % 3]);
    });
  }
}
['textureNums', 'color', 'albedo', 'roughness', 'metallicity', 'emissiveness', 'translucency', 'ior'].forEach(name => {
  Object.defineProperty(Object3D.prototype, name, { set (v) { this._each(c => { c[name] = v; }); }, configurable: true });
});

class Bounding extends Object3D {
  constructor (items) {
    super(items.length);
    items.forEach((item, i) => { this[i] = item; });
  }
}

class Cuboid extends Object3D {                            // scene.js:903-921; index order top right front bottom left back
  constructor (x, x2, y, y2, z, z2) {
    super(6);
    x += NODE_BIAS; y += NODE_BIAS; z += NODE_BIAS;
    x2 -= NODE_BIAS; y2 -= NODE_BIAS; z2 -= NODE_BIAS;
    this.bounding = [x, x2, y, y2, z, z2];
    this.top = new Plane([x, y2, z], [x2, y2, z], [x2, y2, z2], [x, y2, z2]);
    this.right = new Plane([x2, y2, z], [x2, y, z], [x2, y, z2], [x2, y2, z2]);
    this.front = new Plane([x2, y2, z2], [x2, y, z2], [x, y, z2], [x, y2, z2]);
    this.bottom = new Plane([x, y, z2], [x2, y, z2], [x2, y, z], [x, y, z]);
    this.left = new Plane([x, y2, z2], [x, y, z2], [x, y, z], [x, y2, z]);
    this.back = new Plane([x, y2, z], [x, y, z], [x2, y, z], [x2, y2, z]);
    [this.top, this.right, this.front, this.bottom, this.left, this.back].forEach((p, i) => { this[i] = p; });
  }
}

/* ---- an imported object that lives in native code (SURVEY 8f N2: flx_mesh_* of libflexlight_hip.so through the N-API addon) ----
 * Same surface as the Bounding tree importObj returns — transform / material setters, move, scale — and the same arrays at
 * flattening time, bit for bit; parsing, BVH build and flattening of a 45 k-triangle OBJ take tens of milliseconds instead
 * of seconds. */
const MESH_FIELD = { color: 0, albedo: 0, roughness: 1, metallicity: 2, emissiveness: 3, translucency: 4, ior: 5, textureNums: 6 };
class NativeMesh {
  constructor (addon, handle) {
    this.nativeMesh = true;
    this.indexable = false;
    this._addon = addon;
    this._handle = handle;
    this._transform = undefined;
    this.relativePosition = [0, 0, 0];
    const c = addon.meshCounts(handle);
    this.entries = c.entries;
    this.triangles = c.triangles;
  }

  get transformNum () { return this._transform ? this._transform.number : 0; }
  get transform () { return this._transform; }
  set transform (t) { this._transform = t; this._addon.meshSetTransform(this._handle, t ? t.number : 0); }
  move (x, y, z) { this.relativePosition = [x, y, z]; this._addon.meshMove(this._handle, x, y, z); }
  scale (s) { this._addon.meshScale(this._handle, s); }
  get bounding () { return this._addon.meshBounding(this._handle); }
  set bounding (b) { /* recomputed on demand */ }
  /* the block of generateArraysFromGraph: entries [at, at + entries), ids relative to `at` */
  flattenInto (geometryBuffer, sceneBuffer, idBuffer, at, tri) {
    const g = geometryBuffer.subarray(at * ENTRY, (at + this.entries) * ENTRY);
    const a = sceneBuffer.subarray(at * ATTR, (at + this.entries) * ATTR);
    const ids = idBuffer.subarray(tri, tri + this.triangles);
    const box = this._addon.meshFlatten(this._handle, g, a, ids);
    if (at !== 0) for (let i = 0; i < ids.length; i++) ids[i] += at;
    return box;
  }
}
Object.keys(MESH_FIELD).forEach(name => {
  Object.defineProperty(NativeMesh.prototype, name, {
    set (v) {
      const vals = new Float64Array(3);
      if (typeof v === 'number') vals[0] = v; else for (let i = 0; i < 3; i++) vals[i] = v[i];
      this._addon.meshSetMaterial(this._handle, MESH_FIELD[name], vals);
    },
    configurable: true
  });
});

/* ---- images: what the browser keeps in <img>/<canvas>, here {width, height, data: RGBA bytes} -------- */
function imageFromRGBA (array, width, height) {             // scene.js:22-39 (values clamp + round like Uint8ClampedArray)
  return { width, height, data: new Uint8Array(new Uint8ClampedArray(array).buffer) };
}
function imageFromTriples (array, width, height) {          // scene.js:43-50: x255, alpha 255
  const texels = [];
  for (let i = 0; i < array.length; i += 3) texels.push(array[i] * 255, array[i + 1] * 255, array[i + 2] * 255, 255);
  return imageFromRGBA(texels, width, height);
}

/* ---- Scene ------------------------------------------------------------------------------------------ */
class Scene {
  constructor (options) {
    this.primaryLightSources = [[0, 10, 0]];
    this.defaultLightIntensity = 200;
    this.defaultLightVariation = 0.4;
    this.ambientLight = [0.025, 0.025, 0.025];
    this.textures = [];
    this.pbrTextures = [];
    this.translucencyTextures = [];
    this.standardTextureSizes = [1024, 1024];
    this.queue = [];
    const opts = options || {};
    this.assetRoot = opts.assetRoot || process.cwd();
    /* generated assets (build/assets: the synthetic 100k dragon) are looked up before the asset root */
    this.extraAssetRoot = opts.extraAssetRoot || process.env.FLX_EXTRA_ASSETS || path.resolve(__dirname, '..', '..', 'build', 'assets');
    this.readText = opts.readText || (p => {
      const extra = path.resolve(this.extraAssetRoot, p);
      return fs.readFileSync(fs.existsSync(extra) ? extra : path.resolve(this.assetRoot, p), 'utf8');
    });
    // constructors as the API exposes them (scene.js:319-327)
    this.Transform = () => new Transform();
    this.Cuboid = (x, x2, y, y2, z, z2) => new Cuboid(x, x2, y, y2, z, z2);
    this.Plane = (c0, c1, c2, c3) => new Plane(c0, c1, c2, c3);
    this.Triangle = (a, b, c) => new Triangle(a, b, c);
    this.Bounding = array => new Bounding(array);
  }

  async textureFromRGB (array, width, height) { return imageFromRGBA(array, width, height); }
  async textureFromRME (array, width, height) { return imageFromTriples(array, width, height); }
  async textureFromTPO (array, width, height) { return imageFromTriples(array, width, height); }

  fitsInBound (bound, obj) {                               // scene.js:56-59, boundings are [xmin,xmax,ymin,ymax,zmin,zmax]
    const b = obj.bounding;
    return bound[0] <= b[0] && bound[2] <= b[2] && bound[4] <= b[4] && bound[1] >= b[1] && bound[3] >= b[3] && bound[5] >= b[5];
  }

  /* scene.js:157-187: exact vertex bounds on primitives, children's bounds widened by NODE_BIAS on groups */
  updateBoundings (obj) {
    if (obj === undefined) obj = this.queue;
    let box = new Array(6);
    if (obj.nativeMesh) return obj.bounding;
    if (isGroup(obj)) {
      if (obj.length === 0 && !obj.blockError) {
        console.error('problematic object structure', 'isArray:', Array.isArray(obj), 'indexable:', obj.indexable, 'object:', obj);
        obj.blockError = true;
      } else {
        box = this.updateBoundings(obj[0]);
        for (let i = 1; i < obj.length; i++) {
          const b = this.updateBoundings(obj[i]);
          box = box.map((v, k) => (k % 2 === 0 ? Math.min(v, b[k] - NODE_BIAS) : Math.max(v, b[k] + NODE_BIAS)));
        }
      }
    } else {
      const v = obj.vertices;
      box = [v[0], v[0], v[1], v[1], v[2], v[2]];
      for (let i = 3; i < v.length; i++) {
        const a = (i % 3) * 2;
        box[a] = Math.min(box[a], v[i]);
        box[a + 1] = Math.max(box[a + 1], v[i]);
      }
    }
    obj.bounding = box;
    return box;
  }

  /* scene.js:62-154: top-down split at the box centre on the axis with the fewest straddlers, three
   * buckets (below / above / straddling), leaves of <= 4, depth <= log2(n) + 8.  The order of the output
   * is the traversal order on the GPU, so every choice (>= in the axis pick, bucket order) is kept. */
  generateBVH (objects) {
    if (objects === undefined) objects = this.queue;
    const scene = this;
    const minWidth = 1 / 256;
    let top = new Bounding(objects);
    this.updateBoundings(top);
    const maxDepth = Math.log2(top.length) + 8;

    const straddlers = (objs, lo, hi) => {
      let n = 0;
      for (let i = 0; i < objs.length; i++) if (!scene.fitsInBound(lo, objs[i]) && !scene.fitsInBound(hi, objs[i])) n++;
      return n;
    };

    const split = (objs, depth) => {
      if (objs.length <= LEAF_MAX || depth > maxDepth) return objs;
      const bb = objs.bounding;
      const centre = [(bb[0] + bb[1]) / 2, (bb[2] + bb[3]) / 2, (bb[4] + bb[5]) / 2];
      let axis = 0, fewest = Infinity;
      const tried = [];
      for (let a = 0; a < 3; a++) {
        const upper = bb.concat(), lower = bb.concat();
        upper[a * 2] = centre[a];
        lower[a * 2 + 1] = centre[a];
        const room = Math.min(upper[a * 2 + 1] - centre[a], centre[a] - lower[a * 2]);
        const n = straddlers(objs, upper, lower);
        tried.push(n);
        if (fewest >= n && room > minWidth) { axis = a; fewest = n; }
      }
      if (fewest === Infinity) {
        // no axis splits this subtree (every candidate plane leaves no room): it stays a flat list, as in the reference (modules/scene.js:128-132, which also logs it)
        console.warn('flexlight-hip: BVH builder keeps a subtree of ' + objs.length + ' primitives flat (straddlers per axis: ' + tried.join(', ') + ')');
        return objs;
      }
      const bounds = [bb, bb.concat(), bb.concat()];           // bucket 0 shares (and narrows) the parent's array, like the reference
      bounds[0][axis * 2] = centre[axis];
      bounds[1][axis * 2 + 1] = centre[axis];
      const buckets = [[], [], []];
      for (let i = 0; i < objs.length; i++) {
        if (scene.fitsInBound(bounds[0], objs[i])) buckets[0].push(objs[i]);
        else if (scene.fitsInBound(bounds[1], objs[i])) buckets[1].push(objs[i]);
        else buckets[2].push(objs[i]);
      }
      const children = [];
      for (let k = 0; k < 3; k++) {
        if (buckets[k].length === 0) continue;
        const node = new Bounding(buckets[k]);
        scene.updateBoundings(node);
        children.push(split(node, depth + 1));
      }
      const parent = new Bounding(children);
      parent.bounding = objs.bounding;
      return parent;
    };

    top = split(top, 0);
    return top;
  }

  /* scene.js:190-316: depth-first flatten into the skip-list geometry array (12 f32 / entry), the
   * attribute array (28 f32 / entry, same index) and idBuffer (entry index of every triangle). */
  generateArraysFromGraph (root) {
    if (root === undefined) root = this.queue;
    let entries = 0, triangles = 0;
    const measure = item => {
      if (item.nativeMesh) { entries += item.entries; triangles += item.triangles; return; }
      if (isGroup(item)) {
        if (item.length === 0) return;
        entries++;
        for (let i = 0; i < item.length; i++) measure(item[i]);
      } else {
        entries += item.length;
        triangles += item.length;
      }
    };
    measure(root);
    const rowFloatsG = ENTRY * ROW, rowFloatsA = ATTR * ROW;
    const geometryBuffer = new Float32Array(Math.ceil(entries * ENTRY / rowFloatsG) * rowFloatsG);
    const sceneBuffer = new Float32Array(Math.ceil(entries * ATTR / rowFloatsA) * rowFloatsA);
    const idBuffer = new Int32Array(triangles);
    let at = 0, tri = 0;
    const emit = item => {
      if (item.nativeMesh) {
        const box = item.flattenInto(geometryBuffer, sceneBuffer, idBuffer, at, tri);
        at += item.entries; tri += item.triangles;
        return box;
      }
      if (isGroup(item)) {
        if (item.length === 0) return [];
        const self = at++;
        const box = emit(item[0]);                        // [min xyz, max xyz]; tight (no bias here)
        for (let i = 1; i < item.length; i++) {
          const b = emit(item[i]);
          for (let k = 0; k < 3; k++) { box[k] = Math.min(box[k], b[k]); box[k + 3] = Math.max(box[k + 3], b[k + 3]); }
        }
        const g = self * ENTRY;
        for (let k = 0; k < 6; k++) geometryBuffer[g + k] = box[k];
        geometryBuffer[g + 6] = at - self - 1;            // entries to skip on a miss
        const tn = item.transformNum;
        geometryBuffer[g + 9] = (tn !== undefined && tn !== null) ? tn : 0;
        geometryBuffer[g + 10] = 1;
        return box;
      }
      geometryBuffer.set(item.geometryBuffer, at * ENTRY);
      sceneBuffer.set(item.sceneBuffer, at * ATTR);
      for (let i = 0; i < item.length; i++) idBuffer[tri++] = at++;
      const v = item.vertices;
      const box = [v[0], v[1], v[2], v[0], v[1], v[2]];
      for (let i = 3; i < v.length; i += 3) {
        for (let k = 0; k < 3; k++) { box[k] = Math.min(box[k], v[i + k]); box[k + 3] = Math.max(box[k + 3], v[i + k]); }
      }
      return box;
    };
    const minMax = emit(root);
    const height = geometryBuffer.length / rowFloatsG;
    return {
      textureLength: entries, bufferLength: triangles, idBuffer, minMax,
      geometryBufferHeight: height, geometryBuffer,
      sceneBufferHeight: height, sceneBuffer              // the reference derives both heights from the geometry buffer (scene.js:308)
    };
  }

  /* scene.js:330-436.  Quads become Planes (vertex order 3,2,1,0), triangles Triangles (2,1,0); negative
   * indices count from the end; uvs / normals are written into the primitive's live arrays. */
  async importObj (file, materials) {
    if (materials === undefined) materials = [];
    const v = [], vt = [], vn = [];
    let items = [];
    let material;
    const text = await this.readText(file);
    text.split(/\r\n|\r|\n/).forEach(line => {
      const words = line.split(/[\s+]/).filter(w => w.length);     // the reference's class [\\t \\s\\s+] also splits at a literal "+"
      if (words[0] === 'v') v.push([Number(words[1]), Number(words[2]), Number(words[3])]);
      else if (words[0] === 'vt') vt.push([Number(words[1]), Number(words[2])]);
      else if (words[0] === 'vn') vn.push([Number(words[1]), Number(words[2]), Number(words[3])]);
      else if (words[0] === 'usemtl') {
        if (materials[words[1]]) material = words[1];
        else console.warn('Couldn\'t resolve material', material);
      } else if (words[0] === 'f') {
        const corners = words.slice(1).map(w => w.split('/').map(s => {
          let n = Number(s);
          if (n < 0) n = v.length + n + 1;
          return n;
        }));
        let prim, order;
        if (corners.length === 4) {
          prim = new Plane(v[corners[3][0] - 1], v[corners[2][0] - 1], v[corners[1][0] - 1], v[corners[0][0] - 1]);
          order = [3, 2, 1, 1, 0, 3];
        } else {
          prim = new Triangle(v[corners[2][0] - 1], v[corners[1][0] - 1], v[corners[0][0] - 1]);
          order = [2, 1, 0];
        }
        order.forEach((c, i) => {
          const uv = vt[corners[c][1] - 1], nrm = vn[corners[c][2] - 1];
          if (uv !== undefined) prim.uvs.set(uv, i * 2);
          if (nrm !== undefined) prim.normals.set(nrm, i * 3);
        });
        if (material) {
          const m = materials[material];
          const or = (x, d) => ((x !== undefined && x !== null) ? x : d);
          prim.color = or(m.color, [255, 255, 255]);
          prim.emissiveness = or(m.emissiveness, 0);
          prim.metallicity = or(m.metallicity, 0);
          prim.roughness = or(m.roughness, 1);
          prim.translucency = or(m.translucency, 0);
          prim.ior = or(m.ior, 1);
        }
        items.push(prim);
      }
    });
    items = this.generateBVH(items);
    this.updateBoundings(items);
    return items;
  }

  /* importObj in native code: the OBJ (and optionally the MTL) text goes to flx_mesh_import_obj, the result is a NativeMesh */
  async importObjNative (file, mtlFile) {
    const addon = Scene.loadAddon();
    const objText = await this.readText(file);
    const mtlText = mtlFile ? await this.readText(mtlFile) : null;
    return new NativeMesh(addon, addon.meshImport(objText, mtlText));
  }

  static loadAddon () {
    if (!Scene._addon) Scene._addon = require(path.join(__dirname, '..', 'napi', 'flexlight_napi.node'));
    return Scene._addon;
  }

  /* scene.js:438-487: Ka -> colour*255, Ke -> emissiveness = 4*max and colour = 255*Ke/max, Ns/1000 -> metallicity, Ni -> ior */
  async importMtl (file) {
    const materials = [];
    let current;
    const text = await this.readText(file);
    text.split(/\r\n|\r|\n/).forEach(line => {
      const w = line.split(/[\s+]/).filter(s => s.length);
      if (w[0] === 'newmtl') { current = w[1]; materials[current] = {}; }
      else if (w[0] === 'Ka') materials[current].color = la.scaleVec([Number(w[1]), Number(w[2]), Number(w[3])], 255);
      else if (w[0] === 'Ke') {
        const e = Math.max(Number(w[1]), Number(w[2]), Number(w[3]));
        if (e > 0) {
          materials[current].emissiveness = e * 4;
          materials[current].color = la.scaleVec([Number(w[1]), Number(w[2]), Number(w[3])], 255 / e);
        }
      } else if (w[0] === 'Ns') materials[current].metallicity = Number(w[1] / 1000);
      else if (w[0] === 'Ni') materials[current].ior = Number(w[1]);
    });
    return materials;
  }
}

module.exports = { Scene, Transform, Primitive, Triangle, Plane, Object3D, Cuboid, Bounding, NativeMesh };
