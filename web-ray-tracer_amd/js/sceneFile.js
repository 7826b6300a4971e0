'use strict';
/*
 * Assemble the complete render input of one frame (SURVEY.md §8a rows D1–D7) from a FlexLight
 * scene graph and store it as a .flxs file.  Used by tools/make_scenes.js (this repo's host
 * layer) and by tools/ref_goldens.js (arrays emitted by the reference's own scene.js).
 */
const flxs = require('./flxs.js');

// Light array as the reference uploads it (modules/pathtracerWGL2.js:143-165): per light
// [x, y, z, intensity, variation, 0]; a missing intensity/variation property takes the scene default.
function buildLightArray (scene) {
  const lights = scene.primaryLightSources;
  const out = new Float32Array(6 * lights.length);
  lights.forEach((l, i) => {
    const intensity = l.intensity === undefined ? scene.defaultLightIntensity : l.intensity;
    const variation = l.variation === undefined ? scene.defaultLightVariation : l.variation;
    out.set([l[0], l[1], l[2], intensity, variation, 0], i * 6);
  });
  return out;
}

// Atlas as modules/pathtracerWGL2.js:85-104 draws it: every source image nearest-resampled to
// standardTextureSizes and placed at column i % tw, row floor(i / tw); the canvas is
// width*tw by height*list.length (taller than needed; unused cells stay transparent black).
// `images` are {width, height, data: Uint8Array RGBA}.  Nearest resample rule (the browser's is
// unspecified; this is the definition here): destination pixel centre -> floor of source coordinate.
function buildAtlas (images, sizes) {
  if (images.length === 0) return { width: 1, height: 1, data: new Uint8Array(4) };
  const w = sizes[0], h = sizes[1];
  const tw = Math.floor(2048 / w);
  const W = w * tw, H = h * images.length;
  const data = new Uint8Array(W * H * 4);
  images.forEach((img, i) => {
    const ox = w * (i % tw), oy = h * Math.floor(i / tw);
    for (let y = 0; y < h; y++) {
      const sy = Math.min(img.height - 1, Math.floor((y + 0.5) * img.height / h));
      for (let x = 0; x < w; x++) {
        const sx = Math.min(img.width - 1, Math.floor((x + 0.5) * img.width / w));
        const s = (sy * img.width + sx) * 4, d = ((oy + y) * W + ox + x) * 4;
        data[d] = img.data[s]; data[d + 1] = img.data[s + 1]; data[d + 2] = img.data[s + 2]; data[d + 3] = img.data[s + 3];
      }
    }
  });
  return { width: W, height: H, data };
}

// viewMatrix of modules/pathtracerWGL2.js:312-318 (row-major, uploaded with transpose = true).
function buildViewMatrix (camera, width, height) {
  const invFov = 1 / camera.fov;
  const k = height * invFov / width;
  const fx = camera.fx, fy = camera.fy;
  return new Float32Array([
    Math.cos(fx) * k, 0, Math.sin(fx) * k,
    -Math.sin(fx) * Math.sin(fy) * invFov, Math.cos(fy) * invFov, Math.cos(fx) * Math.sin(fy) * invFov,
    -Math.sin(fx) * Math.cos(fy), -Math.sin(fy), Math.cos(fx) * Math.cos(fy)
  ]);
}

function assemble (name, engine, built, transformArrays, frame, extraMeta) {
  const scene = engine.scene, camera = engine.camera;
  const sizes = scene.standardTextureSizes;
  const albedo = buildAtlas(scene.textures, sizes);
  const pbr = buildAtlas(scene.pbrTextures, sizes);
  const tpo = buildAtlas(scene.translucencyTextures, sizes);
  const meta = Object.assign({
    name,
    textureLength: built.textureLength,
    bufferLength: built.bufferLength,
    entriesPadded: built.geometryBuffer.length / 12,
    transforms: transformArrays[1].length / 8,
    lights: scene.primaryLightSources.length,
    camera: { x: camera.x, y: camera.y, z: camera.z, fx: camera.fx, fy: camera.fy, fov: camera.fov },
    ambient: Array.from(scene.ambientLight),
    textureWidth: Math.floor(2048 / sizes[0]),
    atlas: {
      albedo: [albedo.width, albedo.height], pbr: [pbr.width, pbr.height], tpo: [tpo.width, tpo.height]
    },
    frame
  }, extraMeta || {});
  const arrays = {
    geometry: built.geometryBuffer,
    attributes: built.sceneBuffer,
    ids: built.idBuffer,
    rotation: transformArrays[0],
    shift: transformArrays[1],
    lights: buildLightArray(scene),
    viewMatrix: buildViewMatrix(camera, frame.width, frame.height),
    atlasAlbedo: albedo.data,
    atlasPbr: pbr.data,
    atlasTpo: tpo.data
  };
  return { meta, arrays };
}

function save (file, name, engine, built, transformArrays, frame, extraMeta) {
  const s = assemble(name, engine, built, transformArrays, frame, extraMeta);
  flxs.write(file, s.meta, s.arrays);
  return s;
}

/* A scene that replays a .flxs file: what the renderer asks of a Scene (generateArraysFromGraph, light list, ambient light, texture
 * lists, standardTextureSizes) answered from the stored arrays — for boxes without the OBJ / JPEG assets (the GPU box).  The
 * atlases come prebuilt (`prebuiltAtlases`); transforms are NOT replayed: the caller re-creates its Transform objects, so that
 * an animation can move them (tools/js_loop.js). */
function sceneFromFlxs (file) {
  const { meta, arrays } = flxs.read(file);
  const lights = [];
  for (let i = 0; i < arrays.lights.length; i += 6) {
    const l = [arrays.lights[i], arrays.lights[i + 1], arrays.lights[i + 2]];
    l.intensity = arrays.lights[i + 3]; l.variation = arrays.lights[i + 4];
    lights.push(l);
  }
  const atlas = k => ({ width: meta.atlas[k][0], height: meta.atlas[k][1], data: arrays['atlas' + k[0].toUpperCase() + k.slice(1)] });
  return {
    meta,
    primaryLightSources: lights, defaultLightIntensity: 200, defaultLightVariation: 0.4,
    ambientLight: meta.ambient.slice(),
    textures: [], pbrTextures: [], translucencyTextures: [],
    standardTextureSizes: [Math.floor(2048 / meta.textureWidth), Math.floor(2048 / meta.textureWidth)],
    prebuiltAtlases: [atlas('albedo'), atlas('pbr'), atlas('tpo')],
    queue: [],
    generateArraysFromGraph: () => ({ textureLength: meta.textureLength, bufferLength: meta.bufferLength, geometryBuffer: arrays.geometry, sceneBuffer: arrays.attributes, idBuffer: arrays.ids })
  };
}

module.exports = { buildLightArray, buildAtlas, buildViewMatrix, assemble, save, sceneFromFlxs };
