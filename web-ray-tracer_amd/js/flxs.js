'use strict';
/*
 * .flxs — flat scene container: the boundary data of SURVEY.md §8a rows D1–D7 in one file, so a
 * scene built by the JavaScript host can be replayed by the C / Python drivers on a box that has
 * no Node (and no OBJ assets).  Layout (little endian):
 *   "FLXS1\n"  u32 jsonBytes  json  pad-to-16  array0 pad-to-16 array1 ...
 * json = { meta: {...}, arrays: [{name, dtype: 'f32'|'i32'|'u8', count, offset, shape?}] } with
 * offsets relative to the start of the data section.  A '.gz' suffix means the whole file is gzip'd.
 */
const fs = require('fs');
const zlib = require('zlib');

const MAGIC = Buffer.from('FLXS1\n', 'latin1');
const pad16 = n => (n + 15) & ~15;

function write (file, meta, arrays) {
  const entries = [];
  const chunks = [];
  let offset = 0;
  Object.keys(arrays).forEach(name => {
    const a = arrays[name];
    let dtype;
    if (a instanceof Float32Array) dtype = 'f32';
    else if (a instanceof Int32Array) dtype = 'i32';
    else if (a instanceof Uint8Array || a instanceof Uint8ClampedArray) dtype = 'u8';
    else throw new Error('flxs: unsupported array type for ' + name);
    const bytes = Buffer.from(a.buffer, a.byteOffset, a.byteLength);
    entries.push({ name, dtype, count: a.length, offset });
    chunks.push(bytes);
    const padded = pad16(bytes.length);
    if (padded !== bytes.length) chunks.push(Buffer.alloc(padded - bytes.length));
    offset += padded;
  });
  const json = Buffer.from(JSON.stringify({ meta, arrays: entries }), 'utf8');
  const head = Buffer.alloc(pad16(MAGIC.length + 4 + json.length));
  MAGIC.copy(head, 0);
  head.writeUInt32LE(json.length, MAGIC.length);
  json.copy(head, MAGIC.length + 4);
  let blob = Buffer.concat([head].concat(chunks));
  if (file.endsWith('.gz')) blob = zlib.gzipSync(blob, { level: 9 });
  fs.writeFileSync(file, blob);
}

function read (file) {
  let blob = fs.readFileSync(file);
  if (file.endsWith('.gz')) blob = zlib.gunzipSync(blob);
  if (blob.slice(0, MAGIC.length).compare(MAGIC) !== 0) throw new Error('flxs: bad magic in ' + file);
  const jsonLen = blob.readUInt32LE(MAGIC.length);
  const desc = JSON.parse(blob.slice(MAGIC.length + 4, MAGIC.length + 4 + jsonLen).toString('utf8'));
  const base = pad16(MAGIC.length + 4 + jsonLen);
  const arrays = {};
  desc.arrays.forEach(e => {
    const width = e.dtype === 'u8' ? 1 : 4;
    const copy = new Uint8Array(e.count * width);
    blob.copy(Buffer.from(copy.buffer), 0, base + e.offset, base + e.offset + e.count * width);
    arrays[e.name] = e.dtype === 'f32' ? new Float32Array(copy.buffer)
      : e.dtype === 'i32' ? new Int32Array(copy.buffer) : copy;
  });
  return { meta: desc.meta, arrays };
}

module.exports = { write, read };
