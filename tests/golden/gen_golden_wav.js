#!/usr/bin/env node
// Golden-vector generator for the file-intake functions (test infrastructure; run once in the build
// container).  Evaluates the reference's own JavaScript, read by path from /root/reference at
// generation time, on synthetic RIFF/WAVE files built here, and writes the files plus the reference's
// answers into tests/golden/wav/.  Only data is written: no reference source text is stored.
//
//   node tests/golden/gen_golden_wav.js [/root/reference]
//
// Covered reference entry points:
//   modules/fileLoader.js   getWavSampleRate, getWavDuration   (the text of these two functions only:
//                           the rest of the file needs a DOM and syntax newer than this Node)
//   modules/guanoReader.js  extractGuanoMetadata, parseGuanoMetadata
//   modules/cropAudio.js    cropWavBlob
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const { TextDecoder } = require('util');

const REF = process.argv[2] || '/root/reference';
const OUT = path.join(__dirname, 'wav');
fs.mkdirSync(OUT, { recursive: true });

function Blob(parts, opts) { this.parts = parts; this.type = (opts || {}).type; }
function load(file, keep, exportNames) {
  let src = fs.readFileSync(path.join(REF, 'modules', file), 'utf8');
  if (keep) src = keep(src);
  src = src.replace(/^import .*$/mg, '').replace(/export\s+(async\s+)?function/g, '$1function');
  src += `\n;globalThis.__x = { ${exportNames.join(', ')} };\n`;
  const sandbox = { Math, Array, Object, String, Number, DataView, Uint8Array, TextDecoder, Blob, isNaN, parseFloat, console };
  sandbox.globalThis = sandbox;
  vm.runInContext(src, vm.createContext(sandbox), { filename: file });
  return sandbox.__x;
}
// fileLoader.js: only the two leading wav functions (up to the first non-function top-level statement)
const FL = load('fileLoader.js', (src) => src.slice(0, src.indexOf('let lastObjectUrl')), ['getWavSampleRate', 'getWavDuration']);
const GU = load('guanoReader.js', null, ['extractGuanoMetadata', 'parseGuanoMetadata']);
const CR = load('cropAudio.js', null, ['cropWavBlob']);

// ---------------------------------------------------------------- synthetic files
function lcg(seed) { let s = seed >>> 0; return () => (s = (Math.imul(1103515245, s) + 12345) & 0x7fffffff); }
function chunk(id, payload) {
  const pad = payload.length & 1;
  const b = Buffer.alloc(8 + payload.length + pad);
  b.write(id, 0, 4, 'latin1');
  b.writeUInt32LE(payload.length, 4);
  payload.copy(b, 8);
  return b;
}
function fmt(tag, ch, sr, bits, extra) {
  const b = Buffer.alloc(16 + (extra || 0));
  b.writeUInt16LE(tag, 0); b.writeUInt16LE(ch, 2); b.writeUInt32LE(sr, 4);
  b.writeUInt32LE(sr * ch * bits / 8, 8); b.writeUInt16LE(ch * bits / 8, 12); b.writeUInt16LE(bits, 14);
  return b;
}
function pcm(frames, ch, bits, seed, float) {
  const r = lcg(seed), b = Buffer.alloc(frames * ch * bits / 8);
  for (let i = 0; i < frames * ch; i++) {
    const v = (r() / 0x40000000) - 1;   // [-1, 1)
    if (float) b.writeFloatLE(Math.fround(v * 0.5), i * 4);
    else if (bits === 16) b.writeInt16LE(Math.round(v * 32767), i * 2);
    else if (bits === 24) b.writeIntLE(Math.round(v * 8388607), i * 3, 3);
    else if (bits === 32) b.writeInt32LE(Math.round(v * 2147483647), i * 4);
    else if (bits === 8) b.writeUInt8(Math.round(v * 127) + 128, i);
  }
  return b;
}
function riff(chunks) {
  const body = Buffer.concat(chunks);
  const h = Buffer.alloc(12);
  h.write('RIFF', 0, 4, 'latin1'); h.writeUInt32LE(4 + body.length, 4); h.write('WAVE', 8, 4, 'latin1');
  return Buffer.concat([h, body]);
}
const guanoA = 'GUANO|Version: 1.0\nMake: Synthetic\nTimestamp: 2023-05-01 20:15:30+08:00\nLoc Position: 22.3964 114.1095\nNote: a: b\n';   // odd length
const guanoB = 'GUANO|Version:  1.0\r\nTimestamp: 2024-11-30T05:06:07Z\r\nLoc Position: 22.5 -114.25\r\nSamplerate: 384000\r\n';
const guanoC = 'GUANO|Version: 1.0\nLoc Position: 47.1   -122.33\nTimestamp: 2022-01-02 03:04\nTimestamp: 2021-12-31 23:59:59+00:00';
const guanoD = 'GUANO|Version: 1.0\nLoc Position: 10.5 abc\nKeyOnly\n: novalue\n';
const files = {
  'mono16_guano_after': riff([chunk('fmt ', fmt(1, 1, 384000, 16)), chunk('data', pcm(4001, 1, 16, 1)), chunk('guan', Buffer.from(guanoA, 'utf8'))]),
  'mono16_list_before_odd': riff([chunk('LIST', Buffer.from('INFOabc', 'latin1')), chunk('fmt ', fmt(1, 1, 500000, 16, 2)), chunk('guan', Buffer.from(guanoB, 'utf8')),
                                  chunk('data', pcm(3000, 1, 16, 2))]),
  'stereo24': riff([chunk('fmt ', fmt(1, 2, 256000, 24)), chunk('data', pcm(1501, 2, 24, 3)), chunk('guan', Buffer.from(guanoC, 'utf8'))]),
  'float32_mono': riff([chunk('fmt ', fmt(3, 1, 192000, 32)), chunk('fact', Buffer.alloc(4)), chunk('data', pcm(2000, 1, 32, 4, true))]),
  'u8_mono_no_guano': riff([chunk('fmt ', fmt(1, 1, 44100, 8)), chunk('data', pcm(999, 1, 8, 5))]),
  'int32_mono': riff([chunk('fmt ', fmt(1, 1, 96000, 32)), chunk('data', pcm(700, 1, 32, 6))]),
  'no_fmt': riff([chunk('junk', Buffer.alloc(10)), chunk('data', pcm(100, 1, 16, 7))]),
  'data_before_fmt': riff([chunk('data', pcm(500, 1, 16, 8)), chunk('fmt ', fmt(1, 1, 384000, 16))]),
  'guano_fields_odd': riff([chunk('fmt ', fmt(1, 1, 384000, 16)), chunk('guan', Buffer.from(guanoD, 'utf8')), chunk('data', pcm(64, 1, 16, 9))]),
};

function fileOf(buf) {
  const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
  return { arrayBuffer: async () => ab, type: '' };
}

(async () => {
  const manifest = { generator: 'tests/golden/gen_golden_wav.js', node: process.version, files: [], guano_texts: [] };
  for (const [name, buf] of Object.entries(files)) {
    fs.writeFileSync(path.join(OUT, name + '.wav'), buf);
    const f = fileOf(buf);
    const e = { name, file: 'wav/' + name + '.wav', bytes: buf.length };
    e.sample_rate = await FL.getWavSampleRate(f);
    e.duration = await FL.getWavDuration(f);
    try { e.guano_text = await GU.extractGuanoMetadata(f); } catch (err) { e.guano_error = String(err && err.name); }
    if (e.guano_text !== undefined) e.guano_fields = GU.parseGuanoMetadata(e.guano_text);
    e.crops = [];
    const d = e.duration || 0.01;
    for (const [t0, t1] of [[0, d], [d * 0.25, d * 0.5], [0.001, 0.0035], [d * 0.9, d * 2], [d * 0.5, d * 0.5], [d, d * 0.2]]) {
      let blob = null;
      try { blob = await CR.cropWavBlob(f, t0, t1); } catch (err) { blob = { error: String(err && err.name) }; }
      const c = { start: t0, end: t1 };
      if (blob === null) c.result = null;
      else if (blob.error) c.error = blob.error;
      else {
        const out = Buffer.from(blob.parts[0]);
        c.result = `wav/${name}.crop${e.crops.length}.wav`;
        fs.writeFileSync(path.join(__dirname, c.result), out);
        c.bytes = out.length;
      }
      e.crops.push(c);
    }
    manifest.files.push(e);
  }
  // parseGuanoMetadata alone, on texts that exercise its branches
  const texts = [guanoA, guanoB, guanoC, guanoD, '', '(No GUANO metadata found in file)', 'Timestamp: 2020-02-03 04:05:06.789+01:00\nLoc Position: -33.9 113.0',
                 'Loc Position: 22.1 -113\nTimestamp: 2020-1-2 3:4', 'Loc Position: 1.5 -115.0', 'Loc Position: 1.5 -115.0001', 'Loc Position:  7 1e2 extra',
                 'Loc Position: 5', 'Timestamp:  2019-09-09  10:11:12', 'Loc Position: 22.30 114.100', 'Loc Position: 0 -0.000001234', 'Loc Position: 0 123456789012345678901234'];
  for (const t of texts) manifest.guano_texts.push({ text: t, fields: GU.parseGuanoMetadata(t) });
  fs.writeFileSync(path.join(OUT, 'manifest.json'), JSON.stringify(manifest, null, 1));
  let bytes = 0;
  for (const f of fs.readdirSync(OUT)) bytes += fs.statSync(path.join(OUT, f)).size;
  console.log(`wrote ${manifest.files.length} files, ${manifest.guano_texts.length} texts, ${(bytes / 1024).toFixed(0)} KiB into ${OUT}`);
})().catch((e) => { console.error(e); process.exit(1); });
