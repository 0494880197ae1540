'use strict';
// File intake with the reference's function names and results (modules/fileLoader.js:7-63,
// modules/guanoReader.js:3-80, modules/cropAudio.js:1-59), on Buffers / Uint8Arrays instead of
// browser File objects.  The chunk walks run in libsonospec (sono_wav_*).
const { native } = require('./index.js');

function bytesOf(file) {
  if (file == null) return null;
  if (file instanceof Uint8Array) return file;
  if (file instanceof ArrayBuffer) return new Uint8Array(file);
  throw new TypeError('expected a Buffer, Uint8Array or ArrayBuffer');
}

function getWavSampleRate(file) {
  const b = bytesOf(file);
  return b ? native.wavParse(b).sampleRate : 256000;
}
function getWavDuration(file) {
  const b = bytesOf(file);
  return b ? native.wavParse(b).duration : 0;
}
function extractGuanoMetadata(file) {
  const b = bytesOf(file);
  if (!b) return '(no file selected)';
  const [off, len] = native.wavGuano(b);
  if (!len) return '(No GUANO metadata found in file)';
  return Buffer.from(b.buffer, b.byteOffset + off, len).toString('utf8');
}
function parseGuanoMetadata(text) {
  return native.guanoFields(new Uint8Array(Buffer.from(text || '', 'utf8')));
}
// -> Uint8Array of the cropped file, or null (where the reference returns null)
function cropWavBlob(file, startTime, endTime) {
  const b = bytesOf(file);
  return b ? native.wavCrop(b, startTime, endTime) : null;
}
// fileLoader.js:179-181: 'ok' | 'small' | 'long'
function acceptFile(fileBytes, duration, timeExpansion) {
  return ['ok', 'small', 'long'][native.wavAccept(fileBytes, duration, !!timeExpansion)];
}
// the data chunk's bytes (a view, no copy) and its layout
function pcmSamples(file) {
  const b = bytesOf(file), w = native.wavParse(b);
  if (!w.hasFmt || !w.hasData) throw new Error('fmt or data chunk missing');
  return { raw: b.subarray(w.dataOffset, w.dataOffset + w.frames * w.blockAlign), info: w };
}

// One channel of the data chunk as Float32Array, on the host: v / 2^(bits-1) for integer PCM (8-bit is offset binary),
// float32 as stored -- the convention of the device conversion (sono_pcm_to_f32).
function channelData(file, channel = 0) {
  const { raw, info: w } = pcmSamples(file);
  const bits = w.bitsPerSample, step = w.blockAlign, out = new Float32Array(w.frames);
  const dv = new DataView(raw.buffer, raw.byteOffset, raw.byteLength), o0 = channel * (bits >> 3);
  if (channel < 0 || channel >= w.channels) throw new RangeError('no such channel');
  if (w.formatTag === 3 && bits === 32) for (let i = 0; i < w.frames; i++) out[i] = dv.getFloat32(i * step + o0, true);
  else if (w.formatTag === 1 && bits === 16) for (let i = 0; i < w.frames; i++) out[i] = dv.getInt16(i * step + o0, true) / 32768;
  else if (w.formatTag === 1 && bits === 8) for (let i = 0; i < w.frames; i++) out[i] = (dv.getUint8(i * step + o0) - 128) / 128;
  else if (w.formatTag === 1 && bits === 24)
    for (let i = 0; i < w.frames; i++) {
      const p = i * step + o0;
      out[i] = ((dv.getUint8(p) | (dv.getUint8(p + 1) << 8) | (dv.getInt8(p + 2) << 16))) / 8388608;
    }
  else if (w.formatTag === 1 && bits === 32) for (let i = 0; i < w.frames; i++) out[i] = dv.getInt32(i * step + o0, true) / 2147483648;
  else throw new Error('unsupported PCM format (tag ' + w.formatTag + ', ' + bits + ' bits)');
  return out;
}

module.exports = { getWavSampleRate, getWavDuration, extractGuanoMetadata, parseGuanoMetadata, cropWavBlob, acceptFile, pcmSamples, channelData };
