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

module.exports = { getWavSampleRate, getWavDuration, extractGuanoMetadata, parseGuanoMetadata, cropWavBlob, acceptFile, pcmSamples };
