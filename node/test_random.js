'use strict';
// Random plugin option sets through the Node host: reads <dir>/cases.json (written by tests/test_node_addon.py, which
// runs the same cases through the Python host), prints one JSON line with a SHA-1 per case of getFrequencies' bytes,
// of the colour-mapped image with its crop window, and of the canvas image.  Both hosts sit on the same C ABI, so the
// digests must agree exactly: what is compared is the marshalling of every option on the JavaScript side.
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const { Spectrogram } = require('./spectrogram_plugin.js');

const dir = process.argv[2];
const cases = JSON.parse(fs.readFileSync(path.join(dir, 'cases.json'), 'utf8'));
const sha = (u8) => crypto.createHash('sha1').update(Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength)).digest('hex');
const out = [];
for (const c of cases) {
  const raw = fs.readFileSync(path.join(dir, c.pcm));
  const x = new Float32Array(raw.buffer.slice(raw.byteOffset, raw.byteOffset + raw.byteLength));
  const buffer = { sampleRate: c.sampleRate, length: x.length, numberOfChannels: 1, getChannelData: () => x };
  const opts = Object.assign({}, c.options, { canvasWidth: c.width });
  const p = Spectrogram.create(opts);
  try {
    const frames = p.getFrequencies(buffer)[0];
    const img = p.drawSpectrogram(frames, c.width);
    const canvas = p.drawCanvas(frames, c.width);
    out.push({ id: c.id, frames: frames.length, freq: sha(frames.flat), image: sha(img.data), crop: [img.crop.sy, img.crop.sh],
               canvas: sha(canvas.data), canvasHeight: canvas.height });
  } finally {
    p.destroy();
  }
}
console.log(JSON.stringify({ results: out }));
