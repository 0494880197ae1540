'use strict';
// From per-frame spectral features to the `data` object SonoRadar's species matcher reads.
// The reference's autoIdHK(data) (modules/autoid_HK.js:215-264) matches kHz / ms fields that the
// browser takes from user-placed markers (modules/autoIdPanel.js:1388-1468); a batch host measures
// them: SONO_OUT_FEATURES gives one spectral peak per frame, and the library's host functions group
// frames into calls (sono_segment_calls), find where a call turns (sono_call_shape), convert to the
// panel's units (sono_autoid_data_of_call) and run the matcher (sono_autoid_match).  This file only
// names things the JavaScript way and flattens a rule table -- passed in as data, it stays with the
// reference -- into the records the matcher takes.
const N = require('./index.js').native;

const FIELDS = ['highestFreq', 'lowestFreq', 'kneeFreq', 'heelFreq', 'startFreq', 'endFreq', 'cfStart', 'cfEnd', 'duration',
                'bandwidth', 'kneeLowTime', 'kneeLowBandwidth', 'heelLowBandwidth', 'kneeHeelBandwidth'];   // autoid_HK.js:216-221
const CF_CALL_TYPES = ['FM-CF-FM', 'CF-FM'];                                                                 // autoIdPanel.js:1417
const CALL_KEYS = ['startFrame', 'endFrame', 'peakFrame', 'startBin', 'endBin', 'highBin', 'lowBin', 'peakBin', 'peakIdx'];
const SHAPE_KEYS = ['highFrame', 'lowFrame', 'kneeFrame', 'heelFrame', 'cfStartFrame', 'cfEndFrame', 'kneeBin', 'heelBin',
                    'cfStartBin', 'cfEndBin'];
const REL_CODE = { '=': 0, '>': 1, '<': 2, '>=': 3, '=>': 3, '<=': 4 };

// {peakBin:Int32Array, peakIdx:Uint8Array} -> the raw 8-byte records of SONO_OUT_FEATURES (what stft() returns)
function toRecords(features) {
  if (features instanceof Uint8Array) return features;
  const n = features.peakIdx.length, rec = new Uint8Array(8 * n), dv = new DataView(rec.buffer);
  for (let i = 0; i < n; i++) {
    dv.setInt32(8 * i, features.peakBin[i], true);
    rec[8 * i + 4] = features.peakIdx[i];
  }
  return rec;
}
function fromRecords(rec) {
  const n = rec.length >> 3, dv = new DataView(rec.buffer, rec.byteOffset, rec.length);
  const peakBin = new Int32Array(n), peakIdx = new Uint8Array(n);
  for (let i = 0; i < n; i++) {
    peakBin[i] = dv.getInt32(8 * i, true);
    peakIdx[i] = rec[8 * i + 4];
  }
  return { peakBin, peakIdx };
}
const unpack = (keys, nums, at = 0) => Object.fromEntries(keys.map((k, i) => [k, nums[at + i]]));
const pack = (keys, obj) => Float64Array.from(keys, (k) => obj[k]);

// maximal runs of frames with peakIdx >= threshold, at least minFrames long
function segmentCalls(features, threshold, minFrames = 1, firstFrame = 0) {
  const nums = N.segmentCalls(toRecords(features), threshold, minFrames, firstFrame), calls = [];
  for (let at = 0; at < nums.length; at += CALL_KEYS.length) calls.push(unpack(CALL_KEYS, nums, at));
  return calls;
}

// knee, heel, constant-frequency part, first highest / lowest frame of a call (frames absolute, -1 = none)
function callShape(features, call, firstFrame = 0) {
  return unpack(SHAPE_KEYS, N.callShape(toRecords(features), firstFrame, pack(CALL_KEYS, call)));
}

function callToAutoIdData(call, fftN, hop, sampleRate, { callType, harmonic = 0, timeExpansion = 1, shape } = {}) {
  const f = N.autoidDataOfCall(pack(CALL_KEYS, call), shape ? pack(SHAPE_KEYS, shape) : null, fftN, hop, sampleRate, timeExpansion,
                               CF_CALL_TYPES.includes(callType) ? 1 : 0);
  const d = {};
  FIELDS.forEach((name, i) => { if (!Number.isNaN(f[i])) d[name] = f[i]; });
  d.peakFreq = call.peakBin * sampleRate / fftN / 1000 * timeExpansion;
  d.startTime = call.startFrame * hop / sampleRate * 1000 / timeExpansion;
  d.harmonic = harmonic;
  if (callType !== undefined) d.callType = callType;
  return d;
}

// A species rule table (the shape of autoid_HK.js:3-198) flattened into sono_autoid_rule / sono_autoid_cond records.
class RuleTable {
  constructor(speciesRules) {
    this.names = speciesRules.map((s) => s.name);
    this.callTypes = [];
    const rules = [], conds = [], intervals = [], harmonics = [];
    const num = (v) => (v === null ? 0 : (typeof v === 'number' || typeof v === 'boolean') ? Number(v) : NaN);  // as in `val >= v`
    speciesRules.forEach((sp, si) => sp.rules.forEach((rule) => {
      let lo32 = 0, hi32 = 0;
      if (rule.callType) {
        for (const t of rule.callType.split(',').map((s) => s.trim())) {
          if (!this.callTypes.includes(t)) this.callTypes.push(t);
          const bit = this.callTypes.indexOf(t);
          if (bit < 32) lo32 = (lo32 | (1 << bit)) >>> 0; else hi32 = (hi32 | (1 << (bit - 32))) >>> 0;
        }
      }
      const hFirst = harmonics.length, hCount = rule.harmonic ? rule.harmonic.length : 0;
      if (rule.harmonic) harmonics.push(...rule.harmonic.map((h) => h | 0));
      const cFirst = conds.length / 4;
      if (rule.harmonic && rule.harmonic.length === 0) conds.push(0, 2, 0, 0);  // `[].includes(x)` is false: the rule never matches
      FIELDS.forEach((name, fi) => {
        const r = rule[name];
        if (!r) return;
        if (typeof r[0] === 'string' && r.length === 1) {
          const m = r[0].match(/^(=|=>|>=|<|<=|>)\s*(\w+)$/);
          if (m && FIELDS.includes(m[2])) conds.push(fi, 1, REL_CODE[m[1]], FIELDS.indexOf(m[2]));
          else conds.push(fi, 2, 0, 0);
          return;
        }
        const first = intervals.length / 2;
        for (const iv of (Array.isArray(r[0]) ? r : [r])) intervals.push(iv.length > 0 ? num(iv[0]) : NaN, iv.length > 1 ? num(iv[1]) : NaN);
        conds.push(fi, 0, first, intervals.length / 2 - first);
      });
      rules.push([lo32, hi32, hFirst, hCount, cFirst, conds.length / 4 - cFirst, si, 0]);
    }));
    if (this.callTypes.length > 64) throw Error('more than 64 distinct call types');
    this._rules = new Uint8Array(32 * rules.length);
    const dv = new DataView(this._rules.buffer);
    rules.forEach((r, i) => r.forEach((v, j) => dv.setUint32(32 * i + 4 * j, v, true)));
    this._conds = Uint32Array.from(conds);
    this._intervals = Float64Array.from(intervals);
    this._harmonics = Int32Array.from(harmonics);
  }

  // autoIdHK(data): names joined with ' / ', or 'No species matched' (autoid_HK.js:261-263)
  match(data = {}) {
    const fields = Float64Array.from(FIELDS, (f) => (typeof data[f] === 'number' ? data[f] : NaN));
    const ct = this.callTypes.indexOf(data.callType);
    const hit = N.autoidMatch(this._rules, this._conds, this._intervals, this._harmonics, fields, ct,
                              typeof data.harmonic === 'number' ? data.harmonic : null, this.names.length);
    const names = this.names.filter((_, i) => hit[i]);
    return names.length ? names.join(' / ') : 'No species matched';
  }
}

const tables = new WeakMap();
function autoIdHK(data, speciesRules) {
  let t = tables.get(speciesRules);
  if (!t) tables.set(speciesRules, t = new RuleTable(speciesRules));
  return t.match(data);
}

module.exports = { segmentCalls, callShape, callToAutoIdData, autoIdHK, RuleTable, toRecords, fromRecords, FIELDS };
