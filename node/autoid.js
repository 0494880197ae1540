'use strict';
// From per-frame spectral features to the `data` object SonoRadar's species matcher reads.
// The reference's autoIdHK(data) (modules/autoid_HK.js:215-264) matches kHz / ms fields that the
// browser takes from user-placed markers (modules/autoIdPanel.js:1388-1468); a batch host measures
// them: SONO_OUT_FEATURES gives one spectral peak per frame, segmentCalls groups frames into calls,
// callToAutoIdData converts bins and frames with the panel's own definitions
// (duration = (t_max - t_min) * 1000, bandwidth = high - low, autoIdPanel.js:1406-1421).
// autoIdHK below restates the matcher over a rule table passed in as data.
const FIELDS = ['highestFreq', 'lowestFreq', 'kneeFreq', 'heelFreq', 'startFreq', 'endFreq', 'cfStart', 'cfEnd', 'duration',
                'bandwidth', 'kneeLowTime', 'kneeLowBandwidth', 'heelLowBandwidth', 'kneeHeelBandwidth'];   // autoid_HK.js:216-221
const REL = /^(=|=>|>=|<|<=|>)\s*(\w+)$/;

// features: {peakBin:Int32Array, peakIdx:Uint8Array}; maximal runs with peakIdx >= threshold
function segmentCalls(features, threshold, minFrames = 1, firstFrame = 0) {
  const { peakBin, peakIdx } = features, n = peakIdx.length, calls = [];
  let i = 0;
  while (i < n) {
    if (peakIdx[i] < threshold) { i++; continue; }
    const c = { startFrame: firstFrame + i, startBin: peakBin[i], highBin: peakBin[i], lowBin: peakBin[i],
                peakBin: peakBin[i], peakIdx: peakIdx[i], peakFrame: firstFrame + i };
    let j = i;
    while (j < n && peakIdx[j] >= threshold) {
      if (peakBin[j] > c.highBin) c.highBin = peakBin[j];
      if (peakBin[j] < c.lowBin) c.lowBin = peakBin[j];
      if (peakIdx[j] > c.peakIdx) { c.peakIdx = peakIdx[j]; c.peakBin = peakBin[j]; c.peakFrame = firstFrame + j; }
      j++;
    }
    c.endFrame = firstFrame + j - 1;
    c.endBin = peakBin[j - 1];
    if (j - i >= minFrames) calls.push(c);
    i = j;
  }
  return calls;
}

function callToAutoIdData(call, fftN, hop, sampleRate, { callType, harmonic = 0, timeExpansion = 1 } = {}) {
  const khz = (b) => b * sampleRate / fftN / 1000 * timeExpansion;
  const frames = call.endFrame - call.startFrame + 1;
  const d = {
    highestFreq: khz(call.highBin), lowestFreq: khz(call.lowBin), startFreq: khz(call.startBin), endFreq: khz(call.endBin),
    peakFreq: khz(call.peakBin), duration: frames * hop / sampleRate * 1000 / timeExpansion,
    bandwidth: khz(call.highBin) - khz(call.lowBin), harmonic,
    startTime: call.startFrame * hop / sampleRate * 1000 / timeExpansion,
  };
  if (callType !== undefined) d.callType = callType;
  return d;
}

function inRange(val, range) {                                       // autoid_HK.js:200-213
  if (val == null || isNaN(val)) return false;
  if (Array.isArray(range[0])) return range.some((r) => inRange(val, r));
  return val >= range[0] && val <= range[1];
}

function autoIdHK(data = {}, speciesRules) {                         // autoid_HK.js:215-264
  const matches = speciesRules.filter((species) => species.rules.some((rule) => {
    if (rule.callType) {
      const callTypes = rule.callType.split(',').map((s) => s.trim());
      if (!callTypes.includes(data.callType)) return false;
    }
    if (rule.harmonic && !rule.harmonic.includes(data.harmonic)) return false;
    return FIELDS.every((f) => {
      if (!rule[f]) return true;
      if (typeof rule[f][0] === 'string' && rule[f].length === 1) {
        const m = rule[f][0].match(REL);
        if (m) {
          const val = data[f], refVal = data[m[2]];
          if (val == null || refVal == null || isNaN(val) || isNaN(refVal)) return false;
          switch (m[1]) {
            case '=': return val === refVal;
            case '>': return val > refVal;
            case '<': return val < refVal;
            case '>=': case '=>': return val >= refVal;
            case '<=': return val <= refVal;
            default: return false;
          }
        }
      }
      return inRange(data[f], rule[f]);
    });
  })).map((s) => s.name);
  return matches.length ? matches.join(' / ') : 'No species matched';
}

module.exports = { segmentCalls, callToAutoIdData, autoIdHK, FIELDS };
