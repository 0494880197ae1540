// File intake on the host: RIFF/WAVE chunk walk, GUANO metadata, crop by time, and the load filters,
// with the reference's semantics (modules/fileLoader.js:7-63,179-181, modules/guanoReader.js:3-80,
// modules/cropAudio.js:1-59).  No GPU code here; the PCM conversion kernel is in kernels_post.hip.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "sonospec.h"

namespace {

inline uint16_t u16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

bool contains(const uint8_t* hay, uint64_t n, const char* needle) {
  const size_t m = std::strlen(needle);
  if (n < m) return false;
  for (uint64_t i = 0; i + m <= n; i++)
    if (hay[i] == (uint8_t)needle[0] && std::memcmp(hay + i, needle, m) == 0) return true;
  return false;
}

// ---- the pieces of JavaScript string/number behaviour parseGuanoMetadata relies on
bool js_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && js_space((unsigned char)s[a])) a++;
  while (b > a && js_space((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}
std::vector<std::string> split_char(const std::string& s, char c) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    const size_t p = s.find(c, a);
    if (p == std::string::npos) { out.push_back(s.substr(a)); break; }
    out.push_back(s.substr(a, p - a));
    a = p + 1;
  }
  return out;
}
std::vector<std::string> split_ws(const std::string& s) {  // String.prototype.split(/\s+/)
  std::vector<std::string> out;
  size_t a = 0;
  for (size_t i = 0; i <= s.size();) {
    if (i < s.size() && js_space((unsigned char)s[i])) {
      size_t j = i;
      while (j < s.size() && js_space((unsigned char)s[j])) j++;
      out.push_back(s.substr(a, i - a));
      a = i = j;
    } else if (i == s.size()) {
      out.push_back(s.substr(a));
      break;
    } else {
      i++;
    }
  }
  return out;
}
// parseFloat: the longest prefix that is a decimal literal; NaN if there is none
double js_parse_float(const std::string& s) {
  size_t i = 0;
  while (i < s.size() && js_space((unsigned char)s[i])) i++;
  const size_t start = i;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) i++;
  if (s.compare(i, 8, "Infinity") == 0) return s[start] == '-' ? -INFINITY : INFINITY;
  size_t digits = 0;
  while (i < s.size() && s[i] >= '0' && s[i] <= '9') { i++; digits++; }
  if (i < s.size() && s[i] == '.') {
    size_t j = i + 1, frac = 0;
    while (j < s.size() && s[j] >= '0' && s[j] <= '9') { j++; frac++; }
    if (digits + frac > 0) { i = j; digits += frac; }
  }
  if (digits == 0) return NAN;
  if (i < s.size() && (s[i] == 'e' || s[i] == 'E')) {
    size_t j = i + 1;
    if (j < s.size() && (s[j] == '+' || s[j] == '-')) j++;
    size_t ed = 0;
    while (j < s.size() && s[j] >= '0' && s[j] <= '9') { j++; ed++; }
    if (ed) i = j;
  }
  return std::strtod(s.substr(start, i - start).c_str(), nullptr);
}
// Number.prototype.toString(): shortest digits that round-trip, ECMAScript layout
std::string js_number_to_string(double v) {
  if (std::isnan(v)) return "NaN";
  if (v == 0) return "0";
  if (std::isinf(v)) return v < 0 ? "-Infinity" : "Infinity";
  std::string sign = v < 0 ? "-" : "";
  v = std::fabs(v);
  char buf[64];
  std::string dig;
  int e10 = 0;
  for (int p = 1; p <= 17; p++) {
    std::snprintf(buf, sizeof buf, "%.*e", p - 1, v);
    if (std::strtod(buf, nullptr) == v) {
      const char* ep = std::strchr(buf, 'e');
      e10 = std::atoi(ep + 1);
      dig.clear();
      for (const char* c = buf; c < ep; c++)
        if (*c >= '0' && *c <= '9') dig.push_back(*c);
      break;
    }
  }
  while (dig.size() > 1 && dig.back() == '0') dig.pop_back();
  const int k = (int)dig.size(), n = e10 + 1;
  std::string out;
  if (k <= n && n <= 21) {
    out = dig + std::string((size_t)(n - k), '0');
  } else if (0 < n && n <= 21) {
    out = dig.substr(0, (size_t)n) + "." + dig.substr((size_t)n);
  } else if (-6 < n && n <= 0) {
    out = "0." + std::string((size_t)(-n), '0') + dig;
  } else {
    const int e = n - 1;
    out = dig.substr(0, 1);
    if (k > 1) out += "." + dig.substr(1);
    out += (e < 0 ? "e-" : "e+") + std::to_string(e < 0 ? -e : e);
  }
  return sign + out;
}
void put(char* dst, size_t cap, const std::string& s) {
  const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  std::memcpy(dst, s.data(), n);
  dst[n] = 0;
}

}  // namespace

extern "C" {

int sono_wav_parse(const void* file, uint64_t bytes, sono_wav_info* out) {
  if (!file || !out) return SONO_ERR_ARG;
  std::memset(out, 0, sizeof *out);
  out->sample_rate = 256000;  // getWavSampleRate's answer when there is no fmt chunk (fileLoader.js:8,26)
  const uint8_t* b = static_cast<const uint8_t*>(file);
  if (bytes < 12) return SONO_ERR_FORMAT;
  // fileLoader.js:11-25 / :33-56 / cropAudio.js:12-30: chunks from offset 12 while pos < length - 8
  bool rate_taken = false;
  for (uint64_t pos = 12; pos + 8 < bytes;) {
    const uint32_t size = u32(b + pos + 4);
    if (std::memcmp(b + pos, "fmt ", 4) == 0) {
      if (pos + 8 + 16 > bytes) return SONO_ERR_FORMAT;  // the reference's DataView read would throw here
      if (!rate_taken) {  // getWavSampleRate returns at the first fmt chunk, wherever it is
        out->sample_rate = u32(b + pos + 12);
        rate_taken = true;
      }
      if (!out->has_data) {  // getWavDuration and cropWavBlob stop at the data chunk: the last fmt before it counts
        out->has_fmt = 1;
        out->fmt_offset = pos + 8;
        out->fmt_bytes = size;
        out->format_tag = u16(b + pos + 8);
        out->channels = u16(b + pos + 10);
        out->fmt_sample_rate = u32(b + pos + 12);
        out->bits_per_sample = u16(b + pos + 22);
      }
    } else if (std::memcmp(b + pos, "data", 4) == 0 && !out->has_data) {
      out->has_data = 1;
      out->data_offset = pos + 8;
      out->data_bytes = size;
      const uint64_t avail = bytes - (pos + 8);
      out->data_bytes_present = size < avail ? size : avail;
    }
    pos += 8 + (uint64_t)size + (size & 1);
  }
  if (out->has_fmt && out->format_tag == 0xFFFE && out->fmt_bytes >= 26 && out->fmt_offset + 26 <= bytes)
    out->format_tag = u16(b + out->fmt_offset + 24);  // WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes
  // getWavDuration, fileLoader.js:57-62 (declared data size, floating division)
  if (out->has_fmt && out->has_data && out->fmt_sample_rate > 0 && out->data_bytes > 0) {
    const double bytes_per_sample = ((double)out->bits_per_sample / 8.0) * (double)out->channels;
    out->duration_s = ((double)out->data_bytes / bytes_per_sample) / (double)out->fmt_sample_rate;
  }
  const uint64_t block = (uint64_t)out->channels * out->bits_per_sample / 8;
  out->block_align = (uint32_t)block;
  out->frames = (out->has_fmt && out->has_data && block) ? out->data_bytes_present / block : 0;
  return SONO_OK;
}

int sono_wav_guano(const void* file, uint64_t bytes, uint64_t* offset, uint64_t* length) {
  if (!file || !offset || !length) return SONO_ERR_ARG;
  *offset = *length = 0;
  const uint8_t* b = static_cast<const uint8_t*>(file);
  // guanoReader.js:9-31: every chunk from offset 12 while pos < length; the first one whose text has the marker
  for (uint64_t pos = 12; pos < bytes;) {
    if (pos + 8 > bytes) return SONO_ERR_FORMAT;  // the reference throws a RangeError on a truncated chunk header
    const uint32_t size = u32(b + pos + 4);
    if (pos + 8 + (uint64_t)size > bytes) return SONO_ERR_FORMAT;  // ... and on a chunk that runs past the file
    if (contains(b + pos + 8, size, "GUANO|Version:")) {
      *offset = pos + 8;
      *length = size;
      return SONO_OK;
    }
    pos += 8 + (uint64_t)size + (size & 1);
  }
  return SONO_OK;
}

int sono_guano_fields(const char* text, uint64_t bytes, sono_guano_fields_t* out) {
  if (!out || (!text && bytes)) return SONO_ERR_ARG;
  std::memset(out, 0, sizeof *out);
  const std::string t(text ? text : "", (size_t)bytes);
  if (t.empty() || t.compare(0, 9, "(No GUANO") == 0) return SONO_OK;  // guanoReader.js:37 -> {}
  out->present = 1;
  std::map<std::string, std::string> meta;
  size_t a = 0;
  while (a <= t.size()) {  // split(/\r?\n/)
    size_t p = t.find('\n', a);
    std::string line = t.substr(a, p == std::string::npos ? std::string::npos : p - a);
    if (p != std::string::npos && !line.empty() && line.back() == '\r') line.pop_back();
    const size_t c = line.find(':');
    if (c != std::string::npos) meta[trim(line.substr(0, c))] = trim(line.substr(c + 1));
    if (p == std::string::npos) break;
    a = p + 1;
  }
  std::string date, time, lat, lon;
  auto ts = meta.find("Timestamp");
  if (ts != meta.end() && !ts->second.empty()) {  // :48-54
    const std::vector<std::string> parts = split_char(ts->second, ' ');
    const std::string date_part = parts[0];
    const std::string tz = parts.size() > 1 ? parts[1] : "";
    const std::string time_part = split_char(tz, '+')[0];
    date = date_part;
    for (char& ch : date)
      if (ch == '-') ch = '/';
    if (!time_part.empty()) {
      time = time_part.substr(0, 5);
      const size_t c = time.find(':');
      if (c != std::string::npos) time.erase(c, 1);
    }
  }
  auto lp = meta.find("Loc Position");
  if (lp != meta.end() && !lp->second.empty()) {  // :56-72
    const std::vector<std::string> parts = split_ws(lp->second);
    lat = parts[0];
    if (parts.size() > 1 && !parts[1].empty()) {
      double v = js_parse_float(parts[1]);
      if (!std::isnan(v)) {
        if (v < 0 && std::fabs(v) >= 113 && std::fabs(v) <= 115) v = std::fabs(v);
        lon = js_number_to_string(v);
      } else {
        lon = parts[1];
      }
    }
  }
  put(out->date, sizeof out->date, date);
  put(out->time, sizeof out->time, time);
  put(out->latitude, sizeof out->latitude, lat);
  put(out->longitude, sizeof out->longitude, lon);
  return SONO_OK;
}

int64_t sono_wav_crop(const void* file, uint64_t bytes, double start_s, double end_s, void* out, uint64_t capacity) {
  if (!file) return SONO_ERR_ARG;
  if (!(start_s >= 0) || !(end_s >= 0)) return SONO_ERR_ARG;  // (negative times make the reference slice from the file's end)
  sono_wav_info w;
  const int rc = sono_wav_parse(file, bytes, &w);
  if (rc != SONO_OK) return rc;
  if (!w.has_fmt || !w.has_data) return 0;  // cropAudio.js:32 -> null
  const uint8_t* b = static_cast<const uint8_t*>(file);
  // :34-44, in the reference's double arithmetic
  const double channels = u16(b + w.fmt_offset + 2), rate = u32(b + w.fmt_offset + 4), bits = u16(b + w.fmt_offset + 14);
  const double block = channels * bits / 8;
  const double start_sample = std::floor(start_s * rate), end_sample = std::floor(end_s * rate);
  const double start_byte = (double)w.data_offset + start_sample * block;
  double end_byte = (double)w.data_offset + end_sample * block;
  const double data_end = (double)w.data_offset + (double)w.data_bytes;
  if (data_end < end_byte) end_byte = data_end;
  if (end_byte <= start_byte) return 0;  // null
  const double new_len = end_byte - start_byte;
  if (new_len != std::floor(new_len) || start_byte != std::floor(start_byte)) return SONO_ERR_FORMAT;  // fractional block size
  const uint64_t total = w.data_offset + (uint64_t)new_len;
  if (!out) return (int64_t)total;  // size query
  if (capacity < total) return SONO_ERR_CAPACITY;
  uint8_t* o = static_cast<uint8_t*>(out);
  std::memset(o, 0, total);
  std::memcpy(o, b, w.data_offset);  // everything up to the data payload, other chunks included (:47,51)
  const uint64_t sb = (uint64_t)start_byte;
  if (sb < bytes) {                   // ArrayBuffer.slice clips to the file (:48)
    uint64_t n = (uint64_t)new_len;
    if (sb + n > bytes) n = bytes - sb;
    std::memcpy(o + w.data_offset, b + sb, n);
  }
  put32(o + 4, (uint32_t)(total - 8));                 // :55
  put32(o + w.data_offset - 4, (uint32_t)new_len);     // :56
  return (int64_t)total;
}

int sono_wav_accept(uint64_t file_bytes, double duration_s, int time_expansion) {
  if (file_bytes < 200 * 1024) return 1;               // fileLoader.js:179 "skipped due to small file size (<200kb)"
  if (duration_s > 20 && !time_expansion) return 2;    // :181 "longer than 20 seconds are not supported"
  return 0;
}

}  // extern "C"
