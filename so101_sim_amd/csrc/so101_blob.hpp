// Host-side reader of the model blob (so101_sim_amd/model/blob.py): a directory of named int32 / float32 arrays.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct BlobView {
  std::map<std::string, std::pair<const uint8_t*, uint32_t>> ent;
  bool parse(const void* p, size_t bytes, std::string& err) {
    if (bytes < 16) { err = "blob too small"; return false; }
    const uint8_t* b = (const uint8_t*)p;
    uint32_t magic, ver, rb, n;
    memcpy(&magic, b, 4); memcpy(&ver, b + 4, 4); memcpy(&rb, b + 8, 4); memcpy(&n, b + 12, 4);
    if (magic != 0x424D3153u) { err = "bad blob magic"; return false; }
    if (ver != 3) { err = "unsupported blob version"; return false; }
    if (rb != 4) { err = "the HIP library needs the f32 blob"; return false; }
    if (16 + (size_t)48 * n > bytes) { err = "truncated blob directory"; return false; }
    for (uint32_t k = 0; k < n; k++) {
      const uint8_t* e = b + 16 + 48 * k;
      char name[33]; memcpy(name, e, 32); name[32] = 0;
      uint32_t kd, cnt; uint64_t off;
      memcpy(&kd, e + 32, 4); memcpy(&cnt, e + 36, 4); memcpy(&off, e + 40, 8);
      if (off > bytes || (uint64_t)cnt * 4 > bytes - off) { err = std::string("truncated blob entry ") + name; return false; }
      ent[name] = {b + off, cnt};
    }
    return true;
  }
  bool has(const char* n) const { return ent.count(n) != 0; }
  size_t count(const char* n) const { auto it = ent.find(n); return it == ent.end() ? 0 : it->second.second; }
  std::vector<int> I(const char* n) const {
    std::vector<int> v; auto it = ent.find(n);
    if (it != ent.end()) { v.resize(it->second.second); memcpy(v.data(), it->second.first, 4 * v.size()); }
    return v;
  }
  std::vector<float> F(const char* n) const {
    std::vector<float> v; auto it = ent.find(n);
    if (it != ent.end()) { v.resize(it->second.second); memcpy(v.data(), it->second.first, 4 * v.size()); }
    return v;
  }
};
