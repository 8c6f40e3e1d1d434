// Host-side helper of the device Preparator (include/urcco.h, urcco_hash_strings): 64-bit keys of id strings.
// XXH64 (the public xxHash algorithm, restated from its specification) over the UTF-8 bytes of each string, spread over
// a few host threads -- the Python host mirror used to do this in an interpreter loop.  Pure host code.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "urcco_internal.h"

using namespace urcco_detail;

namespace {

constexpr uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull, P4 = 9650029242287828579ull,
                   P5 = 2870177450012600261ull;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }  // little-endian host (x86-64)
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t round1(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t merge(uint64_t acc, uint64_t v) { return (acc ^ round1(0, v)) * P1 + P4; }

uint64_t xxh64(const uint8_t* p, size_t len, uint64_t seed) {
  const uint8_t* const end = p + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    const uint8_t* const limit = end - 32;
    do {
      v1 = round1(v1, rd64(p));
      v2 = round1(v2, rd64(p + 8));
      v3 = round1(v3, rd64(p + 16));
      v4 = round1(v4, rd64(p + 24));
      p += 32;
    } while (p <= limit);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = merge(h, v1);
    h = merge(h, v2);
    h = merge(h, v3);
    h = merge(h, v4);
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) {
    h ^= round1(0, rd64(p));
    h = rotl(h, 27) * P1 + P4;
    p += 8;
  }
  if (p + 4 <= end) {
    h ^= (uint64_t)rd32(p) * P1;
    h = rotl(h, 23) * P2 + P3;
    p += 4;
  }
  while (p < end) {
    h ^= (uint64_t)(*p) * P5;
    h = rotl(h, 11) * P1;
    ++p;
  }
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

}  // namespace

extern "C" int urcco_hash_strings(const uint8_t* bytes, const int64_t* offsets, int64_t n, uint64_t seed, uint64_t* keys) {
  return guarded([&]() -> int {
    if (n < 0 || (n > 0 && (!offsets || !keys)) || (n > 0 && offsets[n] > offsets[0] && !bytes)) return fail(URCCO_BAD_ARG, "urcco_hash_strings: bad argument");
    const unsigned hc = std::thread::hardware_concurrency();
    const int nt = n < (1 << 16) ? 1 : (int)(hc >= 16 ? 8 : (hc >= 4 ? 4 : 1));
    auto work = [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        uint64_t h = xxh64(bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), seed);
        keys[i] = h == ~0ull ? 0ull : h;  // ~0 is the device dictionary's reserved "empty slot" value
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
    work(0, n / nt);
    for (auto& t : th) t.join();
    return URCCO_OK;
  });
}
