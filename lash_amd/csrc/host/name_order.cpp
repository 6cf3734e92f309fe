// name_order.cpp — see name_order.hpp.
#include "name_order.hpp"

#include <cstring>
#include <unordered_map>

namespace lashhost {
namespace {

// XXH3's default 192-byte secret (xxHash v0.8 XXH3_kSecret; xxhash-rust 0.8.15 DEFAULT_SECRET)
const uint8_t K_SECRET[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

constexpr uint64_t P32_1 = 0x9E3779B1u, P32_2 = 0x85EBCA77u, P32_3 = 0xC2B2AE3Du;
constexpr uint64_t P64_1 = 0x9E3779B185EBCA87ull, P64_2 = 0xC2B2AE3D27D4EB4Full, P64_3 = 0x165667B19E3779F9ull,
                   P64_4 = 0x85EBCA77C2B2AE63ull, P64_5 = 0x27D4EB2F165667C5ull;
constexpr uint64_t MX1 = 0x165667919E3779F9ull, MX2 = 0x9FB21C651E98DF25ull;

inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }      // little-endian hosts only
inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fold(uint64_t a, uint64_t b) { const unsigned __int128 m = (unsigned __int128)a * b; return (uint64_t)m ^ (uint64_t)(m >> 64); }
inline uint64_t avalanche3(uint64_t h) { h ^= h >> 37; h *= MX1; return h ^ (h >> 32); }
inline uint64_t avalanche64(uint64_t h) { h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; return h ^ (h >> 32); }
inline uint64_t mix16(const uint8_t *in, const uint8_t *sec, uint64_t seed) { return fold(rd64(in) ^ (rd64(sec) + seed), rd64(in + 8) ^ (rd64(sec + 8) - seed)); }

void accumulate_stripe(uint64_t acc[8], const uint8_t *in, const uint8_t *sec)
{
    for (int i = 0; i < 8; ++i) {
        const uint64_t v = rd64(in + 8 * i), key = v ^ rd64(sec + 8 * i);
        acc[i ^ 1] += v;
        acc[i] += (key & 0xFFFFFFFFull) * (key >> 32);
    }
}

uint64_t hash_long(const uint8_t *p, size_t n, uint64_t seed)
{
    uint8_t sec[192];
    for (int i = 0; i < 12; ++i) {                     // the seed-derived secret (seed == 0 leaves K_SECRET as it is)
        const uint64_t lo = rd64(K_SECRET + 16 * i) + seed, hi = rd64(K_SECRET + 16 * i + 8) - seed;
        memcpy(sec + 16 * i, &lo, 8); memcpy(sec + 16 * i + 8, &hi, 8);
    }
    uint64_t acc[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    const size_t stripes_per_block = (192 - 64) / 8, block = 64 * stripes_per_block, n_blocks = (n - 1) / block;
    for (size_t b = 0; b < n_blocks; ++b) {
        for (size_t s = 0; s < stripes_per_block; ++s) accumulate_stripe(acc, p + b * block + 64 * s, sec + 8 * s);
        for (int i = 0; i < 8; ++i) { uint64_t a = acc[i]; a ^= a >> 47; a ^= rd64(sec + 192 - 64 + 8 * i); acc[i] = a * P32_1; }
    }
    const size_t n_stripes = ((n - 1) - block * n_blocks) / 64;
    for (size_t s = 0; s < n_stripes; ++s) accumulate_stripe(acc, p + n_blocks * block + 64 * s, sec + 8 * s);
    accumulate_stripe(acc, p + n - 64, sec + 192 - 64 - 7);
    uint64_t r = (uint64_t)n * P64_1;
    for (int i = 0; i < 4; ++i) r += fold(acc[2 * i] ^ rd64(sec + 11 + 16 * i), acc[2 * i + 1] ^ rd64(sec + 11 + 16 * i + 8));
    return avalanche3(r);
}

}  // namespace

uint64_t xxh3_64_seeded(const uint8_t *p, size_t n, uint64_t seed)
{
    const uint8_t *s = K_SECRET;
    if (n == 0) return avalanche64(seed ^ (rd64(s + 56) ^ rd64(s + 64)));
    if (n <= 3) {
        const uint32_t combined = ((uint32_t)p[0] << 16) | ((uint32_t)p[n >> 1] << 24) | (uint32_t)p[n - 1] | ((uint32_t)n << 8);
        return avalanche64((uint64_t)combined ^ ((uint64_t)(rd32(s) ^ rd32(s + 4)) + seed));
    }
    if (n <= 8) {
        seed ^= (uint64_t)__builtin_bswap32((uint32_t)seed) << 32;
        const uint64_t in64 = (uint64_t)rd32(p + n - 4) + ((uint64_t)rd32(p) << 32);
        uint64_t h = in64 ^ ((rd64(s + 8) ^ rd64(s + 16)) - seed);
        h ^= rotl(h, 49) ^ rotl(h, 24);
        h *= MX2;
        h ^= (h >> 35) + n;
        h *= MX2;
        return h ^ (h >> 28);
    }
    if (n <= 16) {
        const uint64_t lo = rd64(p) ^ ((rd64(s + 24) ^ rd64(s + 32)) + seed), hi = rd64(p + n - 8) ^ ((rd64(s + 40) ^ rd64(s + 48)) - seed);
        return avalanche3((uint64_t)n + __builtin_bswap64(lo) + hi + fold(lo, hi));
    }
    if (n <= 128) {
        uint64_t acc = (uint64_t)n * P64_1;
        if (n > 32) {
            if (n > 64) {
                if (n > 96) { acc += mix16(p + 48, s + 96, seed); acc += mix16(p + n - 64, s + 112, seed); }
                acc += mix16(p + 32, s + 64, seed); acc += mix16(p + n - 48, s + 80, seed);
            }
            acc += mix16(p + 16, s + 32, seed); acc += mix16(p + n - 32, s + 48, seed);
        }
        acc += mix16(p, s, seed); acc += mix16(p + n - 16, s + 16, seed);
        return avalanche3(acc);
    }
    if (n <= 240) {
        uint64_t acc = (uint64_t)n * P64_1;
        const size_t rounds = n / 16;
        for (size_t i = 0; i < 8; ++i) acc += mix16(p + 16 * i, s + 16 * i, seed);
        acc = avalanche3(acc);
        for (size_t i = 8; i < rounds; ++i) acc += mix16(p + 16 * i, s + 16 * (i - 8) + 3, seed);
        acc += mix16(p + n - 16, s + 136 - 17, seed);
        return avalanche3(acc);
    }
    return hash_long(p, n, seed);
}

std::vector<uint32_t> hashbrown_key_order(const std::vector<std::string> &names, uint64_t seed)
{
    // One 16-byte group read at `pos` sees buckets pos .. pos+15 circularly (the control bytes past the end mirror the
    // first 16, and for tables smaller than a group the padding reads EMPTY and fix_insert_slot restarts from bucket
    // 0), so "lowest empty bit of the group" is the first free bucket of a circular scan of min(16, buckets) from
    // pos; a full group advances pos by 16, 32, 48, ... (triangular probing).  No tombstones: nothing is removed.
    struct Table {
        std::vector<int64_t> slot;      // name index or -1
        size_t items = 0;
        size_t capacity() const { const size_t b = slot.size(); return b == 0 ? 0 : b < 8 + 1 ? b - 1 : b / 8 * 7; }
        void place(uint64_t hash, int64_t v)
        {
            const size_t mask = slot.size() - 1, width = slot.size() < 16 ? slot.size() : 16;
            size_t pos = (size_t)hash & mask, stride = 0;
            for (;;) {
                for (size_t b = 0; b < width; ++b)
                    if (slot[(pos + b) & mask] < 0) { slot[(pos + b) & mask] = v; ++items; return; }
                stride += 16; pos = (pos + stride) & mask;
            }
        }
    };
    auto buckets_for = [](size_t cap) -> size_t {      // capacity_to_buckets, 16-byte (&String, &Sketch) entries
        if (cap < 4) return 4;
        if (cap < 8) return 8;
        if (cap < 15) return 16;
        size_t want = cap * 8 / 7, b = 1;
        while (b < want) b <<= 1;
        return b;
    };
    std::vector<uint64_t> hashes(names.size());
    std::string key;
    for (size_t i = 0; i < names.size(); ++i) {        // Hash for str: the bytes, then 0xFF (hasher.rs:24-26 streams both)
        key.assign(names[i]); key.push_back((char)0xFF);
        hashes[i] = xxh3_64_seeded((const uint8_t *)key.data(), key.size(), seed);
    }
    Table t;
    std::unordered_map<std::string, size_t> first;      // name -> index whose hash / key sits in the table
    std::vector<uint32_t> value(names.size());          // first index -> index of the value it now carries
    for (size_t i = 0; i < names.size(); ++i) {
        // RawTable::find_or_find_insert_slot reserves room for one more entry BEFORE it looks the key up
        if (t.items == t.capacity()) {
            Table grown;
            grown.slot.assign(buckets_for(t.capacity() + 1), -1);
            for (int64_t v : t.slot) if (v >= 0) grown.place(hashes[(size_t)v], v);
            t = std::move(grown);
        }
        auto it = first.find(names[i]);
        if (it != first.end()) { value[it->second] = (uint32_t)i; continue; }
        first.emplace(names[i], i);
        value[i] = (uint32_t)i;
        t.place(hashes[i], (int64_t)i);
    }
    std::vector<uint32_t> order;
    order.reserve(t.items);
    for (int64_t v : t.slot) if (v >= 0) order.push_back(value[(size_t)v]);
    return order;
}

}  // namespace lashhost
