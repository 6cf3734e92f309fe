// name_order.hpp — the order in which the reference's dist walks its sketch maps.
//
// /root/reference/src/utils.rs:111-127 (and 208-209, 307-308) puts the sketches into hashbrown::HashMap<&String, _>
// built with hasher.rs's Xxh3Builder { seed: 93 } "to keep key order deterministic", then takes column order, the
// same-files triangle (utils.rs:135-142, 158-160) and — under rayon with one thread — row order from `.keys()`.
// That order is a function of (a) XXH3-64(seed 93) of the name bytes followed by 0xFF (Hash for str) and (b)
// hashbrown 0.15.4's open-addressing table on x86-64 (16-wide SSE2 groups): insertion in file order, growth
// 0 -> 4 -> 8 -> 16 -> next_pow2(cap * 8 / 7) buckets, re-insertion in bucket order on growth, iteration in bucket
// order.  Both are restated here; neither dependency is vendored under /root/reference (Cargo.lock: hashbrown
// 0.15.4, xxhash-rust 0.8.15).  The hash is pinned by the python-xxhash module in tests/test_host.py; the table walk
// is pinned only by tools/ref_probe's dist cases.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace lashhost {

// XXH3_64bits_withSeed over n bytes (every length class: 0-16, 17-128, 129-240, > 240 with the seed-derived secret).
uint64_t xxh3_64_seeded(const uint8_t *p, size_t n, uint64_t seed);

// Indices into `names` in the order `HashMap::with_hasher(Xxh3Builder{seed}).keys()` yields them after inserting
// names[0], names[1], ... .  A repeated name keeps its first key and its LAST value (HashMap::insert replaces the
// value), so the entry for it is the index of its last occurrence.
std::vector<uint32_t> hashbrown_key_order(const std::vector<std::string> &names, uint64_t seed = 93);

}  // namespace lashhost
