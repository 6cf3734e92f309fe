#!/usr/bin/env python3
"""Generate tests/golden/xxh3_vectors.json with python-xxhash (libxxhash 0.8.2).

Run in the BUILD container only (python-xxhash is a third-party pin for the XXH3 layer that
xxhash-rust 0.8.15 implements, Cargo.lock:2326; /root/reference itself has no vectors).
The reference hashes   (masked as u32).to_le_bytes()  with XXH3-128 (utils.rs:397, inside
hyperminhash) and       masked.to_le_bytes()          with XXH3-64  (utils.rs:412, 428).
Nothing here travels to the GPU box except the JSON it writes.
"""
import json, os, random, struct
import xxhash

SEEDS = [0, 42, 93, 2**63 + 5, 2**64 - 1, 0x0123456789ABCDEF]
rng = random.Random(20260128)
vals64 = [0, 1, 12345, 0xFFFFFFFF, 0x100000000, 2**64 - 1, 0x1be4e4d8, 0x6f93936368] + \
         [rng.getrandbits(64) for _ in range(200)] + [rng.getrandbits(32) for _ in range(48)]
vals32 = [0, 1, 12345, 0xFFFFFFFF, 0x1be4e4d8, 0x36393906, 0x8d8e4e41, 0x93936368] + \
         [rng.getrandbits(32) for _ in range(248)]

out = {"libxxhash": xxhash.XXHASH_VERSION, "python_xxhash": xxhash.VERSION, "seeds": [str(s) for s in SEEDS],
       "xxh3_64_of_le8": [], "xxh3_128_of_le4": []}
for s in SEEDS:
    out["xxh3_64_of_le8"].append([[str(v), str(xxhash.xxh3_64_intdigest(struct.pack("<Q", v), seed=s))] for v in vals64])
    out["xxh3_128_of_le4"].append([[str(v), str(xxhash.xxh3_128_intdigest(struct.pack("<I", v), seed=s))] for v in vals32])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xxh3_vectors.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote", path, os.path.getsize(path), "bytes")
