"""Naive pure-Python restatement of the reference hot path, for SMALL cases only.

Independent of oracle/lash_oracle.c in style: every k-mer is packed from scratch from the filtered
string, the reverse complement is built by string reversal, registers are updated one k-mer at a
time.  Reference lines: utils.rs:33-41 (filter), 57-64 (mask), 457-505 (loops), 395-429 (add_kmer).
"""
import struct

M64 = (1 << 64) - 1
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}

P64_1 = 0x9E3779B185EBCA87
MX1 = 0x165667919E3779F9
MX2 = 0x9FB21C651E98DF25
SEC8, SEC16, SEC24 = 0x1CAD21F72C81017C, 0xDB979083E96DD4DE, 0x1F67B3B7A4A44072


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _short_seed(seed):
    lo = seed & 0xFFFFFFFF
    return seed ^ (int.from_bytes(lo.to_bytes(4, "little"), "big") << 32)


def xxh3_64_8b(v, seed):
    s = _short_seed(seed)
    bitflip = ((SEC8 ^ SEC16) - s) & M64
    keyed = (((v >> 32) | ((v & 0xFFFFFFFF) << 32)) ^ bitflip) & M64
    h = keyed
    h ^= _rotl(h, 49) ^ _rotl(h, 24)
    h = (h * MX2) & M64
    h ^= ((h >> 35) + 8)
    h = (h * MX2) & M64
    return h ^ (h >> 28)


def xxh3_64_8b_inverse(h, seed):
    """The 8-byte value whose xxh3_64 under `seed` is h: every step of the len-4..8 branch is a bijection of 64-bit words (an odd multiplier,
    an xorshift, x ^ ((x >> 35) + 8), and x ^ rotl(x, 49) ^ rotl(x, 24), whose 64th power is the identity).  Tests use it to BUILD k-mers
    with hashes one would wait 2^32 k-mers and more for (tests/test_gpu_rare_hashes.py)."""
    inv = pow(MX2, -1, 1 << 64)
    h ^= (h >> 28) ^ (h >> 56)
    h = (h * inv) & M64
    h ^= ((h >> 35) + 8)                      # (bits 63..35 pass unchanged, and they are all the step reads)
    h = (h * inv) & M64
    for _ in range(63):
        h ^= _rotl(h, 49) ^ _rotl(h, 24)
    s = _short_seed(seed)
    keyed = h ^ (((SEC8 ^ SEC16) - s) & M64)
    return ((keyed >> 32) | ((keyed & 0xFFFFFFFF) << 32)) & M64


def xxh3_128_4b(w, seed):
    s = _short_seed(seed)
    bitflip = ((SEC16 ^ SEC24) + s) & M64
    keyed = ((w | (w << 32)) ^ bitflip) & M64
    m = keyed * ((P64_1 + 16) & M64)
    lo, hi = m & M64, (m >> 64) & M64
    hi = (hi + ((lo << 1) & M64)) & M64
    lo ^= hi >> 3
    lo ^= lo >> 35
    lo = (lo * MX2) & M64
    lo ^= lo >> 28
    hi ^= hi >> 37
    hi = (hi * MX1) & M64
    hi ^= hi >> 32
    return lo, hi


def filter_out_n(seq: str) -> str:
    return "".join(c for c in seq if c in "ACGT")


class Lay:
    """SURVEY App. D's unknowns as data (mirrors oracle_lib.make_layout's keyword arguments)."""

    def __init__(self, codes="ACGT", kmer="msb", hmh_x="high", hmh_reg="le", hll_bucket="low", hmh_hdr="", hll_hdr="azspl",
                 ull_hdr="l"):
        self.code = {letter: c for c, letter in enumerate(codes)}
        self.lsb_first = kmer == "lsb"
        self.x_low = hmh_x == "low"
        self.reg_be = hmh_reg == "be"
        self.bucket_high = hll_bucket == "high"
        self.hdr = {"hmh": hmh_hdr, "hll": hll_hdr, "ull": ull_hdr}

    def header(self, algo, p, n_regs, zero=0, tot=0.0):
        alpha = {4: 0.673, 5: 0.697, 6: 0.709}.get(p, 0.7213 / (1.0 + 1.079 / (1 << p)))
        enc = {"a": ("<d", alpha), "z": ("<Q", zero), "Z": ("<I", zero), "s": ("<d", tot), "p": ("<B", p), "P": ("<I", p),
               "Q": ("<Q", p), "l": ("<Q", n_regs), "L": ("<I", n_regs)}
        return b"".join(struct.pack(*enc[c]) for c in self.hdr[algo])


DEFAULT = Lay()


def pack(kmer: str, lay=DEFAULT) -> int:
    v = 0
    for c in (reversed(kmer) if lay.lsb_first else kmer):      # lsb-first: the first base ends up in the lowest bits
        v = (v << 2) | lay.code[c]
    return v


def canonical_kmers(record: str, k: int, lay=DEFAULT):
    s = filter_out_n(record)
    if len(s) < k:
        return []
    out = []
    for i in range(len(s) - k + 1):
        km = s[i:i + k]
        rc = "".join(COMP[c] for c in reversed(km))
        canon = min(pack(km, lay), pack(rc, lay))
        if k <= 14 or k == 16:
            canon &= 0xFFFFFFFF
        if 2 * k < 64:
            canon &= (1 << (2 * k)) - 1
        out.append(canon)
    return out


def clz64(x):
    return 64 - x.bit_length()


def hmh_sketch(records, k, seed, x_is_low=False, lay=DEFAULT):
    regs = [0] * 16384
    for rec in records:
        for km in canonical_kmers(rec, k, lay):
            lo, hi = xxh3_128_4b(km & 0xFFFFFFFF, seed)
            x, y = (lo, hi) if (x_is_low or lay.x_low) else (hi, lo)
            bucket = x >> 50
            lz = clz64(((x << 14) & M64) ^ 0x3FFF) + 1
            reg = (lz << 10) | (y & 0x3FF)
            if regs[bucket] < reg:
                regs[bucket] = reg
    return lay.header("hmh", 14, 16384) + b"".join(struct.pack(">H" if lay.reg_be else "<H", r) for r in regs)


def hll_sketch(records, k, p, seed, lay=DEFAULT):
    m = [0] * (1 << p)
    for rec in records:
        for km in canonical_kmers(rec, k, lay):
            x = xxh3_64_8b(km, seed)
            if lay.bucket_high:
                j, w = x >> (64 - p), x & ((1 << (64 - p)) - 1)
            else:
                j, w = x & ((1 << p) - 1), x >> p
            rho = (64 - p) - w.bit_length() + 1
            m[j] = max(m[j], rho)
    zero = sum(1 for r in m if r == 0)
    tot = sum(2.0 ** (-r) for r in m)
    return lay.header("hll", p, 1 << p, zero, tot) + bytes(m)


def ull_unpack(r):
    return 0 if r == 0 else (4 | (r & 3)) << ((r >> 2) - 2)


def ull_pack(x):
    top = x.bit_length() - 1
    return (top << 2) | ((x >> (top - 2)) & 3 if top >= 2 else (x << (2 - top)) & 3)


def ull_sketch(records, k, p, seed, lay=DEFAULT):
    st = [0] * (1 << p)
    for rec in records:
        for km in canonical_kmers(rec, k, lay):
            h = xxh3_64_8b(km, seed)
            idx = h >> (64 - p)
            t = (~((~h & M64) << p)) & M64
            nlz = clz64(t)
            st[idx] = ull_pack(ull_unpack(st[idx]) | (1 << (nlz + p - 1)))
    return lay.header("ull", p, 1 << p) + bytes(st)


def sketch_from_masked_kmers(algo, p, seed, values, lay=DEFAULT):
    """the three add_kmer rules (utils.rs:395-434) on a list of masked k-mer values, however they were made (tests/test_amino.py)"""
    class _L(list):
        pass
    if algo == "hmh":
        regs = [0] * 16384
        for km in values:
            lo, hi = xxh3_128_4b(km & 0xFFFFFFFF, seed)
            x, y = (lo, hi) if lay.x_low else (hi, lo)
            reg = ((clz64(((x << 14) & M64) ^ 0x3FFF) + 1) << 10) | (y & 0x3FF)
            if regs[x >> 50] < reg:
                regs[x >> 50] = reg
        return lay.header("hmh", 14, 16384) + b"".join(struct.pack(">H" if lay.reg_be else "<H", r) for r in regs)
    if algo == "hll":
        m = [0] * (1 << p)
        for km in values:
            x = xxh3_64_8b(km, seed)
            j, w = (x >> (64 - p), x & ((1 << (64 - p)) - 1)) if lay.bucket_high else (x & ((1 << p) - 1), x >> p)
            m[j] = max(m[j], (64 - p) - w.bit_length() + 1)
        return lay.header("hll", p, 1 << p, sum(1 for r in m if r == 0), sum(2.0 ** (-r) for r in m)) + bytes(m)
    st = [0] * (1 << p)
    for km in values:
        h = xxh3_64_8b(km, seed)
        t = (~((~h & M64) << p)) & M64
        st[h >> (64 - p)] = ull_pack(ull_unpack(st[h >> (64 - p)]) | (1 << (clz64(t) + p - 1)))
    return lay.header("ull", p, 1 << p) + bytes(st)


# ---- dist side: hyperminhash similarity as published by axiomhq/hyperminhash (the crate hyperminhash 0.1.4 ports it)
# and the Mash-style distance of main.rs:415-423.  [PARITY UNPINNED like every crate-internal rule.]
import math


def hmh_regs(image: bytes):
    return struct.unpack("<16384H", image)


def hmh_cardinality(image: bytes) -> float:
    s, ez = 0.0, 0.0
    for r in hmh_regs(image):
        lz = r >> 10
        if lz == 0:
            ez += 1.0
        s += 1.0 / math.pow(2.0, lz)
    m = 16384.0
    alpha = 0.7213 / (1.0 + 1.079 / m)
    zl = math.log(ez + 1.0)
    beta = (-0.370393911 * ez + 0.070471823 * zl + 0.17393686 * zl ** 2 + 0.16339839 * zl ** 3 - 0.09237745 * zl ** 4
            + 0.03738027 * zl ** 5 - 0.005384159 * zl ** 6 + 0.00042419 * zl ** 7)
    return alpha * m * (m - ez) / (beta + s)


def hmh_expected_collisions(n: float, m: float) -> float:
    p, q, r = 14, 6, 10
    if n < m:
        n, m = m, n
    if n > 2.0 ** (2 ** q + r):
        return 1.8446744073709552e19
    if n > 2.0 ** (p + 5):
        d = (4.0 * n / m) / ((1.0 + n) / m) ** 2
        return 0.169919487159739093975315012348 * 2.0 ** (p - r) * d + 0.5
    x = 0.0
    for i in range(1, 65):
        for j in range(1, 1025):
            if i != 64:
                den = 2.0 ** (p + r + i)
                b1, b2 = (1024 + j) / den, (1024 + j + 1) / den
            else:
                den = 2.0 ** (p + r + i - 1)
                b1, b2 = j / den, (j + 1) / den
            x += (math.pow(1 - b2, n) - math.pow(1 - b1, n)) * (math.pow(1 - b2, m) - math.pow(1 - b1, m))
    return (x * p + 0.5) / p


def hmh_similarity(a: bytes, b: bytes) -> float:
    ra, rb = hmh_regs(a), hmh_regs(b)
    c = sum(1 for x, y in zip(ra, rb) if x != 0 and x == y)
    n = sum(1 for x, y in zip(ra, rb) if x != 0 or y != 0)
    if c == 0:
        return 0.0
    ec = hmh_expected_collisions(hmh_cardinality(a), hmh_cardinality(b))
    return 0.0 if c < ec else (c - ec) / n


def mash_distance(sim: float, k: int, model: int, same_name: bool) -> float:
    sim = max(sim, 0.0)
    frac = 2.0 * sim / (1.0 + sim)
    if same_name:
        return 0.0
    if model == 1:
        return 1.0 if frac == 0.0 else min(-math.log(frac) / k, 1.0)
    return 1.0 - frac ** (1.0 / k)


# ---- HyperLogLog dist side (utils.rs:290-373; streaming_algorithms' HLL++ len(), regimes without the bias tables) ----
HLL_THRESHOLD = [10, 20, 40, 80, 220, 400, 900, 1800, 3100, 6500, 11500, 20000, 50000, 120000, 350000]   # p = 4..18


def hll_alpha(p: int) -> float:
    return {4: 0.673, 5: 0.697, 6: 0.709}.get(p, 0.7213 / (1.0 + 1.079 / (1 << p)))


def hll_estimate_bias(tables, p: int, e: float):
    """streaming_algorithms' estimate_bias: mean bias of the 6 samples of the HLL++ table for p whose raw estimate is
    nearest to e (squared distance; equal distances in table order).  tables = {p: (raw list, bias list)}."""
    if not tables or p not in tables:
        return None
    raw, bias = tables[p]
    near = sorted(range(len(raw)), key=lambda i: ((e - raw[i]) * (e - raw[i]), i))[:6]
    s = 0.0
    for i in near:
        s += bias[i]
    return s / 6.0


def hll_len_from_regs(p: int, regs, tables=None) -> float:
    """None when the estimate falls into the bias-corrected regime and `tables` has nothing for p."""
    m = float(1 << p)
    zero = sum(1 for r in regs if r == 0)
    if zero:
        h = m * math.log(m / zero)
        if h <= HLL_THRESHOLD[p - 4]:
            return h
    e = hll_alpha(p) * m * m / math.fsum(2.0 ** -int(r) for r in regs)
    if e <= 5.0 * m:
        b = hll_estimate_bias(tables, p, e)
        return None if b is None else e - b
    return e


def hll_similarity(p: int, a: bytes, b: bytes, tables=None) -> float:
    ra, rb = a[33:], b[33:]
    la, lb = hll_len_from_regs(p, ra, tables), hll_len_from_regs(p, rb, tables)
    u = hll_len_from_regs(p, [max(x, y) for x, y in zip(ra, rb)], tables)
    return max((la + lb - u) / u, 0.0)


# ---- UltraLogLog dist side (utils.rs:186-288): FGRA and ML estimators of crate ultraloglog 0.1.6 (hash4j), restated per
# REGISTER (the product works on histograms) from the model in Ertl's UltraLogLog paper.  [PARITY UNPINNED]
ULL_ETA = (4.663135422063788, 2.1378502137958524, 2.781144650979996, 0.9824082545153715)
ULL_TAU = 0.8194911375910897
ULL_V = 0.6118931496978437


def _ull_state(r, p):
    """register -> (largest update value u, seen(u-1), seen(u-2)); None for an empty register"""
    if r == 0:
        return None
    u = (r >> 2) - p + 2
    return u, (r >> 1) & 1, r & 1


def _ull_expected_eta(q1, q2):
    """E[eta_bits] when the bit for u-1 is set with probability 1-q1 and the bit for u-2 with probability 1-q2"""
    e = ULL_ETA
    return q1 * q2 * e[0] + q1 * (1 - q2) * e[1] + (1 - q1) * q2 * e[2] + (1 - q1) * (1 - q2) * e[3]


def ull_fgra(regs, p):
    m = 1 << p
    K = 65 - p
    if any(_ull_state(r, p) and _ull_state(r, p)[0] >= K for r in regs):
        raise NotImplementedError("saturated registers: not reachable in tests")
    c0 = sum(1 for r in regs if r == 0)
    c4 = sum(1 for r in regs if r == 4 * p - 4)
    c8 = sum(1 for r in regs if r == 4 * p)
    c10 = sum(1 for r in regs if r == 4 * p + 2)
    z = None
    if c0 or c4 or c8 or c10:
        al, be, ga = m + 3 * (c0 + c4 + c8 + c10), m - c0 - c4, 4 * c0 + 2 * c4 + 3 * c8 + c10
        x = (math.sqrt(be * be + 4 * al * ga) - be) / (2 * al)
        z = x ** 4                                              # e^(-n/m)
    total = 0.0
    for r in regs:
        st = _ull_state(r, p)
        if st is None:
            # empty: the largest VIRTUAL update value is -j (j >= 0) with probability z^(2^j - 1) (1 - z^(2^j))
            if z >= 1.0:
                return 0.0
            s, j = 0.0, 0
            while True:
                zj = z ** (2 ** j)
                term = (2.0 ** (ULL_TAU * j)) * (z ** (2 ** j - 1)) * (1 - zj) * _ull_expected_eta(zj ** 2, zj ** 4)
                s += term
                if term < 1e-18 * max(s, 1e-300) or zj == 0.0:
                    break
                j += 1
            total += s
            continue
        u, b1, b2 = st
        w = 2.0 ** (-ULL_TAU * u)
        if u >= 3:
            total += w * ULL_ETA[(b1 << 1) | b2]
        elif u == 2:                                            # bit for u-2 = virtual value 0: unseen with probability z
            total += w * (z * ULL_ETA[b1 << 1] + (1 - z) * ULL_ETA[(b1 << 1) | 1])
        else:                                                   # u == 1: both lower bits virtual (values 0 and -1)
            total += w * _ull_expected_eta(z, z * z)
    factor = m ** (1 + 1 / ULL_TAU) / (1 + ULL_V * (1 + ULL_TAU) / (2 * m))
    return factor * total ** (-1 / ULL_TAU)


def ull_ml(regs, p):
    """maximum likelihood under the Poisson model, by bisection on the log-likelihood's derivative (the product uses Ertl's
    secant iteration to a relative tolerance of 7.6e-4 / sqrt(m)), then the first-order bias correction"""
    m = 1 << p
    K = 65 - p
    a = 0.0                                                     # coefficient of -lambda in the log-likelihood
    b = {}                                                      # update value k -> how many (1 - e^(-lambda 2^-k')) factors, k' = min(k, K-1)
    for r in regs:
        st = _ull_state(r, p)
        if st is None:
            a += 1.0
            continue
        u, b1, b2 = st
        a += 2.0 ** -u if u < K else 0.0
        for k, seen in ((u, 1), (u - 1, b1), (u - 2, b2)):
            if k < 1:
                continue
            rate = 2.0 ** -min(k, K - 1)
            if seen:
                b[rate] = b.get(rate, 0) + 1
            else:
                a += rate
    if not b:
        return 0.0
    if a == 0.0:
        return float("inf")

    def dlog(lam):                                              # d/d lambda of the log-likelihood
        return -a + sum(cnt * rate / math.expm1(lam * rate) for rate, cnt in b.items())
    lo, hi = 1e-12, 1.0
    while dlog(hi) > 0:
        hi *= 2
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if dlog(mid) > 0:
            lo = mid
        else:
            hi = mid
    return m * 0.5 * (lo + hi) / (1 + 0.48147376527720065 / m)


def ull_merge(a, b):
    return bytes(ull_pack(ull_unpack(x) | ull_unpack(y)) if x and y else (x or y) for x, y in zip(a, b))


# ---- hashbrown 0.15 RawTable, written out control byte by control byte (x86-64: 16-wide groups) -------------------
# The host code (lash_amd/csrc/host/name_order.cpp) uses a simplified circular scan; this restatement keeps the
# control array with its EMPTY padding and mirrored tail, group loads, fix_insert_slot and the resize policy as the
# crate has them, so that the two formulations check each other.  /root/reference/src/utils.rs:111-127 builds
# HashMap<&String, &Sketch, Xxh3Builder{seed: 93}>; Cargo.lock pins hashbrown 0.15.4.
class HashbrownOrder:
    WIDTH = 16
    EMPTY = 0xFF

    def __init__(self, hash_fn):
        self.hash_fn = hash_fn
        self.mask = 0                    # the static empty singleton: one bucket, no capacity
        self.ctrl = [self.EMPTY] * (1 + self.WIDTH)
        self.keys = [None]
        self.vals = [None]
        self.items = 0
        self.growth_left = 0

    @staticmethod
    def _cap(mask):                      # bucket_mask_to_capacity
        return mask if mask < 8 else (mask + 1) // 8 * 7

    @staticmethod
    def _buckets(cap):                   # capacity_to_buckets for 16-byte entries
        if cap < 15:
            cap = max(cap, 3)
            return 4 if cap < 4 else 8 if cap < 8 else 16
        adj = cap * 8 // 7
        b = 1
        while b < adj:
            b <<= 1
        return b

    def _group(self, pos):
        return self.ctrl[pos:pos + self.WIDTH]

    def _set_ctrl(self, i, c):
        i2 = ((i - self.WIDTH) & self.mask) + self.WIDTH
        self.ctrl[i] = c
        self.ctrl[i2] = c

    def _fix(self, idx):                 # fix_insert_slot
        if self.ctrl[idx] & 0x80 == 0:   # bucket is full: only possible when the table is smaller than a group
            assert self.mask < self.WIDTH
            g = self._group(0)
            return next(b for b in range(self.WIDTH) if g[b] & 0x80)
        return idx

    def _probe(self, h, key):
        """find_or_find_insert_slot_inner -> ('found', idx) or ('slot', idx)"""
        tag = (h >> 57) & 0x7F
        pos, stride, slot = h & self.mask, 0, None
        while True:
            g = self._group(pos)
            for b in range(self.WIDTH):
                if g[b] == tag:
                    idx = (pos + b) & self.mask
                    if self.keys[idx] == key:
                        return "found", idx
            if slot is None:
                for b in range(self.WIDTH):
                    if g[b] & 0x80:
                        slot = (pos + b) & self.mask
                        break
            if slot is not None and any(c == self.EMPTY for c in g):
                return "slot", self._fix(slot)
            stride += self.WIDTH
            pos = (pos + stride) & self.mask

    def _find_insert_slot(self, h):
        pos, stride = h & self.mask, 0
        while True:
            g = self._group(pos)
            for b in range(self.WIDTH):
                if g[b] & 0x80:
                    return self._fix((pos + b) & self.mask)
            stride += self.WIDTH
            pos = (pos + stride) & self.mask

    def _resize(self, capacity):
        old = [(self.keys[i], self.vals[i]) for i in range(self.mask + 1) if self.ctrl[i] & 0x80 == 0]
        buckets = self._buckets(capacity)
        self.mask = buckets - 1
        self.ctrl = [self.EMPTY] * (buckets + self.WIDTH)
        self.keys = [None] * buckets
        self.vals = [None] * buckets
        self.growth_left = self._cap(self.mask) - self.items
        for k, v in old:                 # full buckets in index order
            h = self.hash_fn(k)
            idx = self._find_insert_slot(h)
            self._set_ctrl(idx, (h >> 57) & 0x7F)
            self.keys[idx], self.vals[idx] = k, v

    def insert(self, key, val):
        h = self.hash_fn(key)
        if self.growth_left < 1:         # reserve(1) comes before the lookup
            self._resize(max(self.items + 1, self._cap(self.mask) + 1))
        kind, idx = self._probe(h, key)
        if kind == "found":
            self.vals[idx] = val
            return
        self.growth_left -= 1            # the slot was EMPTY (nothing is ever deleted)
        self._set_ctrl(idx, (h >> 57) & 0x7F)
        self.keys[idx], self.vals[idx] = key, val
        self.items += 1

    def values_in_key_order(self):
        return [self.vals[i] for i in range(self.mask + 1) if self.ctrl[i] & 0x80 == 0]


def hashbrown_name_order(names, seed=93, hash_fn=None):
    """indices of `names` in `.keys()` order of the reference's seeded map (Hash for str = bytes + 0xFF)"""
    if hash_fn is None:
        import xxhash
        hash_fn = lambda s: xxhash.xxh3_64_intdigest(s.encode() + b"\xff", seed=seed)
    t = HashbrownOrder(hash_fn)
    for i, n in enumerate(names):
        t.insert(n, i)
    return t.values_in_key_order()
