/*
 * lash_oracle.c — CPU ORACLE (scalar restatement) of lash's sketching hot path.
 * TEST INFRASTRUCTURE ONLY: see lash_oracle.h for who may use it and for the pinning status
 * ("parity unpinned" for the crate-internal rules; XXH3 layer pinned by tests/golden/).
 *
 * Reference map (all paths under /root/reference/src):
 *   filter_out_n ............ utils.rs:33-41
 *   mask_bits ............... utils.rs:57-64
 *   per-record k-mer loop ... utils.rs:457-505   (three k regimes: <=14, ==16, 15 & 17..=32)
 *   add_kmer (HMH/HLL/ULL) .. utils.rs:395-398, 411-413, 427-429
 *   new / save .............. utils.rs:391-393,400-402 / 407-409,415-417 / 422-425,431-433
 *   one sketch per file ..... utils.rs:450-509
 *
 * The oracle deliberately keeps the reference's *structure* (copy-filter the record, 2-bit pack it
 * into a second buffer, slide a window, recompute the reverse complement from scratch per k-mer,
 * apply the sequential register rule), so that it is an independent statement of the result and
 * not a CPU twin of the GPU kernels (which use funnel-shift windows and OR/max-merge algebra).
 */
#include "lash_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * XXH3, inputs of 4..8 bytes (XXH 0.8 spec; xxhash-rust 0.8.15 implements the same function).
 * Closed forms per SURVEY.md Appendix C; pinned by tests/golden/xxh3_vectors.json.
 * ---------------------------------------------------------------------------------------- */
#define XXH_PRIME64_1 0x9E3779B185EBCA87ULL
#define XXH_PRIME_MX1 0x165667919E3779F9ULL
#define XXH_PRIME_MX2 0x9FB21C651E98DF25ULL
/* little-endian u64 words of the default kSecret at byte offsets 8, 16, 24 */
#define XXH_SEC8  0x1cad21f72c81017cULL
#define XXH_SEC16 0xdb979083e96dd4deULL
#define XXH_SEC24 0x1f67b3b7a4a44072ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

static inline uint64_t xxh3_short_seed(uint64_t seed)
{
    /* seed ^= (u64)swap32((u32)seed) << 32  (XXH3_len_4to8_64b / _128b) */
    return seed ^ ((uint64_t)bswap32((uint32_t)seed) << 32);
}

uint64_t lash_or_xxh3_64_8b(uint64_t v, uint64_t seed)
{
    /* XXH3_len_4to8_64b with len == 8: input1 = first 4 bytes, input2 = last 4 bytes */
    uint64_t s = xxh3_short_seed(seed);
    uint32_t in1 = (uint32_t)v, in2 = (uint32_t)(v >> 32);
    uint64_t bitflip = (XXH_SEC8 ^ XXH_SEC16) - s;
    uint64_t in64 = (uint64_t)in2 + ((uint64_t)in1 << 32);
    uint64_t h = in64 ^ bitflip;
    /* XXH3_rrmxmx(h, len = 8) */
    h ^= rotl64(h, 49) ^ rotl64(h, 24);
    h *= XXH_PRIME_MX2;
    h ^= (h >> 35) + 8;
    h *= XXH_PRIME_MX2;
    return h ^ (h >> 28);
}

void lash_or_xxh3_128_4b(uint32_t w, uint64_t seed, uint64_t *lo, uint64_t *hi)
{
    /* XXH3_len_4to8_128b with len == 4: input_lo == input_hi == w */
    uint64_t s = xxh3_short_seed(seed);
    uint64_t in64 = (uint64_t)w + ((uint64_t)w << 32);
    uint64_t bitflip = (XXH_SEC16 ^ XXH_SEC24) + s;
    uint64_t keyed = in64 ^ bitflip;
    unsigned __int128 m = (unsigned __int128)keyed * (XXH_PRIME64_1 + (4u << 2));
    uint64_t l = (uint64_t)m, h = (uint64_t)(m >> 64);
    h += l << 1;
    l ^= h >> 3;
    l ^= l >> 35;            /* XXH_xorshift64(l, 35) */
    l *= XXH_PRIME_MX2;
    l ^= l >> 28;            /* XXH_xorshift64(l, 28) */
    h ^= h >> 37;            /* XXH3_avalanche(h) */
    h *= XXH_PRIME_MX1;
    h ^= h >> 32;
    *lo = l;
    *hi = h;
}

/* ------------------------------------------------------------------------------------------
 * utils.rs:33-41 / 57-64
 * ---------------------------------------------------------------------------------------- */
size_t lash_or_filter_out_n(const uint8_t *seq, size_t n, uint8_t *out)
{
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = seq[i];
        if (c == 'A' || c == 'C' || c == 'T' || c == 'G') out[m++] = c;   /* upper-case only */
    }
    return m;
}

uint64_t lash_or_mask_bits(uint64_t v, int k)
{
    unsigned b = 2u * (unsigned)k;
    if (b == 64) return v;
    return v & ((1ULL << b) - 1);
}

/* ------------------------------------------------------------------------------------------
 * The layout: every unverified crate-internal choice as data (lash_oracle.h)
 * ---------------------------------------------------------------------------------------- */
static const lash_or_layout DEFAULT_LAYOUT = { {0, 1, 2, 3}, 0, 0, 0, 0, "", "azspl", "l", 0, 0, {0} };
static inline const lash_or_layout *lay_of(const lash_or_params *prm) { return prm->layout ? prm->layout : &DEFAULT_LAYOUT; }

void lash_or_layout_default(lash_or_layout *out) { *out = DEFAULT_LAYOUT; }

static size_t field_bytes(char c)
{
    switch (c) {
    case 'a': case 'z': case 's': case 'Q': case 'l': return 8;
    case 'Z': case 'P': case 'L': return 4;
    case 'p': return 1;
    default: return (size_t)-1;
    }
}
static size_t header_len(const char *tpl)
{
    size_t n = 0;
    for (int i = 0; i < 8 && tpl[i]; i++) n += field_bytes(tpl[i]);
    return n;
}
static const char *header_tpl(const lash_or_layout *lay, int algo)
{
    return algo == LASH_OR_HMH ? lay->hmh_header : algo == LASH_OR_HLL ? lay->hll_header : lay->ull_header;
}
size_t lash_or_header_bytes(const lash_or_layout *lay, int algo) { return header_len(header_tpl(lay ? lay : &DEFAULT_LAYOUT, algo)); }

int lash_or_layout_check(const lash_or_layout *lay)
{
    unsigned seen = 0;
    for (int i = 0; i < 4; i++) { if (lay->base_code[i] > 3) return -1; seen |= 1u << lay->base_code[i]; }
    if (seen != 15u) return -1;
    if ((lay->base_code[0] ^ lay->base_code[3]) != (lay->base_code[1] ^ lay->base_code[2])) return -1;   /* always true for a permutation */
    const char *t[3] = { lay->hmh_header, lay->hll_header, lay->ull_header };
    for (int a = 0; a < 3; a++) {
        int i = 0;
        for (; i < 8 && t[a][i]; i++) if (field_bytes(t[a][i]) == (size_t)-1) return -1;
        if (i == 8) return -1;                                           /* must be NUL-terminated inside its 8 bytes */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * kmerutils 0.0.14 restated [UNPINNED, switch U5 = layout.base_code / layout.kmer_lsb_first]: Alphabet2b
 * A=0 C=1 G=2 T=3; Sequence::new(.,2) packs 4 bases per byte; k-mers are built by shift-left-and-OR (first
 * base most significant); reverse_complement = reverse the 2-bit groups, complement, right-align to 2k bits;
 * Ord compares the packed value.
 * ---------------------------------------------------------------------------------------- */
static inline unsigned base_code(const lash_or_layout *lay, uint8_t c)
{
    switch (c) { case 'A': return lay->base_code[0]; case 'C': return lay->base_code[1]; case 'G': return lay->base_code[2];
    default: return lay->base_code[3]; /* 'T' */ }
}

typedef struct { uint8_t *bytes; size_t n_bases; } kseq_t;   /* KSeq::new(&seq, 2) */

static void kseq_pack(const lash_or_layout *lay, kseq_t *ks, const uint8_t *filtered, size_t n)
{
    ks->n_bases = n;
    ks->bytes = (uint8_t *)calloc((n + 3) / 4 + 1, 1);
    for (size_t i = 0; i < n; i++)
        ks->bytes[i >> 2] |= (uint8_t)(base_code(lay, filtered[i]) << (6 - 2 * (i & 3)));
}
static inline unsigned kseq_get(const kseq_t *ks, size_t i)
{
    return (ks->bytes[i >> 2] >> (6 - 2 * (i & 3))) & 3u;
}

/* order of the k 2-bit groups reversed, nothing complemented: plain loop on purpose (the GPU uses bit tricks) */
static inline uint64_t groups_reversed(uint64_t v, int k)
{
    uint64_t o = 0;
    for (int i = 0; i < k; i++) { o = (o << 2) | (v & 3u); v >>= 2; }
    return o;
}
/* complement every base (code -> code of the complementary letter: A<->T, C<->G), reverse their order.
 * kmerutils does this with swaps on the whole word (complement, swap groups inside nibbles, nibbles inside bytes, bytes,
 * then right-align): restated for the default codes, where complement is bitwise NOT; other code assignments (layout
 * switch U5) take the plain loop. */
static inline uint64_t revcomp(const lash_or_layout *lay, uint64_t v, int k)
{
    const unsigned cx = (unsigned)(lay->base_code[0] ^ lay->base_code[3]);      /* == code[C] ^ code[G] */
    if (cx == 3u) {
        uint64_t x = ~v;
        x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
        x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
        x = __builtin_bswap64(x);
        return x >> (64 - 2 * k);
    }
    uint64_t o = 0;
    for (int i = 0; i < k; i++) { o = (o << 2) | ((v & 3u) ^ cx); v >>= 2; }
    return o;
}

typedef void (*kmer_sink)(void *ctx, uint64_t masked);

/* The `while let Some(km) = it.next()` loops, utils.rs:469-476 / 481-488 / 493-498.
 * The three container types differ only in width; min() and mask_bits() see the same numbers
 * (Kmer32bit's 4-bit length tag is equal on both sides of min() and removed by mask_bits). */
static uint64_t iterate_kmers(const lash_or_layout *lay, const kseq_t *ks, int k, kmer_sink sink, void *ctx, uint64_t *out)
{
    size_t L = ks->n_bases;
    if (L < (size_t)k) return 0;
    uint64_t kmask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t fwd = 0, count = 0;
    for (size_t i = 0; i < L; i++) {
        fwd = ((fwd << 2) | kseq_get(ks, i)) & kmask;           /* window with the first base most significant */
        if (i + 1 < (size_t)k) continue;
        uint64_t km = lay->kmer_lsb_first ? groups_reversed(fwd, k) : fwd;   /* the iterator's value (switch U5) */
        uint64_t rc = revcomp(lay, km, k);                       /* km.reverse_complement() */
        uint64_t canon = km < rc ? km : rc;                      /* km.min(rc) */
        uint64_t masked;
        if (k <= 14 || k == 16)                                  /* u32 containers (utils.rs:471-474,483-486) */
            masked = lash_or_mask_bits((uint64_t)(uint32_t)canon, k);
        else                                                     /* Kmer64bit (utils.rs:495-496) */
            masked = lash_or_mask_bits(canon, k);
        if (sink) sink(ctx, masked);
        if (out) out[count] = masked;
        count++;
    }
    return count;
}

uint64_t lash_or_record_kmers(const uint8_t *seq, size_t n, int k, uint64_t *out)
{
    if (k < 1 || k > 32) return 0;
    uint8_t *filt = (uint8_t *)malloc(n + 1);
    size_t m = lash_or_filter_out_n(seq, n, filt);                /* utils.rs:459 */
    uint64_t cnt = 0;
    if (m >= (size_t)k) {                                        /* utils.rs:460-462 */
        kseq_t ks;
        kseq_pack(&DEFAULT_LAYOUT, &ks, filt, m);                /* utils.rs:464 */
        cnt = iterate_kmers(&DEFAULT_LAYOUT, &ks, k, NULL, NULL, out);
        free(ks.bytes);
    }
    free(filt);
    return cnt;
}

/* ------------------------------------------------------------------------------------------
 * Sketch states and the three add_kmer rules.
 * ---------------------------------------------------------------------------------------- */
#define HMH_P 14
#define HMH_M (1u << HMH_P)
#define HMH_R 10

typedef struct {
    int algo, p;
    uint64_t seed;
    int hmh_x_is_low;
    const lash_or_layout *lay;
    uint16_t *hmh;     /* hyperminhash::Sketch: 16384 x u16            [UNPINNED, A.2] */
    uint8_t *reg;      /* HLL m[] / ULL state[]: 2^p x u8              [UNPINNED, A.3/A.4] */
    uint64_t hll_zero; /* streaming_algorithms HyperLogLog.zero        */
    double hll_sum;    /* streaming_algorithms HyperLogLog.sum         */
} sketch_t;

/* hyperminhash add_hash(x, y)  [UNPINNED, A.2; axiomhq/hyperminhash AddHash] */
static inline void hmh_add_hash(uint16_t *regs, uint64_t x, uint64_t y)
{
    uint32_t bucket = (uint32_t)(x >> (64 - HMH_P));
    uint32_t lz = (uint32_t)__builtin_clzll((x << HMH_P) ^ 0x3FFFULL) + 1;   /* 1..=51 */
    uint16_t sig = (uint16_t)(y & ((1u << HMH_R) - 1));
    uint16_t reg = (uint16_t)((lz << HMH_R) | sig);
    if (regs[bucket] < reg) regs[bucket] = reg;
}

/* utils.rs:395-398 -> Sketch::add_bytes_with_seed(&(masked as u32).to_le_bytes(), seed) */
static void hmh_add_kmer(void *ctx, uint64_t masked)
{
    sketch_t *s = (sketch_t *)ctx;
    uint64_t lo, hi;
    lash_or_xxh3_128_4b((uint32_t)masked, s->seed, &lo, &hi);
    if (s->hmh_x_is_low) hmh_add_hash(s->hmh, lo, hi);           /* switch U1 */
    else                 hmh_add_hash(s->hmh, hi, lo);
}

static inline double pow2_neg(unsigned e)
{
    /* 2^-e exactly (the crate's f64 bit-hack produces the same value) */
    uint64_t bits = (uint64_t)(1023 - e) << 52;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

/* utils.rs:411-413 -> push_hash64(xxh3_64_with_seed(&masked.to_le_bytes(), seed))
 * [UNPINNED, A.3: bucket = LOW p bits, rho from the remaining 64-p bits; zero/sum kept incrementally] */
static void hll_add_kmer(void *ctx, uint64_t masked)
{
    sketch_t *s = (sketch_t *)ctx;
    uint64_t x = lash_or_xxh3_64_8b(masked, s->seed);
    uint64_t j, w;
    if (!s->lay->hll_bucket_high) { j = x & ((1ULL << s->p) - 1); w = x >> s->p; }            /* upstream `push` */
    else { j = x >> (64 - s->p); w = x & (~0ULL >> s->p); }                                     /* switch U3, alternative */
    unsigned bitlen = w ? 64u - (unsigned)__builtin_clzll(w) : 0u;
    uint8_t rho = (uint8_t)((64 - s->p) - bitlen + 1);           /* 1..=65-p */
    uint8_t old = s->reg[j];
    uint8_t neu = old > rho ? old : rho;
    s->hll_zero -= (old == 0);
    /* upstream `push` as recalled (SURVEY App. A.3 is RECALLED-UNVERIFIED either way): ONE update per k-mer,
     *     self.sum -= f64::from_bits(..old..) - f64::from_bits(..new..);
     * i.e. the difference of the two powers first (exact unless new - old > 53), then one rounding into `sum`.  Equal to the
     * two-step form (sum -= 2^-old; sum += 2^-new) whenever 2^-old lies on sum's grid — every update outside the `sum` corner, and in
     * the corner unless the bucket's OLD value is above 53 - p as well. */
    s->hll_sum -= pow2_neg(old) - pow2_neg(neu);
    s->reg[j] = neu;
}

/* hash4j UltraLogLog pack/unpack, as ported by crate ultraloglog [UNPINNED, A.4] */
static inline uint64_t ull_unpack(uint8_t r)
{
    if (r < 8) return 0;   /* only r == 0 occurs below 4(p-1) >= 8; Java's mod-64 shift gives 0 for r == 0 */
    return (4ULL | (r & 3u)) << ((r >> 2) - 2);
}
static inline uint8_t ull_pack(uint64_t x)
{
    unsigned nlz1 = (unsigned)__builtin_clzll(x) + 1;            /* x != 0 */
    uint64_t below = (nlz1 >= 64) ? 0 : (x << nlz1) >> 62;
    return (uint8_t)((((64u - nlz1) & 63u) << 2) | (unsigned)below);
}

/* utils.rs:427-429 -> UltraLogLog::add(xxh3_64_with_seed(&masked.to_le_bytes(), seed)) */
static void ull_add_kmer(void *ctx, uint64_t masked)
{
    sketch_t *s = (sketch_t *)ctx;
    uint64_t h = lash_or_xxh3_64_8b(masked, s->seed);
    int p = s->p, q = 64 - p;
    uint64_t idx = h >> q;
    uint64_t t = ~(~h << p);                                      /* low p bits all ones => nlz <= q */
    unsigned nlz = (unsigned)__builtin_clzll(t);
    uint64_t prefix = ull_unpack(s->reg[idx]);
    prefix |= 1ULL << (nlz + (unsigned)p - 1);
    s->reg[idx] = ull_pack(prefix);
}

/* ------------------------------------------------------------------------------------------
 * new() and save()  (byte images).  Each layout is isolated in one function (switches U2-U4).
 * ---------------------------------------------------------------------------------------- */
static int check_params(int algo, int k, int p)
{
    if (k < 1 || k > 32) return -1;                              /* utils.rs:500-502 */
    if (algo == LASH_OR_HMH) return 0;                           /* precision ignored, main.rs:212-213 */
    if (algo == LASH_OR_HLL) return (p >= 4 && p <= 16) ? 0 : -1;  /* get_alpha assert [UNPINNED] */
    if (algo == LASH_OR_ULL) return (p >= 3 && p <= 26) ? 0 : -1;  /* UltraLogLog::new range [UNPINNED] */
    return -1;                                                   /* main.rs:245 */
}

size_t lash_or_image_bytes_layout(const lash_or_layout *lay, int algo, int p)
{
    if (!lay) lay = &DEFAULT_LAYOUT;
    const size_t hdr = header_len(header_tpl(lay, algo));
    switch (algo) {
    case LASH_OR_HMH: return hdr + (size_t)HMH_M * 2;             /* U2: header + 16384 x u16 */
    case LASH_OR_HLL: return hdr + ((size_t)1 << p);              /* U3: bincode(alpha,zero,sum,p,len) + m */
    case LASH_OR_ULL: return hdr + ((size_t)1 << p);              /* U4: bincode(Vec<u8>) = u64 len + bytes */
    default: return 0;
    }
}
size_t lash_or_image_bytes(int algo, int p) { return lash_or_image_bytes_layout(&DEFAULT_LAYOUT, algo, p); }

static void put_u32(uint8_t *dst, uint32_t v) { for (int i = 0; i < 4; i++) dst[i] = (uint8_t)(v >> (8 * i)); }
static void put_u64(uint8_t *dst, uint64_t v) { for (int i = 0; i < 8; i++) dst[i] = (uint8_t)(v >> (8 * i)); }
static void put_f64(uint8_t *dst, double d) { uint64_t b; memcpy(&b, &d, 8); put_u64(dst, b); }

static double hll_alpha(int p)
{
    switch (p) { case 4: return 0.673; case 5: return 0.697; case 6: return 0.709;
    default: return 0.7213 / (1.0 + 1.079 / (double)(1u << p)); }
}

/* writes the header the template describes, returns its length (switches U2-U4) */
static size_t write_header(const char *tpl, uint8_t *image, int p, uint64_t n_regs, uint64_t zero, double sum)
{
    size_t at = 0;
    for (int i = 0; i < 8 && tpl[i]; i++) {
        switch (tpl[i]) {
        case 'a': put_f64(image + at, hll_alpha(p)); break;
        case 'z': put_u64(image + at, zero); break;
        case 'Z': put_u32(image + at, (uint32_t)zero); break;
        case 's': put_f64(image + at, sum); break;
        case 'p': image[at] = (uint8_t)p; break;
        case 'P': put_u32(image + at, (uint32_t)p); break;
        case 'Q': put_u64(image + at, (uint64_t)p); break;
        case 'l': put_u64(image + at, n_regs); break;
        case 'L': put_u32(image + at, (uint32_t)n_regs); break;
        }
        at += field_bytes(tpl[i]);
    }
    return at;
}

static void hmh_save(const lash_or_layout *lay, const uint16_t *regs, uint8_t *image)       /* U2 */
{
    image += write_header(lay->hmh_header, image, HMH_P, HMH_M, 0, 0.0);
    for (uint32_t i = 0; i < HMH_M; i++) {
        const uint8_t lo = (uint8_t)regs[i], hi = (uint8_t)(regs[i] >> 8);
        image[2 * i] = lay->hmh_reg_be ? hi : lo;
        image[2 * i + 1] = lay->hmh_reg_be ? lo : hi;
    }
}
static void hll_save(const lash_or_layout *lay, int p, const uint8_t *m, uint64_t zero, double sum, uint8_t *image)   /* U3 */
{
    size_t n = (size_t)1 << p;
    image += write_header(lay->hll_header, image, p, n, zero, sum);
    memcpy(image, m, n);
}
static void ull_save(const lash_or_layout *lay, int p, const uint8_t *state, uint8_t *image)   /* U4 */
{
    size_t n = (size_t)1 << p;
    image += write_header(lay->ull_header, image, p, n, 0, 0.0);
    memcpy(image, state, n);
}

static int sketch_new(sketch_t *s, const lash_or_params *prm)
{
    memset(s, 0, sizeof *s);
    s->algo = prm->algo; s->p = prm->p; s->seed = prm->seed;
    s->lay = lay_of(prm);
    s->hmh_x_is_low = prm->hmh_x_is_low || s->lay->hmh_x_low;
    if (prm->algo == LASH_OR_HMH) {
        s->hmh = (uint16_t *)calloc(HMH_M, 2);                    /* Sketch::default() */
    } else {
        size_t n = (size_t)1 << prm->p;
        s->reg = (uint8_t *)calloc(n, 1);
        s->hll_zero = n;                                          /* with_p: zero = m, sum = m */
        s->hll_sum = (double)n;
    }
    return 0;
}
static void sketch_free(sketch_t *s) { free(s->hmh); free(s->reg); }

/* Per-thread buffers for the two per-record copies (utils.rs:459 Vec, :464 KSeq).  The reference allocates them
 * per record; the oracle keeps one growing pair per worker thread so that the CPU baseline measures the algorithm
 * and not mmap/page-fault contention of the allocator on many-core hosts. */
typedef struct { uint8_t *filt; size_t filt_cap; uint8_t *packed; size_t packed_cap; } scratch_t;

static void scratch_reserve(scratch_t *sc, size_t n)
{
    if (sc->filt_cap < n + 1) { free(sc->filt); sc->filt_cap = n + n / 4 + 64; sc->filt = (uint8_t *)malloc(sc->filt_cap); }
    size_t pb = (n + 3) / 4 + 1;
    if (sc->packed_cap < pb) { free(sc->packed); sc->packed_cap = pb + pb / 4 + 64; sc->packed = (uint8_t *)malloc(sc->packed_cap); }
}
static void scratch_free(scratch_t *sc) { free(sc->filt); free(sc->packed); memset(sc, 0, sizeof *sc); }

static int sketch_genome_with(const lash_or_params *prm, const uint8_t *seq, const uint64_t *rec_off, uint64_t n_rec,
                              uint8_t *image, scratch_t *sc);

/* ------------------------------------------------------------------------------------------
 * The amino-acid branch (utils.rs:511-563; unreachable in the reference: main.rs:198 hard-wires aa = false, the --aa flag is
 * commented out at main.rs:97-104).  Per record: seq().to_ascii_uppercase() (:521); records whose RAW length is below k are
 * skipped (:523-525, BEFORE the filter, unlike the nucleotide branch); filter_out_a keeps the 20 residue letters
 * (utils.rs:43-55); kmerutils' SequenceAA / KmerAA32bit (k <= 6) / KmerAA64bit (k <= 12) [UNPINNED: 5 bits per residue,
 * codes per layout.aa_code_zero_based, k-mer built by shift-left-and-OR so the first residue is most significant, no
 * k-mer when fewer than k residues survive]; mask_aa_bits (utils.rs:66-76); add_kmer as for nucleotides.
 * ---------------------------------------------------------------------------------------- */
static int aa_code(const lash_or_layout *lay, uint8_t c)           /* -1: deleted by filter_out_a */
{
    static const char letters[] = "ACDEFGHIKLMNPQRSTVWY";
    const char *at = c ? strchr(letters, (int)c) : NULL;
    if (!at) return -1;
    return (int)(at - letters) + (lay->aa_code_zero_based ? 0 : 1);
}
static uint64_t mask_aa_bits(uint64_t v, int k)                    /* utils.rs:66-76 */
{
    const unsigned b = 5u * (unsigned)k;
    if (b == 0) return 0;
    if (b >= 64) return v;
    return v & ((1ull << b) - 1ull);
}
static void sketch_record_aa(const lash_or_params *prm, const lash_or_layout *lay, const uint8_t *rec, size_t n, kmer_sink sink, void *ctx)
{
    const int k = prm->k;
    if (n < (size_t)k) return;                                      /* utils.rs:523-525: the RAW length */
    uint64_t v = 0;
    size_t have = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = rec[i];
        if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);            /* to_ascii_uppercase */
        const int code = aa_code(lay, c);
        if (code < 0) continue;                                     /* filter_out_a */
        v = (v << 5) | (uint64_t)code;                              /* (bits above 5k fall to mask_aa_bits; k <= 12: 60 bits) */
        if (++have >= (size_t)k) sink(ctx, mask_aa_bits(v, k));
    }
}

int lash_or_sketch_genome(const lash_or_params *prm, const uint8_t *seq,
                          const uint64_t *rec_off, uint64_t n_rec, uint8_t *image)
{
    scratch_t sc = {0};
    int rc = sketch_genome_with(prm, seq, rec_off, n_rec, image, &sc);
    scratch_free(&sc);
    return rc;
}

static int sketch_genome_with(const lash_or_params *prm, const uint8_t *seq, const uint64_t *rec_off, uint64_t n_rec,
                              uint8_t *image, scratch_t *sc)
{
    if (check_params(prm->algo, prm->k, prm->p) || lash_or_layout_check(lay_of(prm))) return -1;
    sketch_t s;
    sketch_new(&s, prm);                                         /* utils.rs:454 */
    kmer_sink sink = prm->algo == LASH_OR_HMH ? hmh_add_kmer : prm->algo == LASH_OR_HLL ? hll_add_kmer : ull_add_kmer;
    if (prm->amino && prm->k > 12) { sketch_free(&s); return -1; }   /* utils.rs:554: panic, k must be 1-12 */
    for (uint64_t r = 0; r < n_rec; r++) {                       /* utils.rs:457 */
        const uint8_t *rec = seq + rec_off[r];
        size_t n = (size_t)(rec_off[r + 1] - rec_off[r]);
        if (prm->amino) { sketch_record_aa(prm, s.lay, rec, n, sink, &s); continue; }
        scratch_reserve(sc, n);
        size_t m = lash_or_filter_out_n(rec, n, sc->filt);        /* utils.rs:459 */
        if (m >= (size_t)prm->k) {                               /* utils.rs:460-462 */
            kseq_t ks;
            ks.n_bases = m;                                      /* utils.rs:464 */
            ks.bytes = sc->packed;
            memset(ks.bytes, 0, (m + 3) / 4 + 1);
            for (size_t i = 0; i < m; i++)
                ks.bytes[i >> 2] |= (uint8_t)(base_code(s.lay, sc->filt[i]) << (6 - 2 * (i & 3)));
            iterate_kmers(s.lay, &ks, prm->k, sink, &s, NULL);
        }
    }
    if (prm->algo == LASH_OR_HMH) hmh_save(s.lay, s.hmh, image);
    else if (prm->algo == LASH_OR_HLL) hll_save(s.lay, s.p, s.reg, s.hll_zero, s.hll_sum, image);
    else ull_save(s.lay, s.p, s.reg, image);
    sketch_free(&s);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * files.par_iter() — one task per genome, dynamic scheduling (utils.rs:450-452)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const lash_or_params *prm; const uint8_t *seq; const uint64_t *rec_off, *genome_rec_off;
    uint32_t n_genomes; uint8_t *images; size_t image_bytes; volatile uint32_t *next; int err;
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    scratch_t sc = {0};
    for (;;) {
        uint32_t g = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (g >= j->n_genomes) break;
        uint64_t r0 = j->genome_rec_off[g], r1 = j->genome_rec_off[g + 1];
        if (sketch_genome_with(j->prm, j->seq, j->rec_off + r0, r1 - r0, j->images + (size_t)g * j->image_bytes, &sc))
            j->err = -1;
    }
    scratch_free(&sc);
    return NULL;
}

int lash_or_sketch_genomes(const lash_or_params *prm, const uint8_t *seq, const uint64_t *rec_off,
                           const uint64_t *genome_rec_off, uint32_t n_genomes, uint8_t *images, int threads)
{
    if (check_params(prm->algo, prm->k, prm->p)) return -1;
    if (threads < 1) threads = 1;
    volatile uint32_t next = 0;
    mt_job job = { prm, seq, rec_off, genome_rec_off, n_genomes, images, lash_or_image_bytes_layout(lay_of(prm), prm->algo, prm->p), &next, 0 };
    if (threads == 1) { mt_worker(&job); return job.err; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) { jobs[t] = job; pthread_create(&th[t], NULL, mt_worker, &jobs[t]); }
    int err = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); err |= jobs[t].err; }
    free(th); free(jobs);
    return err;
}

/* ------------------------------------------------------------------------------------------
 * Union of serialized sketches (dist side: utils.rs:171 Sketch::union, :261 UltraLogLog::merge,
 * :357 HyperLogLog::union) — used by tests of the on-device merge.
 * ---------------------------------------------------------------------------------------- */
int lash_or_merge_images_layout(const lash_or_layout *lay, int algo, int p, const uint8_t *a, const uint8_t *b, uint8_t *out)
{
    if (!lay) lay = &DEFAULT_LAYOUT;
    const size_t hdr = header_len(header_tpl(lay, algo));
    if (algo == LASH_OR_HMH) {
        uint16_t *regs = (uint16_t *)malloc(HMH_M * 2);
        const int lo = lay->hmh_reg_be ? 1 : 0, hi = 1 - lo;
        for (uint32_t i = 0; i < HMH_M; i++) {
            uint16_t x = (uint16_t)(a[hdr + 2 * i + lo] | (a[hdr + 2 * i + hi] << 8));
            uint16_t y = (uint16_t)(b[hdr + 2 * i + lo] | (b[hdr + 2 * i + hi] << 8));
            regs[i] = x > y ? x : y;
        }
        hmh_save(lay, regs, out);
        free(regs);
        return 0;
    }
    size_t n = (size_t)1 << p;
    if (algo == LASH_OR_HLL) {
        uint8_t *m = (uint8_t *)malloc(n);
        uint64_t zero = 0; double sum = 0.0;
        for (size_t i = 0; i < n; i++) {
            m[i] = a[hdr + i] > b[hdr + i] ? a[hdr + i] : b[hdr + i];
            zero += (m[i] == 0);
            sum += pow2_neg(m[i]);
        }
        hll_save(lay, p, m, zero, sum, out);
        free(m);
        return 0;
    }
    if (algo == LASH_OR_ULL) {
        uint8_t *st = (uint8_t *)malloc(n);
        for (size_t i = 0; i < n; i++) {
            uint8_t x = a[hdr + i], y = b[hdr + i];
            if (x == 0) st[i] = y;
            else if (y == 0) st[i] = x;
            else st[i] = ull_pack(ull_unpack(x) | ull_unpack(y));
        }
        ull_save(lay, p, st, out);
        free(st);
        return 0;
    }
    return -1;
}
int lash_or_merge_images(int algo, int p, const uint8_t *a, const uint8_t *b, uint8_t *out)
{
    return lash_or_merge_images_layout(&DEFAULT_LAYOUT, algo, p, a, b, out);
}

/* ------------------------------------------------------------------------------------------
 * From file bytes: needletail 0.6.3's behaviour for uncompressed input as lash uses it
 * (utils.rs:453-459) [UNPINNED: restated from needletail's documented format rules].
 *   FASTA: a record is a '>' header line, then sequence lines up to the next line that starts with '>';
 *          seq() strips '\n' and '\r'.  FASTQ: '@' header line, sequence line, '+' line, quality line of the
 *          same length; a record that breaks this ends the iteration with an error (the loop keeps what came before).
 * ---------------------------------------------------------------------------------------- */
typedef struct { const lash_or_params *prm; sketch_t *s; scratch_t *sc; kmer_sink sink; } rec_ctx;

static void sketch_record(rec_ctx *rc, const uint8_t *rec, size_t n)
{
    if (rc->prm->amino) { sketch_record_aa(rc->prm, rc->s->lay, rec, n, rc->sink, rc->s); return; }
    scratch_reserve(rc->sc, n);
    size_t m = lash_or_filter_out_n(rec, n, rc->sc->filt);          /* utils.rs:459 */
    if (m < (size_t)rc->prm->k) return;                             /* utils.rs:460-462 */
    kseq_t ks;
    ks.n_bases = m;
    ks.bytes = rc->sc->packed;
    memset(ks.bytes, 0, (m + 3) / 4 + 1);
    for (size_t i = 0; i < m; i++)
        ks.bytes[i >> 2] |= (uint8_t)(base_code(rc->s->lay, rc->sc->filt[i]) << (6 - 2 * (i & 3)));
    iterate_kmers(rc->s->lay, &ks, rc->prm->k, rc->sink, rc->s, NULL);
}

static const uint8_t *line_end(const uint8_t *p, const uint8_t *end)
{
    const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
    return nl ? nl : end;
}

static int sketch_file_with(const lash_or_params *prm, const uint8_t *buf, uint64_t len, uint8_t *image, scratch_t *sc)
{
    if (check_params(prm->algo, prm->k, prm->p) || lash_or_layout_check(lay_of(prm)) || (prm->amino && prm->k > 12)) return -1;
    if (len == 0 || (buf[0] != '>' && buf[0] != '@')) return -2;    /* parse_fastx_file(..).expect("Invalid input file") */
    sketch_t s;
    sketch_new(&s, prm);
    rec_ctx rc = { prm, &s, sc, prm->algo == LASH_OR_HMH ? hmh_add_kmer : prm->algo == LASH_OR_HLL ? hll_add_kmer : ull_add_kmer };
    const uint8_t *p = buf, *end = buf + len;
    uint8_t *joined = NULL; size_t joined_cap = 0;
    if (buf[0] == '>') {
        while (p < end) {
            const uint8_t *e = line_end(p, end);                    /* header line */
            p = e < end ? e + 1 : end;
            size_t n = 0;
            while (p < end && *p != '>') {                           /* sequence lines, joined without line ends */
                e = line_end(p, end);
                size_t ll = (size_t)(e - p);
                while (ll && p[ll - 1] == '\r') ll--;
                if (joined_cap < n + ll + 1) { joined_cap = (n + ll + 1) * 2; joined = (uint8_t *)realloc(joined, joined_cap); }
                memcpy(joined + n, p, ll);
                n += ll;
                p = e < end ? e + 1 : end;
            }
            sketch_record(&rc, joined ? joined : buf, n);
        }
    } else {
        const int skip_bad = lay_of(prm)->fastq_skip_bad;            /* switch U6: what follows a malformed record */
        while (p < end) {
            const uint8_t *rec = p;
            int ok = 0;
            do {
                if (*p != '@') break;                                /* malformed */
                const uint8_t *e = line_end(p, end);
                if (e >= end) break;
                const uint8_t *sq = e + 1, *se = line_end(sq, end);
                if (se >= end) break;
                const uint8_t *pl = se + 1;
                if (pl >= end || *pl != '+') break;
                const uint8_t *pe = line_end(pl, end);
                if (pe >= end) break;
                const uint8_t *ql = pe + 1, *qe = line_end(ql, end);
                size_t sl = (size_t)(se - sq), qn = (size_t)(qe - ql);
                while (sl && sq[sl - 1] == '\r') sl--;
                while (qn && ql[qn - 1] == '\r') qn--;
                if (sl != qn) break;                                 /* sequence and quality lengths differ */
                sketch_record(&rc, sq, sl);
                p = qe < end ? qe + 1 : end;
                ok = 1;
            } while (0);
            if (ok) continue;
            if (!skip_bad) break;                                    /* the iterator is finished by the error */
            /* resume at the next plausible record start after `rec`: a line that starts with '@' whose line after next starts with '+' */
            const uint8_t *c = line_end(rec, end);
            p = end;
            while (c < end) {
                const uint8_t *ls = c + 1;                           /* a line start */
                if (ls >= end) break;
                if (*ls == '@') {
                    const uint8_t *l1 = line_end(ls, end);
                    const uint8_t *l2 = l1 < end ? line_end(l1 + 1, end) : end;
                    if (l2 < end && l2 + 1 < end && l2[1] == '+') { p = ls; break; }
                }
                c = line_end(ls, end);
            }
        }
    }
    free(joined);
    if (prm->algo == LASH_OR_HMH) hmh_save(s.lay, s.hmh, image);
    else if (prm->algo == LASH_OR_HLL) hll_save(s.lay, s.p, s.reg, s.hll_zero, s.hll_sum, image);
    else ull_save(s.lay, s.p, s.reg, image);
    sketch_free(&s);
    return 0;
}

typedef struct {
    const lash_or_params *prm; const uint8_t *const *bufs; const uint64_t *lens;
    uint32_t n_files; uint8_t *images; size_t image_bytes; volatile uint32_t *next; int err;
} file_job;

static void *file_worker(void *arg)
{
    file_job *j = (file_job *)arg;
    scratch_t sc = {0};
    for (;;) {
        uint32_t f = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (f >= j->n_files) break;
        int rc = sketch_file_with(j->prm, j->bufs[f], j->lens[f], j->images + (size_t)f * j->image_bytes, &sc);
        if (rc) j->err = rc;
    }
    scratch_free(&sc);
    return NULL;
}

int lash_or_sketch_file_buffers(const lash_or_params *prm, const uint8_t *const *bufs, const uint64_t *lens,
                                uint32_t n_files, uint8_t *images, int threads)
{
    if (check_params(prm->algo, prm->k, prm->p)) return -1;
    if (threads < 1) threads = 1;
    volatile uint32_t next = 0;
    file_job job = { prm, bufs, lens, n_files, images, lash_or_image_bytes_layout(lay_of(prm), prm->algo, prm->p), &next, 0 };
    if (threads == 1) { file_worker(&job); return job.err; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    file_job *jobs = (file_job *)malloc(sizeof(file_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) { jobs[t] = job; pthread_create(&th[t], NULL, file_worker, &jobs[t]); }
    int err = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (jobs[t].err) err = jobs[t].err; }
    free(th); free(jobs);
    return err;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic genomes: base i of genome g = 2 bits of a splitmix64 stream.
 * ---------------------------------------------------------------------------------------- */
#define SYNTH_SEED 20260128ULL
#define GOLDEN 0x9E3779B97F4A7C15ULL
static inline uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + GOLDEN;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void lash_or_synth_genome(uint64_t genome, uint64_t n_bases, uint8_t *out_ascii)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    uint64_t base = SYNTH_SEED ^ (genome * GOLDEN);
    for (uint64_t i = 0; i < n_bases; i += 32) {
        uint64_t w = splitmix64(base + (i >> 5));
        uint64_t lim = n_bases - i < 32 ? n_bases - i : 32;
        for (uint64_t j = 0; j < lim; j++) out_ascii[i + j] = (uint8_t)acgt[(w >> (2 * j)) & 3];
    }
}
