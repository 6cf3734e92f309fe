// inflate_fast.cpp — see inflate_fast.hpp.
#include "inflate_fast.hpp"

#include <cstdlib>
#include <cstring>
#include <initializer_list>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lashhost {
namespace {

// ---- table entries -------------------------------------------------------------------------------------------------------
//   bits  7..0   bits to drop from the bit buffer for this symbol: its code (sub-table entries: the whole code) PLUS its
//                extra bits — one shift advances the stream, the extra value is cut out of a saved copy off the critical path
//   bits 11..8   extra-bit count of a length / distance symbol; of a SUB entry: index bits of its sub-table
//   bits 15..12  kind flags
//   bits 31..16  literal byte / length base / distance base / sub-table start
constexpr uint32_t E_LIT = 0x1000u, E_EOB = 0x2000u, E_SUB = 0x4000u, E_BAD = 0x8000u;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

enum Kind { K_PRECODE, K_LITLEN, K_DIST };

inline uint32_t entry_for(Kind kind, unsigned sym, unsigned len)
{
    if (kind == K_PRECODE) return (sym << 16) | len;
    if (kind == K_LITLEN) {
        if (sym < 256) return E_LIT | (sym << 16) | len;
        if (sym == 256) return E_EOB | len;
        if (sym <= 285) return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8) | (len + LEN_EXTRA[sym - 257]);
        return E_BAD | len;                                  // 286, 287: codes exist in the fixed tree, never valid in data
    }
    if (sym < 30) return ((uint32_t)DIST_BASE[sym] << 16) | ((uint32_t)DIST_EXTRA[sym] << 8) | (len + DIST_EXTRA[sym]);
    return E_BAD | len;
}

inline unsigned bit_reverse(unsigned code, unsigned len)
{
    unsigned r = 0;
    for (unsigned i = 0; i < len; ++i) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Canonical Huffman decode table from code lengths (RFC 1951 3.2.2), codes LSB-first as they sit in the bit buffer.
// Accepts what zlib's inflate_table accepts: complete codes; an incomplete code only when its longest code has one bit
// (a single symbol; distance codes of blocks with one distance, or none at all).  Over-subscribed sets are rejected.
bool build_table(const uint8_t *lens, unsigned n_sym, Kind kind, unsigned root, uint32_t *table, unsigned cap)
{
    unsigned count[16] = {0};
    for (unsigned s = 0; s < n_sym; ++s) count[lens[s]]++;
    unsigned max_len = 15;
    while (max_len > 0 && count[max_len] == 0) --max_len;
    const unsigned root_size = 1u << root;
    if (max_len == 0) {                                      // no codes at all
        if (kind != K_DIST) return false;                    // (a block of literals only may come without distance codes)
        for (unsigned i = 0; i < root_size; ++i) table[i] = E_BAD | 1u;
        return true;
    }
    int left = 1;
    for (unsigned len = 1; len <= 15; ++len) {
        left = (left << 1) - (int)count[len];
        if (left < 0) return false;                          // over-subscribed
    }
    if (left > 0 && (kind == K_PRECODE || max_len != 1)) return false;   // incomplete
    unsigned next_code[16];
    {
        unsigned code = 0;
        count[0] = 0;
        for (unsigned len = 1; len <= 15; ++len) { code = (code + count[len - 1]) << 1; next_code[len] = code; }
    }
    for (unsigned i = 0; i < root_size; ++i) table[i] = E_BAD | 1u;
    // codes longer than the root: the longest code under each root-bit prefix sizes that prefix's sub-table
    uint16_t rev[288];
    uint8_t sub_max[1u << 12];                           // root <= 12
    const bool has_long = max_len > root;
    if (has_long) memset(sub_max, 0, root_size);
    for (unsigned s = 0; s < n_sym; ++s) {
        const unsigned len = lens[s];
        if (!len) continue;
        const unsigned r = bit_reverse(next_code[len]++, len);
        rev[s] = (uint16_t)r;
        if (len <= root) {
            const uint32_t e = entry_for(kind, s, len);
            for (unsigned i = r; i < root_size; i += 1u << len) table[i] = e;
        } else {
            uint8_t &m = sub_max[r & (root_size - 1)];
            if (len > m) m = (uint8_t)len;
        }
    }
    if (!has_long) return true;
    unsigned next_free = root_size;
    for (unsigned pfx = 0; pfx < root_size; ++pfx) {
        if (!sub_max[pfx]) continue;
        const unsigned sub_bits = sub_max[pfx] - root, size = 1u << sub_bits;
        if (next_free + size > cap) return false;            // cannot happen for valid deflate parameters
        table[pfx] = E_SUB | (next_free << 16) | (sub_bits << 8) | root;
        for (unsigned i = 0; i < size; ++i) table[next_free + i] = E_BAD | (root + 1u);
        next_free += size;
    }
    for (unsigned s = 0; s < n_sym; ++s) {
        const unsigned len = lens[s];
        if (len <= root) continue;
        const uint32_t pe = table[rev[s] & (root_size - 1)];
        const unsigned start = pe >> 16, size = 1u << ((pe >> 8) & 15u);
        const uint32_t e = entry_for(kind, s, len);
        for (unsigned i = rev[s] >> root; i < size; i += 1u << (len - root)) table[start + i] = e;
    }
    return true;
}

inline uint64_t load_le64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

}  // namespace

void InflateStream::begin()
{
    bitbuf_ = 0;
    bitcnt_ = 0;
    phase_ = PH_HEADER;
    final_ = false;
    stored_left_ = 0;
}

// byte-wise refill (headers, stored blocks, the last bytes of the input); compatible with the branch-free refill of the
// fast loop: bits above bitcnt_ are either zero or genuine upcoming stream bits
bool InflateStream::need_bits(const uint8_t *in, size_t in_n, size_t &ip, unsigned n)
{
    while (bitcnt_ < n) {
        if (ip >= in_n) return false;
        if (bitcnt_ > 56) return true;                       // cannot happen for n <= 56
        bitbuf_ |= (uint64_t)in[ip++] << bitcnt_;
        bitcnt_ += 8;
    }
    return true;
}

bool InflateStream::read_dynamic_header(const uint8_t *in, size_t in_n, size_t &ip, int &err)
{
    err = TRUNCATED;
    if (!need_bits(in, in_n, ip, 14)) return false;
    const unsigned hlit = (unsigned)(bitbuf_ & 31u) + 257u, hdist = (unsigned)((bitbuf_ >> 5) & 31u) + 1u,
                   hclen = (unsigned)((bitbuf_ >> 10) & 15u) + 4u;
    bitbuf_ >>= 14; bitcnt_ -= 14;
    if (hlit > 286 || hdist > 30) { err = BAD_DATA; return false; }
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t pre[19] = {0};
    for (unsigned i = 0; i < hclen; ++i) {
        if (!need_bits(in, in_n, ip, 3)) return false;
        pre[ORDER[i]] = (uint8_t)(bitbuf_ & 7u);
        bitbuf_ >>= 3; bitcnt_ -= 3;
    }
    uint32_t ptab[128];
    if (!build_table(pre, 19, K_PRECODE, 7, ptab, 128)) { err = BAD_DATA; return false; }
    uint8_t lens[286 + 30 + 138];
    unsigned at = 0;
    const unsigned total = hlit + hdist;
    while (at < total) {
        // a code of the code-length alphabet (<= 7 bits) + its extra bits (<= 7): the stream may end inside the 14
        if (!need_bits(in, in_n, ip, 14) && bitcnt_ == 0) return false;
        const uint32_t e = ptab[bitbuf_ & 127u];
        const unsigned len = e & 0xFFu;
        if (e & E_BAD) { err = BAD_DATA; return false; }
        if (len > bitcnt_) return false;
        const unsigned sym = e >> 16;
        unsigned extra_bits = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
        if (len + extra_bits > bitcnt_) return false;
        bitbuf_ >>= len; bitcnt_ -= len;
        if (sym < 16) { lens[at++] = (uint8_t)sym; continue; }
        const unsigned x = (unsigned)(bitbuf_ & ((1u << extra_bits) - 1u));
        bitbuf_ >>= extra_bits; bitcnt_ -= extra_bits;
        unsigned rep, val = 0;
        if (sym == 16) {
            if (at == 0) { err = BAD_DATA; return false; }
            val = lens[at - 1];
            rep = 3 + x;
        } else rep = (sym == 17 ? 3 : 11) + x;
        if (at + rep > total) { err = BAD_DATA; return false; }
        memset(lens + at, (int)val, rep);
        at += rep;
    }
    err = BAD_DATA;
    if (lens[256] == 0) return false;                        // no end-of-block code
    if (!build_table(lens, hlit, K_LITLEN, LL_ROOT, ll_, LL_CAP)) return false;
    if (!build_table(lens + hlit, hdist, K_DIST, D_ROOT, d_, D_CAP)) return false;
    return true;
}

// Sequence text is literal-heavy with 2-3-bit codes, and one table lookup per literal is a serial chain (index -> load ->
// shift -> index ...: ~8 cycles per byte).  ml_[w] therefore holds, for every 11-bit window w of the bit buffer, the run of
// up to three literals whose codes fit entirely inside the window: bytes in bits 23..0, bits consumed in 27..24, count in
// 29..28 (0 = the next symbol is not such a literal: use ll_).  A code of length L depends on the low L bits of its table
// index only, so ll_[(w >> used) & mask] is exact as long as L <= 11 - used.
void InflateStream::build_literal_runs()
{
    constexpr unsigned MASK = (1u << LL_ROOT) - 1u;
    for (unsigned w = 0; w <= MASK; ++w) {
        uint32_t lits = 0;
        unsigned used = 0, cnt = 0;
        while (cnt < 3) {
            const uint32_t e = ll_[(w >> used) & MASK];
            const unsigned len = e & 0xFFu;
            if (!(e & E_LIT) || len > LL_ROOT - used) break;
            lits |= ((e >> 16) & 0xFFu) << (8 * cnt);
            used += len;
            ++cnt;
        }
        ml_[w] = lits | (used << 24) | (cnt << 28);
    }
}

// Huffman-coded block body.  Returns 0 = end of block, 1 = leave the fast loop (FAST) / output full (!FAST), < 0 = Status error.
template <bool FAST>
int InflateStream::decode_block(const uint8_t *in, size_t in_n, size_t &ip_ref, uint8_t *out, size_t &op_ref, size_t out_cap)
{
    uint64_t bb = bitbuf_;
    unsigned bc = bitcnt_;
    size_t ip = ip_ref, op = op_ref;
    const uint32_t *ll = ll_, *dt = d_, *ml = ml_;
    int ret;
#define LASH_SAVE() do { bitbuf_ = bb; bitcnt_ = bc; ip_ref = ip; op_ref = op; } while (0)
#define LASH_REFILL_FAST() do { bb |= load_le64(in + ip) << bc; ip += (63u - bc) >> 3; bc |= 56u; } while (0)
#define LASH_REFILL_SAFE() do { while (bc <= 56u && ip < in_n) { bb |= (uint64_t)in[ip++] << bc; bc += 8u; } } while (0)
    for (;;) {
        if (FAST) {
            if (ip + 16 > in_n || op + 280 > out_cap) { ret = 1; break; }
            LASH_REFILL_FAST();
        } else {
            if (op + 258 > out_cap) { ret = 1; break; }
            LASH_REFILL_SAFE();
        }
        if (FAST) {
            // a literal run first, WITHOUT a branch: in sequence text literals come as singles between short matches, so
            // "literal or match?" is a coin the predictor loses; a run of zero literals stores 4 spare bytes, moves nothing
            const uint32_t m = ml[bb & ((1u << LL_ROOT) - 1u)];
            memcpy(out + op, &m, 4);
            op += m >> 28;
            const unsigned used = (m >> 24) & 15u;
            bb >>= used; bc -= used;
        }
        uint32_t e = ll[bb & ((1u << LL_ROOT) - 1u)];
        if (e & E_SUB) e = ll[(e >> 16) + ((bb >> LL_ROOT) & ((1u << ((e >> 8) & 15u)) - 1u))];
        unsigned n = e & 0xFFu;
        if (!FAST && n > bc) { ret = TRUNCATED; break; }      // (the zero padding past the end may look up anything, E_BAD included)
        if (e & (E_LIT | E_EOB | E_BAD)) {
            if (e & E_BAD) { ret = BAD_DATA; break; }
            bb >>= n; bc -= n;
            if (e & E_EOB) { ret = 0; break; }
            out[op++] = (uint8_t)(e >> 16);
            continue;
        }
        // a match: the entry's low byte covers code + extra bits (<= 15 + 5; present in both modes after the check above)
        uint64_t saved = bb;
        bb >>= n; bc -= n;
        unsigned xb = (e >> 8) & 15u;
        const unsigned len = (e >> 16) + (unsigned)((saved >> (n - xb)) & ((1u << xb) - 1u));
        if (FAST) LASH_REFILL_FAST(); else LASH_REFILL_SAFE();
        e = dt[bb & ((1u << D_ROOT) - 1u)];
        if (e & E_SUB) e = dt[(e >> 16) + ((bb >> D_ROOT) & ((1u << ((e >> 8) & 15u)) - 1u))];
        n = e & 0xFFu;
        if (!FAST && n > bc) { ret = TRUNCATED; break; }
        if (e & E_BAD) { ret = BAD_DATA; break; }
        saved = bb;
        bb >>= n; bc -= n;
        xb = (e >> 8) & 15u;
        const size_t dist = (e >> 16) + (size_t)((saved >> (n - xb)) & ((1u << xb) - 1u));
        if (dist > op) { ret = BAD_DATA; break; }            // before the start of the stream
        uint8_t *dst = out + op;
        const uint8_t *src = dst - dist;
        op += len;
        if (FAST) {
            // >= 280 bytes of room: whole 8-byte words may run past the match's end
            if (dist >= 16) {
                // most matches of sequence text are 3..16 long: two words unconditionally, a loop only beyond that
                memcpy(dst, src, 8); memcpy(dst + 8, src + 8, 8);
                if (len > 16) {
                    uint8_t *const end = dst + len;
                    dst += 16; src += 16;
                    do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < end);
                }
            } else if (dist >= 8) {
                uint8_t *const end = dst + len;
                do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < end);
            } else if (dist == 1) {
                memset(dst, src[0], len);
            } else {
                // short period: widen it to >= 8 bytes by copying byte-wise, then continue in words
                uint8_t *const end = dst + len;
                for (unsigned i = 0; i < 8; ++i) dst[i] = src[i];            // overlapping forward copy, period `dist`
                if (len > 8) {
                    const size_t wide = dist * (8 / dist + (8 % dist ? 1 : 0));  // smallest multiple of the period >= 8
                    uint8_t *d2 = dst + 8;
                    const uint8_t *s2 = d2 - wide;
                    do { memcpy(d2, s2, 8); d2 += 8; s2 += 8; } while (d2 < end);
                }
            }
        } else {
            for (unsigned i = 0; i < len; ++i) dst[i] = src[i];
        }
    }
    LASH_SAVE();
#undef LASH_SAVE
#undef LASH_REFILL_FAST
#undef LASH_REFILL_SAFE
    return ret;
}

InflateStream::Status InflateStream::run(const uint8_t *in, size_t in_n, size_t &in_pos, uint8_t *out, size_t &out_pos, size_t out_cap)
{
    size_t ip = in_pos, op = out_pos;
    Status st = DONE;
    for (;;) {
        if (phase_ == PH_DONE) { st = DONE; break; }
        if (phase_ == PH_HEADER) {
            if (final_) {
                // byte position after the stream: whole unread bytes go back to the input
                ip -= bitcnt_ >> 3;
                bitbuf_ = 0; bitcnt_ = 0;
                phase_ = PH_DONE;
                continue;
            }
            if (!need_bits(in, in_n, ip, 3)) { st = TRUNCATED; break; }
            final_ = (bitbuf_ & 1u) != 0;
            const unsigned type = (unsigned)((bitbuf_ >> 1) & 3u);
            bitbuf_ >>= 3; bitcnt_ -= 3;
            if (type == 0) {
                const unsigned drop = bitcnt_ & 7u;
                bitbuf_ >>= drop; bitcnt_ -= drop;
                if (!need_bits(in, in_n, ip, 32)) { st = TRUNCATED; break; }
                const unsigned len = (unsigned)(bitbuf_ & 0xFFFFu), nlen = (unsigned)((bitbuf_ >> 16) & 0xFFFFu);
                bitbuf_ >>= 32; bitcnt_ -= 32;
                if ((len ^ 0xFFFFu) != nlen) { st = BAD_DATA; break; }
                ip -= bitcnt_ >> 3;                              // the rest of the bit buffer is whole bytes: hand them back
                bitbuf_ = 0; bitcnt_ = 0;
                stored_left_ = len;
                phase_ = PH_STORED;
            } else if (type == 1) {
                uint8_t lens[288 + 32];
                memset(lens, 8, 144); memset(lens + 144, 9, 112); memset(lens + 256, 7, 24); memset(lens + 280, 8, 8);
                memset(lens + 288, 5, 32);
                if (!build_table(lens, 288, K_LITLEN, LL_ROOT, ll_, LL_CAP) || !build_table(lens + 288, 32, K_DIST, D_ROOT, d_, D_CAP)) {
                    st = BAD_DATA; break;
                }
                build_literal_runs();
                phase_ = PH_HUFF;
            } else if (type == 2) {
                int err;
                if (!read_dynamic_header(in, in_n, ip, err)) { st = (Status)err; break; }
                build_literal_runs();
                phase_ = PH_HUFF;
            } else { st = BAD_DATA; break; }
            continue;
        }
        if (phase_ == PH_STORED) {
            if (stored_left_) {
                const size_t room = out_cap - op, have = in_n - ip;
                const size_t take = stored_left_ < room ? (stored_left_ < have ? stored_left_ : have) : (room < have ? room : have);
                if (take) { memcpy(out + op, in + ip, take); op += take; ip += take; stored_left_ -= take; }
                if (stored_left_) { st = (in_n - ip == 0) ? TRUNCATED : OUTPUT_FULL; break; }
            }
            phase_ = PH_HEADER;
            continue;
        }
        // PH_HUFF
        int r = decode_block<true>(in, in_n, ip, out, op, out_cap);
        if (r == 1) r = decode_block<false>(in, in_n, ip, out, op, out_cap);
        if (r == 0) { phase_ = PH_HEADER; continue; }
        st = r == 1 ? OUTPUT_FULL : (Status)r;
        break;
    }
    in_pos = ip;
    out_pos = op;
    return st;
}

// ---- CRC-32 ------------------------------------------------------------------------------------------------------------
namespace {

struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
    }
};
const CrcTables CRC;

uint32_t crc32_slice8(uint32_t crc, const uint8_t *p, size_t n)      // crc: internal (pre-inverted) state
{
    while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) { crc = CRC.t[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8); --n; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= crc;
        crc = CRC.t[7][v & 0xFFu] ^ CRC.t[6][(v >> 8) & 0xFFu] ^ CRC.t[5][(v >> 16) & 0xFFu] ^ CRC.t[4][(v >> 24) & 0xFFu] ^
              CRC.t[3][(v >> 32) & 0xFFu] ^ CRC.t[2][(v >> 40) & 0xFFu] ^ CRC.t[1][(v >> 48) & 0xFFu] ^ CRC.t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) crc = CRC.t[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8);
    return crc;
}

#if defined(__x86_64__)
// Folding with carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"),
// bit-reflected CRC-32: constants are x^N mod P for the fold distances, as tabulated in that paper for this polynomial.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc, const uint8_t *p, size_t n)   // n >= 64, multiple of 16
{
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);   // fold by 512 bits
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);   // fold by 128 bits
    const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x00000001F7011641ll, 0x00000001DB710641ll);   // mu (high), P (low)
    const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
    __m128i x1 = _mm_loadu_si128((const __m128i *)p), x2 = _mm_loadu_si128((const __m128i *)(p + 16)),
            x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64; n -= 64;
    while (n >= 64) {
        __m128i h1 = _mm_clmulepi64_si128(x1, k1k2, 0x11), h2 = _mm_clmulepi64_si128(x2, k1k2, 0x11),
                h3 = _mm_clmulepi64_si128(x3, k1k2, 0x11), h4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x00); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x00); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, h1), _mm_loadu_si128((const __m128i *)p));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, h2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, h3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, h4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    // four accumulators -> one
#define LASH_FOLD128(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x00), _mm_clmulepi64_si128(acc, k3k4, 0x11)), next)
    x1 = LASH_FOLD128(x1, x2);
    x1 = LASH_FOLD128(x1, x3);
    x1 = LASH_FOLD128(x1, x4);
    while (n >= 16) { x1 = LASH_FOLD128(x1, _mm_loadu_si128((const __m128i *)p)); p += 16; n -= 16; }
#undef LASH_FOLD128
    // 128 -> 64 bits, 64 -> 32 bits, Barrett reduction
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), t);
    t = x1;
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_clmulepi64_si128(x1, poly, 0x10);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_clmulepi64_si128(x1, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

// the folding constants are recalled, not derived here: verify them once against the table method before trusting them
bool clmul_usable()
{
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    uint8_t buf[64 + 16 * 5];
    uint32_t s = 0x12345678u;
    for (size_t i = 0; i < sizeof buf; ++i) { s = s * 1664525u + 1013904223u; buf[i] = (uint8_t)(s >> 24); }
    for (size_t n = 64; n <= sizeof buf; n += 16)
        for (uint32_t seed : {0u, 0xFFFFFFFFu, 0xDEADBEEFu})
            if (crc32_clmul(seed, buf, n) != crc32_slice8(seed, buf, n)) return false;
    return true;
}
#endif

}  // namespace

uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n)
{
    uint32_t c = ~crc;
#if defined(__x86_64__)
    static const bool use_clmul = clmul_usable();
    if (use_clmul && n >= 128) {
        const size_t body = n & ~(size_t)15;
        c = crc32_clmul(c, p, body);
        p += body; n -= body;
    }
#endif
    return ~crc32_slice8(c, p, n);
}

// ---- gzip members ------------------------------------------------------------------------------------------------------
ByteSink::~ByteSink() { free(p); }
void ByteSink::release() { free(p); p = nullptr; n = cap = 0; }
bool ByteSink::reserve(size_t want)
{
    if (want <= cap) return true;
    size_t ncap = cap ? cap : (1u << 16);
    while (ncap < want) ncap += ncap / 2 + 4096;
    uint8_t *q = static_cast<uint8_t *>(realloc(p, ncap));
    if (!q) return false;
    p = q; cap = ncap;
    return true;
}

size_t gzip_header_length(const uint8_t *h, size_t avail)
{
    if (avail < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) return 0;
    const unsigned flg = h[3];
    size_t q = 10;
    if (flg & 4) {                                                       // FEXTRA
        if (q + 2 > avail) return 0;
        q += 2 + ((size_t)h[q] | ((size_t)h[q + 1] << 8));
    }
    for (unsigned bit : {8u, 16u})                                       // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            while (q < avail && h[q]) ++q;
            ++q;
        }
    if (flg & 2) q += 2;                                                 // FHCRC
    return q < avail ? q : 0;
}

const char *gunzip_members(const uint8_t *src, size_t n, ByteSink &out, bool one_member, size_t *consumed)
{
    size_t at = 0;
    InflateStream z;
    bool any = false;
    while (at < n) {
        // bytes after a complete member that do not start another one are ignored, as zlib's gz* readers do (padding)
        const uint8_t *h = src + at;
        if (any && (n - at < 18 || h[0] != 0x1f || h[1] != 0x8b)) break;
        if (n - at < 18) return "truncated gzip stream";
        if (h[0] != 0x1f || h[1] != 0x8b) return "not a gzip stream";
        const size_t hl = gzip_header_length(h, n - at);
        if (!hl) return (h[2] != 8 || (h[3] & 0xE0)) ? "unsupported gzip header" : "truncated gzip stream";
        const size_t q = at + hl;
        // deflate body; ISIZE of a single-member file sizes the output up front
        const size_t start = out.n;
        if (out.cap - out.n < (1u << 16)) {
            size_t guess = (n - q) * 4;
            const uint8_t *tl = src + n - 4;
            const size_t isize = (size_t)tl[0] | ((size_t)tl[1] << 8) | ((size_t)tl[2] << 16) | ((size_t)tl[3] << 24);
            if (isize >= (n - q) / 2 && isize <= (n - q) * 1100) guess = isize;       // plausible: looks like the only member
            if (guess > out.limit) guess = out.limit;
            if (!out.reserve(out.n + guess + 4096)) return "out of memory";
        }
        z.begin();
        size_t ip = q;
        for (;;) {
            // history for back-references = this member's own output: the decoder sees out.p + start as its buffer
            size_t op = out.n - start;
            const InflateStream::Status st = z.run(src, n, ip, out.p + start, op, out.cap - start);
            out.n = start + op;
            if (st == InflateStream::DONE) break;
            if (st == InflateStream::OUTPUT_FULL) {
                if (out.n > out.limit) return "member larger than the buffer limit";
                if (!out.reserve(out.cap + out.cap / 2 + (1u << 20))) return "out of memory";
                continue;
            }
            return st == InflateStream::TRUNCATED ? "truncated gzip stream" : "corrupt deflate data";
        }
        if (out.n > out.limit) return "member larger than the buffer limit";
        if (ip + 8 > n) return "truncated gzip stream";
        const uint32_t want_crc = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
        const uint32_t want_len = (uint32_t)src[ip + 4] | ((uint32_t)src[ip + 5] << 8) | ((uint32_t)src[ip + 6] << 16) | ((uint32_t)src[ip + 7] << 24);
        if ((uint32_t)(out.n - start) != want_len) return "gzip length check failed";
        if (crc32_fast(0, out.p + start, out.n - start) != want_crc) return "gzip CRC check failed";
        at = ip + 8;
        any = true;
        if (one_member) break;
    }
    if (!any) return "truncated gzip stream";
    if (consumed) *consumed = at;
    return nullptr;
}

WindowedInflate::WindowedInflate(size_t chunk) : cap_(32768 + chunk + InflateStream::MIN_ROOM) { buf_ = static_cast<uint8_t *>(malloc(cap_)); }
WindowedInflate::~WindowedInflate() { free(buf_); }

void WindowedInflate::begin()
{
    z_.begin();
    op_ = rd_ = 0;
    done_ = false;
    crc_ = 0;
    total_ = 0;
}

long WindowedInflate::read(const uint8_t *in, size_t in_n, size_t &in_pos, uint8_t *dst, size_t n, const char **err)
{
    if (!buf_) { *err = "out of memory"; return -1; }
    size_t got = 0;
    const uint32_t crc_in = crc_;
    const uint64_t total_in = total_;
    while (got < n) {
        if (rd_ < op_) {
            // crc() / total() cover exactly the bytes HANDED OUT: a caller that gives up on this decoder mid-member (pgzip.cpp:
            // zlib takes the member over) can check the replacement's prefix against them
            const size_t take = n - got < op_ - rd_ ? n - got : op_ - rd_;
            if (test_flip_ >= 0 && (uint64_t)test_flip_ >= total_ && (uint64_t)test_flip_ < total_ + take) buf_[rd_ + ((uint64_t)test_flip_ - total_)] ^= 0x20;
            memcpy(dst + got, buf_ + rd_, take);
            crc_ = crc32_fast(crc_, buf_ + rd_, take);
            total_ += take;
            rd_ += take;
            got += take;
            continue;
        }
        if (done_) break;
        if (op_ + InflateStream::MIN_ROOM > cap_) {                      // everything was handed out: keep the last 32 KiB as history
            memmove(buf_, buf_ + op_ - 32768, 32768);
            op_ = rd_ = 32768;
        }
        const InflateStream::Status st = z_.run(in, in_n, in_pos, buf_, op_, cap_);
        if (st == InflateStream::DONE) done_ = true;
        else if (st != InflateStream::OUTPUT_FULL) {
            *err = st == InflateStream::TRUNCATED ? "truncated gzip stream" : "corrupt deflate data";
            crc_ = crc_in;                                                // the caller discards this call's bytes
            total_ = total_in;
            return -1;
        }
    }
    return (long)got;
}

}  // namespace lashhost
