// lash_common.h — structures shared by the host side of liblash_gfx950 and its gfx950 kernels.
#pragma once
#include <stdint.h>

struct lash_hll_bias;      // include/lash_gfx950.h

namespace lash {

// ---- HBM layout of a packed batch (DESIGN.md "Data layout") ----------------------------------------------
// words : 2-bit bases, 16 per u32, FIRST base in bits 31:30 (so a k-mer is a funnel-shift window and its
//         numeric value follows kmerutils' "first base most significant" rule).  Each genome starts on a
//         16-byte boundary (word_off % 4 == 0) and is followed by >= PAD_WORDS readable words.
// brk   : 1 bit per packed base position, LSB-first inside a u32; bit i is set when position i is the first
//         surviving base of a record (k-mers never span records: utils.rs:457-464).
// nvalid: per genome, the number of bases that survived filter_out_n (utils.rs:33-41), device-written.
struct GenomeDesc {
    uint64_t byte_off;    // first byte of the genome's records in seq
    uint64_t byte_len;    // bytes of all its records
    uint64_t word_off;    // into words[]
    uint64_t brk_off;     // into brk[] (u32 units)
    uint64_t rec_begin;   // records [rec_begin, rec_end) of rec_off[]  (format 0 only)
    uint64_t rec_end;
    uint32_t format;      // 0 = record sequences + rec_off table; 1 = raw FASTA file bytes; 2 = raw FASTQ file bytes
    uint32_t handover;   // direct mode: waves of this genome that must find it too dirty before it is handed to stream_sketch_kernel
                         // (about 1 in 32 of the waves that work on it, at least 1; set when the work items are planned)
};

// One workgroup of the sketch kernel = one slice of one genome.
struct WorkItem {
    uint32_t genome;
    uint32_t word_begin;  // slice = packed words [word_begin, word_end) of the genome, multiples of 4
    uint32_t word_end;
    uint32_t slice;       // bits 14:0 slice index inside the genome, bit 15 ITEM_SOLE; bits 31:16 which bucket-space pass
};

constexpr uint32_t ITEM_SOLE        = 0x8000u;   // WorkItem::slice flag: the genome's only work item (writes the image itself)
constexpr int      PAD_WORDS        = 8;       // readable slack after every genome (look-ahead words)
constexpr int      SKETCH_WORDS_PER_THREAD = 4;   // one global_load_dwordx4 per lane per step
constexpr int      HMH_P            = 14;
constexpr uint32_t HMH_M            = 1u << HMH_P;

// The context's lash_layout (include/lash_gfx950.h: SURVEY App. D's unknowns as data) in the form the kernels consume.
struct LayoutDev {
    uint32_t code_lo, code_hi;   // direct route: v_perm tables indexed by (byte & 7): A=1 C=3 T=4 G=7 -> 2-bit code
    uint32_t code_tab4;          // pack route: byte i = layout code of the letter whose kmerutils-hypothesis code is i
    uint32_t comp_mask;          // complement of 16 packed bases = word ^ comp_mask  (code[A]^code[T] in every 2-bit group)
    uint32_t hdr_bytes;          // bytes written before the register array of this algo's image
    uint32_t hmh_reg_be;         // HyperMinHash registers big-endian in images
    uint32_t kmer_lsb_first;     // (informational: layout_dev() has already turned it into the complemented code tables above)
    uint32_t hll_bucket_high;    // (informational: the launchers pick the kernels' bucket-high variant, rule_variant())
    uint32_t aa_code_base;       // amino-acid kernel: code of 'A' (1, or 0 with layout.aa_code_zero_based); the other 19 follow in letter order
    uint64_t hdr_tpl;            // that header's field codes (see lash_layout), first field in the low byte, 0-terminated
                                 // (a scalar, not an array: a dynamically indexed kernel-argument array would live in scratch)
};

// XXH3 constants (XXH 0.8 spec; closed forms in SURVEY.md Appendix C, pinned by tests/golden/xxh3_vectors.json)
constexpr uint64_t XXH_PRIME64_1 = 0x9E3779B185EBCA87ULL;
constexpr uint64_t XXH_PRIME_MX1 = 0x165667919E3779F9ULL;
constexpr uint64_t XXH_PRIME_MX2 = 0x9FB21C651E98DF25ULL;
constexpr uint64_t XXH_SEC8  = 0x1cad21f72c81017cULL;
constexpr uint64_t XXH_SEC16 = 0xdb979083e96dd4deULL;
constexpr uint64_t XXH_SEC24 = 0x1f67b3b7a4a44072ULL;

inline uint64_t xxh3_short_seed(uint64_t seed)
{
    uint32_t lo = (uint32_t)seed;
    uint32_t sw = (lo >> 24) | ((lo >> 8) & 0xFF00u) | ((lo << 8) & 0xFF0000u) | (lo << 24);
    return seed ^ ((uint64_t)sw << 32);
}
// seed-dependent constants folded on the host and passed as kernel arguments
inline uint64_t xxh3_bitflip64(uint64_t seed)  { return (XXH_SEC8 ^ XXH_SEC16) - xxh3_short_seed(seed); }   // 8-byte input, 64-bit hash
inline uint64_t xxh3_bitflip128(uint64_t seed) { return (XXH_SEC16 ^ XXH_SEC24) + xxh3_short_seed(seed); }  // 4-byte input, 128-bit hash

// hyperminhash's expected_collisions(n, m) split where the GPU takes over (dist_estimators.hip): the saturated and the
// closed-form regimes are O(1) (returns true, *out set); below 2^(p+5) the crate walks 65 536 cells (returns false):
// on the host with hmh_ec_cell_walk, or as lash_hmh_pair_expected_collisions' matrix product
bool hmh_ec_closed_form(double n, double m, double *out);
double hmh_ec_from_cell_sum(double x);               // the cell sum -> the value similarity() subtracts
double hmh_ec_cell_walk(double n, double m);         // the crate's loop, term by term, on the host
// per-sketch cardinalities from register histograms made on the GPU (sketch_set.hip; dist_estimators.hip)
double hmh_cardinality_from_hist(const uint32_t *hist64, bool *exact);
int    hll_cardinality_from_hist(const uint32_t *hist256, int p, const lash_hll_bias *tables, double *out);   // LASH_OK / LASH_ERANGE

}  // namespace lash
