/*
 * lash_oracle.h — CPU ORACLE for the lash sketching hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of the reference algorithm
 * (/root/reference/src/utils.rs:33-41, 57-64, 377-434, 439-510, 567-574) and of the
 * crate-internal rules it calls into.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; the product library (liblash_gfx950.so)
 * never links, dlopens or calls anything in this directory.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - XXH3 layer (xxh3_64 of 8 bytes, xxh3_128 of 4 bytes, seeded): PINNED against
 *     libxxhash 0.8.2 via python-xxhash golden vectors committed in tests/golden/.
 *   - kmerutils / hyperminhash / streaming_algorithms / ultraloglog crate rules and
 *     the three `save` byte layouts: PARITY UNPINNED.  The crates are un-vendored
 *     third-party dependencies (Cargo.lock:713,917,1834,2012), no Rust toolchain exists
 *     in this image, and the reference ships no tests or golden vectors.  They are
 *     restated from the published algorithms (axiomhq/hyperminhash, HLL of Flajolet
 *     et al. as in alecmocatta/streaming_algorithms, hash4j UltraLogLog) and every
 *     unverified choice sits behind exactly one function or switch in lash_oracle.c.
 */
#ifndef LASH_ORACLE_H
#define LASH_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LASH_OR_HMH = 0, LASH_OR_HLL = 1, LASH_OR_ULL = 2 };

/* Every crate-internal rule this image cannot verify, as DATA: one struct consumed by every
 * function below, byte-compatible with the product's `lash_layout` (include/lash_gfx950.h).  tools/ref_probe/fit_layout.py
 * searches this space against images written by the real `lash`; NULL / lash_or_layout_default() = the hypotheses of
 * SURVEY App. A.  Header templates are strings of field codes, written in order before the register array:
 *   'a' alpha f64 | 'z' zero u64 | 'Z' zero u32 | 's' sum f64 | 'p' p u8 | 'P' p u32 | 'Q' p u64 |
 *   'l' register count u64 | 'L' register count u32            (all little-endian, bincode fixint style) */
typedef struct {
    uint8_t base_code[4];     /* U5: 2-bit codes of 'A','C','G','T' (kmerutils Alphabet2b); complement(b) = code of the
                                 complementary letter = b ^ (code[A] ^ code[T])                      default {0,1,2,3} */
    uint8_t kmer_lsb_first;   /* U5: 0 = a k-mer's FIRST base sits in its most significant 2 bits (shift-left-and-OR
                                 iterator); 1 = in its least significant 2 bits                              default 0 */
    uint8_t hmh_x_low;        /* U1: which half of xxh3_128 is x (bucket, lz): 0 = high 64 bits, 1 = low     default 0 */
    uint8_t hmh_reg_be;       /* U2: HyperMinHash registers saved as u16 little-endian (0) or big-endian (1) default 0 */
    uint8_t hll_bucket_high;  /* U3: HyperLogLog bucket = low p bits of the hash, rho from the rest (0, as upstream
                                 `push`), or bucket = top p bits, rho from the lower 64-p bits (1)          default 0 */
    char    hmh_header[8];    /* U2: ""        */
    char    hll_header[8];    /* U3: "azspl"   (bincode of alpha, zero, sum, p, Box<[u8]> length prefix)  */
    char    ull_header[8];    /* U4: "l"       (bincode Vec<u8> length prefix)                            */
    uint8_t fastq_skip_bad;   /* U6: what lash's record loop sees after a malformed FASTQ record.  The loop is
                                 `while let Some(res) = reader.next() { if let Ok(rec) = res {..} }` (utils.rs:457-458): it KEEPS CALLING
                                 next() after an Err.  0 = needletail's iterator is finished by the error (next() -> None): the
                                 records before it stand, nothing after it is seen.  1 = the iterator goes on: the malformed
                                 record is dropped and reading resumes at the next line that starts with '@' and whose
                                 line after next starts with '+'                                              default 0 */
    uint8_t aa_code_zero_based; /* U7 (amino-acid sketches, utils.rs:511-563): kmerutils' 5-bit residue codes over "ACDEFGHIKLMNPQRSTVWY"
                                 in that order — 0: A = 1 ... Y = 20 (the crate's match table as recalled), 1: A = 0 ... Y = 19   default 0 */
    uint8_t reserved[6];      /* zero */
} lash_or_layout;             /* 40 bytes */

void   lash_or_layout_default(lash_or_layout *out);
/* 0 if usable: base_code a permutation of 0..3, header templates made of known field codes */
int    lash_or_layout_check(const lash_or_layout *lay);
size_t lash_or_header_bytes(const lash_or_layout *lay, int algo);

typedef struct {
    int algo;            /* LASH_OR_HMH / HLL / ULL   (main.rs:210-246)              */
    int k;               /* 1..=32                    (utils.rs:466-502)             */
    int p;               /* HLL 4..=16, ULL 3..=26; ignored for HMH (main.rs:212-213) */
    uint64_t seed;       /* -s, default 42            (main.rs:88-95)                */
    int hmh_x_is_low;    /* switch U1: 0 => x = high64(xxh3_128), y = low64; OR-ed with layout->hmh_x_low */
    const lash_or_layout *layout;   /* NULL = lash_or_layout_default() */
    int amino;           /* != 0: the amino-acid branch of sketch_files (utils.rs:511-563; `aa`, hard-wired false at main.rs:198):
                            records are upper-cased, shorter-than-k RAW records skipped, every byte outside the 20 residue
                            letters deleted (filter_out_a, utils.rs:43-55), 5-bit codes, k <= 12, no reverse complement,
                            mask_aa_bits (utils.rs:66-76)                                                   */
} lash_or_params;

/* XXH3 short-input closed forms (utils.rs:412,428 and inside hyperminhash for :397). */
uint64_t lash_or_xxh3_64_8b(uint64_t v, uint64_t seed);
void     lash_or_xxh3_128_4b(uint32_t w, uint64_t seed, uint64_t *lo, uint64_t *hi);

/* utils.rs:33-41 — returns number of bytes kept. `out` must hold n bytes. */
size_t   lash_or_filter_out_n(const uint8_t *seq, size_t n, uint8_t *out);
/* utils.rs:57-64 */
uint64_t lash_or_mask_bits(uint64_t v, int k);

/* Masked canonical k-mers of ONE record, in iterator order (utils.rs:459-499).
 * Returns the count (0 if fewer than k bases survive the filter). `out` may be NULL. */
uint64_t lash_or_record_kmers(const uint8_t *seq, size_t n, int k, uint64_t *out);

/* Size in bytes of one serialized sketch (what S::save writes, utils.rs:400-402,415-417,431-433). */
size_t   lash_or_image_bytes(int algo, int p);                                  /* default layout */
size_t   lash_or_image_bytes_layout(const lash_or_layout *lay, int algo, int p);

/* One file == one sketch (utils.rs:452-508): all records of one genome -> one image.
 * rec_off has n_rec+1 byte offsets into seq.  Returns 0 or a negative error. */
int      lash_or_sketch_genome(const lash_or_params *prm, const uint8_t *seq,
                               const uint64_t *rec_off, uint64_t n_rec, uint8_t *image);

/* files.par_iter().map(...).collect() (utils.rs:450-452,507-509): one task per genome,
 * dynamic scheduling over `threads` OS threads; images written in genome order. */
int      lash_or_sketch_genomes(const lash_or_params *prm, const uint8_t *seq,
                                const uint64_t *rec_off, const uint64_t *genome_rec_off,
                                uint32_t n_genomes, uint8_t *images, int threads);

/* Union of two serialized sketches of the same algo/p, as the dist side would form it
 * (hyperminhash union = max, HLL union = max + recomputed zero/sum,
 * UltraLogLog::merge = pack(unpack|unpack)).  utils.rs:171,261,357. */
int      lash_or_merge_images(int algo, int p, const uint8_t *a, const uint8_t *b, uint8_t *out);   /* default layout */
int      lash_or_merge_images_layout(const lash_or_layout *lay, int algo, int p, const uint8_t *a, const uint8_t *b,
                                     uint8_t *out);

/* The whole per-file closure (utils.rs:452-508) from FILE BYTES: a needletail-like parse of uncompressed FASTA ('>',
 * multi-line, CR/LF stripped) or FASTQ ('@', 4-line records) followed by the path above, one sketch per file, one task
 * per file over `threads` threads.  A file whose first byte is neither '>' nor '@' is an error (parse_fastx_file fails,
 * utils.rs:453 `expect`); iteration stops at the first malformed FASTQ record (`while let Some(Ok(..))`, utils.rs:457).
 * This is what bench.py's parse-inclusive CPU figure times. */
int      lash_or_sketch_file_buffers(const lash_or_params *prm, const uint8_t *const *bufs, const uint64_t *lens,
                                     uint32_t n_files, uint8_t *images, int threads);

/* Synthetic genome generator of SURVEY.md §8(d): base i of genome g. */
void     lash_or_synth_genome(uint64_t genome, uint64_t n_bases, uint8_t *out_ascii);

#ifdef __cplusplus
}
#endif
#endif
