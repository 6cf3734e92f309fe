/*
 * lash_gfx950.h — C ABI of liblash_gfx950.so: the MI355X (gfx950) drop-in for lash's sketching hot path.
 *
 * What it replaces.  In the reference the per-k-mer plug-in boundary is
 *     trait KmerSketch { fn new(Option<u32>); fn add_kmer(&mut self, masked: u64, seed: u64); fn save(&self, w) }
 * (/root/reference/src/utils.rs:377-386) with a single caller, the per-file closure of `sketch_files`
 * (utils.rs:452-508).  A per-k-mer FFI call is meaningless for a GPU, so the boundary is that closure widened
 * to a batch:  records of many files in  ->  the byte images `S::save` would have written out
 * (utils.rs:400-402, 415-417, 431-433), in file order (utils.rs:509, 571-573).  A Rust host `write_all`s the
 * images into its zstd encoder unchanged; INTEGRATION.md shows the `extern "C"` block.
 *
 * Rules of the ABI: plain pointers and sizes only; the caller allocates and frees every buffer; the library owns
 * only what hangs off a lash_ctx / lash_packed; no exceptions or aborts cross the boundary (0 = OK, <0 = error);
 * a lash_ctx is used by one host thread at a time; there is NO CPU fallback — without a HIP device every compute
 * entry returns LASH_ENODEV.
 */
#ifndef LASH_GFX950_H
#define LASH_GFX950_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASH_ABI_VERSION 5

/* error codes */
#define LASH_OK       0
#define LASH_EINVAL  (-1)   /* bad algo / k / p / arguments: mirrors the panics at utils.rs:500-502, main.rs:245 and
                               the expect()s at utils.rs:408,423 */
#define LASH_ENODEV  (-2)   /* no usable HIP device */
#define LASH_EHIP    (-3)   /* HIP runtime error; text via lash_ctx_last_error() */
#define LASH_ENOMEM  (-4)
#define LASH_ERANGE  (-6)   /* HyperLogLog estimate <= 5 * 2^p: streaming_algorithms subtracts a bias read from the HLL++
                               empirical tables there; the tables are not in this build, so the case is refused */
#define LASH_EFORMAT (-7)   /* lash_sketch_files_raw_device: a FASTQ file's line structure broke; see lash_ctx_format_errors() */
#define LASH_ELIMIT  (-5)   /* a genome has more than 2^32-64 bases in one call (split it and merge images) */

/* -a {hmh,hll,ull}  (main.rs:69-76, 210-246) */
#define LASH_HMH 0
#define LASH_HLL 1
#define LASH_ULL 2

/* flags */
#define LASH_F_HMH_X_LOW   1u  /* SURVEY App. D switch U1: take x (bucket, lz) from the LOW 64 bits of xxh3_128
                                  (per call; OR-ed with the context layout's hmh_x_low) */
#define LASH_F_ACCUMULATE  2u  /* out_images already hold sketches of the same algo/p: union the new ones in */
#define LASH_F_AMINO       8u  /* the amino-acid branch of sketch_files (utils.rs:511-563; `aa`, which main.rs:198 hard-wires to false and
                                  whose --aa flag is commented out at main.rs:97-104): records are upper-cased, records whose RAW
                                  length is below k skipped, every byte outside the 20 residue letters deleted (filter_out_a,
                                  utils.rs:43-55), 5-bit codes, k 1..=12 (utils.rs:554 panics above), no reverse complement,
                                  mask_aa_bits (utils.rs:66-76), add_kmer as for nucleotides.  lash_sketch_batch[_async/_device] and
                                  lash_sketch_files_raw (host parse); the packed / raw-device entries return LASH_EINVAL */
#define LASH_F_NO_DIRECT   4u  /* lash_sketch_batch[_device]: pack first (ASCII -> 2-bit stream in HBM -> sketch), the route raw
                                  FASTA / FASTQ bytes always take.  By default the sketch kernels read the record bytes
                                  themselves and apply filter_out_n (utils.rs:33-41) on the fly: clean stretches as they are,
                                  deleted bytes by junction walks / in-LDS compaction, soft-masked genomes by a compacting
                                  kernel.  Same images either way; the flag exists for A/B runs */

#define LASH_F_STREAM_ONLY 16u /* lash_sketch_batch[_device]: skip the optimistic pass, every genome goes through the compacting
                                  kernel (stream_sketch_kernel) — what the context does by itself while batches keep turning out
                                  soft-masked.  Same images; for A/B runs and tests */

#define LASH_F_NO_SOLE     32u /* (ABI v5) every sketch entry: do not use the persistent small-genome kernel (sole_kernels.hip), which by
                                  default takes the genomes of a call that are at most LASH_SOLE_MAX bytes long (environment,
                                  default 393216; 0 = never) when the sketch's table fits 64 KiB of LDS — a resident workgroup streams
                                  whole genomes through an LDS ring instead of one workgroup per genome.  Same images; for A/B
                                  runs and tests */

typedef struct lash_ctx lash_ctx;        /* one per (host thread, GPU): stream, workspace, scratch */
typedef struct lash_packed lash_packed;  /* device-resident 2-bit genomes produced by lash_pack_* */

typedef struct {
    int32_t  algo;    /* LASH_HMH / LASH_HLL / LASH_ULL                                        */
    int32_t  k;       /* 1..=32 (utils.rs:466-502)                                             */
    int32_t  p;       /* HLL 4..=16, ULL 3..=26, ignored for HMH (main.rs:212-213)             */
    uint32_t flags;   /* LASH_F_*                                                              */
    uint64_t seed;    /* -s (main.rs:88-95); handed to xxh3 exactly as utils.rs:397,412,428 do  */
} lash_params;

/* ---- layout: the crate-internal rules of the reference's dependencies, as data ----------------------------------------
 * lash's k-mer values, register rules and `save` bytes are fixed inside kmerutils 0.0.14, hyperminhash 0.1.4,
 * streaming_algorithms 0.3.3 and ultraloglog 0.1.6 (Cargo.lock:917,713,1834,2012; called at utils.rs:397,412,428,464-499
 * and 401,416,432).  Those crates are not in the reference tree, so each rule that could not be verified here (SURVEY App. D,
 * U1-U5) is ONE field of this struct; the default is the hypothesis of SURVEY App. A.  tools/ref_probe/ produces images with
 * the real `lash` and fits this struct to them; a mismatch is then a field change, not a code change.
 * Header templates are strings of field codes written before the register array, all little-endian (bincode fixint):
 *   'a' alpha f64 | 'z' zero u64 | 'Z' zero u32 | 's' sum f64 | 'p' p u8 | 'P' p u32 | 'Q' p u64 |
 *   'l' register count u64 | 'L' register count u32 */
typedef struct {
    uint8_t base_code[4];     /* U5  2-bit codes of 'A','C','G','T' (a permutation of 0..3)                 {0,1,2,3} */
    uint8_t kmer_lsb_first;   /* U5  0: a k-mer's first base is its MOST significant 2 bits; 1: least                0 */
    uint8_t hmh_x_low;        /* U1  0: x (bucket, lz) = high 64 bits of xxh3_128, y = low; 1: swapped               0 */
    uint8_t hmh_reg_be;       /* U2  HyperMinHash registers saved as u16 little- (0) or big-endian (1)               0 */
    uint8_t hll_bucket_high;  /* U3  0: HLL bucket = low p bits of the hash, rho from the rest; 1: top p bits        0 */
    char    hmh_header[8];    /* U2  ""       */
    char    hll_header[8];    /* U3  "azspl"  */
    char    ull_header[8];    /* U4  "l"      */
    uint8_t fastq_skip_bad;   /* U6  what lash's record loop sees after a MALFORMED FASTQ record.  The loop is
                                     `while let Some(res) = reader.next() { if let Ok(rec) = res {..} }` (utils.rs:457-458): it
                                     keeps calling next() after an Err.  0: needletail's iterator is finished by the error — the
                                     records before it stand, nothing after it is seen.  1: the iterator goes on — the malformed
                                     record is dropped, reading resumes at the next line that starts with '@' and whose line
                                     after next starts with '+'.  tools/ref_probe carries two such files                  0 */
    uint8_t aa_code_zero_based; /* U7 amino-acid sketches (LASH_F_AMINO): kmerutils' 5-bit residue codes over "ACDEFGHIKLMNPQRSTVWY" in
                                     that order — 0: A = 1 ... Y = 20, 1: A = 0 ... Y = 19                                  0 */
    uint8_t reserved[6];      /*     zero */
} lash_layout;                /* 40 bytes */

/* Sums over every sketch call since lash_ctx_enable_timing(ctx, 1) (HIP events on the ctx stream). */
typedef struct {
    float    pack_ms;           /* ASCII -> 2-bit + record-break bitmap (incl. the small table uploads); calls
                                   that took the direct route spend nothing here: their dirty-genome pack is
                                   queued behind the direct pass and counted in sketch_ms */
    float    sketch_ms;         /* k-mer / xxh3 / register-update kernels (direct pass + the compacting kernel for handed-over genomes) */
    float    finalize_ms;       /* partial-sketch reduction + byte images                                  */
    uint32_t calls;             /* sketch calls summed                                                     */
    uint32_t sketch_launches;   /* launches of the sketch kernel summed                                    */
    uint32_t sketch_workgroups; /* workgroups of the last sketch launch                                    */
    uint32_t direct_launches;   /* launches of the direct (ASCII-reading) sketch kernel summed              */
    uint64_t kmers;             /* valid k-mers hashed, device-counted, summed                             */
    uint64_t bases_last;        /* bases that survived filter_out_n in the last call                       */
    uint64_t packed_bytes;      /* 2-bit words + break bitmap read by the sketch kernel, summed            */
    float    direct_ms;         /* the part of sketch_ms spent in the direct (ASCII-reading) sketch kernel */
    uint32_t defer_launches;    /* of sketch_launches: HyperMinHash with deferred signatures (batches of long work items)  */
    uint32_t sole_launches;     /* (ABI v5) launches of the persistent small-genome kernel summed (sole_kernels.hip)         */
} lash_timing;

/* ---- library / context ---------------------------------------------------------------------------------- */
int         lash_abi_version(void);
int         lash_device_count(void);                       /* 0 when no GPU is visible */
const char *lash_strerror(int code);
int         lash_ctx_create(lash_ctx **out, int device);   /* device >= 0 */
void        lash_ctx_destroy(lash_ctx *ctx);
int         lash_ctx_set_stream(lash_ctx *ctx, void *hip_stream);   /* run on the caller's hipStream_t (NULL = own) */
int         lash_ctx_synchronize(lash_ctx *ctx);
const char *lash_ctx_last_error(lash_ctx *ctx);
int         lash_ctx_enable_timing(lash_ctx *ctx, int on); /* (re)starts the sums; HIP events around each stage */
int         lash_ctx_get_timing(lash_ctx *ctx, lash_timing *out);   /* synchronizes the stream */

/* Page-locked host memory for the buffers handed to lash_sketch_batch (optional: pageable memory works, pinned memory
 * lets the H2D copy run at PCIe speed).  Free with lash_host_free_pinned. */
void       *lash_host_alloc_pinned(size_t bytes);
void        lash_host_free_pinned(void *p);

/* ---- parameters / sizes (host only, no GPU needed) -------------------------------------------------------- */
int         lash_params_check(const lash_params *prm);     /* LASH_OK or LASH_EINVAL */
size_t      lash_sketch_image_bytes(int algo, int p);      /* bytes S::save writes per sketch (default layout); 0 if invalid */

/* ---- layout (host only) ------------------------------------------------------------------------------------------ */
void        lash_layout_default(lash_layout *out);
int         lash_layout_check(const lash_layout *lay);     /* LASH_OK or LASH_EINVAL */
/* "key=value,..." on top of the default: codes=ACGT (the four letters in code order) kmer=msb|lsb hmh_x=high|low
 * hmh_reg=le|be hll_bucket=low|high hmh_hdr= hll_hdr=azspl ull_hdr=l fastq_err=stop|skip aa_codes=one|zero.  NULL / "" = the default. */
int         lash_layout_parse(const char *spec, lash_layout *out);
size_t      lash_layout_header_bytes(const lash_layout *lay, int algo);
size_t      lash_layout_image_bytes(const lash_layout *lay, int algo, int p);   /* lay NULL = default; 0 if invalid */
/* Every entry below that reads or writes images through `ctx` uses the context's layout (default until set). */
int         lash_ctx_set_layout(lash_ctx *ctx, const lash_layout *lay);         /* NULL = back to the default */
int         lash_ctx_get_layout(lash_ctx *ctx, lash_layout *out);

/* ---- the hot path --------------------------------------------------------------------------------------- */
/* Replaces the body of files.par_iter().map(...) (utils.rs:450-509) for a batch of files ("genomes").
 *   seq            concatenated record sequences exactly as needletail's seqrec.seq() yields them (utils.rs:459):
 *                  no headers, no line ends; any byte other than upper-case A C G T is deleted (utils.rs:33-41)
 *   rec_off        n_rec+1 byte offsets into seq (record r = [rec_off[r], rec_off[r+1]))
 *   genome_rec_off n_genomes+1 record indices (genome g owns records [genome_rec_off[g], genome_rec_off[g+1]))
 *   out_images     n_genomes * lash_sketch_image_bytes() bytes, genome order
 * Host-resident buffers; copies in, runs, copies out, synchronizes. */
int lash_sketch_batch(lash_ctx *ctx, const lash_params *prm,
                      const uint8_t *seq, const uint64_t *rec_off, uint64_t n_rec,
                      const uint64_t *genome_rec_off, uint32_t n_genomes, uint8_t *out_images);

/* The same, without the wait: queues copy-in, kernels and copy-out and returns.  Two staging slots and separate copy streams
 * inside the context let the H2D copy of the NEXT call and the D2H copy of the PREVIOUS one overlap this call's kernels, so
 * a host that streams batches runs at the PCIe rate.  seq / rec_off / out_images must stay valid and
 * untouched until lash_ctx_synchronize(ctx) (or until two further _async calls have been made); give them page-locked memory
 * (lash_host_alloc_pinned) — copies from pageable memory are staged by the runtime and do not overlap.  genome_rec_off is
 * consumed before the call returns. */
int lash_sketch_batch_async(lash_ctx *ctx, const lash_params *prm,
                            const uint8_t *seq, const uint64_t *rec_off, uint64_t n_rec,
                            const uint64_t *genome_rec_off, uint32_t n_genomes, uint8_t *out_images);

/* Same with seq / rec_off / out_images already in device memory (asynchronous on the ctx stream).
 * The two small per-genome tables stay on the host: genome_byte_off[g] == rec_off[genome_rec_off[g]]. */
int lash_sketch_batch_device(lash_ctx *ctx, const lash_params *prm,
                             const uint8_t *d_seq, const uint64_t *d_rec_off, uint64_t n_rec,
                             const uint64_t *genome_rec_off, const uint64_t *genome_byte_off,
                             uint32_t n_genomes, uint8_t *d_out_images);

/* File bytes in: the FASTA / FASTQ parse itself runs on the device, so the host only reads (or
 * inflates) files into one buffer.  file_fmt[g] is LASH_FMT_FASTA ('>' files) or LASH_FMT_FASTQ ('@' files, 4-line
 * records); file g is bytes [file_off[g], file_off[g+1]) of raw, uncompressed.  Semantics are needletail's
 * (utils.rs:453-459) followed by the same path as lash_sketch_batch: one sketch per file. */
#define LASH_FMT_FASTA 1
#define LASH_FMT_FASTQ 2
int lash_sketch_files_raw(lash_ctx *ctx, const lash_params *prm, const uint8_t *raw, const uint64_t *file_off,
                          const uint8_t *file_fmt, uint32_t n_files, uint8_t *out_images);
int lash_sketch_files_raw_device(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_raw, const uint64_t *file_off,
                                 const uint8_t *file_fmt, uint32_t n_files, uint8_t *d_out_images);
/* Malformed FASTQ.  needletail's iterator ends with an error at a record that is not header / sequence / '+' / quality and
 * lash keeps the records before it (`while let Some(Ok(..))`, utils.rs:457).  The device parse counts lines modulo 4 and
 * cannot stop mid-file, but it CHECKS needletail's rules in full: a line in phase 0 must start with '@', one in phase 2 with
 * '+' (pack_kernels.hip); a quality line must be as long as its sequence line, CR stripped, and the file must end on a whole
 * record (fastq_check.hip, round 3 — a second scan over newline positions).  Files that fail:
 *   lash_sketch_files_raw         (bytes on the host) re-does each such file through a host parse that stops at the first
 *                                 malformed record (or drops it: layout.fastq_skip_bad) and lash_sketch_batch — exact reference
 *                                 semantics, returns LASH_OK;
 *   lash_sketch_files_raw_device  (bytes only in HBM) leaves their images unreliable; the next lash_ctx_synchronize() returns
 *                                 LASH_EFORMAT.  The flagged set equals the set of files lash_fastq_valid_prefix(buf, n) < n
 *                                 (tests/test_gpu_rawfiles.py holds them side by side).
 * Either way the indices of those files (of the last raw call) are available here: returns how many, copies up to `cap`.
 * The first byte of a file must be '>' or '@' (parse_fastx_file fails otherwise, utils.rs:453): LASH_EINVAL from both. */
uint32_t lash_ctx_format_errors(lash_ctx *ctx, uint32_t *file_index, uint32_t cap);

/* HyperLogLog images carry `sum` = sum_j 2^-m[j] (f64).  streaming_algorithms keeps it incrementally per k-mer (sum -= 2^-old;
 * sum += 2^-new); that is exactly the sum over the final registers as long as every register is <= 53 - p, and the HIP path
 * writes that sum.  A register above 53 - p (one k-mer in 2^(52-p): p = 14 -> 1 in 2.7e11) makes the incremental value depend
 * on the order of its f64 roundings — it can differ from the correctly rounded sum written here by the terms below the
 * 2^(p-53) grid (< 2^(p-52) absolute, ~1e-14 relative: the low bits of those 8 header bytes); registers, `zero` and every
 * other byte are unaffected.  This call synchronizes the context's stream and lists
 * the genomes of the LAST HyperLogLog sketch call that are in that corner (indices into that call's genomes; returns how many,
 * writes at most `cap`).  tests/test_gpu_hll_corner.py holds k-mers that reach it. */
uint32_t lash_ctx_hll_inexact_sums(lash_ctx *ctx, uint32_t *genome_index, uint32_t cap);
/* Round 4: the corner is closed for per-genome sketches.  For a flagged genome the library finds the k-mer(s) that put a register
 * above 53 - p by a binary search over PREFIXES of the genome's records (each probe an ordinary sketch call), takes the incremental
 * sum up to there from the prefix's own (exact) header, performs that k-mer's `sum -= 2^-old; sum += 2^-new` in IEEE double as
 * the crate does, and adds the exact net change of the steps that follow: the 8 bytes then equal the oracle's
 * (tests/test_gpu_hll_corner.py asserts byte equality).  lash_sketch_batch and lash_sketch_files_raw do this themselves before they
 * return (lash_ctx_hll_inexact_sums() then reports nothing); after lash_sketch_batch_device / _async call this with the same
 * arguments once the caller may wait: it synchronizes, patches `sum` in d_images and clears the report.  What stays flagged:
 * calls with LASH_F_ACCUMULATE (the registers already in the image are not the library's to replay — the CLI's streamed chunks
 * of one huge file) and the raw-bytes device entry.  ~25 small sketch calls per flagged genome, i.e. nothing on average. */
int lash_hll_replay_sums_device(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_seq, const uint64_t *d_rec_off, uint64_t n_rec,
                                const uint64_t *genome_rec_off, uint32_t n_genomes, uint8_t *d_images);
/* (ABI v5) The same for ONE FILE STREAMED IN CHUNKS with LASH_F_ACCUMULATE (the CLI's --stream-mb; utils.rs:457-505 reads the file's records
 * in order into one sketch, so the reference's `sum` is the incremental value over the whole file).  Call after each chunk's
 * lash_sketch_files_raw: raw / n_bytes / fmt = the chunk as it was handed over, image_before = the file's image before that call,
 * image_after = after it (host memory, patched in place), carry[2] + *have_carry = state the caller keeps per file (zero-initialised).
 * While no register is above 53 - p nothing happens.  A chunk that lifts one there is replayed — prefix sketches of the chunk united with
 * image_before's registers locate the k-mers, their updates are done in IEEE double as the crate does — and from then on every later
 * chunk's exact net change is added to the carried value: image_after's `sum` equals the reference's after every chunk
 * (tests/test_gpu_hll_corner.py asserts byte equality for --stream-mb 1). */
int lash_hll_replay_streamed_chunk(lash_ctx *ctx, const lash_params *prm, const uint8_t *raw, uint64_t n_bytes, int fmt,
                                   const uint8_t *image_before, uint8_t *image_after, double *carry, int *have_carry);
/* Host-side twin of the device checks (what the library itself runs on a flagged file; the `lash` CLI no longer pre-validates —
 * it hands the bytes over and reads lash_ctx_format_errors): the length of the longest prefix of `buf` that is a sequence of
 * well-formed records — '@' header, sequence, '+' line, quality of EQUAL length (CR stripped); the last record may lack its
 * final newline.  == n for a well-formed file, 0 if the first byte is not '@'.  What follows the prefix is what needletail's
 * iterator never yields; lash_fastq_neutralise_tail overwrites such a tail (inside a batch buffer whose file offsets must stay
 * contiguous) with a well-formed stand-in that contributes no base.  Host only. */
uint64_t lash_fastq_valid_prefix(const uint8_t *buf, uint64_t n);
void     lash_fastq_neutralise_tail(uint8_t *tail, uint64_t n);
/* Both in one, for either setting of layout.fastq_skip_bad: overwrites, in place, every byte needletail's iterator would not
 * turn into a record — skip_bad 0: what follows the first malformed record (lash_fastq_neutralise_tail); skip_bad 1: each
 * malformed stretch up to the next plausible record start ('@' + 'x'..., which makes it part of that record's header line), the
 * last one as a tail — so that the device parse of the buffer yields exactly the reference's records.  File offsets stay as
 * they are.  Returns the number of bytes overwritten (0: a well-formed file).  Host only. */
uint64_t lash_fastq_sanitize(uint8_t *buf, uint64_t n, int skip_bad);

/* Two-stage form for callers that keep genomes resident in HBM as 2-bit (0.28 B/base incl. break bitmap):
 * pack once, sketch many times (other k / algo / seed; a packed batch belongs to the base codes / k-mer bit order of the layout it was packed
 * under: after lash_ctx_set_layout changes those, lash_sketch_packed_device returns LASH_EINVAL).  The pack stage performs filter_out_n + KSeq::new
 * (utils.rs:459,464) and records where records begin so that no k-mer spans two records (utils.rs:457-464). */
int  lash_pack_device(lash_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_rec_off, uint64_t n_rec,
                      const uint64_t *genome_rec_off, const uint64_t *genome_byte_off, uint32_t n_genomes,
                      lash_packed **out);
int  lash_sketch_packed_device(lash_ctx *ctx, const lash_params *prm, const lash_packed *pk, uint8_t *d_out_images);
void lash_packed_free(lash_ctx *ctx, lash_packed *pk);
uint64_t lash_packed_bytes(const lash_packed *pk);          /* device bytes held */

/* Union of serialized sketches, image-wise: dst[i] = dst[i] U src[i]
 * (hyperminhash union / HyperLogLog::union / UltraLogLog::merge, utils.rs:171,261,357). */
int lash_merge_images_device(lash_ctx *ctx, int algo, int p, uint8_t *d_dst, const uint8_t *d_src, uint64_t n_images);
int lash_merge_images(lash_ctx *ctx, int algo, int p, uint8_t *dst, const uint8_t *src, uint64_t n_images);

/* dist side, HyperMinHash: the register scan of hyperminhash's Sketch::similarity for every (reference, query) pair
 * (/root/reference/src/utils.rs:150-167):  out_c[r * n_qry + q] = #{i : a_i != 0 and a_i == b_i},
 * out_n[r * n_qry + q] = #{i : a_i != 0 or b_i != 0}.  Images are 32 768-byte HMH sketches as written by `save`. */
int lash_hmh_pair_counts_device(lash_ctx *ctx, const uint8_t *d_ref_images, uint32_t n_ref, const uint8_t *d_qry_images,
                                uint32_t n_qry, uint32_t *d_out_c, uint32_t *d_out_n);
int lash_hmh_pair_counts(lash_ctx *ctx, const uint8_t *ref_images, uint32_t n_ref, const uint8_t *qry_images,
                         uint32_t n_qry, uint32_t *out_c, uint32_t *out_n);

/* dist side, HyperLogLog: for every (reference, query) pair the two numbers `len()` needs from the union sketch
 * (/root/reference/src/utils.rs:355-363: ref_hll.union(q_hll); ref_hll.len()):
 * out_zero[r * n_qry + q] = #{i : max(a_i, b_i) == 0},  out_sum[r * n_qry + q] = sum_i 2^-max(a_i, b_i).
 * Images are HLL sketches of precision p as written by `save` (33-byte header + 2^p registers).
 * (p >= 10 runs through per-threshold bitmaps and popcounts, which needs the range of register values first: the device entry
 * synchronizes the stream once for that 8-byte read-back.) */
int lash_hll_pair_union_stats_device(lash_ctx *ctx, int p, const uint8_t *d_ref_images, uint32_t n_ref,
                                     const uint8_t *d_qry_images, uint32_t n_qry, uint32_t *d_out_zero, double *d_out_sum);
int lash_hll_pair_union_stats(lash_ctx *ctx, int p, const uint8_t *ref_images, uint32_t n_ref, const uint8_t *qry_images,
                              uint32_t n_qry, uint32_t *out_zero, double *out_sum);

/* dist side, UltraLogLog (/root/reference/src/utils.rs:186-288): for every (reference, query) pair the distinct-count
 * estimate of the merged sketch — UltraLogLog::merge (utils.rs:260-262) followed by get_distinct_count_estimate()
 * (estimator LASH_ULL_FGRA) or MaximumLikelihoodEstimator.estimate() (LASH_ULL_ML), utils.rs:265-269:
 * out_est[r * n_qry + q].  Images are ULL sketches of precision p as written by `save`. */
#define LASH_ULL_FGRA 0
#define LASH_ULL_ML   1
int lash_ull_pair_union_estimates_device(lash_ctx *ctx, int p, int estimator, const uint8_t *d_ref_images, uint32_t n_ref,
                                         const uint8_t *d_qry_images, uint32_t n_qry, double *d_out_est);
int lash_ull_pair_union_estimates(lash_ctx *ctx, int p, int estimator, const uint8_t *ref_images, uint32_t n_ref,
                                  const uint8_t *qry_images, uint32_t n_qry, double *out_est);
/* The same two estimators for ONE sketch (utils.rs:213-217), host only: `registers` = the 2^p state bytes (no header). */
double lash_ull_estimate(const uint8_t *registers, int p, int estimator);

/* ---- dist side, host arithmetic (no GPU): what is O(sketches) or O(pairs) in utils.rs:84-373 ------------------------------
 * HLL++ empirical bias tables.  streaming_algorithms' `len()` (utils.rs:315, 355-363) subtracts, when its raw estimate is
 * <= 5 * 2^p, the mean bias of the 6 nearest samples of a per-precision table of Monte-Carlo measurements (Heule et al. 2013,
 * appendix; the crate carries them as constants).  The numbers are neither in /root/reference nor derivable, so they are NOT
 * in this library: lash_hll_bias_load reads them from a text file ('#' comments; "p <p> <n>" then n lines "<raw> <bias>", any
 * subset of p = 4..18) that tools/ref_probe/extract_hll_bias.py writes from the crate's source.  With `tables` NULL (or no
 * table for p) that regime returns LASH_ERANGE instead of a different estimate. */
typedef struct lash_hll_bias lash_hll_bias;
int  lash_hll_bias_load(const char *path, lash_hll_bias **out);               /* LASH_EINVAL: cannot open; LASH_EFORMAT: malformed */
int  lash_hll_bias_from_arrays(lash_hll_bias **inout, int p, const double *raw, const double *bias, uint32_t n);  /* *inout NULL: created */
int  lash_hll_bias_has(const lash_hll_bias *tables, int p);
void lash_hll_bias_free(lash_hll_bias *tables);
/* Per-sketch cardinalities from the register bytes (no header): hyperminhash's LogLog-beta (`cardinality()`, utils.rs:170-173),
 * streaming_algorithms' `len()` (utils.rs:315), lash_ull_estimate above. */
double lash_hmh_cardinality(const uint8_t *registers, int big_endian);
int    lash_hll_cardinality(const uint8_t *registers, int p, const lash_hll_bias *tables, double *out);
/* The distance the reference prints for every pair of an [n_ref x n_qry] block, from the GPU's pair statistics and the
 * per-sketch cardinalities: similarity (hmh: C, N + expected-collision correction, utils.rs:164; hll: len() of the union from
 * zero / sum, utils.rs:355-362; ull: the union estimate, utils.rs:272) -> .max(0) -> 2s/(1+s) -> model 1: min(-ln(f)/k, 1),
 * model 0: 1 - f^(1/k) (main.rs:415-423), in f32 arithmetic when fp32 (main.rs --fp32).  The caller applies the
 * "same name -> 0" rule (main.rs:452-453).  hmh: c_or_zero = C, n_counts = N; hll: c_or_zero = zero, sum_or_union = sum;
 * ull: sum_or_union = union estimates.  LASH_ERANGE: a union fell into the HLL bias-table regime and `tables` does not cover
 * it (*bad_pair = its index). */
int    lash_dist_rows(int algo, int p, int k, int model, int fp32, uint32_t n_ref, uint32_t n_qry, const double *ref_card,
                      const double *qry_card, const uint32_t *c_or_zero, const uint32_t *n_counts, const double *sum_or_union,
                      const lash_hll_bias *tables, const double *hmh_ec, double *out_dist, uint64_t *bad_pair);
/* hmh_ec (HyperMinHash only, may be NULL): hyperminhash's expected_collisions(n, m) per pair, [n_ref x n_qry], as
 * lash_hmh_pair_expected_collisions / lash_sketch_set_hmh_expected_collisions compute them on the GPU; READ ONLY for pairs whose two
 * cardinalities are both <= 2^19 (every other pair is a closed form, taken here).  NULL: those pairs are computed here on the host,
 * a walk over 65 536 cells with four pow() each (4 ms to 0.2 s PER PAIR, as in the crate): viruses, plasmids, short contigs. */

/* ---- dist side, resident form: all-vs-all on whole collections (BASELINE configs[3]: 10^5 sketches, 5 * 10^9 printed pairs) ----
 * `lash dist` keeps both sketch files in memory for the run (utils.rs:95-127, 202-242, 303-337), takes one cardinality per
 * sketch (utils.rs:170-173, 213-219, 314-315) and walks reference rows x query columns — the lower triangle only when both are
 * the same files (utils.rs:150-180).  A lash_sketch_set is the device-side counterpart: N serialized sketches resident in HBM,
 * uploaded (or adopted from device memory, e.g. the result of an all-gather) ONCE, plus what the pair kernels derive from them
 * once: HyperMinHash register bit planes, HyperLogLog threshold bitmaps.
 *   create          images: n_images serialized sketches of (algo, p) in the context's layout, host memory; member i of the set is
 *                   images[order[i]] (order NULL: member i = image i, n == n_images) — `order` is how a caller puts the set into
 *                   the reference's hash-map key order so that the printed triangle is the set's lower triangle
 *   create_device   adopts d_images (n sketches, set order) without copying; they must outlive the set
 *   cardinalities   per-member distinct-count estimates, the same numbers as lash_hmh_cardinality / lash_hll_cardinality /
 *                   lash_ull_estimate give for each image: register histograms on the GPU, O(histogram) finish on the host.
 *                   LASH_ERANGE: an HLL member fell into the bias-table regime and `tables` does not cover it (*bad_index)
 *   prepare         builds what pair_block needs for this (reference, query) combination (may be the same set); call it once,
 *                   from one thread, before the first pair_block of the combination.  Not needed for correctness: without it
 *                   pair_block falls back to the kernels that read the images themselves (several times slower)
 *   pair_block      statistics of set rows [r0, r1) of `ref` against columns [0, n_cols) of `qry`, row-major [r1 - r0][n_cols]:
 *                   hmh: out_c_or_zero = C, out_n = N (lash_hmh_pair_counts); hll: out_c_or_zero = zero, out_sum_or_union = sum
 *                   (lash_hll_pair_union_stats); ull: out_sum_or_union = the union estimate (lash_ull_pair_union_estimates).
 *                   triangle != 0 (the rows and the columns are the same names in the same order, normally the same set): only
 *                   entries with column <= r0 + row are defined — tiles wholly above the diagonal are skipped
 *                   (utils.rs:158-160).  A caller that wants the triangle passes n_cols = r1.
 *                   Sets are read-only here: several contexts (host threads) of the same device may call pair_block on the
 *                   same prepared sets concurrently.  _device: outputs in device memory, asynchronous on the context's stream. */
typedef struct lash_sketch_set lash_sketch_set;
int      lash_sketch_set_create(lash_ctx *ctx, int algo, int p, const uint8_t *images, uint32_t n_images, const uint32_t *order, uint32_t n,
                                lash_sketch_set **out);
int      lash_sketch_set_create_device(lash_ctx *ctx, int algo, int p, const uint8_t *d_images, uint32_t n, lash_sketch_set **out);
void     lash_sketch_set_free(lash_ctx *ctx, lash_sketch_set *set);
uint32_t lash_sketch_set_size(const lash_sketch_set *set);
int      lash_sketch_set_cardinalities(lash_ctx *ctx, lash_sketch_set *set, int ull_estimator, const lash_hll_bias *tables,
                                       double *out_card, uint32_t *bad_index);
int      lash_sketch_set_prepare(lash_ctx *ctx, lash_sketch_set *ref, lash_sketch_set *qry);
int      lash_sketch_set_pair_block(lash_ctx *ctx, const lash_sketch_set *ref, uint32_t r0, uint32_t r1, const lash_sketch_set *qry,
                                    uint32_t n_cols, int triangle, int ull_estimator, uint32_t *out_c_or_zero, uint32_t *out_n,
                                    double *out_sum_or_union);
int      lash_sketch_set_pair_block_device(lash_ctx *ctx, const lash_sketch_set *ref, uint32_t r0, uint32_t r1, const lash_sketch_set *qry,
                                           uint32_t n_cols, int triangle, int ull_estimator, uint32_t *d_c_or_zero, uint32_t *d_n,
                                           double *d_sum_or_union);

/* HyperMinHash sets: hyperminhash's expected_collisions(n, m) for the pairs of a block in which BOTH sketches hold at most 2^19
 * distinct k-mers (the regime in which the crate walks 65 536 cells per pair; lash_hmh_pair_expected_collisions below):
 * out_ec[(r - r0) * n_cols + c] for exactly those pairs — the other entries are left untouched, lash_dist_rows derives theirs in
 * O(1) — and *n_small_pairs says how many there were (0: out_ec was not touched and may be NULL).  Both sets must have had
 * lash_sketch_set_cardinalities called (the set keeps them); lash_sketch_set_prepare builds the query side's cell vectors once. */
int      lash_sketch_set_hmh_expected_collisions(lash_ctx *ctx, const lash_sketch_set *ref, uint32_t r0, uint32_t r1, const lash_sketch_set *qry,
                                                 uint32_t n_cols, double *out_ec, uint64_t *n_small_pairs);

/* hyperminhash's expected_collisions(n, m) for every pair of an [n_ref x n_qry] block from the per-sketch cardinalities (host
 * arrays in, host array out).  Above 2^19 (either sketch) the crate's closed form; below, the 65 536-cell sum as a product of
 * per-sketch cell-probability vectors: one vector kernel per sketch, one f64 MFMA matrix product per block (dist_kernels.hip).
 * Summation order differs from the crate's loop: agreement ~1e-13 relative.  The query vectors are kept on the device while
 * consecutive calls pass the same qry_card values (row blocks of one `dist` run). */
int    lash_hmh_pair_expected_collisions(lash_ctx *ctx, const double *ref_card, uint32_t n_ref, const double *qry_card, uint32_t n_qry,
                                         double *out_ec);

/* Synthetic genomes of SURVEY.md §8(d) generated in HBM (bench / tests): genome ids first..first+n-1,
 * n_bases ASCII bytes each, written back to back at d_out. */
int lash_synth_genomes_device(lash_ctx *ctx, uint64_t first_genome, uint32_t n_genomes, uint64_t n_bases, uint8_t *d_out);

#ifdef __cplusplus
}
#endif
#endif
