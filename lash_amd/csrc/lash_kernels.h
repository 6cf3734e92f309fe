// lash_kernels.h — host-callable launchers of the gfx950 kernels (implemented in *.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lash_common.h"

namespace lash {

// ---- sketch stage ----------------------------------------------------------------------------------------
struct BinGenome { uint64_t list_off; uint32_t cap; uint32_t pad; };   // binned launches: this genome's lists, bin b = lists + list_off + b * cap
struct SketchArgs {
    const uint32_t   *words;      // packed 2-bit bases
    const uint32_t   *brk;        // record-break bitmap
    const uint32_t   *zero_words; // >= 4 zero words (stand-in bitmap for single-record genomes)
    const GenomeDesc *genomes;
    const uint64_t   *nvalid;     // surviving bases per genome
    const WorkItem   *items;
    const uint32_t   *item_order; // NULL, or a permutation of the item indices: workgroup b takes item item_order[b] (longest first)
    uint8_t          *partials;   // [n_items][partial_stride] partial sketches in image register format
    uint32_t         *gregs;      // [n_items][nreg32] zeroed u32 words, only for the global-register variant
    uint32_t         *item_kmers; // [n_items] valid k-mers of each work item (summed per genome by finalize_kernel)
    uint8_t          *images;     // ITEM_SOLE work items write their genome's image themselves
    uint64_t          image_bytes;
    uint64_t          alpha_bits; // HLL alpha as f64 bits
    int               accumulate;
    // direct mode (sketch_kernel<..., DIRECT>): format-0 genomes are read as ASCII straight from the caller's buffer
    const uint8_t    *seq;        // the caller's record bytes
    const uint32_t   *brk_bytes;  // record-break bitmap in BYTE positions (== base positions while nothing is deleted)
    const uint8_t    *safe;       // >= 128 readable bytes: load target of lanes that are not on the fast path
    uint32_t         *dirty;      // [n_genomes + 1] per genome: the direct pass hands this genome to stream_sketch_kernel (much dense
                                  // dirt, or a gap beyond its look-ahead scan); the stream launch runs only these
    uint32_t         *nslow;      // [n_genomes] waves that found their part of the genome too dirty for the direct pass (hand-over at GenomeDesc::handover)
    uint32_t         *ndel;       // [n_genomes] bytes the direct pass deleted in place (surviving bases = nvalid - ndel while !dirty)
    uint32_t         *ndel2;      // [n_genomes] bytes stream_sketch_kernel deleted in the genomes it took over (zeroed)
    const uint64_t   *rec_off;    // direct mode: the caller's record offsets (uniform-length genomes derive their record starts from them)
    const uint32_t   *nonuniform; // direct mode: [n_genomes], 0 = all records of the genome have the same length (rec_uniform_kernel):
                                  // its record starts are the multiples of that length, no bitmap is made or read for it
    uint32_t         *hll_corner; // NULL or [n_genomes], see FinalizeArgs (genomes whose one work item writes the image itself)
    uint64_t          bitflip;    // xxh3 seed-folded constant (64- or 128-bit variant by algo)
    LayoutDev         lay;
    uint32_t          partial_stride;
    uint32_t          stage_off;  // direct mode: LDS byte offset of the waves' staging areas (dense_tile), after registers + census
    uint32_t          stage_stride; // bytes of one wave's staging area (dense_tile's 1 600, or the lanes' stacks of a deferring launch if larger)
    uint32_t          sigq_depth; // deferring launches: words of one lane's stack (odd: the lanes' slots fall into distinct banks)
    uint32_t          nreg32;     // u32 words of register state (HMH 16384, HLL 2^p, ULL 2*2^p)
    int               k;
    int               p;
    uint32_t          item_base;  // launches over a range of the items: workgroup b takes item item_base + b (item_order == NULL)
    unsigned long long *item_trace;  // diagnostic (LASH_ITEM_TRACE=file, tools/item_trace.py): NULL, or [n_items][4] — the workgroup's first and
                                     // last instruction on the 100 MHz wall clock, its XCC / CU / SIMD ids, 0.  Read by the kernels only in a library
                                     // built with -DLASH_ITEM_TRACE_BUILD (tools/build_trace_lib.sh); the shipped kernels ignore it
    // binned launches (SketchPlan::bins; sketch_kernels.hip "BinRegs"): the sketch kernels append entries, bins_apply_kernel builds registers
    uint32_t         *bin_lists;  // every (genome of the group, bin) list, BinGenome::list_off apart
    uint32_t         *bin_cnt;    // [genomes of the group][bins] fill counters, zeroed
    uint32_t         *bin_slab;   // [genomes of the group][bin_slab_words] full-size fallback tables (zero / 0xFF-filled)
    uint32_t         *bin_spill;  // [genomes of the group][bins] set when that bin's part of the genome's fallback table holds something
    const BinGenome  *bin_genomes;   // [genomes of the group]
    uint32_t          bins, bin_shift, bin_S, bin_sub_shift;
    uint32_t          bin_flush_words;              // the word loop empties the staging rows after every 1 / 2 / 4 words of 16 k-mers per lane
    uint32_t          bin_lds_off, bin_wave_bytes;   // LDS: the waves' counter + staging areas
    uint32_t          bin_slab_words;
    uint32_t          bin_genome0;                  // first genome of the group
};

struct SketchPlan {
    int      algo, k, p;
    bool     variant;             // the register rule's compile-time variant (rule_variant(), lash_ctx.h): HyperMinHash x = low half of
                                  // xxh3_128 (layout.hmh_x_low / LASH_F_HMH_X_LOW), HyperLogLog bucket = top p bits (layout.hll_bucket_high)
    bool     use_lds;
    uint32_t threads;             // 512 (<= 64 KiB of LDS, two workgroups per CU) or 1024
    uint32_t lds_bytes;
    uint32_t nreg32;
    uint32_t parts_log2;          // > 0: the bucket space is covered in 2^parts_log2 passes, nreg32 / lds_bytes are per pass
    uint32_t partial_bytes;       // bytes of one partial sketch (register array only)
    uint32_t partial_stride;      // rounded up to 16
    bool     defer = false;       // direct HyperMinHash launch with deferred signatures (the caller sets it for batches of large work items)
    uint32_t sigq_depth = 7;      // ... and the depth of its lanes' stacks (process_word_defer)
    // register tables beyond 128 KiB of LDS (HLL p = 16, ULL p = 15 .. 22): hash once, scatter entries into 2^bins_log2 bins per genome,
    // one LDS pass per bin (BinRegs / bins_apply_kernel).  use_lds stays true (no per-item global table), lds_bytes holds no table.
    bool     bins = false;
    bool     bytes = false;       // byte registers in LDS with compare-and-swap updates (LdsByteRegs): hll p = 16, ull p = 15 .. 17
    uint32_t bins_log2 = 0, bin_shift = 0, bin_S = 0, bin_sub_shift = 0, bin_flush_words = 1;
};
// per wave: bytes of LDS a binned launch needs for its bin counters and staging rows
uint32_t sketch_bin_wave_bytes(const SketchPlan &plan);
struct BinApplyArgs {
    const uint32_t *lists, *cnt;
    uint32_t *spill;                // one flag per (genome, bin): its part of the fallback table holds something
    uint32_t *slab;                 // (a bin's part is read beside the LDS table and wiped when its flag is up)
    const BinGenome *genomes;
    uint8_t  *partials;           // one partial per GENOME of the call: genome gi of the group at partials + (genome0 + gi) * partial_stride
    uint32_t *item_kmers;         // ... and its k-mer count at item_kmers[virt0 + gi] = the sum over its real items
    const uint32_t *genome_item_begin;   // real items of genome g: [genome_item_begin[g], genome_item_begin[g + 1])
    const WorkItem *items;               // ... of which those that begin beyond the genome's surviving bases never ran (as in finalize_kernel)
    const uint64_t *nvalid;              // NULL: every item ran
    int       k;
    uint64_t  partial_stride;
    uint32_t  virt0, genome0;
    uint32_t  bins, bin_shift, slab_words;
    int       algo, p;
    // UltraLogLog, not accumulating (round 6): the registers leave straight into the caller's images — header by the genome's first bin, k-mer census
    // into kmer_counter — and no finalize launch follows (at p = 22 it read and wrote 4 MiB per genome only to move them: 4.1 of 33 ms)
    uint8_t  *images;             // NULL: into the partials, finalize_kernel does the rest
    uint64_t  image_bytes;
    uint64_t  hdr_tpl;
    uint32_t  hdr_bytes;
    unsigned long long *kmer_counter;
};
hipError_t launch_bins_apply(const BinApplyArgs &args, uint32_t n_group_genomes, hipStream_t stream);

// small_items: the batch's genomes average under ~100 kbp (workgroup shape for small register tables, see the .hip)
SketchPlan make_sketch_plan(int algo, int k, int p, bool variant, bool small_items = false, bool allow_bins = true);
// the genomes flagged in args.dirty, again from their ASCII bytes, compacted through an LDS ring per wave (stream_sketch_kernel)
hipError_t launch_sketch_stream(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream);
// direct launches: bytes of LDS the waves' staging areas take on top of plan.lds_bytes (they start at plan.lds_bytes)
uint32_t sketch_direct_stage_bytes(const SketchPlan &plan);
hipError_t launch_sketch(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream,
                         bool direct = false);
// ---- whole small genomes on persistent workgroups (sole_kernels.hip, round 5) -------------------------------------------------
// One sketch per file whatever its size (utils.rs:450-509): a collection of viruses, plasmids, amplicons or contigs is very many
// genomes of a few kbp.  sketch_kernel gives each a workgroup of its own, whose fixed cost — three dependent loads to find its
// bytes, a 64-byte-per-lane tile that a 10 kbp genome fills with 2.5 waves, the flush — is 20 us against 1..7 us of hashing.
// sole_sketch_kernel keeps workgroups resident instead: each takes CHUNKS of consecutive genomes (planned on the host from the
// genome byte offsets alone: no GenomeDesc, no work items) and streams their bytes through an LDS ring of 2-bit bases — every
// byte converted once, deleted bytes dropped on the way in (filter_out_n, utils.rs:33-41), record starts as bits beside the
// bases — from which every lane hashes ONE word (16 k-mer starts) per round; the next round's bytes (the next genome's, when this
// one ends) are in flight meanwhile, and the image leaves LDS in 16-byte stores.
struct SoleArgs {
    // ASCII source (PACKED = false)
    const uint8_t  *seq;              // the caller's record bytes
    uint64_t        seq_bytes;        // readable bytes of seq
    const uint64_t *genome_byte_off;  // [n_genomes + 1] device copy: genome g = bytes [genome_byte_off[g], genome_byte_off[g + 1])
    const uint32_t *brk_abs;          // NULL (every genome has one record), or bit b set <=> a record starts at byte b of seq (sole_mark_kernel)
    const uint8_t  *safe;             // >= 16 readable bytes: load target of the lanes whose 16 bytes run past seq
    uint32_t       *ndel;             // NULL, or [n_genomes]: bytes deleted from each genome sketched here (calls that also run the sliced launch
                                      // keep their census per genome: lash_timing::bases_last)
    // packed source (PACKED = true): the pack stage's 2-bit stream, break bitmap, descriptors and surviving-base counts
    const uint32_t   *words;
    const uint32_t   *brk;
    const GenomeDesc *genomes;
    const uint64_t   *nvalid;
    // work
    const uint32_t *chunk_begin;      // [n_chunks + 1] genomes [chunk_begin[c], chunk_begin[c + 1]) form chunk c
    uint32_t        n_chunks;
    uint32_t       *ticket;           // zeroed: chunks beyond the first gridDim.x are handed out through it
    uint64_t        max_len;          // genomes longer than this many bytes (packed: words * 16) belong to the sliced launch
    // out
    uint8_t        *images;
    uint64_t        image_bytes;
    uint64_t        alpha_bits;       // HLL alpha as f64 bits
    int             accumulate;
    uint32_t       *hll_corner;       // NULL or [n_genomes], see FinalizeArgs
    unsigned long long *wg_counts;    // [gridDim.x][2]: valid k-mers, surviving bases of the genomes this workgroup sketched
    uint64_t        bitflip;
    LayoutDev       lay;
    uint32_t        nreg32;           // table words (HMH 16384, HLL 2^p, ULL 2 * 2^p)
    int             k, p;
    // LDS layout (byte offsets; the table starts at 0)
    uint32_t        hist_off;         // 80 words: HyperLogLog header histogram
    uint32_t        scan_off;         // 2 x 8 words: the waves' survivor counts of a round (double-buffered)
    uint32_t        ring_off;         // ring_words words of 2-bit bases
    uint32_t        brk_off;          // ring_words / 2 words of record-start bits
    uint32_t        ring_words;       // a power of two, >= 4 * threads
    uint32_t        lds_words;        // everything, for the initial clear
};
struct SolePlan {
    bool     ok;                      // the sketch type has a table the persistent kernel holds (<= 64 KiB of LDS)
    uint32_t threads, lds_bytes, wg_per_cu;
    uint32_t hist_off, scan_off, ring_off, brk_off, ring_words;
};
SolePlan make_sole_plan(int algo, int p, uint32_t n_genomes = 0, uint32_t cu_count = 256);   // n_genomes: of the call, 0 = unknown
hipError_t launch_sole(const SolePlan &plan, int algo, int k, bool variant, bool packed, const SoleArgs &args, uint32_t n_wg, hipStream_t stream);
// workgroups of that launch's kernel variant that one CU holds at a time (hipOccupancyMaxActiveBlocksPerMultiprocessor): the launch is
// sized to what is resident
hipError_t sole_resident_per_cu(const SolePlan &plan, int algo, int k, bool variant, bool packed, uint32_t *out);
// bit b of brk_abs (zeroed, (seq_bytes + 63) / 32 + 2 words) set <=> some record starts at byte b
hipError_t launch_sole_mark(const uint64_t *rec_off, uint64_t n_rec, uint64_t seq_bytes, uint32_t *brk_abs, hipStream_t stream);
// wg_counts -> the context's k-mer census (counter[0]) and surviving-base count (counter[1])
hipError_t launch_sole_census(const unsigned long long *wg_counts, uint32_t n_wg, unsigned long long *counter, unsigned long long *bases, uint32_t *ticket,
                              hipStream_t stream);                          // (and the chunk ticket back to zero for the next launch)

// amino-acid sketches (LASH_F_AMINO; utils.rs:511-563): work items are RECORD ranges of a genome (WorkItem::word_begin / word_end =
// record indices relative to the genome's first), a lane walks one record at a time; args.seq / args.rec_off = the caller's bytes
hipError_t launch_sketch_aa(const SketchPlan &plan, const SketchArgs &args, uint32_t n_items, hipStream_t stream);
constexpr uint32_t AA_RECORDS_PER_ITEM = 4096;
// nonuniform[g] (zeroed before the launch) != 0 <=> the records of multi-record genome g differ in length
hipError_t launch_rec_uniform(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes, uint64_t n_rec, uint32_t *nonuniform,
                              hipStream_t stream);
// record starts of the multi-record format-0 genomes whose records differ in length -> args.brk_bytes; every word of such a
// genome's bitmap is written (no memset needed), genomes with nonuniform[g] == 0 are left alone
// (genomes whose records average a KiB or more — assemblies of contigs — have their bitmaps zeroed by every thread and the few record
// starts OR-ed in; read sets are written from their head records: decided per genome on the device)
hipError_t launch_brk_bytes(const GenomeDesc *genomes, const uint64_t *rec_off, uint32_t n_genomes, uint64_t n_rec, const uint32_t *nonuniform,
                            uint32_t *brk_bytes, hipStream_t stream);
// fastq_check.hip: the FASTQ rule the pack kernel's line-structure check cannot see — a quality line as long as its sequence
// line, a file that ends on a whole record (needletail's Err; /root/reference/src/utils.rs:453-459 stops there).
struct FqFile {
    uint64_t off;        // first byte of the file in the raw buffer
    uint64_t len;        // bytes (< 4 GiB)
    uint32_t block0;     // index of the file's first 4 KiB block among all FASTQ blocks of the call
    uint32_t n_blocks;
    uint32_t index;      // which file_err word to set
    uint32_t pad;
};
hipError_t launch_fastq_check(const uint8_t *d_raw, const FqFile *d_files, uint32_t n_files, uint32_t n_blocks, uint32_t *d_scratch,
                              uint32_t *d_file_err, hipStream_t stream);
size_t fastq_check_scratch_words(uint32_t n_files, uint32_t n_blocks);
uint32_t fastq_check_block_bytes();

struct FinalizeArgs {
    const uint8_t  *partials;
    const WorkItem *items;
    const uint32_t *genome_item_begin;   // n_genomes + 1
    const uint64_t *nvalid;              // NULL => every item is live (merge of images)
    const uint32_t *item_kmers;          // NULL or per-item valid k-mer counts ...
    unsigned long long *kmer_counter;    // ... whose per-genome sums are added here
    uint8_t        *images;
    uint64_t        partial_stride;
    uint64_t        partial_base_off;    // offset of the register array inside each partial (merge: header size)
    uint64_t        image_bytes;
    uint64_t        alpha_bits;          // HLL alpha as f64 bits
    int             algo, p, k;
    int             accumulate;          // union into the registers already in images[]
    uint32_t        parts_log2;          // see SketchPlan: a partial holds only the registers of its item's pass
    LayoutDev       lay;
    uint32_t       *hll_corner;          // NULL, or [n_genomes] zeroed: set when a genome holds a HyperLogLog register > 53 - p
                                         // (write_hll_header, sketch_kernels.hip)
    int             src_images;          // the "partials" are images (lash_merge_images): HMH registers in image byte order
    uint32_t        group;               // 0, or G: launch_reduce_groups() has folded every G consecutive slices (per
                                         // pass) into the first one's partial; only those group heads are read
    const GenomeDesc *descs;             // NULL, or: genomes of at most skip_max_len bytes are not this launch's (the persistent
    uint64_t        skip_max_len;        // small-genome kernel writes their images: sole_kernels.hip)
};
// Many slices per genome (one metagenome-sized input: thousands of work items, one finalize workgroup): fold every R
// consecutive slices into the first one's partial, in place, with one workgroup per (genome, pass, group).
hipError_t launch_reduce_groups(const FinalizeArgs &args, uint32_t n_genomes, uint32_t max_slices, hipStream_t stream);
hipError_t launch_finalize(const FinalizeArgs &args, uint32_t n_genomes, hipStream_t stream);
hipError_t launch_census(const FinalizeArgs &args, uint32_t n_genomes, hipStream_t stream);   // all genomes ITEM_SOLE

// ---- pack stage -------------------------------------------------------------------------------------------
struct PackArgs {
    const uint8_t    *seq;
    const uint8_t    *seq_end;    // one past the caller's buffer (edge loads are bounds-checked)
    const uint64_t   *rec_off;
    const GenomeDesc *genomes;
    uint32_t         *words;
    uint32_t         *brk;        // zeroed before the launch
    uint64_t         *nvalid;
    uint32_t          code_tab4;  // LayoutDev::code_tab4
    uint32_t         *file_err;   // NULL, or [n_genomes] zeroed: raw FASTQ files whose line structure broke (pack_kernels.hip)
};
// one 16 KiB tile of one genome (filled by pack_map_kernel)
struct TileInfo {
    int64_t  toff;      // first byte of the tile, relative to seq (16-byte aligned address; may be < byte_off)
    uint64_t r0;        // first record starting at or after toff
    uint32_t nrec;      // records starting inside the tile
    uint32_t g;         // genome
    uint32_t tb;        // first tile of that genome (look-back stops there)
    int32_t  rel_lo;    // genome bytes occupy [rel_lo, rel_hi) of the tile
    int32_t  rel_hi;
    uint32_t flags;     // TF_FIRST | TF_LAST | TF_FULL
    uint32_t pad;
    uint64_t word_off;  // the genome's offsets into words[] / brk[] (copied from GenomeDesc: saves a dependent load)
    uint64_t brk_off;
};
static_assert(sizeof(TileInfo) == 64, "TileInfo is one 64-byte line");
struct PackV2Args {
    const TileInfo *tiles;           // n_tiles
    uint64_t       *desc;            // n_tiles look-back descriptors, zeroed before the launch
    uint32_t       *desc2;           // n_tiles line-state descriptors (raw FASTA/FASTQ genomes), zeroed
    uint32_t       *ticket;          // PACK_TICKET_SHARDS counters at a 128-byte stride, zeroed before the launch
    uint32_t       *error_flag;      // zeroed; != 0 after the launch means a look-back spin hit its bound
    uint32_t        n_tiles;         // upper bound when n_tiles_dev is set
    uint32_t        n_shards;        // set by launch_pack_v2
    const uint32_t *n_tiles_dev;     // NULL, or the device-side tile count (pack only the genomes flagged dirty)
};
constexpr uint32_t PACK_TICKET_SHARDS = 16;
struct PackMapArgs {
    const uint8_t    *seq;
    const uint64_t   *rec_off;
    const GenomeDesc *genomes;
    const uint32_t   *tile_begin;    // n_genomes + 1: first tile of each genome
    TileInfo         *tiles;
    uint32_t          n_tiles, n_genomes;
    const uint32_t   *n_tiles_dev;   // see PackV2Args
};
// tile_begin_c[g] = tiles of the dirty genomes before g; n_tiles_c = their total (one workgroup, device-side, so the
// host never waits to learn which genomes the direct sketch pass gave up on)
hipError_t launch_dirty_tile_scan(const uint32_t *tile_begin, const uint32_t *dirty, uint32_t n_genomes,
                                  uint32_t *tile_begin_c, uint32_t *n_tiles_c, hipStream_t stream);
uint32_t   pack_v2_tile_bytes();
hipError_t launch_pack_v2(const PackArgs &args, const PackV2Args &v, const PackMapArgs &m, uint32_t cu_count, bool any_raw,
                          hipStream_t stream);

// ---- dist side (HyperMinHash pair statistics) ------------------------------------------------------------------
// images: `hdr` header bytes, then the registers; consecutive images are `stride` bytes apart
// tri (every pair launcher): -1, or the set index of the call's first row when the reference rows and the query columns come
// from the SAME set in the same order (utils.rs:158-160: only pairs with column <= row are printed): tiles wholly above
// the diagonal return at once and leave their outputs unwritten
hipError_t launch_hmh_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, uint32_t hdr,
                            uint64_t stride, uint32_t *d_c, uint32_t *d_n, hipStream_t stream, int64_t tri = -1);
// the same counts through register bit planes (pair_planes.hip).  Row layout T[(w * 17 + b) * ldT + s] (b = 16: the plane of
// non-zero registers) with ldT a multiple of hmh_planes_row_pad(); column layout S[(w * n_pad + s) * 20 + b] with n_pad a multiple
// of hmh_planes_col_pad(); members beyond n read as zero sketches.  Either layout pointer may be NULL (not wanted); buffer sizes
// from hmh_planes_T_words / hmh_planes_S_words.  d_nzcount (NULL, or zeroed [n]) receives each member's number of non-zero registers.
hipError_t launch_hmh_planes(const uint8_t *d_img, uint32_t hdr, uint64_t stride, uint32_t n, uint32_t *d_T, uint32_t ldT, uint32_t *d_S,
                             uint32_t n_pad, uint32_t *d_nzcount, hipStream_t stream);
uint32_t   hmh_planes_col_pad();
uint32_t   hmh_planes_row_pad();
size_t     hmh_planes_T_words(uint32_t ldT);
size_t     hmh_planes_S_words(uint32_t n_pad);
hipError_t launch_hmh_pairs_planes(const uint32_t *d_T, uint32_t ldT, uint32_t row0, uint32_t n_rows, const uint32_t *d_S, uint32_t n_pad,
                                   uint32_t n_cols, bool full, bool triangle, uint32_t *d_c, uint32_t *d_n, uint64_t ld_out, hipStream_t stream);

hipError_t launch_hll_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, int p, uint32_t hdr,
                            uint32_t *d_zero, double *d_sum, hipStream_t stream, int64_t tri = -1);

// the same statistics for p >= 10 through threshold bitmaps (dist_kernels.hip): range of register values -> bm[n][band][m/32] -> pairs
hipError_t launch_hll_minmax(const uint8_t *d_img, uint32_t n, int p, uint32_t hdr, uint32_t *d_lohi, hipStream_t stream);
hipError_t launch_hll_bitmaps(const uint8_t *d_img, uint32_t n, int p, uint32_t hdr, uint32_t lo, uint32_t band, uint32_t *d_bm, hipStream_t stream);
hipError_t launch_hll_pairs_bitmap(const uint32_t *d_bm_ref, uint32_t n_ref, const uint32_t *d_bm_qry, uint32_t n_qry, int p, uint32_t lo,
                                   uint32_t band, uint32_t *d_zero, double *d_sum, hipStream_t stream, int64_t tri = -1);

// estimator: 0 = FGRA, 1 = ML (ull_estimators.h); d_est[r * n_qry + q] = estimated distinct count of the union
hipError_t launch_ull_pairs(const uint8_t *d_ref, uint32_t n_ref, const uint8_t *d_qry, uint32_t n_qry, int p, uint32_t hdr,
                            int estimator, double *d_est, hipStream_t stream, int64_t tri = -1);
// register histograms of n serialized sketches: hist[s][256] (HLL / ULL: register bytes; HyperMinHash: bins 0..63 = the
// registers' leading-zero fields)
hipError_t launch_sketch_hist(const uint8_t *d_img, uint32_t n, int algo, uint32_t n_regs, uint32_t hdr, uint64_t stride, uint32_t hmh_be,
                              uint32_t *d_hist, hipStream_t stream);
// dst row i = src row order[i] (row_bytes each)
hipError_t launch_gather_rows(const uint8_t *d_src, const uint32_t *d_order, uint32_t n, uint64_t row_bytes, uint8_t *d_dst, hipStream_t stream);

// HyperMinHash expected collisions of small sketches (dist_kernels.hip): P[n][65536] cell probabilities of each cardinality,
// X[m][n] = A[m][65536] * B[n][65536]^T
hipError_t launch_collision_vectors(const double *d_card, uint32_t n, double *d_P, hipStream_t stream);
hipError_t launch_collision_gemm(const double *d_A, uint32_t m, const double *d_B, uint32_t n, double *d_X, hipStream_t stream);

// ---- synthetic genomes --------------------------------------------------------------------
hipError_t launch_synth(uint64_t first_genome, uint32_t n_genomes, uint64_t n_bases, uint8_t *d_out, hipStream_t stream);

}  // namespace lash
