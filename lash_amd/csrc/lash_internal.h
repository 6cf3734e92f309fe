// lash_internal.h — what the host-side translation units of liblash_gfx950.so share (round 6: lash_api.hip was 2 400 lines; it is now
//   lash_api.hip         library / context / layout entries, the sketch entries (record batches, packed genomes, raw files), the strict FASTX parse
//   lash_plan.hip        what a sketch call queues: pack, persistent small-genome launch, work-item planning, binned / global tables, amino acids
//   lash_hll_replay.hip  HyperLogLog's incremental `sum` in its order-dependent corner, replayed from prefix sketches
//   lash_dist_api.hip    merge of serialized sketches and the pair-statistics entries of `dist`
//   sketch_set.hip       resident sketch sets).
// Everything here lives in namespace lashi (lash_ctx.h) and is called only from inside the library.
#pragma once
#include <utility>
#include <vector>

#include "lash_ctx.h"

namespace lashi {

// ---- lash_plan.hip --------------------------------------------------------------------------------------------------------------
int pack_into(lash_ctx *ctx, lash_packed *pk, hipStream_t stream, EvSet *ev, const uint8_t *d_seq, const uint8_t *d_seq_end,
              const uint64_t *d_rec_off, uint64_t n_rec, const uint64_t *genome_rec_off, const uint64_t *genome_byte_off,
              uint32_t n_genomes, const uint8_t *formats = nullptr, bool direct = false);
int probe_dirty(lash_ctx *ctx, lash_packed *pk, hipStream_t stream);
uint64_t sole_max_bytes(const lash_ctx *ctx, const lash_params *prm, const SolePlan &sp);
int sole_run(lash_ctx *ctx, const lash_params *prm, const SolePlan &sp, uint64_t max_len, const uint8_t *d_seq, uint64_t seq_bytes,
             const uint64_t *d_rec_off, uint64_t n_rec, bool any_multi, bool rec_identity, const uint64_t *genome_byte_off, const lash_packed *pk,
             uint32_t n_genomes, uint8_t *d_out_images, uint32_t *per_genome_ndel);
int sketch_from(lash_ctx *ctx, const lash_params *prm, const lash_packed *pk, uint8_t *d_out_images, EvSet *ev, bool allow_bins = true);
int sketch_aa(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_seq, const uint64_t *d_rec_off, uint64_t n_rec,
              const uint64_t *genome_rec_off, const uint64_t *genome_byte_off, uint32_t n_genomes, uint8_t *d_out_images, bool allow_bins = true);

// ---- lash_hll_replay.hip --------------------------------------------------------------------------------------------------------
double grid_sum(const uint8_t *regs, size_t m, int p);
int hll_sum_field_offset(const lash_layout &lay);
int hll_replay_one(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_seq, const std::vector<uint64_t> &rec, const uint8_t *fin,
                   const uint8_t *base, double &S, double &G, bool &carry);
int hll_replay_sums(lash_ctx *ctx, const lash_params *prm0, const uint8_t *d_seq, const uint64_t *d_rec_off, const uint64_t *h_rec_off,
                    const uint64_t *genome_rec_off, uint8_t *d_images, uint8_t *h_images, const std::vector<uint32_t> &flagged,
                    std::vector<uint32_t> &left);
std::vector<uint32_t> hll_flagged(lash_ctx *ctx);

// ---- lash_api.hip ---------------------------------------------------------------------------------------------------------------
// needletail's record rules for uncompressed input as lash uses it (utils.rs:453-459), on the host: the exact path for the rare file the device
// parse flags, and the streamed replay's
size_t parse_fastx_strict(const uint8_t *d, size_t n, std::vector<uint8_t> *seq, std::vector<uint64_t> *rec_off, bool skip_bad = false,
                          std::vector<std::pair<size_t, size_t>> *bad = nullptr);

}  // namespace lashi
