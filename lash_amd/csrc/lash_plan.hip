// lash_plan.hip — what a sketch call queues: packing raw bytes, the persistent small-genome launch, work-item planning for the sliced
// kernels (slices, tail split, launch order, deferral), binned / global register tables, the amino-acid branch.  Split out of lash_api.hip
// in round 6; the extern "C" entries that call in are in lash_api.hip.  Reference: the per-file closure of sketch_files
// (/root/reference/src/utils.rs:452-508).
#include "lash_ctx.h"
#include "lash_internal.h"

namespace lashi {


// Packs genomes [0, n_genomes) described by genome_rec_off / genome_byte_off (absolute record indices / byte offsets
// into d_seq) on `stream`.  `ev`, when set, gets its pack-start / pack-end events recorded on that stream.
int pack_into(lash_ctx *ctx, lash_packed *pk, hipStream_t stream, EvSet *ev, const uint8_t *d_seq, const uint8_t *d_seq_end,
              const uint64_t *d_rec_off, uint64_t n_rec, const uint64_t *genome_rec_off, const uint64_t *genome_byte_off,
              uint32_t n_genomes, const uint8_t *formats, bool direct)
{
    // formats == nullptr: record sequences + rec_off table; else per genome LASH_FMT_FASTA / LASH_FMT_FASTQ raw file bytes
    if (n_genomes && (!genome_byte_off || (!formats && !genome_rec_off))) return LASH_EINVAL;
    pk->error_flag = nullptr;
    pk->direct = direct && !formats && n_genomes;
    direct = pk->direct;
    pk->d_seq = d_seq;
    std::vector<GenomeDesc> &descs = pk->h_descs;
    descs.assign(n_genomes, GenomeDesc{});
    pk->byte_len.assign(n_genomes, 0);
    uint64_t wo = 0, bo = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) {
        if (genome_byte_off[g + 1] < genome_byte_off[g]) return LASH_EINVAL;
        if (!formats && (genome_rec_off[g + 1] < genome_rec_off[g] || genome_rec_off[g + 1] > n_rec)) return LASH_EINVAL;
        if (formats && formats[g] != LASH_FMT_FASTA && formats[g] != LASH_FMT_FASTQ) return LASH_EINVAL;
        GenomeDesc &d = descs[g];
        d.byte_off = genome_byte_off[g];
        d.byte_len = genome_byte_off[g + 1] - genome_byte_off[g];
        if (d.byte_len > 0xFFFFFFFFull - 64) return LASH_ELIMIT;
        d.rec_begin = formats ? 0 : genome_rec_off[g];
        d.rec_end = formats ? 0 : genome_rec_off[g + 1];
        d.format = formats ? formats[g] : 0u;
        d.handover = 1;
        d.word_off = wo;
        d.brk_off = bo;
        pk->byte_len[g] = d.byte_len;
        uint64_t nw = (d.byte_len + 15) / 16 + 2 * PAD_WORDS;
        wo += (nw + 3) & ~3ull;                                   // keep every genome 16-byte aligned
        bo += (d.byte_len + 1 + 31) / 32 + 4;                     // +3 words of look-ahead in kmer_valid_mask
    }
    pk->n_genomes = n_genomes;
    pk->total_words = wo + 2 * PAD_WORDS;
    pk->total_brk = bo + 4;
    // tiles of the single-pass pack: genomes are cut at 16-byte-aligned addresses, tiles never straddle genomes
    const uint64_t tile_bytes = pack_v2_tile_bytes();
    std::vector<uint32_t> &tile_begin = pk->h_tile_begin;
    tile_begin.assign(n_genomes + 1, 0);
    uint64_t n_tiles = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) {
        tile_begin[g] = (uint32_t)n_tiles;
        if (descs[g].byte_len) {
            const uint64_t lead = (reinterpret_cast<uintptr_t>(d_seq) + descs[g].byte_off) & 15u;
            n_tiles += (lead + descs[g].byte_len + tile_bytes - 1) / tile_bytes;
        }
        if (n_tiles > 0x7FFFFFFFull) return LASH_ELIMIT;
    }
    tile_begin[n_genomes] = (uint32_t)n_tiles;
    TRACE("pack: tables built");
    int rc;
    // direct mode packs nothing (round 3: the genomes the direct pass gives up are redone from their ASCII bytes by
    // stream_sketch_kernel), so there is no 2-bit stream, no tile table and no look-back state to make room for
    if (!direct) {
        if ((rc = reserve(ctx, pk->words, pk->total_words * 4))) return rc;
        if ((rc = reserve(ctx, pk->brk, pk->total_brk * 4))) return rc;
        if ((rc = reserve(ctx, pk->tiles, (size_t)(n_tiles + 1) * sizeof(TileInfo)))) return rc;
    }
    const size_t lb_bytes = direct ? 0 : (((size_t)(n_tiles + 2) * 12 + PACK_TICKET_SHARDS * 128 + 512) + 15) & ~(size_t)15;
    if ((rc = reserve(ctx, pk->lookback, lb_bytes + (size_t)(5 * (size_t)n_genomes + 2) * 4))) return rc;
    pk->d_dirty = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(pk->lookback.ptr) + lb_bytes);
    if (direct && (rc = reserve(ctx, pk->tile_begin_c, (size_t)(n_genomes + 2) * 4))) return rc;
    if (n_genomes == 0) return LASH_OK;
    TRACE("pack: reserved");
    bool any_multi = false;                                   // single-record genomes never consult the bitmap
    for (uint32_t g = 0; g < n_genomes && !any_multi; ++g)
        any_multi = descs[g].format != 0u || descs[g].rec_end - descs[g].rec_begin > 1;
    pk->any_multi = any_multi;
    pk->d_rec_off = d_rec_off;
    pk->n_rec = n_rec;
    // surviving bases per genome: written by the pack kernel; direct mode starts from "nothing deleted" (= bytes) and
    // the deferred pack overwrites the genomes that turn out dirty
    pk->h_nvalid.assign(n_genomes + 1, 0);
    if (direct) std::copy(pk->byte_len.begin(), pk->byte_len.end(), pk->h_nvalid.begin());
    if (ev) { ev->pack = true; HIPCHK(ctx, hipEventRecord(ev->e[0], stream)); }
    if (!direct) {
        std::vector<Section> sec = {{descs.data(), descs.size() * sizeof(GenomeDesc), 0},
                                    {tile_begin.data(), tile_begin.size() * 4, 0},
                                    {pk->h_nvalid.data(), pk->h_nvalid.size() * 8, 0}};
        const size_t total = layout_sections(sec);
        if ((rc = reserve(ctx, pk->tables, total))) return rc;
        if ((rc = upload_sections(ctx, pk->tables.ptr, sec, total, stream))) return rc;
        uint8_t *tb = static_cast<uint8_t *>(pk->tables.ptr);
        pk->d_descs = reinterpret_cast<GenomeDesc *>(tb + sec[0].off);
        pk->d_tile_begin = reinterpret_cast<uint32_t *>(tb + sec[1].off);
        pk->d_nvalid = reinterpret_cast<uint64_t *>(tb + sec[2].off);
        TRACE("pack: tables uploaded");
        if (any_multi) HIPCHK(ctx, hipMemsetAsync(pk->brk.ptr, 0, pk->total_brk * 4, stream));
        HIPCHK(ctx, hipMemsetAsync(pk->lookback.ptr, 0, lb_bytes + (formats ? (size_t)(5 * (size_t)n_genomes + 2) * 4 : 0), stream));
        TRACE("pack: memsets queued");
    } else {
        // direct mode: tables go up together with the work items (sketch_from), the pack launch follows the direct pass
        pk->d_descs = nullptr; pk->d_tile_begin = nullptr; pk->d_nvalid = nullptr;
        if (any_multi && (rc = reserve(ctx, pk->brk_bytes, pk->total_brk * 4))) return rc;
    }
    PackArgs pa{};
    pa.seq = d_seq;
    pa.seq_end = d_seq_end;
    pa.rec_off = d_rec_off;
    pa.genomes = pk->d_descs;                                 // direct mode: filled in by direct_begin()
    pa.words = static_cast<uint32_t *>(pk->words.ptr);
    pa.brk = static_cast<uint32_t *>(pk->brk.ptr);
    pa.nvalid = pk->d_nvalid;
    pa.code_tab4 = layout_dev(ctx->layout, LASH_HMH).code_tab4;
    pk->code_tab4 = pa.code_tab4;
    pa.file_err = formats ? pk->d_dirty + 3 * (size_t)n_genomes + 1 : nullptr;      // raw files: FASTQ structure flags
    uint64_t *lb = static_cast<uint64_t *>(pk->lookback.ptr);
    PackV2Args v2{};
    v2.tiles = static_cast<const TileInfo *>(pk->tiles.ptr);
    v2.desc = lb;
    v2.error_flag = reinterpret_cast<uint32_t *>(lb + n_tiles);
    v2.desc2 = reinterpret_cast<uint32_t *>(lb + n_tiles + 1);
    v2.ticket = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(v2.desc2 + n_tiles + 1) + 127) & ~(uintptr_t)127);
    v2.n_tiles = (uint32_t)n_tiles;
    PackMapArgs pm{};
    pm.seq = d_seq;
    pm.rec_off = d_rec_off;
    pm.genomes = pa.genomes;
    pm.tile_begin = pk->d_tile_begin;
    pm.tiles = static_cast<TileInfo *>(pk->tiles.ptr);
    pm.n_tiles = (uint32_t)n_tiles;
    pm.n_genomes = n_genomes;
    pk->error_flag = direct ? nullptr : v2.error_flag;        // (direct mode launches no pack kernel)
    if (!direct) {
        HIPCHK(ctx, launch_pack_v2(pa, v2, pm, (uint32_t)ctx->cu_count, formats != nullptr, stream));
        if (formats) {
            // FASTQ files: quality-line lengths and how the file ends, into the same flags (fastq_check.hip)
            std::vector<FqFile> fq;
            uint64_t blocks = 0;
            const uint64_t bb = fastq_check_block_bytes();
            for (uint32_t g = 0; g < n_genomes; ++g) {
                if (formats[g] != LASH_FMT_FASTQ || descs[g].byte_len == 0) continue;
                const uint64_t nb = (descs[g].byte_len + bb - 1) / bb;
                fq.push_back(FqFile{descs[g].byte_off, descs[g].byte_len, (uint32_t)blocks, (uint32_t)nb, g, 0u});
                blocks += nb;
            }
            if (blocks > 0x7FFFFFFFull) return LASH_ELIMIT;
            if (!fq.empty()) {
                std::vector<Section> sec = {{fq.data(), fq.size() * sizeof(FqFile), 0}};
                const size_t total = layout_sections(sec), tab = total;
                if ((rc = reserve(ctx, pk->fq, tab + fastq_check_scratch_words((uint32_t)fq.size(), (uint32_t)blocks) * 4))) return rc;
                if ((rc = upload_sections(ctx, pk->fq.ptr, sec, total, stream))) return rc;
                uint8_t *fb = static_cast<uint8_t *>(pk->fq.ptr);
                HIPCHK(ctx, launch_fastq_check(d_seq, reinterpret_cast<const FqFile *>(fb + sec[0].off), (uint32_t)fq.size(), (uint32_t)blocks,
                                               reinterpret_cast<uint32_t *>(fb + tab), pa.file_err, stream));
            }
        }
    }
    if (ev) HIPCHK(ctx, hipEventRecord(ev->e[1], stream));
    TRACE("pack: done");
    return LASH_OK;
}

// direct mode, feedback: how much of the batch (in 16 KiB tiles) lies in genomes the direct pass gave up — counted on the device,
// copied to a pinned word without waiting; lash_sketch_batch_device looks at it before its next call (dirty_frac).
int probe_dirty(lash_ctx *ctx, lash_packed *pk, hipStream_t stream)
{
    uint32_t *tbc = static_cast<uint32_t *>(pk->tile_begin_c.ptr);
    HIPCHK(ctx, launch_dirty_tile_scan(pk->d_tile_begin, pk->d_dirty, pk->n_genomes, tbc, tbc + pk->n_genomes + 1, stream));
    if (!ctx->probe_host) {
        HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->probe_host), 64, hipHostMallocDefault));
        ctx->probe_host[0] = 0;
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->probe_ev, hipEventDisableTiming));
    }
    if (!ctx->probe_pending) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->probe_host, tbc + pk->n_genomes + 1, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(ctx, hipEventRecord(ctx->probe_ev, stream));
        ctx->probe_pending = true;
        ctx->probe_tiles = pk->h_tile_begin.empty() ? 0 : pk->h_tile_begin.back();
    }
    return LASH_OK;
}

// ---- binned launches (SketchPlan::bins; sketch_kernels.hip "BinRegs") ---------------------------------------------------------
// Register tables beyond 128 KiB of LDS: the sketch kernels hash every k-mer once and append an entry to the list of its bin,
// bins_apply_kernel builds each bin's registers in LDS and writes them into the genome's image (UltraLogLog, not accumulating) or leaves
// ONE partial per genome ("virtual item" n_items + g) for the ordinary finalize stage.  Lists, counters and fallback tables are sized per genome GROUP (a few GiB at a time; the stream orders
// the groups, so the buffers are reused), from an upper bound of the entries each genome's work items push.
// HBM one group of a binned launch (or one chunk of per-item global tables) may take.  LASH_BINS_MB (read per call), else a twelfth of the
// device's memory but no more than 24 GiB nor a third of what was free when the context first asked (round 6; 6 GiB before: 1 000 x 5 Mbp at
// p = 22 ran in ten groups of ~100 genomes — 300 work items for 256 workgroup slots, a tail per group — and each genome's 32 MiB fallback
// table is most of what the budget pays for; 24 GiB: -7 .. -9 %, more buys nothing, profiles/r06/bins_ab.txt)
uint64_t bins_budget_bytes(lash_ctx *ctx)
{
    if (const char *e = getenv("LASH_BINS_MB")) return (uint64_t)std::max(64, atoi(e)) << 20;
    if (!ctx->bins_budget) {
        size_t free_b = 0, total_b = 0;
        uint64_t b = 6144ull << 20;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b)
            b = std::max<uint64_t>(1ull << 30, std::min<uint64_t>({24ull << 30, (uint64_t)total_b / 12, (uint64_t)free_b / 3}));
        else (void)hipGetLastError();
        ctx->bins_budget = b;
    }
    return ctx->bins_budget;
}

struct BinsRun {
    std::vector<uint32_t> group_end;                  // genome index at which each group ends
    std::vector<BinGenome> table;                     // per genome: list offset inside its group's buffer, list capacity
    const BinGenome *d_table = nullptr;
    uint32_t *d_cnt = nullptr, *d_spill = nullptr;    // [max group][bins], [max group]
    WorkItem *d_vitems = nullptr;                     // one virtual item per genome
    uint32_t *d_vbegin = nullptr;                     // 0, 1, ..., n_genomes
    uint32_t slab_words = 0, max_group = 0;
    bool fits = true;
};
static int bins_prepare(lash_ctx *ctx, const SketchPlan &plan, const std::vector<uint64_t> &entries_of_genome, uint32_t n_genomes, BinsRun &br)
{
    const uint32_t B = 1u << plan.bins_log2;
    br.slab_words = plan.nreg32;                                     // HLL: 2^p words, ULL: 2 * 2^p
    const uint64_t budget = bins_budget_bytes(ctx);
    br.table.resize(n_genomes);
    uint64_t bytes = 0, off = 0, group_max_bytes = 0;
    uint32_t in_group = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) {
        // a row's entries leave padded to a multiple of SIX, six to a 16-byte chunk (round 6): (m + 2.5) / m on average for rows of m entries per
        // flush, two thirds of a 32-bit word each; `cap` and the lists' counters are in words
        const uint64_t m_row = std::max<uint64_t>(1, 1024u >> (plan.bins_log2 + plan.bin_sub_shift));
        uint64_t mean = entries_of_genome[g] / B * (2 * m_row + 6) / (2 * m_row) * 2 / 3 + 4;
        uint64_t sq = 1; while (sq * sq < mean) ++sq;
        const uint64_t cap = (mean + mean / 8 + 8 * sq + 1024 + 63) & ~63ull;
        if (cap > 0xFFFFFFFFull) { br.fits = false; return LASH_OK; }
        const uint64_t mine = B * cap * 4 + (uint64_t)br.slab_words * 4;
        if (mine > budget) { br.fits = false; return LASH_OK; }     // one genome beyond the budget: the caller takes the global-table path
        if (in_group && (bytes + mine > budget || in_group == 65535u)) {
            br.group_end.push_back(g);
            br.max_group = std::max(br.max_group, in_group);
            bytes = 0; off = 0; in_group = 0;
        }
        br.table[g] = BinGenome{off, (uint32_t)cap, 0u};
        off += B * cap;
        bytes += mine;
        group_max_bytes = std::max(group_max_bytes, off * 4);
        ++in_group;
    }
    br.group_end.push_back(n_genomes);
    br.max_group = std::max(br.max_group, in_group);
    int rc;
    if ((rc = reserve(ctx, ctx->bins_lists, group_max_bytes + 256))) return rc;
    {
        void *before = ctx->bins_slab.ptr;
        if ((rc = reserve(ctx, ctx->bins_slab, (size_t)br.max_group * br.slab_words * 4 + 256))) return rc;
        if (ctx->bins_slab.ptr != before) ctx->bins_slab_fill = -1;     // new memory: contents unknown
    }
    std::vector<WorkItem> vitems(n_genomes);
    std::vector<uint32_t> vbegin(n_genomes + 1);
    for (uint32_t g = 0; g < n_genomes; ++g) { vitems[g] = WorkItem{g, 0u, 4u, 0u}; vbegin[g] = g; }
    vbegin[n_genomes] = n_genomes;
    std::vector<Section> sec = {{br.table.data(), br.table.size() * sizeof(BinGenome), 0}, {vitems.data(), vitems.size() * sizeof(WorkItem), 0},
                                {vbegin.data(), vbegin.size() * 4, 0}};
    const size_t tabs = layout_sections(sec), cnt_bytes = ((size_t)br.max_group * B * 4 + 255) & ~(size_t)255, spill_bytes = cnt_bytes;   // (one flag per bin)
    if ((rc = reserve(ctx, ctx->bins_meta, tabs + cnt_bytes + spill_bytes + 256))) return rc;
    if ((rc = upload_sections(ctx, ctx->bins_meta.ptr, sec, tabs, ctx->stream))) return rc;
    uint8_t *mb = static_cast<uint8_t *>(ctx->bins_meta.ptr);
    br.d_table = reinterpret_cast<const BinGenome *>(mb + sec[0].off);
    br.d_vitems = reinterpret_cast<WorkItem *>(mb + sec[1].off);
    br.d_vbegin = reinterpret_cast<uint32_t *>(mb + sec[2].off);
    br.d_cnt = reinterpret_cast<uint32_t *>(mb + tabs);
    br.d_spill = reinterpret_cast<uint32_t *>(mb + tabs + cnt_bytes);
    return LASH_OK;
}
// UltraLogLog, not accumulating: bins_apply_kernel writes the images itself and the call queues no finalize launch (BinApplyArgs::images)
static bool bins_write_images(const lash_params *prm) { return prm->algo == LASH_ULL && !(prm->flags & LASH_F_ACCUMULATE); }
// partial sketches a call needs room for: one per work item — but the items of a binned launch leave entries, not partials: one per GENOME then
// (bins_apply_kernel's, for finalize_kernel), or none when it writes the images itself (round 6: a slot per item was 4 MiB x 3 000 items at
// p = 22 for 1 000 genomes, never touched — and out of memory at p = 23)
static size_t bins_partials(const SketchPlan &plan, const lash_params *prm, size_t n_items, size_t n_genomes)
{
    if (!plan.bins) return n_items + 1;
    return bins_write_images(prm) ? 1 : n_genomes + 1;
}
// the launches of one call, group by group: launch(sa, first item, items) queues the sketch kernels of an item range
template <class Launch>
static int bins_run(lash_ctx *ctx, const SketchPlan &plan, const lash_params *prm, SketchArgs sa, const BinsRun &br, const std::vector<uint32_t> &item_begin,
                    uint32_t n_items, const uint32_t *d_item_begin, Launch launch)
{
    const bool to_images = bins_write_images(prm);
    const uint32_t B = 1u << plan.bins_log2;
    sa.bin_lists = static_cast<uint32_t *>(ctx->bins_lists.ptr);
    sa.bin_cnt = br.d_cnt;
    sa.bin_slab = static_cast<uint32_t *>(ctx->bins_slab.ptr);
    sa.bin_spill = br.d_spill;
    sa.bins = B; sa.bin_shift = plan.bin_shift; sa.bin_S = plan.bin_S; sa.bin_sub_shift = plan.bin_sub_shift; sa.bin_flush_words = plan.bin_flush_words; sa.bin_slab_words = br.slab_words;
    sa.item_order = nullptr;
    // the fallback tables: empty at rest (bins_apply_kernel wipes what it folds in); wiped here only when new, or last left by the other sketch type
    {
        const int fill = prm->algo == LASH_ULL ? 0x00 : 0xFF;
        const size_t need = (size_t)br.max_group * br.slab_words * 4;
        if (ctx->bins_slab_fill != fill || ctx->bins_slab_clean < need) {
            HIPCHK(ctx, hipMemsetAsync(ctx->bins_slab.ptr, fill, need, ctx->stream));
            ctx->bins_slab_clean = need;
        }
        ctx->bins_slab_fill = -1;                                      // (until this call's last bins_apply_kernel is queued)
    }
    uint32_t g0 = 0;
    for (uint32_t g1 : br.group_end) {
        const uint32_t ng = g1 - g0;
        HIPCHK(ctx, hipMemsetAsync(br.d_cnt, 0, (size_t)ng * B * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(br.d_spill, 0, (size_t)ng * B * 4, ctx->stream));
        sa.bin_genomes = br.d_table + g0;
        sa.bin_genome0 = g0;
        sa.item_base = item_begin[g0];
        int rc = launch(sa, item_begin[g0], item_begin[g1] - item_begin[g0]);
        if (rc) return rc;
        BinApplyArgs ba{};
        ba.lists = sa.bin_lists; ba.cnt = br.d_cnt; ba.slab = sa.bin_slab; ba.spill = br.d_spill; ba.genomes = sa.bin_genomes;
        ba.partials = sa.partials; ba.item_kmers = sa.item_kmers; ba.genome_item_begin = d_item_begin;
        ba.items = sa.items; ba.nvalid = sa.nvalid; ba.k = prm->k;
        ba.partial_stride = sa.partial_stride; ba.virt0 = n_items + g0; ba.genome0 = g0;      // (k-mer count at item_kmers[virt0 + gi], partial sketch at partials[genome0 + gi])
        ba.bins = B; ba.bin_shift = plan.bin_shift; ba.slab_words = br.slab_words; ba.algo = prm->algo; ba.p = prm->p;
        if (to_images) {
            ba.images = sa.images; ba.image_bytes = sa.image_bytes; ba.hdr_tpl = sa.lay.hdr_tpl; ba.hdr_bytes = sa.lay.hdr_bytes;
            ba.kmer_counter = static_cast<unsigned long long *>(ctx->counter.ptr);
        }
        HIPCHK(ctx, launch_bins_apply(ba, ng, ctx->stream));
        g0 = g1;
    }
    ctx->bins_slab_fill = prm->algo == LASH_ULL ? 0x00 : 0xFF;
    return LASH_OK;
}

// UltraLogLog p >= 23: every work item updates a table of its own in global memory (2^p x 8 bytes: 64 MiB at p = 23, 512 MiB at
// p = 26).  The items of a call run a chunk at a time so that the tables of one chunk fit a budget (round 4; a table per item of
// the whole call was 256 GB for 200 genomes at p = 23); launch(sa, items) queues the sketch kernels of an item range.
template <class Launch>
static int global_run(lash_ctx *ctx, const SketchPlan &plan, SketchArgs sa, uint32_t n_items, Launch launch)
{
    const uint64_t budget = bins_budget_bytes(ctx);
    const uint64_t table = (uint64_t)plan.nreg32 * 4;
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_items ? n_items : 1, budget / table));
    int rc;
    if ((rc = reserve(ctx, ctx->gregs, (size_t)per * table + 256))) return rc;
    sa.gregs = static_cast<uint32_t *>(ctx->gregs.ptr);
    sa.item_order = nullptr;
    for (uint32_t i0 = 0; i0 < n_items; i0 += per) {
        const uint32_t n = std::min(per, n_items - i0);
        HIPCHK(ctx, hipMemsetAsync(ctx->gregs.ptr, 0, (size_t)n * table, ctx->stream));
        sa.item_base = i0;
        if ((rc = launch(sa, n))) return rc;
    }
    return LASH_OK;
}

// ---- whole small genomes on persistent workgroups (sole_kernels.hip, round 5) --------------------------------------------------
// Which genomes of a call the persistent kernel takes: those of at most LASH_SOLE_MAX bytes (0 = none), when the sketch's table
// fits its LDS budget and nothing asks for another route.
uint64_t sole_max_bytes(const lash_ctx *ctx, const lash_params *prm, const SolePlan &sp)
{
    if (!sp.ok || (prm->flags & (LASH_F_NO_SOLE | LASH_F_AMINO | LASH_F_STREAM_ONLY))) return 0;
    if (getenv("LASH_STREAM_FIRST")) return 0;                       // (A/B knob of tools/: every genome through stream_sketch_kernel)
    const char *e = getenv("LASH_SOLE_MAX");                         // read per call: tests and tools flip it in-process
    const long long v = e ? atoll(e) : 393216;
    // (the kernel holds a genome's length and offsets in 32 bits: anything that large belongs to the sliced kernels anyway)
    return v > 0 ? (uint64_t)std::min<long long>(v, 64ll << 20) : 0;
}

// chunks per workgroup: the launch's tail is one chunk long (tools/: LASH_SOLE_CHUNKS)
uint32_t sole_chunks_per_wg()
{
    const char *e = getenv("LASH_SOLE_CHUNKS");
    const int v = e ? atoi(e) : 24;
    return (uint32_t)std::max(1, std::min(v, 4096));
}

// Chunks of consecutive genomes of about equal cost, planned from the genome byte offsets alone: cost = bytes + a fixed part per
// genome (its flush).  off[g] = first byte (or any monotone position) of genome g, off[n] = the end.
void sole_chunks(const uint64_t *off, uint32_t n_genomes, uint64_t fixed, uint32_t want, std::vector<uint32_t> &chunk_begin)
{
    const uint32_t n_chunks = std::max(1u, std::min(want, n_genomes));
    chunk_begin.resize(n_chunks + 1);
    const long double total = (long double)(off[n_genomes] - off[0]) + (long double)fixed * n_genomes;
    uint32_t g = 0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        chunk_begin[c] = g;
        const long double goal = total * (c + 1) / n_chunks;
        // first genome whose PREFIX cost reaches the goal: binary search (cost is monotone in g)
        uint32_t lo = g, hi = n_genomes;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            const long double cost = (long double)(off[mid + 1] - off[0]) + (long double)fixed * (mid + 1);
            if (cost < goal) lo = mid + 1; else hi = mid;
        }
        g = std::min(n_genomes, std::max(lo + 1, g + 1));            // at least one genome per chunk
        if (n_genomes - g < n_chunks - 1 - c) g = n_genomes - (n_chunks - 1 - c);   // ... and one left for each chunk to come
    }
    chunk_begin[n_chunks] = n_genomes;
}

// Queues the persistent kernel over every genome of at most `max_len` bytes (+ the record-start marks it reads, + its census).
// ASCII source: d_seq / d_rec_off / host genome_byte_off; packed source: pk.  per_genome_ndel: the direct pass's per-genome
// deleted-byte counts (calls that also run the sliced launch keep lash_timing::bases_last per genome), else NULL.
int sole_run(lash_ctx *ctx, const lash_params *prm, const SolePlan &sp, uint64_t max_len, const uint8_t *d_seq, uint64_t seq_bytes,
             const uint64_t *d_rec_off, uint64_t n_rec, bool any_multi, bool rec_identity, const uint64_t *genome_byte_off, const lash_packed *pk,
             uint32_t n_genomes, uint8_t *d_out_images, uint32_t *per_genome_ndel)
{
    if (n_genomes == 0) return LASH_OK;
    const bool packed = pk != nullptr;
    const uint64_t image_bytes = ::image_bytes(ctx->layout, prm->algo, prm->p);
    const bool x_low = rule_variant(ctx->layout, prm->algo, prm->flags);   // (HyperMinHash x = low half / HyperLogLog bucket = top bits)
    // as many workgroups as are RESIDENT at a time (the kernel variant's registers, LDS and wave slots taken together): chunks are handed
    // out to running workgroups, one that started late would only hold its first chunk back
    uint32_t per_cu = sp.wg_per_cu;
    HIPCHK(ctx, sole_resident_per_cu(sp, prm->algo, prm->k, x_low, packed, &per_cu));
    uint32_t n_wg = (uint32_t)std::min<uint64_t>((uint64_t)ctx->cu_count * per_cu, n_genomes);
    // (tests and the randomized runners: FEW workgroups, so that a test batch of a hundred genomes walks the paths of a collection of a
    //  million — several genomes per chunk, one after the other on the same rings and table, the next one's bytes in flight.  Without
    //  this every genome of a small batch has a workgroup of its own; a stale ring pointer survived round 5's suite that way.)
    if (const char *e = getenv("LASH_SOLE_WGS")) n_wg = (uint32_t)std::max(1, std::min<int>((int)n_wg, atoi(e)));
    // chunks: a couple of dozen per workgroup, so that the tail of the launch is a few percent of a workgroup's share — but none
    // smaller than ~100 us of a workgroup's time (a chunk starts with a few dependent loads: 3..5 us)
    std::vector<uint32_t> chunk_begin;
    std::vector<uint64_t> off_tmp;
    const uint64_t *off = genome_byte_off;
    if (packed) {
        off_tmp.resize((size_t)n_genomes + 1);
        off_tmp[0] = 0;
        for (uint32_t g = 0; g < n_genomes; ++g) off_tmp[g + 1] = off_tmp[g] + pk->byte_len[g];
        off = off_tmp.data();
    }
    {
        const uint64_t fixed = image_bytes / 4 + 256;
        const uint64_t total = off[n_genomes] - off[0] + fixed * n_genomes;
        const uint64_t min_cost = 512ull * sp.threads;                     // 256 KiB for eight waves, 32 KiB for one
        const uint64_t by_cost = std::max<uint64_t>(n_wg, total / min_cost);
        sole_chunks(off, n_genomes, fixed, (uint32_t)std::min<uint64_t>((uint64_t)n_wg * sole_chunks_per_wg(), by_cost), chunk_begin);
    }
    const uint32_t n_chunks = (uint32_t)chunk_begin.size() - 1;
    int rc;
    std::vector<Section> sec = {{chunk_begin.data(), chunk_begin.size() * 4, 0}};
    // every genome exactly one record (the usual case: one sequence per file): its byte offsets ARE the record offsets, which are
    // resident already — no per-genome table goes up at all
    // (identity mapping only: genome_rec_off = [0, 1, 1] with two records has n_rec == n_genomes and no multi-record genome either — ADVICE r5)
    const bool gbo_is_rec_off = !packed && !any_multi && n_rec == n_genomes && d_rec_off != nullptr && rec_identity;
    if (!packed && !gbo_is_rec_off) sec.push_back({genome_byte_off, ((size_t)n_genomes + 1) * 8, 0});
    const size_t tabs = layout_sections(sec), counts_bytes = ((size_t)n_wg * 16 + 255) & ~(size_t)255;
    if ((rc = reserve(ctx, ctx->sole_tab, tabs + counts_bytes + 256))) return rc;
    if ((rc = upload_sections(ctx, ctx->sole_tab.ptr, sec, tabs, ctx->stream))) return rc;
    uint8_t *tb = static_cast<uint8_t *>(ctx->sole_tab.ptr);
    if (!ctx->sole_state.ptr) {
        // the chunk ticket: zero at rest (sole_census_kernel, which follows every launch on the stream, puts it back)
        if ((rc = reserve(ctx, ctx->sole_state, 256))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->sole_state.ptr, 0, 256, ctx->stream));
    }
    if ((rc = reserve(ctx, ctx->counter, 256))) return rc;
    if (!ctx->counter_zeroed) {
        HIPCHK(ctx, hipMemsetAsync(ctx->counter.ptr, 0, 256, ctx->stream));
        ctx->counter_zeroed = true;
    }
    SoleArgs sa{};
    sa.chunk_begin = reinterpret_cast<const uint32_t *>(tb + sec[0].off);
    sa.n_chunks = n_chunks;
    sa.ticket = static_cast<uint32_t *>(ctx->sole_state.ptr);
    sa.wg_counts = reinterpret_cast<unsigned long long *>(tb + tabs);
    sa.max_len = max_len;
    sa.safe = static_cast<const uint8_t *>(ctx->counter.ptr) + 128;
    if (!packed) {
        sa.seq = d_seq;
        sa.seq_bytes = seq_bytes;
        sa.genome_byte_off = gbo_is_rec_off ? d_rec_off : reinterpret_cast<const uint64_t *>(tb + sec[1].off);
        sa.ndel = per_genome_ndel;
        if (any_multi) {
            // some genome has more than one record: record starts as bits at absolute byte positions (16 spare bytes: a lane reads
            // its 16 bits with one 4-byte load at any alignment)
            const size_t bm_bytes = ((seq_bytes + 63) / 32 + 2) * 4 + 2048;    // (+ a round of the widest workgroup: the prefetch past the last genome)
            if ((rc = reserve(ctx, ctx->sole_brk, bm_bytes))) return rc;
            HIPCHK(ctx, hipMemsetAsync(ctx->sole_brk.ptr, 0, bm_bytes, ctx->stream));
            HIPCHK(ctx, launch_sole_mark(d_rec_off, n_rec, seq_bytes, static_cast<uint32_t *>(ctx->sole_brk.ptr), ctx->stream));
            sa.brk_abs = static_cast<const uint32_t *>(ctx->sole_brk.ptr);
        }
    } else {
        sa.words = static_cast<const uint32_t *>(pk->words.ptr);
        sa.brk = static_cast<const uint32_t *>(pk->brk.ptr);
        sa.genomes = pk->d_descs;
        sa.nvalid = pk->d_nvalid;
    }
    sa.images = d_out_images;
    sa.image_bytes = image_bytes;
    {
        const double alpha0 = hll_alpha(prm->p);
        memcpy(&sa.alpha_bits, &alpha0, 8);
    }
    sa.accumulate = (prm->flags & LASH_F_ACCUMULATE) ? 1 : 0;
    sa.hll_corner = prm->algo == LASH_HLL ? static_cast<uint32_t *>(ctx->hll_flags.ptr) : nullptr;
    sa.bitflip = prm->algo == LASH_HMH ? xxh3_bitflip128(prm->seed) : xxh3_bitflip64(prm->seed);
    sa.lay = layout_dev(ctx->layout, prm->algo);
    sa.nreg32 = prm->algo == LASH_HMH ? HMH_M : prm->algo == LASH_HLL ? (1u << prm->p) : (2u << prm->p);
    sa.k = prm->k;
    sa.p = prm->p;
    HIPCHK(ctx, launch_sole(sp, prm->algo, prm->k, x_low, packed, sa, n_wg, ctx->stream));
    unsigned long long *ctr = static_cast<unsigned long long *>(ctx->counter.ptr);
    HIPCHK(ctx, launch_sole_census(sa.wg_counts, n_wg, ctr, ctr + 1, sa.ticket, ctx->stream));
    ctx->last.sole_launches += 1;
    return LASH_OK;
}

int sketch_from(lash_ctx *ctx, const lash_params *prm, const lash_packed *pk, uint8_t *d_out_images, EvSet *ev, bool allow_bins)
{
    const uint32_t n_genomes = pk->n_genomes;
    // Genomes of at most sole_max bytes go to the persistent kernel (sole_kernels.hip), the others are cut into work items as ever;
    // blen() is a genome's length as the planning below sees it (0 = not this launch's)
    const SolePlan sole_plan = make_sole_plan(prm->algo, prm->p, n_genomes, (uint32_t)ctx->cu_count);
    uint64_t sole_max = sole_max_bytes(ctx, prm, sole_plan);
    if (pk->direct && n_genomes && pk->h_descs[n_genomes - 1].byte_off + pk->h_descs[n_genomes - 1].byte_len < 16) sole_max = 0;   // (the kernel loads 16 bytes at a time, from inside the buffer)
    auto blen = [&](uint32_t g) -> uint64_t { return pk->byte_len[g] <= sole_max && sole_max ? 0 : pk->byte_len[g]; };
    uint32_t n_sole = 0;
    if (sole_max) for (uint32_t g = 0; g < n_genomes; ++g) n_sole += pk->byte_len[g] <= sole_max;
    const bool x_low = rule_variant(ctx->layout, prm->algo, prm->flags);   // (HyperMinHash x = low half / HyperLogLog bucket = top bits)
    // what the persistent kernel's launch needs from a batch in direct mode (ASCII in the caller's buffer)
    std::vector<uint64_t> sole_gbo;
    auto sole_launch = [&](uint32_t *ndel) -> int {
        if (pk->direct) {
            sole_gbo.resize((size_t)n_genomes + 1);
            for (uint32_t g = 0; g < n_genomes; ++g) sole_gbo[g] = pk->h_descs[g].byte_off;
            sole_gbo[n_genomes] = pk->h_descs[n_genomes - 1].byte_off + pk->h_descs[n_genomes - 1].byte_len;
            bool identity = pk->n_rec == n_genomes;                        // genome g IS record g (then the resident record offsets serve as byte offsets)
            for (uint32_t g = 0; g < n_genomes && identity; ++g) identity = pk->h_descs[g].rec_begin == g && pk->h_descs[g].rec_end == g + 1u;
            return sole_run(ctx, prm, sole_plan, sole_max, pk->d_seq, sole_gbo[n_genomes], pk->d_rec_off, pk->n_rec, pk->any_multi, identity, sole_gbo.data(),
                            nullptr, n_genomes, d_out_images, ndel);
        }
        return sole_run(ctx, prm, sole_plan, sole_max, nullptr, 0, nullptr, 0, false, false, nullptr, pk, n_genomes, d_out_images, nullptr);
    };
    if (n_sole == n_genomes && n_genomes && !pk->direct) {
        // a packed batch of small genomes only (lash_sketch_packed_device, raw files, LASH_F_NO_DIRECT): no work items at all
        int rc;
        ctx->hll_flags_n = 0;
        ctx->hll_flags_on_host = false;
        if (prm->algo == LASH_HLL) {                                    // (every genome's flag is written by the kernel: nothing to clear)
            if ((rc = reserve(ctx, ctx->hll_flags, (size_t)n_genomes * 4))) return rc;
            ctx->hll_flags_n = n_genomes;
        }
        if (ev) HIPCHK(ctx, hipEventRecord(ev->e[2], ctx->stream));
        if ((rc = sole_launch(nullptr))) return rc;
        if (ev) { HIPCHK(ctx, hipEventRecord(ev->e[3], ctx->stream)); HIPCHK(ctx, hipEventRecord(ev->e[4], ctx->stream)); ev->done = true; }
        ctx->last_packed.push_back(pk);
        ctx->last.sketch_launches += 1;
        ctx->last.sketch_workgroups = (uint32_t)std::min<uint64_t>((uint64_t)ctx->cu_count * sole_plan.wg_per_cu, n_genomes);
        return LASH_OK;
    }
    uint64_t total_bytes = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) total_bytes += blen(g);
    const bool small_items = n_genomes > n_sole && total_bytes / (n_genomes - n_sole) < 100000u;
    SketchPlan plan = make_sketch_plan(prm->algo, prm->k, prm->p, x_low, small_items, allow_bins);
    if (plan.bins) {
        // a binned launch keeps ~4.6 bytes per input byte of one genome group in HBM: a single genome beyond the budget (a multi-Gbp
        // input in one call, which the CLI would have streamed in chunks) takes the table-in-global-memory path instead
        const uint64_t budget = bins_budget_bytes(ctx);
        uint64_t big = 0;
        for (uint32_t g = 0; g < n_genomes; ++g) big = std::max<uint64_t>(big, pk->byte_len[g]);
        if (big * 5 + (uint64_t)plan.nreg32 * 4 + (64u << 20) > budget)
            plan = make_sketch_plan(prm->algo, prm->k, prm->p, x_low, small_items, false);
    }
    const uint64_t image_bytes = ::image_bytes(ctx->layout, prm->algo, prm->p);

    // ---- plan work items: slices of genomes, enough of them to keep every CU's workgroup slots busy ----
    const bool defer_eligible = prm->algo == LASH_HMH && plan.use_lds && plan.parts_log2 == 0;   // (see plan_d below)
    const uint32_t lds_wg = plan.lds_bytes + ((pk->direct || defer_eligible || plan.bytes) ? sketch_direct_stage_bytes(plan) : 0u);   // + the waves' staging areas / lists
    const uint32_t wg_per_cu = plan.use_lds ? std::max(1u, (160u * 1024u) / std::max(lds_wg, 1u)) : 4u;   // 64 KiB + census -> 2
    const uint64_t slots = (uint64_t)ctx->cu_count * std::min(wg_per_cu, 2048u / plan.threads);
    uint64_t total_words = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) total_words += (blen(g) + 15) / 16;
    const uint64_t step = (uint64_t)plan.threads * SKETCH_WORDS_PER_THREAD;
    const uint64_t min_slice = step * 8;                           // amortise the LDS clear + flush
    static const uint64_t slice_factor_env = getenv("LASH_SLICE_FACTOR") ? std::max(1, atoi(getenv("LASH_SLICE_FACTOR"))) : 0;
    uint64_t len_lo = ~0ull, len_hi = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) { len_lo = std::min<uint64_t>(len_lo, blen(g)); len_hi = std::max<uint64_t>(len_hi, blen(g)); }
    const bool equal_genomes = n_genomes > 0 && len_hi <= len_lo + len_lo / 4;
    // HyperMinHash launches that may defer their signatures (below) like long items — a slice starts with an empty table, and the share
    // of k-mers that pass the filter is 2.8 % over a whole 5 Mbp genome, 7.7 % over a third of one — and the split tail (below) has
    // taken over what the many small slices were for: 2x the slots there (1 000 x 5 Mbp: 4.39 -> 4.33 ms, round 3), and for batches of
    // EQUAL genomes 1x: BASELINE configs[1], 1 000 x 5 Mbp, runs whole genomes in two rounds instead of thirds in six (4.12 -> 3.95 ms,
    // profiles/r04/cfg1_slicing.txt; a collection of unequal genomes loses a third with that: it needs the item cap below)
    const uint64_t slice_factor = slice_factor_env ? slice_factor_env : (defer_eligible ? (equal_genomes ? 1 : 2) : 4);   // tuning knob: the
    // sketch time is flat from 2x to 24x the slots (4.87-4.91 ms on the default bench), the finalize time grows with it
    uint64_t target = total_words / (slots * slice_factor) + 1;
    // a table in global memory per work item (UltraLogLog p >= 23: 64 .. 512 MiB each, zeroed before and read back after): few, long items
    if (!plan.use_lds) target = total_words / std::max<uint64_t>(1, slots / 4) + 1;
    target = std::max(target, min_slice);
    {
        // When some genome is cut anyway (so partials and the finalize pass exist whatever the slicing), items of at most 1 MiB: a
        // large batch would otherwise get multi-megabyte items, and the few genomes handed to stream_sketch_kernel — one or two
        // items each — would run on a fraction of the chip (2 000 mixed genomes: that launch 1.6 ms -> 0.5 ms).
        // (2 MiB where the launch may defer signatures: that kernel wants long items — the same collection 7.83 -> 7.50 ms, while the
        // HyperLogLog kernel loses with the larger items, 8.09 -> 8.49 ms)
        static const uint64_t cap_env = getenv("LASH_ITEM_CAP_WORDS") ? std::max(1024, atoi(getenv("LASH_ITEM_CAP_WORDS"))) : 0;
        const uint64_t cap = cap_env ? cap_env : (defer_eligible ? 131072 : 65536);
        // (only for batches of unequal genomes: a batch of equal ones keeps its few large items — when those are soft-masked they all
        // are, every item is busy in both launches, and smaller items only add ramp-up: -3 % on bench.py --dirty lower)
        bool any_cut = false;
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t g = 0; g < n_genomes; ++g) {
            any_cut = any_cut || (((blen(g) + 15) / 16 + 3) & ~3ull) > target;
            lo = std::min<uint64_t>(lo, blen(g)); hi = std::max<uint64_t>(hi, blen(g));
        }
        if (any_cut && hi > lo + lo / 4) target = std::max(min_slice, std::min(target, cap));
    }
    std::vector<WorkItem> items;
    uint32_t max_slices = 0;                                       // most slices any genome is cut into
    bool all_sole = plan.parts_log2 == 0 && plan.use_lds && !plan.bins && n_genomes > 0;   // every genome has exactly one work item
    std::vector<uint32_t> item_begin(n_genomes + 1, 0);
    items.reserve(n_genomes * 2);
    auto slicing = [&](uint32_t g, uint64_t &nw, uint64_t &ns, uint64_t &per) {
        nw = ((blen(g) + 15) / 16 + 3) & ~3ull;
        ns = nw ? (nw + target - 1) / target : 0;
        per = nw ? (((nw + ns - 1) / ns) + 3) & ~3ull : 0;
    };
    // The tail of a launch: equal items run in lockstep rounds of `slots`, and the last round is as long as a full one however
    // few items it holds (600 x 5 Mbp in 2 400 items = 4.7 rounds took the time of 5; 12 500 whole genomes 24.4 -> 25).  The last
    // round's worth of slices is therefore cut into quarters: the launch ends on a quarter-round boundary instead.  Index order is
    // launch order, so the small items are the ones handed out last.
    static const uint32_t tail_split_env = getenv("LASH_TAIL_SPLIT") ? (uint32_t)std::max(1, atoi(getenv("LASH_TAIL_SPLIT"))) : 0u;
    // (halves where whole genomes may defer signatures: a quarter of a genome fills its table four times over)
    // (none where every genome goes straight to stream_sketch_kernel — recent batches were soft-masked: a wave of that kernel walks a
    //  contiguous sixteenth of its item and pays per part: its ring's warm-up, the look-ahead past its part, a last batch under a
    //  mask.  10 kb blocks 2.94 -> 2.80 ms, 2.5 kb blocks 3.39 -> 3.27 ms per 1 000 x 5 Mbp)
    const uint32_t tail_split = tail_split_env ? tail_split_env : (pk->direct && pk->stream_first ? 1u : (defer_eligible && equal_genomes ? 2u : 4u));
    const uint64_t tail_min = min_slice / 8;                        // 32 kb of sequence: 15 us of a workgroup's time
    const bool tail_geo = !(getenv("LASH_TAIL_GEO") && atoi(getenv("LASH_TAIL_GEO")) == 0);   // A/B knob (read per call): 0 = the uniform split of rounds 3-5
    uint64_t n_coarse = 0, fine_from = ~0ull;
    uint64_t c_lo = ~0ull, c_hi = 0;                               // smallest and largest slice
    for (uint32_t g = 0; g < n_genomes; ++g) {
        uint64_t nw, ns, per;
        slicing(g, nw, ns, per);
        if (!nw) continue;
        const uint64_t cnt = (nw + per - 1) / per, last = nw - (cnt - 1) * per;
        n_coarse += cnt;
        c_lo = std::min(c_lo, last); c_hi = std::max(c_hi, cnt > 1 ? per : last);
    }
    const bool unequal = c_hi > c_lo + c_lo / 4;                   // then the launch goes longest first (below) and ends on its small items anyway
    // (two or three rounds' worth in quarters, or halves / eighths: the same within noise)
    if (plan.use_lds && tail_split > 1 && n_coarse > slots && !unequal) fine_from = n_coarse - slots;
    uint64_t ci = 0;                                               // coarse slice counter over the batch
    for (uint32_t g = 0; g < n_genomes; ++g) {
        item_begin[g] = (uint32_t)items.size();
        uint64_t nw, ns, per;
        slicing(g, nw, ns, per);
        if (nw == 0) { all_sole = false; continue; }               // no work item at all: finalize writes the empty image (or it is the
                                                                   // persistent kernel's: FinalizeArgs::skip_max_len)
        uint32_t s = 0;
        const bool whole = ns == 1 && plan.parts_log2 == 0 && plan.use_lds && !plan.bins;
        for (uint64_t b = 0; b < nw; b += per, ++ci) {
            const uint64_t e = std::min(nw, b + per);
            uint64_t sub = e - b;                                  // this slice as one item, or as tail_split smaller ones
            if (ci >= fine_from) {
                // ... and the later HALF of that last round in twice as many parts, its last QUARTER in four times as many (round 6): sizes
                // are what the host balances by, but a byte's cost varies sixfold with what it holds — the soft-masked half of a genome runs
                // at 0.07 us per kB, the clean half at 0.41 — so a launch of equal halves can still end on one full half running alone
                // (profiles/r06/dirty_2500000_trace.txt: the last items started at 2.5 of 3.5 ms).  Ever smaller items towards the end
                // bound that tail whatever the bytes cost; on clean input it is neutral (12 500 x 5 Mbp, 1 000 x 5 Mbp: profiles/r06/tail_geo_ab.txt)
                uint64_t split = tail_split;
                const uint64_t from_end = n_coarse - 1 - ci;
                if (tail_geo && from_end < slots / 2) split *= 2;
                if (tail_geo && from_end < slots / 4) split *= 2;
                while (split > 1 && (e - b) / split < tail_min) split /= 2;
                if (split > 1) sub = ((((e - b) + split - 1) / split) + 3) & ~3ull;
            }
            const bool sole = whole && sub == e - b;
            for (uint64_t bb = b; bb < e; bb += sub, ++s)
                for (uint32_t part = 0; part < (1u << plan.parts_log2); ++part)               // slice index | pass << 16
                    items.push_back(WorkItem{g, (uint32_t)bb, (uint32_t)std::min(e, bb + sub), (s & 0x7FFFu) | (part << 16) | (sole ? ITEM_SOLE : 0u)});
        }
        max_slices = std::max<uint32_t>(max_slices, s);
        if (s != 1) all_sole = false;
    }
    item_begin[n_genomes] = (uint32_t)items.size();
    const uint32_t n_items = (uint32_t)items.size();
    if (pk->direct) {
        // how many waves must judge a genome too dirty before it is handed over: one for a genome of a few items, 1 in 32 for a 3 Gbp
        // read set cut into thousands (where SOME wave always meets four reads with an N among its first tiles)
        lash_packed *mpk = const_cast<lash_packed *>(pk);
        for (uint32_t g = 0; g < n_genomes; ++g)
            mpk->h_descs[g].handover = std::max<uint32_t>(1u, (item_begin[g + 1] - item_begin[g]) * (plan.threads / 64u) / 32u);
    }
    // Launch order: longest items first when their sizes differ (a collection of 0.6 .. 12 Mbp genomes lost 11 % to the tail of a
    // launch in genome order: the hardware hands workgroups out in index order, and a 3.6 MB item that starts last runs alone).
    // A bucket sort on the size's leading bits: O(items), stable inside a bucket (neighbouring items still share cache lines).
    std::vector<uint32_t> order;
    {
        if (n_items > slots && unequal && !plan.bins && plan.use_lds) { // (binned / global-table launches run their items range by range, in order)
            auto bucket = [&](uint32_t n) {                             // 8 buckets per octave, larger sizes first
                const uint32_t e = 31u - (uint32_t)__builtin_clz(n | 1u);
                const uint32_t m = e >= 3 ? (n >> (e - 3)) & 7u : 0u;
                return 255u - (e * 8u + m);
            };
            uint32_t count[257] = {0};
            for (const WorkItem &w : items) ++count[bucket(w.word_end - w.word_begin) + 1];
            for (int b = 0; b < 256; ++b) count[b + 1] += count[b];
            order.resize(n_items);
            for (uint32_t i = 0; i < n_items; ++i) order[count[bucket(items[i].word_end - items[i].word_begin)]++] = i;
        }
    }
    // HyperMinHash with deferred signatures (process_word_defer) pays off when a work item's table fills up early in the item, i.e.
    // when items are long: the share of k-mers that can still change their bucket is 2.8 % at 5 Mbp per item, 10 % at 1 Mbp
    // (profiles/r03/defer/ab.txt: -12 % of the kernel's time at 5 Mbp per item, -5.5 % at 1 Mbp, -3 % at 1.25 Mbp slices, +3 % at
    // 0.73 Mbp, +17 % at 0.26 Mbp; with round 4's threshold words and per-lane stacks, profiles/r04/defer/items.txt: -13.8 % at
    // 2 Mbp, -6.5 % at 1 Mbp, -2.6 % at 750 kbp, -0.7 % at 600 kbp, +1 % at 500 kbp, +6.5 % at 400 kbp, +19 % at 200 kbp)
    SketchPlan plan_d = plan;
    {
        const char *dm_env = getenv("LASH_DEFER_MIN");                       // (read per call: the tests flip it in-process)
        const int64_t defer_min = dm_env ? atoll(dm_env) : 600000;          // bases per work item; < 0: never
        // (judged on the slices as first cut: the quarters at the launch's tail would pull the mean of a few-round launch under the line)
        plan_d.defer = defer_eligible && n_coarse > 0 && defer_min >= 0 && total_words * 16 / n_coarse >= (uint64_t)defer_min;
    }
    TRACE("sketch: planned");

    int rc;
    const size_t n_virtual = plan.bins ? n_genomes : 0;               // binned launches: one k-mer count per genome behind the items'
    if ((rc = reserve(ctx, ctx->partials, bins_partials(plan, prm, n_items, n_genomes) * plan.partial_stride))) return rc;
    if ((rc = reserve(ctx, ctx->item_kmers, (size_t)(n_items + n_virtual + 1) * 4))) return rc;
    if ((rc = reserve(ctx, ctx->counter, 256))) return rc;
    BinsRun bins_run_state;
    if (plan.bins) {
        // entries a genome's work items push: 16 per lane and word for every tile a wave takes part in (masked positions and the
        // idle lanes of a busy wave included); what dirt adds on top (junction walks, a second pass by the compacting kernel) goes to
        // the genome's fallback table if its lists run full
        std::vector<uint64_t> entries(n_genomes, 0);
        const uint64_t tile_words = (uint64_t)plan.threads * SKETCH_WORDS_PER_THREAD;
        for (const WorkItem &w : items) entries[w.genome] += ((w.word_end - w.word_begin + tile_words - 1) / tile_words) * tile_words * 16;
        if ((rc = bins_prepare(ctx, plan, entries, n_genomes, bins_run_state))) return rc;
        // a genome whose lists outgrow the budget after all (the estimate above is coarser than bins_prepare's sizing: ADVICE r4): the
        // call is planned again without bins — a table in global memory per work item, as the comment above promises.  Nothing has
        // been queued yet.
        if (!bins_run_state.fits) return sketch_from(ctx, prm, pk, d_out_images, ev, false);
    }
    const WorkItem *d_items;
    const uint32_t *d_item_begin, *d_item_order = nullptr;
    {
        std::vector<Section> sec = {{items.data(), (size_t)n_items * sizeof(WorkItem), 0},
                                    {item_begin.data(), (size_t)(n_genomes + 1) * 4, 0}};
        lash_packed *mpk = const_cast<lash_packed *>(pk);
        if (pk->direct) {                                          // everything this call needs in ONE copy
            sec.push_back({pk->h_descs.data(), pk->h_descs.size() * sizeof(GenomeDesc), 0});
            sec.push_back({pk->h_tile_begin.data(), pk->h_tile_begin.size() * 4, 0});
            sec.push_back({pk->h_nvalid.data(), pk->h_nvalid.size() * 8, 0});
        }
        const size_t order_sec = sec.size();
        if (!order.empty()) sec.push_back({order.data(), order.size() * 4, 0});
        const size_t total = layout_sections(sec);
        DevBuf &dst = pk->direct ? mpk->tables : ctx->items;
        if ((rc = reserve(ctx, dst, total + 256))) return rc;
        if ((rc = upload_sections(ctx, dst.ptr, sec, total, ctx->stream))) return rc;
        if (ev) HIPCHK(ctx, hipEventRecord(ev->e[2], ctx->stream));            // the sketch stage: record-start bitmaps included
        uint8_t *tb = static_cast<uint8_t *>(dst.ptr);
        d_items = reinterpret_cast<const WorkItem *>(tb + sec[0].off);
        d_item_begin = reinterpret_cast<const uint32_t *>(tb + sec[1].off);
        d_item_order = order.empty() ? nullptr : reinterpret_cast<const uint32_t *>(tb + sec[order_sec].off);
        if (pk->direct) {
            mpk->d_descs = reinterpret_cast<GenomeDesc *>(tb + sec[2].off);
            mpk->d_tile_begin = reinterpret_cast<uint32_t *>(tb + sec[3].off);
            mpk->d_nvalid = reinterpret_cast<uint64_t *>(tb + sec[4].off);
            // flags and counters of the direct pass: dirty [n+1] | nslow [n] | ndel [n] | nonuniform [n] +1 | ndel2 [n]
            HIPCHK(ctx, hipMemsetAsync(pk->d_dirty, 0, (size_t)(5 * (size_t)n_genomes + 2) * 4, ctx->stream));
            if (pk->any_multi) {
                // record starts in BYTE positions: genomes whose records are all equally long (read sets) get theirs computed in
                // the sketch kernel, the others a bitmap every word of which brk_bytes_kernel writes (no memset).  The
                // packed-position bitmap of the fallback is cleared only for the genomes that take it (pack_dirty)
                uint32_t *nonuni = pk->d_dirty + 3 * (size_t)n_genomes + 1;
                HIPCHK(ctx, launch_rec_uniform(pk->d_descs, pk->d_rec_off, n_genomes, pk->n_rec, nonuni, ctx->stream));
                HIPCHK(ctx, launch_brk_bytes(pk->d_descs, pk->d_rec_off, n_genomes, pk->n_rec, nonuni, static_cast<uint32_t *>(pk->brk_bytes.ptr),
                                             ctx->stream));
            }
        }
    }
    if (!ctx->counter_zeroed) {
        HIPCHK(ctx, hipMemsetAsync(ctx->counter.ptr, 0, 256, ctx->stream));     // [0,8) k-mer census, [16,64) zero words,
                                                                                 // [128,256) direct mode's safe load target
        ctx->counter_zeroed = true;
    }
    TRACE("sketch: items uploaded");

    SketchArgs sa{};
    sa.words = static_cast<const uint32_t *>(pk->words.ptr);
    sa.brk = static_cast<const uint32_t *>(pk->brk.ptr);
    sa.zero_words = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(ctx->counter.ptr) + 16);   // zeroed once, never written
    sa.genomes = pk->d_descs;
    sa.nvalid = pk->d_nvalid;
    sa.items = d_items;
    sa.item_order = d_item_order;
    sa.partials = static_cast<uint8_t *>(ctx->partials.ptr);
    sa.gregs = static_cast<uint32_t *>(ctx->gregs.ptr);
    sa.item_kmers = static_cast<uint32_t *>(ctx->item_kmers.ptr);
    sa.safe = static_cast<const uint8_t *>(ctx->counter.ptr) + 128;
    sa.images = d_out_images;
    sa.image_bytes = image_bytes;
    {
        const double alpha0 = hll_alpha(prm->p);
        memcpy(&sa.alpha_bits, &alpha0, 8);
    }
    sa.accumulate = (prm->flags & LASH_F_ACCUMULATE) ? 1 : 0;
    sa.bitflip = prm->algo == LASH_HMH ? xxh3_bitflip128(prm->seed) : xxh3_bitflip64(prm->seed);
    sa.lay = layout_dev(ctx->layout, prm->algo);
    sa.partial_stride = plan.partial_stride;
    sa.nreg32 = plan.nreg32 >> plan.parts_log2;                 // register words of one pass
    sa.k = prm->k;
    sa.p = prm->p;
    ctx->hll_flags_n = 0;
    ctx->hll_flags_on_host = false;
    if (prm->algo == LASH_HLL) {                                // which genomes end with a register above 53 - p (write_hll_header)
        if ((rc = reserve(ctx, ctx->hll_flags, (size_t)n_genomes * 4))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->hll_flags.ptr, 0, (size_t)n_genomes * 4, ctx->stream));
        sa.hll_corner = static_cast<uint32_t *>(ctx->hll_flags.ptr);
        ctx->hll_flags_n = n_genomes;
    }
    if (pk->direct) {
        sa.seq = pk->d_seq;
        sa.brk_bytes = static_cast<const uint32_t *>(pk->brk_bytes.ptr);
        sa.dirty = pk->d_dirty;
        sa.rec_off = pk->d_rec_off;
        sa.nonuniform = pk->d_dirty + 3 * (size_t)n_genomes + 1;
        sa.nslow = pk->d_dirty + n_genomes + 1;
        sa.ndel = sa.nslow + n_genomes;
        sa.ndel2 = pk->d_dirty + 4 * (size_t)n_genomes + 2;
        if (pk->stream_first)   // recent batches were full of finely fragmented dirt: every genome goes straight to the compacting kernel
            HIPCHK(ctx, hipMemsetAsync(pk->d_dirty, 0x01, (size_t)n_genomes * 4, ctx->stream));
        if (!plan.use_lds) {
            if (ev) HIPCHK(ctx, hipEventRecord(ev->e[6], ctx->stream));
            rc = global_run(ctx, plan, sa, n_items, [&](const SketchArgs &a, uint32_t cnt) -> int {
                if (!pk->stream_first) HIPCHK(ctx, launch_sketch(plan_d, a, cnt, ctx->stream, true));
                HIPCHK(ctx, launch_sketch_stream(plan, a, cnt, ctx->stream));
                return LASH_OK;
            });
            if (rc) return rc;
            if (ev) { HIPCHK(ctx, hipEventRecord(ev->e[5], ctx->stream)); ev->direct = true; }
            if (!pk->stream_first) {
                if ((rc = probe_dirty(ctx, const_cast<lash_packed *>(pk), ctx->stream))) return rc;
                ctx->last.direct_launches += n_items ? 1 : 0;
            }
        } else if (plan.bins) {
            if (ev) HIPCHK(ctx, hipEventRecord(ev->e[6], ctx->stream));
            rc = bins_run(ctx, plan, prm, sa, bins_run_state, item_begin, n_items, d_item_begin, [&](const SketchArgs &a, uint32_t, uint32_t cnt) -> int {
                if (!pk->stream_first) HIPCHK(ctx, launch_sketch(plan_d, a, cnt, ctx->stream, true));
                HIPCHK(ctx, launch_sketch_stream(plan, a, cnt, ctx->stream));
                return LASH_OK;
            });
            if (rc) return rc;
            if (ev) { HIPCHK(ctx, hipEventRecord(ev->e[5], ctx->stream)); ev->direct = true; }
            if (!pk->stream_first) {
                if ((rc = probe_dirty(ctx, const_cast<lash_packed *>(pk), ctx->stream))) return rc;
                ctx->last.direct_launches += n_items ? 1 : 0;
            }
        } else {
            if (!pk->stream_first) {
                // diagnostic, LASH_ITEM_TRACE=file (tools/item_trace.py): when and where every workgroup of this launch ran, appended as text
                const char *trace_path = getenv("LASH_ITEM_TRACE");
                unsigned long long *d_trace = nullptr;
                if (trace_path && n_items) {
                    HIPCHK(ctx, hipMalloc(&d_trace, (size_t)n_items * 32));
                    HIPCHK(ctx, hipMemsetAsync(d_trace, 0, (size_t)n_items * 32, ctx->stream));
                    sa.item_trace = d_trace;
                }
                if (ev) HIPCHK(ctx, hipEventRecord(ev->e[6], ctx->stream));       // direct_ms: this one launch
                HIPCHK(ctx, launch_sketch(plan_d, sa, n_items, ctx->stream, true)); // ASCII in; sparse and coarse dirt handled in place
                if (ev) { HIPCHK(ctx, hipEventRecord(ev->e[5], ctx->stream)); ev->direct = true; }
                if (d_trace) {
                    std::vector<unsigned long long> h((size_t)n_items * 4);
                    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                    HIPCHK(ctx, hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost));
                    (void)hipFree(d_trace);
                    sa.item_trace = nullptr;
                    if (FILE *f = fopen(trace_path, "a")) {
                        fprintf(f, "# launch: %u items, %u threads, order %s; columns: slot item genome word_begin word_end start_10ns end_10ns hw_id xcc_id\n",
                                n_items, plan_d.threads, order.empty() ? "item" : "longest first");
                        for (uint32_t b = 0; b < n_items; ++b) {
                            const uint32_t i = order.empty() ? b : order[b];
                            fprintf(f, "%u %u %u %u %u %llu %llu %u %u\n", b, i, items[i].genome, items[i].word_begin, items[i].word_end, h[4ull * i], h[4ull * i + 1],
                                    (unsigned)(h[4ull * i + 2] & 0xFFFFFFFFu), (unsigned)(h[4ull * i + 2] >> 32));
                        }
                        fclose(f);
                    }
                }
                if ((rc = probe_dirty(ctx, const_cast<lash_packed *>(pk), ctx->stream))) return rc;
                ctx->last.direct_launches += n_items ? 1 : 0;
                ctx->last.defer_launches += (n_items && plan_d.defer) ? 1 : 0;
            }
            HIPCHK(ctx, launch_sketch_stream(plan_d, sa, n_items, ctx->stream));    // the flagged genomes, compacted on the fly
        }
    } else if (plan.bins) {
        rc = bins_run(ctx, plan, prm, sa, bins_run_state, item_begin, n_items, d_item_begin, [&](const SketchArgs &a, uint32_t, uint32_t cnt) -> int {
            HIPCHK(ctx, launch_sketch(plan_d, a, cnt, ctx->stream));
            return LASH_OK;
        });
        if (rc) return rc;
    } else if (!plan.use_lds) {
        rc = global_run(ctx, plan, sa, n_items, [&](const SketchArgs &a, uint32_t cnt) -> int { HIPCHK(ctx, launch_sketch(plan_d, a, cnt, ctx->stream)); return LASH_OK; });
        if (rc) return rc;
    } else {
        HIPCHK(ctx, launch_sketch(plan_d, sa, n_items, ctx->stream));
        ctx->last.defer_launches += (n_items && plan_d.defer) ? 1 : 0;
    }
    // the small genomes, whole, on resident workgroups.  Their deleted-byte counts go where lash_ctx_get_timing() will look: with the learnt
    // stream_first every genome's flag is up (d_dirty = 0x01..) and the statistic subtracts ndel2, else ndel
    if (n_sole && (rc = sole_launch(pk->direct ? (pk->stream_first ? sa.ndel2 : sa.ndel) : nullptr))) return rc;
    if (ev) HIPCHK(ctx, hipEventRecord(ev->e[3], ctx->stream));
    TRACE("sketch: launched");

    FinalizeArgs fa{};
    fa.partials = static_cast<const uint8_t *>(ctx->partials.ptr);
    fa.items = d_items;
    fa.genome_item_begin = d_item_begin;
    fa.nvalid = pk->d_nvalid;
    fa.item_kmers = static_cast<const uint32_t *>(ctx->item_kmers.ptr);
    if (plan.bins) {                                               // one partial per genome (bins_apply_kernel), its k-mer count behind the items'
        fa.item_kmers += n_items;
        fa.items = bins_run_state.d_vitems;
        fa.genome_item_begin = bins_run_state.d_vbegin;
        max_slices = 1;
    }
    fa.kmer_counter = static_cast<unsigned long long *>(ctx->counter.ptr);
    fa.images = d_out_images;
    fa.partial_stride = plan.partial_stride;
    fa.partial_base_off = 0;
    fa.image_bytes = image_bytes;
    const double alpha = hll_alpha(prm->p);
    memcpy(&fa.alpha_bits, &alpha, 8);
    fa.algo = prm->algo;
    fa.p = prm->p;
    fa.k = prm->k;
    fa.accumulate = (prm->flags & LASH_F_ACCUMULATE) ? 1 : 0;
    fa.parts_log2 = plan.parts_log2;
    fa.lay = sa.lay;
    fa.src_images = 0;
    fa.hll_corner = sa.hll_corner;
    fa.descs = n_sole ? pk->d_descs : nullptr;
    fa.skip_max_len = n_sole ? sole_max : 0;
    // one finalize workgroup walks all of a genome's partials: fine for a handful of slices, 20 ms for the 4 096 slices of
    // a metagenome-sized input (BASELINE configs[4]) -> fold groups of 32 slices first (until <= 16 heads remain)
    fa.group = 0;
    // (from 9 slices on: finalize_kernel's walk is serial — a dependent load per slice and word — and the tail quarters give a
    // genome up to 16: 300 x 5 Mbp, finalize stage 0.27 -> 0.15 ms)
    // (the fold's grid spans every genome of the batch: with many thousands of genomes and ONE long one, wait for the 33rd slice as before)
    if (max_slices > (n_genomes <= 4096u ? 8u : 32u) && n_genomes <= 65535u)
        for (fa.group = 32u; (max_slices + fa.group - 1) / fa.group > 16u; fa.group *= 32u) {}
    if (plan.bins && bins_write_images(prm)) {
        // bins_apply_kernel wrote the images, headers and census itself
    } else if (all_sole) {
        HIPCHK(ctx, launch_census(fa, n_genomes, ctx->stream));            // every image was written by its one work item
    } else {
        HIPCHK(ctx, launch_reduce_groups(fa, n_genomes, max_slices, ctx->stream));
        HIPCHK(ctx, launch_finalize(fa, n_genomes, ctx->stream));
    }
    if (ev) { HIPCHK(ctx, hipEventRecord(ev->e[4], ctx->stream)); ev->done = true; }
    TRACE("finalize: launched");
    ctx->last_packed.push_back(pk);
    ctx->last.sketch_launches += n_items ? 1 : 0;
    ctx->last.sketch_workgroups = n_items;
    for (uint32_t g = 0; g < n_genomes; ++g)
        ctx->last.packed_bytes += (pk->byte_len[g] + 15) / 16 * 4 + (pk->byte_len[g] + 31) / 32 * 4;
    return LASH_OK;
}

// The amino-acid branch (LASH_F_AMINO; utils.rs:511-563): no pack stage — a lane of aa_sketch_kernel reads a record's bytes itself.
// Work items are ranges of a genome's records; partials and finalize as for nucleotides.
int sketch_aa(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_seq, const uint64_t *d_rec_off, uint64_t n_rec,
              const uint64_t *genome_rec_off, const uint64_t *genome_byte_off, uint32_t n_genomes, uint8_t *d_out_images, bool allow_bins)
{
    int rc;
    if (allow_bins && (rc = timing_begin(ctx))) return rc;            // (the second attempt keeps the first one's event set)
    EvSet *ev = ctx->cur_ev;
    const bool x_low = rule_variant(ctx->layout, prm->algo, prm->flags);   // (HyperMinHash x = low half / HyperLogLog bucket = top bits)
    SketchPlan plan = make_sketch_plan(prm->algo, prm->k, prm->p, x_low, false, allow_bins);
    if (plan.bins) {                                                  // (as in sketch_from: one genome beyond the binned launch's budget)
        const uint64_t budget = bins_budget_bytes(ctx);
        uint64_t big = 0;
        for (uint32_t g = 0; g < n_genomes; ++g)
            big = std::max<uint64_t>(big, (genome_byte_off[g + 1] - genome_byte_off[g]) + 32 * (genome_rec_off[g + 1] - genome_rec_off[g]));
        if (big * 6 + (uint64_t)plan.nreg32 * 4 + (64u << 20) > budget) plan = make_sketch_plan(prm->algo, prm->k, prm->p, x_low, false, false);
    }
    const uint64_t image_bytes = ::image_bytes(ctx->layout, prm->algo, prm->p);
    std::vector<GenomeDesc> descs(n_genomes, GenomeDesc{});
    std::vector<WorkItem> items;
    std::vector<uint32_t> item_begin(n_genomes + 1, 0);
    uint32_t max_slices = 0;
    for (uint32_t g = 0; g < n_genomes; ++g) {
        if (genome_rec_off[g + 1] < genome_rec_off[g] || genome_rec_off[g + 1] > n_rec) return LASH_EINVAL;
        GenomeDesc &d = descs[g];
        d.byte_off = genome_byte_off[g];
        d.byte_len = genome_byte_off[g + 1] - genome_byte_off[g];
        d.rec_begin = genome_rec_off[g];
        d.rec_end = genome_rec_off[g + 1];
        item_begin[g] = (uint32_t)items.size();
        const uint64_t nr = d.rec_end - d.rec_begin;
        if (nr > 0xFFFFFFFFull) return LASH_ELIMIT;
        uint32_t s = 0;
        for (uint64_t r0 = 0; r0 < nr; r0 += AA_RECORDS_PER_ITEM, ++s)
            for (uint32_t part = 0; part < (1u << plan.parts_log2); ++part)
                items.push_back(WorkItem{g, (uint32_t)r0, (uint32_t)std::min<uint64_t>(nr, r0 + AA_RECORDS_PER_ITEM), (s & 0x7FFFu) | (part << 16)});
        max_slices = std::max(max_slices, s);
    }
    item_begin[n_genomes] = (uint32_t)items.size();
    const uint32_t n_items = (uint32_t)items.size();
    const size_t n_virtual = plan.bins ? n_genomes : 0;
    if ((rc = reserve(ctx, ctx->partials, bins_partials(plan, prm, n_items, n_genomes) * plan.partial_stride))) return rc;
    if ((rc = reserve(ctx, ctx->item_kmers, (size_t)(n_items + n_virtual + 1) * 4))) return rc;
    if ((rc = reserve(ctx, ctx->counter, 256))) return rc;
    BinsRun bins_run_state;
    if (plan.bins) {
        // a lane pushes 16 entries per trip of its loop — 16 residues of a record, or the fetch of the next one — and the idle lanes of a
        // busy wave push along: residues + 32 per record, and a quarter on top
        std::vector<uint64_t> entries(n_genomes);
        for (uint32_t g = 0; g < n_genomes; ++g) {
            const uint64_t e = descs[g].byte_len + 32 * (descs[g].rec_end - descs[g].rec_begin);
            entries[g] = e + e / 4 + (uint64_t)plan.threads * 256;
        }
        if ((rc = bins_prepare(ctx, plan, entries, n_genomes, bins_run_state))) return rc;
        if (!bins_run_state.fits)                                    // (as in sketch_from: planned again without bins; nothing has been queued yet)
            return sketch_aa(ctx, prm, d_seq, d_rec_off, n_rec, genome_rec_off, genome_byte_off, n_genomes, d_out_images, false);
    }
    std::vector<Section> sec = {{items.data(), (size_t)n_items * sizeof(WorkItem), 0}, {item_begin.data(), (size_t)(n_genomes + 1) * 4, 0},
                                {descs.data(), descs.size() * sizeof(GenomeDesc), 0}};
    const size_t total = layout_sections(sec);
    if ((rc = reserve(ctx, ctx->items, total + 256))) return rc;
    if ((rc = upload_sections(ctx, ctx->items.ptr, sec, total, ctx->stream))) return rc;
    uint8_t *tb = static_cast<uint8_t *>(ctx->items.ptr);
    if (!ctx->counter_zeroed) {
        HIPCHK(ctx, hipMemsetAsync(ctx->counter.ptr, 0, 256, ctx->stream));
        ctx->counter_zeroed = true;
    }
    if (ev) HIPCHK(ctx, hipEventRecord(ev->e[2], ctx->stream));
    SketchArgs sa{};
    sa.seq = d_seq;
    sa.rec_off = d_rec_off;
    sa.genomes = reinterpret_cast<const GenomeDesc *>(tb + sec[2].off);
    sa.items = reinterpret_cast<const WorkItem *>(tb + sec[0].off);
    sa.partials = static_cast<uint8_t *>(ctx->partials.ptr);
    sa.gregs = static_cast<uint32_t *>(ctx->gregs.ptr);
    sa.item_kmers = static_cast<uint32_t *>(ctx->item_kmers.ptr);
    sa.images = d_out_images;
    sa.image_bytes = image_bytes;
    const double alpha = hll_alpha(prm->p);
    memcpy(&sa.alpha_bits, &alpha, 8);
    sa.accumulate = (prm->flags & LASH_F_ACCUMULATE) ? 1 : 0;
    sa.bitflip = prm->algo == LASH_HMH ? xxh3_bitflip128(prm->seed) : xxh3_bitflip64(prm->seed);
    sa.lay = layout_dev(ctx->layout, prm->algo);
    sa.partial_stride = plan.partial_stride;
    sa.nreg32 = plan.nreg32 >> plan.parts_log2;
    sa.k = prm->k;
    sa.p = prm->p;
    ctx->hll_flags_n = 0;
    ctx->hll_flags_on_host = false;
    if (prm->algo == LASH_HLL) {
        if ((rc = reserve(ctx, ctx->hll_flags, (size_t)n_genomes * 4))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->hll_flags.ptr, 0, (size_t)n_genomes * 4, ctx->stream));
        sa.hll_corner = static_cast<uint32_t *>(ctx->hll_flags.ptr);
        ctx->hll_flags_n = n_genomes;
    }
    if (plan.bins) {
        rc = bins_run(ctx, plan, prm, sa, bins_run_state, item_begin, n_items, reinterpret_cast<const uint32_t *>(tb + sec[1].off),
                      [&](const SketchArgs &a, uint32_t, uint32_t cnt) -> int { HIPCHK(ctx, launch_sketch_aa(plan, a, cnt, ctx->stream)); return LASH_OK; });
        if (rc) return rc;
    } else if (!plan.use_lds) {
        rc = global_run(ctx, plan, sa, n_items, [&](const SketchArgs &a, uint32_t cnt) -> int { HIPCHK(ctx, launch_sketch_aa(plan, a, cnt, ctx->stream)); return LASH_OK; });
        if (rc) return rc;
    } else {
        HIPCHK(ctx, launch_sketch_aa(plan, sa, n_items, ctx->stream));
    }
    if (ev) HIPCHK(ctx, hipEventRecord(ev->e[3], ctx->stream));
    FinalizeArgs fa{};
    fa.partials = static_cast<const uint8_t *>(ctx->partials.ptr);
    fa.items = sa.items;
    fa.genome_item_begin = reinterpret_cast<const uint32_t *>(tb + sec[1].off);
    fa.nvalid = nullptr;                                           // every item is live
    fa.item_kmers = static_cast<const uint32_t *>(ctx->item_kmers.ptr);
    if (plan.bins) {
        fa.item_kmers += n_items;
        fa.items = bins_run_state.d_vitems;
        fa.genome_item_begin = bins_run_state.d_vbegin;
        max_slices = 1;
    }
    fa.kmer_counter = static_cast<unsigned long long *>(ctx->counter.ptr);
    fa.images = d_out_images;
    fa.partial_stride = plan.partial_stride;
    fa.partial_base_off = 0;
    fa.image_bytes = image_bytes;
    memcpy(&fa.alpha_bits, &alpha, 8);
    fa.algo = prm->algo;
    fa.p = prm->p;
    fa.k = prm->k;
    fa.accumulate = sa.accumulate;
    fa.parts_log2 = plan.parts_log2;
    fa.lay = sa.lay;
    fa.src_images = 0;
    fa.hll_corner = sa.hll_corner;
    fa.group = 0;
    // (from 9 slices on: finalize_kernel's walk is serial — a dependent load per slice and word — and the tail quarters give a
    // genome up to 16: 300 x 5 Mbp, finalize stage 0.27 -> 0.15 ms)
    if (max_slices > 8u && n_genomes <= 65535u)
        for (fa.group = 32u; (max_slices + fa.group - 1) / fa.group > 16u; fa.group *= 32u) {}
    if (!(plan.bins && bins_write_images(prm))) {                      // (else bins_apply_kernel wrote the images, headers and census itself)
        HIPCHK(ctx, launch_reduce_groups(fa, n_genomes, max_slices, ctx->stream));
        HIPCHK(ctx, launch_finalize(fa, n_genomes, ctx->stream));
    }
    if (ev) { HIPCHK(ctx, hipEventRecord(ev->e[4], ctx->stream)); ev->done = true; }
    ctx->cur_ev = nullptr;
    ctx->last.sketch_launches += n_items ? 1 : 0;
    ctx->last.sketch_workgroups = n_items;
    return LASH_OK;
}

}  // namespace lashi
