// lash_ctx.h — what hangs off a lash_ctx / lash_packed, and the small host helpers every translation unit of liblash_gfx950
// that implements part of the C ABI shares (lash_api.hip: sketch side; sketch_set.hip: resident dist side).  Private to the library.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "../../include/lash_gfx950.h"
#include "lash_kernels.h"
#include "ull_estimators.h"

using namespace lash;

namespace lashi {

struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct EvSet {
    hipEvent_t e[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // pack start/end; sketch stage start/end;
    bool pack = false, done = false, direct = false;                            // finalize end; end [5] and start [6] of the direct kernel
};

// LASH_TRACE_HOST=1: host-side microsecond marks of one call on stderr (tools/, DESIGN.md "Host cost of a call").
// The object lives in the context that makes the call: contexts on different host threads never share it.
struct HostTrace {
    bool on = false;
    std::chrono::steady_clock::time_point t0;
    void begin()
    {
        static const bool env_on = getenv("LASH_TRACE_HOST") != nullptr;
        on = env_on;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void end() { on = false; }
    void mark(const char *what) const
    {
        if (!on) return;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[lash host] %8.1f us  %s\n", us, what);
    }
};

struct HostStage {
    void *ptr = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
};

const lash_layout kDefaultLayout = {{0, 1, 2, 3}, 0, 0, 0, 0, "", "azspl", "l", 0, 0, {0, 0, 0, 0, 0, 0}};

}  // namespace lashi
using namespace lashi;

struct lash_packed {
    uint32_t n_genomes = 0;
    DevBuf words, brk;
    DevBuf tables;                        // one upload: [descs | tile_begin | nvalid] (+ the work items in direct mode)
    GenomeDesc *d_descs = nullptr;        // sections of `tables`
    uint32_t *d_tile_begin = nullptr;
    uint64_t *d_nvalid = nullptr;
    DevBuf tiles, lookback;               // pack scratch (lookback: descriptors + flag + ticket counters + dirty flags)
    uint64_t total_words = 0, total_brk = 0;
    std::vector<uint64_t> byte_len;      // per genome, host copy (upper bound of surviving bases)
    uint32_t *error_flag = nullptr;      // device word set by the pack kernel if a look-back spin hit its bound
    // direct mode (lash_sketch_batch[_device] without LASH_F_NO_DIRECT): the pack launch is deferred and restricted,
    // on the device, to the genomes the direct sketch pass flagged dirty
    bool direct = false, any_multi = false;
    const uint8_t *d_seq = nullptr;
    uint32_t *d_dirty = nullptr;         // inside `lookback` (zeroed by the same memset): [n+1] dirty flags, then [n] slow
                                         // wave-tile counts and [n] in-place deleted-byte counts of the direct pass
    bool stream_first = false;       // direct mode: skip the optimistic pass, every genome goes to stream_sketch_kernel
    DevBuf tile_begin_c, brk_bytes, fq;                       // fq: FASTQ file table + scratch of fastq_check.hip
    std::vector<GenomeDesc> h_descs;     // host copies, uploaded together with the work items
    std::vector<uint32_t> h_tile_begin;
    std::vector<uint64_t> h_nvalid;
    const uint64_t *d_rec_off = nullptr;
    uint64_t n_rec = 0;
    bool owned_by_ctx = false;           // the scratch instance reused by lash_sketch_batch_device
    uint32_t code_tab4 = 0;              // the 2-bit code table the words were packed under (layout.base_code, swapped for kmer_lsb_first): a
                                         // packed batch belongs to that table; lash_sketch_packed_device refuses another (round 6)
};

struct lash_ctx;
// N serialized sketches resident in HBM + what the pair kernels derive from them once (sketch_set.hip)
struct lash_sketch_set {
    int device = 0, algo = 0, p = 0;
    uint32_t n = 0;
    uint32_t hdr = 0;
    uint64_t stride = 0;                   // bytes between consecutive images
    uint32_t hmh_be = 0;
    const uint8_t *d_images = nullptr;     // [n][stride], set order
    DevBuf images;                         // owned copy (lash_sketch_set_create); empty when borrowed (.._create_device)
    // HyperMinHash: register bit planes (pair_planes.hip)
    DevBuf S, T, nzcount;
    uint32_t ldT = 0, n_pad = 0;
    bool have_S = false, have_T = false, full = false;   // full: every register of every member is non-zero
    // HyperLogLog: range of register values, threshold bitmaps [n][band][m/32] for the range the set was prepared with
    DevBuf lohi, bm;
    uint32_t lo = 0xFFFFFFFFu, hi = 0, bm_lo = 0, bm_band = 0;
    bool have_range = false, have_bm = false;
    // per-member cardinalities (lash_sketch_set_cardinalities keeps them); HyperMinHash: the members at or below 2^19 distinct
    // k-mers ("small": hyperminhash walks 65 536 cells for a pair of them) and their cell-probability vectors [n_small][65536] f64
    std::vector<double> card;
    std::vector<uint32_t> small_idx;
    DevBuf ec_vec;
    bool have_ec_vec = false;
};

int lash_set_build_planes(lash_ctx *ctx, lash_sketch_set *s, bool want_T);   // sketch_set.hip

struct lash_ctx {
    int device = 0;
    int cu_count = 256;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    lash_layout layout = kDefaultLayout;  // SURVEY App. D's unknowns as data (lash_ctx_set_layout)
    HostTrace trace;
    bool timing = false;
    std::deque<EvSet> ev_pool;           // one event set per timed (chunk of a) call; deque: stable addresses on growth
    size_t ev_used = 0;
    EvSet *cur_ev = nullptr;
    lash_timing last{};
    HostStage ring[32];                  // pinned staging for the small per-call tables
    unsigned ring_next = 0;
    std::vector<const lash_packed *> last_packed;   // what the last sketch call consumed (for bases_last / error flags)
    DevBuf items, item_begin, item_kmers, partials, gregs, counter;   // items: [work items | item_begin] of a sketch call
    uint64_t bins_budget = 0;     // bytes one group of a binned launch may take (0: not asked yet; lash_plan.hip bins_budget_bytes)
    int bins_slab_fill = -1; size_t bins_slab_clean = 0;   // the fallback tables rest EMPTY between launches: with which byte (0x00 ull / 0xFF hll; -1: unknown), how far
    DevBuf bins_lists, bins_meta, bins_slab;   // binned launches (SketchPlan::bins): entry lists, tables + counters, fallback tables of one genome group
    bool counter_zeroed = false;
    DevBuf sole_tab, sole_brk, sole_state;           // persistent small-genome launches (sole_kernels.hip): [chunks | genome byte offsets | per-workgroup
                                         // counts | ticket]; record starts as bits at absolute byte positions
    bool last_sole_only = false;         // the last sketch call ran on that kernel alone (lash_timing::bases_last comes from its census)
    DevBuf st_seq, st_rec, st_img;       // staging for the synchronous host-buffer entries (files_raw, merge, pair statistics)
    // lash_sketch_batch[_async]: two staging slots and two copy streams, so that the H2D copy of batch n+1 and the D2H copy of
    // batch n-1 run while the kernels of batch n do (PCIe Gen5 moves 1 B/base: the host-buffer entry is link-bound)
    struct AsyncSlot { DevBuf seq, rec, img; hipEvent_t h2d = nullptr, kern = nullptr, d2h = nullptr; bool busy = false; };
    AsyncSlot slot[2];
    unsigned slot_next = 0;
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    lash_packed scratch;                 // packed batch of lash_sketch_batch[_device]
    lash_sketch_set pl_ref, pl_qry;      // lash_hmh_pair_counts[_device]: bit planes of the call's images (buffers reused across calls)
    DevBuf hll_bm_ref, hll_bm_qry, hll_lohi;   // lash_hll_pair_union_stats*: threshold bitmaps [n][band][m/32], range of register values
    DevBuf ec_ref, ec_qry, ec_x, ec_card;   // lash_hmh_pair_expected_collisions: cell vectors [n][65536] f64, products, cardinalities
    std::vector<double> ec_qry_cards;    // the small query cardinalities whose vectors ec_qry holds (reused across row blocks)
    DevBuf hll_flags;                    // [hll_flags_n] per genome of the last HyperLogLog sketch call: a register > 53 - p
    uint32_t hll_flags_n = 0;            // (lash_ctx_hll_inexact_sums)
    bool hll_flags_on_host = false;      // the list below stands for the flags (hll_replay_sums has dealt with the others)
    std::vector<uint32_t> hll_left;      // genomes of the last call still in the corner after the replay
    DevBuf replay_rec, replay_img;       // hll_replay_sums: record offsets and image of a prefix sketch
    std::vector<uint32_t> bad_files;     // lash_ctx_format_errors(): files of the last raw call whose FASTQ structure broke
    uint32_t raw_files_pending = 0;      // files of a lash_sketch_files_raw_device call whose error flags have not been read yet
    // direct-mode feedback: the dirty-tile count of the last direct call comes back through a pinned word, is looked at
    // (never waited for) by the next call, and switches the optimistic pass off while batches keep turning out dirty
    uint32_t *probe_host = nullptr;      // pinned: [0] = dirty tiles of the last probed call
    hipEvent_t probe_ev = nullptr;
    bool probe_pending = false;
    uint32_t probe_tiles = 0;            // all tiles of that call
    float dirty_frac = 0.f;              // last observed fraction of tiles in dirty genomes
    uint32_t direct_skipped = 0;         // calls since the optimistic pass was last tried
};

namespace {

int fail(lash_ctx *ctx, int code, const char *what, hipError_t e)
{
    if (ctx) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        ctx->err = buf;
    }
    return code;
}

#define TRACE(what) ctx->trace.mark(what)
// marks end with the entry point that began them, on every return path
struct TraceScope {
    HostTrace &t;
    explicit TraceScope(HostTrace &tr) : t(tr) { t.begin(); }
    ~TraceScope() { t.end(); }
};

#define HIPCHK(ctx, expr)                                              \
    do {                                                               \
        hipError_t e__ = (expr);                                       \
        if (e__ != hipSuccess) return fail((ctx), e__ == hipErrorOutOfMemory ? LASH_ENOMEM : LASH_EHIP, #expr, e__); \
    } while (0)

// grow-only device buffer; growing synchronizes the stream first because queued kernels may still use the old one
int reserve(lash_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return LASH_OK;
    if (b.ptr) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipMalloc(&b.ptr, want));
    b.cap = want;
    return LASH_OK;
}

void release(DevBuf &b)
{
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.cap = 0;
}

// Small host tables go through a ring of pinned buffers, so the async copy never reads a dead std::vector and a
// call never has to drain the stream.
int upload(lash_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, hipStream_t stream = nullptr)
{
    if (bytes == 0) return LASH_OK;
    if (!stream) stream = ctx->stream;
    HostStage &hs = ctx->ring[ctx->ring_next++ % 32];
    if (hs.pending) { HIPCHK(ctx, hipEventSynchronize(hs.done)); hs.pending = false; }
    if (!hs.done) HIPCHK(ctx, hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
    if (hs.cap < bytes) {
        if (hs.ptr) HIPCHK(ctx, hipHostFree(hs.ptr));
        hs.ptr = nullptr;
        hs.cap = 0;
        HIPCHK(ctx, hipHostMalloc(&hs.ptr, bytes + bytes / 4 + 4096, hipHostMallocDefault));
        hs.cap = bytes + bytes / 4 + 4096;
    }
    memcpy(hs.ptr, h_src, bytes);
    HIPCHK(ctx, hipMemcpyAsync(d_dst, hs.ptr, bytes, hipMemcpyHostToDevice, stream));
    HIPCHK(ctx, hipEventRecord(hs.done, stream));
    hs.pending = true;
    return LASH_OK;
}

// several small tables, one pinned staging buffer, one copy: sec[i] lands at d_base + off[i] (offsets 256-B aligned)
struct Section { const void *src; size_t bytes; size_t off; };
size_t layout_sections(std::vector<Section> &sec)
{
    size_t at = 0;
    for (Section &x : sec) { x.off = at; at += (x.bytes + 255) & ~(size_t)255; }
    return at;
}
int upload_sections(lash_ctx *ctx, void *d_base, const std::vector<Section> &sec, size_t total, hipStream_t stream)
{
    if (total == 0) return LASH_OK;
    HostStage &hs = ctx->ring[ctx->ring_next++ % 32];
    if (hs.pending) { HIPCHK(ctx, hipEventSynchronize(hs.done)); hs.pending = false; }
    if (!hs.done) HIPCHK(ctx, hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
    if (hs.cap < total) {
        if (hs.ptr) HIPCHK(ctx, hipHostFree(hs.ptr));
        hs.ptr = nullptr;
        hs.cap = 0;
        HIPCHK(ctx, hipHostMalloc(&hs.ptr, total + total / 4 + 4096, hipHostMallocDefault));
        hs.cap = total + total / 4 + 4096;
    }
    for (const Section &x : sec)
        if (x.bytes) memcpy(static_cast<uint8_t *>(hs.ptr) + x.off, x.src, x.bytes);
    HIPCHK(ctx, hipMemcpyAsync(d_base, hs.ptr, total, hipMemcpyHostToDevice, stream));
    HIPCHK(ctx, hipEventRecord(hs.done, stream));
    hs.pending = true;
    return LASH_OK;
}

int timing_begin(lash_ctx *ctx)      // sets ctx->cur_ev to a fresh event set (or nullptr when timing is off / exhausted)
{
    ctx->cur_ev = nullptr;
    if (!ctx->timing) return LASH_OK;
    if (ctx->ev_used == ctx->ev_pool.size()) {
        if (ctx->ev_pool.size() >= 16384) return LASH_OK;      // stop recording, keep running
        EvSet s;
        for (auto &e : s.e) HIPCHK(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(s);
    }
    EvSet *s = &ctx->ev_pool[ctx->ev_used++];
    s->pack = false;
    s->done = false;
    s->direct = false;
    ctx->cur_ev = s;
    return LASH_OK;
}

double hll_alpha(int p)
{
    switch (p) {
    case 4: return 0.673;
    case 5: return 0.697;
    case 6: return 0.709;
    default: return 0.7213 / (1.0 + 1.079 / (double)(1u << p));
    }
}

// ---- layout (include/lash_gfx950.h) ----
size_t field_bytes(char c)
{
    switch (c) {
    case 'a': case 'z': case 's': case 'Q': case 'l': return 8;
    case 'Z': case 'P': case 'L': return 4;
    case 'p': return 1;
    default: return (size_t)-1;
    }
}
const char *header_tpl(const lash_layout &lay, int algo)
{
    return algo == LASH_HMH ? lay.hmh_header : algo == LASH_HLL ? lay.hll_header : lay.ull_header;
}
uint64_t header_bytes(const lash_layout &lay, int algo)
{
    uint64_t n = 0;
    const char *t = header_tpl(lay, algo);
    for (int i = 0; i < 8 && t[i]; ++i) n += field_bytes(t[i]);
    return n;
}
bool layout_ok(const lash_layout &lay)
{
    unsigned seen = 0;
    for (int i = 0; i < 4; ++i) { if (lay.base_code[i] > 3) return false; seen |= 1u << lay.base_code[i]; }
    if (seen != 15u) return false;
    for (int a = 0; a < 3; ++a) {
        const char *t = header_tpl(lay, a);
        int i = 0;
        for (; i < 8 && t[i]; ++i) if (field_bytes(t[i]) == (size_t)-1) return false;
        if (i == 8) return false;
    }
    return true;
}
// The register rule's compile-time variant of the sketch kernels (sketch_rules.h, add_kmer): HyperMinHash with x = the LOW half of
// xxh3_128 (SURVEY App. D switch U1: the context layout, or LASH_F_HMH_X_LOW per call), HyperLogLog with the bucket in the TOP p bits
// (U3).  Every kernel family — direct, stream, persistent small-genome, packed, amino-acid, deferring — exists in both forms.
bool rule_variant(const lash_layout &lay, int algo, uint32_t flags)
{
    if (algo == LASH_HMH) return (flags & LASH_F_HMH_X_LOW) != 0 || lay.hmh_x_low;
    return algo == LASH_HLL && lay.hll_bucket_high;
}

LayoutDev layout_dev(const lash_layout &lay, int algo)
{
    LayoutDev d{};
    // layout.kmer_lsb_first (U5: a k-mer's first base in its LEAST significant bits): the lsb-first value of a k-mer is the msb-first
    // value of its reverse complement under the code table with every letter's code replaced by its complement's (sketch_rules.h,
    // above process_word), and the canonical k-mer is the minimum over {k-mer, reverse complement} either way — so the kernels
    // run unchanged on the swapped table.  comp_mask (code[A] ^ code[T]) is the same for both tables.
    const bool sw = lay.kmer_lsb_first != 0;
    const uint32_t A = lay.base_code[sw ? 3 : 0], Cc = lay.base_code[sw ? 2 : 1], G = lay.base_code[sw ? 1 : 2], T = lay.base_code[sw ? 0 : 3];
    d.code_lo = (A << 8) | (Cc << 24);                   // keys (byte & 7): A = 1, C = 3
    d.code_hi = T | (G << 24);                           // T = 4, G = 7
    d.code_tab4 = A | (Cc << 8) | (G << 16) | (T << 24); // index = hypothesis code A,C,G,T = 0,1,2,3
    d.comp_mask = (A ^ T) * 0x55555555u;
    d.hdr_bytes = (uint32_t)header_bytes(lay, algo);
    const char *t = header_tpl(lay, algo);
    for (int i = 0; i < 8 && t[i]; ++i) d.hdr_tpl |= (uint64_t)(uint8_t)t[i] << (8 * i);
    d.hmh_reg_be = lay.hmh_reg_be;
    d.kmer_lsb_first = lay.kmer_lsb_first;
    d.hll_bucket_high = lay.hll_bucket_high;
    d.aa_code_base = lay.aa_code_zero_based ? 0u : 1u;
    return d;
}

size_t image_bytes(const lash_layout &lay, int algo, int p)
{
    switch (algo) {
    case LASH_HMH: return header_bytes(lay, algo) + (size_t)HMH_M * 2;                            // [header] 16384 x u16
    case LASH_HLL: return (p >= 4 && p <= 16) ? header_bytes(lay, algo) + ((size_t)1 << p) : 0;   // bincode(alpha, zero, sum, p, len) + m
    case LASH_ULL: return (p >= 3 && p <= 26) ? header_bytes(lay, algo) + ((size_t)1 << p) : 0;   // bincode(Vec<u8>)
    default: return 0;
    }
}

}  // namespace
