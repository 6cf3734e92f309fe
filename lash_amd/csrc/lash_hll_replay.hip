// lash_hll_replay.hip — HyperLogLog's incremental `sum` in the corner where it depends on the order of its roundings (a register above 53 - p),
// replayed from prefix sketches so that the 8 header bytes equal what streaming_algorithms' push leaves (utils.rs:411-413 -> push_hash64).
// Split out of lash_api.hip in round 6; the sketch entries call hll_replay_sums() through lash_internal.h.
#include "lash_ctx.h"
#include "lash_internal.h"

namespace lashi {

// ---- HyperLogLog: the incremental `sum` of genomes in the > 53 - p corner, replayed -------------------------------------------
// streaming_algorithms keeps `sum` per k-mer: sum -= 2^-old; sum += 2^-new (utils.rs:411-413 -> push_hash64).
// Every such pair is exact in f64 — the terms are multiples of 2^(p-53) and the sum only falls — except where a term BELOW that
// grid is involved: the k-mer that lifts a register above 53 - p (one in 2^(52-p)), or one that later overwrites such a register.
// There the result depends on the value `sum` had at that moment, i.e. on the registers of the genome's PREFIX.  So:
//   * the registers of the final image name the buckets above 53 - p;
//   * a bucket's value in the sketch of a prefix is monotone in the prefix length: a binary search over cut positions — each probe
//     one ordinary sketch call on the records cut at that byte — finds the k-mer that did it (prefixes are cut by BYTES, so
//     deleted bytes, records and k-mer order need no special care: a k-mer belongs to a prefix iff its last base does);
//   * the sketch of the prefix just before it carries the incremental sum up to there (no sub-grid term yet: its header IS
//     exact) and the register's old value; the two f64 operations of that k-mer are then done here, on the host, in IEEE double;
//   * from there to the genome's end (or the next such k-mer) every step is exact, so the net change is the difference of the
//     on-grid parts of the two register states — one more exact addition.
// Returns the number of genomes redone; `left` lists those it had to leave (accumulating calls hold registers the replay cannot
// see).  Synchronous; runs ~25 small sketch calls per flagged genome (one genome in ~10^4 at p = 14).
double grid_sum(const uint8_t *regs, size_t m, int p)
{
    uint32_t hist[72] = {0};
    for (size_t i = 0; i < m; ++i) ++hist[regs[i] < 71 ? regs[i] : 71];
    double s = 0.0;                                                // multiples of 2^(p-53) below 2^p: exact in any order
    for (int r = 0; r <= 53 - p; ++r) s += (double)hist[r] * ldexp(1.0, -r);
    return s;
}
int hll_sum_field_offset(const lash_layout &lay)
{
    const char *t = header_tpl(lay, LASH_HLL);
    int at = 0;
    for (int i = 0; i < 8 && t[i]; ++i) {
        switch (t[i]) {
        case 's': return at;
        case 'a': case 'z': case 'Q': case 'l': at += 8; break;
        case 'Z': case 'P': case 'L': at += 4; break;
        case 'p': at += 1; break;
        default: break;
        }
    }
    return -1;
}

// the replay's probes are ordinary sketch calls: whatever they leave behind in the context — timing switch and sums, the list of
// packed batches the user's call consumed, the direct pass's dirt feedback — is put back on EVERY way out
struct ReplayRestore {
    lash_ctx *c;
    lash_timing last; bool timing; std::vector<const lash_packed *> packed; float dirty_frac; uint32_t direct_skipped; bool sole_only;
    explicit ReplayRestore(lash_ctx *x) : c(x), last(x->last), timing(x->timing), packed(x->last_packed), dirty_frac(x->dirty_frac),
                                          direct_skipped(x->direct_skipped), sole_only(x->last_sole_only) {}
    ~ReplayRestore()
    {
        c->last = last; c->timing = timing; c->last_packed = packed; c->dirty_frac = dirty_frac; c->direct_skipped = direct_skipped;
        c->last_sole_only = sole_only;
        c->probe_pending = false;                                     // (a probe's feedback is not the user's batch's)
    }
};

// One genome (or one streamed chunk of a file) of the replay.  `rec`: its records' absolute offsets into d_seq; `fin`: its image
// AFTER (header + registers, on the host); `base`: NULL, or the registers the sketch held BEFORE these records (a streamed file's
// earlier chunks: every prefix sketch is united with them before it is looked at).  (S, G, carry): the incremental sum and the on-grid
// sum of the registers at the moment S was last brought up to date — carried from chunk to chunk of a streamed file; in: carry == false
// means "no register has been above 53 - p so far" (S is then the exact sum, taken from the registers).  Out: S is the reference's
// incremental value after these records, G the on-grid sum of `fin`'s registers, carry = true.
int hll_replay_one(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_seq, const std::vector<uint64_t> &rec, const uint8_t *fin,
                          const uint8_t *base, double &S, double &G, bool &carry)
{
    const int p = prm->p;
    const size_t hdr = header_bytes(ctx->layout, LASH_HLL), m = (size_t)1 << p, ib = hdr + m;
    int rc;
    if ((rc = reserve(ctx, ctx->replay_img, ib + 64))) return rc;
    if ((rc = reserve(ctx, ctx->replay_rec, (rec.size() + 1) * 8))) return rc;
    // the registers after the genome's records cut at byte `cut` (absolute offset into d_seq), united with `base` -> out
    auto prefix = [&](uint64_t cut, std::vector<uint8_t> &out) -> int {
        size_t i = (size_t)(std::upper_bound(rec.begin(), rec.end(), cut) - rec.begin());   // records [0, i-1) lie wholly before the cut
        if (i == 0) i = 1;
        std::vector<uint64_t> pr(rec.begin(), rec.begin() + i);
        if (pr.back() < cut) pr.push_back(cut);                                             // the record the cut falls into, truncated
        const uint64_t n = pr.size() - 1, goff[2] = {0, n}, gbo[2] = {pr.front(), pr.back()};
        HIPCHK(ctx, hipMemcpy(ctx->replay_rec.ptr, pr.data(), pr.size() * 8, hipMemcpyHostToDevice));
        int r = lash_sketch_batch_device(ctx, prm, d_seq, static_cast<const uint64_t *>(ctx->replay_rec.ptr), n, goff, gbo, 1,
                                         static_cast<uint8_t *>(ctx->replay_img.ptr));
        if (r) return r;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        out.resize(ib);
        HIPCHK(ctx, hipMemcpy(out.data(), ctx->replay_img.ptr, ib, hipMemcpyDeviceToHost));
        if (base) for (size_t j = 0; j < m; ++j) out[hdr + j] = std::max(out[hdr + j], base[j]);
        return LASH_OK;
    };
    struct Event { uint64_t cut; uint32_t j; uint8_t neu, old; std::vector<uint8_t> before; };
    std::vector<Event> events;
    std::vector<std::pair<uint32_t, uint8_t>> todo;                 // (bucket, value above the grid) whose k-mer is to be found
    for (size_t j = 0; j < m; ++j)
        if (fin[hdr + j] > 53 - p && (!base || fin[hdr + j] != base[j])) todo.push_back({(uint32_t)j, fin[hdr + j]});
    std::vector<uint8_t> probe;
    while (!todo.empty()) {
        const auto [j, val] = todo.back();
        todo.pop_back();
        uint64_t lo = rec.front(), hi = rec.back();                // prefix(lo) lacks the value, prefix(hi) has it
        while (hi - lo > 1) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if ((rc = prefix(mid, probe))) return rc;
            if (probe[hdr + j] >= val) hi = mid; else lo = mid;
        }
        Event e;
        e.cut = hi; e.j = j; e.neu = val;
        if ((rc = prefix(hi - 1, e.before))) return rc;
        e.old = e.before[hdr + j];
        // an earlier k-mer OF THESE RECORDS had already put this bucket above the grid (one of an earlier chunk is `base`'s: no event here)
        if (e.old > 53 - p && (!base || e.old != base[j])) todo.push_back({j, e.old});
        events.push_back(std::move(e));
    }
    if (events.empty()) {                                           // nothing of these records touches the corner
        if (carry) { S += grid_sum(fin + hdr, m, p) - G; }
        else memcpy(&S, fin + hll_sum_field_offset(ctx->layout), 8);
        G = grid_sum(fin + hdr, m, p);
        return LASH_OK;
    }
    std::sort(events.begin(), events.end(), [](const Event &a, const Event &b) { return a.cut < b.cut; });
    // up to the first such k-mer every step was exact: the sum is that of the registers (on their grid), or the carried value plus
    // the exact net change since it was taken
    if (carry) S += grid_sum(events[0].before.data() + hdr, m, p) - G;
    else S = grid_sum(events[0].before.data() + hdr, m, p);
    double grid_after = 0.0;
    for (size_t i = 0; i < events.size(); ++i) {
        const Event &e = events[i];
        if (i) S += grid_sum(e.before.data() + hdr, m, p) - grid_after;      // exact steps in between: their net change
        // the k-mer's own update, rounded as the crate's is: ONE operation, sum -= 2^-old - 2^-new (the difference is exact unless
        // new - old > 53; ADVICE r4: the two-step form differs when the bucket's old value lies above 53 - p as well)
        S -= ldexp(1.0, -(int)e.old) - ldexp(1.0, -(int)e.neu);
        std::vector<uint8_t> after(e.before.begin() + hdr, e.before.end());
        after[e.j] = e.neu;
        grid_after = grid_sum(after.data(), m, p);
    }
    G = grid_sum(fin + hdr, m, p);
    S += G - grid_after;
    carry = true;
    return LASH_OK;
}

int hll_replay_sums(lash_ctx *ctx, const lash_params *prm0, const uint8_t *d_seq, const uint64_t *d_rec_off, const uint64_t *h_rec_off,
                           const uint64_t *genome_rec_off, uint8_t *d_images, uint8_t *h_images, const std::vector<uint32_t> &flagged,
                           std::vector<uint32_t> &left)
{
    left.clear();
    if (flagged.empty()) return LASH_OK;
    const int p = prm0->p, sum_at = hll_sum_field_offset(ctx->layout);
    const size_t hdr = header_bytes(ctx->layout, LASH_HLL), m = (size_t)1 << p, ib = hdr + m;
    if (sum_at < 0 || (prm0->flags & (LASH_F_ACCUMULATE | LASH_F_AMINO))) { left = flagged; return LASH_OK; }
    lash_params prm = *prm0;
    ReplayRestore restore(ctx);
    ctx->timing = false;
    int rc = LASH_OK;
    std::vector<uint8_t> fin(ib);
    for (uint32_t g : flagged) {
        const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1], nr = r1 - r0;
        std::vector<uint64_t> rec(nr + 1);
        if (h_rec_off) memcpy(rec.data(), h_rec_off + r0, (nr + 1) * 8);
        else HIPCHK(ctx, hipMemcpy(rec.data(), d_rec_off + r0, (nr + 1) * 8, hipMemcpyDeviceToHost));
        if (h_images) memcpy(fin.data(), h_images + (size_t)g * ib, ib);
        else HIPCHK(ctx, hipMemcpy(fin.data(), d_images + (size_t)g * ib, ib, hipMemcpyDeviceToHost));
        double S = 0.0, G = 0.0;
        bool carry = false;
        if ((rc = hll_replay_one(ctx, &prm, d_seq, rec, fin.data(), nullptr, S, G, carry))) break;
        if (h_images) memcpy(h_images + (size_t)g * ib + sum_at, &S, 8);
        if (d_images) HIPCHK(ctx, hipMemcpy(d_images + (size_t)g * ib + sum_at, &S, 8, hipMemcpyHostToDevice));
    }
    if (rc) return rc;
    ctx->hll_flags_n = 0;
    ctx->hll_flags_on_host = true;
    ctx->hll_left = left;
    return LASH_OK;
}

std::vector<uint32_t> hll_flagged(lash_ctx *ctx)
{
    std::vector<uint32_t> idx(lash_ctx_hll_inexact_sums(ctx, nullptr, 0));
    if (!idx.empty()) lash_ctx_hll_inexact_sums(ctx, idx.data(), (uint32_t)idx.size());
    return idx;
}

}  // namespace lashi

extern "C" {

uint32_t lash_ctx_hll_inexact_sums(lash_ctx *ctx, uint32_t *genome_index, uint32_t cap)
{
    if (ctx && ctx->hll_flags_on_host) {                          // a replay has run: what it could not redo
        for (uint32_t i = 0; i < ctx->hll_left.size() && i < cap && genome_index; ++i) genome_index[i] = ctx->hll_left[i];
        return (uint32_t)ctx->hll_left.size();
    }
    if (!ctx || !ctx->hll_flags_n || !ctx->hll_flags.ptr) return 0;
    (void)hipSetDevice(ctx->device);
    std::vector<uint32_t> fl(ctx->hll_flags_n);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(fl.data(), ctx->hll_flags.ptr, fl.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        ctx->err = "lash_ctx_hll_inexact_sums: reading the flags failed";
        return 0;
    }
    uint32_t n = 0;
    for (uint32_t g = 0; g < fl.size(); ++g)
        if (fl[g]) { if (genome_index && n < cap) genome_index[n] = g; ++n; }
    return n;
}

int lash_hll_replay_sums_device(lash_ctx *ctx, const lash_params *prm, const uint8_t *d_seq, const uint64_t *d_rec_off, uint64_t n_rec,
                                const uint64_t *genome_rec_off, uint32_t n_genomes, uint8_t *d_images)
{
    (void)n_rec;
    if (!ctx || !prm || !genome_rec_off || (n_genomes && !d_images)) return LASH_EINVAL;
    if (prm->algo != LASH_HLL) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->hll_flags_on_host) return LASH_OK;                     // already done for this call
    if (ctx->hll_flags_n != n_genomes) return LASH_EINVAL;          // not the arguments of the last HyperLogLog call
    const std::vector<uint32_t> flagged = hll_flagged(ctx);          // (synchronizes the stream)
    std::vector<uint32_t> left;
    if (flagged.empty()) { ctx->hll_flags_on_host = true; ctx->hll_left.clear(); return LASH_OK; }
    return hll_replay_sums(ctx, prm, d_seq, d_rec_off, nullptr, genome_rec_off, d_images, nullptr, flagged, left);
}

int lash_hll_replay_streamed_chunk(lash_ctx *ctx, const lash_params *prm, const uint8_t *raw, uint64_t n_bytes, int fmt, const uint8_t *image_before,
                                   uint8_t *image_after, double *carry, int *have_carry)
{
    if (!ctx || !prm || !image_before || !image_after || !carry || !have_carry || (n_bytes && !raw)) return LASH_EINVAL;
    int rc = lash_params_check(prm);
    if (rc) return rc;
    if (prm->algo != LASH_HLL || (prm->flags & LASH_F_AMINO) || (fmt != LASH_FMT_FASTA && fmt != LASH_FMT_FASTQ)) return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    const int p = prm->p, sum_at = hll_sum_field_offset(ctx->layout);
    if (sum_at < 0) return LASH_OK;                                  // (a layout without the field: nothing to keep)
    const size_t hdr = header_bytes(ctx->layout, LASH_HLL), m = (size_t)1 << p;
    bool event = false;                                              // did THIS chunk lift a register above 53 - p (or one that was there, further)?
    for (size_t j = 0; j < m && !event; ++j) event = image_after[hdr + j] > 53 - p && image_after[hdr + j] != image_before[hdr + j];
    double S = carry[0], G = carry[1];
    bool have = *have_carry != 0;
    if (!event) {
        if (!have) return LASH_OK;                                   // still on the grid: the header's sum is exact
        const double g = grid_sum(image_after + hdr, m, p);          // every step of this chunk was exact: its net change
        S += g - G;
        G = g;
    } else {
        // the chunk's records as needletail yields them (the library's own host parse), resident for the prefix sketches
        std::vector<uint8_t> seq;
        std::vector<uint64_t> rec(1, 0);
        if (fmt == LASH_FMT_FASTA && n_bytes && raw[0] != '>') {
            // a later chunk of a record that outgrew its chunk begins with sequence lines (the carried bases first): the device parse
            // takes them as a record's lines; the host parse wants the header line it would have had
            std::vector<uint8_t> with_hdr;
            with_hdr.reserve((size_t)n_bytes + 3);
            with_hdr.push_back('>'); with_hdr.push_back('c'); with_hdr.push_back('\n');
            with_hdr.insert(with_hdr.end(), raw, raw + n_bytes);
            parse_fastx_strict(with_hdr.data(), with_hdr.size(), &seq, &rec, ctx->layout.fastq_skip_bad != 0);
        } else {
            parse_fastx_strict(raw, (size_t)n_bytes, &seq, &rec, ctx->layout.fastq_skip_bad != 0);
        }
        if (rec.back() == rec.front()) { ctx->err = "hll replay: the chunk holds an event but no base"; return LASH_EINVAL; }
        if ((rc = reserve(ctx, ctx->st_seq, seq.size() + 64))) return rc;
        if (!seq.empty()) HIPCHK(ctx, hipMemcpy(ctx->st_seq.ptr, seq.data(), seq.size(), hipMemcpyHostToDevice));
        ReplayRestore restore(ctx);
        ctx->timing = false;
        lash_params pr = *prm;
        pr.flags &= ~(uint32_t)LASH_F_ACCUMULATE;                    // (a prefix is sketched by itself; the registers before it are `image_before`'s)
        if ((rc = hll_replay_one(ctx, &pr, static_cast<const uint8_t *>(ctx->st_seq.ptr), rec, image_after, image_before + hdr, S, G, have))) return rc;
    }
    memcpy(image_after + sum_at, &S, 8);
    carry[0] = S; carry[1] = G;
    *have_carry = have ? 1 : 0;
    ctx->hll_flags_n = 0;                                            // (the caller's image carries the incremental value now)
    ctx->hll_flags_on_host = true;
    ctx->hll_left.clear();
    return LASH_OK;
}

}  // extern "C"
