// lash_dist_api.hip — the extern "C" entries on finished images: union of serialized sketches (Sketch::merge / HyperLogLog::union /
// UltraLogLog::merge as `lash dist` uses them, utils.rs:84-373) and the pair statistics of the all-vs-all drivers.  Split out of lash_api.hip in round 6.
#include "lash_ctx.h"
#include "lash_internal.h"

extern "C" {

int lash_merge_images_device(lash_ctx *ctx, int algo, int p, uint8_t *d_dst, const uint8_t *d_src, uint64_t n_images)
{
    if (!ctx || (n_images && (!d_dst || !d_src)) || n_images > 0x7FFFFFFFull) return LASH_EINVAL;
    lash_params prm{algo, 16, p, 0, 0};
    int rc = lash_params_check(&prm);
    if (rc) return rc;
    if (n_images == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    // one pseudo work item per image: "partials" are the source images themselves (registers after the header)
    std::vector<WorkItem> items((size_t)n_images);
    std::vector<uint32_t> begin((size_t)n_images + 1);
    for (uint64_t i = 0; i < n_images; ++i) { items[i] = WorkItem{(uint32_t)i, 0, 0, 0}; begin[i] = (uint32_t)i; }
    begin[n_images] = (uint32_t)n_images;
    if ((rc = reserve(ctx, ctx->items, (size_t)(n_images + 1) * sizeof(WorkItem)))) return rc;
    if ((rc = reserve(ctx, ctx->item_begin, (size_t)(n_images + 1) * 4))) return rc;
    if ((rc = upload(ctx, ctx->items.ptr, items.data(), items.size() * sizeof(WorkItem)))) return rc;
    if ((rc = upload(ctx, ctx->item_begin.ptr, begin.data(), begin.size() * 4))) return rc;
    FinalizeArgs fa{};
    fa.partials = d_src;
    fa.items = static_cast<const WorkItem *>(ctx->items.ptr);
    fa.genome_item_begin = static_cast<const uint32_t *>(ctx->item_begin.ptr);
    fa.nvalid = nullptr;
    fa.images = d_dst;
    fa.image_bytes = image_bytes(ctx->layout, algo, p);
    fa.partial_stride = fa.image_bytes;
    fa.partial_base_off = header_bytes(ctx->layout, algo);
    fa.lay = layout_dev(ctx->layout, algo);
    fa.src_images = 1;
    const double alpha = hll_alpha(p);
    memcpy(&fa.alpha_bits, &alpha, 8);
    fa.algo = algo;
    fa.p = p;
    fa.k = 16;
    fa.accumulate = 1;
    HIPCHK(ctx, launch_finalize(fa, (uint32_t)n_images, ctx->stream));
    return LASH_OK;
}

int lash_merge_images(lash_ctx *ctx, int algo, int p, uint8_t *dst, const uint8_t *src, uint64_t n_images)
{
    if (!ctx || (n_images && (!dst || !src))) return LASH_EINVAL;
    const size_t ib = image_bytes(ctx->layout, algo, p);
    if (!ib) return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    const size_t bytes = ib * (size_t)n_images;
    int rc;
    if ((rc = reserve(ctx, ctx->st_img, bytes + 64))) return rc;
    if ((rc = reserve(ctx, ctx->st_seq, bytes + 64))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_img.ptr, dst, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->st_seq.ptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = lash_merge_images_device(ctx, algo, p, static_cast<uint8_t *>(ctx->st_img.ptr),
                                  static_cast<const uint8_t *>(ctx->st_seq.ptr), n_images);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(dst, ctx->st_img.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LASH_OK;
}

int lash_hmh_pair_counts_device(lash_ctx *ctx, const uint8_t *d_ref_images, uint32_t n_ref, const uint8_t *d_qry_images,
                                uint32_t n_qry, uint32_t *d_out_c, uint32_t *d_out_n)
{
    if (!ctx || ((n_ref && n_qry) && (!d_ref_images || !d_qry_images || !d_out_c || !d_out_n))) return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    if (n_ref == 0 || n_qry == 0) return LASH_OK;
    const uint32_t hdr = (uint32_t)header_bytes(ctx->layout, LASH_HMH);
    const uint64_t stride = image_bytes(ctx->layout, LASH_HMH, 0);
    static const bool words_kernel = getenv("LASH_HMH_PAIRS_WORDS") != nullptr;      // A/B knob: the u16-pair kernel on the images
    if (words_kernel) {
        HIPCHK(ctx, launch_hmh_pairs(d_ref_images, n_ref, d_qry_images, n_qry, hdr, stride, d_out_c, d_out_n, ctx->stream));
        return LASH_OK;
    }
    // register bit planes of the call's images (pair_planes.hip; one read-back of the non-zero counts: synchronizes once)
    const bool same = d_ref_images == d_qry_images && n_ref == n_qry;
    lash_sketch_set *sets[2] = {&ctx->pl_ref, same ? &ctx->pl_ref : &ctx->pl_qry};
    for (int i = 0; i < (same ? 1 : 2); ++i) {
        lash_sketch_set *s = sets[i];
        s->device = ctx->device; s->algo = LASH_HMH; s->p = 0; s->hdr = hdr; s->stride = stride;
        s->n = i ? n_qry : n_ref;
        s->d_images = i ? d_qry_images : d_ref_images;
        s->have_S = s->have_T = false;                               // (the buffers are kept, their contents are this call's)
    }
    int rc;
    if ((rc = lash_set_build_planes(ctx, sets[0], true))) return rc;                 // (row and column layout in one pass when the sets coincide)
    if ((rc = lash_set_build_planes(ctx, sets[1], false))) return rc;
    HIPCHK(ctx, launch_hmh_pairs_planes(static_cast<const uint32_t *>(sets[0]->T.ptr), sets[0]->ldT, 0, n_ref, static_cast<const uint32_t *>(sets[1]->S.ptr),
                                        sets[1]->n_pad, n_qry, sets[0]->full && sets[1]->full, false, d_out_c, d_out_n, n_qry, ctx->stream));
    return LASH_OK;
}

int lash_hmh_pair_counts(lash_ctx *ctx, const uint8_t *ref_images, uint32_t n_ref, const uint8_t *qry_images,
                         uint32_t n_qry, uint32_t *out_c, uint32_t *out_n)
{
    if (!ctx || ((n_ref && n_qry) && (!ref_images || !qry_images || !out_c || !out_n))) return LASH_EINVAL;
    if (n_ref == 0 || n_qry == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    const size_t ib = image_bytes(ctx->layout, LASH_HMH, 0), rb = ib * n_ref, qb = ib * n_qry, pb = (size_t)n_ref * n_qry * 4;
    int rc;
    if ((rc = reserve(ctx, ctx->st_seq, rb + qb + 64))) return rc;
    if ((rc = reserve(ctx, ctx->st_img, 2 * pb + 64))) return rc;
    uint8_t *d_r = static_cast<uint8_t *>(ctx->st_seq.ptr), *d_q = d_r + rb;
    uint32_t *d_c = static_cast<uint32_t *>(ctx->st_img.ptr), *d_n = d_c + (size_t)n_ref * n_qry;
    HIPCHK(ctx, hipMemcpyAsync(d_r, ref_images, rb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_q, qry_images, qb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = lash_hmh_pair_counts_device(ctx, d_r, n_ref, d_q, n_qry, d_c, d_n))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_c, d_c, pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out_n, d_n, pb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LASH_OK;
}

int lash_hmh_pair_expected_collisions(lash_ctx *ctx, const double *ref_card, uint32_t n_ref, const double *qry_card, uint32_t n_qry,
                                      double *out_ec)
{
    if (!ctx || ((n_ref && n_qry) && (!ref_card || !qry_card || !out_ec))) return LASH_EINVAL;
    if (n_ref == 0 || n_qry == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    // O(1) regimes on the host; what is left needs the cell sum: pairs whose LARGER sketch is at or below 2^(p+5), i.e. both are
    std::vector<uint32_t> rs, qs;
    std::vector<uint8_t> rsmall(n_ref), qsmall(n_qry);
    double dummy;
    for (uint32_t i = 0; i < n_ref; ++i) rsmall[i] = !hmh_ec_closed_form(ref_card[i], ref_card[i], &dummy);
    for (uint32_t j = 0; j < n_qry; ++j) qsmall[j] = !hmh_ec_closed_form(qry_card[j], qry_card[j], &dummy);
    for (uint32_t j = 0; j < n_qry; ++j) if (qsmall[j]) qs.push_back(j);
    if (!qs.empty()) for (uint32_t i = 0; i < n_ref; ++i) if (rsmall[i]) rs.push_back(i);
    for (uint32_t i = 0; i < n_ref; ++i) {
        double *row = out_ec + (size_t)i * n_qry;
        for (uint32_t j = 0; j < n_qry; ++j)
            if (!(rsmall[i] && qsmall[j])) (void)hmh_ec_closed_form(qry_card[j], ref_card[i], &row[j]);
    }
    if (rs.empty()) return LASH_OK;
    constexpr size_t VEC = 65536 * sizeof(double);
    constexpr size_t Q_CHUNK = (24ull << 30) / VEC, R_CHUNK = (4ull << 30) / VEC;      // <= 24 + 4 GiB of vectors at a time
    int rc;
    std::vector<double> cards, x;
    for (size_t q0 = 0; q0 < qs.size(); q0 += Q_CHUNK) {
        const uint32_t nq = (uint32_t)std::min(Q_CHUNK, qs.size() - q0);
        cards.resize(nq);
        for (uint32_t j = 0; j < nq; ++j) cards[j] = qry_card[qs[q0 + j]];
        if ((rc = reserve(ctx, ctx->ec_card, (size_t)(nq + R_CHUNK) * 8))) return rc;
        double *d_card = static_cast<double *>(ctx->ec_card.ptr);
        const bool cached = qs.size() <= Q_CHUNK && ctx->ec_qry.ptr && cards == ctx->ec_qry_cards;
        if (!cached) {
            ctx->ec_qry_cards.clear();
            if ((rc = reserve(ctx, ctx->ec_qry, (size_t)nq * VEC))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(d_card, cards.data(), (size_t)nq * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, launch_collision_vectors(d_card, nq, static_cast<double *>(ctx->ec_qry.ptr), ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));    // (`cards` is reused below)
            if (qs.size() <= Q_CHUNK) ctx->ec_qry_cards = cards;
        }
        for (size_t r0 = 0; r0 < rs.size(); r0 += R_CHUNK) {
            const uint32_t nr = (uint32_t)std::min(R_CHUNK, rs.size() - r0);
            std::vector<double> rcards(nr);
            for (uint32_t i = 0; i < nr; ++i) rcards[i] = ref_card[rs[r0 + i]];
            if ((rc = reserve(ctx, ctx->ec_ref, (size_t)nr * VEC))) return rc;
            if ((rc = reserve(ctx, ctx->ec_x, (size_t)nr * nq * 8))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(d_card + nq, rcards.data(), (size_t)nr * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, launch_collision_vectors(d_card + nq, nr, static_cast<double *>(ctx->ec_ref.ptr), ctx->stream));
            HIPCHK(ctx, launch_collision_gemm(static_cast<const double *>(ctx->ec_ref.ptr), nr, static_cast<const double *>(ctx->ec_qry.ptr), nq,
                                              static_cast<double *>(ctx->ec_x.ptr), ctx->stream));
            x.resize((size_t)nr * nq);
            HIPCHK(ctx, hipMemcpyAsync(x.data(), ctx->ec_x.ptr, x.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            for (uint32_t i = 0; i < nr; ++i) {
                const uint32_t ri = rs[r0 + i];
                for (uint32_t j = 0; j < nq; ++j) {
                    const uint32_t qj = qs[q0 + j];
                    out_ec[(size_t)ri * n_qry + qj] = hmh_ec_from_cell_sum(x[(size_t)i * nq + j]);
                }
            }
        }
    }
    return LASH_OK;
}

int lash_hll_pair_union_stats_device(lash_ctx *ctx, int p, const uint8_t *d_ref_images, uint32_t n_ref,
                                     const uint8_t *d_qry_images, uint32_t n_qry, uint32_t *d_out_zero, double *d_out_sum)
{
    if (!ctx || p < 4 || p > 16 || ((n_ref && n_qry) && (!d_ref_images || !d_qry_images || !d_out_zero || !d_out_sum)))
        return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    const uint32_t hdr = (uint32_t)header_bytes(ctx->layout, LASH_HLL);
    static const bool byte_kernel_only = getenv("LASH_HLL_PAIRS_BYTEWISE") != nullptr;
    if (p >= 10 && n_ref && n_qry && !byte_kernel_only) {
        // threshold-bitmap form (dist_kernels.hip): needs the range of register values first — one 8-byte read-back
        int rc;
        if ((rc = reserve(ctx, ctx->hll_lohi, 8))) return rc;
        uint32_t *d_lohi = static_cast<uint32_t *>(ctx->hll_lohi.ptr);
        HIPCHK(ctx, hipMemsetAsync(d_lohi, 0xFF, 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d_lohi + 1, 0, 4, ctx->stream));
        HIPCHK(ctx, launch_hll_minmax(d_ref_images, n_ref, p, hdr, d_lohi, ctx->stream));
        const bool same = d_ref_images == d_qry_images && n_ref == n_qry;
        if (!same) HIPCHK(ctx, launch_hll_minmax(d_qry_images, n_qry, p, hdr, d_lohi, ctx->stream));
        uint32_t lohi[2] = {0, 0};
        HIPCHK(ctx, hipMemcpyAsync(lohi, d_lohi, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t lo = lohi[0], hi = lohi[1];
        if (hi > lo && hi <= 64u) {                              // (all registers equal, or values no sketch can hold: the byte-wise kernel)
            const uint32_t band = hi - lo;
            const size_t per = (size_t)band * ((size_t)1 << p) / 8;
            if ((rc = reserve(ctx, ctx->hll_bm_qry, (size_t)n_qry * per))) return rc;
            uint32_t *bq = static_cast<uint32_t *>(ctx->hll_bm_qry.ptr), *br = bq;
            HIPCHK(ctx, launch_hll_bitmaps(d_qry_images, n_qry, p, hdr, lo, band, bq, ctx->stream));
            if (!same) {
                if ((rc = reserve(ctx, ctx->hll_bm_ref, (size_t)n_ref * per))) return rc;
                br = static_cast<uint32_t *>(ctx->hll_bm_ref.ptr);
                HIPCHK(ctx, launch_hll_bitmaps(d_ref_images, n_ref, p, hdr, lo, band, br, ctx->stream));
            }
            HIPCHK(ctx, launch_hll_pairs_bitmap(br, n_ref, bq, n_qry, p, lo, band, d_out_zero, d_out_sum, ctx->stream));
            return LASH_OK;
        }
    }
    HIPCHK(ctx, launch_hll_pairs(d_ref_images, n_ref, d_qry_images, n_qry, p, hdr, d_out_zero, d_out_sum, ctx->stream));
    return LASH_OK;
}

int lash_hll_pair_union_stats(lash_ctx *ctx, int p, const uint8_t *ref_images, uint32_t n_ref, const uint8_t *qry_images,
                              uint32_t n_qry, uint32_t *out_zero, double *out_sum)
{
    if (!ctx || p < 4 || p > 16 || ((n_ref && n_qry) && (!ref_images || !qry_images || !out_zero || !out_sum))) return LASH_EINVAL;
    if (n_ref == 0 || n_qry == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    const size_t ib = image_bytes(ctx->layout, LASH_HLL, p), rb = ib * n_ref, qb = ib * n_qry, np = (size_t)n_ref * n_qry;
    int rc;
    if ((rc = reserve(ctx, ctx->st_seq, rb + qb + 64))) return rc;
    if ((rc = reserve(ctx, ctx->st_img, np * 12 + 64))) return rc;
    uint8_t *d_r = static_cast<uint8_t *>(ctx->st_seq.ptr), *d_q = d_r + rb;
    double *d_s = static_cast<double *>(ctx->st_img.ptr);
    uint32_t *d_z = reinterpret_cast<uint32_t *>(d_s + np);
    HIPCHK(ctx, hipMemcpyAsync(d_r, ref_images, rb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_q, qry_images, qb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = lash_hll_pair_union_stats_device(ctx, p, d_r, n_ref, d_q, n_qry, d_z, d_s))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_zero, d_z, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out_sum, d_s, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LASH_OK;
}

int lash_ull_pair_union_estimates_device(lash_ctx *ctx, int p, int estimator, const uint8_t *d_ref_images, uint32_t n_ref,
                                         const uint8_t *d_qry_images, uint32_t n_qry, double *d_out_est)
{
    if (!ctx || p < 3 || p > 26 || (estimator != LASH_ULL_FGRA && estimator != LASH_ULL_ML) ||
        ((n_ref && n_qry) && (!d_ref_images || !d_qry_images || !d_out_est)))
        return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, launch_ull_pairs(d_ref_images, n_ref, d_qry_images, n_qry, p, (uint32_t)header_bytes(ctx->layout, LASH_ULL), estimator,
                                 d_out_est, ctx->stream));
    return LASH_OK;
}

int lash_ull_pair_union_estimates(lash_ctx *ctx, int p, int estimator, const uint8_t *ref_images, uint32_t n_ref,
                                  const uint8_t *qry_images, uint32_t n_qry, double *out_est)
{
    if (!ctx || p < 3 || p > 26 || ((n_ref && n_qry) && (!ref_images || !qry_images || !out_est))) return LASH_EINVAL;
    if (n_ref == 0 || n_qry == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    const size_t ib = image_bytes(ctx->layout, LASH_ULL, p), rb = ib * n_ref, qb = ib * n_qry, np = (size_t)n_ref * n_qry;
    int rc;
    if ((rc = reserve(ctx, ctx->st_seq, rb + qb + 64))) return rc;
    if ((rc = reserve(ctx, ctx->st_img, np * 8 + 64))) return rc;
    uint8_t *d_r = static_cast<uint8_t *>(ctx->st_seq.ptr), *d_q = d_r + rb;
    double *d_e = static_cast<double *>(ctx->st_img.ptr);
    HIPCHK(ctx, hipMemcpyAsync(d_r, ref_images, rb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_q, qry_images, qb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = lash_ull_pair_union_estimates_device(ctx, p, estimator, d_r, n_ref, d_q, n_qry, d_e))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out_est, d_e, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LASH_OK;
}

double lash_ull_estimate(const uint8_t *registers, int p, int estimator)
{
    if (!registers || p < 3 || p > 26) return -1.0;
    uint32_t hist[256] = {0};
    for (size_t i = 0, m = (size_t)1 << p; i < m; ++i) hist[registers[i]]++;
    auto h = [&](uint32_t r) { return hist[r]; };
    return estimator == LASH_ULL_ML ? lash::ull::ml(h, p) : lash::ull::fgra(h, p);
}

}  // extern "C"
