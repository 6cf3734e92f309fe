// sketch_set.hip — the resident form of the dist side (include/lash_gfx950.h: lash_sketch_set_*): N serialized sketches kept in
// HBM for the whole of an all-vs-all run, with what the pair kernels derive from them built ONCE.
//
// Reference: `lash dist` loads both sketch files into two hash maps and keeps them for the run (/root/reference/src/utils.rs:
// 95-127, 202-242, 303-337), computes one cardinality per sketch (utils.rs:170-173, 213-219, 314-315) and walks reference rows
// x query columns, lower triangle only when both are the same files (utils.rs:150-180).  At BASELINE configs[3] that is 10^5
// sketches (3.3 GB of HyperMinHash images) and 5 * 10^9 printed pairs: the set is uploaded (or adopted from an all-gather) once,
// cardinalities come from register histograms made on the GPU, and lash_sketch_set_pair_block hands out the statistics of one
// block of rows against a prefix of the columns with the tiles above the diagonal skipped.
#include "lash_ctx.h"

namespace lash {

// hist[s][256]: HLL / ULL count register bytes; HyperMinHash counts the 6-bit leading-zero field of its u16 registers (bins 0..63)
__global__ void __launch_bounds__(256) sketch_hist_kernel(const uint8_t *__restrict__ img, int algo, uint32_t n_regs, uint32_t hdr,
                                                          uint64_t stride, uint32_t hmh_be, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *src = img + (uint64_t)blockIdx.x * stride + hdr;
    if (algo == LASH_HMH) {
        for (uint32_t i = threadIdx.x; i < n_regs; i += 256u) atomicAdd(&h[src[2u * i + (hmh_be ? 0u : 1u)] >> 2], 1u);   // reg >> 10
    } else {
        for (uint32_t i = threadIdx.x; i < n_regs; i += 256u) atomicAdd(&h[src[i]], 1u);
    }
    __syncthreads();
    hist[(uint64_t)blockIdx.x * 256u + threadIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const uint8_t *__restrict__ src, const uint32_t *__restrict__ order,
                                                          uint64_t row_bytes, uint8_t *__restrict__ dst)
{
    const uint8_t *s = src + (uint64_t)order[blockIdx.x] * row_bytes;
    uint8_t *d = dst + (uint64_t)blockIdx.x * row_bytes;
    if ((((uintptr_t)s | (uintptr_t)d | row_bytes) & 15u) == 0) {
        for (uint64_t i = threadIdx.x; i < row_bytes / 16; i += 256u) reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(s)[i];
    } else {
        for (uint64_t i = threadIdx.x; i < row_bytes; i += 256u) d[i] = s[i];
    }
}

hipError_t launch_sketch_hist(const uint8_t *d_img, uint32_t n, int algo, uint32_t n_regs, uint32_t hdr, uint64_t stride, uint32_t hmh_be,
                              uint32_t *d_hist, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sketch_hist_kernel, dim3(n), dim3(256), 0, stream, d_img, algo, n_regs, hdr, stride, hmh_be, d_hist);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const uint8_t *d_src, const uint32_t *d_order, uint32_t n, uint64_t row_bytes, uint8_t *d_dst, hipStream_t stream)
{
    if (n == 0 || row_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, stream, d_src, d_order, row_bytes, d_dst);
    return hipGetLastError();
}

}  // namespace lash

namespace {

int set_geometry(lash_ctx *ctx, lash_sketch_set *s, int algo, int p, uint32_t n)
{
    lash_params prm{algo, 16, p, 0, 0};
    const int rc = lash_params_check(&prm);
    if (rc) return rc;
    s->device = ctx->device;
    s->algo = algo;
    s->p = algo == LASH_HMH ? 0 : p;
    s->n = n;
    s->hdr = (uint32_t)header_bytes(ctx->layout, algo);
    s->stride = image_bytes(ctx->layout, algo, p);
    s->hmh_be = ctx->layout.hmh_reg_be;
    return LASH_OK;
}

uint32_t set_regs(const lash_sketch_set *s) { return s->algo == LASH_HMH ? HMH_M : (1u << s->p); }

}  // namespace

// HyperMinHash: column layout (+ non-zero counts -> `full`), optionally the row layout.  Synchronizes the stream.
int lash_set_build_planes(lash_ctx *ctx, lash_sketch_set *s, bool want_T)
{
    int rc;
    const bool need_S = !s->have_S, need_T = want_T && !s->have_T;
    if (!need_S && !need_T) return LASH_OK;
    if (need_S) {
        const uint32_t pad = hmh_planes_col_pad();
        s->n_pad = (s->n + pad - 1) / pad * pad;
        if ((rc = reserve(ctx, s->S, hmh_planes_S_words(s->n_pad) * 4))) return rc;
        if ((rc = reserve(ctx, s->nzcount, (size_t)s->n * 4 + 4))) return rc;
        HIPCHK(ctx, hipMemsetAsync(s->nzcount.ptr, 0, (size_t)s->n * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(static_cast<uint32_t *>(s->S.ptr) + hmh_planes_S_words(s->n_pad) - 1024, 0, 4096, ctx->stream));   // the slack
    }
    if (need_T) {
        const uint32_t pad = hmh_planes_row_pad();
        s->ldT = (s->n + pad - 1) / pad * pad;
        if (s->ldT >= (1u << 24)) return LASH_ELIMIT;
        if ((rc = reserve(ctx, s->T, hmh_planes_T_words(s->ldT) * 4))) return rc;
    }
    HIPCHK(ctx, launch_hmh_planes(s->d_images, s->hdr, s->stride, s->n, need_T ? static_cast<uint32_t *>(s->T.ptr) : nullptr, s->ldT,
                                  need_S ? static_cast<uint32_t *>(s->S.ptr) : nullptr, s->n_pad, need_S ? static_cast<uint32_t *>(s->nzcount.ptr) : nullptr,
                                  ctx->stream));
    if (need_S) {
        std::vector<uint32_t> nz(s->n);
        HIPCHK(ctx, hipMemcpyAsync(nz.data(), s->nzcount.ptr, (size_t)s->n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        s->full = true;
        for (uint32_t v : nz) s->full = s->full && v == HMH_M;
        s->have_S = true;
    } else {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (need_T) s->have_T = true;
    return LASH_OK;
}

namespace {

int hll_range(lash_ctx *ctx, lash_sketch_set *s)
{
    if (s->have_range) return LASH_OK;
    int rc;
    if ((rc = reserve(ctx, s->lohi, 8))) return rc;
    uint32_t *d = static_cast<uint32_t *>(s->lohi.ptr);
    HIPCHK(ctx, hipMemsetAsync(d, 0xFF, 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d + 1, 0, 4, ctx->stream));
    HIPCHK(ctx, launch_hll_minmax(s->d_images, s->n, s->p, s->hdr, d, ctx->stream));
    uint32_t lohi[2] = {0, 0};
    HIPCHK(ctx, hipMemcpyAsync(lohi, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    s->lo = lohi[0];
    s->hi = lohi[1];
    s->have_range = true;
    return LASH_OK;
}

int hll_bitmaps(lash_ctx *ctx, lash_sketch_set *s, uint32_t lo, uint32_t band)
{
    if (s->have_bm && s->bm_lo == lo && s->bm_band == band) return LASH_OK;
    int rc;
    const size_t per = (size_t)band * ((size_t)1 << s->p) / 8;
    if ((rc = reserve(ctx, s->bm, (size_t)s->n * per))) return rc;
    HIPCHK(ctx, launch_hll_bitmaps(s->d_images, s->n, s->p, s->hdr, lo, band, static_cast<uint32_t *>(s->bm.ptr), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    s->bm_lo = lo;
    s->bm_band = band;
    s->have_bm = true;
    return LASH_OK;
}

}  // namespace

extern "C" {

int lash_sketch_set_create_device(lash_ctx *ctx, int algo, int p, const uint8_t *d_images, uint32_t n, lash_sketch_set **out)
{
    if (!ctx || !out || (n && !d_images)) return LASH_EINVAL;
    *out = nullptr;
    lash_sketch_set *s = new (std::nothrow) lash_sketch_set();
    if (!s) return LASH_ENOMEM;
    const int rc = set_geometry(ctx, s, algo, p, n);
    if (rc) { delete s; return rc; }
    s->d_images = d_images;
    *out = s;
    return LASH_OK;
}

int lash_sketch_set_create(lash_ctx *ctx, int algo, int p, const uint8_t *images, uint32_t n_images, const uint32_t *order, uint32_t n,
                           lash_sketch_set **out)
{
    if (!ctx || !out || (n && !images) || (!order && n != n_images)) return LASH_EINVAL;
    *out = nullptr;
    if (order)
        for (uint32_t i = 0; i < n; ++i)
            if (order[i] >= n_images) return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    lash_sketch_set *s = new (std::nothrow) lash_sketch_set();
    if (!s) return LASH_ENOMEM;
    int rc = set_geometry(ctx, s, algo, p, n);
    auto bail = [&](int code) { lash_sketch_set_free(ctx, s); return code; };
    if (rc) return bail(rc);
    if ((rc = reserve(ctx, s->images, (size_t)n * s->stride + 64))) return bail(rc);
    s->d_images = static_cast<const uint8_t *>(s->images.ptr);
    if (n == 0) { *out = s; return LASH_OK; }
    hipError_t e;
    if (!order) {
        e = hipMemcpyAsync(s->images.ptr, images, (size_t)n * s->stride, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return bail(fail(ctx, LASH_EHIP, "lash_sketch_set_create: upload", e));
    } else {
        // the whole file goes up once, the members are gathered into set order on the device
        DevBuf all, ord;
        rc = reserve(ctx, all, (size_t)n_images * s->stride + 64);
        if (!rc) rc = reserve(ctx, ord, (size_t)n * 4);
        if (!rc) {
            e = hipMemcpyAsync(all.ptr, images, (size_t)n_images * s->stride, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ord.ptr, order, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess)
                e = launch_gather_rows(static_cast<const uint8_t *>(all.ptr), static_cast<const uint32_t *>(ord.ptr), n, s->stride,
                                       static_cast<uint8_t *>(s->images.ptr), ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, LASH_EHIP, "lash_sketch_set_create: upload + gather", e);
        }
        release(all);
        release(ord);
        if (rc) return bail(rc);
    }
    *out = s;
    return LASH_OK;
}

void lash_sketch_set_free(lash_ctx *ctx, lash_sketch_set *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf *b : {&s->images, &s->S, &s->T, &s->nzcount, &s->lohi, &s->bm, &s->ec_vec}) release(*b);
    delete s;
}

uint32_t lash_sketch_set_size(const lash_sketch_set *s) { return s ? s->n : 0; }

int lash_sketch_set_cardinalities(lash_ctx *ctx, lash_sketch_set *s, int ull_estimator, const lash_hll_bias *tables, double *out_card,
                                  uint32_t *bad_index)
{
    if (!ctx || !s || (s->n && !out_card)) return LASH_EINVAL;
    if (s->algo == LASH_ULL && ull_estimator != LASH_ULL_FGRA && ull_estimator != LASH_ULL_ML) return LASH_EINVAL;
    if (s->n == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    int rc;
    const size_t hb = (size_t)s->n * 256 * 4;
    if ((rc = reserve(ctx, ctx->st_seq, hb))) return rc;
    HIPCHK(ctx, launch_sketch_hist(s->d_images, s->n, s->algo, set_regs(s), s->hdr, s->stride, s->hmh_be, static_cast<uint32_t *>(ctx->st_seq.ptr),
                                   ctx->stream));
    std::vector<uint32_t> hist((size_t)s->n * 256);
    HIPCHK(ctx, hipMemcpyAsync(hist.data(), ctx->st_seq.ptr, hb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint8_t> one;
    for (uint32_t i = 0; i < s->n; ++i) {
        const uint32_t *h = hist.data() + (size_t)i * 256;
        if (s->algo == LASH_HMH) {
            bool exact = true;
            out_card[i] = hmh_cardinality_from_hist(h, &exact);
            if (!exact) {                                     // a register with lz > 39: the crate's register-order sum, on the host
                one.resize(s->stride);
                HIPCHK(ctx, hipMemcpy(one.data(), s->d_images + (uint64_t)i * s->stride, s->stride, hipMemcpyDeviceToHost));
                out_card[i] = lash_hmh_cardinality(one.data() + s->hdr, (int)s->hmh_be);
            }
        } else if (s->algo == LASH_HLL) {
            if (hll_cardinality_from_hist(h, s->p, tables, &out_card[i]) != LASH_OK) { if (bad_index) *bad_index = i; return LASH_ERANGE; }
        } else {
            auto hh = [&](uint32_t r) { return h[r]; };
            out_card[i] = ull_estimator == LASH_ULL_ML ? lash::ull::ml(hh, s->p) : lash::ull::fgra(hh, s->p);
        }
    }
    s->card.assign(out_card, out_card + s->n);
    s->small_idx.clear();
    s->have_ec_vec = false;
    if (s->algo == LASH_HMH) {
        double dummy;
        for (uint32_t i = 0; i < s->n; ++i)
            if (!hmh_ec_closed_form(out_card[i], out_card[i], &dummy)) s->small_idx.push_back(i);
    }
    return LASH_OK;
}

int lash_sketch_set_prepare(lash_ctx *ctx, lash_sketch_set *ref, lash_sketch_set *qry)
{
    if (!ctx || !ref || !qry || ref->algo != qry->algo || ref->p != qry->p || ref->device != ctx->device || qry->device != ctx->device)
        return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    int rc;
    if (ref->n == 0 || qry->n == 0) return LASH_OK;
    if (ref->algo == LASH_HMH) {
        // expected collisions of small pairs (lash_sketch_set_hmh_expected_collisions): the query side's cell vectors, once, when
        // both sides have small members and the vectors fit a third of the free memory (else they are made per block)
        if (!ref->small_idx.empty() && !qry->small_idx.empty() && !qry->have_ec_vec) {
            size_t free_b = 0, total_b = 0;
            const size_t need = qry->small_idx.size() * (size_t)65536 * 8;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 3) {
                std::vector<double> cards(qry->small_idx.size());
                for (size_t j = 0; j < cards.size(); ++j) cards[j] = qry->card[qry->small_idx[j]];
                if ((rc = reserve(ctx, qry->ec_vec, need))) return rc;
                if ((rc = reserve(ctx, ctx->ec_card, cards.size() * 8))) return rc;
                HIPCHK(ctx, hipMemcpyAsync(ctx->ec_card.ptr, cards.data(), cards.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                HIPCHK(ctx, launch_collision_vectors(static_cast<const double *>(ctx->ec_card.ptr), (uint32_t)cards.size(),
                                                     static_cast<double *>(qry->ec_vec.ptr), ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                qry->have_ec_vec = true;
            }
        }
        static const bool words_kernel = getenv("LASH_HMH_PAIRS_WORDS") != nullptr;      // A/B knob: the u16-pair kernel on the images
        if (words_kernel) return LASH_OK;
        if ((rc = lash_set_build_planes(ctx, ref, true))) return rc;                     // (both layouts in one pass when ref == qry)
        if ((rc = lash_set_build_planes(ctx, qry, false))) return rc;
    } else if (ref->algo == LASH_HLL && ref->p >= 10) {
        static const bool byte_kernel_only = getenv("LASH_HLL_PAIRS_BYTEWISE") != nullptr;
        if (byte_kernel_only) return LASH_OK;
        if ((rc = hll_range(ctx, ref))) return rc;
        if (qry != ref && (rc = hll_range(ctx, qry))) return rc;
        const uint32_t lo = std::min(ref->lo, qry->lo), hi = std::max(ref->hi, qry->hi);
        if (hi > lo && hi <= 64u) {                          // (all registers equal, or values no sketch can hold: the byte-wise kernel)
            if ((rc = hll_bitmaps(ctx, qry, lo, hi - lo))) return rc;
            if (qry != ref && (rc = hll_bitmaps(ctx, ref, lo, hi - lo))) return rc;
        } else {
            ref->have_bm = qry->have_bm = false;
        }
    }
    return LASH_OK;
}

int lash_sketch_set_pair_block_device(lash_ctx *ctx, const lash_sketch_set *ref, uint32_t r0, uint32_t r1, const lash_sketch_set *qry,
                                      uint32_t n_cols, int triangle, int ull_estimator, uint32_t *d_c_or_zero, uint32_t *d_n,
                                      double *d_sum_or_union)
{
    if (!ctx || !ref || !qry || ref->algo != qry->algo || ref->p != qry->p || r0 > r1 || r1 > ref->n || n_cols > qry->n) return LASH_EINVAL;
    if (ref->device != ctx->device || qry->device != ctx->device) return LASH_EINVAL;
    const uint32_t nr = r1 - r0;
    if (nr == 0 || n_cols == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    const int64_t tri = triangle ? (int64_t)r0 : -1;
    const uint8_t *rimg = ref->d_images + (uint64_t)r0 * ref->stride;
    switch (ref->algo) {
    case LASH_HMH:
        if (!d_c_or_zero || !d_n) return LASH_EINVAL;
        if (ref->have_T && qry->have_S) {
            HIPCHK(ctx, launch_hmh_pairs_planes(static_cast<const uint32_t *>(ref->T.ptr), ref->ldT, r0, nr, static_cast<const uint32_t *>(qry->S.ptr),
                                                qry->n_pad, n_cols, ref->full && qry->full, triangle != 0, d_c_or_zero, d_n, n_cols, ctx->stream));
        } else {
            HIPCHK(ctx, launch_hmh_pairs(rimg, nr, qry->d_images, n_cols, ref->hdr, ref->stride, d_c_or_zero, d_n, ctx->stream, tri));
        }
        return LASH_OK;
    case LASH_HLL:
        if (!d_c_or_zero || !d_sum_or_union) return LASH_EINVAL;
        if (ref->have_bm && qry->have_bm && ref->bm_lo == qry->bm_lo && ref->bm_band == qry->bm_band) {
            const size_t per = (size_t)ref->bm_band * (((size_t)1 << ref->p) / 32);
            HIPCHK(ctx, launch_hll_pairs_bitmap(static_cast<const uint32_t *>(ref->bm.ptr) + (size_t)r0 * per, nr, static_cast<const uint32_t *>(qry->bm.ptr),
                                                n_cols, ref->p, ref->bm_lo, ref->bm_band, d_c_or_zero, d_sum_or_union, ctx->stream, tri));
        } else {
            HIPCHK(ctx, launch_hll_pairs(rimg, nr, qry->d_images, n_cols, ref->p, ref->hdr, d_c_or_zero, d_sum_or_union, ctx->stream, tri));
        }
        return LASH_OK;
    case LASH_ULL:
        if (!d_sum_or_union || (ull_estimator != LASH_ULL_FGRA && ull_estimator != LASH_ULL_ML)) return LASH_EINVAL;
        HIPCHK(ctx, launch_ull_pairs(rimg, nr, qry->d_images, n_cols, ref->p, ref->hdr, ull_estimator, d_sum_or_union, ctx->stream, tri));
        return LASH_OK;
    default: return LASH_EINVAL;
    }
}

int lash_sketch_set_pair_block(lash_ctx *ctx, const lash_sketch_set *ref, uint32_t r0, uint32_t r1, const lash_sketch_set *qry, uint32_t n_cols,
                               int triangle, int ull_estimator, uint32_t *out_c_or_zero, uint32_t *out_n, double *out_sum_or_union)
{
    if (!ctx || !ref || r0 > r1) return LASH_EINVAL;
    const size_t np = (size_t)(r1 - r0) * n_cols;
    if (np == 0) return LASH_OK;
    (void)hipSetDevice(ctx->device);
    int rc;
    // [sum_or_union f64 | c_or_zero u32 | n u32]
    if ((rc = reserve(ctx, ctx->st_img, np * 16 + 64))) return rc;
    double *d_u = static_cast<double *>(ctx->st_img.ptr);
    uint32_t *d_c = reinterpret_cast<uint32_t *>(d_u + np), *d_n = d_c + np;
    if ((rc = lash_sketch_set_pair_block_device(ctx, ref, r0, r1, qry, n_cols, triangle, ull_estimator, d_c, d_n, d_u))) return rc;
    if (ref->algo != LASH_ULL) {
        if (!out_c_or_zero) return LASH_EINVAL;
        HIPCHK(ctx, hipMemcpyAsync(out_c_or_zero, d_c, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (ref->algo == LASH_HMH) {
        if (!out_n) return LASH_EINVAL;
        HIPCHK(ctx, hipMemcpyAsync(out_n, d_n, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        if (!out_sum_or_union) return LASH_EINVAL;
        HIPCHK(ctx, hipMemcpyAsync(out_sum_or_union, d_u, np * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return LASH_OK;
}

int lash_sketch_set_hmh_expected_collisions(lash_ctx *ctx, const lash_sketch_set *ref, uint32_t r0, uint32_t r1, const lash_sketch_set *qry,
                                            uint32_t n_cols, double *out_ec, uint64_t *n_small_pairs)
{
    if (n_small_pairs) *n_small_pairs = 0;
    if (!ctx || !ref || !qry || ref->algo != LASH_HMH || qry->algo != LASH_HMH || r0 > r1 || r1 > ref->n || n_cols > qry->n) return LASH_EINVAL;
    if (ref->card.size() != ref->n || qry->card.size() != qry->n) return LASH_EINVAL;          // lash_sketch_set_cardinalities first
    // the block's small rows, the small columns below n_cols (small_idx ascends)
    const auto rb = std::lower_bound(ref->small_idx.begin(), ref->small_idx.end(), r0), re = std::lower_bound(rb, ref->small_idx.end(), r1);
    const uint32_t nrs = (uint32_t)(re - rb);
    const uint32_t nqs = (uint32_t)(std::lower_bound(qry->small_idx.begin(), qry->small_idx.end(), n_cols) - qry->small_idx.begin());
    if (nrs == 0 || nqs == 0) return LASH_OK;
    if (!out_ec) return LASH_EINVAL;
    (void)hipSetDevice(ctx->device);
    constexpr size_t VEC = 65536 * sizeof(double);
    int rc;
    std::vector<double> cards(nrs), x;
    for (uint32_t i = 0; i < nrs; ++i) cards[i] = ref->card[rb[i]];
    if ((rc = reserve(ctx, ctx->ec_ref, (size_t)nrs * VEC))) return rc;
    if ((rc = reserve(ctx, ctx->ec_card, (size_t)(nrs + 4096) * 8))) return rc;
    double *d_card = static_cast<double *>(ctx->ec_card.ptr);
    HIPCHK(ctx, hipMemcpyAsync(d_card, cards.data(), (size_t)nrs * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, launch_collision_vectors(d_card, nrs, static_cast<double *>(ctx->ec_ref.ptr), ctx->stream));
    // the query vectors: the set's (prepare), or made here 4 096 at a time when they did not fit
    const uint32_t q_step = qry->have_ec_vec ? nqs : 4096u;
    std::vector<double> qc;
    for (uint32_t q0 = 0; q0 < nqs; q0 += q_step) {
        const uint32_t nq = std::min(q_step, nqs - q0);
        const double *d_B = static_cast<const double *>(qry->ec_vec.ptr);
        if (!qry->have_ec_vec) {
            qc.resize(nq);
            for (uint32_t j = 0; j < nq; ++j) qc[j] = qry->card[qry->small_idx[q0 + j]];
            if ((rc = reserve(ctx, ctx->ec_qry, (size_t)nq * VEC))) return rc;
            HIPCHK(ctx, hipMemcpyAsync(d_card + nrs, qc.data(), (size_t)nq * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(ctx, launch_collision_vectors(d_card + nrs, nq, static_cast<double *>(ctx->ec_qry.ptr), ctx->stream));
            d_B = static_cast<const double *>(ctx->ec_qry.ptr);
        }
        if ((rc = reserve(ctx, ctx->ec_x, (size_t)nrs * nq * 8))) return rc;
        HIPCHK(ctx, launch_collision_gemm(static_cast<const double *>(ctx->ec_ref.ptr), nrs, d_B, nq, static_cast<double *>(ctx->ec_x.ptr), ctx->stream));
        x.resize((size_t)nrs * nq);
        HIPCHK(ctx, hipMemcpyAsync(x.data(), ctx->ec_x.ptr, x.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t i = 0; i < nrs; ++i) {
            double *row = out_ec + (size_t)(rb[i] - r0) * n_cols;
            for (uint32_t j = 0; j < nq; ++j) row[qry->small_idx[q0 + j]] = hmh_ec_from_cell_sum(x[(size_t)i * nq + j]);
        }
    }
    if (n_small_pairs) *n_small_pairs = (uint64_t)nrs * nqs;
    return LASH_OK;
}

}  // extern "C"
