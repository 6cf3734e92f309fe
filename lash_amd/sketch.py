"""Python face of the C ABI (include/lash_gfx950.h).  Mirrors the reference's sketch interface for the hot path:
`sketch_files::<S>(precision, files, k, out, threads, seed, aa)` (utils.rs:439-447) becomes
`Context.sketch_batch(algo, k, p, seed, records...)` returning the bytes `S::save` would write (utils.rs:571-573).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Layout, Params, Timing

ALGOS = {"hmh": _lib.HMH, "hll": _lib.HLL, "ull": _lib.ULL}
ULL_ESTIMATORS = {"fgra": 0, "ml": 1}


class HllBias:
    """HLL++ empirical bias tables (include/lash_gfx950.h: lash_hll_bias).  They are not part of this repository: load the
    text file tools/ref_probe/extract_hll_bias.py writes, or build from arrays (tests).  Without them the estimate <= 5 * 2^p
    regime of streaming_algorithms' len() is refused with ERANGE."""

    def __init__(self, path=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        if path is not None:
            rc = self._lib.lash_hll_bias_load(str(path).encode(), C.byref(self._h))
            if rc != _lib.OK:
                raise LashError(rc, "%s: %s" % (path, self._lib.lash_strerror(rc).decode()))

    def set(self, p, raw, bias):
        r = np.ascontiguousarray(raw, dtype=np.float64)
        b = np.ascontiguousarray(bias, dtype=np.float64)
        assert r.shape == b.shape and r.ndim == 1
        rc = self._lib.lash_hll_bias_from_arrays(C.byref(self._h), int(p), r.ctypes.data, b.ctypes.data, len(r))
        if rc != _lib.OK:
            raise LashError(rc, self._lib.lash_strerror(rc).decode())
        return self

    def has(self, p):
        return bool(self._lib.lash_hll_bias_has(self._h, int(p)))

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lash_hll_bias_free(self._h)
            self._h = C.c_void_p()


def _bias_handle(hll_bias):
    return None if hll_bias is None else hll_bias._h


def sketch_cardinality(algo, p, image, layout=None, estimator="fgra", hll_bias=None):
    """Distinct-count estimate of ONE serialized sketch, as `lash dist` computes it per sketch (utils.rs:101-103, 213-217, 314-315):
    hmh LogLog-beta, hll `len()` (LashError(ERANGE) in the bias-table regime unless `hll_bias` covers p), ull FGRA / ML.  Host only."""
    lib = _lib.load()
    a = _algo(algo)
    lay = parse_layout(layout)
    img = np.ascontiguousarray(image, dtype=np.uint8).reshape(-1)
    regs = img[header_bytes(a, lay):]
    if a == _lib.HMH:
        return float(lib.lash_hmh_cardinality(regs.ctypes.data, int(lay.hmh_reg_be)))
    if a == _lib.ULL:
        return float(lib.lash_ull_estimate(regs.ctypes.data, int(p), ULL_ESTIMATORS[estimator]))
    out = C.c_double()
    rc = lib.lash_hll_cardinality(regs.ctypes.data, int(p), _bias_handle(hll_bias), C.byref(out))
    if rc != _lib.OK:
        raise LashError(rc, lib.lash_strerror(rc).decode())
    return out.value


def dist_rows(algo, p, k, model, ref_card, qry_card, c_or_zero=None, n_counts=None, sum_or_union=None, fp32=False, hll_bias=None,
              hmh_ec=None):
    """The distances the reference prints for an [n_ref, n_qry] block, from the GPU's pair statistics and the per-sketch
    cardinalities (lash_dist_rows; utils.rs:164-167, 272-278, 355-365 + main.rs:415-423).  numpy in, float64 [n_ref, n_qry] out."""
    lib = _lib.load()
    rc_ = np.ascontiguousarray(ref_card, dtype=np.float64)
    qc_ = np.ascontiguousarray(qry_card, dtype=np.float64)
    nr, nq = len(rc_), len(qc_)
    a = None if c_or_zero is None else np.ascontiguousarray(c_or_zero, dtype=np.uint32)
    b = None if n_counts is None else np.ascontiguousarray(n_counts, dtype=np.uint32)
    d = None if sum_or_union is None else np.ascontiguousarray(sum_or_union, dtype=np.float64)
    out = np.zeros((nr, nq), dtype=np.float64)
    bad = C.c_uint64()
    ec = None if hmh_ec is None else np.ascontiguousarray(hmh_ec, dtype=np.float64)       # (kept alive across the call)
    assert ec is None or ec.size == nr * nq
    rc = lib.lash_dist_rows(_algo(algo), int(p or 0), int(k), int(model), 1 if fp32 else 0, nr, nq, rc_.ctypes.data, qc_.ctypes.data,
                            None if a is None else a.ctypes.data, None if b is None else b.ctypes.data,
                            None if d is None else d.ctypes.data, _bias_handle(hll_bias),
                            None if ec is None else ec.ctypes.data, out.ctypes.data, C.byref(bad))
    if rc != _lib.OK:
        raise LashError(rc, lib.lash_strerror(rc).decode() + (" (pair %d)" % bad.value if rc == _lib.ERANGE else ""))
    return out


class PinnedArray:
    """numpy view of page-locked host memory from lash_host_alloc_pinned (freed with the object)"""

    def __init__(self, nbytes, dtype=np.uint8):
        self._lib = _lib.load()
        self._p = self._lib.lash_host_alloc_pinned(int(max(nbytes, 1)))
        if not self._p:
            raise MemoryError("lash_host_alloc_pinned(%d)" % nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * int(max(nbytes, 1))).from_address(self._p)).view(dtype)

    def __del__(self):
        try:
            if self._p:
                self._lib.lash_host_free_pinned(self._p)
                self._p = None
        except Exception:
            pass


def ull_estimate(registers, p, estimator="fgra"):
    """FGRA / ML distinct-count estimate of ONE UltraLogLog sketch from its 2^p register bytes (host only; utils.rs:213-217)."""
    regs = np.ascontiguousarray(registers, dtype=np.uint8)
    assert regs.size == 1 << int(p)
    return float(_lib.load().lash_ull_estimate(regs.ctypes.data, int(p), ULL_ESTIMATORS[estimator]))


class LashError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lash error %d: %s" % (code, msg))
        self.code = code


def _algo(a):
    if isinstance(a, str):
        if a not in ALGOS:
            # main.rs:245 panics with this text
            raise LashError(_lib.EINVAL, "Algorithm must be either hmh, ull, or hll")
        return ALGOS[a]
    return int(a)


def parse_layout(spec):
    """"key=value,..." (see lash_layout_parse in include/lash_gfx950.h) -> Layout; None / "" = the default."""
    if isinstance(spec, Layout):
        return spec
    lay = Layout()
    rc = _lib.load().lash_layout_parse((spec or "").encode(), C.byref(lay))
    if rc != _lib.OK:
        raise LashError(rc, "bad layout spec %r" % (spec,))
    return lay


def image_bytes(algo, p=0, layout=None):
    if layout is None:
        return int(_lib.load().lash_sketch_image_bytes(_algo(algo), int(p)))
    return int(_lib.load().lash_layout_image_bytes(C.byref(parse_layout(layout)), _algo(algo), int(p)))


def header_bytes(algo, layout=None):
    lay = parse_layout(layout)
    return int(_lib.load().lash_layout_header_bytes(C.byref(lay), _algo(algo)))


def params_check(algo, k, p=0):
    prm = Params(_algo(algo), int(k), int(p), 0, 0)
    return int(_lib.load().lash_params_check(C.byref(prm)))


def records_to_arrays(genomes):
    """genomes: list of lists of record byte strings -> (seq u8, rec_off u64, genome_rec_off u64)."""
    recs = [r for g in genomes for r in g]
    seq = np.frombuffer(b"".join(recs), dtype=np.uint8).copy() if recs else np.zeros(0, np.uint8)
    rec_off = np.zeros(len(recs) + 1, dtype=np.uint64)
    if recs:
        rec_off[1:] = np.cumsum([len(r) for r in recs], dtype=np.uint64)
    goff = np.zeros(len(genomes) + 1, dtype=np.uint64)
    if genomes:
        goff[1:] = np.cumsum([len(g) for g in genomes], dtype=np.uint64)
    return seq, rec_off, goff


def _ptr(x):
    """device pointer of a torch tensor / raw int, or host pointer of a numpy array"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


class Packed:
    """Device-resident 2-bit genomes (lash_packed*)."""

    def __init__(self, ctx, handle, n_genomes):
        self._ctx, self._h, self.n_genomes = ctx, handle, n_genomes

    @property
    def device_bytes(self):
        return int(_lib.load().lash_packed_bytes(self._h))

    def free(self):
        if self._h:
            _lib.load().lash_packed_free(self._ctx._h, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One lash_ctx: a GPU, a stream and the HBM workspace.  Not thread-safe (like the C object)."""

    def __init__(self, device=0, stream=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.lash_ctx_create(C.byref(h), int(device))
        if rc != _lib.OK:
            raise LashError(rc, self._lib.lash_strerror(rc).decode())
        self._h = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _lib.OK:
            detail = self._lib.lash_ctx_last_error(self._h).decode() if rc in (_lib.EHIP, _lib.ENOMEM) else ""
            raise LashError(rc, self._lib.lash_strerror(rc).decode() + (": " + detail if detail else ""))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lash_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream):
        """stream: a raw hipStream_t as int, or an object with .cuda_stream (torch.cuda.Stream); None = own stream"""
        ptr = None if stream is None else int(getattr(stream, "cuda_stream", stream))
        self._check(self._lib.lash_ctx_set_stream(self._h, ptr))

    def synchronize(self):
        self._check(self._lib.lash_ctx_synchronize(self._h))

    def set_layout(self, layout=None):
        """layout: None (default), a spec string, or a Layout.  Every later call reads / writes images in that layout."""
        self._layout = None if layout is None else parse_layout(layout)
        self._check(self._lib.lash_ctx_set_layout(self._h, None if self._layout is None else C.byref(self._layout)))

    def image_bytes(self, algo, p=0):
        return image_bytes(algo, p, getattr(self, "_layout", None))

    def enable_timing(self, on=True):
        self._check(self._lib.lash_ctx_enable_timing(self._h, 1 if on else 0))

    def timing(self):
        t = Timing()
        self._check(self._lib.lash_ctx_get_timing(self._h, C.byref(t)))
        return {f[0]: getattr(t, f[0]) for f in Timing._fields_ if f[0] != "reserved"}

    @staticmethod
    def _params(algo, k, p, seed, flags):
        return Params(_algo(algo), int(k), int(p or 0), int(flags), int(seed) & (2**64 - 1))

    # -- hot path ---------------------------------------------------------------------------------------------
    def sketch_batch(self, algo, k, p, seed, seq, rec_off, genome_rec_off, flags=0, out=None):
        """Host buffers in, images[n_genomes, image_bytes] (numpy uint8) out."""
        prm = self._params(algo, k, p, seed, flags)
        self._check(self._lib.lash_params_check(C.byref(prm)))
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        goff = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
        n_g, n_rec = len(goff) - 1, len(rec_off) - 1
        ib = self.image_bytes(prm.algo, prm.p)
        if out is None:
            out = np.zeros((n_g, ib), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size == n_g * ib and out.flags.c_contiguous
        seq_p = seq.ctypes.data if seq.size else None
        self._check(self._lib.lash_sketch_batch(self._h, C.byref(prm), seq_p, rec_off.ctypes.data, n_rec,
                                                goff.ctypes.data, n_g, out.ctypes.data if out.size else None))
        return out

    def sketch_batch_async(self, algo, k, p, seed, seq, rec_off, genome_rec_off, out, flags=0):
        """lash_sketch_batch_async: queue one batch (host buffers in, images into `out`) and return; synchronize() before
        reading `out` or touching seq / rec_off.  Pass page-locked arrays (pinned_array) for the copies to overlap."""
        prm = self._params(algo, k, p, seed, flags)
        assert seq.dtype == np.uint8 and rec_off.dtype == np.uint64 and out.dtype == np.uint8
        assert seq.flags.c_contiguous and rec_off.flags.c_contiguous and out.flags.c_contiguous
        goff = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
        self._keep_async = getattr(self, "_keep_async", [])[-4:] + [(seq, rec_off, out)]
        self._check(self._lib.lash_sketch_batch_async(self._h, C.byref(prm), seq.ctypes.data if seq.size else None, rec_off.ctypes.data,
                                                      len(rec_off) - 1, goff.ctypes.data, len(goff) - 1, out.ctypes.data if out.size else None))

    def sketch_files_raw(self, algo, k, p, seed, files_bytes, flags=0):
        """files_bytes: list of uncompressed FASTA / FASTQ file contents (bytes).  The parse runs on the GPU.
        Returns images[n_files, image_bytes]."""
        prm = self._params(algo, k, p, seed, flags)
        self._check(self._lib.lash_params_check(C.byref(prm)))
        raw = np.frombuffer(b"".join(files_bytes), dtype=np.uint8) if files_bytes else np.zeros(0, np.uint8)
        off = np.zeros(len(files_bytes) + 1, dtype=np.uint64)
        if files_bytes:
            off[1:] = np.cumsum([len(f) for f in files_bytes], dtype=np.uint64)
        for f in files_bytes:                                   # parse_fastx_file(..).expect("Invalid input file"), utils.rs:453
            if f[:1] not in (b">", b"@"):
                raise LashError(_lib.EINVAL, "Invalid input file: the first byte must be '>' (FASTA) or '@' (FASTQ)")
        fmt = np.array([_lib.FMT_FASTQ if f[:1] == b"@" else _lib.FMT_FASTA for f in files_bytes], dtype=np.uint8)
        ib = self.image_bytes(prm.algo, prm.p)
        out = np.zeros((len(files_bytes), ib), dtype=np.uint8)
        self._check(self._lib.lash_sketch_files_raw(self._h, C.byref(prm), raw.ctypes.data if raw.size else None,
                                                    off.ctypes.data, fmt.ctypes.data if fmt.size else None,
                                                    len(files_bytes), out.ctypes.data if out.size else None))
        return out

    def format_errors(self):
        """indices of the files of the last raw call whose FASTQ structure broke (lash_ctx_format_errors)"""
        n = int(self._lib.lash_ctx_format_errors(self._h, None, 0))
        idx = (C.c_uint32 * max(n, 1))()
        self._lib.lash_ctx_format_errors(self._h, idx, n)
        return [int(idx[i]) for i in range(n)]

    def hll_inexact_sums(self):
        """genomes of the last HyperLogLog sketch call with a register above 53 - p (include/lash_gfx950.h:
        lash_ctx_hll_inexact_sums); synchronizes the stream"""
        n = int(self._lib.lash_ctx_hll_inexact_sums(self._h, None, 0))
        idx = (C.c_uint32 * max(n, 1))()
        self._lib.lash_ctx_hll_inexact_sums(self._h, idx, n)
        return [int(idx[i]) for i in range(n)]

    def hll_replay_sums_device(self, k, p, seed, d_seq, d_rec_off, n_rec, genome_rec_off, d_images, flags=0):
        """After sketch_batch_device("hll", ...) with the same arguments: the genomes hll_inexact_sums() lists get the `sum` the
        reference's incremental rule leaves (include/lash_gfx950.h: lash_hll_replay_sums_device); synchronizes."""
        prm = self._params("hll", k, p, seed, flags)
        goff = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
        self._check(self._lib.lash_hll_replay_sums_device(self._h, C.byref(prm), _ptr(d_seq), _ptr(d_rec_off), int(n_rec),
                                                          goff.ctypes.data_as(C.c_void_p), len(goff) - 1, _ptr(d_images)))

    def sketch_batch_device(self, algo, k, p, seed, d_seq, d_rec_off, n_rec, genome_rec_off, genome_byte_off, d_out,
                            flags=0):
        """Device-resident records in, device images out; asynchronous on the context's stream."""
        prm = self._params(algo, k, p, seed, flags)
        goff = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
        gbo = np.ascontiguousarray(genome_byte_off, dtype=np.uint64)
        assert len(goff) == len(gbo)
        self._keep = (goff, gbo)
        self._check(self._lib.lash_sketch_batch_device(self._h, C.byref(prm), _ptr(d_seq), _ptr(d_rec_off), int(n_rec),
                                                       goff.ctypes.data, gbo.ctypes.data, len(goff) - 1, _ptr(d_out)))

    def pack_device(self, d_seq, d_rec_off, n_rec, genome_rec_off, genome_byte_off):
        goff = np.ascontiguousarray(genome_rec_off, dtype=np.uint64)
        gbo = np.ascontiguousarray(genome_byte_off, dtype=np.uint64)
        h = C.c_void_p()
        self._check(self._lib.lash_pack_device(self._h, _ptr(d_seq), _ptr(d_rec_off), int(n_rec), goff.ctypes.data,
                                               gbo.ctypes.data, len(goff) - 1, C.byref(h)))
        return Packed(self, h, len(goff) - 1)

    def sketch_packed_device(self, algo, k, p, seed, packed, d_out, flags=0):
        prm = self._params(algo, k, p, seed, flags)
        self._check(self._lib.lash_sketch_packed_device(self._h, C.byref(prm), packed._h, _ptr(d_out)))

    def merge_images_device(self, algo, p, d_dst, d_src, n_images):
        self._check(self._lib.lash_merge_images_device(self._h, _algo(algo), int(p or 0), _ptr(d_dst), _ptr(d_src),
                                                       int(n_images)))

    def merge_images(self, algo, p, dst, src):
        """dst[i] = dst[i] U src[i] on the GPU; numpy uint8 arrays [n, image_bytes]; returns dst."""
        assert dst.dtype == np.uint8 and src.dtype == np.uint8 and dst.shape == src.shape
        assert dst.flags.c_contiguous and src.flags.c_contiguous
        n = dst.shape[0] if dst.ndim == 2 else 1
        self._check(self._lib.lash_merge_images(self._h, _algo(algo), int(p or 0), dst.ctypes.data, src.ctypes.data, n))
        return dst

    def hmh_pair_expected_collisions(self, ref_card, qry_card):
        """hyperminhash's expected_collisions(n, m) for every pair, float64 [n_ref, n_qry]; the 65 536-cell regime (both
        cardinalities <= 2^19) runs on the GPU (include/lash_gfx950.h)"""
        r = np.ascontiguousarray(ref_card, dtype=np.float64)
        q = np.ascontiguousarray(qry_card, dtype=np.float64)
        out = np.zeros((len(r), len(q)), dtype=np.float64)
        self._check(self._lib.lash_hmh_pair_expected_collisions(self._h, r.ctypes.data, len(r), q.ctypes.data, len(q), out.ctypes.data))
        return out

    def hmh_pair_counts(self, ref_images, qry_images):
        """HyperMinHash pair statistics (C, N) of serialized sketches: numpy uint8 [n, 32768] in, two uint32
        [n_ref, n_qry] arrays out (hyperminhash Sketch::similarity's register scan, utils.rs:164)."""
        ref = np.ascontiguousarray(ref_images, dtype=np.uint8)
        qry = np.ascontiguousarray(qry_images, dtype=np.uint8)
        ib = self.image_bytes("hmh")
        assert ref.ndim == 2 and qry.ndim == 2 and ref.shape[1] == ib and qry.shape[1] == ib
        c = np.zeros((ref.shape[0], qry.shape[0]), dtype=np.uint32)
        n = np.zeros_like(c)
        self._check(self._lib.lash_hmh_pair_counts(self._h, ref.ctypes.data, ref.shape[0], qry.ctypes.data, qry.shape[0],
                                                   c.ctypes.data, n.ctypes.data))
        return c, n

    def hll_pair_union_stats(self, p, ref_images, qry_images):
        """HyperLogLog union statistics of serialized sketches: numpy uint8 [n, 33 + 2^p] in; (zero uint32, sum float64)
        [n_ref, n_qry] out — what `len()` reads after `union` (utils.rs:355-363)."""
        ref = np.ascontiguousarray(ref_images, dtype=np.uint8)
        qry = np.ascontiguousarray(qry_images, dtype=np.uint8)
        ib = self.image_bytes("hll", p)
        assert ref.ndim == 2 and qry.ndim == 2 and ref.shape[1] == ib and qry.shape[1] == ib
        zero = np.zeros((ref.shape[0], qry.shape[0]), dtype=np.uint32)
        usum = np.zeros((ref.shape[0], qry.shape[0]), dtype=np.float64)
        self._check(self._lib.lash_hll_pair_union_stats(self._h, int(p), ref.ctypes.data, ref.shape[0], qry.ctypes.data,
                                                        qry.shape[0], zero.ctypes.data, usum.ctypes.data))
        return zero, usum

    # device-resident forms (asynchronous on the context's stream): image blocks and outputs are device pointers / torch tensors
    def hmh_pair_counts_device(self, d_ref, n_ref, d_qry, n_qry, d_c, d_n):
        self._check(self._lib.lash_hmh_pair_counts_device(self._h, _ptr(d_ref), int(n_ref), _ptr(d_qry), int(n_qry), _ptr(d_c), _ptr(d_n)))

    def hll_pair_union_stats_device(self, p, d_ref, n_ref, d_qry, n_qry, d_zero, d_sum):
        self._check(self._lib.lash_hll_pair_union_stats_device(self._h, int(p), _ptr(d_ref), int(n_ref), _ptr(d_qry), int(n_qry),
                                                               _ptr(d_zero), _ptr(d_sum)))

    def ull_pair_union_estimates_device(self, p, estimator, d_ref, n_ref, d_qry, n_qry, d_est):
        self._check(self._lib.lash_ull_pair_union_estimates_device(self._h, int(p), ULL_ESTIMATORS[estimator], _ptr(d_ref), int(n_ref),
                                                                   _ptr(d_qry), int(n_qry), _ptr(d_est)))

    def ull_pair_union_estimates(self, p, ref_images, qry_images, estimator="fgra"):
        """UltraLogLog: estimated distinct count of merge(ref_i, qry_j) for every pair (utils.rs:260-270): numpy uint8
        [n, header + 2^p] in, float64 [n_ref, n_qry] out."""
        ref = np.ascontiguousarray(ref_images, dtype=np.uint8)
        qry = np.ascontiguousarray(qry_images, dtype=np.uint8)
        ib = self.image_bytes("ull", p)
        assert ref.ndim == 2 and qry.ndim == 2 and ref.shape[1] == ib and qry.shape[1] == ib
        est = np.zeros((ref.shape[0], qry.shape[0]), dtype=np.float64)
        self._check(self._lib.lash_ull_pair_union_estimates(self._h, int(p), ULL_ESTIMATORS[estimator], ref.ctypes.data, ref.shape[0],
                                                            qry.ctypes.data, qry.shape[0], est.ctypes.data))
        return est

    # -- dist side, resident form (include/lash_gfx950.h: lash_sketch_set_*) ---------------------------------------------
    def sketch_set(self, algo, p, images, order=None):
        """N serialized sketches resident on the context's GPU: numpy uint8 [n, image_bytes] (uploaded once; `order` = member
        indices into images) or a CUDA torch tensor (adopted, not copied: keep it alive)."""
        return SketchSet(self, algo, p, images, order)

    def synth_genomes_device(self, first_genome, n_genomes, n_bases, d_out):
        self._check(self._lib.lash_synth_genomes_device(self._h, int(first_genome), int(n_genomes), int(n_bases),
                                                        _ptr(d_out)))


class SketchSet:
    """lash_sketch_set: the sketches of a `dist` run kept in HBM (reference: the two hash maps of utils.rs:107-127), with
    per-member cardinalities from GPU register histograms and pair statistics by row block, lower triangle aware."""

    def __init__(self, ctx, algo, p, images, order=None):
        self._ctx, self._lib = ctx, ctx._lib
        self.algo, self.p = _algo(algo), int(p or 0)
        h = C.c_void_p()
        ib = ctx.image_bytes(self.algo, self.p)
        if isinstance(images, np.ndarray):
            img = np.ascontiguousarray(images, dtype=np.uint8)
            assert img.ndim == 2 and img.shape[1] == ib
            o = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
            n = img.shape[0] if o is None else len(o)
            ctx._check(self._lib.lash_sketch_set_create(ctx._h, self.algo, self.p, img.ctypes.data if img.size else None, img.shape[0],
                                                        None if o is None else o.ctypes.data, n, C.byref(h)))
        else:
            assert order is None and images.is_cuda and images.is_contiguous() and images.numel() % ib == 0
            self._keep = images
            n = images.numel() // ib
            ctx._check(self._lib.lash_sketch_set_create_device(ctx._h, self.algo, self.p, images.data_ptr() if n else None, n, C.byref(h)))
        self._h, self.n = h, n

    def free(self):
        if getattr(self, "_h", None) and self._ctx._h:
            self._lib.lash_sketch_set_free(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def cardinalities(self, estimator="fgra", hll_bias=None):
        out = np.zeros(self.n, dtype=np.float64)
        bad = C.c_uint32()
        rc = self._lib.lash_sketch_set_cardinalities(self._ctx._h, self._h, ULL_ESTIMATORS[estimator], _bias_handle(hll_bias),
                                                     out.ctypes.data, C.byref(bad))
        if rc == _lib.ERANGE:
            raise LashError(rc, self._lib.lash_strerror(rc).decode() + " (sketch %d)" % bad.value)
        self._ctx._check(rc)
        return out

    def prepare(self, qry=None):
        self._ctx._check(self._lib.lash_sketch_set_prepare(self._ctx._h, self._h, (qry or self)._h))

    def hmh_expected_collisions(self, r0, r1, qry=None, n_cols=None):
        """hyperminhash's expected collisions of the block's SMALL pairs (both sketches <= 2^19 distinct k-mers) as a float64
        [r1 - r0, n_cols] array for lash_dist_rows' hmh_ec (other entries are not read), or None when the block has none.
        cardinalities() must have run on both sets."""
        q = qry or self
        nc = q.n if n_cols is None else int(n_cols)
        out = np.empty((int(r1) - int(r0), nc), dtype=np.float64)
        cnt = C.c_uint64()
        self._ctx._check(self._lib.lash_sketch_set_hmh_expected_collisions(self._ctx._h, self._h, int(r0), int(r1), q._h, nc, out.ctypes.data, C.byref(cnt)))
        return out if cnt.value else None

    def pair_block(self, r0, r1, qry=None, n_cols=None, triangle=False, estimator="fgra", out=None):
        """statistics of rows [r0, r1) against columns [0, n_cols) of `qry` (default: this set) as the dict lash_dist_rows takes.
        `out`: optional dict of preallocated (e.g. pinned) flat arrays 'c', 'n' (uint32), 'u' (float64) of >= (r1-r0)*n_cols."""
        q = qry or self
        nc = q.n if n_cols is None else int(n_cols)
        nr = int(r1) - int(r0)
        np_ = nr * nc

        def buf(key, dt):
            if out is not None and key in out:
                return out[key][:np_].reshape(nr, nc)
            return np.empty((nr, nc), dtype=dt)
        c = buf("c", np.uint32) if self.algo != _lib.ULL else None
        n = buf("n", np.uint32) if self.algo == _lib.HMH else None
        u = buf("u", np.float64) if self.algo != _lib.HMH else None
        self._ctx._check(self._lib.lash_sketch_set_pair_block(self._ctx._h, self._h, int(r0), int(r1), q._h, nc, 1 if triangle else 0,
                                                              ULL_ESTIMATORS[estimator], None if c is None else c.ctypes.data,
                                                              None if n is None else n.ctypes.data, None if u is None else u.ctypes.data))
        st = {}
        if c is not None:
            st["c_or_zero"] = c
        if n is not None:
            st["n_counts"] = n
        if u is not None:
            st["sum_or_union"] = u
        return st
