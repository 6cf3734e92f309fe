#include "fastx.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "codec_dl.hpp"
#include "inflate_fast.hpp"
#include "pgzip.hpp"
#include "zstd_dl.hpp"

namespace lashhost {

struct ByteStream::Impl {
    int kind = 0;               // 0 plain, 2 zstd, 3 bzip2 / xz, 4 gzip
    FILE *f = nullptr;
    ParallelGzip pg;                // kind 4: multi-member aware; set_threads() > 1 adds speculative inflate threads
    int threads = 1;
    ZstdReader z;
    DlDecoder d;
};

ByteStream::ByteStream() : impl_(new Impl()) {}
ByteStream::~ByteStream()
{
    if (impl_->f) fclose(impl_->f);
    delete impl_;
}

void ByteStream::set_threads(int t) { impl_->threads = t < 1 ? 1 : t; }

std::string ByteStream::open(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "Invalid input file: cannot open " + path;
    unsigned char m[6] = {0, 0, 0, 0, 0, 0};
    const size_t got = fread(m, 1, 6, f);
    if (got >= 2 && m[0] == 0x1f && m[1] == 0x8b) {              // gzip: pgzip.hpp, sequential when threads == 1
        fclose(f);
        impl_->kind = 4;
        return impl_->pg.open(path, impl_->threads);
    }
    rewind(f);
    if (got >= 3 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h') { impl_->kind = 3; return impl_->d.open(f, Codec::BZIP2); }
    if (got >= 6 && m[0] == 0xfd && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z' && m[5] == 0) { impl_->kind = 3; return impl_->d.open(f, Codec::XZ); }
    if (got >= 4 && m[0] == 0x28 && m[1] == 0xb5 && m[2] == 0x2f && m[3] == 0xfd) {
        impl_->kind = 2;
        return impl_->z.open(f);
    }
    impl_->f = f;
    return "";
}

long ByteStream::read(uint8_t *dst, size_t n, std::string &err)
{
    if (impl_->kind == 0) return (long)fread(dst, 1, n, impl_->f);
    if (impl_->kind == 2) return impl_->z.read(dst, n, err);
    if (impl_->kind == 3) return impl_->d.read(dst, n, err);
    return impl_->pg.read(dst, n, err);
}

std::string slurp_maybe_compressed(const std::string &path, std::vector<uint8_t> &out)
{
    out.clear();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "Invalid input file: cannot open " + path;
    unsigned char magic[6] = {0, 0, 0, 0, 0, 0};
    size_t got = fread(magic, 1, 6, f);
    if (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        // inflate_fast.hpp first (1.5-1.7x zlib on sequence text); zlib stays the arbiter: whatever the fast decoder
        // rejects is read again with gzread, so a defect there costs time, never data
        if (!getenv("LASH_NO_FAST_INFLATE")) {
            std::vector<uint8_t> gz(magic, magic + got);
            {
                std::vector<uint8_t> buf(1 << 22);
                size_t n;
                while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) gz.insert(gz.end(), buf.begin(), buf.begin() + n);
            }
            ByteSink sink;
            const char *e = gunzip_members(gz.data(), gz.size(), sink, false, nullptr);
            if (!e) {
                fclose(f);
                out.assign(sink.p, sink.p + sink.n);
                return "";
            }
            if (getenv("LASH_INFLATE_VERBOSE")) fprintf(stderr, "[lash] %s: fast inflate said \"%s\", reading again with zlib\n", path.c_str(), e);
        }
        fclose(f);
        gzFile g = gzopen(path.c_str(), "rb");
        if (!g) return "Invalid input file: gzopen failed for " + path;
        gzbuffer(g, 1 << 20);
        std::vector<uint8_t> buf(1 << 22);
        for (;;) {
            int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n < 0) { gzclose(g); return "Invalid input file: corrupt gzip stream in " + path; }
            if (n == 0) break;
            out.insert(out.end(), buf.begin(), buf.begin() + n);
        }
        gzclose(g);
        return "";
    }
    // rest of the file
    std::vector<uint8_t> raw(magic, magic + got);
    {
        std::vector<uint8_t> buf(1 << 22);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) raw.insert(raw.end(), buf.begin(), buf.begin() + n);
        fclose(f);
    }
    if (got >= 4 && magic[0] == 0x28 && magic[1] == 0xb5 && magic[2] == 0x2f && magic[3] == 0xfd) {
        std::string err = zstd_decompress_all(raw.data(), raw.size(), out);
        return err.empty() ? "" : "Invalid input file: " + err + " (" + path + ")";
    }
    const bool bz = got >= 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h';
    const bool xz = got >= 6 && magic[0] == 0xfd && magic[1] == '7' && magic[2] == 'z' && magic[3] == 'X' && magic[4] == 'Z' && magic[5] == 0;
    if (bz || xz) {
        std::vector<uint8_t>().swap(raw);
        ByteStream bs;
        std::string err = bs.open(path);
        if (!err.empty()) return err;
        std::vector<uint8_t> buf(1 << 22);
        for (;;) {
            const long n = bs.read(buf.data(), buf.size(), err);
            if (n < 0) return err + " (" + path + ")";
            if (n == 0) break;
            out.insert(out.end(), buf.begin(), buf.begin() + n);
        }
        return "";
    }
    out.swap(raw);
    return "";
}

// needletail's record rules as lash sees them (utils.rs:453-459; the same rules as the library's host path for files the
// device parse flags, and as the oracle): the first byte decides FASTA / FASTQ, anything else is "Invalid input file";
// FASTQ iteration stops at the first record that is not header / sequence / '+' / quality of equal length.
std::string parse_fastx_buffer(const uint8_t *d, size_t n, RecordBatch &out)
{
    if (n == 0) return "";                                          // empty file: no records
    auto line_end = [&](size_t p) { const void *q = memchr(d + p, '\n', n - p); return q ? (size_t)((const uint8_t *)q - d) : n; };
    size_t i = 0;
    if (d[0] == '>') {
        while (i < n) {
            i = line_end(i);                                        // skip the header line
            if (i < n) ++i;
            while (i < n && d[i] != '>') {                          // sequence lines
                size_t e = line_end(i), stop = e;
                while (stop > i && d[stop - 1] == '\r') --stop;
                out.seq.insert(out.seq.end(), d + i, d + stop);
                i = e < n ? e + 1 : n;
            }
            out.rec_off.push_back(out.seq.size());
        }
        return "";
    }
    if (d[0] == '@') {
        while (i < n) {
            if (d[i] != '@') return "";                             // malformed: records so far stand (utils.rs:457)
            const size_t e = line_end(i);                           // header
            if (e >= n) break;
            const size_t s = e + 1, se = line_end(s);               // sequence line
            if (se >= n) break;
            const size_t p = se + 1;
            if (p >= n || d[p] != '+') return "";
            const size_t pe = line_end(p);                          // '+' line
            if (pe >= n) break;
            const size_t qs = pe + 1, qe = line_end(qs);            // quality line
            size_t sl = se - s, ql = qe - qs;
            while (sl && d[s + sl - 1] == '\r') --sl;
            while (ql && d[qs + ql - 1] == '\r') --ql;
            if (sl != ql) return "";
            out.seq.insert(out.seq.end(), d + s, d + s + sl);
            out.rec_off.push_back(out.seq.size());
            i = qe < n ? qe + 1 : n;
        }
        return "";
    }
    return "Invalid input file: neither FASTA ('>') nor FASTQ ('@')";
}

std::string read_fastx_file(const std::string &path, RecordBatch &out)
{
    std::vector<uint8_t> data;
    std::string err = slurp_maybe_compressed(path, data);
    if (!err.empty()) return err;
    return parse_fastx_buffer(data.data(), data.size(), out);
}

}  // namespace lashhost
