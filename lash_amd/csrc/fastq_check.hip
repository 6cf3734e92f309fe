// fastq_check.hip — needletail's OTHER FASTQ error on the device: a quality line whose length differs from its sequence line's
// (and a file that ends inside a record).
//
// The pack stage's raw-file parse (pack_kernels.hip) counts newlines modulo 4 and checks that phase-0 lines start with '@' and
// phase-2 lines with '+' (/root/reference/src/utils.rs:453-459: needletail's iterator ends with an Err at a malformed record).
// A length mismatch keeps that four-line structure intact, so it went unseen on the device: lash_sketch_files_raw_device then
// sketched such a file to its end where the reference stops (round 2's documented divergence; only the `lash` CLI closed it, on
// host threads).  Line lengths are differences of newline POSITIONS, a second scan; it is done here in three small kernels that
// only read the file bytes, beside the pack kernel rather than inside its look-back:
//   1. fq_count_kernel  per 4 KiB block: newline count, positions of its last three newlines;
//   2. fq_scan_kernel   per file: exclusive scan of the counts (a block's first line number);
//   3. fq_check_kernel  per block: every newline that ends a QUALITY line (line number = 3 mod 4) looks up the three newlines
//                       before it — in the block's bitmap, else in the earlier blocks' last-three tables — and compares
//                       CR-stripped lengths; the thread on the file's last byte checks how the file ends.
// A mismatch sets file_err[g], the flag the structure check already uses: the host-buffer entry re-does such a file through the
// exact host parse, the device-buffer entry reports LASH_EFORMAT.  Cost: two more reads of the FASTQ bytes at the copy rate.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lash_kernels.h"

namespace lash {

constexpr uint32_t FQ_BLOCK = 4096;          // bytes per block = 256 threads x 16
constexpr uint32_t NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t nl_mask16(const uint8_t *p, uint64_t avail)
{
    // bit i: byte i of the 16 at p is '\n' (bytes at or beyond `avail` never are)
    uint32_t m = 0;
    if (avail >= 16) {
        uint4 q;
        __builtin_memcpy(&q, p, 16);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t z = w[j] ^ 0x0A0A0A0Au;
            const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;                 // high bit of a byte set <=> the byte is non-zero
            m |= ((((~nz >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xFu) << (4 * j);
        }
    } else {
        for (uint32_t i = 0; i < (uint32_t)avail; ++i) m |= (p[i] == 0x0Au ? 1u : 0u) << i;
    }
    return m;
}

__device__ __forceinline__ uint32_t file_of_block(const FqFile *files, uint32_t n_files, uint32_t b)
{
    uint32_t lo = 0, hi = n_files;                       // the last file whose block0 <= b
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (files[mid].block0 <= b) lo = mid; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(256) fq_count_kernel(const uint8_t *__restrict__ raw, const FqFile *__restrict__ files, uint32_t n_files,
                                                       uint32_t *__restrict__ cnt, uint32_t *__restrict__ last3)
{
    __shared__ uint32_t bm[FQ_BLOCK / 32];
    __shared__ uint32_t n_sh;
    const uint32_t b = blockIdx.x, f = file_of_block(files, n_files, b);
    const FqFile ff = files[f];
    const uint64_t boff = (uint64_t)(b - ff.block0) * FQ_BLOCK, at = boff + threadIdx.x * 16ull;
    const uint32_t m = at < ff.len ? nl_mask16(raw + ff.off + at, ff.len - at) : 0u;
    if ((threadIdx.x & 1u) == 0u) bm[threadIdx.x >> 1] = 0;
    if (threadIdx.x == 0) n_sh = 0;
    __syncthreads();
    if (m) { atomicOr(&bm[threadIdx.x >> 1], m << (16u * (threadIdx.x & 1u))); atomicAdd(&n_sh, (uint32_t)__builtin_popcount(m)); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t found = 0, pos[3] = {NONE, NONE, NONE};
        for (int w = FQ_BLOCK / 32 - 1; w >= 0 && found < 3; --w) {
            uint32_t x = bm[w];
            while (x && found < 3) { const uint32_t i = 31u - (uint32_t)__builtin_clz(x); pos[found++] = (uint32_t)boff + 32u * (uint32_t)w + i; x &= ~(1u << i); }
        }
        cnt[b] = n_sh;
        last3[3ull * b] = pos[0]; last3[3ull * b + 1] = pos[1]; last3[3ull * b + 2] = pos[2];      // nearest first
    }
}

// one workgroup per file: base[b] = newlines of the file before block b; total[f] = all of them
__global__ void __launch_bounds__(1024) fq_scan_kernel(const FqFile *__restrict__ files, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ base,
                                                       uint32_t *__restrict__ total)
{
    __shared__ uint32_t part[1024];
    const FqFile ff = files[blockIdx.x];
    const uint32_t n = ff.n_blocks, per = (n + 1023u) / 1024u, b0 = threadIdx.x * per, b1 = min(n, b0 + per);
    uint32_t s = 0;
    for (uint32_t b = b0; b < b1; ++b) s += cnt[ff.block0 + b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (uint32_t i = 0; i < 1024; ++i) { const uint32_t v = part[i]; part[i] = acc; acc += v; } total[blockIdx.x] = acc; }
    __syncthreads();
    uint32_t acc = part[threadIdx.x];
    for (uint32_t b = b0; b < b1; ++b) { base[ff.block0 + b] = acc; acc += cnt[ff.block0 + b]; }
}

__global__ void __launch_bounds__(256) fq_check_kernel(const uint8_t *__restrict__ raw, const FqFile *__restrict__ files, uint32_t n_files,
                                                       const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ last3,
                                                       const uint32_t *__restrict__ base, const uint32_t *__restrict__ total,
                                                       uint32_t *__restrict__ file_err)
{
    __shared__ uint32_t bm[FQ_BLOCK / 32];
    __shared__ uint32_t wsum[4];
    const uint32_t b = blockIdx.x, f = file_of_block(files, n_files, b);
    const FqFile ff = files[f];
    const uint8_t *fp = raw + ff.off;
    const uint32_t lb = b - ff.block0, boff = lb * FQ_BLOCK;
    const uint64_t at = (uint64_t)boff + threadIdx.x * 16ull;
    const uint32_t m = at < ff.len ? nl_mask16(fp + at, ff.len - at) : 0u;
    if ((threadIdx.x & 1u) == 0u) bm[threadIdx.x >> 1] = 0;
    __syncthreads();
    if (m) atomicOr(&bm[threadIdx.x >> 1], m << (16u * (threadIdx.x & 1u)));
    __syncthreads();
    // position of the newline before file position `pos` (NONE: there is none): this block's bitmap, then earlier blocks' tables
    auto prev_nl = [&](uint32_t pos) -> uint32_t {
        uint32_t q = pos - boff;                                             // bits [0, q) of this block are before pos
        if (pos >= boff) {
            int w = (int)(q >> 5);
            uint32_t x = w < (int)(FQ_BLOCK / 32) ? bm[w] & ((q & 31u) ? ((1u << (q & 31u)) - 1u) : 0u) : 0u;
            for (;;) {
                if (x) return boff + 32u * (uint32_t)w + 31u - (uint32_t)__builtin_clz(x);
                if (--w < 0) break;
                x = bm[w];
            }
        }
        // earlier blocks: the nearest newline before pos in block j < lb is last3[j][0] — unless pos itself lies in an earlier block
        // (a walk that has already left this block), then the entries of that block that are below pos
        for (int64_t j = (int64_t)min(lb, pos / FQ_BLOCK + 1u) - 1; j >= 0; --j) {
            const uint64_t t = 3ull * (ff.block0 + (uint32_t)j);
            for (int e = 0; e < 3; ++e) { const uint32_t v = last3[t + e]; if (v != NONE && v < pos) return v; }
            if (cnt[ff.block0 + (uint32_t)j] > 3u && (uint32_t)j == pos / FQ_BLOCK) {
                // more than three newlines in the block that holds pos, all three known ones at or after pos: scan its bytes (rare)
                for (uint32_t i = pos; i > (uint32_t)j * FQ_BLOCK; --i) if (fp[i - 1] == 0x0Au) return i - 1;
            }
        }
        return NONE;
    };
    auto stripped_len = [&](uint32_t start, uint32_t end) -> uint32_t {      // [start, end) without trailing '\r' (needletail strips them)
        while (end > start && fp[end - 1] == 0x0Du) --end;
        return end - start;
    };
    // a quality line ends at `qend` (a newline, or the end of a file without a final newline), `before` = the newline before it
    auto check_record = [&](uint32_t qend) {
        const uint32_t n2 = prev_nl(qend);                                   // ends the '+' line
        if (n2 == NONE) { file_err[ff.index] = 1u; return; }
        const uint32_t n1 = prev_nl(n2);                                     // ends the sequence line
        if (n1 == NONE) { file_err[ff.index] = 1u; return; }
        const uint32_t n0 = prev_nl(n1);                                     // ends the header line
        if (n0 == NONE) { file_err[ff.index] = 1u; return; }
        if (stripped_len(n0 + 1u, n1) != stripped_len(n2 + 1u, qend)) file_err[ff.index] = 1u;
    };
    // line number of a lane's first newline = newlines of the file before it: block base + waves before + lanes before
    const uint32_t mine = (uint32_t)__builtin_popcount(m);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if ((int)(threadIdx.x & 63u) >= d) incl += v; }
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (m) {
        uint32_t before = base[b] + incl - mine;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
        const uint32_t q = threadIdx.x * 16u;
        uint32_t mm = m;
        while (mm) {
            const uint32_t i = (uint32_t)__builtin_ctz(mm);
            mm &= mm - 1u;
            if ((before & 3u) == 3u) check_record(boff + q + i);             // this newline ends line number `before`
            ++before;
        }
    }
    // how the file ends: complete <=> a final newline after a whole number of records, or no final newline with the quality line
    // as the (non-empty) last line.  Anything else is a record cut short: flagged (the host parse decides what exactly stands).
    if (ff.len && at <= ff.len - 1 && ff.len - 1 < at + 16) {
        const uint32_t T = total[f];
        const bool ends_nl = fp[ff.len - 1] == 0x0Au;
        if ((T & 3u) == 3u) check_record((uint32_t)ff.len);                  // the last quality line has no newline (it may be empty)
        else if (!ends_nl || (T & 3u) != 0u) file_err[ff.index] = 1u;
    }
}

hipError_t launch_fastq_check(const uint8_t *d_raw, const FqFile *d_files, uint32_t n_files, uint32_t n_blocks, uint32_t *d_scratch,
                              uint32_t *d_file_err, hipStream_t stream)
{
    if (n_files == 0 || n_blocks == 0) return hipSuccess;
    uint32_t *cnt = d_scratch, *base = cnt + n_blocks, *total = base + n_blocks, *last3 = total + ((n_files + 3u) & ~3u);
    hipLaunchKernelGGL(fq_count_kernel, dim3(n_blocks), dim3(256), 0, stream, d_raw, d_files, n_files, cnt, last3);
    hipLaunchKernelGGL(fq_scan_kernel, dim3(n_files), dim3(1024), 0, stream, d_files, cnt, base, total);
    hipLaunchKernelGGL(fq_check_kernel, dim3(n_blocks), dim3(256), 0, stream, d_raw, d_files, n_files, cnt, last3, base, total, d_file_err);
    return hipGetLastError();
}

size_t fastq_check_scratch_words(uint32_t n_files, uint32_t n_blocks) { return (size_t)n_blocks * 5 + ((n_files + 3u) & ~3u) + 16; }
uint32_t fastq_check_block_bytes() { return FQ_BLOCK; }

}  // namespace lash
