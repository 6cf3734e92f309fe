// sketch_rules.h — the device code every sketch kernel of liblash_gfx950 shares: register tables in LDS, the three add_kmer
// rules (/root/reference/src/utils.rs:395-398, 411-413, 427-429), k-mer windows and validity masks, ASCII -> 2-bit conversion,
// filter_out_n (utils.rs:33-41) in its in-place forms, and the end-of-item flush.  Included by sketch_kernels.hip (work items =
// slices of genomes) and sole_kernels.hip (round 5: persistent workgroups over whole small genomes).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "lash_device.h"
#include "lash_kernels.h"


// how far the four-word loop of a tile is unrolled (1: one body, the window rotates through register copies)
constexpr uint32_t BINS_APPLY_RARE = 62;   // bins_apply_kernel: room for the UltraLogLog entries with 32 and more leading zeros of one (genome, bin)
#ifndef LASH_BINS_APPLY_LOADS
#define LASH_BINS_APPLY_LOADS 4         // 16-byte loads a lane of bins_apply_kernel has in flight (8, 16: the same or slower)
#endif
#ifndef LASH_WORD_UNROLL
#define LASH_WORD_UNROLL 1
#endif

namespace lash {

enum { KM_16 = 0, KM_LT16 = 1, KM_GT16 = 2 };

// ------------------------------------------------------------------------------------------------------------
// register spaces: LDS (the normal case) or global memory (2^p too large for 160 KiB of LDS)
//
// What a table word holds WHILE a work item is hashed is chosen for the update's instruction count — one VALU instruction is
// ~4.2 cycles of a SIMD whatever it does (tools/ubench_isa, profiles/r04/isa_cost) — and turned into the register of the image
// by the flush (finish_item), once per item:
//   HyperMinHash   raw = (lz - 1) << 10 | sig  under a SIGNED maximum, -1 = empty  (the rule's "+ 0x400" moves to the flush)
//   HyperLogLog    raw = rho - 1               under a SIGNED maximum, -1 = empty  (v_ffbh's "nothing found" IS the no-op)
//   UltraLogLog    a 64-bit bitmap of the nlz values seen (bit n: some k-mer had nlz = n); hash4j's prefix is that << (p - 1).
//                  nlz < 32 whenever the fast form applies, so the fast form always hits the pair's first word
//   HyperMinHash, deferring launches (LdsThrRegs): the rank as a THRESHOLD, see there
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t RANK_EMPTY = 0xFFFFFFFFu;                   // HyperMinHash / HyperLogLog table word of an empty register

struct LdsRegs {
    uint32_t *base;
    static constexpr bool BINS = false, BYTES = false, QUEUED = false;
    // The register table starts at LDS address 0 (the kernel has no static __shared__; checked at kernel entry), so the
    // word index goes straight into the DS address.  Through a pointer hipcc adds the table's link-time base (0) with a
    // v_add_u32 per k-mer; the update itself is fire-and-forget, nothing in the hashing loop reads the table back, and
    // lds_wait() drains the counter before the flush reads it.
    static constexpr bool THR = false;
#ifdef LASH_ABL_NO_ATOMIC   // timing-only diagnostic build (tools/variants.sh): results are wrong by construction
    __device__ __forceinline__ void smax(uint32_t i, uint32_t v) const { asm volatile("" ::"v"(i), "v"(v)); }
    __device__ __forceinline__ void bor_b(uint32_t b, uint32_t v) const { asm volatile("" ::"v"(b), "v"(v)); }
#else
    __device__ __forceinline__ void smax(uint32_t i, uint32_t v) const
    { asm volatile("ds_max_i32 %0, %1" ::"v"(i << 2), "v"(v) : "memory"); }
    __device__ __forceinline__ void bor_b(uint32_t byte_addr, uint32_t v) const
    { asm volatile("ds_or_b32 %0, %1" ::"v"(byte_addr), "v"(v) : "memory"); }
#endif
    // UltraLogLog: OR `v` into word `w` (0 / 1) of register idx's bitmap pair; hh = the hash's high word (idx = its top p bits)
    __device__ __forceinline__ void bor_first(uint32_t hh, int p, uint32_t v) const { bor_b((hh >> (29 - p)) & ~7u, v); }
    __device__ __forceinline__ void bor_pair(uint32_t idx, uint32_t w, uint32_t v) const { bor_b((idx << 3) | (w << 2), v); }
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return base[i]; }
    static __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};
struct GlobalRegs {                                            // (UltraLogLog p >= 23 only: a zeroed slab per work item)
    uint32_t *base;
    static constexpr bool BINS = false, BYTES = false, QUEUED = false;
    static constexpr bool THR = false;
    __device__ __forceinline__ void smax(uint32_t, uint32_t) const {}
    __device__ __forceinline__ void bor_pair(uint32_t idx, uint32_t w, uint32_t v) const
    { if (v) (void)__hip_atomic_fetch_or(base + 2u * idx + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void bor_first(uint32_t hh, int p, uint32_t v) const { bor_pair(hh >> (32 - p), 0u, v); }
    __device__ __forceinline__ uint32_t get(uint32_t i) const
    { return __hip_atomic_load(base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void lds_wait() {}
};

// A register table larger than 128 KiB of LDS (HyperLogLog p = 16; UltraLogLog p = 15 .. 22) is not updated by the sketch kernels
// at all (round 4; rounds 1-3 hashed every k-mer once per 128 KiB part of the table, 2 .. 16 times, and from p = 19 fell back to
// one L2 atomic per k-mer: 1.7e10 k-mers/s).  Every k-mer is hashed ONCE and leaves an entry, register index << 6 | value
// (HyperLogLog: rho - 1; UltraLogLog: nlz; 63 = nothing), in the list of its BIN — the genome's registers cut into bins of 2^15
// of them, one LDS table's worth of 32-bit words (round 6; before: 2^14 for UltraLogLog's 64-bit bitmaps).  bins_apply_kernel then reads each
// list once and builds its bin's registers in LDS.  In the lists only the bits an entry does not share with its bin travel: six 21-bit
// fields per 16-byte chunk.  HBM: ~3 - 6 bytes written + as many read per k-mer beside the 1 byte of input (profiles/r06/large_tables/);
// no atomics in the data path.
//   The scatter is staged per WAVE: push() takes a slot in the wave's LDS area of its bin (returning ds_add on the bin's counter),
// and after each word of 16 k-mers per lane the wave reserves room in the bins' lists (one returning global atomic per bin holding
// entries, issued by 64 lanes at once) and copies the staged entries out, a bin at a time, lanes side by side.  Entries that find
// their staging row or their list full go straight into the genome's full-size table in global memory (the rounds 1-3 path: exact,
// slow, and only met by genomes whose k-mers pile into few buckets — a satellite repeat); bins_apply_kernel folds that table in
// when the genome's flag is up.
// words from one staging row to the next: the row's S slots + the spare one (BinRegs::push, mode 1), made ODD — lane l of a flush reads row l, the
// pushes of a word land at about the same rank in every row: with the round-5 stride S + 4 (64 words at p = 18) all of that met in ONE bank
// (SQ_LDS_BANK_CONFLICT: 83 % of the LDS pipe's active cycles, profiles/r06/bins_ab.txt section 6)
__host__ __device__ constexpr uint32_t bin_row_stride(uint32_t S) { return (S + 1u) | 1u; }
struct BinRegs {
    static constexpr bool THR = false, BINS = true, BYTES = false, QUEUED = false;
    uint32_t cnt_b, stage_b;      // LDS byte addresses of this wave's row counters [V] and staging rows [V][S]
    uint32_t S, RS, V, sub_shift; // slots per row, words from one row to the next (bin_row_stride); rows: V = bins << sub_shift (few bins: each has 2^sub_shift rows, a lane uses row
    uint32_t sub_lane;            //   lane & (2^sub_shift - 1) of its bin — 64 lanes on 2 counters would be 32-way LDS atomic conflicts)
    uint32_t bin_shift;           // register index >> bin_shift = bin
    uint32_t cap;                 // entries per list
    uint32_t *lists, *cnt;        // this genome's lists and their fill counters [bins]
    uint32_t *slab, *spill;       // this genome's full-size fallback table, and its "look there too" flag
    int algo;
    // mode (wave-uniform): 0 = careful — an entry that finds its row full is applied to the fallback table there and then (junction
    // walks, dense tiles: lanes push one at a time); 1 = the hot loops' form — no branch between the 16 pushes of a word (with one, each
    // push waits for its own returning LDS atomic: 16 round trips per word): a full row's entries land in the row's spare slot, `ovf`
    // remembers it, and the caller runs the word again in mode 2; 2 = the entries of the rows that ran full straight to the fallback table
    // (max / OR are idempotent: what was staged the first time does no harm)
    uint32_t mode;
    mutable uint32_t ovf;
    // exact, slow: one global atomic.  (Static, everything by value: a member function that is not inlined takes `this`, the struct
    // then lives in scratch memory and every push reads its fields back from there — 6.7 vector memory reads per k-mer, found with
    // SQ_INSTS_VMEM_RD)
    // (`spill` holds one flag per BIN of the genome: bins_apply_kernel folds in — and wipes — only the parts of the fallback table that were used)
    static __device__ __noinline__ void spill_to(uint32_t *slab, uint32_t *spill, int algo, uint32_t bin_shift, uint32_t e)
    {
        const uint32_t v = e & 63u, idx = e >> 6;
        if (v == 63u) return;
        if (algo == 2) (void)__hip_atomic_fetch_or(slab + 2u * idx + (v >> 5), 1u << (v & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_fetch_max(reinterpret_cast<int *>(slab) + idx, (int)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(spill + (idx >> bin_shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void spill_entry(uint32_t e) const { spill_to(slab, spill, algo, bin_shift, e); }
    __device__ __forceinline__ void push(uint32_t idx, uint32_t v) const
    {
        const uint32_t row = ((idx >> bin_shift) << sub_shift) | sub_lane, e = (idx << 6) | (v & 63u);
        if (mode == 2u) {
            // the word's second run: only the rows that ran full the first time (their counters still say so: flush has not reset them) lost
            // entries — and only THEIR bins get the "look in the fallback table" flag.  (Round 6: every entry of the word went there, 1 024
            // global atomics and a flag on EVERY bin of the genome for one full row — at p = 22 about half of all bins were flagged, and
            // bins_apply_kernel read and wiped 256 KiB of fallback table for each: profiles/r06/bins_ab.txt section 8)
            if (*(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(cnt_b + row * 4u) > S) spill_entry(e);
            return;
        }
        const uint32_t rank = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)(uintptr_t)(cnt_b + row * 4u), 1u,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (mode == 1u) {
            ovf = ovf > rank ? ovf : rank;
            *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(stage_b + (row * RS + (rank < S ? rank : S)) * 4u) = e;
            return;
        }
        if (__builtin_expect(rank < S, 1)) *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(stage_b + (row * RS + rank) * 4u) = e;
        else spill_entry(e);
    }
    // after a word pushed in mode 1: did a row run full?  (wave-uniform answer; clears the mark)
    __device__ __forceinline__ bool overflowed() const
    {
        const bool o = __builtin_amdgcn_ballot_w64(ovf >= S) != 0ull;
        ovf = 0u;
        return o;
    }
    // staged entries -> the bins' lists: a lane owns a row, reserves room for its entries in the bin's list and copies them out 16 bytes
    // at a time (scattered 4-byte stores were what bound the first version).  Round 6: inside a bin the high bits of the register index are the
    // bin's own, so a list entry needs bin_shift + 6 = 20 (UltraLogLog) or 21 (HyperLogLog) bits: SIX entries leave in one 16-byte chunk (two
    // 64-bit words of three fields each; rows are padded to six entries with "nothing" = 63), a third less list traffic in both passes wherever
    // a row holds more than a handful of entries per flush (profiles/r06/floor_bins.md).  The rows of one bin sit in neighbouring lanes and share
    // ONE returning global atomic (a prefix sum over the wave; 32 lanes adding to the same two counters was the other thing that bound it).
    // (Reserving chunks ahead — one atomic per ~8 flushes — was tried and lost: more atomics on fewer counters where bins are few,
    // and the filling of unused tails.)  The lists' counters count 32-bit WORDS (four per chunk).  Every lane of the wave must call it.
    __device__ __forceinline__ void flush(uint32_t lane) const
    {
        auto lds = [](uint32_t b) { return (__attribute__((address_space(3))) uint32_t *)(uintptr_t)b; };
        const uint32_t F = bin_shift + 6u;                                 // bits of a list entry: index inside the bin << 6 | value
        const uint32_t fmask = (1u << F) - 1u;
        // 64 rows and more (64 bins and more, a row per lane): up to four groups of 64 rows; ALL their reservations are issued before the
        // first copy waits for one (p = 22, 256 rows: four returning atomics one after the other per word of 16 k-mers were most of its pass)
        // 32 rows (p = 18 .. 20): lanes l and l + 32 share row l and copy its even and odd chunks — with a row per lane half the wave sat out
        // the copy loop, a quarter of the pass's vector instructions
        const bool twin = V == 32u;
        const uint32_t half = twin ? lane >> 5 : 0u;
        uint32_t n_[4], base_[4];
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) {
            n_[g] = 0u; base_[g] = 0u;
            if (g * 64u >= V) continue;
            const uint32_t row = twin ? (lane & 31u) : g * 64u + lane;
            uint32_t n = 0;
            if (row < V) {
                n = *lds(cnt_b + row * 4u);                                // (both lanes of a twin read in this one instruction, before the reset)
                if (n && half == 0u) *lds(cnt_b + row * 4u) = 0u;
                n = n < S ? n : S;                                         // (what went beyond the row has been spilled / redone)
            }
            const uint32_t n4 = half ? 0u : ((n + 5u) / 6u) * 4u, bin = row >> sub_shift;   // words: four per chunk of six entries (reserved by the row's first lane)
            uint32_t base;
#ifdef LASH_ABL_BINS_NO_ATOMIC  // timing-only diagnostic build: every row lands at the start of its list
            if (true) { base = 0u; (void)bin; } else
#endif
            if (sub_shift == 0u) {
                base = n4 ? __hip_atomic_fetch_add(cnt + bin, n4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            } else {
                uint32_t incl = n4;                                        // inclusive prefix sum over the wave's lanes (six DPP adds)
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);
                const uint32_t sub = (1u << sub_shift) - 1u, gs = lane & ~sub, ge = gs | sub;
                const uint32_t before = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(gs * 4u), (int)(incl - n4));     // words of the bins before this one
                const uint32_t total = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ge * 4u), (int)incl) - before;
                uint32_t bin_base = 0;
                if (lane == gs && total) bin_base = __hip_atomic_fetch_add(cnt + bin, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bin_base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(gs * 4u), (int)bin_base);
                base = bin_base + (incl - n4) - before;
            }
            if (twin) base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane & 31u) * 4u), (int)base);
            n_[g] = n; base_[g] = base;
        }
#pragma unroll
        for (uint32_t g = 0; g < 4u; ++g) {
            if (g * 64u >= V) continue;
            const uint32_t row = twin ? (lane & 31u) : g * 64u + lane, n = n_[g], base = base_[g];
            const uint32_t bin = row >> sub_shift;
            uint32_t *dst = lists + (uint64_t)bin * cap;
            const uint32_t src = stage_b + row * RS * 4u;
            const uint32_t di = twin ? 12u : 6u, dat = twin ? 8u : 4u;
            for (uint32_t i = 6u * half, at = base + 4u * half; i < n; i += di, at += dat) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                // six staged entries (rows are multiples of six slots long; their stride is ODD, so only 4-byte aligned: three ds_read2_b32)
                typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
                const u32x2 r0 = *(__attribute__((address_space(3))) u32x2_a4 *)(uintptr_t)(src + i * 4u);
                const u32x2 r1 = *(__attribute__((address_space(3))) u32x2_a4 *)(uintptr_t)(src + i * 4u + 8u);
                const u32x2 r2 = *(__attribute__((address_space(3))) u32x2_a4 *)(uintptr_t)(src + i * 4u + 16u);
                uint32_t e[6] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y};
#pragma unroll
                for (uint32_t j = 1; j < 6u; ++j) e[j] = i + j < n ? e[j] : 0xFFFFFFFFu;       // the padding (the row holds older entries there)
#ifdef LASH_ABL_BINS_NO_STORE   // timing-only diagnostic build (tools/build_variant_lib.sh): results are wrong by construction
                asm volatile("" ::"v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(at));
#else
                if (at + 4u <= cap) {
                    const uint64_t lo = (uint64_t)(e[0] & fmask) | ((uint64_t)(e[1] & fmask) << F) | ((uint64_t)(e[2] & fmask) << (2u * F));
                    const uint64_t hi = (uint64_t)(e[3] & fmask) | ((uint64_t)(e[4] & fmask) << F) | ((uint64_t)(e[5] & fmask) << (2u * F));
                    *reinterpret_cast<uint4 *>(dst + at) = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 6u; ++j) spill_entry(e[j]);                        // (the list is full: the full entries, to the fallback table)
                }
#endif
            }
        }
    }
    __device__ __forceinline__ void finish(uint32_t lane) const { flush(lane); }
    __device__ __forceinline__ uint32_t get(uint32_t) const { return 0u; }
    __device__ __forceinline__ void smax(uint32_t, uint32_t) const {}       // (never instantiated for a launch: HyperMinHash's table fits)
    __device__ __forceinline__ void bor_first(uint32_t, int, uint32_t) const {}
    __device__ __forceinline__ void bor_pair(uint32_t, uint32_t, uint32_t) const {}
    static __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};

// HyperLogLog p = 16 and UltraLogLog p = 15 .. 17 (round 4): the registers as BYTES — 2^p of them, 32 .. 128 KiB of LDS, ONE pass over
// the genome where rounds 1-3 ran 2 .. 8 bucket-partitioned passes over 32-bit / 64-bit table words (and binning pays only from 8
// bins on).  LDS has no byte atomics: the update reads the byte and, if the k-mer changes it, runs a compare-and-swap on its word.
//   HyperLogLog: the byte is rho - 1 as a signed value, -1 = empty (as in the 32-bit tables); a k-mer changes it iff its own is larger.
//   UltraLogLog: the byte is the register itself; hash4j's add — pack(unpack(r) | 1 << (nlz + p - 1)) — is the merge of r with the
//   register 4 * (nlz + p - 1) of a sketch that holds only this k-mer (its unpack() is that single bit): ull_merge_fast, lash_device.h.
struct LdsByteRegs {
    static constexpr bool THR = false, BINS = false, BYTES = true, QUEUED = false;
    int p;
    static __device__ __forceinline__ uint32_t *word(uint32_t byte_addr) { return (uint32_t *)(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(byte_addr & ~3u); }
    __device__ __forceinline__ void hll_max(uint32_t j, uint32_t raw) const        // raw = rho - 1; all ones = nothing
    {
        __attribute__((address_space(3))) uint32_t *w = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)(j & ~3u);
        const uint32_t sh = (j & 3u) * 8u;
        uint32_t cur = *w;
        while ((int)raw > (int)(int8_t)(cur >> sh)) {
            const uint32_t want = (cur & ~(0xFFu << sh)) | ((raw & 0xFFu) << sh);
            if (__hip_atomic_compare_exchange_strong(w, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
    }
    __device__ __forceinline__ void ull_add(uint32_t idx, uint32_t nlz) const       // nlz & 63 == 63: nothing
    {
        if ((nlz & 63u) == 63u) return;
        __attribute__((address_space(3))) uint32_t *w = (__attribute__((address_space(3))) uint32_t *)(uintptr_t)(idx & ~3u);
        const uint32_t sh = (idx & 3u) * 8u, single = 4u * ((nlz & 63u) + (uint32_t)p - 1u);
        uint32_t cur = *w;
        for (;;) {
            const uint32_t r = (cur >> sh) & 0xFFu, nr = ull_merge_fast(r, single);
            if (nr == r) break;
            const uint32_t want = (cur & ~(0xFFu << sh)) | (nr << sh);
            if (__hip_atomic_compare_exchange_strong(w, &cur, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
    }
    __device__ __forceinline__ uint32_t get_word(uint32_t i) const { return *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(i * 4u); }
    __device__ __forceinline__ uint32_t get(uint32_t) const { return 0u; }
    __device__ __forceinline__ void smax(uint32_t, uint32_t) const {}
    __device__ __forceinline__ void bor_first(uint32_t, int, uint32_t) const {}
    __device__ __forceinline__ void bor_pair(uint32_t, uint32_t, uint32_t) const {}
    static __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};

// The byte tables behind a filter (round 4, second form; sketch_kernel only).  The compare-and-swap update above costs a k-mer its
// own LDS round trip and its own loop — 47 instructions in a block of their own per k-mer, nothing of the next k-mer's hash issued
// meanwhile (hll p = 16: 7.1e11 k-mers/s where p = 14's ds_max runs 1.15e12).  But nine k-mers in ten leave their register as it
// is, and whether one does is a question to the register's byte alone:
//   HyperLogLog: it changes the byte iff rho - 1 > byte (signed, -1 = empty);
//   UltraLogLog: with t = byte - 4 * (nlz + p - 1) = 4 * (top - bit) + the two bits below the top:  t < 0 (a new top), t = 4, 5
//                (one below the top, that bit unset), t = 8, 10 (two below, unset); everything else leaves it alone.
// So the word loop only ASKS — ds_read_u8, a compare (HLL) / subtract, clamp, table shift (ULL) — and a k-mer that would change
// its register waits on its lane's stack (the deferring HyperMinHash launches' layout: SigQueue below) as index << 8 | rho - 1 or
// index << 6 | nlz, appended with an unconditional ds_write and a conditional pointer step.  No branch between the k-mers of a
// group of four; when some lane cannot take another group, every lane that has an entry pops one and runs the exact update (the
// compare-and-swap above).  A stale byte (another wave is updating it) only makes the filter more permissive: registers grow
// monotonically, and what the older value already represents the newer one represents or has dropped below its window.
//   "Nothing" (a masked position; the fast form's undecided k-mer): HLL's -1 never passes; ULL's nlz = -1 can, as entry value 63,
// which the update skips.
template <int ALGO>
struct LdsByteQRegs : LdsByteRegs {
    static constexpr bool QUEUED = true;
    mutable uint32_t qptr;        // per lane: LDS byte address of its next free slot
    uint32_t q_lane_b, q_lim;     // its first slot; beyond q_lim the lane cannot take another group of four
    static __device__ __forceinline__ uint32_t ld(uint32_t b) { return *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)b; }
    static __device__ __forceinline__ void st(uint32_t b, uint32_t v) { *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)b = v; }
    __device__ __forceinline__ void queue_init(uint32_t wave_base_b, uint32_t depth, uint32_t lane)
    {
        q_lane_b = wave_base_b + lane * depth * 4u;
        qptr = q_lane_b;
        q_lim = q_lane_b + 4u * (depth - 4u);
    }
    __device__ __forceinline__ void hll_max(uint32_t j, uint32_t raw) const
    {
        const int cur = *(__attribute__((address_space(3))) int8_t *)(uintptr_t)j;                 // ds_read_i8
        st(qptr, (j << 8) | (raw & 0xFFu));
        qptr += (int)raw > cur ? 4u : 0u;
    }
    __device__ __forceinline__ void ull_add(uint32_t idx, uint32_t nlz) const
    {
        const uint32_t r = *(__attribute__((address_space(3))) uint8_t *)(uintptr_t)idx;            // ds_read_u8
        int ts = (int)(r - ((nlz << 2) + 4u * (uint32_t)p - 5u));                                    // t + 1
        asm("v_med3_i32 %0, %1, 0, 15" : "=v"(ts) : "v"(ts));                                         // t < 0 -> 0;  t >= 14 -> 15
        st(qptr, (idx << 6) | (nlz & 63u));
        qptr += (0x2984u >> ts) & 4u;                                                                  // t + 1 in {0, 5, 6, 9, 11}
    }
    // (Two entries per round with their compare-and-swap chains side by side — three LDS round trips for two entries instead of
    // three each, no branch, idle lanes swapping a slot of their own stack for itself — measured the same or slower: hll p = 16
    // 5.97 -> 5.91 ms, ull p = 15 / 16 / 17 6.03 -> 6.16, 6.62 -> 6.69, 7.53 -> 7.68.)
    template <bool ALL>
    __device__ __forceinline__ void drain() const
    {
        do {
            if (qptr != q_lane_b) {
                qptr -= 4u;
                const uint32_t e = ld(qptr);
                if constexpr (ALGO == 1) LdsByteRegs::hll_max(e >> 8, e & 0xFFu);
                else LdsByteRegs::ull_add(e >> 6, e & 63u);
            }
        } while (__builtin_amdgcn_ballot_w64(ALL ? qptr != q_lane_b : qptr > q_lim) != 0ull);
    }
    // after every fourth k-mer of a word
    __device__ __forceinline__ void check() const { if (__builtin_amdgcn_ballot_w64(qptr > q_lim) != 0ull) drain<false>(); }
};

// HyperMinHash, launches that defer signatures (process_word_defer).  The filter asks one question per k-mer — can its rank still
// win its bucket? — so the table word answers it with ONE compare: the rank is stored as the largest value of the hash's rank bits
// that still passes,
//     bits 31:16   thr = 0xFFFF >> min(lz - 1, 16)        (x16, the 16 rank bits below the bucket, passes <=> x16 <= thr)
//     bits 15:0    0xFFFE - ((lz - 1 - min(lz - 1, 16)) << 10 | sig)
// under an UNSIGNED MINIMUM, 0xFFFFFFFF = empty (passes everything; every real word is smaller).  A smaller word is a larger
// (lz, sig) in the reference's order (utils.rs:395-398 -> hyperminhash add_hash), so ds_min_u32 is its max; beyond 16 leading
// zeros the threshold is 0 (x16 must be 0) and the full update decides.  get() turns the word back into lz << 10 | sig.
//   The x = low variant (layout.hmh_x_low; round 6) caps the threshold's shift at 14 instead of 16: its filter compares the rank half of
// the hash WITHOUT the final xorshift (xxh3_128_4b_hmh_rank_xlow), whose bits 3:0 — the two lowest of the 16 rank bits — are not yet
// the hash's.  A threshold of at least 3 never looks at them; ranks beyond the cap live in the low half, as they do beyond 16.
template <uint32_t CAP_>
struct LdsThrRegsT {
    uint32_t *base;
    static constexpr bool THR = true, BINS = false, BYTES = false, QUEUED = false;
    static constexpr uint32_t CAP = CAP_;
    static constexpr bool XLOW = CAP_ != 16u;
    static constexpr uint32_t REDO_BELOW = 1u << (31u - CAP_);              // the 18-bit fast form's t18 below this: rank beyond the cap, re-run exactly
    static __device__ __forceinline__ uint32_t encode(uint32_t lzm1, uint32_t sig)
    {
        const uint32_t m = lzm1 < CAP ? lzm1 : CAP;
        return ((0xFFFFu >> m) << 16) | (0xFFFEu - (((lzm1 - m) << 10) | sig));
    }
    __device__ __forceinline__ void push_rank(uint32_t bucket, uint32_t lzm1, uint32_t sig, uint32_t vm) const
    { asm volatile("ds_min_u32 %0, %1" ::"v"(bucket << 2), "v"(encode(lzm1, sig) | ~vm) : "memory"); }
    // lzm1 <= 18 (the 18-bit fast form): exact up to the cap, what lies beyond goes in as the cap — an under-estimate, and the caller
    // re-runs those (REDO_BELOW) — which makes the word three instructions (four with the cap at 14)
    __device__ __forceinline__ void push_rank_fast(uint32_t bucket, uint32_t lzm1, uint32_t sig, uint32_t vm) const
    {
        uint32_t thr = (0xFFFF0000u >> lzm1) & 0xFFFF0000u;
        if constexpr (CAP < 16u) thr |= (0xFFFF0000u >> CAP) & 0xFFFF0000u;
        asm volatile("ds_min_u32 %0, %1" ::"v"(bucket << 2), "v"((thr | (0xFFFEu - sig)) | ~vm) : "memory");
    }
    __device__ __forceinline__ uint32_t get(uint32_t i) const                 // -> the table word LdsRegs would hold
    {
        const uint32_t w = base[i];
        if (w == 0xFFFFFFFFu) return RANK_EMPTY;
        const uint32_t thr = w >> 16, t = 0xFFFEu - (w & 0xFFFFu);
        const uint32_t m = thr ? (uint32_t)__builtin_clz(thr) - 16u : 16u;
        return ((m + (t >> 10)) << 10) | (t & 0x3FFu);
    }
    static __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};
using LdsThrRegs = LdsThrRegsT<16u>;      // x = high half (the default)
using LdsThrRegsX = LdsThrRegsT<14u>;     // x = low half

enum { REGS_LDS = 0, REGS_GLOBAL = 1, REGS_BINS = 2, REGS_BYTES = 3 };

// ------------------------------------------------------------------------------------------------------------
// the three add_kmer rules.  `vm` is 0 or ~0: invalid k-mers degrade to no-ops (max with "empty", OR 0).
//
// FAST: every rule needs a 64-bit count-leading-zeros whose answer is < 32 unless the top word `t` of the
// counted value is 0 (probability 2^-32 per k-mer).  The fast form counts in the top word only
// (one v_ffbh_u32) and returns `t`; when t == 0 it has pushed a harmless under-estimate (max) or nothing (OR),
// and the caller re-runs the word with FAST = false — legal because max/OR are idempotent.
//
// XLOW is the rule's VARIANT (round 6: every unpinned register rule is a compile-time form of the same kernels, not a separate slow
// route): HyperMinHash — x (bucket, rank) is the LOW half of xxh3_128 and y (signature) the high one (layout.hmh_x_low, SURVEY App. D
// U1); HyperLogLog — the bucket is the TOP p bits of the hash and rho counts the zeros below them (layout.hll_bucket_high, U3);
// UltraLogLog has no variant.
// ------------------------------------------------------------------------------------------------------------
template <int ALGO, bool XLOW, bool MASKED, bool FAST, class Regs, bool VSH = false>
__device__ __forceinline__ uint32_t add_kmer(const Regs &regs, uint32_t c_lo, uint32_t c_hi, uint32_t vm,
                                             BitFlip bitflip, int p, uint32_t ull_sh28 = 0)
{
    constexpr bool HLL_HIGH = ALGO == 1 && XLOW;
    if constexpr (ALGO == 0) {
        // utils.rs:395-398: Sketch::add_bytes_with_seed(&(masked as u32).to_le_bytes(), seed)
        (void)c_hi;                                      // k > 16: only the low 32 bits are hashed
        if constexpr (FAST) {
            // rank from the 18 bits of x that share a word with the bucket: t18 = those bits, left-aligned, padded
            // with ones -> clz(t18) = lz - 1 when any of them is set, else 18 (an under-estimate; re-run by caller)
            uint32_t xh, sig;
            if constexpr (XLOW) xxh3_128_4b_hmh_fast_xlow(c_lo, bitflip, xh, sig);
            else xxh3_128_4b_hmh_fast(c_lo, bitflip, xh, sig);
            const uint32_t t18 = (xh << 14) | 0x3FFFu;
            if constexpr (Regs::THR) {
                regs.push_rank_fast(xh >> 18, ffbh_u32(t18), sig, MASKED ? vm : 0xFFFFFFFFu);
            } else {
                uint32_t raw = (ffbh_u32(t18) << 10) | sig;
                if constexpr (MASKED) raw |= ~vm;
                regs.smax(xh >> 18, raw);
            }
            return t18;                                  // < 0x4000 <=> all 18 rank bits were zero (LdsThrRegs: < REDO_BELOW, a rank beyond its cap)
        } else {
            uint64_t lo, hi;
            xxh3_128_4b(c_lo, bitflip, lo, hi);
            const uint64_t x = XLOW ? lo : hi, y = XLOW ? hi : lo;
            const uint32_t xh = (uint32_t)(x >> 32), xl = (uint32_t)x;
            const uint32_t bucket = xh >> 18;                                // x >> 50
            const uint32_t th = alignbit(xh, xl, 18);                        // high word of (x << 14) ^ 0x3FFF
            const uint32_t tl = (xl << 14) | 0x3FFFu;                        // low word, never 0
            const uint32_t lzm1 = clz64_nz(th, tl);                          // lz = lzm1 + 1 = 1..=51
            if constexpr (Regs::THR) {
                regs.push_rank(bucket, lzm1, (uint32_t)y & 0x3FFu, MASKED ? vm : 0xFFFFFFFFu);
            } else {
                uint32_t raw = (lzm1 << 10) | ((uint32_t)y & 0x3FFu);        // flush: + 0x400 = (lz << 10) | sig
                if constexpr (MASKED) raw |= ~vm;
                regs.smax(bucket, raw);
            }
            return th;
        }
    } else if constexpr (ALGO == 1) {
        // utils.rs:411-413: push_hash64(xxh3_64(masked.to_le_bytes(), seed)): bucket = low p bits,
        // rho = 1 + leading zeros of the remaining 64-p bits = clz64(h | (2^p - 1)) + 1; the table holds rho - 1
        const uint32_t pm = (1u << p) - 1u;
        if constexpr (FAST && !HLL_HIGH) {
            // the hash's last step is h ^= h >> 28, which cannot move the leading one of the high word: the rank comes from the
            // high word BEFORE it, and the bucket needs only the low p bits of the low word after it
            const uint64_t g = xxh3_64_8b_pre(c_lo, c_hi, bitflip);
            const uint32_t gh = (uint32_t)(g >> 32), gl = (uint32_t)g;
            const uint32_t j = (gl ^ alignbit(gh, gl, 28)) & pm;
            uint32_t raw = ffbh_u32(gh);                                        // gh == 0 -> "empty" (nothing happens; re-run by the caller)
            if constexpr (MASKED) raw |= ~vm;
            if constexpr (Regs::BINS) regs.push(j, raw);                        // (its low six bits: 63 = nothing)
            else if constexpr (Regs::BYTES) regs.hll_max(j, raw);
            else regs.smax(j, raw);
            return gh;
        }
        if constexpr (FAST && HLL_HIGH) {
            // layout.hll_bucket_high (SURVEY App. D, U3 alternative): bucket = top p bits, rho - 1 = leading zeros of the 64 - p bits
            // below them — UltraLogLog's (index, nlz) to the letter (the all-zero tail gives 64 - p under both paddings), so its fast
            // form applies: the top word of h << p without materialising the last xorshift, the index from above the xorshift's reach
            const uint64_t g = xxh3_64_8b_pre(c_lo, c_hi, bitflip);
            const uint32_t gh = (uint32_t)(g >> 32), gl = (uint32_t)g;
            const uint32_t t28 = gh >> (VSH ? ull_sh28 : (uint32_t)(28 - p));
            const uint32_t th = alignbit(gh, gl, 32 - p) ^ t28;
            uint32_t raw = ffbh_u32(th);                                        // th == 0 -> "empty" (nothing happens; re-run by the caller)
            if constexpr (MASKED) raw |= ~vm;
            const uint32_t jh = gh >> (32 - p);
            if constexpr (Regs::BINS) regs.push(jh, raw);
            else if constexpr (Regs::BYTES) regs.hll_max(jh, raw);
            else regs.smax(jh, raw);
            return th;
        }
        const uint64_t h = xxh3_64_8b(c_lo, c_hi, bitflip);
        const uint32_t hh = (uint32_t)(h >> 32), hl = (uint32_t)h;
        if constexpr (HLL_HIGH) {
            // the exact form: rho = 1 + leading zeros of the lower 64-p bits = clz64((h << p) | 2^(p-1)) + 1
            const uint32_t jh = hh >> (32 - p);
            uint32_t raw = clz64_nz(alignbit(hh, hl, 32 - p), (hl << p) | (1u << (p - 1)));
            if constexpr (MASKED) raw |= ~vm;
            if constexpr (Regs::BINS) regs.push(jh, raw);
            else if constexpr (Regs::BYTES) regs.hll_max(jh, raw);
            else regs.smax(jh, raw);
            return 1u;
        }
        const uint32_t j = hl & pm;
        uint32_t raw;
        if constexpr (FAST) raw = ffbh_u32(hh);                              // hh == 0 -> "empty"
        else raw = clz64_nz(hh, hl | pm);
        if constexpr (MASKED) raw |= ~vm;
        if constexpr (Regs::BINS) regs.push(j, raw);
        else if constexpr (Regs::BYTES) regs.hll_max(j, raw);
        else regs.smax(j, raw);
        return hh;
    } else {
        // utils.rs:427-429: UltraLogLog::add(h): idx = top p bits, nlz = leading zeros of the 64-p bits below; hash4j sets bit
        // (nlz + p - 1) of the register's prefix and packs; sequential pack(unpack(old) | bit) == pack(OR of all bits)
        // (SURVEY §7.3), so: OR the nlz values into a bitmap now, shift by p - 1 and pack at the flush
        uint32_t hh, hl, th;
        if constexpr (FAST) {
            // without materialising the hash's last step h' = g ^ (g >> 28): bits [63-p, 32-p] of h' are those bits of g xor the
            // same bits of g >> 28, i.e. of the high word shifted right by 28 - p (p <= 26); the index sits above the xorshift's reach
            const uint64_t g = xxh3_64_8b_pre(c_lo, c_hi, bitflip);
            hh = (uint32_t)(g >> 32); hl = (uint32_t)g;
            const uint32_t t28 = hh >> (VSH ? ull_sh28 : (uint32_t)(28 - p));   // (VSH: the amount sits in a vector register, KParams)
            th = alignbit(hh, hl, 32 - p) ^ t28;
            // th != 0 -> nlz = v_ffbh(th) < 32: always the pair's first word; th == 0 -> push nothing (re-run by the caller)
            if constexpr (Regs::BINS || Regs::BYTES) {
                uint32_t nlz = ffbh_u32(th);                                     // th == 0 -> all ones -> 63 = nothing
                if constexpr (MASKED) nlz |= ~vm;
                if constexpr (Regs::BINS) regs.push(hh >> (32 - p), nlz);
                else regs.ull_add(hh >> (32 - p), nlz);
                return th;
            }
            uint32_t one;
            if constexpr (MASKED) asm("v_min3_u32 %0, %1, 1, %2" : "=v"(one) : "v"(th), "v"(vm));   // th == 0 or an invalid k-mer: nothing
            else one = th < 1u ? th : 1u;
            // the register's byte address (hh >> (29 - p)) & ~7 from the same shifted word: ONE amount lives in a vector register (two
            // cost the reads kernel a register it did not have: 9 -> 13 scratch instructions, one reload per tile)
            if constexpr (VSH && std::is_same<Regs, LdsRegs>::value) regs.bor_b((t28 >> 1) & ~7u, one << (ffbh_u32(th) & 31u));
            else regs.bor_first(hh, p, one << (ffbh_u32(th) & 31u));
            return th;
        } else {
            const uint64_t h = xxh3_64_8b(c_lo, c_hi, bitflip);
            hh = (uint32_t)(h >> 32); hl = (uint32_t)h;
            th = alignbit(hh, hl, 32 - p);                                   // high word of ~(~h << p), p >= 3
            const uint32_t tl = (hl << p) | pm_of(p);
            const uint32_t nlz = clz64_nz(th, tl);                           // 0..=64-p
            if constexpr (Regs::BINS) {
                regs.push(hh >> (32 - p), MASKED ? (nlz | ~vm) : nlz);
                return th;
            }
            if constexpr (Regs::BYTES) {
                regs.ull_add(hh >> (32 - p), MASKED ? (nlz | ~vm) : nlz);
                return th;
            }
            uint32_t val = 1u << (nlz & 31u);
            if constexpr (MASKED) val &= vm;
            regs.bor_pair(hh >> (32 - p), nlz >> 5, val);
            return th;
        }
    }
}

// what a FAST form's return value must exceed for its update to stand (the callers re-run the exact form at or below it): HyperMinHash
// looks at the 18 rank bits of the bucket's word (t18 < 0x4000: all zero; threshold tables: a rank beyond their cap), the others at 32 bits
template <int ALGO, class Regs>
__host__ __device__ constexpr uint32_t z_redo()
{
    if constexpr (ALGO != 0) return 0u;
    else if constexpr (Regs::THR) return Regs::REDO_BELOW - 1u;
    else return 0x3FFFu;
}

// ------------------------------------------------------------------------------------------------------------
// which k-mer start positions of a lane's 64 are real k-mers of the reference's iterator?
// position i is valid iff i + k <= L (genome end) and no record begins in (i, i+k-1].
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t kmer_valid_mask(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t pos0, uint32_t nk, int k)
{
    // b0..b2: break bits of positions pos0 .. pos0+95 (a genome has fewer than 2^32 - 64 bases)
    const uint32_t lim = nk - pos0;                       // caller guarantees pos0 < nk
    const uint64_t kvm = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
    if ((b0 | b1 | b2) == 0u || k == 1) return kvm;
    // S(i) = OR_{d=1..k-1} B(i+d) by doubling on the 96-bit window
    uint64_t lo = (uint64_t)b0 | ((uint64_t)b1 << 32), hi = b2;
    lo = (lo >> 1) | (hi << 63);
    hi >>= 1;
    const int n = k - 1;
    int len = 1;
    while (len * 2 <= n) {
        lo |= (lo >> len) | (hi << (64 - len));
        hi |= hi >> len;
        len *= 2;
    }
    if (len < n) {
        const int s = n - len;
        lo |= (lo >> s) | (hi << (64 - s));
    }
    return kvm & ~lo;
}

// Equal-length records (rec_uniform_kernel): record starts are the multiples of RL, the genome's first record excepted.
// Bit i of the result: position pos + i starts a record, i < n_bits <= 96.
struct Brk96 { uint32_t b0, b1, b2; };
__device__ __forceinline__ Brk96 uniform_breaks(uint32_t pos, uint32_t RL, uint32_t n_bits)
{
    const uint32_t m = pos % RL;
    uint32_t o = m ? RL - m : 0u;                       // the first record start at or after pos
    if (pos == 0u && o == 0u) o = RL;                   // (a genome's first record is no barrier)
    Brk96 b{0u, 0u, 0u};
    for (; o < n_bits; o += RL) {
        const uint32_t bit = 1u << (o & 31u);
        if (o < 32u) b.b0 |= bit; else if (o < 64u) b.b1 |= bit; else b.b2 |= bit;
    }
    return b;
}

// The same mask for equal-length records, without the bitmap detour: a record start at b (a multiple of RL, b > 0) invalidates the
// k-1 windows that begin in [b-k+1, b-1].  One modulo and, per record start within reach of the lane's 64 windows (at most one
// when RL >= 64 + k), two shifts — instead of uniform_breaks' bit loop plus the doubling OR above (90 instructions per lane and tile).
__device__ __forceinline__ uint64_t uniform_valid_mask(uint32_t pos0, uint32_t RL, uint32_t nk, int k)
{
    const uint32_t lim = nk - pos0;                       // caller guarantees pos0 < nk
    const uint64_t kvm = lim >= 64 ? ~0ull : ((1ull << lim) - 1ull);
    if (k == 1) return kvm;
    const uint32_t m = pos0 % RL, reach = 64u + (uint32_t)k - 1u;
    uint64_t bad = 0;
    for (uint32_t b = RL - m; b < reach; b += RL) {       // (m == 0: the record that starts AT pos0 is no barrier for its own windows)
        const uint32_t lo = b >= (uint32_t)k - 1u ? b - ((uint32_t)k - 1u) : 0u, hi = b - 1u < 63u ? b - 1u : 63u;   // windows lo..hi span the start
        if (lo <= hi) bad |= (hi - lo >= 63u ? ~0ull : ((1ull << (hi - lo + 1u)) - 1ull)) << lo;
    }
    return kvm & ~bad;
}

// ------------------------------------------------------------------------------------------------------------
// one packed word = 16 k-mer start positions, fully unrolled: 16 independent instruction streams per lane
// ------------------------------------------------------------------------------------------------------------
struct KParams {
    BitFlip bitflip;
    uint64_t mask_gt;      // KM_GT16: low 2k bits
    uint32_t sh_lt;        // KM_LT16: 32 - 2k            (these two and mask_hi sit in VECTOR registers in the hot kernels: a VOP2
    uint32_t mask_lt;      // KM_LT16: low 2k bits          with a scalar source issues in 4.4 cycles, with two vector sources in 2.6-2.9)
    uint32_t mask_hi;      // KM_GT16: bits 63:32 of mask_gt (its low word is all ones: 2k > 32)
    uint32_t sh_gt;        // KM_GT16: 64 - 2k
    uint32_t ull_sh28 = 0; // UltraLogLog fast form: 28 - p as a VECTOR register (round 5: `v_lshrrev_b32 v, s, v` issues in 4.42 cycles,
                           // `v, v, v` in 2.75; the second shift of the rule, by 29 - p, is taken from the first's result)
    int p;
    // per-kernel constants of VOP2 instructions as vector registers (see BitFlip, lash_device.h)
    __device__ __forceinline__ void to_vector_registers()
    {
#ifdef LASH_SCALAR_CONSTS   // A/B build (tools/variants.sh)
        return;
#endif
        asm volatile("v_mov_b32 %0, %1" : "=v"(sh_lt) : "s"(sh_lt));
        asm volatile("v_mov_b32 %0, %1" : "=v"(mask_lt) : "s"(mask_lt));
        asm volatile("v_mov_b32 %0, %1" : "=v"(mask_hi) : "s"(mask_hi));
        if (p <= 26) {
            ull_sh28 = 28u - (uint32_t)p;
            asm volatile("v_mov_b32 %0, %1" : "=v"(ull_sh28) : "s"(ull_sh28));
        }
    }
};

// The canonical k-mer of start position r (0..15) of a word for 16 < k <= 32: the window's value, right-aligned, against the
// mirrored window of the reverse-complemented stream (utils.rs:493-494).  K = 0: k at run time — two funnel shifts per stream,
// one 64-bit shift, one mask.  K = k at compile time (21: BASELINE configs[2], the reference's usual HyperLogLog setting): every
// half is a field at a known place — one v_bfe_u32 while it lies inside one stream word, one v_alignbit for the low words —
// 4.4 instead of 6 instructions per k-mer in front of the 64-bit minimum.
template <int K>
__device__ __forceinline__ void canon_gt16(int r, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t r0, uint32_t r1, uint32_t r2, const KParams &kp,
                                           uint32_t &can_lo, uint32_t &can_hi)
{
    uint32_t f_lo, f_hi, q_lo, q_hi;
    if constexpr (K == 0) {
        const uint32_t fh = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
        const uint32_t fl = r ? alignbit(c1, c2, 32 - 2 * r) : c1;
        const uint64_t fwd = (((uint64_t)fh << 32) | fl) >> kp.sh_gt;
        f_lo = (uint32_t)fwd; f_hi = (uint32_t)(fwd >> 32);
        q_lo = r ? alignbit(r1, r0, 2 * r) : r0;
        q_hi = (r ? alignbit(r2, r1, 2 * r) : r1) & kp.mask_hi;                // (the mask's low word is all ones)
    } else {
        static_assert(K > 16 && K < 32, "compile-time k of the 64-bit window");
        constexpr int D = 2 * K - 32;                                          // bits of the k-mer above its low word
        const int s = 2 * r + D;                                               // stream bit where its low word starts
        f_lo = s < 32 ? alignbit(c0, c1, 32 - s) : s == 32 ? c1 : alignbit(c1, c2, 64 - s);
        f_hi = s <= 32 ? (uint32_t)__builtin_amdgcn_ubfe(c0, 32 - s, D) : alignbit(c0, c1, 32 - 2 * r) >> (32 - D);
        q_lo = r ? alignbit(r1, r0, 2 * r) : r0;
        q_hi = s <= 32 ? (uint32_t)__builtin_amdgcn_ubfe(r1, 2 * r, D) : alignbit(r2, r1, 2 * r) & ((1u << D) - 1u);
    }
    min_u64(f_lo, f_hi, q_lo, q_hi, can_lo, can_hi);                           // km.min(rc), utils.rs:494
}

// layout.kmer_lsb_first (SURVEY App. D, U5 alternative: a k-mer's FIRST base in its least significant bits) needs no code here:
// the iterator's value is then the group-reversed window, and with cm = the complement mask on 2k bits
//     groups_reversed(fwd) = rc ^ cm,   its reverse complement = fwd ^ cm,
// i.e. the pair {fwd, rc} of the stream c ^ cm.  x ^ cm maps every base's code to its complement's, so the lsb-first k-mers ARE the
// msb-first k-mers under the complemented code table — layout_dev() hands the kernels that table (round 6; rounds 2-5 ran a separate
// always-masked, packed-only kernel family for it at 0.55-0.72 of the default's rate).
template <int ALGO, int KMODE, bool XLOW, bool MASKED, bool FAST, class Regs, int K = 0>
__device__ __forceinline__ uint32_t process_word(const Regs &regs, const KParams &kp, uint32_t c0, uint32_t c1,
                                                 uint32_t c2, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t kvw)
{
    uint32_t zacc = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t vm = MASKED ? (uint32_t)__builtin_amdgcn_sbfe((int)kvw, r, 1) : 0xFFFFFFFFu;   // 0 or ~0
        uint32_t can_lo, can_hi = 0;
        if constexpr (KMODE == KM_GT16) {
            canon_gt16<K>(r, c0, c1, c2, r0, r1, r2, kp, can_lo, can_hi);
        } else {
            uint32_t fwd = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
            uint32_t rc = r ? alignbit(r1, r0, 2 * r) : r0;
            if constexpr (KMODE == KM_LT16) { fwd >>= kp.sh_lt; rc &= kp.mask_lt; }
            can_lo = fwd < rc ? fwd : rc;                                        // utils.rs:470,482
        }
        const uint32_t t = add_kmer<ALGO, XLOW, MASKED, FAST, Regs, true>(regs, can_lo, can_hi, vm, kp.bitflip, kp.p, kp.ull_sh28);
        zacc = zacc < t ? zacc : t;
        if constexpr (Regs::QUEUED) { if ((r & 3) == 3) regs.check(); }        // (LdsByteQRegs: is some lane's stack full?)
    }
    return zacc;
}

// A QUARTER of a word: the four k-mer start positions r0 .. r0 + 3 of it, r0 = 0, 4, 8, 12 at RUN time (round 5, sole_kernels.hip).
// The last words of a small genome are few: handed out a quarter per lane they cost a dependent chain of 4 k-mers instead of 16.
// Same canonical k-mers, same rule: v_alignbit takes its shift from a register as readily as from an immediate (r = 0 is a select,
// as in process_word); bit r of kvw says whether position r is a k-mer.
template <int ALGO, int KMODE, bool XLOW, bool FAST, class Regs>
__device__ __forceinline__ uint32_t process_quarter(const Regs &regs, const KParams &kp, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t r0, uint32_t r1,
                                                    uint32_t r2, uint32_t kvw, uint32_t rq)
{
    uint32_t zacc = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (int)rq + j;
        const uint32_t vm = (uint32_t)__builtin_amdgcn_sbfe((int)kvw, r, 1);    // 0 or ~0
        uint32_t can_lo, can_hi = 0;
        if constexpr (KMODE == KM_GT16) {
            canon_gt16<0>(r, c0, c1, c2, r0, r1, r2, kp, can_lo, can_hi);
        } else {
            uint32_t fwd = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
            uint32_t rc = r ? alignbit(r1, r0, 2 * r) : r0;
            if constexpr (KMODE == KM_LT16) { fwd >>= kp.sh_lt; rc &= kp.mask_lt; }
            can_lo = fwd < rc ? fwd : rc;
        }
        const uint32_t t = add_kmer<ALGO, XLOW, true, FAST, Regs, true>(regs, can_lo, can_hi, vm, kp.bitflip, kp.p, kp.ull_sh28);
        zacc = zacc < t ? zacc : t;
    }
    return zacc;
}

// ------------------------------------------------------------------------------------------------------------
// HyperMinHash with the signature deferred.  A k-mer changes its bucket only if its rank is at least the bucket's current one —
// one k-mer in 35 of a 5 Mbp genome — and the rank needs only the x half of the hash.  The filter computes that half, reads the
// bucket's word back (LdsThrRegs: the word IS the threshold; a stale value is a larger one, the test only gets more permissive),
// and the k-mers that pass wait for the full update in LDS.
//
// Round 4: every LANE keeps its own little stack (depth a.sigq_depth, an odd number of words: the lanes' slots then fall into
// distinct banks).  Appending is three vector instructions and no scalar one:
//     ds_write_b32 ptr, k-mer          every lane, no exec mask: a k-mer that does not pass is overwritten by the next
//     v_cmp_le_u32_sdwa vcc, x16.WORD_0, word.WORD_1     (x16 = the 16 rank bits below the bucket)
//     v_cndmask_b32 inc, 0, 4, vcc ; v_add_u32 ptr, ptr, inc
// Round 3 compacted the passing lanes into one wave-wide list instead (ballot, two v_mbcnt, address, exec save / restore, scalar
// position arithmetic, a readfirstlane per word): 9 vector + 6 scalar instructions per k-mer for test and append, now 5 + 0.
// The price is the drain: when some lane could not take another group of k-mers, every lane that has one pops ONE and runs the
// full update — two thirds of the lanes on average at depth 7 (the compacted list ran them 64 at a time), nearly all at the depth
// a 1024-thread workgroup has room for.  max() is commutative and idempotent: order and repetition do not matter.
// ------------------------------------------------------------------------------------------------------------
struct SigQueue {
    uint32_t ptr;        // per lane: LDS byte address of its next free slot
    uint32_t lane_b;     // per lane: LDS byte address of its first slot
    uint32_t lim;        // per lane: lane_b + 4 * (depth - SIGQ_GROUP); beyond it the lane cannot take another group
};
#ifndef LASH_SIGQ_GROUP
#define LASH_SIGQ_GROUP 4                                   // (tools/variants.sh: groups of two measured 3 % slower)
#endif
constexpr int SIGQ_GROUP = LASH_SIGQ_GROUP;                 // k-mers between two "is a lane full?" checks
constexpr uint32_t SIGQ_MIN_DEPTH = 7;                      // 512-thread workgroups, two per CU: 64 x 7 words = 1 792 bytes per wave
static_assert(16 % SIGQ_GROUP == 0 && SIGQ_GROUP < (int)SIGQ_MIN_DEPTH, "a lane must hold a whole group of k-mers");

__device__ __forceinline__ uint32_t lds_load(uint32_t byte_addr) { return *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)byte_addr; }
__device__ __forceinline__ void lds_store(uint32_t byte_addr, uint32_t v) { *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)byte_addr = v; }

__device__ __forceinline__ void sigq_init(SigQueue &q, uint32_t wave_base_b, uint32_t depth, uint32_t lane)
{
    q.lane_b = wave_base_b + lane * depth * 4u;
    q.ptr = q.lane_b;
    q.lim = q.lane_b + 4u * (depth - (uint32_t)SIGQ_GROUP);
}
// every lane that has a k-mer waiting pops one and runs the full update on it; ALL: until every stack is empty (end of the
// item, or the staging area is needed), else until every lane can take another group
template <bool ALL, class Regs>
__device__ __forceinline__ void sigq_drain(const Regs &regs, BitFlip bitflip, int p, SigQueue &q)
{
    do {
        if (q.ptr != q.lane_b) {
            q.ptr -= 4u;
            const uint32_t c = lds_load(q.ptr);
            const uint32_t t = add_kmer<0, Regs::XLOW, false, true>(regs, c, 0u, 0xFFFFFFFFu, bitflip, p);
            if (t < Regs::REDO_BELOW) (void)add_kmer<0, Regs::XLOW, false, false>(regs, c, 0u, 0xFFFFFFFFu, bitflip, p);   // a rank beyond the threshold's cap: exact form
        }
    } while (__builtin_amdgcn_ballot_w64(ALL ? q.ptr != q.lane_b : q.ptr > q.lim) != 0ull);
}

template <int KMODE, bool MASKED, class Regs, int K = 0>
__device__ __forceinline__ void process_word_defer(const Regs &regs, const KParams &kp, uint32_t c0, uint32_t c1, uint32_t c2,
                                                   uint32_t r0, uint32_t r1, uint32_t r2, uint32_t kvw, SigQueue &q)
{
    static_assert(Regs::THR, "the filter reads thresholds");
#pragma unroll
    for (int g = 0; g < 16; g += SIGQ_GROUP) {
        // four k-mers as one straight line: four rank halves, four words read back, then four appends — no branch in between: with
        // a branch per k-mer the scheduler had one hash chain at a time, 16 % slower than not deferring at all
        uint32_t can[SIGQ_GROUP], x16[SIGQ_GROUP], cur[SIGQ_GROUP];
#pragma unroll
        for (int j = 0; j < SIGQ_GROUP; ++j) {
            const int r = g + j;
            if constexpr (KMODE == KM_GT16) {
                uint32_t can_hi;
                canon_gt16<K>(r, c0, c1, c2, r0, r1, r2, kp, can[j], can_hi);
            } else {
                uint32_t fwd = r ? alignbit(c0, c1, 32 - 2 * r) : c0;
                uint32_t rc = r ? alignbit(r1, r0, 2 * r) : r0;
                if constexpr (KMODE == KM_LT16) { fwd >>= kp.sh_lt; rc &= kp.mask_lt; }
                can[j] = fwd < rc ? fwd : rc;
            }
            const uint32_t xh = Regs::XLOW ? xxh3_128_4b_hmh_rank_xlow(can[j], kp.bitflip) : xxh3_128_4b_hmh_rank(can[j], kp.bitflip);
#ifdef LASH_ABL_DEFER_NO_READ
            cur[j] = (xh >> 16) & 0xFFFCu;
#else
            cur[j] = lds_load((xh >> 16) & 0xFFFCu);                                              // the register table starts at LDS address 0
#endif
            x16[j] = xh >> 2;                                                                     // bits 15:0 = the 16 rank bits below the bucket
        }
#pragma unroll
        for (int j = 0; j < SIGQ_GROUP; ++j) {
#ifdef LASH_ABL_DEFER_NO_APPEND   // timing-only diagnostic builds (tools/variants.sh): results are wrong by construction
            asm volatile("" ::"v"(x16[j]), "v"(cur[j]), "v"(can[j]));
#else
            lds_store(q.ptr, can[j]);
            uint32_t inc;
            asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:WORD_1\n\tv_cndmask_b32_e64 %0, 0, 4, vcc"
                : "=v"(inc) : "v"(x16[j]), "v"(cur[j]) : "vcc");
            if constexpr (MASKED) inc &= (uint32_t)__builtin_amdgcn_sbfe((int)kvw, g + j, 1);       // not a k-mer: never stays
            q.ptr += inc;
#endif
        }
#ifndef LASH_ABL_DEFER_NO_DRAIN
        if (__builtin_amdgcn_ballot_w64(q.ptr > q.lim) != 0ull) sigq_drain<false>(regs, kp.bitflip, kp.p, q);
#else
        q.ptr = q.ptr > q.lim ? q.lane_b : q.ptr;
#endif
    }
}

// ------------------------------------------------------------------------------------------------------------
// which work item a workgroup takes.  The hardware hands workgroup b to XCD b mod 8 (eight dies, each with its own workgroup slots and L2:
// nothing moves between them once dispatched) and, inside the die, to its four shader engines in turn ((b / 8) mod 4, measured with
// LASH_ITEM_TRACE: profiles/r06/dirty_2500000_trace_*.txt) — 32 static classes of workgroup slots; only the CU inside a class is chosen
// dynamically.  A launch whose items alternate between dear and cheap with a period that divides 32 — the two halves of genomes that are clean up
// to the middle and soft-masked after it; (C, C, M, M) quarters — therefore puts all the dear ones on the same dies, or on the same half of every
// die's CUs: 1 000 x 5 Mbp with 2.5 Mb lower-case blocks ran its clean halves on 128 of the 256 CUs, 3.6 ms for half the k-mers of a 3.9 ms clean
// run.  Row j of 32 workgroups therefore takes its 32 items rotated by j: a bijection inside the row (the launch's last, incomplete row stays
// as it is), three scalar instructions, and every short-period pattern of costs meets every class equally often.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcd_skewed(uint32_t b, uint32_t n)
{
    const uint32_t j = b >> 5;
    return b < (n & ~31u) ? (b & ~31u) | ((b + j) & 31u) : b;
}

// ------------------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------------------
#ifndef LASH_SKETCH_WAVES_PER_EU
#define LASH_SKETCH_WAVES_PER_EU_ATTR
#else
#define LASH_SKETCH_WAVES_PER_EU_ATTR __attribute__((amdgpu_waves_per_eu(LASH_SKETCH_WAVES_PER_EU, LASH_SKETCH_WAVES_PER_EU)))
#endif

// ---- helpers shared by the sketch kernel's flush and the finalize kernels (ull_merge_reg: lash_device.h) ----
__device__ __forceinline__ uint32_t load_u32_any(const uint8_t *p)
{
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) return *reinterpret_cast<const uint32_t *>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ void store_u32_any(uint8_t *p, uint32_t v)
{
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) { *reinterpret_cast<uint32_t *>(p) = v; return; }
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

template <int ALGO>
__device__ __forceinline__ uint32_t merge_word(uint32_t a, uint32_t b)
{
    if constexpr (ALGO == 0) {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));   // two u16 registers per word: one packed maximum (v_pk_max_u16; the
        const u16x2 m = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));   // mask / compare form: two SDWA maxima and a shift-or)
        return __builtin_bit_cast(uint32_t, m);
    } else {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = (a >> (8 * i)) & 0xFFu, y = (b >> (8 * i)) & 0xFFu;
            const uint32_t m = ALGO == 1 ? (x > y ? x : y) : ull_merge_reg(x, y);
            o |= m << (8 * i);
        }
        return o;
    }
}

// Image header by template (lash_layout header codes; defaults: HMH none, HLL "azspl" = bincode of streaming_algorithms'
// alpha f64, zero u64, sum f64, p u8 + the Box<[u8]> length prefix, ULL "l" = bincode Vec<u8> length prefix).
__device__ __forceinline__ void put_le(uint8_t *dst, uint64_t v, int n) { for (int b = 0; b < n; ++b) dst[b] = (uint8_t)(v >> (8 * b)); }
inline __device__ __noinline__ void write_header(uint8_t *img, uint64_t tpl, uint64_t alpha_bits, uint64_t n_regs, uint64_t zero,
                                          double sum, int p)
{
    uint32_t at = 0;
    for (; tpl & 0xFFu; tpl >>= 8) {
        switch ((uint32_t)(tpl & 0xFFu)) {
        case 'a': put_le(img + at, alpha_bits, 8); at += 8; break;
        case 'z': put_le(img + at, zero, 8); at += 8; break;
        case 'Z': put_le(img + at, zero, 4); at += 4; break;
        case 's': put_le(img + at, (uint64_t)__double_as_longlong(sum), 8); at += 8; break;
        case 'p': img[at] = (uint8_t)p; at += 1; break;
        case 'P': put_le(img + at, (uint64_t)p, 4); at += 4; break;
        case 'Q': put_le(img + at, (uint64_t)p, 8); at += 8; break;
        case 'l': put_le(img + at, n_regs, 8); at += 8; break;
        case 'L': put_le(img + at, n_regs, 4); at += 4; break;
        default: break;
        }
    }
}
__device__ __forceinline__ uint32_t wave_sum_dpp(uint32_t v)      // every lane in, the total in lane 63 (full exec mask)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
// The header's histogram of an image being written: every register's rho counted in LDS.  One atomic per register put thousands of
// them on the two or three words of the common ranks (hist[1], hist[2], ...): 4 096 registers of a 10 kbp genome took 7 us, a quarter
// of its workgroup's time.  Ranks below 8 are tallied per lane in eight 8-bit fields first and leave as one atomic per wave and rank.
struct HllTally {
    uint64_t packed = 0;           // eight counts of 8 bits: rho 0 .. 7
    uint32_t words = 0;            // words tallied since the last flush (4 registers each: 63 words cannot overflow a field)
    __device__ __forceinline__ void add(uint32_t *hist, uint32_t v)          // v: four registers
    {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t rho = (v >> (8 * b)) & 0xFFu;
            if (rho < 8u) packed += 1ull << (8u * rho);
            else atomicAdd(&hist[rho < 71u ? rho : 71u], 1u);
        }
        if (++words == 63u) flush(hist);
    }
    __device__ __forceinline__ void flush(uint32_t *hist)                    // (lanes may arrive here at different times: no wave-wide step inside a branch)
    {
        if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {                     // the whole wave is here: one atomic per rank
#pragma unroll
            for (uint32_t r = 0; r < 8u; ++r) {
                const uint32_t tot = wave_sum_dpp((uint32_t)(packed >> (8u * r)) & 0xFFu);
                if ((threadIdx.x & 63u) == 63u && tot) atomicAdd(&hist[r], tot);
            }
        } else {
#pragma unroll
            for (uint32_t r = 0; r < 8u; ++r) {
                const uint32_t c = (uint32_t)(packed >> (8u * r)) & 0xFFu;
                if (c) atomicAdd(&hist[r], c);
            }
        }
        packed = 0; words = 0;
    }
};
// HLL: zero and sum are recomputed from the final registers' histogram.  The reference keeps `sum` incrementally
// (sum -= 2^-old; sum += 2^-new per k-mer).  While every register is <= 53 - p each of those
// updates is exact — sum < 2^p after the first one and every term is a multiple of 2^(p-53), so nothing ever needs more
// than 53 bits — and the incremental value IS sum_j 2^-m[j], which the histogram gives exactly in any order.  A register
// above 53 - p (one k-mer in 2^(52-p)) brings terms below the grid: the reference's value then depends on the order of
// its roundings, the one here is the correctly rounded exact sum; `corner` reports the genome (lash_ctx_hll_inexact_sums).
__device__ __forceinline__ void write_hll_header(uint8_t *img, uint64_t tpl, const uint32_t *hist, uint64_t alpha_bits, int p,
                                                 uint32_t *corner);
// The same header by the 64 lanes of a workgroup's FIRST wave (every lane must call it).  One thread walking the 72 ranks — load,
// branch, convert, multiply, add, each waiting for the last — held its workgroup's slot for ~5 us: 18 % of a 10 kbp genome's time.
// While no rank lies above 53 - p every term hist[r] * 2^-r is a multiple of 2^-(53-p) and the sum is an integer below 2^54 in those
// units: lane r shifts its count into place, three 22-bit limbs go through a DPP wave sum each, lane 63 converts — the same double
// the sequential loop gets, because nothing in either is rounded.  A rank above 53 - p (the `sum` corner): the sequential loop.
__device__ __forceinline__ void write_hll_header_wave(uint8_t *img, uint64_t tpl, const uint32_t *hist, uint64_t alpha_bits, int p,
                                                      uint32_t *corner)
{
    const uint32_t lane = threadIdx.x & 63u;
    const int R = 53 - p;                                                   // 37 .. 49
    const uint32_t h0 = hist[lane], h1 = lane < 8u ? hist[64u + lane] : 0u;
    const bool above = ((int)lane > R && h0 != 0u) || h1 != 0u;
    if (__builtin_amdgcn_ballot_w64(above) != 0ull) {
        if (lane == 0u) write_hll_header(img, tpl, hist, alpha_bits, p, corner);
        return;
    }
    const uint64_t v = (int)lane <= R ? (uint64_t)h0 << (uint32_t)(R - (int)lane) : 0ull;   // < 2^(p + 53 - p) = 2^53 (+ the others: < 2^54)
    const uint32_t a = wave_sum_dpp((uint32_t)v & 0x3FFFFFu), b = wave_sum_dpp((uint32_t)(v >> 22) & 0x3FFFFFu), c = wave_sum_dpp((uint32_t)(v >> 44));
    if (lane == 63u) {
        const uint64_t S = (uint64_t)a + ((uint64_t)b << 22) + ((uint64_t)c << 44);
        const double sum = (double)S * __longlong_as_double((long long)(1023 - R) << 52);      // S <= 2^53: exact; the scaling too
        write_header(img, tpl, alpha_bits, 1ull << p, hist[0], sum, p);
    }
}
__device__ __forceinline__ void write_hll_header(uint8_t *img, uint64_t tpl, const uint32_t *hist, uint64_t alpha_bits, int p,
                                                 uint32_t *corner)
{
    double sum = 0.0;
    uint32_t above = 0;
    for (int r = 71; r >= 0; --r) {
        if (hist[r]) sum += (double)hist[r] * __longlong_as_double((long long)(1023 - r) << 52);
        if (r > 53 - p) above |= hist[r];
    }
    if (corner && above) *corner = 1u;
    write_header(img, tpl, alpha_bits, 1ull << p, hist[0], sum, p);
}
// HyperMinHash registers travel as native little-endian u16 pairs; images may hold them big-endian (layout.hmh_reg_be)
__device__ __forceinline__ uint32_t hmh_img_order(uint32_t v, uint32_t be) { return be ? (((v & 0x00FF00FFu) << 8) | ((v >> 8) & 0x00FF00FFu)) : v; }

// ---- direct mode: 2-bit words straight from ASCII ------------------------------------------------------------
// While a genome holds nothing but upper-case ACGT, filter_out_n (utils.rs:33-41) deletes nothing, base i IS byte i,
// and the pack stage's scan has nothing to compute: the sketch kernel can read the caller's bytes itself and save
// the 2-bit round trip through HBM.  Codes are kmerutils' 2-bit alphabet A,C,G,T -> 0,1,2,3; bytes
// outside the alphabet are deleted on the fly (junction walks, dense_tile below), and a genome with many of them is
// handed to stream_sketch_kernel in the same call (lash_api.hip) — since round 3 nothing here goes through the pack stage.
__device__ __forceinline__ uint4 load16_any(const uint8_t *p)    // any alignment (gfx950 global loads take it)
{
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
struct CodeTabs { uint32_t lo, hi; };   // LayoutDev::code_lo / code_hi (defaults 0x01000000 / 0x02000003: A,C,G,T = 0,1,2,3)
__device__ __forceinline__ uint32_t ascii4_gather(uint32_t x, uint32_t &bad, const CodeTabs ct)
{
    // The low 3 bits tell the four letters apart (A 1, C 3, T 4, G 7), so they index two 8-entry byte tables held in
    // v_perm operands: the letter that key stands for (0xFF for the keys no letter has: never equal to x, whose low
    // bits ARE the key) and its 2-bit code (the context layout's base codes: kernel arguments, same instruction count).
    const uint32_t key = x & 0x07070707u;
    bad = __builtin_amdgcn_bitop3_b32(bad, x, __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, key), 0xF6);   // bad | (x ^ letters); tables: entries 7..4 | 3..0
    const uint32_t code = __builtin_amdgcn_perm(ct.hi, ct.lo, key);
    return code * 0x40100401u;                                               // bits 31:24 = b0<<6 | b1<<4 | b2<<2 | b3 (no carries); the rest is junk
}
__device__ __forceinline__ uint32_t ascii4_to_2bit(uint32_t x, uint32_t &bad, const CodeTabs ct) { return ascii4_gather(x, bad, ct) >> 24; }
__device__ __forceinline__ uint32_t ascii16_to_word(const uint4 q, uint32_t &bad, const CodeTabs ct)
{
    // the four products' top bytes, first chunk most significant: three byte permutes (hipcc's shift / mask / or form: six)
    const uint32_t r0 = ascii4_gather(q.x, bad, ct), r1 = ascii4_gather(q.y, bad, ct), r2 = ascii4_gather(q.z, bad, ct), r3 = ascii4_gather(q.w, bad, ct);
    const uint32_t t01 = __builtin_amdgcn_perm(r0, r1, 0x07030303u), t23 = __builtin_amdgcn_perm(r2, r3, 0x03030703u);
    return __builtin_amdgcn_perm(t01, t23, 0x07060100u);
}
// tail lanes (the last <= 2 lanes of a genome, whose 96 bytes are not all inside it): bytes at or past L read as 'A' (their
// k-mers are masked).  All 96 byte loads are unconditional on a clamped index, so they are issued together: one memory
// round trip per genome tail instead of 96 (many small genomes: 41 us -> a few us per genome).
struct TailWords { uint32_t w[6]; uint32_t bad; };
inline __device__ __noinline__ TailWords ascii96_tail(const uint8_t *gseq, uint64_t o, uint64_t L, const CodeTabs ct)
{
    uint32_t d[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        uint32_t x = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint64_t pos = o + (uint64_t)(4 * i + b);
            const uint32_t ch = gseq[pos < L ? pos : L - 1];               // L >= 1: the lane is active
            x |= (pos < L ? ch : 0x41u) << (8 * b);
        }
        d[i] = x;
    }
    TailWords t;
    t.bad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) t.w[j] = ascii16_to_word(make_uint4(d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3]), t.bad, ct);
    return t;
}

// ---- direct mode, sparse dirt in place ---------------------------------------------------------------------
// filter_out_n (utils.rs:33-41) DELETES a byte that is not upper-case ACGT and joins the flanks.  Seen from a k-mer's
// first base b (a surviving byte): if the k bytes [b, b+k) all survive, the k-mer is the raw window (the fast path, with
// every window that holds a deleted byte masked out); otherwise it is b plus the next k-1 SURVIVING bytes, wherever they
// are — a "junction" k-mer.  Junction k-mers are few (at most k-1 per run of deleted bytes) and belong to the lane that
// owns b; that lane walks forward from its first junction start, skipping deleted bytes, and hashes them one by one.
// A sprinkle of IUPAC codes or an N in a read therefore costs a few microseconds of one lane.  Tiles where that would not be
// cheap — much of the tile deleted, a junction k-mer whose bases lie beyond its lane's 96 bytes — are compacted by their wave
// instead (dense_tile, below); WALK_MAX is a safety net behind those tests.
constexpr uint32_t WALK_MAX = 4096;          // bytes a junction walk may read past its lane's 64 positions

__device__ __forceinline__ uint32_t inv4(uint32_t x)            // bit j: byte j is not one of A C G T
{
    const uint32_t z = x ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x & 0x07070707u);
    const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
    return ((((nz >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__device__ __forceinline__ uint32_t inv16(const uint4 q) { return inv4(q.x) | (inv4(q.y) << 4) | (inv4(q.z) << 8) | (inv4(q.w) << 12); }
// The same in 8 instructions per word instead of 11 (stream_sketch_kernel, round 5; the kernels above keep the form their register
// allocation was tuned with — tests/test_kernel_budget.py): bit 7 of every byte says "that byte of z is not zero", and ONE multiply
// gathers the four flags (7+21, 15+14, 23+7, 31+0 = bits 28..31; every other partial product lands on a bit of its own below them)
__device__ __forceinline__ uint32_t inv4s(uint32_t x)
{
    const uint32_t z = x ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, x & 0x07070707u);
    const uint32_t nz = __builtin_amdgcn_bitop3_b32((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu, z, 0x80808080u, 0xA8);   // (a | b) & c
    return (nz * 0x00204081u) >> 28;
}
__device__ __forceinline__ uint32_t inv16s(const uint4 q) { return inv4s(q.x) | (inv4s(q.y) << 4) | (inv4s(q.z) << 8) | (inv4s(q.w) << 12); }

struct InvMask { uint64_t lo; uint32_t hi; };                  // deleted bytes among the lane's 96: own 64 + look-ahead 32
// re-reads the lane's bytes (L1 / L2 hits: they were loaded a moment ago) so that the fast path keeps no raw bytes alive
inline __device__ __noinline__ InvMask lane_inv_mask(const uint8_t *gseq, uint64_t o, uint64_t L)
{
    InvMask m{0, 0};
    if (o + 96 <= L) {
        const uint8_t *src = gseq + o;
        m.lo = (uint64_t)inv16(load16_any(src)) | ((uint64_t)inv16(load16_any(src + 16)) << 16) |
               ((uint64_t)inv16(load16_any(src + 32)) << 32) | ((uint64_t)inv16(load16_any(src + 48)) << 48);
        m.hi = inv16(load16_any(src + 64)) | (inv16(load16_any(src + 80)) << 16);
    } else {
        for (uint32_t i = 0; i < 96 && o + i < L; ++i) {         // the genome's last lanes: bytes at or past L are nobody's
            const uint32_t c = gseq[o + i];
            const bool ok = c == 0x41u || c == 0x43u || c == 0x47u || c == 0x54u;
            if (!ok) { if (i < 64) m.lo |= 1ull << i; else m.hi |= 1u << (i - 64); }
        }
    }
    return m;
}
// bit i = OR of bits i .. i+n-1 of the 96-bit value {hi, lo}, for i < 64 and 1 <= n <= 32
__device__ __forceinline__ uint64_t window_or(uint64_t lo, uint32_t hi32, int n)
{
    uint64_t hi = hi32;
    int len = 1;
    while (len * 2 <= n) {
        lo |= (lo >> len) | (hi << (64 - len));
        hi |= hi >> len;
        len *= 2;
    }
    if (len < n) {
        const int s = n - len;
        lo |= (lo >> s) | (hi << (64 - s));
    }
    return lo;
}

// The junction k-mers whose first base is one of this lane's 64 positions (set J; `starts` = the lane's surviving bytes
// from the first junction start on).  Bases are collected in order, deleted bytes skipped, a record start resets the
// window (k-mers never span records, utils.rs:457-464); every completed k-mer consumes the lowest remaining start, and
// is hashed iff that start is in J (the others are whole windows the fast path has done).  Returns how many it added.
template <int ALGO, bool XLOW, class Regs>
inline __device__ __noinline__ uint32_t junction_walk(const Regs regs, const uint8_t *gseq, uint64_t L, uint64_t pos0, uint64_t J,
                                               uint64_t starts, const uint32_t *bk, uint32_t RL, int k, BitFlip bitflip, int p,
                                               uint32_t cmask, const CodeTabs ct, uint32_t *dirty)
{
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    const uint32_t s0 = (uint32_t)__builtin_ctzll(J);
    uint64_t pos = pos0 + s0, fwd = 0;
    uint32_t have = 0, added = 0;
    while (starts && pos < L) {
        if (pos - pos0 > 64 + WALK_MAX) {                       // a long run of deleted bytes: leave the genome to the pack stage
            atomicOr(dirty, 1u);
            break;
        }
        // next 16 bytes (clamped at the genome's end), their codes, which of them survive, which start a record
        uint4 q;
        uint32_t nb = 16;
        if (pos + 16 <= L) q = load16_any(gseq + pos);
        else {
            nb = (uint32_t)(L - pos);
            uint32_t d[4] = {0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u};
            for (uint32_t i = 0; i < nb; ++i) d[i >> 2] = (d[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t)gseq[pos + i] << (8 * (i & 3)));
            q = make_uint4(d[0], d[1], d[2], d[3]);
        }
        uint32_t bad = 0;
        const uint32_t codes = ascii16_to_word(q, bad, ct);      // byte j -> bits 31-2j..30-2j
        uint32_t ok = ~inv16(q) & ((1u << nb) - 1u);
        uint32_t brk = 0;
        if (RL) brk = uniform_breaks((uint32_t)pos, RL, 16u).b0;          // equal-length records: starts are multiples of RL
        else if (bk) {
            const uint64_t two = ((uint64_t)bk[(pos >> 5) + 1] << 32) | bk[pos >> 5];
            brk = (uint32_t)(two >> (pos & 31)) & 0xFFFFu;
        }
        for (uint32_t j = 0; j < nb && starts; ++j) {
            if ((brk >> j) & 1u) {                               // a record starts at this byte
                have = 0;
                const uint64_t rel = pos + j - pos0;
                starts = rel < 64 ? starts & ~((1ull << rel) - 1ull) : 0ull;
            }
            if (!((ok >> j) & 1u)) continue;
            fwd = ((fwd << 2) | ((codes >> (30 - 2 * j)) & 3u)) & kmask;
            if (++have < (uint32_t)k || !starts) continue;
            const uint32_t st = (uint32_t)__builtin_ctzll(starts);
            starts &= starts - 1;
            if (!((J >> st) & 1ull)) continue;
            const uint64_t x = fwd << (64 - 2 * k);              // left-aligned: rcword() of each half, halves swapped
            const uint64_t rc = ((((uint64_t)rcword((uint32_t)x, cmask)) << 32) | rcword((uint32_t)(x >> 32), cmask)) & kmask;
            const uint64_t can = fwd < rc ? fwd : rc;            // km.min(km.reverse_complement()), utils.rs:470,482,494
            (void)add_kmer<ALGO, XLOW, false, false>(regs, (uint32_t)can, (uint32_t)(can >> 32), 0xFFFFFFFFu, bitflip, p);
            ++added;
        }
        pos += nb;
    }
    return added;
}

// ---- direct mode, dense dirt in place ----------------------------------------------------------------------
// A wave-tile (64 lanes x 64 bytes) with many deleted bytes — a soft-masked block, the flank of an assembly gap — is not worth
// per-lane junction walks, and lanes that own deleted bytes would idle through the hashing.  The wave COMPACTS the tile instead:
// every lane classifies its 64 bytes, a wave prefix sum gives each lane's offset among the survivors, the survivors go to a
// wave-private LDS staging area as the same 2-bit stream the pack stage would have written to HBM (plus the record-break bits
// moved to compacted positions), followed by the next k-1 surviving bases after the tile (found by the whole wave, 1 KiB per
// step, however long the deleted run is, up to DENSE_SCAN_MAX).  Then the lanes share the survivors EVENLY — ceil(words / 64)
// packed words each, 1..4 — and hash them with the same process_word as the clean path: a half-deleted tile costs half a tile.
// The k-mers of a tile are those whose FIRST base is one of the tile's surviving bytes (filter_out_n joins across deleted
// bytes, utils.rs:33-41; records still separate, utils.rs:457-464), so tiles stay independent: no carry, no look-back.
constexpr uint32_t DENSE_STAGE_CODE_WORDS = 264;      // 4096 + 31 + 16 bases at 16 per word, rounded up; reads reach word 4*63+5
constexpr uint32_t DENSE_STAGE_BRK_WORDS = 136;       // the same positions, one bit each; reads reach word 2*63+3 (+1)
constexpr uint32_t DENSE_STAGE_WORDS = DENSE_STAGE_CODE_WORDS + DENSE_STAGE_BRK_WORDS;   // 1600 bytes per wave
constexpr uint32_t DENSE_SCAN_MAX = 4u << 20;         // bytes of look-ahead scan before the genome is left to the pack stage

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    // Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows
    // 2, 3 (row_bcast:15 / :31): six DPP adds, no LDS round trips (a __shfl_up is a ds_bpermute)
    uint32_t incl = v;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);
    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    return incl - v;
}

// 16 bytes -> 2-bit codes (byte j in bits 31-2j..30-2j; garbage where deleted) and which bytes are NOT one of A C G T (bit j)
__device__ __forceinline__ uint32_t classify16(const uint4 q, const CodeTabs ct, uint32_t &inv)
{
    auto four = [&](uint32_t x, uint32_t &iv) {
        const uint32_t key = x & 0x07070707u;
        const uint32_t z = x ^ __builtin_amdgcn_perm(0x47FFFF54u, 0x43FF41FFu, key);     // zero byte <=> the letter its low bits stand for
        const uint32_t nz = ((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z;
        iv = ((((nz >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
        return (__builtin_amdgcn_perm(ct.hi, ct.lo, key) * 0x40100401u) >> 24;
    };
    uint32_t i0, i1, i2, i3;
    const uint32_t c = (four(q.x, i0) << 24) | (four(q.y, i1) << 16) | (four(q.z, i2) << 8) | four(q.w, i3);
    inv = i0 | (i1 << 4) | (i2 << 8) | (i3 << 12);
    return c;
}
// certainly no A C G T among the 16 bytes?  Those four letters have bits 5 and 3 clear; lower-case letters have bit 5 set, 'N' bit 3:
// if every byte has one of the two, nothing survives.  (A cheap sufficient test for the long runs; "false" means "look closer".)
__device__ __forceinline__ uint32_t hopeless_bits(const uint4 q)
{
    return (q.x | (q.x << 2)) & (q.y | (q.y << 2)) & (q.z | (q.z << 2)) & (q.w | (q.w << 2));
}
__device__ __forceinline__ bool hopeless16(const uint4 q) { return (hopeless_bits(q) & 0x20202020u) == 0x20202020u; }

// 16 classified bytes -> their survivors, first in bits 31:30 (`codes`: byte j in bits 31-2j..30-2j; `m`: bit j = byte j survives)
__device__ __forceinline__ uint32_t compact16(uint32_t codes, uint32_t m, uint32_t recv, uint32_t &cb)
{
    // recv: bit j = the survivor at byte j opens a record (compacted alongside into cb, LSB first)
    if (m == 0xFFFFu) { cb = recv; return codes; }
    uint32_t bits = 0, cnt = 0;
    cb = 0;
    while (m) {
        const uint32_t j = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        bits |= ((codes >> (30 - 2 * j)) & 3u) << (30 - 2 * cnt);
        cb |= ((recv >> j) & 1u) << cnt;
        ++cnt;
    }
    return bits;
}

// The same when the survivors of the 16 bytes are ONE RUN of neighbours (m = 0..01..10..0 — the edge of a soft-masked block, of a
// gap, of the wave's part; m = 0 counts): the compaction is a shift.  compact16's loop runs once per survivor for the whole wave as
// soon as one lane has a partial mask — 120 instructions for the one lane that holds a block's edge (round 5).
__device__ __forceinline__ bool is_run16(uint32_t m)
{
    const uint32_t t = m | (m - 1u);                                       // everything below the lowest survivor filled in
    return (t & (t + 1u)) == 0u;                                           // 0..01..1 (m = 0: all ones)
}
__device__ __forceinline__ uint32_t compact16_run(uint32_t codes, uint32_t m, uint32_t recv, uint32_t &cb)
{
    const uint32_t n = (uint32_t)__builtin_popcount(m);
    const uint32_t a = m ? (uint32_t)__builtin_ctz(m) : 0u;
    cb = recv >> a;                                                         // (recv holds survivors only: nothing beyond the run)
    return n ? (codes << (2u * a)) & (0xFFFFFFFFu << (32u - 2u * n)) : 0u;
}

// `n` compacted bases (`bits`, first in 31:30) and their break bits to stream position s of the wave's staging area (LDS byte address)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void lds_or(uint32_t byte_addr, uint32_t v) { asm volatile("ds_or_b32 %0, %1" ::"v"(byte_addr), "v"(v) : "memory"); }
__device__ __forceinline__ void stage_put(uint32_t stage_b, uint32_t s, uint32_t bits, uint32_t cb, uint32_t n, bool breaks)
{
    if (n == 0) return;
    const uint32_t w = s >> 4, sh = 2u * (s & 15u);
    lds_or(stage_b + 4u * w, bits >> sh);
    if (sh && (s & 15u) + n > 16u) lds_or(stage_b + 4u * w + 4u, bits << (32u - sh));
    if (breaks && cb) {
        const uint32_t bm = stage_b + 4u * DENSE_STAGE_CODE_WORDS;
        const uint32_t bw = s >> 5, bs = s & 31u;
        lds_or(bm + 4u * bw, cb << bs);
        if (bs && bs + n > 32u) lds_or(bm + 4u * bw + 4u, cb >> (32u - bs));
    }
}

template <int ALGO, int KMODE, bool XLOW, class Regs>
inline __device__ __noinline__ uint32_t dense_tile(const Regs regs, const KParams kp, const uint8_t *gseq, uint64_t L, uint64_t P0, uint64_t E,
                                            const uint32_t *bk, uint32_t RL, int k, uint32_t cmask, const CodeTabs ct, uint32_t stage_b,
                                            uint32_t *dirty, uint32_t *ndel, bool have_raw, uint4 q0, uint4 q1, uint4 q2, uint4 q3)
{
    // have_raw (per lane): q0..q3 are the lane's 64 bytes, still in registers from the tile load
    const uint32_t lane = threadIdx.x & 63u;
    // the KiB after the tile, asked for now: its round trip runs under the classification and the staging
    const bool la_fast = E + 1024 <= L;                                           // uniform
    uint4 la0 = make_uint4(0, 0, 0, 0);
    if (la_fast) la0 = load16_any(gseq + E + 16ull * lane);
    lds_u32 *const stage = (lds_u32 *)(uintptr_t)stage_b;                                    // this wave's staging area (LDS byte address)
    const bool breaks = bk != nullptr || RL != 0u;
    // ---- 1. the lane's own 64 bytes: codes, survivors, record starts ----
    const uint64_t o = P0 + 64ull * lane;
    const uint32_t own = o >= E ? 0u : (E - o >= 64 ? 64u : (uint32_t)(E - o));
    uint4 q[4] = {q0, q1, q2, q3};
    if (have_raw) {
    } else if (o + 64 <= L) {
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = load16_any(gseq + o + 16 * c);
    } else {
        uint32_t d[16];
        for (int i = 0; i < 16; ++i) d[i] = 0x4E4E4E4Eu;                         // 'N': beyond the genome nothing survives
        for (uint32_t i = 0; i < 64 && o + i < L; ++i) d[i >> 2] = (d[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t)gseq[o + i] << (8 * (i & 3)));
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = make_uint4(d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]);
    }
    uint32_t codes[4], vm[4];
    const uint64_t ownmask = own >= 64 ? ~0ull : ((1ull << own) - 1ull);
    uint64_t v64 = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t iv;
        codes[c] = classify16(q[c], ct, iv);
        vm[c] = ~iv & 0xFFFFu & (uint32_t)(ownmask >> (16 * c));
        v64 |= (uint64_t)vm[c] << (16 * c);
    }
    const uint32_t n_mine = (uint32_t)__builtin_popcountll(v64);
    uint32_t T;
    const uint32_t off = wave_excl_scan(n_mine, T);
    if (ndel) {
        // the census of deleted bytes (lash_timing::bases_last): the owned region, and the genome's last < k bytes when they start
        // beyond this tile — no k-mer starts there, so no wave will look at them
        uint32_t nd = own - n_mine;
        for (int d = 32; d > 0; d >>= 1) nd += __shfl_xor(nd, d, 64);
        if (lane == 0) {
            if (E < L && L - E < (uint64_t)k)
                for (uint64_t i = E; i < L; ++i) { const uint32_t c = gseq[i]; nd += !(c == 0x41u || c == 0x43u || c == 0x47u || c == 0x54u); }
            if (nd) atomicAdd(ndel, nd);
        }
    }
    if (T == 0u) return 0u;
    // record starts land on the first survivor at or after them: a carry through the deleted bytes (~v + starts), lanes chained
    uint64_t recv = 0;
    bool pend_tile = false;                                                       // a record starts after the tile's last survivor
    if (breaks) {
        uint64_t rb = 0;
        if (own) {
            if (RL) { const Brk96 ub = uniform_breaks((uint32_t)o, RL, 64u); rb = (uint64_t)ub.b0 | ((uint64_t)ub.b1 << 32); }
            else rb = (uint64_t)bk[o >> 5] | ((uint64_t)bk[(o >> 5) + 1] << 32);
            rb &= ownmask;
        }
        const uint64_t nv = ~v64 & ownmask;                                       // owned and deleted
        // carry out of the lane: a start after its last survivor (own < 64: positions at or above `own` absorb nothing and
        // propagate, like deleted ones)
        const uint64_t fill = nv | ~ownmask;
        const uint64_t rbd = rb & fill;                                           // starts AT a survivor need no carry (OR-ed in below)
        const bool gen = fill + rbd < rbd;                                        // carry out with no carry in
        const uint64_t G = __builtin_amdgcn_ballot_w64(gen), Z = __builtin_amdgcn_ballot_w64(n_mine == 0u);
        const uint64_t below = (1ull << lane) - 1ull;
        const uint64_t nz = ~Z & below;
        const uint64_t from = nz ? ~((1ull << (63 - __builtin_clzll(nz))) - 1ull) : ~0ull;   // lanes hs .. lane-1
        const bool pend_in = (G & below & from) != 0ull;
        recv = ((fill + rbd + (pend_in ? 1ull : 0ull)) | rb) & v64;
        const uint64_t nzall = ~Z;                                                // T > 0: some lane has survivors
        const int hs = 63 - __builtin_clzll(nzall);
        pend_tile = (G >> hs) != 0ull;
    }
    // ---- 2. staging: zero, own survivors, look-ahead ----
    {
        lds_u32 *z = stage + 4u * lane;
        z[0] = 0; z[1] = 0; z[2] = 0; z[3] = 0;
        if (lane < (DENSE_STAGE_WORDS - 256u) / 4u) { z[256] = 0; z[257] = 0; z[258] = 0; z[259] = 0; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        uint32_t s = off;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t cb;
            const uint32_t bits = compact16(codes[c], vm[c], (uint32_t)(recv >> (16 * c)) & 0xFFFFu, cb);
            const uint32_t n = (uint32_t)__builtin_popcount(vm[c]);
            stage_put(stage_b, s, bits, cb, n, breaks);
            s += n;
        }
    }
    uint32_t A = 0;                                                               // surviving bases found after the tile
    if (E < L && !pend_tile && k > 1) {
        const uint32_t want = (uint32_t)k - 1u;
        uint64_t base = E;
        for (;;) {
            if (base - E >= DENSE_SCAN_MAX) {                                    // a very long gap: the pack stage takes the genome
                if (lane == 0) atomicOr(dirty, 1u);
                return 0u;
            }
            const uint64_t at = base + 16ull * lane;
            uint4 x;
            if (base == E && la_fast) x = la0;
            else if (at + 16 <= L) x = load16_any(gseq + at);
            else {
                uint32_t d[4] = {0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu};
                for (uint32_t i = 0; i < 16 && at + i < L; ++i) d[i >> 2] = (d[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t)gseq[at + i] << (8 * (i & 3)));
                x = make_uint4(d[0], d[1], d[2], d[3]);
            }
            uint32_t iv;
            const uint32_t cw = classify16(x, ct, iv);
            uint32_t m = ~iv & 0xFFFFu;
            bool stop = base + 1024 >= L;
            uint32_t rb = 0;
            if (breaks && at < L) {
                if (RL) rb = uniform_breaks((uint32_t)at, RL, 16u).b0;
                else rb = (bk[at >> 5] >> (at & 31u)) & 0xFFFFu;                  // `at` is a multiple of 16 (E < L: E is one of 64)
            }
            const uint64_t Bm = __builtin_amdgcn_ballot_w64(rb != 0u);
            if (Bm) {                                                             // the next record: nothing from its start on
                const uint32_t fb = (uint32_t)__builtin_ctzll(Bm);
                if (lane > fb) m = 0;
                else if (lane == fb) m &= (1u << __builtin_ctz(rb)) - 1u;
                stop = true;
            }
            const uint32_t n = (uint32_t)__builtin_popcount(m);
            uint32_t tot;
            const uint32_t at_s = A + wave_excl_scan(n, tot);
            if (n && at_s < want) {
                uint32_t cb;
                const uint32_t bits = compact16(cw, m, 0u, cb);
                stage_put(stage_b, T + at_s, bits, 0u, n, false);
            }
            A += tot;
            if (A >= want || stop) break;
            base += 1024;
            if (tot != 0u) continue;
            // a whole KiB deleted: a long run (soft-masked block, assembly gap).  Skip ahead 8 KiB per round trip to the first KiB
            // that holds a survivor or a record start; the loop above takes it from there.
            for (;;) {
                if (base + 8192 + 16 > L || base - E >= DENSE_SCAN_MAX) break;    // near the genome's end: KiB by KiB (above)
                uint4 y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = load16_any(gseq + base + 1024ull * j + 16ull * lane);
                uint32_t first = 8;
#pragma unroll
                for (int j = 7; j >= 0; --j) {
                    bool hit = !hopeless16(y[j]);
                    if (breaks) {
                        const uint64_t aj = base + 1024ull * j + 16ull * lane;
                        hit = hit || (RL ? uniform_breaks((uint32_t)aj, RL, 16u).b0 != 0u : ((bk[aj >> 5] >> (aj & 31u)) & 0xFFFFu) != 0u);
                    }
                    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) first = (uint32_t)j;
                }
                base += 1024ull * first;
                if (first < 8u) break;
            }
        }
        if (A > want) A = want;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 3. the survivors, shared evenly, through the clean path's hashing ----
    const uint32_t S = T + A;
    const uint32_t nk = S >= (uint32_t)k ? (T < S - (uint32_t)k + 1u ? T : S - (uint32_t)k + 1u) : 0u;   // k-mers that start in the tile
    if (nk == 0u) return 0u;
    const uint32_t wpl = ((nk + 15u) / 16u + 63u) / 64u;                          // packed words per lane: 1..4
    const uint32_t fw = wpl * lane, pos0 = 16u * fw;
    const bool active = pos0 < nk;
    const uint32_t junk = (threadIdx.x + 1u) * 0x9E3779B1u;
    uint32_t c0 = junk, c1 = ~junk, c2 = junk, c3 = ~junk, c4 = junk, c5 = ~junk;
    uint64_t kv = 0;
    if (active) {
        c0 = stage[fw]; c1 = stage[fw + 1]; c2 = stage[fw + 2]; c3 = stage[fw + 3]; c4 = stage[fw + 4]; c5 = stage[fw + 5];
        uint32_t b0 = 0, b1 = 0, b2 = 0;
        if (breaks) {
            const lds_u32 *bm = stage + DENSE_STAGE_CODE_WORDS + (pos0 >> 5);
            const uint32_t sh = pos0 & 31u;                                       // 0 or 16
            const uint32_t w0 = bm[0], w1 = bm[1], w2 = bm[2], w3 = bm[3];
            b0 = sh ? (w0 >> 16) | (w1 << 16) : w0;
            b1 = sh ? (w1 >> 16) | (w2 << 16) : w1;
            b2 = sh ? (w2 >> 16) | (w3 << 16) : w2;
        }
        kv = kmer_valid_mask(b0, b1, b2, pos0, nk, k);
        if (wpl < 4u) kv &= (1ull << (16u * wpl)) - 1ull;
    }
    const uint32_t added = (uint32_t)__builtin_popcountll(kv);
    const uint64_t full = wpl < 4u ? (1ull << (16u * wpl)) - 1ull : ~0ull;
    const bool all_valid = __builtin_amdgcn_ballot_w64(kv != full) == 0ull;
    uint32_t r0 = rcword(c0, cmask), r1 = rcword(c1, cmask), r2 = (KMODE == KM_GT16) ? rcword(c2, cmask) : 0u;
#pragma unroll 1
    for (uint32_t wi = 0; wi < wpl; ++wi) {
        uint32_t z;
        if (all_valid) {
            z = process_word<ALGO, KMODE, XLOW, false, true>(regs, kp, c0, c1, c2, r0, r1, r2, 0u);
        } else {
            uint32_t kvw = (uint32_t)kv;
            asm volatile("" : "+v"(kvw));
            z = process_word<ALGO, KMODE, XLOW, true, true>(regs, kp, c0, c1, c2, r0, r1, r2, kvw);
        }
        constexpr uint32_t Z_REDO = z_redo<ALGO, Regs>();
        if (z <= Z_REDO) {
            uint32_t kvw = (uint32_t)kv;
            asm volatile("" : "+v"(kvw));
            (void)process_word<ALGO, KMODE, XLOW, true, false>(regs, kp, c0, c1, c2, r0, r1, r2, kvw);
        }
        c0 = c1; c1 = c2; c2 = c3; c3 = c4; c4 = c5; c5 = 0;
        r0 = r1;
        if constexpr (KMODE == KM_GT16) { r1 = r2; r2 = rcword(c2, cmask); } else { r1 = rcword(c1, cmask); }
        kv >>= 16;
    }
    return added;
}

// ------------------------------------------------------------------------------------------------------------
// end of a work item, shared by the nucleotide and the amino-acid kernel: k-mer census, then the registers leave LDS
// ------------------------------------------------------------------------------------------------------------
// sum of v over the wave's lanes, the same value in every lane's hands (six DPP adds + a readlane; cf. wave_excl_scan)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    uint32_t total;
    (void)wave_excl_scan(v, total);
    return total;
}

// this wave's BinRegs for the genome of a work item (REGS_BINS launches): counters cleared, nothing staged
template <int ALGO>
__device__ __forceinline__ BinRegs bin_regs_of(const SketchArgs &a, uint32_t genome)
{
    BinRegs r;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    r.sub_shift = a.bin_sub_shift;
    r.V = a.bins << r.sub_shift;
    r.sub_lane = lane & ((1u << r.sub_shift) - 1u);
    r.cnt_b = a.bin_lds_off + wave * a.bin_wave_bytes;
    r.stage_b = r.cnt_b + r.V * 4u;
    r.S = a.bin_S; r.RS = bin_row_stride(a.bin_S); r.bin_shift = a.bin_shift; r.algo = ALGO;
    r.mode = 0u; r.ovf = 0u;

    const uint32_t gi = genome - a.bin_genome0;
    const BinGenome bg = a.bin_genomes[gi];
    r.cap = bg.cap;
    r.lists = a.bin_lists + bg.list_off;
    r.cnt = a.bin_cnt + (uint64_t)gi * a.bins;
    r.slab = a.bin_slab + (uint64_t)gi * a.bin_slab_words;
    r.spill = a.bin_spill + (uint64_t)gi * a.bins;
    for (uint32_t b = lane; b < r.V; b += 64u) *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(r.cnt_b + b * 4u) = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return r;
}

template <int ALGO, int REGS, class Regs>
__device__ __forceinline__ void finish_item(const SketchArgs &a, const WorkItem &it, const Regs &regs, uint32_t *census, uint32_t part,
                                            uint32_t my_kmers, int p, uint32_t item)
{
    // `item`: the work item's index (its slot in partials / item_kmers / gregs) — blockIdx.x unless the launch is ordered (a.item_order)
    constexpr bool USE_LDS = REGS != REGS_GLOBAL;
    // valid k-mer census (tests compare it with the oracle's iterator count): wave reduce, LDS, one store per item
    Regs::lds_wait();
    if ((threadIdx.x & 63) == 0) census[threadIdx.x >> 6] = my_kmers;          // my_kmers: the WAVE's count (wave-uniform)
    if constexpr (!USE_LDS) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) tot += census[i];
        a.item_kmers[item] = part == 0u ? (uint32_t)tot : 0u;   // < 2^32 per slice; summed by finalize_kernel (the
                                                                      // passes of one slice count the same k-mers: once)
    }

    if constexpr (REGS == REGS_BINS) {
        regs.finish(threadIdx.x & 63u);                                    // what the wave still holds staged; the registers are bins_apply_kernel's job
        return;
    }
    // flush in image register format (u16 LE for HMH, u8 for HLL / ULL): into this item's partial sketch, or — when the
    // item is the only one of its genome (ITEM_SOLE: many small genomes) — straight into the genome's image, header
    // included, so that neither a partial nor a finalize pass is needed for it
    const bool sole = (it.slice & ITEM_SOLE) != 0u;
    uint32_t *out = reinterpret_cast<uint32_t *>(a.partials + (uint64_t)item * a.partial_stride);
    uint8_t *img = a.images + (uint64_t)it.genome * a.image_bytes;
    const uint32_t HDR = a.lay.hdr_bytes, reg_be = ALGO == 0 ? a.lay.hmh_reg_be : 0u;
    uint32_t *hist = census + 16;                                          // 72 words after the census (HLL header)
    if constexpr (ALGO == 1) {
        if (sole) {
            if (threadIdx.x < 72) hist[threadIdx.x] = 0;
            __syncthreads();
        }
    }
    HllTally tally;
    auto put = [&](uint32_t i, uint32_t v) {
        if (!sole) { out[i] = v; return; }
        uint8_t *dst = img + HDR + 4ull * i;
        if (a.accumulate) v = merge_word<ALGO>(hmh_img_order(load_u32_any(dst), reg_be), v);
        store_u32_any(dst, hmh_img_order(v, reg_be));
        if constexpr (ALGO == 1) tally.add(hist, v);
    };
    // table words -> registers of the image (see "register spaces" at the top of the file)
    auto hmh_reg = [](uint32_t raw) { return (int32_t)raw < 0 ? 0u : raw + 0x400u; };       // (lz - 1) << 10 | sig, -1 = empty -> lz << 10 | sig, 0
    auto hll_reg = [](uint32_t raw) { return raw + 1u; };                                     // rho - 1, -1 = empty -> rho, 0
    if constexpr (REGS == REGS_BYTES) {
        // byte tables: UltraLogLog bytes are the registers; HyperLogLog bytes are rho - 1 (0xFF = empty)
        for (uint32_t i = threadIdx.x; i < ((1u << p) >> 2); i += blockDim.x) {
            uint32_t w = regs.get_word(i);
            if constexpr (ALGO == 1) w = ((w & 0x7F7F7F7Fu) + 0x01010101u) ^ (w & 0x80808080u);      // + 1 in every byte, no carry across (0xFF -> 0)
            put(i, w);
        }
        if constexpr (ALGO == 1) {
            if (sole) {
                tally.flush(hist);
                __syncthreads();
                if (threadIdx.x < 64u) write_hll_header_wave(img, a.lay.hdr_tpl, hist, a.alpha_bits, p, a.hll_corner ? a.hll_corner + it.genome : nullptr);
            }
        } else {
            if (sole && threadIdx.x == 0) write_header(img, a.lay.hdr_tpl, a.alpha_bits, 1ull << p, 0, 0.0, p);
        }
        return;
    }
    if constexpr (ALGO == 0) {
        for (uint32_t i = threadIdx.x; i < HMH_M / 2; i += blockDim.x)
            put(i, hmh_reg(regs.get(2 * i)) | (hmh_reg(regs.get(2 * i + 1)) << 16));
        if (sole && threadIdx.x == 0 && HDR) write_header(img, a.lay.hdr_tpl, a.alpha_bits, HMH_M, 0, 0.0, HMH_P);
    } else if constexpr (ALGO == 1) {
        const uint32_t nw = (1u << p) >> 2;
        for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x)
            put(i, hll_reg(regs.get(4 * i)) | (hll_reg(regs.get(4 * i + 1)) << 8) | (hll_reg(regs.get(4 * i + 2)) << 16) | (hll_reg(regs.get(4 * i + 3)) << 24));
        if (sole) {
            tally.flush(hist);
            __syncthreads();
            if (threadIdx.x < 64u) write_hll_header_wave(img, a.lay.hdr_tpl, hist, a.alpha_bits, p, a.hll_corner ? a.hll_corner + it.genome : nullptr);
        }
    } else {
        const uint32_t nw = (1u << p) >> 2;
        for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t lo = regs.get(8 * i + 2 * b), hi = regs.get(8 * i + 2 * b + 1);
                uint32_t r = 0;
                if (lo | hi) {
                    // the nlz values seen -> hash4j's prefix (bit nlz + p - 1; nlz <= 64 - p); pack(): r = 4 * (index of the
                    // leading one) + the two bits below it
                    const uint64_t x = (((uint64_t)hi << 32) | lo) << (p - 1);
                    const uint32_t top = 63u - (uint32_t)__builtin_clzll(x);
                    const uint32_t below = top >= 2 ? (uint32_t)(x >> (top - 2)) & 3u : (uint32_t)(x << (2 - top)) & 3u;
                    r = (top << 2) | below;
                }
                o |= r << (8 * b);
            }
            put(i, o);
        }
        if (sole && threadIdx.x == 0) write_header(img, a.lay.hdr_tpl, a.alpha_bits, 1ull << p, 0, 0.0, p);   // switch U4
    }
}

}  // namespace lash
