// dev_fml.h -- device side of the BFC half of the FermiAssembler window pipeline: bit-plane packing of the reads,
// k-mer counting into per-window hash tables, the count histogram, error correction (bfc_ec1) and the unique-k-mer filter (max_streak).
//
// Reference behaviour: what fermi-lite's fml_correct / fml_fltuniq do for /root/reference/src/FermiAssembler.cpp:133-138 and what
// fml_count / bfc_ch_hist / kmer_correct do for /root/reference/src/BFC.cpp:262-270,315,351 (fermi-lite is an un-vendored, empty
// submodule of the reference: the behaviour is the one DESIGN.md section 8 writes down, which the parity tests hold this file to).
//
// Layout.  All reads of a batch of windows are one flat ASCII text in HBM.  k_fml_pack turns every 64 positions into five 64-bit
// words -- low bit and high bit of the base code, "is N", "quality >= q", and (k_fml_starts) "first base of a read" -- with the LATER
// position in the LOWER bit, which is how BFC holds a k-mer (last base in bit 0): the k-mer ending at any position is then a 128-bit
// funnel shift of two neighbouring words and a mask, its reverse complement a bit reversal, its validity (no N, no read boundary
// inside) and its "all bases high quality" flag two more masked shifts.  No per-read loop carries a rolling k-mer: one lane per
// POSITION, every lane busy, the text read once through the planes (5 bits per base).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct FmlSlot { unsigned long long key, cnt; };      // key + 1 (0 = empty); cnt = occurrences | high-quality occurrences << 32

struct FmlWin {                // per window, device-visible
    unsigned long long tab_off;         // first slot of the window's table
    unsigned int tab_mask;              // slots - 1 (a power of two)
    int k, min_cov, mode;
    long long read0, read1;             // reads [read0, read1)
    long long pos0;                     // first text position of the window (= offs[read0])
    unsigned int part0, part_mask;      // partitioned counting (k_fml_bin / k_fml_part): first partition of the window, partitions - 1 (a power of two)
};

struct FmlPlanes { const unsigned long long *p0, *p1, *pn, *pq, *ps; };      // index = block + 1 (block -1 is a guard: all N)

__device__ __forceinline__ int fml_nt5(int c)
{
    c &= 0xdf;          // upper case
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4;
}

__device__ __forceinline__ uint64_t fml_mix64(uint64_t h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
    return h;
}

// the k bits of a plane that end at position p (later position = lower bit)
__device__ __forceinline__ uint32_t fml_extract(const unsigned long long *pl, long long p, uint32_t mask)
{
    const long long b = p >> 6;
    const int s = 63 - (int)(p & 63);
    unsigned long long v = pl[b + 1] >> s;
    if (s) v |= pl[b] << (64 - s);
    return (uint32_t)v & mask;
}

// canonical key of a k-mer given its forward planes: bfc_kmer_hash's strand choice (the middle base decides; for odd k the strand whose
// middle base is A or C)
__device__ __forceinline__ unsigned long long fml_key(int k, uint32_t x0, uint32_t x1, uint32_t mask)
{
    const uint32_t x2 = __brev(~x0 & mask) >> (32 - k), x3 = __brev(~x1 & mask) >> (32 - k);
    const int t = k >> 1;
    const bool u = ((x1 >> t) & 1u) > ((x3 >> t) & 1u);
    return u ? ((unsigned long long)x3 << 32 | x2) : ((unsigned long long)x1 << 32 | x0);
}

__device__ __forceinline__ int fml_val(unsigned long long cnt)          // the 14-bit value bfc keeps: (occurrences - 1) capped at 255 | high capped at 63
{
    const unsigned int lo = (unsigned int)cnt, hi = (unsigned int)(cnt >> 32);
    return (int)((lo - 1 > 255u ? 255u : lo - 1) | (hi > 63u ? 63u : hi) << 8);
}

__device__ __forceinline__ int fml_get(const FmlSlot *tab, const FmlWin &w, unsigned long long key)          // bfc_ch_get: -1 = absent
{
    unsigned int i = (unsigned int)fml_mix64(key) & w.tab_mask;
    const FmlSlot *t = tab + w.tab_off;
    while (true) {
        const unsigned long long kk = t[i].key;
        if (kk == key + 1) return fml_val(t[i].cnt);
        if (kk == 0) return -1;
        i = (i + 1) & w.tab_mask;
    }
}

template <typename T>
__device__ __forceinline__ long long fml_upper(const T *a, long long n, T v)          // number of entries <= v in ascending a[0..n)
{
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------------------------- planes

static __global__ void k_fml_starts(const unsigned long long *offs, long long n_reads, unsigned long long *ps)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const unsigned long long p = offs[r];
    if (offs[r + 1] == p) return;
    atomicOr(&ps[(p >> 6) + 1], 1ULL << (63 - (int)(p & 63)));
}

// one wave per 64 positions: ballots ARE the planes
static __global__ void __launch_bounds__(256) k_fml_pack(const char *bases, const char *quals, long long total, int q,
                                                  unsigned long long *p0, unsigned long long *p1, unsigned long long *pn, unsigned long long *pq)
{
    const long long blk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blk * 64 >= total) return;
    const long long p = blk * 64 + 63 - lane;
    int c = 4, hq = 0;
    if (p < total) {
        c = fml_nt5((unsigned char)bases[p]);
        hq = quals ? ((int)(unsigned char)quals[p] - 33 >= q) : 1;
    }
    const unsigned long long b0 = __ballot(c < 4 && (c & 1)), b1 = __ballot(c < 4 && (c & 2)), bn = __ballot(c > 3), bq = __ballot(hq != 0);
    if (lane == 0) { p0[blk + 1] = b0; p1[blk + 1] = b1; pn[blk + 1] = bn; pq[blk + 1] = bq; }
}

// ---------------------------------------------------------------------------------------------------------------- counting

// fml_count: one lane per position = per k-mer ending there.  stats[0] = k-mers inserted, stats[1] = a table ran full (the host
// enlarges the tables and counts again; it also does so when the histogram that follows shows a table more than 70 % full).
static __global__ void __launch_bounds__(256) k_fml_count(FmlPlanes pl, long long total, const unsigned long long *offs, long long n_reads,
                                                   const FmlWin *wins, int n_win, FmlSlot *tab, unsigned long long *stats)
{
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = p < total;
    unsigned long long key = 0, inc = 1;
    FmlWin w;
    if (ok) {
        int lo = 0, hi = n_win;          // the window holding position p (windows tile the text: no need to find the read -- that was a
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].pos0 <= p) lo = mid; else hi = mid; }   // 20-step search per lane)
        w = wins[lo];
        const int k = w.k;
        const uint32_t mask = (uint32_t)((1ULL << k) - 1);
        ok = k > 0 && fml_extract(pl.pn, p, mask) == 0 && (fml_extract(pl.ps, p, mask) & (mask >> 1)) == 0;
        if (ok) {
            key = fml_key(k, fml_extract(pl.p0, p, mask), fml_extract(pl.p1, p, mask), mask);
            if (fml_extract(pl.pq, p, mask) == mask) inc |= 1ULL << 32;
        }
    }
    if (ok) {
        FmlSlot *t = tab + w.tab_off;
        unsigned int i = (unsigned int)fml_mix64(key) & w.tab_mask, probes = 0;
        while (true) {
            unsigned long long old = t[i].key;
            if (old == 0) old = atomicCAS(&t[i].key, 0ULL, key + 1);
            if (old == 0 || old == key + 1) { atomicAdd(&t[i].cnt, inc); break; }
            i = (i + 1) & w.tab_mask;
            if (++probes > w.tab_mask) { stats[1] = 1; break; }
        }
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&stats[0], (unsigned long long)__popcll(m));
}

// ---- fml_count in two passes, without the contended global atomics of k_fml_count.  At 30x coverage a k-mer is inserted ~25 times, and
// each insert of k_fml_count is a device-scope read-modify-write on a 16-byte slot somewhere in a table of hundreds of megabytes: the
// memory side serves ~4.3 G of them per second, whatever the table's size.  Here the k-mers are first BINNED by the high bits of their
// hash into partitions of ~8-16 K k-mers (k_fml_bin: a block counts its tile's k-mers per partition in LDS, reserves a stretch of each
// partition with ONE global atomic, and writes the 8-byte items there), then each partition is counted by one block in an LDS hash table
// (k_fml_part: LDS atomics), and every distinct k-mer is inserted into the window's table once, uncontended.  The items a full partition or a
// full LDS table cannot take go straight into the window's table (one global atomic each, as k_fml_count would): no flag, no recount.
#define FML_BIN_TILE 16384          // text positions per block of k_fml_bin
#define FML_PART_CAP 20736          // items a partition holds (1.25 x the 16 K mean at most + slack)
#define FML_PART_SLOTS 4096         // LDS table of k_fml_part
#define FML_PART_MAX 4096           // partitions per window at most (LDS counters of k_fml_bin)
#define FML_PART_PROBES 64          // LDS probes before an item of k_fml_part spills into the window's table (a full LDS table must not cost 4096 probes per item)

// one insert into a window's table (what k_fml_count does per k-mer); false = the table is full
__device__ __forceinline__ bool fml_insert(FmlSlot *t, unsigned int tab_mask, unsigned long long key, unsigned long long inc)
{
    unsigned int i = (unsigned int)fml_mix64(key) & tab_mask, probes = 0;
    while (true) {
        unsigned long long old = t[i].key;
        if (old == 0) old = atomicCAS(&t[i].key, 0ULL, key + 1);
        if (old == 0 || old == key + 1) { atomicAdd(&t[i].cnt, inc); return true; }
        i = (i + 1) & tab_mask;
        if (++probes > tab_mask) return false;
    }
}

__device__ __forceinline__ bool fml_kmer_at(const FmlPlanes &pl, long long p, const FmlWin &w, unsigned long long &key, bool &hq)
{
    const int k = w.k;
    const uint32_t mask = (uint32_t)((1ULL << k) - 1);
    if (!(k > 0 && fml_extract(pl.pn, p, mask) == 0 && (fml_extract(pl.ps, p, mask) & (mask >> 1)) == 0)) return false;
    key = fml_key(k, fml_extract(pl.p0, p, mask), fml_extract(pl.p1, p, mask), mask);
    hq = fml_extract(pl.pq, p, mask) == mask;
    return true;
}

static __global__ void __launch_bounds__(256) k_fml_bin(FmlPlanes pl, long long total, const FmlWin *wins, int n_win, unsigned int *cursor, unsigned long long *items,
                                                 FmlSlot *tab, unsigned long long *stats)
{
    __shared__ unsigned int s_cnt[FML_PART_MAX], s_base[FML_PART_MAX];
    __shared__ unsigned int s_valid;
    const long long t0 = (long long)blockIdx.x * FML_BIN_TILE;
    int lo = 0, hi = n_win;                                     // the window of the tile's first position: its partitions are counted in LDS
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].pos0 <= t0) lo = mid; else hi = mid; }
    const FmlWin w0 = wins[lo];
    const long long w0_end = lo + 1 < n_win ? wins[lo + 1].pos0 : total;
    for (unsigned int i = threadIdx.x; i <= w0.part_mask; i += 256) s_cnt[i] = 0;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    unsigned int n_valid = 0;
    for (int it = 0; it < FML_BIN_TILE / 256; ++it) {
        const long long p = t0 + it * 256 + threadIdx.x;
        if (p >= total || p >= w0_end) continue;                // (positions of later windows: second pass)
        unsigned long long key; bool hq;
        if (!fml_kmer_at(pl, p, w0, key, hq)) continue;
        ++n_valid;
        atomicAdd(&s_cnt[(unsigned int)(fml_mix64(key) >> 40) & w0.part_mask], 1u);
    }
    __syncthreads();
    for (unsigned int i = threadIdx.x; i <= w0.part_mask; i += 256) {
        const unsigned int c = s_cnt[i];
        s_base[i] = c ? atomicAdd(&cursor[w0.part0 + i], c) : 0u;
        s_cnt[i] = 0;
    }
    __syncthreads();
    for (int it = 0; it < FML_BIN_TILE / 256; ++it) {
        const long long p = t0 + it * 256 + threadIdx.x;
        if (p >= total) continue;
        if (p < w0_end) {
            unsigned long long key; bool hq;
            if (!fml_kmer_at(pl, p, w0, key, hq)) continue;
            const unsigned int lp = (unsigned int)(fml_mix64(key) >> 40) & w0.part_mask;
            const unsigned int at = s_base[lp] + atomicAdd(&s_cnt[lp], 1u);
            if (at < FML_PART_CAP) items[(unsigned long long)(w0.part0 + lp) * FML_PART_CAP + at] = key | (hq ? 1ULL << 63 : 0ULL);
            else if (!fml_insert(tab + w0.tab_off, w0.tab_mask, key, hq ? (1ULL | 1ULL << 32) : 1ULL)) stats[1] = 1;      // a full partition (one k-mer thousands of times): straight into the table
        } else {                                                // a tile that runs into the next window(s): those k-mers one global atomic each
            int l2 = lo, h2 = n_win;
            while (h2 - l2 > 1) { const int mid = (l2 + h2) >> 1; if (wins[mid].pos0 <= p) l2 = mid; else h2 = mid; }
            const FmlWin w = wins[l2];
            unsigned long long key; bool hq;
            if (!fml_kmer_at(pl, p, w, key, hq)) continue;
            ++n_valid;
            const unsigned int gp = w.part0 + ((unsigned int)(fml_mix64(key) >> 40) & w.part_mask);
            const unsigned int at = atomicAdd(&cursor[gp], 1u);
            if (at < FML_PART_CAP) items[(unsigned long long)gp * FML_PART_CAP + at] = key | (hq ? 1ULL << 63 : 0ULL);
            else if (!fml_insert(tab + w.tab_off, w.tab_mask, key, hq ? (1ULL | 1ULL << 32) : 1ULL)) stats[1] = 1;
        }
    }
    atomicAdd(&s_valid, n_valid);
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) atomicAdd(&stats[0], (unsigned long long)s_valid);
}

static __global__ void __launch_bounds__(256) k_fml_part(const unsigned int *cursor, const unsigned long long *items, const FmlWin *wins, int n_win, FmlSlot *tab,
                                                  unsigned long long *stats)
{
    __shared__ unsigned long long s_key[FML_PART_SLOTS];
    __shared__ unsigned int s_val[FML_PART_SLOTS];              // occurrences | high-quality occurrences << 16 (a partition holds < 2^16 items)
    const unsigned int gp = blockIdx.x;
    unsigned int n = cursor[gp];
    if (n == 0) return;
    if (n > FML_PART_CAP) n = FML_PART_CAP;                     // (the overflow is flagged by k_fml_bin)
    int lo = 0, hi = n_win;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].part0 <= gp) lo = mid; else hi = mid; }
    const FmlWin w = wins[lo];
    for (int i = threadIdx.x; i < FML_PART_SLOTS; i += 256) { s_key[i] = 0; s_val[i] = 0; }
    __syncthreads();
    const unsigned long long *src = items + (unsigned long long)gp * FML_PART_CAP;
    for (unsigned int i = threadIdx.x; i < n; i += 256) {
        const unsigned long long it = src[i];
        const unsigned long long key = it & ~(1ULL << 63);
        const unsigned int inc = 1u | (unsigned int)(it >> 63) << 16;
        unsigned int s = (unsigned int)(fml_mix64(key) >> 20) & (FML_PART_SLOTS - 1), probes = 0;
        while (true) {
            unsigned long long old = s_key[s];
            if (old == 0) old = atomicCAS(&s_key[s], 0ULL, key + 1);
            if (old == 0 || old == key + 1) { atomicAdd(&s_val[s], inc); break; }
            s = (s + 1) & (FML_PART_SLOTS - 1);
            if (++probes >= FML_PART_PROBES) {                     // the LDS table is (nearly) full around here -- a partition of mostly distinct k-mers: this one goes straight into the table
                if (!fml_insert(tab + w.tab_off, w.tab_mask, key, (unsigned long long)(inc & 0xffffu) | (unsigned long long)(inc >> 16) << 32)) stats[1] = 1;
                break;
            }
        }
    }
    __syncthreads();
    FmlSlot *t = tab + w.tab_off;
    for (int i = threadIdx.x; i < FML_PART_SLOTS; i += 256) {
        const unsigned long long k1 = s_key[i];
        if (!k1) continue;
        const unsigned int v = s_val[i];
        unsigned int j = (unsigned int)fml_mix64(k1 - 1) & w.tab_mask, probes = 0;
        while (true) {                                          // every k-mer lives in ONE partition: its slot is claimed once (the overflow paths may have been there first)
            unsigned long long old = t[j].key;
            if (old == 0) old = atomicCAS(&t[j].key, 0ULL, k1);
            if (old == 0 || old == k1) { atomicAdd(&t[j].cnt, (unsigned long long)(v & 0xffffu) | (unsigned long long)(v >> 16) << 32); break; }
            j = (j + 1) & w.tab_mask;
            if (++probes > w.tab_mask) { stats[1] = 1; break; }
        }
    }
}

// bfc_ch_hist: blocks of 1024 slots never straddle two windows (tables are powers of two >= 1024, laid end to end)
static __global__ void __launch_bounds__(256) k_fml_hist(const FmlSlot *tab, unsigned long long n_slots, const FmlWin *wins, int n_win, unsigned long long *hist /* n_win x 320 */)
{
    __shared__ unsigned int h[320];
    for (int i = threadIdx.x; i < 320; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned long long base = (unsigned long long)blockIdx.x * 1024;
    for (int j = 0; j < 4; ++j) {
        const unsigned long long s = base + j * 256 + threadIdx.x;
        if (s < n_slots && tab[s].key) {
            const int v = fml_val(tab[s].cnt);
            atomicAdd(&h[v & 0xff], 1u);
            atomicAdd(&h[256 + (v >> 8)], 1u);
        }
    }
    __syncthreads();
    int lo = 0, hi = n_win;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].tab_off <= base) lo = mid; else hi = mid; }
    for (int i = threadIdx.x; i < 320; i += 256)
        if (h[i]) atomicAdd(&hist[(size_t)lo * 320 + i], (unsigned long long)h[i]);
}

static __global__ void k_fml_dump(const FmlSlot *tab, FmlWin w, unsigned long long *keys, unsigned short *vals, unsigned long long cap, unsigned long long *n)
{
    const unsigned long long s = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > w.tab_mask) return;
    const FmlSlot e = tab[w.tab_off + s];
    if (!e.key) return;
    const unsigned long long at = atomicAdd(n, 1ULL);
    if (at < cap) { keys[at] = e.key - 1; vals[at] = (unsigned short)fml_val(e.cnt); }
}

// ---------------------------------------------------------------------------------------------------------------- correction

#define FML_EC_HIST 5
#define FML_EC_HIST_HIGH 2
#define FML_HEAP_CAP 112
#define FML_STACK_CAP(n) (8 * (n) + 64)          // [CHOICE] shared with the oracle: a search that pushes more states is given up

struct FmlEcOpt { int q, win_multi_ec, max_end_ext, w_ec, w_ec_high, w_absent, w_absent_high, max_heap; };

struct __align__(16) FmlHeap1 {          // echeap1_t, 48 bytes
    int tot_pen, i, k;
    int eh[FML_EC_HIST_HIGH];
    int ep[FML_EC_HIST];
    uint32_t x0, x1;          // forward planes of the last k - 1 (then k) bases; the reverse strand is derived when a key is needed
};

// Per-lane work areas of the correction kernels, INTERLEAVED across the 64 lanes of a wave: element j of a lane's array sits at
// index j * 64 + lane, so that lanes working on the same position of their reads -- the common case, every lane advances one base
// per step -- touch one contiguous stretch.  The pointers below are already offset by the lane; index with FML_L(j).
//
// Everything the walk knows about a base is ONE 32-bit word (a step of the walk is one or two coalesced loads):
//   bits 0-2 b, 3 q, 4-6 ob                    bfc's ecbase_t
//   bits 7-8  cls     the table's answer for the k-mer of the read that ENDS here, as far as bfc_ec1dir looks at it: 0 = absent or fewer than min_cov
//                     occurrences, 1 = min_cov, 2 = more, FML_CLS_ASK = no such k-mer (N, read start) or a base under it has changed since: probe
//   bit 9 solid, bit 10 high                    bfc_ec_kcov's two flags of that k-mer
//   bits 11-16 lcov, 17-22 hcov                 solid (and solid high) k-mers over this base
//   bits 23-25 alt0, bit 29 "alt0 known"        that k-mer with its LAST base replaced (bit (b - own - 1) & 3): in the table with min_cov occurrences or more?
//   bits 26-28 alt1, bit 30 "alt1 known"        ... with its FIRST base replaced: what the walk towards the read's start asks
// cls / solid / high / alt* come from k_fml_occ, a lane per POSITION, before the walks (a lane per READ) begin.
#define FML_L(j) ((size_t)(j) * 64)
#define FML_CLS_ASK 3u
#define FML_W_CLS(v) ((v) >> 7 & 3u)
#define FML_W_SOLID(v) ((v) >> 9 & 1u)
#define FML_W_HIGH(v) ((v) >> 10 & 1u)
#define FML_W_LCOV(v) ((int)((v) >> 11 & 63u))
#define FML_W_HCOV(v) ((int)((v) >> 17 & 63u))
#define FML_W_FORGET(v) (((v) | FML_CLS_ASK << 7) & ~(3u << 29))
struct FmlEcScratch {
    unsigned int *W;           // per base: see above
    struct FmlHeap1 *heap;     // heap entries 1 .. (entry 0, the top, lives in registers)
    uint2 *stack;              // x = parent, y = i | b << 16
};

__host__ __device__ inline size_t fml_scratch_bytes(int max_len)          // per LANE (a wave owns 64 times this, contiguous)
{
    size_t b = (4 * (size_t)max_len + 15) & ~(size_t)15;
    size_t h = sizeof(FmlHeap1) * FML_HEAP_CAP;
    size_t s = ((size_t)(FML_STACK_CAP(max_len) + 8) * 8 + 15) & ~(size_t)15;
    return b + h + s;
}

__device__ __forceinline__ FmlEcScratch fml_scratch_of(unsigned char *scratch, size_t lane_bytes, int max_len, size_t wave, int lane)
{
    unsigned char *p = scratch + wave * 64 * lane_bytes;
    FmlEcScratch sc;
    sc.W = (unsigned int *)p + lane; p += 64 * ((4 * (size_t)max_len + 15) & ~(size_t)15);
    sc.heap = (FmlHeap1 *)p + lane; p += 64 * sizeof(FmlHeap1) * FML_HEAP_CAP;
    sc.stack = (uint2 *)p + lane;
    return sc;
}

__device__ __forceinline__ void fml_append(int k, uint32_t &x0, uint32_t &x1, int c)
{
    const uint32_t mask = (uint32_t)((1ULL << k) - 1);
    x0 = (x0 << 1 | (uint32_t)(c & 1)) & mask;
    x1 = (x1 << 1 | (uint32_t)(c >> 1)) & mask;
}

__device__ __forceinline__ int fml_occ(const FmlSlot *tab, const FmlWin &w, uint32_t x0, uint32_t x1)
{
    const uint32_t mask = (uint32_t)((1ULL << w.k) - 1);
    return fml_get(tab, w, fml_key(w.k, x0, x1, mask));
}

// bfc_ch_get for up to N keys at once: the first slots' loads (16 bytes: key and count) leave together -- one round trip for all where a lane asking one
// after the other waits N times -- then each key's probe sequence is finished on its own.  want: bit j = key j is asked; r[j] = bfc_ch_get's value, -1 = absent.
template <int N>
__device__ __forceinline__ void fml_get_n(const FmlSlot *tab, const FmlWin &w, const unsigned long long (&key)[N], unsigned int want, int (&r)[N])
{
    typedef unsigned long long fml_slot2 __attribute__((ext_vector_type(2)));
    const fml_slot2 *t = (const fml_slot2 *)(tab + w.tab_off);
    unsigned int at[N];
    fml_slot2 s0[N];
    for (int j = 0; j < N; ++j) { at[j] = (unsigned int)fml_mix64(key[j]) & w.tab_mask; s0[j] = (want >> j & 1u) ? t[at[j]] : fml_slot2{0ULL, 0ULL}; }
    for (int j = 0; j < N; ++j) {
        r[j] = -1;
        if (!(want >> j & 1u)) continue;
        fml_slot2 sl = s0[j];
        unsigned int i = at[j];
        while (sl.x != key[j] + 1 && sl.x != 0) { i = (i + 1) & w.tab_mask; sl = t[i]; }
        if (sl.x != 0) r[j] = fml_val(sl.y);
    }
}
__device__ __forceinline__ unsigned int fml_cls_of(int r, int min_cov) { return r < 0 || (r & 0xff) < min_cov ? 0u : (r & 0xff) == min_cov ? 1u : 2u; }

// The table's answers for the k-mer that ends at every text position, one lane per POSITION: bfc_ec_kcov asks for each of them, and
// bfc_ec1dir asks again for every position its path has not changed (all of them, for a read without errors), plus -- wherever the base is
// not "fixed" (low quality, or a k-mer seen too rarely) -- for the three k-mers with another base at the path's end.  A lane that walks its
// read asks one dependent probe at a time, and a wave's 64 walks take their turns at it; asked here, by every lane of the chip at once, the
// walks find the answers beside their bases and probe the table only where the path differs from the read.
// occ[p]: bits 0-1 cls, 2 solid, 3 high, 4-6 alt0, 7-9 alt1, 10 alt0 known, 11 alt1 known (the word's fields, see above).
static __global__ void __launch_bounds__(256) k_fml_occ(FmlPlanes pl, long long total, const FmlWin *wins, int n_win, const FmlSlot *tab, unsigned short *occ)
{
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    int lo = 0, hi = n_win;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].pos0 <= p) lo = mid; else hi = mid; }
    const FmlWin w = wins[lo];
    const int k = w.k;
    unsigned int v = FML_CLS_ASK;
    if (k > 0) {
        const uint32_t mask = (uint32_t)((1ULL << k) - 1);
        if (fml_extract(pl.pn, p, mask) == 0 && (fml_extract(pl.ps, p, mask) & (mask >> 1)) == 0) {
            const uint32_t x0 = fml_extract(pl.p0, p, mask), x1 = fml_extract(pl.p1, p, mask), xq = fml_extract(pl.pq, p, mask);
            const int r = fml_get(tab, w, fml_key(k, x0, x1, mask));
            const int cnt = r & 0xff;
            const unsigned int cls = r < 0 || cnt < w.min_cov ? 0u : cnt == w.min_cov ? 1u : 2u;
            v = cls | (r >= 0 && cnt >= w.min_cov ? 4u : 0u) | (r >= 0 && (r >> 8 & 0x3f) >= w.min_cov + 1 ? 8u : 0u);
            // the other bases are asked about where bfc_ec1dir would: not at a high-quality base whose k-mer occurs more than min_cov times (lcov may still
            // make it ask there, a few bases from the ends of a read: the walk probes for itself then)
            for (int side = 0; side < 2; ++side) {
                const int at = side ? k - 1 : 0;          // the bit of the base that is replaced: the last base (walk to the end), the first (walk to the start)
                if ((xq >> at & 1u) && cls == 2u) continue;
                const int own = (int)((x1 >> at & 1u) << 1 | (x0 >> at & 1u));
                unsigned int bits = 0;
                unsigned long long key3[3];
                int r3[3];
                for (int d = 0; d < 3; ++d) {
                    const int b = (own + 1 + d) & 3;
                    const uint32_t y0 = (x0 & ~(1u << at)) | (uint32_t)(b & 1) << at, y1 = (x1 & ~(1u << at)) | (uint32_t)(b >> 1) << at;
                    key3[d] = fml_key(k, y0, y1, mask);
                }
                fml_get_n<3>(tab, w, key3, 7u, r3);          // (the three probes in flight together)
                for (int d = 0; d < 3; ++d) if (fml_cls_of(r3[d], w.min_cov) != 0u) bits |= 1u << d;
                v |= bits << (4 + 3 * side) | 1u << (10 + side);
            }
        }
    }
    occ[p] = (unsigned short)v;
}

// bfc_seq_conv of one base, with what k_fml_occ found for the k-mer that ends there
__device__ __forceinline__ unsigned int fml_word0(int ch, int qch, bool has_q, int q_min, unsigned int oc)
{
    const int c = fml_nt5(ch);
    int q = !has_q ? 1 : (qch - 33 >= q_min ? 1 : 0);
    if (c > 3) q = 0;
    return (unsigned int)(c | q << 3 | c << 4) | (oc & 0xfu) << 7 | (oc >> 4 & 0xffu) << 23;
}
__device__ __forceinline__ char fml_out_base(unsigned int v) { const int b = v & 7; return b != (int)(v >> 4 & 7) ? "acgtn"[b] : "ACGTN"[b]; }
__device__ __forceinline__ char fml_out_qual(unsigned int v) { const int ob = v >> 4 & 7; return (int)(v & 7) != ob ? (char)(34 + ob) : ((v >> 3 & 1) ? '?' : '+'); }

// a base of the read changes at (forward) position fi: the k-mers that end at fi .. fi + k - 1 are no longer the ones k_fml_occ asked about
template <class SC>
__device__ __forceinline__ void fml_set_base(const SC &sc, int fi, int nb, int k, int n)
{
    sc.W[FML_L(fi)] = (sc.W[FML_L(fi)] & ~7u) | (unsigned int)nb;
    const int to = fi + k < n ? fi + k : n;
    for (int j = fi; j < to; ++j) sc.W[FML_L(j)] = FML_W_FORGET(sc.W[FML_L(j)]);
}

// klib's ks_heapdown / ks_heapup on tot_pen (smallest on top), operation for operation -- the order in which equal penalties leave the
// heap decides between equally cheap paths -- over a heap whose entry 0 is the register `top` and whose entries 1 .. are mem[0 ..].
// Nearly always the heap holds one state (the read's own path), and then no heap traffic reaches memory at all.
// (memory entries through an address-space-1 pointer, three 16-byte vectors each: with a plain pointer the optimiser merges "top = e" / "mem[i] = e" into one
// access through a CHOSEN address, and a `top` that has an address lives in scratch memory -- every pop and push of the one-state heap a round trip)
typedef unsigned int fml_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) fml_u4 fml_g_u4;
struct FmlHeapMem {          // heap entries 1 .. in memory, entry e of a lane 64 entries after entry e - 1 (FML_L)
    fml_g_u4 *p;
    __device__ __forceinline__ FmlHeap1 get(int idx) const
    {
        const size_t at = FML_L(idx) * 3;
        const fml_u4 a = p[at], b = p[at + 1], c = p[at + 2];
        FmlHeap1 e;
        e.tot_pen = (int)a.x; e.i = (int)a.y; e.k = (int)a.z; e.eh[0] = (int)a.w;
        e.eh[1] = (int)b.x; e.ep[0] = (int)b.y; e.ep[1] = (int)b.z; e.ep[2] = (int)b.w;
        e.ep[3] = (int)c.x; e.ep[4] = (int)c.y; e.x0 = c.z; e.x1 = c.w;
        return e;
    }
    __device__ __forceinline__ void put(int idx, const FmlHeap1 &e) const
    {
        const size_t at = FML_L(idx) * 3;
        p[at] = fml_u4{(unsigned int)e.tot_pen, (unsigned int)e.i, (unsigned int)e.k, (unsigned int)e.eh[0]};
        p[at + 1] = fml_u4{(unsigned int)e.eh[1], (unsigned int)e.ep[0], (unsigned int)e.ep[1], (unsigned int)e.ep[2]};
        p[at + 2] = fml_u4{(unsigned int)e.ep[3], (unsigned int)e.ep[4], e.x0, e.x1};
    }
};
template <class HM>
struct FmlHeap {
    FmlHeap1 top;
    HM mem;
    int n;
    // (no get(i) / set(i) that pick between `top` and memory by index: a choice between the two ADDRESSES puts `top` in scratch memory)
    __device__ __forceinline__ FmlHeap1 pop()          // z = l[0]; l[0] = l[--n]; ks_heapdown(0, n, l)
    {
        const FmlHeap1 z = top;
        --n;
        if (n > 0) {
            const FmlHeap1 tmp = mem.get(n - 1);
            int i = 0, k = 0;
            while ((k = (k << 1) + 1) < n) {          // (k >= 1: the children are in memory)
                FmlHeap1 ck = mem.get(k - 1);
                if (k != n - 1) { const FmlHeap1 c2 = mem.get(k); if (ck.tot_pen > c2.tot_pen) { ++k; ck = c2; } }
                if (ck.tot_pen > tmp.tot_pen) break;
                if (i == 0) top = ck; else mem.put(i - 1, ck);
                i = k;
            }
            if (i == 0) top = tmp; else mem.put(i - 1, tmp);
        }
        return z;
    }
    __device__ __forceinline__ void push(const FmlHeap1 &r)          // l[n++] = r; ks_heapup(n, l)
    {
        int i = n++;
        while (i > 0) {
            const int k = (i - 1) >> 1;
            if (k == 0) {
                if (r.tot_pen > top.tot_pen) break;
                mem.put(i - 1, top); i = 0;
            } else {
                const FmlHeap1 pk = mem.get(k - 1);
                if (r.tot_pen > pk.tot_pen) break;
                mem.put(i - 1, pk); i = k;
            }
        }
        if (i == 0) top = r; else mem.put(i - 1, r);
    }
};

// what a walk works on: its words, its stack, the memory of its heap
struct FmlEcMem {          // k_fml_ec: everything in memory (FmlEcScratch)
    typedef FmlHeapMem HM;
    unsigned int *W;
    uint2 *stack;
    HM hm;
};
__device__ __forceinline__ FmlEcMem fml_ec_mem(const FmlEcScratch &g)
{
    FmlEcMem sc;
    sc.W = g.W; sc.stack = g.stack; sc.hm.p = (fml_g_u4 *)g.heap;
    return sc;
}

struct FmlPen { int ec, ec_high, absent, absent_high, b; };
__device__ __forceinline__ unsigned int fml_pen_pack(const FmlPen &p) { return (unsigned int)(p.ec | p.ec_high << 1 | p.absent << 2 | p.absent_high << 3 | p.b << 4); }
__device__ __forceinline__ FmlPen fml_pen_unpack(unsigned int v) { return FmlPen{(int)(v & 1), (int)(v >> 1 & 1), (int)(v >> 2 & 1), (int)(v >> 3 & 1), (int)(v >> 4 & 3)}; }

// bfc_ec1dir in search coordinates: position i is base i of the read (dir 0) or base n - 1 - i of it, complemented (dir 1); the
// corrected read is written back in place.
//
// The search is a loop of steps (pop the cheapest state, extend it by one base), and a step needs the table only where the path has left the read:
// behind a substitution, at a base k_fml_occ did not ask the other three bases for.  Such a step costs dependent probes of a table in memory, a step
// without one a load or two of the lane's own words -- and a wave's 64 walks each meet theirs at different steps: taken as they come, nearly every step of the wave waits for some
// lane's probe.  So a step comes in two forms.  fml_step<false> LOOKS at the state on top of the heap and, if extending it needs the table, leaves
// everything as it is and reports "blocked"; the wave runs such steps until every lane is blocked or done, then the blocked lanes take one
// fml_step<true> TOGETHER (their probes in flight at the same time), and so on: as many waits per wave as the busiest lane has, not as all have together.

template <class SC>
struct FmlWalk {
    FmlHeap<typename SC::HM> hp;
    int n, dir, end, n_stack, n_kept, n_failures, path, rv;          // rv: 0 = a path was found (path = its last base-changing step), < 0 = bfc_ec1dir's failures
    // what fml_lookahead asked the table: the classes (2 bits each) of the k-mers that end at la_pos0 .. la_pos0 + la_len - 1 on the path that had its last two
    // substitutions at la_s (base la_bs) and la_s1 and follows the read from la_pos0 on
    int la_pos0, la_len, la_s, la_s1, la_bs;
    unsigned long long la_cls;
};

#define FML_LA_MAX 30          // positions per lookahead (2 bits each in la_cls)
#ifndef FML_LA_CHUNK
#define FML_LA_CHUNK 3         // probes in flight at a time (ten at a time hold 80 VGPRs for the keys and slots: the kernel's peak, a wave per SIMD less)
#endif
// Behind a substitution the path's next k - 1 k-mers are not the read's: k_fml_occ knows nothing about them, and asking one per step makes k - 1 waits of every
// correction (and, in the second walk, of every base the first one changed: the read's k-mers there are new).  They are all known the moment the path gets there --
// the state's k - 1 bases and the read's bases that follow -- so the first step that needs one asks for all of them, FML_LA_CHUNK probes in flight at a time.
// Exact: a state finds its k-mer here only if its last two substitutions are the record's (positions, and the base at the last one while that lies under
// the k-mer), the one before them lay outside every k-mer of the run when it was built, and no N of the read lies under any of them (a substituted N is not in ep[]).
template <class SC>
__device__ __forceinline__ bool fml_lookahead(const FmlSlot *tab, const FmlWin &w, const SC &sc, FmlWalk<SC> &S, const FmlHeap1 &zt)
{
    const int k = w.k, n = S.n, dir = S.dir, pos0 = zt.i, s = zt.ep[0];
    const uint32_t mask = (uint32_t)((1ULL << k) - 1);
    if (zt.ep[1] >= 0 && zt.ep[1] > pos0 - k) return false;
    for (int q = pos0 - k + 1 > 0 ? pos0 - k + 1 : 0; q < pos0; ++q)
        if ((sc.W[FML_L(dir ? n - 1 - q : q)] >> 4 & 7u) == 4u) return false;
    int len = s > pos0 - k ? s + k - pos0 : k;
    if (len > n - pos0) len = n - pos0;
    if (len > FML_LA_MAX) len = FML_LA_MAX;
    uint32_t x0 = zt.x0, x1 = zt.x1;
    unsigned long long classes = 0;
    int got = 0;
    bool open = true;
    for (int c = 0; c < len && open; c += FML_LA_CHUNK) {
        unsigned long long key[FML_LA_CHUNK];
        int r[FML_LA_CHUNK];
        unsigned int want = 0;
        for (int j = 0; j < FML_LA_CHUNK; ++j) {
            key[j] = 0;
            if (!open || c + j >= len) continue;
            const unsigned int v = sc.W[FML_L(dir ? n - 1 - (pos0 + c + j) : pos0 + c + j)];
            int b = (int)(v & 7u);
            if (b > 3 || (v >> 4 & 7u) == 4u) { open = false; continue; }          // (an N of the read, substituted or not: the run ends before it)
            if (dir) b = 3 - b;
            fml_append(k, x0, x1, b);
            key[j] = fml_key(k, x0, x1, mask);
            want |= 1u << j;
        }
        fml_get_n<FML_LA_CHUNK>(tab, w, key, want, r);
        for (int j = 0; j < FML_LA_CHUNK; ++j)
            if (want >> j & 1u) { classes |= (unsigned long long)fml_cls_of(r[j], w.min_cov) << (2 * (c + j)); ++got; }
    }
    if (got == 0) return false;
    S.la_pos0 = pos0; S.la_len = got; S.la_s = s; S.la_s1 = zt.ep[1]; S.la_cls = classes;
    S.la_bs = s > pos0 - k && s >= 0 ? (int)((zt.x1 >> (pos0 - 1 - s) & 1u) << 1 | (zt.x0 >> (pos0 - 1 - s) & 1u)) : -1;
    return true;
}


// 0 = stepped, 1 = blocked (only fml_step<false>; nothing has changed), 2 = the search is over (S.rv, S.path)
template <bool SLOW, class SC>
__device__ __forceinline__ int fml_step(const FmlSlot *tab, const FmlWin &w, const FmlEcOpt &o, const SC &sc, FmlWalk<SC> &S)
{
    const int k = w.k, n = S.n, dir = S.dir;
    if (S.hp.n == 0) { S.rv = -2; return 2; }
    const FmlHeap1 zt = S.hp.top;          // (the state pop() will return)
    bool stop = zt.i - S.end > o.max_end_ext;
    const bool have = zt.i < n;
    int cb = 4, cq = 0, cob = 4, lc = 0, hc = 0;
    unsigned int cls = 0;          // of the path's k-mer with the read's base at i (stays 0 = "absent" where there is none)
    bool fixed = false, alt_known = false;
    unsigned int alt_ok = 0;          // bit b: the path's k-mer with base b at i is in the table min_cov times or more
    if (!stop) {
        // m: the word that holds k_fml_occ's answers for the READ's k-mer that ends at search position i (dir 1: it ends k - 1 bases further on in the read).
        // They are the path's while the path is the read there: no substitution within the k-mer's first k - 1 bases (ep[0]; a substituted N is not in
        // ep[], but no k-mer of the read lies over an N: FML_CLS_ASK, nothing known) and, for the k-mer with the read's own base, that base at its end
        unsigned int m = FML_CLS_ASK << 7;
        if (have) {
            const unsigned int v = sc.W[FML_L(dir ? n - 1 - zt.i : zt.i)];
            cb = v & 7; cq = v >> 3 & 1; cob = v >> 4 & 7;
            if (dir) { cb = cb < 4 ? 3 - cb : 4; cob = cob < 4 ? 3 - cob : 4; }
            lc = FML_W_LCOV(v); hc = FML_W_HCOV(v);
            m = dir ? (unsigned int)sc.W[FML_L(n - zt.i + k - 2)] : v;
            if (zt.ep[0] > zt.i - k) m = FML_CLS_ASK << 7;
        }
        if (have && cb < 4) {
            cls = FML_W_CLS(m);
            if (cls == FML_CLS_ASK) {
                auto in_record = [&]() -> bool {
                    if (!((unsigned int)(zt.i - S.la_pos0) < (unsigned int)S.la_len && zt.ep[0] == S.la_s && zt.ep[1] == S.la_s1)) return false;
                    if (S.la_s <= zt.i - k || S.la_s < 0) return true;
                    const int sh = zt.i - 1 - S.la_s;
                    return (int)((zt.x1 >> sh & 1u) << 1 | (zt.x0 >> sh & 1u)) == S.la_bs;
                };
                bool hit = in_record();
                if (!hit) {
                    if (!SLOW) return 1;
                    hit = fml_lookahead(tab, w, sc, S, zt);
                }
                if (hit) cls = (unsigned int)(S.la_cls >> (2 * (zt.i - S.la_pos0))) & 3u;
                else {
                    uint32_t x0 = zt.x0, x1 = zt.x1;
                    fml_append(k, x0, x1, cb);
                    cls = fml_cls_of(fml_occ(tab, w, x0, x1), w.min_cov);
                }
            }
            if (cq && cls == 2u && lc >= w.min_cov + 1) fixed = true;
            else if ((double)hc > k * .75) fixed = true;
        }
        // the other bases: all or none of them pass bfc's tests on the path's history
        bool others = !(fixed && have);
        if (have && others) {
            if (cq && zt.eh[FML_EC_HIST_HIGH - 1] >= 0 && zt.i - zt.eh[FML_EC_HIST_HIGH - 1] < o.win_multi_ec) others = false;
            if (zt.ep[FML_EC_HIST - 1] >= 0 && zt.i - zt.ep[FML_EC_HIST - 1] < o.win_multi_ec) others = false;
        }
        if (others) {
            alt_known = (m >> (29 + dir) & 1u) != 0;
            if (alt_known) {
                const unsigned int bits = m >> (23 + 3 * dir) & 7u;
                for (int b = 0; b < 4; ++b) if (b != cb && (bits >> ((dir ? cb - b - 1 : b - cb - 1) & 3) & 1u)) alt_ok |= 1u << b;
            } else {
                if (!SLOW) return 1;
                const uint32_t mask = (uint32_t)((1ULL << k) - 1);
                unsigned long long key[4];          // the other bases (all four past the read's end or at an N): their probes together
                int r4[4];
                unsigned int want = 0;
                for (int b = 0; b < 4; ++b) {
                    key[b] = 0;
                    if (have && b == cb) continue;
                    uint32_t x0 = zt.x0, x1 = zt.x1;
                    fml_append(k, x0, x1, b);
                    key[b] = fml_key(k, x0, x1, mask);
                    want |= 1u << b;
                }
                fml_get_n<4>(tab, w, key, want, r4);
                for (int b = 0; b < 4; ++b) if ((want >> b & 1u) && fml_cls_of(r4[b], w.min_cov) != 0u) alt_ok |= 1u << b;
            }
        }
    }
    const FmlHeap1 z = S.hp.pop();
    if (!stop) {
        int n_added = 0, other_ext = 0;
        unsigned int added = 0;          // a byte per candidate: FmlPen packed (no indexed private array: that would live in scratch memory)
        for (int b = 0; b < 4; ++b) {
            FmlPen pen;
            if (!have || b != cb) {
                if (!(alt_ok >> b & 1u)) continue;          // (fixed, the history tests, a k-mer the table does not hold often enough: alt_ok has no bit)
                pen.ec = have && cob < 4 ? 1 : 0;
                pen.ec_high = pen.ec ? cq : 0;
                pen.absent = pen.absent_high = 0;
                pen.b = b;
                added |= fml_pen_pack(pen) << (8 * n_added++);
                ++other_ext;
            } else {
                pen.ec = pen.ec_high = 0;
                pen.absent = cls == 0u ? 1 : 0;
                pen.absent_high = pen.absent ? cq : 0;
                pen.b = b;
                added |= fml_pen_pack(pen) << (8 * n_added++);
            }
        }
        if (!fixed && other_ext == 0) ++S.n_failures;
        if (S.n_failures > n * 2 || S.n_stack > FML_STACK_CAP(n)) { S.rv = -3; return 2; }
        if (have || n_added == 1) {
            int first = 0, last = n_added;
            if (n_added > 1 && S.hp.n > o.max_heap) {
                int min_b = -1, mn = 0x7fffffff;
                for (int b = 0; b < n_added; ++b) {
                    const FmlPen ab = fml_pen_unpack(added >> (8 * b));
                    const int t = o.w_ec * ab.ec + o.w_ec_high * ab.ec_high + o.w_absent * ab.absent + o.w_absent_high * ab.absent_high;
                    if (mn > t) mn = t, min_b = b;
                }
                first = min_b; last = min_b + 1;
            }
            for (int a = first; a < last; ++a) {          // buf_update
                const FmlPen pen = fml_pen_unpack(added >> (8 * a));
                FmlHeap1 r = z;
                // bfc's stack holds every step of every path, and the winner is read back through its parent links: a chain of dependent reads of memory as
                // long as the read.  Only the steps that CHANGE a base of the read have anything to say then, so only those are kept (k = the path's last
                // such step); n_stack still counts every step, for the cap.
                if (have && pen.b != cb) {
                    sc.stack[FML_L(S.n_kept)] = make_uint2((unsigned int)z.k, (unsigned int)z.i | (unsigned int)pen.b << 16);
                    r.k = S.n_kept++;
                }
                ++S.n_stack;
                r.tot_pen = z.tot_pen + o.w_ec * pen.ec + o.w_ec_high * pen.ec_high + o.w_absent * pen.absent + o.w_absent_high * pen.absent_high;
                r.i = z.i + 1;
                if (pen.ec_high) { r.eh[1] = z.eh[0]; r.eh[0] = z.i; }
                if (pen.ec) { r.ep[4] = z.ep[3]; r.ep[3] = z.ep[2]; r.ep[2] = z.ep[1]; r.ep[1] = z.ep[0]; r.ep[0] = z.i; }
                fml_append(k, r.x0, r.x1, pen.b);
                S.hp.push(r);
            }
        } else stop = true;
    }
    if (stop) { S.path = z.k; S.rv = 0; return 2; }
    return 0;
}

// the first state of a walk: the k - 1 bases before the k-th good base from `start` on; false = bfc_ec1dir's "no k-mer before `end`"
template <class SC>
__device__ __forceinline__ bool fml_walk_begin(const SC &sc, const FmlWin &w, FmlWalk<SC> &S, int n, int dir, int start, int end)
{
    const int k = w.k;
    S.n = n; S.dir = dir; S.end = end; S.n_stack = S.n_kept = S.n_failures = 0; S.path = -1; S.rv = -1;
    S.hp.mem = sc.hm; S.hp.n = 0;
    S.la_len = 0; S.la_pos0 = 0; S.la_s = S.la_s1 = S.la_bs = -1; S.la_cls = 0;
    FmlHeap1 z;
    int l;
    z.tot_pen = 0; z.x0 = z.x1 = 0; z.k = -1;
    for (z.i = start, l = 0; z.i < end; ++z.i) {
        int b = (int)(sc.W[FML_L(dir ? n - 1 - z.i : z.i)] & 7u);
        if (dir) b = b < 4 ? 3 - b : 4;
        if (b < 4) {
            if (++l == k) break;
            fml_append(k, z.x0, z.x1, b);
        } else l = 0, z.x0 = z.x1 = 0;
    }
    if (z.i >= end) return false;
    for (int i = 0; i < FML_EC_HIST; ++i) z.ep[i] = -1;
    for (int i = 0; i < FML_EC_HIST_HIGH; ++i) z.eh[i] = -1;
    S.hp.push(z);
    return true;
}

// the winning path into the read: its base-changing steps, last one first
template <class SC>
__device__ __forceinline__ void fml_walk_apply(const SC &sc, const FmlWin &w, const FmlWalk<SC> &S)
{
    const int n = S.n, dir = S.dir;
    for (int l = S.path; l >= 0; l = (int)sc.stack[FML_L(l)].x) {
        const uint2 e = sc.stack[FML_L(l)];
        const int i = (int)(e.y & 0xffff), b = (int)(e.y >> 16);
        if (i < n) {
            const int fi = dir ? n - 1 - i : i, nb = dir ? 3 - b : b;
            const unsigned int old = sc.W[FML_L(fi)];
            if ((int)(old & 7u) != nb) {
                if (!dir) fml_set_base(sc, fi, nb, w.k, n);          // (the second direction reads k_fml_occ's answers; nothing does after it)
                else sc.W[FML_L(fi)] = (old & ~7u) | (unsigned int)nb;
            }
        }
    }
}

// Both bfc_ec1dir calls of a read -- from the island to the read's end, then (on the result) from the island to its start -- as ONE loop of the wave:
// a lane whose first walk is over goes on with its second while its neighbours are still in their first (the island lies anywhere in a read: the two
// walks of a read add up to its length, each alone is anything).  0 = both walks done, -13 / -14 = bfc_ec1's failures.
template <class SC>
__device__ __forceinline__ int fml_ec_walks(const FmlSlot *tab, const FmlWin &w, const FmlEcOpt &o, const SC &sc, int n, int start, int end)
{
    FmlWalk<SC> S;
    if (!fml_walk_begin(sc, w, S, n, 0, start, n)) return -13;
    int state = 0, rc = 0;          // 0 running, 1 blocked, 2 this walk is over, 3 all over
    while (true) {
        while (true) {
            if (state == 0) state = fml_step<false>(tab, w, o, sc, S);
            // go on while it pays: a step of the blocked lanes (their probes) costs four to five steps of the running ones, and about one step in ten blocks --
            // waiting for more blocked lanes than running ones idles more lanes in the steps than it saves in the rounds (measured: 4 : 1 was the worse rule)
            const int run = __popcll(__ballot(state == 0)), blk = __popcll(__ballot(state == 1));
            if (run == 0 || blk >= run) break;
        }
        if (state == 1) state = fml_step<true>(tab, w, o, sc, S);
        if (state == 2) {
            if (S.rv < 0) { rc = S.dir ? -14 : -13; state = 3; }          // (nothing has been written to the text yet: the read stays as it was)
            else {
                fml_walk_apply(sc, w, S);
                if (S.dir) state = 3;
                else if (fml_walk_begin(sc, w, S, n, 1, n - end, n)) state = 0;
                else { rc = -14; state = 3; }
            }
        }
        if (!__ballot(state == 0)) break;
    }

    return rc;
}

// bfc_ec1 between bfc_seq_conv and the write-back: the read is in sc.W (fml_word0), n_n of its bases are N; 0 = sc.W holds the corrected read
template <class SC>
__device__ __forceinline__ int fml_ec_core(const FmlSlot *tab, const FmlWin &w, const FmlEcOpt &o, SC sc, int n, int n_n)
{
    const int k = w.k;
    if ((double)n_n > n * .05) return -10;
    if (n < k) return -11;
    {          // bfc_ec_kcov: the number of solid (and solid high-quality) k-mers over each base as a sliding count of k_fml_occ's two flags
        int lc = 0, hc = 0;          // ends in [j, j + k - 1]
        for (int i = 0; i < k - 1 && i < n; ++i) { const unsigned int f = sc.W[FML_L(i)]; lc += FML_W_SOLID(f); hc += FML_W_SOLID(f) & FML_W_HIGH(f); }
        for (int j = 0; j < n; ++j) {
            if (j + k - 1 < n) { const unsigned int f = sc.W[FML_L(j + k - 1)]; lc += FML_W_SOLID(f); hc += FML_W_SOLID(f) & FML_W_HIGH(f); }
            const unsigned int fj = sc.W[FML_L(j)];
            sc.W[FML_L(j)] = (fj & ~(0xfffu << 11)) | (unsigned int)lc << 11 | (unsigned int)hc << 17;
            lc -= FML_W_SOLID(fj); hc -= FML_W_SOLID(fj) & FML_W_HIGH(fj);
        }
    }
    int start = 0, end = 0;
    {          // bfc_ec_best_island
        int l = 0, mx = 0, mx_i = -1, i;
        for (i = k - 1; i < n; ++i) {
            if (!FML_W_SOLID((unsigned int)sc.W[FML_L(i)])) {
                if (l > mx) mx = l, mx_i = i;
                l = 0;
            } else ++l;
        }
        if (l > mx) mx = l, mx_i = i;
        if (mx > 0) { start = mx_i - mx - k + 1; end = mx_i; }
        else {          // no solid k-mer: bfc_ec_first_kmer + bfc_ec_greedy_k
            int ec = -1;
            uint32_t x0 = 0, x1 = 0;
            while (true) {
                int ll = 0;
                x0 = x1 = 0;
                for (end = start; end < n; ++end) {
                    const int c = (int)(sc.W[FML_L(end)] & 7u);
                    if (c < 4) {
                        fml_append(k, x0, x1, c);
                        if (++ll == k) break;
                    } else ll = 0, x0 = x1 = 0;
                }
                if (end >= n) break;
                {
                    int mx1 = 0, mx_ec = -1, mx2 = 0;
                    for (int d = 0; d < k; ++d) {
                        const int c = (int)((x1 >> d & 1) << 1 | (x0 >> d & 1));
                        for (int j = 0; j < 4; ++j) {
                            if (j == c) continue;
                            const uint32_t y0 = (uint32_t)(j & 1) << d | (x0 & ~(1u << d)), y1 = (uint32_t)(j >> 1) << d | (x1 & ~(1u << d));
                            const int ret = fml_occ(tab, w, y0, y1);
                            if (ret < 0) continue;
                            if ((mx1 & 0xff) < (ret & 0xff)) mx2 = mx1, mx1 = ret, mx_ec = d << 2 | j;
                            else if ((mx2 & 0xff) < (ret & 0xff)) mx2 = ret;
                        }
                    }
                    ec = (mx1 & 0xff) * 3 > w.mode && (mx2 & 0xff) < 3 ? mx_ec : -1;
                }
                if (ec >= 0) break;
                if (end + (k >> 1) >= n) break;
                start = end - (k >> 1);
            }
            if (ec < 0 || end >= n) return -12;
            fml_set_base(sc, end - (ec >> 2), ec & 3, k, n);
            ++end; start = end - k;
        }
    }
    return fml_ec_walks(tab, w, o, sc, n, start, end);
}

// kmer_correct (flt_uniq = 0): persistent waves, 64 consecutive reads at a time each (handed out by an atomic counter), a lane per read: the lanes then
// walk reads of the same window and (nearly always) the same length in step.  The text of the 64 reads is one stretch of memory and enters (bfc_seq_conv)
// and leaves as such, lane = byte; what a lane works on in between is its column of the wave's words (FmlEcScratch).
// (The words were tried in LDS -- a wave per block, four blocks per CU at 150 bp: the walks are bound by instruction issue and by the waits of the blocked
// steps, and twelve waves per CU on words in memory overlap both better than four on words in LDS: 130 ms against 163 per 6.4 M reads.)
#ifndef FML_EC_WAVES
#define FML_EC_WAVES 4          // waves per SIMD the compiler is asked to leave room for (measured per 6.4 M reads, lookahead chunks of 10: 2 = 189 VGPRs, 116 ms; 3 = 168 VGPRs, 89 ms; 4 = 128 VGPRs and 440 bytes of spills, 119 ms -- chunks of 3: 4 = 128 VGPRs, 12 bytes of spills, 81 ms; 5 = 96 VGPRs, 140 bytes, 112 ms)
#endif
static __global__ void __launch_bounds__(256, FML_EC_WAVES) k_fml_ec(const FmlSlot *tab, const FmlWin *wins, int n_win, FmlEcOpt o, char *bases, char *quals,
                                                const unsigned long long *offs, long long n_reads, const unsigned short *occ, unsigned char *scratch, size_t lane_bytes,
                                                int max_len, unsigned long long *next, int *status)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const FmlEcScratch gsc = fml_scratch_of(scratch, lane_bytes, max_len, wave, lane);
    const FmlEcMem sc = fml_ec_mem(gsc);
    unsigned int *wW = gsc.W - lane;          // the wave's words: base i of lane j at wW[FML_L(i) + j]
    while (true) {
        unsigned long long r0 = 0;
        if (lane == 0) r0 = atomicAdd(next, 64ULL);
        r0 = __shfl(r0, 0);
        if ((long long)r0 >= n_reads) break;
        const long long r = (long long)r0 + lane;
        const bool live = r < n_reads;
        const int cnt = (int)((n_reads - (long long)r0) < 64 ? (n_reads - (long long)r0) : 64);
        const unsigned long long b0 = offs[r0];
        const int rel = (int)(offs[live ? r : (long long)r0 + cnt] - b0);          // where the lane's read starts in the wave's text
        const int n = live ? (int)(offs[r + 1] - b0) - rel : 0;
        const int tot = (int)(offs[(long long)r0 + cnt] - b0);
        // byte t of the wave's text is base t - rel_j of read j, the last read that starts at or before t (six shuffles find it): bfc_seq_conv
        auto read_of = [&](int t) -> int {
            int lo = 0;
            for (int s2 = 32; s2 > 0; s2 >>= 1) {          // (every lane takes part in every shuffle: a lane that sat one out would hand its neighbours nothing)
                const int mid = lo + s2, v = __shfl(rel, mid < cnt ? mid : 0);
                if (mid < cnt && v <= t) lo = mid;
            }
            return lo;
        };
        for (int t0 = 0; t0 < tot; t0 += 64) {
            const int t = t0 + lane;
            const int j = read_of(t < tot ? t : tot - 1);
            const int rel_j = __shfl(rel, j);
            if (t < tot) wW[FML_L(t - rel_j) + j] = fml_word0((unsigned char)bases[b0 + t], quals ? (int)(unsigned char)quals[b0 + t] : 0, quals != nullptr, o.q, occ[b0 + t]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int rc = 1;
        if (n > 0) {
            int lo = 0, hi = n_win;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].read0 <= r) lo = mid; else hi = mid; }
            const FmlWin w = wins[lo];
            int n_n = 0;
            for (int i = 0; i < n; ++i) n_n += (sc.W[FML_L(i)] & 7u) > 3u;
            rc = w.k > 0 ? fml_ec_core(tab, w, o, sc, n, n_n) : 1;
        }
        if (status && live) status[r] = rc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (int t0 = 0; t0 < tot; t0 += 64) {          // the corrected reads back into the text
            const int t = t0 + lane;
            const int j = read_of(t < tot ? t : tot - 1);
            const int rel_j = __shfl(rel, j), rc_j = __shfl(rc, j);
            if (t < tot && rc_j == 0) {
                const unsigned int v = wW[FML_L(t - rel_j) + j];
                bases[b0 + t] = fml_out_base(v);
                if (quals) quals[b0 + t] = fml_out_qual(v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    }
}

// worker_ec with flt_uniq, first half: "a k-mer of a read ends here and the table holds it more than once" for every text position, as one more plane
// (one wave per 64 positions, the ballot is the word -- k_fml_pack's bit order).  A lane per POSITION asks the table; a lane per read asking its 130
// k-mers one after the other was 37 ms of mostly waiting per 6.4 M reads.
static __global__ void __launch_bounds__(256) k_fml_multi(FmlPlanes pl, long long total, const FmlWin *wins, int n_win, const FmlSlot *tab, unsigned long long *pm)
{
    const long long blk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blk * 64 >= total) return;
    const long long p = blk * 64 + 63 - lane;
    bool ok = false;
    if (p < total) {
        int lo = 0, hi = n_win;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].pos0 <= p) lo = mid; else hi = mid; }
        const FmlWin w = wins[lo];
        unsigned long long key; bool hq;
        if (fml_kmer_at(pl, p, w, key, hq)) ok = fml_get(tab, w, key) > 0;
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) pm[blk + 1] = m;
}

// ... second half: max_streak per read over that plane, and the keep / trim / drop decision
static __global__ void __launch_bounds__(256) k_fml_streak(const unsigned long long *pm, const FmlWin *wins, int n_win, const unsigned long long *offs,
                                                    long long n_reads, float min_trim_frac, int *new_start, int *new_len)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const unsigned long long b = offs[r];
    const int n = (int)(offs[r + 1] - b);
    int ns = 0, nl = 0;
    if (n > 0) {
        int lo = 0, hi = n_win;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].read0 <= r) lo = mid; else hi = mid; }
        const int k = wins[lo].k;
        if (k > 0) {
            unsigned long long mx = 0, t = 0, word = 0;
            for (int i = 0; i < n; ++i) {
                const unsigned long long p = b + (unsigned long long)i;
                if (i == 0 || (p & 63) == 0) word = pm[(p >> 6) + 1];
                if (word >> (63 - (int)(p & 63)) & 1ULL) t += 1ULL << 32;          // (a k-mer ends here -- no N, k bases into the read -- and occurs more than once)
                else t = (unsigned long long)(i + 1);
                mx = mx > t ? mx : t;
            }
            if (mx >> 32 && (double)((mx >> 32) + k - 1) / n > (double)min_trim_frac) {
                const int start = (int)(unsigned int)mx, end = start + (int)(mx >> 32);
                ns = start - (k - 1); nl = end - ns;
            }
        }
    }
    new_start[r] = ns; new_len[r] = nl;
}

// the trimmed reads as a new flat text (fml_fltuniq's memmove): read r keeps [new_start, new_start + new_len)
static __global__ void __launch_bounds__(256) k_fml_trim_copy(const char *src, const char *srcq, const unsigned long long *offs, const int *new_start, const unsigned long long *new_offs,
                                                       long long n_reads, char *dst, char *dstq)
{
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const unsigned long long s = offs[r] + (unsigned long long)new_start[r], d = new_offs[r];
    const int n = (int)(new_offs[r + 1] - d);
    for (int i = threadIdx.x & 63; i < n; i += 64) { dst[d + i] = src[s + i]; if (srcq) dstq[d + i] = srcq[s + i]; }
}
