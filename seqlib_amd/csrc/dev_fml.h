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

// Per-lane work areas of the correction kernel, INTERLEAVED across the 64 lanes of a wave: element j of a lane's array sits at
// index j * 64 + lane, so that lanes working on the same position of their reads -- the common case, every lane advances one base
// per step -- touch one contiguous stretch.  The pointers below are already offset by the lane; index with FML_L(j).
#define FML_L(j) ((size_t)(j) * 64)
typedef __attribute__((address_space(3))) unsigned char fml_lds_u8;
typedef __attribute__((address_space(3))) unsigned short fml_lds_u16;
struct FmlEcScratchLds {          // the same with the per-base arrays in LDS (k_fml_ec_lds): address-space pointers, so that the walk's loads are ds_read and
    fml_lds_u8 *B;                // never wait behind the stack's stores to memory (one counter, vmcnt, orders both on this chip; a flat load would join it)
    fml_lds_u16 *cv, *oc;
    struct FmlHeap1 *heap;
    uint2 *stack;
};
struct FmlEcScratch {
    unsigned char *B;          // per base: b | q << 3 | ob << 4
    unsigned short *cv;        // per base: lcov | hcov << 6 | solid_end << 12 | high_end << 13
    unsigned short *oc;        // per base: the table's value for the k-mer of B that ENDS here (k_fml_occ), FML_OC_ABSENT, or FML_OC_ASK (no such k-mer / no longer known)
    FmlHeap1 *heap;            // heap entries 1 .. (entry 0, the top, lives in registers)
    uint2 *stack;              // x = parent, y = i | b << 16
};

__host__ __device__ inline size_t fml_scratch_bytes(int max_len)          // per LANE (a wave owns 64 times this, contiguous)
{
    size_t b = ((size_t)max_len + 15) & ~(size_t)15;
    size_t c = (2 * (size_t)max_len + 15) & ~(size_t)15;
    size_t h = sizeof(FmlHeap1) * FML_HEAP_CAP;
    size_t s = ((size_t)(FML_STACK_CAP(max_len) + 8) * 8 + 15) & ~(size_t)15;
    return b + 2 * c + h + s;
}

__device__ __forceinline__ FmlEcScratch fml_scratch_of(unsigned char *scratch, size_t lane_bytes, int max_len, size_t wave, int lane)
{
    unsigned char *p = scratch + wave * 64 * lane_bytes;
    FmlEcScratch sc;
    sc.B = p + lane; p += 64 * (((size_t)max_len + 15) & ~(size_t)15);
    sc.cv = (unsigned short *)p + lane; p += 64 * ((2 * (size_t)max_len + 15) & ~(size_t)15);
    sc.oc = (unsigned short *)p + lane; p += 64 * ((2 * (size_t)max_len + 15) & ~(size_t)15);
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

// The table's answer for the k-mer that ends at every text position, one lane per POSITION: bfc_ec_kcov asks for each of them, and
// bfc_ec1dir asks again for every position its path has not changed (all of them, for a read without errors) -- from a lane that walks
// its read one dependent probe at a time.  Asked here once, by every lane of the chip at the same time, the walk finds the answers in
// its own scratch and probes the table only where its k-mer differs from the read's.
#define FML_OC_ABSENT 0xffffu          // the k-mer is not in the table (bfc_ch_get: -1)
#define FML_OC_ASK 0xfffeu             // no k-mer of the read ends here (N, read start), or a base under it has changed since: probe
static __global__ void __launch_bounds__(256) k_fml_occ(FmlPlanes pl, long long total, const FmlWin *wins, int n_win, const FmlSlot *tab, unsigned short *occ)
{
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    int lo = 0, hi = n_win;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].pos0 <= p) lo = mid; else hi = mid; }
    const FmlWin w = wins[lo];
    unsigned long long key; bool hq;
    unsigned int v = FML_OC_ASK;
    if (fml_kmer_at(pl, p, w, key, hq)) { const int r = fml_get(tab, w, key); v = r < 0 ? FML_OC_ABSENT : (unsigned int)r; }
    occ[p] = (unsigned short)v;
}

// a base of B changes at (forward) position fi: the k-mers that end at fi .. fi + k - 1 are no longer the ones k_fml_occ asked about
template <class SC>
__device__ __forceinline__ void fml_oc_forget(const SC &sc, int fi, int k, int n)
{
    const int to = fi + k < n ? fi + k : n;
    for (int j = fi; j < to; ++j) sc.oc[FML_L(j)] = (unsigned short)FML_OC_ASK;
}

// klib's ks_heapdown / ks_heapup on tot_pen (smallest on top), operation for operation -- the order in which equal penalties leave the
// heap decides between equally cheap paths -- over a heap whose entry 0 is the register `top` and whose entries 1 .. are mem[0 ..].
// Nearly always the heap holds one state (the read's own path), and then no heap traffic reaches memory at all.
// (memory entries through an address-space-1 pointer, three 16-byte vectors each: with a plain pointer the optimiser merges "top = e" / "mem[i] = e" into one
// access through a CHOSEN address, and a `top` that has an address lives in scratch memory -- every pop and push of the one-state heap a round trip)
typedef unsigned int fml_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) fml_u4 fml_g_u4;
struct FmlHeapMem {
    fml_g_u4 *p;
    __device__ __forceinline__ FmlHeap1 operator[](size_t at) const
    {
        const fml_u4 a = p[at * 3], b = p[at * 3 + 1], c = p[at * 3 + 2];
        FmlHeap1 e;
        e.tot_pen = (int)a.x; e.i = (int)a.y; e.k = (int)a.z; e.eh[0] = (int)a.w;
        e.eh[1] = (int)b.x; e.ep[0] = (int)b.y; e.ep[1] = (int)b.z; e.ep[2] = (int)b.w;
        e.ep[3] = (int)c.x; e.ep[4] = (int)c.y; e.x0 = c.z; e.x1 = c.w;
        return e;
    }
    __device__ __forceinline__ void put(size_t at, const FmlHeap1 &e) const
    {
        p[at * 3] = fml_u4{(unsigned int)e.tot_pen, (unsigned int)e.i, (unsigned int)e.k, (unsigned int)e.eh[0]};
        p[at * 3 + 1] = fml_u4{(unsigned int)e.eh[1], (unsigned int)e.ep[0], (unsigned int)e.ep[1], (unsigned int)e.ep[2]};
        p[at * 3 + 2] = fml_u4{(unsigned int)e.ep[3], (unsigned int)e.ep[4], e.x0, e.x1};
    }
};
struct FmlHeap {
    FmlHeap1 top;
    FmlHeapMem mem;
    int n;
    // (no get(i) / set(i) that pick between `top` and memory by index: a choice between the two ADDRESSES puts `top` in scratch memory)
    __device__ __forceinline__ FmlHeap1 pop()          // z = l[0]; l[0] = l[--n]; ks_heapdown(0, n, l)
    {
        const FmlHeap1 z = top;
        --n;
        if (n > 0) {
            const FmlHeap1 tmp = mem[FML_L(n - 1)];
            int i = 0, k = 0;
            while ((k = (k << 1) + 1) < n) {          // (k >= 1: the children are in memory)
                FmlHeap1 ck = mem[FML_L(k - 1)];
                if (k != n - 1) { const FmlHeap1 c2 = mem[FML_L(k)]; if (ck.tot_pen > c2.tot_pen) { ++k; ck = c2; } }
                if (ck.tot_pen > tmp.tot_pen) break;
                if (i == 0) top = ck; else mem.put(FML_L(i - 1), ck);
                i = k;
            }
            if (i == 0) top = tmp; else mem.put(FML_L(i - 1), tmp);
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
                mem.put(FML_L(i - 1), top); i = 0;
            } else {
                const FmlHeap1 pk = mem[FML_L(k - 1)];
                if (r.tot_pen > pk.tot_pen) break;
                mem.put(FML_L(i - 1), pk); i = k;
            }
        }
        if (i == 0) top = r; else mem.put(FML_L(i - 1), r);
    }
};

struct FmlPen { int ec, ec_high, absent, absent_high, b; };
__device__ __forceinline__ unsigned int fml_pen_pack(const FmlPen &p) { return (unsigned int)(p.ec | p.ec_high << 1 | p.absent << 2 | p.absent_high << 3 | p.b << 4); }
__device__ __forceinline__ FmlPen fml_pen_unpack(unsigned int v) { return FmlPen{(int)(v & 1), (int)(v >> 1 & 1), (int)(v >> 2 & 1), (int)(v >> 3 & 1), (int)(v >> 4 & 3)}; }

// bfc_ec1dir in search coordinates: position i is base i of the read (dir 0) or base n - 1 - i of it, complemented (dir 1); the
// corrected read is written back in place
template <class SC>
__device__ __forceinline__ int fml_ec1dir(const FmlSlot *tab, const FmlWin &w, const FmlEcOpt &o, SC sc, int n, int dir, int start, int end)
{
    const int k = w.k;
    FmlHeap1 z;
    FmlHeap hp;
    int l, n_stack = 0, n_failures = 0, path = -1;
    bool found = false;
    hp.mem.p = (fml_g_u4 *)sc.heap; hp.n = 0;
    auto base_at = [&](int i, int &b, int &q, int &ob, int &lc, int &hc) {
        const int fi = dir ? n - 1 - i : i;
        const int v = sc.B[FML_L(fi)];
        b = v & 7; q = v >> 3 & 1; ob = v >> 4 & 7;
        if (dir) { b = b < 4 ? 3 - b : 4; ob = ob < 4 ? 3 - ob : 4; }
        const int c = sc.cv[FML_L(fi)];
        lc = c & 63; hc = c >> 6 & 63;
    };
    // the table's value for the path's k-mer that ends at search position i with base b there: the read's own answer (sc.oc) while the k-mer is
    // the read's -- b is the read's base and the path's last substitution lies before the k-mer (sc.oc is FML_OC_ASK wherever an original N lies
    // under the k-mer: substitutions of an N are the ones ep[] does not record) -- else a probe
    auto occ_of = [&](const FmlHeap1 &s, bool own, uint32_t x0, uint32_t x1) -> int {
        if (own && s.ep[0] <= s.i - k) {
            const unsigned int m = sc.oc[FML_L(dir ? n - s.i + k - 2 : s.i)];
            if (m != FML_OC_ASK) return m == FML_OC_ABSENT ? -1 : (int)m;
        }
        return fml_occ(tab, w, x0, x1);
    };
    z.tot_pen = 0; z.x0 = z.x1 = 0; z.k = -1;
    for (z.i = start, l = 0; z.i < end; ++z.i) {
        int b, q, ob, lc, hc;
        base_at(z.i, b, q, ob, lc, hc);
        if (b < 4) {
            if (++l == k) break;
            fml_append(k, z.x0, z.x1, b);
        } else l = 0, z.x0 = z.x1 = 0;
    }
    if (z.i >= end) return -1;
    for (int i = 0; i < FML_EC_HIST; ++i) z.ep[i] = -1;
    for (int i = 0; i < FML_EC_HIST_HIGH; ++i) z.eh[i] = -1;
    hp.push(z);
    while (true) {
        bool stop = false;
        if (hp.n == 0) return -2;
        z = hp.pop();
        if (z.i - end > o.max_end_ext) stop = true;
        if (!stop) {
            const bool have = z.i < n;
            int cb = 4, cq = 0, cob = 4, lc = 0, hc = 0;
            if (have) base_at(z.i, cb, cq, cob, lc, hc);
            int os = -1, n_added = 0, other_ext = 0;
            bool fixed = false;
            unsigned int added = 0;          // a byte per candidate: FmlPen packed (no indexed private array: that would live in scratch memory)
            if (have && cb < 4) {
                uint32_t x0 = z.x0, x1 = z.x1;
                fml_append(k, x0, x1, cb);
                os = occ_of(z, true, x0, x1);
                if (cq && os >= 0 && (os & 0xff) >= w.min_cov + 1 && lc >= w.min_cov + 1) fixed = true;
                else if ((double)hc > k * .75) fixed = true;
            }
            for (int b = 0; b < 4; ++b) {
                FmlPen pen;
                if (fixed && have && b != cb) continue;
                if (!have || b != cb) {
                    if (have) {
                        if (cq && z.eh[FML_EC_HIST_HIGH - 1] >= 0 && z.i - z.eh[FML_EC_HIST_HIGH - 1] < o.win_multi_ec) continue;
                        if (z.ep[FML_EC_HIST - 1] >= 0 && z.i - z.ep[FML_EC_HIST - 1] < o.win_multi_ec) continue;
                    }
                    uint32_t x0 = z.x0, x1 = z.x1;
                    fml_append(k, x0, x1, b);
                    const int s = fml_occ(tab, w, x0, x1);
                    if (s < 0 || (s & 0xff) < w.min_cov) continue;
                    pen.ec = have && cob < 4 ? 1 : 0;
                    pen.ec_high = pen.ec ? cq : 0;
                    pen.absent = pen.absent_high = 0;
                    pen.b = b;
                    added |= fml_pen_pack(pen) << (8 * n_added++);
                    ++other_ext;
                } else {
                    pen.ec = pen.ec_high = 0;
                    pen.absent = (os < 0 || (os & 0xff) < w.min_cov) ? 1 : 0;
                    pen.absent_high = pen.absent ? cq : 0;
                    pen.b = b;
                    added |= fml_pen_pack(pen) << (8 * n_added++);
                }
            }
            if (!fixed && other_ext == 0) ++n_failures;
            if (n_failures > n * 2 || n_stack > FML_STACK_CAP(n)) return -3;
            if (have || n_added == 1) {
                int first = 0, last = n_added;
                if (n_added > 1 && hp.n > o.max_heap) {
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
                    sc.stack[FML_L(n_stack)] = make_uint2((unsigned int)z.k, (unsigned int)z.i | (unsigned int)pen.b << 16);
                    r.tot_pen = z.tot_pen + o.w_ec * pen.ec + o.w_ec_high * pen.ec_high + o.w_absent * pen.absent + o.w_absent_high * pen.absent_high;
                    r.i = z.i + 1;
                    r.k = n_stack++;
                    if (pen.ec_high) { r.eh[1] = z.eh[0]; r.eh[0] = z.i; }
                    if (pen.ec) { r.ep[4] = z.ep[3]; r.ep[3] = z.ep[2]; r.ep[2] = z.ep[1]; r.ep[1] = z.ep[0]; r.ep[0] = z.i; }
                    fml_append(k, r.x0, r.x1, pen.b);
                    hp.push(r);
                }
            } else stop = true;
        }
        if (stop) { path = z.k; found = true; break; }
    }
    if (!found) return -1;
    for (l = path; l >= 0; l = (int)sc.stack[FML_L(l)].x) {
        const uint2 e = sc.stack[FML_L(l)];
        const int i = (int)(e.y & 0xffff), b = (int)(e.y >> 16);
        if (i < n) {
            const int fi = dir ? n - 1 - i : i;
            const int old = sc.B[FML_L(fi)], nb = dir ? 3 - b : b;
            if ((old & 7) != nb) {
                sc.B[FML_L(fi)] = (unsigned char)((old & ~7) | nb);
                if (!dir) fml_oc_forget(sc, fi, k, n);          // (the second direction reads sc.oc; nothing does after it)
            }
        }
    }
    return 0;
}

// bfc_ec1 for one read, in place in the ASCII text
// bfc_ec1 between bfc_seq_conv and the write-back: the read is in sc.B / sc.oc, n_n of its bases are N; 0 = sc.B holds the corrected read
template <class SC>
__device__ __forceinline__ int fml_ec_core(const FmlSlot *tab, const FmlWin &w, const FmlEcOpt &o, SC sc, int n, int n_n)
{
    const int k = w.k;
    if ((double)n_n > n * .05) return -10;
    if (n < k) return -11;
    {          // bfc_ec_kcov: solid / high ends, then the number of solid (and solid high-quality) k-mers over each base as a sliding count
        for (int i = 0; i < n; ++i) {          // (the k-mers' counts: k_fml_occ's answers)
            const int r = sc.oc[FML_L(i)];
            int f = 0;
            if (r < (int)FML_OC_ASK) {
                const int high = (r >> 8 & 0x3f) >= w.min_cov + 1;
                if ((r & 0xff) >= w.min_cov) f = 1 << 12 | high << 13;
                else f = high << 13;
            }
            sc.cv[FML_L(i)] = (unsigned short)f;
        }
        int lc = 0, hc = 0;          // ends in [j, j + k - 1]
        for (int i = 0; i < k - 1 && i < n; ++i) { const int f = sc.cv[FML_L(i)]; lc += f >> 12 & 1; hc += (f >> 12 & 1) & (f >> 13 & 1); }
        for (int j = 0; j < n; ++j) {
            if (j + k - 1 < n) { const int f = sc.cv[FML_L(j + k - 1)]; lc += f >> 12 & 1; hc += (f >> 12 & 1) & (f >> 13 & 1); }
            const int fj = sc.cv[FML_L(j)];
            sc.cv[FML_L(j)] = (unsigned short)((fj & 0x3000) | lc | hc << 6);
            lc -= fj >> 12 & 1; hc -= (fj >> 12 & 1) & (fj >> 13 & 1);
        }
    }
    int start = 0, end = 0;
    {          // bfc_ec_best_island
        int l = 0, mx = 0, mx_i = -1, i;
        for (i = k - 1; i < n; ++i) {
            if (!(sc.cv[FML_L(i)] >> 12 & 1)) {
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
                    const int c = sc.B[FML_L(end)] & 7;
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
            const int at = end - (ec >> 2);
            sc.B[FML_L(at)] = (unsigned char)((sc.B[FML_L(at)] & ~7) | (ec & 3));
            fml_oc_forget(sc, at, k, n);
            ++end; start = end - k;
        }
    }
    if (fml_ec1dir(tab, w, o, sc, n, 0, start, n) < 0) return -13;
    if (fml_ec1dir(tab, w, o, sc, n, 1, n - end, n) < 0) return -14;          // (nothing has been written to the text yet: the read stays as it was)
    return 0;
}

__device__ __forceinline__ unsigned char fml_conv1(int ch, int qch, bool has_q, int q_min)          // bfc_seq_conv of one base: b | q << 3 | ob << 4
{
    const int c = fml_nt5(ch);
    int q = !has_q ? 1 : (qch - 33 >= q_min ? 1 : 0);
    if (c > 3) q = 0;
    return (unsigned char)(c | q << 3 | c << 4);
}
__device__ __forceinline__ char fml_out_base(int v) { const int b = v & 7; return b != (v >> 4 & 7) ? "acgtn"[b] : "ACGTN"[b]; }
__device__ __forceinline__ char fml_out_qual(int v) { const int ob = v >> 4 & 7; return (v & 7) != ob ? (char)(34 + ob) : ((v >> 3 & 1) ? '?' : '+'); }

// bfc_ec1 for one read, in place in the ASCII text
__device__ __noinline__ int fml_ec_read(const FmlSlot *tab, const FmlWin &w, const FmlEcOpt &o, FmlEcScratch sc, char *seq, char *qual, const unsigned short *occ, int n)
{
    int n_n = 0;
    for (int i = 0; i < n; ++i) {          // bfc_seq_conv
        const unsigned char v = fml_conv1((unsigned char)seq[i], qual ? (int)(unsigned char)qual[i] : 0, qual != nullptr, o.q);
        n_n += (v & 7) > 3;
        sc.B[FML_L(i)] = v;
        sc.oc[FML_L(i)] = occ[i];
    }
    const int rc = fml_ec_core(tab, w, o, sc, n, n_n);
    if (rc < 0) return rc;
    for (int i = 0; i < n; ++i) {
        const int v = sc.B[FML_L(i)];
        seq[i] = fml_out_base(v);
        if (qual) qual[i] = fml_out_qual(v);
    }
    return 0;
}

// kmer_correct (flt_uniq = 0): persistent lanes, one read at a time each, reads handed out by an atomic counter
static __global__ void __launch_bounds__(256) k_fml_ec(const FmlSlot *tab, const FmlWin *wins, int n_win, FmlEcOpt o, char *bases, char *quals,
                                                const unsigned long long *offs, long long n_reads, const unsigned short *occ, unsigned char *scratch, size_t lane_bytes,
                                                int max_len, unsigned long long *next, int *status)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const FmlEcScratch sc = fml_scratch_of(scratch, lane_bytes, max_len, wave, threadIdx.x & 63);
    while (true) {
        // a wave takes 64 consecutive reads: its lanes then walk reads of the same window and (nearly always) the same length in step
        unsigned long long r0 = 0;
        if ((threadIdx.x & 63) == 0) r0 = atomicAdd(next, 64ULL);
        r0 = __shfl(r0, 0);
        if ((long long)r0 >= n_reads) break;
        const long long r = (long long)r0 + (threadIdx.x & 63);
        if (r >= n_reads) continue;
        const unsigned long long b = offs[r];
        const int n = (int)(offs[r + 1] - b);
        int rc = 1;
        if (n > 0) {
            int lo = 0, hi = n_win;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].read0 <= r) lo = mid; else hi = mid; }
            const FmlWin w = wins[lo];
            rc = w.k > 0 ? fml_ec_read(tab, w, o, sc, bases + b, quals ? quals + b : nullptr, occ + b, n) : 1;
        }
        if (status) status[r] = rc;
    }
}

// The same for short reads (the sequencer's: every read of the batch at most FML_EC_LDS_MAX bases), the walk's per-base arrays in LDS.  k_fml_ec
// keeps them in memory, and with a megabyte of scratch per wave every pass over a read comes from HBM again: 25 KB of traffic per 150 bp read,
// a dependent round trip to memory per step of the walk.  Here a block is ONE wave and owns 5 bytes x 64 lanes per base of LDS (B, cv, oc; the
// stack of the search, written once and read back along one path, and the heap beyond its top entry stay in memory), its 64 consecutive reads
// enter and leave as whole coalesced rows (lane = base, one read at a time) instead of a byte per lane per step.
#define FML_EC_LDS_MAX 480          // 5 x 64 x 480 = 150 KB of the CU's 160
static __global__ void __launch_bounds__(64) k_fml_ec_lds(const FmlSlot *tab, const FmlWin *wins, int n_win, FmlEcOpt o, char *bases, char *quals,
                                                    const unsigned long long *offs, long long n_reads, const unsigned short *occ, unsigned char *scratch, size_t lane_bytes,
                                                    int max_len, unsigned long long *next, int *status)
{
    extern __shared__ unsigned char fml_lds[];
    const int lane = threadIdx.x;
    const size_t row = ((size_t)max_len + 15) & ~(size_t)15;          // bases per lane, as fml_scratch_of rounds
    const FmlEcScratch gsc = fml_scratch_of(scratch, lane_bytes, max_len, blockIdx.x, lane);          // heap and stack: memory
    fml_lds_u8 *lB = (fml_lds_u8 *)fml_lds;
    fml_lds_u16 *lcv = (fml_lds_u16 *)(lB + 64 * row), *loc = (fml_lds_u16 *)(lB + 64 * row * 3);
    FmlEcScratchLds sc;
    sc.B = lB + lane; sc.cv = lcv + lane; sc.oc = loc + lane; sc.heap = gsc.heap; sc.stack = gsc.stack;
    while (true) {
        unsigned long long r0 = 0;
        if (lane == 0) r0 = atomicAdd(next, 64ULL);
        r0 = __shfl(r0, 0);
        if ((long long)r0 >= n_reads) break;
        const long long r = (long long)r0 + lane;
        const bool live = r < n_reads;
        const unsigned long long b = live ? offs[r] : 0;
        const int n = live ? (int)(offs[r + 1] - b) : 0;
        int n_n = 0;
        const int cnt = (int)((n_reads - (long long)r0) < 64 ? (n_reads - (long long)r0) : 64);
        for (int j = 0; j < cnt; ++j) {          // bfc_seq_conv, read j of the wave's: lane = base
            const unsigned long long bj = __shfl(b, j);
            const int nj = __shfl(n, j);
            int nn = 0;
            for (int i = lane; i < nj; i += 64) {
                const unsigned char v = fml_conv1((unsigned char)bases[bj + i], quals ? (int)(unsigned char)quals[bj + i] : 0, quals != nullptr, o.q);
                nn += (v & 7) > 3;
                lB[FML_L(i) + j] = v;
                loc[FML_L(i) + j] = occ[bj + i];
            }
            for (int d = 32; d > 0; d >>= 1) nn += __shfl_xor(nn, d);
            if (lane == j) n_n = nn;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int rc = 1;
        if (n > 0) {
            int lo = 0, hi = n_win;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wins[mid].read0 <= r) lo = mid; else hi = mid; }
            const FmlWin w = wins[lo];
            rc = w.k > 0 ? fml_ec_core(tab, w, o, sc, n, n_n) : 1;
        }
        if (status && live) status[r] = rc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = 0; j < cnt; ++j) {          // the corrected reads back into the text
            if (__shfl(rc, j) != 0) continue;
            const unsigned long long bj = __shfl(b, j);
            const int nj = __shfl(n, j);
            for (int i = lane; i < nj; i += 64) {
                const int v = lB[FML_L(i) + j];
                bases[bj + i] = fml_out_base(v);
                if (quals) quals[bj + i] = fml_out_qual(v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// worker_ec with flt_uniq: max_streak per read and the keep / trim / drop decision
static __global__ void __launch_bounds__(256) k_fml_streak(const FmlSlot *tab, const FmlWin *wins, int n_win, const char *bases, const unsigned long long *offs,
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
        const FmlWin w = wins[lo];
        const int k = w.k;
        if (k > 0) {
            unsigned long long mx = 0, t = 0;
            uint32_t x0 = 0, x1 = 0;
            int l = 0;
            for (int i = 0; i < n; ++i) {
                const int c = fml_nt5((unsigned char)bases[b + i]);
                if (c < 4) {
                    fml_append(k, x0, x1, c);
                    if (++l >= k) {
                        if (fml_occ(tab, w, x0, x1) > 0) t += 1ULL << 32;
                        else t = (unsigned long long)(i + 1);
                    } else t = (unsigned long long)(i + 1);
                } else l = 0, x0 = x1 = 0, t = (unsigned long long)(i + 1);
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
