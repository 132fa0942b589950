// dev_seed4.h -- SMEM seeding passes 1 + 2 (bwa's mem_collect_intv / bwt_smem1a, SURVEY.md A.3/A.4, reached from
// /root/reference/src/BWAAligner.cpp:104 -> mem_align1 -> mem_chain) with WAVE-LEVEL MODES.
//
// A per-lane state machine with one bwt_extend per trip of a wave-uniform loop (the first design, experiments/dev_seed2_state_machine.h)
// spends ~340 instructions around every extend, because each trip walks through every phase block that any lane of the wave is in;
// measurements showed that kernel bound by exactly that, not by the memory system, while a plain lock-step loop
// costs a third per extend.  Here the lanes still own one read each and still fetch the next read when theirs is done, but the wave runs
// TIGHT LOOPS, one kind of step at a time:
//     F loop   forward extension steps (bwt_smem1a's first loop) for the lanes that are in one
//     D loop   forward steps of an interval that is down to ONE occurrence: the read against the reference text, no rank (below)
//     B loop   backward steps (one work-list entry against the row's base) for the lanes that are in a backward row
//     events   read assignment and the start of the next bwt_smem1a call, between the loops
// A loop keeps going while enough of its lanes are left; lanes of the other kind idle meanwhile, which makes the lanes of a wave fall
// into step (forward together, then backward together).  The price is idle lanes, the gain is that a step costs little more than its
// extend: the forward step ranks its symbol and "everything greater" (two masked popcounts per block, for both strands' interval
// starts), the backward step ranks one symbol and never touches x[1] -- no consumer reads it.
// Measured on C3 (SEED4_DEBUG): per read ~72 backward, ~12 rank-based forward and ~5 direct steps.  Which loop runs next decides how many
// lanes each step carries.  "Largest group first, each loop until half of its lanes are through" levels the three groups out at a third
// of the wave each, and the backward loop -- 86 % of the wave steps -- ran with 15 of 64 lanes.  SEED4_SCHED = 1 works in batches: a short
// loop (forward, direct) is entered when SEED4_BATCH lanes wait for it and runs until an eighth of them is left; the backward loop keeps
// going until a batch is ready elsewhere (20 lanes per step, 10 % fewer wave steps, 46 -> 44 ms).  Also measured and dropped: smaller
// batches (three times the rounds, each paying the events block), and ONE loop in which every lane takes whichever step it is due (28
// lanes per trip and a third fewer trips, but a trip that carries all three kinds of step costs 1.6 x a single-kind one even with every
// load issued up front: 59 ms against 55).
#pragma once
#include "dev_occ.h"
#include "dev_pack.h"

// one rank block in flight.  Plain scalar members on purpose: selecting a vector element or an array slot by the per-lane symbol makes
// the compiler park the struct in scratch memory.
template <typename I>
struct RankLd {
    uint32_t c0, c1, c2, c3;  // running counts of A, C, G, T before the block
    uint32_t l0, l1, h0, h1;  // low / high bit planes of its 64 symbols
    I s0, s1, s2, s3;         // u64 index: base counts of the 2^32-symbol super-block
    uint32_t n;               // symbols of the block that count: 1..64
};

template <typename I>
__device__ __forceinline__ void rank_issue(const DevFM<I> &fm, I pos, RankLd<I> &r)
{   // pos = BWT position whose inclusive rank is wanted (bwt_occ's k), not yet mapped past the sentinel
    const I kk = pos - (pos >= fm.primary ? 1 : 0);
    const uint4 *blk = fm.occ + ((size_t)(kk >> 6) << 1);
    const uint4 cn = blk[0], pl = blk[1];
    r.c0 = cn.x; r.c1 = cn.y; r.c2 = cn.z; r.c3 = cn.w;
    r.l0 = pl.x; r.l1 = pl.y; r.h0 = pl.z; r.h1 = pl.w;
    r.n = ((uint32_t)kk & 63) + 1;
    if (sizeof(I) == 8) {
        const ulonglong2 *sp = (const ulonglong2 *)(fm.sup + ((uint64_t)kk >> 32) * 4);
        const ulonglong2 a = sp[0], b = sp[1];
        r.s0 = (I)a.x; r.s1 = (I)a.y; r.s2 = (I)b.x; r.s3 = (I)b.y;
    } else r.s0 = r.s1 = r.s2 = r.s3 = 0;
}

template <typename T>
__device__ __forceinline__ T sel4(int c, T v0, T v1, T v2, T v3) { const T lo = (c & 1) ? v1 : v0, hi = (c & 1) ? v3 : v2; return (c & 2) ? hi : lo; }

// occurrences of symbol c, and (GT) of the symbols greater than c, in BWT[0..pos]
template <typename I, bool GT>
__device__ __forceinline__ void rank_finish(const RankLd<I> &r, int c, I &cnt_c, I &cnt_gt)
{
    const uint64_t m = ~0ull >> (64 - r.n);
    const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
    const uint32_t a0 = (c & 1) ? r.l0 : ~r.l0, a1 = (c & 1) ? r.l1 : ~r.l1;
    const uint32_t b0 = (c & 2) ? r.h0 : ~r.h0, b1 = (c & 2) ? r.h1 : ~r.h1;
    cnt_c = (I)(sel4<uint32_t>(c, r.c0, r.c1, r.c2, r.c3) + (uint32_t)__popc(a0 & b0 & m0) + (uint32_t)__popc(a1 & b1 & m1));
    if (sizeof(I) == 8) cnt_c += sel4<I>(c, r.s0, r.s1, r.s2, r.s3);
    if (GT) {
        // symbols > c:  c = 0: low | high,  c = 1: high,  c = 2: low & high,  c = 3: none
        const uint32_t g0 = sel4<uint32_t>(c, r.l0 | r.h0, r.h0, r.l0 & r.h0, 0u);
        const uint32_t g1 = sel4<uint32_t>(c, r.l1 | r.h1, r.h1, r.l1 & r.h1, 0u);
        cnt_gt = (I)(sel4<uint32_t>(c, r.c1 + r.c2 + r.c3, r.c2 + r.c3, r.c3, 0u) + (uint32_t)__popc(g0 & m0) + (uint32_t)__popc(g1 & m1));
        if (sizeof(I) == 8) cnt_gt += sel4<I>(c, r.s1 + r.s2 + r.s3, r.s2 + r.s3, r.s3, (I)0);
    } else cnt_gt = 0;
}

template <typename I>
__device__ __forceinline__ I l2_of(const DevFM<I> &fm, int c) { return sel4<I>(c, fm.L2[0], fm.L2[1], fm.L2[2], fm.L2[3]); }

// bwt_set_intv: the bidirectional interval of the single base c
template <typename I>
__device__ __forceinline__ void set_intv4(const DevFM<I> &fm, int c, I &k0, I &k1, I &k2)
{
    k0 = l2_of<I>(fm, c) + 1; k1 = l2_of<I>(fm, 3 - c) + 1;
    k2 = (c == 3 ? fm.L2[4] : l2_of<I>(fm, c + 1)) - l2_of<I>(fm, c);
}

// base of the indexed text T = forward ++ reverse complement (what the FM-index is over) at position p < 2 l_pac, through an 8-byte
// window of the 2-bit pac (32 bases; the reverse strand walks the same bytes downwards)
struct RWin { uint64_t bits; int64_t chunk; };
// do the four read bases at codes[a .. a + 4) equal the four text bases at p .. p + 3?  Answered only when those are the four bases of ONE byte of pac (forward
// strand: p a multiple of 4; reverse strand: the mirror position ends a byte) and lie on one strand; false otherwise and on any ambiguous read base -- the caller
// then compares base by base as before.
__device__ __forceinline__ bool dir4_match(const DevRef &R, const uint8_t *codes, uint64_t a, int64_t p, RWin &w)
{
    const bool rev = p >= R.l_pac;
    if (!rev && p + 3 >= R.l_pac) return false;
    if (p + 3 >= (R.l_pac << 1)) return false;
    const int64_t f = rev ? (R.l_pac << 1) - 1 - p : p;          // position of the first of the four on the forward strand; the others follow at f + 1 .. (rev: f - 1 ..)
    if ((f & 3) != (rev ? 3 : 0)) return false;
    const int64_t fb = rev ? f - 3 : f;
    const int64_t ch = fb >> 5;
    if (ch != w.chunk) { w.bits = *(const uint64_t *)(R.pac + (ch << 3)); w.chunk = ch; }
    const uint32_t tb = (uint32_t)(w.bits >> (((fb >> 2) & 7) << 3)) & 0xffu;          // b(fb) << 6 | b(fb + 1) << 4 | b(fb + 2) << 2 | b(fb + 3)
    uint32_t x;
    __builtin_memcpy(&x, codes + a, 4);
    if (x & 0xfcfcfcfcu) return false;
    const uint32_t q0 = x & 3u, q1 = (x >> 8) & 3u, q2 = (x >> 16) & 3u, q3 = x >> 24;
    const uint32_t qq = rev ? ((q3 << 6 | q2 << 4 | q1 << 2 | q0) ^ 0xffu) : (q0 << 6 | q1 << 4 | q2 << 2 | q3);
    return qq == tb;
}

// ... and the same for 32 bases against ONE 8-byte chunk of pac (forward strand: p a multiple of 32; reverse strand: the mirror position ends a chunk): a contig
// that matches the reference is one SMEM of tens of kilobases, and four bases per step still walked it for ~80 ms on its lane.  Four read bases pack into a
// byte with one multiply (x = q0 | q1 << 8 | q2 << 16 | q3 << 24; x * (2^30 + 2^20 + 2^10 + 1) >> 24 = q0 << 6 | q1 << 4 | q2 << 2 | q3: the products that
// land in bits 24-31 are exactly these four), eight such bytes are pac's chunk; on the reverse strand the chunk holds the complement of the read's bases in
// reverse order: bytes swapped, the four pairs of every byte swapped, all bits flipped.
__device__ __forceinline__ bool dir32_match(const DevRef &R, const uint8_t *codes, uint64_t a, int64_t p, RWin &w)
{
    const bool rev = p >= R.l_pac;
    if (!rev && p + 31 >= R.l_pac) return false;
    if (p + 31 >= (R.l_pac << 1)) return false;
    const int64_t f = rev ? (R.l_pac << 1) - 1 - p : p;
    if ((f & 31) != (rev ? 31 : 0)) return false;
    const int64_t ch = f >> 5;
    if (ch != w.chunk) { w.bits = *(const uint64_t *)(R.pac + (ch << 3)); w.chunk = ch; }
    uint32_t x[8];
    __builtin_memcpy(x, codes + a, 32);
    uint32_t amb = 0;
    uint64_t q = 0;
    for (int j = 0; j < 8; ++j) {
        amb |= x[j];
        q |= (uint64_t)((x[j] * 0x40100401u) >> 24) << (8 * j);
    }
    if (amb & 0xfcfcfcfcu) return false;
    if (rev) {
        q = __builtin_bswap64(q);
        q = ((q & 0x0303030303030303ULL) << 6) | ((q & 0x0c0c0c0c0c0c0c0cULL) << 2) | ((q >> 2) & 0x0c0c0c0c0c0c0c0cULL) | ((q >> 6) & 0x0303030303030303ULL);
        q = ~q;
    }
    return q == w.bits;
}

__device__ __forceinline__ int text_at(const DevRef &R, int64_t p, RWin &w)
{
    const bool rev = p >= R.l_pac;
    const int64_t f = rev ? (R.l_pac << 1) - 1 - p : p;
    const int64_t ch = f >> 5;
    if (ch != w.chunk) { w.bits = *(const uint64_t *)(R.pac + (ch << 3)); w.chunk = ch; }
    const int b = (int)((w.bits >> ((((f >> 2) & 7) << 3) + ((~f & 3) << 1))) & 3);
    return rev ? 3 - b : b;
}

// one forward extension (bwt_extend on the swapped interval with the complemented base) of (k0, k1, k2) by query base cq
template <typename I>
__device__ __forceinline__ void fwd_step(const DevFM<I> &fm, I k0, I k1, I k2, int cq, I &o0, I &o1, I &o2)
{
    const int c = 3 - cq;
    RankLd<I> rk, rl;
    rank_issue<I>(fm, k1 - 1, rk); rank_issue<I>(fm, k1 - 1 + k2, rl);
    I ckc, ckg, clc, clg;
    rank_finish<I, true>(rk, c, ckc, ckg); rank_finish<I, true>(rl, c, clc, clg);
    o2 = clc - ckc;
    o1 = l2_of<I>(fm, c) + 1 + ckc;
    o0 = k0 + ((k1 <= fm.primary && k1 + k2 - 1 >= fm.primary) ? 1 : 0) + (clg - ckg);
}

// ---------------------------------------------------------------- k-mer table
// T[W] = SA interval of the K-mer W, computed with the very forward steps the seeding loops run, so that a jump through the table and
// K - 1 steps leave the same numbers.  What it replaces, exactly (K <= min_seed_len, checked at launch):
//   forward:  the first K - 1 steps of a bwt_smem1a / bwt_seed_strategy1 call cannot stop (no ambiguous base among the K, the read has
//             them, and the K-mer's interval is still >= min_intv -- sizes only shrink), so the call starts at T[W] / T[revcomp W].
//             The entries those steps would have pushed (prefixes of 1 .. K-1 bases) become VIRTUAL: they are not on the work list.
//   backward: a work-list entry is the SA interval of a substring, and rows extend all entries by the same base, so the prefix of k bases
//             after t rows is the interval of q[sx-t, sx+k).  Nothing an entry does before it is K bases long can be seen: it cannot
//             be reported (shorter than min_seed_len), `last_start` only compares within a row, and entries influence only SHORTER
//             ones (nc == 0, last_sz).  So the prefix of k bases joins its row when it reaches K bases -- row t = K - k, as the row's last
//             entry, straight from T -- and the tests it meets there (>= min_intv, size differs from the last kept entry's: nested
//             intervals of equal size are equal, and stay equal) drop it exactly when bwa would have dropped it in that row or before.
//             Rows go on while virtual entries are left even when the list is empty.
// This removes the triangle of (prefix, row) steps every call spends on intervals of thousands of occurrences: about K (K - 1) / 2
// backward and K - 1 forward extends per call, for K - 1 + 2 table reads.
template <typename I>
__global__ void k_kmer_lut(DevFM<I> fm, int K, LutE<I> *out)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (1ull << (2 * K))) return;
    I k0, k1, k2;
    set_intv4<I>(fm, (int)((w >> (2 * (K - 1))) & 3), k0, k1, k2);
    for (int t = 1; t < K; ++t) {
        I o0, o1, o2;
        fwd_step<I>(fm, k0, k1, k2, (int)((w >> (2 * (K - 1 - t))) & 3), o0, o1, o2);
        k0 = o0; k1 = o1; k2 = o2;
        if (k2 == 0) break;
    }
    LutE<I> e; e.lo = k2 ? k0 : (I)1; e.sz = k2;
    out[w] = e;
}

// ---------------------------------------------------------------- repeat filter
// Pass 2 of mem_collect_intv re-seeds from the middle of every long SMEM with few occurrences: bwt_smem1a(mid, min_intv = occ + 1), of whose
// results only the intervals of >= min_seed_len bases are kept.  For a read from unique sequence (occ = 1) that is HALF of the steps of
// passes 1 + 2 (k_seed12m without it: 21.9 ms for 8.3 M reads, with it 44.3) -- ~14 backward rows of 3-4 entries -- and it nearly always
// ends with nothing to keep (3.7 % of the reads of the C3 workload keep anything): sequence that occurs twice AND is >= 19 bases long is
// what repeats are made of.  A kept interval [s, e) covers mid, is >= min_seed_len long and occurs >= 2 times, so it contains a window of
// rep_k <= min_seed_len bases that covers mid and occurs >= 2 times.  The filter holds one bit per hashed rep_k-mer of the indexed text
// that occurs at least twice (two suffixes adjacent in the suffix array share their first rep_k bases): when none of the <= rep_k windows
// covering mid is in it, the call cannot keep anything, and it has no other effect -- it is not made (k_seed2_select).  A false positive
// only means the call is made as before.  Exact for occ = 1 (min_intv = 2) calls; the others (parent SMEMs with 2..split_width
// occurrences) are always made.
template <typename I>
__global__ void k_rep_filter(DevFM<I> fm, DevRef R, int kf, uint32_t *bits, uint64_t mask)
{
    const uint64_t n = (uint64_t)fm.seq_len;           // ranks 0..n (rank 0 is the sentinel's suffix)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t p = (uint64_t)fm.sa_dense[i], q = (uint64_t)fm.sa_dense[i + 1];
        if (p + (uint64_t)kf > n || q + (uint64_t)kf > n) continue;
        RWin wp, wq; wp.bits = wq.bits = 0; wp.chunk = wq.chunk = -1;
        uint64_t kmer = 0;
        bool same = true;
        for (int t = 0; t < kf; ++t) {
            const int a = text_at(R, (int64_t)(p + t), wp), b = text_at(R, (int64_t)(q + t), wq);
            if (a != b) { same = false; break; }
            kmer = (kmer << 2) | (uint64_t)a;
        }
        if (!same) continue;
        const uint64_t h = rep_hash(kmer) & mask;
        atomicOr(bits + (h >> 5), 1u << (h & 31));
    }
}

// Which pass-2 calls of a read are made: one lane per read, after pass 1 (k_seed12m, mode 1).  For every interval pass 1 kept -- in the order
// pass 2 walks them -- bwa's own conditions (>= split_len bases, <= split_width occurrences, a base at the middle position) and, for a
// parent with one occurrence, the repeat filter above.  p2mask[r]: bit k < 63 = the call for interval k is made; bit 63 = intervals from
// the 63rd on are left to the kernel's own tests (a read with that many intervals is rare).  Reads with any call go on `list`.
template <typename I>
__global__ void __launch_bounds__(128) k_seed2_select(DevFM<I> fm, Chunk ck, DevOpt dopt, unsigned long long *p2mask, int *list, unsigned int *n_list,
                                                      uint32_t *items, unsigned int *n_items, uint32_t cap_items)
{
    const slx_opt &opt = dopt.o;
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = ck.spread > 1 ? t_ / ck.spread : t_;          // (a small chunk: one read per wave, on its lane 0)
    const bool live = r < ck.n_reads && (ck.spread <= 1 || t_ % ck.spread == 0);
    const int n = live ? (int)ck.intv_n[r] : 0;
    unsigned long long mask = 0;
    if (n > 0) {
        const uint64_t qoff = ck.offs[r];
        const int len = (int)(ck.offs[r + 1] - qoff);
        const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
        QWin win; win.bits = 0; win.chunk = 0xffffffffu;
        auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
        const bool use_rep = fm.rep != nullptr && fm.rep_k <= opt.min_seed_len && fm.rep_k > 0;
        const int kf = fm.rep_k;
        const uint64_t kmask = kf < 32 ? (1ull << (2 * kf)) - 1 : ~0ull;
        for (int k = 0; k < n; ++k) {
            const size_t o = (size_t)r * ck.cap_intv + k;
            const qp_t inf = ((const qp_t *)ck.intv_info)[o];
            const I s = ((const I *)ck.intv_x2)[o];
            const int start = QP_HI(inf), end = QP_LO(inf);
            if (end - start < split_len || s > (I)opt.split_width) continue;
            if (k >= 63) { mask |= 1ull << 63; break; }
            const int mid = (start + end) >> 1;
            if (qb(mid) > 3) continue;                  // bwt_smem1a returns at once on an ambiguous start
            bool take = true;
            if (s == (I)1 && use_rep) {                 // does any window of rep_k bases that covers mid occur twice in the text?
                const int a_lo = mid - kf + 1 > 0 ? mid - kf + 1 : 0, a_hi = mid < len - kf ? mid : len - kf;
                uint64_t kmer = 0;
                int good = 0;                           // valid bases at the end of the current window
                for (int p = a_lo; p < a_lo + kf - 1; ++p) {
                    const int c = qb(p);
                    if (c > 3) { good = 0; kmer = 0; } else { kmer = ((kmer << 2) | (uint64_t)c) & kmask; ++good; }
                }
                take = false;
                for (int a = a_lo; a <= a_hi; ++a) {
                    const int c = qb(a + kf - 1);
                    if (c > 3) { good = 0; kmer = 0; continue; }
                    kmer = ((kmer << 2) | (uint64_t)c) & kmask; ++good;
                    if (good < kf) continue;
                    const uint64_t h = rep_hash(kmer) & fm.rep_mask;
                    if ((fm.rep[h >> 5] >> (h & 31)) & 1u) { take = true; break; }
                }
            }
            if (take) mask |= 1ull << k;
        }
    }
    if (live) p2mask[r] = mask;
    // The calls of a read are independent of each other (each reads pass 1's intervals and appends its own), so they go out as single ITEMS
    // (read << 6 | interval), one lane each in k_seed12m<2>: the reads in multi-copy repeats make a dozen calls of ~175 backward steps, and
    // taken one read per lane they were a tail of 4 busy lanes per wave.  A read with calls beyond the mask, or whose items do not fit,
    // goes on `list` and is walked by one lane as before.
    const uint32_t cnt = (uint32_t)__popcll(mask & ~(1ull << 63));
    const bool whole = (mask >> 63) != 0;
    uint32_t base = 0xffffffffu;
    {   // wave-aggregated reservation of cnt item slots per lane
        const uint32_t want = (live && !whole) ? cnt : 0u;
        uint32_t incl = want;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)(threadIdx.x & 63) >= d) incl += u; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        uint32_t wbase = 0;
        if (total) {
            if ((threadIdx.x & 63) == 0) wbase = atomicAdd(n_items, total);
            wbase = (uint32_t)__shfl((int)wbase, 0, 64);
            base = wbase + incl - want;
        }
    }
    if (!live || !mask) return;
    if (!whole && cnt && base + cnt <= cap_items) {
        unsigned long long m = mask;
        for (uint32_t t = 0; t < cnt; ++t) { const int k = __ffsll((long long)m) - 1; m &= m - 1; items[base + t] = (uint32_t)r << 6 | (uint32_t)k; }
    } else {
        if (!whole) for (uint32_t t = 0; t < cnt && base + t < cap_items; ++t) items[base + t] = 0xffffffffu;   // reserved but not used: k_seed12m skips them
        list[wave_fetch_inc(n_list)] = r;
    }
}

// the K-mer at q[p, p + K) as a table index and its reverse complement's; returns the offset of the first ambiguous base, or -1
template <typename QB>
__device__ __forceinline__ int kmer_codes(QB &qb, int p, int K, uint32_t &fw, uint32_t &rc)
{
    fw = 0; rc = 0;
    for (int t = 0; t < K; ++t) {
        const int c = qb(p + t);
        if (c > 3) return t;
        fw = (fw << 2) | (uint32_t)c;
        rc |= (uint32_t)(3 - c) << (2 * t);
    }
    return -1;
}

#ifndef SEED4_MIN_WAVES
#define SEED4_MIN_WAVES 4
#endif
#ifndef SEED4_MIN_WAVES_U64
#define SEED4_MIN_WAVES_U64 3
#endif
#ifndef SEED4_FLOOR_SHIFT
#define SEED4_FLOOR_SHIFT 1   // a mode loop ends when its active lanes drop to (lanes at entry) >> this
#endif
#ifndef SEED4_WL_LDS
#define SEED4_WL_LDS 6        // work-list entries per list and lane held in LDS (u32 index: 18 KB per 128-lane block, 8 blocks per CU)
#endif
#ifndef SEED4_WL_LDS_U64
#define SEED4_WL_LDS_U64 4    // ... u64 index: 20 KB per block at 6 blocks per CU
#endif
#ifndef SEED4_SCHED
#define SEED4_SCHED 1         // 1 = batches (below), 0 = largest group first, each loop until half of its lanes are through
#endif
#ifndef SEED4_BATCH
#define SEED4_BATCH 20
#endif
#if SEED4_SCHED == 1
#ifndef SEED4_SHORT_FLOOR_SHIFT
#define SEED4_SHORT_FLOOR_SHIFT 3
#endif
#ifndef SEED4_ITERS_B
#define SEED4_ITERS_B 96
#endif
#else
#define SEED4_SHORT_FLOOR_SHIFT SEED4_FLOOR_SHIFT
#define SEED4_ITERS_B SEED4_ITERS
#endif
#ifndef SEED4_ITERS
#define SEED4_ITERS 24        // steps a mode loop runs at most before the wave looks at events and modes again
#endif

#ifndef SEED2C_MIN
#define SEED2C_MIN 12         // pass-2 calls with at least this many entries after their forward phase go to k_seed2_coop (one wave per call)
#endif
#define SEED2C_CAP 256        // ... entries per list in its LDS (a call whose forward list + K does not fit stays on its lane)
#ifndef SEED4_DEBUG_MODE
#define SEED4_DEBUG_MODE 0    // 1 / 2: the statistics of that launch only (pass 1 / pass 2)
#endif
#ifndef SEED4_DEBUG
#define SEED4_DEBUG 0         // 1 + SLX_DEBUG_CYC=3: loop statistics (steps and active lanes per mode) on stderr
#endif

// Direct forward steps.  Once a forward interval holds a single occurrence (the rule after ~14 bases of a read from unique sequence), a
// forward extend can only keep it or empty it: x[2] stays 1 and x[0] stays put while the next base of the read equals the next base of
// the text behind that occurrence (the sentinel adjustment and the "greater symbols" count are both 0 for a kept single occurrence), and
// x[1] is read by nothing but the next rank-based forward step.  So from there the walk needs one suffix-array read (the occurrence's
// position, dense SA) and the 2-bit reference -- 32 bases per 8-byte read -- instead of a rank per base.  A lane switches after two
// consecutive steps at one occurrence (chance matches next to a mismatch die at once: not worth the two reads).
#ifndef SEED4_DIRECT_AFTER
#define SEED4_DIRECT_AFTER 2
#endif
enum Seed4Phase : int { S4_FETCH = 0, S4_INIT, S4_START, S4_FWD, S4_DIR, S4_ROW, S4_BWD, S4_DONE };

// MODE 1: pass 1 of every read of the chunk (all SMEMs).  MODE 2: pass 2 (re-seeding) of the reads on `list`, for the intervals
// k_seed2_select marked in p2mask -- the two passes are separate launches so that the selection between them runs one lane per read.
#ifdef SEED4_NUM_VGPR
#define SEED4_VGPR_ATTR __attribute__((amdgpu_num_vgpr(SEED4_NUM_VGPR)))
#else
#define SEED4_VGPR_ATTR
#endif
template <typename I, int MODE>
__global__ void SEED4_VGPR_ATTR __launch_bounds__(128, (sizeof(I) == 8 ? SEED4_MIN_WAVES_U64 : SEED4_MIN_WAVES)) k_seed12m(DevFM<I> fm, DevRef R, Chunk ck, DevOpt dopt, unsigned int *queue, uint32_t wave_quota,
                                                                                                              const int *list, const unsigned int *n_list, const unsigned long long *p2mask,
                                                                                                              const uint32_t *items, const unsigned int *n_items, uint32_t cap_items,
                                                                                                              uint32_t *long_items, unsigned int *n_long, uint32_t cap_long)
{
    // wave_quota: reads a wave takes from the queue before it stops fetching and drains (0 = until the queue is empty: persistent waves).
    // With a quota the launch has many more blocks than fit the chip and slots keep coming free, so the other workers' kernels -- the
    // dozens of small launches of a chunk above all -- get onto the CUs while a seeding launch is in flight instead of behind it.
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x & 63;
    const uint32_t pool = ck.spread > 1 ? 1u : (uint32_t)SEED_POOL;          // reads a wave takes from the queue at a time: ONE for a small chunk, so that every read gets a wave of its own
    WorkLists<I> wl;
    wl.base = (IntvE<I> *)ck.lists + (size_t)blockIdx.x * ((size_t)2 * ck.cap_list * 128) + threadIdx.x; wl.stride = 128; wl.cap = ck.cap_list;
    // the first SEED4_WL_LDS entries of both work lists live in LDS (entry-major: lanes at the same entry hit different banks); with the
    // k-mer table a list rarely grows beyond a handful of entries (the forward phase only pushes where the interval size changes past the
    // k-th base), so the global lists -- 3 KB of partial-line writes per read before -- are left to the repeats
    constexpr int WL = sizeof(I) == 8 ? SEED4_WL_LDS_U64 : SEED4_WL_LDS;
    __shared__ I s_wx0[2 * WL * 128], s_wx2[2 * WL * 128];
    __shared__ uint32_t s_winf[2 * WL * 128];
    auto wl_put = [&](int list, int e, I a, I sz, uint32_t inf) {
        if (e < WL) { const int o = (list * WL + e) * 128 + (int)threadIdx.x; s_wx0[o] = a; s_wx2[o] = sz; s_winf[o] = inf; }
        else { IntvE<I> v; v.x0 = a; v.x1 = 0; v.x2 = sz; v.info = inf; wl.at(list, e) = v; }
    };
    auto wl_get = [&](int list, int e, I &a, I &sz, uint32_t &inf) {
        if (e < WL) { const int o = (list * WL + e) * 128 + (int)threadIdx.x; a = s_wx0[o]; sz = s_wx2[o]; inf = s_winf[o]; }
        else { const IntvE<I> v = wl.at(list, e); a = v.x0; sz = v.x2; inf = v.info; }
    };
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    // MODE 2: the work units are the single-call items first, then the whole reads of `list` (k_seed2_select)
    uint32_t n_it = 0;
    if (MODE == 2) { n_it = (uint32_t)__builtin_amdgcn_readfirstlane((int)*n_items); n_it = n_it < cap_items ? n_it : cap_items; }   // (reads whose items did not fit are on `list`)
    const uint32_t n_reads = MODE == 2 ? n_it + (uint32_t)__builtin_amdgcn_readfirstlane((int)*n_list) : (uint32_t)ck.n_reads;
    unsigned long long p2m = 0;                    // MODE 2: this unit's pass-2 calls
    bool single = false;                           // MODE 2: one call of a read whose other calls run on other lanes: results are appended atomically
    // ---- wave-level pool of read indices: [pool_next, pool_end) in use, [res_next, res_end) in reserve
    uint32_t pool_next = 0, pool_end = 0, res_next = 0, res_end = 0;
    uint32_t pend_base = 0, taken = 0;
    bool pending = false, exhausted = false;
    // ---- per-lane state
    int phase = S4_FETCH;
    uint32_t r = 0;
    uint64_t o0 = 0, o1 = 0, qoff = 0;
    int len = 0, n_out = 0;
    bool out_ovf = false, list_ovf = false;
    int pass = 1, x = 0, k2 = 0, old_n = 0;
    int sx = 0, i = 0, n = 0, ret = 0, bi = 0, cb = 0, cur = 1, np = 0, rev = 0, j = 0, nc = 0, last_start = 0;
    I min_intv = 1, last_sz = 0;
    I ik0 = 0, ik1 = 0, ik2 = 0;                   // the forward interval (both strands); its end in the query is i
    I h0 = 0, h2 = 0; uint32_t hinf = 0;           // first entry of the backward row being read
    I nh0 = 0, nh2 = 0; uint32_t nhinf = 0;        // ... and of the row being written
    I p0 = 0, p2 = 0; uint32_t pinf = 0;           // the entry this backward step extends
    const int K = (fm.lut && fm.lut_k <= opt.min_seed_len) ? fm.lut_k : 0;   // k-mer table (see k_kmer_lut)
    const LutE<I> *lut = (const LutE<I> *)fm.lut;
    int nv = 0;                                    // virtual entries of this call: K - 1 after a jump through the table
    uint32_t vcode = 0;                            // K-mer at the backward row's position
    I v0 = 0, v2 = 0;                              // ... and its interval, loaded at the start of the row
    QWin win; win.bits = 0; win.chunk = 0xffffffffu;
    const bool direct_ok = fm.sa_dense != nullptr;  // (the sampled-SA walk would cost more than it saves)
    const bool bwd_direct = MODE == 1 && direct_ok && fm.bwd_direct != 0;          // backward steps of one-occurrence entries against the text (below)
    int ones = 0;                                  // consecutive forward steps that ended at one occurrence
    int64_t dpos = 0;                              // S4_DIR: text position of that occurrence of q[sx, i)
    RWin rwin; rwin.bits = 0; rwin.chunk = -1;
#if SEED4_DEBUG
    const unsigned long long d_t0 = __builtin_readcyclecounter();
    unsigned int d_call = 0;                      // backward steps of the call in hand
    auto dbg_call_end = [&]() {                    // [16 + 2b] calls, [17 + 2b] steps by bucket b: min_intv == 2 ? 0 : 1;  [20 + 2h], [21 + 2h] by length class h
        if (!(ck.dbg_cyc && ck.dbg_stage == 3) || !(SEED4_DEBUG_MODE == 0 || SEED4_DEBUG_MODE == MODE)) return;
        const int b = min_intv <= (I)2 ? 0 : 1;
        atomicAdd(ck.dbg_cyc + 16 + 2 * b, 1ull); atomicAdd(ck.dbg_cyc + 17 + 2 * b, (unsigned long long)d_call);
        const int h = d_call < 64 ? 0 : d_call < 256 ? 1 : d_call < 1024 ? 2 : 3;
        atomicAdd(ck.dbg_cyc + 20 + 2 * h, 1ull); atomicAdd(ck.dbg_cyc + 21 + 2 * h, (unsigned long long)d_call);
        atomicMax(ck.dbg_cyc + 30, (unsigned long long)d_call);
        d_call = 0;
    };
    unsigned long long d_fsteps = 0, d_flanes = 0, d_bsteps = 0, d_blanes = 0, d_rounds = 0, d_ev = 0, d_dsteps = 0, d_dlanes = 0, d_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
    auto push_fwd = [&]() {                        // the current forward interval, ending at i (x[1] is never read back)
        if (n < wl.cap) wl_put(1, n, ik0, ik2, (uint32_t)i); else list_ovf = true;
        ++n;
    };
    auto out_push = [&](int start, int end, I a, I s) {
        if (MODE == 2 && single) {                  // a slot of the read's interval array, shared with the lanes that run its other calls
            const uint32_t slot = atomicAdd(ck.intv_n + r, 1u);
            if (slot < (uint32_t)ck.cap_intv) {
                const size_t o = (size_t)r * ck.cap_intv + slot;
                ((qp_t *)ck.intv_info)[o] = QP_PACK(start, end); ((I *)ck.intv_x0)[o] = a; ((I *)ck.intv_x2)[o] = s;
            } else { atomicSub(ck.intv_n + r, 1u); out_ovf = true; }   // (the count never stays above the capacity: later kernels index with it)
        } else if (n_out < ck.cap_intv) {
            const size_t o = (size_t)r * ck.cap_intv + n_out;
            ((qp_t *)ck.intv_info)[o] = QP_PACK(start, end); ((I *)ck.intv_x0)[o] = a; ((I *)ck.intv_x2)[o] = s;
            ++n_out;
        } else out_ovf = true;
    };
    auto emit_mem = [&](I a, I s, uint32_t end) {  // a MEM [bi+1, end) ends the backward walk of an entry
        if (bi + 1 < last_start) {                 // not contained in a longer match already reported by this call
            last_start = bi + 1;
            if ((int)end - (bi + 1) >= opt.min_seed_len) out_push(bi + 1, (int)end, a, s);
        }
    };
    auto finish_read = [&]() {                     // passes 1 + 2 done (or abandoned on a work-list overflow)
        if (!(MODE == 2 && single)) ck.intv_n[r] = (uint32_t)n_out;
        if (out_ovf) atomicOr(ck.flags, OVF_INTV);
        if (list_ovf) atomicOr(ck.flags, OVF_LIST);
        phase = S4_FETCH;
    };
    auto begin_bwd = [&]() {                       // the forward phase is over: its last push is the longest extension
        if (list_ovf) { finish_read(); return; }
        if (MODE == 2 && single && cap_long && n >= SEED2C_MIN && n + K <= SEED2C_CAP) {
            // a re-seeding call inside a repeat: rows of dozens of entries (rows x entries = thousands of steps on this lane).  Its rows are
            // data-parallel: the call goes to k_seed2_coop, one wave per call, and this lane takes its next unit
            const uint32_t idx = wave_fetch_inc(n_long);
            if (idx < cap_long) { long_items[idx] = r << 6 | (uint32_t)(k2 - 1); phase = S4_START; return; }
        }
        ret = i;                                    // = info of the entry pushed last = next start of pass 1
        h0 = ik0; h2 = ik2; hinf = (uint32_t)i;
        cur = 1; np = n; rev = 1; last_start = 0x7fffffff; bi = sx - 1;
        phase = S4_ROW;
    };

    for (;;) {
        // ------------------------------------------------ events: read assignment, next bwt_smem1a call
        const unsigned long long m_ev = __ballot(phase == S4_FETCH || phase == S4_INIT || phase == S4_START);
#if SEED4_DEBUG
        ++d_rounds; if (m_ev) ++d_ev;
#endif
        if (m_ev) {
            if (pending) {
                const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend_base);
                pending = false;
                if (base >= n_reads) exhausted = true;
                else {
                    res_next = base; res_end = base + pool < n_reads ? base + pool : n_reads;
                    taken += pool;
                    if (wave_quota && taken >= wave_quota) exhausted = true;      // this wave's share: what is in hand is finished, nothing more is fetched
                }
            }
            if (pool_next == pool_end && res_next != res_end) { pool_next = res_next; pool_end = res_end; res_next = res_end = 0; }
            if (res_next == res_end && !exhausted) {
                if (lane == 0) pend_base = atomicAdd(queue, (unsigned int)pool);
                pending = true;
            }
            if (phase == S4_INIT) {                     // offsets requested one round ago
                qoff = o0; len = (int)(o1 - o0);
                out_ovf = false; list_ovf = false;
                if (MODE == 2) {                        // pass 2 appends to what pass 1 left
                    if (single) { k2 = (int)p2m; old_n = k2 + 1; p2m = 1ull << k2; n_out = 0; }      // (p2m held the interval number since the fetch)
                    else { n_out = (int)ck.intv_n[r]; p2m = p2mask[r]; k2 = 0; old_n = n_out; }
                    pass = 2; phase = S4_START;
                } else {
                    n_out = 0;
                    if (len < opt.min_seed_len) finish_read();
                    else { pass = 1; x = 0; phase = S4_START; }
                }
            }
            {
                const unsigned long long want = __ballot(phase == S4_FETCH);
                if (want) {
                    const uint32_t avail = pool_end - pool_next;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    if (phase == S4_FETCH) {
                        if (rank < avail) {
                            if (MODE == 2) {
                                const uint32_t u = pool_next + rank;
                                single = u < n_it;
                                uint32_t it = 0;
                                if (single) { it = items[u]; r = it >> 6; p2m = it & 63u; }
                                else r = (uint32_t)list[u - n_it];
                                if (it != 0xffffffffu) { o0 = ck.offs[r]; o1 = ck.offs[r + 1]; phase = S4_INIT; }     // (else: a slot left empty by k_seed2_select)
                            } else { r = pool_next + rank; o0 = ck.offs[r]; o1 = ck.offs[r + 1]; phase = S4_INIT; }
                        } else if (exhausted && !pending && res_next == res_end) phase = S4_DONE;
                    }
                    const uint32_t cnt = (uint32_t)__popcll(want);
                    pool_next += cnt < avail ? cnt : avail;
                }
            }
            if (phase == S4_START) {
                int xs = -1;
                I mi = 1;
                if (MODE == 1) {
                    while (x < len && qb(x) > 3) ++x;
                    if (x < len) xs = x;
                    else finish_read();                 // pass 1 is done; pass 2 is another launch (MODE 2)
                }
                if (MODE == 2 && pass == 2) {           // re-seed from the middle of long SMEMs with few occurrences
                    while (k2 < old_n) {
                        const int k = k2++;
                        if (k < 63 && !((p2m >> k) & 1ull)) continue;           // k_seed2_select: not a candidate, or the repeat filter rules a result out
                        if (k >= 63 && !(p2m >> 63)) { k2 = old_n; break; }
                        const size_t o = (size_t)r * ck.cap_intv + k;
                        const qp_t inf = ((const qp_t *)ck.intv_info)[o];
                        const I s = ((const I *)ck.intv_x2)[o];
                        const int start = QP_HI(inf), end = QP_LO(inf);
                        const int mid = (start + end) >> 1;
                        if (k >= 63) {                  // beyond the mask: bwa's own conditions here
                            if (end - start < split_len || s > (I)opt.split_width) continue;
                            if (qb(mid) > 3) continue;  // bwt_smem1a returns at once on an ambiguous start
                        }
                        xs = mid; mi = s + 1;
                        break;
                    }
                    if (xs < 0) finish_read();
                }
                if (xs >= 0) {
                    sx = xs; min_intv = mi < 1 ? (I)1 : mi;
                    nv = 0;
                    if (K && xs + K <= len) {
                        uint32_t fw, rc;
                        if (kmer_codes(qb, xs, K, fw, rc) < 0) {
                            const LutE<I> e = lut[fw];
                            if (e.sz >= min_intv) { ik0 = e.lo; ik2 = e.sz; ik1 = lut[rc].lo; i = xs + K; nv = K - 1; vcode = fw; }
                        }
                    }
                    if (!nv) { set_intv4<I>(fm, qb(xs), ik0, ik1, ik2); i = xs + 1; }
                    n = 0; ones = 0;
                    phase = S4_FWD;
                }
            }
        }
        if (__all(phase == S4_DONE)) break;
#if SEED4_DEBUG
        for (int ph = 0; ph < 8; ++ph) d_ph[ph] += (unsigned long long)__popcll(__ballot(phase == ph));
#endif
        const int nF = __popcll(__ballot(phase == S4_FWD)), nB = __popcll(__ballot(phase == S4_ROW || phase == S4_BWD));
        const int nD = __popcll(__ballot(phase == S4_DIR));
        if (nF == 0 && nB == 0 && nD == 0) continue;
#if SEED4_SCHED == 1
        // batches: the short loops (forward, direct) are entered when SEED4_BATCH lanes wait for them (or nothing else can run) and run until
        // nearly all of their lanes are through; the backward loop -- where a read spends most of its steps -- keeps going until that many
        // lanes are ready for something else
        const bool goD = nD >= SEED4_BATCH || (nD && nD >= nB && nD >= nF);
        const bool goF = !goD && (nF >= SEED4_BATCH || (nF && nF >= nB));
#else
        const bool goD = nD >= nF && nD >= nB;
        const bool goF = !goD && nF >= nB;
#endif
        if (goD) {
            // ------------------------------------------------ D loop: up to 32 bases against the reference text per step
            const int floor_ = nD >> SEED4_SHORT_FLOOR_SHIFT;
            for (int it = 0; it < SEED4_ITERS; ++it) {
                if (__popcll(__ballot(phase == S4_DIR)) <= floor_ && it) break;
#if SEED4_DEBUG
                ++d_dsteps; d_dlanes += (unsigned long long)__popcll(__ballot(phase == S4_DIR));
#endif
                if (phase == S4_DIR) {
                    for (int t = 0; t < 32; ++t) {
                        int cq = 4;
                        const int64_t tp = dpos + (int64_t)(i - sx);
                        // thirty-two bases at once where the text side starts a chunk of pac (a step of the loop's budget each: up to 1 024 bases per turn)
                        if (i + 32 <= len && dir32_match(R, ck.codes, qoff + (uint64_t)i, tp, rwin)) { i += 32; continue; }
                        // four bases at once where the text side starts a byte of pac: a contig's direct walk is hundreds of thousands of bases on one lane
                        if (t + 4 <= 32 && i + 4 <= len && dir4_match(R, ck.codes, qoff + (uint64_t)i, tp, rwin)) { i += 4; t += 3; continue; }
                        if (i < len && (cq = qb(i)) <= 3 && tp < (R.l_pac << 1) && text_at(R, tp, rwin) == cq) { ++i; continue; }
                        // end of the read, an ambiguous base, or the occurrence ends here (mismatch / end of the text): bwt_smem1a pushes the
                        // interval and stops -- in the last two cases because the extended interval is empty
                        push_fwd(); begin_bwd();
                        break;
                    }
                }
            }
        } else if (goF) {
            // ------------------------------------------------ F loop: one forward extend per step
            const int floor_ = nF >> SEED4_SHORT_FLOOR_SHIFT;           // leave when that share of the lanes that entered is left (the others wait in other phases)
            for (int it = 0; it < SEED4_ITERS; ++it) {
                bool act = phase == S4_FWD;
                int cq = 4;
                if (act) {
                    if (i >= len || (cq = qb(i)) > 3) { push_fwd(); begin_bwd(); act = false; }
                }
                if (__popcll(__ballot(act)) <= floor_ && it) break;
#if SEED4_DEBUG
                ++d_fsteps; d_flanes += (unsigned long long)__popcll(__ballot(act));
#endif
                if (act) {
                    I o0_, o1, o2;
                    fwd_step<I>(fm, ik0, ik1, ik2, cq, o0_, o1, o2);
                    bool stop = false;
                    if (o2 != ik2) {
                        push_fwd();
                        if (o2 < min_intv) stop = true;
                    }
                    if (stop) begin_bwd();
                    else {
                        ik0 = o0_; ik1 = o1; ik2 = o2; ++i;
                        ones = ik2 == 1 ? ones + 1 : 0;
                        if (direct_ok && ones >= SEED4_DIRECT_AFTER) { dpos = fm_sa<I>(fm, ik0); phase = S4_DIR; }
                    }
                }
            }
        } else {
            // ------------------------------------------------ B loop: one work-list entry against the row's base per step
            const int floor_ = nB >> SEED4_FLOOR_SHIFT; (void)floor_;
            for (int it = 0; it < SEED4_ITERS_B; ++it) {
                if (phase == S4_ROW) {                  // start of the backward row at query position bi; its first entry is in registers
                    int t = -1;
                    if (bi >= 0) { t = qb(bi); if (t > 3) t = -1; }
                    cb = t;
                    j = 0; nc = 0; last_sz = 0;
                    p0 = h0; p2 = h2; pinf = hinf;
                    if (cb < 0) {                       // beginning of the read or an ambiguous base: every entry ends here, only the first can be new
                        if (np > 0) emit_mem(p0, p2, pinf);   // (np == 0: only virtual entries are left, too short to report)
                        if (MODE == 1) x = ret;
                        phase = S4_START;
#if SEED4_DEBUG
                        dbg_call_end();
#endif
                    } else {
                        if (sx - bi <= nv) {            // this row's virtual entry: the K-mer starting at bi
                            vcode = (vcode >> 2) | ((uint32_t)cb << (2 * (K - 1)));
                            const LutE<I> e = lut[vcode];
                            v0 = e.lo; v2 = e.sz;
                        }
                        phase = S4_BWD;
                    }
                }
                const bool act = phase == S4_BWD;
#if SEED4_SCHED == 1
                {
                    const int n_fwd = __popcll(__ballot(phase == S4_START || phase == S4_INIT || phase == S4_FWD || (phase == S4_FETCH && pool_next != pool_end)));
                    const int n_dir = __popcll(__ballot(phase == S4_DIR));
                    if (it && (!__any(act) || n_fwd >= SEED4_BATCH || n_dir >= SEED4_BATCH)) break;
                }
#else
                if (__popcll(__ballot(act)) <= floor_ && it) break;
#endif
#if SEED4_DEBUG
                ++d_bsteps; d_blanes += (unsigned long long)__popcll(__ballot(act));
                if (act) ++d_call;
#endif
                if (act) {
                    I nx0 = 0, nx2 = 0; uint32_t nxinf = 0;
                    if (j + 1 < np) wl_get(cur, rev ? np - 2 - j : j + 1, nx0, nx2, nxinf);   // the next entry, in flight behind this step's rank reads
                    if (j < np) {                       // (np == 0: a row of the virtual entry alone)
                        I o2, o0_;
                        if (bwd_direct && p2 == (I)1) {
                            // ONE occurrence: the extension by cb exists iff the text has cb before it -- one (mostly cached) text read instead of two rank reads.
                            // The entry carries its text position from here on (flag bit in x0); its rank is never needed again: an emitted interval
                            // is only ever turned into positions (intv_pos).
                            if (!(p0 & pos_flag<I>())) p0 = (I)fm_sa<I>(fm, p0) | pos_flag<I>();
                            const int64_t pos = (int64_t)(p0 & ~pos_flag<I>());
                            o2 = (pos > 0 && text_at(R, pos - 1, rwin) == cb) ? (I)1 : (I)0;
                            o0_ = (I)(pos - 1) | pos_flag<I>();
                        } else {
                        RankLd<I> rk, rl;
                        rank_issue<I>(fm, p0 - 1, rk); rank_issue<I>(fm, p0 - 1 + p2, rl);
                        I ckc, clc, dummy;
                        rank_finish<I, false>(rk, cb, ckc, dummy); rank_finish<I, false>(rl, cb, clc, dummy);
                        o2 = clc - ckc;
                        o0_ = l2_of<I>(fm, cb) + 1 + ckc;
                        }
                        if (o2 < min_intv) { if (nc == 0) emit_mem(p0, p2, pinf); }
                        else if (nc == 0 || o2 != last_sz) {
                            if (nc == 0) { nh0 = o0_; nh2 = o2; nhinf = pinf; }
                            wl_put(1 - cur, nc++, o0_, o2, pinf);
                            last_sz = o2;
                        }
                    }
                    ++j;
                    if (j >= np) {
                        const int t = sx - bi;          // rows done with this one
                        if (t <= nv && v2 >= min_intv && (nc == 0 || v2 != last_sz)) {   // the virtual entry [bi, bi + K) joins as the row's last
                            if (nc < wl.cap) {
                                if (nc == 0) { nh0 = v0; nh2 = v2; nhinf = (uint32_t)(bi + K); }
                                wl_put(1 - cur, nc++, v0, v2, (uint32_t)(bi + K));
                            } else list_ovf = true;
                        }
                        if (list_ovf) finish_read();
                        else if (nc == 0 && t >= nv) {
                            if (MODE == 1) x = ret;
                            phase = S4_START;
#if SEED4_DEBUG
                            dbg_call_end();
#endif
                        }
                        else { cur = 1 - cur; np = nc; rev = 0; --bi; h0 = nh0; h2 = nh2; hinf = nhinf; phase = S4_ROW; }
                    } else { p0 = nx0; p2 = nx2; pinf = nxinf; }
                }
            }
        }
        }
#if SEED4_DEBUG
    if (ck.dbg_cyc && ck.dbg_stage == 3 && lane == 0 && (SEED4_DEBUG_MODE == 0 || SEED4_DEBUG_MODE == MODE)) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        ck.dbg_cyc[64 + (size_t)blockIdx.x * 2 + (threadIdx.x >> 6)] = t_end - d_t0;   // this wave's lifetime
        atomicMax(ck.dbg_cyc + 31, t_end - d_t0);
        for (int ph = 0; ph < 8; ++ph) atomicAdd(ck.dbg_cyc + 8 + ph, d_ph[ph]);
        atomicAdd(ck.dbg_cyc + 0, d_fsteps); atomicAdd(ck.dbg_cyc + 1, d_flanes); atomicAdd(ck.dbg_cyc + 2, d_bsteps); atomicAdd(ck.dbg_cyc + 3, d_blanes);
        atomicAdd(ck.dbg_cyc + 4, d_rounds); atomicAdd(ck.dbg_cyc + 5, d_ev); atomicAdd(ck.dbg_cyc + 6, d_dsteps); atomicAdd(ck.dbg_cyc + 7, d_dlanes);
    }
#endif
}

// Pass-2 calls inside repeats, one WAVE per call (the items k_seed12m<2> put aside: >= SEED2C_MIN entries after the forward phase).  The
// forward phase is run again, every lane the same steps (a chain of dependent rank reads either way); then a backward row is one wave
// step per 64 entries: every lane extends its entry by the row's base, and what bwt_smem1a does with the results in list order is
//     entry 0 below min_intv             -> its match ends here: the call's next MEM (later failing entries start at the same position)
//     entry j kept                       <=> size >= min_intv and (entry j-1 failed or has a different size)
// because the entries of a row are nested intervals in order of growing size: failing entries are a prefix of the list, equal sizes are
// neighbours, and the size of the last KEPT entry is the size of entry j-1 whenever entry j-1 did not fail.  Kept entries are compacted
// into the next row's list in LDS by a ballot prefix count.  The k-mer table's virtual entries join as in k_seed12m.
template <typename I>
__global__ void __launch_bounds__(64) k_seed2_coop(DevFM<I> fm, DevRef R, Chunk ck, DevOpt dopt, unsigned int *queue, const uint32_t *items, const unsigned int *n_items, uint32_t cap_items)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    __shared__ I s_x0[2][SEED2C_CAP], s_x2[2][SEED2C_CAP];
    __shared__ uint32_t s_inf[2][SEED2C_CAP];
    uint32_t n_it = (uint32_t)__builtin_amdgcn_readfirstlane((int)*n_items);
    n_it = n_it < cap_items ? n_it : cap_items;
    const int K = (fm.lut && fm.lut_k <= opt.min_seed_len) ? fm.lut_k : 0;
    const LutE<I> *lut = (const LutE<I> *)fm.lut;
    for (;;) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(queue, 1u);
        u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= n_it) break;
        const uint32_t it = items[u];
        const uint32_t r = it >> 6;
        const uint64_t qoff = ck.offs[r];
        const int len = (int)(ck.offs[r + 1] - qoff);
        const size_t po = (size_t)r * ck.cap_intv + (it & 63u);
        const qp_t pinf0 = ((const qp_t *)ck.intv_info)[po];
        const int sx = (QP_HI(pinf0) + QP_LO(pinf0)) >> 1;
        const I min_intv = ((const I *)ck.intv_x2)[po] + 1;
        QWin win; win.bits = 0; win.chunk = 0xffffffffu;
        auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
        // ---- forward phase (bwt_smem1a's first loop), the same on every lane; lane 0 keeps the list
        I ik0, ik1, ik2;
        int i, nv = 0, n = 0;
        uint32_t vcode = 0;
        bool list_ovf = false;
        if (K && sx + K <= len) {
            uint32_t fw, rc;
            if (kmer_codes(qb, sx, K, fw, rc) < 0) {
                const LutE<I> e = lut[fw];
                if (e.sz >= min_intv) { ik0 = e.lo; ik2 = e.sz; ik1 = lut[rc].lo; i = sx + K; nv = K - 1; vcode = fw; }
            }
        }
        if (!nv) { set_intv4<I>(fm, qb(sx), ik0, ik1, ik2); i = sx + 1; }
        auto push_fwd = [&]() {
            if (n < ck.cap_list && n < SEED2C_CAP) { if (lane == 0) { s_x0[1][n] = ik0; s_x2[1][n] = ik2; s_inf[1][n] = (uint32_t)i; } }
            else list_ovf = true;
            ++n;
        };
        for (;;) {
            int cq = 4;
            if (i >= len || (cq = qb(i)) > 3) { push_fwd(); break; }
            I o0_, o1, o2;
            fwd_step<I>(fm, ik0, ik1, ik2, cq, o0_, o1, o2);
            if (o2 != ik2) { push_fwd(); if (o2 < min_intv) break; }
            ik0 = o0_; ik1 = o1; ik2 = o2; ++i;          // (min_intv >= 2 here: an interval never stays at one occurrence, no direct steps)
        }
        __syncthreads();
        if (list_ovf) { if (lane == 0) atomicOr(ck.flags, OVF_LIST); continue; }
        // ---- backward rows
        int cur = 1, np = n, rev = 1, last_start = 0x7fffffff, bi = sx - 1;
        bool out_ovf = false;
        auto emit_mem = [&](I a, I sz, uint32_t end) {      // lane 0
            if (bi + 1 < last_start) {
                last_start = bi + 1;
                if ((int)end - (bi + 1) >= opt.min_seed_len) {
                    const uint32_t slot = atomicAdd(ck.intv_n + r, 1u);
                    if (slot < (uint32_t)ck.cap_intv) {
                        const size_t o = (size_t)r * ck.cap_intv + slot;
                        ((qp_t *)ck.intv_info)[o] = QP_PACK(bi + 1, end); ((I *)ck.intv_x0)[o] = a; ((I *)ck.intv_x2)[o] = sz;
                    } else { atomicSub(ck.intv_n + r, 1u); out_ovf = true; }
                }
            }
        };
        for (;;) {
            int cb = -1;
            if (bi >= 0) { cb = qb(bi); if (cb > 3) cb = -1; }
            if (cb < 0) {                                   // beginning of the read or an ambiguous base: every entry ends here, only the first can be new
                if (np > 0 && lane == 0) { const int h = rev ? np - 1 : 0; emit_mem(s_x0[cur][h], s_x2[cur][h], s_inf[cur][h]); }
                break;
            }
            const int t = sx - bi;                          // rows done with this one
            I v0 = 0, v2 = 0;
            if (t <= nv) {
                vcode = (vcode >> 2) | ((uint32_t)cb << (2 * (K - 1)));
                const LutE<I> e = lut[vcode];
                v0 = e.lo; v2 = e.sz;
            }
            int nc = 0;
            I last_sz = 0;                                  // size of the last entry that did not fail
            bool prev_ok = false;                           // ... and whether the entry before this chunk did not fail
            for (int c0 = 0; c0 < np; c0 += 64) {
                const int j = c0 + lane;
                const bool live = j < np;
                I p0 = 0, p2 = 0, o2 = 0, o0_ = 0; uint32_t pinf = 0;
                if (live) {
                    const int idx = rev ? np - 1 - j : j;
                    p0 = s_x0[cur][idx]; p2 = s_x2[cur][idx]; pinf = s_inf[cur][idx];
                    RankLd<I> rk, rl;
                    rank_issue<I>(fm, p0 - 1, rk); rank_issue<I>(fm, p0 - 1 + p2, rl);
                    I ckc, clc, dummy;
                    rank_finish<I, false>(rk, cb, ckc, dummy); rank_finish<I, false>(rl, cb, clc, dummy);
                    o2 = clc - ckc;
                    o0_ = l2_of<I>(fm, cb) + 1 + ckc;
                }
                const bool ok = live && o2 >= min_intv;
                if (c0 == 0 && lane == 0 && !ok) emit_mem(p0, p2, pinf);      // (np > 0 here, so lane 0 is live)
                // the entry before mine: lane - 1, or the last entry of the previous chunk
                I left_o2 = (I)__shfl_up((long long)o2, 1, 64);
                int left_ok = __shfl_up((int)ok, 1, 64);
                if (lane == 0) { left_o2 = last_sz; left_ok = prev_ok; }
                const bool kept = ok && (!left_ok || o2 != left_o2);
                const unsigned long long mk = __ballot(kept), mo = __ballot(ok);
                if (kept) {
                    const int pos = nc + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                    s_x0[1 - cur][pos] = o0_; s_x2[1 - cur][pos] = o2; s_inf[1 - cur][pos] = pinf;
                }
                nc += __popcll(mk);
                const int last_lane = np - c0 < 64 ? np - c0 - 1 : 63;
                last_sz = (I)__shfl((long long)o2, last_lane, 64);
                prev_ok = ((mo >> last_lane) & 1ull) != 0;
            }
            // the row's virtual entry [bi, bi + K) joins as its last (k_seed12m); last_sz only matters when an entry was kept, and then the
            // list's last entry did not fail
            if (t <= nv && v2 >= min_intv && (nc == 0 || v2 != last_sz)) {
                if (nc < ck.cap_list) { if (lane == 0) { s_x0[1 - cur][nc] = v0; s_x2[1 - cur][nc] = v2; s_inf[1 - cur][nc] = (uint32_t)(bi + K); } ++nc; }
                else list_ovf = true;
            }
            __syncthreads();
            if (list_ovf) break;
            if (nc == 0 && t >= nv) break;
            cur = 1 - cur; np = nc; rev = 0; --bi;
        }
        if (lane == 0) {
            if (list_ovf) atomicOr(ck.flags, OVF_LIST);
            if (out_ovf) atomicOr(ck.flags, OVF_INTV);
        }
        __syncthreads();
    }
}

// pass 3 (bwt_seed_strategy1: forward-only LAST-like seeds); the per-read epilogue follows in k_seed_epi.  One lane per read, one forward extend per base: lanes of a wave run
// in lock step for reads of equal length, with the forward step of this file (two masked popcounts per block instead of all four
// symbol counts).
template <typename I>
__global__ void __launch_bounds__(128) k_seed3m(DevFM<I> fm, DevRef R, Chunk ck, DevOpt dopt, int par_min_len = 0x7fffffff)
{
    const slx_opt &opt = dopt.o;
    const int t_ = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = ck.spread > 1 ? t_ / ck.spread : t_;          // (a small chunk: one read per wave, on its lane 0)
    bool live = r < ck.n_reads && (ck.spread <= 1 || t_ % ck.spread == 0);
    const uint64_t qoff = live ? ck.offs[r] : 0;
    int len = live ? (int)(ck.offs[r + 1] - qoff) : 0;
    if (len >= par_min_len) { live = false; len = 0; }          // a contig: k_seed3_next + k_seed3_chase below
    const size_t ob = (size_t)(live ? r : 0) * ck.cap_intv;
    qp_t *oinfo = (qp_t *)ck.intv_info + ob;
    I *ox0 = (I *)ck.intv_x0 + ob, *ox2 = (I *)ck.intv_x2 + ob;
    int n_out = live ? (int)ck.intv_n[r] : 0;
    bool out_ovf = false;
    QWin win; win.bits = 0; win.chunk = 0xffffffffu;
    auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
    if (opt.max_mem_intv > 0) {                      // bwt_seed_strategy1 from every position a seed ended at
        int x = len >= opt.min_seed_len ? 0 : len, i = 0;
        bool fresh = true, dir = false;                 // dir: the interval is one occurrence at text position dpos (see the D loop of k_seed12m)
        int64_t dpos = 0;
        RWin rwin; rwin.bits = 0; rwin.chunk = -1;
        const bool direct_ok = fm.sa_dense != nullptr;
        I k0 = 0, k1 = 0, k2 = 0;
        const int K = (fm.lut && fm.lut_k <= opt.min_seed_len) ? fm.lut_k : 0;   // k-mer table (see k_kmer_lut)
        const LutE<I> *lut = (const LutE<I> *)fm.lut;
        for (;;) {
            if (fresh) {
                while (x < len) {
                    if (qb(x) > 3) { ++x; continue; }
                    if (!K) { set_intv4<I>(fm, qb(x), k0, k1, k2); i = x + 1; fresh = false; break; }
                    // no seed is reported before it is min_seed_len >= K bases long: the first K - 1 steps only move the interval
                    if (x + K > len) { x = len; break; }                // fewer than K bases left: nothing more to report
                    uint32_t fw, rc;
                    const int bad = kmer_codes(qb, x, K, fw, rc);
                    if (bad >= 0) { x += bad + 1; continue; }           // an ambiguous base restarts the search behind it
                    const LutE<I> e = lut[fw];
                    k0 = e.lo; k2 = e.sz; k1 = lut[rc].lo; i = x + K; fresh = false;
                    break;
                }
            }
            const bool act = !fresh && i < len;
            if (!__any(act)) break;
            if (act) {
                const int cq = qb(i);
                if (cq > 3) { x = i + 1; fresh = true; dir = false; }
                else {
                    I o0 = k0, o1 = k1, o2 = 0;
                    if (dir) {                          // one occurrence: kept while the text goes on like the read
                        const int64_t tp = dpos + (int64_t)(i - x);
                        o2 = (k2 != 0 && tp < (R.l_pac << 1) && text_at(R, tp, rwin) == cq) ? (I)1 : (I)0;
                    } else if (k2 != 0) fwd_step<I>(fm, k0, k1, k2, cq, o0, o1, o2);      // (an empty interval stays empty: bwa walks on to min_seed_len)
                    if (o2 < (I)opt.max_mem_intv && i - x >= opt.min_seed_len) {
                        if (o2 > 0) {
                            if (n_out < ck.cap_intv) { oinfo[n_out] = QP_PACK(x, i + 1); ox0[n_out] = o0; ox2[n_out] = o2; ++n_out; }
                            else out_ovf = true;
                        }
                        x = i + 1; fresh = true; dir = false;
                    } else {
                        k0 = o0; k1 = o1; k2 = o2; ++i;
                        if (!dir && direct_ok && k2 == 1 && x + opt.min_seed_len - i >= 2) { dpos = fm_sa<I>(fm, k0); dir = true; }
                    }
                }
            }
        }
    }
    if (!live) return;
    ck.intv_n[r] = (uint32_t)n_out;
    if (out_ovf) atomicOr(ck.flags, OVF_INTV);
}

// Pass 3 for contigs.  bwt_seed_strategy1 is restarted where the previous seed ended, so the walk over a read is a chain x -> next(x) --
// ten thousand dependent walks for a 300 kb contig on its one lane (240 ms).  But what a walk from x does (where it ends, what it reports)
// depends on x alone: k_seed3_next runs the walk from EVERY position of the long reads, one lane per position (the same state machine as
// k_seed3m, started fresh at x), and k_seed3_chase follows the chain from 0 through the stored results, one lane per read, appending the
// seeds of the positions it visits.  ~25 x the rank queries, all of them independent.  (The order the seeds are appended in does not
// matter: k_seed_epi sorts them.)
template <typename I>
struct Seed3Next { int next, end; I x0, x2; };      // x2 = 0: nothing reported from this start

template <typename I>
__global__ void __launch_bounds__(256) k_seed3_next(DevFM<I> fm, DevRef R, Chunk ck, DevOpt dopt, int par_min_len, uint64_t n_bases, Seed3Next<I> *out)
{
    const slx_opt &opt = dopt.o;
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_bases || opt.max_mem_intv <= 0) return;
    int lo = 0, hi = ck.n_reads;                        // the read holding base p
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ck.offs[mid] <= p) lo = mid; else hi = mid; }
    const int r = lo;
    const uint64_t qoff = ck.offs[r];
    const int len = (int)(ck.offs[r + 1] - qoff);
    if (len < par_min_len) return;
    const int x0s = (int)(p - qoff);
    QWin win; win.bits = 0; win.chunk = 0xffffffffu;
    auto qb = [&](int q) { return q_at(ck.codes, qoff + (uint64_t)q, win); };
    RWin rwin; rwin.bits = 0; rwin.chunk = -1;
    const bool direct_ok = fm.sa_dense != nullptr;
    const int K = (fm.lut && fm.lut_k <= opt.min_seed_len) ? fm.lut_k : 0;
    const LutE<I> *lut = (const LutE<I> *)fm.lut;
    Seed3Next<I> res; res.next = len; res.end = 0; res.x0 = 0; res.x2 = 0;
    // ---- the entry of a fresh walk at x (k_seed3m's `fresh` block, one position only: where it moves x, that is this start's next)
    int x = x0s, i = 0;
    I k0 = 0, k1 = 0, k2 = 0;
    bool started = false;
    if (qb(x) > 3) res.next = x + 1;
    else if (!K) { set_intv4<I>(fm, qb(x), k0, k1, k2); i = x + 1; started = true; }
    else if (x + K > len) res.next = len;
    else {
        uint32_t fw, rc;
        const int bad = kmer_codes(qb, x, K, fw, rc);
        if (bad >= 0) res.next = x + bad + 1;
        else { const LutE<I> e = lut[fw]; k0 = e.lo; k2 = e.sz; k1 = lut[rc].lo; i = x + K; started = true; }
    }
    // ---- the walk
    bool dir = false;
    int64_t dpos = 0;
    while (started) {
        if (i >= len) { res.next = len; break; }            // the read ends inside the walk: nothing reported
        const int cq = qb(i);
        if (cq > 3) { res.next = i + 1; break; }
        I o0 = k0, o1 = k1, o2 = 0;
        if (dir) {
            const int64_t tp = dpos + (int64_t)(i - x);
            o2 = (k2 != 0 && tp < (R.l_pac << 1) && text_at(R, tp, rwin) == cq) ? (I)1 : (I)0;
        } else if (k2 != 0) fwd_step<I>(fm, k0, k1, k2, cq, o0, o1, o2);
        if (o2 < (I)opt.max_mem_intv && i - x >= opt.min_seed_len) {
            if (o2 > 0) { res.end = i + 1; res.x0 = o0; res.x2 = o2; }
            res.next = i + 1;
            break;
        }
        k0 = o0; k1 = o1; k2 = o2; ++i;
        if (!dir && direct_ok && k2 == 1 && x + opt.min_seed_len - i >= 2) { dpos = fm_sa<I>(fm, k0); dir = true; }
    }
    out[p] = res;
}

template <typename I>
__global__ void __launch_bounds__(64) k_seed3_chase(Chunk ck, DevOpt dopt, int par_min_len, const Seed3Next<I> *nx)
{
    const slx_opt &opt = dopt.o;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ck.n_reads || opt.max_mem_intv <= 0) return;
    const uint64_t qoff = ck.offs[r];
    const int len = (int)(ck.offs[r + 1] - qoff);
    if (len < par_min_len) return;
    const size_t ob = (size_t)r * ck.cap_intv;
    qp_t *oinfo = (qp_t *)ck.intv_info + ob;
    I *ox0 = (I *)ck.intv_x0 + ob, *ox2 = (I *)ck.intv_x2 + ob;
    int n_out = (int)ck.intv_n[r];
    bool out_ovf = false;
    int x = len >= opt.min_seed_len ? 0 : len;
    while (x < len) {
        const Seed3Next<I> e = nx[qoff + (uint64_t)x];
        if (e.x2 > 0) {
            if (n_out < ck.cap_intv) { oinfo[n_out] = QP_PACK(x, e.end); ox0[n_out] = e.x0; ox2[n_out] = e.x2; ++n_out; }
            else out_ovf = true;
        }
        x = e.next;
    }
    ck.intv_n[r] = (uint32_t)n_out;
    if (out_ovf) atomicOr(ck.flags, OVF_INTV);
}

// per-read epilogue of mem_collect_intv / prologue of mem_chain: sort the kept intervals by (start, end) -- entries with equal keys are
// identical intervals, so any exact sort matches ks_introsort -- then the repetitive length and the number of seed occurrences to look
// up.  One lane per read.  A read's ~13 intervals sit in its own 3 x 160-byte slice of the interval arrays, so an insertion sort in
// place is a chain of dependent, uncoalesced loads and read-modify-write stores (it was more than half of k_seed3m's time and 28x write
// amplification); here the slice is loaded once (independent loads), sorted in LDS (entry e of lane l at e * 64 + l: conflict-free) and
// written back once.  Reads with more than SEED_EPI_N intervals (repeats) sort in place as before.
#ifndef SEED_EPI_N
#define SEED_EPI_N 20
#endif
template <typename I>
__global__ void __launch_bounds__(64) k_seed_epi(Chunk ck, DevOpt dopt, unsigned int *max_cnt, unsigned int heavy_thr, unsigned int *n_heavy)
{
    const slx_opt &opt = dopt.o;
    __shared__ qp_t s_info[SEED_EPI_N * 64];
    __shared__ I s_x0[SEED_EPI_N * 64], s_x2[SEED_EPI_N * 64];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * 64 + lane;
    if (r >= ck.n_reads) return;
    const size_t ob = (size_t)r * ck.cap_intv;
    qp_t *oinfo = (qp_t *)ck.intv_info + ob;
    I *ox0 = (I *)ck.intv_x0 + ob, *ox2 = (I *)ck.intv_x2 + ob;
    const int n_out = (int)ck.intv_n[r];
    int b = 0, e = 0, l_rep = 0;
    uint32_t cnt = 0;
    auto account = [&](qp_t inf, I s) {          // mem_chain prologue, entries in sorted order
        if (s > (I)opt.max_occ) {
            const int sb = QP_HI(inf), se = QP_LO(inf);
            if (sb > e) { l_rep += e - b; b = sb; e = se; }
            else e = e > se ? e : se;
            const I step = s / (I)opt.max_occ;
            const I cc = (s + step - 1) / step;
            cnt += (uint32_t)(cc < (I)opt.max_occ ? cc : (I)opt.max_occ);
        } else cnt += (uint32_t)s;
    };
    if (n_out <= SEED_EPI_N) {
        for (int a = 0; a < n_out; a += 4) {         // four entries' loads in flight at a time
            qp_t ki[4]; I v0[4], v2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (a + u < n_out) { ki[u] = oinfo[a + u]; v0[u] = ox0[a + u]; v2[u] = ox2[a + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (a + u < n_out) { s_info[(a + u) * 64 + lane] = ki[u]; s_x0[(a + u) * 64 + lane] = v0[u]; s_x2[(a + u) * 64 + lane] = v2[u]; }
        }
        bool moved = false;
        for (int a = 1; a < n_out; ++a) {
            const qp_t ki = s_info[a * 64 + lane]; const I k0 = s_x0[a * 64 + lane], k2 = s_x2[a * 64 + lane];
            int p = a - 1;
            while (p >= 0 && s_info[p * 64 + lane] > ki) {
                s_info[(p + 1) * 64 + lane] = s_info[p * 64 + lane]; s_x0[(p + 1) * 64 + lane] = s_x0[p * 64 + lane]; s_x2[(p + 1) * 64 + lane] = s_x2[p * 64 + lane];
                --p;
            }
            if (p + 1 != a) { s_info[(p + 1) * 64 + lane] = ki; s_x0[(p + 1) * 64 + lane] = k0; s_x2[(p + 1) * 64 + lane] = k2; moved = true; }
        }
        for (int a = 0; a < n_out; ++a) {
            const qp_t ki = s_info[a * 64 + lane]; const I k0 = s_x0[a * 64 + lane], k2 = s_x2[a * 64 + lane];
            if (moved) { oinfo[a] = ki; ox0[a] = k0; ox2[a] = k2; }
            account(ki, k2);
        }
    } else {
        for (int a = 1; a < n_out; ++a) {
            const qp_t ki = oinfo[a]; const I s0 = ox0[a], s2 = ox2[a];
            int p = a - 1;
            while (p >= 0 && oinfo[p] > ki) { oinfo[p + 1] = oinfo[p]; ox0[p + 1] = ox0[p]; ox2[p + 1] = ox2[p]; --p; }
            oinfo[p + 1] = ki; ox0[p + 1] = s0; ox2[p + 1] = s2;
        }
        for (int k = 0; k < n_out; ++k) account(oinfo[k], ox2[k]);
    }
    l_rep += e - b;
    ck.l_rep[r] = l_rep;
    ck.seed_cnt[r] = (unsigned long long)cnt;
    if (cnt > 512u) atomicMax(max_cnt, cnt);      // the host skips the big-table launches of later stages when no read can need them
    // how many reads the heavy list will hold: the host sizes the sort of that list (and the scans over it) by it instead of by the chunk
    if (cnt >= heavy_thr) wave_fetch_inc(n_heavy);
}
