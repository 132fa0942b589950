// dev_seed3.h -- production SMEM seeding, multi-slot: bwa's mem_collect_intv (reached from
// /root/reference/src/BWAAligner.cpp:104 -> mem_align1 -> mem_chain) with up to THREE independent bwt_extend per lane and trip.
//
// Measured on this machine (scripts/ubench/rand32.hip): the memory system serves 220-370 G dependent random 32-byte reads per
// second from an index that fits L2 / Infinity Cache (49 G from HBM), and it is saturated by two waves per SIMD with one read in
// flight per lane.  The one-extend-per-trip state machine of dev_seed2.h reaches a quarter of that: a lane spends its trips on
// bookkeeping, not on loads.  So this kernel puts more rank queries behind every wait:
//   slot A   the smem machine of passes 1 + 2 (bwt_smem1a): one forward extend, or the first entry of a backward row
//   slot B   the second entry of the same backward row -- the entries of a row are independent (most rows hold 5-16 of them)
//   slot C   pass 3 (bwt_seed_strategy1), which depends on nothing but the read: it runs beside passes 1 + 2 instead of after them
// Every trip issues the rank reads of all active slots, waits once, and applies the results in the order the scalar code would.
// Other changes that follow from looking at what each consumer needs:
//   * a backward extension never reads x[1] of its interval and no MEM ever reports it, so work-list entries are (x0, x2, end) --
//     12 bytes instead of 16 -- and the backward step computes the rank of ONE symbol at k and l (two masked popcounts per block);
//   * the forward step needs the rank of its symbol and of "all greater symbols" (for the other strand's start), again two masked
//     popcounts per block instead of all four counts.
// Pass 3's intervals are collected from the top of the read's slot region downwards and joined to the others when the read is
// done; k_seed_fin sorts the set by (start, end) and derives the repetitive length and the seed-occurrence bound (mem_chain's
// prologue), as k_seed3 did.
#pragma once
#include "dev_seed2.h"

template <typename I>
struct LEnt { I x0, x2; uint32_t info; };        // one backward-phase interval: 12 bytes for u32 indexes

template <typename I>
struct WorkLists3 {
    LEnt<I> *base;            // already offset by the lane's slot
    size_t stride;            // n_threads
    int cap;
    __device__ __forceinline__ LEnt<I> &at(int list, int e) { return base[((size_t)list * cap + e) * stride]; }
};

// one rank block in flight: the two 16-byte halves of a 32-byte occ-plane block (+ the super-block row of a u64 index).
// Plain scalar members on purpose: selecting a vector element or an array slot by the (per-lane) symbol makes the compiler park the
// whole struct in scratch memory.
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

#ifndef SEED3_MIN_WAVES
#define SEED3_MIN_WAVES 3
#endif
#ifndef SEED3_MIN_WAVES_U64
#define SEED3_MIN_WAVES_U64 2
#endif

template <typename I>
__global__ void __launch_bounds__(128, (sizeof(I) == 8 ? SEED3_MIN_WAVES_U64 : SEED3_MIN_WAVES)) k_seed_ms(DevFM<I> fm, Chunk ck, DevOpt dopt, unsigned int *queue)
{
    const slx_opt &opt = dopt.o;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    WorkLists3<I> wl;
    // the two lists of a lane-block sit in ONE contiguous region ([list][entry][128 lanes]): the entries a wave walks through then share a
    // few pages instead of one page per (list, entry) pair spread over gigabytes
    wl.base = (LEnt<I> *)ck.lists + (size_t)blockIdx.x * ((size_t)2 * ck.cap_list * 128) + threadIdx.x; wl.stride = 128; wl.cap = ck.cap_list;
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    const uint32_t n_reads = (uint32_t)ck.n_reads;
    // ---- wave-level pool of read indices: [pool_next, pool_end) in use, [res_next, res_end) in reserve
    uint32_t pool_next = 0, pool_end = 0, res_next = 0, res_end = 0;
    uint32_t pend_base = 0;
    bool pending = false, exhausted = false;
    // ---- per-lane state: the smem machine (passes 1 + 2)
    int phase = S2_FETCH;
    uint32_t r = 0;
    uint64_t o0 = 0, o1 = 0, qoff = 0;
    int len = 0, n_out = 0;
    bool out_ovf = false, list_ovf = false;
    int pass = 1, x = 0, k2 = 0, old_n = 0;
    int sx = 0, i = 0, n = 0, ret = 0, bi = 0, cb = 0, cq = 0, cur = 1, np = 0, rev = 0, j = 0, nc = 0, last_start = 0;
    I min_intv = 1, last_sz = 0;
    IntvE<I> ik;                                   // the forward interval (both strands)
    ik.x0 = ik.x1 = ik.x2 = 0; ik.info = 0;
    LEnt<I> psrc, psrc2, nsrc, nsrc2;
    psrc.x0 = psrc.x2 = 0; psrc.info = 0;
    psrc2 = nsrc = nsrc2 = psrc;
    I nh_x0 = 0, nh_x2 = 0, nh2_x0 = 0, nh2_x2 = 0;       // first two entries of the row being written (plain scalars: structs written inside a
    uint32_t nh_info = 0, nh2_info = 0;                   // lambda through a reference end up in scratch)
    QWin win; win.bits = 0; win.chunk = 0xffffffffu;
    // ---- per-lane state: pass 3
    bool p3_run = false, p3_fresh = true;
    int p3_x = 0, p3_i = 0, n3 = 0;
    I p3_x0 = 0, p3_x1 = 0, p3_x2 = 0;
    QWin win3; win3.bits = 0; win3.chunk = 0xffffffffu;
    uint32_t trip = 0;

    auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
    auto qb3 = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win3); };
    auto push_fwd = [&](const IntvE<I> &v) {
        if (n < wl.cap) { LEnt<I> e; e.x0 = v.x0; e.x2 = v.x2; e.info = v.info; wl.at(1, n) = e; } else list_ovf = true;
        ++n;
    };
    auto out_push = [&](int start, int end, I a, I s) {       // passes 1 + 2: from the bottom of the read's slots
        if (n_out + n3 < ck.cap_intv) {
            const size_t o = (size_t)r * ck.cap_intv + n_out;
            ck.intv_info[o] = ((uint32_t)start << 16) | (uint32_t)end; ((I *)ck.intv_x0)[o] = a; ((I *)ck.intv_x2)[o] = s;
            ++n_out;
        } else out_ovf = true;
    };
    auto out_push3 = [&](int start, int end, I a, I s) {      // pass 3: from the top down
        if (n_out + n3 < ck.cap_intv) {
            const size_t o = (size_t)r * ck.cap_intv + (ck.cap_intv - 1 - n3);
            ck.intv_info[o] = ((uint32_t)start << 16) | (uint32_t)end; ((I *)ck.intv_x0)[o] = a; ((I *)ck.intv_x2)[o] = s;
            ++n3;
        } else out_ovf = true;
    };
    auto emit_mem = [&](const LEnt<I> &p) {        // a MEM [bi+1, p.info) ends the backward walk of entry p
        if (bi + 1 < last_start) {                 // not contained in a longer match already reported by this call
            last_start = bi + 1;
            if ((int)p.info - (bi + 1) >= opt.min_seed_len) out_push(bi + 1, (int)p.info, p.x0, p.x2);
        }
    };
    auto finish_read = [&]() {                     // both machines are done (or the read is abandoned on an overflow)
        // pass 3's intervals join the others: the topmost ones drop into the gap, the rest already touch it
        const size_t ob = (size_t)r * ck.cap_intv;
        const int gap = ck.cap_intv - n3 - n_out;
        const int mv = gap < n3 ? gap : n3;
        for (int t = 0; t < mv; ++t) {
            const size_t s = ob + (size_t)(ck.cap_intv - 1 - t), d = ob + (size_t)(n_out + t);
            ck.intv_info[d] = ck.intv_info[s]; ((I *)ck.intv_x0)[d] = ((I *)ck.intv_x0)[s]; ((I *)ck.intv_x2)[d] = ((I *)ck.intv_x2)[s];
        }
        ck.intv_n[r] = (uint32_t)(n_out + n3);
        if (out_ovf) atomicOr(ck.flags, OVF_INTV);
        if (list_ovf) atomicOr(ck.flags, OVF_LIST);
        phase = S2_FETCH;
    };
    auto smem_done = [&]() { phase = S2_DONE + 1; };     // passes 1 + 2 finished: wait for pass 3 (S2_WAIT3)
    constexpr int S2_WAIT3 = S2_DONE + 1;

    for (;;) {
        // The rare, long blocks (read assignment, the start of the next bwt_smem1a call) run every SEED2_EV_EVERY-th trip only -- or at
        // once when enough lanes wait or no lane has an extend to do; a waiting lane idles for a trip or two (as in dev_seed2.h).
        if (phase == S2_WAIT3 && !p3_run) finish_read();
        const unsigned long long m_ev = __ballot(phase == S2_FETCH || phase == S2_INIT || phase == S2_START);
        const unsigned long long m_run = __ballot((phase != S2_FETCH && phase != S2_INIT && phase != S2_START && phase != S2_DONE && phase != S2_WAIT3) || p3_run);
        ++trip;
        if ((trip & (SEED2_EV_EVERY - 1)) == 0 || m_run == 0 || __popcll(m_ev) >= SEED2_EV_LANES) {
            // ------------------------------------------------ read assignment (wave-uniform bookkeeping)
            if (pending) {
                const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend_base);
                pending = false;
                if (base >= n_reads) exhausted = true;
                else { res_next = base; res_end = base + SEED2_POOL < n_reads ? base + SEED2_POOL : n_reads; }
            }
            if (pool_next == pool_end && res_next != res_end) { pool_next = res_next; pool_end = res_end; res_next = res_end = 0; }
            if (res_next == res_end && !exhausted) {
                if (lane == 0) pend_base = atomicAdd(queue, (unsigned int)SEED2_POOL);
                pending = true;
            }
            if (phase == S2_INIT) {                     // offsets requested one trip ago
                qoff = o0; len = (int)(o1 - o0);
                n_out = 0; n3 = 0; out_ovf = false; list_ovf = false;
                if (len < opt.min_seed_len) { p3_run = false; finish_read(); }
                else {
                    pass = 1; x = 0; phase = S2_START;
                    p3_run = opt.max_mem_intv > 0; p3_fresh = true; p3_x = 0;
                }
            }
            {
                const unsigned long long want = __ballot(phase == S2_FETCH);
                if (want) {
                    const uint32_t avail = pool_end - pool_next;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    if (phase == S2_FETCH) {
                        if (rank < avail) {
                            r = pool_next + rank;
                            o0 = ck.offs[r]; o1 = ck.offs[r + 1];
                            phase = S2_INIT;
                        } else if (exhausted && !pending && res_next == res_end) phase = S2_DONE;
                    }
                    const uint32_t cnt = (uint32_t)__popcll(want);
                    pool_next += cnt < avail ? cnt : avail;
                }
            }
            // ------------------------------------------------ next bwt_smem1a call of this read
            if (phase == S2_START) {
                int xs = -1;
                I mi = 1;
                if (pass == 1) {
                    while (x < len && qb(x) > 3) ++x;
                    if (x < len) xs = x;
                    else { pass = 2; k2 = 0; old_n = n_out; }
                }
                if (pass == 2) {                        // re-seed from the middle of long SMEMs with few occurrences
                    while (k2 < old_n) {
                        const size_t o = (size_t)r * ck.cap_intv + k2;
                        const uint32_t inf = ck.intv_info[o];
                        const I s = ((const I *)ck.intv_x2)[o];
                        ++k2;
                        const int start = (int)(inf >> 16), end = (int)(inf & 0xffff);
                        if (end - start < split_len || s > (I)opt.split_width) continue;
                        const int mid = (start + end) >> 1;
                        if (qb(mid) > 3) continue;      // bwt_smem1a returns at once on an ambiguous start
                        xs = mid; mi = s + 1;
                        break;
                    }
                    if (xs < 0) smem_done();
                }
                if (xs >= 0) {
                    sx = xs; min_intv = mi < 1 ? (I)1 : mi;
                    const int c0 = qb(xs);
                    ik.x0 = l2_of<I>(fm, c0) + 1; ik.x1 = l2_of<I>(fm, 3 - c0) + 1;                     // bwt_set_intv
                    ik.x2 = (c0 == 3 ? fm.L2[4] : l2_of<I>(fm, c0 + 1)) - l2_of<I>(fm, c0);
                    ik.info = (uint32_t)(xs + 1);
                    i = xs + 1; n = 0;
                    phase = S2_FWD0;
                }
            }
        }
        if (phase == S2_FWD0) {                     // is there a base to extend with?
            if (list_ovf) { p3_run = false; finish_read(); }
            else if (i >= len || (cq = qb(i)) > 3) { push_fwd(ik); phase = S2_BEGIN_BWD; }
            else phase = S2_FWD;
        }
        if (phase == S2_BEGIN_BWD) {
            if (list_ovf) { p3_run = false; finish_read(); }
            else {
                ret = (int)ik.info;                 // the entry pushed last = longest forward extension = next start of pass 1
                nh_x0 = ik.x0; nh_x2 = ik.x2; nh_info = ik.info;
                if (n >= 2) { const LEnt<I> e2 = wl.at(1, n - 2); nh2_x0 = e2.x0; nh2_x2 = e2.x2; nh2_info = e2.info; }   // the entry pushed before it
                cur = 1; np = n; rev = 1; last_start = 0x7fffffff; bi = sx - 1;
                phase = S2_ROW;
            }
        }
        if (phase == S2_ROW) {                      // start of the backward row at query position bi: its first two entries are in registers
            int t = -1;
            if (bi >= 0) { t = qb(bi); if (t > 3) t = -1; }
            cb = t;
            j = 0; nc = 0; last_sz = 0;
            psrc.x0 = nh_x0; psrc.x2 = nh_x2; psrc.info = nh_info;
            psrc2.x0 = nh2_x0; psrc2.x2 = nh2_x2; psrc2.info = nh2_info;
            if (cb < 0) {                            // beginning of the read or an ambiguous base: every entry ends here, only the first can be new
                emit_mem(psrc);
                if (pass == 1) x = ret;
                phase = S2_START;
            } else phase = S2_BWD;
        }
        // ------------------------------------------------ pass 3: position its cursor
        bool actC = false;
        int cC = 0;
        if (p3_run) {
            if (p3_fresh) {
                while (p3_x < len && qb3(p3_x) > 3) ++p3_x;
                if (p3_x < len) {
                    const int c0 = qb3(p3_x);
                    p3_x0 = l2_of<I>(fm, c0) + 1; p3_x1 = l2_of<I>(fm, 3 - c0) + 1;
                    p3_x2 = (c0 == 3 ? fm.L2[4] : l2_of<I>(fm, c0 + 1)) - l2_of<I>(fm, c0);
                    p3_i = p3_x + 1; p3_fresh = false;
                } else p3_run = false;
            }
            if (p3_run) {
                if (p3_i >= len) p3_run = false;
                else {
                    const int c = qb3(p3_i);
                    if (c > 3) { p3_x = p3_i + 1; p3_fresh = true; }
                    else { actC = true; cC = 3 - c; }
                }
            }
        }
        if (__all(phase == S2_DONE)) break;
        // ------------------------------------------------ this trip's rank reads: all issued before the one wait
        const bool fwd = phase == S2_FWD, bwd = phase == S2_BWD;
        const bool actB = bwd && j + 1 < np;
        if (bwd && j + 2 < np) nsrc = wl.at(cur, rev ? np - 3 - j : j + 2);      // next trip's entries, in flight behind this trip's rank reads
        if (bwd && j + 3 < np) nsrc2 = wl.at(cur, rev ? np - 4 - j : j + 3);
        const I xinA = fwd ? ik.x1 : psrc.x0, x2A = fwd ? ik.x2 : psrc.x2;
        const int cA = fwd ? 3 - cq : cb;
        RankLd<I> rAk, rAl, rBk, rBl, rCk, rCl;
        if (fwd || bwd) { rank_issue<I>(fm, xinA - 1, rAk); rank_issue<I>(fm, xinA - 1 + x2A, rAl); }
        if (actB) { rank_issue<I>(fm, psrc2.x0 - 1, rBk); rank_issue<I>(fm, psrc2.x0 - 1 + psrc2.x2, rBl); }
        if (actC) { rank_issue<I>(fm, p3_x1 - 1, rCk); rank_issue<I>(fm, p3_x1 - 1 + p3_x2, rCl); }
        // ------------------------------------------------ slot A (+ B)
        if (fwd) {
            I ckc, ckg, clc, clg;
            rank_finish<I, true>(rAk, cA, ckc, ckg); rank_finish<I, true>(rAl, cA, clc, clg);
            IntvE<I> ok;
            ok.x2 = clc - ckc;
            ok.x1 = l2_of<I>(fm, cA) + 1 + ckc;
            ok.x0 = ik.x0 + ((ik.x1 <= fm.primary && ik.x1 + ik.x2 - 1 >= fm.primary) ? 1 : 0) + (clg - ckg);
            bool stop = false;
            if (ok.x2 != ik.x2) {
                push_fwd(ik);
                if (ok.x2 < min_intv) stop = true;
            }
            if (stop) phase = S2_BEGIN_BWD;
            else { ok.info = (uint32_t)(i + 1); ik = ok; ++i; phase = S2_FWD0; }
        } else if (bwd) {
            // first entry, then (same row, same base) the second: each is dropped, reported or carried over exactly as the scalar loop would
            auto apply_b = [&](const LEnt<I> &src, const RankLd<I> &rk, const RankLd<I> &rl) {
                I ckc, clc, dummy;
                rank_finish<I, false>(rk, cb, ckc, dummy); rank_finish<I, false>(rl, cb, clc, dummy);
                LEnt<I> ok;
                ok.x2 = clc - ckc;
                ok.x0 = l2_of<I>(fm, cb) + 1 + ckc;
                ok.info = src.info;
                if (ok.x2 < min_intv) { if (nc == 0) emit_mem(src); }
                else if (nc == 0 || ok.x2 != last_sz) {
                    if (nc == 0) { nh_x0 = ok.x0; nh_x2 = ok.x2; nh_info = ok.info; }
                    else if (nc == 1) { nh2_x0 = ok.x0; nh2_x2 = ok.x2; nh2_info = ok.info; }
                    wl.at(1 - cur, nc++) = ok;
                    last_sz = ok.x2;
                }
                ++j;
            };
            apply_b(psrc, rAk, rAl);
            if (actB) apply_b(psrc2, rBk, rBl);
            if (j >= np) {
                if (nc == 0) { if (pass == 1) x = ret; phase = S2_START; }
                else { cur = 1 - cur; np = nc; rev = 0; --bi; phase = S2_ROW; }
            } else { psrc = nsrc; psrc2 = nsrc2; }
        }
        // ------------------------------------------------ slot C
        if (actC) {
            I ckc, ckg, clc, clg;
            rank_finish<I, true>(rCk, cC, ckc, ckg); rank_finish<I, true>(rCl, cC, clc, clg);
            const I ox2 = clc - ckc;
            const I ox1 = l2_of<I>(fm, cC) + 1 + ckc;
            const I ox0 = p3_x0 + ((p3_x1 <= fm.primary && p3_x1 + p3_x2 - 1 >= fm.primary) ? 1 : 0) + (clg - ckg);
            if (ox2 < (I)opt.max_mem_intv && p3_i - p3_x >= opt.min_seed_len) {
                if (ox2 > 0) out_push3(p3_x, p3_i + 1, ox0, ox2);
                p3_x = p3_i + 1; p3_fresh = true;
            } else { p3_x0 = ox0; p3_x1 = ox1; p3_x2 = ox2; ++p3_i; }
        }
    }
}

// per-read epilogue of mem_collect_intv / prologue of mem_chain: sort by (start, end), repetitive length, seed-occurrence bound
template <typename I>
__global__ void __launch_bounds__(128) k_seed_fin(Chunk ck, DevOpt dopt)
{
    const slx_opt &opt = dopt.o;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ck.n_reads) return;
    const size_t ob = (size_t)r * ck.cap_intv;
    uint32_t *oinfo = ck.intv_info + ob;
    I *ox0 = (I *)ck.intv_x0 + ob, *ox2 = (I *)ck.intv_x2 + ob;
    const int n_out = (int)ck.intv_n[r];
    // entries with equal keys are identical intervals, so any exact sort matches ks_introsort
    for (int a = 1; a < n_out; ++a) {
        const uint32_t ki = oinfo[a]; const I k0 = ox0[a], kk2 = ox2[a];
        int b = a - 1;
        while (b >= 0 && oinfo[b] > ki) { oinfo[b + 1] = oinfo[b]; ox0[b + 1] = ox0[b]; ox2[b + 1] = ox2[b]; --b; }
        oinfo[b + 1] = ki; ox0[b + 1] = k0; ox2[b + 1] = kk2;
    }
    int b = 0, e = 0, l_rep = 0;
    uint32_t cnt = 0;
    for (int k = 0; k < n_out; ++k) {
        const I s = ox2[k];
        if (s > (I)opt.max_occ) {
            const int sb = (int)(oinfo[k] >> 16), se = (int)(oinfo[k] & 0xffff);
            if (sb > e) { l_rep += e - b; b = sb; e = se; }
            else e = e > se ? e : se;
            const I step = s / (I)opt.max_occ;
            const I cc = (s + step - 1) / step;
            cnt += (uint32_t)(cc < (I)opt.max_occ ? cc : (I)opt.max_occ);
        } else cnt += (uint32_t)s;
    }
    l_rep += e - b;
    ck.l_rep[r] = l_rep;
    ck.seed_cnt[r] = (unsigned long long)cnt;
}
