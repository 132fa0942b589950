// experiments/dev_seed2_state_machine.h -- the round-1 / early round-2 seeding kernels (k_seed12: passes 1 + 2 as a per-lane state
// machine with one bwt_extend per trip of a wave-uniform loop; k_seed3: pass 3 + epilogue), kept for reference only.  Superseded by
// seqlib_amd/csrc/dev_seed4.h (wave-level mode loops + k-mer table): the state machine spent ~340 instructions around every extend
// (DESIGN.md section 4).  Not compiled into the library; it would need dev_occ.h's includes and the all-four-symbols rank below.
#pragma once
#include "../seqlib_amd/csrc/dev_occ.h"
#define SEED2_POOL SEED_POOL

// counts of A,C,G,T in BWT[0..k] (bwt_occ4), k already mapped past the sentinel
template <typename I>
__device__ __forceinline__ void occp_rank(const DevFM<I> &fm, I kk, I t[4])
{
    const uint4 *blk = fm.occ + ((size_t)(kk >> 6) << 1);
    const uint4 c = blk[0], p = blk[1];
    I b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (sizeof(I) == 8) {                                            // super-block base: a cached 32-byte table row, in flight with the block
        const ulonglong2 *sp = (const ulonglong2 *)(fm.sup + ((uint64_t)kk >> 32) * 4);
        const ulonglong2 s0 = sp[0], s1 = sp[1];
        b0 = (I)s0.x; b1 = (I)s0.y; b2 = (I)s1.x; b3 = (I)s1.y;
    }
    const uint32_t n = ((uint32_t)kk & 63) + 1;
    const uint64_t m = ~0ull >> (64 - n);
    const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
    const uint32_t l0 = p.x & m0, l1 = p.y & m1, h0 = p.z & m0, h1 = p.w & m1;
    const uint32_t sl = __popc(l0) + __popc(l1), sh = __popc(h0) + __popc(h1), st = __popc(l0 & h0) + __popc(l1 & h1);
    t[0] = b0 + (I)(c.x + (n + st - sl - sh)); t[1] = b1 + (I)(c.y + (sl - st)); t[2] = b2 + (I)(c.z + (sh - st)); t[3] = b3 + (I)(c.w + st);
}

// bwt_extend for the one output symbol the caller needs (same contract as fm_extend in dev_fm.h)
template <typename I>
__device__ __forceinline__ void fm_extend_p(const DevFM<I> &fm, const IntvE<I> &ik, int c, int is_back, IntvE<I> &ok)
{
    I tk[4], tl[4];
    const I xin = is_back ? ik.x0 : ik.x1;
    const I xot = is_back ? ik.x1 : ik.x0;
    const I k = xin - 1, l = xin - 1 + ik.x2;
    const bool k_none = (k == (I)-1), l_none = (l == (I)-1);
    occp_rank<I>(fm, k_none ? (I)0 : k - (k >= fm.primary ? 1 : 0), tk);
    occp_rank<I>(fm, l_none ? (I)0 : l - (l >= fm.primary ? 1 : 0), tl);
    if (k_none) tk[0] = tk[1] = tk[2] = tk[3] = 0;
    if (l_none) tl[0] = tl[1] = tl[2] = tl[3] = 0;
    const I s0 = tl[0] - tk[0], s1 = tl[1] - tk[1], s2 = tl[2] - tk[2], s3 = tl[3] - tk[3];
    I base = xot + ((xin <= fm.primary && xin + ik.x2 - 1 >= fm.primary) ? 1 : 0);
    I sz = s3, nin = fm.L2[3] + 1 + tk[3];
    if (c <= 2) { base += s3; sz = s2; nin = fm.L2[2] + 1 + tk[2]; }
    if (c <= 1) { base += s2; sz = s1; nin = fm.L2[1] + 1 + tk[1]; }
    if (c == 0) { base += s1; sz = s0; nin = fm.L2[0] + 1 + tk[0]; }
    if (is_back) { ok.x0 = nin; ok.x1 = base; } else { ok.x1 = nin; ok.x0 = base; }
    ok.x2 = sz;
}


#ifndef SEED2_MIN_WAVES
#define SEED2_MIN_WAVES 6
#endif
#ifndef SEED2_MIN_WAVES_U64
#define SEED2_MIN_WAVES_U64 4      // u64 intervals: 6 waves/SIMD would spill (152 bytes of scratch); 4 keeps everything in registers
#endif
#ifndef SEED2_POOL
#define SEED2_POOL 16
#endif
#ifndef SEED2_DEBUG
#define SEED2_DEBUG 0
#endif
#ifndef SEED2_EV_EVERY
#define SEED2_EV_EVERY 4      // power of two
#endif
#ifndef SEED2_EV_LANES
#define SEED2_EV_LANES 8
#endif

enum Seed2Phase : int { S2_FETCH = 0, S2_INIT, S2_START, S2_FWD0, S2_BEGIN_BWD, S2_ROW, S2_FWD, S2_BWD, S2_DONE };

template <typename I>
__global__ void __launch_bounds__(128, (sizeof(I) == 8 ? SEED2_MIN_WAVES_U64 : SEED2_MIN_WAVES)) k_seed12(DevFM<I> fm, Chunk ck, DevOpt dopt, unsigned int *queue)
{
    const slx_opt &opt = dopt.o;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    WorkLists<I> wl;
    // the two lists of a lane-block sit in ONE contiguous region ([list][entry][128 lanes]): the entries a wave walks through then share a
    // few pages instead of one page per (list, entry) pair spread over gigabytes
    wl.base = (IntvE<I> *)ck.lists + (size_t)blockIdx.x * ((size_t)2 * ck.cap_list * 128) + threadIdx.x; wl.stride = 128; wl.cap = ck.cap_list;
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    const uint32_t n_reads = (uint32_t)ck.n_reads;
    // ---- wave-level pool of read indices: [pool_next, pool_end) in use, [res_next, res_end) in reserve
    uint32_t pool_next = 0, pool_end = 0, res_next = 0, res_end = 0;
    uint32_t pend_base = 0;
    bool pending = false, exhausted = false;
    // ---- per-lane state
    int phase = S2_FETCH;
    uint32_t r = 0;
    uint64_t o0 = 0, o1 = 0, qoff = 0;
    int len = 0, n_out = 0;
    bool out_ovf = false, list_ovf = false;
    int pass = 1, x = 0, k2 = 0, old_n = 0;
    int sx = 0, i = 0, n = 0, ret = 0, bi = 0, cb = 0, cq = 0, cur = 1, np = 0, rev = 0, j = 0, nc = 0, last_start = 0;
    I min_intv = 1, last_sz = 0;
    IntvE<I> ik, head, nhead, psrc, nsrc;
    ik.x0 = ik.x1 = ik.x2 = 0; ik.info = 0;
    head = nhead = psrc = nsrc = ik;
    QWin win; win.bits = 0; win.chunk = 0xffffffffu;
    uint32_t trip = 0;
#if SEED2_DEBUG
    unsigned long long dbg_ext = 0, dbg_ev = 0, dbg_done = 0;   // (three more live values cost this kernel a register tier: off in production)
#endif

    auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
    auto push_fwd = [&](const IntvE<I> &v) { if (n < wl.cap) wl.at(1, n) = v; else list_ovf = true; ++n; };
    auto out_push = [&](int start, int end, I a, I s) {
        if (n_out < ck.cap_intv) {
            const size_t o = (size_t)r * ck.cap_intv + n_out;
            ck.intv_info[o] = ((uint32_t)start << 16) | (uint32_t)end; ((I *)ck.intv_x0)[o] = a; ((I *)ck.intv_x2)[o] = s;
            ++n_out;
        } else out_ovf = true;
    };
    auto emit_mem = [&](const IntvE<I> &p) {       // a MEM [bi+1, p.info) ends the backward walk of entry p
        if (bi + 1 < last_start) {                 // not contained in a longer match already reported by this call
            last_start = bi + 1;
            if ((int)p.info - (bi + 1) >= opt.min_seed_len) out_push(bi + 1, (int)p.info, p.x0, p.x2);
        }
    };
    auto finish_read = [&]() {                     // passes 1+2 done (or abandoned on a work-list overflow)
        ck.intv_n[r] = (uint32_t)n_out;
        if (out_ovf) atomicOr(ck.flags, OVF_INTV);
        if (list_ovf) atomicOr(ck.flags, OVF_LIST);
        phase = S2_FETCH;
    };

    for (;;) {
        // Read assignment and the start of the next bwt_smem1a call are the long, rarely needed blocks of this loop (a dozen times per
        // read against ~300 extends), yet with 64 lanes at different points of their reads some lane wants them on nearly every trip,
        // and every lane then pays their issue slots.  So they run every SEED2_EV_EVERY-th trip only -- or at once when enough lanes
        // wait or no lane has an extend to do; a waiting lane idles for a trip or two.
        const unsigned long long m_ev = __ballot(phase == S2_FETCH || phase == S2_INIT || phase == S2_START);
        const unsigned long long m_run = __ballot(phase != S2_FETCH && phase != S2_INIT && phase != S2_START && phase != S2_DONE);
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
                n_out = 0; out_ovf = false; list_ovf = false;
                if (len < opt.min_seed_len) finish_read();
                else { pass = 1; x = 0; phase = S2_START; }
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
                    if (xs < 0) finish_read();
                }
                if (xs >= 0) {
                    sx = xs; min_intv = mi < 1 ? (I)1 : mi;
                    set_intv<I>(fm, qb(xs), ik);
                    ik.info = (uint32_t)(xs + 1);
                    i = xs + 1; n = 0;
                    phase = S2_FWD0;
                }
            }
        }
        if (phase == S2_FWD0) {                     // is there a base to extend with?
            if (list_ovf) finish_read();
            else if (i >= len || (cq = qb(i)) > 3) { push_fwd(ik); phase = S2_BEGIN_BWD; }
            else phase = S2_FWD;
        }
        if (phase == S2_BEGIN_BWD) {
            if (list_ovf) finish_read();
            else {
                ret = (int)ik.info;                 // the entry pushed last = longest forward extension = next start of pass 1
                head = ik;
                cur = 1; np = n; rev = 1; last_start = 0x7fffffff; bi = sx - 1;
                phase = S2_ROW;
            }
        }
        if (phase == S2_ROW) {                      // start of the backward row at query position bi
            int t = -1;
            if (bi >= 0) { t = qb(bi); if (t > 3) t = -1; }
            cb = t;
            j = 0; nc = 0; last_sz = 0;
            if (cb < 0) {                            // beginning of the read or an ambiguous base: every entry ends here, only the first can be new
                emit_mem(head);
                if (pass == 1) x = ret;
                phase = S2_START;
            } else { psrc = head; phase = S2_BWD; }
        }
        if (__all(phase == S2_DONE)) break;
        // ------------------------------------------------ the one bwt_extend of this trip
        const bool fwd = phase == S2_FWD, bwd = phase == S2_BWD;
#if SEED2_DEBUG
        if (ck.dbg_cyc && ck.dbg_stage == 3) {       // -DSEED2_DEBUG=1 + SLX_DEBUG_CYC=3: lanes extending / waiting for an event block / done, per trip
            dbg_ext += (unsigned long long)__popcll(__ballot(fwd || bwd));
            dbg_ev += (unsigned long long)__popcll(__ballot(phase == S2_FETCH || phase == S2_INIT || phase == S2_START));
            dbg_done += (unsigned long long)__popcll(__ballot(phase == S2_DONE));
        }
#endif
        if (bwd && j + 1 < np) nsrc = wl.at(cur, rev ? np - 2 - j : j + 1);       // next trip's entry, in flight behind this trip's rank reads
        IntvE<I> ok;
        ok.x0 = ok.x1 = ok.x2 = 0; ok.info = 0;
        if (fwd || bwd) {                           // ONE inlined copy of the extend for both directions: a wave nearly always holds both kinds of lane
            IntvE<I> src = psrc;
            int c = cb;
            if (fwd) { src = ik; c = 3 - cq; }
            fm_extend_p<I>(fm, src, c, bwd ? 1 : 0, ok);
        }
        // ------------------------------------------------ apply it
        if (fwd) {
            bool stop = false;
            if (ok.x2 != ik.x2) {
                push_fwd(ik);
                if (ok.x2 < min_intv) stop = true;
            }
            if (stop) phase = S2_BEGIN_BWD;
            else { ok.info = (uint32_t)(i + 1); ik = ok; ++i; phase = S2_FWD0; }
        } else if (bwd) {
            if (ok.x2 < min_intv) { if (nc == 0) emit_mem(psrc); }
            else if (nc == 0 || ok.x2 != last_sz) {
                ok.info = psrc.info;
                if (nc == 0) nhead = ok;
                wl.at(1 - cur, nc++) = ok;
                last_sz = ok.x2;
            }
            ++j;
            if (j >= np) {
                if (nc == 0) { if (pass == 1) x = ret; phase = S2_START; }
                else { cur = 1 - cur; np = nc; rev = 0; --bi; head = nhead; phase = S2_ROW; }
            } else psrc = nsrc;
        }
    }
#if SEED2_DEBUG
    if (ck.dbg_cyc && ck.dbg_stage == 3 && lane == 0) {
        atomicAdd(ck.dbg_cyc + 0, (unsigned long long)trip); atomicAdd(ck.dbg_cyc + 1, dbg_ext); atomicAdd(ck.dbg_cyc + 2, dbg_ev); atomicAdd(ck.dbg_cyc + 3, dbg_done);
    }
#endif
}

// pass 3 + per-read epilogue; one lane per read
template <typename I>
__global__ void __launch_bounds__(128) k_seed3(DevFM<I> fm, Chunk ck, DevOpt dopt)
{
    const slx_opt &opt = dopt.o;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < ck.n_reads;
    const uint64_t qoff = live ? ck.offs[r] : 0;
    const int len = live ? (int)(ck.offs[r + 1] - qoff) : 0;
    const size_t ob = (size_t)(live ? r : 0) * ck.cap_intv;
    uint32_t *oinfo = ck.intv_info + ob;
    I *ox0 = (I *)ck.intv_x0 + ob, *ox2 = (I *)ck.intv_x2 + ob;
    int n_out = live ? (int)ck.intv_n[r] : 0;
    bool out_ovf = false;
    QWin win; win.bits = 0; win.chunk = 0xffffffffu;
    auto qb = [&](int p) { return q_at(ck.codes, qoff + (uint64_t)p, win); };
    if (opt.max_mem_intv > 0) {                      // bwt_seed_strategy1 from every position a seed ended at
        int x = len >= opt.min_seed_len ? 0 : len, i = 0;
        bool fresh = true;
        IntvE<I> ik, ok;
        ik.x0 = ik.x1 = ik.x2 = 0; ik.info = 0;
        for (;;) {
            if (fresh) {
                while (x < len && qb(x) > 3) ++x;
                if (x < len) { set_intv<I>(fm, qb(x), ik); i = x + 1; fresh = false; }
            }
            const bool act = !fresh && i < len;
            if (!__any(act)) break;
            if (act) {
                const int c = qb(i);
                if (c > 3) { x = i + 1; fresh = true; }
                else {
                    fm_extend_p<I>(fm, ik, 3 - c, 0, ok);
                    if (ok.x2 < (I)opt.max_mem_intv && i - x >= opt.min_seed_len) {
                        if (ok.x2 > 0) {
                            if (n_out < ck.cap_intv) { oinfo[n_out] = ((uint32_t)x << 16) | (uint32_t)(i + 1); ox0[n_out] = ok.x0; ox2[n_out] = ok.x2; ++n_out; }
                            else out_ovf = true;
                        }
                        x = i + 1; fresh = true;
                    } else { ik = ok; ++i; }
                }
            }
        }
    }
    if (!live) return;
    // sort by (start, end): entries with equal keys are identical intervals, so any exact sort matches ks_introsort
    for (int a = 1; a < n_out; ++a) {
        const uint32_t ki = oinfo[a]; const I k0 = ox0[a], kk2 = ox2[a];
        int b = a - 1;
        while (b >= 0 && oinfo[b] > ki) { oinfo[b + 1] = oinfo[b]; ox0[b + 1] = ox0[b]; ox2[b + 1] = ox2[b]; --b; }
        oinfo[b + 1] = ki; ox0[b + 1] = k0; ox2[b + 1] = kk2;
    }
    // mem_chain prologue: repetitive fraction and the number of seed occurrences to look up
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
    ck.intv_n[r] = (uint32_t)n_out;
    ck.l_rep[r] = l_rep;
    ck.seed_cnt[r] = (unsigned long long)cnt;
    if (out_ovf) atomicOr(ck.flags, OVF_INTV);
}
