// dev_seed_sm.h -- SMEM seeding as a per-lane STATE MACHINE (production kernel; dev_seed.h's nested-loop
// kernel is kept as a test reference, seed_mode=0).  Same algorithm -- bwa's mem_collect_intv: bwt_smem1a over all
// start positions, re-seeding inside long rare SMEMs, bwt_seed_strategy1, reached from
// /root/reference/src/BWAAligner.cpp:104 -- but flattened so that every trip of ONE wave-uniform loop performs
// exactly one bwt_extend (two 64-byte rank-line reads + popcounts) per lane, whatever that lane is doing
// (forward step, backward step of a work-list entry, LAST-like step).  The nested data-dependent loops of the
// reference leave most lanes of a wave idle; here the only divergent code is the short bookkeeping between
// extends, and a lane that finishes its read pulls the next one from a device-wide queue.
#pragma once
#include "dev_seed.h"

enum SeedPhase : int {
    PH_FETCH = 0, PH_P1_NEXT, PH_FWD, PH_FWD_END, PH_BWD_ROW, PH_BWD, PH_CALL_END, PH_P2_INIT, PH_P2_NEXT, PH_P3_INIT, PH_P3_NEXT,
    PH_P3_FWD, PH_FINISH, PH_DONE
};

#ifndef SEED_MIN_WAVES
#define SEED_MIN_WAVES 6
#endif
template <typename I>
__global__ void __launch_bounds__(128, SEED_MIN_WAVES) k_seed_sm(DevFM<I> fm, Chunk ck, DevOpt dopt, unsigned int *queue)
{
    const slx_opt &opt = dopt.o;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    WorkLists<I> wl;
    wl.base = (IntvE<I> *)ck.lists + tid; wl.stride = (size_t)ck.n_threads; wl.cap = ck.cap_list;
    const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
    // ---- per-lane state
    int phase = PH_FETCH;
    int r = 0, len = 0;
    const uint8_t *q = nullptr;
    SeedOut<I> out;
    out.info = nullptr; out.x0 = nullptr; out.x2 = nullptr; out.n = 0; out.cap = ck.cap_intv; out.overflow = false;
    bool list_ovf = false;
    int pass = 1, x = 0, k2 = 0, old_n = 0;
    // one bwt_smem1a call
    int sx = 0, i = 0, n = 0, ret = 0, bi = 0, cb = 0, cur = 1, np = 0, rev = 0, j = 0, nc = 0, last_start = 0;
    I min_intv = 1, last_sz = 0;
    IntvE<I> ik;
    ik.x0 = ik.x1 = ik.x2 = 0; ik.info = 0;

    auto start_smem = [&](int xs, I mi) {
        sx = xs; min_intv = mi < 1 ? (I)1 : mi;
        set_intv<I>(fm, q[xs], ik);
        ik.info = (uint32_t)(xs + 1);
        i = xs + 1; n = 0;
    };
    auto push_fwd = [&](const IntvE<I> &v) { if (n < wl.cap) wl.at(1, n) = v; else list_ovf = true; ++n; };
    auto emit_mem = [&](const IntvE<I> &p) {       // a MEM [bi+1, p.info) ends the backward walk of entry p
        if (bi + 1 < last_start) {                 // not contained in a longer match already reported by this call
            last_start = bi + 1;
            if ((int)p.info - (bi + 1) >= opt.min_seed_len) out.push(bi + 1, (int)p.info, p.x0, p.x2);
        }
    };

    while (true) {
        // ------------------------------------------------ bookkeeping until this lane has an extend to do (or is done)
        bool need = false;
        IntvE<I> src = ik;
        int c = 0, is_back = 0;
        while (true) {
            if (phase == PH_FETCH) {
                r = (int)atomicAdd(queue, 1u);
                if (r >= ck.n_reads) { phase = PH_DONE; break; }
                q = ck.codes + ck.offs[r];
                len = (int)(ck.offs[r + 1] - ck.offs[r]);
                out.info = ck.intv_info + (size_t)r * ck.cap_intv;
                out.x0 = (I *)ck.intv_x0 + (size_t)r * ck.cap_intv;
                out.x2 = (I *)ck.intv_x2 + (size_t)r * ck.cap_intv;
                out.n = 0; out.overflow = false; list_ovf = false;
                if (len < opt.min_seed_len) phase = PH_FINISH;
                else { pass = 1; x = 0; phase = PH_P1_NEXT; }
            } else if (phase == PH_P1_NEXT) {       // pass 1: next start position with an A/C/G/T base
                while (x < len && q[x] > 3) ++x;
                if (x >= len) phase = PH_P2_INIT;
                else { start_smem(x, (I)1); phase = PH_FWD; }
            } else if (phase == PH_FWD) {
                if (list_ovf) { phase = PH_FINISH; }
                else if (i >= len || q[i] > 3) { push_fwd(ik); phase = PH_FWD_END; }
                else { need = true; src = ik; c = 3 - q[i]; is_back = 0; break; }
            } else if (phase == PH_FWD_END) {
                if (list_ovf) { phase = PH_FINISH; }
                else {
                    ret = (int)wl.at(1, n - 1).info;   // longest forward extension = next start of pass 1
                    cur = 1; np = n; rev = 1; last_start = 0x7fffffff; bi = sx - 1;
                    phase = PH_BWD_ROW;
                }
            } else if (phase == PH_BWD_ROW) {       // start of the backward row at query position bi
                cb = bi < 0 ? -1 : (q[bi] < 4 ? (int)q[bi] : -1);
                j = 0; nc = 0; last_sz = 0;
                if (cb < 0) {                        // beginning of the read or an ambiguous base: every entry ends here, only the first can be new
                    if (np > 0) emit_mem(wl.at(cur, rev ? np - 1 : 0));
                    phase = PH_CALL_END;
                } else phase = PH_BWD;
            } else if (phase == PH_BWD) {
                if (j >= np) {
                    if (nc == 0) phase = PH_CALL_END;
                    else { cur = 1 - cur; np = nc; rev = 0; --bi; phase = PH_BWD_ROW; }
                } else { need = true; src = wl.at(cur, rev ? np - 1 - j : j); c = cb; is_back = 1; break; }
            } else if (phase == PH_CALL_END) {
                if (pass == 1) { x = ret; phase = PH_P1_NEXT; }
                else phase = PH_P2_NEXT;
            } else if (phase == PH_P2_INIT) {       // pass 2: re-seed from the middle of long SMEMs with few occurrences
                pass = 2; k2 = 0; old_n = out.n;
                phase = PH_P2_NEXT;
            } else if (phase == PH_P2_NEXT) {
                int xs = -1; I mi = 1;
                while (k2 < old_n) {
                    const int start = (int)(out.info[k2] >> 16), end = (int)(out.info[k2] & 0xffff);
                    const I s = out.x2[k2];
                    ++k2;
                    if (end - start < split_len || s > (I)opt.split_width) continue;
                    xs = (start + end) >> 1; mi = s + 1;
                    break;
                }
                if (xs < 0) phase = PH_P3_INIT;
                else if (q[xs] > 3) phase = PH_P2_NEXT;   // bwt_smem1a returns at once on an ambiguous start
                else { start_smem(xs, mi); phase = PH_FWD; }
            } else if (phase == PH_P3_INIT) {
                if (opt.max_mem_intv > 0) { pass = 3; x = 0; phase = PH_P3_NEXT; }
                else phase = PH_FINISH;
            } else if (phase == PH_P3_NEXT) {       // pass 3: LAST-like forward-only seeds (bwt_seed_strategy1)
                while (x < len && q[x] > 3) ++x;
                if (x >= len) phase = PH_FINISH;
                else { set_intv<I>(fm, q[x], ik); i = x + 1; phase = PH_P3_FWD; }
            } else if (phase == PH_P3_FWD) {
                if (i >= len) { x = len; phase = PH_P3_NEXT; }
                else if (q[i] > 3) { x = i + 1; phase = PH_P3_NEXT; }
                else { need = true; src = ik; c = 3 - q[i]; is_back = 0; break; }
            } else if (phase == PH_FINISH) {
                // sort by (start, end): entries with equal keys are identical intervals, so any exact sort matches ks_introsort
                for (int a = 1; a < out.n; ++a) {
                    const uint32_t ki = out.info[a]; const I k0 = out.x0[a], kk2 = out.x2[a];
                    int b = a - 1;
                    while (b >= 0 && out.info[b] > ki) { out.info[b + 1] = out.info[b]; out.x0[b + 1] = out.x0[b]; out.x2[b + 1] = out.x2[b]; --b; }
                    out.info[b + 1] = ki; out.x0[b + 1] = k0; out.x2[b + 1] = kk2;
                }
                // mem_chain prologue: repetitive fraction and the number of seed occurrences to look up
                int b = 0, e = 0, l_rep = 0;
                uint32_t cnt = 0;
                for (int k = 0; k < out.n; ++k) {
                    const I s = out.x2[k];
                    if (s > (I)opt.max_occ) {
                        const int sb = (int)(out.info[k] >> 16), se = (int)(out.info[k] & 0xffff);
                        if (sb > e) { l_rep += e - b; b = sb; e = se; }
                        else e = e > se ? e : se;
                        const I step = s / (I)opt.max_occ;
                        const I cc = (s + step - 1) / step;
                        cnt += (uint32_t)(cc < (I)opt.max_occ ? cc : (I)opt.max_occ);
                    } else cnt += (uint32_t)s;
                }
                l_rep += e - b;
                ck.intv_n[r] = (uint32_t)out.n;
                ck.l_rep[r] = l_rep;
                ck.seed_cnt[r] = (unsigned long long)cnt;
                if (out.overflow) atomicOr(ck.flags, OVF_INTV);
                if (list_ovf) atomicOr(ck.flags, OVF_LIST);
                phase = PH_FETCH;
            } else break;                            // PH_DONE
        }
        if (__all(phase == PH_DONE)) break;
        // ------------------------------------------------ the one bwt_extend of this trip
        IntvE<I> ok;
        ok.x0 = ok.x1 = ok.x2 = 0; ok.info = 0;
        if (need) fm_extend<I>(fm, src, c, is_back, ok);
        // ------------------------------------------------ apply it
        if (need) {
            if (phase == PH_FWD) {
                bool stop = false;
                if (ok.x2 != ik.x2) {
                    push_fwd(ik);
                    if (ok.x2 < min_intv) stop = true;
                }
                if (stop) phase = PH_FWD_END;
                else { ok.info = (uint32_t)(i + 1); ik = ok; ++i; }
            } else if (phase == PH_BWD) {
                if (ok.x2 < min_intv) { if (nc == 0) emit_mem(src); }
                else if (nc == 0 || ok.x2 != last_sz) {
                    ok.info = src.info;
                    wl.at(1 - cur, nc++) = ok;
                    last_sz = ok.x2;
                }
                ++j;
            } else {                                 // PH_P3_FWD
                if (ok.x2 < (I)opt.max_mem_intv && i - x >= opt.min_seed_len) {
                    if (ok.x2 > 0) out.push(x, i + 1, ok.x0, ok.x2);
                    x = i + 1;
                    phase = PH_P3_NEXT;
                } else { ik = ok; ++i; }
            }
        }
    }
}
