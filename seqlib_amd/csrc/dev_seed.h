// dev_seed.h -- SMEM seeding kernel: bwa's mem_collect_intv (three passes: all SMEMs, re-seeding
// inside long rare SMEMs, LAST-like forward seeds) as reached from
// /root/reference/src/BWAAligner.cpp:104 -> mem_align1 -> mem_chain.  SURVEY.md Appendix A.3/A.4.
// One lane owns one read; the bidirectional work lists live in a lane-interleaved HBM scratch so
// that lanes walking their lists in step touch neighbouring 16-byte slots.
#pragma once
#include "dev_fm.h"
#include "dev_types.h"

template <typename I>
struct SeedOut {              // per-read output slots
    uint32_t *info; I *x0; I *x2;
    int n, cap;
    bool overflow;
    __device__ __forceinline__ void push(int start, int end, I a, I s)
    {
        if (n < cap) { info[n] = ((uint32_t)start << 16) | (uint32_t)end; x0[n] = a; x2[n] = s; ++n; }
        else overflow = true;
    }
};

template <typename I>
struct WorkLists {
    IntvE<I> *base;           // already offset by the lane's slot
    size_t stride;            // n_threads
    int cap;
    __device__ __forceinline__ IntvE<I> &at(int list, int e) { return base[((size_t)list * cap + e) * stride]; }
};

// bwt_smem1a with max_intv = 0.  Emits every MEM found (start, end, interval) through `out` when it
// is at least min_seed_len long; returns the next start position.
template <typename I>
__device__ int dev_smem1(const DevFM<I> &fm, const uint8_t *q, int len, int x, I min_intv, int min_seed_len,
                         WorkLists<I> &wl, SeedOut<I> &out, bool &list_ovf)
{
    if (q[x] > 3) return x + 1;
    if (min_intv < 1) min_intv = 1;
    IntvE<I> ik, ok;
    set_intv<I>(fm, q[x], ik);
    ik.info = (uint32_t)(x + 1);
    int n = 0, i;
    auto push = [&](const IntvE<I> &v) { if (n < wl.cap) wl.at(1, n) = v; else list_ovf = true; ++n; };
    for (i = x + 1; i < len; ++i) {           // forward search
        if (q[i] < 4) {
            fm_extend<I>(fm, ik, 3 - q[i], 0, ok);
            if (ok.x2 != ik.x2) {
                push(ik);
                if (ok.x2 < min_intv) break;
            }
            ok.info = (uint32_t)(i + 1);
            ik = ok;
        } else { push(ik); break; }
    }
    if (i == len) push(ik);
    if (list_ovf) return len;                  // chunk will be re-run with longer lists
    const int ret = (int)wl.at(1, n - 1).info; // longest forward extension
    int cur = 1, np = n, rev = 1, last_start = 0x7fffffff;
    for (i = x - 1; i >= -1; --i) {            // backward search for MEMs
        const int c = i < 0 ? -1 : (q[i] < 4 ? (int)q[i] : -1);
        int nc = 0;
        I last_sz = 0;
        for (int j = 0; j < np; ++j) {
            IntvE<I> p = wl.at(cur, rev ? np - 1 - j : j);
            if (c >= 0) fm_extend<I>(fm, p, c, 1, ok);
            if (c < 0 || ok.x2 < min_intv) {
                if (nc == 0) {
                    if (i + 1 < last_start) {   // not contained in a longer match already reported
                        last_start = i + 1;
                        if ((int)p.info - (i + 1) >= min_seed_len) out.push(i + 1, (int)p.info, p.x0, p.x2);
                    }
                }
            } else if (nc == 0 || ok.x2 != last_sz) {
                ok.info = p.info;
                wl.at(1 - cur, nc++) = ok;
                last_sz = ok.x2;
            }
        }
        if (nc == 0) break;
        cur = 1 - cur; np = nc; rev = 0;
    }
    return ret;
}

template <typename I>
__global__ void __launch_bounds__(128) k_seed(DevFM<I> fm, Chunk ck, DevOpt dopt)
{
    const slx_opt &opt = dopt.o;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    WorkLists<I> wl;
    wl.base = (IntvE<I> *)ck.lists + tid; wl.stride = (size_t)ck.n_threads; wl.cap = ck.cap_list;
    for (int r = tid; r < ck.n_reads; r += ck.n_threads) {
        const uint8_t *q = ck.codes + ck.offs[r];
        const int len = (int)(ck.offs[r + 1] - ck.offs[r]);
        SeedOut<I> out;
        out.info = ck.intv_info + (size_t)r * ck.cap_intv;
        out.x0 = (I *)ck.intv_x0 + (size_t)r * ck.cap_intv;
        out.x2 = (I *)ck.intv_x2 + (size_t)r * ck.cap_intv;
        out.n = 0; out.cap = ck.cap_intv; out.overflow = false;
        bool list_ovf = false;
        if (len >= opt.min_seed_len) {
            // pass 1: all SMEMs
            int x = 0;
            while (x < len) {
                if (q[x] < 4) x = dev_smem1<I>(fm, q, len, x, (I)1, opt.min_seed_len, wl, out, list_ovf);
                else ++x;
            }
            // pass 2: re-seed from the middle of long SMEMs with few occurrences
            const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
            const int old_n = out.n;
            for (int k = 0; k < old_n; ++k) {
                const int start = (int)(out.info[k] >> 16), end = (int)(out.info[k] & 0xffff);
                const I s = out.x2[k];
                if (end - start < split_len || s > (I)opt.split_width) continue;
                dev_smem1<I>(fm, q, len, (start + end) >> 1, s + 1, opt.min_seed_len, wl, out, list_ovf);
            }
            // pass 3: LAST-like forward-only seeds (bwt_seed_strategy1)
            if (opt.max_mem_intv > 0) {
                x = 0;
                while (x < len) {
                    if (q[x] < 4) {
                        IntvE<I> ik, ok;
                        set_intv<I>(fm, q[x], ik);
                        int i, nx = len;
                        for (i = x + 1; i < len; ++i) {
                            if (q[i] < 4) {
                                fm_extend<I>(fm, ik, 3 - q[i], 0, ok);
                                if (ok.x2 < (I)opt.max_mem_intv && i - x >= opt.min_seed_len) {
                                    if (ok.x2 > 0) out.push(x, i + 1, ok.x0, ok.x2);
                                    nx = i + 1;
                                    break;
                                }
                                ik = ok;
                            } else { nx = i + 1; break; }
                        }
                        x = nx;
                    } else ++x;
                }
            }
            // sort by (start, end): entries with equal keys are identical intervals, so any exact sort matches ks_introsort
            for (int a = 1; a < out.n; ++a) {
                uint32_t ki = out.info[a]; I k0 = out.x0[a], k2 = out.x2[a];
                int b = a - 1;
                while (b >= 0 && out.info[b] > ki) { out.info[b + 1] = out.info[b]; out.x0[b + 1] = out.x0[b]; out.x2[b + 1] = out.x2[b]; --b; }
                out.info[b + 1] = ki; out.x0[b + 1] = k0; out.x2[b + 1] = k2;
            }
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
                I c = (s + step - 1) / step;
                cnt += (uint32_t)(c < (I)opt.max_occ ? c : (I)opt.max_occ);
            } else cnt += (uint32_t)s;
        }
        l_rep += e - b;
        ck.intv_n[r] = (uint32_t)out.n;
        ck.l_rep[r] = l_rep;
        ck.seed_cnt[r] = (unsigned long long)cnt;
        if (out.overflow) atomicOr(ck.flags, OVF_INTV);
        if (list_ovf) atomicOr(ck.flags, OVF_LIST);
    }
}
