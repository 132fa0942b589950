// dev_long.h -- what reads of ~727 bp and more need beyond the 150 bp pipeline (contigs realigned through BWAAligner,
// /root/reference/src/seqtools/seqtools.cpp:198-210, reach mem_align1 from /root/reference/src/BWAAligner.cpp:104 just like reads):
//
//   k_long_list / k_flt_seeds   bwa's mem_flt_chained_seeds (bwamem.c): every seed shorter than 100 bp of every kept chain is
//                               re-scored by an exact affine-gap local alignment of the seed +- 50 bp (mem_seed_sw -> ksw_align2,
//                               whose 16-bit SSE2 kernel returns the plain Smith-Waterman optimum); seeds scoring below
//                               min_HSP_score leave the chain, the others carry their score into mem_chain2aln's seed order.
//                               One wave per read, one seed per lane (a 199 x 199 cell matrix at most, rows in lane-private memory).
//   k_cig_long                  mem_reg2aln's bwa_gen_cigar2 -> ksw_global2 + traceback for queries beyond the register-resident
//                               wave kernels of dev_fin2.h: one lane per job, H/E rows in a per-thread global scratch.
// The other stages run on the general kernels: seeding and chaining as for any read, extension on k_extend_reg with its H/E row in
// LDS (wave_ksw_extend2, query codes re-read per tile), regions on k_regs with the same per-thread scratch.
// SURVEY.md A.6.
#pragma once
#include "dev_fin2.h"

// reads of the chunk the seed filter applies to (kept chains only)
__global__ void k_long_list(Chunk ck, DevOpt dopt, int *list, unsigned int *n_list)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ck.n_reads) return;
    const int len = (int)(ck.offs[r + 1] - ck.offs[r]);
    if (ck.n_chain[r] <= 0 || len <= 0) return;
    if (len >= ck.log_lut_n) { atomicOr(ck.flags, ERR_LOGLUT); return; }
    if (flt_live(dopt.o, len, ck.log_lut[len], nullptr)) list[atomicAdd(n_list, 1u)] = r;
}

// mem_seed_sw: -1 = no alignment needed (long seed, or its +-50 bp window reaches 200 bp)
__device__ int dev_seed_sw(const DevRef &R, const slx_opt &o, const uint8_t *query, int l_query, int s_qbeg, int s_len, int64_t s_rbeg)
{
    if (s_len >= MEM_SHORT_LEN) return -1;
    int qb = s_qbeg, qe = s_qbeg + s_len;
    int64_t rb = s_rbeg, re = s_rbeg + s_len;
    const int64_t mid = (rb + re) >> 1, l_pac = R.l_pac;
    qb -= MEM_SHORT_EXT; qb = qb > 0 ? qb : 0;
    qe += MEM_SHORT_EXT; qe = qe < l_query ? qe : l_query;
    rb -= MEM_SHORT_EXT; rb = rb > 0 ? rb : 0;
    re += MEM_SHORT_EXT; re = re < l_pac << 1 ? re : l_pac << 1;
    if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
    if (qe - qb >= MEM_SHORT_LEN || re - rb >= MEM_SHORT_LEN) return -1;
    {   // bns_fetch_seq: clip to the contig (on its strand) that holds mid
        int is_rev;
        const int rid = dev_pos2rid(R, dev_depos(R, mid, &is_rev));
        int64_t far_beg = R.ann_off[rid], far_end = far_beg + R.ann_len[rid];
        if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
        rb = rb > far_beg ? rb : far_beg;
        re = re < far_end ? re : far_end;
    }
    // ksw_align2(..., KSW_XSTART, 0).score: affine-gap local alignment, E/F fed by H, everything floored at 0
    const int qlen = qe - qb, tlen = (int)(re - rb);
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    int H[MEM_SHORT_LEN], E[MEM_SHORT_LEN];
    for (int j = 0; j <= qlen; ++j) { H[j] = 0; E[j] = 0; }
    int best = 0;
    for (int i = 0; i < tlen; ++i) {
        const int8_t *row = o.mat + ref_base(R, rb + i) * 5;
        int f = 0, hdiag = 0;
        for (int j = 1; j <= qlen; ++j) {
            int h = hdiag + row[query[qb + j - 1]];
            int e = E[j];
            hdiag = H[j];
            h = h > e ? h : e;
            h = h > f ? h : f;
            h = h > 0 ? h : 0;
            H[j] = h;
            best = best > h ? best : h;
            e -= o.e_del; { const int x = h - oe_del; e = e > x ? e : x; } e = e > 0 ? e : 0; E[j] = e;
            f -= o.e_ins; { const int x = h - oe_ins; f = f > x ? f : x; } f = f > 0 ? f : 0;
        }
    }
    return best;
}

// mem_flt_chained_seeds for the listed reads, in two steps.  k_flt_score: the local alignment of every seed of every kept chain (the
// flattened lists of a read are one contiguous slice of c_w), one seed per lane, FLT_PARTS blocks per read -- a contig has thousands
// of short seeds, and on one wave per read a hundred contigs left the chip idle for 100 ms.  k_flt_seeds: the seeds below min_HSP_score
// leave their chains (order kept), one wave per read.
#define FLT_PARTS 64
__global__ void __launch_bounds__(64) k_flt_score(DevRef R, Chunk ck, DevOpt dopt, const int *list, const unsigned int *n_list)
{
    const slx_opt &opt = dopt.o;
    const unsigned int t = blockIdx.x / FLT_PARTS, part = blockIdx.x % FLT_PARTS;
    if (t >= *n_list) return;
    const int r = list[t];
    ReadWS w = make_ws_uniform(ck, r);
    const int n_chn = ck.n_chain[r];
    if (n_chn <= 0) return;
    const int c_last = w.ia[n_chn - 1];
    const int total = w.c_first[c_last] + w.c_n[c_last];           // the kept chains' lists are laid out back to back from c_w[0]
    const uint8_t *query = ck.codes + ck.offs[r];
    const int l_query = (int)(ck.offs[r + 1] - ck.offs[r]);
    for (int i = (int)part * 64 + (int)threadIdx.x; i < total; i += FLT_PARTS * 64) {
        const int s = w.c_w[i];
        w.s_score[s] = dev_seed_sw(R, opt, query, l_query, w.s_qbeg(s), w.s_len(s), w.s_rbeg[s]);
    }
}

__global__ void __launch_bounds__(64) k_flt_seeds(DevRef R, Chunk ck, DevOpt dopt, const int *list, const unsigned int *n_list, unsigned int *queue)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const unsigned int n_todo = *n_list;
    for (;;) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(queue, 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= n_todo) break;
        const int r = list[t];
        ReadWS w = make_ws_uniform(ck, r);
        const int l_query = (int)(ck.offs[r + 1] - ck.offs[r]);
        int min_hsp = 0;
        flt_live(opt, l_query, ck.log_lut[l_query], &min_hsp);
        const int n_chn = ck.n_chain[r];
        for (int ci = lane; ci < n_chn; ci += 64) {              // a chain per lane
            const int c = w.ia[ci];
            const int n = w.c_n[c];
            int *cs = w.c_w + w.c_first[c];
            int k = 0;
            for (int j = 0; j < n; ++j) {
                const int s = cs[j];
                const int sc = w.s_score[s];
                if (sc < 0 || sc >= min_hsp) { w.s_score[s] = sc < 0 ? w.s_len(s) * opt.a : sc; cs[k++] = s; }
            }
            w.c_n[c] = k;
        }
    }
}

// one bwa_gen_cigar2 sequence of mem_reg2aln (up to three band widths), traceback, NM, position: one LANE per job
__global__ void __launch_bounds__(128) k_cig_long(DevRef R, Chunk ck, DevOpt dopt, FinLists fl)
{
    const slx_opt &opt = dopt.o;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= ck.long_threads) return;
    int *eh_h = ck.long_scratch + (size_t)tid * 2 * ck.long_stride, *eh_e = eh_h + ck.long_stride;
    const unsigned int n_jobs = *fl.n_dp;
    for (;;) {
        const unsigned int t = atomicAdd(fl.q_dp, 1u);
        if (t >= n_jobs) break;
        const uint32_t slot = fl.dp_list[t];
        const DJob j = fl.jobs[slot];
        DHit h = ck.hits[slot];
        const uint8_t *query = ck.codes + ck.offs[j.r];
        const int l_query = (int)(ck.offs[j.r + 1] - ck.offs[j.r]);
        const int lq = j.qe - j.qb;
        const uint8_t *qseg = query + j.qb;
        int w2 = j.w2, it = 0, score, last_sc = -(1 << 30);
        GenCig g;
        do {
            w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
            g = dev_gen_cigar2<0>(R, opt, ck, w2, lq, qseg, j.rb, j.re, true, eh_h, eh_e);
            score = g.score;
            if (score == last_sc || w2 == opt.w << 2) break;
            last_sc = score;
            w2 <<= 1;
        } while (++it < 3 && score < j.truesc - opt.a);
        if (!g.valid && (*ck.flags & OVF_ZARENA)) continue;        // the chunk is re-run with a larger arena
        int n_ops = 0;
        if (g.valid) {
            if (g.fast) n_ops = 1;
            else dev_traceback(g.z, g.n_col, g.qlen, g.tlen, g.w, [&](int, int) { ++n_ops; });
        }
        const unsigned long long need = (unsigned long long)n_ops + 2;
        const unsigned long long base = atomicAdd(ck.cigused, need);
        if (base + need > ck.cigcap) { atomicOr(ck.flags, OVF_CIGAR); continue; }
        uint32_t *cg = ck.cigpool + base + 1;
        if (g.valid) {
            if (g.fast) cg[0] = (uint32_t)g.qlen << 4;
            else { int wp = n_ops; dev_traceback(g.z, g.n_col, g.qlen, g.tlen, g.w, [&](int op, int len) { cg[--wp] = (uint32_t)len << 4 | (uint32_t)op; }); }
            // NM = mismatches in M + inserted + deleted bases (a D that is the first or last op is not counted)
            const bool rev = g.rev;
            int x = 0, y = 0, n_mm = 0, n_gap = 0;
            for (int k = 0; k < n_ops; ++k) {
                const int op = (int)(cg[k] & 0xf), len = (int)(cg[k] >> 4);
                if (op == 0) {
                    for (int u = 0; u < len; ++u) {
                        const int qc = rev ? qseg[lq - 1 - (x + u)] : qseg[x + u];
                        const int tc = rev ? ref_base(R, j.re - 1 - (y + u)) : ref_base(R, j.rb + y + u);
                        if (qc != tc) ++n_mm;
                    }
                    x += len; y += len;
                } else if (op == 2) { if (k > 0 && k < n_ops - 1) n_gap += len; y += len; }
                else if (op == 1) { x += len; n_gap += len; }
            }
            h.nm = n_mm + n_gap;
        }
        dev_finish_hit(R, ck, j, l_query, h, (int64_t)base + 1, n_ops);
        ck.hits[slot] = h;
    }
}
