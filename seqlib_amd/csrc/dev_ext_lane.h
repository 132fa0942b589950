// dev_ext_lane.h -- the ahead-of-time extensions of the heavy reads, ONE LANE PER SEED.
//
// A read inside a repeat family keeps ~500 one-seed chains, and mem_chain2aln extends nearly every one of them (they lie at different
// reference positions, so no earlier region covers them): ~1 000 ksw_extend2 calls of ~60 rows per read, which on the wave-per-read
// kernel is a serial walk of 10-40 ms on one wave (the extension stage's tail) and on the wave-per-seed job kernel (k_extend_cand)
// ~10 000 wave instructions per extension.  The extensions of different seeds do not depend on each other, and for these reads there
// are hundreds of them with the same query: here a wave takes 64 seeds and every lane runs ksw_extend2's scalar loops on its own -- the
// H/E row of a lane lives in LDS, column-major across the lanes (cell j * 64 + lane: no bank conflicts; layouts below) -- ~36 instructions
// per cell for 64 extensions at once instead of ~170 per row for one.
// k_extend_reg then replays mem_chain2aln's decisions and takes the regions from the table (cand), as it does after k_extend_cand.
// /root/reference/src/BWAAligner.cpp:104 -> mem_align1 -> mem_chain2aln -> ksw_extend2.
#pragma once
#include "dev_seed4.h"
#include "dev_ext_reg.h"

struct alignas(8) LaneJob {    // one seed extension (k_cand_lane_prep -> k_ext_lanes)
    int64_t s_rbeg, rmax0, rmax1;
    uint64_t q_off;             // the read's codes
    int l_query, s_qbeg, s_len, rid;
    float frac_rep;
    int out;                    // slot of the region in the table (cand_base[r] + seed index)
    int r, c;                   // read and chain (for seedcov)
};

#define LANE_SCORE_LIMIT 16000  // (read length) x max(mat) has to stay below this for the lane kernel to be used (host check)

// Where a lane keeps its H/E row.  WIDE: one 32-bit word per column -- 14-bit H, 14-bit E, the column's query code above them.  NARROW: when
// no score can reach 256 (150 bp reads with bwa's default scores: read length x max(mat) < 256) a column is 8-bit H + 8-bit E in a 16-bit
// word and the query codes sit apart, eight 4-bit codes per word: half the LDS per wave, twice the waves per CU -- and the kernel's time
// is inversely proportional to its waves per CU (measured by padding the rows: 4 / 3 / 2 waves per CU -> 30.8 / 40.4 / 60.3 ms).
struct LaneWide {
    uint32_t *eh;                                   // this lane's word of column 0; stride WAVE words
    static __host__ __device__ size_t bytes(int cols) { return (size_t)cols * WAVE * 4; }
    __device__ __forceinline__ void init(uint32_t *base, int, int lane) { eh = base + lane; }
    __device__ __forceinline__ void put_all(int j, int h, int e, int q) { eh[j * WAVE] = (uint32_t)h | (uint32_t)e << 14 | (uint32_t)q << 28; }
    __device__ __forceinline__ uint32_t get(int j) const { return eh[j * WAVE]; }
    __device__ __forceinline__ int q_of(uint32_t v, int) const { return (int)(v >> 28); }
    static __device__ __forceinline__ int h_of(uint32_t v) { return (int)(v & 0x3fffu); }
    static __device__ __forceinline__ int e_of(uint32_t v) { return (int)((v >> 14) & 0x3fffu); }
    __device__ __forceinline__ void put(int j, int h, int e, uint32_t old) { eh[j * WAVE] = (uint32_t)h | (uint32_t)e << 14 | (old & 0xf0000000u); }
    static __device__ __forceinline__ bool zero(uint32_t v) { return (v & 0x0fffffffu) == 0; }
};
struct LaneNarrow {
    uint16_t *eh;                                   // 16-bit cells, stride WAVE
    uint32_t *qa;                                   // eight 4-bit query codes per word, stride WAVE words
    static __host__ __device__ size_t bytes(int cols) { return (size_t)cols * WAVE * 2 + (size_t)((cols + 7) / 8) * WAVE * 4; }
    __device__ __forceinline__ void init(uint32_t *base, int cols, int lane) { eh = (uint16_t *)base + lane; qa = base + (size_t)cols * WAVE / 2 + lane; }
    __device__ __forceinline__ void put_all(int j, int h, int e, int q)
    {
        eh[j * WAVE] = (uint16_t)(h | e << 8);
        uint32_t w = (j & 7) ? qa[(j >> 3) * WAVE] : 0u;        // (columns are written in ascending order: a word starts at its column 0)
        w |= (uint32_t)q << ((j & 7) * 4);
        qa[(j >> 3) * WAVE] = w;
    }
    __device__ __forceinline__ uint32_t get(int j) const { return eh[j * WAVE]; }
    __device__ __forceinline__ int q_of(uint32_t, int j) const { return (int)((qa[(j >> 3) * WAVE] >> ((j & 7) * 4)) & 7u); }
    static __device__ __forceinline__ int h_of(uint32_t v) { return (int)(v & 0xffu); }
    static __device__ __forceinline__ int e_of(uint32_t v) { return (int)(v >> 8); }
    __device__ __forceinline__ void put(int j, int h, int e, uint32_t) { eh[j * WAVE] = (uint16_t)(h | e << 8); }
    static __device__ __forceinline__ bool zero(uint32_t v) { return v == 0; }
};

// ksw_extend2, scalar, one extension per lane; L = the lane's row (not initialised by the caller)
template <typename L, typename QF, typename TF>
__device__ ExtResult lane_ksw_extend2(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int w, int end_bonus, int h0, L &row)
{
    const int o_del = o.o_del, e_del = o.e_del, o_ins = o.o_ins, e_ins = o.e_ins, zdrop = o.zdrop;
    const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
    // row -1: eh[0].h = h0, then the insertion ramp while it stays positive; the query code of column j rides along
    for (int j = 0; j <= qlen; ++j) {
        const int v = h0 - oe_ins - (j - 1) * e_ins;
        const int h = j == 0 ? h0 : (v > 0 ? v : 0);
        row.put_all(j, h, 0, j < qlen ? qf(j) : 0);
    }
    int max = 0;
    for (int i = 0; i < 25; ++i) max = max > o.mat[i] ? max : o.mat[i];
    int max_ins = (int)((double)(qlen * max + end_bonus - o_ins) / e_ins + 1.);
    max_ins = max_ins > 1 ? max_ins : 1;
    w = w < max_ins ? w : max_ins;
    int max_del = (int)((double)(qlen * max + end_bonus - o_del) / e_del + 1.);
    max_del = max_del > 1 ? max_del : 1;
    w = w < max_del ? w : max_del;
    const int tail_top = ext_tail_bound0(o, qlen, h0, max);
    max = h0;
    int max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, beg = 0, end = qlen;
    for (int i = 0; i < tlen; ++i) {
        if (i >= qlen && ext_tail_done(tail_top - (i - qlen) * e_del, max, gscore)) break;      // dev_ext_wave.h: rows that cannot matter
        const int t = tf(i);
        const uint32_t rowp = t == 0 ? mr.packed[0] : t == 1 ? mr.packed[1] : t == 2 ? mr.packed[2] : t == 3 ? mr.packed[3] : mr.packed[4];
        const int row4 = t == 0 ? mr.q4[0] : t == 1 ? mr.q4[1] : t == 2 ? mr.q4[2] : t == 3 ? mr.q4[3] : mr.q4[4];
        int f = 0, m = 0, mj = -1;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        int h1 = 0;
        if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; }
        uint32_t cur = beg < end ? row.get(beg) : 0u;
        for (int j = beg; j < end; ++j) {
            const uint32_t nxt = row.get(j + 1);                    // (column j + 1 <= qlen exists; read ahead of this cell's arithmetic)
            int M = L::h_of(cur), e = L::e_of(cur);
            const uint32_t q = (uint32_t)row.q_of(cur, j);
            const int s = q < 4 ? __builtin_amdgcn_sbfe((int)rowp, q << 3, 8u) : row4;
            M = M ? M + s : 0;
            int h = M > e ? M : e;
            h = h > f ? h : f;
            const int hl = h1;                                      // H(i, j-1): what eh[j].h holds for the next row
            h1 = h;
            mj = m > h ? mj : j;
            m = m > h ? m : h;
            int t2 = M - oe_del; t2 = t2 > 0 ? t2 : 0;
            e -= e_del; e = e > t2 ? e : t2;
            row.put(j, hl, e, cur);
            t2 = M - oe_ins; t2 = t2 > 0 ? t2 : 0;
            f -= e_ins; f = f > t2 ? f : t2;
            cur = nxt;
        }
        row.put(end, h1, 0, beg < end ? cur : row.get(end));        // eh[end].h = h1; eh[end].e = 0 (the column keeps its query code)
        if ((end > beg ? end : beg) == qlen) {                       // (the scalar loop's j after its last trip)
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        if (m == 0) break;
        if (m > max) {
            max = m; max_i = i; max_j = mj;
            const int off = mj - i < 0 ? i - mj : mj - i;
            max_off = max_off > off ? max_off : off;
        } else if (zdrop > 0) {
            if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
            else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
        }
        int j;
        for (j = beg; j < end && L::zero(row.get(j)); ++j) {}
        beg = j;
        for (j = end; j >= beg && L::zero(row.get(j)); --j) {}
        end = j + 2 < qlen ? j + 2 : qlen;
    }
    ExtResult r;
    r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
    return r;
}

// dev_extend_core, one seed per lane (the same statements, the scalar extension above in place of the wave-wide one)
template <typename L>
__device__ DReg lane_extend_core(const DevRef &R, const slx_opt &opt, const MatRows &mr, const uint8_t *query, int l_query, int s_qbeg, int s_len,
                                 int64_t s_rbeg, int64_t rmax0, int64_t rmax1, int rid, float frac_rep, L &eh)
{
    DReg a;
    a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0;
    a.n_comp = 0; a.hash = 0;
    int aw0 = opt.w, aw1 = opt.w, i;
    a.w = opt.w; a.score = a.truesc = -1; a.rid = rid;
    RWin rw; rw.bits = 0; rw.chunk = -1;             // the reference through a 32-base window: one 8-byte read per 32 rows
    if (s_qbeg) {
        const int64_t tmp = s_rbeg - rmax0;
        ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
        for (i = 0; i < 2; ++i) {
            const int prev = a.score;
            aw0 = opt.w << i;
            er = lane_ksw_extend2(s_qbeg, [&](int j) { return (int)query[s_qbeg - 1 - j]; }, (int)tmp,
                                  [&](int t) { return text_at(R, s_rbeg - 1 - t, rw); }, opt, mr, aw0, opt.pen_clip5, s_len * opt.a, eh);
            a.score = er.score;
            if (a.score == prev || er.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
        }
        if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip5) { a.qb = s_qbeg - er.qle; a.rb = s_rbeg - er.tle; a.truesc = a.score; }
        else { a.qb = 0; a.rb = s_rbeg - er.gtle; a.truesc = er.gscore; }
    } else { a.score = a.truesc = s_len * opt.a; a.qb = 0; a.rb = s_rbeg; }
    if (s_qbeg + s_len != l_query) {
        const int sc0 = a.score, qe = s_qbeg + s_len;
        const int64_t re0 = s_rbeg + s_len;
        ExtResult er; er.score = -1; er.qle = er.tle = er.gtle = er.gscore = er.max_off = 0;
        for (i = 0; i < 2; ++i) {
            const int prev = a.score;
            aw1 = opt.w << i;
            er = lane_ksw_extend2(l_query - qe, [&](int j) { return (int)query[qe + j]; }, (int)(rmax1 - re0),
                                  [&](int t) { return text_at(R, re0 + t, rw); }, opt, mr, aw1, opt.pen_clip3, sc0, eh);
            a.score = er.score;
            if (a.score == prev || er.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
        }
        if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip3) { a.qe = qe + er.qle; a.re = re0 + er.tle; a.truesc += a.score - sc0; }
        else { a.qe = l_query; a.re = re0 + er.gtle; a.truesc += er.gscore - sc0; }
    } else { a.qe = l_query; a.re = s_rbeg + s_len; }
    a.w = aw0 > aw1 ? aw0 : aw1;
    a.seedlen0 = s_len;
    a.frac_rep = frac_rep;
    return a;
}

// one wave per selected heavy read: a job for every seed of every kept chain (chains across the lanes: the reference window of a chain
// -- rmax, clipped to its contig -- is computed once and copied into its seeds' jobs)
template <int MAXQ>
__global__ void __launch_bounds__(64) k_cand_lane_prep(DevRef R, Chunk ck, DevOpt dopt, const int *heavy, const unsigned int *n_heavy, const unsigned int *job_off, LaneJob *jobs)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    __shared__ int gap_lut[MAXQ + 2];
    for (int q = lane; q < MAXQ + 2; q += WAVE) gap_lut[q] = dev_cal_max_gap(opt, q);
    __syncthreads();
    auto max_gap_of = [&](int q) { return gap_lut[q < 0 ? 0 : (q > MAXQ + 1 ? MAXQ + 1 : q)]; };
    const int64_t l_pac = R.l_pac;
    const unsigned int nh = (unsigned int)__builtin_amdgcn_readfirstlane((int)*n_heavy);
    for (unsigned int s = blockIdx.x; s < nh; s += gridDim.x) {
        const int r = heavy[s];
        const int base = __builtin_amdgcn_readfirstlane(ck.cand_base[r]);
        if (base < 0 || job_off[s + 1] == job_off[s]) continue;
        const unsigned int job0 = job_off[s];
        const ReadWS w = make_ws_uniform(ck, r);
        const uint64_t q_off = ck.offs[r];
        const int l_query = (int)(ck.offs[r + 1] - q_off);
        const int n_chn = ck.n_chain[r];
        const float frac_rep = ck.frac_rep[r];
        unsigned int running = 0;
        for (int cb = 0; cb < n_chn; cb += WAVE) {
            const int ci = cb + lane;
            const int c = ci < n_chn ? w.ia[ci] : -1;
            const int n = c >= 0 ? w.c_n[c] : 0;
            int incl = n;
            for (int d = 1; d < WAVE; d <<= 1) { const int u = __shfl_up(incl, d, WAVE); if (lane >= d) incl += u; }
            const unsigned int my = running + (unsigned int)(incl - n);
            running += (unsigned int)__shfl(incl, WAVE - 1, WAVE);
            if (n <= 0) continue;
            const int *cs = w.c_w + w.c_first[c];
            int64_t rmax0 = l_pac << 1, rmax1 = 0;
            for (int i = 0; i < n; ++i) {
                const int sd = cs[i];
                const int qb = w.s_qbeg(sd), sl = w.s_len(sd);
                const int64_t b = w.s_rbeg[sd] - (qb + max_gap_of(qb));
                const int64_t e = w.s_rbeg[sd] + sl + ((l_query - qb - sl) + max_gap_of(l_query - qb - sl));
                rmax0 = rmax0 < b ? rmax0 : b;
                rmax1 = rmax1 > e ? rmax1 : e;
            }
            rmax0 = rmax0 > 0 ? rmax0 : 0;
            rmax1 = rmax1 < l_pac << 1 ? rmax1 : l_pac << 1;
            const int64_t first_rbeg = w.s_rbeg[cs[0]];
            if (rmax0 < l_pac && l_pac < rmax1) {
                if (first_rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac;
            }
            {
                int is_rev;
                const int rid = dev_pos2rid(R, dev_depos(R, first_rbeg, &is_rev));
                int64_t far_beg = R.ann_off[rid], far_end = far_beg + R.ann_len[rid];
                if (is_rev) { const int64_t t = far_beg; far_beg = (l_pac << 1) - far_end; far_end = (l_pac << 1) - t; }
                rmax0 = rmax0 > far_beg ? rmax0 : far_beg;
                rmax1 = rmax1 < far_end ? rmax1 : far_end;
            }
            const int rid_c = w.c_rid[c];
            for (int i = 0; i < n; ++i) {
                const int sd = cs[i];
                LaneJob j;
                j.s_rbeg = w.s_rbeg[sd]; j.rmax0 = rmax0; j.rmax1 = rmax1; j.q_off = q_off; j.l_query = l_query;
                j.s_qbeg = w.s_qbeg(sd); j.s_len = w.s_len(sd); j.rid = rid_c; j.frac_rep = frac_rep; j.out = base + sd; j.r = r; j.c = c;
                jobs[job0 + my + (unsigned int)i] = j;
            }
        }
    }
}

// (dynamic LDS: columns 0 .. longest query of an extension = longest read - min_seed_len: 34 KB per wave for 150 bp reads in the wide layout -- four waves
// per CU -- 21 KB in the narrow one)
template <typename L>
__global__ void __launch_bounds__(64) k_ext_lanes(DevRef R, Chunk ck, DevOpt dopt, const unsigned int *n_heavy, const unsigned int *job_off, unsigned int *queue,
                                                   const LaneJob *jobs, DReg *cand, int cols)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    extern __shared__ uint32_t lane_lds[];
    L row;
    row.init(lane_lds, cols, lane);
    const unsigned int nh = (unsigned int)__builtin_amdgcn_readfirstlane((int)*n_heavy);
    const unsigned int n_jobs = nh ? (unsigned int)__builtin_amdgcn_readfirstlane((int)job_off[nh]) : 0u;
    for (;;) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(queue, (unsigned int)WAVE);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_jobs) break;
        const unsigned int job = base + (unsigned int)lane;
        if (job < n_jobs) {
            const LaneJob j = jobs[job];
            const uint8_t *query = ck.codes + j.q_off;
            DReg a = lane_extend_core<L>(R, opt, mr, query, j.l_query, j.s_qbeg, j.s_len, j.s_rbeg, j.rmax0, j.rmax1, j.rid, j.frac_rep, row);
            // seedcov: the chain's seeds that lie inside the region
            const ReadWS w = make_ws(ck, j.r);
            const int n = w.c_n[j.c];
            const int *cs = w.c_w + w.c_first[j.c];
            int cov = 0;
            for (int i = 0; i < n; ++i) {
                const int t = cs[i];
                const int t_qbeg = w.s_qbeg(t), t_len = w.s_len(t);
                const int64_t t_rbeg = w.s_rbeg[t];
                if (t_qbeg >= a.qb && t_qbeg + t_len <= a.qe && t_rbeg >= a.rb && t_rbeg + t_len <= a.re) cov += t_len;
            }
            a.seedcov = cov;
            cand[j.out] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------- light reads: the top-seed extensions that need the dynamic program, one lane per job
// k_ext_first gives each of them a wave (~11 000 wave instructions per job of which a band of a few dozen columns uses a fraction of the lanes).  The jobs are independent
// and come with everything in one 64-byte descriptor, so they can run as the heavy reads' seeds do above, 64 per wave on lane_extend_core -- but a wave then lasts as long
// as its longest job, and the work of a job goes with the SQUARE of its extension queries (0 .. read length - seed length bases on either side of the seed).  Hence the
// bins: the jobs k_first_diag left are counting-sorted by sqrt(left^2 + right^2) into 64 classes, most work first, so that the 64 jobs of a wave are alike and the long
// ones start early (three small launches, no host round trip; the order inside a class is whatever the atomics give: every job writes its own table entry).
__device__ __forceinline__ int first_job_bin(const FirstJob &j, int max_len)
{
    const int left = j.s_qbeg, right = j.l_query - j.s_qbeg - j.s_len;
    const float w = sqrtf((float)(left * left + right * right));
    int b = (int)(w * 64.f / (float)(max_len > 0 ? max_len : 1));
    b = b > 63 ? 63 : (b < 0 ? 0 : b);
    return 63 - b;
}
__global__ void __launch_bounds__(256) k_first_bin_count(const FirstJob *jobs, const unsigned int *dp_list, const unsigned int *n_dp, int max_len, unsigned int *bins)
{
    __shared__ unsigned int s[64];
    if (threadIdx.x < 64) s[threadIdx.x] = 0;
    __syncthreads();
    const unsigned int n = *n_dp;
    for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) atomicAdd(&s[first_job_bin(jobs[dp_list[t]], max_len)], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && s[threadIdx.x]) atomicAdd(&bins[threadIdx.x], s[threadIdx.x]);
}
__global__ void k_first_bin_scan(unsigned int *bins)          // bins[64 + b] = where class b starts (then its cursor)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned int acc = 0; for (int b = 0; b < 64; ++b) { bins[64 + b] = acc; acc += bins[b]; } }
}
__global__ void __launch_bounds__(256) k_first_bin_scatter(const FirstJob *jobs, const unsigned int *dp_list, const unsigned int *n_dp, int max_len, unsigned int *bins, unsigned int *out)
{
    __shared__ unsigned int s_cnt[64], s_base[64];
    const unsigned int n = *n_dp;
    for (unsigned int t0 = blockIdx.x * blockDim.x; t0 < n; t0 += gridDim.x * blockDim.x) {          // (block-uniform trip count: the barriers below are reached by all)
        if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const unsigned int t = t0 + threadIdx.x;
        int b = -1;
        unsigned int job = 0, my = 0;
        if (t < n) { job = dp_list[t]; b = first_job_bin(jobs[job], max_len); my = atomicAdd(&s_cnt[b], 1u); }
        __syncthreads();
        if (threadIdx.x < 64 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&bins[64 + threadIdx.x], s_cnt[threadIdx.x]);
        __syncthreads();
        if (b >= 0) out[s_base[b] + my] = job;
        __syncthreads();
    }
}

template <typename L>
__global__ void __launch_bounds__(64) k_first_lanes(DevRef R, Chunk ck, DevOpt dopt, unsigned int *queue, const FirstJob *jobs, DReg *first, const unsigned int *list,
                                                     const unsigned int *n_list, int cols)
{
    const slx_opt &opt = dopt.o;
    const int lane = threadIdx.x;
    const MatRows mr = make_matrows(opt.mat);
    extern __shared__ uint32_t lane_lds[];
    L row;
    row.init(lane_lds, cols, lane);
    const unsigned int n_jobs = (unsigned int)__builtin_amdgcn_readfirstlane((int)*n_list);
    for (;;) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(queue, (unsigned int)WAVE);
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_jobs) break;
        const unsigned int t = base + (unsigned int)lane;
        if (t < n_jobs) {
            const unsigned int job = list[t];
            const FirstJob j = jobs[job];
            const DReg a = lane_extend_core<L>(R, opt, mr, ck.codes + j.q_off, j.l_query, j.s_qbeg, j.s_len, j.s_rbeg, j.rmax0, j.rmax1, j.rid, j.frac_rep, row);
            first[job] = a;                              // seedcov is filled in by k_ext_replay
        }
    }
}

// ---------------------------------------------------------------------------------------------- light reads: the diagonal, one lane per job
// Most top-seed extensions of the light reads are answered by the diagonal alone (diag_extend, dev_ext_reg.h: at most oe - 1 lost against a
// perfect match).  One wave per job that is ~1 000 instructions around two short scans -- 12.8 of k_ext_first's 20 ms per 8.3 M reads;
// one LANE per job it is a loop over the query with both sequences read through register windows.  k_first_diag answers those jobs (the
// region goes straight into the table) and lists the others for k_ext_first, which keeps the dynamic program.
template <typename QF, typename TF>
__device__ __forceinline__ bool lane_diag_extend(int qlen, QF qf, int tlen, TF tf, const slx_opt &o, const MatRows &mr, int amax, int h0, ExtResult &out)
{   // the conditions and the result of diag_extend, computed by one lane
    if (tlen < qlen || qlen > 3 * WAVE || qlen < 1) return false;
    if ((long long)h0 + (long long)qlen * amax >= (1 << 22)) return false;
    const int oe_del = o.o_del + o.e_del, oe_ins = o.o_ins + o.e_ins;
    const int oe = oe_del < oe_ins ? oe_del : oe_ins;
    int loss = 0, run = h0, best = h0, best_p = -1;
    for (int p = 0; p < qlen; ++p) {
        const int t = tf(p), q = qf(p);
        const uint32_t rowp = t == 0 ? mr.packed[0] : t == 1 ? mr.packed[1] : t == 2 ? mr.packed[2] : mr.packed[3];
        const int row4 = t == 0 ? mr.q4[0] : t == 1 ? mr.q4[1] : t == 2 ? mr.q4[2] : mr.q4[3];
        const int sc = q < 4 ? __builtin_amdgcn_sbfe((int)rowp, (uint32_t)q << 3, 8u) : row4;
        loss += amax - sc;
        if (loss > oe - 1) return false;                 // (the loss only grows: amax is the largest entry)
        run += sc;
        if (run > best) { best = run; best_p = p; }      // the largest prefix above h0 at its FIRST position
    }
    if (h0 <= loss) return false;
    out.score = best; out.qle = best_p + 1; out.tle = best_p + 1; out.gtle = qlen; out.gscore = run; out.max_off = 0;
    return true;
}

__global__ void __launch_bounds__(256) k_first_diag(DevRef R, Chunk ck, DevOpt dopt, int n, const unsigned int *first_off, unsigned int cap, const FirstJob *jobs, DReg *first,
                                                    unsigned int *dp_list, unsigned int *n_dp)
{
    const slx_opt &opt = dopt.o;
    const MatRows mr = make_matrows(opt.mat);
    int amax = 0;
    for (int i = 0; i < 25; ++i) amax = amax > opt.mat[i] ? amax : opt.mat[i];
    unsigned int n_jobs = first_off[n];
    if (n_jobs > cap) n_jobs = cap;
    const unsigned int job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= n_jobs) return;
    const FirstJob j = jobs[job];
    if (j.l_query <= 0) return;                           // an empty job (a read whose slots pass the table end): k_extend_reg has the read
    const uint8_t *codes = ck.codes;
    QWin qw; qw.bits = 0; qw.chunk = 0xffffffffu;
    RWin rw; rw.bits = 0; rw.chunk = -1;
    DReg a;
    a.rb = a.re = 0; a.qb = a.qe = 0; a.sub = a.csub = a.sub_n = 0; a.seedcov = 0; a.secondary = 0;
    a.n_comp = 0; a.hash = 0;
    a.score = a.truesc = -1; a.rid = j.rid;
    // the diagonal's max_off is 0, so the band-doubling loop of dev_extend_core ends after its first trip (band opt.w) -- unless the band is so
    // narrow that (w >> 1) + (w >> 2) is 0: then it takes the second trip, finds the same score and ends with band 2 w
    const int aw_ext = ((opt.w >> 1) + (opt.w >> 2)) > 0 ? opt.w : opt.w << 1;
    int aw0 = opt.w, aw1 = opt.w;
    bool ok = true;
    const int s_qbeg = j.s_qbeg, s_len = j.s_len, l_query = j.l_query;
    const int64_t s_rbeg = j.s_rbeg;
    if (s_qbeg) {
        aw0 = aw_ext;
        ExtResult er;
        ok = lane_diag_extend(s_qbeg, [&](int p) { return q_at(codes, j.q_off + (uint64_t)(s_qbeg - 1 - p), qw); }, (int)(s_rbeg - j.rmax0),
                              [&](int t) { return text_at(R, s_rbeg - 1 - t, rw); }, opt, mr, amax, s_len * opt.a, er);
        if (ok) {
            a.score = er.score;
            if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip5) { a.qb = s_qbeg - er.qle; a.rb = s_rbeg - er.tle; a.truesc = a.score; }
            else { a.qb = 0; a.rb = s_rbeg - er.gtle; a.truesc = er.gscore; }
        }
    } else { a.score = a.truesc = s_len * opt.a; a.qb = 0; a.rb = s_rbeg; }
    if (ok) {
        if (s_qbeg + s_len != l_query) {
            aw1 = aw_ext;
            const int sc0 = a.score, qe = s_qbeg + s_len;
            const int64_t re0 = s_rbeg + s_len;
            ExtResult er;
            ok = lane_diag_extend(l_query - qe, [&](int p) { return q_at(codes, j.q_off + (uint64_t)(qe + p), qw); }, (int)(j.rmax1 - re0),
                                  [&](int t) { return text_at(R, re0 + t, rw); }, opt, mr, amax, sc0, er);
            if (ok) {
                a.score = er.score;
                if (er.gscore <= 0 || er.gscore <= a.score - opt.pen_clip3) { a.qe = qe + er.qle; a.re = re0 + er.tle; a.truesc += a.score - sc0; }
                else { a.qe = l_query; a.re = re0 + er.gtle; a.truesc += er.gscore - sc0; }
            }
        } else { a.qe = l_query; a.re = s_rbeg + s_len; }
    }
    if (ok) {
        a.w = aw0 > aw1 ? aw0 : aw1;
        a.seedlen0 = s_len;
        a.frac_rep = j.frac_rep;
        first[job] = a;                                   // seedcov is filled in by k_ext_replay
    } else dp_list[wave_fetch_inc(n_dp)] = job;
}
