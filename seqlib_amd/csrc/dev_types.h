// dev_types.h -- device-side views of one chunk of reads and its per-read work areas.
// Vocabulary follows bwa / SeqLib: intervals (SMEMs), seeds, chains, regions (mem_alnreg_t), hits
// (mem_aln_t after the glue's sort + filters).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "seqlib_amd.h"

// overflow / error bits raised by kernels for a chunk (the host re-runs the chunk with larger caps)
enum : uint32_t {
    OVF_INTV = 1u << 0,      // more kept intervals than cap_intv
    OVF_LIST = 1u << 1,      // SMEM work list longer than cap_list
    OVF_ZARENA = 1u << 2,    // traceback arena exhausted
    OVF_CIGAR = 1u << 3,     // cigar pool exhausted
    ERR_LOGLUT = 1u << 4,    // log() argument outside the host-built table
    ERR_INTERNAL = 1u << 5,
};

struct DevRef {               // packed reference + contig table (bntseq_t)
    const uint8_t *pac;       // forward strand, 2 bit/base
    int64_t l_pac;
    int n_seqs;
    const int64_t *ann_off;   // n_seqs
    const int32_t *ann_len;
    const uint8_t *ann_alt;   // bntann1_t::is_alt per contig (from <prefix>.alt); null when the index has no ALT contig
};

__device__ __forceinline__ int ref_is_alt(const DevRef &R, int rid) { return R.ann_alt && rid >= 0 ? (int)R.ann_alt[rid] : 0; }

struct DevOpt {               // mem_opt_t as the kernels see it (plain copy of slx_opt)
    slx_opt o;
};

struct alignas(8) DReg {      // mem_alnreg_t
    int64_t rb, re;
    int qb, qe;
    int rid;
    int score, truesc, sub, csub, sub_n;
    int w, seedcov, secondary, seedlen0;
    int n_comp;
    float frac_rep;
    uint64_t hash;
};

struct alignas(8) DHit {      // mem_aln_t after reg2aln
    int64_t pos;
    int rid;
    int flag;                 // 0x100 secondary, 0x10 reverse already folded in
    int mapq;
    int score;
    int nm;
    int n_cigar;
    int64_t cig_start;        // word offset in the cigar pool
};

struct alignas(8) FirstJob {   // one top-seed extension of a light read (k_first_prep -> k_ext_first)
    int64_t s_rbeg, rmax0, rmax1;
    uint64_t q_off;             // the read's codes
    int l_query, s_qbeg, s_len, rid;
    float frac_rep;
    int pad;
};

// Long reads (contigs), extension in rounds (k_extend_reg<MAXQ > 704>, slx_align.hip: launch_tail).  What extending a seed yields depends
// on the seed and its chain only, never on the regions found before it -- only WHETHER mem_chain2aln extends it does.  So the per-read
// walk does no dynamic programming itself: where it needs the region of a seed it takes it from the memo (memo_idx[seed slot] -> memo_tab),
// and where the memo has none it emits a job (memo_jobs, round_list), treats the seed as extended-with-unknown-region and walks on to
// find the other seeds it will probably need (at most `budget` per read and round; a seed close to the diagonal of a pending one is
// taken to be covered by it).  The jobs of a round run one wave each (k_ext_first), then the reads that missed anything are walked again
// from the start.  A walk that misses nothing has made every decision on real regions: it IS mem_chain2aln's, and only such a walk
// writes the read's result.
struct ExtSpec {
    int *memo_idx = nullptr;           // per seed slot (ck.seed_off[r] + s): job index, or -1
    FirstJob *memo_jobs = nullptr;     // job descriptors, one per emitted seed
    const DReg *memo_tab = nullptr;    // their regions (seedcov not filled in)
    unsigned int *n_jobs = nullptr;    // jobs emitted so far (all rounds)
    unsigned int *round_list = nullptr, *n_round = nullptr;     // this round's jobs
    int *todo_next = nullptr;          // reads that missed something this round
    unsigned int *n_todo_next = nullptr;
    int budget = 0;                    // 0: this mode is off
    int predict = 0;                   // 1 (the first round): the top seed of a chain, not extended yet, is given a guessed region (its diagonal, the whole read)
    int guess = 0;                     // 1: once a seed is pending, seeds within the band of its diagonal are taken to be covered by its region
};

// one chunk of reads, all device pointers
struct Chunk {
    int n_reads;
    int spread;               // 1, or 64 for a small chunk: the lane-per-read kernels of seeding put ONE read on each wave (read = thread / 64, lane 0 works)
    const uint8_t *codes;     // nt4 codes, reads concatenated
    const uint64_t *offs;     // n_reads + 1 (relative to codes)
    uint64_t first_ordinal;   // lrand48 draw index of read 0
    uint64_t rng_state;
    // stage 1 -> 2
    int cap_intv;
    uint32_t *intv_n;         // [n_reads]
    uint32_t *intv_info;      // [n_reads * cap_intv]  qbeg << 16 | qend
    void *intv_x0;            // idx_t
    void *intv_x2;            // idx_t
    int32_t *l_rep;           // [n_reads]
    unsigned long long *seed_cnt; // [n_reads + 1] upper bound of seeds this read will generate
    uint64_t *seed_off;       // [n_reads + 1] exclusive scan of seed_cnt
    // per-seed-slot work areas (indexed seed_off[r] + i)
    int64_t *s_rbeg;
    uint32_t *s_ql;           // qbeg << 16 | len
    int32_t *s_next;
    int32_t *s_score;         // mem_seed_t::score, only for chunks with reads long enough for mem_flt_chained_seeds (else null: score = length)
    int64_t *c_pos;
    int32_t *c_head, *c_tail, *c_n, *c_rid, *c_w, *c_first;
    int8_t *c_kept;
    int32_t *ia, *ib, *ic;    // int scratch lists (orders, kept lists, sort handles)
    uint64_t *srt;            // seed sort keys
    DReg *regs;
    DHit *hits;
    // regions of heavy reads' chains extended ahead of time (k_extend_cand): cand[cand_base[r] + chain index], cand_base[r] < 0 = none
    const DReg *cand;
    const int32_t *cand_base;
    unsigned long long *dbg_cyc;   // SLX_DEBUG_CYC: cycles a kernel spent on each read, 4 arrays of n_reads (null otherwise)
    int dbg_stage;                 // 1 = the extension kernel fills them, 2 = the region kernel
    // per-read results
    int32_t *n_chain;         // kept chains
    int32_t *n_reg;
    int32_t *n_hit;
    int32_t *na;              // regs.n after dedup (NA tag)
    float *frac_rep;
    // arenas
    uint8_t *zarena; unsigned long long zcap; unsigned long long *zused;
    uint32_t *cigpool; unsigned long long cigcap; unsigned long long *cigused;
    // misc
    const double *log_lut; int log_lut_n;
    uint32_t *flags;          // OVF_* bits
    // SMEM work lists: [2 lists][cap_list][n_threads] interleaved by thread
    void *lists; int cap_list; int n_threads;
    // long-read chunks: per-thread H/E rows of the lane-per-read alignment kernels, 2 * long_stride ints per thread
    int *long_scratch; int long_stride; int long_threads;
    // chunks with a read beyond the LDS rows of k_extend_reg (8 004 columns): three rows of huge_stride ints per block of that kernel, in HBM
    int *huge_rows; int huge_stride;
    // glue parameters (src/BWAAligner.cpp:89-95)
    int hardclip; double keepSecFrac; int maxSecondary;
    // SLX_F_REG2SAM: bwa's own record selection (mem_reg2sam / mem_gen_alt) instead of the SeqLib glue's sort + filters
    int sam_mode;
};

// One atomic per wave instead of one per lane: the lanes that reach this call together (whatever subset of the wave that
// is) share a single fetch-add on the wave-uniform counter and take consecutive values.  3 M same-address atomics in a
// kernel otherwise serialise in one L2 channel (measured: 35 ms for a 3.3 M-read launch).
__device__ __forceinline__ uint32_t wave_fetch_inc(uint32_t *ctr)
{
    const unsigned long long m = __ballot(1);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t base = 0;
    if (rank == 0) base = atomicAdd(ctr, (uint32_t)__popcll(m));
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + rank;
}
__device__ __forceinline__ unsigned long long wave_fetch_add_u64(unsigned long long *ctr, unsigned long long each)
{
    const unsigned long long m = __ballot(1);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    unsigned long long base = 0;
    if (rank == 0) base = atomicAdd(ctr, each * (unsigned long long)__popcll(m));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    return (((unsigned long long)hi << 32) | lo) + each * rank;
}

__device__ __forceinline__ int ref_base(const DevRef &R, int64_t p)
{   // base at coordinate p of forward ++ reverse-complement (bns_get_seq)
    if (p >= R.l_pac) { int64_t f = (R.l_pac << 1) - 1 - p; return 3 - ((R.pac[f >> 2] >> ((~f & 3) << 1)) & 3); }
    return (R.pac[p >> 2] >> ((~p & 3) << 1)) & 3;
}

__device__ __forceinline__ int64_t dev_depos(const DevRef &R, int64_t pos, int *is_rev)
{
    return (*is_rev = (pos >= R.l_pac)) ? (R.l_pac << 1) - 1 - pos : pos;
}

__device__ inline int dev_pos2rid(const DevRef &R, int64_t pos_f)
{   // bns_pos2rid
    if (pos_f >= R.l_pac) return -1;
    int left = 0, mid = 0, right = R.n_seqs;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= R.ann_off[mid]) {
            if (mid == R.n_seqs - 1) break;
            if (pos_f < R.ann_off[mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

__device__ inline int dev_intv2rid(const DevRef &R, int64_t rb, int64_t re)
{   // bns_intv2rid
    int is_rev;
    if (rb < R.l_pac && re > R.l_pac) return -2;
    int rid_b = dev_pos2rid(R, dev_depos(R, rb, &is_rev));
    int rid_e = rb < re ? dev_pos2rid(R, dev_depos(R, re - 1, &is_rev)) : rid_b;
    return rid_b == rid_e ? rid_b : -1;
}

// mem_flt_chained_seeds' guard (bwamem.c): the seed filter runs when  min_l <= MEM_SEEDSW_COEF * l_query,
// min_l = min_chain_weight ? MEM_HSP_COEF * min_chain_weight : MEM_MINSC_COEF * log(l_query)  (reads >= ~727 bp at the defaults).
// C's promotions kept: float * int -> float, float * double -> double, double > float compared in double.
#define MEM_SHORT_EXT 50
#define MEM_SHORT_LEN 200
__host__ __device__ __forceinline__ bool flt_live(const slx_opt &o, int l_query, double log_l, int *min_hsp_score)
{
    const double min_l = o.min_chain_weight ? (double)(1.1f * (float)o.min_chain_weight) : (double)5.5f * log_l;
    if (min_hsp_score) *min_hsp_score = (int)(o.a * min_l + .499);
    return !(min_l > (double)(0.05f * (float)l_query));
}

__device__ __forceinline__ int dev_cal_max_gap(const slx_opt &o, int qlen)
{
    int l_del = (int)((double)(qlen * o.a - o.o_del) / o.e_del + 1.);
    int l_ins = (int)((double)(qlen * o.a - o.o_ins) / o.e_ins + 1.);
    int l = l_del > l_ins ? l_del : l_ins;
    l = l > 1 ? l : 1;
    return l < o.w << 1 ? l : o.w << 1;
}
