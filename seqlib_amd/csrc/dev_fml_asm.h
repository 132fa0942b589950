// dev_fml_asm.h -- device side of fml_assemble's data-parallel half: the overlap graph of a window's reads.
//
// Reference behaviour: what fermi-lite's fml_seq2fmi + fml_fmi2mag compute for /root/reference/src/FermiAssembler.cpp:26-31,140-143 --
// which reads are duplicates of or contained in others, every exact suffix-prefix overlap of at least min_asm_ovlp bases, and of those
// the irreducible ones -- as DEFINED in DESIGN.md section 8 (fermi-lite is not in the reference tree).
//
// fermi walks an FM-index of the reads one read at a time.  Here the same relations come out of a JOIN: every string (a read or its
// reverse complement, 1 byte per base in one flat text) is keyed by its first 16 bases, the keys are radix-sorted, and ONE LANE PER TEXT
// POSITION looks its own 16-mer up among the keys and verifies the candidates base by base: a string that ends inside the one it was found
// in is contained (or, over the full length at position 0, a duplicate), one that runs past its end is an overlap.  Transitive
// reduction is one wave per vertex over its sorted overlap list.  Everything is per window through a 16-bit window tag in the key.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FML_SEED_MAX 16          // bases per seed key: 3 bits each + the window tag in the top 16 bits

struct FmlStr {                  // one string of the text
    unsigned long long off;      // first byte in the text
    unsigned long long src;      // first base of the kept stretch of its read in the ASCII text
    int len;
    int win;
};

struct FmlEdge { int v, len; };

__device__ __forceinline__ int fml_code6(int c)
{
    c &= 0xdf;
    return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 5;
}

// the text: string 2 i = kept stretch of read i, string 2 i + 1 = its reverse complement; one wave per string
static __global__ void __launch_bounds__(256) k_asm_strings(const char *bases, const FmlStr *strs, long long n_str, unsigned char *text)
{
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_str) return;
    const FmlStr s = strs[t];
    for (int i = threadIdx.x & 63; i < s.len; i += 64) {
        int c;
        if (t & 1) { c = fml_code6((unsigned char)bases[s.src + (unsigned long long)(s.len - 1 - i)]); c = c < 5 ? 5 - c : 5; }
        else c = fml_code6((unsigned char)bases[s.src + (unsigned long long)i]);
        text[s.off + (unsigned long long)i] = (unsigned char)c;
    }
}

__device__ __forceinline__ unsigned long long fml_seed_key(const unsigned char *p, int kk, int win)
{
    unsigned long long k = 0;
    for (int i = 0; i < kk; ++i) k = k << 3 | p[i];
    return k | (unsigned long long)win << 48;
}

static __global__ void __launch_bounds__(256) k_asm_keys(const unsigned char *text, const FmlStr *strs, long long n_str, int kk, unsigned long long *keys, unsigned int *vals,
                                                         int *rep, unsigned char *contained)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_str) return;
    const FmlStr s = strs[t];
    keys[t] = fml_seed_key(text + s.off, kk, s.win);
    vals[t] = (unsigned int)t;
    rep[t] = (int)t;
    // a string equal to its own reverse complement is no vertex
    const FmlStr o = strs[t ^ 1];
    bool same = true;
    for (int i = 0; i < s.len && same; ++i) same = text[s.off + i] == text[o.off + i];
    contained[t] = same ? 1 : 0;
}

struct FmlTriple { int u, v, len; };

// An index over the sorted seed keys: key -> position of its first occurrence, open addressing (slot key + 1, 0 = empty).  One probe
// sequence replaces a 20-step binary search per text position; most positions' seeds start no string and end at the first empty slot.
static __global__ void __launch_bounds__(256) k_asm_index(const unsigned long long *keys, long long n_str, unsigned long long *hkey, unsigned int *hval, unsigned int hmask)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_str) return;
    const unsigned long long key = keys[i];
    if (i > 0 && keys[i - 1] == key) return;
    unsigned int h = (unsigned int)(key * 0x9E3779B97F4A7C15ULL >> 32) & hmask;
    while (atomicCAS(&hkey[h], 0ULL, key + 1) != 0) h = (h + 1) & hmask;          // keys are distinct here: a slot taken is another key's
    hval[h] = (unsigned int)i;
}

__device__ __forceinline__ long long fml_index_find(const unsigned long long *hkey, const unsigned int *hval, unsigned int hmask, unsigned long long key)
{
    unsigned int h = (unsigned int)(key * 0x9E3779B97F4A7C15ULL >> 32) & hmask;
    while (true) {
        const unsigned long long k = hkey[h];
        if (k == key + 1) return (long long)hval[h];
        if (k == 0) return -1;
        h = (h + 1) & hmask;
    }
}

__device__ __forceinline__ unsigned long long fml_load8(const unsigned char *p)          // 8 bytes from any address
{
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// equal over [from, to)?  32 bytes per step -- eight independent loads in flight, one branch -- then 8, then a masked tail (both texts
// have 8 readable bytes past any string: the buffers are padded).  A candidate that shares its 16-base seed with the text nearly
// always matches to the end, so the loop is a chain of round trips to memory: fewer, wider steps are what count.
__device__ __forceinline__ bool fml_same(const unsigned char *x, const unsigned char *y, int from, int to)
{
    int i = from;
    for (; i + 32 <= to; i += 32) {
        const unsigned long long a0 = fml_load8(x + i), a1 = fml_load8(x + i + 8), a2 = fml_load8(x + i + 16), a3 = fml_load8(x + i + 24);
        const unsigned long long b0 = fml_load8(y + i), b1 = fml_load8(y + i + 8), b2 = fml_load8(y + i + 16), b3 = fml_load8(y + i + 24);
        if ((a0 ^ b0) | (a1 ^ b1) | (a2 ^ b2) | (a3 ^ b3)) return false;
    }
    for (; i + 8 <= to; i += 8)
        if (fml_load8(x + i) != fml_load8(y + i)) return false;
    if (i < to) {
        const unsigned long long m = ~0ULL >> (8 * (8 - (to - i)));
        if ((fml_load8(x + i) ^ fml_load8(y + i)) & m) return false;
    }
    return true;
}

// The join, one wave per string u, one lane per position p of it: the strings whose first kk bases are u[p .. p + kk) come from a binary
// search in the sorted keys and are verified against u[p ..).  A string that ends inside u is contained (over the whole of u at p = 0: a
// duplicate, the smaller index stands for both); one that runs past the end of u overlaps it by |u| - p bases.  Overlaps leave as
// (u, v, length) triples, a wave reserving room for its lanes' finds with one atomic; cnt[u] counts them per source.
#define FML_TRI_CHUNK 1024          // triples a wave reserves at a time: one atomic on the shared counter per chunk, not per round (12.7 M strings x 2 rounds of
                                    // same-address atomics serialise in one L2 channel at ~10 ns each: that, not the probing, was 3/4 of this kernel's time)
static __global__ void __launch_bounds__(256) k_asm_join(const unsigned char *text, const FmlStr *strs, long long n_str, int kk, int min_match,
                                                         const unsigned long long *keys, const unsigned int *vals, const unsigned long long *hkey, const unsigned int *hval, unsigned int hmask,
                                                         int *rep, unsigned char *contained, unsigned int *cnt, FmlTriple *tri, unsigned long long tri_cap, unsigned long long *tri_n)
{
    const int lane = threadIdx.x & 63;
    unsigned long long my_base = 0;          // this wave's stretch of tri[] (wave-uniform)
    unsigned int my_left = 0;
    for (long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); u < n_str; u += (long long)gridDim.x * 4) {
    const FmlStr su = strs[u];
    const unsigned char *ut = text + su.off;
    unsigned int n_mine = 0;
    for (int p0 = 0; p0 + min_match <= su.len; p0 += 64) {
        const int p = p0 + lane, rest = su.len - p;
        FmlTriple found[4];
        int nf = 0;
        if (rest >= min_match) {
            const unsigned long long key = fml_seed_key(ut + p, kk, su.win);
            long long a = fml_index_find(hkey, hval, hmask, key);
            for (; a >= 0 && a < n_str && keys[a] == key; ++a) {
                const int v = (int)vals[a];
                if (v == (int)u) continue;
                const FmlStr sv = strs[v];
                const int m = rest < sv.len ? rest : sv.len;
                if (!fml_same(ut + p, text + sv.off, kk, m)) continue;
                if (sv.len <= rest) {          // v lies inside u
                    if (p == 0 && sv.len == su.len) { if (v < (int)u) atomicMin(&rep[u], v); }
                    else contained[v] = 1;
                } else if (p > 0) {          // v runs past the end of u
                    if (nf < 4) found[nf] = FmlTriple{(int)u, v, rest};
                    else { const unsigned long long at = atomicAdd(tri_n, 1ULL); if (at < tri_cap) tri[at] = FmlTriple{(int)u, v, rest}; }
                    ++nf;
                }
            }
        }
        // the wave's finds of this round (at most four per lane; the rare rest went out one by one above) into the wave's own stretch
        const int mine = nf < 4 ? nf : 4;
        int incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        const int total = __shfl(incl, 63);
        if (total) {
            if ((unsigned int)total > my_left) {          // a new stretch; what is left of the old one is marked empty (u = -1: k_asm_scatter skips it)
                for (unsigned int i = (unsigned int)lane; i < my_left; i += 64) if (my_base + i < tri_cap) tri[my_base + i] = FmlTriple{-1, 0, 0};
                unsigned long long nb = 0;
                const unsigned int want = total > FML_TRI_CHUNK ? (unsigned int)total : FML_TRI_CHUNK;
                if (lane == 0) nb = atomicAdd(tri_n, (unsigned long long)want);
                my_base = __shfl(nb, 0); my_left = want;
            }
            const unsigned long long base = my_base + (unsigned long long)(incl - mine);
            for (int j = 0; j < mine; ++j) if (base + j < tri_cap) tri[base + j] = found[j];
            my_base += (unsigned long long)total; my_left -= (unsigned int)total;
        }
        n_mine += (unsigned int)nf;
    }
    for (int o = 32; o > 0; o >>= 1) n_mine += __shfl_xor(n_mine, o);
    if (lane == 0 && n_mine) cnt[u] = n_mine;
    }
    for (unsigned int i = (unsigned int)lane; i < my_left; i += 64) if (my_base + i < tri_cap) tri[my_base + i] = FmlTriple{-1, 0, 0};
}

// the overlaps between vertices, grouped by source (eoff = exclusive scan of cnt)
static __global__ void __launch_bounds__(256) k_asm_scatter(const FmlTriple *tri, unsigned long long n_tri, const int *rep, const unsigned char *contained,
                                                            const unsigned long long *eoff, unsigned int *cur, FmlEdge *edges)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tri) return;
    const FmlTriple t = tri[i];
    if (t.u < 0) return;          // (the unused end of a wave's stretch)
    if (rep[t.u] != t.u || contained[t.u] || rep[t.v] != t.v || contained[t.v]) return;
    const unsigned int slot = atomicAdd(&cur[t.u], 1u);
    edges[eoff[t.u] + slot] = FmlEdge{t.v, t.len};
}

// Transitive reduction, one wave per source vertex: edges ordered by (overlap descending, target ascending), one edge per target (the
// longest), and u -> v_j dropped when a longer overlap u -> v_k exists whose string agrees with v_j's wherever both lie beyond the end of
// u.  Up to 64 edges (the rule: ~coverage x (1 - min_overlap / read length)) live one per lane in registers -- ranks by 64 shuffles, the
// sort as one cross-lane push, the witnesses' descriptors by shuffle; more go through memory (k_asm_reduce_big).  The irreducible edges of
// u go to out[] at an offset reserved with one atomic per wave; n_irr / irr_off say where.
#define FML_OUT_CHUNK 128           // irreducible edges a wave reserves at a time (as FML_TRI_CHUNK: one same-address atomic per chunk instead of one per vertex)
static __global__ void __launch_bounds__(256) k_asm_reduce(const unsigned char *text, const FmlStr *strs, long long n_str, const unsigned long long *eoff, const unsigned int *cur,
                                                           const FmlEdge *edges, unsigned int *n_irr, unsigned long long *irr_off, FmlEdge *out, unsigned long long *out_n,
                                                           int *big_list, unsigned int *n_big)
{
    const int lane = threadIdx.x & 63;
    unsigned long long my_base = 0;
    unsigned int my_left = 0;
    for (long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); u < n_str; u += (long long)gridDim.x * 4) {
    const int d = (int)cur[u];
    if (d == 0) { if (lane == 0) { n_irr[u] = 0; irr_off[u] = 0; } continue; }
    if (d > 64) { if (lane == 0) big_list[atomicAdd(n_big, 1u)] = (int)u; continue; }
    const FmlEdge me = lane < d ? edges[eoff[u] + (unsigned long long)lane] : FmlEdge{0x7fffffff, -1};
    int r = 0;
    for (int j = 0; j < d; ++j) {
        const int ov = __shfl(me.v, j), ol = __shfl(me.len, j);
        r += (ol > me.len) || (ol == me.len && ov < me.v);
    }
    if (lane >= d) r = lane;
    // lane r receives the edge of rank r
    const int sv = __builtin_amdgcn_ds_permute(r << 2, me.v), sl = __builtin_amdgcn_ds_permute(r << 2, me.len);
    bool dup = false;
    for (int k = 0; k < d; ++k) { const int vk = __shfl(sv, k); dup |= (k < lane && vk == sv); }
    const FmlStr su = strs[u];
    FmlStr sj = su;
    if (lane < d) sj = strs[sv];
    const int aj = su.len - sl;
    bool drop = false;
    for (int k = 0; k < d; ++k) {
        const int lk = __shfl(sl, k), slen_k = __shfl(sj.len, k);
        const unsigned long long off_k = __shfl(sj.off, k);
        const bool dup_k = __shfl((int)dup, k) != 0;
        if (k < lane && lane < d && !dup && !drop && !dup_k && lk != sl) {
            const int ak = su.len - lk, end_k = ak + slen_k, end_j = aj + sj.len;
            const int end = end_k < end_j ? end_k : end_j;
            // both strings equal u up to its end: compare what lies beyond
            if (fml_same(text + off_k + (su.len - ak), text + sj.off + (su.len - aj), 0, end - su.len)) drop = true;
        }
    }
    const bool keep = lane < d && !dup && !drop;
    const unsigned long long m = __ballot(keep);
    const int n_keep = __popcll(m);
    if ((unsigned int)n_keep > my_left) {
        unsigned long long nb = 0;
        if (lane == 0) nb = atomicAdd(out_n, (unsigned long long)FML_OUT_CHUNK);
        my_base = __shfl(nb, 0); my_left = FML_OUT_CHUNK;
    }
    if (lane == 0) { n_irr[u] = (unsigned int)n_keep; irr_off[u] = my_base; }
    if (keep) out[my_base + __popcll(m & ((1ULL << lane) - 1))] = FmlEdge{sv, sl};
    my_base += (unsigned long long)n_keep; my_left -= (unsigned int)n_keep;
    }
}

// The same for the vertices with more than 64 overlaps (reads inside repeats and low-complexity tracts: hundreds to thousands of
// overlaps each): one BLOCK per vertex, the edge list in LDS.  Rank and "a longer overlap with the same target exists" come out of one
// all-against-all pass (both are order-free); the witnesses of the reduction are tried longest overlap first.  More than
// FML_BIG_CAP edges: k_asm_huge_*, through memory.
#define FML_BIG_CAP 4096
static __global__ void __launch_bounds__(256) k_asm_reduce_big(const unsigned char *text, const FmlStr *strs, const int *big_list, const unsigned int *n_big,
                                                               const unsigned long long *eoff, const unsigned int *cur, const FmlEdge *edges,
                                                               unsigned int *n_irr, unsigned long long *irr_off, FmlEdge *out, unsigned long long *out_n,
                                                               int *huge_list, unsigned int *n_huge, int big_cap)
{
    __shared__ int ev[FML_BIG_CAP], el[FML_BIG_CAP], sv[FML_BIG_CAP], sl[FML_BIG_CAP];
    __shared__ unsigned char fl[FML_BIG_CAP];
    __shared__ int wave_cnt[4];
    __shared__ unsigned long long at_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (blockIdx.x >= *n_big) return;
    const long long u = big_list[blockIdx.x];
    const int d = (int)cur[u];
    if (d > big_cap) { if (tid == 0) huge_list[atomicAdd(n_huge, 1u)] = (int)u; return; }
    const FmlEdge *e = edges + eoff[u];
    for (int i = tid; i < d; i += 256) { const FmlEdge x = e[i]; ev[i] = x.v; el[i] = x.len; }
    __syncthreads();
    for (int i = tid; i < d; i += 256) {
        const int v = ev[i], l = el[i];
        int r = 0, dup = 0;
        for (int j = 0; j < d; ++j) {
            const int ov = ev[j], ol = el[j];
            r += (ol > l) || (ol == l && ov < v);
            dup |= (ov == v && ol > l);
        }
        sv[r] = v; sl[r] = l; fl[r] = (unsigned char)dup;
    }
    __syncthreads();
    const FmlStr su = strs[u];
    for (int j = tid; j < d; j += 256) {
        if (fl[j] == 1) continue;
        const int lj = sl[j];
        const FmlStr sj = strs[sv[j]];
        const int aj = su.len - lj;
        bool drop = false;
        for (int k = 0; k < j && !drop; ++k) {
            if (fl[k] == 1 || sl[k] == lj) continue;          // (fl[k] may be turning from 0 into 2 right now: either reads as "not a copy")
            const FmlStr sk = strs[sv[k]];
            const int ak = su.len - sl[k], end_k = ak + sk.len, end_j = aj + sj.len;
            const int end = end_k < end_j ? end_k : end_j;
            drop = fml_same(text + sk.off + (su.len - ak), text + sj.off + (su.len - aj), 0, end - su.len);
        }
        if (drop) fl[j] = 2;
    }
    __syncthreads();
    int keep = 0;
    for (int j = tid; j < d; j += 256) keep += fl[j] == 0;
    for (int o = 32; o > 0; o >>= 1) keep += __shfl_xor(keep, o);
    if (lane == 0) wave_cnt[wv] = keep;
    __syncthreads();
    if (tid == 0) {
        const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        at_s = atomicAdd(out_n, (unsigned long long)tot);
        n_irr[u] = (unsigned int)tot; irr_off[u] = at_s;
    }
    __syncthreads();
    int done = 0;
    for (int j0 = 0; j0 < d; j0 += 256) {          // in sorted order: per round the waves' counts, then a ballot prefix inside each wave
        const int j = j0 + tid;
        const bool k = j < d && fl[j] == 0;
        const unsigned long long m = __ballot(k);
        __syncthreads();
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w2 = 0; w2 < wv; ++w2) before += wave_cnt[w2];
        if (k) out[at_s + done + before + __popcll(m & ((1ULL << lane) - 1))] = FmlEdge{sv[j], sl[j]};
        done += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

// The same through memory, for a vertex with more overlaps than k_asm_reduce_big's LDS holds (a few dozen per batch, 4 000 - 10 000 overlaps each: reads
// inside low-complexity tracts).  Rank, "a longer overlap with the same target exists" and the witness search are all-against-all over one vertex's
// list: a vertex is FML_HUGE_SLICES blocks in each of three launches (a launch per phase: each needs the one before it complete), then one wave
// writes its irreducible edges out in order.
#define FML_HUGE_SLICES 16
static __global__ void __launch_bounds__(256) k_asm_huge_rank(const int *list, const unsigned long long *eoff, const unsigned int *cur, const FmlEdge *edges, FmlEdge *sorted)
{
    const long long u = list[blockIdx.x];
    const int d = (int)cur[u];
    const FmlEdge *e = edges + eoff[u];
    FmlEdge *s = sorted + eoff[u];
    for (int i = (int)blockIdx.y * 256 + (int)threadIdx.x; i < d; i += 256 * FML_HUGE_SLICES) {
        const FmlEdge me = e[i];
        int r = 0;
        for (int j = 0; j < d; ++j) {
            const FmlEdge o = e[j];
            r += (o.len > me.len) || (o.len == me.len && o.v < me.v);
        }
        s[r] = me;
    }
}
static __global__ void __launch_bounds__(256) k_asm_huge_dup(const int *list, const unsigned long long *eoff, const unsigned int *cur, const FmlEdge *sorted, unsigned char *flags)
{
    const long long u = list[blockIdx.x];
    const int d = (int)cur[u];
    const FmlEdge *s = sorted + eoff[u];
    unsigned char *fl = flags + eoff[u];
    for (int j = (int)blockIdx.y * 256 + (int)threadIdx.x; j < d; j += 256 * FML_HUGE_SLICES) {
        const int vj = s[j].v;
        int f = 0;
        for (int k = 0; k < j; ++k)
            if (s[k].v == vj) { f = 1; break; }
        fl[j] = (unsigned char)f;
    }
}
static __global__ void __launch_bounds__(256) k_asm_huge_witness(const unsigned char *text, const FmlStr *strs, const int *list, const unsigned long long *eoff, const unsigned int *cur,
                                                                 const FmlEdge *sorted, unsigned char *flags)
{
    const long long u = list[blockIdx.x];
    const int d = (int)cur[u];
    const FmlEdge *s = sorted + eoff[u];
    unsigned char *fl = flags + eoff[u];
    const FmlStr su = strs[u];
    for (int j = (int)blockIdx.y * 256 + (int)threadIdx.x; j < d; j += 256 * FML_HUGE_SLICES) {
        if (fl[j] == 1) continue;
        const FmlEdge ej = s[j];
        const FmlStr sj = strs[ej.v];
        const int aj = su.len - ej.len;
        bool drop = false;
        for (int k = 0; k < j && !drop; ++k) {
            if (fl[k] == 1) continue;          // (another thread may be turning fl[k] from 0 into 2 right now: either reads as "not a copy")
            const FmlEdge ek = s[k];
            if (ek.len == ej.len) continue;
            const FmlStr sk = strs[ek.v];
            const int ak = su.len - ek.len, end_k = ak + sk.len, end_j = aj + sj.len;
            const int end = end_k < end_j ? end_k : end_j;
            drop = fml_same(text + sk.off + (su.len - ak), text + sj.off + (su.len - aj), 0, end - su.len);
        }
        if (drop) fl[j] = 2;
    }
}
static __global__ void __launch_bounds__(256) k_asm_huge_emit(const int *list, const unsigned int *n_list, const unsigned long long *eoff, const unsigned int *cur, const FmlEdge *sorted,
                                                              const unsigned char *flags, unsigned int *n_irr, unsigned long long *irr_off, FmlEdge *out, unsigned long long *out_n)
{
    const long long b = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= (long long)*n_list) return;
    const long long u = list[b];
    const int d = (int)cur[u];
    const FmlEdge *s = sorted + eoff[u];
    const unsigned char *fl = flags + eoff[u];
    int keep = 0;
    for (int j = lane; j < d; j += 64) keep += fl[j] == 0;
    for (int o = 32; o > 0; o >>= 1) keep += __shfl_xor(keep, o);
    unsigned long long at = 0;
    if (lane == 0) { at = atomicAdd(out_n, (unsigned long long)keep); n_irr[u] = (unsigned int)keep; irr_off[u] = at; }
    at = __shfl(at, 0);
    int done = 0;
    for (int j0 = 0; j0 < d; j0 += 64) {
        const int j = j0 + lane;
        const bool k = j < d && fl[j] == 0;
        const unsigned long long m = __ballot(k);
        if (k) out[at + done + __popcll(m & ((1ULL << lane) - 1))] = s[j];
        done += __popcll(m);
    }
}
