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

// The join.  PASS 1: duplicates (rep), containment, overlaps counted per source string.  PASS 2: the overlaps between vertices written out.
template <int PASS>
static __global__ void __launch_bounds__(256) k_asm_join(const unsigned char *text, unsigned long long text_len, const FmlStr *strs, const unsigned long long *str_off /* n_str + 1 */,
                                                         long long n_str, int kk, int min_match, const unsigned long long *keys, const unsigned int *vals,
                                                         int *rep, unsigned char *contained, unsigned int *cnt, const unsigned long long *eoff, unsigned int *cur, FmlEdge *edges)
{
    const unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= text_len) return;
    // the string holding text position g
    long long lo = 0, hi = n_str;
    while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if (str_off[mid] <= g) lo = mid; else hi = mid; }
    const int u = (int)lo;
    const FmlStr su = strs[u];
    const int p = (int)(g - su.off), rest = su.len - p;
    if (rest < min_match) return;
    if (PASS == 2 && (p == 0 || rep[u] != u || contained[u])) return;
    const unsigned long long key = fml_seed_key(text + g, kk, su.win);
    long long a = 0, b = n_str;
    while (a < b) { const long long mid = (a + b) >> 1; if (keys[mid] < key) a = mid + 1; else b = mid; }
    for (; a < n_str && keys[a] == key; ++a) {
        const int v = (int)vals[a];
        if (v == u) continue;
        const FmlStr sv = strs[v];
        const int m = rest < sv.len ? rest : sv.len;
        const unsigned char *x = text + g, *y = text + sv.off;
        bool same = true;
        for (int i = kk; i < m; ++i)
            if (x[i] != y[i]) { same = false; break; }
        if (!same) continue;
        if (sv.len <= rest) {          // v lies inside u
            if (PASS == 1) {
                if (p == 0 && sv.len == su.len) { if (v < u) atomicMin(&rep[u], v); }
                else contained[v] = 1;
            }
        } else if (p > 0) {          // v runs past the end of u: an overlap of `rest` bases
            if (PASS == 1) atomicAdd(&cnt[u], 1u);
            else if (rep[v] == v && !contained[v]) {
                const unsigned int slot = atomicAdd(&cur[u], 1u);
                edges[eoff[u] + slot] = FmlEdge{v, rest};
            }
        }
    }
}

// Transitive reduction, one wave per source vertex: edges ordered by (overlap descending, target ascending), one edge per target (the longest),
// and u -> v_j dropped when a longer overlap u -> v_k exists whose string agrees with v_j's wherever both lie beyond the end of u.
// The irreducible edges of u go to out[] at an offset reserved with one atomic per wave; n_irr / irr_off say where.
static __global__ void __launch_bounds__(256) k_asm_reduce(const unsigned char *text, const FmlStr *strs, long long n_str, const unsigned long long *eoff, const unsigned int *cur,
                                                           FmlEdge *edges, FmlEdge *sorted, unsigned char *flags, unsigned int *n_irr, unsigned long long *irr_off,
                                                           FmlEdge *out, unsigned long long *out_n)
{
    const long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (u >= n_str) return;
    const int d = (int)cur[u];
    if (d == 0) { if (lane == 0) { n_irr[u] = 0; irr_off[u] = 0; } return; }
    const unsigned long long base = eoff[u];
    const FmlEdge *e = edges + base;
    FmlEdge *s = sorted + base;
    unsigned char *fl = flags + base;
    // rank by counting: (len desc, v asc); equal (v, len) pairs cannot occur (one hit per position and target)
    for (int i = lane; i < d; i += 64) {
        const FmlEdge me = e[i];
        int r = 0;
        for (int j = 0; j < d; ++j) {
            const FmlEdge o = e[j];
            r += (o.len > me.len) || (o.len == me.len && o.v < me.v);
        }
        s[r] = me;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one edge per target: of several overlap lengths with the same string the longest (the first in this order) stays
    for (int j = lane; j < d; j += 64) {
        const int vj = s[j].v;
        int f = 0;
        for (int k = 0; k < j; ++k)
            if (s[k].v == vj) { f = 1; break; }
        fl[j] = (unsigned char)f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const FmlStr su = strs[u];
    for (int j = lane; j < d; j += 64) {
        if (fl[j] == 1) continue;
        const FmlEdge ej = s[j];
        const FmlStr sj = strs[ej.v];
        const int aj = su.len - ej.len;
        bool drop = false;
        for (int k = 0; k < j && !drop; ++k) {
            if (fl[k] == 1) continue;          // (another lane may be turning fl[k] from 0 into 2 right now: either reads as "not a copy")
            const FmlEdge ek = s[k];
            if (ek.len == ej.len) continue;
            const FmlStr sk = strs[ek.v];
            const int ak = su.len - ek.len, end_k = ak + sk.len, end_j = aj + sj.len;
            const int end = end_k < end_j ? end_k : end_j;
            bool ok = true;
            for (int x = su.len; x < end; ++x)
                if (text[sk.off + (unsigned long long)(x - ak)] != text[sj.off + (unsigned long long)(x - aj)]) { ok = false; break; }
            drop = ok;
        }
        if (drop) fl[j] = 2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // count, reserve, compact in order
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
