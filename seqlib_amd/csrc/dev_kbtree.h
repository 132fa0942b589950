// dev_kbtree.h -- bwa's ordered chain set, exactly: klib's kbtree as mem_chain instantiates it (KBTREE_INIT(chn, mem_chain_t,
// chain_cmp), kb_init(chn, 512) => t = 5, at most 9 keys per node; keys compared by chain pos, duplicates allowed).
// Two things about it are visible in the output: which chain kb_intervalp returns as `lower` when several chains share a position
// (the first copy the root-to-leaf search meets), and the in-order traversal that hands the chains to mem_chain_flt's unstable
// sort.  With at most 9 chains the tree is a single leaf, which a sorted array reproduces (dev_chain.h keeps that fast path); from
// the 10th chain on the shape matters as soon as two chains share a position, so the tree is built operation for operation:
// __kb_getp_aux, kb_intervalp, __kb_split, __kb_putp_aux, kb_putp, __kb_traverse.  Nodes live in the read's own region slots
// (unused until extension): 40 ints per node, at most (n_chains - 1) / 4 + 1 nodes.
// Reached from /root/reference/src/BWAAligner.cpp:104 (mem_align1 -> mem_chain); SURVEY.md A.3.
#pragma once
#include <stdint.h>

#define KB_T 5
#define KB_NODE_INTS 40

struct KbTree {
    int *mem;                 // node k at mem + k * KB_NODE_INTS
    int n_nodes, root;
    // node layout (ints): [0] n, [1] is_internal, [2..19] int64 pos[9], [20..28] handle[9], [29..38] child[10]
    __device__ __forceinline__ int *node(int k) const { return mem + (size_t)k * KB_NODE_INTS; }
    __device__ __forceinline__ static int64_t *npos(int *x) { return (int64_t *)(x + 2); }
    __device__ __forceinline__ static int *nkey(int *x) { return x + 20; }
    __device__ __forceinline__ static int *nptr(int *x) { return x + 29; }
    __device__ __forceinline__ int alloc(int is_internal) { int *x = node(n_nodes); x[0] = 0; x[1] = is_internal; return n_nodes++; }

    // __kb_getp_aux: index of the last key <= pos in node x (first of equal keys), r = sign of cmp(pos, that key) as klib leaves it
    __device__ static int getp_aux(int *x, int64_t pos, int *r)
    {
        const int n = x[0];
        const int64_t *p = npos(x);
        int begin = 0, end = n;
        if (n == 0) return -1;
        while (begin < end) { const int mid = (begin + end) >> 1; if (p[mid] < pos) begin = mid + 1; else end = mid; }
        if (begin == n) { *r = 1; return n - 1; }
        *r = (p[begin] < pos) - (pos < p[begin]);
        if (*r < 0) --begin;
        return begin;
    }
    // kb_intervalp's `lower`: chain handle or -1
    __device__ int lower(int64_t pos) const
    {
        int xk = root, low = -1;
        for (;;) {
            int *x = node(xk);
            int r = 0;
            const int i = getp_aux(x, pos, &r);
            if (i >= 0 && r == 0) return nkey(x)[i];
            if (i >= 0) low = nkey(x)[i];
            if (!x[1]) return low;
            xk = nptr(x)[i + 1];
        }
    }
    // __kb_split: x internal, y = child i of x, full
    __device__ void split(int xk, int i, int yk)
    {
        const int zk = alloc(node(yk)[1]);
        int *x = node(xk), *y = node(yk), *z = node(zk);
        z[0] = KB_T - 1;
        for (int k = 0; k < KB_T - 1; ++k) { npos(z)[k] = npos(y)[KB_T + k]; nkey(z)[k] = nkey(y)[KB_T + k]; }
        if (y[1]) for (int k = 0; k < KB_T; ++k) nptr(z)[k] = nptr(y)[KB_T + k];
        y[0] = KB_T - 1;
        for (int k = x[0]; k > i; --k) nptr(x)[k + 1] = nptr(x)[k];
        nptr(x)[i + 1] = zk;
        for (int k = x[0]; k > i; --k) { npos(x)[k] = npos(x)[k - 1]; nkey(x)[k] = nkey(x)[k - 1]; }
        npos(x)[i] = npos(y)[KB_T - 1]; nkey(x)[i] = nkey(y)[KB_T - 1];
        ++x[0];
    }
    // kb_putp
    __device__ void put(int64_t pos, int handle)
    {
        if (node(root)[0] == 2 * KB_T - 1) {
            const int s = alloc(1);
            nptr(node(s))[0] = root;
            split(s, 0, root);
            root = s;
        }
        int xk = root;
        for (;;) {                                   // __kb_putp_aux, iteratively
            int *x = node(xk);
            int r;
            if (!x[1]) {
                const int i = getp_aux(x, pos, &r);
                for (int k = x[0] - 1; k > i; --k) { npos(x)[k + 1] = npos(x)[k]; nkey(x)[k + 1] = nkey(x)[k]; }
                npos(x)[i + 1] = pos; nkey(x)[i + 1] = handle;
                ++x[0];
                return;
            }
            int i = getp_aux(x, pos, &r) + 1;
            if (node(nptr(x)[i])[0] == 2 * KB_T - 1) {
                split(xk, i, nptr(x)[i]);
                x = node(xk);
                if (pos > npos(x)[i]) ++i;
            }
            xk = nptr(x)[i];
        }
    }
    // a single leaf holding the chains of a sorted array (what the tree is while it has at most 9 keys)
    __device__ void from_array(int *scratch, const int *ord, const int64_t *c_pos, int n)
    {
        mem = scratch; n_nodes = 0;
        root = alloc(0);
        int *x = node(root);
        x[0] = n;
        for (int k = 0; k < n; ++k) { npos(x)[k] = c_pos[ord[k]]; nkey(x)[k] = ord[k]; }
    }
    // __kb_traverse: handles in order -> out[]; returns the count
    __device__ int traverse(int *out) const
    {
        int stack_node[12], stack_i[12], sp = 0, n_out = 0;
        stack_node[0] = root; stack_i[0] = 0;
        while (sp >= 0) {
            int *x = node(stack_node[sp]);
            const int i = stack_i[sp];
            if (!x[1]) {                             // leaf: all keys, then pop
                for (int k = 0; k < x[0]; ++k) out[n_out++] = nkey(x)[k];
                --sp;
                continue;
            }
            // internal: child i, then key i, ..., last child x[0]
            if (i > x[0]) { --sp; continue; }
            if (i > 0) out[n_out++] = nkey(x)[i - 1];  // key i-1 sits between child i-1 and child i
            stack_i[sp] = i + 1;
            ++sp;
            stack_node[sp] = nptr(x)[i]; stack_i[sp] = 0;
        }
        return n_out;
    }
};
