// dev_sort.h -- device emulations of the two host sorts whose tie behaviour is observable in the
// reference's output:
//   * klib's ks_introsort (bwa's ksort.h), used on chains by weight, regions by end / score / hash
//     (mem_chain_flt, mem_sort_dedup_patch, mem_mark_primary_se)
//   * libstdc++'s std::sort, used on the hits at /root/reference/src/BWAAligner.cpp:133
// Both are unstable, so the exact sequence of compares and swaps is reproduced.  They sort an
// array of int handles; `lt(a, b)` compares the objects the handles name.
#pragma once
#include <hip/hip_runtime.h>

template <typename LT>
__device__ void ks_insertsort_idx(int *s, int *t, LT lt)
{
    for (int *i = s + 1; i < t; ++i)
        for (int *j = i; j > s && lt(*j, *(j - 1)); --j) { int sw = *j; *j = *(j - 1); *(j - 1) = sw; }
}

template <typename LT>
__device__ void ks_combsort_idx(int n, int *a, LT lt)
{
    const double shrink = 1.2473309501039786540366528676643;
    int do_swap, gap = n;
    do {
        if (gap > 2) {
            gap = (int)(gap / shrink);
            if (gap == 9 || gap == 10) gap = 11;
        }
        do_swap = 0;
        for (int *i = a; i < a + n - gap; ++i) {
            int *j = i + gap;
            if (lt(*j, *i)) { int tmp = *i; *i = *j; *j = tmp; do_swap = 1; }
        }
    } while (do_swap || gap > 2);
    if (gap != 1) ks_insertsort_idx(a, a + n, lt);
}

template <typename LT>
__device__ void ks_introsort_idx(int n, int *a, LT lt)
{
    if (n < 1) return;
    if (n == 2) {
        if (lt(a[1], a[0])) { int sw = a[0]; a[0] = a[1]; a[1] = sw; }
        return;
    }
    int d;
    for (d = 2; (1u << d) < (unsigned)n; ++d);
    int st_l[40], st_r[40], st_d[40], top = 0;
    int s = 0, t = n - 1;
    d <<= 1;
    while (true) {
        if (s < t) {
            if (--d == 0) { ks_combsort_idx(t - s + 1, a + s, lt); t = s; continue; }
            int i = s, j = t, k = i + ((j - i) >> 1) + 1;
            if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
            else k = lt(a[j], a[i]) ? i : j;
            int rp = a[k];
            if (k != t) { int sw = a[k]; a[k] = a[t]; a[t] = sw; }
            for (;;) {
                do ++i; while (lt(a[i], rp));
                do --j; while (i <= j && lt(rp, a[j]));
                if (j <= i) break;
                int sw = a[i]; a[i] = a[j]; a[j] = sw;
            }
            { int sw = a[i]; a[i] = a[t]; a[t] = sw; }
            if (i - s > t - i) {
                if (i - s > 16) { st_l[top] = s; st_r[top] = i - 1; st_d[top] = d; ++top; }
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) { st_l[top] = i + 1; st_r[top] = t; st_d[top] = d; ++top; }
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (top == 0) { ks_insertsort_idx(a, a + n, lt); return; }
            --top; s = st_l[top]; t = st_r[top]; d = st_d[top];
        }
    }
}

// ---------------------------------------------------------------- libstdc++ std::sort
template <typename LT>
__device__ void std_unguarded_linear_insert(int *a, int last, LT lt)
{
    int val = a[last], next = last - 1;
    while (lt(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
}

template <typename LT>
__device__ void std_insertion_sort(int *a, int first, int last, LT lt)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (lt(a[i], a[first])) {
            int val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else std_unguarded_linear_insert(a, i, lt);
    }
}

template <typename LT>
__device__ void std_adjust_heap(int *a, int first, int hole, int len, int value, LT lt)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(a[first + child], a[first + child - 1])) --child;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;   // __push_heap
    while (hole > top && lt(a[first + parent], value)) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

template <typename LT>
__device__ void std_heapsort(int *a, int first, int last, LT lt)
{   // partial_sort(first, last, last) = __heap_select (make_heap only) + sort_heap
    int len = last - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            std_adjust_heap(a, first, parent, len, a[first + parent], lt);
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        int value = a[last];
        a[last] = a[first];
        std_adjust_heap(a, first, 0, last - first, value, lt);
    }
}

template <typename LT>
__device__ void std_sort_idx(int n, int *a, LT lt)
{
    if (n <= 0) return;
    // __introsort_loop with an explicit stack for the right-hand recursions
    int st_f[48], st_l[48], st_d[48], top = 0;
    int first = 0, last = n, depth = 0;
    for (int m = n; m > 1; m >>= 1) ++depth;
    depth *= 2;
    while (true) {
        while (last - first > 16) {
            if (depth == 0) { std_heapsort(a, first, last, lt); break; }
            --depth;
            // __unguarded_partition_pivot
            int mid = first + (last - first) / 2;
            int ra = first + 1, rb = mid, rc = last - 1, res = first, pick;
            if (lt(a[ra], a[rb])) { if (lt(a[rb], a[rc])) pick = rb; else if (lt(a[ra], a[rc])) pick = rc; else pick = ra; }
            else if (lt(a[ra], a[rc])) pick = ra; else if (lt(a[rb], a[rc])) pick = rc; else pick = rb;
            { int sw = a[res]; a[res] = a[pick]; a[pick] = sw; }
            int f = first + 1, l = last;
            while (true) {
                while (lt(a[f], a[first])) ++f;
                --l;
                while (lt(a[first], a[l])) --l;
                if (!(f < l)) break;
                int sw = a[f]; a[f] = a[l]; a[l] = sw;
                ++f;
            }
            int cut = f;
            st_f[top] = cut; st_l[top] = last; st_d[top] = depth; ++top;   // recurse on [cut, last) later
            last = cut;
        }
        // NOTE: libstdc++ recurses into the right part FIRST, then loops on the left; the partitions are
        // disjoint so the order of processing does not change the outcome.
        if (top == 0) break;
        --top; first = st_f[top]; last = st_l[top]; depth = st_d[top];
    }
    // __final_insertion_sort
    if (n > 16) {
        std_insertion_sort(a, 0, 16, lt);
        for (int i = 16; i < n; ++i) std_unguarded_linear_insert(a, i, lt);
    } else std_insertion_sort(a, 0, n, lt);
}
