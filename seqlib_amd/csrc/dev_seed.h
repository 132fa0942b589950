// dev_seed.h -- SMEM seeding kernel: bwa's mem_collect_intv (three passes: all SMEMs, re-seeding
// inside long rare SMEMs, LAST-like forward seeds) as reached from
// /root/reference/src/BWAAligner.cpp:104 -> mem_align1 -> mem_chain.  SURVEY.md Appendix A.3/A.4.
// One lane owns one read; the bidirectional work lists live in a lane-interleaved HBM scratch so
// that lanes walking their lists in step touch neighbouring 16-byte slots.
#pragma once
#include "dev_fm.h"
#include "dev_types.h"

template <typename I>
struct WorkLists {
    IntvE<I> *base;           // already offset by the lane's slot
    size_t stride;            // n_threads
    int cap;
    __device__ __forceinline__ IntvE<I> &at(int list, int e) { return base[((size_t)list * cap + e) * stride]; }
};


