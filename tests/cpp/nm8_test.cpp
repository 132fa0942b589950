// nm8_chunk (seqlib_amd/csrc/dev_nm8.h: the NM count of k_cig_fast) against the definition -- bns_get_seq's base at every coordinate -- for every
// start in a small packed text, every length 1..8, both strands, including the windows at both ends of the text.  Plain g++, no GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../seqlib_amd/csrc/dev_nm8.h"
#ifndef NM8_PAD
#define NM8_PAD 16
#endif

static int base_at(const std::vector<uint8_t> &pac, int64_t l_pac, int64_t p)
{
    if (p >= l_pac) { const int64_t f = (l_pac << 1) - 1 - p; return 3 - ((pac[f >> 2] >> ((~f & 3) << 1)) & 3); }
    return (pac[p >> 2] >> ((~p & 3) << 1)) & 3;
}

int main()
{
    unsigned long long bad = 0, n = 0;
    for (int64_t l_pac : {1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 101, 257}) {
        std::vector<uint8_t> pac((size_t)(l_pac + 3) / 4 + NM8_PAD, 0);
        srand48((long)l_pac);
        for (int64_t i = 0; i < l_pac; ++i) pac[(size_t)(i >> 2)] |= (uint8_t)((lrand48() & 3) << ((~i & 3) << 1));
        for (size_t i = (size_t)(l_pac + 3) / 4; i < pac.size(); ++i) pac[i] = (uint8_t)lrand48();          // the padding holds anything
        for (int64_t rb = 0; rb < 2 * l_pac; ++rb)
            for (int nv = 1; nv <= 8; ++nv) {
                const int64_t lim = rb < l_pac ? l_pac : 2 * l_pac;
                if (rb + nv > lim) continue;
                for (int rep = 0; rep < 6; ++rep) {
                    uint8_t q[16];
                    int want = 0;
                    for (int k = 0; k < 16; ++k) q[k] = (uint8_t)(lrand48() % 5);
                    for (int k = 0; k < nv; ++k) {
                        if (rep < 3 && (lrand48() & 3)) q[k] = (uint8_t)base_at(pac, l_pac, rb + k);          // mostly matching, as real reads are
                        want += q[k] != base_at(pac, l_pac, rb + k);
                    }
                    const int got = nm8_chunk(pac.data(), l_pac, q, rb, nv);
                    ++n;
                    if (got != want) { if (++bad < 10) std::printf("l_pac %lld rb %lld nv %d: got %d want %d\n", (long long)l_pac, (long long)rb, nv, got, want); }
                }
            }
    }
    std::printf("%llu cases, %llu bad\n", n, bad);
    return bad ? 1 : 0;
}
