// The product's host-side graph stage (seqlib_amd/csrc/fml_graph.h: unitig chaining + mag_g_clean + fml_mag2utg) run WITHOUT a GPU on an overlap graph
// the CPU checker dumped (oracle/orc_fml_asm.c: orc_fml_set_overlap_dump), printed one unitig per line for tests/test_fml_graph.py to compare with
// the checker's own unitigs.  Test infrastructure: nothing in the product includes this file.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../seqlib_amd/csrc/fml_graph.h"

static void rd(void *p, size_t n, FILE *fp) { if (fread(p, 1, n, fp) != n) { fprintf(stderr, "short dump\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: fml_graph_test <dump>\n"); return 2; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 2; }
    int32_t hdr[2];
    rd(hdr, 8, fp);
    const int n = hdr[0];
    slx_magopt mo;
    rd(&mo, sizeof(mo), fp);
    std::vector<int> len((size_t)n + 1), rep((size_t)n + 1);
    std::vector<std::vector<unsigned char>> text((size_t)n + 1);
    std::vector<const unsigned char *> txt((size_t)n + 1);
    for (int t = 0; t < n; ++t) {
        int32_t l; rd(&l, 4, fp);
        len[(size_t)t] = l; text[(size_t)t].resize((size_t)l + 1); rd(text[(size_t)t].data(), (size_t)l, fp);
        txt[(size_t)t] = text[(size_t)t].data();
    }
    for (int t = 0; t < n; ++t) { int32_t r; rd(&r, 4, fp); rep[(size_t)t] = r; }
    std::vector<unsigned char> cont((size_t)n + 1);
    rd(cont.data(), (size_t)n, fp);
    std::vector<unsigned int> nirr((size_t)n + 1);
    std::vector<unsigned long long> off((size_t)n + 1);
    std::vector<int> ev, el;
    for (int t = 0; t < n; ++t) {
        int32_t k; rd(&k, 4, fp);
        nirr[(size_t)t] = (unsigned)k; off[(size_t)t] = ev.size();
        for (int j = 0; j < k; ++j) { int32_t e[2]; rd(e, 8, fp); ev.push_back(e[0]); el.push_back(e[1]); }
    }
    fclose(fp);
    ev.push_back(0); el.push_back(0);
    fmlg::Overlaps O;
    O.n_str = n; O.len = len.data(); O.text = txt.data(); O.rep = rep.data(); O.contained = cont.data();
    O.n_irr = nirr.data(); O.irr_off = off.data(); O.edge_v = ev.data(); O.edge_len = el.data(); O.min_match = hdr[1];
    fmlg::Graph g;
    g.build(O);
    g.clean_graph(mo);
    int n_utg = 0;
    slx_fml_utg *u = g.to_utgs(&n_utg);
    for (int i = 0; i < n_utg; ++i) {
        printf("UTG\t%d\t%d\t%s\t%s\t%d\t%d", u[i].len, u[i].nsr, u[i].seq, u[i].cov, u[i].n_ovlp[0], u[i].n_ovlp[1]);
        for (int j = 0; j < u[i].n_ovlp[0] + u[i].n_ovlp[1]; ++j) printf("\t%u:%u:%u:%u", (unsigned)u[i].ovlp[j].len, (unsigned)u[i].ovlp[j].from, (unsigned)u[i].ovlp[j].id, (unsigned)u[i].ovlp[j].to);
        printf("\n");
    }
    return 0;
}
