// fml_graph.h -- host side of fml_assemble: from the irreducible overlaps the GPU found (dev_fml_asm.h) to unitigs, and
// fermi-lite's graph cleaning (mag.c: mag_g_clean and the passes it runs) on the unitig graph, to fml_utg_t records.
//
// Reference behaviour: fml_fmi2mag's chaining of reads into unitigs, fml_mag_clean, fml_mag2utg as reached from
// /root/reference/src/FermiAssembler.cpp:26-44,140-151; defined (fermi-lite is not in the reference tree) in DESIGN.md section 8.
// The graph of a window has 10^2..10^4 vertices after chaining: control-plane work, done here on the host, one window at a time;
// the data-parallel work (10^7 positions per window) is on the GPU.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "seqlib_amd_fml.h"

namespace fmlg {

struct Nei { uint64_t x, y; };          // x: the neighbouring end (a string index), y: the overlap
static const uint64_t DEL = (uint64_t)-2;
inline void mark_del(Nei &e) { e.x = DEL; e.y = 0; }
inline bool is_del(const Nei &e) { return e.x == DEL || e.y == 0; }

struct Vtx {
    int len = -1, nsr = 0;
    uint64_t k[2] = {0, 0};
    std::vector<Nei> nei[2];
    std::string seq, cov;          // seq in codes 1..5
};

// what the device hands over for one window (string indices local to the window)
struct Overlaps {
    int n_str = 0;
    const int *len = nullptr;                  // per string
    const unsigned char *const *text = nullptr;          // per string: codes
    const int *rep = nullptr;
    const unsigned char *contained = nullptr;
    const unsigned int *n_irr = nullptr;
    const unsigned long long *irr_off = nullptr;
    const int *edge_v = nullptr, *edge_len = nullptr;          // irreducible edges, targets local to the window
    int min_match = 0;
};

struct Graph {
    std::vector<Vtx> v;
    std::vector<int64_t> idd;          // end name -> vertex << 1 | side, -1 = none
    int min_ovlp = 0;

    int64_t tid2idd(uint64_t tid) const { assert((int64_t)tid >= 0 && tid < idd.size() && idd[tid] >= 0); return idd[tid]; }

    void destroy(Vtx &p) { p = Vtx(); }

    void flip(int i)
    {
        Vtx &p = v[(size_t)i];
        std::reverse(p.seq.begin(), p.seq.end());
        for (char &c : p.seq) c = c < 5 ? (char)(5 - c) : (char)5;
        std::reverse(p.cov.begin(), p.cov.end());
        std::swap(p.k[0], p.k[1]);
        std::swap(p.nei[0], p.nei[1]);
        idd[p.k[0]] = (int64_t)i << 1 | 0;
        idd[p.k[1]] = (int64_t)i << 1 | 1;
    }

    void eh_add(uint64_t u, uint64_t w, int ovlp)
    {
        if ((int64_t)u < 0) return;
        const int64_t x = tid2idd(u);
        std::vector<Nei> &r = v[(size_t)(x >> 1)].nei[x & 1];
        for (const Nei &e : r) if (e.x == w) return;
        r.push_back(Nei{w, (uint64_t)ovlp});
    }

    void eh_markdel(uint64_t u, uint64_t w)
    {
        if ((int64_t)u < 0) return;
        const int64_t x = tid2idd(u);
        for (Nei &e : v[(size_t)(x >> 1)].nei[x & 1]) if (e.x == w) mark_del(e);
    }

    void v_del(int i)
    {
        Vtx &p = v[(size_t)i];
        if (p.len < 0) return;
        for (int s = 0; s < 2; ++s)
            for (const Nei &e : p.nei[s])
                if (!is_del(e) && e.x != p.k[0] && e.x != p.k[1]) eh_markdel(e.x, p.k[s]);
        idd[p.k[0]] = idd[p.k[1]] = -1;
        destroy(p);
    }

    void v_transdel(int i, int min_ov)
    {
        Vtx &p = v[(size_t)i];
        if (!p.nei[0].empty() && !p.nei[1].empty()) {
            // (eh_add may grow other vertices' lists, never p's: p's own ends are skipped)
            for (size_t a = 0; a < p.nei[0].size(); ++a) {
                const Nei ea = p.nei[0][a];
                if (is_del(ea) || ea.x == p.k[0] || ea.x == p.k[1]) continue;
                for (size_t b = 0; b < p.nei[1].size(); ++b) {
                    const Nei eb = p.nei[1][b];
                    if (is_del(eb) || eb.x == p.k[0] || eb.x == p.k[1]) continue;
                    const int ov = (int)(ea.y + eb.y) - p.len;
                    if (ov >= min_ov) { eh_add(ea.x, eb.x, ov); eh_add(eb.x, ea.x, ov); }
                }
            }
        }
        v_del(i);
    }

    static void clean(std::vector<Nei> &r)
    {
        size_t j = 0;
        for (size_t i = 0; i < r.size(); ++i) if (!is_del(r[i])) r[j++] = r[i];
        r.resize(j);
    }

    static void rmdup(std::vector<Nei> &r, int min_ov)
    {
        if (r.size() > 1) std::sort(r.begin(), r.end(), [](const Nei &a, const Nei &b) { return a.x != b.x ? a.x < b.x : a.y > b.y; });
        size_t l = 0;
        int cnt = 0;
        for (; l < r.size(); ++l) {
            if (is_del(r[l]) || (int)r[l].y < min_ov) { mark_del(r[l]); ++cnt; }
            else break;
        }
        if (l == r.size()) { r.clear(); return; }
        uint64_t x = r[l].x;
        for (++l; l < r.size(); ++l) {
            if (is_del(r[l]) || (int)r[l].y < min_ov) { mark_del(r[l]); ++cnt; }
            else if (x == r[l].x) { mark_del(r[l]); ++cnt; }
            else x = r[l].x;
        }
        if (cnt) clean(r);
    }

    int merge_try(int i, int min_merge_len)
    {
        Vtx &p = v[(size_t)i];
        if (p.nei[1].size() != 1) return -1;
        if ((int64_t)p.nei[1][0].x < 0) return -2;
        if ((int)p.nei[1][0].y < min_merge_len) return -5;
        int64_t iq = tid2idd(p.nei[1][0].x);
        const int qi = (int)(iq >> 1);
        if (qi == i) return -3;
        if (v[(size_t)qi].nei[iq & 1].size() != 1) return -4;
        if (iq & 1) flip(qi);
        Vtx &q = v[(size_t)qi];
        idd[p.k[1]] = -1; idd[q.k[0]] = -1;
        assert(p.k[1] == q.nei[0][0].x && q.k[0] == p.nei[1][0].x && p.nei[1][0].y == q.nei[0][0].y);
        const int ov = (int)p.nei[1][0].y;
        assert(p.len >= ov && q.len >= ov);
        p.nsr += q.nsr;
        const int new_l = p.len + q.len - ov;
        p.seq.resize((size_t)new_l); p.cov.resize((size_t)new_l);
        for (int a = p.len - ov, b = 0; b < q.len; ++a, ++b) {
            p.seq[(size_t)a] = q.seq[(size_t)b];
            if (a < p.len) {
                const int c = (int)p.cov[(size_t)a] + (q.cov[(size_t)b] - 33);
                p.cov[(size_t)a] = (char)(c > 126 ? 126 : c);
            } else p.cov[(size_t)a] = q.cov[(size_t)b];
        }
        p.len = new_l;
        p.nei[1] = std::move(q.nei[1]); p.k[1] = q.k[1];
        idd[p.k[1]] = (int64_t)i << 1 | 1;
        destroy(q);
        return 0;
    }

    void merge(bool dedup, int min_merge_len)
    {
        for (Vtx &p : v) {
            if (p.len < 0) continue;
            if (dedup) { rmdup(p.nei[0], min_ovlp); rmdup(p.nei[1], min_ovlp); }
            else { clean(p.nei[0]); clean(p.nei[1]); }
        }
        for (int i = 0; i < (int)v.size(); ++i) {
            if (v[(size_t)i].len < 0) continue;
            while (merge_try(i, min_merge_len) == 0) {}
            flip(i);
            while (merge_try(i, min_merge_len) == 0) {}
        }
    }

    std::vector<int> by_support(const std::vector<int> &ids) const          // (nsr, len, position) ascending
    {
        std::vector<int> a(ids);
        std::sort(a.begin(), a.end(), [&](int x, int y) {
            const Vtx &p = v[(size_t)x], &q = v[(size_t)y];
            if (p.nsr != q.nsr) return p.nsr < q.nsr;
            if (p.len != q.len) return p.len < q.len;
            return x < y;
        });
        return a;
    }

    bool is_tip(const Vtx &p, int min_len, int min_nsr) const { return (p.nei[0].empty() || p.nei[1].empty()) && p.len < min_len && p.nsr < min_nsr; }

    void rm_vext(int min_len, int min_nsr)
    {
        std::vector<int> a;
        for (int i = 0; i < (int)v.size(); ++i) {
            const Vtx &p = v[(size_t)i];
            if (p.len < 0 || (!p.nei[0].empty() && !p.nei[1].empty())) continue;
            if (p.len >= min_len || p.nsr >= min_nsr) continue;
            a.push_back(i);
        }
        for (int i : by_support(a)) v_del(i);
    }

    void rm_vint(int min_len, int min_nsr, int min_ov)
    {
        std::vector<int> a;
        for (int i = 0; i < (int)v.size(); ++i) {
            const Vtx &p = v[(size_t)i];
            if (p.len >= 0 && p.len < min_len && p.nsr < min_nsr) a.push_back(i);
        }
        for (int i : by_support(a)) v_transdel(i, min_ov);
    }

    void rm_edge(int min_ov, double min_ratio, int min_len, int min_nsr)
    {
        std::vector<int> a;
        for (int i = 0; i < (int)v.size(); ++i) {
            const Vtx &p = v[(size_t)i];
            if (p.len < 0 || is_tip(p, min_len, min_nsr)) continue;
            a.push_back(i);
        }
        a = by_support(a);
        for (size_t n = a.size(); n > 0; --n) {
            Vtx &p = v[(size_t)a[n - 1]];
            for (int s = 0; s < 2; ++s) {
                std::vector<Nei> &r = p.nei[s];
                int max_ov = min_ov, max_k = -1;
                if (r.empty()) continue;
                for (int k = 0; k < (int)r.size(); ++k)
                    if (max_ov < (int)r[(size_t)k].y) max_ov = (int)r[(size_t)k].y, max_k = k;
                if (max_k >= 0) {
                    const int64_t x = tid2idd(r[(size_t)max_k].x);
                    const Vtx &q = v[(size_t)(x >> 1)];
                    if (q.len >= 0 && is_tip(q, min_len, min_nsr)) max_ov = min_ov;
                }
                for (Nei &e : r) {
                    if (is_del(e)) continue;
                    if ((int)e.y < min_ov || (double)e.y / max_ov < min_ratio) { eh_markdel(e.x, p.k[s]); mark_del(e); }
                }
            }
        }
    }

    // ksw_align(..., 4, mat(5 / -4), gapo, gape, xtra = 0).score: the local-alignment optimum, a gap of k bases costing gapo + k * gape, cells capped at
    // 32767 as the 16-bit striped kernel's saturating adds cap them.  Codes 0..3 = ACGT, anything else never matches.
    // Computed column by column of the query (the checker walks rows of the target): same cells, same values.
    static int sw_local(const std::string &qry, const std::string &tgt, int match, int mismatch, int gapo, int gape)
    {
        const size_t n = tgt.size();
        std::vector<int> H(n + 1, 0), F(n + 1, 0);          // H of the previous query column; F = best score ending in a gap that consumes query bases
        int best = 0;
        for (size_t j = 0; j < qry.size(); ++j) {
            int diag = 0, e = 0;          // e = best score ending in a gap that consumes target bases, within this column
            const char cq = qry[j];
            for (size_t i = 1; i <= n; ++i) {
                const int up = H[i];
                int h = diag + ((tgt[i - 1] == cq && cq < 4) ? match : mismatch);
                h = std::max(std::max(h, F[i]), std::max(e, 0));
                h = std::min(h, 32767);
                diag = up;
                H[i] = h;
                best = std::max(best, h);
                const int open = h - gapo - gape;
                F[i] = std::max(std::max(F[i] - gape, open), 0);
                e = std::max(std::max(e - gape, open), 0);
            }
        }
        return best;
    }

    // ksw_extend (fermi-lite's ksw.c: one gap cost, no z-drop, no end bonus), as called by mag_v_pop_open: 5 x 5 matrix, 5 / -4, code 4 scores 0
    static int extend(const std::string &qry, const std::string &tgt, int gapo, int gape, int w, int h0, int *qle)
    {
        const int qlen = (int)qry.size(), tlen = (int)tgt.size(), gapoe = gapo + gape;
        struct Cell { int h = 0, e = 0; };
        std::vector<Cell> eh((size_t)qlen + 2);
        auto sc = [](char a, char b) { return (a > 3 || b > 3) ? 0 : (a == b ? 5 : -4); };
        eh[0].h = h0;
        if (qlen >= 1) eh[1].h = h0 > gapoe ? h0 - gapoe : 0;
        for (int j = 2; j <= qlen && eh[(size_t)j - 1].h > gape; ++j) eh[(size_t)j].h = eh[(size_t)j - 1].h - gape;
        int max_gap = (int)((double)(qlen * 5 - gapo) / gape + 1.);
        max_gap = std::max(max_gap, 1);
        w = std::min(w, max_gap);
        int best = h0, best_j = -1, beg = 0, end = qlen;
        for (int i = 0; i < tlen; ++i) {
            int f = 0, row_max = 0, row_j = -1;
            beg = std::max(beg, i - w);
            end = std::min(std::min(end, i + w + 1), qlen);
            int h1 = beg == 0 ? std::max(h0 - (gapo + gape * (i + 1)), 0) : 0;
            int j = beg;
            for (; j < end; ++j) {
                Cell &c = eh[(size_t)j];
                int M = c.h, e = c.e;
                c.h = h1;
                M = M ? M + sc(tgt[(size_t)i], qry[(size_t)j]) : 0;
                const int h = std::max(std::max(M, e), f);
                h1 = h;
                if (!(row_max > h)) row_j = j;
                row_max = std::max(row_max, h);
                const int t = std::max(M - gapoe, 0);
                c.e = std::max(e - gape, t);
                f = std::max(f - gape, t);
            }
            eh[(size_t)end].h = h1; eh[(size_t)end].e = 0;
            if (row_max == 0) break;
            if (row_max > best) best = row_max, best_j = row_j;
            for (j = beg; j < end && eh[(size_t)j].h == 0 && eh[(size_t)j].e == 0; ++j) {}
            beg = j;
            for (j = end; j >= beg && eh[(size_t)j].h == 0 && eh[(size_t)j].e == 0; --j) {}
            end = std::min(j + 2, qlen);
        }
        *qle = best_j + 1;
        return best;
    }

    // the l bases of q that follow its overlap when it is entered through `side`, as codes 0..4
    static std::string branch(const Vtx &q, int side, int ovlp, int l)
    {
        std::string s((size_t)std::max(l, 0), 0);
        for (int i = 0; i < l; ++i) {
            const int at = side == 0 ? ovlp + i : q.len - 1 - ovlp - i;
            const char c = q.seq[(size_t)at];
            s[(size_t)i] = (char)((side == 0 ? c : (c < 5 ? (char)(5 - c) : (char)5)) - 1);
        }
        return s;
    }

    // bubble.c: mag_vh_pop_simple
    void pop_simple_at(int64_t x0, float max_cov, float max_frac, bool aggressive)
    {
        const int pi = (int)(x0 >> 1), dir = (int)(x0 & 1);
        const Vtx &p = v[(size_t)pi];
        const float max_n_diff = aggressive ? 2.01 * 2. : 2.01, max_r_diff = aggressive ? 0.1 * 2. : 0.1;
        if (p.len < 0 || p.nei[dir].size() != 2) return;
        int qi[2], side[2], l[2];
        float avg[2];
        for (int j = 0; j < 2; ++j) {
            const Nei &e = p.nei[dir][(size_t)j];
            if ((int64_t)e.x < 0) return;
            const int64_t x = tid2idd(e.x);
            side[j] = (int)(x & 1); qi[j] = (int)(x >> 1);
            const Vtx &q = v[(size_t)qi[j]];
            if (q.nei[0].size() != 1 || q.nei[1].size() != 1) return;
            l[j] = q.len - (int)(q.nei[0][0].y + q.nei[1][0].y);
        }
        if (v[(size_t)qi[0]].nei[side[0] ^ 1][0].x != v[(size_t)qi[1]].nei[side[1] ^ 1][0].x) return;
        std::string seq[2];
        for (int j = 0; j < 2; ++j) {
            const Vtx &q = v[(size_t)qi[j]];
            int beg = (int)q.nei[0][0].y, end = q.len - (int)q.nei[1][0].y;
            if (l[j] > 0) seq[j] = branch(q, side[j], (int)q.nei[side[j]][0].y, l[j]);
            else if (beg > end) std::swap(beg, end);
            if (beg < end) {          // l > 0: exactly the stretch between the overlaps; the sum of small integers is exact in float whatever its order
                int s = 0;
                for (int i = beg; i < end; ++i) s += q.cov[(size_t)i] - 33;
                avg[j] = (float)s / (float)(end - beg);
            } else avg[j] = (float)(q.cov[(size_t)beg] - 33);
        }
        float n_diff, r_diff;
        if (l[0] > 0 && l[1] > 0) {
            const int score = sw_local(seq[0], seq[1], 5, -4, 5, 2);
            n_diff = (float)((std::min(l[0], l[1]) * 5. - score) / (5. + 4.));
            r_diff = (float)(n_diff / ((l[0] + l[1]) / 2.));
        } else n_diff = (float)(std::abs(l[0] - l[1]) * 0.2), r_diff = 1.f;
        if (n_diff < max_n_diff || r_diff < max_r_diff) {
            const int j = avg[0] < avg[1] ? 0 : 1;
            if (aggressive || (avg[j] / (avg[j ^ 1] + avg[j]) < max_frac && avg[j] < max_cov)) v_del(qi[j]);
        }
    }

    void pop_simple(float max_cov, float max_frac, int min_merge_len, bool aggressive)
    {
        for (int64_t i = 0; i < (int64_t)v.size(); ++i) {
            pop_simple_at(i << 1 | 0, max_cov, max_frac, aggressive);
            pop_simple_at(i << 1 | 1, max_cov, max_frac, aggressive);
        }
        merge(false, min_merge_len);
    }

    // mag.c: mag_v_pop_open (the tip is extended against every sibling with ksw_extend)
    void pop_open_at(int pi, int min_elen)
    {
        const Vtx &p = v[(size_t)pi];
        if (p.len < 0 || p.len >= min_elen) return;
        if (p.nei[0].size() + p.nei[1].size() != 1) return;
        const int dir = p.nei[0].empty() ? 1 : 0;
        const Nei e0 = p.nei[dir][0];
        if ((int64_t)e0.x < 0) return;
        const int64_t x = tid2idd(e0.x);
        const int qi = (int)(x >> 1);
        if (qi == pi) return;
        const std::vector<Nei> &r = v[(size_t)qi].nei[x & 1];
        const int lq = p.len - (int)e0.y;
        if (lq <= 0) return;
        const std::string qs = branch(p, dir, (int)e0.y, lq);
        const int h0 = (int)e0.y * 5;
        bool kill = false;
        for (size_t i = 0; i < r.size() && !kill; ++i) {
            if ((int64_t)r[i].x < 0) continue;
            const int64_t y = tid2idd(r[i].x);
            const int ti = (int)(y >> 1);
            const Vtx &t = v[(size_t)ti];
            if (ti == pi || ti == qi || t.len < 0) continue;
            const int lt = t.len - (int)r[i].y;
            if (lt <= 0) continue;
            int qle = 0;
            const int sc = extend(qs, branch(t, (int)(y & 1), (int)r[i].y, lt), 5, 2, 50, h0, &qle);
            if (qle == lq && sc - h0 >= 5 * lq - 9 * std::max(lq / 10, 1)) kill = true;
        }
        if (kill) v_del(pi);
    }

    void pop_open(int min_elen) { for (int i = 0; i < (int)v.size(); ++i) pop_open_at(i, min_elen); }

    // Closed bubbles with more than two paths (fermi's bubble.c: mag_g_simplify_bubble; what FermiAssembler::SetSimplifyBubble turns on by clearing
    // MAG_F_NO_SIMPL, /root/reference/SeqLib/FermiAssembler.h:88-90).  From an end with two or more neighbours the graph is walked in topological order -- an end
    // is expanded once every edge into it has been seen -- carrying per end reached the two best-supported paths from the start (n: reads on the path, d: its
    // length with the overlaps taken off, and where it came from).  More than max_vtx vertices, a path beyond max_dist, a dead end or a way back to the start:
    // no bubble.  When the front shrinks to one end with nothing pending, that end closes the bubble: the vertices on its two best paths stay, the others the
    // walk touched go.  The CPU checker's restatement of mag_vh_simplify_bubble (test infrastructure) is the definition; this is the same walk on this graph's types.
    struct Tri { int64_t id; int cnt[2]; int n[2][2], d[2][2]; int64_t bx[2][2]; int br[2][2]; };
    struct BubbleAux { std::vector<Tri> a; std::vector<int64_t> stack; std::vector<int> slot; std::vector<unsigned char> keep; };
    static Tri &tri_get(BubbleAux &A, int64_t x)
    {
        int &sl = A.slot[(size_t)(x >> 1)];
        if (sl >= 0) return A.a[(size_t)sl];
        Tri t;
        t.id = x; t.cnt[0] = t.cnt[1] = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { t.n[i][j] = t.d[i][j] = -0x40000000; t.bx[i][j] = -1; t.br[i][j] = 0; }
        sl = (int)A.a.size();
        A.a.push_back(t);
        return A.a.back();
    }
    void simplify_bubble_at(int64_t start, int max_vtx, int max_dist, BubbleAux &A)
    {
        {
            const Vtx &p0 = v[(size_t)(start >> 1)];
            if (p0.len < 0 || p0.nei[start & 1].size() < 2) return;
        }
        A.a.clear(); A.stack.clear();
        int n_pending = 0;
        bool failed = false;
        {
            const Vtx &p0 = v[(size_t)(start >> 1)];
            Tri &t0 = tri_get(A, start);
            t0.d[(start & 1) ^ 1][0] = -p0.len;
            t0.n[(start & 1) ^ 1][0] = -p0.nsr;
        }
        A.stack.push_back(start ^ 1);
        while (!A.stack.empty()) {
            if (A.stack.size() == 1 && A.stack[0] != (start ^ 1) && n_pending == 0) break;          // the other end of the bubble
            const int64_t x = A.stack.back();
            A.stack.pop_back();
            const Vtx &p = v[(size_t)(x >> 1)];
            const std::vector<Nei> &r = p.nei[(x & 1) ^ 1];
            {
                const Tri &tp = tri_get(A, x);
                if ((int)A.a.size() > max_vtx || tp.d[x & 1][0] > max_dist || tp.d[x & 1][1] > max_dist || r.empty()) { failed = true; break; }
            }
            for (size_t i = 0; i < r.size(); ++i) {
                if ((int64_t)r[i].x < 0 || is_del(r[i])) continue;
                const int64_t y = tid2idd(r[i].x);
                if (y == (start ^ 1)) { A.stack.clear(); failed = true; break; }          // a loop through the start
                if (A.slot[(size_t)(y >> 1)] < 0) { ++n_pending; clean(v[(size_t)(y >> 1)].nei[y & 1]); }
                (void)tri_get(A, y);                              // (may move the table)
                const Tri &tp = tri_get(A, x);                    // (a reference, as in the checker: on a self-loop tp and tq are one record)
                Tri &tq = tri_get(A, y);
                int nsr = tp.n[x & 1][0] + p.nsr, dist = tp.d[x & 1][0] + p.len - (int)r[i].y, which = 0;
                if (nsr > tq.n[y & 1][0]) {
                    tq.n[y & 1][1] = tq.n[y & 1][0]; tq.d[y & 1][1] = tq.d[y & 1][0]; tq.bx[y & 1][1] = tq.bx[y & 1][0]; tq.br[y & 1][1] = tq.br[y & 1][0];
                    tq.n[y & 1][0] = nsr; tq.d[y & 1][0] = dist; tq.bx[y & 1][0] = x ^ 1; tq.br[y & 1][0] = 0;
                    nsr = tp.n[x & 1][1] + p.nsr;
                    dist = tp.d[x & 1][1] + p.len - (int)r[i].y;
                    which = 1;
                }
                if (nsr > tq.n[y & 1][1]) { tq.n[y & 1][1] = nsr; tq.d[y & 1][1] = dist; tq.bx[y & 1][1] = x ^ 1; tq.br[y & 1][1] = which; }
                if (++tq.cnt[y & 1] == (int)v[(size_t)(y >> 1)].nei[y & 1].size()) { A.stack.push_back(y); --n_pending; }
            }
            if (failed) break;
        }
        if (!failed && n_pending == 0 && A.stack.size() == 1 && A.stack[0] != (start ^ 1)) {
            const int64_t x = A.stack[0];
            A.keep.assign(A.a.size(), 0);
            A.keep[(size_t)A.slot[(size_t)(start >> 1)]] = 1; A.keep[(size_t)A.slot[(size_t)(x >> 1)]] = 1;
            for (int rank = 0; rank < 2; ++rank) {
                int64_t at = x;
                int rk = rank;
                for (;;) {
                    const Tri &t = A.a[(size_t)A.slot[(size_t)(at >> 1)]];
                    const int64_t px = t.bx[at & 1][rk];
                    const int prk = t.br[at & 1][rk];
                    if (px < 0 || px == start) break;
                    A.keep[(size_t)A.slot[(size_t)(px >> 1)]] = 1;
                    at = px ^ 1; rk = prk;
                }
            }
            for (size_t i = 0; i < A.a.size(); ++i)
                if (!A.keep[i]) v_del((int)(A.a[i].id >> 1));
        }
        for (const Tri &t : A.a) A.slot[(size_t)(t.id >> 1)] = -1;
    }
    void simplify_bubble(int max_vtx, int max_dist)
    {
        BubbleAux A;
        A.slot.assign(v.size() + 1, -1);
        for (int64_t i = 0; i < (int64_t)v.size(); ++i) {
            simplify_bubble_at(i << 1 | 0, max_vtx, max_dist, A);
            simplify_bubble_at(i << 1 | 1, max_vtx, max_dist, A);
        }
        merge(false, 0);
    }

    void clean_graph(const slx_magopt &o)          // mag.c: mag_g_clean
    {
        if (min_ovlp < o.min_ovlp) min_ovlp = o.min_ovlp;
        for (int j = 2; j <= o.min_ensr; ++j) rm_vext(o.min_elen, j);
        merge(false, o.min_merge_len);
        rm_edge(min_ovlp, o.min_dratio1, o.min_elen, o.min_ensr);
        merge(true, o.min_merge_len);
        for (int j = 2; j <= o.min_ensr; ++j) rm_vext(o.min_elen, j);
        merge(false, o.min_merge_len);
        if (o.flag & SLX_MAG_F_POPOPEN) { pop_open(o.min_elen); merge(false, o.min_merge_len); }
        if (!(o.flag & SLX_MAG_F_NO_SIMPL)) simplify_bubble(o.max_bvtx, o.max_bdist);
        pop_simple(o.max_bcov, o.max_bfrac, o.min_merge_len, (o.flag & SLX_MAG_F_AGGRESSIVE) != 0);
        rm_vint(o.min_elen, o.min_insr, min_ovlp);
        rm_edge(min_ovlp, o.min_dratio1, o.min_elen, o.min_ensr);
        merge(true, o.min_merge_len);
        rm_vext(o.min_elen, o.min_ensr);
        merge(false, o.min_merge_len);
        if (o.flag & SLX_MAG_F_POPOPEN) { pop_open(o.min_elen); merge(false, o.min_merge_len); }
        rm_vext(o.min_elen, o.min_ensr);
        merge(false, o.min_merge_len);
    }

    // chains of vertices over edges that are the only way out of their source and the only way into their target
    void build(const Overlaps &O)
    {
        const int n = O.n_str;
        min_ovlp = O.min_match;
        idd.assign((size_t)n + 1, -1);
        v.clear();
        auto vertex = [&](int t) { return O.rep[t] == t && !O.contained[t]; };
        auto deg = [&](int t) { return vertex(t) ? (int)O.n_irr[t] : 0; };
        auto target = [&](int t, int j) { return O.edge_v[O.irr_off[t] + (unsigned long long)j]; };
        auto ovl = [&](int t, int j) { return O.edge_len[O.irr_off[t] + (unsigned long long)j]; };
        std::vector<unsigned char> used((size_t)n + 1, 0);
        std::vector<int> right, left, all, dcov;
        for (int t = 0; t < n; ++t) {
            if (!vertex(t) || used[(size_t)t]) continue;
            used[(size_t)t] = used[(size_t)(t ^ 1)] = 1;
            right.clear(); left.clear();
            for (int cur = t; deg(cur) == 1;) {
                const int w = target(cur, 0);
                if (deg(w ^ 1) != 1 || used[(size_t)w]) break;
                used[(size_t)w] = used[(size_t)(w ^ 1)] = 1;
                right.push_back(w); cur = w;
            }
            for (int cur = t ^ 1; deg(cur) == 1;) {
                const int w = target(cur, 0);
                if (deg(w ^ 1) != 1 || used[(size_t)w]) break;
                used[(size_t)w] = used[(size_t)(w ^ 1)] = 1;
                left.push_back(w); cur = w;
            }
            all.clear();
            for (size_t i = left.size(); i > 0; --i) all.push_back(left[i - 1] ^ 1);
            all.push_back(t);
            for (int w : right) all.push_back(w);
            int tot = O.len[all[0]];
            for (size_t i = 1; i < all.size(); ++i) {
                assert(deg(all[i - 1]) == 1 && target(all[i - 1], 0) == all[i]);
                tot += O.len[all[i]] - ovl(all[i - 1], 0);
            }
            Vtx p;
            p.len = tot; p.nsr = (int)all.size();
            p.seq.assign((size_t)tot, 0); p.cov.assign((size_t)tot, 33);
            // the unitig's bases: a string agrees with the one before it over their overlap (the overlaps are exact), so only what lies beyond it is
            // new; its coverage: +1 where a string starts, -1 behind its end, summed once (33 + the number of strings over a base, capped at 126)
            dcov.assign((size_t)tot + 1, 0);
            int pos = 0;
            for (size_t i = 0; i < all.size(); ++i) {
                const int ov = i ? ovl(all[i - 1], 0) : 0;
                if (i) pos += O.len[all[i - 1]] - ov;
                const unsigned char *s = O.text[all[i]];
                const int l = O.len[all[i]];
                if (l > ov) memcpy(&p.seq[(size_t)(pos + ov)], s + ov, (size_t)(l - ov));
                ++dcov[(size_t)pos]; --dcov[(size_t)(pos + l)];
            }
            for (int x = 0, c = 0; x < tot; ++x) { c += dcov[(size_t)x]; p.cov[(size_t)x] = (char)(33 + c > 126 ? 126 : 33 + c); }
            const int first = all.front(), last = all.back();
            p.k[0] = (uint64_t)(first ^ 1); p.k[1] = (uint64_t)last;
            for (int j = 0; j < deg(first ^ 1); ++j) p.nei[0].push_back(Nei{(uint64_t)(target(first ^ 1, j) ^ 1), (uint64_t)ovl(first ^ 1, j)});
            for (int j = 0; j < deg(last); ++j) p.nei[1].push_back(Nei{(uint64_t)(target(last, j) ^ 1), (uint64_t)ovl(last, j)});
            idd[p.k[0]] = (int64_t)v.size() << 1 | 0;
            idd[p.k[1]] = (int64_t)v.size() << 1 | 1;
            v.push_back(std::move(p));
        }
        // an edge must be answered from the other side (mag_g_amend)
        for (Vtx &p : v)
            for (int s = 0; s < 2; ++s) {
                for (Nei &e : p.nei[s]) {
                    bool ok = false;
                    if (idd[e.x] >= 0) {
                        const int64_t y = idd[e.x];
                        for (const Nei &b : v[(size_t)(y >> 1)].nei[y & 1]) if (b.x == p.k[s]) ok = true;
                    }
                    if (!ok) mark_del(e);
                }
                clean(p.nei[s]);
            }
    }

    // fml_mag2utg; the records are malloc'ed for the C-ABI (slx_fml_utgs_free)
    slx_fml_utg *to_utgs(int *n_utg) const
    {
        std::vector<int64_t> newid(idd.size(), -1);
        int n = 0;
        for (const Vtx &p : v) {
            if (p.len < 0) continue;
            newid[p.k[0]] = (int64_t)n << 1 | 0; newid[p.k[1]] = (int64_t)n << 1 | 1;
            ++n;
        }
        *n_utg = n;
        slx_fml_utg *utg = (slx_fml_utg *)calloc((size_t)std::max(n, 1), sizeof(slx_fml_utg));
        int j = 0;
        for (const Vtx &p : v) {
            if (p.len < 0) continue;
            slx_fml_utg &q = utg[j++];
            q.len = p.len; q.nsr = p.nsr;
            q.seq = (char *)malloc((size_t)p.len + 1); q.cov = (char *)malloc((size_t)p.len + 1);
            for (int a = 0; a < p.len; ++a) { q.seq[a] = "$ACGTN"[(int)p.seq[(size_t)a]]; q.cov[a] = p.cov[(size_t)a]; }
            q.seq[p.len] = q.cov[p.len] = 0;
            for (int from = 0; from < 2; ++from) {
                q.n_ovlp[from] = 0;
                for (const Nei &e : p.nei[from]) if (!is_del(e) && newid[e.x] >= 0) ++q.n_ovlp[from];
            }
            q.ovlp = (slx_fml_ovlp *)calloc((size_t)(q.n_ovlp[0] + q.n_ovlp[1] + 1), sizeof(slx_fml_ovlp));
            int a = 0;
            for (int from = 0; from < 2; ++from)
                for (const Nei &e : p.nei[from])
                    if (!is_del(e) && newid[e.x] >= 0) {
                        slx_fml_ovlp &o = q.ovlp[a++];
                        o.len = (uint32_t)e.y; o.from = (uint32_t)from;
                        o.id = (uint32_t)(newid[e.x] >> 1); o.to = (uint32_t)(newid[e.x] & 1);
                    }
        }
        return utg;
    }
};

}          // namespace fmlg
