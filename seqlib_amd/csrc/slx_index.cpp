// slx_index.cpp -- host side of the index: the bwa on-disk formats, the 2-bit packed reference and
// the annotation table.  Mirrors /root/reference/src/BWAIndex.cpp:28-33 (LoadIndex -> bwa_idx_load),
// :83-180 (ConstructIndex), :183-302 (pac building), :360-406 (WriteIndex).  File layouts follow
// SURVEY.md Appendix B, which was byte-checked against the reference's tests/data/tiny.fa.*.
// The suffix sort / BWT / Occ construction itself runs on the GPU (slx_index_gpu.hip).
#include "slx_internal.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>

static thread_local char g_err[1024] = "";

void slx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *slx_last_error(void) { return g_err; }
extern "C" const char *slx_version(void) { return "seqlib_amd 0.1 (gfx950)"; }

// ---------------------------------------------------------------- libc lrand48 stream
static const uint64_t LCG_A = 0x5DEECE66DULL, LCG_C = 0xBULL, LCG_M = (1ULL << 48) - 1;

extern "C" uint64_t slx_lrand48_advance(uint64_t state, uint64_t n)
{
    uint64_t a = LCG_A, c = LCG_C, ra = 1, rc = 0;
    while (n) {
        if (n & 1) { ra = (ra * a) & LCG_M; rc = (rc * a + c) & LCG_M; }
        c = ((a + 1) * c) & LCG_M;
        a = (a * a) & LCG_M;
        n >>= 1;
    }
    return (ra * (state & LCG_M) + rc) & LCG_M;
}

extern "C" uint64_t slx_lrand48_peek_libc(void)
{
    // seed48 returns a pointer to the previous 48-bit state; put it straight back
    unsigned short zero[3] = {0, 0, 0};
    unsigned short *old = seed48(zero);
    unsigned short keep[3] = {old[0], old[1], old[2]};
    seed48(keep);
    return (uint64_t)keep[0] | (uint64_t)keep[1] << 16 | (uint64_t)keep[2] << 32;
}

extern "C" void slx_lrand48_skip_libc(uint64_t n)
{
    uint64_t s = slx_lrand48_advance(slx_lrand48_peek_libc(), n);
    unsigned short st[3] = {(unsigned short)(s & 0xffff), (unsigned short)((s >> 16) & 0xffff), (unsigned short)((s >> 32) & 0xffff)};
    seed48(st);
}

// ---------------------------------------------------------------- options
extern "C" void slx_fill_scmat(int a, int b, int8_t mat[25])
{
    int k = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[k++] = (int8_t)(i == j ? a : -b);
        mat[k++] = -1;
    }
    for (int j = 0; j < 5; ++j) mat[k++] = -1;
}

extern "C" void slx_opt_init(slx_opt *o)
{
    memset(o, 0, sizeof *o);
    o->a = 1; o->b = 4;
    o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1;
    o->w = 100; o->T = 30; o->zdrop = 100;
    o->pen_unpaired = 17; o->pen_clip5 = o->pen_clip3 = 5;
    o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10; o->max_occ = 500;
    o->max_chain_gap = 10000;
    o->mask_level = 0.50f; o->drop_ratio = 0.50f; o->split_factor = 1.5f; o->mask_level_redun = 0.95f;
    o->min_chain_weight = 0; o->max_chain_extend = 1 << 30;
    o->mapQ_coef_len = 50; o->mapQ_coef_fac = 3;   // (int)log(50)
    o->flag = 0x200;                               // MEM_F_SOFTCLIP
    o->XA_drop_ratio = 0.80f; o->max_XA_hits = 5; o->max_XA_hits_alt = 200;
    slx_fill_scmat(o->a, o->b, o->mat);
}

// ---------------------------------------------------------------- accessors
extern "C" int slx_index_nseq(const slx_index *idx) { return idx ? (int)idx->anns.size() : 0; }
extern "C" const char *slx_index_name(const slx_index *idx, int i)
{
    if (!idx || i < 0 || i >= (int)idx->anns.size()) return nullptr;
    return idx->anns[i].name.c_str();
}
extern "C" int64_t slx_index_len(const slx_index *idx, int i)
{
    if (!idx || i < 0 || i >= (int)idx->anns.size()) return -1;
    return idx->anns[i].len;
}
extern "C" int64_t slx_index_l_pac(const slx_index *idx) { return idx ? idx->l_pac : 0; }
extern "C" int slx_index_n_holes(const slx_index *idx) { return idx ? (int)idx->ambs.size() : 0; }
extern "C" int slx_index_fetch(const slx_index *idx, int rid, int64_t beg, int64_t len, char *out)
{
    if (!idx || !out || rid < 0 || rid >= (int)idx->anns.size() || beg < 0 || len < 0 || beg + len > idx->anns[(size_t)rid].len) {
        slx_set_error("slx_index_fetch: bad argument");
        return SLX_EINVAL;
    }
    const int64_t p0 = idx->anns[(size_t)rid].offset + beg;
    for (int64_t i = 0; i < len; ++i) {
        const int64_t p = p0 + i;
        out[i] = "ACGT"[idx->pac[(size_t)(p >> 2)] >> ((~p & 3) << 1) & 3];
    }
    return SLX_OK;
}
extern "C" void slx_index_free(slx_index *idx) { delete idx; }

// ---------------------------------------------------------------- build
static const uint8_t NT4[256] = {
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4, 4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
    4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4};

extern "C" int slx_index_build(const char *const *names, const char *const *seqs, const int64_t *lens, int n, slx_index **out)
{
    if (!out) { slx_set_error("slx_index_build: out is null"); return SLX_EINVAL; }
    *out = nullptr;
    if (n <= 0 || !names || !seqs) { slx_set_error("slx_index_build: no reference sequences"); return SLX_EINVAL; }
    int64_t total = 0;
    std::vector<int64_t> L((size_t)n);
    for (int i = 0; i < n; ++i) {
        L[i] = lens ? lens[i] : (seqs[i] ? (int64_t)strlen(seqs[i]) : 0);
        if (!names[i] || !names[i][0] || !seqs[i] || L[i] <= 0) {
            slx_set_error("BWAIndex::Construct each reference must have non-empty Name and Seq");
            return SLX_EINVAL;
        }
        total += L[i];
    }
    // (the reference's in-memory builder stops at 2^31 - 1 BWT symbols -- is_bwt takes an int; larger indexes reach it as files
    // made by `bwa index`.  Here texts of 2^32 - 1 symbols and more take the 64-bit suffix sorter, slx_index_gpu64.hip.)
    const bool wide = (uint64_t)total * 2 + 1 >= (1ULL << 32) || getenv("SLX_BUILD64") != nullptr;
    slx_index *idx = new slx_index();
    idx->l_pac = total;
    // forward pac (first pass over the sequences; every N draws lrand48()&3)
    idx->pac.assign((size_t)(total / 4 + 2), 0);
    int64_t l = 0;
    for (int i = 0; i < n; ++i)
        for (int64_t k = 0; k < L[i]; ++k, ++l) {
            int c = NT4[(uint8_t)seqs[i][k]];
            if (c >= 4) c = (int)(lrand48() & 3);
            idx->pac[(size_t)(l >> 2)] |= (uint8_t)(c << ((~l & 3) << 1));
        }
    // BWT text: second pass (fresh draws at N, as the reference's second seqlib_make_pac call) ++ reverse complement
    std::vector<uint8_t> text((size_t)total * 2);
    l = 0;
    for (int i = 0; i < n; ++i)
        for (int64_t k = 0; k < L[i]; ++k, ++l) {
            int c = NT4[(uint8_t)seqs[i][k]];
            if (c >= 4) c = (int)(lrand48() & 3);
            text[(size_t)l] = (uint8_t)c;
        }
    for (int64_t k = total - 1; k >= 0; --k, ++l) text[(size_t)l] = (uint8_t)(3 - text[(size_t)k]);
    int rc = wide ? slx_gpu_build_fm64(idx, text.data(), (uint64_t)total * 2) : slx_gpu_build_fm(idx, text.data(), (uint64_t)total * 2);
    std::vector<uint8_t>().swap(text);
    if (rc != SLX_OK) { delete idx; return rc; }
    idx->seed = 11;
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        slx_ann a;
        a.offset = off; a.len = (int32_t)L[i]; a.n_ambs = 0; a.gi = 0; a.is_alt = 0;
        a.name = names[i]; a.anno = "(null)";
        idx->anns.push_back(a);
        off += L[i];
    }
    *out = idx;
    return SLX_OK;
}

// ---------------------------------------------------------------- write
static bool write_all(const std::string &fn, const void *p, size_t bytes, const char *mode, FILE **keep = nullptr)
{
    FILE *fp = fopen(fn.c_str(), mode);
    if (!fp) return false;
    bool ok = bytes == 0 || fwrite(p, 1, bytes, fp) == bytes;
    if (keep) *keep = fp; else fclose(fp);
    return ok;
}

extern "C" int slx_index_write(const slx_index *idx, const char *prefix)
{
    if (!idx) { slx_set_error("BWAIndex::writeIndex: no index loaded"); return SLX_EINVAL; }
    struct stat st;
    if (stat(prefix, &st) == 0 && S_ISDIR(st.st_mode)) { slx_set_error("BWAIndex::writeIndex: prefix refers to a directory"); return SLX_EIO; }
    std::string p(prefix);
    FILE *fp = nullptr;
    // .bwt : primary, L2[1..4], interleaved words
    if (!write_all(p + ".bwt", &idx->primary, 8, "wb", &fp)) { slx_set_error("cannot write %s.bwt", prefix); return SLX_EIO; }
    fwrite(idx->L2 + 1, 8, 4, fp); fwrite(idx->bwt.data(), 4, idx->bwt.size(), fp); fclose(fp);
    // .sa : primary, L2[1..4], sa_intv, seq_len, samples 1..n_sa-1
    if (!write_all(p + ".sa", &idx->primary, 8, "wb", &fp)) { slx_set_error("cannot write %s.sa", prefix); return SLX_EIO; }
    uint64_t v = (uint64_t)idx->sa_intv;
    fwrite(idx->L2 + 1, 8, 4, fp); fwrite(&v, 8, 1, fp); fwrite(&idx->seq_len, 8, 1, fp);
    fwrite(idx->sa.data() + 1, 8, idx->sa.size() - 1, fp); fclose(fp);
    // .ann / .amb
    fp = fopen((p + ".ann").c_str(), "w");
    if (!fp) { slx_set_error("cannot write %s.ann", prefix); return SLX_EIO; }
    fprintf(fp, "%lld %d %u\n", (long long)idx->l_pac, (int)idx->anns.size(), idx->seed);
    for (const slx_ann &a : idx->anns) {
        fprintf(fp, "%d %s", (int)a.gi, a.name.c_str());
        if (!a.anno.empty()) fprintf(fp, " %s\n", a.anno.c_str()); else fprintf(fp, "\n");
        fprintf(fp, "%lld %d %d\n", (long long)a.offset, a.len, a.n_ambs);
    }
    fclose(fp);
    fp = fopen((p + ".amb").c_str(), "w");
    if (!fp) { slx_set_error("cannot write %s.amb", prefix); return SLX_EIO; }
    fprintf(fp, "%lld %d %u\n", (long long)idx->l_pac, (int)idx->anns.size(), (unsigned)idx->ambs.size());
    for (const slx_amb &a : idx->ambs) fprintf(fp, "%lld %d %c\n", (long long)a.offset, a.len, a.amb);
    fclose(fp);
    // .pac : ceil(l_pac/4) bytes, a zero byte if l_pac%4==0, then l_pac%4
    size_t nb = (size_t)((idx->l_pac >> 2) + ((idx->l_pac & 3) == 0 ? 0 : 1));
    if (!write_all(p + ".pac", idx->pac.data(), nb, "wb", &fp)) { slx_set_error("cannot write %s.pac", prefix); return SLX_EIO; }
    uint8_t ct = 0;
    if (idx->l_pac % 4 == 0) fwrite(&ct, 1, 1, fp);
    ct = (uint8_t)(idx->l_pac % 4);
    fwrite(&ct, 1, 1, fp);
    fclose(fp);
    return SLX_OK;
}

// ---------------------------------------------------------------- load
static bool file_exists(const std::string &fn) { struct stat st; return stat(fn.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

extern "C" int slx_index_load(const char *prefix_, slx_index **out)
{
    if (!out) return SLX_EINVAL;
    *out = nullptr;
    std::string prefix(prefix_ ? prefix_ : "");
    // bwa_idx_infer_prefix: <prefix>.64.bwt is probed first
    if (file_exists(prefix + ".64.bwt")) prefix += ".64";
    else if (!file_exists(prefix + ".bwt")) { slx_set_error("Failed to load BWA index"); return SLX_EIO; }
    slx_index *idx = new slx_index();
    auto fail = [&](const char *what) { slx_set_error("Failed to load BWA index (%s%s)", prefix.c_str(), what); delete idx; return SLX_EIO; };
    FILE *fp = fopen((prefix + ".bwt").c_str(), "rb");
    if (!fp) return fail(".bwt");
    fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    if (sz < 40) { fclose(fp); return fail(".bwt"); }
    idx->bwt.resize((size_t)(sz - 40) >> 2);
    if (fread(&idx->primary, 8, 1, fp) != 1 || fread(idx->L2 + 1, 8, 4, fp) != 4 ||
        fread(idx->bwt.data(), 4, idx->bwt.size(), fp) != idx->bwt.size()) { fclose(fp); return fail(".bwt"); }
    fclose(fp);
    idx->seq_len = idx->L2[4];
    fp = fopen((prefix + ".sa").c_str(), "rb");
    if (!fp) return fail(".sa");
    uint64_t primary2, skip[4], intv, seqlen2;
    if (fread(&primary2, 8, 1, fp) != 1 || fread(skip, 8, 4, fp) != 4 || fread(&intv, 8, 1, fp) != 1 ||
        fread(&seqlen2, 8, 1, fp) != 1 || primary2 != idx->primary || seqlen2 != idx->seq_len || intv == 0 || (intv & (intv - 1))) { fclose(fp); return fail(".sa"); }
    idx->sa_intv = (int)intv;
    idx->sa.assign((size_t)((idx->seq_len + intv) / intv), 0);
    idx->sa[0] = (uint64_t)-1;
    if (fread(idx->sa.data() + 1, 8, idx->sa.size() - 1, fp) != idx->sa.size() - 1) { fclose(fp); return fail(".sa"); }
    fclose(fp);
    fp = fopen((prefix + ".ann").c_str(), "r");
    if (!fp) return fail(".ann");
    long long ll; int a, b; unsigned u;
    if (fscanf(fp, "%lld%d%u", &ll, &a, &u) != 3 || a < 0) { fclose(fp); return fail(".ann"); }
    idx->l_pac = ll; idx->seed = u;
    int n_seqs = a;
    for (int i = 0; i < n_seqs; ++i) {
        slx_ann an;
        char name[4096], line[8192]; char *q = line; int c;
        if (fscanf(fp, "%u%4095s", &an.gi, name) != 2) { fclose(fp); return fail(".ann"); }
        an.name = name;
        c = fgetc(fp);
        while (c != '\n' && c != EOF && q - line < (long)sizeof(line) - 1) { *q++ = (char)c; c = fgetc(fp); }
        *q = 0;
        an.anno = (q - line > 1) ? line + 1 : line;
        if (fscanf(fp, "%lld%d%d", &ll, &a, &b) != 3) { fclose(fp); return fail(".ann"); }
        an.offset = ll; an.len = a; an.n_ambs = b; an.is_alt = 0;
        idx->anns.push_back(an);
    }
    fclose(fp);
    fp = fopen((prefix + ".alt").c_str(), "r");   // bns_restore: the first field of every non-header line names an ALT contig
    if (fp) {
        std::string tok;
        int c;
        while ((c = fgetc(fp)) != EOF) {
            if (c == '\t' || c == '\n' || c == '\r') {
                if (tok.empty() || tok[0] != '@')   // (bwa's name hash keeps the LAST contig of a given name)
                    for (size_t k = idx->anns.size(); k-- > 0;) if (idx->anns[k].name == tok) { idx->anns[k].is_alt = 1; break; }
                while (c != '\n' && c != EOF) c = fgetc(fp);
                tok.clear();
            } else tok.push_back((char)c);
        }
        fclose(fp);
    }
    fp = fopen((prefix + ".amb").c_str(), "r");
    if (!fp) return fail(".amb");
    if (fscanf(fp, "%lld%d%d", &ll, &a, &b) != 3) { fclose(fp); return fail(".amb"); }
    for (int i = 0; i < b; ++i) {
        slx_amb am; char ch[8];
        if (fscanf(fp, "%lld%d%7s", &ll, &a, ch) != 3) { fclose(fp); return fail(".amb"); }
        am.offset = ll; am.len = a; am.amb = ch[0];
        idx->ambs.push_back(am);
    }
    fclose(fp);
    fp = fopen((prefix + ".pac").c_str(), "rb");
    if (!fp) return fail(".pac");
    idx->pac.assign((size_t)(idx->l_pac / 4 + 2), 0);
    if (fread(idx->pac.data(), 1, (size_t)(idx->l_pac / 4 + 1), fp) < (size_t)((idx->l_pac + 3) / 4)) { fclose(fp); return fail(".pac"); }
    fclose(fp);
    if (idx->seq_len != (uint64_t)idx->l_pac * 2) return fail(": .bwt and .ann disagree");
    *out = idx;
    return SLX_OK;
}

// ---- how host threads wait for a stream (slx_internal.h: slx_wait_stream) ----
bool slx_wait_sleeps()
{
    const char *e = getenv("SEQLIB_AMD_WAIT");
    return e && !strcmp(e, "sleep");
}
