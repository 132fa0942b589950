#include "orc_fml.h"
orc_fml_utg *orc_fml_assemble(const orc_fml_opt *opt, int n, orc_fseq *seqs, int *n_utg) { *n_utg = 0; return 0; }
orc_fml_utg *orc_fml_direct_assemble(orc_fml_opt *opt, float kcov, int n, orc_fseq *seqs, int *n_utg) { *n_utg = 0; return 0; }
void orc_fml_utg_destroy(int n_utg, orc_fml_utg *utg) {}
