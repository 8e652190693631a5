// Internal interface of the ARPA n-gram language model (ngram_lm.cpp) used by beam_search.cpp.
#pragma once
#include <stdint.h>

struct hctr_ngram;

namespace hctr {
// log10 P(word | context). ctx holds the previous word ids, most recent LAST; only the last
// (order - 1) are used. Word id -1 = out of vocabulary (scored as <unk>).
double ngram_word_logp(const hctr_ngram* lm, const int32_t* ctx, int nctx, int32_t word);
int ngram_order(const hctr_ngram* lm);
int32_t ngram_bos(const hctr_ngram* lm);
}  // namespace hctr
