// ARPA back-off n-gram language model for the beam search's `ngram.score(sentence, eos=False)` hook
// (utils/ctc_codec.py:276-281; the reference uses the kenlm Python module there, which is not
// vendored: third-party/README.md:25). Host code; part of libhctr_hip.so.
//
// Semantics follow kenlm's Model.score(sentence, bos=True, eos=False): the sentence is split on
// whitespace, scoring starts in the <s> context, every word contributes log10 P(w | history) with
// Katz back-off over the ARPA tables (longest matching n-gram, plus the back-off weights of the
// contexts that had to be shortened), out-of-vocabulary words score as <unk> (or -100 if the model
// has none, kenlm's unknown_missing_logprob). Parity with the real kenlm binary is UNPINNED (module
// absent here); the scorer is pinned by oracle/ctc_ref.py's ArpaRef and hand-computed cases.
#include "ngram_lm.h"

#include "../../include/hctr_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <new>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

struct hctr_ngram {
    int order = 0;
    std::unordered_map<std::string, int32_t> vocab;
    std::vector<std::string> words;
    struct Entry { float logp; float backoff; };
    // key: the n-gram's word ids packed as raw int32 bytes
    std::vector<std::unordered_map<std::string, Entry>> tables;   // tables[n-1]
    int32_t bos = -1, eos = -1, unk = -1;
    std::string err;
};

namespace {

std::string g_ngram_error;

inline std::string key_of(const int32_t* ids, int n) { return std::string((const char*)ids, (size_t)n * 4); }

int32_t intern(hctr_ngram* lm, const std::string& w) {
    auto it = lm->vocab.find(w);
    if (it != lm->vocab.end()) return it->second;
    const int32_t id = (int32_t)lm->words.size();
    lm->vocab.emplace(w, id);
    lm->words.push_back(w);
    return id;
}

}  // namespace

namespace hctr {

int ngram_order(const hctr_ngram* lm) { return lm->order; }
int32_t ngram_bos(const hctr_ngram* lm) { return lm->bos; }

double ngram_word_logp(const hctr_ngram* lm, const int32_t* ctx, int nctx, int32_t word) {
    if (word < 0) word = lm->unk;
    if (word < 0) return -100.0;                       // no <unk> in the model
    int use = nctx < lm->order - 1 ? nctx : lm->order - 1;
    const int32_t* c = ctx + (nctx - use);             // most recent `use` words
    double backoff = 0.0;
    int32_t buf[16];
    for (int n = use; n >= 0; --n) {                   // n context words + the word itself
        if (n > 0) memcpy(buf, c + (use - n), (size_t)n * 4);      // (ctx may be NULL when nctx == 0)
        buf[n] = word;
        const auto& tab = lm->tables[n];
        auto it = tab.find(key_of(buf, n + 1));
        if (it != tab.end()) return backoff + (double)it->second.logp;
        if (n > 0) {                                   // shorten the context: pay its back-off weight
            auto bt = lm->tables[n - 1].find(key_of(c + (use - n), n));
            if (bt != lm->tables[n - 1].end()) backoff += (double)bt->second.backoff;
        }
    }
    // the unigram itself is missing (a word id without a 1-gram line): score as <unk>
    if (word != lm->unk && lm->unk >= 0) {
        auto it = lm->tables[0].find(key_of(&lm->unk, 1));
        if (it != lm->tables[0].end()) return backoff + (double)it->second.logp;
    }
    return backoff - 100.0;
}

}  // namespace hctr

extern "C" {

const char* hctr_ngram_last_error(void) { return g_ngram_error.c_str(); }

// No C++ exception crosses the ABI (include/hctr_hip.h): the loaders and scorers below catch everything; an
// allocation failure while reading a model is HCTR_ERR_NOMEM, while scoring it is a NaN score / word id -1.
static int ngram_load_impl(const char* arpa_path, hctr_ngram** out);

int hctr_ngram_load(const char* arpa_path, hctr_ngram** out) {
    if (!arpa_path || !out) {
        try { g_ngram_error = "hctr_ngram_load: arpa_path / out is NULL"; } catch (...) {}
        return HCTR_ERR_ARG;
    }
    *out = nullptr;
    try {
        return ngram_load_impl(arpa_path, out);
    } catch (const std::bad_alloc&) {
        try { g_ngram_error = "out of host memory while reading the ARPA file"; } catch (...) {}
        return HCTR_ERR_NOMEM;
    } catch (...) {
        try { g_ngram_error = "unexpected C++ exception while reading the ARPA file"; } catch (...) {}
        return HCTR_ERR_STATE;
    }
}

static int ngram_load_impl(const char* arpa_path, hctr_ngram** out) {
    std::ifstream f(arpa_path);
    if (!f) { g_ngram_error = std::string("cannot open ARPA file: ") + arpa_path; return HCTR_ERR_ARG; }
    std::unique_ptr<hctr_ngram> holder(new hctr_ngram());
    hctr_ngram* lm = holder.get();
    std::string line;
    int section = 0;                                    // 0 = header, n = inside \n-grams:
    bool saw_data = false;
    size_t lineno = 0;
    while (std::getline(f, line)) {
        ++lineno;
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        if (line.empty()) continue;
        if (line == "\\data\\") { saw_data = true; continue; }
        if (line == "\\end\\") break;
        if (line[0] == '\\') {
            int n = 0;
            if (sscanf(line.c_str(), "\\%d-grams:", &n) == 1 && n >= 1 && n <= 15) {
                section = n;
                if ((int)lm->tables.size() < n) lm->tables.resize(n);
                if (n > lm->order) lm->order = n;
                continue;
            }
            g_ngram_error = "unrecognised ARPA section at line " + std::to_string(lineno);
            return HCTR_ERR_ARG;
        }
        if (section == 0) continue;                      // "ngram N=count" lines
        // logp <TAB> w1 ... wn [<TAB> backoff]
        std::istringstream ss(line);
        std::vector<std::string> tok;
        std::string t;
        while (ss >> t) tok.push_back(t);
        if ((int)tok.size() != section + 1 && (int)tok.size() != section + 2) {
            g_ngram_error = "malformed " + std::to_string(section) + "-gram at line " + std::to_string(lineno);
            return HCTR_ERR_ARG;
        }
        hctr_ngram::Entry e;
        e.logp = strtof(tok[0].c_str(), nullptr);
        e.backoff = (int)tok.size() == section + 2 ? strtof(tok[section + 1].c_str(), nullptr) : 0.f;
        int32_t ids[16];
        for (int i = 0; i < section; ++i) ids[i] = intern(lm, tok[1 + i]);
        lm->tables[section - 1][key_of(ids, section)] = e;
    }
    if (!saw_data || lm->order == 0) {
        g_ngram_error = std::string("not an ARPA file (no \\data\\ / n-gram sections): ") + arpa_path;
        return HCTR_ERR_ARG;
    }
    auto find = [&](const char* w) { auto it = lm->vocab.find(w); return it == lm->vocab.end() ? -1 : it->second; };
    lm->bos = find("<s>");
    lm->eos = find("</s>");
    lm->unk = find("<unk>");
    *out = holder.release();
    return HCTR_OK;
}

void hctr_ngram_free(hctr_ngram* lm) { delete lm; }

int hctr_ngram_order(const hctr_ngram* lm) { return lm ? lm->order : 0; }

int32_t hctr_ngram_word_id(const hctr_ngram* lm, const char* word_utf8) {
    if (!lm || !word_utf8) return -1;
    try {
        auto it = lm->vocab.find(word_utf8);
        return it == lm->vocab.end() ? -1 : it->second;
    } catch (...) {
        return -1;
    }
}

// kenlm.Model.score(sentence, bos, eos): log10 probability of a whitespace-separated sentence
static double ngram_score_impl(const hctr_ngram* lm, const char* sentence_utf8, int bos, int eos);

double hctr_ngram_score(const hctr_ngram* lm, const char* sentence_utf8, int bos, int eos) {
    if (!lm || !sentence_utf8) return 0.0;
    try {
        return ngram_score_impl(lm, sentence_utf8, bos, eos);
    } catch (...) {
        return std::numeric_limits<double>::quiet_NaN();
    }
}

static double ngram_score_impl(const hctr_ngram* lm, const char* sentence_utf8, int bos, int eos) {
    std::vector<int32_t> ctx;
    if (bos && lm->bos >= 0) ctx.push_back(lm->bos);
    double total = 0.0;
    std::istringstream ss(sentence_utf8);
    std::string w;
    while (ss >> w) {
        auto it = lm->vocab.find(w);
        const int32_t id = it == lm->vocab.end() ? -1 : it->second;
        total += hctr::ngram_word_logp(lm, ctx.data(), (int)ctx.size(), id);
        ctx.push_back(id < 0 ? lm->unk : id);
    }
    if (eos && lm->eos >= 0) total += hctr::ngram_word_logp(lm, ctx.data(), (int)ctx.size(), lm->eos);
    return total;
}

}  // extern "C"
