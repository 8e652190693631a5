// Host prefix beam search of the hctr engine (part of libhctr_hip.so; pure host code).
//
// Behavioural contract: utils/ctc_codec.py:124-285 and Beam :288-307 of the reference -
// __cbs_full__, __cbs_skip__ and __context_beam_search__ - on the device front end's output
// (log-softmax top-k, blank log-prob, thresholded candidate lists). Hypotheses are nodes of a
// label-id prefix trie instead of Python strings (the vocabulary is assumed duplicate-free, as the
// reference's char->index dict also assumes). Everything that decides a tie is kept:
//   * scores are float64 sums of float32 log-probs, merged with numpy's logaddexp formula;
//   * new hypotheses are created in first-touch order (Python dict insertion order, :233-265);
//   * the cut to beam_size is a STABLE descending sort on total() (sorted(..., reverse=True), :283);
//   * the skip variant updates beams in place with neither merge nor re-sort (:147-171).
// Build with -ffp-contract=off so a*b+c is never fused (Python evaluates it unfused).
#include "../../include/hctr_hip.h"
#include "ngram_lm.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

const double kNegInf = -std::numeric_limits<double>::infinity();
const double kLogE2 = 0.693147180559945309417232121458176568;   // numpy NPY_LOGE2

// numpy's npy_logaddexp (numpy/_core/src/npymath/npy_math_internal.h.src)
inline double logaddexp(double x, double y) {
    if (x == y) return x + kLogE2;
    const double tmp = x - y;
    if (tmp > 0) return x + std::log1p(std::exp(-tmp));
    if (tmp <= 0) return y + std::log1p(std::exp(tmp));
    return tmp;   // NaN
}

// Prefixes live in a per-line trie: a hypothesis is a node id, extending a prefix is one hash lookup
// and two prefixes are equal iff their node ids are (the reference compares/concatenates Python
// strings of up to ~1000 characters at every step; same semantics, O(1) instead of O(length)).
struct Node {
    int parent;
    int32_t label;      // -1 at the root
    int len;
    double toy;         // built-in LMs: left-to-right running score of this prefix (toy bigram or n-gram)
    uint32_t cp;        // toy LM: code point of `label` (0 at the root); n-gram LM: its word id
};

// (parent node, label) -> child node: open addressing in one flat array. The search of a 2000-column line creates
// ~200 000 nodes; a node-based std::unordered_map made one allocation per insert, and 64 lines decoding at once spent
// their time in the allocator and in page faults (80 -> 270 ms per 64-line chunk from run to run). The table is
// reused from line to line by its worker thread (reset keeps the capacity).
struct KidMap {
    static constexpr uint64_t kEmpty = ~0ull;              // (node < 2^31, so no key is all ones)
    std::vector<uint64_t> keys;
    std::vector<int> vals;
    size_t mask = 0, count = 0;
    static size_t mix(uint64_t k) {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
        return (size_t)k;
    }
    void reset() {
        if (keys.empty()) { keys.assign(1 << 12, kEmpty); vals.assign(1 << 12, 0); }
        else std::fill(keys.begin(), keys.end(), kEmpty);
        mask = keys.size() - 1;
        count = 0;
    }
    int find(uint64_t key) const {
        for (size_t i = mix(key) & mask;; i = (i + 1) & mask) {
            if (keys[i] == key) return vals[i];
            if (keys[i] == kEmpty) return -1;
        }
    }
    void insert(uint64_t key, int val) {                   // key must be absent
        if ((count + 1) * 2 > keys.size()) {
            std::vector<uint64_t> ok(keys.size() * 2, kEmpty);
            std::vector<int> ov(keys.size() * 2, 0);
            ok.swap(keys); ov.swap(vals);
            mask = keys.size() - 1;
            for (size_t j = 0; j < ok.size(); ++j)
                if (ok[j] != kEmpty) {
                    size_t i = mix(ok[j]) & mask;
                    while (keys[i] != kEmpty) i = (i + 1) & mask;
                    keys[i] = ok[j]; vals[i] = ov[j];
                }
        }
        size_t i = mix(key) & mask;
        while (keys[i] != kEmpty) i = (i + 1) & mask;
        keys[i] = key; vals[i] = val;
        ++count;
    }
};

struct Trie {
    std::vector<Node> nodes;
    KidMap kids;
    const int32_t* cps = nullptr;
    const hctr_ngram* lm = nullptr;       // built-in ARPA LM (cps then maps label -> word id)
    void reset(const int32_t* codepoints, const hctr_ngram* ngram = nullptr) {
        cps = codepoints; lm = ngram;
        nodes.clear();
        nodes.push_back(Node{-1, -1, 0, 0.0, 0u});
        kids.reset();
    }
    // the last (order-1) word ids of prefix(node), oldest first, preceded by <s> when the prefix is short
    int lm_context(int node, int32_t* ctx) const {
        const int want = hctr::ngram_order(lm) - 1;
        int n = 0;
        int32_t rev[16];
        for (int nd = node; nd > 0 && n < want; nd = nodes[nd].parent) rev[n++] = (int32_t)nodes[nd].cp;
        if (n < want && hctr::ngram_bos(lm) >= 0) rev[n++] = hctr::ngram_bos(lm);
        for (int i = 0; i < n; ++i) ctx[i] = rev[n - 1 - i];
        return n;
    }
    // one bigram term of the deterministic toy LM (same formula as oracle/ctc_ref.py toy_bigram_score)
    static double toy_term(uint64_t prev, uint64_t c) {
        uint64_t h = (prev * 2654435761ull + c * 40503ull + 12345ull) & 0xFFFFFFFFull;
        h ^= h >> 15;
        h = (h * 2246822519ull) & 0xFFFFFFFFull;
        h ^= h >> 13;
        return -4.0 * ((double)(h & 0xFFFFull) / 65536.0);
    }
    int child(int node, int32_t label) {
        const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)label;
        const int hit = kids.find(key);
        if (hit >= 0) return hit;
        const Node p = nodes[node];
        const uint32_t cp = cps ? (uint32_t)cps[label] : 0u;
        double sc = 0.0;
        if (lm) {
            int32_t ctx[16];
            const int n = lm_context(node, ctx);
            sc = p.toy + hctr::ngram_word_logp(lm, ctx, n, (int32_t)cp);
        } else if (cps) {
            sc = p.toy + toy_term(p.cp, cp);
        }
        nodes.push_back(Node{node, label, p.len + 1, sc, cp});
        kids.insert(key, (int)nodes.size() - 1);
        return (int)nodes.size() - 1;
    }
    void append_labels(int node, std::vector<int32_t>& out) const {      // root -> node order
        const size_t at = out.size();
        out.resize(at + nodes[node].len);
        for (int n = node, i = nodes[node].len - 1; n > 0; n = nodes[n].parent, --i) out[at + i] = nodes[n].label;
    }
    // n-gram score of prefix(node) + suffix: the cached prefix score, then the suffix words in order
    double ngram_score(int node, const std::vector<int32_t>& suffix) const {
        double s = nodes[node].toy;
        int32_t ctx[32];
        int n = lm_context(node, ctx);
        for (int32_t l : suffix) {
            const int32_t w = cps[l];
            s += hctr::ngram_word_logp(lm, ctx, n, w);
            if (n == 31) { memmove(ctx, ctx + 1, 30 * sizeof(int32_t)); n = 30; }
            ctx[n++] = w;
        }
        return s;
    }
    // toy LM score of prefix(node) + suffix, summed left to right exactly like the oracle
    double toy_score(int node, const std::vector<int32_t>& suffix) const {
        double s = nodes[node].toy;
        uint64_t prev = nodes[node].cp;
        for (int32_t l : suffix) {
            const uint64_t c = (uint32_t)cps[l];
            s += toy_term(prev, c);
            prev = c;
        }
        return s;
    }
};

struct Hyp {
    int node;
    double pb, pnb, pt;
    double prob() const { return logaddexp(pb, pnb); }
    double total() const { return logaddexp(pb, pnb) + pt; }
};

inline Hyp fresh_hyp() { return Hyp{0, 0.0, kNegInf, 0.0}; }      // Beam(), :289-297

struct LineInput {
    int W, B, C, k, b;
    const int32_t* topk_idx;
    const float* topk_logp;
    const float* blank_logp;
    const int64_t* cand_off;
    const int32_t* cand_idx;
    const float* cand_logp;
    const float* full_logp;     // optional [W][B][C] log-probs (needed for LM-proposed candidates)
};

struct Scratch {                 // per line, reused across steps
    std::vector<Hyp> gen;
    std::vector<int> slot_of;    // node id -> index in gen for the current step (valid iff stamp matches)
    std::vector<int> stamp;
    int epoch = 0;
    std::vector<int32_t> ids, offs, ling;
    int ling_stride = 0;         // slots per beam in ling (search_depth, or more when the LM returned longer lists)
    std::vector<double> scores, tot;
    std::vector<int> order;
};

// one prefix-beam step (__context_beam_search__, :212-285). cands/plog: visual candidates and their
// log-probs at this time step. Returns HCTR_OK or a callback failure code.
int beam_step(const hctr_beam_params& P, const LineInput& in, int t, Trie& trie, Scratch& S, std::vector<Hyp>& beams,
              const int32_t* cands, const float* plog, int ncand, const std::vector<int32_t>& suffix) {
    const int unk = in.C - 1;
    // Step 1: optional LM-proposed candidates per beam (:215-227)
    if (P.next_cb) {
        S.ids.clear();
        S.offs.assign(1, 0);
        for (const Hyp& h : beams) {
            trie.append_labels(h.node, S.ids);
            S.offs.push_back((int32_t)S.ids.size());
        }
        // the reference chains WHATEVER list the LM returns (:225-226): the callback fills up to `slots` labels per
        // beam (short lists padded with the <unknown> id, which is skipped below) and returns 0, or - when some list is
        // longer - the number of slots it needs, and is then called again with that many
        int slots = std::max(P.search_depth, S.ling_stride);
        for (int attempt = 0;; ++attempt) {
            S.ling.assign((size_t)beams.size() * slots, 0);
            const int rc = P.next_cb(P.user, (int)beams.size(), S.ids.data(), S.offs.data(), slots, S.ling.data());
            if (rc == 0) break;
            if (rc < 0) return rc;
            if (rc <= slots || attempt >= 2) return HCTR_ERR_ARG;       // (a callback that keeps asking is broken)
            slots = rc;
        }
        S.ling_stride = slots;
    }
    // Step 2: extend (:229-265). gen keeps first-touch order (Python dict insertion order).
    std::vector<Hyp>& gen = S.gen;
    gen.clear();
    ++S.epoch;
    auto slot = [&](int node) -> int {
        if ((size_t)node >= S.stamp.size()) {
            S.stamp.resize(trie.nodes.size() + 64, 0);
            S.slot_of.resize(S.stamp.size(), 0);
        }
        if (S.stamp[node] == S.epoch) return S.slot_of[node];
        S.stamp[node] = S.epoch;
        S.slot_of[node] = (int)gen.size();
        gen.push_back(Hyp{node, kNegInf, kNegInf, 0.0});
        return (int)gen.size() - 1;
    };
    for (size_t bi = 0; bi < beams.size(); ++bi) {
        const Hyp h = beams[bi];
        const double hprob = h.prob();
        const int tail = trie.nodes[h.node].label;               // -1 for the empty prefix
        const int nl = (P.next_cb && h.node != 0) ? S.ling_stride : 0;
        for (int ci = 0; ci < ncand + nl; ++ci) {
            int idx;
            double p;
            if (ci < ncand) {
                idx = cands[ci];
                p = (double)plog[ci];
            } else {
                idx = S.ling[bi * (size_t)S.ling_stride + (ci - ncand)];
                if (idx < 0 || idx >= in.C) return HCTR_ERR_ARG;
                if (idx >= unk) continue;
                p = (double)in.full_logp[((size_t)t * in.B + in.b) * in.C + idx];
            }
            if (idx >= unk) continue;                                  // ignore <unknown> (:238-239)
            const int ps = slot(h.node);
            if (idx == 0) {                                            // blank: only pb (:246-249)
                gen[ps].pb = logaddexp(gen[ps].pb, hprob + p);
                continue;
            }
            const int es = slot(trie.child(h.node, idx));
            if (idx != tail) {
                gen[es].pnb = logaddexp(gen[es].pnb, hprob + p);        // (:256-258)
            } else {
                gen[es].pnb = logaddexp(gen[es].pnb, h.pb + p);         // not merged (:260-262)
                gen[ps].pnb = logaddexp(gen[ps].pnb, h.pnb + p);        // merged     (:263-265)
            }
        }
    }
    // Step 3: LM score + length bonus, stable sort, cut (:267-285)
    const size_t n = gen.size();
    if (n) {
        S.scores.assign(n, 0.0);
        if (P.builtin_lm == 2) {
            for (size_t i = 0; i < n; ++i) S.scores[i] = trie.toy_score(gen[i].node, suffix);
        } else if (P.builtin_lm == 3) {
            for (size_t i = 0; i < n; ++i) S.scores[i] = trie.ngram_score(gen[i].node, suffix);
        } else if (P.builtin_lm == 0) {
            S.ids.clear();
            S.offs.assign(1, 0);
            for (const Hyp& g : gen) {
                trie.append_labels(g.node, S.ids);
                S.ids.insert(S.ids.end(), suffix.begin(), suffix.end());
                S.offs.push_back((int32_t)S.ids.size());
            }
            const int rc = P.score_cb(P.user, (int)n, S.ids.data(), S.offs.data(), S.scores.data());
            if (rc != 0) return rc;
        }
        for (size_t i = 0; i < n; ++i)
            gen[i].pt = S.scores[i] * P.lm_panelty + (double)trie.nodes[gen[i].node].len * P.len_bonus;
    }
    S.tot.resize(n);
    S.order.resize(n);
    for (size_t i = 0; i < n; ++i) { S.tot[i] = gen[i].total(); S.order[i] = (int)i; }
    std::stable_sort(S.order.begin(), S.order.end(), [&](int a, int b) { return S.tot[a] > S.tot[b]; });
    const size_t keep = std::min<size_t>(n, (size_t)std::max(P.beam_size, 0));
    beams.clear();
    for (size_t i = 0; i < keep; ++i) beams.push_back(gen[S.order[i]]);
    return HCTR_OK;
}

// everything a line's search allocates, owned by a worker thread and reused from line to line (capacities are kept;
// Scratch's epoch keeps counting, so stamps left by an earlier line never match)
struct LineWork {
    Trie trie;
    Scratch S;
    std::vector<Hyp> beams;
    std::vector<int32_t> suffix, line_lab, line_t, best;
};

int decode_line(const hctr_beam_params& P, const LineInput& in, int32_t* out_labels, int32_t* out_len, LineWork& wk) {
    const int W = in.W, B = in.B, C = in.C, k = in.k, b = in.b;
    const int unk = C - 1;
    *out_len = 0;
    // greedy pass with time stamps (:133-140, :188-195)
    std::vector<int32_t>& line_lab = wk.line_lab;
    std::vector<int32_t>& line_t = wk.line_t;
    line_lab.clear();
    line_t.clear();
    int prev = -1;
    for (int t = 0; t < W; ++t) {
        const int c1 = in.topk_idx[((size_t)t * B + b) * k];
        if (c1 != 0 && c1 != unk && !(t > 0 && prev == c1)) { line_lab.push_back(c1); line_t.push_back(t); }
        prev = c1;
    }
    if (line_lab.empty()) return HCTR_ERR_EMPTY_LINE;              // top_line[-1] -> IndexError (:143,198)
    int end_step = line_t.back() + 4;
    if (end_step >= W) end_step = W;
    Trie& trie = wk.trie;
    trie.reset(P.builtin_lm == 2 ? P.label_codepoints : (P.builtin_lm == 3 ? P.label_words : nullptr),
               P.builtin_lm == 3 ? P.ngram : nullptr);
    Scratch& S = wk.S;
    std::vector<Hyp>& beams = wk.beams;
    beams.assign(1, fresh_hyp());
    std::vector<int32_t>& suffix = wk.suffix;
    size_t first_after = 0;                                         // first greedy entry with ts > t
    const int depth = std::min(P.search_depth, k);
    for (int t = 0; t < end_step; ++t) {
        while (first_after < line_t.size() && line_t[first_after] <= t) ++first_after;
        const size_t r = (size_t)t * B + b;
        auto make_suffix = [&]() {
            suffix.clear();
            for (size_t i = first_after; i < line_lab.size() && suffix.size() < 4; ++i) suffix.push_back(line_lab[i]);
        };
        if (!P.skip_search) {
            make_suffix();
            const int rc = beam_step(P, in, t, trie, S, beams, in.topk_idx + r * k, in.topk_logp + r * k, depth, suffix);
            if (rc != HCTR_OK) return rc;
            continue;
        }
        const int64_t c0 = in.cand_off[r], n = in.cand_off[r + 1] - c0;
        if (n != 1) {
            make_suffix();
            const int rc = beam_step(P, in, t, trie, S, beams, in.cand_idx + c0, in.cand_logp + c0, (int)n, suffix);
            if (rc != HCTR_OK) return rc;
            continue;
        }
        // exactly one class above the prune threshold: in-place update, no LM (:147-171)
        const int c = in.cand_idx[c0];
        if (c >= unk) continue;
        const double pc = (double)in.cand_logp[c0];
        const double p0 = (double)in.blank_logp[r];
        for (Hyp& h : beams) {
            const int tail = trie.nodes[h.node].label;
            if (c == 0) {
                h.pb = h.prob() + pc;                                 // pc == row[0] here
            } else if (c != tail) {
                const double pr = h.prob();
                h.node = trie.child(h.node, c);
                h.pnb = pr + pc;
                h.pb = kNegInf;
            } else if (h.pb != kNegInf) {
                h.node = trie.child(h.node, c);
                h.pnb = h.pb + pc;
                h.pb = kNegInf;
            } else {
                h.pb = h.prob() + p0;
                h.pnb = h.pnb + pc;
            }
        }
    }
    if (beams.empty()) return HCTR_ERR_EMPTY_LINE;                  // kept_beams[0] -> IndexError (:179,208)
    std::vector<int32_t>& best = wk.best;
    best.clear();
    trie.append_labels(beams[0].node, best);
    *out_len = (int32_t)best.size();
    if (!best.empty()) memcpy(out_labels, best.data(), best.size() * sizeof(int32_t));
    return HCTR_OK;
}

}  // namespace

extern "C" int hctr_beam_search(const hctr_beam_params* p, int W, int B, int C, int k,
                                const int32_t* topk_idx, const float* topk_logp, const float* blank_logp,
                                const int64_t* cand_off, const int32_t* cand_idx, const float* cand_logp,
                                const float* full_logp_wbc,
                                int32_t* out_labels, int32_t* out_lengths, int32_t* line_status) {
    if (!p || W < 0 || B < 0 || C < 2 || k < 1) return HCTR_ERR_ARG;
    if (B == 0) return HCTR_OK;
    if (!topk_idx || !topk_logp || !blank_logp || !out_labels || !out_lengths || !line_status) return HCTR_ERR_ARG;
    if (p->skip_search && (!cand_off || (cand_off[(size_t)W * B] > 0 && (!cand_idx || !cand_logp)))) return HCTR_ERR_ARG;
    if (p->builtin_lm == 0 && !p->score_cb) return HCTR_ERR_ARG;
    if (p->builtin_lm == 2 && !p->label_codepoints) return HCTR_ERR_ARG;
    if (p->builtin_lm == 3 && (!p->ngram || !p->label_words)) return HCTR_ERR_ARG;
    if (p->builtin_lm < 0 || p->builtin_lm > 3) return HCTR_ERR_ARG;
    if (p->next_cb && !full_logp_wbc) return HCTR_ERR_ARG;
    if (p->search_depth < 1 || p->beam_size < 0) return HCTR_ERR_ARG;
    const bool callbacks = p->builtin_lm == 0 || p->next_cb != nullptr;
    int nthreads = callbacks ? 1 : std::max(1, p->num_threads);
    nthreads = std::min(nthreads, B);
    for (int b = 0; b < B; ++b) { line_status[b] = HCTR_OK; out_lengths[b] = 0; }
    // No C++ exception leaves this function or a worker thread (include/hctr_hip.h): a line whose search runs out
    // of host memory reports HCTR_ERR_NOMEM, stops the remaining work and becomes the return value.
    std::atomic<int> next(0);
    std::atomic<int> fault(HCTR_OK);
    auto worker = [&]() noexcept {
        LineWork* wk = nullptr;
        try { wk = new LineWork(); } catch (...) { fault.store(HCTR_ERR_NOMEM); next.store(B); return; }
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            int st;
            try {
                LineInput in{W, B, C, k, b, topk_idx, topk_logp, blank_logp, cand_off, cand_idx, cand_logp, full_logp_wbc};
                st = decode_line(*p, in, out_labels + (size_t)b * W, out_lengths + b, *wk);
            } catch (const std::bad_alloc&) {
                st = HCTR_ERR_NOMEM;
            } catch (...) {
                st = HCTR_ERR_STATE;
            }
            line_status[b] = st;
            if (st == HCTR_ERR_NOMEM || st == HCTR_ERR_STATE) {
                fault.store(st);
                next.store(B);
            }
        }
        delete wk;
    };
    if (nthreads == 1) {
        worker();
    } else {
        // a thread that cannot be created (EAGAIN under a process / address-space limit) is not fatal: the threads
        // that did start - or this one - work through the lines
        std::vector<std::thread> th;
        try {
            th.reserve((size_t)nthreads);
            for (int i = 0; i < nthreads; ++i) th.emplace_back(worker);
        } catch (...) {
        }
        if (th.empty()) worker();
        for (auto& t : th) t.join();
    }
    if (fault.load() != HCTR_OK) return fault.load();
    for (int b = 0; b < B; ++b)
        if (line_status[b] != HCTR_OK) return line_status[b];
    return HCTR_OK;
}
