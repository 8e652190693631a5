// Host prefix beam search of the hctr engine (part of libhctr_hip.so; pure host code).
//
// Behavioural contract: utils/ctc_codec.py:124-285 and Beam :288-307 of the reference -
// __cbs_full__, __cbs_skip__ and __context_beam_search__ - on the device front end's output
// (log-softmax top-k, blank log-prob, thresholded candidate lists). Hypotheses are label-id
// sequences instead of Python strings (the vocabulary is assumed duplicate-free, as the reference's
// char->index dict also assumes). Everything that decides a tie is kept:
//   * scores are float64 sums of float32 log-probs, merged with numpy's logaddexp formula;
//   * new hypotheses are created in first-touch order (Python dict insertion order, :233-265);
//   * the cut to beam_size is a STABLE descending sort on total() (sorted(..., reverse=True), :283);
//   * the skip variant updates beams in place with neither merge nor re-sort (:147-171).
// Build with -ffp-contract=off so a*b+c is never fused (Python evaluates it unfused).
#include "../../include/hctr_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <thread>
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

struct Hyp {
    std::vector<int32_t> prefix;
    double pb, pnb, pt;
    double prob() const { return logaddexp(pb, pnb); }
    double total() const { return logaddexp(pb, pnb) + pt; }
};

inline Hyp fresh_hyp() { return Hyp{{}, 0.0, kNegInf, 0.0}; }      // Beam(), :289-297

// deterministic toy bigram LM (same formula as oracle/ctc_ref.py toy_bigram_score)
double toy_bigram(const int32_t* ids, int n, const int32_t* cps) {
    double s = 0.0;
    uint64_t prev = 0;
    for (int i = 0; i < n; ++i) {
        const uint64_t c = (uint64_t)(uint32_t)cps[ids[i]];
        uint64_t h = (prev * 2654435761ull + c * 40503ull + 12345ull) & 0xFFFFFFFFull;
        h ^= h >> 15;
        h = (h * 2246822519ull) & 0xFFFFFFFFull;
        h ^= h >> 13;
        s += -4.0 * ((double)(h & 0xFFFFull) / 65536.0);
        prev = c;
    }
    return s;
}

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

// one prefix-beam step (__context_beam_search__, :212-285). cands/plog: visual candidates and their
// log-probs at this time step. Returns HCTR_OK or a callback failure code.
int beam_step(const hctr_beam_params& P, const LineInput& in, int t, std::vector<Hyp>& beams,
              const int32_t* cands, const float* plog, int ncand, const std::vector<int32_t>& suffix) {
    const int unk = in.C - 1;
    // Step 1: optional LM-proposed candidates per beam (:215-227)
    std::vector<int32_t> ling;
    if (P.next_cb) {
        std::vector<int32_t> ids, offs(1, 0);
        for (const Hyp& h : beams) {
            ids.insert(ids.end(), h.prefix.begin(), h.prefix.end());
            offs.push_back((int32_t)ids.size());
        }
        ling.assign((size_t)beams.size() * P.search_depth, 0);
        const int rc = P.next_cb(P.user, (int)beams.size(), ids.data(), offs.data(), P.search_depth, ling.data());
        if (rc != 0) return rc;
    }
    // Step 2: extend (:229-265). gen keeps first-touch order; index maps prefix -> slot.
    std::vector<Hyp> gen;
    std::map<std::vector<int32_t>, int> index;
    auto slot = [&](const std::vector<int32_t>& pre) -> int {
        auto it = index.find(pre);
        if (it != index.end()) return it->second;
        gen.push_back(Hyp{pre, kNegInf, kNegInf, 0.0});
        index.emplace(pre, (int)gen.size() - 1);
        return (int)gen.size() - 1;
    };
    std::vector<int32_t> ext;
    for (size_t bi = 0; bi < beams.size(); ++bi) {
        const Hyp& h = beams[bi];
        const double hprob = h.prob();
        const int nl = (P.next_cb && !h.prefix.empty()) ? P.search_depth : 0;
        for (int ci = 0; ci < ncand + nl; ++ci) {
            int idx;
            double p;
            if (ci < ncand) {
                idx = cands[ci];
                p = (double)plog[ci];
            } else {
                idx = ling[bi * P.search_depth + (ci - ncand)];
                if (idx < 0 || idx >= in.C) return HCTR_ERR_ARG;
                if (idx >= unk) continue;
                p = (double)in.full_logp[((size_t)t * in.B + in.b) * in.C + idx];
            }
            if (idx >= unk) continue;                                  // ignore <unknown> (:238-239)
            const int ps = slot(h.prefix);
            if (idx == 0) {                                            // blank: only pb (:246-249)
                gen[ps].pb = logaddexp(gen[ps].pb, hprob + p);
                continue;
            }
            const int tail = h.prefix.empty() ? -1 : h.prefix.back();
            ext = h.prefix;
            ext.push_back(idx);
            const int es = slot(ext);
            if (idx != tail) {
                gen[es].pnb = logaddexp(gen[es].pnb, hprob + p);        // (:256-258)
            } else {
                gen[es].pnb = logaddexp(gen[es].pnb, h.pb + p);         // not merged (:260-262)
                gen[ps].pnb = logaddexp(gen[ps].pnb, h.pnb + p);        // merged     (:263-265)
            }
        }
    }
    // Step 3: LM score + length bonus, stable sort, cut (:267-285)
    if (!gen.empty()) {
        std::vector<double> scores(gen.size(), 0.0);
        if (P.builtin_lm == 2) {
            std::vector<int32_t> sent;
            for (size_t i = 0; i < gen.size(); ++i) {
                sent = gen[i].prefix;
                sent.insert(sent.end(), suffix.begin(), suffix.end());
                scores[i] = toy_bigram(sent.data(), (int)sent.size(), P.label_codepoints);
            }
        } else if (P.builtin_lm == 0) {
            std::vector<int32_t> ids, offs(1, 0);
            for (const Hyp& g : gen) {
                ids.insert(ids.end(), g.prefix.begin(), g.prefix.end());
                ids.insert(ids.end(), suffix.begin(), suffix.end());
                offs.push_back((int32_t)ids.size());
            }
            const int rc = P.score_cb(P.user, (int)gen.size(), ids.data(), offs.data(), scores.data());
            if (rc != 0) return rc;
        }
        for (size_t i = 0; i < gen.size(); ++i)
            gen[i].pt = scores[i] * P.lm_panelty + (double)gen[i].prefix.size() * P.len_bonus;
    }
    std::vector<double> tot(gen.size());
    std::vector<int> order(gen.size());
    for (size_t i = 0; i < gen.size(); ++i) { tot[i] = gen[i].total(); order[i] = (int)i; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tot[a] > tot[b]; });
    const size_t keep = std::min<size_t>(order.size(), (size_t)std::max(P.beam_size, 0));
    std::vector<Hyp> out;
    out.reserve(keep);
    for (size_t i = 0; i < keep; ++i) out.push_back(std::move(gen[order[i]]));
    beams.swap(out);
    return HCTR_OK;
}

int decode_line(const hctr_beam_params& P, const LineInput& in, int32_t* out_labels, int32_t* out_len) {
    const int W = in.W, B = in.B, C = in.C, k = in.k, b = in.b;
    const int unk = C - 1;
    *out_len = 0;
    // greedy pass with time stamps (:133-140, :188-195)
    std::vector<int32_t> line_lab, line_t;
    int prev = -1;
    for (int t = 0; t < W; ++t) {
        const int c1 = in.topk_idx[((size_t)t * B + b) * k];
        if (c1 != 0 && c1 != unk && !(t > 0 && prev == c1)) { line_lab.push_back(c1); line_t.push_back(t); }
        prev = c1;
    }
    if (line_lab.empty()) return HCTR_ERR_EMPTY_LINE;              // top_line[-1] -> IndexError (:143,198)
    int end_step = line_t.back() + 4;
    if (end_step >= W) end_step = W;
    std::vector<Hyp> beams(1, fresh_hyp());
    std::vector<int32_t> suffix;
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
            const int rc = beam_step(P, in, t, beams, in.topk_idx + r * k, in.topk_logp + r * k, depth, suffix);
            if (rc != HCTR_OK) return rc;
            continue;
        }
        const int64_t c0 = in.cand_off[r], n = in.cand_off[r + 1] - c0;
        if (n != 1) {
            make_suffix();
            const int rc = beam_step(P, in, t, beams, in.cand_idx + c0, in.cand_logp + c0, (int)n, suffix);
            if (rc != HCTR_OK) return rc;
            continue;
        }
        // exactly one class above the prune threshold: in-place update, no LM (:147-171)
        const int c = in.cand_idx[c0];
        if (c >= unk) continue;
        const double pc = (double)in.cand_logp[c0];
        const double p0 = (double)in.blank_logp[r];
        for (Hyp& h : beams) {
            const int tail = h.prefix.empty() ? -1 : h.prefix.back();
            if (c == 0) {
                h.pb = h.prob() + pc;                                 // pc == row[0] here
            } else if (c != tail) {
                const double pr = h.prob();
                h.prefix.push_back(c);
                h.pnb = pr + pc;
                h.pb = kNegInf;
            } else if (h.pb != kNegInf) {
                h.prefix.push_back(c);
                h.pnb = h.pb + pc;
                h.pb = kNegInf;
            } else {
                h.pb = h.prob() + p0;
                h.pnb = h.pnb + pc;
            }
        }
    }
    if (beams.empty()) return HCTR_ERR_EMPTY_LINE;                  // kept_beams[0] -> IndexError (:179,208)
    const std::vector<int32_t>& best = beams[0].prefix;
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
    if (p->builtin_lm < 0 || p->builtin_lm > 2) return HCTR_ERR_ARG;
    if (p->next_cb && !full_logp_wbc) return HCTR_ERR_ARG;
    if (p->search_depth < 1 || p->beam_size < 0) return HCTR_ERR_ARG;
    const bool callbacks = p->builtin_lm == 0 || p->next_cb != nullptr;
    int nthreads = callbacks ? 1 : std::max(1, p->num_threads);
    nthreads = std::min(nthreads, B);
    std::atomic<int> next(0);
    auto worker = [&]() {
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            LineInput in{W, B, C, k, b, topk_idx, topk_logp, blank_logp, cand_off, cand_idx, cand_logp, full_logp_wbc};
            line_status[b] = decode_line(*p, in, out_labels + (size_t)b * W, out_lengths + b);
        }
    };
    if (nthreads == 1) {
        worker();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nthreads; ++i) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    for (int b = 0; b < B; ++b)
        if (line_status[b] != HCTR_OK) return line_status[b];
    return HCTR_OK;
}
