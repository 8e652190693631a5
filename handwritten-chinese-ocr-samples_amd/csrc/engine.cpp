// libhctr_hip.so - host side of the MI355X hctr engine: context, checkpoint ingest (BatchNorm
// folding + repack to kernel layouts), workspace, forward orchestration, and the C ABI declared in
// include/hctr_hip.h. Reference citations are relative to the reference repository root.
#include "../../include/hctr_hip.h"
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace hctr;

namespace {

constexpr int kImgH = 128;                       // hctr_model.img_height, models/handwritten_ctr_model.py:159
constexpr int kFeat = 2048;                      // 512 channels x 4 rows, :168-169
constexpr double kBnEps = 1e-5;                  // nn.BatchNorm2d default
constexpr int kStagePlanes[4] = {128, 256, 512, 512};
constexpr int kStageBlocks[4] = {2, 4, 5, 1};    // ResNet(1, 512, BasicBlock, [2,4,5,1]), :166
constexpr int kStageH[5] = {128, 64, 32, 16, 8}; // rows entering stage s (stage 0 = stem)
constexpr int64_t kDefaultMaxCols = 131072;      // pixel columns (B*W) per internal pass

std::string g_create_error;

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    mutable bool used = false;
};

struct ConvW {
    half_t* w = nullptr;    // [taps][coutPad][cin] fp16, MFMA row order
    float* bias = nullptr;  // [coutPad]
    int cin = 0, cout = 0, coutPad = 0, taps = 0;
};

struct SeW {
    float* w1 = nullptr;    // [c/16][c]
    float* w2 = nullptr;    // [c][c/16]
    int c = 0;
};

struct BlockW {
    ConvW conv1, conv2, ds;
    SeW se;
    bool has_ds = false;
};

struct Workspace {
    int B = 0, W = 0, Wa = 0;
    void* img = nullptr;            // staged input (u8 or f32)
    int32_t* widths = nullptr;
    half_t* s0 = nullptr;           // conv0_1 output [B][130][Wa][64]
    half_t* x[5] = {};              // x[s]: input of stage s (s = 1..4), padded NHWC
    half_t* p[5][3] = {};           // rotating block buffers of stage s
    half_t* headin = nullptr;       // [B*W][2048], k = h*512 + c
    float* logits = nullptr;        // [B*W][Cpad]
    float* se_part = nullptr;
    float* se_scale = nullptr;
    float* se_border = nullptr;     // [B][4][8 segments][512]
    float* se_mean = nullptr;       // [B][512]
    int32_t* se_counter = nullptr;  // [B] last-block-done tickets of se_premean (zero between launches)
    int32_t* colidx = nullptr;      // [B*W]
    float* amax_val = nullptr;      // [P][B*W] fused head argmax partials, P <= Cpad/64
    int32_t* amax_idx = nullptr;
    int32_t* labels = nullptr;      // [B][W]
    int32_t* lengths = nullptr;     // [B]
    // fused beam front end (WS_BEAM, carved on first use): see ConvArgs in kernels.h
    float* psum = nullptr;          // [P][B*W]
    float* blank_logit = nullptr;   // [B*W]
    float* row_thr = nullptr;       // [B*W][2]
    int32_t* emit_cnt = nullptr;    // [B*W]
    int32_t* emit_list = nullptr;   // [B*W][kBeamCap][2]
    double* esum = nullptr;         // [P][B*W]
    int32_t* overflow = nullptr;    // [1]
    int32_t* bm_idx = nullptr;      // [B*W][kBeamMaxK] outputs of the fused front end (rows r = t*B + b), before the D2H
    float* bm_lp = nullptr;         // [B*W][kBeamMaxK]
    float* bm_bl = nullptr;         // [B*W]
    float* bm_st = nullptr;         // [B*W][2]
    int32_t* bm_cnt = nullptr;      // [B*W]
    int32_t* tile_ctrs = nullptr;   // [64 launches][8 XCDs] tile queues of the persistent conv variant (HCTR_PERSIST=2)
    // guarded precision (WS_GUARD): runner-up / |logit| partials of the fused head, per-column and per-line figures
    float* amax_val2 = nullptr;     // [P][B*W]
    float* amax_abs = nullptr;      // [P][B*W]
    float* col_margin = nullptr;    // [B*W] top-1 minus top-2 logit
    float* col_abs = nullptr;       // [B*W] max |logit|
    float* line_guard = nullptr;    // [B][2] = {min margin, max |logit|} per line
    int features = 0;               // WS_* sets carved into this layout
    bool split = false;             // layout of the f16x3 planes (3x the activation channels)
    bool aliased = false;           // activation buffers shared between the stages (see ensure_workspace)
};

// optional parts of a workspace layout (carved behind the core buffers, so adding one moves nothing)
enum WsFeature { WS_S0 = 1, WS_LOGITS = 2, WS_BEAM = 4, WS_GUARD = 8 };

// one resident set of device weights in kernel layouts; set 0 = f16, set 1 = f16x3 ([w_hi | w_hi | w_lo] rows)
struct WeightSet {
    float* stem_w = nullptr;
    float* stem_b = nullptr;
    ConvW conv0_2;
    std::vector<BlockW> blocks[4];
    ConvW stage_conv[4];
    ConvW head;
    bool built = false;
};

struct ProfEntry {
    std::string name;
    hipEvent_t e0, e1;
};

}  // namespace

struct hctr_ctx {
    int device = 0;
    int num_classes = 0;
    int cpad = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    // device weights: the sets the precision mode chosen before hctr_finalize_weights needs (mode 2 keeps both)
    WeightSet wset[2];
    const WeightSet& wts() const { return wset[split ? 1 : 0]; }
    std::vector<void*> wallocs;
    // ONE device arena holds the workspace of whatever (lines, width) shape is active: a new shape re-carves the
    // pointers and re-zeroes the stored conv borders (a small kernel), it does not allocate - ragged workloads present a
    // new padded width with almost every batch. The arena grows to the largest layout seen.
    Workspace ws;
    char* arena = nullptr;
    size_t arena_cap = 0;
    int ws_sticky = 0;                  // optional parts this context has needed so far (kept in later layouts)
    int64_t arena_reallocs = 0, ws_recarves = 0;
    bool rpre = true;                   // HCTR_RPRE=0/1 with a -DRPRE=1 build: conv2's residual fetched during the last K step (A/B, neutral)
    bool rtouch = false;                // HCTR_RTOUCH=1: residual pre-touch in conv2's K loop (A/B; measured neutral)
    int ws_alias = -1;                  // HCTR_WS_ALIAS: 1 = stages share four activation buffers, 0 = never, -1 (default) = when
                                        // the dedicated layout would exceed ws_dedicated_max bytes or the device's free memory
    size_t ws_dedicated_max = (size_t)16 << 30;      // HCTR_WS_DEDICATED_MAX_GB
    int64_t max_cols = kDefaultMaxCols;
    bool big_tiles = true;
    int halo_mode = 2;
    // f16x3 precision mode: every activation and weight is carried as hi + lo fp16 pairs; a conv sees
    // tripled input channels [x_hi | x_lo | x_hi] against weight rows [w_hi | w_hi | w_lo], so the MFMA
    // main loops are unchanged and the products w_hi*x_hi + w_hi*x_lo + w_lo*x_hi are summed in fp32.
    // `split` is the arithmetic of the pass being run; `mode` is what the caller asked for: 0 = f16, 1 = f16x3,
    // 2 = guarded ("auto"): every line runs in f16, the fused head also yields each column's top-1/top-2 logit margin,
    // and the lines with a column whose margin is within twice the f16 logit tolerance (guard_rel * max|logit of the
    // line| + guard_abs - the tolerance the parity suite asserts) are run again in f16x3 at the same padded width.
    int mode = 0;
    bool split = false;
    // HCTR_X3_MASK (diagnostic, tests/diag_precision_attribution.py): which classes of rounding points the f16x3 mode
    // carries as hi + lo; a cleared bit rounds that class to one fp16 value like the f16 mode. 1 = conv weights,
    // 2 = conv1 outputs of the blocks, 4 = block outputs (the residual stream), 8 = stem and stage-conv outputs,
    // 16 = head input and head weights. Default 31 = the f16x3 mode proper.
    int x3_mask = 31;
    int out_class = 8;               // rounding-point class of the conv being launched (set by run_block / run_forward)
    double guard_rel = 0.01, guard_abs = 0.05;
    // guard figures of the last call in mode 2 (per line of that call's batch)
    std::vector<float> g_margin, g_scale;
    std::vector<uint8_t> g_flag;
    int64_t g_flagged_total = 0, g_lines_total = 0;      // running totals over the context's lifetime
    std::vector<std::vector<int32_t>> h_widths;          // gathered widths of this call's passes (source of async copies:
                                                         // kept until the next call; every call drains the stream first)
    std::vector<int32_t> h_labels, h_lengths;            // labels of a re-run pass before they are scattered
    char* pin = nullptr;                                 // pinned host staging of the beam front end's D2H copies
    size_t pin_cap = 0;
    int chm() const { return split ? 3 : 1; }      // channel multiplier of activation buffers
    bool fuse_se = true;
    std::string stamp_layer;         // hctr_debug_stamps: layer whose workgroups are time-stamped (diagnostic)
    unsigned long long* stamp_buf = nullptr;
    int64_t stamp_cap = 0, stamp_n = 0;
    bool fuse_ds = true;             // 1x1 downsample inside conv2's K loop (HCTR_FUSE_DS=0: own launch + residual)
    bool fuse_argmax = true;         // greedy: argmax in the head GEMM's epilogue (HCTR_FUSE_ARGMAX=0: separate pass)
    bool persist_dynamic = false;    // HCTR_PERSIST=2: persistent conv workgroups drawing tiles from an atomic queue
    int conv_seq = 0;                // conv launches of the current forward (one queue each)
    bool fuse_stem = true;           // conv0_1 inside conv0_2's loader (HCTR_FUSE_STEM=0: own launch + 16 kB/column buffer)
    bool fuse_beam = true;           // beam front end without stored logits (HCTR_FUSE_BEAM=0: logits + row_topk)
    int64_t beam_fallbacks = 0;      // passes that overflowed a row list and were redone through the logits
    // profiling
    bool profiling = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // beam front-end scratch + the candidate lists of the last hctr_beam_frontend call
    std::vector<void*> beam_allocs;
    std::vector<int64_t> cand_off;
    std::vector<int32_t> cand_idx;
    std::vector<float> cand_logp;
};

namespace {

int fail(hctr_ctx* c, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

// The ABI promises that no C++ exception crosses it (include/hctr_hip.h): every extern "C" body runs inside
// guard(), which maps std::bad_alloc to HCTR_ERR_NOMEM and anything else to HCTR_ERR_STATE. Setting the message
// may itself allocate, so that is attempted under its own handler.
int fail_nothrow(hctr_ctx* c, int code, const char* what) noexcept {
    try {
        return fail(c, code, "%s", what);
    } catch (...) {
        return code;
    }
}

template <typename R = int, typename F>
R guard(hctr_ctx* c, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return (R)fail_nothrow(c, HCTR_ERR_NOMEM, "out of host memory (std::bad_alloc)");
    } catch (const std::exception& e) {
        return (R)fail_nothrow(c, HCTR_ERR_STATE, e.what());
    } catch (...) {
        return (R)fail_nothrow(c, HCTR_ERR_STATE, "unknown C++ exception");
    }
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(ctx, HCTR_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,          \
                        hipGetErrorString(_e));                                                   \
    } while (0)

#define TRY(expr)                   \
    do {                            \
        int _r = (expr);            \
        if (_r != HCTR_OK) return _r; \
    } while (0)

template <typename T>
int dev_alloc(hctr_ctx* c, std::vector<void*>& pool, T** out, size_t count, bool zero, size_t* acct = nullptr) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return fail(c, HCTR_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    pool.push_back(p);
    if (acct) *acct += bytes;
    if (zero) HIP_TRY(c, hipMemsetAsync(p, 0, bytes, c->stream));
    *out = (T*)p;
    return HCTR_OK;
}

void free_pool(std::vector<void*>& pool) {
    for (void* p : pool) (void)hipFree(p);
    pool.clear();
}

struct PoolGuard {          // frees a call's temporaries on every exit path, exceptions included
    std::vector<void*>& pool;
    ~PoolGuard() { free_pool(pool); }
};

// ---------------------------------------------------------------------------------------------
// checkpoint ingest
// ---------------------------------------------------------------------------------------------
const HostTensor* find(hctr_ctx* c, const std::string& key) {
    auto it = c->host.find(key);
    return it == c->host.end() ? nullptr : &it->second;
}

int need(hctr_ctx* c, const std::string& key, std::vector<int64_t> shape, const HostTensor** out) {
    const HostTensor* t = find(c, key);
    if (!t) return fail(c, HCTR_ERR_KEY, "Missing key(s) in state_dict: \"%s\"", key.c_str());
    if (t->shape != shape) {
        std::string got, want;
        for (auto v : t->shape) got += std::to_string(v) + ",";
        for (auto v : shape) want += std::to_string(v) + ",";
        return fail(c, HCTR_ERR_SHAPE, "size mismatch for %s: checkpoint [%s] vs model [%s]", key.c_str(),
                    got.c_str(), want.c_str());
    }
    t->used = true;
    *out = t;
    return HCTR_OK;
}

// stored row s of a 64-row block holds cout perm64(s): MFMA tile j = s/16, row ra = s%16 (lane group
// q = ra>>2, element i = ra&3) of that tile is cout (j>>1)*32 + q*8 + (j&1)*4 + i, so an accumulator lane
// owns two runs of 8 consecutive couts and the four lanes of a pixel store 64 contiguous bytes at a time.
inline int perm64(int s) {
    const int j = s >> 4, ra = s & 15;
    return (j >> 1) * 32 + (ra >> 2) * 8 + (j & 1) * 4 + (ra & 3);
}

// Conv2d + eval BatchNorm2d folded: w' = w * g/sqrt(v+eps), b' = (b - mean) * g/sqrt(v+eps) + beta
// (models/handwritten_ctr_model.py:37-40, 73-92, 104-108). Folded in float64, stored fp16 / fp32.
int build_conv(hctr_ctx* c, const std::string& ck, const std::string& bk, int cin, int cout, int ks,
               bool has_bias, int pad_to, ConvW* out) {
    const HostTensor *w, *b = nullptr, *g, *beta, *mean, *var;
    TRY(need(c, ck + ".weight", {cout, cin, ks, ks}, &w));
    if (has_bias) TRY(need(c, ck + ".bias", {cout}, &b));
    TRY(need(c, bk + ".weight", {cout}, &g));
    TRY(need(c, bk + ".bias", {cout}, &beta));
    TRY(need(c, bk + ".running_mean", {cout}, &mean));
    TRY(need(c, bk + ".running_var", {cout}, &var));
    const int taps = ks * ks;
    const int coutPad = (cout + pad_to - 1) / pad_to * pad_to;
    const int cinp = c->chm() * cin;               // stored row length ([w_hi | w_hi | w_lo] when split)
    std::vector<half_t> hw((size_t)taps * coutPad * cinp, (half_t)0.f);
    std::vector<float> hb(coutPad, 0.f);
    double wmax = 0.0;
    for (int blk = 0; blk < coutPad / 64; ++blk)
        for (int s = 0; s < 64; ++s) {
            const int co = blk * 64 + perm64(s);
            if (co >= cout) continue;
            const double sc = (double)g->data[co] / std::sqrt((double)var->data[co] + kBnEps);
            for (int t = 0; t < taps; ++t)
                for (int ci = 0; ci < cin; ++ci) {
                    const double v = (double)w->data[((size_t)co * cin + ci) * taps + t] * sc;
                    wmax = std::max(wmax, std::fabs(v));
                    half_t* row = &hw[((size_t)t * coutPad + blk * 64 + s) * cinp];
                    const half_t hi = (half_t)(float)v;
                    row[ci] = hi;
                    if (c->split) {
                        row[cin + ci] = hi;
                        row[2 * cin + ci] = (c->x3_mask & 1) ? (half_t)(float)(v - (double)(float)hi) : (half_t)0.f;
                    }
                }
        }
    for (int co = 0; co < cout; ++co) {
        const double sc = (double)g->data[co] / std::sqrt((double)var->data[co] + kBnEps);
        const double bb = has_bias ? (double)b->data[co] : 0.0;
        hb[co] = (float)((bb - (double)mean->data[co]) * sc + (double)beta->data[co]);
    }
    if (!(wmax < 65000.0))      // also catches NaN; fp16 storage cannot hold it (degenerate running_var?)
        return fail(c, HCTR_ERR_ARG, "%s: BatchNorm-folded weight magnitude %.3g does not fit fp16", ck.c_str(), wmax);
    out->cin = cinp; out->cout = cout; out->coutPad = coutPad; out->taps = taps;
    TRY(dev_alloc(c, c->wallocs, &out->w, hw.size(), false));
    TRY(dev_alloc(c, c->wallocs, &out->bias, hb.size(), false));
    HIP_TRY(c, hipMemcpyAsync(out->w, hw.data(), hw.size() * sizeof(half_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out->bias, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // host vectors die at scope exit
    return HCTR_OK;
}

int build_se(hctr_ctx* c, const std::string& key, int ch, SeW* out) {
    const HostTensor *w1, *w2;
    TRY(need(c, key + ".fc.0.weight", {ch / 16, ch}, &w1));
    TRY(need(c, key + ".fc.2.weight", {ch, ch / 16}, &w2));
    out->c = ch;
    TRY(dev_alloc(c, c->wallocs, &out->w1, w1->data.size(), false));
    TRY(dev_alloc(c, c->wallocs, &out->w2, w2->data.size(), false));
    HIP_TRY(c, hipMemcpyAsync(out->w1, w1->data.data(), w1->data.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out->w2, w2->data.data(), w2->data.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HCTR_OK;
}

int build_stem(hctr_ctx* c, WeightSet& ws) {
    const HostTensor *w, *b, *g, *beta, *mean, *var;
    TRY(need(c, "cnn.conv0_1.weight", {64, 1, 3, 3}, &w));
    TRY(need(c, "cnn.conv0_1.bias", {64}, &b));
    TRY(need(c, "cnn.bn0_1.weight", {64}, &g));
    TRY(need(c, "cnn.bn0_1.bias", {64}, &beta));
    TRY(need(c, "cnn.bn0_1.running_mean", {64}, &mean));
    TRY(need(c, "cnn.bn0_1.running_var", {64}, &var));
    std::vector<float> hw(64 * 9), hb(64);
    for (int co = 0; co < 64; ++co) {
        const double sc = (double)g->data[co] / std::sqrt((double)var->data[co] + kBnEps);
        for (int t = 0; t < 9; ++t) hw[co * 9 + t] = (float)((double)w->data[co * 9 + t] * sc);
        hb[co] = (float)(((double)b->data[co] - (double)mean->data[co]) * sc + (double)beta->data[co]);
    }
    TRY(dev_alloc(c, c->wallocs, &ws.stem_w, hw.size(), false));
    TRY(dev_alloc(c, c->wallocs, &ws.stem_b, hb.size(), false));
    HIP_TRY(c, hipMemcpyAsync(ws.stem_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ws.stem_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HCTR_OK;
}

// self.linear (models/handwritten_ctr_model.py:169,175). The reference's feature index is
// d = c*4 + h (flatten(1,2) of [B,C,H,W], :173); the engine's head input is [pixel][h*512 + c].
int build_head(hctr_ctx* c, WeightSet& ws) {
    const HostTensor *w, *b;
    const int C = c->num_classes;
    TRY(need(c, "linear.weight", {C, kFeat}, &w));
    TRY(need(c, "linear.bias", {C}, &b));
    const int cpad = c->cpad;
    const int kf = c->chm() * kFeat;               // [h][hi | lo-slot | hi-slot][512] when split
    std::vector<half_t> hw((size_t)cpad * kf, (half_t)0.f);
    std::vector<float> hb(cpad, 0.f);
    for (int blk = 0; blk < cpad / 64; ++blk)
        for (int s = 0; s < 64; ++s) {
            const int n = blk * 64 + perm64(s);
            if (n >= C) continue;
            half_t* dst = &hw[(size_t)(blk * 64 + s) * kf];
            const float* src = &w->data[(size_t)n * kFeat];
            for (int ch = 0; ch < 512; ++ch)
                for (int h = 0; h < 4; ++h) {
                    const float v = src[ch * 4 + h];
                    const half_t hi = (half_t)v;
                    if (!c->split) {
                        dst[h * 512 + ch] = hi;
                    } else {                         // feature planes [x_hi | x_lo | x_hi] x [w_hi | w_hi | w_lo]
                        half_t* r = dst + h * 1536;
                        r[ch] = hi;
                        r[512 + ch] = hi;
                        r[1024 + ch] = (c->x3_mask & 16) ? (half_t)(v - (float)hi) : (half_t)0.f;
                    }
                }
        }
    for (int n = 0; n < C; ++n) hb[n] = b->data[n];
    ws.head.cin = kf; ws.head.cout = C; ws.head.coutPad = cpad; ws.head.taps = 1;
    TRY(dev_alloc(c, c->wallocs, &ws.head.w, hw.size(), false));
    TRY(dev_alloc(c, c->wallocs, &ws.head.bias, hb.size(), false));
    HIP_TRY(c, hipMemcpyAsync(ws.head.w, hw.data(), hw.size() * sizeof(half_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ws.head.bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return HCTR_OK;
}

// ---------------------------------------------------------------------------------------------
// workspace: padded NHWC fp16 activations, one set of buffers per stage (borders stay zero)
// ---------------------------------------------------------------------------------------------
inline int64_t act_elems(int B, int H, int Wa, int C) { return (int64_t)B * (H + 2) * Wa * C; }

// class parts per row written by the fused head epilogues: n-tiles x wave columns. The 256x256 head tile gives
// cpad/128 parts, the 128x128 tile (HCTR_BIG_TILES=0, or cpad not a multiple of 256) cpad/64: buffers are sized for
// the larger figure, launches use head_parts(c).
inline int head_parts_max(const hctr_ctx* c) { return c->cpad / 64; }
inline int head_parts(const hctr_ctx* c) {
    const int hbm = (c->big_tiles && c->cpad % 256 == 0) ? 256 : 128;
    return (c->cpad / hbm) * kLinearWN;
}

int ensure_workspace(hctr_ctx* c, int B, int W, int features = 0) {
    c->ws_sticky |= features;
    bool same_shape = c->ws.B == B && c->ws.W == W && c->ws.split == c->split;
    int feat = c->ws_sticky;
    if (c->fuse_stem && !c->split) feat &= ~WS_S0;        // (mode 2 alternates layouts: conv0_1's buffer only where it is used)
    if (same_shape && (c->ws.features & feat) == feat) return HCTR_OK;
    const int tilesW = (W + kTileW - 1) / kTileW;        // 16-column tiles (upper bound on tiles per row)
    const int Wa = (W + 31) / 32 * 32 + 2;               // room for the widest (32-column) tile + border
    const size_t cols = (size_t)B * W;
    const int m = c->chm();
    // ---- layout: byte offsets into the arena, 256-byte aligned; core buffers first, optional parts behind them.
    // Activation buffers, DEDICATED layout (default): every stage owns its input x[s] and three rotating block buffers,
    // so a buffer keeps one geometry and its stored zero border survives from forward to forward (~245 kB per pixel
    // column). ALIASED layout (HCTR_WS_ALIAS=1, or automatically when the dedicated one does not fit): four buffers of
    // the largest stage geometry serve all stages - x[s] is pool 0 for every s (a stage input is dead after the stage's
    // first block, long before conv_s+pool writes the next one), p[s][i] is pool 1 + i (dead once conv_s+pool has read
    // the stage's last output) - ~95 kB per column; a buffer then changes geometry from stage to stage, so run_forward
    // re-zeroes the borders at every stage entry (+15 small launches, ~0.5 % of a step).
    Workspace ws;
    size_t off = 0;
    std::vector<std::pair<void**, size_t>> slots;
    half_t* pool4[4] = {};
    auto layout = [&](bool alias) {
        ws = Workspace();
        ws.B = B; ws.W = W; ws.features = feat; ws.split = c->split; ws.Wa = Wa; ws.aliased = alias;
        off = 0;
        slots.clear();
        auto A = [&](auto** out, size_t count) {
            slots.emplace_back((void**)out, off);
            off += (std::max<size_t>(count * sizeof(**out), 16) + 255) & ~(size_t)255;
        };
        A((char**)&ws.img, cols * kImgH * 4);
        A(&ws.widths, (size_t)B);
        int cin = 64;
        size_t se_max = 0, pool_elems = 0;
        for (int s = 1; s <= 4; ++s) {
            const int H = kStageH[s], planes = kStagePlanes[s - 1];
            if (!alias) {
                A(&ws.x[s], (size_t)act_elems(B, H, Wa, cin * m));
                const int nbuf = (s == 4) ? 2 : 3;
                for (int i = 0; i < nbuf; ++i) A(&ws.p[s][i], (size_t)act_elems(B, H, Wa, planes * m));
            }
            pool_elems = std::max(pool_elems, (size_t)act_elems(B, H, Wa, std::max(cin, planes) * m));
            se_max = std::max(se_max, (size_t)B * (H / 8) * tilesW * planes);
            cin = planes;
        }
        if (alias)
            for (int i = 0; i < 4; ++i) A(&pool4[i], pool_elems);
        A(&ws.headin, cols * kFeat * m);
        A(&ws.se_part, se_max);
        A(&ws.se_scale, (size_t)B * 512);
        A(&ws.se_border, (size_t)B * 5 * 8 * 512);
        A(&ws.se_mean, (size_t)B * 512);
        A(&ws.se_counter, (size_t)B);
        A(&ws.colidx, cols);
        const size_t P = (size_t)head_parts_max(c);           // class parts per row of the fused head epilogues
        A(&ws.amax_val, cols * P);
        A(&ws.amax_idx, cols * P);
        A(&ws.labels, cols);
        A(&ws.lengths, (size_t)B);
        A(&ws.tile_ctrs, (size_t)64 * 8);
        // conv0_1's output (16 kB per column): only the unfused / f16x3 stem path
        if (feat & WS_S0) A(&ws.s0, (size_t)act_elems(B, 128, Wa, 64 * m));
        // [B*W][cpad] fp32 logits (3.8 GB at config 2): hctr_forward_logits and the HCTR_FUSE_* = 0 A/B paths only
        if (feat & WS_LOGITS) A(&ws.logits, cols * c->cpad);
        if (feat & WS_GUARD) {                               // guarded precision: runner-up / |logit| partials and figures
            A(&ws.amax_val2, cols * P);
            A(&ws.amax_abs, cols * P);
            A(&ws.col_margin, cols);
            A(&ws.col_abs, cols);
            A(&ws.line_guard, (size_t)2 * B);
        }
        if (feat & WS_BEAM) {                                // scratch of the fused beam front end (kernels.h ConvArgs)
            A(&ws.psum, P * cols);
            A(&ws.blank_logit, cols);
            A(&ws.row_thr, 2 * cols);
            A(&ws.emit_cnt, cols);
            A(&ws.esum, P * cols);
            A(&ws.overflow, (size_t)1);
            A(&ws.bm_idx, cols * kBeamMaxK);
            A(&ws.bm_lp, cols * kBeamMaxK);
            A(&ws.bm_bl, cols);
            A(&ws.bm_st, 2 * cols);
            A(&ws.bm_cnt, cols);
            A(&ws.emit_list, cols * kBeamCap * 2);
        }
    };
    bool alias = c->ws_alias == 1;
    layout(alias);
    if (c->ws_alias < 0) {
        // automatic: small shapes keep every stage's own buffers (no per-stage border zeroing, every debug tap readable);
        // a layout beyond ws_dedicated_max (16 GiB: about 30 lines of 2000 columns), or one that would not fit into the
        // device's free memory, uses the shared buffers (a third of the bytes, +0.3 % time)
        bool want = off > c->ws_dedicated_max;
        if (!want && off > c->arena_cap) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && off > free_b + c->arena_cap) want = true;
        }
        if (want) {
            alias = true;
            layout(true);
        }
    }
    bool fresh = false;
    if (off > c->arena_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->arena) (void)hipFree(c->arena);
        c->arena = nullptr; c->arena_cap = 0;
        c->ws = Workspace();
        const size_t want = off + off / 16;              // a little headroom: the next slightly wider batch fits too
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) e = hipMalloc(&p, off);
        if (e != hipSuccess)
            return fail(c, HCTR_ERR_NOMEM, "hipMalloc(%zu bytes) for the workspace of %d lines x %d columns failed: %s", off,
                        B, W, hipGetErrorString(e));
        c->arena = (char*)p;
        c->arena_cap = e == hipSuccess && want >= off ? want : off;
        ++c->arena_reallocs;
        fresh = true;
    }
    for (auto& sl : slots) *sl.first = c->arena + sl.second;
    if (ws.aliased) {
        for (int s = 1; s <= 4; ++s) {
            ws.x[s] = pool4[0];
            for (int i = 0; i < ((s == 4) ? 2 : 3); ++i) ws.p[s][i] = pool4[1 + i];
        }
    }
    if (c->ws.aliased != ws.aliased) same_shape = false;
    // ---- stored conv borders: every activation buffer's border rows / columns must read as zero. The interiors are
    //      rewritten by each forward, so after a shape change (or a new arena) only the borders are cleared. ----
    auto zero_act = [&](half_t* p, int H, int C) { return launch_zero_borders(p, B, H, W, Wa, C * m, c->stream); };
    if (fresh || !same_shape) {
        int ci = 64;
        for (int s = 1; s <= 4 && !ws.aliased; ++s) {        // (aliased layout: run_forward zeroes at every stage entry)
            const int H = kStageH[s], planes = kStagePlanes[s - 1];
            HIP_TRY(c, zero_act(ws.x[s], H, ci));
            const int nbuf = (s == 4) ? 2 : 3;
            for (int i = 0; i < nbuf; ++i) HIP_TRY(c, zero_act(ws.p[s][i], H, planes));
            ci = planes;
        }
        HIP_TRY(c, hipMemsetAsync(ws.se_counter, 0, (size_t)B * 4, c->stream));
        ++c->ws_recarves;
    }
    if (ws.s0 && (fresh || !same_shape || !c->ws.s0)) HIP_TRY(c, zero_act(ws.s0, 128, 64));
    c->ws = ws;
    return HCTR_OK;
}

// ---------------------------------------------------------------------------------------------
// forward orchestration
// ---------------------------------------------------------------------------------------------
struct Prof {
    hctr_ctx* c;
    bool on;
    explicit Prof(hctr_ctx* ctx) : c(ctx), on(ctx->profiling) {}
    hipEvent_t ev() {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            c->ev_pool.push_back(e);
        }
        return c->ev_pool[c->ev_used++];
    }
    void begin(const char* name) {
        if (!on) return;
        ProfEntry pe{name, ev(), ev()};
        (void)hipEventRecord(pe.e0, c->stream);
        c->prof.push_back(pe);
    }
    void end() {
        if (!on) return;
        (void)hipEventRecord(c->prof.back().e1, c->stream);
    }
};

// start of an API call: forget the previous call's per-layer events (a call may run several internal passes,
// whose entries of the same name hctr_last_profile adds up)
inline void prof_reset(hctr_ctx* c) {
    if (c->profiling) { c->prof.clear(); c->ev_used = 0; }
    c->h_widths.clear();                              // (the previous call drained the stream before it returned)
}

struct ActDesc {            // a padded NHWC activation
    half_t* p;
    int H, C;
};

// Block tile per layer: 256x256 (8 waves) halves the bytes staged per MFMA, so it is used wherever
// the layer shape allows (Cout % 256 == 0, H % 16 == 0: stages 2 and 3 = 82 % of the FLOPs).
ConvTile pick_tile(const hctr_ctx* c, const ConvW& cw, int H) {
    if (cw.cout == 64) return TILE_64x256;
    // 3x3 layers with Cout % 128 == 0 and H % 16 == 0 (stages 1-3): two 4-wave halo workgroups per CU
    if (c->halo_mode == 2 && cw.taps == 9 && cw.coutPad % 128 == 0 && H % 16 == 0) return TILE_HALO4;
    if (c->halo_mode == 2 && cw.taps == 9 && cw.coutPad % 128 == 0 && H % 8 == 0) return TILE_HALO4_8x32;   // stage 4
    if (c->big_tiles && cw.coutPad % 256 == 0 && H % 16 == 0) return TILE_256x256;
    return TILE_128x128;
}

int run_conv(hctr_ctx* c, Prof& pf, const char* name, const ConvW& cw, ActDesc in, half_t* out, int outH,
             bool relu, bool pool, float* se_part, bool to_head, const float* se_scale = nullptr,
             const half_t* resid = nullptr, const ConvW* ds = nullptr, const half_t* ds_in = nullptr,
             int stem_img_f32 = -1, bool stem_widths = false) {
    const Workspace& ws = c->ws;
    ConvArgs a{};
    a.x = in.p; a.w = cw.w; a.bias = cw.bias; a.y = out; a.se_part = se_part;
    a.se_scale = se_scale; a.resid = resid;
    a.rtouch = c->rtouch ? 1 : 0;
    a.rpre = c->rpre ? 1 : 0;
    a.split = c->split ? 1 : 0;
    a.drop_lo = (c->split && !(c->x3_mask & c->out_class)) ? 1 : 0;
    const int m = c->chm();                        // cw.cin already counts the tripled input channels
    a.H = in.H; a.W = ws.W; a.Cin = cw.cin; a.Cout = cw.cout; a.CoutPad = cw.coutPad;
    const ConvTile tile = pick_tile(c, cw, in.H);
    const int rows = conv_tile_rows(tile);
    if (in.H % rows != 0 || cw.cin % kBK != 0)
        return fail(c, HCTR_ERR_ARG, "conv %s: H=%d or Cin=%d not tileable", name, in.H, cw.cin);
    const int cols = conv_tile_cols(tile);
    a.tilesW = (ws.W + cols - 1) / cols;
    a.tilesH = in.H / rows;
    a.in_sb = (int64_t)(in.H + 2) * ws.Wa * cw.cin;
    a.in_sh = ws.Wa * cw.cin;
    if (a.in_sb * 2 >= ((int64_t)1 << 32) || (int64_t)(outH + 2) * ws.Wa * cw.cout * m * 2 >= ((int64_t)1 << 32))
        return fail(c, HCTR_ERR_ARG, "line width %d too large: one image's activation exceeds the kernels' 32-bit "
                    "in-image byte offsets (conv %s)", ws.W, name);
    if (to_head) {            // conv4 + pool -> head input [B][W][4][512] (x3 planes when split)
        a.out_sb = (int64_t)ws.W * kFeat * m; a.out_sh = 512 * m; a.out_sw = kFeat * m; a.out_off = 0;
        a.out_wlimit = ws.W;
    } else {
        a.out_sb = (int64_t)(outH + 2) * ws.Wa * cw.cout * m;
        a.out_sh = ws.Wa * cw.cout * m; a.out_sw = cw.cout * m;
        a.out_off = (int64_t)(ws.Wa + 1) * cw.cout * m;
        a.out_wlimit = a.tilesW * cols;
    }
    a.relu = relu; a.pool = pool;
    if (ds) {                 // fused 1x1 downsample of the block input (same H, W, Wa; ds->cin channels)
        if (tile != TILE_HALO4 || !se_scale || resid || ds->cin % kBK != 0 || ds->coutPad != cw.coutPad)
            return fail(c, HCTR_ERR_STATE, "conv %s: downsample fusion not applicable", name);
        a.ds_x = ds_in; a.ds_w = ds->w; a.ds_bias = ds->bias; a.ds_cin = ds->cin;
        a.ds_in_sh = ws.Wa * ds->cin;
        a.ds_in_sb = (int64_t)(in.H + 2) * ws.Wa * ds->cin;
    }
    a.mtiles = ws.B * a.tilesH * a.tilesW;
    a.ntiles = cw.coutPad / conv_tile_couts(tile);
    if (c->persist_dynamic && c->conv_seq < 64) a.tile_counter = ws.tile_ctrs + 8 * c->conv_seq++;
    if (c->stamp_buf && c->stamp_layer == name && (tile == TILE_HALO4 || tile == TILE_HALO4_8x32) &&
        (int64_t)a.mtiles * a.ntiles <= c->stamp_cap) {
        a.stamps = c->stamp_buf;
        c->stamp_n = (int64_t)a.mtiles * a.ntiles;
    }
    static const char* dbg_layer = getenv("HCTR_DBG_LAYER");          // timing experiments on ONE layer (kernels.hip env_dbg)
    if (dbg_layer && strcmp(dbg_layer, name) == 0)
        if (const char* e = getenv("HCTR_DBG")) a.dbg = atoi(e);
    pf.begin(name);
    if (stem_img_f32 >= 0) {      // conv0_2 with conv0_1 computed in its loader from the staged image (kernels.hip)
        if (tile != TILE_64x256 || cw.taps != 9 || cw.cin != 64 || cw.coutPad != 64 || c->split)
            return fail(c, HCTR_ERR_STATE, "conv %s: stem fusion not applicable", name);
        a.img = ws.img; a.img_f32 = stem_img_f32; a.img_widths = stem_widths ? ws.widths : nullptr;
        a.stem_w = c->wts().stem_w; a.stem_b = c->wts().stem_b;
        HIP_TRY(c, launch_stem_conv0_2(a, c->stream));
    } else {
        HIP_TRY(c, launch_conv(a, tile, cw.taps, false, c->stream));
    }
    pf.end();
    return HCTR_OK;
}

// BasicBlock.forward (models/handwritten_ctr_model.py:47-60). Fused form (default): the SE scale is
// computed from statistics of conv1's output before conv2 runs (kernels.hip, se_premean), and conv2's
// epilogue applies relu(acc * scale + residual) itself. HCTR_FUSE_SE=0 selects the unfused form
// (conv2 -> o, SE on sums of o, separate se_apply pass) for A/B runs.
int run_block(hctr_ctx* c, Prof& pf, const std::string& name, const BlockW& bw, ActDesc in, half_t* t,
              half_t* o, half_t* r, int planes) {
    const Workspace& ws = c->ws;
    const int H = in.H;
    auto tiles_of = [&](const ConvW& cw) {
        const ConvTile t = pick_tile(c, cw, H);
        return (H / conv_tile_rows(t)) * ((ws.W + conv_tile_cols(t) - 1) / conv_tile_cols(t));
    };
    const float inv_hw = 1.0f / ((float)H * (float)ws.W);
    const half_t* res = in.p;
    // first block of stages 1-3: the 1x1 downsample branch runs inside conv2's K loop (kernels.hip DSFUSE) when
    // conv2 is on the default halo kernel; otherwise (A/B paths, f16x3) as its own launch writing the residual r
    const bool fuse_ds = bw.has_ds && c->fuse_ds && (c->fuse_se || c->split) && bw.ds.cin % kBK == 0 &&
                         pick_tile(c, bw.conv2, H) == TILE_HALO4;
    if (bw.has_ds && !fuse_ds) {
        c->out_class = 4;
        TRY(run_conv(c, pf, (name + ".downsample").c_str(), bw.ds, in, r, H, false, false, nullptr, false));
        res = r;
    }
    if (c->fuse_se || c->split) {
        const int tiles1 = tiles_of(bw.conv1);
        c->out_class = 2;
        TRY(run_conv(c, pf, (name + ".conv1").c_str(), bw.conv1, in, t, H, true, false, ws.se_part, false));
        c->out_class = 4;
        pf.begin((name + ".se_stats").c_str());
        HIP_TRY(c, launch_se_border(t, ws.B, H, ws.W, ws.Wa, planes, c->split, ws.se_part, tiles1, ws.se_border, c->stream));
        pf.end();
        pf.begin((name + ".se_scale").c_str());
        HIP_TRY(c, launch_se_premean(ws.se_border, t, bw.conv2.w, bw.conv2.bias, ws.B, H, ws.W, ws.Wa, planes,
                                     bw.conv2.coutPad, c->split, ws.se_mean, bw.se.w1, bw.se.w2, ws.se_scale,
                                     ws.se_counter, c->stream));
        pf.end();
        if (fuse_ds)
            TRY(run_conv(c, pf, (name + ".conv2+se+ds").c_str(), bw.conv2, ActDesc{t, H, planes}, o, H, true, false,
                         nullptr, false, ws.se_scale, nullptr, &bw.ds, in.p));
        else
            TRY(run_conv(c, pf, (name + ".conv2+se").c_str(), bw.conv2, ActDesc{t, H, planes}, o, H, true, false,
                         nullptr, false, ws.se_scale, res));
        return HCTR_OK;
    }
    c->out_class = 2;
    TRY(run_conv(c, pf, (name + ".conv1").c_str(), bw.conv1, in, t, H, true, false, nullptr, false));
    c->out_class = 4;
    TRY(run_conv(c, pf, (name + ".conv2").c_str(), bw.conv2, ActDesc{t, H, planes}, o, H, false, false,
                 ws.se_part, false));
    const int tiles = tiles_of(bw.conv2);
    pf.begin((name + ".se_fc").c_str());
    HIP_TRY(c, launch_se_fc(ws.se_part, tiles, bw.se.w1, bw.se.w2, ws.se_scale, ws.B, planes, inv_hw, c->stream));
    pf.end();
    pf.begin((name + ".se_apply").c_str());
    HIP_TRY(c, launch_se_apply(o, res, ws.se_scale, (int64_t)(H + 2) * ws.Wa * planes, ws.B, planes, c->stream));
    pf.end();
    return HCTR_OK;
}

// trunk + head for the staged batch in ws.img: leaves [B*W][cpad] fp32 logits in ws.logits, or - greedy
// decode, fused_argmax - only the per-column argmax in ws.colidx (the 29 kB-per-column logits never exist).
// ResNet.forward :115-153 and hctr_model.forward :171-176; np.argmax(preds, 2) utils/ctc_codec.py:75.
enum HeadMode { HEAD_LOGITS = 0, HEAD_ARGMAX = 1, HEAD_BEAM = 2 };

// optional workspace parts a forward in this head mode uses. Callers pass them to ensure_workspace BEFORE the input
// is staged: carving a part may move the arena, which would lose an image already copied into it.
int ws_need(const hctr_ctx* c, HeadMode mode, bool guarded = false) {
    int need = 0;
    if (mode == HEAD_LOGITS) need |= WS_LOGITS;
    if (mode == HEAD_BEAM) need |= WS_BEAM;
    if (!(c->fuse_stem && !c->split)) need |= WS_S0;
    if (guarded) need |= WS_GUARD;
    return need;
}

// the head projection's launch arguments for the active workspace (models/handwritten_ctr_model.py:175)
void head_args(hctr_ctx* c, ConvArgs* out, ConvTile* tile) {
    const Workspace& ws = c->ws;
    ConvArgs a{};
    a.x = ws.headin; a.w = c->wts().head.w; a.bias = c->wts().head.bias; a.y = ws.logits;
    a.Cin = c->wts().head.cin; a.Cout = c->num_classes; a.CoutPad = c->cpad;
    a.M = (int64_t)ws.B * ws.W; a.ldo = c->cpad;
    *tile = (c->big_tiles && c->cpad % 256 == 0) ? TILE_256x256 : TILE_128x128;
    const int hbm = *tile == TILE_256x256 ? 256 : 128;
    a.mtiles = (int)((a.M + hbm - 1) / hbm); a.ntiles = c->cpad / hbm;
    *out = a;
}

// guarded: also leave each line's {smallest top-1/top-2 logit margin, largest |logit|} in ws.line_guard (mode 2's
// f16 passes; the figures come from the head epilogue's partials, or from the stored logits in HEAD_LOGITS mode)
int run_forward(hctr_ctx* c, int img_f32, bool have_widths, HeadMode mode = HEAD_LOGITS, bool guarded = false) {
    if ((c->ws.features & ws_need(c, mode, guarded)) != ws_need(c, mode, guarded) || c->ws.split != c->split)
        return fail(c, HCTR_ERR_STATE, "workspace lacks a part this forward needs (ensure_workspace before staging)");
    Workspace& ws = c->ws;
    const WeightSet& wt = c->wts();
    if (!wt.built) return fail(c, HCTR_ERR_STATE, "the weight set of this precision mode is not resident");
    Prof pf(c);
    if (c->persist_dynamic) {
        HIP_TRY(c, hipMemsetAsync(ws.tile_ctrs, 0, 64 * 8 * 4, c->stream));
        c->conv_seq = 0;
    }
    // aliased workspace: a buffer changes geometry from stage to stage, so the stored zero borders of a stage's buffers
    // are re-established when the stage is entered (the buffers' previous users are dead by then, see ensure_workspace)
    auto zero_stage = [&](int s, bool input, bool blocks) -> int {
        if (!ws.aliased) return HCTR_OK;
        const int H = kStageH[s], planes = kStagePlanes[s - 1], cin_s = s == 1 ? 64 : kStagePlanes[s - 2];
        const int mm = c->chm();
        pf.begin("zero_borders");
        if (input) HIP_TRY(c, launch_zero_borders(ws.x[s], ws.B, H, ws.W, ws.Wa, cin_s * mm, c->stream));
        for (int i = 0; blocks && i < ((s == 4) ? 2 : 3); ++i)
            HIP_TRY(c, launch_zero_borders(ws.p[s][i], ws.B, H, ws.W, ws.Wa, planes * mm, c->stream));
        pf.end();
        return HCTR_OK;
    };
    TRY(zero_stage(1, true, true));
    c->out_class = 8;
    if (c->fuse_stem && !c->split) {
        // conv0_1's output (16 kB per pixel column) never reaches HBM: it is computed into conv0_2's LDS halo
        TRY(run_conv(c, pf, "stem+conv0_2+pool", wt.conv0_2, ActDesc{nullptr, 128, 64}, ws.x[1], 64, true, true, nullptr,
                     false, nullptr, nullptr, nullptr, nullptr, img_f32 ? 1 : 0, have_widths));
    } else {
        pf.begin("stem.conv0_1");
        HIP_TRY(c, launch_stem(ws.img, img_f32, have_widths ? ws.widths : nullptr, wt.stem_w, wt.stem_b, ws.s0, ws.B,
                               ws.W, ws.Wa, c->split ? ((c->x3_mask & 8) ? 1 : 2) : 0, c->stream));
        pf.end();
        TRY(run_conv(c, pf, "conv0_2+pool", wt.conv0_2, ActDesc{ws.s0, 128, 64}, ws.x[1], 64, true, true, nullptr, false));
    }
    int cin = 64;
    for (int s = 1; s <= 4; ++s) {
        const int H = kStageH[s], planes = kStagePlanes[s - 1];
        ActDesc cur{ws.x[s], H, cin};
        int ci = -1;                              // index of the buffer holding `cur`; -1 = stage input
        for (size_t i = 0; i < wt.blocks[s - 1].size(); ++i) {
            const BlockW& bw = wt.blocks[s - 1][i];
            char nm[48];
            snprintf(nm, sizeof(nm), "block%d.%zu", s, i);
            // three rotating buffers: t and o go to the two that do not hold the block input; the
            // 1x1-downsample residual (first block of stages 1-3 only) uses the third.
            const int ti = ci < 0 ? 0 : (ci + 1) % 3, oi = ci < 0 ? 1 : (ci + 2) % 3;
            if (bw.has_ds && ci >= 0) return fail(c, HCTR_ERR_STATE, "downsample only expected on a stage's first block");
            TRY(run_block(c, pf, nm, bw, cur, ws.p[s][ti], ws.p[s][oi], bw.has_ds ? ws.p[s][2] : nullptr, planes));
            cur = ActDesc{ws.p[s][oi], H, planes};
            ci = oi;
        }
        char nm[32];
        snprintf(nm, sizeof(nm), "conv%d+pool", s);
        c->out_class = s < 4 ? 8 : 16;            // (conv4+pool writes the head input)
        if (s < 4) {
            TRY(zero_stage(s + 1, true, false));                 // x[s+1] = pool 0: dead since this stage's first block
            TRY(run_conv(c, pf, nm, wt.stage_conv[s - 1], cur, ws.x[s + 1], H / 2, true, true, nullptr, false));
            TRY(zero_stage(s + 1, false, true));                 // the block buffers: dead once the launch above has read `cur`
        } else
            TRY(run_conv(c, pf, nm, wt.stage_conv[s - 1], cur, ws.headin, H / 2, true, true, nullptr, true));
        cin = planes;
    }
    // head GEMM
    ConvArgs a;
    ConvTile htile;
    head_args(c, &a, &htile);
    auto line_guard = [&]() -> int {
        pf.begin("line_guard");
        HIP_TRY(c, launch_line_guard(ws.col_margin, ws.col_abs, ws.B, ws.W, ws.line_guard, c->stream));
        pf.end();
        return HCTR_OK;
    };
    if (mode == HEAD_ARGMAX || mode == HEAD_BEAM) {
        a.y = nullptr;
        a.amax_val = ws.amax_val;
        a.amax_idx = ws.amax_idx;
        if (guarded) { a.amax_val2 = ws.amax_val2; a.amax_abs = ws.amax_abs; }
        if (mode == HEAD_BEAM) { a.psum = ws.psum; a.blank_logit = ws.blank_logit; }
        pf.begin(mode == HEAD_BEAM ? "head.linear+partials" : "head.linear+argmax");
        HIP_TRY(c, launch_conv(a, htile, 1, true, c->stream));
        pf.end();
        if (mode == HEAD_BEAM && !guarded) return HCTR_OK;
        pf.begin("argmax_partials");
        HIP_TRY(c, launch_argmax_partials(ws.amax_val, ws.amax_idx, a.ntiles * kLinearWN, a.M,
                                          mode == HEAD_BEAM ? nullptr : ws.colidx, c->stream, a.amax_val2, a.amax_abs,
                                          guarded ? ws.col_margin : nullptr, guarded ? ws.col_abs : nullptr));
        pf.end();
        if (guarded) TRY(line_guard());
        return HCTR_OK;
    }
    pf.begin("head.linear");
    HIP_TRY(c, launch_conv(a, htile, 1, true, c->stream));
    pf.end();
    if (guarded) {
        pf.begin("row_guard");
        HIP_TRY(c, launch_row_guard(ws.logits, c->cpad, a.M, c->num_classes, ws.col_margin, ws.col_abs, c->stream));
        pf.end();
        TRY(line_guard());
    }
    return HCTR_OK;
}

// second half of the fused beam front end for the active workspace (after run_forward(HEAD_BEAM)): thresholds, the
// head GEMM once more with the list-emitting epilogue, selection. Device outputs are indexed r = t*B + b.
int beam_finish(hctr_ctx* c, int k, bool want_candidates, double thresh, int32_t* d_idx, float* d_lp, float* d_bl,
                float* d_st, int32_t* d_cnt) {
    Workspace& ws = c->ws;
    Prof pf(c);
    ConvArgs a;
    ConvTile htile;
    head_args(c, &a, &htile);
    const int P = a.ntiles * kLinearWN;
    pf.begin("beam_thresholds");
    HIP_TRY(c, launch_beam_thresholds(ws.amax_val, ws.psum, P, a.M, k, thresh, want_candidates ? 1 : 0, ws.row_thr,
                                      ws.emit_cnt, c->stream));
    pf.end();
    a.y = nullptr;
    a.row_thr = ws.row_thr; a.emit_cnt = ws.emit_cnt; a.emit_list = ws.emit_list; a.emit_cap = kBeamCap;
    a.esum = ws.esum;
    pf.begin("head.linear+lists");
    HIP_TRY(c, launch_conv(a, htile, 1, true, c->stream));
    pf.end();
    HIP_TRY(c, hipMemsetAsync(ws.overflow, 0, 4, c->stream));
    pf.begin("beam_select");
    HIP_TRY(c, launch_beam_select(ws.row_thr, ws.emit_cnt, ws.emit_list, kBeamCap, ws.esum, P, ws.blank_logit, ws.B, ws.W,
                                  k, thresh, d_idx, d_lp, d_bl, d_st, d_cnt, ws.overflow, c->stream));
    pf.end();
    return HCTR_OK;
}

// Copy the lines `lines[0..nb)` of the caller's batch (and their widths) into the workspace. Runs of consecutive
// lines become one copy each, so an ordinary pass (a contiguous range) is a single copy.
int stage_input(hctr_ctx* c, const void* img, int img_dtype, int img_on_device, const int32_t* widths,
                const int* lines, int nb, int W) {
    Workspace& ws = c->ws;
    const size_t esz = img_dtype == HCTR_F32 ? 4 : 1;
    const size_t per = (size_t)kImgH * W * esz;
    const hipMemcpyKind kind = img_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    for (int i = 0; i < nb;) {
        int j = i + 1;
        while (j < nb && lines[j] == lines[j - 1] + 1) ++j;
        HIP_TRY(c, hipMemcpyAsync((char*)ws.img + (size_t)i * per, (const char*)img + (size_t)lines[i] * per,
                                  per * (size_t)(j - i), kind, c->stream));
        i = j;
    }
    if (widths) {
        c->h_widths.emplace_back((size_t)nb);
        std::vector<int32_t>& hw = c->h_widths.back();
        for (int i = 0; i < nb; ++i) {
            const int32_t w = widths[lines[i]];
            if (w < 1 || w > W) return fail(c, HCTR_ERR_ARG, "widths[%d]=%d outside [1,%d]", lines[i], w, W);
            hw[(size_t)i] = w;
        }
        HIP_TRY(c, hipMemcpyAsync(ws.widths, hw.data(), (size_t)nb * 4, hipMemcpyHostToDevice, c->stream));
    }
    return HCTR_OK;
}

int check_forward_args(hctr_ctx* c, const void* img, int img_dtype, int B, int W) {
    if (!c) return HCTR_ERR_ARG;
    if (!c->finalized) return fail(c, HCTR_ERR_STATE, "weights not finalized: call hctr_finalize_weights first");
    if (!img) return fail(c, HCTR_ERR_ARG, "img is NULL");
    if (img_dtype != HCTR_U8 && img_dtype != HCTR_F32) return fail(c, HCTR_ERR_ARG, "img_dtype must be U8 or F32");
    if (B < 0 || W < 1) return fail(c, HCTR_ERR_ARG, "bad batch shape B=%d W=%d", B, W);
    return HCTR_OK;
}

// lines per internal pass: at most max_cols pixel columns, balanced so every pass of a batch has the
// same shape (one cached workspace per (B, W) instead of a second one for a short tail)
int sub_batch(hctr_ctx* c, int B, int W, bool split) {
    int64_t nb = (c->max_cols / (split ? 3 : 1)) / W;
    if (nb < 1) nb = 1;
    if (nb >= B) return B;
    const int64_t passes = (B + nb - 1) / nb;
    return (int)((B + passes - 1) / passes);
}

// mode 2: which lines of the finished f16 sweep must run again in f16x3. gbuf = [B][2] {min margin, max |logit|}.
// A line is certain when EVERY column's top-1/top-2 margin exceeds twice the f16 logit tolerance
// tol = guard_rel * (max |logit| of the line) + guard_abs: a flip needs err(top1) + err(other) > margin, and each
// error is bounded by tol (the bound tests/test_gpu_parity.py asserts for the f16 mode). NaN / inf figures flag.
std::vector<int> guard_decide(hctr_ctx* c, const std::vector<float>& gbuf, int B) {
    c->g_margin.resize((size_t)B); c->g_scale.resize((size_t)B); c->g_flag.assign((size_t)B, 0);
    std::vector<int> flagged;
    for (int b = 0; b < B; ++b) {
        const float mg = gbuf[2 * (size_t)b], sc = gbuf[2 * (size_t)b + 1];
        c->g_margin[(size_t)b] = mg;
        c->g_scale[(size_t)b] = sc;
        const double thr = 2.0 * (c->guard_rel * (double)sc + c->guard_abs);
        if (!((double)mg > thr)) {
            c->g_flag[(size_t)b] = 1;
            flagged.push_back(b);
        }
    }
    c->g_lines_total += B;
    c->g_flagged_total += (int64_t)flagged.size();
    return flagged;
}

void guard_clear(hctr_ctx* c) {
    c->g_margin.clear(); c->g_scale.clear(); c->g_flag.clear();
}

// the arithmetic of a call's passes returns to the mode's own on every exit path
struct SplitScope {
    hctr_ctx* c;
    explicit SplitScope(hctr_ctx* ctx) : c(ctx) { c->split = c->mode == 1; }
    ~SplitScope() { c->split = c->mode == 1; }
};

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

#ifndef HCTR_SRC_HASH
#define HCTR_SRC_HASH "unhashed-build000"
#endif
// ends in the hash of the sources this binary was built from (_lib.py source_hash: stale-binary detection)
const char* hctr_version(void) {
    return "hctr-hip 0.3 (gfx950, f16 storage / f16 MFMA / f32 accumulate; f16x3 split precision; guarded mode) "
           "hctr-src=" HCTR_SRC_HASH;
}

int hctr_create(hctr_ctx** out, int device, int num_classes) {
    return guard(nullptr, [&]() -> int {
        if (!out) return fail(nullptr, HCTR_ERR_ARG, "out is NULL");
        *out = nullptr;
        if (num_classes < 3) return fail(nullptr, HCTR_ERR_ARG, "num_classes must be >= 3 (blank + chars + unknown)");
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0)
            return fail(nullptr, HCTR_ERR_HIP, "no HIP device available (%s): the hctr engine has no CPU fallback",
                        hipGetErrorString(e));
        if (device < 0 || device >= ndev) return fail(nullptr, HCTR_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
        e = hipSetDevice(device);
        if (e != hipSuccess) return fail(nullptr, HCTR_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
        hctr_ctx* c = new hctr_ctx();
        c->device = device;
        c->num_classes = num_classes;
        c->cpad = (num_classes + 255) / 256 * 256;
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(nullptr, HCTR_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        if (const char* bt = getenv("HCTR_BIG_TILES")) c->big_tiles = atoi(bt) != 0;
        if (const char* hm = getenv("HCTR_HALO")) c->halo_mode = atoi(hm);
        if (const char* pr = getenv("HCTR_PRECISION")) {
            const std::string m(pr);
            c->mode = m == "f16x3" ? 1 : (m == "auto" ? 2 : 0);
            c->split = c->mode == 1;
        }
        if (const char* fs = getenv("HCTR_FUSE_SE")) c->fuse_se = atoi(fs) != 0;
        if (const char* fa = getenv("HCTR_FUSE_ARGMAX")) c->fuse_argmax = atoi(fa) != 0;
        if (const char* fd = getenv("HCTR_FUSE_DS")) c->fuse_ds = atoi(fd) != 0;
        if (const char* fb = getenv("HCTR_FUSE_BEAM")) c->fuse_beam = atoi(fb) != 0;
        if (const char* fs2 = getenv("HCTR_FUSE_STEM")) c->fuse_stem = atoi(fs2) != 0;
        if (const char* ps = getenv("HCTR_PERSIST")) c->persist_dynamic = atoi(ps) == 2;
        if (const char* xm = getenv("HCTR_X3_MASK")) c->x3_mask = atoi(xm) & 31;
        if (const char* rt = getenv("HCTR_RTOUCH")) c->rtouch = atoi(rt) != 0;
        if (const char* rp = getenv("HCTR_RPRE")) c->rpre = atoi(rp) != 0;
        if (const char* wa = getenv("HCTR_WS_ALIAS")) c->ws_alias = atoi(wa) != 0 ? 1 : 0;
        if (const char* wd = getenv("HCTR_WS_DEDICATED_MAX_GB")) {
            const double g = atof(wd);
            if (g >= 0) c->ws_dedicated_max = (size_t)(g * 1073741824.0);
        }
        if (const char* mc = getenv("HCTR_MAX_COLS")) {
            const long long v = atoll(mc);
            if (v > 0) c->max_cols = v;
        }
        *out = c;
        return HCTR_OK;
    });
}

void hctr_destroy(hctr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->arena) (void)hipFree(c->arena);
    if (c->pin) (void)hipHostFree(c->pin);
    free_pool(c->wallocs);
    free_pool(c->beam_allocs);
    if (c->stamp_buf) (void)hipFree(c->stamp_buf);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* hctr_last_error(const hctr_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int hctr_load_tensor(hctr_ctx* c, const char* key, const void* host_ptr, const int64_t* shape, int ndim, int dtype) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (!key || (!host_ptr && ndim > 0) || ndim < 0 || ndim > 8) return fail(c, HCTR_ERR_ARG, "bad tensor arguments");
        if (c->finalized) return fail(c, HCTR_ERR_STATE, "weights already finalized");
        const std::string k(key);
        if (dtype == HCTR_I64) {       // num_batches_tracked: accepted, unused in eval mode
            if (k.size() < 19 || k.compare(k.size() - 19, 19, "num_batches_tracked") != 0)
                return fail(c, HCTR_ERR_KEY, "Unexpected key(s) in state_dict: \"%s\" (int64)", key);
            return HCTR_OK;
        }
        if (dtype != HCTR_F32) return fail(c, HCTR_ERR_ARG, "tensor %s: only float32 parameters are supported", key);
        HostTensor t;
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) {
            if (shape[i] < 0) return fail(c, HCTR_ERR_ARG, "negative dimension");
            t.shape.push_back(shape[i]);
            n *= shape[i];
        }
        t.data.assign((const float*)host_ptr, (const float*)host_ptr + n);
        c->host[k] = std::move(t);
        return HCTR_OK;
    });
}

int hctr_finalize_weights(hctr_ctx* c) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (c->finalized) return fail(c, HCTR_ERR_STATE, "weights already finalized");
        HIP_TRY(c, hipSetDevice(c->device));
        // the f16 set for modes 0 and 2, the f16x3 set for modes 1 and 2 (build_* read c->split for the row layout)
        auto build_set = [&](bool split) -> int {
            c->split = split;
            WeightSet& ws = c->wset[split ? 1 : 0];
            TRY(build_stem(c, ws));
            TRY(build_conv(c, "cnn.conv0_2", "cnn.bn0_2", 64, 64, 3, true, 64, &ws.conv0_2));
            int inpl = 64;
            for (int s = 0; s < 4; ++s) {
                const int planes = kStagePlanes[s];
                ws.blocks[s].assign(kStageBlocks[s], BlockW());
                for (int i = 0; i < kStageBlocks[s]; ++i) {
                    BlockW& bw = ws.blocks[s][i];
                    const std::string p = "cnn.block" + std::to_string(s + 1) + "." + std::to_string(i);
                    if (i == 0 && inpl != planes) {
                        bw.has_ds = true;
                        TRY(build_conv(c, p + ".downsample.0", p + ".downsample.1", inpl, planes, 1, false, 128, &bw.ds));
                    }
                    TRY(build_conv(c, p + ".conv1", p + ".bn1", inpl, planes, 3, true, 128, &bw.conv1));
                    TRY(build_conv(c, p + ".conv2", p + ".bn2", planes, planes, 3, true, 128, &bw.conv2));
                    TRY(build_se(c, p + ".se", planes, &bw.se));
                    inpl = planes;
                }
                const std::string k = "cnn.conv" + std::to_string(s + 1);
                TRY(build_conv(c, k, "cnn.bn" + std::to_string(s + 1), planes, planes, 3, true, 128, &ws.stage_conv[s]));
            }
            TRY(build_head(c, ws));
            ws.built = true;
            return HCTR_OK;
        };
        int rc = HCTR_OK;
        if (c->mode != 1) rc = build_set(false);
        if (rc == HCTR_OK && c->mode != 0) rc = build_set(true);
        c->split = c->mode == 1;
        TRY(rc);
        // strict load: no unexpected float keys (load_state_dict(strict=True), test.py:153)
        for (auto& kv : c->host)
            if (!kv.second.used)
                return fail(c, HCTR_ERR_KEY, "Unexpected key(s) in state_dict: \"%s\"", kv.first.c_str());
        c->host.clear();
        c->finalized = true;
        return HCTR_OK;
    });
}

int hctr_set_precision(hctr_ctx* c, int mode) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (mode < 0 || mode > 2) return fail(c, HCTR_ERR_ARG, "precision mode must be 0 (f16), 1 (f16x3) or 2 (guarded: f16, "
                                                                 "uncertain lines again in f16x3)");
        if (c->finalized) {
            // after the weights are resident the mode may still move between the modes whose weight set(s) exist
            // (a context finalized in mode 2 holds both sets and serves all three)
            const bool need0 = mode != 1, need1 = mode != 0;
            if ((need0 && !c->wset[0].built) || (need1 && !c->wset[1].built))
                return fail(c, HCTR_ERR_STATE, "precision mode %d needs a weight set this context did not build: choose it "
                                               "(or mode 2) before hctr_finalize_weights", mode);
        }
        c->mode = mode;
        c->split = mode == 1;
        return HCTR_OK;
    });
}

int hctr_set_guard(hctr_ctx* c, double rel, double abs_tol) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (!(rel >= 0.0) || !(abs_tol >= 0.0)) return fail(c, HCTR_ERR_ARG, "guard tolerances must be >= 0");
        c->guard_rel = rel;
        c->guard_abs = abs_tol;
        return HCTR_OK;
    });
}

int hctr_last_guard(hctr_ctx* c, int64_t* lines, int64_t* flagged, uint8_t* flags, float* min_margin, float* scale,
                    int64_t cap) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        const int64_t n = (int64_t)c->g_flag.size();
        int64_t nf = 0;
        for (uint8_t f : c->g_flag) nf += f ? 1 : 0;
        if (lines) *lines = n;
        if (flagged) *flagged = nf;
        const int64_t m = n < cap ? n : (cap < 0 ? 0 : cap);
        if (flags && m) memcpy(flags, c->g_flag.data(), (size_t)m);
        if (min_margin && m) memcpy(min_margin, c->g_margin.data(), (size_t)m * 4);
        if (scale && m) memcpy(scale, c->g_scale.data(), (size_t)m * 4);
        return HCTR_OK;
    });
}

int hctr_workspace_stats(hctr_ctx* c, int64_t* arena_bytes, int64_t* arena_allocations, int64_t* recarves) {
    if (!c) return HCTR_ERR_ARG;
    if (arena_bytes) *arena_bytes = (int64_t)c->arena_cap;
    if (arena_allocations) *arena_allocations = c->arena_reallocs;
    if (recarves) *recarves = c->ws_recarves;
    return HCTR_OK;
}

int hctr_lines_per_pass(hctr_ctx* c, int B, int W, int f16x3) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (B < 1 || W < 1) return fail(c, HCTR_ERR_ARG, "bad batch shape B=%d W=%d", B, W);
        return sub_batch(c, B, W, f16x3 != 0);
    });
}

int hctr_set_profiling(hctr_ctx* c, int enabled) {
    if (!c) return HCTR_ERR_ARG;
    c->profiling = enabled != 0;
    return HCTR_OK;
}

int hctr_last_profile(hctr_ctx* c, char* names_buf, int cap, float* ms, int max_n) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::string names;
        std::vector<std::string> order;
        int n = 0;
        for (auto& pe : c->prof) {
            float t = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&t, pe.e0, pe.e1));
            size_t i = 0;
            while (i < order.size() && order[i] != pe.name) ++i;      // internal passes repeat the layer names
            if (i == order.size()) {
                if (n >= max_n) break;
                order.push_back(pe.name);
                ms[n++] = 0.f;
                names += pe.name;
                names += '\n';
            }
            ms[i] += t;
        }
        if (names_buf && cap > 0) {
            strncpy(names_buf, names.c_str(), (size_t)cap - 1);
            names_buf[cap - 1] = 0;
        }
        return n;
    });
}

int hctr_forward_logits(hctr_ctx* c, const void* img, int img_dtype, int img_on_device, const int32_t* widths,
                        int B, int W, float* out_wbc, int out_on_device) {
    return guard(c, [&]() -> int {
        TRY(check_forward_args(c, img, img_dtype, B, W));
        if (!out_wbc) return fail(c, HCTR_ERR_ARG, "out_wbc is NULL");
        if (B == 0) return HCTR_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        const int C = c->num_classes;
        const bool guarded = c->mode == 2;
        SplitScope scope(c);
        prof_reset(c);
        guard_clear(c);
        float* dev_out = out_wbc;
        std::vector<void*> tmp;
        PoolGuard tmp_guard{tmp};
        if (!out_on_device) TRY(dev_alloc(c, tmp, &dev_out, (size_t)B * W * C, false));
        std::vector<float> gbuf(guarded ? (size_t)2 * B : 0);
        // one pass: the lines `lines[0..nb)` -> their rows of the [W][B][C] output
        auto pass = [&](const int* lines, int nb, bool guard_pass, float* guard_dst) -> int {
            TRY(ensure_workspace(c, nb, W, ws_need(c, HEAD_LOGITS, guard_pass)));
            TRY(stage_input(c, img, img_dtype, img_on_device, widths, lines, nb, W));
            TRY(run_forward(c, img_dtype == HCTR_F32, widths != nullptr, HEAD_LOGITS, guard_pass));
            for (int i = 0; i < nb;) {             // [nb*W][cpad] -> out[t][line][C], one launch per run of consecutive lines
                int j = i + 1;
                while (j < nb && lines[j] == lines[j - 1] + 1) ++j;
                HIP_TRY(c, launch_logits_to_wbc(c->ws.logits + (size_t)i * W * c->cpad, c->cpad, j - i, W, C, dev_out, B,
                                                lines[i], c->stream));
                i = j;
            }
            if (guard_pass)
                HIP_TRY(c, hipMemcpyAsync(guard_dst, c->ws.line_guard, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
            return HCTR_OK;
        };
        std::vector<int> all((size_t)B);
        for (int b = 0; b < B; ++b) all[(size_t)b] = b;
        int rc = HCTR_OK;
        const int nbmax = sub_batch(c, B, W, c->split);
        for (int b0 = 0; b0 < B && rc == HCTR_OK; b0 += nbmax)
            rc = pass(all.data() + b0, std::min(nbmax, B - b0), guarded, guarded ? gbuf.data() + 2 * (size_t)b0 : nullptr);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(e));
        if (rc == HCTR_OK && guarded) {
            const std::vector<int> flagged = guard_decide(c, gbuf, B);
            if (!flagged.empty()) {               // the uncertain lines once more, in f16x3, into the same output rows
                c->split = true;
                const int nf = (int)flagged.size(), nb3 = sub_batch(c, nf, W, true);
                for (int o = 0; o < nf && rc == HCTR_OK; o += nb3) rc = pass(flagged.data() + o, std::min(nb3, nf - o), false, nullptr);
            }
        }
        if (rc == HCTR_OK && !out_on_device) {
            e = hipMemcpyAsync(out_wbc, dev_out, (size_t)B * W * C * 4, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "D2H logits: %s", hipGetErrorString(e));
        }
        e = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(e));
        return rc;
    });
}

int hctr_greedy(hctr_ctx* c, const void* img, int img_dtype, int img_on_device, const int32_t* widths, int B, int W,
                int32_t* labels, int32_t* lengths) {
    return guard(c, [&]() -> int {
        TRY(check_forward_args(c, img, img_dtype, B, W));
        if (!labels || !lengths) return fail(c, HCTR_ERR_ARG, "labels/lengths is NULL");
        if (B == 0) return HCTR_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        const int C = c->num_classes;
        const bool guarded = c->mode == 2;
        SplitScope scope(c);
        prof_reset(c);
        guard_clear(c);
        std::vector<float> gbuf(guarded ? (size_t)2 * B : 0);
        const HeadMode hm = c->fuse_argmax ? HEAD_ARGMAX : HEAD_LOGITS;
        // every pass queues async copies into host buffers: on a failure the stream is still drained before
        // returning, so nothing is in flight into (or out of) caller memory after an error
        auto pass = [&](const int* lines, int nb, bool guard_pass, int32_t* lab_dst, int32_t* len_dst, float* guard_dst) -> int {
            TRY(ensure_workspace(c, nb, W, ws_need(c, hm, guard_pass)));
            TRY(stage_input(c, img, img_dtype, img_on_device, widths, lines, nb, W));
            TRY(run_forward(c, img_dtype == HCTR_F32, widths != nullptr, hm, guard_pass));
            Workspace& ws = c->ws;
            Prof pf(c);
            if (!c->fuse_argmax) {
                pf.begin("argmax_rows");
                HIP_TRY(c, launch_argmax_rows(ws.logits, c->cpad, (int64_t)nb * W, C, ws.colidx, 0, 0, c->stream));
                pf.end();
            }
            pf.begin("ctc_collapse");
            HIP_TRY(c, launch_ctc_collapse(ws.colidx, nb, W, C, ws.labels, ws.lengths, c->stream));
            pf.end();
            HIP_TRY(c, hipMemcpyAsync(lab_dst, ws.labels, (size_t)nb * W * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(len_dst, ws.lengths, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
            if (guard_pass)
                HIP_TRY(c, hipMemcpyAsync(guard_dst, ws.line_guard, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
            return HCTR_OK;
        };
        std::vector<int> all((size_t)B);
        for (int b = 0; b < B; ++b) all[(size_t)b] = b;
        int rc = HCTR_OK;
        const int nbmax = sub_batch(c, B, W, c->split);
        for (int b0 = 0; b0 < B && rc == HCTR_OK; b0 += nbmax)
            rc = pass(all.data() + b0, std::min(nbmax, B - b0), guarded, labels + (size_t)b0 * W, lengths + b0,
                      guarded ? gbuf.data() + 2 * (size_t)b0 : nullptr);
        hipError_t es = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && es != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(es));
        if (rc != HCTR_OK || !guarded) return rc;
        // guarded precision: the lines the f16 sweep cannot certify run again in f16x3 at the SAME padded width (a line's
        // result depends on nothing but its own pixels and W), and their labels replace the f16 ones
        const std::vector<int> flagged = guard_decide(c, gbuf, B);
        if (flagged.empty()) return HCTR_OK;
        c->split = true;
        const int nf = (int)flagged.size(), nb3 = sub_batch(c, nf, W, true);
        c->h_labels.resize((size_t)nf * W);
        c->h_lengths.resize((size_t)nf);
        for (int o = 0; o < nf && rc == HCTR_OK; o += nb3)
            rc = pass(flagged.data() + o, std::min(nb3, nf - o), false, c->h_labels.data() + (size_t)o * W,
                      c->h_lengths.data() + o, nullptr);
        es = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && es != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(es));
        if (rc != HCTR_OK) return rc;
        for (int i = 0; i < nf; ++i) {
            const int n = c->h_lengths[(size_t)i];
            lengths[flagged[(size_t)i]] = n;
            if (n > 0) memcpy(labels + (size_t)flagged[(size_t)i] * W, c->h_labels.data() + (size_t)i * W, (size_t)n * 4);
        }
        return HCTR_OK;
    });
}

int hctr_decode_greedy_logits(hctr_ctx* c, const float* logits_wbc, int on_device, int W, int B, int C,
                              int32_t* labels, int32_t* lengths) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (W < 0 || B < 0 || C < 2) return fail(c, HCTR_ERR_ARG, "bad logits shape W=%d B=%d C=%d", W, B, C);
        if ((int64_t)W * B == 0) return HCTR_OK;     // reference: zero-length lines produce no output (:85-86)
        if (!logits_wbc || !labels || !lengths) return fail(c, HCTR_ERR_ARG, "NULL pointer");
        HIP_TRY(c, hipSetDevice(c->device));
        std::vector<void*> tmp;
        PoolGuard tmp_guard{tmp};
        const size_t n = (size_t)W * B * C;
        const float* dev = logits_wbc;
        float* up = nullptr;
        int32_t *idx = nullptr, *dl = nullptr, *dn = nullptr;
        int rc = HCTR_OK;
        if (!on_device) {
            rc = dev_alloc(c, tmp, &up, n, false);
            if (rc == HCTR_OK) {
                hipError_t e = hipMemcpyAsync(up, logits_wbc, n * 4, hipMemcpyHostToDevice, c->stream);
                if (e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "H2D logits: %s", hipGetErrorString(e));
            }
            dev = up;
        }
        if (rc == HCTR_OK) rc = dev_alloc(c, tmp, &idx, (size_t)W * B, false);
        if (rc == HCTR_OK) rc = dev_alloc(c, tmp, &dl, (size_t)W * B, false);
        if (rc == HCTR_OK) rc = dev_alloc(c, tmp, &dn, (size_t)B, false);
        if (rc == HCTR_OK) {
            // rows of the WBC tensor are r = t*B + b; the argmax kernel writes idx as [b][t]
            hipError_t e = launch_argmax_rows(dev, C, (int64_t)W * B, C, idx, B, W, c->stream);
            if (e == hipSuccess) e = launch_ctc_collapse(idx, B, W, C, dl, dn, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(labels, dl, (size_t)W * B * 4, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(lengths, dn, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "decode_greedy_logits: %s", hipGetErrorString(e));
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(e));
        free_pool(tmp);
        return rc;
    });
}

int hctr_beam_frontend(hctr_ctx* c, const void* img, int img_dtype, int img_on_device, const int32_t* widths,
                       const float* logits_wbc, int logits_on_device, int B, int W, int C, int k,
                       int want_candidates, int32_t* topk_idx, float* topk_logp, float* blank_logp,
                       int64_t* num_candidates) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (B < 0 || W < 0 || k < 1) return fail(c, HCTR_ERR_ARG, "bad shape B=%d W=%d k=%d", B, W, k);
        if (!topk_idx || !topk_logp || !blank_logp) return fail(c, HCTR_ERR_ARG, "NULL output pointer");
        c->cand_off.assign((size_t)W * B + 1, 0);
        c->cand_idx.clear();
        c->cand_logp.clear();
        if (num_candidates) *num_candidates = 0;
        if ((int64_t)B * W == 0) return HCTR_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        const bool from_img = img != nullptr;
        if (from_img) {
            TRY(check_forward_args(c, img, img_dtype, B, W));
            if (C != c->num_classes) return fail(c, HCTR_ERR_ARG, "C=%d differs from the model's %d classes", C, c->num_classes);
        } else if (!logits_wbc) {
            return fail(c, HCTR_ERR_ARG, "neither img nor logits given");
        }
        if (k > C) return fail(c, HCTR_ERR_ARG, "k=%d exceeds C=%d", k, C);
        const double thresh = std::log(0.001);        // utils/ctc_codec.py:128
        const bool guarded = from_img && c->mode == 2;
        SplitScope scope(c);
        prof_reset(c);
        guard_clear(c);
        // candidate lists of one pass (lines in pass order); owner[line] = the pass whose lists are current for the line
        struct PassOut { std::vector<int> lines; std::vector<int64_t> loff; std::vector<int32_t> ci; std::vector<float> cl; };
        std::vector<PassOut> outs;
        std::vector<int> owner((size_t)B, -1);
        std::vector<int32_t> counts((size_t)W * B, 0);
        std::vector<float> gbuf(guarded ? (size_t)2 * B : 0);
        std::vector<void*>& pool = c->beam_allocs;
        // one pass over the lines `lines[0..nb)`: forward (or the caller's logits), log-softmax + top-k (+ lists), results
        // scattered to the lines' (t, line) rows of the host outputs
        auto run_pass = [&](const int* lines, int nb, bool guard_pass, float* guard_dst) -> int {
            const int64_t rows = (int64_t)nb * W;
            free_pool(pool);
            const float* rowsrc = nullptr;
            int64_t ld = 0;
            // fused front end (default): the logits are never stored; the head GEMM runs twice with reducing epilogues
            // (kernels.h ConvArgs). Not for k beyond the part count / kBeamMaxK, and a pass in which some row needed
            // more than kBeamCap list slots (near-uniform logits) is redone through the stored-logits kernels.
            bool fused = from_img && c->fuse_beam && k <= kBeamMaxK && k <= head_parts(c);
            if (from_img) {
                const HeadMode hm = fused ? HEAD_BEAM : HEAD_LOGITS;
                TRY(ensure_workspace(c, nb, W, ws_need(c, hm, guard_pass)));
                TRY(stage_input(c, img, img_dtype, img_on_device, widths, lines, nb, W));
                TRY(run_forward(c, img_dtype == HCTR_F32, widths != nullptr, hm, guard_pass));
                if (guard_pass)
                    HIP_TRY(c, hipMemcpyAsync(guard_dst, c->ws.line_guard, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
                rowsrc = c->ws.logits; ld = c->cpad;
            } else {
                float *up = nullptr, *rowsbuf = nullptr;
                const float* dev = logits_wbc;
                if (!logits_on_device) {
                    TRY(dev_alloc(c, pool, &up, (size_t)rows * C, false));
                    HIP_TRY(c, hipMemcpyAsync(up, logits_wbc, (size_t)rows * C * 4, hipMemcpyHostToDevice, c->stream));
                    dev = up;
                }
                TRY(dev_alloc(c, pool, &rowsbuf, (size_t)rows * C, false));
                HIP_TRY(c, launch_wbc_to_rows(dev, nb, W, C, rowsbuf, C, c->stream));
                rowsrc = rowsbuf; ld = C;
            }
            // device outputs (rows r = t*nb + b): inside the arena on the fused path (no allocation per pass)
            int32_t *d_idx = nullptr, *d_cnt = nullptr;
            float *d_lp = nullptr, *d_bl = nullptr, *d_st = nullptr;
            auto own_outputs = [&]() -> int {
                TRY(dev_alloc(c, pool, &d_idx, (size_t)rows * k, false));
                TRY(dev_alloc(c, pool, &d_lp, (size_t)rows * k, false));
                TRY(dev_alloc(c, pool, &d_bl, (size_t)rows, false));
                TRY(dev_alloc(c, pool, &d_st, (size_t)rows * 2, false));
                TRY(dev_alloc(c, pool, &d_cnt, (size_t)rows, false));
                return HCTR_OK;
            };
            if (fused) {
                const Workspace& w0 = c->ws;
                d_idx = w0.bm_idx; d_lp = w0.bm_lp; d_bl = w0.bm_bl; d_st = w0.bm_st; d_cnt = w0.bm_cnt;
                TRY(beam_finish(c, k, want_candidates != 0, thresh, d_idx, d_lp, d_bl, d_st, d_cnt));
                int32_t ovf = 0;
                HIP_TRY(c, hipMemcpyAsync(&ovf, c->ws.overflow, 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (ovf) {                       // redo this pass with stored logits (re-staged: carving may move the arena)
                    fused = false;
                    ++c->beam_fallbacks;
                    TRY(ensure_workspace(c, nb, W, ws_need(c, HEAD_LOGITS)));
                    TRY(stage_input(c, img, img_dtype, img_on_device, widths, lines, nb, W));
                    TRY(run_forward(c, img_dtype == HCTR_F32, widths != nullptr, HEAD_LOGITS));
                    rowsrc = c->ws.logits;
                    TRY(own_outputs());
                }
            } else {
                TRY(own_outputs());
            }
            if (!fused) {
                Prof pf(c);
                pf.begin("row_topk");
                const hipError_t e = launch_row_topk(rowsrc, ld, nb, W, C, k, thresh, d_idx, d_lp, d_bl, d_st, d_cnt, c->stream);
                pf.end();
                if (e != hipSuccess) return fail(c, HCTR_ERR_HIP, "row_topk: %s (C=%d)", hipGetErrorString(e), C);
            }
            // D2H through a pinned staging buffer (grown on demand): a pageable destination is copied in small staged pieces
            const size_t nk = (size_t)rows * k, need_pin = (2 * nk + 2 * (size_t)rows) * 4;
            if (c->pin_cap < need_pin) {
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                if (c->pin) (void)hipHostFree(c->pin);
                c->pin = nullptr; c->pin_cap = 0;
                void* hp = nullptr;
                if (hipHostMalloc(&hp, need_pin + need_pin / 8, hipHostMallocDefault) != hipSuccess)
                    return fail(c, HCTR_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", need_pin);
                c->pin = (char*)hp; c->pin_cap = need_pin + need_pin / 8;
            }
            int32_t* p_idx = (int32_t*)c->pin;
            float* p_lp = (float*)(p_idx + nk);
            float* p_bl = p_lp + nk;
            int32_t* p_cnt = (int32_t*)(p_bl + rows);
            HIP_TRY(c, hipMemcpyAsync(p_idx, d_idx, nk * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(p_lp, d_lp, nk * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(p_bl, d_bl, (size_t)rows * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(p_cnt, d_cnt, (size_t)rows * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const int32_t* h_idx_p = p_idx; const float* h_lp_p = p_lp; const float* h_bl_p = p_bl; const int32_t* h_cnt_p = p_cnt;
            for (int t = 0; t < W; ++t)
                for (int b = 0; b < nb; ++b) {
                    const size_t src = (size_t)t * nb + b, dst = (size_t)t * B + lines[b];
                    memcpy(topk_idx + dst * k, h_idx_p + src * k, (size_t)k * 4);
                    memcpy(topk_logp + dst * k, h_lp_p + src * k, (size_t)k * 4);
                    blank_logp[dst] = h_bl_p[src];
                    counts[dst] = h_cnt_p[src];
                }
            if (!want_candidates) return HCTR_OK;
            outs.emplace_back();
            PassOut& po = outs.back();
            po.lines.assign(lines, lines + nb);
            for (int b = 0; b < nb; ++b) owner[(size_t)lines[b]] = (int)outs.size() - 1;
            po.loff.resize(rows + 1);
            int64_t tot = 0;
            for (int64_t r = 0; r < rows; ++r) { po.loff[r] = tot; tot += h_cnt_p[r]; }
            po.loff[rows] = tot;
            po.ci.resize(tot); po.cl.resize(tot);
            int64_t* d_off = nullptr;
            int32_t* d_ci = nullptr;
            float* d_cl = nullptr;
            TRY(dev_alloc(c, pool, &d_off, (size_t)rows + 1, false));
            TRY(dev_alloc(c, pool, &d_ci, (size_t)std::max<int64_t>(tot, 1), false));
            TRY(dev_alloc(c, pool, &d_cl, (size_t)std::max<int64_t>(tot, 1), false));
            HIP_TRY(c, hipMemcpyAsync(d_off, po.loff.data(), (size_t)(rows + 1) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, fused ? launch_beam_candidates(c->ws.emit_cnt, c->ws.emit_list, kBeamCap, d_st, nb, W, thresh, d_off, d_ci,
                                                      d_cl, c->stream)
                             : launch_row_candidates(rowsrc, ld, nb, W, C, thresh, d_st, d_off, d_ci, d_cl, c->stream));
            if (tot) {
                HIP_TRY(c, hipMemcpyAsync(po.ci.data(), d_ci, (size_t)tot * 4, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(c, hipMemcpyAsync(po.cl.data(), d_cl, (size_t)tot * 4, hipMemcpyDeviceToHost, c->stream));
            }
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            return HCTR_OK;
        };
        std::vector<int> all((size_t)B);
        for (int b = 0; b < B; ++b) all[(size_t)b] = b;
        int rc = HCTR_OK;
        const int nbmax = from_img ? sub_batch(c, B, W, c->split) : B;
        for (int b0 = 0; b0 < B && rc == HCTR_OK; b0 += nbmax)
            rc = run_pass(all.data() + b0, std::min(nbmax, B - b0), guarded, guarded ? gbuf.data() + 2 * (size_t)b0 : nullptr);
        if (rc == HCTR_OK && guarded) {
            // guarded precision: lines with a column the f16 sweep cannot certify (same criterion as hctr_greedy) once
            // more in f16x3; their rows of every output are replaced
            const std::vector<int> flagged = guard_decide(c, gbuf, B);
            if (!flagged.empty()) {
                c->split = true;
                const int nf = (int)flagged.size(), nb3 = sub_batch(c, nf, W, true);
                for (int o = 0; o < nf && rc == HCTR_OK; o += nb3) rc = run_pass(flagged.data() + o, std::min(nb3, nf - o), false, nullptr);
            }
        }
        (void)hipStreamSynchronize(c->stream);
        free_pool(pool);
        if (rc != HCTR_OK || !want_candidates) return rc;
        // merge the per-pass lists into one CSR in (t*B + b) order, held by the context until fetched
        int64_t tot = 0;
        for (int64_t r = 0; r < (int64_t)W * B; ++r) { c->cand_off[r] = tot; tot += counts[r]; }
        c->cand_off[(size_t)W * B] = tot;
        c->cand_idx.resize(tot); c->cand_logp.resize(tot);
        for (size_t pi = 0; pi < outs.size(); ++pi) {
            const PassOut& po = outs[pi];
            const int nb = (int)po.lines.size();
            for (int t = 0; t < W; ++t)
                for (int b = 0; b < nb; ++b) {
                    if (owner[(size_t)po.lines[(size_t)b]] != (int)pi) continue;      // a later (f16x3) pass re-did this line
                    const size_t src = (size_t)t * nb + b, dst = (size_t)t * B + po.lines[(size_t)b];
                    const int64_t n = po.loff[src + 1] - po.loff[src];
                    if (n) {
                        memcpy(c->cand_idx.data() + c->cand_off[dst], po.ci.data() + po.loff[src], (size_t)n * 4);
                        memcpy(c->cand_logp.data() + c->cand_off[dst], po.cl.data() + po.loff[src], (size_t)n * 4);
                    }
                }
        }
        if (num_candidates) *num_candidates = tot;
        return HCTR_OK;
    });
}

int hctr_beam_fetch_candidates(hctr_ctx* c, int64_t* cand_off, int32_t* cand_idx, float* cand_logp) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (c->cand_off.empty()) return fail(c, HCTR_ERR_STATE, "no candidate lists: call hctr_beam_frontend(want_candidates=1)");
        if (!cand_off) return fail(c, HCTR_ERR_ARG, "cand_off is NULL");
        memcpy(cand_off, c->cand_off.data(), c->cand_off.size() * 8);
        if (!c->cand_idx.empty()) {
            if (!cand_idx || !cand_logp) return fail(c, HCTR_ERR_ARG, "cand_idx/cand_logp is NULL");
            memcpy(cand_idx, c->cand_idx.data(), c->cand_idx.size() * 4);
            memcpy(cand_logp, c->cand_logp.data(), c->cand_logp.size() * 4);
        }
        return HCTR_OK;
    });
}

int hctr_log_softmax(hctr_ctx* c, const float* logits_wbc, int on_device, int W, int B, int C, float* out_host) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (W < 0 || B < 0 || C < 1) return fail(c, HCTR_ERR_ARG, "bad shape");
        const int64_t rows = (int64_t)W * B;
        if (rows == 0) return HCTR_OK;
        if (!logits_wbc || !out_host) return fail(c, HCTR_ERR_ARG, "NULL pointer");
        HIP_TRY(c, hipSetDevice(c->device));
        std::vector<void*> tmp;
        PoolGuard tmp_guard{tmp};
        const float* dev = logits_wbc;
        float *up = nullptr, *y = nullptr;
        int rc = HCTR_OK;
        if (!on_device) {
            rc = dev_alloc(c, tmp, &up, (size_t)rows * C, false);
            if (rc == HCTR_OK) {
                hipError_t e = hipMemcpyAsync(up, logits_wbc, (size_t)rows * C * 4, hipMemcpyHostToDevice, c->stream);
                if (e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "H2D logits: %s", hipGetErrorString(e));
            }
            dev = up;
        }
        if (rc == HCTR_OK) rc = dev_alloc(c, tmp, &y, (size_t)rows * C, false);
        if (rc == HCTR_OK) {
            hipError_t e = launch_log_softmax_rows(dev, rows, C, y, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(out_host, y, (size_t)rows * C * 4, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "log_softmax: %s", hipGetErrorString(e));
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(e));
        free_pool(tmp);
        return rc;
    });
}

int hctr_resize_lines(hctr_ctx* c, const uint8_t* packed_src, int64_t packed_bytes, const int64_t* offsets,
                      const int32_t* heights, const int32_t* widths, const int32_t* channels, int n, int out_height,
                      const int32_t* out_widths, int out_W, uint8_t* out, int out_on_device) {
    return guard(c, [&]() -> int {
        if (!c) return HCTR_ERR_ARG;
        if (n < 0 || out_height < 1 || out_W < 0 || packed_bytes < 0) return fail(c, HCTR_ERR_ARG, "bad shape");
        if (n == 0 || out_W == 0) return HCTR_OK;
        if (!packed_src || !offsets || !heights || !widths || !channels || !out_widths || !out)
            return fail(c, HCTR_ERR_ARG, "NULL pointer");
        if (n > 65535) return fail(c, HCTR_ERR_ARG, "at most 65535 images per call");
        std::vector<ResizeLine> lines((size_t)n);
        for (int i = 0; i < n; ++i) {
            const int ch = channels[i];
            if (ch != 1 && ch != 3 && ch != -3) return fail(c, HCTR_ERR_ARG, "image %d: channels must be 1, 3 (BGR) or -3 (RGB)", i);
            if (heights[i] < 1 || widths[i] < 1) return fail(c, HCTR_ERR_SHAPE, "image %d: empty source", i);
            // cv2.resize asserts !dsize.empty(): a line so narrow that int(128 * w / h) == 0 fails there too
            if (out_widths[i] < 1) return fail(c, HCTR_ERR_SHAPE, "image %d: destination width %d < 1", i, out_widths[i]);
            const int64_t bytes = (int64_t)heights[i] * widths[i] * (ch < 0 ? -ch : ch);
            if (offsets[i] < 0 || offsets[i] + bytes > packed_bytes)
                return fail(c, HCTR_ERR_ARG, "image %d: [%lld, +%lld) outside the packed buffer of %lld bytes", i,
                            (long long)offsets[i], (long long)bytes, (long long)packed_bytes);
            ResizeLine& L = lines[(size_t)i];
            L.src_off = offsets[i];
            L.sh = heights[i];
            L.sw = widths[i];
            L.ch = ch;
            L.dw = out_widths[i];
            // dispatch of cv::resize for INTER_AREA (oracle/resize_ref.py resize_area)
            L.inv_x = (double)L.dw / (double)L.sw;
            L.inv_y = (double)out_height / (double)L.sh;
            L.scale_x = 1.0 / L.inv_x;
            L.scale_y = 1.0 / L.inv_y;
            L.ix = (int)std::nearbyint(L.scale_x);
            L.iy = (int)std::nearbyint(L.scale_y);
            if (L.scale_x >= 1.0 && L.scale_y >= 1.0) {
                const bool fast = std::fabs(L.scale_x - L.ix) < DBL_EPSILON && std::fabs(L.scale_y - L.iy) < DBL_EPSILON;
                L.mode = fast ? 1 : 0;
            } else {
                L.mode = 2;
            }
        }
        HIP_TRY(c, hipSetDevice(c->device));
        std::vector<void*> tmp;
        PoolGuard tmp_guard{tmp};
        uint8_t *src = nullptr, *dst = nullptr;
        ResizeLine* dl = nullptr;
        int rc = dev_alloc(c, tmp, &src, (size_t)std::max<int64_t>(packed_bytes, 1), false);
        if (rc == HCTR_OK) rc = dev_alloc(c, tmp, &dl, (size_t)n, false);
        const size_t out_bytes = (size_t)n * out_height * out_W;
        if (rc == HCTR_OK && !out_on_device) rc = dev_alloc(c, tmp, &dst, out_bytes, false);
        if (rc == HCTR_OK) {
            uint8_t* target = out_on_device ? out : dst;
            hipError_t e = hipMemcpyAsync(src, packed_src, (size_t)packed_bytes, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(dl, lines.data(), sizeof(ResizeLine) * (size_t)n, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = launch_resize_lines(src, dl, n, target, out_height, out_W, c->stream);
            if (e == hipSuccess && !out_on_device)
                e = hipMemcpyAsync(out, dst, out_bytes, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "resize_lines: %s", hipGetErrorString(e));
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (rc == HCTR_OK && e != hipSuccess) rc = fail(c, HCTR_ERR_HIP, "stream sync: %s", hipGetErrorString(e));
        free_pool(tmp);
        return rc;
    });
}

int64_t hctr_debug_stamps(hctr_ctx* c, const char* layer, uint64_t* out, int64_t cap_wgs) {
    return guard<int64_t>(c, [&]() -> int64_t {
        if (!c || cap_wgs < 0) return HCTR_ERR_ARG;
        if (hipSetDevice(c->device) != hipSuccess) return fail(c, HCTR_ERR_HIP, "hipSetDevice");
        if (layer) {                                  // arm: later forwards stamp this layer's workgroups
            if (c->stamp_buf) { (void)hipFree(c->stamp_buf); c->stamp_buf = nullptr; }
            c->stamp_layer = layer;
            c->stamp_cap = cap_wgs;
            c->stamp_n = 0;
            if (cap_wgs > 0 && hipMalloc((void**)&c->stamp_buf, (size_t)cap_wgs * 128) != hipSuccess)
                return fail(c, HCTR_ERR_NOMEM, "stamp buffer");
            return 0;
        }
        if (!out || !c->stamp_buf) return fail(c, HCTR_ERR_STATE, "stamping not armed");
        const int64_t n = std::min(c->stamp_n, cap_wgs);
        if (hipMemcpy(out, c->stamp_buf, (size_t)n * 128, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(c, HCTR_ERR_HIP, "stamp copy");
        return n;
    });
}

int64_t hctr_debug_activation(hctr_ctx* c, const char* name, float* out, int64_t cap, int* Cout, int* Hout) {
    return guard<int64_t>(c, [&]() -> int64_t {
        if (!c || !name) return HCTR_ERR_ARG;
        const Workspace& ws = c->ws;
        if (ws.B == 0) return fail(c, HCTR_ERR_STATE, "no forward has run");
        const half_t* p = nullptr;
        int H = 0, C = 0;
        bool head = false;
        const std::string n(name);
        if (n == "conv0_1") {
            if (!ws.s0) return fail(c, HCTR_ERR_STATE, "conv0_1 is fused into conv0_2 (no buffer): use HCTR_FUSE_STEM=0");
            p = ws.s0; H = 128; C = 64;
        }
        else if (n == "stage0") { p = ws.x[1]; H = 64; C = 64; }
        else if (n == "stage1") { p = ws.x[2]; H = 32; C = 128; }
        else if (n == "stage2") { p = ws.x[3]; H = 16; C = 256; }
        else if (n == "stage3") { p = ws.x[4]; H = 8; C = 512; }
        else if (n == "stage4") { p = ws.headin; H = 4; C = 512; head = true; }
        else if (n.size() == 4 && n[0] == 'p' && n[2] == '.' && n[1] >= '1' && n[1] <= '4' && n[3] >= '0' && n[3] <= '2') {
            // raw rotating block buffer i of stage s ("p<s>.<i>"): what it holds depends on the block count
            const int st = n[1] - '0', bi = n[3] - '0';
            p = ws.p[st][bi]; H = kStageH[st]; C = kStagePlanes[st - 1];
            if (!p) return fail(c, HCTR_ERR_ARG, "buffer %s not allocated", name);
        }
        else return fail(c, HCTR_ERR_ARG, "unknown activation '%s'", name);
        if (ws.aliased && !(n == "stage3" || n == "stage4" || (n.size() == 4 && n[0] == 'p' && n[1] == '4')))
            return fail(c, HCTR_ERR_STATE, "activation '%s' was overwritten by a later stage: the stages share their buffers "
                        "in this workspace layout (HCTR_WS_ALIAS=0 keeps every stage's own)", name);
        const int64_t total = (int64_t)ws.B * C * H * ws.W;
        if (Cout) *Cout = C;
        if (Hout) *Hout = H;
        if (!out || cap < total) return total;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const int m = ws.split ? 3 : 1;                // [hi | lo | hi] planes in f16x3 mode: report hi + lo
        const int64_t elems = head ? (int64_t)ws.B * ws.W * kFeat * m : act_elems(ws.B, H, ws.Wa, C * m);
        std::vector<half_t> host((size_t)elems);
        HIP_TRY(c, hipMemcpy(host.data(), p, (size_t)elems * sizeof(half_t), hipMemcpyDeviceToHost));
        for (int b = 0; b < ws.B; ++b)
            for (int ch = 0; ch < C; ++ch)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < ws.W; ++w) {
                        const int64_t src = head ? (((int64_t)b * ws.W + w) * 4 + h) * (512 * m) + ch
                                                 : (((int64_t)b * (H + 2) + h + 1) * ws.Wa + w + 1) * (C * m) + ch;
                        float v = (float)host[(size_t)src];
                        if (m == 3) v += (float)host[(size_t)src + C];
                        out[(((int64_t)b * C + ch) * H + h) * ws.W + w] = v;
                    }
        return total;
    });
}

}  // extern "C"
