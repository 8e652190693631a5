// Launch interface between the engine (engine.cpp) and the gfx950 kernels (kernels.hip).
// Internal to libhctr_hip.so; the public C ABI is include/hctr_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hctr {

typedef _Float16 half_t;

constexpr int kTileW = 16;      // output columns per pixel tile (all conv layers)
constexpr int kBK = 64;         // input channels per k-step

// One MFMA implicit-GEMM launch: 3x3 / 1x1 convolution over padded NHWC fp16 activations, or the
// head projection (MODE linear). See DESIGN.md "conv_mfma" for the tiling.
struct ConvArgs {
    const half_t* x;        // input base: padded NHWC [B][H+2][Wa][Cin] (conv) or [M][Cin] (linear)
    const half_t* w;        // [taps][CoutPad][Cin], rows permuted to MFMA order within 64-blocks
    const float* bias;      // [CoutPad] folded BatchNorm bias
    void* y;                // output base (fp16, or fp32 for the head)
    float* se_part;         // [B][tilesH*tilesW][Cout] per-tile channel sums of the STORED (fp16-rounded,
                            // post-ReLU, column-masked) outputs, or nullptr
    const float* se_scale;  // fused squeeze-excite: [B][Cout] channel scales, or nullptr
    const half_t* resid;    // fused residual (same geometry as the output), used with se_scale
    int H, W;               // conv output rows / valid columns (pre-pool)
    int Cin, Cout, CoutPad;
    int tilesW, tilesH;     // pixel tiles per image
    int64_t in_sb;          // input batch stride (elements)
    int in_sh;              // input row stride (elements); pixel stride is Cin
    int64_t out_sb;         // output batch stride (elements)
    int out_sh, out_sw;     // output row / column strides (elements); row index is post-pool
    int64_t out_off;        // element offset of output pixel (b=0,h=0,w=0)
    int out_wlimit;         // columns [W, out_wlimit) are written as zeros (keeps the zero border)
    int relu, pool;
    int64_t M;              // linear mode: number of rows
    int64_t ldo;            // linear mode: output leading dimension (elements)
    int mtiles, ntiles;
    // halo4 kernel prologue without runtime divisions (set by launch_conv): ntiles and tilesH as shifts when they are
    // powers of two (-1 otherwise: the kernel divides), mt / tilesW as (mt * tw_magic) >> 40 (exact for mt * tilesW < 2^40)
    int nt_shift, th_shift;
    uint64_t tw_magic;
    // linear mode, fused greedy argmax (np.argmax(preds, 2), utils/ctc_codec.py:75): when amax_idx is set the
    // logits are NOT stored; every wave column (nt, wn) writes its best (value, class) of each row to
    // amax_val/amax_idx[(nt*WN + wn) * M + m]; launch_argmax_partials picks the first maximum per row.
    float* amax_val;
    int32_t* amax_idx;
    // guarded precision (hctr_set_precision mode 2): when amax_val2 is set the same epilogue also writes, per
    // (part, row), the RUNNER-UP value of the part (amax_val2, -inf for a one-class part) and the part's largest
    // |logit| (amax_abs); launch_argmax_partials turns them into the row's top-1/top-2 margin and max |logit|.
    float* amax_val2;
    float* amax_abs;
    // linear mode, fused beam front end (log_softmax + top-k + p > 0.001 lists, utils/ctc_codec.py:65,127-128,144,186)
    // without storing the logits. PASS 1 = the argmax partials above plus, when psum is set, per (part, row) the sum of
    // expf(v - part max) and the logit of class 0 per row. launch_beam_thresholds turns those into per-row
    // {vmin, gmax}: every logit >= vmin is among the row's top-k or above the candidate threshold.
    // PASS 2 (emit_cnt set; the same GEMM again, 2.5 % of a forward) appends every (class, logit >= vmin) to the
    // row's list (atomic slot counter, emit_cap slots) and writes per (part, row) the float64 sum of
    // expf(v - gmax) - the same terms the unfused row_topk kernel sums. launch_beam_select finishes the rows.
    float* psum;
    float* blank_logit;
    const float* row_thr;     // [M][2] = {vmin, gmax}
    int32_t* emit_cnt;        // [M], zeroed by launch_beam_thresholds
    int32_t* emit_list;       // [M][emit_cap][2] = {class, float bits}
    int emit_cap;
    double* esum;             // [P][M]
    // launch_stem_conv0_2 only: the raw line images (u8 [B][128][W] or f32 [B][1][128][W]), each line's valid width
    // (NULL = W) and conv0_1's folded fp32 weights [64][9] / bias [64]; x is unused there
    const void* img;
    int img_f32;
    const int32_t* img_widths;
    const float* stem_w;
    const float* stem_b;
    int split;              // f16x3 mode: activations are [hi | lo | hi] fp16 planes of Cout channels each
                            // (input side: Cin already counts the tripled channels)
    int rpre;               // halo4 kernel, conv2 of an identity block: fetch the residual during the last K step (HCTR_RPRE)
    int rtouch;             // halo4 kernel, conv2 of an identity block: pre-touch the residual tile's cache lines (HCTR_RTOUCH)
    int drop_lo;            // f16x3 diagnostic (HCTR_X3_MASK, precision attribution): round this layer's output to ONE fp16
                            // value like the f16 mode does (the lo plane is written as zeros)
    // fused 1x1 downsample of a block's input (first block of stages 1-3): the halo4 kernel first accumulates
    // ds_w * ds_x (centre tap, ds_cin channels, same H/W/Wa geometry as x), turns it into the residual term and
    // continues with the 3x3 taps; resid must then be NULL and se_scale set. NULL = not used.
    const half_t* ds_x;
    const half_t* ds_w;       // [1][CoutPad][ds_cin], rows permuted like w
    const float* ds_bias;     // folded BN of the downsample
    int ds_cin;
    int ds_in_sh;             // elements per image row of ds_x
    int64_t ds_in_sb;         // elements per image of ds_x
    int32_t* tile_counter;        // persistent halo4 variant with a dynamic tile queue: 8 ints (one per XCD), zero at launch
    unsigned long long* stamps;   // diagnostic instance only (hctr_debug_stamps): 16 x u64 per workgroup, else NULL
    // timing experiments only (HCTR_DBG), results INVALID. 8-wave/generic kernels: 1 = DMA from fixed hot addresses,
    // 2 = no DMA in the loop. halo4 kernel, bit mask: 32 = no halo reload at chunk boundaries, 64 = no K loop,
    // 128 = no epilogue, 256 = no output stores, 512 = no weight DMA inside the K loop, 1024 = no per-step barrier.
    int dbg;
};

// tile configurations: (couts x pixels) per 256-thread block
// TILE_HALO4 / TILE_HALO4_8x32: 128 couts x (16x16 | 8x32) pixels, 4 waves, halo-reuse 3x3 kernel
// (two workgroups per CU)
enum ConvTile { TILE_64x256 = 0, TILE_128x128 = 1, TILE_256x256 = 2, TILE_HALO4 = 3, TILE_HALO4_8x32 = 4 };
inline int conv_tile_rows(ConvTile t) { return (t == TILE_128x128 || t == TILE_HALO4_8x32) ? 8 : 16; }
inline int conv_tile_cols(ConvTile t) { return t == TILE_HALO4_8x32 ? 32 : 16; }
inline int conv_tile_couts(ConvTile t) {
    return t == TILE_64x256 ? 64 : (t == TILE_256x256 ? 256 : 128);
}

hipError_t launch_conv(const ConvArgs& a, ConvTile tile, int taps, bool linear_f32, hipStream_t s);
constexpr int kLinearWN = 2;          // wave columns of both linear-mode tiles (partials per n-tile)
// rows [P][M] of (value, class) partials -> idx[M]: first maximum, 0 for an all-(-inf)/NaN row (idx may be NULL).
// With val2 / absp (the guard partials, see ConvArgs) also margin[M] = top-1 minus top-2 logit of the row over all
// classes (0 for an exact tie, NaN when a NaN is involved) and rowabs[M] = max |logit| of the row.
hipError_t launch_argmax_partials(const float* val, const int32_t* cls, int P, int64_t M, int32_t* idx, hipStream_t s,
                                  const float* val2 = nullptr, const float* absp = nullptr, float* margin = nullptr,
                                  float* rowabs = nullptr);
// the same two per-row figures from stored logits rows [M][ld] (hctr_forward_logits in guarded mode)
hipError_t launch_row_guard(const float* logits, int64_t ld, int64_t M, int C, float* margin, float* rowabs, hipStream_t s);
// per line b: out[b][0] = min over its W columns of margin (NaN if any is NaN), out[b][1] = max of rowabs
hipError_t launch_line_guard(const float* margin, const float* rowabs, int B, int W, float* out, hipStream_t s);
size_t conv_lds_bytes(ConvTile tile);

// NormalizePAD + conv0_1 + bn0_1 + ReLU computed into the LDS halo of conv0_2's tile, then conv0_2 + bn0_2 + ReLU +
// (2,1) max-pool from it: conv0_1's 16 kB-per-column output never exists in HBM (f16 mode; a.w/bias = conv0_2's)
hipError_t launch_stem_conv0_2(const ConvArgs& a, hipStream_t s);

// zero the stored conv border of a padded NHWC activation [B][H+2][Wa][C]: rows 0 and H+1, column 0 and columns
// > W of every other row (the interior is rewritten by every forward; see engine.cpp ensure_workspace)
hipError_t launch_zero_borders(half_t* p, int B, int H, int W, int Wa, int C, hipStream_t s);

// split: 0 = fp16 output, 1 = [hi | lo | hi] planes, 2 = planes with the lo plane zero (HCTR_X3_MASK diagnostic)
hipError_t launch_stem(const void* img, int img_f32, const int32_t* widths_dev, const float* w9,
                       const float* bias, half_t* y, int B, int W, int Wa, int split, hipStream_t s);

// statistics of a padded NHWC activation t for the fused squeeze-excite: out[b][5][8 segments][C] = channel sums of
// {row 0, row H-1, column 0, column W-1, the whole image}; the whole-image sums come from conv1's per-tile sums
hipError_t launch_se_border(const half_t* t, int B, int H, int W, int Wa, int C, int split, const float* tsum_part,
                            int tiles, float* out, hipStream_t s);
// SE mean of conv2(t) from those statistics (linearity of the convolution) and, by the last block of each image,
// the SELayer FCs -> channel scales; counter: int32 [B], zero before the first launch (the kernel re-zeroes it)
hipError_t launch_se_premean(const float* border, const half_t* t, const half_t* w, const float* bias, int B, int H,
                             int W, int Wa, int C, int CoutPad, int split, float* mean, const float* w1,
                             const float* w2, float* scale, int32_t* counter, hipStream_t s);

hipError_t launch_se_fc(const float* se_part, int tiles_per_img, const float* w1, const float* w2,
                        float* scale, int B, int C, float inv_hw, hipStream_t s);

hipError_t launch_se_apply(half_t* o, const half_t* r, const float* scale, int64_t img_elems,
                           int B, int C, hipStream_t s);

// tB > 0: rows are in WBC order (r = t*tB + b) and idx is written as [b][t] with row length tW
hipError_t launch_argmax_rows(const float* logits, int64_t ld, int64_t M, int C, int32_t* idx,
                              int tB, int tW, hipStream_t s);

// ---- fused beam front end (see ConvArgs): P = ntiles * kLinearWN class parts per row, rows m = b*W + t ----
constexpr int kBeamCap = 256;         // list slots per row; a row that needs more makes the engine fall back
constexpr int kBeamMaxK = 32;         // largest top-k the fused path serves
// after pass 1: row_thr[m] = {min(k-th largest part maximum, candidate value bound), row max}; emit_cnt[m] = 0.
// cand_thresh: log-prob threshold of the candidate lists (ln 0.001) or +inf when they are not wanted.
hipError_t launch_beam_thresholds(const float* pmax, const float* psum, int P, int64_t M, int k, double cand_thresh,
                                  int want_candidates, float* row_thr, int32_t* emit_cnt, hipStream_t s);
// after pass 2: exact float32 log-softmax terms of the listed classes (row max, float64 exp-sum as row_topk computes
// them), top-k by (log-prob desc, class asc), log-prob of class 0, count of classes above cand_thresh; outputs are
// indexed r = t*B + b like launch_row_topk's. overflow[0] is set to 1 if any row needed more than cap slots.
hipError_t launch_beam_select(const float* row_thr, const int32_t* emit_cnt, const int32_t* emit_list, int cap,
                              const double* esum, int P, const float* blank_logit, int B, int W, int k,
                              double cand_thresh, int32_t* topk_idx, float* topk_logp, float* blank_logp,
                              float* stats, int32_t* cand_count, int32_t* overflow, hipStream_t s);
// candidate lists (ascending class order, log-prob > cand_thresh) out of the per-row lists, to cand_off[r] (r = t*B + b)
hipError_t launch_beam_candidates(const int32_t* emit_cnt, const int32_t* emit_list, int cap, const float* stats,
                                  int B, int W, double cand_thresh, const int64_t* cand_off, int32_t* cand_idx,
                                  float* cand_logp, hipStream_t s);

// raw per-column indices [B][W] (row m = b*W + t) -> collapsed labels [B][W] + lengths [B]
hipError_t launch_ctc_collapse(const int32_t* idx, int B, int W, int C, int32_t* labels,
                               int32_t* lengths, hipStream_t s);

// sub-batch rows [B*W][ld] -> out[t][b0 + b][C] of a [W][Bfull][C] tensor
hipError_t launch_logits_to_wbc(const float* logits, int64_t ld, int B, int W, int C, float* out,
                                int Bfull, int b0, hipStream_t s);
// [W][B][C] -> [B*W][ld] (for caller-supplied logits)
hipError_t launch_wbc_to_rows(const float* wbc, int B, int W, int C, float* rows, int64_t ld,
                              hipStream_t s);

// log-softmax + top-k (+ count of candidates above thresh) per (t,b) row of [B*W][ld]; outputs are
// indexed r = t*B + b. stats receives (row max, log-sum) pairs for launch_row_candidates.
hipError_t launch_row_topk(const float* logits, int64_t ld, int B, int W, int C, int k, double thresh,
                           int32_t* topk_idx, float* topk_logp, float* blank_logp, float* stats,
                           int32_t* cand_count, hipStream_t s);
hipError_t launch_row_candidates(const float* logits, int64_t ld, int B, int W, int C, double thresh,
                                 const float* stats, const int64_t* cand_off, int32_t* cand_idx,
                                 float* cand_logp, hipStream_t s);

// contiguous rows [rows][C] -> float32 log-softmax per row
hipError_t launch_log_softmax_rows(const float* x, int64_t rows, int C, float* y, hipStream_t s);

// ---- line preprocessing (preprocess.hip): cv2.resize(..., INTER_AREA) of ragged u8 images to height out_h ----
struct ResizeLine {
    int64_t src_off;          // byte offset of the image inside the packed source buffer
    int32_t sh, sw;           // source height, width
    int32_t ch;               // 1 = gray, 3 = BGR (cv2.imread order), -3 = RGB (PIL order)
    int32_t dw;               // destination width (columns >= dw of the line's output row are zeroed)
    int32_t mode;             // 0 area, 1 integer decimation (ix, iy), 2 enlarging bilinear variant
    int32_t ix, iy;
    double scale_x, scale_y;  // 1 / inv_*
    double inv_x, inv_y;      // dw / sw, out_h / sh
};
hipError_t launch_resize_lines(const uint8_t* packed, const ResizeLine* lines, int n, uint8_t* out, int out_h, int out_w,
                               hipStream_t s);

}  // namespace hctr
