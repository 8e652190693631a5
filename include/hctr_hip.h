/*
 * hctr_hip.h - C ABI of libhctr_hip.so, the MI355X (gfx950) engine for the reference's
 * hctr CNN+CTC inference path.
 *
 * The reference (AndrewCullacino/handwritten-chinese-ocr-samples) has no FFI layer: the seam
 * for this path is two Python classes plus a checkpoint dict. Each entry point
 * below names the reference interface it stands behind (file:line relative to the reference
 * root). The Python shims in handwritten-chinese-ocr-samples_amd/{model,codec}.py bind these
 * with ctypes and keep the reference's class/method signatures; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; the caller owns every host buffer for the duration of a call;
 *     the library owns device memory, workspace and the context;
 *   - every function returns 0 (HCTR_OK) or a negative hctr_status; the message for the last
 *     failure on a context is hctr_last_error(ctx); no C++ exception crosses the ABI;
 *   - a context is bound to one device and one HIP stream and is NOT re-entrant; use one context
 *     per thread/GPU. Calls are synchronous at return unless documented otherwise;
 *   - "WBC" = [W][B][C] row-major float32, the layout hctr_model.forward returns
 *     (models/handwritten_ctr_model.py:171-178).
 */
#ifndef HCTR_HIP_H
#define HCTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hctr_ctx hctr_ctx;

typedef enum {
    HCTR_OK = 0,
    HCTR_ERR_ARG = -1,        /* bad argument (-> ValueError) */
    HCTR_ERR_HIP = -2,        /* HIP runtime failure, message has file:line (-> RuntimeError) */
    HCTR_ERR_STATE = -3,      /* call order violated, e.g. forward before finalize (-> RuntimeError) */
    HCTR_ERR_KEY = -4,        /* unknown / missing state-dict key (-> KeyError, as load_state_dict strict) */
    HCTR_ERR_SHAPE = -5,      /* tensor shape mismatch (-> RuntimeError, as load_state_dict) */
    HCTR_ERR_EMPTY_LINE = -6, /* beam search on a line whose greedy decode is empty, or whose beam set
                                 became empty: the reference raises IndexError (utils/ctc_codec.py:143,198,179) */
    HCTR_ERR_NOMEM = -7
} hctr_status;

typedef enum { HCTR_U8 = 0, HCTR_F32 = 1, HCTR_I64 = 2 } hctr_dtype;

/* ---- context: replaces hctr_model(num_classes) + .cuda(gpu) ---------------------------------
 * models/handwritten_ctr_model.py:156-169, test.py:143-148. num_classes = 1 + len(chars) + 1. */
int hctr_create(hctr_ctx** out, int device, int num_classes);
void hctr_destroy(hctr_ctx* ctx);
const char* hctr_last_error(const hctr_ctx* ctx);   /* ctx may be NULL: last create() failure */
const char* hctr_version(void);

/* ---- weight ingest: replaces model.load_state_dict(checkpoint['state_dict']) -----------------
 * test.py:152-153; key schema main.py:349-356 / SURVEY.md 8b (254 entries). Call once per entry
 * (any order), then hctr_finalize_weights, which checks completeness, folds eval-mode BatchNorm
 * (eps 1e-5) into the preceding conv, converts to the kernel layouts and uploads. fp32 tensors are
 * HCTR_F32; num_batches_tracked entries are HCTR_I64 and ignored. */
int hctr_load_tensor(hctr_ctx* ctx, const char* key, const void* host_ptr,
                     const int64_t* shape, int ndim, int dtype);
int hctr_finalize_weights(hctr_ctx* ctx);

/* ---- precision mode ----------------------------------------------------------------------------------
 * 0 = f16 (default): fp16 storage and MFMA inputs, fp32 accumulation - the 10-bit mantissa of the TF32
 *     mode the reference enables on its GPUs (main.py:37-41).
 * 1 = f16x3: every activation and weight is carried as a hi + lo fp16 pair and each product is formed
 *     as w_hi*x_hi + w_hi*x_lo + w_lo*x_hi in fp32 (about 3x the matrix work and memory); logits then
 *     agree with the fp32 CPU reference to ~1e-5 relative: its argmax / text is the reference's wherever
 *     two fp32 summation orders agree.
 * 2 = guarded ("auto"): every line runs in f16 and the head's fused epilogue also yields, per pixel column,
 *     the margin between the largest and second largest logit. A line is CERTAIN when every one of its
 *     columns (pad columns included - the reference decodes them too, utils/ctc_codec.py:75 over the padded
 *     batch of test.py:170-186) has margin > 2 * (rel * max|logit of the line| + abs): no error within the
 *     f16 logit tolerance rel * scale + abs can then change an argmax. Every other line is run again in
 *     f16x3 at the same padded width (a line's result depends only on its own pixels and that width) and
 *     its results replace the f16 ones - in hctr_greedy, hctr_forward_logits and hctr_beam_frontend alike.
 *     Text is then the f16x3 mode's wherever f16 cannot be trusted, at f16 speed for lines with peaky
 *     logits. Both weight sets stay resident (about 0.42 GB).
 * Before hctr_finalize_weights the mode decides which weight set(s) are built (0: f16, 1: f16x3, 2: both);
 * afterwards it may still be changed among the modes whose set(s) are resident (a context finalized in
 * mode 2 serves all three).
 * hctr_set_guard: rel / abs of mode 2's criterion (defaults 0.01 / 0.05 = the f16 logit tolerance the parity
 *     suite asserts, tests/test_gpu_parity.py LOGIT_RTOL / LOGIT_ATOL).
 * hctr_last_guard: figures of the last mode-2 call, per line of its batch: flags[i] = 1 if line i was
 *     run again in f16x3, min_margin[i] = its smallest column margin, scale[i] = its largest |logit| (f16 sweep).
 *     Any output pointer may be NULL; at most cap entries are written. */
int hctr_set_precision(hctr_ctx* ctx, int mode);
int hctr_set_guard(hctr_ctx* ctx, double rel, double abs_tol);
int hctr_last_guard(hctr_ctx* ctx, int64_t* lines, int64_t* flagged, uint8_t* flags, float* min_margin,
                    float* scale, int64_t cap);

/* ---- forward: replaces hctr_model.forward --------------------------------------------------
 * models/handwritten_ctr_model.py:171-178 (trunk :115-153). Input: a batch of B line images of
 * height 128 and common width W, either HCTR_F32 [B][1][128][W] already normalised to [-1,1]
 * (what test.py:179-193 feeds), or HCTR_U8 [B][128][W] raw grey levels, in which case the engine
 * applies NormalizePAD itself (utils/dataset.py:83-93): x/255, (x-0.5)/0.5, and columns
 * >= widths[b] replicate column widths[b]-1 (widths may be NULL = all W).
 * img_on_device / out_on_device: the pointer is a device pointer on ctx's device.
 * Output: float32 logits in WBC layout, B*W*num_classes values. */
int hctr_forward_logits(hctr_ctx* ctx, const void* img, int img_dtype, int img_on_device,
                        const int32_t* widths, int B, int W,
                        float* out_wbc, int out_on_device);

/* ---- fused forward + greedy decode: replaces model(x) -> codec.decode(...) greedy ------------
 * test.py:191-194 with utils/ctc_codec.py:70-99. Logits never leave the device; argmax takes the
 * first maximum (np.argmax), a column is kept iff idx!=0 && idx!=C-1 && idx!=previous raw idx.
 * labels: int32 [B][W] (first lengths[b] entries valid), lengths: int32 [B]; host pointers. */
int hctr_greedy(hctr_ctx* ctx, const void* img, int img_dtype, int img_on_device,
                const int32_t* widths, int B, int W,
                int32_t* labels, int32_t* lengths);

/* ---- decode of caller-supplied logits: replaces ctc_codec.decode(ndarray) greedy -------------
 * utils/ctc_codec.py:63-99. logits: float32 WBC with C classes (host or device pointer). */
int hctr_decode_greedy_logits(hctr_ctx* ctx, const float* logits_wbc, int on_device,
                              int W, int B, int C, int32_t* labels, int32_t* lengths);

/* ---- beam-search front end on the device -----------------------------------------------------
 * utils/ctc_codec.py:65 (log_softmax), :127/:186 (top search_depth by descending log-prob), :128,144
 * (candidates with log-prob > ln 0.001 for the "skip" variant).
 * Runs the forward on img (or, when img == NULL, takes caller logits in WBC layout), then
 * log-softmax and top-k per (t, b), and copies to the host buffers
 *   topk_idx  int32 [W][B][k], topk_logp float32 [W][B][k]   (descending; ties: lower index first)
 *   blank_logp float32 [W][B]                                 (log-prob of class 0).
 * With want_candidates != 0 it also builds, per (t, b), the ascending list of classes whose
 * log-prob exceeds ln(0.001); the context keeps those lists until hctr_beam_fetch_candidates copies
 * them out: cand_off int64 [W*B+1] (CSR, row r = t*B + b), cand_idx int32 / cand_logp float32 of
 * *num_candidates entries. */
int hctr_beam_frontend(hctr_ctx* ctx, const void* img, int img_dtype, int img_on_device,
                       const int32_t* widths, const float* logits_wbc, int logits_on_device,
                       int B, int W, int C, int k, int want_candidates,
                       int32_t* topk_idx, float* topk_logp, float* blank_logp,
                       int64_t* num_candidates);
int hctr_beam_fetch_candidates(hctr_ctx* ctx, int64_t* cand_off, int32_t* cand_idx, float* cand_logp);

/* ---- log-softmax of caller logits: replaces scipy.special.log_softmax(preds, axis=2) ------------
 * utils/ctc_codec.py:65. float32 WBC in (host or device), float32 WBC out (host). Only needed when
 * the language model proposes candidates (use_tfm_pred), because those can be any class. */
int hctr_log_softmax(hctr_ctx* ctx, const float* logits_wbc, int on_device, int W, int B, int C,
                     float* out_host);

/* ---- host prefix beam search: replaces ctc_codec.__cbs_full__/__cbs_skip__ -------------------
 * utils/ctc_codec.py:124-285 (Beam :288-307), float64 accumulators over float32 log-probs.
 * The language model stays behind callbacks, as in the reference (kenlm / transformer objects are
 * duck-typed there, utils/ctc_codec.py:216-219,269-281):
 *   score_cb: called once per time step with n sentences (label ids of prefix+suffix, CSR in
 *             ids/offs); must fill scores[n]. Replaces ngram.score(' '.join(chars), eos=False)
 *             / transformer.score(batch, char_based=True).
 *   next_cb:  optional (use_tfm_pred): for n beam prefixes fill out_ids[n][k] with the LM's next labels
 *             (utils/ctc_codec.py:216-227; k = search_depth on the first call of a step). The reference chains
 *             whatever list the LM returns (:225-226): pad a shorter list with the <unknown> id C-1 (skipped, :238-239)
 *             and return 0; if some list is LONGER than k return the number of slots needed (> k) and the search
 *             calls again with that k. Negative = failure (propagated). NULL disables.
 * builtin_lm: 0 = callbacks, 1 = zero LM, 2 = toy hashed bigram over code points (needs
 *             label_codepoints[C]), 3 = ARPA n-gram (needs ngram + label_words[C]); built-ins make
 *             the multi-threaded path callback-free.
 * full_logp_wbc: float32 [W][B][C] log-probs, required only with next_cb (LM-proposed labels can be
 *             any class); NULL otherwise. cand_* are required only when skip_search != 0.
 * Per-line results: out_labels int32 [B][W] + out_lengths [B]. line_status[b] is HCTR_OK or
 * HCTR_ERR_EMPTY_LINE; the return value is the first non-OK line status. */
typedef int (*hctr_lm_score_cb)(void* user, int n, const int32_t* ids, const int32_t* offs, double* scores);
typedef int (*hctr_lm_next_cb)(void* user, int n, const int32_t* ids, const int32_t* offs, int k, int32_t* out_ids);

/* ---- ARPA back-off n-gram LM: stands in for kenlm.Model (utils/ctc_codec.py:121-122,276-281) ---
 * hctr_ngram_score == kenlm.Model.score(sentence, bos, eos): log10 probability of a whitespace-
 * separated UTF-8 sentence (Katz back-off, OOV -> <unk>). hctr_ngram_word_id maps a token to the
 * model's word id (-1 = out of vocabulary) so the beam search can score label sequences natively
 * (hctr_beam_params.builtin_lm == 3, .ngram, .label_words). Parity with the kenlm binary: unpinned. */
typedef struct hctr_ngram hctr_ngram;
int hctr_ngram_load(const char* arpa_path, hctr_ngram** out);
void hctr_ngram_free(hctr_ngram* lm);
int hctr_ngram_order(const hctr_ngram* lm);
int32_t hctr_ngram_word_id(const hctr_ngram* lm, const char* word_utf8);
double hctr_ngram_score(const hctr_ngram* lm, const char* sentence_utf8, int bos, int eos);
const char* hctr_ngram_last_error(void);

typedef struct {
    int skip_search;        /* utils/ctc_codec.py:40,66 */
    int beam_size;          /* :37 */
    int search_depth;       /* :36 */
    double lm_panelty;      /* :34 (spelling is the reference's) */
    double len_bonus;       /* :35 */
    int builtin_lm;
    const int32_t* label_codepoints;  /* [C] unicode code point per label, for builtin_lm == 2 */
    hctr_lm_score_cb score_cb;
    hctr_lm_next_cb next_cb;
    void* user;
    int num_threads;        /* lines in parallel; forced to 1 when callbacks are used */
    const hctr_ngram* ngram;          /* builtin_lm == 3: ARPA model ... */
    const int32_t* label_words;       /* ... and [C] LM word id per label (-1 = OOV) */
} hctr_beam_params;

int hctr_beam_search(const hctr_beam_params* p, int W, int B, int C, int k,
                     const int32_t* topk_idx, const float* topk_logp, const float* blank_logp,
                     const int64_t* cand_off, const int32_t* cand_idx, const float* cand_logp,
                     const float* full_logp_wbc,
                     int32_t* out_labels, int32_t* out_lengths, int32_t* line_status);

/* ---- line preprocessing on the device: replaces read_resize_image / pil_loader -----------------
 * test.py:207-216 (cv2.cvtColor BGR2GRAY + cv2.resize(src, (tw, 128), interpolation=cv2.INTER_AREA)) and
 * utils/dataset.py:47-60 (the same resize inside ImageDataset.pil_loader). Image files are decoded by
 * the caller; this call takes the decoded u8 pixels of n ragged images packed back to back in one host
 * buffer (image i: heights[i] x widths[i] x |channels[i]| bytes at offsets[i]; channels 1 = gray,
 * 3 = BGR as cv2.imread returns it, -3 = RGB as PIL returns it), converts colour to gray, resizes image i
 * to out_height x out_widths[i] and writes it into row i of out[n][out_height][out_W] (uint8). Columns
 * >= out_widths[i] are zero - hctr_greedy / hctr_forward_logits replicate the last real column from
 * `widths`, which is NormalizePAD, utils/dataset.py:83-93; columns >= out_W of a wider line are dropped,
 * which is AlignCollate's max_width crop, utils/dataset.py:118-145 (pass widths min(out_widths[i], out_W)
 * on). out may be device memory (out_on_device), so the resized batch feeds hctr_greedy without touching
 * the host again.
 * Pixel parity with OpenCV is unpinned (opencv-python is not installed here): the kernel follows the
 * published OpenCV 4.x algorithm as restated in oracle/resize_ref.py and is bit-exact against that. */
int hctr_resize_lines(hctr_ctx* ctx, const uint8_t* packed_src, int64_t packed_bytes, const int64_t* offsets,
                      const int32_t* heights, const int32_t* widths, const int32_t* channels, int n,
                      int out_height, const int32_t* out_widths, int out_W, uint8_t* out, int out_on_device);

/* ---- multi-GPU result gather for plain-C callers -----------------------------------
 * The path shards by lines: one process per GPU runs hctr_greedy on its contiguous range of the (globally padded)
 * batch, and the decoded label sequences meet in ONE collective. The reference has no counterpart (single-device
 * inference, test.py:143-148; NCCL only in training DDP, main.py:226-237). Python callers use torch.distributed
 * (handwritten-chinese-ocr-samples_amd/dist.py); these entry points do the same over RCCL (xGMI), which is bound
 * with dlopen at the first call. Protocol as in NCCL: rank 0 calls hctr_comm_unique_id and hands the 128 bytes to
 * the other ranks out of band (file, socket, MPI); every rank then calls hctr_comm_create.
 * hctr_gather_labels: every rank passes its n_local decoded lines (labels: int32 [n_local][row_stride], the first
 * lengths[i] entries of row i valid - exactly hctr_greedy's outputs with row_stride = W) and the common
 * lines_per_rank = ceil(global lines / world) and cap (longest label sequence allowed). One ncclAllGather of the
 * packed [lines_per_rank][1 + cap] int32 buffer; on return EVERY rank holds out_labels int32 [world*lines_per_rank][cap]
 * and out_lengths [world*lines_per_rank] in rank order (rows >= a rank's n_local have length 0). */
#define HCTR_COMM_ID_BYTES 128
typedef struct hctr_comm hctr_comm;
int hctr_comm_unique_id(void* id128);
int hctr_comm_create(hctr_comm** out, const void* id128, int rank, int world, int device);
void hctr_comm_destroy(hctr_comm* comm);
int hctr_gather_labels(hctr_comm* comm, const int32_t* labels, const int32_t* lengths, int n_local, int row_stride,
                       int lines_per_rank, int cap, int32_t* out_labels, int32_t* out_lengths);
const char* hctr_comm_last_error(void);

/* ---- introspection used by bench.py / tests ---------------------------------------------------
 * Per-layer device time of the last forward (HIP events on the context's stream), in call order.
 * names: '\n'-separated layer names written into buf (cap bytes); ms: float array of n entries.
 * Returns the number of layers recorded (or a negative status). Enabled by hctr_set_profiling. */
int hctr_set_profiling(hctr_ctx* ctx, int enabled);
int hctr_last_profile(hctr_ctx* ctx, char* names_buf, int cap, float* ms, int max_n);
/* Workspace arena of the context: bytes currently allocated, how often it was (re)allocated - it only ever grows, to
 * the largest layout seen - and how often its layout was re-carved for a new (lines, width, precision) shape. */
int hctr_workspace_stats(hctr_ctx* ctx, int64_t* arena_bytes, int64_t* arena_allocations, int64_t* recarves);
/* Lines per internal pass for a batch of B lines of width W (a batch beyond HCTR_MAX_COLS pixel columns - a third
 * of that in f16x3 - runs in balanced passes; hctr_last_profile adds the passes' entries of one name up). */
int hctr_lines_per_pass(hctr_ctx* ctx, int B, int W, int f16x3);
/* Diagnostic build of the 3x3 conv kernel: with layer != NULL, arms time-stamping of that layer's
 * workgroups (a separate kernel instance; results of the forward are unchanged) for up to cap_wgs
 * workgroups and returns 0. With layer == NULL copies the last forward's stamps to out[n][16]
 * (u64: entry, prologue issued, operands landed, K loop done, epilogue done, stores drained - 100 MHz
 * ticks - HW_ID and XCC_ID registers, then four stamps inside the epilogue) and returns n. tools/gpu_stamps.py is the consumer. */
int64_t hctr_debug_stamps(hctr_ctx* ctx, const char* layer, uint64_t* out, int64_t cap_wgs);

/* Debug taps for bisecting parity: copy an intermediate activation of the last forward to the host
 * as float32 NCHW [B][C][H][W]. name: "stage0".."stage4", "conv0_1". Returns element count or <0. */
int64_t hctr_debug_activation(hctr_ctx* ctx, const char* name, float* out, int64_t cap,
                              int* C, int* H);

#ifdef __cplusplus
}
#endif
#endif /* HCTR_HIP_H */
