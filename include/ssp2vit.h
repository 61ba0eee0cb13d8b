/*
 * libssp2vit — C ABI of the MI355X-native 2SSP-for-ViT hot path.
 *
 * The reference (zvezdvv/2ssp-X-vit) has NO native boundary: its plug-in surface is Python
 * (SURVEY.md §8b).  Each entry point below names the reference interface whose device work it
 * replaces; the Python host layer (2ssp-x-vit_amd/ssp2vit) re-exposes the reference's own function
 * names on top of these calls, and INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions: every function returns 0 on success, a negative SSP2_E* code otherwise
 * (ssp2_last_error() gives the message).  All pointers named *_dev are device (HBM) pointers owned by
 * the caller; nothing here allocates or frees caller memory.  All work is enqueued on the handle's HIP
 * stream (ssp2_set_stream; default = the NULL stream) and is asynchronous unless stated.  A handle is
 * thread-compatible, not thread-safe.  No torch types appear anywhere.
 */
#ifndef SSP2VIT_H
#define SSP2VIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSP2_ABI_VERSION 5

enum {
  SSP2_OK = 0,
  SSP2_EINVAL = -1,      /* bad argument / shape the kernels do not support   */
  SSP2_EHIP = -2,        /* a HIP runtime call failed                          */
  SSP2_ENOMEM = -3,
  SSP2_ESTATE = -4       /* weights missing, capacity exceeded, ...            */
};

/* tensor kinds for ssp2_load_tensor (layer index ignored for the model-level ones) */
enum {
  SSP2_T_PATCH_W = 0, SSP2_T_PATCH_B, SSP2_T_CLS, SSP2_T_POS,
  SSP2_T_LN1_G, SSP2_T_LN1_B, SSP2_T_QKV_W, SSP2_T_QKV_B, SSP2_T_PROJ_W, SSP2_T_PROJ_B,
  SSP2_T_LN2_G, SSP2_T_LN2_B, SSP2_T_FC1_W, SSP2_T_FC1_B, SSP2_T_FC2_W, SSP2_T_FC2_B,
  SSP2_T_LNF_G, SSP2_T_LNF_B, SSP2_T_HEAD_W, SSP2_T_HEAD_B,
  SSP2_T_COUNT
};

/* where the stage-1 score is read (reference hook site, src/vit_pruning.py:130 vs :135) */
enum { SSP2_SCORE_NONE = 0, SSP2_SCORE_PRE_GELU = 1 /* timm: fc1 out */, SSP2_SCORE_POST_GELU = 2 /* HF: intermediate out */ };
/* OR-ed into score_site: the caller wants the scores only — x_dev is scratch after the call.  The last block of the range then
 * stops behind its hooked activation (its fc2 + residual, whose result nothing reads, is not run).  The reference runs the whole
 * model and discards the logits (src/vit_pruning.py:180); the scores are the same bits either way. */
#define SSP2_SCORE_ONLY 0x10

/* score arithmetic: fp32 accumulators end to end, or the reference's CPU-autocast bf16 rounding points
 * (src/vit_pruning.py:151-157: per-(sample,neuron) norm -> bf16, batch sum -> bf16)                     */
enum { SSP2_CHAIN_FP32 = 0, SSP2_CHAIN_BF16_REF = 1 };

typedef struct ssp2_engine* ssp2_handle;

typedef struct {
  int32_t img;          /* input resolution (square)                               */
  int32_t patch;        /* patch size; tokens N = (img/patch)^2 + 1                */
  int32_t dim;          /* hidden size, multiple of 64                             */
  int32_t heads;        /* dim/heads in {16, 64, 80}; tokens = (img/patch)^2 + 1 must fall into an instantiated key-tile count:
                         * d_h 64: up to 288 tokens (ViT-Ti/S/B/L at /16 and /32 up to 256 pixels; ViT-L/14); d_h 80: 257..288 (ViT-H/14);
                         * d_h 16: up to 32 — anything else is refused by ssp2_create with SSP2_EINVAL */
  int32_t depth;        /* encoder blocks L                                        */
  int32_t classes;
  float   ln_eps;       /* 1e-6 timm, 1e-12 HF                                     */
  int32_t max_images;   /* capacity of one forward call (workspace is sized once)  */
  const int32_t* d_int; /* [depth] FFN width per block (differs after width prune) */
} ssp2_vit_desc;

int ssp2_abi_version(void);
const char* ssp2_last_error(void);

/* Replaces: model construction + `.to(device)` of the third-party ViT the reference calls
 * (src/vit_pruning.py:180, :354).  Allocates bf16 weight storage and the activation workspace. */
int ssp2_create(const ssp2_vit_desc* desc, ssp2_handle* out);
int ssp2_destroy(ssp2_handle h);
int ssp2_set_stream(ssp2_handle h, void* hip_stream);
/* Cap the grids of the handle's persistent kernels (large-tile GEMMs, d_h = 64 attention) at n_cu workgroups (<= 0 or
 * more than the device has: all CUs).  Two handles on two streams with half the CUs each run side by side instead of
 * taking turns — the point of bench.py --two-streams. */
int ssp2_set_cu_limit(ssp2_handle h, int n_cu);
/* Run-time switches of a handle (kernel routing and launch order; NONE of them changes a result bit — the GPU suite flips
 * each one and compares).  Defaults come from the environment variable named beside each, read once in ssp2_create.
 *   SSP2_OPT_ZIGZAG         1  SSP2_ZIGZAG          large launches walk their row panels opposite to the previous launch
 *   SSP2_OPT_ATTN_PERSIST   1  SSP2_ATTN_PERSIST    d_h = 64 / 80 attention on the persistent producer / consumer kernel
 * (LAB-BUILD ONLY — lib/libssp2vit_lab.so, -DSSP2_LAB; the product library refuses a non-zero value with SSP2_ESTATE: SSP2_OPT_LN_FUSION,
 *  SSP2_OPT_GROUP256, SSP2_OPT_DEFER_RESID.  All three were measured slower than the default and are kept for their bit-identity tests.)
 *   SSP2_OPT_LN_FUSION      0  SSP2_LN_FUSION       LayerNorm inside the residual GEMM (the panel's last-arriving workgroup normalises it): 0 off, 1 / 2 on for every eligible launch
 *   SSP2_OPT_BIG_TILES      1  SSP2_NO_BIG_TILES    launches with >= 4096 rows on the persistent 256 x 256 GEMM
 *   SSP2_OPT_FC1_BIG_TILES  1  SSP2_FC1_SMALL_TILES fc1 of such launches too
 *   SSP2_OPT_GROUP256       0  SSP2_GROUP256        tile order of the 256 x 256 GEMM: 0 = N fastest, 100 * GM + GN = column groups
 *   SSP2_OPT_PATCH_LDS      1  SSP2_PATCH_LDS       patch embed stages its patch tiles pixels -> LDS inside the GEMM kernel (0: im2col image in HBM)
 *   SSP2_OPT_ATTN_STAGGER   0  SSP2_ATTN_STAGGER    persistent attention: waves 4.. start their tile this many x 64 cycles behind their SIMD partners
 *   SSP2_OPT_FP8_PROJ       1  SSP2_FP8_PROJ        fp8 mode only (the one switch that changes results, within that mode's tolerance): the attention
 *                                                   output is written as e4m3 and the out-projection runs on e4m3 operands too
 *   SSP2_OPT_BIG_TILE_MIN_ROWS 4096 SSP2_BIG_TILE_MIN_ROWS  launches with at least this many token rows go to the persistent 256 x 256 GEMM (and, in
 *                                                   fp8 mode, to e4m3 operands); >= 256.  The default is the measured break-even; the GPU suite lowers
 *                                                   it so that test-sized launches meet the large kernel
 *   SSP2_OPT_NT_STORES      1  SSP2_NT_STORES       the 256 x 256 GEMM writes its bf16 / e4m3 activation outputs (QKV, fc1) with non-temporal stores
 *   SSP2_OPT_DEFER_RESID    0  SSP2_DEFER_RESID     residual projections on the 256 x 256 GEMM park bf16(acc + bias) and add it to x during the NEXT
 *                                                   tile's main loop (bf16 operands, dim a multiple of 256); same bits as the direct epilogue
 *   SSP2_OPT_ATTN_LIVE      1  SSP2_ATTN_LIVE       persistent attention at 197 / 257 tokens on the instantiation that knows at compile time that the
 *                                                   last key tile holds ONE live register group (5 / 1 valid keys): the padding's maxima,
 *                                                   exponentials, sums and P V half are not emitted; 0 = the general instantiation */
enum { SSP2_OPT_ZIGZAG = 0, SSP2_OPT_ATTN_PERSIST, SSP2_OPT_LN_FUSION, SSP2_OPT_BIG_TILES, SSP2_OPT_FC1_BIG_TILES, SSP2_OPT_GROUP256,
       SSP2_OPT_PATCH_LDS, SSP2_OPT_ATTN_STAGGER, SSP2_OPT_FP8_PROJ, SSP2_OPT_BIG_TILE_MIN_ROWS, SSP2_OPT_NT_STORES,
       SSP2_OPT_DEFER_RESID, SSP2_OPT_ATTN_LIVE, SSP2_OPT_COUNT };
int ssp2_set_option(ssp2_handle h, int option, int value);
int ssp2_get_option(ssp2_handle h, int option);                             /* >= 0, or SSP2_EINVAL */

/* fp32 HOST data in nn.Linear/Conv2d layout ([out,in], conv [dim,3,p,p]); matrices are rounded to bf16
 * (RNE) exactly as torch.autocast casts them, biases are rounded to bf16 and kept as fp32 values, LayerNorm
 * parameters / cls / pos stay fp32.  Synchronous. */
int ssp2_load_tensor(ssp2_handle h, int kind, int layer, const float* host, size_t numel);
/* The same from fp32 data that already lives in HBM (a module on the GPU, as the reference keeps it:
 * `model.to(device)`, adaptation-for-Pures-framework/auto_2ssp.py:693): no host copy, the bf16 rounding and the
 * padding run in a kernel on the handle's stream.  Asynchronous: dev_ptr must stay valid until the stream has passed
 * this point. */
int ssp2_load_tensor_dev(ssp2_handle h, int kind, int layer, const float* dev_ptr, size_t numel);
/* ABI 5: MANY device-resident tensors in one call — entry i is (kinds[i], layers[i], dev_ptrs[i], numels[i]) with the meaning of
 * ssp2_load_tensor_dev, and the bits it leaves are that call's.  This is how a live module is taken over (the reference hands the
 * plug-in an nn.Module whose ~150 parameters sit in HBM: `model.to(device)`, adaptation-for-Pures-framework/auto_2ssp.py:693; the
 * interface reads them, mask_conjunction.py:237-248): the engine issues ceil(count / 64) launches instead of `count`.  Every entry is
 * validated before anything is enqueued (SSP2_EINVAL names the first wrong one; the engine is then untouched).  Asynchronous on the
 * handle's stream: every dev_ptrs[i] must stay valid until the stream has passed this point; the four arrays are host memory and are
 * free to go when the call returns. */
int ssp2_load_tensors_dev(ssp2_handle h, int count, const int* kinds, const int* layers, const float* const* dev_ptrs, const size_t* numels);

/* Arithmetic of the three large projections (QKV, fc1, fc2) of launches with >= 4096 token rows.
 *   SSP2_PREC_BF16  (default) bf16 operands — the reference's CPU-autocast arithmetic, the parity mode.
 *   SSP2_PREC_FP8   opt-in, BASELINE.json configs[4]: OCP e4m3 operands on the CDNA4 scaled-MFMA path (2x the bf16 matrix
 *                   rate): weights quantised per output row (scale = amax / 448) from their bf16 image, activations cast
 *                   directly (LayerNorm output and GELU output are O(1)), fp32 accumulation, scale applied before the bias.
 *                   The reference has no fp8 arithmetic: this mode is checked by tolerance against the bf16 engine.
 * Call after the weights are loaded; synchronous with respect to later launches on the handle's stream. */
enum { SSP2_PREC_BF16 = 0, SSP2_PREC_FP8 = 1 };
int ssp2_set_precision(ssp2_handle h, int mode);
/* fp8 mode, optional CALIBRATION of the one hand-off that carries a fixed scale: the attention output reaches the e4m3 out-projection as
 * e4m3(o x s_l), s_l = 16 by default (fine for |o| <= 28; beyond that values clip — counted by SSP2_Q_FP8_SATURATED).
 *   ssp2_fp8_calibrate_begin(h)          from now on forwards run the out-projection on bf16 and record max |o| per block
 *   ... one or more ssp2_layers / forward calls over representative images (launches of >= SSP2_OPT_BIG_TILE_MIN_ROWS rows) ...
 *   ssp2_fp8_calibrate_end(h, headroom)  sets s_l = the largest power of two with  max|o|_l x headroom x s_l <= 448  for every block that
 *                                        was seen (2^-8 <= s_l <= 2^10; others keep 16), switches the recording off; synchronises
 *   ssp2_fp8_attn_scale(h, l)            the scale in force for block l (e.g. to store it with a checkpoint); ssp2_fp8_set_attn_scale
 *                                        sets it directly (a power of two > 0)
 * Explicit and deterministic: the scales depend only on the images handed over between begin and end, never on launch order. */
int ssp2_fp8_calibrate_begin(ssp2_handle h);
int ssp2_fp8_calibrate_end(ssp2_handle h, float headroom);
float ssp2_fp8_attn_scale(ssp2_handle h, int layer);                     /* > 0, or 0 on a bad argument */
int ssp2_fp8_set_attn_scale(ssp2_handle h, int layer, float scale);

/* Row layout of the token matrix x.  group <= 0 or >= n: images contiguous, ssp2_rows = n*N.  0 < group < n: SLABS of
 * `group` images (one dataloader batch each), every slab padded to a multiple of 128 rows (the hook's tile) — a sample then sits at the
 * same offset inside its slab whichever call it is part of, so per-tile partial sums (hence stage-1 scores) do not
 * depend on how many batches share a call.  ssp2_embed / ssp2_layers / ssp2_head of one forward take the same group. */
long ssp2_rows(ssp2_handle h, int n, int group);

/* a3 (patch-embed conv k=s=p + cls + pos):  pixels_dev f32 NCHW [n,3,img,img]  ->  x_dev f32 [ssp2_rows(n,group), dim] */
int ssp2_embed(ssp2_handle h, const float* pixels_dev, int n, float* x_dev, int group);

/* a3/a2/a6: run encoder blocks [l_begin, l_end) in place on the residual stream x_dev [n*N, dim].
 *   attn_skip      host bytes [depth] or NULL; attn_skip[l]!=0 == the reference's zero-output attention
 *                  bypass (src/vit_pruning.py:416-429): the block computes x <- x + MLP(LN2(x)) only.
 *   score_group    the row layout group of x (see ssp2_rows) AND the scoring group:
 *   score_site     SSP2_SCORE_*: if not NONE, the call's n samples are cut into groups of score_group
 *                  consecutive samples (= the reference dataloader's batches; the last group may be short) and
 *                  batch_scores_dev[g][l][:] receives sum_{s in group g} || act_l[s,:,j] ||_2 (the hook body,
 *                  :151-152), samples added in index order (deterministic, independent of launch geometry and
 *                  of how many batches share one call).
 *   batch_scores_dev f32 [ceil(n/score_group), depth, score_ld] (rows outside [l_begin,l_end) untouched),
 *                  score_ld >= max d_int;  score_group <= 0 means one group of n. */
int ssp2_layers(ssp2_handle h, float* x_dev, int n, int l_begin, int l_end, const uint8_t* attn_skip,
                int score_site, int score_chain, int score_group, float* batch_scores_dev, int score_ld);
/* The same with the residual stream ENTERING block l_begin read from x_in_dev (same shape and row layout as x_dev, not
 * overlapping it; NULL = x_dev): every LayerNorm in front of the first residual add and that add itself read x_in_dev, the add
 * writes x_dev, everything after it runs in place on x_dev.  x_in_dev is left untouched.  This is how the depth search starts
 * candidate l from the baseline's stream without copying it (reference: a deep copy of the whole model per candidate,
 * src/vit_pruning.py:477-480). */
int ssp2_layers_from(ssp2_handle h, const float* x_in_dev, float* x_dev, int n, int l_begin, int l_end, const uint8_t* attn_skip,
                     int score_site, int score_chain, int score_group, float* batch_scores_dev, int score_ld);

/* The same with the stage-1 hook on the LEADING score_images images of the launch only (score_images == n: ssp2_layers_from).
 * score_images < n needs the slab layout (0 < score_group < n) and a whole number of slabs.  This is how ONE dense pass serves both
 * stages: the reference's Auto2SSPInterface walks one loader with one batch_limit for the stage-1 scores and for the search's baseline
 * (adaptation-for-Pures-framework/mask_conjunction.py:276-281, :327, :345); here the baseline of the layer-major search sits in the
 * leading slabs of the launch and is hooked, the candidates' streams behind it are not.  Scores and streams are the bits of the
 * separate calls (a row's arithmetic does not depend on the launch it is part of; a slab's position pins its partial sums).
 *   batch_scores_dev f32 [ceil(score_images/score_group), depth, score_ld] */
int ssp2_layers_prefix(ssp2_handle h, const float* x_in_dev, float* x_dev, int n, int l_begin, int l_end, const uint8_t* attn_skip,
                       int score_site, int score_chain, int score_group, int score_images, float* batch_scores_dev, int score_ld);

/* a3 tail + a4: final LayerNorm on the CLS rows, classifier, first-max-index argmax (torch.argmax rule),
 * comparison with labels.  Any of logits_dev [n,classes] f32, pred_dev [n] i32, labels_dev [n] i64 +
 * correct_dev [1] i64 (ACCUMULATED into, caller zeroes) may be NULL. */
int ssp2_head(ssp2_handle h, const float* x_dev, int n, int group, float* logits_dev, int32_t* pred_dev,
              const int64_t* labels_dev, int64_t* correct_dev);

/* Evaluation tail (a4, every pass of the a5 search): the LAST encoder block followed by the head, computed for
 * the CLS rows only — keys/values are formed for every token, but the query, the attention output, the
 * out-projection and the whole MLP run on n rows instead of n*N.  Bit-identical to
 * ssp2_layers(x, depth-1, depth) + ssp2_head(x) (same per-row instruction sequences); x_dev is NOT modified and
 * must be in the contiguous layout (group 0).
 * attn_skip_last != 0 bypasses the last block's attention. */
int ssp2_tail(ssp2_handle h, const float* x_dev, int n, int attn_skip_last, float* logits_dev, int32_t* pred_dev,
              const int64_t* labels_dev, int64_t* correct_dev);
/* The same for `slots` residual streams of n_slot images each, side by side in x_dev (the baseline and the candidates of the
 * layer-major depth search: they all meet the same last block and classifier): one launch sequence over slots * n_slot images.
 * labels_dev [n_slot] is shared by the slots; correct_dev [slots] (int64, accumulated); logits_dev / pred_dev [slots * n_slot, ...]. */
int ssp2_tail_slots(ssp2_handle h, const float* x_dev, int n_slot, int slots, int attn_skip_last, float* logits_dev, int32_t* pred_dev,
                    const int64_t* labels_dev, int64_t* correct_dev);

/* ... and for streams in the SLAB layout (group as in ssp2_rows; n_slot a multiple of group when slots > 1, so that slot s begins at slab
 * s * n_slot / group): the tail of a search whose baseline was laid out for the stage-1 hook (ssp2_layers_prefix).  group <= 0: ssp2_tail_slots. */
int ssp2_tail_group(ssp2_handle h, const float* x_dev, int n_slot, int slots, int group, int attn_skip_last, float* logits_dev,
                    int32_t* pred_dev, const int64_t* labels_dev, int64_t* correct_dev);

/* a8 on the device: keep only the listed FFN neurons of block `layer` — rows of fc1 (+bias) and
 * columns of fc2 are gathered in HBM (src/vit_pruning.py:297-311 does `W_int[keep]`, `B_int[keep]`, `W_out[:,keep]`
 * on the module); keep_host is ascending, 0 < n_keep <= current d_int.  The engine then runs with the smaller
 * d_int exactly as an engine created from the sliced weights would (bit-identical).  Stream-ordered device work on the handle's
 * stream.  keep_host (pageable OR pinned) is copied into an engine-owned staging buffer before the call returns: the caller may
 * free or rewrite it immediately (the same holds for ssp2_prune_ffn_into). */
int ssp2_prune_ffn(ssp2_handle h, int layer, const int32_t* keep_host, int n_keep);
/* a6/a9 applied for good: block `layer` loses its attention sub-module (always bypassed from now on). */
int ssp2_drop_attention(ssp2_handle h, int layer);
int ssp2_restore_attention(ssp2_handle h, int layer);      /* undo ssp2_drop_attention (the weights were never freed) */
int ssp2_d_int(ssp2_handle h, int layer);

/* a8 / a9 "apply" without touching the dense engine — what keeps a pruned model next to the original
 * (the reference deep-copies for that, mask_conjunction.py:339): `dst` is a second engine of the same architecture
 * whose blocks were created with the PRUNED widths.
 *   ssp2_clone_weights   copies every tensor whose shape agrees (embeddings, attention, norms, head, and the FFN of
 *                        blocks with equal d_int) device to device on dst's stream;
 *   ssp2_prune_ffn_into  gathers the kept neurons of src's block `layer` (fc1 rows + bias, fc2 columns) into dst's
 *                        block, n_keep == dst's d_int of that block; asynchronous on dst's stream, no allocation after
 *                        the first call, bit-identical to ssp2_prune_ffn on a copy. */
int ssp2_clone_weights(ssp2_handle dst, ssp2_handle src);
int ssp2_prune_ffn_into(ssp2_handle dst, ssp2_handle src, int layer, const int32_t* keep_host, int n_keep);

/* a2 standalone (the hook body on an activation tensor that already sits in HBM):
 *   act_dev bf16 (dtype 0) or f32 (dtype 1), [n, tokens, ld] with the first d columns used;
 *   out_dev f32 [ceil(n/group), out_stride] : out[g][j] = sum_{s in group g} sqrt(sum_t act[s,t,j]^2)
 *   (group <= 0: one group of n); norms_ws_dev f32 [2, n, ld] scratch (sums of squares of the two token halves).
 * HBM-bound: n*tokens*d*sizeof(act) bytes read once. */
int ssp2_act_l2_accum(void* hip_stream, const void* act_dev, int dtype, int n, int tokens, int d, int ld,
                      int score_chain, int group, float* norms_ws_dev, float* out_dev, size_t out_stride);

/* a3's building block on its own — one nn.Linear of the forward with its fused epilogue, on caller-owned device buffers
 * (what `F.linear` under autocast computes at src/vit_pruning.py:180 inside the third-party model):
 *   a_dev bf16 [M, lda] (K used), w_dev bf16 [ceil256(N), ldw] (nn.Linear [out,in]; rows past N zero), bias_dev f32
 *   [ceil256(N)] holding bf16-representable values; K % 64 == 0, N % 64 == 0.
 *   SSP2_EPI_BF16  out_dev[M, ldo] = bf16(acc + bias)                     QKV projection
 *   SSP2_EPI_GELU  out_dev[M, ldo] = bf16(gelu_erf(bf16(acc + bias)))     fc1 (+ the activation)
 *   SSP2_EPI_RESID x_dev[M, ldx] (f32) += float(bf16(acc + bias))         out-proj / fc2 (+ the residual add)
 * kernel: 0 = the routing of the forward (M >= 4096 rows: persistent 256 x 256 tiles), 1 / 2 force the 128 x 128 /
 * the 256 x 256 kernel; all give identical bits. */
enum { SSP2_EPI_BF16 = 0, SSP2_EPI_RESID = 1, SSP2_EPI_GELU = 2 };
int ssp2_linear_bf16(void* hip_stream, int epilogue, const uint16_t* a_dev, int lda, const uint16_t* w_dev, int ldw,
                     const float* bias_dev, int M, int N, int K, uint16_t* out_dev, int ldo, float* x_dev, int ldx, int kernel);

/* f4 — input pipeline on the device, the step in front of the path (adaptation-for-Pures-framework/auto_2ssp.py:
 * 290-301): uint8 HWC images -> bicubic resize to out x out (Pillow's 8-bit fixed-point resampler, bit-identical) ->
 * optional horizontal flip of the resized image -> /255 -> (x - mean) / std -> fp32 NCHW.
 *   img_dev u8 [n, in_h, in_w, 3]; hflip_dev u8 [n] or NULL; tmp_dev u8 [n, in_h, out, 3] scratch;
 *   out_dev f32 [n, 3, out, out]; out_u8_dev u8 [n, out, out, 3] or NULL (the resized image itself, for checks). */
typedef struct ssp2_preproc* ssp2_preproc_handle;
int ssp2_preproc_create(int in_h, int in_w, int out_size, const float* mean3, const float* std3, ssp2_preproc_handle* out);
int ssp2_preproc_run(ssp2_preproc_handle p, void* hip_stream, const uint8_t* img_dev, int n, const uint8_t* hflip_dev,
                     uint8_t* tmp_dev, float* out_dev, uint8_t* out_u8_dev);
int ssp2_preproc_destroy(ssp2_preproc_handle p);

/* per-kernel-class HIP-event timing (bench.py roofline leg).  klass: see SSP2_K_* */
enum { SSP2_K_GEMM_FC1 = 0, SSP2_K_GEMM_FC2, SSP2_K_GEMM_QKV, SSP2_K_GEMM_PROJ, SSP2_K_GEMM_PATCH,
       SSP2_K_GEMM_HEAD, SSP2_K_ATTN, SSP2_K_LN, SSP2_K_SCORE_FINISH, SSP2_K_ACT_L2, SSP2_K_OTHER,
       SSP2_K_COUNT };
int ssp2_profile_begin(ssp2_handle h, int klass);                       /* start recording event pairs; SSP2_K_COUNT = every class */
/* totals of one class since ssp2_profile_begin: device time, launches, algorithmic flops (GEMMs 2*M*N*K, attention 4*N*N*d_h per
 * head and image) and algorithmic HBM bytes (LayerNorm, attention).  Synchronises the stream; the recording stays on. */
int ssp2_profile_query(ssp2_handle h, int klass, double* total_ms, int64_t* launches, double* flops, double* bytes);
/* synchronises the stream; gemm_flops = sum of the algorithmic 2*M*N*K of the recorded launches (GEMM classes) */
int ssp2_profile_end(ssp2_handle h, double* total_ms, int64_t* launches, double* gemm_flops);

/* workspace / capacity queries */
/* SSP2_Q_FP8_SATURATED (fp8 mode): how many waves have clipped a value at the e4m3 range in the attention output's hand-off to the
 * out-projection — it is written as e4m3(o x 16), so |o| > 28 clips — since the handle was created or the counter was last read with
 * SSP2_Q_FP8_SATURATED_RESET (which returns the count and zeroes it).  Synchronises the handle's stream.  0 on a model whose
 * attention outputs stay in range; > 0 means: run that checkpoint with SSP2_OPT_FP8_PROJ = 0 (or in bf16).  (The LayerNorm outputs
 * carry per-row scales and cannot clip; the GELU output that fc2 reads is cast unscaled — it clips beyond 448; counting in that epilogue
 * costs the fc1 kernels the registers they do not have (20-188 spilled bytes per lane when it was tried), so it is counted during CALIBRATION
 * passes instead: SSP2_Q_FP8_FC2_TOP_CODES.) */
enum { SSP2_Q_DIM = 0, SSP2_Q_DEPTH, SSP2_Q_CLASSES, SSP2_Q_SCORE_LD /* max ceil64(d_int) */, SSP2_Q_MAX_IMAGES, SSP2_Q_TOKENS, SSP2_Q_IMG,
       SSP2_Q_FP8_SATURATED, SSP2_Q_FP8_SATURATED_RESET,
       SSP2_Q_FP8_FC2_TOP_CODES /* fp8 mode: e4m3 bytes of the fc1 -> fc2 hand-off (the GELU output, cast unscaled and saturating) that the forwards between
                                 * ssp2_fp8_calibrate_begin and _end found ON the top code +-448 — values at or beyond the e4m3 range.  Counted by one more read of
                                 * the activation per block during calibration passes only (the fc1 epilogue has no register left to count in: 253 VGPRs); 0 on a
                                 * checkpoint whose FFN activations stay below 448; > 0 says: run it in bf16.  Saturates at 2^31 - 1. */,
       SSP2_Q_LAB_BUILD /* 1: this library was built with -DSSP2_LAB (carries the opt-in kernel forms SSP2_OPT_LN_FUSION / _DEFER_RESID / _GROUP256); 0: the product build */ };
int ssp2_query(ssp2_handle h, int what);                                  /* >= 0, or SSP2_EINVAL */
int ssp2_tokens(ssp2_handle h);
size_t ssp2_workspace_bytes(ssp2_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SSP2VIT_H */
