/* mmrag.h -- C-ABI of the MI355X-native embed-and-retrieve engine (libmmrag.so).
 *
 * Drop-in boundary for the hot path of someone-in-somewhere/multimodal_rag
 * (app/utils/embedder.py).  The reference has no FFI of its own: its arithmetic lives in
 * two third-party engines that embedder.py calls through Python.  Each entry point below
 * replaces one of those engine calls; INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add at the cited call site.
 *
 * Conventions
 *   - plain pointers and sizes only; every `dev` pointer is HIP device memory owned by the
 *     caller (in this repo: torch tensors' data_ptr()); nothing is retained past return;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued on it and the call returns without synchronising;
 *   - return value: 0 = ok, otherwise an MMRAG_E* code; mmrag_last_error() gives the
 *     message for the calling thread.  No entry point aborts or throws;
 *   - re-entrant: no global mutable scratch; callers pass a workspace sized by the matching
 *     *_workspace_bytes() query (embedder.py:368/595 call the engines from thread-pool
 *     workers, SURVEY.md section 8b "Threading").
 *
 * There is NO CPU fallback behind these symbols: without a gfx950 device they fail.
 */
#ifndef MMRAG_H
#define MMRAG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMRAG_OK 0
#define MMRAG_EINVAL 1   /* bad argument (shape, alignment, dtype, k) */
#define MMRAG_EWORKSPACE 2 /* workspace too small */
#define MMRAG_EHIP 3     /* a HIP runtime call failed */
#define MMRAG_EUNSUPPORTED 4

/* storage dtype of vectors / weights / activations */
#define MMRAG_F32 0
#define MMRAG_F16 1
#define MMRAG_BF16 2

#define MMRAG_MAX_K 20 /* api.py:163  top_k: int = Field(5, ge=1, le=20) */

int mmrag_abi_version(void);
const char *mmrag_last_error(void);

/* Leading dimension (in elements) the corpus / query matrices must be padded to for `d`
 * logical columns of `dtype`: rows are streamed in 128-byte K-slabs.  Pad columns are 0. */
int64_t mmrag_padded_dim(int d, int dtype);

/* ---------------------------------------------------------------------------------------
 * Retrieve.  Replaces chromadb's  collection.query(query_embeddings=[v], n_results=k, ...)
 * reference call site: app/utils/embedder.py:595-601 (and :900-905), result flattening
 * :604-609.  Exact batched inner-product k-NN (rows are unit-norm => cosine), fused
 * Q x corpus^T MFMA GEMM + per-query top-k selection; the [B, n] score matrix never
 * reaches HBM.
 *
 *   q          dev [B, ld]      queries, dtype `dtype`, pad columns zero
 *   corpus     dev [n, ld]      this shard's rows, same dtype / ld, pad columns zero
 *   ld         = mmrag_padded_dim(d, dtype) or any larger multiple of it
 *   k          1..MMRAG_MAX_K
 *   row_offset added to local row numbers => global row ids (shard base)
 *   alive_bits dev, optional (NULL = all alive): bit r%32 of word r/32 set <=> row r may be
 *              returned (delete_document tombstones embedder.py:619-656, `where` filters :599)
 *   out_scores dev [B, k] float32, descending; ties -> lower row first
 *   out_rows   dev [B, k] int64 global rows; (-inf, -1) padding when fewer than k live rows
 *   workspace  dev, >= mmrag_cosine_topk_workspace_bytes(B, n, k) bytes, 16-byte aligned
 * ------------------------------------------------------------------------------------- */
size_t mmrag_cosine_topk_workspace_bytes(int B, int64_t n, int k);

int mmrag_cosine_topk(const void *q, const void *corpus, int B, int64_t n, int d, int64_t ld,
                      int dtype, int k, int64_t row_offset, const uint32_t *alive_bits,
                      float *out_scores, int64_t *out_rows, void *workspace,
                      size_t workspace_bytes, void *stream);

/* The same search as two stream-ordered phases, so a caller can time, overlap or graph-capture
 * them separately: phase 1 = the fused GEMM + per-lane selection kernel (reads the corpus once,
 * leaves candidate lists in `workspace`); phase 2 = the small per-query merge of those lists.
 * mmrag_cosine_topk(...) == lists(...) then select(...) with the same B, n, k, workspace. */
int mmrag_cosine_topk_lists(const void *q, const void *corpus, int B, int64_t n, int d, int64_t ld,
                            int dtype, int k, const uint32_t *alive_bits, void *workspace,
                            size_t workspace_bytes, void *stream);
int mmrag_cosine_topk_select(int B, int64_t n, int k, int64_t row_offset, const void *workspace,
                             float *out_scores, int64_t *out_rows, void *stream);

/* Merge G shards' local top-k (layout [G, B, k_in], as produced by an all-gather of
 * mmrag_cosine_topk outputs) into the global top-k [B, k].  Device version (one tiny
 * kernel) and host version (north star: "final host merge"); identical ordering rule.
 * Replaces nothing in the reference (it is single-process); it is the exchange step of the
 * row-sharded index. */
int mmrag_merge_topk(const float *scores, const int64_t *rows, int G, int B, int k_in, int k,
                     float *out_scores, int64_t *out_rows, void *stream);
int mmrag_merge_topk_host(const float *scores, const int64_t *rows, int G, int B, int k_in,
                          int k, float *out_scores, int64_t *out_rows);
/* Same merge over G rank blocks laid out [rows B*k_in i64 | scores B*k_in f32 | pad to 8 bytes]
 * each, i.e. the result of ONE all-gather of a packed per-rank buffer. */
int mmrag_merge_topk_host_packed(const void *blocks, int G, int B, int k_in, int k,
                                 float *out_scores, int64_t *out_rows);

/* ---------------------------------------------------------------------------------------
 * Store.  Replaces chromadb's  collection.add(embeddings=..., ...)  vector half
 * (app/utils/embedder.py:514-523): cast + copy `m` new float32 rows into the shard matrix
 * at row n_used (the id/metadata/document table stays on the host).
 *   corpus  dev [capacity, ld] dtype   new_rows dev [m, d] float32 (tightly packed)
 * ------------------------------------------------------------------------------------- */
int mmrag_append_rows(void *corpus, int64_t capacity, int64_t ld, int dtype, int64_t n_used,
                      const float *new_rows, int64_t m, int d, void *stream);

/* Stable compaction after deletes (collection.delete, embedder.py:639-642):
 * dst[i, :] = src[keep_rows[i], :] for i < m.  dst and src must not overlap. */
int mmrag_gather_rows(void *dst, const void *src, int64_t ld, int dtype,
                      const int64_t *keep_rows, int64_t m, void *stream);

/* Fetch stored vectors as float32 (collection.get(ids, include=['embeddings']),
 * embedder.py:887-897): out[i, :d] = float(corpus[rows[i], :d]). */
int mmrag_fetch_rows_f32(const void *corpus, int64_t ld, int dtype, const int64_t *rows,
                         int64_t m, int d, float *out, void *stream);

/* Stream-ordered copy of a result block into (pinned) host memory: the last step of a query batch
 * (`results['ids'][0]` ... reach the host, embedder.py:604-609).  Thin wrapper so a serving loop can stay
 * on raw stream handles. */
int mmrag_copy_to_host_async(void *dst_host, const void *src_dev, size_t bytes, void *stream);


/* ---------------------------------------------------------------------------------------
 * Embed.  Replaces SentenceTransformer.encode(texts, batch_size=len(texts),
 * convert_to_numpy=True, normalize_embeddings=True)   app/utils/embedder.py:397-403
 * (tokenisation stays on the host; these entry points take token ids).
 * fp16 weights/activations, fp32 accumulate and fp32 LayerNorm / softmax / pooling statistics.
 * Sequences are PACKED: token t of sequence b is row cu_seqlens[b] + t; no padded tokens.
 * ------------------------------------------------------------------------------------- */
#define MMRAG_ACT_NONE 0
#define MMRAG_ACT_GELU 1       /* erf GELU (BERT) */
#define MMRAG_ACT_QUICK_GELU 2 /* x * sigmoid(1.702 x) (CLIP) */

#define MMRAG_ARCH_BERT 0  /* post-LN blocks, learned absolute positions, token-type 0, embedding LN */
#define MMRAG_ARCH_PRELN 1 /* pre-LN blocks (CLIP towers), final LN, bias-free projection */

#define MMRAG_POOL_MEAN 0  /* masked mean over the sequence's tokens (all-MiniLM-L6-v2) */
#define MMRAG_POOL_FIRST 1 /* first token: [CLS] (bge-base-en-v1.5, CLIP vision) */
#define MMRAG_POOL_SELECT 2 /* token sel[b] of each sequence (CLIP text: EOS position) */

typedef struct mmrag_encoder_desc {
    int32_t arch;
    int32_t n_layers, hidden, n_heads, intermediate, vocab, max_pos;
    int32_t pool, act, causal, normalize;
    int32_t out_dim; /* == hidden for BERT; projection width for MMRAG_ARCH_PRELN */
    float ln_eps;
    int32_t image, patch; /* vision tower only (mmrag_vit_forward): square image side, patch side */
} mmrag_encoder_desc;

/* Weight table `w` (device pointers; matrices fp16 stored [out_features][in_features], i.e.
 * the transpose of a torch Linear's .weight.T -- exactly nn.Linear.weight; biases and LayerNorm
 * parameters fp32):
 *   w[0] tok_emb [vocab,H]  w[1] pos_emb [max_pos,H]  w[2] type_emb row 0 [H] (NULL if none)
 *   w[3], w[4] embedding LayerNorm gamma, beta (both NULL for MMRAG_ARCH_PRELN text)
 *   then per layer l, at w[5 + 12 l ...]:
 *     wqkv [3H,H], bqkv [3H], wo [H,H], bo [H], ln1_g, ln1_b, w1 [I,H], b1 [I], w2 [H,I], b2 [H],
 *     ln2_g, ln2_b           (ln1 = attention LN, ln2 = MLP LN; pre- or post- per arch)
 *   MMRAG_ARCH_PRELN tail at w[5 + 12 L ...]: final_ln_g, final_ln_b, proj [out_dim, H]
 *
 *   ids, pos_ids  dev [T] int32 packed token ids / position ids
 *   cu_seqlens    dev [B+1] int32 row offsets, cu_seqlens[B] == T
 *   sel           dev [B] int32 (MMRAG_POOL_SELECT only)
 *   out           dev [B, out_dim] float32, L2-normalised when desc.normalize != 0 */
size_t mmrag_encoder_workspace_bytes(const mmrag_encoder_desc *desc, int64_t T, int B);
int mmrag_encoder_forward(const mmrag_encoder_desc *desc, const void *const *w, const int32_t *ids,
                          const int32_t *pos_ids, const int32_t *cu_seqlens, const int32_t *sel,
                          int64_t T, int B, int max_len, float *out, void *workspace,
                          size_t workspace_bytes, void *stream);

/* The encoder at the reference's own precision (opt-in; MMRAG_ENCODER_PRECISION=fp32).  SentenceTransformer.encode runs
 * in float32 (app/utils/embedder.py:397-403, device pick :204-210, no autocast anywhere); this entry point computes as
 * it does: float32 weights and activations, every contraction on the exact float32 matrix instruction
 * (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain), LayerNorm / softmax / erf-GELU / pooling in float32.  BERT family
 * only.  Same arguments as mmrag_encoder_forward; the weight table has the same order with EVERY entry float32
 * (matrices [out_features][in_features]).  Throughput is bound by the float32 matrix rate (1/16 of fp16).
 *   mmrag_linear_f32   the GEMM of that mode on its own (parity tests): out = act(x . wt^T + bias) (+ resid) */
size_t mmrag_encoder_f32_workspace_bytes(const mmrag_encoder_desc *desc, int64_t T, int B);
int mmrag_encoder_forward_f32(const mmrag_encoder_desc *desc, const void *const *w, const int32_t *ids,
                              const int32_t *pos_ids, const int32_t *cu_seqlens, const int32_t *sel, int64_t T, int B,
                              int max_len, float *out, void *workspace, size_t workspace_bytes, void *stream);
int mmrag_linear_f32(const float *x, int64_t M, int K, const float *wt, int N, const float *bias, int act,
                     const float *resid, float *out, void *stream);

/* Vision tower (CLIP ViT-B/32 shape; BASELINE config 4 -- no reference behaviour):
 * patchify (+ fused uint8 -> normalised fp16 preprocessing) -> patch-embedding GEMM -> class token +
 * positions -> pre-LN -> the same pre-LN blocks / final LN / projection / L2 normalise as the text tower.
 * Weight table: w[0] patch kernel [H, 3*P*P] (conv weight flattened), w[1] pos_emb [NP+1, H],
 * w[2] class_embedding [H], w[3], w[4] pre-LN gamma/beta, layers and tail as for MMRAG_ARCH_PRELN.
 *   pixels       dev: fp16 [B,3,image,image] already normalised (MMRAG_PIXELS_F16_CHW) or uint8
 *                [B,image,image,3] raw crops (MMRAG_PIXELS_U8_HWC; CLIP mean/std applied on the GPU)
 *   cu_seqlens   dev [B+1] int32 = b * (NP+1)
 *   workspace    >= mmrag_encoder_workspace_bytes(desc, B*(NP+1), B) */
#define MMRAG_PIXELS_F16_CHW 0
#define MMRAG_PIXELS_U8_HWC 1
int mmrag_vit_forward(const mmrag_encoder_desc *desc, const void *const *w, const void *pixels, int pixel_kind,
                      const int32_t *cu_seqlens, int B, float *out, void *workspace, size_t workspace_bytes,
                      void *stream);

/* Host-side WordPiece tokenizer (multi-threaded).  Replaces the tokenisation SentenceTransformer.encode performs
 * in native code (Hugging Face `tokenizers`) before the model runs -- app/utils/embedder.py:397-403.  BERT uncased
 * BasicTokenizer + greedy longest-match WordPiece over a caller-supplied vocabulary; strings travel as UTF-32.
 *   create   vocab token i = cps[offsets[i] .. offsets[i+1]), id = i; [CLS]/[SEP]/[UNK] looked up by name
 *            (defaults 101/102/100).  Returns NULL on error.
 *   encode   text i = cps[offsets[i] .. offsets[i+1]); ids [n, max_length] int32: row i holds
 *            [CLS] pieces... [SEP] truncated to max_length, lens[i] its length; n_threads host threads. */
void *mmrag_wordpiece_create(const uint32_t *cps, const int64_t *offsets, int n_tokens, int lower);
void mmrag_wordpiece_destroy(void *tokenizer);
int mmrag_wordpiece_encode_batch(const void *tokenizer, const uint32_t *cps, const int64_t *offsets, int n,
                                 int max_length, int32_t *ids, int32_t *lens, int n_threads);

/* CLIP byte-level BPE (the text tower's tokenizer, BASELINE config 4; the reference only names CLIP in config.py:106).
 * Host code, multi-threaded; equals multimodal_rag_amd/tokenizer.py:ClipBpeTokenizer, which tests pin to
 * transformers.CLIPTokenizer.  The caller passes text already NFC-normalised, whitespace-collapsed and lower-cased.
 *   mmrag_clip_bpe_create   vocabulary entries as UTF-32 strings with their ids, merges ("first second") in rank order;
 *                           NULL (see mmrag_last_error) without <|startoftext|> / <|endoftext|>
 *   mmrag_clip_bpe_encode_batch  ids [n, max_length] int32, rows [sot] ids[: max_length - 2] [eot]; lens [n] */
void *mmrag_clip_bpe_create(const uint32_t *vocab_cps, const int64_t *vocab_offsets, const int32_t *vocab_ids,
                            int n_vocab, const uint32_t *merge_cps, const int64_t *merge_offsets, int n_merges);
void mmrag_clip_bpe_destroy(void *tokenizer);
int mmrag_clip_bpe_encode_batch(const void *tokenizer, const uint32_t *cps, const int64_t *offsets, int n,
                                int max_length, int32_t *ids, int32_t *lens, int n_threads);

/* Image front end of the vision tower (BASELINE config 4; the reference has no image encoder):
 * CLIP's preprocessing = shortest edge -> 224 with PIL bicubic, centre crop, done on uint8.  Bit-exact with
 * Pillow's 8-bit resampler (two integer passes, 22-bit fixed-point taps).
 *   mmrag_resample_ksize / mmrag_resample_coeffs   HOST: taps of output indices [first, first+count) of an
 *       in_size -> out_size resample; bounds [count,2] = (first input index, tap count), taps [count, ksize].
 *   mmrag_resize_crop_u8   DEVICE: src [H,W,3] uint8 (row stride src_row_bytes) -> dst [out_h,out_w,3] uint8 with
 *       horizontal taps (bx,kx: per output COLUMN of the crop) then vertical taps (by,ky: per output ROW of the
 *       crop); [y_lo,y_hi) = source rows the vertical taps touch; tmp holds (y_hi-y_lo)*out_w*3 bytes.
 *       All table and image pointers are device pointers. */
int mmrag_resample_ksize(int in_size, int out_size);
int mmrag_resample_coeffs(int in_size, int out_size, int first, int count, int32_t *bounds, int32_t *taps);
int mmrag_resize_crop_u8(const uint8_t *src, int H, int W, int64_t src_row_bytes, const int32_t *bx,
                         const int32_t *kx, int ksx, const int32_t *by, const int32_t *ky, int ksy, int out_h,
                         int out_w, int y_lo, int y_hi, uint8_t *tmp, uint8_t *dst, void *stream);

/* Measured peaks for the roofline report ("a measured stream-copy bandwidth and a measured MFMA
 * micro-benchmark peak, fractions against both the vendor and the measured peaks").  Not on the product path.
 *   mmrag_device_info        CU count, maximum shader clock (MHz), HBM bytes of the current device
 *   mmrag_bench_stream_copy  one 16-byte-per-lane copy of `bytes` bytes dst <- src (time it with events: moves 2 x bytes)
 *   mmrag_bench_stream_read  read-only stream: every lane loads 16 bytes per step (non-temporal) and sums them; one
 *                            float per thread goes to `out` (dev, 8 x 256 floats per CU).  The READ peak a scan of
 *                            the corpus is measured against (half of a copy's rate is not one).
 *   mmrag_bench_stream_write write-only stream: 16-byte non-temporal stores of a constant over `bytes` bytes
 *   mmrag_bench_mfma_f16     `iters` x 4 back-to-back v_mfma_f32_32x32x16_f16 per wave, one wave per SIMD on every CU,
 *                            operands from `seed` (dev, 256 x 16 bytes of fp16 data; use random values: the clock the
 *                            chip holds depends on them); `out` dev, 256 floats per CU; *flops = work of one launch
 *   mmrag_bench_mfma_f16_16x16x32  the same output tile per wave (64 x 64) on v_mfma_f32_16x16x32_f16: 16 per step */
int mmrag_device_info(int *n_cus, int *max_clock_mhz, int64_t *hbm_bytes);
int mmrag_bench_stream_copy(void *dst, const void *src, int64_t bytes, void *stream);
int mmrag_bench_stream_read(const void *src, int64_t bytes, float *out, void *stream);
int mmrag_bench_stream_write(void *dst, int64_t bytes, void *stream);
int mmrag_bench_mfma_f16(const void *seed, float *out, int iters, int64_t *flops, void *stream);
int mmrag_bench_mfma_f16_16x16x32(const void *seed, float *out, int iters, int64_t *flops, void *stream);

/* The encoder's building blocks, exported so each kernel can be parity-tested on its own. */
int mmrag_linear_f16(const void *x, int64_t M, int K, const void *wt, int N, const float *bias, int act,
                     const void *resid, void *out, void *stream);
int mmrag_layernorm_f16(const void *x, void *out, const float *gamma, const float *beta, int64_t T, int H,
                        float eps, void *stream);
int mmrag_embed_ln_f16(const int32_t *ids, const int32_t *pos_ids, const void *tok, const void *pos,
                       const void *type0, const float *gamma, const float *beta, void *out, int64_t T,
                       int H, int vocab, int max_pos, float eps, void *stream);
int mmrag_attention_f16(const void *qkv, const int32_t *cu_seqlens, void *ctx, int B, int max_len, int H,
                        int n_heads, int causal, void *stream);
int mmrag_pool_normalize_f16(const void *x, const int32_t *cu_seqlens, const int32_t *sel, float *out,
                             int B, int H, int pool, int normalize, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMRAG_H */
