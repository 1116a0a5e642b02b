/*
 * libmi355_hotpath.so -- C ABI of the MI355X (gfx950 / CDNA4) SGLang hot path.
 *
 * Every entry point:
 *   - takes plain device pointers, sizes and strides (in ELEMENTS unless a name says bytes)
 *     and the hipStream_t to launch on (passed as void*; NULL = the null stream);
 *   - returns 0 (MI355_OK) or a negative MI355_ERR_* code, never throws; the text of the
 *     last error on the calling thread is available from mi355_last_error();
 *   - validates shapes/dtypes/alignment on the host BEFORE launching anything;
 *   - is asynchronous: no hipMalloc/hipFree, no host synchronisation, safe under HIP graph
 *     capture; all scratch memory is passed in by the caller (the one exception,
 *     mi355_decode_attention_static, says so).
 *
 * "file:line" citations name the reference interface (under the reference checkout's root)
 * that each entry point replaces.
 */
#ifndef MI355_HOTPATH_H_
#define MI355_HOTPATH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_MAX_KV_SPLITS 4096 /* upper bound of num_splits in the decode attention entry points */
#define MI355_ERR_INVALID (-1)     /* bad argument: null pointer, negative size, misaligned */
#define MI355_ERR_UNSUPPORTED (-2) /* dtype / head_dim / shape outside what the kernels cover */
#define MI355_ERR_LAUNCH (-3)      /* hipGetLastError() after the launch was not hipSuccess */

/* element type codes */
#define MI355_F32 0
#define MI355_BF16 1
#define MI355_F16 2
/* index type codes */
#define MI355_I32 0
#define MI355_I64 1
/* OR into the `dtype` argument of the attention / KV-write entry points that document it: the KV
 * cache buffers hold OCP e4m3fn bytes (kv_cache_dtype fp8_e4m3, mem_cache/memory_pool.py:118-122,
 * 436-446) while q / k / v / o keep the 16-bit dtype in the low byte.  Values are stored as
 * clamp(x, +-448) rounded to nearest even and read back by exact widening.  head_dim 128 (16-bit q, causal or
 * full mask) runs on the matrix-core kernels; any other head dim, a sliding window, a logit cap or a custom mask
 * on the wave-per-head decode kernel and the vector-ALU prefill kernel (same values, slower). */
#define MI355_KV_FP8 0x100
/* OR into the same `dtype` argument: k_buffer / v_buffer are laid out [Hkv, num_slots, D] (head-major: the rows
 * one kv head's workgroup streams are adjacent in memory) instead of [num_slots, Hkv, D].  The same slots, heads
 * and values -- MHATokenToKVPool.get_key_buffer(layer) (mem_cache/memory_pool.py:395-418) seen through
 * buffer.permute(1, 0, 2); num_slots must then be the buffer's true slot count (it is the head stride).  Latent
 * (MLA) caches have one head and ignore the flag. */
#define MI355_KV_HEAD_MAJOR 0x200
/* With MI355_KV_HEAD_MAJOR: k_buffer and v_buffer are the two halves of ONE allocation [Hkv, num_slots, 2, D]
 * (v_buffer = k_buffer + D elements; head_dim == v_head_dim): a token's key and value row of a head are adjacent,
 * so a decode workgroup reads one 2 D-element run per token instead of two D-element runs from two regions. */
#define MI355_KV_INTERLEAVED 0x400

#define MI355_ABI_VERSION 1
int mi355_abi_version(void);
const char* mi355_last_error(void);

/* ---- K8  RMSNorm ------------------------------------------------------------------------
 * out[r,:] = T( (x[r,:] * rsqrt(mean(x[r,:]^2) + eps)) * w ), fp32 math.
 * replaces: RMSNorm.forward_native / forward_hip (python/sglang/srt/layers/layernorm.py:113-147),
 *           sgl_kernel.rmsnorm (sgl-kernel/csrc/elementwise/ -> common_extension.cc:59),
 *           rmsnorm_cpu (sgl-kernel/csrc/cpu/norm.cpp:68-134). */
int mi355_rmsnorm(void* out, const void* x, const void* weight, int64_t rows, int64_t hidden,
                  int64_t x_row_stride, int64_t out_row_stride, float eps, int dtype,
                  void* stream);

/* In place on BOTH tensors: s = f32(x)+f32(residual); residual = T(s);
 * x = T( (s * rsqrt(mean(s^2)+eps)) * w ).
 * replaces: layernorm.py:121-123,137-143; fused_add_rmsnorm
 *           (sgl-kernel/csrc/elementwise/fused_add_rms_norm_kernel.cu:24-59). */
int mi355_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t rows,
                            int64_t hidden, float eps, int dtype, void* stream);

/* L2 warm-up hint.  Row-norm and RoPE launches leave HBM idle; the next such launch made by the
 * calling thread (mi355_rmsnorm, mi355_fused_add_rmsnorm, mi355_rope_inplace) carries extra
 * workgroups that read the first budget_bytes of the packed weight image w_packed into L2, laid
 * out the way the skinny GEMM y[m, n] = x[m, k] W^T (mi355_linear_packed, *_silu_mul,
 * mi355_fp8_scaled_mm_packed; w_elem_bytes 2 or 1) will ask for them, so that GEMM starts on L2
 * hits.  Loads only: results never change.  w_packed == NULL cancels a pending hint; a shape the
 * ring GEMM does not serve is ignored.  No counterpart in the reference (weights there are read
 * cold by every GEMM). */
int mi355_prefetch_next(const void* w_packed, int64_t m, int64_t n, int64_t k, int w_elem_bytes,
                        int64_t budget_bytes);

/* ---- K10 SiLU-and-mul -------------------------------------------------------------------
 * out[r, j] = T( f32(T(silu(x[r, j]))) * x[r, d + j] ), x is [rows, 2d] contiguous.
 * replaces: SiluAndMul.forward_native/forward_cuda (python/sglang/srt/layers/activation.py:55-65),
 *           silu_and_mul (sgl-kernel/csrc/elementwise/activation.cu:22-50). */
int mi355_silu_and_mul(void* out, const void* x, int64_t rows, int64_t d, int dtype,
                       void* stream);

/* ---- K12 epilogue: final-logit soft cap ---------------------------------------------------
 * logits[i] <- cap * tanh(logits[i] / cap), in place on n contiguous fp32 values, cap > 0.
 * replaces: fused_softcap / fused_softcap_kernel (python/sglang/srt/layers/logits_processor.py:579-618),
 *           called from LogitsProcessor._get_logits (:492-493). */
int mi355_softcap_inplace(float* logits, int64_t n, float cap, void* stream);

/* ---- K9 (+K7) RoPE from a cos|sin cache, in place, optional fused KV-cache write ----------
 * q: [T, Hq, D] view with row stride q_stride; k: [T, Hkv, D] with k_stride; rotates the first
 * rotary_dim of every head with cos_sin_cache[positions[t]] ([max_pos, rotary_dim], first half
 * cos, second half sin; cache_dtype = MI355_F32 or the tensor dtype).  Rounding follows
 * _apply_rotary_emb: cos/sin are cast to T, and every multiply/add rounds to T.
 * If k_buffer != NULL also writes k (after rotation) and v rows to
 * k_buffer[loc[t]] / v_buffer[loc[t]] ([slots, Hkv, D] contiguous) -- set_kv_buffer fused in.
 * replaces: RotaryEmbedding.forward_native/forward_cuda (layers/rotary_embedding.py:138-166,
 *           218-244), apply_rope_with_cos_sin_cache_inplace (sgl-kernel/csrc/elementwise/rope.cu:22-89),
 *           MHATokenToKVPool.set_kv_buffer (mem_cache/memory_pool.py:420-458). */
int mi355_rope_inplace(void* q, void* k, const void* v, const int64_t* positions,
                       const void* cos_sin_cache, int cache_dtype, int64_t tokens, int hq,
                       int hkv, int head_dim, int rotary_dim, int64_t q_stride, int64_t k_stride,
                       int64_t v_stride, int is_neox, void* k_buffer, void* v_buffer,
                       const void* loc, int loc_itype, int64_t num_slots, int dtype, void* stream);

/* ---- K7 KV-cache write --------------------------------------------------------------------
 * k_buffer[loc[t], :, :] = k[t], v_buffer[loc[t]] = v[t]; row = heads*dim elements.
 * replaces: MHATokenToKVPool.set_kv_buffer (mem_cache/memory_pool.py:420-458),
 *           decode_set_kv_buffer (sgl-kernel/csrc/cpu/decode.cpp:770-809). */
int mi355_set_kv_buffer(void* k_buffer, void* v_buffer, const void* k, const void* v,
                        const void* loc, int loc_itype, int64_t tokens, int64_t k_row_elems,
                        int64_t v_row_elems, int64_t k_stride, int64_t v_stride,
                        int64_t num_slots, int dtype, void* stream);

/* The same scatter with the row seen as `heads` heads of k_head_dim / v_head_dim features, so that it can follow either
 * cache layout: [num_slots, heads, d] or, under dtype | MI355_KV_HEAD_MAJOR, [heads, num_slots, d] (and e4m3fn bytes
 * under MI355_KV_FP8, same arithmetic).  16-bit k / v, head dims multiples of 8.
 * replaces: MHATokenToKVPool.set_kv_buffer (mem_cache/memory_pool.py:420-458). */
int mi355_set_kv_buffer_heads(void* k_buffer, void* v_buffer, const void* k, const void* v, const void* loc,
                              int loc_itype, int64_t tokens, int heads, int k_head_dim, int v_head_dim, int64_t k_stride,
                              int64_t v_stride, int64_t num_slots, int dtype, void* stream);

/* ---- K15 embedding gather -----------------------------------------------------------------
 * out[t,:] = (vocab_start <= ids[t] < vocab_end) ? table[ids[t]-vocab_start,:] : 0
 * replaces: VocabParallelEmbedding.forward (layers/vocab_parallel_embedding.py:453-475). */
int mi355_embedding(void* out, const int64_t* ids, const void* table, int64_t tokens,
                    int64_t hidden, int64_t vocab_start, int64_t vocab_end, int dtype,
                    void* stream);

/* ---- K6 paged-KV index gather -------------------------------------------------------------
 * kv_indices[kv_indptr[b] + j] = req_to_token[req_pool_indices[b], kv_start[b] + j],
 * j < page_kernel_lens[b]  (kv_start_idx may be NULL = 0).
 * replaces: create_flashinfer_kv_indices_triton (layers/attention/utils.py:5-41). */
int mi355_create_kv_indices(const int32_t* req_to_token, const void* req_pool_indices,
                            int pool_itype, const void* page_kernel_lens, int lens_itype,
                            const int32_t* kv_indptr, const int32_t* kv_start_idx,
                            int32_t* kv_indices, int64_t req_to_token_stride, int64_t bs,
                            void* stream);

/* ---- K16 req->token table write (extend) ---------------------------------------------------
 * req_to_token[req_pool_indices[b], pre_lens[b] + j] = out_cache_loc[cumsum(extend_lens)[b] + j]
 * replaces: write_req_to_token_pool_triton (managers/schedule_batch.py:1900-1934). */
int mi355_write_req_to_token(int32_t* req_to_token, const int64_t* req_pool_indices,
                             const int64_t* pre_lens, const int64_t* seq_lens,
                             const int64_t* extend_lens, const int64_t* out_cache_loc,
                             int64_t req_to_token_stride, int64_t bs, void* stream);

/* ---- K17 positions for an extend batch -----------------------------------------------------
 * positions[start[b] + j] = prefix_lens[b] + j, extend_start_loc[b] = start[b] = excl. cumsum.
 * replaces: compute_position_triton/_torch (model_executor/forward_batch_info.py:703-773). */
int mi355_compute_position(const int32_t* extend_prefix_lens, const int32_t* extend_seq_lens,
                           int64_t* positions, int32_t* extend_start_loc, int64_t bs,
                           void* stream);

/* dst[i][0..n) = src[i][0..n) for i < count <= 5 int64 device vectors, one launch: the refresh of a
 * captured decode step's static inputs (input_ids, req_pool_indices, seq_lens, out_cache_loc,
 * positions).  dst / src are HOST arrays of device pointers.
 * replaces: the five tensor.copy_ calls of CudaGraphRunner.replay_prepare
 *           (model_executor/cuda_graph_runner.py:652-676). */
int mi355_copy_decode_inputs(int64_t* const* dst, const int64_t* const* src, int count, int64_t n, void* stream);

/* ---- K18 paged slot allocation (page_size > 1) ----------------------------------------------
 * Page-aligned KV slots for a batch, taken in request order from the front of free_pages (page
 * ids; slot = page * page_size + offset).  alloc_extend: request b extends from prefix_lens[b] to
 * seq_lens[b] tokens; its new slots first fill the rest of its last page (last_loc[b] + 1, ...),
 * then whole new pages, then the head of one more; out_indices is the flat [sum extend] list,
 * totals[0] = pages consumed, totals[1] = tokens written.  alloc_decode: one token per request
 * (seq_lens already counts it): last_loc[b] + 1, or the first slot of a new page.  The caller drops
 * totals[0] pages from its free list -- and must have checked that many are available: free_pages
 * is read up to that index (it may be null when the list is empty and no new page is needed).
 * replaces: alloc_extend_kernel / alloc_decode_kernel (mem_cache/allocator.py:276-401). */
int mi355_alloc_extend(const int64_t* prefix_lens, const int64_t* seq_lens, const int64_t* last_loc,
                       const int64_t* free_pages, int64_t* out_indices, int64_t* totals, int64_t bs,
                       int64_t page_size, void* stream);
int mi355_alloc_decode(const int64_t* seq_lens, const int64_t* last_loc, const int64_t* free_pages,
                       int64_t* out_indices, int64_t* totals, int64_t bs, int64_t page_size, void* stream);

/* ---- decode step bookkeeping (one launch) ---------------------------------------------------
 * seq_lens[b] += 1; positions[b] = max(seq_lens[b]-1, 0);
 * req_to_token[req_pool_indices[b], seq_lens[b]-1] = out_cache_loc[b]
 * replaces: ScheduleBatch.prepare_for_decode's tensor updates (managers/schedule_batch.py:1543-1573)
 *           + clamp_position (forward_batch_info.py:776-778). */
int mi355_prepare_decode(int32_t* req_to_token, const int64_t* req_pool_indices,
                         int64_t* seq_lens, const int64_t* out_cache_loc, int64_t* positions,
                         int64_t req_to_token_stride, int64_t bs, void* stream);

/* ---- K1+K2 decode attention (split-KV flash decoding) ---------------------------------------
 * q [B, Hq, D] with row stride q_stride (elements), o [B, Hq, Dv] contiguous; k_buffer/v_buffer [slots, Hkv, D|Dv] contiguous.
 * The KV slots of sequence b are page_table[page_off(b) + j], j < len(b), where
 *   mode 0 (req_to_token, as decode_attention_cpu): page_off = req_pool_indices[b]*table_stride,
 *          len = seq_lens[b]   (indptr unused)
 *   mode 1 (kv_indices, as decode_attention_fwd):   page_off = kv_indptr[b],
 *          len = kv_indptr[b+1]-kv_indptr[b]         (req_pool_indices/seq_lens unused)
 * scratch: attn_logits fp32 [B, Hq, num_splits, Dv], attn_lse fp32 [B, Hq, num_splits]
 * (unused when num_splits == 1; num_splits <= MI355_MAX_KV_SPLITS).  logit_cap > 0 applies cap*tanh(s/cap).
 * split_tickets (may be NULL): B * Hq uint32 arrival counters, ZERO when the call is made and zero
 * again when its kernels have finished (hipMemset them once, then hand the same array to every call on
 * one stream).  With them the head_dim-128 16-bit kernel merges the KV splits itself -- the last split
 * workgroup of a (sequence, head group) to finish does it -- and no second launch follows; without them,
 * above 64 splits, and for every other shape, a merge kernel is launched behind the first.  Same bits
 * either way.  Calls that may overlap on the GPU need arrays of their own.
 * kv_chunk_keys (may be NULL): ONE int32 in device memory, read by the kernels when they run: the batch's
 * keys per split.  Sequence b is then cut into ceil(len(b) / *kv_chunk_keys) splits (at most num_splits; a
 * value too small to cover a sequence in num_splits pieces is raised to the even cut), so a ragged batch --
 * or a captured graph planned for the longest context -- does not cut its short sequences into num_splits
 * slivers.  The grid and the scratch layout do not change; the host writes the value (normally
 * ceil(longest sequence / num_splits)) before each launch or graph replay.  This is the per-sequence
 * num_kv_splits of the reference (get_num_kv_splits_triton, layers/attention/triton_backend.py:875-924: one
 * chunk size for the batch, cdiv(seq_len, chunk) splits per sequence).
 * replaces: decode_attention_fwd (layers/attention/triton_ops/decode_attention.py:677-728),
 *           decode_attention_cpu (sgl-kernel/csrc/cpu/decode.cpp:1375-1575),
 *           TorchNativeAttnBackend._run_sdpa_forward_decode (torch_native_backend.py:112-180). */
int mi355_decode_attention(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                           const int32_t* page_table, int mode, const int64_t* req_pool_indices,
                           const int64_t* seq_lens, const int32_t* kv_indptr,
                           int64_t table_stride, float* attn_logits, float* attn_lse,
                           uint32_t* split_tickets, const int32_t* kv_chunk_keys, int num_splits, int64_t bs,
                           int hq, int hkv, int head_dim, int v_head_dim, int64_t q_stride, int64_t num_slots,
                           float sm_scale, float logit_cap, int dtype, void* stream);

/* Number of KV splits mi355_decode_attention should be given for this batch (host-side
 * planner; never exceeds max_splits).
 * replaces: get_num_kv_splits_triton (layers/attention/triton_backend.py:875-924). */
int mi355_decode_num_splits(int64_t bs, int hkv, int64_t max_seq_len, int max_splits);

/* ---- K4 extend / prefill attention -----------------------------------------------------------
 * q_extend [T, Hq, D], k_extend/v_extend [T, Hkv, D|Dv] (row strides given), o_extend [T,Hq,Dv]
 * contiguous.  Sequence b owns rows qo_indptr[b]..qo_indptr[b+1]; its cached prefix is
 * k_buffer[kv_indices[kv_indptr[b]..kv_indptr[b+1]]].  Phase 1 attends the prefix (no mask),
 * phase 2 the new tokens (causal if is_causal).  custom_mask (uint8, may be NULL) follows the
 * reference layout: mask_indptr[b] + row * (prefix+extend) + col, applied to phase 2 always
 * and to phase 1 unless skip_prefix_custom_mask.  sliding_window_size < 0 = off.
 * replaces: extend_attention_fwd (layers/attention/triton_ops/extend_attention.py:306-438),
 *           extend_attention_cpu (sgl-kernel/csrc/cpu/extend.cpp:579-722),
 *           TorchNativeAttnBackend._run_sdpa_forward_extend (torch_native_backend.py:27-110). */
int mi355_extend_attention(const void* q_extend, const void* k_extend, const void* v_extend,
                           void* o_extend, const void* k_buffer, const void* v_buffer,
                           const int32_t* qo_indptr, const int32_t* kv_indptr,
                           const int32_t* kv_indices, const uint8_t* custom_mask,
                           const int64_t* mask_indptr, int is_causal, int skip_prefix_custom_mask,
                           int64_t bs, int64_t max_len_extend, int hq, int hkv, int head_dim,
                           int v_head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                           int64_t num_slots, float sm_scale, float logit_cap,
                           int sliding_window_size, int dtype, void* stream);

/* mi355_decode_attention for callers that cannot allocate device memory or pass a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the split-KV
 * scratch (attn_logits / attn_lse of the call above) is a fixed array inside the library,
 * MI355_DECODE_STATIC_SCRATCH_BYTES per GPU, held from the first call on that GPU until the process
 * ends.  Everything else -- arguments, kernels, results bit for bit -- is mi355_decode_attention.
 * The call itself allocates nothing (no hipMalloc / hipFree); the first call on a device looks the
 * array's device address up (hipGetSymbolAddress), so make one call per device before capturing.
 * bs * hq * num_splits * (v_head_dim + 1) * 4 bytes (partials rounded up to 256 B) beyond the limit
 * -> MI355_ERR_UNSUPPORTED.  All calls on one GPU share the array: calls that may overlap on a GPU
 * (different streams, different threads) race on it; order them or use mi355_decode_attention.
 * The array keeps the previous call's contents; no cell is read before this call has written it.
 * No counterpart in the reference (its backends allocate attn_logits / attn_lse per forward,
 * layers/attention/triton_backend.py:210-219). */
#define MI355_DECODE_STATIC_SCRATCH_BYTES (16 << 20)
int mi355_decode_attention_static(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                  const int32_t* page_table, int mode, const int64_t* req_pool_indices,
                                  const int64_t* seq_lens, const int32_t* kv_indptr, int64_t table_stride,
                                  int num_splits, int64_t bs, int hq, int hkv, int head_dim, int v_head_dim,
                                  int64_t q_stride, int64_t num_slots, float sm_scale, float logit_cap,
                                  int dtype, void* stream);

/* Decode attention fused with everything between the qkv GEMM and the attention kernel.
 * qkv_part: the fp32 split-K partials [nsplit][bs][(hq + 2 hkv) * 128] of the qkv projection
 * (mi355_linear_packed_partials, or mi355_fp8_packed_partials together with that GEMM's per-token
 * and per-channel scales; both NULL otherwise).  The kernel sums them (split order, scales applied
 * as (sum * x_scale[b]) * w_scale[n], one rounding to the tensor dtype = the GEMM's own result), applies neox RoPE over the whole head to q and the new k with
 * cos_sin_cache[positions[b]] (rounding points of mi355_rope_inplace), writes the new k / v rows to
 * slot out_cache_loc[b] (which the page table must name for position seq_len - 1, except for padding
 * rows, whose writes go to the sink slot they name), and runs mi355_decode_attention.  Output and
 * cache contents are bit-identical to mi355_linear_packed -> mi355_rope_inplace (fused KV write) ->
 * mi355_decode_attention (split_tickets, kv_chunk_keys as there).  Needs head_dim 128, a 16-bit dtype, hq / hkv in {1, 2, 4, 8}, no logit cap.
 * replaces: LlamaAttention.forward's rotary_emb + attn sequence (models/llama.py:180-191) on the
 *           decode path; the reference has no fused counterpart. */
int mi355_decode_attention_fused_qkv(const float* qkv_part, int nsplit, const float* part_x_scale,
                                     const float* part_w_scale, const int64_t* positions,
                                     const int64_t* out_cache_loc, const void* cos_sin_cache, int cache_dtype, void* k_buffer,
                                     void* v_buffer, void* o, const int32_t* page_table, int mode,
                                     const int64_t* req_pool_indices, const int64_t* seq_lens,
                                     const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                                     float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys, int num_splits, int64_t bs,
                                     int hq, int hkv, int head_dim, int64_t num_slots, float sm_scale,
                                     int dtype, void* stream);

/* MLA (compressed-KV) decode: every query head attends over ONE latent row per token,
 * kv_buffer [num_slots, 1, kv_lora_rank + rope_dim]; the key is the whole row, the value its first
 * kv_lora_rank features (read from the same buffer, never materialised).  q [B, hq, rank + rope],
 * o [B, hq, kv_lora_rank]; addressing modes, split scratch, split_tickets and kv_chunk_keys as mi355_decode_attention
 * (the rank-512 + rope-64 16-bit kernel merges the splits itself when given tickets).
 * replaces: attn_mqa = RadixAttention(H, 576, scaling, num_kv_heads=1, v_head_dim=512) on
 *           MLATokenToKVPool (python/sglang/srt/models/deepseek_v2.py:866-875,
 *           mem_cache/memory_pool.py:769-970: get_value_buffer = kv_buffer[..., :kv_lora_rank]). */
int mi355_decode_attention_mla(const void* q, const void* kv_buffer, void* o,
                               const int32_t* page_table, int mode,
                               const int64_t* req_pool_indices, const int64_t* seq_lens,
                               const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                               float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys, int num_splits, int64_t bs, int hq,
                               int kv_lora_rank, int rope_dim, int64_t q_stride, int64_t num_slots,
                               float sm_scale, float logit_cap, int dtype, void* stream);

/* MLA decode with RoPE and the cache write inside the launch: q's rope part (features rank .. rank + rope of
 * every head) and k_pe [B, rope] arrive UNROTATED, k_nope [B, rank] is the normed latent; the launch rotates
 * both at positions[b] (cos_sin_cache [max_pos, rope]: cos | sin, fp32 or the tensor dtype; interleaved
 * pairs, or neox pairs when is_neox), writes the new row [k_nope | rope(k_pe)] to kv_buffer[out_cache_loc[b]]
 * and attends over the sequence INCLUDING that row (page_table must already name out_cache_loc[b] as the
 * last key; padded rows of a captured batch point out_cache_loc at the sink slot 0).  Bit for bit
 * mi355_rope_inplace on (q rope part, k_pe) + mi355_set_kv_buffer + mi355_decode_attention_mla, in one launch
 * instead of three.  Needs rank 512 + rope 64 and a 16-bit dtype (..._supported returns 1); the cache is 16-bit or
 * e4m3fn (dtype | MI355_KV_FP8: the new row is quantised with mi355_set_kv_buffer's arithmetic); everything else
 * as mi355_decode_attention_mla.
 * replaces: decode_attention_fwd_grouped_rope (layers/attention/triton_ops/rocm_mla_decode_rope.py:402-439),
 *           called from DeepseekV2AttentionMLA.forward_absorb_fused_mla_rope_{prepare,core}
 *           (python/sglang/srt/models/deepseek_v2.py:1337-1449, 1502-1536; dispatched at :973-980). */
int mi355_decode_attention_mla_fused_rope_supported(int kv_lora_rank, int rope_dim, int dtype, int cache_is_fp8);
int mi355_decode_attention_mla_fused_rope(const void* q, const void* k_nope, int64_t k_nope_stride, const void* k_pe,
                                          int64_t k_pe_stride, const int64_t* positions, const int64_t* out_cache_loc,
                                          const void* cos_sin_cache, int cache_dtype, int is_neox, void* kv_buffer,
                                          void* o, const int32_t* page_table, int mode,
                                          const int64_t* req_pool_indices, const int64_t* seq_lens,
                                          const int32_t* kv_indptr, int64_t table_stride, float* attn_logits,
                                          float* attn_lse, uint32_t* split_tickets, const int32_t* kv_chunk_keys,
                                          int num_splits, int64_t bs, int hq, int kv_lora_rank, int rope_dim,
                                          int64_t q_stride, int64_t num_slots, float sm_scale, float logit_cap,
                                          int dtype, void* stream);

/* The same attention for an EXTEND batch over cached rows: request b's new tokens are rows
 * qo_indptr[b] .. qo_indptr[b + 1] - 1 of q [T, hq, rank + rope] / o [T, hq, rank]; token i of the request
 * attends to the first (seq_lens[b] - n_new_b) + i + 1 rows of the request's req_to_token row (causal; the
 * new tokens' latent rows are in kv_buffer already).  Four consecutive tokens of a request share one pass
 * over the rows (64 query rows per fetched row instead of the 16 of a per-token decode pass).
 * max_extend_len = max_b n_new_b (host knowledge, sizes the grid).  Needs rank 512 + rope 64 and a 16-bit
 * dtype (mi355_extend_attention_mla_supported returns 1); other shapes: one mi355_decode_attention_mla item
 * per new token gives the same result.
 * replaces: attn_mqa's forward_extend in DeepseekV2AttentionMLA.forward_absorb
 *           (python/sglang/srt/models/deepseek_v2.py:1264-1335, layers/attention/triton_backend.py:632-685). */
int mi355_extend_attention_mla_supported(int hq, int kv_lora_rank, int rope_dim, int dtype);
int mi355_extend_attention_mla(const void* q, const void* kv_buffer, void* o, const int32_t* req_to_token,
                               const int64_t* req_pool_indices, const int64_t* seq_lens,
                               const int32_t* qo_indptr, int64_t table_stride, int64_t bs,
                               int64_t max_extend_len, int hq, int kv_lora_rank, int rope_dim,
                               int64_t q_stride, int64_t num_slots, float sm_scale, float logit_cap,
                               int dtype, void* stream);

/* ---- the small steps of an MLA decode layer between its projections, one launch each ------------------
 * mi355_mla_latent_norms: a row of fused_qkv_a_proj_with_mqa's output is [q_a (q_rank) | c_kv (kv_rank) | k_pe
 * (rope_dim)].  Given x [rows, >= width] (16-bit, row stride x_row_stride) OR that GEMM's split-K partials
 * part [nsplit, rows, width] fp32 (with part_x_scale [rows] / part_w_scale [width] when the GEMM was fp8 W8A8),
 * writes q_out [rows, q_rank] = RMSNorm(q_a) * q_weight, kv_out [rows, kv_rank] = RMSNorm(c_kv) * kv_weight and
 * pe_out [rows, rope_dim] = k_pe, all dense.  q_rank == 0 (a model without q_lora: x is kv_a_proj_with_mqa's
 * output) skips the q part (q_out, q_weight NULL).  Same bits as split + contiguous + mi355_rmsnorm per part.
 * q_fp8_out [rows, q_rank] e4m3fn + q_fp8_scale [rows] (both or neither): also the per-token fp8 quantisation
 * of q_out's rows (= mi355_per_token_quant_fp8 on them), for an fp8 W8A8 q_b_proj.
 * Needs a 16-bit dtype, q_rank <= 2048, kv_rank + rope_dim <= 2048, all three % 8 == 0.
 * replaces: q_a_layernorm / kv_a_layernorm around the split of fused_qkv_a_proj_with_mqa in
 *           DeepseekV2AttentionMLA.forward_absorb_prepare (python/sglang/srt/models/deepseek_v2.py:1193-1232). */
int mi355_mla_latent_norms(void* q_out, void* kv_out, void* pe_out, const void* x, int64_t x_row_stride,
                           const float* part, int nsplit, const float* part_x_scale, const float* part_w_scale,
                           const void* q_weight, const void* kv_weight, uint8_t* q_fp8_out, float* q_fp8_scale,
                           int64_t rows, int q_rank, int kv_rank, int rope_dim, float eps, int dtype, void* stream);

/* mi355_mla_absorb: out[m, h, 0..n) = sum_k a[m, h, k] * w_t[h, j, k] for m < M <= 64 rows and every head, fp32
 * accumulation, rounded once to the 16-bit dtype; a[m, h, k .. k + tail) is copied to out[m, h, n .. n + tail).
 * a / out: element (m, h, f) at m * row_stride + h * head_stride + f; w_t [heads, n, k] contiguous (each head's
 * matrix TRANSPOSED, the reduction dimension contiguous).  Serves both absorbed products of MLA decode:
 * q_nope . w_kc with q's rope part carried along (n = kv_lora_rank, k = qk_nope_head_dim, tail = rope) and
 * attn . w_vc written straight into o_proj's input layout (n = v_head_dim, k = kv_lora_rank, tail = 0,
 * out_head_stride = v_head_dim).  Needs n % 16 == 0, k % 32 == 0, tail % 8 == 0, strides % 8 == 0.
 * replaces: torch.bmm(q_nope.transpose(0, 1), w_kc) + the concatenation with q_pe, and
 *           torch.bmm(attn_output.transpose(0, 1), w_vc) + the transpose back
 *           (python/sglang/srt/models/deepseek_v2.py:1234-1262, 1296-1333). */
int mi355_mla_absorb(void* out, const void* a, const void* w_t, int64_t m, int heads, int n, int k, int tail,
                     int64_t a_row_stride, int64_t a_head_stride, int64_t out_row_stride, int64_t out_head_stride,
                     int dtype, void* stream);

/* mi355_mla_prefill_assemble: the glue of MLA's multi-head prefill in one launch.  In place: q [tokens, heads, nope +
 * rope] (row stride q_stride) gets its rope part rotated at positions[t].  Out: k [tokens, heads, nope + rope] =
 * [kv's nope part | rope(k_pe[t])] and v [tokens, heads, v_dim] (both dense) from kv [tokens, heads, nope + v_dim]
 * (kv_b_proj's output, dense); the latent row [c_kv[t] (kv_lora_rank, the normed latent) | rope(k_pe[t])] is written to
 * kv_cache[out_cache_loc[t]] (16-bit rows, or e4m3fn with mi355_set_kv_buffer's arithmetic under dtype | MI355_KV_FP8).
 * cos_sin_cache [max_pos, rope]: cos | sin, fp32 or the tensor dtype; interleaved pairs, or neox pairs when is_neox.
 * Same bits as mi355_rope_inplace on (q rope part, k_pe) + the copies + mi355_set_kv_buffer.
 * replaces: the rotary_emb call, the k / v assembly and set_kv_buffer of DeepseekV2AttentionMLA.forward_normal
 *           (python/sglang/srt/models/deepseek_v2.py:1131-1179). */
int mi355_mla_prefill_assemble(void* q, void* k, void* v, const void* kv, const void* k_pe, int64_t k_pe_stride,
                               const void* c_kv, int64_t c_kv_stride, const int64_t* positions, const void* cos_sin_cache,
                               int cache_dtype, int is_neox, void* kv_cache, const int64_t* out_cache_loc, int64_t tokens,
                               int heads, int nope_dim, int rope_dim, int v_dim, int kv_lora_rank, int64_t q_stride,
                               int64_t num_slots, int dtype, void* stream);

/* ---- K12/K13 dense projection y = x @ W^T ------------------------------------------------------
 * x [M, K] (row stride x_stride), W [N, K] contiguous (nn.Linear layout), y [M, N] contiguous.
 * M <= 64 uses the weight-streaming MFMA kernel (bf16/f16); fp32 uses an exact-fp32 tiled kernel
 * for any M.  out_dtype = dtype, or MI355_F32 to get T-rounded values widened to fp32 (the
 * `.float()` of LogitsProcessor).  bias may be NULL.
 * Workspace contract (split-K): `workspace` must hold mi355_linear_workspace_bytes(M,N,K) bytes,
 * 16-B aligned.  Its first 4096 bytes are arrival counters: the caller zeroes them ONCE when it
 * allocates the buffer, every launch leaves them zero again, the rest needs no initialisation.
 * Launches that share a workspace must be ordered on one stream; nothing else may write to it.
 * replaces: UnquantizedLinearMethod.apply / F.linear (layers/quantization/unquant.py:110-122),
 *           LogitsProcessor._get_logits matmul (layers/logits_processor.py:453-455,490). */
int64_t mi355_linear_workspace_bytes(int64_t m, int64_t n, int64_t k);
int mi355_linear(void* y, const void* x, const void* w, const void* bias, int64_t m, int64_t n,
                 int64_t k, int64_t x_stride, void* workspace, int64_t workspace_bytes,
                 int dtype, int out_dtype, void* stream);

/* Decode-time weight layout.  mi355_pack_weight re-orders an [N, K] 16-bit weight into MFMA
 * fragment order [ceil(N/64)][K/32][4][64][8] (mi355_packed_weight_elems(n,k) elements, zero
 * padded): one wave-instruction reads a contiguous 1-KiB block, the four waves of a workgroup
 * read 4 KiB side by side, and a workgroup walks K as one contiguous stream;
 * mi355_linear_packed is mi355_linear (M <= 64) on such a weight.  The packing plays the role of
 * the reference's process_weights_after_loading hook, which repacks weights for its own CPU GEMM
 * (python/sglang/srt/layers/amx_utils.py:_amx_process_weight_after_loading,
 *  layers/quantization/unquant.py:105-108). */
int64_t mi355_packed_weight_elems(int64_t n, int64_t k);
int mi355_pack_weight(void* dst, const void* src, int64_t n, int64_t k, int dtype, void* stream);
int mi355_linear_packed(void* y, const void* x, const void* w_packed, const void* bias, int64_t m,
                        int64_t n, int64_t k, int64_t x_stride, void* workspace,
                        int64_t workspace_bytes, int dtype, int out_dtype, void* stream);

/* Deferred split-K reduction.  mi355_linear_packed_partials runs the packed skinny GEMM of
 * mi355_linear_packed but leaves each K split's fp32 partial in part[split][m][n] (row-major,
 * mi355_linear_packed_splits(m, n, k) splits; 0 = shape not served: needs K % 256 == 0, N < 65536)
 * for the kernel that consumes y to sum: mi355_fused_add_rmsnorm_partials(x, residual, weight,
 * part, nsplit, ...) computes y = T(sum over splits, in split order) and then exactly what
 * mi355_fused_add_rmsnorm does with x = y (x receives the normalised rows).  Bit-identical to the
 * GEMM followed by the norm; saves the GEMM's arrival ticket and second pass.  No counterpart in
 * the reference (its row-parallel GEMMs and norms are separate library calls, layers/linear.py:1277-1303,
 * layers/layernorm.py:76-78). */
/* The same deferred reduction for the fp8 W8A8 GEMM (mi355_fp8_scaled_mm_packed): raw fp32
 * accumulator partials [splits][m][n]; the consumer applies (sum * x_scale[m]) * w_scale[n] and
 * rounds once.  mi355_rmsnorm_ex is the general row norm behind it: x or split-K partials (with
 * optional fp8 scales) in, optional fused add of `residual` (updated in place), normalised rows to
 * `out`, and optionally their per-token fp8 quantisation (q_out [rows, hidden] e4m3fn + q_scale
 * [rows]; identical to mi355_per_token_quant_fp8 applied to `out`) for the fp8 GEMM that follows.
 * mi355_silu_and_mul_quant_fp8 = mi355_silu_and_mul followed by mi355_per_token_quant_fp8 in one
 * launch (the 16-bit activation is never written); q_stride as in mi355_per_token_quant_fp8.
 * replaces: apply_fp8_linear's quantise-then-GEMM pairs (layers/quantization/fp8_utils.py:676-749)
 *           around RMSNorm / SiluAndMul on the decode path; no fused counterpart in the reference. */
int mi355_fp8_packed_splits(int64_t m, int64_t n, int64_t k);
int mi355_fp8_packed_partials(float* part, const uint8_t* xq, const uint8_t* wq_packed, int64_t m, int64_t n,
                              int64_t k, void* stream);
int mi355_rmsnorm_ex(void* out, const void* x, void* residual, const void* weight, const float* part,
                     int nsplit, const float* part_x_scale, const float* part_w_scale, uint8_t* q_out,
                     float* q_scale, int64_t rows, int64_t hidden, float eps, int dtype, void* stream);
int mi355_silu_and_mul_quant_fp8(uint8_t* q_out, float* q_scale, const void* x, int64_t rows, int64_t d,
                                 int64_t q_stride, int dtype, void* stream);
int mi355_linear_packed_splits(int64_t m, int64_t n, int64_t k);
int mi355_linear_packed_partials(float* part, const void* x, const void* w_packed, int64_t m, int64_t n,
                                 int64_t k, int64_t x_stride, int dtype, void* stream);
int mi355_fused_add_rmsnorm_partials(void* x, void* residual, const void* weight, const float* part,
                                     int nsplit, int64_t rows, int64_t hidden, float eps, int dtype,
                                     void* stream);

/* gate_up projection with SiLU-and-mul fused into the epilogue: y[m, j] = T(silu(g_j)) * u_j with
 * g = x W_gate^T, u = x W_up^T each rounded to T first (the rounding points of the unfused ops).
 * w_packed = mi355_pack_weight of the [2d, K] matrix whose 16-row tile t holds gate rows
 * 8t..8t+7 followed by up rows 8t..8t+7; n = 2d; y is [m, d].  Needs K % 256 == 0, n < 65536.
 * replaces: LlamaMLP.forward gate_up_proj + act_fn (python/sglang/srt/models/llama.py:94-98,
 *           layers/activation.py:55-65). */
int mi355_linear_packed_silu_mul(void* y, const void* x, const void* w_packed, int64_t m, int64_t n,
                                 int64_t k, int64_t x_stride, void* workspace,
                                 int64_t workspace_bytes, int dtype, void* stream);

/* Large-M (prefill) projection: y = x @ W^T with 16-bit operands, fp32 accumulation on the matrix
 * cores, one rounding to the output dtype (= dtype).  x [M, K] (row stride x_stride, a multiple
 * of 8), W [N, K] contiguous, bias (may be NULL) [N].  epilogue 0: y [M, N] = T(acc + bias);
 * epilogue 1 (W = [gate; up], N = 2 I): y [M, I], y[m, j] = T(silu(T(g_j))) * T(u_j) with the rounding
 * points of the unfused GEMM + SiluAndMul pair.  Needs K % 64 == 0, K >= 128, N % 16 == 0
 * (mi355_linear_tiled_supported returns 1); no workspace, nothing allocated.
 * replaces: UnquantizedLinearMethod.apply / F.linear at prefill sizes
 *           (layers/quantization/unquant.py:110-122) and, with epilogue 1, LlamaMLP.forward's
 *           gate_up_proj + act_fn (models/llama.py:94-98, layers/activation.py:55-65). */
int mi355_linear_tiled_supported(int64_t m, int64_t n, int64_t k, int dtype, int epilogue);
int mi355_linear_tiled(void* y, const void* x, const void* w, const void* bias, int64_t m, int64_t n,
                       int64_t k, int64_t x_stride, int dtype, int epilogue, void* stream);

/* ---- K14 fp8 W8A8 ------------------------------------------------------------------------------
 * Per-token dynamic quantisation to OCP e4m3fn (max 448): s[r] = max(absmax(x[r,:]), 1e-10)/448,
 * q = clamp(x/s, +-448) -> fp8.  Row r of xq starts at byte r * q_stride (q_stride >= k); bytes k .. q_stride of
 * every row are set to zero, so that a reduction length the packed GEMM does not stream (K % 512 != 0) can be
 * rounded up with zero weights and zero activations behind it.
 * replaces: per_token_group_quant_fp8 with group = K (layers/quantization/fp8_kernel.py:115-155,
 *           fp8_utils.py:676-678), sgl_per_token_quant_fp8 (sgl-kernel/csrc/gemm/per_token_quant_fp8.cu). */
int mi355_per_token_quant_fp8(uint8_t* xq, float* scales, const void* x, int64_t rows, int64_t k,
                              int64_t x_stride, int64_t q_stride, int dtype, void* stream);

/* y[m,n] = T( (sum_k fp8(xq[m,k]) * fp8(wq[n,k])) * x_scale[m] * w_scale[n] (+ bias[n]) ),
 * wq is [N, K] e4m3fn (row per output channel), fp32 accumulation on the fp8 MFMA; bias (may be
 * NULL) has the output dtype; workspace must hold mi355_fp8_workspace_bytes(m,n,k) bytes.
 * replaces: apply_fp8_linear (layers/quantization/fp8_utils.py:510-749), fp8_scaled_mm
 *           (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu). */
int64_t mi355_fp8_workspace_bytes(int64_t m, int64_t n, int64_t k);
int mi355_fp8_scaled_mm(void* y, const uint8_t* xq, const uint8_t* wq, const float* x_scale,
                        const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                        void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* mi355_fp8_scaled_mm at prefill sizes: the 256 x 256-tile kernel of mi355_linear_tiled on fp8 operands
 * (v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: twice the 16-bit matrix rate), the
 * per-token / per-channel scales applied to the fp32 accumulator as (acc * x_scale[m]) * w_scale[n],
 * then bias, one rounding to out_dtype; epilogue 1 = SiLU-and-mul over wq = [gate; up] as in
 * mi355_linear_tiled.  xq [M, K] and wq [N, K] contiguous e4m3fn bytes.  Needs K % 128 == 0, K >= 256,
 * N % 16 == 0 (mi355_fp8_scaled_mm_tiled_supported returns 1); no workspace.
 * replaces: apply_fp8_linear's torch._scaled_mm / fp8_scaled_mm call at prefill sizes
 *           (layers/quantization/fp8_utils.py:676-749, sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu). */
int mi355_fp8_scaled_mm_tiled_supported(int64_t m, int64_t n, int64_t k, int out_dtype, int epilogue);
int mi355_fp8_scaled_mm_tiled(void* y, const uint8_t* xq, const uint8_t* wq, const float* x_scale,
                              const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                              int out_dtype, int epilogue, void* stream);

/* The qkv projection of a prefill batch with RoPE and the KV-cache write in the GEMM's epilogue:
 *   qkv = x @ w^T (w [(hq + 2 hkv) * 128, K], rows q | k | v), rounded to the dtype;
 *   neox RoPE at positions[m] over the whole head on the q and k heads, with the rounding points of
 *   mi355_rope_inplace (so the result is bit-identical to mi355_linear_tiled followed by
 *   mi355_rope_inplace with its cache write); qkv [M, (hq + 2 hkv) * 128] receives rotated q, rotated k
 *   and v; k_buffer[loc[m]] / v_buffer[loc[m]] ([slots, hkv, 128]) receive the rotated k and v rows.
 * x_scale != NULL selects the fp8 form: x, w are e4m3fn bytes with per-token x_scale[M] and per-channel
 * w_scale[N] applied as in mi355_fp8_scaled_mm_tiled (x_stride must then equal k).
 * Needs head_dim = rotary_dim = 128, is_neox, an even hq + 2 hkv, a 16-bit dtype, K % 64 == 0 and K >= 128
 * (fp8: K % 128 == 0, K >= 256), no bias; cos_sin_cache [max_pos, 128] fp32 or the dtype; loc int32 / int64.
 * replaces: LlamaAttention.forward's qkv_proj -> rotary_emb -> set_kv_buffer sequence at prefill
 *           (models/llama.py:187-199, layers/rotary_embedding.py:222-246, mem_cache/memory_pool.py:425-458). */
int mi355_qkv_proj_rope_tiled_supported(int64_t m, int64_t k, int hq, int hkv, int head_dim, int rotary_dim,
                                        int is_neox, int dtype, int fp8);
int mi355_qkv_proj_rope_tiled(void* qkv, const void* x, const void* w, const float* x_scale,
                              const float* w_scale, int64_t m, int64_t k, int64_t x_stride, int hq, int hkv,
                              int head_dim, const int64_t* positions, const void* cos_sin_cache,
                              int cache_dtype, int rotary_dim, int is_neox, void* k_buffer, void* v_buffer,
                              const void* loc, int loc_itype, int64_t num_slots, int dtype, void* stream);

/* ---- K11 sampling --------------------------------------------------------------------------------
 * ids[b] = first index of max(logits[b,:])   (torch.argmax tie rule)
 * replaces: Sampler.forward greedy branch (layers/sampler.py:72-76). */
int mi355_argmax(int64_t* ids, const float* logits, int64_t rows, int64_t vocab,
                 int64_t row_stride, void* stream);

/* The same arg-max for a handful of rows (a decode batch): each row is cut into chunks handled by separate
 * workgroups (one workgroup per row reads at what one CU can draw), the chunks' (maximum, first index)
 * pairs meet in a 64-bit cell per row through an order-independent atomic max, and the last chunk to
 * arrive writes ids[row].  Same indices as mi355_argmax, same tie rule, deterministic.
 * workspace: mi355_argmax_workspace_bytes(rows) bytes, 8-byte aligned, ZERO when first used (the launch
 * hands it back zeroed); calls that may overlap on one GPU need workspaces of their own.  NULL, or enough
 * rows to fill the chip, runs mi355_argmax's kernel.
 * replaces: torch.argmax(logits, -1) in Sampler.forward (python/sglang/srt/layers/sampler.py:74). */
int64_t mi355_argmax_workspace_bytes(int64_t rows);
int mi355_argmax_chunked(int64_t* ids, const float* logits, int64_t rows, int64_t vocab, int64_t row_stride,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* In place: logits[b,:] = softmax(logits[b,:] / temperature[b])
 * replaces: Sampler.forward lines 79-81 (layers/sampler.py). */
int mi355_softmax_temperature(float* logits, const float* temperatures, int64_t rows,
                              int64_t vocab, int64_t row_stride, void* stream);

/* Joint top-k / top-p / min-p filter + one draw per row by inverse CDF over the kept set taken
 * in index order, with the caller's uniform u[b] in [0,1).  A token is kept iff it is among the
 * top_k largest probabilities AND the probability mass strictly above it is <= top_p AND
 * (min_p == 0 or p >= min_p * pmax) -- the mask of top_k_top_p_min_p_sampling_from_probs_torch.
 * replaces: top_k_top_p_min_p_sampling_from_probs_torch (layers/sampler.py:154-178),
 *           top_k_top_p_sampling_from_probs / min_p_sampling_from_probs
 *           (sgl-kernel/python/sgl_kernel/sampling.py, common_extension.cc:314-333). */
int mi355_top_k_top_p_min_p_sample(int32_t* ids, const float* probs, const int32_t* top_ks,
                                   const float* top_ps, const float* min_ps, const float* uniform,
                                   int64_t rows, int64_t vocab, int64_t row_stride, void* stream);

/* fp8_scaled_mm for M <= 64 on a weight packed by mi355_pack_weight (pass the fp8 bytes as an
 * [N, K/2] 16-bit matrix: the packed layout is defined on 16-byte fragments): the bf16 decode GEMM's
 * single-launch weight-streaming kernel with fp8 MFMA and the scale epilogue.  K % 512 == 0,
 * N < 65536.  Workspace contract as mi355_linear (first 4096 bytes zero at rest).
 * replaces: apply_fp8_linear (layers/quantization/fp8_utils.py:510-749) at decode batch sizes. */
int64_t mi355_fp8_packed_workspace_bytes(int64_t m, int64_t n, int64_t k);
int mi355_fp8_scaled_mm_packed(void* y, const uint8_t* xq, const uint8_t* wq_packed, const float* x_scale,
                               const float* w_scale, const void* bias, int64_t m, int64_t n, int64_t k,
                               void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* ---- sum all-reduce over the GPUs of one xGMI hive (tensor parallelism) ------------------------
 * One process per GPU.  Every rank allocates one shared region, publishes its IPC handle, opens
 * its peers' handles, and builds a communicator over the n region addresses (its own at [rank]).
 * A call is one kernel launch (capturable in a HIP graph: it touches only `in`, `out` and the
 * regions): the input is copied into the rank's region, same-numbered workgroups of all ranks meet
 * at a flag barrier, partial tensors are pulled over xGMI and summed in fp32 in rank order 0..n-1
 * with one rounding, so every rank gets the same bits and both algorithms give the same bits.
 * `in == out` is allowed.  A barrier wait gives up after the communicator's timeout (20 s unless
 * mi355_ar_set_timeout_ms changed it; the value in force at launch -- or at capture -- applies): the
 * wait is counted, the call's output is filled with NaN (a sum over data that never arrived must not
 * look like a number) and mi355_ar_status reports it; the reference's kernels wait for ever
 * instead (custom_all_reduce_hip.cuh:155-248).  Calls of one communicator must be issued in the same order, with the same count,
 * dtype and algo, on every rank, and must not overlap each other.
 * replaces: init_custom_ar / all_reduce_reg / all_reduce_unreg / dispose / meta_size /
 *           allocate_meta_buffer / get_meta_buffer_ipc_handle (sgl-kernel/csrc/allreduce/
 *           custom_all_reduce.hip:13-51,89-110,112-117,146-172), the kernels
 *           cross_device_reduce_1stage / _2stage (custom_all_reduce_hip.cuh:260-343) and the
 *           algorithm choice of CustomAllreduce::allreduce (custom_all_reduce_hip.cuh:499-568). */
int64_t mi355_ar_region_bytes(int64_t max_bytes);           /* max_bytes: positive multiple of 256; -1 otherwise */
int mi355_ar_alloc_region(void** region, int64_t max_bytes); /* uncached device memory, zeroed; synchronous */
int mi355_ar_free_region(void* region);
int mi355_ar_ipc_handle_bytes(void);                         /* 64 */
int mi355_ar_get_ipc_handle(const void* region, uint8_t* handle);
int mi355_ar_open_ipc_handle(const uint8_t* handle, void** region);
int mi355_ar_close_ipc_handle(void* region);
/* regions[r]: rank r's region as mapped in this process; world in {2, 4, 6, 8} */
int mi355_ar_init(void** comm, void* const* regions, int rank, int world, int64_t max_bytes);
int mi355_ar_dispose(void* comm);
int mi355_ar_set_timeout_ms(void* comm, int64_t ms);        /* 1 .. 3 600 000; applies to later launches / captures */
/* 1 = one-shot, 2 = two-shot (reduce-scatter + all-gather) for `bytes` on `world` ranks */
int mi355_ar_choose_algo(int world, int64_t bytes);
/* count * element size: a multiple of 16, at most max_bytes; algo 0 = mi355_ar_choose_algo */
int mi355_ar_all_reduce(void* comm, void* out, const void* in, int64_t count, int dtype, int algo,
                        void* stream);
/* out[r * bytes ...] = rank r's `in` (bytes: a multiple of 16, at most max_bytes; out holds
 * world * bytes; not in place).  Same launch structure and ordering rules as the all-reduce.
 * replaces: GroupCoordinator.all_gather on the logits (distributed/parallel_state.py:528-597) at
 *           decode sizes; the reference has no custom kernel for it and goes through RCCL. */
int mi355_ar_all_gather(void* comm, void* out, const void* in, int64_t bytes, void* stream);
/* In place: x[rows, hidden] holds this rank's partial sums on entry and
 * RMSNorm(sum over ranks + residual) * weight on return; residual becomes sum + residual.  Bit-identical
 * to mi355_ar_all_reduce(x) followed by mi355_fused_add_rmsnorm(x, residual, weight).
 * hidden * element size a multiple of 16, at most 2048 16-byte vectors per row; rows * hidden within max_bytes.
 * replaces: RMSNorm.forward_with_allreduce_fusion (layers/layernorm.py:166-190). */
int mi355_ar_fused_add_rmsnorm(void* comm, void* x, void* residual, const void* weight, int64_t rows,
                               int64_t hidden, float eps, int dtype, void* stream);
/* waits for `stream`, then MI355_OK, or MI355_ERR_LAUNCH if any barrier wait has timed out so far */
int mi355_ar_status(void* comm, void* stream);
/* the same answer WITHOUT waiting for any work stream (the counter is read over a private
 * non-blocking stream): for a caller that has already waited for the step it asks about and has
 * the next one in flight (overlap scheduling). */
int mi355_ar_poll_status(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* MI355_HOTPATH_H_ */
