/* fvk_amd.h — C ABI of libfvk_amd.so: the MI355X (gfx950 / CDNA4) kernels behind FastVideo's
 * per-step Wan video-DiT path.
 *
 * Drop-in boundary.  The reference's compiled boundary is the pybind11 module
 * `fastvideo_kernel._C.fastvideo_kernel_ops` (fastvideo-kernel/csrc/common_extension.cpp:42-69,
 * functions over torch::Tensor) plus eager PyTorch layer code (fastvideo/layers, fastvideo/attention).
 * This library replaces what is underneath both with plain pointers + sizes: no torch types, no
 * allocation, no host synchronisation; every call enqueues on the caller's `hipStream_t` (passed as
 * `void* stream`, NULL = default stream) and is re-entrant per stream.  All pointers are DEVICE
 * pointers unless the name ends in `_host`.  Each function returns 0 on success or a negative FVK_ERR_*;
 * `fvk_last_error()` returns a thread-local message (the Python host raises RuntimeError from it —
 * the analogue of TORCH_CHECK in fastvideo-kernel/csrc/attention/st_attn_h100.cu:386-411).
 *
 * bf16 tensors are raw 16-bit storage (`void*`), fp32 tensors `float*`, indices `int32_t*`.
 * Citations `ref:` are into /root/reference.
 */
#ifndef FVK_AMD_H
#define FVK_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FVK_OK 0
#define FVK_ERR_ARG (-1)     /* unsupported shape / null pointer / bad flag   */
#define FVK_ERR_LAUNCH (-2)  /* hipLaunch / hip runtime error                  */

const char* fvk_last_error(void);
int fvk_abi_version(void);                 /* bumps when a signature changes or entry points are added (5 = round 3, 6 = round 4, 7 = round 5's fvk_gemm_vt_bf16 + round 6's fvk_mfma_sustained_probe_bf16, 8 = fvk_attn_block_sparse_ws_bf16 / _workspace_bytes, fvk_block_mean_gather_bf16) */
int fvk_device_arch(char* buf, int len);   /* gcnArchName of the current device ("gfx950...") */
int fvk_is_probe_build(void);              /* 0: the product library; 1: the measurement build (scripts/probes/libfvk_probe.so) */
/* Integer knobs for within-process A/B measurements (scripts/microbench.py); 0 = shipped configuration.
 * The PRODUCT library contains the shipped configuration only: it accepts the known names with value 0 and refuses any other value
 * (FVK_ERR_ARG).  The non-shipping kernels, schedules and timing ablations live in the measurement build of the same sources
 * (-DFVK_PROBE_BUILD + scripts/probes/{attn_pp,attn_vsa}.hip -> scripts/probes/libfvk_probe.so; FVK_PROBE_LIB=1 makes the Python
 * binding load it), where:
 *   "gemm_impl": 0 auto (gemm_w1 for K % 128 == 0, else gemm_ph / gemm_pp; fp8: gemm_w1 for K % 256 == 0), 1 force the 128x128
 *                register-staged kernel, 2 / 3 gemm_pp (also its fp8 kernel), 4 + 8 * VAR gemm_ph variants (228 = its shipped schedule),
 *                5 + 8 * VAR gemm_w1 variants (125 / 1149 = the shipped configuration without / with streaming stores)
 *   "vae_conv_impl": 0 auto (halo-reuse kernel for 3x3 spatial taps), 1 force the per-tap gather kernel, 2 lockstep 96-channel schedule
 *   "attn_impl": 0 auto (attn_w16 from 2048 keys, else the 8-wave kernel attn_pp2), 1 force the 4-wave kernel, 2..98 attn_pp, 99.. attn_pp2
 *                and its variants / ablations, 200.. attn_w64 and variants, 300.. attn_w16 and variants (31x = timing ablations)
 *   "vsa_impl": 1 block-per-row top-k kernel */
int fvk_set_tunable(const char* name, int value);

/* ------------------------------------------------------------------ norm / modulate family (HBM-bound)
 * ref: fastvideo/layers/layernorm.py:115-125 (FP32LayerNorm), :128-213 (ScaleResidualLayerNormScaleShift),
 *      :216-273 (LayerNormScaleShift), :91-109 (ScaleResidual); call sites fastvideo/models/dits/wanvideo.py:393,
 *      :414-431, :746-756.  One fused pass:
 *          r   = residual ? residual + x * gate : x          (fp32; gate NULL = 1)
 *          r   = bf16(r)            if FVK_LN_ROUND_RESIDUAL  (bf16+bf16 add of the cross-attn path)
 *          n   = LayerNorm_fp32(r) [* ln_w + ln_b]
 *          n   = bf16(n)            if FVK_LN_ROUND_NORM      (FP32LayerNorm on a bf16 input casts back)
 *          out = bf16(n * mul + add)                          (mul = 1+scale, add = shift; NULL = identity)
 *          res_out = bf16(r)                                  (optional)
 * x, residual, out, res_out: bf16 [M, d] contiguous.  gate, mul, add: fp32 [M / rows_per_batch, d].
 * ln_w, ln_b: fp32 [d].  d % 8 == 0, d <= 8192. */
#define FVK_LN_ROUND_RESIDUAL 1
#define FVK_LN_ROUND_NORM 2
int fvk_ln_modulate_bf16(const void* x, const void* residual, const float* gate, const float* ln_w,
                         const float* ln_b, const float* mul, const float* add, void* res_out, void* out,
                         int M, int d, int rows_per_batch, float eps, int flags, void* stream);

/* fvk_ln_modulate_bf16 that ALSO (out != NULL) or ONLY (out == NULL) writes the per-token e4m3 quantisation of its bf16 output row:
 * q_out [M, d] e4m3 bytes, q_scale [M] fp32 — byte- and bit-identical to fvk_fp8_quantize_bf16(rowwise = 1) applied to the bf16 output
 * (ref: _quantize_rowwise, fastvideo/layers/quantization/fp8_config.py:62-68; the fp8_channel linear that consumes the row then needs no
 * separate absmax + quantise passes). */
int fvk_ln_modulate_fp8_bf16(const void* x, const void* residual, const float* gate, const float* ln_w, const float* ln_b, const float* mul,
                             const float* add, void* res_out, void* out, void* q_out, float* q_scale, int M, int d, int rows_per_batch,
                             float eps, int flags, void* stream);
/* out = bf16(residual + x * gate) — ScaleResidual (ref: layernorm.py:91-109). gate fp32 [M/rows_per_batch, d]. */
int fvk_scale_residual_bf16(const void* residual, const void* x, const float* gate, void* out, int M, int d,
                            int rows_per_batch, void* stream);

/* QK-RMSNorm-across-heads and/or 3-D RoPE on rows of `width` = n_heads*head_dim bf16 elements.
 * ref: fastvideo/layers/layernorm.py:48-83 (RMSNorm.forward_native: fp32 normalise -> cast bf16 -> * weight),
 *      fastvideo/layers/rotary_embedding.py:105-135 (_apply_rotary_emb, interleaved pairs, fp32, cast back),
 *      call sites wanvideo.py:398-401 and fastvideo/attention/layer.py:130-132.
 * Processes `n_tensors` (1..3) tensors per launch: in[i]/out[i] rows start `in_stride`/`out_stride`
 * elements apart (so q,k slices of a fused [M,3d] QKV buffer work in place of separate tensors).
 * weight[i]: bf16 [width] or NULL (skip norm).  cos/sin: fp32 [seq_len, head_dim] or NULL (skip RoPE);
 * row m uses position (m + pos_offset) % seq_len (pos_offset = first global token of a sequence-parallel shard).
 * head_dim % 8 == 0. */
int fvk_rmsnorm_rope_bf16(const void* const* in, void* const* out, const void* const* weight, int n_tensors,
                          const float* cos, const float* sin, int M, int width, int head_dim, int seq_len,
                          int pos_offset, long in_stride, long out_stride, float eps, void* stream);

/* The same pass with SCATTERED output rows: row m of tensor i lands in row row_map[i][m] of out[i] (row_map[i] NULL = row m; a negative
 * entry drops the row).  Folds the tile-major permutation of the sparse attention paths — VideoSparseAttentionImpl.tile
 * (fastvideo/attention/backends/video_sparse_attn.py:254-264) and the sliding-tile layout (fastvideo_kernel/ops.py:21-62 expects
 * tile-major q, k, v) — into the norm / RoPE pass that writes q and k anyway: no separate gather, no extra S x d round trip. */
int fvk_rmsnorm_rope_scatter_bf16(const void* const* in, void* const* out, const void* const* weight, int n_tensors,
                                  const float* cos, const float* sin, int M, int width, int head_dim, int seq_len,
                                  int pos_offset, long in_stride, long out_stride, float eps,
                                  const int32_t* const* row_map, void* stream);

/* Sequence-parallel exchange #1 packing fused into the QK-norm / RoPE pass (one read of the fused QKV buffer, no torch.cat):
 * q,k: RMS-norm across heads (weights wq/wk, NULL = skip) + RoPE (cos/sin NULL = skip); v: copied.  Row m of the rank's shard
 * (Sl rows, `in_stride` elements apart, `width` = heads*head_dim columns) is split into G head groups of W = width/G columns; group
 * g goes to every rank rp = g + G*u' (u' < U) of the 2-D Ulysses grid (fastvideo_amd/distributed.py) as the message row
 *     send[((rp*Sl + m)*3 + slot)*W .. +W],  slot 0 = K, 1 = V, 2 = Q
 * i.e. send is [G*U ranks, Sl, 3, W] bf16: equal-size per-peer blocks for ONE all_to_all_single, and the received buffer
 * [ranks*Sl, 3, heads/G, head_dim] is a fused-QKV-shaped tensor the attention kernels read in place through strides.
 * replaces: torch.cat([q,k,v]) + all_to_all_4D's transpose().contiguous() pack
 *   (ref: fastvideo/attention/layer.py:117-124, fastvideo/distributed/device_communicators/base_device_communicator.py:147-165). */
int fvk_qkv_norm_rope_pack_bf16(const void* q, const void* k, const void* v, const void* wq, const void* wk, const float* cos,
                                const float* sin, void* send, int Sl, int width, int head_dim, int seq_len, int pos_offset,
                                long in_stride, int G, int U, float eps, void* stream);
/* The same pass writing TWO send buffers — the first `heads_a` heads of every head group into send_a [G*U, Sl, 3, heads_a*head_dim], the rest into
 * send_b [G*U, Sl, 3, (heads/G - heads_a)*head_dim] — for the pipelined exchange (two head chunks: chunk B's all-to-all overlaps chunk A's attention,
 * chunk A's output exchange chunk B's attention; fastvideo_amd/distributed.py: attention_packed_pipelined).  Heads are independent in attention, so the chunks' outputs are
 * the un-chunked output's columns bit for bit. */
int fvk_qkv_norm_rope_pack2_bf16(const void* q, const void* k, const void* v, const void* wq, const void* wk, const float* cos,
                                 const float* sin, void* send_a, void* send_b, int heads_a, int Sl, int width, int head_dim, int seq_len,
                                 int pos_offset, long in_stride, int G, int U, float eps, void* stream);
/* The same pass with a FOURTH per-token, per-head tensor copied along (the VSA compress gate `to_gate_compress(x)`, which the reference
 * sends through the same all-to-all as q, k, v: fastvideo/attention/layer.py:172-245): message row = [K | V | Q | gate], send is
 * [G*U ranks, Sl, 4, W].  The gate travels with Q — to every rank of its head group's column — so video-sparse attention runs on any
 * G x U grid (12 heads on 8 GPUs: G4 x U2), where the reference needs num_heads % sp_size == 0. */
int fvk_qkvg_norm_rope_pack_bf16(const void* q, const void* k, const void* v, const void* gate, const void* wq, const void* wk,
                                 const float* cos, const float* sin, void* send, int Sl, int width, int head_dim, int seq_len,
                                 int pos_offset, long in_stride, int G, int U, float eps, void* stream);

/* V[b, s, h, :] at v + b*in_batch_stride + s*in_stride + h*in_head_stride (elements) -> Vt [B, H, D, S_pad] bf16,
 * S_pad = a multiple of 64 >= S (the host wrapper uses round_up(S, 128): whole tiles of the 128-key-tile kernel), pad columns zero.  Within every aligned group of 16 keys the key order is
 * permuted by swapping bits 2 and 3 of the in-group index (the MFMA-B-operand order of fvk_attn_*; see
 * DESIGN.md "Vt layout").  D == 128. */
int fvk_v_transpose_bf16(const void* v, void* vt, int B, int S, int H, int D, long in_stride, long in_batch_stride,
                         long in_head_stride, int S_pad, void* stream);

/* The same with GATHERED source rows: key position p of Vt takes row src_rows[p] of v (int32 [S_pad]; negative = a zero column) — V goes
 * from token order straight to the tile-major, zero-padded V^T the sparse kernels read (the `tile` gather of V folded in). */
int fvk_v_transpose_gather_bf16(const void* v, void* vt, const int32_t* src_rows, int B, int S, int H, int D, long in_stride,
                                long in_batch_stride, long in_head_stride, int S_pad, void* stream);

/* ------------------------------------------------------------------ dense GEMM (MFMA-bound)
 * ref: fastvideo/layers/linear.py:146-156 (UnquantizedLinearMethod.apply = F.linear), mlp.py:47-51,
 *      wanvideo.py:394-396, :411, :430-431.
 *   y   = bf16( sum_k x[m,k] * w[n,k] + bias[n] )        x bf16 [M,K] (row stride lda), w bf16 [N,K]
 *   FVK_EPI_GELU_TANH : out = bf16(gelu_tanh(float(y)))
 *   FVK_EPI_SILU      : out = bf16(silu(float(y)))
 *   FVK_EPI_RESIDUAL_GATE : out = bf16(float(residual[m,n]) + float(y) * gate[m / rows_per_batch, n])
 *                           (gate NULL = 1; = ScaleResidual fused into the out-projection)
 * K % 64 == 0 (K % 32 == 0 on the large-M path).  Any M, N (tails masked).  out row stride ldc. */
#define FVK_EPI_NONE 0
#define FVK_EPI_GELU_TANH 1
#define FVK_EPI_SILU 2
#define FVK_EPI_RESIDUAL_GATE 3
#define FVK_EPI_DIV 4 /* out = bf16(float(y) / epi_scalar) — the `scores / dim**0.5` of fastvideo_kernel/ops.py:113 */
int fvk_gemm_bf16(const void* x, const void* w, const void* bias, void* out, int M, int N, int K, long lda,
                  long ldc, int epilogue, const void* residual, const float* gate, int rows_per_batch,
                  void* stream);
/* V projection written STRAIGHT into the attention kernels' V^T layout (round 5; replaces `to_v` = F.linear, linear.py:146-156, followed by the
 * layout pass fvk_v_transpose_bf16):
 *   vt[b, n, p] = bf16( sum_k wv[n,k] * x[b, key(p), k] + bias[n] )   for key(p) < S,   0 for the padding positions up to S_pad
 * wv bf16 [d, K] (the V rows of the fused QKV weight), x bf16 [B, S, K] (dense rows, batch stride x_bstride elements), bias bf16 [d] or NULL,
 * vt bf16 [B, d / 128, 128, S_pad]; key(p) = p with bits 2 and 3 of its index exchanged (fvk_v_transpose_bf16's order).  Same products, same
 * k order and the same rounding points as fvk_gemm_bf16 followed by fvk_v_transpose_bf16 — the MFMA operands trade places (bit-identical:
 * tests/test_gpu_kernels.py).  Needs K % 128 == 0, d % 128 == 0 and > 128, S % 8 == 0, S_pad % 64 == 0 in [S, round_up(S, 256)]. */
int fvk_gemm_vt_bf16(const void* wv, const void* x, const void* bias, void* vt, int B, int S, int d, int K, long ldx,
                     long x_bstride, int S_pad, void* stream);
/* `batch` independent GEMMs (no bias): operand b at x + b*x_bstride etc.  Used for the VSA coarse scores
 * q_c · k_c^T / sqrt(D) per (batch, head) (ref: fastvideo_kernel/ops.py:113).  epilogue NONE or DIV. */
int fvk_gemm_bf16_batched(const void* x, const void* w, void* out, int batch, int M, int N, int K, long lda, long ldc,
                          long x_bstride, long w_bstride, long out_bstride, int epilogue, float epi_scalar, void* stream);

/* ------------------------------------------------------------------ fp8 (OCP e4m3fn) linear path (MFMA-bound GEMM, HBM-bound quantisation)
 * ref: fastvideo/layers/quantization/fp8_config.py:55-68 (_quantize_tensorwise/_rowwise), :119-158 (FP8QuantizeMethod.apply),
 *      :211-245 (convert_model_to_fp8 — the same arithmetic applied once to the weights).
 * fvk_fp8_quantize_bf16:  absmax over the tensor (rowwise = 0) or per row (rowwise = 1: per token / per output channel);
 *      scale = max(absmax / 448, 1/(448*512)) -> scale[1] or scale[M] (fp32);
 *      q = e4m3fn( clamp( bf16( float(x) / float(bf16(scale)) ), -448, 448 ) ) -> q [M, K] bytes (row stride K).
 *      absmax_scratch: device fp32 [1] or [M] (overwritten).  K, lda multiples of 8.
 * fvk_gemm_fp8:  out = epilogue( bf16( bf16( (x_fp8 · w_fp8^T)_fp32 * scale_a * scale_b ) + bias ) )  — the two roundings of
 *      torch._scaled_mm(out_dtype=bf16) followed by `out + bias`; epilogues as fvk_gemm_bf16 (NONE, GELU_TANH, SILU, RESIDUAL_GATE).
 *      x_fp8 [M,K], w_fp8 [N,K] e4m3 bytes, K % 64 == 0; scale_a [1] or [M] (a_rowwise), scale_b [1] or [N] (b_rowwise). */
int fvk_fp8_quantize_bf16(const void* x, void* q, float* scale, float* absmax_scratch, int M, int K, long lda, int rowwise, void* stream);
int fvk_gemm_fp8(const void* x_fp8, const void* w_fp8, const float* scale_a, const float* scale_b, const void* bias, void* out, int M, int N,
                 int K, long ldc, int a_rowwise, int b_rowwise, int epilogue, const void* residual, const float* gate, int rows_per_batch,
                 void* stream);

/* ------------------------------------------------------------------ attention (MFMA-bound)
 * Flash-style forward, head_dim 128, non-causal, fp32 online softmax in the exp2 domain, P rounded to bf16
 * before P·V (ref numerics: block_sparse_attn_triton.py:124-158; st_attn_triton.py:60-89).
 * q [B, Sq, H, 128], k [B, Skv, H, 128] with explicit strides (elements): *_bs batch, *_ss sequence row,
 * *_hs head — so both the backend layout [B,S,H,D] (ref: fastvideo/attention/backends/abstract.py AttentionImpl.forward)
 * and the kernel-package layout [B,H,S,D] (ref: fastvideo_kernel/ops.py) are served without copies.
 * vt = fvk_v_transpose_bf16 output [B, H, 128, Skv_pad].  o has q's shape with its own strides.
 * lse: optional fp32 [B, H, Sq] = m*log2e*scale + log2(l) (block_sparse_attn_triton.py:155-157), or NULL. */
typedef struct {
    const void* q; const void* k; const void* vt; void* o; float* lse;
    int B, H, Sq, Skv, Skv_pad;
    long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, o_bs, o_ss, o_hs;
    float scale;
    int qk_dim; /* depth of the q·k contraction: 0 or 128 (the DiT), 384 = dense only: the VAE mid-block's single 384-wide head,
                   run as H = 3 output slices of 128 columns that share q and k (q_hs = k_hs = 0, o_hs = 128;
                   ref: WanAttentionBlock.forward, fastvideo/models/vaes/wanvae.py:479-507) */
} fvk_attn_args;

/* dense: ref fastvideo/attention/backends/sdpa.py:122-147 / flash_attn.py:247-345 (the path replaced).
 * lse of the long-key kernel (Skv >= 2048): the row sum is the sum of the bf16-rounded probabilities (the values the numerator uses), i.e.
 * accurate to ~2^-9 relative (3e-3 in log2 units), not to fp32 rounding. */
int fvk_attn_dense_bf16(const fvk_attn_args* a, void* stream);
/* The same call with the long-key kernel named: kernel 0 = the default (1), 1 = attn_w16 (16x16x32 MFMAs: less energy per FLOP, ~6 % more
 * matrix-pipe cycles), 2 = attn_w64 (32x32x16 MFMAs).  Same result to rounding (different summation order inside the MFMA; kernel 2 sums the
 * fp32 probabilities for lse).  Which is faster depends on the clock the device reaches in the CALLER'S launch sequence: back to back kernel 1
 * is ~5 % faster, between the GEMMs of a DiT block it is anywhere from 5 % faster to 5 % slower (profiles/r03_attn_context.md); a caller that
 * cares times both in place once.  Key axes below 2048 ignore the choice (8-wave kernel).
 * kernel 3 = attn_pp2 at ANY key length: the 8-wave kernel with the reference kernels' ONLINE softmax (running row max, rescale of O) — for a
 * parity-critical caller that wants the flash-attention rounding points (P rounded after subtracting the running maximum, as
 * flash_attn.py / st_attn_triton.py do) rather than the long-key kernels' fixed softmax reference; ~10 % slower at 32 760 keys. */
int fvk_attn_dense_kernel_bf16(const fvk_attn_args* a, int kernel, void* stream);
/* The same attention with the KEY axis cut into n_split runs of whole 128-key stages, each (query block, head, batch, run) its own workgroup,
 * and a merge pass: for grids too small to fill the 256 CUs — the per-rank shapes of sequence parallelism (SP = 8 on Wan2.1-1.3B: 3 heads x
 * 64 query blocks = 192 workgroups of 256 rows; 4 runs make 768 = three full rounds).  Workspace (device, overwritten): o_part fp32
 * [n_split, B, H, Sq, 128] (each run's normalised output), lse_part fp32 [n_split, B, H, Sq] (its base-2 log-sum-exp); the merge is
 * o = sum_r 2^(lse_r - max) o_r / sum_r 2^(lse_r - max), a->lse (optional) receives the merged LSE.  Sq >= 256, head_dim 128.
 * (No reference counterpart: flash-attn's split-KV decode path is the same idea; fastvideo/attention/backends/flash_attn.py calls it through
 * flash_attn_func.) */
int fvk_attn_dense_split_bf16(const fvk_attn_args* a, int n_split, float* o_part, float* lse_part, void* stream);

/* block-sparse (VSA sparse branch; sliding-tile windows on arbitrary canvases): query block i (q_block = 64 or 128 rows) attends KV
 * blocks q2k_idx[b,h,i,0..q2k_num[b,h,i]) (64 keys each, of which the first kv_block_sizes[j] are valid).
 * ref: fastvideo-kernel/csrc/attention/block_sparse_h100.cu:66-272, triton_kernels/block_sparse_attn_triton.py:32-160.
 * q2k_idx int32 [B,H,Nq,max_kv], q2k_num int32 [B,H,Nq] with Nq = Sq / q_block, kv_block_sizes int32 [Nkv]. Skv multiple of 64.
 * A list may be empty (q2k_num = 0: the block's output rows are written as zeros) and a listed block may have size 0 (as the FIRST block of a
 * 64-row list it sends the rows through the exact pass: it would have set the softmax reference).  q_block 64 (round 6, attn_bs16.hip): one wave per list, lists read with scalar loads (any
 * length up to max_kv; ids outside [0, Skv / 64) are clamped).  q_block 128: block ids must be < 2^24; the first 2048 entries of a list are
 * kept in LDS (longer lists fall back to global reads past that point).
 * Every entry point taking fvk_attn_args refuses (FVK_ERR_ARG) a (batch, head) K slice whose extent reaches 4 GiB: the K / V^T
 * streams are addressed through 32-bit buffer descriptors. */
int fvk_attn_block_sparse_bf16(const fvk_attn_args* a, const int32_t* q2k_idx, const int32_t* q2k_num,
                               const int32_t* kv_block_sizes, int max_kv, int q_block, void* stream);
/* The same call with a device WORKSPACE (q_block 64; round 6): the launch runs one 4-list workgroup per CU, so a list count that is not a
 * multiple of 4 x CUs leaves the last round partly empty (Wan2.1-1.3B at 81f x 480p: 1 872 workgroups = 7.31 rounds of 256).  With a workspace of
 * fvk_attn_block_sparse_workspace_bytes(a, max_kv, q_block) bytes (0 = no split planned; 16-byte aligned, overwritten) the workgroups of the
 * last round walk their lists in 2-4 PARTS (whole blocks, an even count per part) on as many workgroups, each part's result is written un-merged
 * (normalised O as fp32 rows + base-2 LSE) and merged as fvk_attn_dense_split_bf16 does: o = sum_p 2^(lse_p - max) o_p / sum_p 2^(lse_p - max).
 * The rows of split lists agree with the unsplit call to rounding (a part's softmax reference is the row maximum of ITS first block), every
 * other row is bit-identical.  workspace NULL / too small: exactly fvk_attn_block_sparse_bf16.  No reference counterpart (flash-attn's
 * split-KV decode is the same idea). */
long fvk_attn_block_sparse_workspace_bytes(const fvk_attn_args* a, int max_kv, int q_block);
int fvk_attn_block_sparse_ws_bf16(const fvk_attn_args* a, const int32_t* q2k_idx, const int32_t* q2k_num, const int32_t* kv_block_sizes,
                                  int max_kv, int q_block, void* workspace, long workspace_bytes, void* stream);
/* The same attention for 64-row lists with the two lists of a workgroup's neighbouring query blocks (2p, 2p + 1) walked as ONE merged list
 * (round 4): fvk_vsa_union_lists merges the ascending lists of fvk_map_to_index into u_idx [B*H, ceil(nq/2), 2*max_kv] packed entries
 * (block id | valid keys << 22 | halves << 29) + u_num [B*H, ceil(nq/2)]; fvk_attn_block_sparse_union_bf16 walks them — a KV tile both blocks
 * selected is fetched once.  Output bit-identical to fvk_attn_block_sparse_bf16 (q_block 64) on the same lists.  max_kv <= 2048. */
int fvk_vsa_union_lists(const int32_t* q2k_idx, const int32_t* q2k_num, const int32_t* kv_block_sizes, int32_t* u_idx, int32_t* u_num, int rows,
                        int nq, int max_kv, void* stream);
int fvk_attn_block_sparse_union_bf16(const fvk_attn_args* a, const int32_t* u_idx, const int32_t* u_num, int max_u, void* stream);

/* block-sparse with SHARED lists — sliding-tile attention on arbitrary canvases: every rows_per_list consecutive query rows (one
 * sliding tile in tile-major order, 384 tokens for the reference's (6,8,8) tile) attend the same KV blocks, list i =
 * q2k_idx[b,h,i,0..q2k_num[b,h,i]) of 64-key blocks with kv_block_sizes valid keys each, exactly as above.  Same result as
 * fvk_attn_block_sparse_bf16 with the list repeated per 128-row block (tests/test_gpu_kernels.py), but 256 of a list's rows share one
 * workgroup on the dense kernel's ping-pong schedule (128-key steps = two listed blocks, each half masked by its own size); a 128-row
 * remainder (384 = 256 + 128) runs on the 4-wave kernel.
 * ref: fastvideo_kernel.sliding_tile_attention (fastvideo-kernel/python/fastvideo_kernel/ops.py:21-62), mask rule
 *      fastvideo-kernel/tests/support_flex_sta.py:29-59, kernels st_attn_triton.py:19-121 / csrc/attention/st_attn_h100.cu.
 * q2k_idx int32 [B,H,Nl,max_kv], q2k_num int32 [B,H,Nl], Nl = Sq / rows_per_list; rows_per_list a multiple of 128, >= 256; max_kv <= 4096.
 * q_rows_valid (optional, int32 [Nl]): real query rows at the head of each list's rows — 256- / 128-row groups that start at or past it
 * hold only padding and are written as zeros without touching K / V.
 * o_rows (optional, int32 [Sq]; rows_per_list a multiple of 256): query row r's output is stored at row o_rows[r] of o (negative:
 * dropped) instead of row r — the un-tiling gather (VideoSparseAttentionImpl.untile, video_sparse_attn.py:266-272) folded into the
 * store; o then has as many rows as the map addresses, not Sq. */
int fvk_attn_tile_lists_bf16(const fvk_attn_args* a, const int32_t* q2k_idx, const int32_t* q2k_num,
                             const int32_t* kv_block_sizes, int max_kv, int rows_per_list, const int32_t* q_rows_valid,
                             const int32_t* o_rows, void* stream);

/* sliding-tile attention: tokens in tile-major order, tile = tile_t*tile_h*tile_w tokens (multiple of 64),
 * canvas of (ct,ch,cw) tiles; head h uses window (win[3h], win[3h+1], win[3h+2]) tiles with the clamped
 * centre rule of ref: fastvideo-kernel/tests/support_flex_sta.py:29-59 ≡ st_attn_triton.py:158-176:
 *   kv tile in window  <=>  |clamp(q_tile, k/2, n-1-k/2) - kv_tile| <= k/2   per axis, INTEGER k/2
 * so window sizes may be odd or even (>= 1; an even k selects like k+1 — the reference's own test uses (3,1,10),
 * fastvideo-kernel/tests/test_sta.py:40).  win_host: HOST int32 [3*H].  No text tokens (Wan T2V self-attention has none). */
int fvk_attn_sta_bf16(const fvk_attn_args* a, int ct, int ch, int cw, int tile_tokens, const int32_t* win_host,
                      void* stream);

/* ------------------------------------------------------------------ VSA coarse stage + index construction
 * ref: fastvideo/attention/backends/video_sparse_attn.py:31-114, 170-189, 285-290;
 *      fastvideo_kernel/triton_kernels/fused_compress_topk.py:22-60 (block mean), :211-277 (top-k mask);
 *      triton_kernels/index.py:33-61 (map_to_index).  All integer outputs are bit-exact targets. */
/* host-side metadata (CPU, pure integer): fills perm/rev/non_pad/untile [T*H*W] and vbs [n_tiles]; any may be NULL. */
int fvk_vsa_build_metadata_host(int T, int H, int W, int tt, int th, int tw, int32_t* perm, int32_t* rev,
                                int32_t* vbs, int32_t* non_pad, int32_t* untile);
/* dst[b, dst_index[i], :] = src[b, src_index[i], :] for i < n (NULL index = identity); rows of `row_elems` bf16
 * (multiple of 8).  tile(): dst zero-initialised buffer, dst_index=non_pad, src_index=perm.
 * untile(): dst_index=NULL, src_index=untile_combined.  ref: video_sparse_attn.py:170-189, 285-290. */
int fvk_gather_rows_bf16(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int B, int n,
                         int row_elems, long src_batch_stride, long dst_batch_stride, void* stream);
/* same with explicit row strides (elements, multiples of 8): gathers straight out of a column block of the fused QKV(+gate) buffer
 * (src_row_stride = 3d or 4d) into a persistent, pre-zeroed tile buffer — no `.contiguous()` copy, no per-layer memset. */
int fvk_gather_rows_strided_bf16(const void* src, void* dst, const int32_t* src_index, const int32_t* dst_index, int B, int n,
                                 int row_elems, long src_row_stride, long dst_row_stride, long src_batch_stride, long dst_batch_stride,
                                 void* stream);
/* x [B,S_pad,H,D] with strides -> out [B,H,Nblk,D] bf16 = bf16(fp32 sum over 64 rows / vbs[blk]). */
int fvk_block_mean_bf16(const void* x, void* out, const int32_t* vbs, int B, int H, int n_blocks, int block, int D,
                        long x_bs, long x_ss, long x_hs, void* stream);
/* the same means over rows GATHERED on the fly (round 6): tile-major row p is row src_rows[p] of x (int32 [n_blocks * block]; negative = a
 * padding row, i.e. zeros) — tile() of video_sparse_attn.py:254-281 folded in; bit-identical to fvk_block_mean_bf16 on the gathered copy. */
int fvk_block_mean_gather_bf16(const void* x, void* out, const int32_t* vbs, const int32_t* src_rows, int B, int H, int n_blocks, int block,
                               int D, long x_bs, long x_ss, long x_hs, void* stream);
/* scores bf16 or fp32 [rows, n] -> mask uint8 [rows, n] with exactly min(topk,n) ones per row (bisection + first-come ties). */
int fvk_topk_mask(const void* scores, int scores_is_fp32, uint8_t* mask, int rows, int n, int topk, void* stream);
/* mask uint8 [rows, n] -> idx int32 [rows, n] (ascending, tail zero), num int32 [rows]. */
int fvk_map_to_index(const uint8_t* mask, int32_t* idx, int32_t* num, int rows, int n, void* stream);
/* row softmax over the last dim: in/out bf16 [rows, n] (fp32 math, one rounding) — VSA coarse attn weights. */
int fvk_softmax_rows_bf16(const void* in, void* out, int rows, int n, void* stream);
/* out[b,s,h,:] = bf16( bf16(float(out_c[b,h,s/block,:]) * float(gate[b,s,h,:])) + float(out_s[b,s,h,:]) )
 * ref: fastvideo_kernel/ops.py:120-133.  out_s/gate/out share strides (bs, ss, hs); out_c is [B,H,Nblk,D] contiguous. */
int fvk_vsa_combine_bf16(const void* out_c, const void* out_s, const void* gate, void* out, int B, int S, int H, int D,
                         int block, long bs, long ss, long hs, void* stream);
/* The same with the tile gather of `gate` and the un-tiling of the result folded in: tile-major row s of out_c / out_s belongs to token
 * token_of_row[s] (int32 [S]; negative = padding row, skipped); gate is read and out written at the TOKEN's row with their own strides
 * (elements): gate[b*g_bs + tok*g_ss + h*g_hs], out[b*o_bs + tok*o_ss + h*o_hs].
 * ref: VideoSparseAttentionImpl.tile / untile (fastvideo/attention/backends/video_sparse_attn.py:254-272) around fastvideo_kernel/ops.py:120-133. */
int fvk_vsa_combine_scatter_bf16(const void* out_c, const void* out_s, const void* gate, void* out, const int32_t* token_of_row,
                                 int B, int S, int H, int D, int block, long bs, long ss, long hs, long g_bs, long g_ss, long g_hs,
                                 long o_bs, long o_ss, long o_hs, void* stream);

/* ------------------------------------------------------------------ Wan VAE decode (causal 3-D conv; MFMA-bound convs, HBM-bound norm)
 * Activations are channels-last bf16 [frames, H, W, C].  ref: fastvideo/models/vaes/wanvae.py:160-207 (WanCausalConv3d),
 * :251-380 (WanResample), :383-462 (WanResidualBlock), :857-993 (WanDecoder3d), :1189-1215 (decode).
 * Implicit-GEMM convolution:  out[t,h,w,co] = bias[co] + sum_{dt,dh,dw,ci} w[co][((dt*KH+dh)*KW+dw)*Cin+ci] *
 *                                              in[slot(t+dt)][h+dh-KH/2][w+dw-KW/2][ci]      (zero outside the image)
 *   in      : ring of `ring` frames [ring, Hin, Win, Cin]; logical frame l lives in slot (ring_start + l) % ring.  For KT = 3
 *             logical frames 0,1 are the causal history (the reference's feat_cache / zero padding) and frame 2+t is the
 *             chunk's t-th frame; for KT = 1 logical frame t is the t-th frame.
 *   upsample2x: the input is at half resolution (Hin = H/2, Win = W/2) and nearest-exact 2x upsampling is folded into the
 *             gather (KT must be 1)  — WanUpsample + Conv2d of WanResample.
 *   output pixel (t,h,w) -> out + t*out_frame_stride + (h*W+w)*Cout (strides in elements; lets the two halves of an
 *             upsample3d time_conv interleave their frames, wanvae.py:354-356), same for residual with res_frame_stride.
 *   epilogue: 0 bias, 1 bias + residual (WanResidualBlock `x + h`), 2 final: out_f32[co*plane_stride + t*H*W + h*W + w] =
 *             clamp(acc + bias, -1, 1) in fp32 (the decoder's `.float().clamp(-1,1)` and NCTHW layout).
 * Cin % 32 == 0 (pad z_dim 16 -> 32 with zero weights); KT in {1,3}; KH = KW in {1,3}; Cout % 8 == 0 unless epilogue 2. */
int fvk_vae_conv_bf16(const void* in, const void* w, const void* bias, void* out, const void* residual, float* out_f32, int T,
                      int H, int W, int Cin, int Cout, int KT, int KH, int KW, int ring, int ring_start, long out_frame_stride,
                      long res_frame_stride, long plane_stride, int upsample2x, int epilogue, void* stream);
/* fvk_vae_conv_bf16 (3x3 spatial taps, bias [+ residual]) with the CONSUMER's WanRMS_norm (+ SiLU) fused into the epilogue, Cout == 96
 * (the full-resolution stage, where the separate norm pass moves 613 MB per call) or 192 (the half-resolution stage: the two waves that
 * hold a pixel's 192 channels swap their partial sums of squares through LDS): the normalised tensor is written straight into the
 * consumer conv's input ring norm_out [norm_ring, H*W, Cout] at frame slots (norm_slot0 + t) % norm_ring.  out == NULL drops the un-normed
 * store (conv1 -> norm2 -> conv2 inside a residual block, wanvae.py:418-431); otherwise both are written (the raw tensor feeds the next
 * block's shortcut).  Arithmetic = fvk_vae_rmsnorm_silu_bf16 applied to the bf16-rounded conv output. */
int fvk_vae_conv_norm_bf16(const void* in, const void* w, const void* bias, void* out, const void* residual, int T, int H, int W, int Cin,
                           int Cout, int KT, int ring, int ring_start, long out_frame_stride, long res_frame_stride, int upsample2x,
                           const float* norm_gamma, void* norm_out, int norm_ring, int norm_slot0, int norm_silu, void* stream);
/* WanRMS_norm (+ SiLU): out = [silu]( x / max(||x||_2, 1e-12) * sqrt(C) * gamma ) per pixel (ref: wanvae.py:231-232, :418-419).
 * x [n_pix, C] bf16, gamma fp32 [C]; pixel p = (t = p / HW, hw) is written to frame slot (slot0 + t) % ring of `out`
 * ([ring, HW, C]) — i.e. straight into the consumer conv's input ring.  C % 8 == 0, C <= 512. */
int fvk_vae_rmsnorm_silu_bf16(const void* x, const float* gamma, void* out, long n_pix, int C, int HW, int ring, int slot0,
                              int silu, void* stream);

/* ------------------------------------------------------------------ denoising-step tail (HBM-bound, one pass)
 * CFG combine + FlowUniPC multistep update.  ref: fastvideo/pipelines/stages/denoising.py:575-596,
 * fastvideo/models/schedulers/scheduling_flow_unipc_multistep.py:296-347 (convert), :364-489 (UniP), :491-617 (UniC), :649-724 (step).
 *   np   = uncond ? bf16(uncond + bf16(g * bf16(text - uncond))) : text                     (bf16 tensors: every op rounds)
 *   x0   = sample - float(bf16(sigma_t * np))                                               -> x0_out   (history for later steps)
 *   xc   = corr_order ? cc_x*last_sample - cc_m0*m0 - cc_B*([c_rho0*(m1-m0)/c_rk +] c_rho_last*(x0-m0)) : sample  -> sample_c_out
 *   next = pc_x*xc - pc_m0*x0 [- pc_B*(p_rho0*(m0-x0)/p_rk)]  (pred_order 2)               -> next_out (fp32), next_bf16_out (optional)
 * fp32 arithmetic in exactly this order without contraction (bit-identical to the eager reference).  n elements; m0 / m1 = x0_out of
 * the previous / second-previous step.  coef_host: HOST float[13] = g, sigma_t, cc_x, cc_m0, cc_B, c_rho0, c_rho_last, c_rk, pc_x,
 * pc_m0, pc_B, p_rho0, p_rk.  pred_order bit 8 (| 0x100): the two divisions by rk are evaluated as x * (1.0f / rk) — what torch's eager
 * GPU kernels do with a 0-d CPU divisor ("may lose one bit") — instead of a true division (torch's eager CPU kernels). */
int fvk_cfg_unipc_step(const void* noise_text, const void* noise_uncond, const float* sample, const float* last_sample, const float* m0,
                       const float* m1, float* x0_out, float* sample_c_out, float* next_out, void* next_bf16_out, long n,
                       const float* coef_host, int corr_order, int pred_order, void* stream);

/* DMD few-step sampling step (HBM-bound, one pass): x0 prediction and re-noising to the next timestep.
 * ref: fastvideo/models/utils.py:138-175 (pred_noise_to_pred_video), fastvideo/models/schedulers/scheduling_flow_match_euler_discrete.py:601-635
 *      (add_noise); call sites fastvideo/pipelines/stages/denoising.py:1382-1395 (DmdDenoisingStage), causal_denoising.py:273-310.
 *   video = bf16( float( double(noisy) - sigma_t[f] * double(pred_noise) ) )          fp64, product and difference rounded separately
 *   next  = bf16( (1 - sigma_next[f]) * float(video) + sigma_next[f] * float(noise) )  fp32, no contraction   (optional: last step has none)
 * pred_noise, noise, outputs: bf16 [frames, per_frame]; noisy_latent bf16 or fp32 (noisy_is_f32); sigma_t (fp64) / sigma_next (fp32): DEVICE
 * arrays [frames] holding the scheduler's fp32 table values.  Bit-identical to the eager reference ops. */
int fvk_dmd_step(const void* pred_noise, const void* noisy_latent, int noisy_is_f32, const double* sigma_t, const void* noise,
                 const float* sigma_next, void* pred_video_out, void* next_out, long frames, long per_frame, void* stream);

/* ------------------------------------------------------------------ VAE tile cross-fade + pixel post-processing (HBM-bound)
 * ref: fastvideo/models/vaes/common.py:94-114 (blend_v / blend_h / blend_t: `extent` python-loop slice assignments per tile edge),
 *      fastvideo/pipelines/stages/decoding.py:210, fastvideo/entrypoints/video_generator.py:912-913.
 * fvk_vae_blend_f32: a, b fp32 of logical shape [outer0, outer1, len, inner] with explicit strides (inner contiguous; e.g. a planar
 *   [C,T,H,W] tile or its [:, 1:] view blended along T, H or W), in place on b:
 *   b[.., i, :] = a[.., len_a - e + i, :] * float(1 - i/e) + b[.., i, :] * float(i/e),  i < e = min(extent, len_a, len_b);
 *   products and sum rounded separately (no FMA) = the eager reference bit for bit.
 * fvk_vae_postprocess_u8: planar fp32 [3, T, H, W] in [-1, 1] (plane_stride elements between channels) -> u8 frames [T, H, W, 3]:
 *   uint8(trunc(clamp(clamp(x / 2 + 0.5, 0, 1) * 255, 0, 255))). */
int fvk_vae_blend_f32(const float* a, float* b, long outer0, long outer1, long inner, int len_a, int len_b, int extent, long a_stride0,
                      long a_stride1, long a_axis_stride, long b_stride0, long b_stride1, long b_axis_stride, void* stream);
int fvk_vae_postprocess_u8(const float* pixels, void* frames_u8, int T, int H, int W, long plane_stride, void* stream);

/* ------------------------------------------------------------------ patch / time embedding glue
 * ref: fastvideo/layers/visual_embedding.py:46-55 (PatchEmbed k=s=(1,2,2)), :136-157 (timestep_embedding),
 *      wanvideo.py:689-690, :761-764. */
int fvk_patchify_bf16(const void* latent, void* out, int B, int C, int T, int Hh, int W, int pt, int ph, int pw,
                      void* stream);
int fvk_unpatchify_bf16(const void* x, void* latent, int B, int C, int T, int Hh, int W, int pt, int ph, int pw,
                        void* stream);
/* out bf16 [B, dim] = bf16([cos(t*f) | sin(t*f)]), f_i = exp(-ln(max_period) * i / (dim/2)) in fp32; t fp32 [B]. */
int fvk_timestep_embedding_bf16(const float* t, void* out, int B, int dim, float max_period, void* stream);
/* y = bf16(silu(float(x))) elementwise, n % 8 == 0. */
int fvk_silu_bf16(const void* x, void* y, long n, void* stream);

/* ------------------------------------------------------------------ measurement: the matrix pipe's sustained bf16 rate on THIS device
 * (no reference counterpart; SURVEY §8d asks for "the measured peaks on the box" next to every result).  `workgroups` x 4 waves (one per
 * SIMD) each issue iters x 64 register-only v_mfma_f32_16x16x32_bf16 on normal-like operands (data = 1) or zeros (data = 0): 16 384 FLOP per
 * instruction, no LDS, no memory traffic in the loop.  out: >= workgroups * 256 floats of scratch.  bench.py times it after the timed
 * region and prints roofline.sustained_matrix_rate_at_cap_tf with the socket power and shader clock it ran at. */
int fvk_mfma_sustained_probe_bf16(float* out, int workgroups, int iters, int data, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FVK_AMD_H */
