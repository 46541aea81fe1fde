/* ladcast_hip.h -- C ABI of libladcast_hip.so (MI355X / gfx950 HIP kernels for the
 * LaDCast autoregressive latent-diffusion rollout).
 *
 * The reference (tonyzyl/ladcast) is pure Python and has NO FFI: its plug-in surface
 * is Python duck typing (pipeline API, diffusers-style model/scheduler surface,
 * attention-processor protocol).  This header is therefore
 * build-defined.  Each entry point names the reference code it replaces
 * (file:line under the reference tree) so a maintainer can see which torch ops
 * the call stands in for.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (the caller --
 *     normally torch -- owns all memory; the library never allocates);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is
 *     enqueued on it, nothing synchronises the host; safe under hipGraph capture;
 *   - return value: 0 on success, LDC_ERR_* (<0) on bad arguments, or
 *     -(1000 + hipError_t) if the launch failed.  Nothing throws across the ABI;
 *   - all matrices are row-major fp32 unless stated; "ld*" are row strides in
 *     ELEMENTS; float4 paths need pointers 16-byte aligned and K, ld* % 4 == 0;
 *   - no global mutable state: reentrant per stream.
 */
#ifndef LADCAST_HIP_H
#define LADCAST_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LDC_ABI_VERSION 5

#define LDC_OK 0
#define LDC_ERR_ARG (-1)       /* null pointer / non-positive size */
#define LDC_ERR_ALIGN (-2)     /* pointer or stride breaks the 16-byte rule */
#define LDC_ERR_UNSUPPORTED (-3)

enum ldc_act { LDC_ACT_NONE = 0, LDC_ACT_SILU = 1, LDC_ACT_GELU_TANH = 2, LDC_ACT_RELU = 3 };
/* `act_in` of ldc_linear_small / _grouped / _mod only: x holds ONE timestep per row (x_rows floats, any alignment) and the input
 * row is its 256-wide sinusoidal embedding [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(1e4) k / 128) (K must be 256): diffusers
 * Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0) fused into the TimestepEmbedding's first Linear
 * (models/LaDCast_3D_model.py:362-364,673); same values as ldc_timestep_embedding. */
#define LDC_ACT_IN_TIMESTEP_SINCOS 16

int ldc_abi_version(void);
/* name of the code-object architecture the library was built for ("gfx950") */
const char* ldc_build_arch(void);

/* ---------------------------------------------------------------------------
 * Dense contraction  C[b] = epilogue(A[b] . W^T)        (fp32 MFMA, exact fp32)
 *   A[b]: M x K (lda), W: N x K (ldw, shared by all batches), C[b]: M x N (ldc)
 *   v = acc + bias[n]; v = act(v); v *= gate[b*gate_bs + n]; v += R[b][m*ldr + n]
 *   (bias / gate / R may be NULL).  R may alias C.
 * Replaces nn.Linear + activation + gated residual on the token streams:
 *   models/LaDCast_3D_model.py:92-94,175-177,214,219 (q/k/v, added q/k/v, to_out,
 *   to_add_out), :271-276,503-512,558-563 (FeedForward + gate), :422-424,442,
 *   460-462 (proj_mlp / proj_out of the single block), :1045 (proj_out),
 *   models/embeddings.py:52-58 (k=1 Conv3d patch embed), and the NHWC 1x1 convs /
 *   Linear layers of models/DCAE.py:129-131,142-144,286,296.
 * ------------------------------------------------------------------------- */
typedef struct ldc_gemm_desc {
  int M, N, K, batch;
  int lda, ldw, ldc, ldr;
  long long a_bs, c_bs, r_bs; /* batch strides (elements) */
  int gate_bs;                /* batch stride of gate (elements) */
  int act;                    /* enum ldc_act */
  int flags;                  /* LDC_GEMM_* bits below; 0 = plain fp32 A and C */
  int reserved;               /* 0 */
} ldc_gemm_desc;
/* Split-bf16 activation format (only ldc_gemm_grouped_bf16x3 takes these; every other entry rejects them):
 * columns 8c..8c+7 of a row occupy the SAME 32 bytes as 8 floats would, holding [hi x8 | lo x8] bf16
 * (hi = bf16(x), lo = bf16(x - hi)), so every pointer / stride / column offset that is a multiple of 8
 * elements addresses the same data in both formats.
 *   LDC_GEMM_A_SPLIT  A is in that format (made by a producer kernel that split it once) instead of fp32:
 *                     the GEMM then skips its own split of every A fragment; results are bit-identical.
 *   LDC_GEMM_C_SPLIT  C is written in that format (after bias / activation / gate / residual). */
#define LDC_GEMM_A_SPLIT 1
#define LDC_GEMM_C_SPLIT 2
/*   LDC_GEMM_BF16_1TERM  single-term bf16 contraction bf16(A).bf16(W) (fp32 accumulate): the "bf16" mixed-precision mode
 *                     (BASELINE configs[4] "fp16/bf16 mixed"; what torch.autocast(bfloat16) does to an nn.Linear's
 *                     operands, the reference's fp32 islands - models/LaDCast_3D_model.py:953, models/DCAE.py:162,180 -
 *                     never reach a GEMM here).  In this mode the operands are PLAIN bf16 ROWS (LDC_FMT_BF16): a row of
 *                     K values occupies the first 2 K bytes of the 4 K bytes its fp32 row would (same pointers, same
 *                     strides, half the bytes moved); LDC_GEMM_A_SPLIT (required) then means "A is in that format" and
 *                     LDC_GEMM_C_SPLIT "write C in that format"; W comes from ldc_pack_weight_bf16 ([N][K] bf16).
 *                     K % 64 == 0; all problems of one call must agree.  Measured 3.2e-3 rel-L2 per 375M forward instead of 6e-6
 *                     (stated tolerance 7e-3 = measured x 2; the table of all stated tolerances is ladcast_amd/precision.py). */
#define LDC_GEMM_BF16_1TERM 4
/*   LDC_GEMM_F32_REGSTAGE  (ldc_gemm_grouped, exact-fp32 problems only) keep the launch on the register-staged stream-K kernel
 *                     (v_mfma_f32_32x32x2_f32) even where the LDS-DMA ring kernel would serve it (K % 32 == 0, contiguous weights,
 *                     16-byte rows): the second exact-fp32 implementation as a caller's choice - cross-checks, and shapes a caller
 *                     knows to be launch-bound.  Same results to fp32 rounding (another summation order). */
#define LDC_GEMM_F32_REGSTAGE 8
/* activation formats of the producers' `out_split` / `x_fmt` arguments: fp32, split-bf16 groups (LDC_GEMM_A_SPLIT), plain bf16 rows */
#define LDC_FMT_F32 0
#define LDC_FMT_SPLIT 1
#define LDC_FMT_BF16 2
int ldc_sizeof_gemm_desc(void);
int ldc_gemm_bias_act(const float* A, const float* W, const float* bias, const float* gate,
                      const float* R, float* C, const ldc_gemm_desc* d, void* stream);

/* Same contraction for up to LDC_GEMM_MAX_PROBLEMS independent problems in ONE persistent launch
 * with stream-K scheduling (work = (tile, 32-deep k-step) units cut into equal contiguous ranges);
 * tiles whose K range was split are summed INSIDE the launch by the piece that arrives last, in
 * workgroup order (bitwise reproducible; no second launch since ABI 2).  Problems with K % 32 == 0,
 * ldw == K, lda % 4 == 0 and 16-byte aligned A / W run on the LDS-DMA ring kernel (fp32 operand rows,
 * v_mfma_f32_16x16x4_f32, one workgroup per CU); the others - same results up to the summation
 * order - on the register-staged kernel (2 workgroups per CU).  Exact fp32 products either way.  Built for the
 * small grids of the AR transformer (e.g. the pred- and cond-stream projections of a dual block,
 * models/LaDCast_3D_model.py:92-94,175-177,558-563).  `workspace` is caller-owned device scratch of
 * at least ldc_gemm_grouped_workspace_bytes() bytes (16-byte aligned), initialised ONCE with
 * ldc_gemm_grouped_workspace_init (zeroes the tile-arrival counters in its first MiB; every call
 * leaves them zero) and then reused by the calls of ONE stream.  C of one problem must not overlap
 * A/R of another problem of the same call. */
#define LDC_GEMM_MAX_PROBLEMS 4
typedef struct ldc_gemm_problem {
  const float* A;
  const float* W;
  const float* bias;
  const float* gate;
  const float* R;
  float* C;
  ldc_gemm_desc d;
} ldc_gemm_problem;
int ldc_sizeof_gemm_problem(void);
long long ldc_gemm_grouped_workspace_bytes(void);
int ldc_gemm_grouped_workspace_init(void* workspace, long long workspace_bytes, void* stream);
int ldc_gemm_grouped(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                     void* stream);

/* Split-bf16 ("bf16x3") form of ldc_gemm_grouped: same problems / epilogue / scheduling, but every
 * problem's W points to weights pre-split by ldc_pack_weight_bf16x2 ([N][K/8][hi x8 | lo x8] bf16,
 * N*K*4 bytes; d.ldw is ignored, K % 8 == 0) and the contraction is Ah.Wh + Ah.Wl + Al.Wh on the bf16
 * matrix cores with fp32 accumulation (hi = bf16(x), lo = bf16(x - hi); activations are split on
 * load).  Error vs exact fp32: ~4e-6 rel-L2 per model forward. */
int ldc_gemm_grouped_bf16x3(const ldc_gemm_problem* problems, int n, void* workspace, long long workspace_bytes,
                            void* stream);
int ldc_pack_weight_bf16x2(const float* W, void* out, int N, int K, int ldw, void* stream);
/* fp32 [N][K] (row stride ldw) -> plain bf16 [N][K] (2 N K bytes), the W operand of the LDC_GEMM_BF16_1TERM mode */
int ldc_pack_weight_bf16(const float* W, void* out, int N, int K, int ldw, void* stream);

/* Small-M linear (M = rows <= 64): y[r] = act_out(W . act_in(x[r % x_rows]) + bias)
 *                                           + add[r % add_rows]
 * HBM-bound weight streaming; replaces the timestep / text / AdaLN projection
 * Linears on (B, D) vectors: diffusers TimestepEmbedding, PixArtAlphaTextProjection,
 * AdaLayerNormZero/-Single/-Continuous .linear and HunyuanVideoAdaNorm.linear
 * (models/LaDCast_3D_model.py:224-238,362-364,441,524-529,673-678,1044).
 * Two kernels, chosen from (rows, N, K) only: up to 8 rows, or N < 4096, or K % 64 != 0 -> the VALU kernel (W streamed once per 8 rows);
 * more rows on a wide output (the AdaLN modulation of a sampler chunk's conditioning batch: 20 rows per member x 38 D) -> exact-fp32
 * products on v_mfma_f32_16x16x4_f32 (W streamed once per 32 rows).  Same formula, different summation order (fp32 rounding); inside either
 * kernel a row's result does not depend on the other rows of the launch. */
int ldc_linear_small(const float* x, int x_rows, const float* W, const float* bias,
                     const float* add, int add_rows, float* y, int rows, int N, int K,
                     int act_in, int act_out, void* stream);

/* ldc_linear_small followed by y = y * (1 + mod[r % mod_rows][n]) + mod[r % mod_rows][N + n] (mod: [mod_rows][2 N]): the
 * time-elapsed modulation of the conditioning embedding, temb * (1 + scale) + shift (models/LaDCast_3D_model.py:958-969), as the
 * epilogue of the text embedder's second Linear; always the VALU kernel: bit-identical to ldc_linear_small + ldc_temb_modulate where that
 * takes the VALU kernel too (see above). */
int ldc_linear_small_mod(const float* x, int x_rows, const float* W, const float* bias, const float* add, int add_rows,
                         const float* mod, int mod_rows, float* y, int rows, int N, int K, int act_in, int act_out,
                         void* stream);

/* Up to LDC_LINEAR_SMALL_MAX_GROUPED independent small linears (same formula, the per-output arithmetic of
 * ldc_linear_small's VALU kernel: results are bit-identical to single launches that take that kernel) in ONE launch: the two MLPs of
 * CombinedTimestepTextProjEmbeddings (timestep_embedder / text_embedder, used twice per forward:
 * models/LaDCast_3D_model.py:362-364,953-969) have independent first and second layers.  No problem's y may be
 * another problem's x / add / y (LDC_ERR_ARG). */
#define LDC_LINEAR_SMALL_MAX_GROUPED 4
typedef struct ldc_linear_small_problem {
  const float* x;    /* [x_rows][K] */
  const float* W;    /* [N][K] */
  const float* bias; /* [N] or NULL */
  const float* add;  /* [add_rows][N] or NULL */
  float* y;          /* [rows][N] */
  int x_rows, add_rows, rows, N, K, act_in, act_out, reserved;
} ldc_linear_small_problem;
int ldc_linear_small_grouped(const ldc_linear_small_problem* problems, int n, void* stream);

/* ---------------------------------------------------------------------------
 * Attention   O = softmax(Q K^T / sqrt(128) + key_bias) V   per (batch, head)
 *   Q,K,V: token-major [B][S][H][128] views with row stride ld_qkv and batch stride
 *   qkv_bs (so they can be column slices of one fused QKV buffer);
 *   O: [B][S][H*128] with row stride ldo, batch stride o_bs;
 *   key_bias: NULL, or [S] floats added to every query's score of that key - the float
 *   attention mask of `scale_attn_by_lat` (models/LaDCast_3D_model.py:683-693,873-882:
 *   a (1, 1, 1, keys) tensor, i.e. a function of the key alone).
 * Replaces F.scaled_dot_product_attention at models/LaDCast_3D_model.py:199-203
 * (incl. the transpose/flatten back to [B, N, C]).  head_dim must be 128.
 * ------------------------------------------------------------------------- */
int ldc_attn_fwd(const float* Q, const float* K, const float* V, float* O, int B, int S, int H,
                 int ld_qkv, long long qkv_bs, int ldo, long long o_bs, const float* key_bias, void* stream);
/* (ABI 4) the same with a workspace of ldc_attn_fwd_workspace_bytes() bytes (16-byte aligned, ZERO-FILLED ONCE by the caller, one per
 * stream: the launch leaves its counters zero).  With it, a launch whose query blocks x heads x batch do not fill whole rounds of the 256
 * CUs is cut into one contiguous range of (unit, key tile) items per CU - 216 units (one member of the 375M model) take 0.84 of a round,
 * 288 / 432 units 1.13 / 1.69 rounds instead of two; pieces of a unit meet through write-through slabs and are combined by the last
 * arriver in range order (bitwise reproducible).  workspace == NULL (or too small): the one-unit-per-workgroup grids of ldc_attn_fwd. */
long long ldc_attn_fwd_workspace_bytes(void);
int ldc_attn_fwd_ws(const float* Q, const float* K, const float* V, float* O, int B, int S, int H, int ld_qkv, long long qkv_bs,
                    int ldo, long long o_bs, const float* key_bias, void* workspace, long long workspace_bytes, void* stream);

/* (ABI 2: the second-generation split attention - ldc_attn_pack_bf16x3 / ldc_attn_fwd_packed_bf16x3 / ldc_attn_packed_bytes, a
 * pack pass writing LDS tile images + the attention on them - was removed; ldc_attn_qkv_prepare_split / the fused QKV epilogue +
 * ldc_attn_fwd_split below serve every caller it had.)
 *
 * `flags` of ldc_attn_fwd_split is a bit set: LDC_ATTN_OUT_SPLIT = O is written in the split activation format of LDC_GEMM_A_SPLIT (ldo, o_bs
 * multiples of 8); LDC_ATTN_BF16_1TERM = single-term bf16 products (Qh.Kh, Ph.Vh; fp32 scores, softmax and accumulation):
 * the attention of the "bf16" mixed-precision mode (see LDC_GEMM_BF16_1TERM). */
#define LDC_ATTN_OUT_SPLIT 1
#define LDC_ATTN_BF16_1TERM 2
#define LDC_ATTN_OUT_BF16 4 /* ldc_attn_fwd_split only: O as plain bf16 rows (LDC_FMT_BF16) */

/* Split-bf16 attention on ROW-MAJOR operand rows (third generation; no packed copy of the operands):
 *   Q, K, V are views into a fused [B][S][3][H][128]-float buffer (row stride ld_qkv, batch stride qkv_bs, in floats) whose
 *   512 bytes per (token, head) hold, instead of 128 floats,
 *     q, k: 16 groups of 8 head-dim values, each [hi x8 | lo x8] bf16 (the LDC_GEMM_A_SPLIT format), with the per-head
 *           RMSNorm + rotary embedding already applied and q multiplied by log2(e) / sqrt(128);
 *     v   : [hi x128 | lo x128] bf16.
 *   Producers: the QKV projection itself (ldc_gemm_grouped_bf16x3_qkv below: GEMM epilogue) or, from an fp32 buffer in place,
 *   ldc_attn_qkv_prepare_split (token rows [0, split_row) use wq0 / wk0 / cos0 / sin0, rows [split_row, S) use wq1 / wk1 / cos1 / sin1 with table row = row - split_row; NULL weights = no norm, NULL tables = no RoPE).
 *   ldc_attn_fwd_split: O = softmax(q.k^T) v per (batch, head); `flags`: the bit set above.
 * Replaces attn.norm_q/norm_k/norm_added_q/norm_added_k + apply_rotary_emb + F.scaled_dot_product_attention,
 * models/LaDCast_3D_model.py:103-169,183-203. */
/* The QKV projection with those operand rows as its OUTPUT: ldc_gemm_grouped_bf16x3 (pre-split activations, K % 32 == 0) whose
 * epilogue, for every problem i with epi[i].heads > 0 (N = 3 * heads * 128, C = the fused buffer, no act / gate / residual), adds
 * the bias, applies RMSNorm(128, eps) * wq | wk and the rotary embedding (compact table, row = rope_row0 + m; NULL = none) to the q and k
 * heads, multiplies q by qscale (0 = log2(e) / sqrt(128)) and writes the split rows - to_q/to_k/to_v (+ add_*_proj), norm_q/k,
 * apply_rotary_emb of models/LaDCast_3D_model.py:92-169,175-190 in one launch.  Problems with heads == 0 get the ordinary epilogue.
 * LDC_ERR_UNSUPPORTED: shape not served by that kernel - run the plain GEMM, then ldc_attn_qkv_prepare_split. */
typedef struct ldc_qkv_epilogue {
  const float* wq;   /* [128] */
  const float* wk;   /* [128] */
  const float* rope; /* [rows][64][2] = (cos_i, sin_i) of rotary pair i: the compact form of the reference's [rows][128] cos / sin
                        tables, which hold every value twice (repeat_interleave(2)); NULL = no rotary embedding */
  const float* reserved; /* NULL */
  float eps, qscale;
  int heads, rope_row0;
} ldc_qkv_epilogue;
int ldc_sizeof_qkv_epilogue(void);
int ldc_gemm_grouped_bf16x3_qkv(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace,
                                long long workspace_bytes, void* stream);
/* (ABI 4) exact-fp32 counterpart: the same grouped launch on fp32 operands (ldc_gemm_grouped's problems, flags 0) where problem i with
 * epi[i].heads > 0 is a QKV projection whose epilogue applies bias -> per-head RMSNorm * weight -> adjacent-pair rotary embedding to the q
 * and k heads (v: bias only) and writes PLAIN fp32 rows - what ldc_qk_rmsnorm_rope does in place afterwards, without its launches and its
 * round trip (pass qscale = 1: ldc_attn_fwd scales q itself).  LDC_ERR_UNSUPPORTED (K % 32 != 0, strided weights, rows not 16-byte
 * aligned): run ldc_gemm_grouped + ldc_qk_rmsnorm_rope. */
int ldc_gemm_grouped_qkv_f32(const ldc_gemm_problem* problems, const ldc_qkv_epilogue* epi, int n, void* workspace,
                             long long workspace_bytes, void* stream);
int ldc_attn_qkv_prepare_split(float* Q, float* K, float* V, int B, int S, int H, int ld_qkv, long long qkv_bs, int split_row,
                               const float* wq0, const float* wk0, const float* cos0, const float* sin0, const float* wq1,
                               const float* wk1, const float* cos1, const float* sin1, float eps, void* stream);
/* workspace (ABI 3): bytes ldc_attn_fwd_split can use for this call shape (0: none needed).  When query blocks x heads x batch exceed the
 * 256 CUs by at most half a round (the 1.6B model: 16 heads x 18 query blocks of 128 = 288 units), the kernel runs one persistent
 * workgroup per CU - its whole units first, then ONE key slice of a left-over unit - and a second small launch merges the slices'
 * (O, m, l) in slice order (bitwise reproducible): 288 units take ~1.13 rounds instead of two uneven ones.  workspace == NULL (or too
 * small) is legal: the call then runs the plain one-unit-per-workgroup grids. */
long long ldc_attn_fwd_split_workspace_bytes(int B, int S, int H);
/* upper bound of the above over every call shape (17.8 MB): allocate this once per stream when launches are captured into hipGraphs -
 * a workspace that is replaced by a larger one later leaves the captured launches writing through a dangling pointer */
long long ldc_attn_fwd_split_workspace_max_bytes(void);
int ldc_attn_fwd_split(const float* Q, const float* K, const float* V, float* O, int B, int S, int H, int ld_qkv,
                       long long qkv_bs, int ldo, long long o_bs, const float* key_bias, int flags, void* workspace,
                       long long workspace_bytes, void* stream);
/* key_bias: as ldc_attn_fwd's, but 32 * ceil(S / 32) floats (16-byte aligned; entries past S are ignored). */

/* In-place per-head RMSNorm(128, eps, weight) on q and k followed by the
 * adjacent-pair rotary embedding (cos/sin tables [rows][128], NULL = no RoPE),
 * for token rows [row0, row0+rows) of every batch of a fused QKV buffer.
 * Replaces attn.norm_q/norm_k/norm_added_q/norm_added_k + apply_rotary_emb,
 * models/LaDCast_3D_model.py:103-169,183-186. */
int ldc_qk_rmsnorm_rope(float* q, float* k, int B, int row0, int rows, int H, int ld, long long bs,
                        const float* wq, const float* wk, float eps, const float* cos_tab,
                        const float* sin_tab, void* stream);

/* Row LayerNorm (no affine inside) followed by y = n * mul + add with
 *   mode 0: mul = 1 + scale[b][c], add = shift[b][c]   (AdaLN modulate)
 *   mode 1: mul = scale[c],        add = shift[c]      (plain affine LayerNorm)
 * x: [B][rows][D] (ldx, x_bs) -> y (ldy, y_bs); mod_bs = batch stride of scale/shift.
 * Replaces AdaLayerNormZero/-Single/-Continuous .norm + modulation, nn.LayerNorm,
 * models/LaDCast_3D_model.py:257,270,287,299,441,502,507,524-529,546-555,1044. */
int ldc_layernorm_mod(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs,
                      int ldy, long long y_bs, const float* scale, const float* shift, int mod_bs,
                      int mode, float eps, int out_split, void* stream);
/* out_split != 0: y is written in the split activation format of LDC_GEMM_A_SPLIT (the consumer GEMM then
 * does not split it again); D, ldy, y_bs multiples of 8. */
/* The same over two row segments in one launch: rows [0, split_row) use scale / shift, rows [split_row, rows)
 * use scale2 / shift2 (same mode, eps and mod_bs).  One call per dual block norm instead of one per stream:
 * norm1 + norm1_context (models/LaDCast_3D_model.py:524-529) and norm2 + norm2_context (:546-555) when the
 * two token streams are adjacent rows of one buffer.  Row results are those of ldc_layernorm_mod, bit for bit. */
int ldc_layernorm_mod2(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs,
                       int ldy, long long y_bs, const float* scale, const float* shift, int split_row,
                       const float* scale2, const float* shift2, int mod_bs, int mode, float eps,
                       int out_split, void* stream);

/* y[b][c] = mean over rows of x[b][r][c]   (hidden_states.mean(dim=1),
 * models/LaDCast_3D_model.py:382,955) */
int ldc_mean_rows(const float* x, float* y, int B, int rows, int D, int ldx, long long x_bs, void* stream);
/* The same mean (bit-identical y) that also writes x in the split activation format of LDC_GEMM_A_SPLIT to
 * x_split (row stride lds, batch stride s_bs, in floats; D % 8 == 0), so that the Linear consuming x
 * (context_refiner.proj_in, models/LaDCast_3D_model.py:362-380) takes the pre-split GEMM kernel. */
int ldc_mean_rows_split(const float* x, float* y, float* x_split, int B, int rows, int D, int ldx,
                        long long x_bs, int lds, long long s_bs, int fmt /* LDC_FMT_SPLIT | LDC_FMT_BF16 */, void* stream);

/* out[b][r][c] = resid[b][r][c] + gate[b][c] * y[b][r][c]  (refiner gated residual on the
 * projection-less attention output, models/LaDCast_3D_model.py:296-297) */
int ldc_gate_residual(const float* resid, const float* y, const float* gate, float* out, int B,
                      int rows, int D, int ld_res, long long res_bs, int ld_y, long long y_bs,
                      int gate_bs, void* stream);
/* ldc_gate_residual in place (resid += gate * y) followed by out = LayerNorm(resid; eps) * weight + bias (out_split: written in
 * the LDC_GEMM_A_SPLIT format) in ONE launch: the refiner block's gated attention residual + norm2
 * (models/LaDCast_3D_model.py:296-300); bit-identical to ldc_gate_residual + ldc_layernorm_mod(mode 1). */
int ldc_gate_residual_layernorm(float* resid, const float* y, const float* gate, float* out, int B, int rows, int D, int ld_res,
                                long long res_bs, int ld_y, long long y_bs, int gate_bs, int ld_out, long long out_bs,
                                const float* weight, const float* bias, float eps, int out_split, void* stream);

/* Layout changes between the reference's channel-major latents (B, C, T*H*W) and
 * token-major (B, T*H*W, ld) (models/embeddings.py:56-59 flatten/transpose and
 * models/LaDCast_3D_model.py:1047-1062 un-patchify; also NCHW <-> NHWC for the DCAE).
 * Columns [C, fill_cols) of the token-major output are zero-filled (fill_cols <= ldo; lets two
 * calls concatenate channels into one padded row). */
int ldc_chan_to_token(const float* in, float* out, int B, int C, int N, int ldo, int fill_cols, void* stream);
/* out_split != 0: the token rows are written in the split activation format of LDC_GEMM_A_SPLIT
 * (fill_cols, ldo multiples of 8; out 32-byte aligned). */
int ldc_chan_to_token_split(const float* in, float* out, int B, int C, int N, int ldo, int fill_cols,
                            int out_split, void* stream);
int ldc_token_to_chan(const float* in, float* out, int B, int C, int N, int ldi, void* stream);

/* Sinusoidal timestep embedding [cos | sin], 256 wide (diffusers Timesteps(256,
 * flip_sin_to_cos=True, shift=0); models/LaDCast_3D_model.py:362-364,673). */
int ldc_timestep_embedding(const float* t, float* out, int n, void* stream);

/* temb[b][c] = temb[b][c] * (1 + te[b % te_rows][c]) + te[b % te_rows][D + c]
 * (time-elapsed scale/shift, models/LaDCast_3D_model.py:966-969) */
int ldc_temb_modulate(float* temb, const float* te, int B, int D, int te_rows, void* stream);

/* Per-channel affine on (outer, C, inner): forward (x-mean[c])/std[c]*t, inverse
 * x/t*std[c]+mean[c]  (dataloader/utils.py:223-240). */
int ldc_chan_affine(const float* x, float* y, const float* mean, const float* std_, float target_std,
                    long long outer, int C, long long inner, int inverse, void* stream);

/* ---------------------------------------------------------------------------
 * Sampler state updates (pipelines/edm_sampler.py:60-113; fp64 state, fp32 model I/O)
 * and the DPM-Solver++(2M) step of diffusers.EDMDPMSolverMultistepScheduler.step
 * (used by pipelines/pipeline_AR.py:100-102).  Scalars are computed on the host
 * in fp32 exactly as the scheduler does and passed here widened to double.
 * ------------------------------------------------------------------------- */
int ldc_edm_scale_f64_to_f32(const double* x, double c_in, float* out, long long n, void* stream);
int ldc_edm_init_state(const float* noise, double sigma0, double* x, long long n, void* stream);
/* stochastic churn of the non-deterministic sampler (pipelines/edm_sampler.py:67-76): x_hat = x_cur + coef * noise (fp64),
 * coef = sqrt(t_hat^2 - t_cur^2) * S_noise computed in fp32 on the host as the reference does */
int ldc_edm_churn(const double* x_cur, const double* noise, double coef, double* x_hat, long long n, void* stream);
/* den = c_skip*x + c_out*F; d = (x-den)/t_hat; x_next = x + dt*d */
int ldc_edm_euler(const double* x_hat, const float* F, double c_skip, double c_out, double t_hat,
                  double dt, double* x_next, double* d_cur, long long n, void* stream);
/* den = c_skip*x_next + c_out*F; d' = (x_next-den)/t_next; x_next = x_hat + dt*(0.5 d_cur + 0.5 d') */
int ldc_edm_heun(const double* x_hat, double* x_next, const float* F, const double* d_cur,
                 double c_skip, double c_out, double t_next, double dt, long long n, void* stream);
int ldc_f64_to_f32(const double* x, float* y, long long n, void* stream);
/* x0 = c_skip*sample + c_out*F; order 1: prev = a*sample - b*x0;
 * order 2: prev = a*sample - b*x0 - (0.5*b) * (inv_r0 * (x0 - m1))   (all fp32) */
int ldc_dpm_step(const float* sample, const float* F, const float* m1, float* x0, float* prev,
                 float c_skip, float c_out, float a, float b, float inv_r0, int order, long long n,
                 void* stream);
/* One solver step of the scheduler classes the reference's pipeline loop names (pipelines/pipeline_AR.py:19-21,100-102:
 * diffusers DDIMScheduler.step / DDPMScheduler.step, v0.32.1) as one fused fp32 kernel; coefficients are the host's fp32 0-dim
 * tensor arithmetic, widened nowhere.  prediction_type: 0 epsilon, 1 sample, 2 v_prediction; clip_range <= 0: no clamp of x0;
 * noise == NULL: no noise term (DDIM eta = 0; DDPM at t = 0).
 *   x0 = (s - sqrt_beta_t F) / sqrt_alpha_t | F | sqrt_alpha_t s - sqrt_beta_t F   (clamped)
 *   ldc_ddim_step: prev = sqrt_alpha_prev x0 + dir_coef eps [+ std_dev noise], eps = F | (s - sqrt_alpha_t x0)/sqrt_beta_t |
 *                  sqrt_alpha_t F + sqrt_beta_t s  (use_clipped_model_output: eps re-derived from the clamped x0)
 *   ldc_ddpm_step: prev = x0_coef x0 + sample_coef s [+ std_dev noise] */
int ldc_ddim_step(const float* sample, const float* F, const float* noise, float* x0, float* prev, float sqrt_alpha_t,
                  float sqrt_beta_t, float sqrt_alpha_prev, float dir_coef, float std_dev, float clip_range,
                  int prediction_type, int use_clipped_model_output, long long n, void* stream);
int ldc_ddpm_step(const float* sample, const float* F, const float* noise, float* x0, float* prev, float sqrt_alpha_t,
                  float sqrt_beta_t, float x0_coef, float sample_coef, float std_dev, float clip_range, int prediction_type,
                  long long n, void* stream);
int ldc_scale_f32(const float* x, float s, float* y, long long n, void* stream);
/* out = a*x + b*y (fp32): scheduler.precondition_outputs (c_skip*sample + c_out*F) and
 * add_noise (x0 + sigma*noise) on whole tensors */
int ldc_axpby_f32(const float* x, float a, const float* y, float b, float* out, long long n, void* stream);

/* ---------------------------------------------------------------------------
 * DCAE encoder / decoder (models/DCAE.py, models/sphere_conv.py) in NHWC layout
 * ([B][H][W][C], channel stride 1, pixel stride ld*).
 * ------------------------------------------------------------------------- */
/* Dense SphereConv2d (stride 1, ksize 3 or 5, even W) as an implicit GEMM on the fp32 MFMA:
 *   Y[pix][co] = act(bias[co] + sum_{ky,kx,ci} Wt[co][(ky*ks+kx)*cin + ci] * X[src(pix,ky,kx)][ci]) (+ R[pix][co])
 * with the reference's pole padding / kernel-row flip (models/sphere_conv.py:62-129,174-192).
 * Wt is the conv weight repacked [cout][ks*ks][cin] (cin % 4 == 0, zero-padded if needed). */
int ldc_sphere_conv_nhwc(const float* X, const float* Wt, const float* bias, const float* R, float* Y, int B,
                         int H, int W, int cin, int ldx, int cout, int ldy, int ldr, int ksize, int act,
                         void* stream);
/* Depthwise SphereConv2d, weights [ks*ks][C]; glu != 0: y[c] = d[c] * silu(d[c + C/2]) for c < C/2
 * (GLUMBConv conv_depth + gate, models/DCAE.py:287-295,311-313; multiscale proj_in, :77-85). */
int ldc_sphere_dwconv_nhwc(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                           int C, int ldx, int ldy, int ksize, int glu, void* stream);
/* Grouped 1x1 conv with 32 channels per group, weights [groups*32][32] (models/DCAE.py:86-88). */
int ldc_grouped_conv1x1_nhwc(const float* x, const float* wt, float* y, long long M, int groups, int ldx, int ldy,
                             void* stream);
/* ReLU linear attention over consecutive 96-channel groups (q|k|v = 32|32|32) of qkv[B][P][ldq];
 * y[B][P][groups*32] (models/DCAE.py:158-175,239-253; fp32, eps 1e-15). */
int ldc_relu_linear_attn_nhwc(const float* qkv, float* y, int B, int P, int groups, int ldq, int ldy, float eps,
                              void* workspace, long long workspace_bytes, void* stream);
/* Both contractions on the fp32 matrix core (exact fp32 products).  P >= 1024 and `workspace` >= ldc_relu_linear_attn_workspace_bytes(B, P,
 * groups) bytes (16-byte aligned; 0 bytes for P < 1024): two launches over 128-pixel slices - partial KV matrices (33 x 32 per slice) to the workspace,
 * then KV = the partials added in slice order and the output of the same pixels; the order only depends on P, so a frame's result
 * does not depend on the batch it is in (bitwise reproducible).  With a NULL / smaller workspace: one launch, a 16-wave workgroup
 * per (batch, group), partials added in wave order - the same result up to fp32 rounding. */
long long ldc_relu_linear_attn_workspace_bytes(int B, int P, int groups);
/* y = act(RMSNorm_C(x) * w + b (+ resid)) per pixel row (models/DCAE.py:259-260,317-322,371-377,729-730). */
int ldc_rmsnorm_rows(const float* x, const float* w, const float* b, const float* resid, float* y, long long rows,
                     int C, int ldx, int ldr, int ldy, float eps, int act, void* stream);
/* DCDownBlock2d tail: pixel_unshuffle(cv) + channel-group-mean(pixel_unshuffle(x)) (models/DCAE.py:477-490).
 * cv: [B][2*H2][2*W2][cout/4], x: [B][2*H2][2*W2][cin], y: [B][H2][W2][cout]. */
int ldc_pixel_unshuffle_shortcut(const float* cv, const float* x, float* y, int B, int H2, int W2, int cout, int cin,
                                 void* stream);
/* DCUpBlock2d tail: pixel_shuffle(cv) + pixel_shuffle(repeat_interleave(x)) (models/DCAE.py:526-532).
 * cv: [B][H][W][4*cout], x: [B][H][W][cin], y: [B][2H][2W][cout]. */
int ldc_pixel_shuffle_shortcut(const float* cv, const float* x, float* y, int B, int H, int W, int cout, int cin,
                               void* stream);
/* Channel regroup of [M][cin] -> [M][cout]: group mean (cin > cout; encoder out shortcut, models/DCAE.py:624-627)
 * or repeat_interleave (cin < cout; decoder in shortcut, :720-722). */
int ldc_chan_regroup(const float* x, float* y, long long M, int cin, int cout, void* stream);

/* ldc_sphere_conv_nhwc (dense SphereConv2d, same contract) on PRE-SPLIT activations (the DCAE's `bf16x3` path since round 2; round 1's
 * fp32-rows-in entry point ldc_sphere_conv_nhwc_bf16x3 left the ABI in version 3 - it left the tree with round 1's kernel in round 6): X holds NHWC rows in the split format
 * (in_fmt = LDC_FMT_SPLIT: columns 8g..8g+7 in 32 bytes [hi x8 | lo x8]; ldx % 8 == 0, ldx >= cin rounded up to 8, pad columns
 * zero), written by the producers below, so the conv's main loop is the pre-split GEMM kernel (gemm_bf16x3_v3.hip: 16x16x32 MFMA, no
 * VALU in the loop) with the sphere gather as per-lane LDS-DMA source addresses.  Wp: ldc_pack_weight_bf16x2 of the tap-major weight
 * [cout][k*k][cin_p], cin_p = 32 * 2^j >= cin with zeros behind cin.  out_fmt: LDC_FMT_F32 (Y fp32 rows) or LDC_FMT_SPLIT (Y split rows for the next conv, ldy % 8 == 0,
 * ldy >= cout rounded up to 8; cout % 4 == 0, the pad half of a last half-filled group is written as zeros).  cin % 8 need not
 * hold (252: the last group's pad columns are zero in X and in Wp).  ksize 1 / 3 / 5.
 * in_fmt = LDC_FMT_BF16 is the single-term `bf16` mode of the same conv (see LDC_GEMM_BF16_1TERM): X plain bf16 rows (same ldx, in
 * floats), Wp = ldc_pack_weight_bf16 of the [cout][k*k][cin rounded up to 64 * 2^j] tap-major weight, out_fmt LDC_FMT_F32 | LDC_FMT_BF16.
 * in_fmt = LDC_FMT_F32 (ABI 3) is the EXACT-fp32 conv on the same kernel (v_mfma_f32_16x16x4_f32 on plain fp32 operand rows, the conv
 * gather unchanged): X plain fp32 rows (ldx % 4 == 0, cin % 4 == 0), Wp the plain fp32 [cout][k*k][cin rounded up to 32 * 2^j]
 * tap-major weight with zeros behind cin, out_fmt LDC_FMT_F32 - what AutoencoderDC runs in its fp32 mode (ldc_sphere_conv_nhwc, the
 * tile-per-workgroup kernel, stays for channel counts that are no multiple of 4 and as the second implementation in the tests).
 * Replaces models/sphere_conv.py:62-192 + the nn.Conv2d / nn.Linear 1x1 layers of models/DCAE.py:96-324. */
int ldc_sphere_conv_nhwc_split(const float* X, const void* Wp, const float* bias, const float* R, float* Y, int B, int H, int W,
                               int cin, int ldx, int cout, int ldy, int ldr, int ksize, int act, int in_fmt, int out_fmt,
                               void* workspace, long long workspace_bytes, void* stream);
/* ABI 4: 3 x 3 convs in the split-bf16 / bf16 formats run a second kernel where the shape fills its tiles (conv_halo.hip): a workgroup
 * owns a TH x TW pixel tile and DMAs the tile's (TH + 2) x (TW + 2) halo image into LDS once per channel chunk - sphere padding folded
 * into the source addresses - instead of once per (tap, chunk); same arguments, same arithmetic per output element (k order: chunk-major
 * instead of tap-major, so results differ from the gathered kernel in the last bits of the fp32 accumulation only).
 * ldc_sphere_conv_plan tells which kernel a call shape gets: returns 1 and the tile (rows = TH x TW: 128 | 256; width TW) for the
 * halo-staged kernel, 0 for the gathered one (ksize != 3, fp32 rows, images that would leave most of the tiles' rows empty).  The plan
 * assumes the caller passes the full grouped-GEMM workspace (ldc_gemm_grouped_workspace_bytes): shapes that put two workgroups on a
 * tile need LDC_GEMM_COUNTER_BYTES + 2 * tiles * 128 KiB of it, and a call with less runs the gathered kernel whatever the plan says.
 * Both kernels validate alike: X 32-byte / Wp and workspace 16-byte aligned, ldx % 8 == 0 (LDC_ERR_ALIGN), act in range
 * (LDC_ERR_UNSUPPORTED), ldr >= cout when R is given (LDC_ERR_ARG). */
int ldc_sphere_conv_plan(int B, int H, int W, int cin, int cout, int ksize, int in_fmt, int* tile_rows, int* tile_w);
/* Producers of operand rows.  `ys` = operand copy in format `fmt` (LDC_FMT_SPLIT | LDC_FMT_BF16; lds % 8 == 0, >= C rounded up to
 * 8, 32-byte aligned; pad columns zeroed), `y` = fp32 copy (the residual stream / inputs of depthwise convs); either may be NULL,
 * not both. */
int ldc_rmsnorm_rows_split(const float* x, const float* w, const float* b, const float* resid, float* y, float* ys, long long rows,
                           int C, int ldx, int ldr, int ldy, int lds, int fmt, float eps, int act, void* stream);
int ldc_pixel_unshuffle_shortcut_split(const float* cv, const float* x, float* y, float* ys, int B, int H2, int W2, int cout,
                                       int cin, int lds, int fmt, void* stream);
int ldc_pixel_shuffle_shortcut_split(const float* cv, const float* x, float* y, float* ys, int B, int H, int W, int cout, int cin,
                                     int lds, int fmt, void* stream);
/* ABI 4: `cv` of ldc_pixel_shuffle_shortcut_split may be NULL - y is then the block's shortcut term alone,
 * pixel_shuffle(x.repeat_interleave(4 cout / cin, dim=1), 2) (models/DCAE.py:526-530), which the interpolate form of the up block adds to
 * its conv as the residual operand R.  ldc_upsample_nearest2x_rows: F.interpolate(scale_factor=2, mode="nearest") on NHWC rows
 * (models/DCAE.py:519-522, upsample_block_type = "interpolate"): y[b, 2h+i, 2w+j, :] = x[b, h, w, :], as fp32 rows (y, ldy) and / or operand
 * rows (ys, lds, fmt) for the conv that follows; C % 4 == 0. */
int ldc_upsample_nearest2x_rows(const float* x, float* y, float* ys, int B, int H, int W, int C, int ldx, int ldy, int lds, int fmt, void* stream);
/* ABI 5 (round 6): the DC-AE family's `layers_per_block[0] == 0` form (models/DCAE.py:559-579,702-712: no stage at full resolution - the encoder's
 * conv_in is a DCDownBlock2d and the decoder's conv_out a DCUpBlock2d, both WITHOUT shortcut).  `x` of ldc_pixel_unshuffle_shortcut_split may be
 * NULL: y = pixel_unshuffle(cv) alone.  ldc_pixel_shuffle_to_chan: pixel_shuffle of the conv's NHWC rows cv [B][H][W][4 cout] written straight as
 * the NCHW result out [B][keep][2H][2W], the first `keep` <= cout channels (any cout: field counts need not be multiples of 4). */
int ldc_pixel_shuffle_to_chan(const float* cv, float* out, int B, int H, int W, int cout, int keep, void* stream);
int ldc_split_rows(const float* x, float* ys, long long rows, int C, int ldx, int lds, int fmt, void* stream);
/* ldc_sphere_dwconv_nhwc / ldc_relu_linear_attn_nhwc with the output format as an argument (LDC_FMT_F32 | _SPLIT | _BF16). */
int ldc_sphere_dwconv_nhwc_fmt(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W, int C, int ldx,
                               int ldy, int ksize, int glu, int out_fmt, void* stream);
int ldc_relu_linear_attn_nhwc_fmt(const float* qkv, float* y, int B, int P, int groups, int ldq, int ldy, float eps, int out_fmt,
                                  void* workspace, long long workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Ensemble scoring of one lead time of a decoded forecast:
 *   forecast [M members][C][H*W] with member / channel strides in elements (the H*W plane contiguous), so a
 *   `[:, :, t]` slice of the reference's (ens, C, T, H, W) array needs no copy; truth / clim [C][H*W] with a channel
 *   stride (clim NULL = no ACC); lat_weight [H]; M <= 64.
 *   out [5][C] = ens_acc, ens_mse, crps_spread, crps_skill, crps, each the (nan)mean over the grid of the
 *   latitude-weighted point values; channel `nan_channel` (the SST channel, NaN over land) uses nanmean, the others
 *   mean (-1 = none); ACC always nanmean.  skill_map / spread_map: optional [C][H*W] unweighted point values.
 * Replaces pointwise_crps_skill / pointwise_crps_spread / get_crps / get_acc (ladcast/evaluate/utils.py:51-149) and
 * the per-lead-time block of ladcast/evaluate/evaluate_ens_gpu.py:339-425; one pass over the forecast.
 * workspace: ldc_ensemble_scores_workspace_bytes(C, H, W) bytes of device scratch.
 * ------------------------------------------------------------------------- */
long long ldc_ensemble_scores_workspace_bytes(int C, int H, int W);
int ldc_ensemble_scores(const float* forecast, long long member_stride, long long channel_stride,
                        const float* truth, long long truth_channel_stride, const float* clim,
                        long long clim_channel_stride, const float* lat_weight, int M, int C, int H, int W,
                        int nan_channel, float* out, float* skill_map, float* spread_map, void* workspace,
                        long long workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LADCAST_HIP_H */
