/*
 * bifold_hip.h -- C ABI of libbifold_hip.so: the MI355X (gfx950 / CDNA4) kernels
 * behind BiFold's VLM training step.
 *
 * The reference (Barbany/bifold) has NO native boundary on this path: every FLOP
 * is a stock ATen kernel reached through nn.Module.forward.  This ABI is the
 * boundary the build introduces *beneath* the reference's Python plugin surface
 * (bifold.models / bifold.losses / bifold.optim); each entry point names the
 * reference call site(s) it replaces (paths relative to the reference's bifold/).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 *    name ends in _host; the caller owns all memory: the library never allocates, frees
 *    or synchronises;
 *  - process-wide state, all of it internal and none of it data: a thread-local error
 *    string; ONE table of four knobs (csrc/tuning.h: kernel selection only, results are the same up to
 *    summation order) filled from BIFOLD_GEMM_PIPE / BIFOLD_P8_GRID / BIFOLD_ATTN_PIPE / BIFOLD_ATTN_BWD_FUSED_P exactly once per process under std::call_once,
 *    plus the bf_gemm_set_variant hook, which stores into an atomic of that table and may be called at any time from
 *    any thread; and, for the persistent
 *    GEMM, a __device__ array of work-ticket counters (one copy per device) with a host-side table per
 *    device that hands each stream its own ring of counter sets: 16 streams per device hold rings at a
 *    time, a further stream takes over the least recently used ring after one hipDeviceSynchronize();
 *    the table is keyed by the device current at the call, so one process may drive several devices;
 *  - every launch goes to the hipStream_t passed as `stream` (void*; NULL = the
 *    legacy default stream);
 *  - tensors are row-major and contiguous unless a leading dimension is given;
 *    pointers must be 16-byte aligned;
 *  - dtype codes: BF_F32 = 0 (IEEE fp32), BF_BF16 = 1 (bfloat16), BF_F16 = 2 (IEEE binary16, CLIP's native
 *    dtype, models/clip/model.py:480-506 `convert_weights`; same MFMA rate as bf16 on gfx950, 2^-11 instead of
 *    2^-9 operand rounding; wherever a comment below says "bf16" for a 16-bit tensor, BF_F16 is accepted too and
 *    all 16-bit tensors of one call share one type).  "cdt" below is
 *    the compute dtype of a call: operands of matrix products; accumulation,
 *    LayerNorm statistics, softmax, the residual stream, losses and optimiser
 *    state are always fp32;
 *  - return value: 0 on success, a negative BF_E* code otherwise;
 *    bf_last_error() returns a human-readable description for this thread.
 */
#ifndef BIFOLD_HIP_H
#define BIFOLD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_ABI_VERSION 12

enum { BF_F32 = 0, BF_BF16 = 1, BF_F16 = 2 };

enum {
    BF_OK = 0,
    BF_EINVAL = -1,     /* bad shape / dtype / flag combination        */
    BF_EALIGN = -2,     /* pointer or leading dimension not aligned    */
    BF_ELAUNCH = -3,    /* hipGetLastError() != hipSuccess after launch */
    BF_EWORKSPACE = -4  /* workspace too small                          */
};

typedef void* bf_stream_t;

int bf_abi_version(void);
/* sizeof of the structs that cross the boundary (0: bf_gemm_args, 1: bf_gemm_tn_problem, 2: bf_lora_slots; else 0): a
 * binding checks its mirror against it once. */
size_t bf_struct_size(int which);
const char* bf_last_error(void);
int bf_last_hip_error(void);

/* ------------------------------------------------------------------------
 * LayerNorm (eps inside the sqrt, biased variance; fp32 statistics).
 * Replaces nn.LayerNorm at models/clip/model.py:178-184 (ln_pre, ln_1, ln_2,
 * ln_post, ln_final) and models/modules.py:11 (PreNorm.norm).
 *   x [rows, cols] x_dtype -> y [rows, cols] y_dtype; mean/rstd [rows] fp32 (may
 *   be NULL when not needed for a backward pass).
 * ---------------------------------------------------------------------- */
int bf_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                     int y_dtype, float* mean, float* rstd, int rows, int cols, float eps,
                     bf_stream_t stream);

/* Residual add fused with the NEXT LayerNorm (models/clip/model.py:217-218 + :178-184; transformer.py:62-63 +
 * modules.py:11):  x_out = x + delta ;  y = LN(x_out).  delta = the 16-bit output (delta_dtype = BF_BF16 / BF_F16, bias
 * included; a 16-bit y has the same type) of the branch's last
 * GEMM -- out-proj or MLP fc2 -- which can then run its plain-bf16 epilogue instead of loading a fp32 residual tile
 * and storing a fp32 one.  x_out may alias x (frozen towers update the stream in place).  cols in {256, 512, 768,
 * 1024}; mean / rstd optional. */
int bf_add_layernorm_fwd(const float* x, const void* delta, int delta_dtype, const float* gamma, const float* beta,
                         float* x_out, void* y, int y_dtype, float* mean, float* rstd, int rows, int cols,
                         float eps, bf_stream_t stream);

/* The same with a SECOND branch and an optional x_out (ABI 9): x_sum = (x + delta) + delta2 ; y = LN(x_sum) ; x_out = x_sum
 * when x_out is not NULL.  For a FROZEN tower (no backward reads the stream between the two branches of a block,
 * models/clip/model.py:215-219) the pass after the attention branch then only produces ln_2's output (x_out = NULL:
 * 8 bytes per element instead of 12) and the pass after the MLP adds both branches to the block's input at once
 * (delta2 = the MLP's output: 14 bytes per element instead of 12): 22 instead of 24 per block, the fp32 stream written
 * once.  The sum is taken in the order the two separate passes take it (same bits).  delta2 may be NULL (= the entry
 * point above). */
int bf_add2_layernorm_fwd(const float* x, const void* delta, const void* delta2, int delta_dtype, const float* gamma,
                          const float* beta, float* x_out, void* y, int y_dtype, float* mean, float* rstd, int rows,
                          int cols, float eps, bf_stream_t stream);

/* Backward of the above fused with the residual branch:
 *   dx = (dres ? dres : 0) + dLN(dy)          fp32 [rows, cols]
 *   dx_cdt (optional, cdt_dtype) = cast(dx)   -- operand for the next dX/dW GEMMs
 *   dgamma_part/dbeta_part: fp32 [n_part, cols] per-workgroup partial sums,
 *   n_part = bf_layernorm_bwd_parts(rows); reduce them with bf_colsum.
 *   dres_sum_part / dx_sum_part (optional, both or neither): partial column sums of dres and
 *   of dx -- the bias gradients of the two residual-branch Linears around this LayerNorm
 *   (FeedForward.net.3 and Attention.to_out), which this kernel streams anyway.
 *   dx == NULL: the 16-bit gradient stream (fp16 mode of the fusion stack's backward) -- `dres`, if given, points to a
 *   cdt_dtype tensor (not fp32), dy is of cdt_dtype too, and the sum leaves ONLY as dx_cdt (required, 16-bit): the
 *   residual stream's gradient is then one 16-bit tensor per block boundary instead of an fp32 tensor and its copy. */
int bf_layernorm_bwd_parts(int rows);
int bf_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma,
                     const float* mean, const float* rstd, const float* dres, float* dx,
                     void* dx_cdt, int cdt_dtype, float* dgamma_part, float* dbeta_part,
                     float* dres_sum_part, float* dx_sum_part, int rows, int cols,
                     bf_stream_t stream);

/* ------------------------------------------------------------------------
 * GEMM with fused epilogue:  C = epilogue(alpha * op(A) * op(B)).
 * Replaces every nn.Linear / Conv2d-as-matmul on the path:
 *   conv1 patch-embed (models/clip/model.py:250-256,297-299), MHA in/out
 *   projections (:196,214-217), c_fc/c_proj (:198-204), project
 *   (models/rgb_clip.py:34,54), to_qkv/to_out (models/transformer.py:19-24),
 *   FeedForward (models/modules.py:21-27), and their autograd backward.
 *
 * Layouts (all row-major):
 *   trans_a = 0: A is [M, K]           trans_a = 1: A is stored [K, M]
 *   b_kn    = 0: B is [N, K] (weights) b_kn    = 1: B is stored [K, N]
 *   forward  y = x W^T      : trans_a 0, b_kn 0
 *   backward dx = dy W      : trans_a 0, b_kn 1
 *   backward dW = dy^T x    : trans_a 1, b_kn 1   (split-K over tokens, see workspace)
 * cdt = dtype of A and B.  Epilogue, applied in this order on the fp32 accumulator:
 *   v = alpha*acc + bias[n]
 *   act: BF_ACT_NONE | BF_ACT_QUICK_GELU (x*sigmoid(1.702x), clip/model.py:187-189)
 *        | BF_ACT_ERF_GELU (nn.GELU, modules.py:23); `pre` (cdt, optional) receives v
 *        before the activation (saved for backward);
 *        BF_ACT_ERF_GELU_GRAD / BF_ACT_TANH_GELU_GRAD: v *= gelu'(aux[m,n]) (aux = saved
 *        pre-activation, cdt)
 *   v += residual[r(m), n]  (fp32; row r = m % resid_mod if resid_mod > 0, else the
 *        output row)
 *   out row = (m / rows_in) * rows_out + row_off + m % rows_in   (rows_in = 0: identity)
 *   C[out row, n] (c_dtype) = v   (or += v when accumulate != 0: fp32 math on the stored value, one more
 *        rounding for a bf16 C);
 *   c2 (optional, BF16) receives a bf16 copy of v at the same position.
 * ---------------------------------------------------------------------- */
enum { BF_ACT_NONE = 0, BF_ACT_QUICK_GELU = 1, BF_ACT_ERF_GELU = 2, BF_ACT_ERF_GELU_GRAD = 3,
       BF_ACT_TANH_GELU = 4 /* gelu_pytorch_tanh: the MLP of transformers' SiglipModel */,
       BF_ACT_TANH_GELU_GRAD = 5 /* v *= d/dx gelu_pytorch_tanh at aux[m,n] (LoRA: tower backward) */,
       BF_ACT_ERF_GELU_SAVEGRAD = 6 /* C = gelu(v) and `pre` (required) receives gelu'(v) INSTEAD of v: the erf is
                                     * evaluated once for both, and the backward GEMM's epilogue becomes ... */,
       BF_ACT_MUL_AUX = 7 /* ... v *= aux[m,n] (aux = the saved gelu'(v), cdt) */ };

typedef struct bf_gemm_args {
    const void* a;
    const void* b;
    void* c;
    int m, n, k;
    int lda, ldb, ldc;       /* leading dimensions in elements                     */
    int cdt;                 /* dtype of a, b (and of pre / aux)                   */
    int c_dtype;             /* dtype of c                                         */
    int trans_a, b_kn;
    float alpha;
    const float* bias;       /* [n] or NULL                                        */
    int act;
    void* pre;               /* optional [m, ldc] cdt                              */
    const void* aux;         /* BF_ACT_*_GELU_GRAD: [m, ld_aux] cdt                */
    int ld_aux;
    const float* residual;   /* optional fp32, leading dimension ld_res            */
    int ld_res;
    int resid_mod;
    int rows_in, rows_out, row_off;
    int accumulate;
    void* c2;                /* optional bf16 copy, leading dimension ldc          */
    int split_k;             /* >1: K is split, fp32 partials go through workspace  */
    void* workspace;         /* split_k * m * n * 4 bytes when split_k > 1          */
    size_t workspace_bytes;
    float* colsum_out;       /* optional fp32 [n]: column sums of c (the bias gradient of the layer that produced
                              * c's upstream: d_bias = sum over rows of dY), overwritten.  Fused into the epilogue
                              * where the kernel can (16-bit GELU-grad path: the fp32 values are summed before they
                              * are rounded for the store), otherwise the stored c is summed by a bf_colsum launch
                              * after the GEMM; needs rows_in == 0 */
    /* ABI 11: optional LOW-RANK ADDEND of the product, c = epilogue(alpha * (a b^T + lr_u lr_w^T)): lr_u [m, lr_k],
     * lr_w [n, lr_k], both cdt, row-major and dense, lr_k = 16 or 32.  The LoRA branch of an adapted projection riding
     * in the projection's GEMM (models/siglip.py:33-40 -> peft: `result + lora_B(lora_A(dropout(x))) * scaling`):
     * lr_u = the adapters' down-projections side by side (bf_lora_down's 16-bit copy), lr_w = scaling * lora_B placed in
     * the rows and columns of its projection, zero elsewhere.  One extra MFMA K-step per output tile in front of the
     * epilogue instead of a read-modify-write pass over c.  Carried by the persistent 256 x 256 kernel's plain 16-bit
     * epilogue only: bf_gemm_lowrank_ok says whether a call qualifies (large-M NT GEMM, k % 64 == 0, 16-bit c, no
     * residual / aux / c2 / accumulate / split); bf_gemm refuses lr_u on a call that does not. */
    const void* lr_u;
    const void* lr_w;
    int lr_k;
} bf_gemm_args;

int bf_gemm(const bf_gemm_args* args, bf_stream_t stream);
/* 1 when bf_gemm would carry args' low-rank addend (pointers are not looked at; lr_k must be set), else 0. */
int bf_gemm_lowrank_ok(const bf_gemm_args* args);

/* Tuning hook: force a bf16 kernel variant for every later bf_gemm call of this process
 * (0 = 128x128x64, 4 = 128x128x32 with 4 workgroups per CU, 8 = persistent 256x256 staggered;
 * a variant that cannot run a shape falls back to the automatic choice), -1 = automatic.
 * Same effect as the BIFOLD_GEMM_PIPE environment variable. */
/* Test hook: launches `blocks` (<= 256) workgroups that each take one CU (150 KB of LDS) and spin for `cycles`
 * shader clocks -- a stand-in for a collective's resident kernels when measuring how bf_gemm behaves with part
 * of the chip taken (tools/gemm_disturb.py). */
int bf_debug_spin(int blocks, long long cycles, bf_stream_t stream);
int bf_gemm_set_variant(int variant);
/* The bf16 kernel the calling thread's last bf_gemm launched: 8 / 4 / 0 as above, 1 = the
 * register-staged fallback, -1 = the exact fp32 kernel (profiling label, bench.py). */
int bf_gemm_last_variant(void);

/* A GROUP of deep weight-gradient GEMMs c_p[m_p, n_p] = alpha * dy_p^T x_p that share the contraction length k (the
 * token count): the four dW of one transformer layer (models/transformer.py:19-24, models/modules.py:21-27 under
 * autograd) in ONE launch of the persistent kernel plus one reduction.  On their own each of these GEMMs has too few
 * 256 x 256 output tiles for 256 CUs and is split 16 - 64 ways along k, writing as many fp32 partial copies of its
 * result; the union of a layer's tiles needs 5.  dy_p: [k, ld_dy] cdt (token-major dY), x_p: [k, ld_x] cdt, c_p fp32
 * [m_p, ldc], overwritten.  bf_gemm_tn_grouped_workspace_bytes returns 0 when the group does not qualify (fp32
 * operands, k < 16 384 or not a multiple of 32, m / n / ld not multiples of 8, more tiles than half the CUs, more than
 * 8 problems): the caller then runs the problems through bf_gemm one by one.  Partials are summed in split order:
 * the result is reproducible. */
typedef struct bf_gemm_tn_problem {
    const void* dy;
    const void* x;
    float* c;
    int m, n, ld_dy, ld_x, ldc;
} bf_gemm_tn_problem;
size_t bf_gemm_tn_grouped_workspace_bytes(const bf_gemm_tn_problem* problems, int count, int k, int cdt);
int bf_gemm_tn_grouped(const bf_gemm_tn_problem* problems, int count, int k, int cdt, float alpha, void* workspace,
                       size_t workspace_bytes, bf_stream_t stream);

/* Optional workspace for an un-split (split_k <= 1) bf16 GEMM: when the persistent 256x256 kernel
 * would leave most CUs idle in its last round of tiles, bf_gemm cuts the rows of that round off
 * and runs them split along K through `workspace`.  Returns the bytes to pass (0: not applicable;
 * passing no workspace is always valid and just skips the optimisation). */
size_t bf_gemm_tail_workspace_bytes(const bf_gemm_args* g);

/* ------------------------------------------------------------------------
 * fp8 (OCP e4m3) forward path of the frozen towers (BASELINE.json configs[4]; no reference counterpart: the
 * reference is fp32 end to end).  Replaces, for the towers only, the LayerNorm -> in-projection and
 * LayerNorm -> MLP fc1 pairs of models/clip/model.py:192-219 (and of transformers' SiglipEncoderLayer):
 *   bf_layernorm_fwd_fp8   y8[m, :] = e4m3(LN(x[m, :]) / s[m]),  s[m] = max |LN(x[m, :])| / 448     (x fp32)
 *   bf_quantize_rows_fp8   q8[n, :] = e4m3(w[n, :] / s[n]),       s[n] = max |w[n, :]| / 448         (once per load)
 *   bf_gemm_fp8            C[m, n] = act(sa[m] sb[n] sum_k a8[m, k] b8[n, k] + bias[n])   (c_dtype BF_BF16 / BF_F32)
 * on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate, half the operand bytes); k % 128 == 0,
 * lda / ldb in bytes = elements, multiples of 16.  act: NONE / QUICK_GELU / ERF_GELU / TANH_GELU. */
int bf_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_fp8, float* row_scale,
                         int rows, int cols, float eps, bf_stream_t stream);
int bf_quantize_rows_fp8(const void* w, int w_dtype, void* q_fp8, float* row_scale, int rows, int cols, int ld,
                         bf_stream_t stream);
int bf_gemm_fp8(const void* a_fp8, const float* a_scale, const void* b_fp8, const float* b_scale, void* c,
                int c_dtype, const float* bias, int act, int m, int n, int k, int lda, int ldb, int ldc,
                bf_stream_t stream);

/* Column sums  out[n] (+)= sum_m x[m, n]  (bias gradients, LayerNorm partials). */
int bf_colsum(const void* x, int x_dtype, float* out, int rows, int cols, int ld, int accumulate,
              bf_stream_t stream);
/* Column sums of up to four stacked fp32 partial arrays parts[n_arrays][nparts][cols] in one launch,
 * each written (not accumulated) to its own out vector [cols]: the reduction of bf_layernorm_bwd's
 * dgamma / dbeta / dres-sum / dx-sum partials. */
int bf_colsum_multi(const float* parts, int n_arrays, int nparts, int cols, float* out0, float* out1,
                    float* out2, float* out3, bf_stream_t stream);

/* ------------------------------------------------------------------------
 * Multi-head attention on a packed QKV buffer.
 * Replaces F.multi_head_attention_forward under nn.MultiheadAttention
 * (models/clip/model.py:196,207-214; causal mask :414-420) and
 * Attention.forward (models/transformer.py:26-41) incl. its query-row mask.
 *   qkv [B, L, 3, H, D] cdt (the output of the in-projection GEMM, untouched)
 *   out [B, L, H*D] cdt      lse [B, H, L] fp32 (log-sum-exp of the scaled, masked
 *   scores; NULL in inference)
 *   mask_mode: BF_MASK_NONE | BF_MASK_CAUSAL (key j > query i -> -inf)
 *            | BF_MASK_QUERY_ROW (row_mask[b, i] == 0 -> the whole score row is
 *              filled with -1e5, i.e. that query attends uniformly; this is what
 *              the reference does, transformer.py:33-35)
 *   D in {32, 48, 64}.
 * ---------------------------------------------------------------------- */
enum { BF_MASK_NONE = 0, BF_MASK_CAUSAL = 1, BF_MASK_QUERY_ROW = 2 };

int bf_attention_fwd(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch,
                     int seq, int heads, int head_dim, float scale, int mask_mode, int cdt,
                     bf_stream_t stream);

/* dqkv [B, L, 3, H, D] cdt from dout [B, L, H*D] cdt; delta_ws fp32 [B, H, L] scratch (rowsum(dO . O), written by the
 * dQ kernel, read by the dK / dV kernel).  ABI 10: no other workspace -- the chunked kernel that summed dQ across key
 * chunks with fp32 atomics (dq_ws, bf_attention_bwd_needs_dq_ws) and round 5's opt-in one-pass kernel
 * (bf_attention_bwd_ws_elems) are gone; every sequence beyond one workgroup runs a dQ / dK+dV kernel pair that needs
 * neither.  Replaces the autograd of models/transformer.py:28-41 and of the towers' attention under LoRA. */
int bf_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                     const int32_t* row_mask, void* dqkv, float* delta_ws, int batch,
                     int seq, int heads, int head_dim, float scale, int mask_mode, int cdt,
                     bf_stream_t stream);

/* ABI 10: nn.Dropout on the attention PROBABILITIES (models/transformer.py:18,36: `attn = self.dropout(attn)` after the
 * softmax, active in training when the model's `dropout` > 0; every shipped YAML says 0).  Same tensors as
 * bf_attention_fwd / _bwd; element (b, h, i, j) of P is kept iff bf_dropout_mask's function keeps index
 * ((b H + h) L + i) L + j under `seed`, kept probabilities are scaled by 1 / (1 - p) into P V while the softmax normaliser
 * sums them as they are; the backward regenerates the mask.  Correct-first: fp32 math on the vector units for every
 * storage type (the MFMA kernels carry no dropout). */
int bf_attention_dropout_fwd(const void* qkv, void* out, float* lse, const int32_t* row_mask, int batch, int seq,
                             int heads, int head_dim, float scale, int mask_mode, int cdt, float p,
                             unsigned long long seed, bf_stream_t stream);
int bf_attention_dropout_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                             const int32_t* row_mask, void* dqkv, float* delta_ws, int batch, int seq, int heads,
                             int head_dim, float scale, int mask_mode, int cdt, float p, unsigned long long seed,
                             bf_stream_t stream);

/* ------------------------------------------------------------------------
 * Token / patch plumbing.
 * ---------------------------------------------------------------------- */
/* im2col for the stride=patch conv (models/clip/model.py:297-299):
 * img [B, 3, S, S] fp32 -> cols [B*g*g, 3*P*P] cdt, column order (c, ky, kx) =
 * conv1.weight.flatten(1), row order = row-major patches. */
int bf_patchify(const float* img, void* cols, int cdt, int batch, int size, int patch,
                bf_stream_t stream);
/* Same for any EVEN patch size (ViT-L/14) with a padded row length: cols [B*g*g, ld], columns
 * [3 p^2, ld) zero, so that the patch-embed GEMM's K is a multiple of its tile (588 -> 640). */
int bf_patchify_ld(const float* img, void* cols, int cdt, int batch, int size, int patch, int ld,
                   bf_stream_t stream);

/* Vision-token assembly + ln_pre (models/clip/model.py:300-309):
 * x[b,0,:] = cls + pos[0]; x[b,1+p,:] = patches[b*n+p,:] + pos[1+p]; x = LN(x). */
int bf_vision_assemble(const float* patches, const float* cls, const float* pos,
                       const float* gamma, const float* beta, float* x, int batch, int n_patch,
                       int width, float eps, bf_stream_t stream);

/* K1 of SURVEY.md 2.2 (models/clip/model.py:297-309 + the first block's ln_1, :206-207): the patch-embed GEMM
 * (bf_gemm with rows_in = n, rows_out = n + 1, row_off = 1, residual = pos[1:], resid_mod = n) has written
 * patches + pos[1 + p] into rows b*(n+1) + 1 + p of x.  In place: row b*(n+1) = cls + pos0; x = ln_pre(x) (g0, b0);
 * h = ln_1(x) (g1, b1) in h_dtype.  One pass over the stream instead of assembly, ln_pre and ln_1. */
int bf_vision_lnpre_ln1(float* x, const float* cls, const float* pos0, const float* g0, const float* b0,
                        const float* g1, const float* b1, void* h, int h_dtype, int batch, int n_patch,
                        int width, float eps, bf_stream_t stream);

/* Token-embedding gather + positional add (models/clip/model.py:448-450):
 * x[b,t,:] = table[tokens[b,t],:] + pos[t,:]; ids out of [0, vocab) are an error
 * flag written to err_flag (device int, may be NULL). */
int bf_embed_tokens(const int32_t* tokens, const float* table, const float* pos, float* x,
                    int batch, int ctx, int width, int vocab, int* err_flag, bf_stream_t stream);

/* Fusion-input assembly (models/rgb_clip.py:55,59-66 + models/fusion.py:45-56):
 *   X[b, 0, :]        = text_token + text_pos[0] + type[0]
 *   X[b, 1+t, :]      = text_feat[b,t,:] + text_pos[1+t] + type[0]      t < n_text
 *   X[b, 1+n_text+p,:]= rgb_feat[b,p,:] + rgb_pos[p] + type[1]          p < n_rgb
 * X is fp32 [B, 1+n_text+n_rgb, dim]; text_feat fp32 (frozen ln_final output),
 * rgb_feat fp32 (project output incl. bias). */
int bf_fusion_assemble(const float* text_feat, const float* rgb_feat, const float* text_token,
                       const float* text_pos, const float* rgb_pos, const float* type_emb, float* x,
                       int batch, int n_text, int n_rgb, int dim, bf_stream_t stream);

/* Its backward: dX [B, L, dim] (dx_dtype: fp32, or -- the fp16 mode's 16-bit gradient stream -- fp16 together with an
 * fp16 d_rgb_feat) -> d_rgb_feat (cdt, [B*n_rgb, dim]: the dY operand of
 * project's backward GEMMs) and the batch-reduced parameter gradients (fp32, overwritten):
 * d_text_token [dim], d_text_pos [1+n_text, dim], d_rgb_pos [n_rgb, dim], d_type [2, dim]. */
int bf_fusion_assemble_bwd(const void* dx, int dx_dtype, void* d_rgb_feat, int cdt, float* d_text_token,
                           float* d_text_pos, float* d_rgb_pos, float* d_type, int batch,
                           int n_text, int n_rgb, int dim, bf_stream_t stream);

/* Generic fusion-input segment (models/siglip.py:56-70,85-137 + models/fusion.py:45-56): one
 * modality segment of 1+n tokens written into the concatenated fusion input X [B, seq_total, dim]:
 *   X[b, row_off, :]       = lead[:]            (+ pos[0, :])   + type_vec[:]
 *   X[b, row_off+1+t, :]   = src[b, t, :]       (+ pos[1+t, :]) + type_vec[:]       t < n
 * src fp32 with batch stride src_bstride elements (a context frame of a [B, T, n, dim] tensor is
 * a segment with src offset t*n*dim and stride T*n*dim); lead = the learned image/text token;
 * pos may be NULL (SigLip) or the frame's rows of context_pos_embedding (SiglipSequential). */
int bf_segment_assemble(const float* src, long src_bstride, const float* lead, const float* pos,
                        const float* type_vec, float* x, int batch, int n, int seq_total, int row_off,
                        int dim, bf_stream_t stream);
/* Its backward w.r.t. the parameters (src comes from a frozen tower):
 *   tmp[t, :] = sum_b dx[b, row_off+t, :]    t < 1+n       (tmp_ws fp32 [(1+n), dim])
 *   d_pos[t, :]  = tmp[t, :]   (if d_pos != NULL; overwritten)
 *   d_lead[:]   += tmp[0, :] ;  d_type[:] += sum_t tmp[t, :]      (accumulated: callers zero first) */
int bf_segment_assemble_bwd(const float* dx, float* d_lead, float* d_pos, float* d_type, float* tmp_ws,
                            int batch, int n, int seq_total, int row_off, int dim, bf_stream_t stream);

/* ------------------------------------------------------------------------
 * Heatmap decoder (collapsed) + BCE.
 * ConvDecoder (models/modules.py:33-74) has no nonlinearity: its five 1x1 convs fold
 * into one vector w_eff[D] and scalar b_eff, and its four bilinear x2 upsamples into a
 * fixed separable operator U [16g, g] (rows sum to 1).  logits = U (tok . w_eff + b_eff) U^T.
 * ---------------------------------------------------------------------- */
/* w[5] / b[5]: the conv weights [Cout, Cin] / biases, channels D, D/2, D/2, D/4, D/4, 1.
 * rows_ws: fp32 scratch >= 4*D floats (keeps the partial products r2..r5 for backward). */
int bf_decoder_collapse_fwd(const float* const* w, const float* const* b, float* w_eff,
                            float* b_eff, float* rows_ws, int dim, bf_stream_t stream);
int bf_decoder_collapse_bwd(const float* const* w, const float* const* b, const float* rows_ws,
                            const float* g_w_eff, const float* g_b_eff, float* const* dw,
                            float* const* db, int dim, bf_stream_t stream);

/* low[h, b, p] = tok[b, 1+p, :] . w_eff[h, :] + b_eff[h]     (models/pickplace.py:71 drops
 * token 0 and lays tokens out as a row-major g x g grid).  tok is fp32 with rows of `dim`
 * and a batch stride of tok_bstride elements (so the last-modality slice of the fusion
 * output is used in place); low is head-major fp32 [n_heads, B, n_patch]. */
int bf_decoder_project(const float* tok, long tok_bstride, const float* w_eff, const float* b_eff,
                       float* low, int batch, int n_heads, int n_patch, int dim, bf_stream_t stream);
/* heat[i, :, :] = sigmoid(U low[i] U^T) for i < n_maps (= n_heads * B, head-major)
 * (models/pickplace.py:108-123), fp32 [n_maps, S, S], S = 16 g;  U fp32 [S, g].
 * U is the composition of the decoder's four bilinear x2 steps and therefore BANDED: U[x][j] == 0 unless
 * |j - x / 16| <= 1.  Both this kernel and its backward sum over that band only (three terms per pixel and axis instead
 * of g); bifold/models/decoder_math.py:upsample_operator builds U and checks the band. */
int bf_decoder_upsample_sigmoid(const float* low, const float* U, float* heat, int n_maps, int grid,
                                bf_stream_t stream);
/* Backward of sigmoid o upsample: g_low[i] = U^T (d_heat[i] * p (1-p)) U, fp32 [n_maps, g*g]. */
int bf_decoder_upsample_sigmoid_bwd(const float* heat, const float* d_heat, const float* U,
                                    float* g_low, int n_maps, int grid, bf_stream_t stream);

/* nn.BCELoss(mean) per head (losses/__init__.py:61,100-103) incl. the -100 log clamp.
 * heat/target fp32 [B, n_heads, pixels] addressed with the given strides (elements);
 * loss_part fp32 [n_heads, n_part] partial sums (n_part = bf_bce_parts(batch));
 * loss[h] = sum * inv_count is finished by bf_bce_finish. */
int bf_bce_parts(int batch);
int bf_bce_fwd(const float* heat, const float* target, float* loss_part, int batch, int n_heads,
               int pixels, long heat_bstride, long heat_hstride, long tgt_bstride,
               long tgt_hstride, bf_stream_t stream);
int bf_bce_finish(const float* loss_part, float* loss, int n_heads, int n_part, float inv_count,
                  bf_stream_t stream);
/* Backward of BCE(mean) w.r.t. the probabilities, exactly as torch computes it:
 *   d_heat = gscale[0] * inv_count * (p - t) / max(p (1-p), 1e-12)     (count elements) */
int bf_bce_bwd(const float* heat, const float* target, const float* gscale, float* d_heat,
               size_t count, float inv_count, bf_stream_t stream);

/* Backward of bf_decoder_project:
 *   d_tok[b, 1+p, :] = sum_h g_low[h,b,p] w_eff[h,:], d_tok[b,0,:] = 0   (dtok_dtype: fp32, or fp16 when the
 *   fusion backward carries its residual gradient in fp16; overwritten; batch stride dtok_bstride elements)
 *   g_w_eff[h,:] = sum_{b,p} g_low[h,b,p] tok[b,1+p,:], g_b_eff[h] = sum g_low[h,:,:]
 *   part_ws: fp32 scratch >= bf_decoder_project_bwd_ws(batch, n_heads, dim) floats. */
size_t bf_decoder_project_bwd_ws(int batch, int n_heads, int dim);
int bf_decoder_project_bwd(const float* tok, long tok_bstride, const float* w_eff, const float* g_low,
                           void* d_tok, int dtok_dtype, long dtok_bstride, float* g_w_eff, float* g_b_eff,
                           float* part_ws, int batch, int n_heads, int n_patch, int dim,
                           bf_stream_t stream);

/* ------------------------------------------------------------------------
 * The mask head (PickPlaceConvDecoder(compute_mask=True), models/pickplace.py:25-30,66-69,76-98,107-119; ABI 12).
 * The head itself is one more collapsed ConvDecoder: bf_decoder_collapse_fwd / bf_decoder_project /
 * bf_decoder_upsample_sigmoid with n_heads = 1 over the image tokens the fusion model was handed (type embedding
 * included: models/fusion.py:52 adds it in place).  What is new is the product and the losses on `mask_heatmap`.
 *
 * bf_mask_mul_fwd: out[h, i] = s[h, i] * mask[i], h < n_heads (the pick heads), i < per_head (= B * S * S); fp32.
 * bf_mask_mul_bwd: d_s[h] = d_out[h] * mask (zeros where d_out[h] == NULL: a head the loss ignores);
 *   d_mask = d_mask_in (NULL = 0: the loss does not read mask_heatmap) + sum_h d_out[h] * s[h] -- the sum is left
 *   out with detach != 0 (models/pickplace.py:79-87,110-114: `.detach()`); d_mask NULL = not wanted.
 *   d_out: host array of n_heads device pointers (n_heads <= 4). */
int bf_mask_mul_fwd(const float* s, const float* mask, float* out, int n_heads, size_t per_head, bf_stream_t stream);
int bf_mask_mul_bwd(const float* const* d_out, const float* s, const float* mask, const float* d_mask_in, float* d_s,
                    float* d_mask, int n_heads, size_t per_head, int detach, bf_stream_t stream);
/* dst[b, i] += src[b, i] for b < batch, i < row_elems (batch strides in elements; dst fp32 or fp16): the mask head's
 * token gradient added to the gradient of the fusion INPUT's image rows. */
int bf_rows_accumulate(const float* src, long src_bstride, void* dst, int dst_dtype, long dst_bstride, int batch,
                       long row_elems, bf_stream_t stream);
/* DiceLoss (losses/__init__.py:126-136): prob / target fp32 [batch, pixels];
 *   stats fp32 [batch, 4] <- {sum p t, sum p, sum t, 0} per sample (kept for the backward),
 *   loss[0] = sum_b 1 - (2 stats0 + 1) / (stats1 + stats2 + 1).
 * bf_dice_bwd: d_prob[b, j] = gscale[0] * -(2 t_j D_b - A_b) / D_b^2 with A, D the numerator / denominator above. */
int bf_dice_fwd(const float* prob, const float* target, float* stats, float* loss, int batch, int pixels,
                bf_stream_t stream);
int bf_dice_bwd(const float* target, const float* stats, const float* gscale, float* d_prob, int batch, int pixels,
                bf_stream_t stream);
/* FocalLoss (losses/__init__.py:139-155) on fp32 [count] probabilities / targets:
 *   f = alpha_t * BCE(p, t) * (1 - p_t)^gamma (BCE with torch's -100 clamp of each log; alpha < 0: no alpha_t),
 *   loss[0] = inv_rows * sum f   (`.mean(1).sum()` on [B, S, S]: inv_rows = 1 / S);
 *   loss_part: fp32 scratch of n_part (<= 1024) floats.
 * bf_focal_bwd: d_prob = gscale[0] * inv_rows * d f / d p, the BCE factor differentiated as torch does
 *   ((p - t) / max(p (1 - p), 1e-12)). */
int bf_focal_fwd(const float* prob, const float* target, float* loss_part, float* loss, size_t count, int n_part,
                 float inv_rows, float alpha, float gamma, bf_stream_t stream);
int bf_focal_bwd(const float* prob, const float* target, const float* gscale, float* d_prob, size_t count,
                 float inv_rows, float alpha, float gamma, bf_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimiser (optim/__init__.py:4-14 -> torch.optim.Adam / AdamW) over a flat fp32
 * arena: all trainable parameters, gradients and both moments live in four
 * contiguous fp32 buffers of `count` elements, so one launch updates the model.
 *   step is 1-based; decoupled != 0 selects AdamW's decay.
 *   shadow (optional, [count] of shadow_dtype = BF_BF16 / BF_F16) receives the 16-bit copy of the new
 *   parameters (the cdt operands of next step's GEMMs).
 *   grad_scale_dev (optional device fp32 scalar) multiplies the gradient first
 *   (gradient clipping, __main__.py:148-149; DDP averaging).
 *   skip_flag_dev (optional device counter): non-zero -> the launch leaves param / moments / shadow untouched
 *   (the BF_F16 mode's guarded step: a backward that produced a non-finite gradient is dropped like
 *   torch.cuda.amp.GradScaler drops optimizer.step(), with no host read; the reference trains in fp32,
 *   __main__.py:138-152, and has no counterpart).  skipped_dev (optional device fp32 scalar) = steps dropped so
 *   far; `step` is the caller's count minus the `skipped_known` drops it knew of when it computed the count, and
 *   the kernel takes skipped_dev[0] - skipped_known more off for the bias corrections (a dropped step does not
 *   advance Adam's t).
 * ---------------------------------------------------------------------- */
int bf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* shadow,
                 int shadow_dtype, size_t count, int step, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int decoupled, const float* grad_scale_dev,
                 const unsigned* skip_flag_dev, const float* skipped_dev, float skipped_known,
                 bf_stream_t stream);

/* sum of squares of x (fp32 [count]) -> out[0] (fp32; partial_ws >= 1024 floats). */
int bf_sumsq(const float* x, size_t count, float* partial_ws, float* out, bf_stream_t stream);
/* clip coefficient: out[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))
 * (torch.nn.utils.clip_grad_norm_, __main__.py:148-149). */
int bf_clip_coef(const float* sumsq, float max_norm, float* out, bf_stream_t stream);
/* torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) (__main__.py:148-149) for gradients that
 * live in ONE flat fp32 arena: norm_out[0] = ||grad||_2 (the value torch returns), then
 * grad *= min(1, max_norm / (norm + 1e-6)) in place; nothing crosses to the host.
 * ws: fp32 scratch >= 1026 floats. */
int bf_clip_grad_norm(float* grad, size_t count, float max_norm, float* ws, float* norm_out,
                      bf_stream_t stream);

/* x[i] *= scale in place (fp32 [count]); *nonfinite_count (optional device counter, never reset here) += the number
 * of results that are inf / nan.  The two ends of the BF_F16 mode's loss scaling: the gradient that enters the
 * trainable tail's backward is multiplied by S so that the 16-bit gradient tensors stay inside binary16's range
 * (BCE's mean puts 1 / (B S^2) in front of every logit gradient -- 8e-8 at B = 256, below binary16's smallest
 * subnormal), and each slice of the gradient arena by 1 / S as the backward completes it.  The reference trains in
 * fp32 (__main__.py:135-150) and has no counterpart. */
int bf_scale_check(float* x, size_t count, float scale, unsigned* nonfinite_count, bf_stream_t stream);
/* The same with the factor read from device memory (scale_dev[0]): the dynamic loss scale lives on the device. */
int bf_scale_check_dev(float* x, size_t count, const float* scale_dev, unsigned* nonfinite_count,
                       bf_stream_t stream);
/* End of a guarded optimiser step (after bf_adam_step with skip_flag_dev = nonfinite_flag): device-side schedule of
 * the loss scale, no host read.  state (fp32 [8]): [0] S, [1] 1 / S, [2] clean steps since the last change, [3] steps
 * dropped so far, [4] non-finite gradient elements seen so far, [5] overflows that found S already at min_scale.
 * flag != 0: S = max(S * backoff, min_scale), [3] += 1; else after growth_interval clean steps S = min(S * growth,
 * max_scale).  The flag is cleared.  Counterpart of torch.cuda.amp.GradScaler.update(); nothing in the reference
 * (fp32 training, __main__.py:138-152). */
int bf_gradscale_update(float* state, unsigned* nonfinite_flag, float backoff, float growth, int growth_interval,
                        float min_scale, float max_scale, bf_stream_t stream);

/* Inverted dropout with a replayable mask: y[i] (+)= keep(seed, i) ? x[i] / (1 - p) : 0, keep = a pure function
 * of (seed, element index) (splitmix64 of index / 4, 15 of its bits per element against p * 2^15), so the backward pass regenerates the mask from
 * the seed instead of storing it.  Replaces the nn.Dropout(lora_dropout) PEFT puts in front of lora_A
 * (models/siglip.py:33-40 -> LoraConfig(lora_dropout=...)).  x, y: dtype BF_F32 / BF_BF16, count % 4 == 0;
 * accumulate != 0 adds into y (the backward: dX += mask * dT / (1 - p)).  bf_dropout_mask writes the same
 * decisions as bytes (1 = kept) for a checker that has to replay them. */
int bf_dropout(const void* x, void* y, int dtype, size_t count, float p, unsigned long long seed, int accumulate,
               bf_stream_t stream);
int bf_dropout_mask(uint8_t* mask, size_t count, float p, unsigned long long seed, bf_stream_t stream);
/* ABI 10: dropout on rows [row0, row0 + nrows) of every group of x [batch, group_rows, cols], IN PLACE, then += add[cols]
 * (NULL: nothing added): the embeddings' nn.Dropout of RGBOnly (models/rgb_clip.py:39-42,56,65) on the text / image rows of
 * the assembled fusion input, followed by the modality's type embedding (models/fusion.py:52).  The mask is that of the
 * range's own [batch, nrows, cols] tensor under bf_dropout_mask.  p = 0: only the add. */
int bf_dropout_rows(void* x, int dtype, const float* add, int batch, int group_rows, int row0, int nrows, int cols,
                    float p, unsigned long long seed, bf_stream_t stream);

/* ------------------------------------------------------------------------
 * ABI 11: the LoRA adapter branch of the adapted projections of one layer, UNMERGED (training with lora_dropout > 0:
 * models/siglip.py:33-40 -> peft LoraLayer.forward, `result + lora_B(lora_A(dropout(x))) * scaling`;
 * conf/model/siglip_sequential.yaml:10-14).  One launch serves every adapted projection ("slot") of the layer; each wide
 * [rows, width] tensor is read once per pass; masks are regenerated from (seed, element index) like bf_dropout's.
 * 16-bit storage only (the fp32 mode composes the branch from bf_dropout + bf_gemm); rank 8 or 16, n_slots * rank <= 32,
 * width % 16 == 0.  u tensors are fp32 [rows, n_slots * rank], slot s in columns [s rank, (s + 1) rank).
 * ---------------------------------------------------------------------- */
typedef struct bf_lora_slots {
    int n_slots;                  /* adapted projections handled by the launch: 1..3 */
    int rank;                     /* r */
    int col_off[3];               /* element offset of slot s inside a row of the wide tensor (multiple of 8) */
    float p;                      /* lora_dropout of the masks; 0: no mask */
    unsigned long long seed[3];   /* p > 0: element (row, c) of slot s is element row * width + c of bf_dropout_mask(seed[s]) */
} bf_lora_slots;
/* u[:, slot s] = alpha (mask_s . x[:, col_off[s] : col_off[s] + width]) wt[s rank : (s + 1) rank, :]^T
 * forward: x = LayerNorm output, wt = lora_A rows stacked, alpha = 1 / (1 - p);
 * backward: x = d(packed projection output), col_off = the slot's column block, wt = scaling * lora_B^T, p = 0.
 * u_half (may be NULL): the same values in the compute dtype, [rows, n_slots * rank] -- bf_gemm's lr_u. */
int bf_lora_down(const void* x, int ldx, const void* wt, float* u, void* u_half, int rows, int width,
                 const bf_lora_slots* slots, float alpha, int cdt, bf_stream_t stream);
/* y[:, col_off[s] + c] += alpha mask_s . (u[:, slot s] wt[slot s rows, c]); slots either share col_off (one
 * read-modify-write for all: the backward's d(LayerNorm output), wt = lora_A, alpha = 1 / (1 - p)) or name disjoint column
 * blocks (the forward: y = packed projection output, wt = scaling * lora_B^T, p = 0). */
int bf_lora_up(const float* u, const void* wt, void* y, int ldy, int rows, int width, const bf_lora_slots* slots,
               float alpha, int cdt, bf_stream_t stream);
/* out[s] (fp32, [rank, width]; transposed != 0: [width, rank]) = alpha u[:, slot s]^T (mask_s . x[:, col_off[s] : + width]),
 * WRITTEN.  lora_A's gradient (x = LayerNorm output, u = du, masks) and lora_B's (x = d(projection output), u = the forward's
 * u, transposed, alpha = scaling).  ws: fp32 scratch of bf_lora_tn_workspace_elems elements (partial sums of row chunks,
 * added in a fixed order: results are reproducible bit for bit). */
size_t bf_lora_tn_workspace_elems(int rows, int width, const bf_lora_slots* slots);
int bf_lora_tn(const void* x, int ldx, const float* u, float* const* out, int transposed, float* ws, size_t ws_elems,
               int rows, int width, const bf_lora_slots* slots, float alpha, int cdt, bf_stream_t stream);

/* The backward's two masked passes in one walk over the rows (n_slots * rank <= 16; every slot on the same columns):
 *   out[s] (fp32 [rank, width], WRITTEN) = alpha du[:, slot s]^T (mask_s . x)           -- lora_A's gradient
 *   y += alpha sum_s mask_s . (du[:, slot s] wt[slot s rows])                            -- d(LayerNorm output)
 * = bf_lora_tn(x, du) followed by bf_lora_up(du, wt, y) with the same slots, at one mask evaluation instead of two. */
int bf_lora_bwd_input(const void* x, int ldx, const float* du, const void* wt, void* y, int ldy, float* const* out, float* ws,
                      size_t ws_elems, int rows, int width, const bf_lora_slots* slots, float alpha, int cdt,
                      bf_stream_t stream);

/* Elementwise cast between fp32 and bf16 (count elements). */
int bf_cast(const void* src, int src_dtype, void* dst, int dst_dtype, size_t count,
            bf_stream_t stream);

/* ----------------------------------------------------------------------
 * Either side of the step: inference post-processing and target synthesis.
 *
 * bf_heatmap_pick: per map, the arg-max pixel of `heat` (fp32 [n_maps, size, size], first
 *   maximum in row-major order like numpy.argmax), optionally snapped to the nearest
 *   non-zero pixel of `mask` (fp32 [n_maps, size, size] or NULL; squared Euclidean distance,
 *   first minimum in row-major order like np.argwhere(...).argmin()).  Replaces
 *   sample_from_heatmap + nearest_to_mask (bifold/models/utils.py:60-83).
 *   xy int32 [n_maps, 2] = (x = column, y = row); conf fp32 [n_maps] = heat at that pixel.
 *   A map whose mask is empty gets xy = (-2, -2) (the reference raises there).
 * bf_gaussian_targets: the training target maps of Processor._make_gaussmap
 *   (bifold/data/processor.py:405-439).  points fp32 [n_maps, max_points, 2] as (x, y)
 *   pixels, counts int32 [n_maps] valid points per map (NULL = all max_points); count 0
 *   gives the all-zero map (:409-410), count 1 or strategy FIRST the peak-1 Gaussian at
 *   the rounded first point (:413-417), strategy GMM the max-normalised sum (:418-426).
 *   strategy FIT the density of the bivariate normal fitted to the UNROUNDED points (mean, ddof = 1 covariance;
 *   scipy.stats.multivariate_normal.pdf, :427-436; not normalised; a singular covariance -- fewer than three
 *   non-collinear points -- gives NaN where scipy raises).  fp64 arithmetic like the reference's numpy, stored as fp32.
 * ---------------------------------------------------------------------- */
#define BF_MAX_TARGET_POINTS 16
enum { BF_TARGETS_FIRST = 0, BF_TARGETS_GMM = 1, BF_TARGETS_FIT = 2 };
/* bf_preprocess_rgb: camera frames raw uint8 [B, S, S, 3] (+ optional 0/1 mask fp32 [B, S, S]) -> network
 *   input fp32 [B, 3, S, S]: mask fill with 77 (bifold/data/processor.py:255-258), scale to [0, 1] and
 *   per-channel normalisation (:87-95; mean3 / std3 are HOST pointers to the 3 values of
 *   conf/processor/default.yaml).  Frames must already have the model's resolution (v2.Resize /
 *   CenterCrop are then the identity); bf_resize_normalize_rgb handles any resolution. */
int bf_preprocess_rgb(const uint8_t* raw, const float* mask, float* out, int batch, int size,
                      const float* mean3, const float* std3, bf_stream_t stream);
/* bf_resize_normalize_rgb: the whole rgb_transform of the reference's Processor (bifold/data/processor.py:45-50,
 *   87-95, 255-258) for frames of ANY resolution: optional mask fill with 77, v2.Resize(size, BICUBIC, antialias) --
 *   the shorter side becomes `size` -- v2.CenterCrop(size), rounding to uint8 as the reference's uint8 image does,
 *   v2.ToDtype(float32, scale=True), v2.Normalize(mean, std), HWC -> CHW.  raw uint8 [B, H, W, 3], mask fp32
 *   [B, H, W] or NULL, tmp_ws >= B * H * size * 3 bytes, out fp32 [B, 3, size, size]; mean3 / std3 are HOST
 *   pointers.  Antialiased bicubic = Keys' cubic (a = -0.5) stretched by the scale factor, columns first, then
 *   rows, uint8 rounding after each pass (PIL's two-pass resample, which torch's uint8 interpolate follows).
 *   torchvision is not in the image: PARITY UNPINNED, checked within 1 LSB against torch's own
 *   F.interpolate(uint8, bicubic, antialias=True).
 * bf_affine_nearest: the image half of Processor._spatial_augmentation (processor.py:333-341):
 *   v2.functional.affine(x, angle, translate=[dx, dy], scale=1, shear=0) with its defaults (nearest neighbour,
 *   zero fill, rotation about the centre).  in / out fp32 [n_images, channels, H, W] (distinct buffers), params
 *   fp32 [n_images, 3] = (angle in degrees, dx, dy) on the DEVICE.  PARITY UNPINNED (torchvision absent): checked
 *   against torch's grid_sample driven by torchvision's documented inverse-matrix construction. */
int bf_resize_normalize_rgb(const uint8_t* raw, const float* mask, uint8_t* tmp_ws, float* out, int batch,
                            int height, int width, int size, const float* mean3, const float* std3,
                            bf_stream_t stream);
int bf_affine_nearest(const float* in, float* out, const float* params, int n_images, int channels,
                      int height, int width, bf_stream_t stream);

/* Depth frames of the sample dict on the device (data/processor.py:59-78, 122-124): DepthScale + MaskDepth --
 * out = (depth + shift[b]) * mask, data/depth_augmentations.py:35-51 (shift [batch] or NULL, mask [batch, pixels] or
 * NULL) -- and TruncatedDepthStandardization (data/depth_augmentations.py:6-17): per frame, with the frame sorted,
 * y = (x - mean(sorted[lo:hi])) / sqrt(var(sorted[lo:hi]) + 1e-6), var unbiased; lo = int(thresh * pixels),
 * hi = int((1 - thresh) * pixels) as the reference computes them (the caller passes the integers).  stats (optional,
 * [batch, 2]) receives the trimmed mean and variance.  x may alias y.  The torchvision Resize / CenterCrop between the
 * two is the identity at the model's own resolution, which is the case this path takes. */
int bf_depth_prepare(const float* depth, const float* mask, const float* shift, float* out, int batch, int pixels,
                     bf_stream_t stream);
int bf_depth_standardize(const float* x, float* y, float* stats, int batch, int pixels, int lo, int hi,
                         bf_stream_t stream);
int bf_heatmap_pick(const float* heat, const float* mask, int32_t* xy, float* conf, int n_maps,
                    int size, bf_stream_t stream);
int bf_gaussian_targets(const float* points, const int32_t* counts, float* heat, int n_maps,
                        int max_points, int size, double sigma, int strategy, bf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BIFOLD_HIP_H */
