/*
 * mtk.h — C ABI of libmtk_hip.so: the MI355X (gfx950) implementation of midst-toolkit's
 * diffusion hot path (GaussianMultinomialDiffusion core + MLPDiffusion denoiser).
 *
 * The reference has no FFI; its seam is two Python nn.Module classes.  Each entry point below
 * names the reference method(s) it replaces.  Paths are relative to
 * src/midst_toolkit/ in the reference checkout:
 *   GMD   = models/clavaddpm/gaussian_multinomial_diffusion.py
 *   DU    = models/clavaddpm/diffusion_utils.py
 *   MODEL = models/clavaddpm/model.py
 *   TF    = attacks/tartan_federer/tartan_federer_attack.py
 *
 * Conventions
 *   - every function returns 0 on success or a negative mtk_status; mtk_last_error() gives the
 *     thread-local message of the last failure;
 *   - pointers are DEVICE pointers unless the name ends in _host;
 *   - nothing synchronises; all work is enqueued on the hipStream_t passed in (a void* here so the
 *     header needs no HIP include; pass torch's current stream handle);
 *   - matrices are described by (pointer, row stride, column stride) in ELEMENTS, so both the
 *     reference's row-major [B, D] tensors (rs = D, cs = 1) and the column-major [D][B] layout the
 *     fused path prefers (rs = 1, cs = B) are accepted;
 *   - a context is bound to one device and may be used from one host thread at a time.
 */
#ifndef MTK_H
#define MTK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtk_ctx mtk_ctx;
typedef void* mtk_stream; /* hipStream_t */

enum mtk_status {
    MTK_OK = 0,
    MTK_ERR_ARG = -1,         /* bad shape / null pointer / unsupported size  (Python: AssertionError) */
    MTK_ERR_UNSUPPORTED = -2, /* unsupported enum / configuration             (Python: ValueError)     */
    MTK_ERR_HIP = -3,         /* a HIP runtime call failed                    (Python: RuntimeError)   */
    MTK_ERR_STATE = -4        /* call order violated (e.g. forward before set_weights)                  */
};

/* Row order of tables_host passed to mtk_ctx_create: MTK_NTAB rows of T doubles each.  Rows 0..9
 * hold the reference's fp32 buffers (exactly representable in a double); rows 10 and 11 are the fp64
 * log-variance and variance rows of the reverse chain, which the reference never casts to fp32
 * (GMD:223, 387-394). */
enum mtk_table {
    MTK_TAB_SQRT_ACP = 0,       /* sqrt_alphas_cumprod            GMD:263 */
    MTK_TAB_SQRT_1M_ACP = 1,    /* sqrt_one_minus_alphas_cumprod  GMD:264 */
    MTK_TAB_LOG_ALPHA = 2,      /* log_alpha                      GMD:254 */
    MTK_TAB_LOG_1M_ALPHA = 3,   /* log_1_min_alpha                GMD:257 */
    MTK_TAB_LOG_ACP = 4,        /* log_cumprod_alpha              GMD:255 */
    MTK_TAB_LOG_1M_ACP = 5,     /* log_1_min_cumprod_alpha        GMD:258 */
    MTK_TAB_SQRT_RECIP_ACP = 6, /* sqrt_recip_alphas_cumprod      GMD:265 */
    MTK_TAB_SQRT_RECIPM1_ACP = 7, /* sqrt_recipm1_alphas_cumprod  GMD:266 */
    MTK_TAB_POST_COEF1 = 8,     /* posterior_mean_coef1           GMD:226 */
    MTK_TAB_POST_COEF2 = 9,     /* posterior_mean_coef2           GMD:228-230 */
    MTK_TAB_MODEL_LOGVAR64 = 10, /* log([posterior_variance[1], (1-alphas)[1:]]) in fp64, GMD:387-394 */
    MTK_TAB_MODEL_VAR64 = 11,   /* [posterior_variance[1], (1-alphas)[1:]] in fp64 (the "variance" of GMD:425, used by condition_mean GMD:615) */
    MTK_TAB_ACP = 12,           /* alphas_cumprod                 GMD:260 */
    MTK_TAB_ACP_PREV = 13,      /* alphas_cumprod_prev            GMD:261 (DDIM, GMD:1106-1107, 1240-1241) */
    MTK_TAB_POST_LOGVAR_CLIPPED = 14, /* posterior_log_variance_clipped (fp32 values) GMD:224-225 (KL Gaussian loss, GMD:459) */
    MTK_NTAB = 15
};

/* Denoiser arithmetic. */
enum mtk_precision {
    MTK_PREC_FP32 = 0, /* f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 FMA chains */
    MTK_PREC_BF16 = 1, /* bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16): one fused kernel for dim_t = 128,
                          4 x 512 hidden, D <= 128; layer-by-layer GEMMs with bf16 activations for every other shape */
    MTK_PREC_BF16X3 = 2 /* split-bf16: every fp32 weight and activation is carried as hi + lo bf16 numbers and every product
                          as three bf16 MFMAs (hi.hi + lo.hi + hi.lo) with fp32 accumulation, layer by layer: meets the fp32
                          parity bar (<= 1e-4 relative; ~1e-5 measured) at several times the rate of MTK_PREC_FP32 */
};

/* Arithmetic of the diffusion kernels (mtk_q_sample, mtk_loss_rows, mtk_p_sample_step, mtk_sample_init). */
enum mtk_math {
    MTK_MATH_REFERENCE = 0, /* accurate expf/logf, every fp32 op rounded as the reference's eager ops, the block
                               log-sum-exp restated as torch-CPU computes it (fp64 cumulative scan + fp32 log-sub-exp,
                               DU:250-289): the parity mode, default */
    MTK_MATH_FAST = 1       /* hardware exp/log, exact per-block log-sum-exp, closed forms for the true posterior and
                               the prior: same formulas without the reference's fp32 cancellation noise; meant for
                               runs with in-kernel Philox noise (no reference random stream to match) */
};

const char* mtk_last_error(void);
int mtk_version(void);

/* Replaces GaussianMultinomialDiffusion.__init__ table set-up (GMD:156-239) on the device side.
 * K_host[C] are the category counts (C = 0 or K_host[0] == 0 means "no categorical block",
 * the shipped pipeline's num_classes=[0]); logK_host[C] = float32 log(K) as the caller's torch
 * computes it (GMD:739); tables_host is MTK_NTAB x T doubles (see enum mtk_table). */
int mtk_ctx_create(int device, int T, int Nn, int C, const int32_t* K_host, const float* logK_host,
                   const double* tables_host, mtk_ctx** out);
void mtk_ctx_destroy(mtk_ctx* ctx);
/* No reference counterpart: selects enum mtk_math for this context's diffusion kernels. */
int mtk_ctx_set_math(mtk_ctx* ctx, int mode);
/* The constructor arguments gaussian_parametrization / parametrization of GaussianMultinomialDiffusion (GMD:164-165, 199-201).
 * gaussian_x0 != 0 = GaussianParametrization.X0: the denoiser's numeric output is the predicted x_0 itself (GMD:404-405) - read
 * by the reverse steps (ancestral, DDIM, reverse DDIM), the KL Gaussian loss and its gradient; the MSE loss compares the output
 * with the noise either way (GMD:531-532).  multinomial_direct != 0 = Parametrization.DIRECT: the per-block log-softmax of the
 * categorical output is log p(x_{t-1} | x_t) itself, no posterior (GMD:828-829) - read by the reverse step, the multinomial
 * loss and its gradient.  Defaults 0 / 0 = EPS / X0, the reference's. */
int mtk_ctx_set_parametrization(mtk_ctx* ctx, int gaussian_x0, int multinomial_direct);

/* No reference counterpart (capture-ready contract): sizes every workspace of the context for calls of up to
 * max_rows rows through the denoiser (B for mtk_mlp_forward and the training entry points, B * P for mtk_tf_score).  Afterwards
 * those entry points neither allocate nor synchronise for such calls, so they can be captured into a HIP graph.  The call
 * itself may synchronise `stream` and allocate; the reservation only grows and survives mtk_mlp_set_weights. */
int mtk_ctx_reserve(mtk_ctx* ctx, int64_t max_rows, mtk_stream stream);
/* No reference counterpart: the reverse-chain kernels (mtk_sample_init, mtk_p_sample_step[_guided], mtk_ddim_step) take their
 * Philox seed and first global row from the device pair {seed, row0} at seed_row0_dev instead of from their arguments (NULL:
 * back to the arguments) - a captured chain is replayed with new noise by rewriting 16 bytes. */
int mtk_ctx_set_seed_source(mtk_ctx* ctx, const uint64_t* seed_row0_dev);
/* No reference counterpart: a counter that moves whenever the context frees or (re)allocates a device buffer its kernels are
 * handed - an activation workspace growing for a larger call (or through mtk_ctx_reserve), the weight images being rebuilt by
 * mtk_mlp_set_weights.  A HIP graph captured from this context's entry points holds raw addresses of those buffers: its owner
 * records the value at capture time and re-captures (never replays) once it differs.  -1 for a null context. */
int64_t mtk_ctx_workspace_generation(mtk_ctx* ctx);
/* No reference counterpart: how many hipMalloc calls the context's weight path (mtk_mlp_set_weights and the image packers it
 * runs) has made so far.  A second mtk_mlp_set_weights with unchanged layer shapes rewrites the images in place - it takes
 * its blocks back from the context's pool, allocates nothing, frees nothing and does not synchronise the stream - so the
 * count does not move across it (tests/test_gpu_parity.py checks exactly that).  -1 for a null context. */
int64_t mtk_ctx_alloc_count(mtk_ctx* ctx);

/* MODEL:583-585, 610-612 (IsTargetConditioned.EMBEDDING with num_classes > 0): the denoiser adds
 * silu(label_embedding.weight[y]) to the time embedding of every record.  Here a (label, timestep) pair is ONE row of the
 * time-embedding table: create the context with num_timesteps = (n_labels + 1) * period table rows, hand over the raw
 * [n_labels][dim] fp32 embedding weight (device pointer, copied) BEFORE mtk_mlp_set_weights, and pass row index
 * (1 + y) * period + t wherever mtk_mlp_forward takes a timestep (rows below `period`: y = None, MODEL:610).  n_labels = 0
 * clears.  Such a context serves the denoiser forward only: mtk_train_attach refuses, and its schedule tables are not those of
 * a diffusion object (the Python drop-in keeps a separate context for the diffusion kernels). */
int mtk_mlp_set_label_embedding(mtk_ctx* ctx, const float* weight_dev, int n_labels, int dim, int period, mtk_stream stream);

/* Replaces MLPDiffusion's parameter storage (MODEL:548-594).  W[i] / b[i] are device pointers to
 * fp32 PyTorch-layout weights [out_i][in_i] / [out_i], in this order:
 *   0 proj, 1 timestep_embedding.0, 2 timestep_embedding.2, 3.. mlp.blocks.*.linear, last mlp.head
 * dims_host = {D, dim_t, h_1 .. h_n} (n = n_hidden).  The call packs the weights into MFMA operand
 * order (fp32 and bf16 images) and rebuilds the [T, dim_t] time-embedding table
 * emb[t] = Linear(SiLU(Linear(sinusoid(t)))) (MODEL:123-144, 608).  freqs_host[dim_t/2] are the
 * sinusoid frequencies as the caller's torch computes them (MODEL:136).  Call again after every
 * optimizer step that changes the weights. */
int mtk_mlp_set_weights(mtk_ctx* ctx, int n_hidden, const int32_t* dims_host, const float* const* W_host_ptrs,
                        const float* const* b_host_ptrs, const float* freqs_host, mtk_stream stream);

/* Replaces MLPDiffusion.forward(x, timesteps) (MODEL:596-615) with is_target_conditioned = NONE.
 * x: [B, D] via (x_rs, x_cs); t: per-row int64 timesteps, or NULL to use t_scalar for every row;
 * out: [B, D] via (o_rs, o_cs). */
int mtk_mlp_forward(mtk_ctx* ctx, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t,
                    int t_scalar, float* out, int64_t o_rs, int64_t o_cs, int precision, mtk_stream stream);

/* Replaces the denoiser call of the reverse loop, GMD:1406: `self._denoise_fn(torch.cat([z_norm, log_z], dim=1).float(), t, ...)`.
 * The reference materialises that input every step; here the denoiser reads the chain's state itself - z [Nn][state_ld] fp64 and
 * z_idx [C][state_ld] int32 (column-major, as mtk_sample_init / mtk_p_sample_step keep them) - and forms float(z) and the
 * log-one-hot rows on the fly, so the step kernels can be called with x_in_next = NULL and write no input matrix at all
 * (480 of their 1,360 bytes per row at 20 numeric + 10 x 10 categorical columns).  One timestep t for all B rows; out as in
 * mtk_mlp_forward.  Same arithmetic and rounding as mtk_mlp_forward on the materialised input: same output bits.
 * mtk_mlp_state_input_ok says (1 / 0) whether the context's weights and `precision` have this path (today: MTK_PREC_BF16 on the
 * layer-by-layer kernels, D <= 128, Nn <= 32, C <= 16); without it mtk_mlp_forward_state returns MTK_ERR_UNSUPPORTED. */
int mtk_mlp_state_input_ok(mtk_ctx* ctx, int precision);
int mtk_mlp_forward_state(mtk_ctx* ctx, int64_t B, const double* z, const int32_t* z_idx, int64_t state_ld, int t_scalar, float* out,
                          int64_t o_rs, int64_t o_cs, int precision, mtk_stream stream);

/* No counterpart in the reference: mtk_mlp_forward(MTK_PREC_FP32) for callers that cannot allocate device memory -
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and results as
 * mtk_mlp_forward with MTK_PREC_FP32, but the activation ping-pong lives in a file-scope device array of the library
 * (2 x MTK_NOALLOC_MAX_ROWS x MTK_NOALLOC_MAX_WIDTH floats = 32 MiB, held by every GPU the library runs on until the
 * process ends) instead of the context's hipMalloc'ed workspace; the call allocates, frees and synchronises nothing.
 * B > MTK_NOALLOC_MAX_ROWS or a hidden layer wider than MTK_NOALLOC_MAX_WIDTH returns MTK_ERR_UNSUPPORTED.  All calls on
 * one GPU share that GPU's array: calls that overlap on one GPU (two streams, two contexts) race. */
#define MTK_NOALLOC_MAX_ROWS 4096
#define MTK_NOALLOC_MAX_WIDTH 1024
int mtk_mlp_forward_noalloc(mtk_ctx* ctx, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t,
                            int t_scalar, float* out, int64_t o_rs, int64_t o_cs, mtk_stream stream);

/* Replaces the noising half of mixed_loss (GMD:1025-1040): gaussian_q_sample (GMD:302),
 * index_to_log_onehot (DU:209), q_pred (GMD:742), log_sample_categorical (GMD:852), q_sample (GMD:873).
 * x0: [B, Nn + C] row-major with leading dimension x0_ld (numeric columns, then category codes as
 * floats).  eps: [B, Nn] row-major (ld = Nn) or NULL → Philox normal(seed, row, col);  u: [B, S]
 * row-major (ld = S) or NULL → Philox uniform.  Outputs: x_in = denoiser input [B, D]
 * (numeric x_t, then the log-one-hot of the sampled categories) via (xi_rs, xi_cs);
 * xt_idx: [C][B] int32 sampled category codes (column-major).  row0 is the global index of row 0
 * (Philox counters use row0 + i so that results do not depend on how rows are sharded). */
int mtk_q_sample(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                 const float* u, float* x_in, int64_t xi_rs, int64_t xi_cs, int32_t* xt_idx, uint64_t seed,
                 uint64_t row0, mtk_stream stream);

/* Replaces the loss half of mixed_loss (GMD:1044-1066): _gaussian_loss MSE branch (GMD:505-532),
 * predict_start (GMD:761), q_posterior (GMD:781) with sliced_logsumexp (DU:268) restated as the
 * fp64 running log-cum-sum-exp + fp32 log-sub-exp it is on torch-CPU, multinomial_kl (GMD:709),
 * log_categorical (DU:195), kl_prior (GMD:887), compute_lt (GMD:907), _multinomial_loss (GMD:984).
 * Per-record outputs (the reference only returns their means): loss_multi[B] already divided by C,
 * loss_gauss[B].  Either may be NULL when that side is absent.  eps NULL → regenerate the Philox
 * draws of mtk_q_sample(seed, row0). */
int mtk_loss_rows(mtk_ctx* ctx, int64_t B, const float* model_out, int64_t mo_rs, int64_t mo_cs, const float* x0,
                  int64_t x0_ld, const int32_t* xt_idx, const float* eps, const int64_t* t, float* loss_multi,
                  float* loss_gauss, uint64_t seed, uint64_t row0, mtk_stream stream);

/* Replaces tartan_federer_attack.mixed_loss (TF:40-95) for n_t timesteps at once:
 * score[k][r][p] = mean_d (noise[p][d] - eps_hat(x_t(r, p, t_k + add_t), t_k)[d])^2.
 * x: [B, >=Nn] row-major (ld x_ld), noise: [P, Nn] row-major, t_list_host: n_t ints,
 * score: [n_t][B][P] fp32.  Requires S == 0 (TF feeds only the numeric block, TF:86-91). */
int mtk_tf_score(mtk_ctx* ctx, int64_t B, int P, const float* x, int64_t x_ld, const float* noise, int n_t,
                 const int32_t* t_list_host, int add_t, float* score, int precision, mtk_stream stream);

/* Replaces one iteration of the reverse loop of sample() (GMD:1403-1417) after the denoiser call:
 * gaussian_p_sample (GMD:663) = gaussian_p_mean_variance (GMD:358) + _predict_features_start_from_eps
 * (GMD:545) + gaussian_q_posterior_mean_variance (GMD:323), and p_sample (GMD:837) = p_pred (GMD:810)
 * + log_sample_categorical (GMD:852).
 * z: [Nn][B] fp64 state, updated in place (first_step != 0: the state holds fp32-representable
 * values and the mean is formed in fp32 arithmetic, as torch's promotion does at t = T-1);
 * z_idx: [C][B] int32, updated in place.  noise: [B, Nn] row-major, fp32 if noise_is_f64 == 0 else
 * fp64, or NULL → Philox;  u: [B, S] row-major or NULL → Philox.  x_in_next: the next denoiser
 * input [B, D] (float(z'), log-one-hot(z_idx')) via (xi_rs, xi_cs), may be NULL. */
int mtk_p_sample_step(mtk_ctx* ctx, int64_t B, int t, int first_step, const float* model_out, int64_t mo_rs,
                      int64_t mo_cs, double* z, int32_t* z_idx, const void* noise, int noise_is_f64, const float* u,
                      float* x_in_next, int64_t xi_rs, int64_t xi_cs, uint64_t seed, uint64_t row0,
                      mtk_stream stream);

/* The same step with classifier guidance: replaces gaussian_p_sample's conditioning_function branch (GMD:699-702)
 * = condition_mean (GMD:585-615): mean <- float(mean) + variance * grad, where grad [B, Nn] row-major fp32 is what the
 * caller's conditioning_function(features, timestep) returned for the CURRENT state z (the callback itself - a torch
 * classifier and its autograd - stays on the caller's side) and variance is the fp64 model variance of t.  grad may be
 * NULL (then identical to mtk_p_sample_step).  Used by sample(..., conditioning_function) and conditional_sample
 * (GMD:1319). */
int mtk_p_sample_step_guided(mtk_ctx* ctx, int64_t B, int t, int first_step, const float* model_out, int64_t mo_rs,
                             int64_t mo_cs, double* z, int32_t* z_idx, const void* noise, int noise_is_f64, const float* u,
                             const float* grad, float* x_in_next, int64_t xi_rs, int64_t xi_cs, uint64_t seed, uint64_t row0,
                             mtk_stream stream);

/* Replaces one iteration of sample_ddim's loop (GMD:1292-1308) after the denoiser call: gaussian_ddim_step (GMD:1070-1118)
 * and multinomial_ddim_step (GMD:1211-1257) with eta = 0, the only value sample_ddim passes (no noise term; the
 * categorical mixture keeps the x0 and the uniform branch).  Same state buffers as mtk_p_sample_step; the numeric state
 * stays fp32-representable (the reference's DDIM chain is fp32 throughout).  u: [B, S] row-major or NULL -> Philox. */
int mtk_ddim_step(mtk_ctx* ctx, int64_t B, int t, const float* model_out, int64_t mo_rs, int64_t mo_cs, double* z,
                  int32_t* z_idx, const float* u, float* x_in_next, int64_t xi_rs, int64_t xi_cs, uint64_t seed, uint64_t row0,
                  mtk_stream stream);
/* The same step for any eta >= 0 and with score-based guidance: gaussian_ddim_step's sigma / noise term (GMD:1109-1116;
 * noise [B, Nn] row-major fp32 as torch.randn_like(features) returned it, or NULL -> Philox), multinomial_ddim_step's
 * "stay" branch coef1 = sigma (GMD:1241-1256), and condition_score (GMD:617-661, called at GMD:1104-1105): grad [B, Nn]
 * row-major fp32 = conditioning_function(features, timestep) for the CURRENT state, or NULL. */
int mtk_ddim_step_ex(mtk_ctx* ctx, int64_t B, int t, double eta, const float* model_out, int64_t mo_rs, int64_t mo_cs, double* z,
                     int32_t* z_idx, const float* noise, const float* u, const float* grad, float* x_in_next, int64_t xi_rs,
                     int64_t xi_cs, uint64_t seed, uint64_t row0, mtk_stream stream);
/* Replaces one iteration of gaussian_ddim_reverse_sample's loop (GMD:1209-1213) after the denoiser call:
 * gaussian_ddim_reverse_step (GMD:1160-1186), numeric features only, on the same fp64 state buffer z [Nn][B]. */
int mtk_ddim_reverse_step(mtk_ctx* ctx, int64_t B, int t, const float* model_out, int64_t mo_rs, int64_t mo_cs, double* z,
                          float* x_in_next, int64_t xi_rs, int64_t xi_cs, mtk_stream stream);

/* Replaces the initial draw of sample() (GMD:1392-1399): z0 [B, Nn] row-major fp32 (or NULL → Philox)
 * into the fp64 state z [Nn][B]; categories by Gumbel-argmax of uniform logits from u0 [B, S]
 * (or NULL → Philox) into z_idx [C][B]; and the first denoiser input (x_in may be NULL when the denoiser reads the state:
 * mtk_mlp_forward_state). */
int mtk_sample_init(mtk_ctx* ctx, int64_t B, const float* z0, const float* u0, double* z, int32_t* z_idx,
                    float* x_in, int64_t xi_rs, int64_t xi_cs, uint64_t seed, uint64_t row0, mtk_stream stream);

/* Replaces the decode at the end of sample() (GMD:1419-1423; one_hot_encoding_to_categories, DU:121):
 * sample[B, Nn + C] fp64 row-major = (z, category codes). */
int mtk_sample_decode(mtk_ctx* ctx, int64_t B, const double* z, const int32_t* z_idx, double* sample,
                      mtk_stream stream);

/* Replaces dataset.numerical_transform.inverse_transform of the synthesiser (models/clavaddpm/synthesizer.py:251-277)
 * for the quantile normaliser the pipeline fits (common/dataset_transformations.py:57-66): scikit-learn 1.7.2
 * QuantileTransformer._transform_col(inverse=True) per column - Phi(x) when output_normal != 0, numpy-style linear
 * interpolation of quantiles [n_quantiles][n_columns] (row-major, the fitted quantiles_) over references
 * [n_quantiles] (references_), and the BOUNDS_THRESHOLD overrides.  x, out: [B, n_columns] fp64 row-major with
 * leading dimensions x_ld / out_ld (out may alias x); NaN stays NaN.  Needs no context: device is the HIP ordinal. */
int mtk_quantile_inverse(int device, int64_t B, int n_columns, const double* x, int64_t x_ld, const double* quantiles,
                         const double* references, int n_quantiles, int output_normal, double* out, int64_t out_ld,
                         mtk_stream stream);

/* The same seam for the other two normalisers the reference can fit (common/dataset_transformations.py:66-69), in
 * scikit-learn's operation order: kind 0 = StandardScaler.inverse_transform, out = x * p0 + p1 (p0 = scale_, p1 = mean_);
 * kind 1 = MinMaxScaler.inverse_transform, out = (x - p0) / p1 (p0 = min_, p1 = scale_).  p0, p1: [n_columns] fp64. */
int mtk_affine_inverse(int device, int64_t B, int n_columns, const double* x, int64_t x_ld, int kind, const double* p0,
                       const double* p1, double* out, int64_t out_ld, mtk_stream stream);

/* Replaces _gaussian_loss with GaussianLossType.KL (GMD:534-541 -> _vb_terms_bpd GMD:431-484; normal_kl DU:9-45,
 * discretized_gaussian_log_likelihood DU:62-91) for the EPS parametrisation: per record, in bits per dimension, the
 * decoder NLL where t == 0 and KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) elsewhere.  model_out / x_t: the numeric block
 * of the denoiser output / input, element (row, j) at [row * rs + j * cs]; x0: [B, x0_ld] row-major (numeric columns
 * first); t [B] int64; loss [B] fp64 (the reference's result is fp64: its model log-variance row is). */
int mtk_gauss_kl_rows(mtk_ctx* ctx, int64_t B, const float* model_out, int64_t mo_rs, int64_t mo_cs, const float* x0,
                      int64_t x0_ld, const float* x_t, int64_t xt_rs, int64_t xt_cs, const int64_t* t, double* loss,
                      mtk_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Training iteration.  Replaces ClavaDDPMTrainer._train_step /
 * _anneal_learning_rate / update_ema (models/clavaddpm/trainer.py:58-94, 138-155).
 *
 * The five parameter images — 0 weights, 1 gradients, 2 Adam exp_avg, 3 Adam exp_avg_sq, 4 EMA — share the
 * padded layout of the forward GEMMs and sit back to back in ONE caller-owned device buffer `state` of
 * 5 * mtk_train_image_elems(ctx) floats, so that a multi-GPU host all-reduces image 1
 * (state + n .. state + 2n) between mtk_train_step and mtk_train_apply (one exchange per iteration).
 * ------------------------------------------------------------------------------------------- */
/* Floats in one image for the weights set by mtk_mlp_set_weights (0 when none are set). */
int64_t mtk_train_image_elems(mtk_ctx* ctx);
/* TRAINER:39-51: copies the current weights into image 0 and image 4 (the EMA starts as a deepcopy), zeroes
 * images 1-3 (fresh AdamW state) and makes the context's fp32 denoiser read its weights from image 0.
 * freqs_host as in mtk_mlp_set_weights.  mtk_mlp_set_weights detaches. */
int mtk_train_attach(mtk_ctx* ctx, float* state, const float* freqs_host, mtk_stream stream);
/* TRAINER:88-91 (zero_grad, mixed_loss GMD:984-1068, backward): fills image 1 with
 * d[(sum_rows loss_multi + loss_gauss) * inv_global_batch] / d weights for the B rows given and writes the
 * per-record losses (either pointer may be NULL).  t / eps / u / seed / row0 as in mtk_q_sample.
 * inv_global_batch = 1 / (rows over all ranks): after a SUM all-reduce image 1 is the gradient of the
 * global batch mean, which is what the reference's loss.backward() produces on one device. */
int mtk_train_step(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                   const float* u, double inv_global_batch, float* loss_multi, float* loss_gauss, uint64_t seed,
                   uint64_t row0, mtk_stream stream);
/* The two remaining per-iteration device launches a host framework would add around mtk_train_step, folded in (no reference
 * counterpart as entry points; the operations are TRAINER:88 -> GMD:1011 sample_time(UNIFORM) and GMD:1066-1068 the batch means):
 *  - mtk_train_step / mtk_train_forward with t = NULL draw the timesteps in the noising kernel, uniform over [0, T)
 *    (Philox-4x32-10 stream 6, word 0 of record row0 + row, scaled by T >> 32: like the in-kernel noise, not torch's stream);
 *    the backward phase of such a forward takes t = NULL too, and mtk_train_timesteps copies the B drawn values out (int64,
 *    device) - what a loss-aware sampler's history update (GMD:1062) reads.
 *  - mtk_train_set_loss_means(ctx, means): every later forward phase that is given loss_multi / loss_gauss also writes their
 *    batch means to means[0] / means[1] (device floats, caller-owned, 0 for an absent vector; fp64 sums added in a fixed order,
 *    one workgroup on the iteration's side stream).  NULL (default) switches it off. */
int mtk_train_set_loss_means(mtk_ctx* ctx, float* means_dev);
int mtk_train_timesteps(mtk_ctx* ctx, int64_t B, int64_t* t_dev, mtk_stream stream);
/* No reference counterpart (an execution detail of TRAINER:88-94 on one device): with deferred reduction on, mtk_train_step
 * (deterministic mode, no bucket events) leaves the split-batch partial sums of the weight / bias gradients in its slabs and
 * mtk_train_apply's optimizer pass adds them - in the same order, so the same bits - as it reads the gradient, writing the sums
 * to image 1 on the way: one launch and one pass over the gradient image fewer per iteration.  Between the two calls image 1
 * is incomplete; mtk_train_export(image 1) and switching the mode off complete it first.  Not for multi-GPU runs, whose
 * all-reduce reads image 1 between the two calls.  Off by default. */
int mtk_train_set_deferred_reduce(mtk_ctx* ctx, int on, mtk_stream stream);
/* No reference counterpart beyond the reference pipeline's torch.use_deterministic_algorithms (common/random.py:11-55): on
 * (default) = the split-batch reductions of the weight gradients are added in a fixed order (partial tiles in slabs) and the
 * time-embedding scatter accumulates in 64-bit fixed point, so gradients are bit-identical run to run; off = fp32 hardware
 * atomics (order varies). */
int mtk_train_set_deterministic(mtk_ctx* ctx, int on);
/* MODEL:162-180 (MLP.Block.forward = dropout(activation(linear(x))), nn.Dropout in training mode): the iteration drops every
 * hidden activation with probability p and scales the kept ones by 1 / (1 - p).  The mask of element (row0 + row, column n) of
 * hidden layer l is a pure function of (seed, l, row, n) (Philox-4x32-10, stream 5), so it does not depend on how a batch is
 * split over ranks or calls; the backward needs no second draw (a saved activation is > 0 exactly where the unit was positive
 * and kept).  torch's own Dropout stream cannot be reproduced bit for bit on any other device; the contract matched is the
 * distribution and the gradient given the mask (tests feed mtk_train_dropout_mask's masks to the oracle's autograd).  p = 0
 * (default) switches it off; the caller passes a fresh seed per iteration. */
int mtk_train_set_dropout(mtk_ctx* ctx, double p, uint64_t seed);
/* MODEL:583-585, 610-612 (IsTargetConditioned.EMBEDDING with num_classes > 0) under training: the denoiser adds
 * silu(label_embedding.weight[y]) to every record's time embedding, and label_embedding.weight is trained.  Call
 * mtk_train_set_label_classes(n) after mtk_mlp_set_weights and before mtk_train_image_elems / mtk_train_attach: the training image
 * gains a layer [n x dim_t] (W = the embedding table, its bias slot unused) placed after the time-embedding MLP - so the order of
 * the tensors of mtk_train_export / mtk_train_import becomes proj, temb.0, temb.2, label_embedding, blocks..., head - zero until
 * imported.  mtk_train_set_labels hands over the int64 labels [B] (device memory, caller-owned, read by the next mtk_train_step /
 * mtk_train_forward / mtk_train_backward); labels outside [0, n) are clamped.  The table's gradient is the scatter of
 * d act[0] over the labels times silu' (64-bit fixed point, order-independent).  Inference with such a network keeps using
 * mtk_mlp_set_label_embedding's table rows.  n = 0 (and every mtk_mlp_set_weights) removes the layer. */
int mtk_train_set_label_classes(mtk_ctx* ctx, int n_labels);
int mtk_train_set_labels(mtk_ctx* ctx, const int64_t* y_dev);
/* The keep mask (1 = kept) hidden layer `layer` uses for rows [row0, row0 + B) under the current mtk_train_set_dropout setting:
 * uint8 [B][width of the layer], row-major. */
int mtk_train_dropout_mask(mtk_ctx* ctx, int layer, int64_t B, uint64_t row0, uint8_t* keep, mtk_stream stream);
/* No reference counterpart (the reference trains in fp32; north_star part 1 "bf16 / fp32 accumulate"): MTK_PREC_BF16 makes every
 * GEMM of the training iteration (forward, data gradients, weight gradients) round its operands to bf16 and multiply on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the master weights, the gradient image, AdamW and EMA stay fp32.
 * MTK_PREC_FP32 (default) is the parity path.  Stated tolerance: tests/test_gpu_train.py. */
int mtk_train_set_precision(mtk_ctx* ctx, int precision);
/* Which Gaussian loss the training iteration evaluates and differentiates: 0 = MSE (GMD:506-532, default), 1 = the KL / decoder-NLL
 * term of the variational bound in bits per dimension (GaussianLossType.KL: GMD:431-484, 534-541; DU:9-91).  loss_gauss then holds
 * that term per record (fp32 here; mtk_gauss_kl_rows gives it in fp64) and the backward differentiates it, the clamp at 1e-12 and
 * the three bins of the discretised likelihood included. */
int mtk_train_set_gaussian_loss(mtk_ctx* ctx, int kl);
/* No reference counterpart (the reference trains on one device; north_star part 1 / SURVEY 2c "bucketed per layer, overlapped"):
 * data-parallel overlap of the gradient exchange with the backward.  The gradient image falls into buckets in the order the
 * backward completes them - head, hidden layers last to first, proj, time-embedding MLP.  mtk_train_bucket_layout returns their
 * number and, when off / len are given, each bucket's element offset and length inside one image.  mtk_train_set_bucket_events
 * hands the library one hipEvent_t per bucket (n = 0 clears): mtk_train_step / mtk_train_backward then record event i on their
 * stream as soon as bucket i of the gradient image is final, so the caller can all-reduce that slice on another stream
 * (RCCL) while the rest of the backward runs.  Values are those of the unbucketed call, bit for bit. */
int mtk_train_bucket_layout(mtk_ctx* ctx, int max_buckets, int64_t* off, int64_t* len);
int mtk_train_set_bucket_events(mtk_ctx* ctx, int n, void* const* events);
/* The two halves of mtk_train_step as separate calls: the autograd seam of the drop-in's mixed_loss (TRAINER:89-91 calls
 * mixed_loss and then .backward() on the sum of its two scalars; attacks/ensemble/clavaddpm_fine_tuning.py does the same).
 * mtk_train_forward = noising + denoiser forward with every activation kept + per-record losses.  mtk_train_backward fills
 * image 1 with d[w_multi * sum_rows loss_multi + w_gauss * sum_rows loss_gauss] / d weights from the activations the LAST
 * mtk_train_forward on this context left behind; x0 / t / eps / seed / row0 must be that call's (B is checked).  With
 * w_multi = w_gauss = 1 / B this is the gradient of (mean loss_multi + mean loss_gauss), what loss.backward() produces. */
int mtk_train_forward(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps, const float* u,
                      float* loss_multi, float* loss_gauss, uint64_t seed, uint64_t row0, mtk_stream stream);
int mtk_train_backward(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps, double w_multi,
                       double w_gauss, uint64_t seed, uint64_t row0, mtk_stream stream);
/* The same backward with the two weights read on the device: w_dev = {w_multi, w_gauss} (2 floats).  What an autograd seam has
 * in hand are the incoming gradients of the two scalars as device tensors (TRAINER:90: loss.backward() hands 1 to each); reading
 * them on the host would synchronise the stream once per iteration. */
int mtk_train_backward_dev(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                           const float* w_dev, uint64_t seed, uint64_t row0, mtk_stream stream);
/* TRAINER:92 + :133: torch.optim.AdamW.step() (decoupled decay, bias correction, step is 1-based) on images
 * 0/2/3 from image 1, then EMA image 4 <- ema_rate * image 4 + (1 - ema_rate) * image 0, then the
 * time-embedding table is rebuilt from the new weights.  The caller owns the learning-rate anneal
 * (TRAINER:58-68: lr0 * (1 - step / steps), applied AFTER the step). */
int mtk_train_apply(mtk_ctx* ctx, double lr, double weight_decay, double beta1, double beta2, double eps,
                    int64_t step, double ema_rate, mtk_stream stream);
/* Copies image `image` to / from PyTorch-layout tensors (order as in mtk_mlp_set_weights): how the host
 * reads back the trained weights / the EMA model, or restores optimizer state. */
int mtk_train_export(mtk_ctx* ctx, int image, float* const* W, float* const* b, mtk_stream stream);
int mtk_train_import(mtk_ctx* ctx, int image, const float* const* W, const float* const* b, mtk_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Classifier guidance.  Replaces the eval-mode forward of `Classifier`
 * (models/clavaddpm/model.py:45-120) and the input gradient its conditioning function takes through autograd
 * (models/clavaddpm/synthesizer.py:429-451): grad = scale * d log_softmax(classifier(x, t))[y] / d x, what
 * condition_mean (GMD:585-615) adds to the posterior mean through mtk_p_sample_step_guided.
 *
 * Network handed over: h0 = proj(x) + emb[t]; then n_lin Linear layers, layer i < n_lin - 1 followed by ReLU
 * (slopes_host[i] == 0) or LeakyReLU (its negative slope).  widths_host[n_lin] = output widths of those layers (the last
 * is the number of classes).  W / b (device, fp32, PyTorch layout [out][in] / [out]) in the order proj,
 * timestep_embedding.0, timestep_embedding.2, then the n_lin layers - with each eval-mode BatchNorm1d (an affine map
 * per feature) folded into the Linear that FOLLOWS it by the caller; Dropout is the identity in eval mode; the
 * nn.Transformer the reference constructs (MODEL:73) is never called by its forward.  freqs_host as in
 * mtk_mlp_set_weights.  The [T, dim_t] time-embedding table is built by the call.
 * ------------------------------------------------------------------------------------------- */
typedef struct mtk_clf mtk_clf;
int mtk_clf_create(int device, int T, int D_in, int dim_t, int n_lin, const int32_t* widths_host, const float* slopes_host,
                   const float* const* W, const float* const* b, const float* freqs_host, mtk_stream stream, mtk_clf** out);
void mtk_clf_destroy(mtk_clf* clf);
/* MTK_PREC_FP32 (default: fp32 MFMA, the reference's arithmetic for SYNTH:429-451) or MTK_PREC_BF16X3: every GEMM of the guidance as a
 * split-bf16 product (operands carried as hi + lo bf16 numbers, three bf16 MFMAs per product, fp32 accumulation and storage) -
 * logits and input gradients within ~1e-3 of their scale at several times the fp32 rate.  Plain bf16 operands are refused: the
 * input gradient through the classifier's layers is a sum with heavy cancellation.  No counterpart in the reference. */
int mtk_clf_set_precision(mtk_clf* clf, int precision);
/* x: [B, x_ld] row-major fp32 (x_ld >= D_in); t: [B] int64; y: [B] int64 class indices (needed for grad).  logits: [B, n_out]
 * row-major or NULL; grad: [B, D_in] row-major or NULL.  Rows whose timestep lies outside [0, T) come back NaN. */
int mtk_clf_guidance(mtk_clf* clf, int64_t B, const float* x, int64_t x_ld, const int64_t* t, const int64_t* y, double scale,
                     float* logits, float* grad, mtk_stream stream);

/* ---- The guidance classifier in TRAINING mode (models/clavaddpm/model.py:45-120 with the module in train(), as
 * models/clavaddpm/train.py:310-430 and :560-627 drive it: logits = classifier(x_t, timesteps); cross_entropy(...).backward()).
 * Per hidden stage: Linear -> ReLU (first stage) / LeakyReLU -> BatchNorm1d on BATCH statistics -> Dropout; then the output Linear.
 * mtk_clft_forward returns the logits and every stage's batch mean / biased batch variance (the caller updates its running
 * statistics as torch does: momentum 0.1, unbiased variance); mtk_clft_backward turns d objective / d logits into the gradients of
 * all Linear weights / biases and BatchNorm gamma / beta - the batch-norm backward goes through the batch statistics
 * (d a = gamma istd / B (B dy - sum dy - xhat sum dy xhat)).  Dropout masks are Philox (counter = row, stage, column; key = seed), as
 * in mtk_train_set_dropout.  The time embedding is evaluated per record, so its layers get plain weight gradients.  fp32 MFMA
 * GEMMs; one reduction chunk per GEMM and fixed-order column sums: identical run to run.  The object owns padded copies of the
 * parameters (mtk_clft_set_params, in place) and the activations of ONE forward (mtk_clft_backward's B must match).
 * widths_host: n_hidden hidden widths, then n_out; slopes_host [n_hidden]: 0 for ReLU, 0.01 for nn.LeakyReLU();
 * W / b: n_hidden + 4 tensors in the order proj, timestep_embedding.0, timestep_embedding.2, model's Linear layers. */
typedef struct mtk_clft mtk_clft;
int mtk_clft_create(int device, int D_in, int dim_t, int n_hidden, const int32_t* widths_host, const float* slopes_host,
                    const float* freqs_host, mtk_stream stream, mtk_clft** out);
void mtk_clft_destroy(mtk_clft* clft);
int mtk_clft_set_params(mtk_clft* clft, const float* const* W, const float* const* b, const float* const* gamma, const float* const* beta,
                        mtk_stream stream);
int mtk_clft_forward(mtk_clft* clft, int64_t B, const float* x, int64_t x_ld, const int64_t* t, double dropout_p, uint64_t seed, double bn_eps,
                     float* logits, float* const* batch_mean, float* const* batch_var, mtk_stream stream);
int mtk_clft_backward(mtk_clft* clft, int64_t B, const float* dlogits, int64_t g_ld, float* const* dW, float* const* db, float* const* dgamma,
                      float* const* dbeta, mtk_stream stream);
/* The keep mask (1 = kept) Dropout stage `stage` used in the last mtk_clft_forward: uint8 [B][width of the stage] (tests feed it to
 * the oracle's autograd: torch's own Bernoulli stream cannot be reproduced off-device). */
int mtk_clft_dropout_mask(mtk_clft* clft, int stage, int64_t B, uint8_t* keep, mtk_stream stream);

/* ---- TabSyn's EDM denoiser.  Replaces models/tabsyn/model/modules.py:
 *   mtk_edm_forward  = MLPDiffusion.forward (modules.py:79-95): out = mlp(proj(x) + time_embed(swap(PositionalEmbedding(labels))))
 *   mtk_edm_denoise  = Preconditioner.forward (modules.py:121-144): c_skip x + c_out F(c_in x, log(sigma) / 4)
 *   mtk_edm_loss     = EDMLoss.__call__ around the Preconditioner (modules.py:165-185) on the caller's draws: rnd_normal [B]
 *                      (torch.randn(B)) and noise [B, d_in] (torch.randn_like(data)); loss [B, d_in] = weight (D(y + n, sigma) - y)^2
 * W / b: device fp32 PyTorch-layout weights in the order proj, time_embed.0, time_embed.2, mlp.0, mlp.2, mlp.4, mlp.6;
 * freqs_host [dim_t / 2]: PositionalEmbedding's frequencies as the caller's torch computes them (modules.py:46-48).
 * x / y: [B, ld] row-major fp32; labels / sigma: fp32, element r at [r * stride] (stride 0 broadcasts one value, as the
 * TabSyn sampler's scalar t_hat does, utils.py:77); out / loss: [B, d_in] row-major fp32.  Calls on one object serialise on
 * its workspaces. */
typedef struct mtk_edm mtk_edm;
int mtk_edm_create(int device, int d_in, int dim_t, const float* const* W, const float* const* b, const float* freqs_host,
                   mtk_stream stream, mtk_edm** out);
void mtk_edm_destroy(mtk_edm* edm);
int mtk_edm_forward(mtk_edm* edm, int64_t B, const float* x, int64_t x_ld, const float* noise_labels, int64_t label_stride, float* out,
                    mtk_stream stream);
int mtk_edm_denoise(mtk_edm* edm, int64_t B, const float* x, int64_t x_ld, const float* sigma, int64_t sigma_stride, double sigma_data,
                    float* out, mtk_stream stream);
int mtk_edm_loss(mtk_edm* edm, int64_t B, const float* y, int64_t y_ld, const float* rnd_normal, const float* noise, double p_mean,
                 double p_std, double sigma_data, float* loss, mtk_stream stream);
/* EDMLoss under autograd (modules.py:165-185 as the reference's training loop differentiates it, models/tabsyn/pipeline.py's
 * loss.backward()): mtk_edm_loss_keep is mtk_edm_loss that also keeps, on the handle, every layer's input, the pre-activations of
 * the four SiLUs and d loss / d F; mtk_edm_backward then turns gout [B, d_in] (= d objective / d loss, g_ld its row stride) into
 * d objective / d parameters: dW[i] [out, in] and db[i] [out], unpadded device tensors in mtk_edm_create's order.  Hand-written
 * chain (softmax-free: 2 w (D - y) c_out -> seven weight-gradient GEMMs, five input-gradient GEMMs, SiLU derivatives from the saved
 * pre-activations) on the fp32 MFMA GEMM of the training iteration; one reduction chunk per GEMM, so gradients are identical
 * run to run.  One kept forward per handle at a time (B must match).  mtk_edm_set_weights writes new weights of the same
 * shapes into the handle in place, stream-ordered - an optimizer step does not rebuild the handle. */
int mtk_edm_loss_keep(mtk_edm* edm, int64_t B, const float* y, int64_t y_ld, const float* rnd_normal, const float* noise, double p_mean,
                      double p_std, double sigma_data, float* loss, mtk_stream stream);
int mtk_edm_backward(mtk_edm* edm, int64_t B, const float* gout, int64_t g_ld, float* const* dW, float* const* db, mtk_stream stream);
int mtk_edm_set_weights(mtk_edm* edm, const float* const* W, const float* const* b, mtk_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MTK_H */
