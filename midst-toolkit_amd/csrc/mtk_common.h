// mtk_common.h — shared host/device definitions for libmtk_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "mtk.h"

// ----------------------------------------------------------------------------- errors
void mtk_set_error(const char* fmt, ...);
#define MTK_FAIL(code, ...)         \
    do {                            \
        mtk_set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)
#define MTK_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) MTK_FAIL(MTK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// ----------------------------------------------------------------------------- schedule rows
// One 32-byte row per timestep: the eight fp32 tables the per-row-t kernels gather.
struct SchedRow {
    float sqrt_acp, sqrt_1m_acp, log_alpha, log_1m_alpha, log_acp, log_1m_acp, sqrt_recip_acp, sqrt_recipm1_acp;
};
static_assert(sizeof(SchedRow) == 32, "SchedRow must be 32 bytes");

// Device-side view of a context's constant data (passed to kernels by value).
struct DevTables {
    const SchedRow* rows;  // [T]
    const int32_t* off;    // [C+1] one-hot column offsets
    const float* logK;     // [C]
    int T, Nn, C, S;
    int gx0, direct;  // GaussianParametrization.X0 (the denoiser's numeric output is x_0, GMD:404) / Parametrization.DIRECT (its
                      // per-block log-softmax is the reverse distribution itself, GMD:828): mtk_ctx_set_parametrization
};

constexpr int MTK_LDS_T_MAX = 2048;  // schedule rows are staged in LDS up to this T (64 KB)
constexpr int MTK_C_MAX = 1024;      // categorical columns staged in LDS

// ----------------------------------------------------------------------------- denoiser storage
struct MlpLayer {
    int in = 0, out = 0;          // logical sizes
    int in_pad = 0, out_pad = 0;  // padded to the fp32 GEMM tile (K: 32, N: 128)
    float* w32 = nullptr;         // [out_pad][in_pad] fp32, zero padded
    float* b32 = nullptr;         // [out_pad]
    bool owned = true;            // false while the layer reads from an attached training image
};

// Split-batch partials of an iteration's weight / bias gradients: segment i of the gradient image (dst, 4 n4 floats) is the sum,
// in slice order, of `splits` slices of a slab (slice stride 4 n4 floats).  splits = 0: nothing to add, dst already holds the gradient.
constexpr int MTK_MAX_SEGS = 48;
struct MtkSegs {
    int n;
    float* dst[MTK_MAX_SEGS];
    const float* slab[MTK_MAX_SEGS];
    int splits[MTK_MAX_SEGS];
    int64_t n4[MTK_MAX_SEGS];
};

// Training state (mtk_train.hip, a19).  The five parameter images (weights, gradients, Adam m, Adam v, EMA) live in
// ONE caller-owned buffer so the host can all-reduce the gradient image; activations are context-owned.
struct TrainLayerRef {
    MlpLayer* l = nullptr;
    size_t w_off = 0, b_off = 0;  // element offsets inside one image
};
struct TrainState {
    bool attached = false;
    float* state = nullptr;  // caller-owned, 5 * n_img floats
    size_t n_img = 0;
    std::vector<TrainLayerRef> layers;  // proj, temb0, temb2, blocks..., head
    // data-parallel overlap: one event per gradient bucket (head, blocks last..first, proj, time-embedding MLP), recorded on the
    // compute stream when that bucket of the gradient image is final (mtk_train_set_bucket_events); empty: no events
    std::vector<hipEvent_t> bucket_events;
    // mtk_train_set_dropout (MODEL:162-180, MLP.Block = Linear -> ReLU -> Dropout in training mode): probability and the Philox key
    // of the iteration's masks
    float drop_p = 0.0f;
    uint64_t drop_seed = 0;
    // Label-conditioned denoiser under training (MODEL:583-585, 610-612: emb += silu(label_embedding(y))).  The embedding table is a
    // layer of the training image ([n_label rows][dim_t], no bias in use), added to every record's row of act[0] by the proj
    // GEMM's epilogue and differentiated by a scatter of d act[0] over the labels (mtk_train_set_label_classes / _set_labels).
    int n_label = 0;
    MlpLayer label_l;                // points into the image once attached
    const int64_t* yvec = nullptr;   // labels of the next iteration's rows (device, caller-owned)
    long long* dlab_fix = nullptr;   // [label_l.out_pad][emb_ld] fixed-point accumulator, zero between iterations
    int bf16 = 0;  // mtk_train_set_precision: the iteration's GEMMs round their operands to bf16 (fp32 accumulate, fp32 master weights)
    // bf16 mode stores what it multiplies as bf16 (half the bytes through the L2 -> CU pipe that bounds these GEMMs; the values
    // were rounded to bf16 before every product anyway): saved activations, activation gradients, and a copy of the weight
    // image that mtk_train_apply / attach / import keep in step with image 0
    // and, because a tile whose reduction index is the slow one in memory costs twice the LDS-write work of one read along
    // its rows, the TRANSPOSES the backward multiplies: W^T for the input gradients, act^T and dY^T for the weight gradients
    // (written by the epilogue of the GEMM that produces act / dY).
    uint16_t* w16 = nullptr;           // [n_img], layout of one image
    uint16_t* w16t = nullptr;          // [n_img]: layer at w_off as [in_pad][out_pad] (hidden layers and head)
    std::vector<uint16_t*> act16;      // act16[i] ~ act[i], row-major [cap16][out_pad]
    std::vector<uint16_t*> act16t;     // act16t[i] = act16[i]^T, [out_pad][Mpad of the call]
    std::vector<uint16_t*> dact16, dact16t;  // ~ dact[i] and its transpose
    int64_t cap16 = 0;
    // time-embedding tables kept for the backward pass, [Tpad][tpad]
    float *sin_tab = nullptr, *z0_tab = nullptr, *e1_tab = nullptr, *demb = nullptr, *dtab = nullptr;
    int Tpad = 0;
    // activations for `cap` rows (a multiple of 128)
    int64_t cap = 0;
    int64_t fwd_rows = 0;          // rows of the last forward phase (mtk_train_backward must match it)
    // xin / dout keep "rows >= B and padded features are zero" without a memset per iteration: they are zeroed when
    // allocated, the kernels write rows < B only, and a call with fewer rows than the last one zeroes them again first
    int64_t xin_rows = 0, dout_rows = 0;
    float* xin = nullptr;          // [round_up(D,128)][cap] column-major model input
    uint16_t *xin16 = nullptr, *dout16 = nullptr;  // bf16 copies of xin / dout, same layout and zero padding: in bf16 mode the noising and
                                   // loss-gradient kernels write them too, and the proj / head weight gradients read those (LDS-DMA GEMM)
    int32_t* xt_idx = nullptr;     // [C][cap]
    int64_t* tdraw = nullptr;      // [cap] timesteps drawn by the library (mtk_train_step with t = null)
    bool t_drawn = false;          // the last forward phase drew its timesteps (mtk_train_timesteps may copy them out)
    float* loss_means = nullptr;   // caller's [2] device buffer for the batch means of the two losses (mtk_train_set_loss_means)
    std::vector<float*> act;       // act[0] = proj output, act[i] = block i output, row-major [cap][out_pad]
    float *out = nullptr, *dout = nullptr;  // column-major [Dpad128][cap]
    std::vector<float*> dact;      // dact[i] = d loss / d act[i], row-major [cap][width of act[i]] (one per layer: the weight
                                   // gradients run after the whole input-gradient chain, see train_phases)
    // Side stream of the iteration: the time-embedding MLP's backward (six small launches) runs under the weight-gradient GEMMs,
    // and the table refresh after the optimizer step under the optimizer pass over the big layers.  Forked and joined inside
    // the call, by events, so the caller's stream sees one ordered piece of work.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // deterministic reductions (default): split-K partial tiles / row sums land in slabs and are added in a fixed order;
    // the time-embedding scatter accumulates in 64-bit fixed point.  false: fp32 hardware atomics (order varies run to run).
    bool deterministic = true;
    float* slab = nullptr;      // [splits][out_pad][in_pad] of the largest layer, sized with the activations
    size_t slab_elems = 0;
    float* slab_rs = nullptr;   // [splits][out_pad]
    size_t slab_rs_elems = 0;
    long long* demb_fix = nullptr;  // [Tpad][emb_ld] fixed-point accumulator of the time-embedding gradient
    // mtk_train_set_deferred_reduce: mtk_train_step leaves the split-batch partials in the slabs (`pending`) and mtk_train_apply's
    // optimizer pass adds them as it reads the gradient (and writes the sums to the gradient image): one launch and one pass over
    // the gradient image fewer per iteration.  Anything else that reads the image first (mtk_train_export) adds them itself.
    bool defer_reduce = false, segs_pending = false;
    MtkSegs pending;
};

struct mtk_ctx {
    int device = 0;
    int T = 0, Nn = 0, C = 0, S = 0, D = 0;
    std::vector<int32_t> K, off;
    std::vector<float> logK;
    std::vector<double> tables;  // MTK_NTAB x T host copy
    SchedRow* d_rows = nullptr;
    int32_t* d_off = nullptr;
    float* d_logK = nullptr;
    bool train_gauss_kl = false;  // mtk_train_set_gaussian_loss
    double* d_kl = nullptr;  // [T][4]: posterior_mean_coef1/2, posterior_log_variance_clipped, model log-variance (KL Gaussian loss; built on first use)
    const uint64_t* seed_src = nullptr;  // device {seed, row0} read by the reverse-chain kernels instead of their arguments (graph replay)
    int64_t reserved_rows = 0;           // mtk_ctx_reserve: workspaces stay sized for calls of up to this many rows
    // Bumped whenever a context-owned device buffer that kernels are handed is freed or (re)allocated: activation workspaces
    // growing, the packed weight images being rebuilt.  A captured HIP graph holds raw addresses of those buffers; its owner
    // compares mtk_ctx_workspace_generation() with the value at capture time and re-captures when it moved.
    uint64_t ws_gen = 0;
    // Recycling of the weight images (mtk_alloc / mtk_release below).  mtk_mlp_set_weights releases the old images into `pool`
    // instead of hipFree'ing them and takes blocks of the same size back out: with unchanged layer shapes it performs no
    // hipMalloc, no hipFree and no stream synchronisation (hipFree is a device-wide one), and the rewrite is ordered on the
    // caller's stream like any other work.  A block that sat in the pool through a whole mtk_mlp_set_weights call unclaimed
    // (the shapes changed) is freed at the start of the next one.
    struct PoolBlock { void* p; size_t bytes; uint64_t gen; };
    std::vector<PoolBlock> pool;
    std::unordered_map<void*, size_t> alloc_bytes;  // blocks handed out by mtk_alloc
    uint64_t pool_gen = 0;
    uint64_t n_malloc = 0;  // hipMalloc calls made by mtk_alloc (mtk_ctx_alloc_count)
    bool pooling = false;  // mtk_release keeps blocks (inside mtk_mlp_set_weights) or frees them (everywhere else)
    bool fast_math = false;  // MTK_MATH_FAST: diffusion kernels on hardware transcendentals, exact block LSE
    int gauss_x0 = 0, direct = 0;  // mtk_ctx_set_parametrization
    // denoiser
    bool have_weights = false;
    int dim_t = 0, n_hidden = 0;
    std::vector<int> hidden;
    MlpLayer proj, temb0, temb2, head;
    std::vector<MlpLayer> blocks;
    float* d_emb = nullptr;    // [T][dim_t_pad] time-embedding table
    float* d_label = nullptr;  // [n_label][dim_t] label-embedding rows (IsTargetConditioned.EMBEDDING, MODEL:610-612); null: none
    int n_label = 0, t_period = 0, label_dim = 0;  // table row r = (1 + label) * t_period + timestep; rows below t_period carry no label
    int emb_ld = 0;
    // fused bf16 image (mtk_mlp_bf16.hip)
    void* d_wfrag = nullptr;   // packed MFMA A-fragments, consumption order
    size_t wfrag_bytes = 0;
    float* d_bias_all = nullptr;  // concatenated padded biases for the fused kernel
    float* d_embb = nullptr;      // [T][emb_ld] time-embedding table + proj bias (fused kernel)
    // Tartan-Federer mode of the fused kernel: proj folded into the first hidden layer (no nonlinearity sits between them):
    // its own fragment stream (W1 Wp, then the hidden layers and the head) and the table W1 (emb[t] + b_proj) + b1, [T][H]
    void* d_wfrag_tf = nullptr;
    float* d_embf = nullptr;
    int tf_chunks = 0;
    bool fused_ok = false;
    // layered bf16 path (mtk_mlp_bf16_layered.hip): bf16 weight images per layer + activation ping-pong
    std::vector<void*> wb16;    // plain bf16 weight images (shapes the fused kernel does not cover)
    std::vector<void*> wb16x;   // split (hi | lo) bf16 weight images, MTK_PREC_BF16X3 (every shape)
    std::vector<float*> bb16;
    void* wb16_fold = nullptr;  // bf16 image of W1 Wp (proj folded into the first block; launches with one timestep)
    float* embf_l = nullptr;    // [T][embf_ld]: W1 (emb[t] + b_proj) + b1, the folded layer's bias row per timestep
    int embf_ld = 0;
    void* ws16[2] = {nullptr, nullptr};
    size_t ws16_elems = 0;
    int fused_h = 0;
    // workspace for the layered fp32 path
    float* ws[2] = {nullptr, nullptr};
    size_t ws_elems = 0;
    // workspace of the unfused Tartan-Federer path (x_t and eps_hat, row-major [rows][Nn])
    float* tf_ws[2] = {nullptr, nullptr};
    size_t tf_ws_elems = 0;
    int64_t* tf_tvec = nullptr;  // per-row timesteps of the unfused Tartan-Federer path
    size_t tf_tvec_elems = 0;
    bool trusted_t = false;      // set around internal denoiser calls whose per-row timesteps the host has range-checked: skip the NaN pass
    float* tf_pack = nullptr;  // compact [B][round_up(Nn,4)] copy of the records for the fused Tartan-Federer kernel
    size_t tf_pack_elems = 0;
    TrainState train;

    DevTables dev() const { return DevTables{d_rows, d_off, d_logK, T, Nn, C, S, gauss_x0, direct}; }
    double tab(int which, int t) const { return tables[(size_t)which * T + t]; }
};

// mtk_capi.hip.  mtk_alloc: a block of exactly `bytes` from the context's pool, else hipMalloc.  mtk_release: back to the pool while
// `pooling`, else hipFree (null is ignored).  mtk_pool_drain: hipFree the pool's blocks older than `min_gen` (all: UINT64_MAX).
int mtk_alloc(mtk_ctx* c, void** p, size_t bytes);
void mtk_release(mtk_ctx* c, void* p);
bool mtk_pool_has_stale(const mtk_ctx* c, uint64_t min_gen);
void mtk_pool_drain(mtk_ctx* c, uint64_t min_gen);
template <typename T> inline int mtk_alloc_t(mtk_ctx* c, T** p, size_t bytes) { return mtk_alloc(c, reinterpret_cast<void**>(p), bytes); }

// ----------------------------------------------------------------------------- device helpers
#define MTK_LOG_TINY (-0x1.144f6ap+6f) /* fp32 log(1e-30f) = -69.07755, DU:227 */
#define MTK_TINY (0x1.4484cp-100f)     /* 1e-30f */

// Per-row timesteps come from device memory the host never sees.  Every kernel clamps them into the schedule tables
// for addressing (mtk_tclamp) and the entry point then overwrites the outputs of rows whose timestep was outside
// [0, T) with NaN (mtk_poison_bad_t, mtk_diffusion.hip): an out-of-range timestep is loud, never an out-of-bounds read.
// The reference's sinusoid embedding accepts any t; here the embedding is a [T] table.
int mtk_ensure_kl_tab(mtk_ctx* ctx);
// mtk_q_sample with the timesteps drawn in the kernel when t_out is given (written there for the kernels that follow; t is not read)
// and a bf16 copy of x_in (same strides) when x16 is given
int mtk_q_sample_t(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps, const float* u, float* x_in,
                   int64_t xi_rs, int64_t xi_cs, int32_t* xt_idx, uint64_t seed, uint64_t row0, int64_t* t_out, uint16_t* x16, hipStream_t stream);
int mtk_poison_bad_t(const int64_t* t, int64_t B, int T, float* p0, int64_t rs0, int64_t cs0, int n0, float* p1, int64_t rs1,
                     int64_t cs1, int n1, double* p2, hipStream_t s);

#ifdef __HIPCC__
__device__ __forceinline__ int mtk_tclamp(int64_t tv, int T) { return tv < 0 ? 0 : (tv >= T ? T - 1 : (int)tv); }
// Philox-4x32-10 (Salmon et al., SC'11).  Counter = (row_lo, row_hi, group, stream), key = seed.
__device__ __forceinline__ uint4 mtk_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float mtk_u01(uint32_t x) { return (float)(x >> 8) * 0x1p-24f; }  // [0, 1)
// Box-Muller on two 32-bit words -> two standard normals.
__device__ __forceinline__ void mtk_normal2(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 0x1p-24f;  // (0, 1]
    const float u2 = (float)(b >> 8) * 0x1p-24f;
    const float r = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
}
// element j & 3 of (a, b, c, d) by masks.  A select chain over the members of a uint4 (or four floats of a struct) is
// turned into a dynamically indexed stack slot by the compiler: a scratch store, load and vmcnt wait per use.
__device__ __forceinline__ uint32_t mtk_pick4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, int j) {
    const uint32_t m1 = 0u - (uint32_t)(j & 1), m2 = 0u - (uint32_t)((j >> 1) & 1);
    const uint32_t lo = (a & ~m1) | (b & m1), hi = (c & ~m1) | (d & m1);
    return (lo & ~m2) | (hi & m2);
}
__device__ __forceinline__ float mtk_pick4f(float a, float b, float c, float d, int j) {
    return __uint_as_float(mtk_pick4(__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d), j));
}
enum { MTK_STREAM_EPS = 1, MTK_STREAM_U = 2, MTK_STREAM_Z0 = 3, MTK_STREAM_U0 = 4, MTK_STREAM_DROP = 5, MTK_STREAM_T = 6, MTK_STREAM_STEP = 16 };
// element j of the normal stream for `row`
__device__ __forceinline__ float mtk_philox_normal(uint64_t seed, uint64_t row, uint32_t stream, int j) {
    const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)(j >> 2), stream, seed);
    float n0, n1;
    if ((j & 2) == 0) mtk_normal2(r.x, r.y, n0, n1); else mtk_normal2(r.z, r.w, n0, n1);
    return (j & 1) ? n1 : n0;
}
__device__ __forceinline__ float mtk_philox_uniform(uint64_t seed, uint64_t row, uint32_t stream, int j) {
    const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)(j >> 2), stream, seed);
    return mtk_u01(mtk_pick4(r.x, r.y, r.z, r.w, j));
}
// the record's timestep when the library draws it (mtk_train_step with t = null; the uniform sample_time of GMD:945-953): the
// first word of the record's MTK_STREAM_T block scaled to [0, T)
__device__ __forceinline__ int64_t mtk_draw_t(uint64_t seed, uint64_t row, int T) {
    const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), 0u, MTK_STREAM_T, seed);
    return (int64_t)(((uint64_t)r.x * (uint64_t)(uint32_t)T) >> 32);
}
#endif
