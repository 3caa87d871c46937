// mtk_mlp_f32.hip — MLPDiffusion denoiser (MODEL:548-615), exact-fp32 path + weight ingestion.
//
// Layer-by-layer GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain per
// output element), 128x128x32 workgroup tile, 4 waves as 2x2, each wave 2x2 tiles of 32x32.
// This is the parity path (fp32 in, fp32 accumulate, fp32 out); the throughput path is the fused
// bf16 kernel in mtk_mlp_bf16.hip.  Activations ping-pong through two context-owned buffers.
#include <math.h>

#include "mtk_common.h"

int mtk_pack_fused(mtk_ctx* c, hipStream_t s);    // mtk_mlp_bf16.hip
void mtk_free_fused(mtk_ctx* c, bool keep_ws);    // mtk_mlp_bf16.hip
void mtk_free_train(mtk_ctx* c);                  // mtk_train.hip
void mtk_free_layered(mtk_ctx* c, bool keep_ws);  // mtk_mlp_bf16_layered.hip
int mtk_pack_layered(mtk_ctx* c, hipStream_t s, bool plain);  // mtk_mlp_bf16_layered.hip
int mtk_layered_bf16_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                             float* out, int64_t o_rs, int64_t o_cs, hipStream_t s);  // mtk_mlp_bf16_layered.hip
bool mtk_layered_state_input_ok(const mtk_ctx* c);
int mtk_layered_bf16_forward_state(mtk_ctx* c, int64_t B, const double* z, const int32_t* z_idx, int64_t ld, int t_scalar, float* out, int64_t o_rs,
                                   int64_t o_cs, hipStream_t s);
int mtk_fused_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                      float* out, int64_t o_rs, int64_t o_cs, hipStream_t s);  // mtk_mlp_bf16.hip
int mtk_layered_bf16x3_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                               float* out, int64_t o_rs, int64_t o_cs, hipStream_t s);  // mtk_mlp_bf16_layered.hip

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

struct GemmArgs {
    int64_t M;          // logical rows
    int N, K;           // logical output / input widths
    int Kpad;           // multiple of BK
    const float* X; int64_t x_rs, x_cs;  // generic input (XGEN) or row-major with ld = x_rs (x_cs ignored)
    const float* W; int ldw;             // [Npad][Kpad]
    const float* bias;                   // [Npad]
    const float* emb; int emb_ld;        // optional [T][emb_ld] gathered by t
    const int64_t* tvec; int t_scalar;
    int T;                               // rows of emb
    int act;
    float* Y; int64_t y_rs, y_cs;        // generic output (OGEN) or row-major with ld = y_rs
};

template <bool XGEN, bool OGEN>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BM][LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN][LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // Operand tiles move global -> registers -> LDS, the registers one k-block ahead of the MFMAs (native vectors in
    // named variables: a HIP float4 array here ends up in a stack slot, see mtk_train.hip).
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f xa0, xa1, xa2, xa3, wb0, wb1, wb2, wb3;
    auto fetch_q = [&](int k0, int q, v4f& xv, v4f& wv) {
        const int idx = tid + 256 * q, r = idx >> 3, c = (idx & 7) * 4;
        if (XGEN) {
            const int64_t row = m0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k0 + c + j;
                xv[j] = (row < g.M && kk < g.K) ? g.X[row * g.x_rs + (int64_t)kk * g.x_cs] : 0.0f;
            }
        } else {
            xv = *reinterpret_cast<const v4f*>(g.X + (m0 + r) * g.x_rs + k0 + c);
        }
        wv = *reinterpret_cast<const v4f*>(g.W + (int64_t)(n0 + r) * g.ldw + k0 + c);
    };
    auto fetch = [&](int k0) {
        fetch_q(k0, 0, xa0, wb0);
        fetch_q(k0, 1, xa1, wb1);
        fetch_q(k0, 2, xa2, wb2);
        fetch_q(k0, 3, xa3, wb3);
    };
    auto stash_q = [&](int q, const v4f& xv, const v4f& wv) {
        const int idx = tid + 256 * q, r = idx >> 3, c = (idx & 7) * 4;
        *reinterpret_cast<v4f*>(&As[r][c]) = xv;
        *reinterpret_cast<v4f*>(&Bs[r][c]) = wv;
    };
    fetch(0);
    for (int k0 = 0; k0 < g.Kpad; k0 += BK) {
        stash_q(0, xa0, wb0);
        stash_q(1, xa1, wb1);
        stash_q(2, xa2, wb2);
        stash_q(3, xa3, wb3);
        __syncthreads();
        if (k0 + BK < g.Kpad) fetch(k0 + BK);  // next k-block's loads fly while this one is multiplied
        // lane (li, hh) owns k in [16 hh, 16 hh + 16) of this K block for both operands
        float a[2][16], b[2][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&As[wr * 64 + mt * 32 + li][16 * hh + 4 * q]);
                a[mt][4 * q] = v.x; a[mt][4 * q + 1] = v.y; a[mt][4 * q + 2] = v.z; a[mt][4 * q + 3] = v.w;
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&Bs[wc * 64 + nt * 32 + li][16 * hh + 4 * q]);
                b[nt][4 * q] = v.x; b[nt][4 * q + 1] = v.y; b[nt][4 * q + 2] = v.z; b[nt][4 * q + 3] = v.w;
            }
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA: column (n) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wc * 64 + nt * 32 + li;
            const float bias = g.bias[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float v = acc[mt][nt][r] + bias;
                if (g.emb) {
                    // per-row timesteps are clamped into the [T] table; the host poisons such rows afterwards (mtk_tclamp)
                    const int tr = (g.tvec && row < g.M) ? mtk_tclamp(g.tvec[row], g.T) : (g.tvec ? 0 : g.t_scalar);
                    v += g.emb[(int64_t)tr * g.emb_ld + n];
                }
                if (g.act == ACT_RELU) v = fmaxf(v, 0.0f);
                else if (g.act == ACT_SILU) v = v / (1.0f + expf(-v));
                if (OGEN) {
                    if (row < g.M && n < g.N) g.Y[row * g.y_rs + (int64_t)n * g.y_cs] = v;
                } else {
                    g.Y[row * g.y_rs + n] = v;
                }
            }
        }
}

__global__ void k_pad_copy(float* dst, int rows_pad, int cols_pad, const float* src, int rows, int cols) {
    const int64_t n = (int64_t)rows_pad * cols_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
        dst[i] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.0f;
    }
}

// MODEL:123-144: [cos(t f_i), sin(t f_i)] (+ a zero column when dim is odd), rows padded to BM
// `period` > 0: row r belongs to timestep r % period (label-conditioned tables, mtk_mlp_set_label_embedding)
__global__ void k_sinusoid(float* dst, int Tpad, int ld, int T, int dim, const float* freqs, int period) {
    const int half = dim / 2;
    const int64_t n = (int64_t)Tpad * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / ld), c = (int)(i % ld);
        float v = 0.0f;
        if (t < T && c < 2 * half) {
            const float arg = (float)(period > 0 ? t % period : t) * freqs[c < half ? c : c - half];
            v = c < half ? cosf(arg) : sinf(arg);
        }
        dst[i] = v;
    }
}

// MODEL:612: embeddings += silu(label_embedding(y)), as rows of the table: row r >= period gets label r / period - 1
__global__ void k_add_label_rows(float* emb, int ld, int T, int period, const float* label, int dim) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)T * dim; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / dim), c = (int)(i - (int64_t)r * dim);
        if (r < period) continue;
        const float v = label[(int64_t)(r / period - 1) * dim + c];
        emb[(int64_t)r * ld + c] += v / (1.0f + expf(-v));
    }
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

void free_layer(mtk_ctx* c, MlpLayer& l) {
    if (l.owned) {
        mtk_release(c, l.w32);
        mtk_release(c, l.b32);
    }
    l = MlpLayer();
}

int make_layer(mtk_ctx* c, MlpLayer& l, int in, int out, int in_pad, const float* W, const float* b, hipStream_t s) {
    free_layer(c, l);
    l.in = in;
    l.out = out;
    l.in_pad = in_pad;
    l.out_pad = round_up(out, BN);
    int rc;
    if ((rc = mtk_alloc_t(c, &l.w32, sizeof(float) * (size_t)l.out_pad * l.in_pad))) return rc;
    if ((rc = mtk_alloc_t(c, &l.b32, sizeof(float) * (size_t)l.out_pad))) return rc;
    hipLaunchKernelGGL(k_pad_copy, dim3(256), dim3(256), 0, s, l.w32, l.out_pad, l.in_pad, W, out, in);
    hipLaunchKernelGGL(k_pad_copy, dim3(4), dim3(256), 0, s, l.b32, 1, l.out_pad, b, 1, out);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

template <bool XGEN, bool OGEN>
int launch_gemm(const GemmArgs& g, int64_t Mpad, int Npad, hipStream_t s) {
    hipLaunchKernelGGL((k_gemm_f32<XGEN, OGEN>), dim3((unsigned)(Mpad / BM), (unsigned)(Npad / BN)), dim3(256), 0, s, g);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

}  // namespace

// keep_ws: the activation workspaces stay (they are scratch sized in elements; the ensure_* functions grow them when a new
// shape needs more) - what mtk_mlp_set_weights asks for, so that a weight update leaves nothing to reallocate.
void mtk_free_mlp(mtk_ctx* c, bool keep_ws) {
    ++c->ws_gen;
    free_layer(c, c->proj);
    free_layer(c, c->temb0);
    free_layer(c, c->temb2);
    free_layer(c, c->head);
    for (auto& l : c->blocks) free_layer(c, l);
    c->blocks.clear();
    mtk_release(c, c->d_emb);
    c->d_emb = nullptr;
    if (!keep_ws) {
        for (int i = 0; i < 2; ++i) {
            if (c->ws[i]) (void)hipFree(c->ws[i]);
            c->ws[i] = nullptr;
        }
        c->ws_elems = 0;
        for (int i = 0; i < 2; ++i) {
            if (c->tf_ws[i]) (void)hipFree(c->tf_ws[i]);
            c->tf_ws[i] = nullptr;
        }
        c->tf_ws_elems = 0;
        if (c->tf_tvec) (void)hipFree(c->tf_tvec);
        c->tf_tvec = nullptr;
        c->tf_tvec_elems = 0;
    }
    mtk_free_fused(c, keep_ws);
    mtk_free_layered(c, keep_ws);
    mtk_free_train(c);
    c->have_weights = false;
}

static int reserve_all(mtk_ctx* c, int64_t rows, hipStream_t s);  // below: re-applies an mtk_ctx_reserve reservation

// IsTargetConditioned.EMBEDDING (MODEL:583-585, 610-612).  The denoiser adds silu(label_embedding[y]) to the time embedding;
// here the pair (label, timestep) is one row of the time-embedding table: the context is created with
// T = (n_labels + 1) * period table rows, row (1 + y) * period + t carries label y, rows below `period` none (y = None), and
// the caller passes that row index wherever a timestep goes.  Call before mtk_mlp_set_weights (which builds the table);
// n_labels = 0 clears.  Contexts with a label table serve the denoiser forward only (no training image).
extern "C" int mtk_mlp_set_label_embedding(mtk_ctx* c, const float* rows_dev, int n_labels, int dim, int period, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_label_embedding: null context");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    MTK_HIP(hipStreamSynchronize(s));
    if (c->d_label) (void)hipFree(c->d_label);
    c->d_label = nullptr;
    c->n_label = 0;
    c->t_period = 0;
    c->label_dim = 0;
    c->have_weights = false;  // the table built by the last mtk_mlp_set_weights no longer matches: forward calls fail until it is called again
    if (n_labels == 0) return MTK_OK;
    if (!rows_dev || n_labels < 0 || dim < 2 || period < 1) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_label_embedding: bad argument");
    if ((int64_t)(n_labels + 1) * period != c->T)
        MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_label_embedding: the context has %d table rows, (n_labels + 1) * period = %lld", c->T,
                 (long long)(n_labels + 1) * period);
    MTK_HIP(hipMalloc(&c->d_label, sizeof(float) * (size_t)n_labels * dim));
    MTK_HIP(hipMemcpyAsync(c->d_label, rows_dev, sizeof(float) * (size_t)n_labels * dim, hipMemcpyDeviceToDevice, s));
    MTK_HIP(hipStreamSynchronize(s));
    c->n_label = n_labels;
    c->t_period = period;
    c->label_dim = dim;
    return MTK_OK;
}

extern "C" int mtk_mlp_set_weights(mtk_ctx* c, int n_hidden, const int32_t* dims_host, const float* const* W,
                                   const float* const* b, const float* freqs_host, mtk_stream stream) {
    if (!c || !dims_host || !W || !b || !freqs_host) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_weights: null argument");
    if (n_hidden < 1 || n_hidden > 16) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_set_weights: n_hidden = %d not in [1, 16]", n_hidden);
    const int D = dims_host[0], dim_t = dims_host[1];
    if (D != c->D) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_weights: denoiser width %d != Nn + S = %d", D, c->D);
    if (dim_t < 2 || dim_t > 1024) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_set_weights: dim_t = %d", dim_t);
    if (c->d_label && c->label_dim != dim_t) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_weights: label embedding is %d wide, dim_t = %d", c->label_dim, dim_t);
    for (int i = 0; i < n_hidden; ++i)
        if (dims_host[2 + i] < 1 || dims_host[2 + i] > 8192) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_set_weights: hidden[%d] = %d", i, dims_host[2 + i]);
    for (int i = 0; i < n_hidden + 4; ++i)
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_set_weights: W[%d] or b[%d] is null", i, i);
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    // The old images go to the context's pool and come back as the new ones when the shapes are unchanged (mtk_common.h): the
    // rewrite is then plain stream-ordered work - no allocation, no free, no synchronisation.  Blocks the LAST call left unclaimed
    // (its shapes differed from the ones before) are freed now; work queued on this stream may still read them, hence the wait.
    if (mtk_pool_has_stale(c, c->pool_gen)) {
        MTK_HIP(hipStreamSynchronize(s));
        mtk_pool_drain(c, c->pool_gen);
    }
    ++c->pool_gen;
    if (c->train.attached) MTK_HIP(hipStreamSynchronize(s));  // the training image's buffers are freed outright
    c->pooling = true;
    struct Unpool { mtk_ctx* c; ~Unpool() { c->pooling = false; } } unpool{c};
    mtk_free_mlp(c, true);
    c->dim_t = dim_t;
    c->n_hidden = n_hidden;
    c->hidden.assign(dims_host + 2, dims_host + 2 + n_hidden);
    const int tpad = round_up(dim_t, BN);
    int rc;
    if ((rc = make_layer(c, c->proj, D, dim_t, round_up(D, BK), W[0], b[0], s))) return rc;
    if ((rc = make_layer(c, c->temb0, dim_t, dim_t, tpad, W[1], b[1], s))) return rc;
    if ((rc = make_layer(c, c->temb2, dim_t, dim_t, tpad, W[2], b[2], s))) return rc;
    c->blocks.resize(n_hidden);
    int prev = dim_t, prev_pad = tpad;
    for (int i = 0; i < n_hidden; ++i) {
        if ((rc = make_layer(c, c->blocks[i], prev, c->hidden[i], prev_pad, W[3 + i], b[3 + i], s))) return rc;
        prev = c->hidden[i];
        prev_pad = c->blocks[i].out_pad;
    }
    if ((rc = make_layer(c, c->head, prev, D, prev_pad, W[3 + n_hidden], b[3 + n_hidden], s))) return rc;

    // time-embedding table emb[t] = Linear2(SiLU(Linear0(sinusoid(t))))  (MODEL:608)
    const int Tpad = round_up(c->T, BM);
    float *d_freq = nullptr, *e0 = nullptr, *e1 = nullptr;  // scratch of this call: released to the pool, found there by the next call
    if ((rc = mtk_alloc_t(c, &d_freq, sizeof(float) * (dim_t / 2)))) return rc;
    if ((rc = mtk_alloc_t(c, &e0, sizeof(float) * (size_t)Tpad * tpad))) return rc;
    if ((rc = mtk_alloc_t(c, &e1, sizeof(float) * (size_t)Tpad * tpad))) return rc;
    if ((rc = mtk_alloc_t(c, &c->d_emb, sizeof(float) * (size_t)Tpad * tpad))) return rc;
    c->emb_ld = tpad;
    MTK_HIP(hipMemcpyAsync(d_freq, freqs_host, sizeof(float) * (dim_t / 2), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sinusoid, dim3(256), dim3(256), 0, s, e0, Tpad, tpad, c->T, dim_t, d_freq, c->d_label ? c->t_period : 0);
    GemmArgs g{};
    g.M = Tpad; g.N = dim_t; g.K = dim_t; g.Kpad = tpad;
    g.X = e0; g.x_rs = tpad; g.W = c->temb0.w32; g.ldw = tpad; g.bias = c->temb0.b32;
    g.act = ACT_SILU; g.Y = e1; g.y_rs = tpad;
    if ((rc = launch_gemm<false, false>(g, Tpad, tpad, s))) return rc;
    g.X = e1; g.W = c->temb2.w32; g.bias = c->temb2.b32; g.act = ACT_NONE; g.Y = c->d_emb;
    if ((rc = launch_gemm<false, false>(g, Tpad, tpad, s))) return rc;
    if (c->d_label) {
        hipLaunchKernelGGL(k_add_label_rows, dim3(512), dim3(256), 0, s, c->d_emb, tpad, c->T, c->t_period, c->d_label, dim_t);
        MTK_HIP(hipGetLastError());
    }
    mtk_release(c, d_freq);
    mtk_release(c, e0);
    mtk_release(c, e1);
    c->have_weights = true;
    if ((rc = mtk_pack_fused(c, s))) return rc;
    // split-bf16 weight images for MTK_PREC_BF16X3 (every shape); shapes the fused kernel does not cover also get plain bf16
    // images for the layer-by-layer bf16 MFMA path
    if ((rc = mtk_pack_layered(c, s, !c->fused_ok))) return rc;
    return c->reserved_rows > 0 ? reserve_all(c, c->reserved_rows, s) : MTK_OK;
}

// Exact-fp32 forward for `B` rows (at most `chunk` at a pass) through the activation ping-pong buffers ws0 / ws1
// ([>= min(round_up(B, 128), chunk)][widest padded layer] floats each).  Every GEMM writes the whole padded tile region
// the next one reads, so the buffers need no clearing.
static int forward_f32_ws(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                          float* out, int64_t o_rs, int64_t o_cs, float* ws0, float* ws1, int64_t chunk, hipStream_t s) {
    float* ws[2] = {ws0, ws1};
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t M = (B - r0 < chunk) ? B - r0 : chunk;
        const int64_t Mpad = (M + BM - 1) / BM * BM;
        int rc;
        GemmArgs g{};
        g.M = M; g.N = c->proj.out; g.K = c->proj.in; g.Kpad = c->proj.in_pad;
        g.X = x + r0 * x_rs; g.x_rs = x_rs; g.x_cs = x_cs;
        g.W = c->proj.w32; g.ldw = c->proj.in_pad; g.bias = c->proj.b32;
        g.emb = c->d_emb; g.emb_ld = c->emb_ld; g.tvec = t ? t + r0 : nullptr; g.t_scalar = t_scalar; g.T = c->T;
        g.act = ACT_NONE; g.Y = ws[0]; g.y_rs = c->proj.out_pad;
        if ((rc = launch_gemm<true, false>(g, Mpad, c->proj.out_pad, s))) return rc;
        int cur = 0;
        for (auto& l : c->blocks) {
            GemmArgs h{};
            h.M = M; h.N = l.out; h.K = l.in; h.Kpad = l.in_pad;
            h.X = ws[cur]; h.x_rs = l.in_pad; h.W = l.w32; h.ldw = l.in_pad; h.bias = l.b32;
            h.act = ACT_RELU; h.Y = ws[cur ^ 1]; h.y_rs = l.out_pad;
            if ((rc = launch_gemm<false, false>(h, Mpad, l.out_pad, s))) return rc;
            cur ^= 1;
        }
        GemmArgs h{};
        h.M = M; h.N = c->head.out; h.K = c->head.in; h.Kpad = c->head.in_pad;
        h.X = ws[cur]; h.x_rs = c->head.in_pad; h.W = c->head.w32; h.ldw = c->head.in_pad; h.bias = c->head.b32;
        h.act = ACT_NONE; h.Y = out + r0 * o_rs; h.y_rs = o_rs; h.y_cs = o_cs;
        if ((rc = launch_gemm<false, true>(h, Mpad, c->head.out_pad, s))) return rc;
    }
    return mtk_poison_bad_t(c->trusted_t ? nullptr : t, B, c->T, out, o_rs, o_cs, c->D, nullptr, 0, 0, 0, nullptr, s);
}

static int max_pad_of(const mtk_ctx* c) {
    int maxpad = c->proj.out_pad;
    for (auto& l : c->blocks) maxpad = l.out_pad > maxpad ? l.out_pad : maxpad;
    return maxpad;
}

constexpr int64_t kChunkF32 = 65536;  // rows per pass of the fp32 layer chain

// Sizes the fp32 activation ping-pong for calls of up to B rows (the only place the fp32 path allocates: it synchronises
// the stream and reallocates when a larger B arrives - ahead of time through mtk_ctx_reserve, or on demand).
static int ensure_ws_f32(mtk_ctx* c, int64_t B, hipStream_t s) {
    const int64_t need_rows = (B < kChunkF32 ? (B + BM - 1) / BM * BM : kChunkF32);
    const size_t need = (size_t)need_rows * max_pad_of(c);
    if (need <= c->ws_elems) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 2; ++i) {
        if (c->ws[i]) (void)hipFree(c->ws[i]);
        c->ws[i] = nullptr;
        MTK_HIP(hipMalloc(&c->ws[i], sizeof(float) * need));
    }
    c->ws_elems = need;
    ++c->ws_gen;
    return MTK_OK;
}

// Exact-fp32 forward through the context-owned workspace (grown on demand, or ahead of time by mtk_ctx_reserve).
static int forward_f32(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                       float* out, int64_t o_rs, int64_t o_cs, hipStream_t s) {
    const int rc = ensure_ws_f32(c, B, s);
    if (rc) return rc;
    return forward_f32_ws(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, c->ws[0], c->ws[1], kChunkF32, s);
}

int mtk_reserve_layered(mtk_ctx* c, int64_t rows, hipStream_t s);  // mtk_mlp_bf16_layered.hip
int mtk_reserve_tf(mtk_ctx* c, int64_t rows, hipStream_t s);       // mtk_tf.hip
int mtk_reserve_train(mtk_ctx* c, int64_t rows, hipStream_t s);    // mtk_train.hip

int64_t mtk_tf_chunk_rows(int64_t BP);  // mtk_tf.hip

static int reserve_all(mtk_ctx* c, int64_t rows, hipStream_t s) {
    int rc;
    // a numeric-only context may score `rows` = B * P rows per timestep (mtk_tf_score): its unfused path hands the denoiser
    // chunks that span timesteps, up to mtk_tf_chunk_rows(rows) rows
    const int64_t tf_rows = (c->S == 0 && c->Nn > 0) ? mtk_tf_chunk_rows(rows) : 0;
    const int64_t dn_rows = rows > tf_rows ? rows : tf_rows;
    if ((rc = ensure_ws_f32(c, dn_rows, s))) return rc;
    if ((rc = mtk_reserve_layered(c, dn_rows, s))) return rc;
    if ((rc = mtk_reserve_tf(c, rows, s))) return rc;
    return mtk_reserve_train(c, rows, s);
}

// No reference counterpart ("scratch sized at create/first use ... never synchronises"): sizes every workspace
// of the context for calls of up to max_rows rows through the denoiser (B for mtk_mlp_forward / mtk_train_step, B * P for
// mtk_tf_score), so that none of those entry points allocates or synchronises afterwards - what a caller needs before
// capturing them into a HIP graph.  This call itself may synchronise the stream and allocate.  The reservation survives
// mtk_mlp_set_weights (re-applied there) and only ever grows.
extern "C" int mtk_ctx_reserve(mtk_ctx* c, int64_t max_rows, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_reserve: null context");
    if (max_rows < 0) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_reserve: negative row count");
    if (max_rows > c->reserved_rows) c->reserved_rows = max_rows;
    if (!c->have_weights || c->reserved_rows == 0) return MTK_OK;  // applied when the weights are set
    MTK_HIP(hipSetDevice(c->device));
    return reserve_all(c, c->reserved_rows, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// mtk_mlp_forward_noalloc: the exact-fp32 denoiser for callers that cannot allocate (HIP graph capture, tight serving
// loops).  Same kernels as mtk_mlp_forward(MTK_PREC_FP32); the activation ping-pong lives in a file-scope device array
// instead of a hipMalloc'ed workspace.  Each GPU holds its own copy of the array at its own address: the address is
// looked up once per device (mtk_noalloc_init, called from mtk_ctx_create with that device current) and only that
// pointer - never the array's name, which in host code is a stand-in and not a device address - reaches a kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int64_t kNoallocRows = MTK_NOALLOC_MAX_ROWS;   // 4096 (a multiple of the 128-row tile)
constexpr int kNoallocWidth = MTK_NOALLOC_MAX_WIDTH;     // 1024
static __device__ __align__(256) float mtk_noalloc_scratch[2 * kNoallocRows * kNoallocWidth];  // 32 MiB per GPU
static float* g_noalloc_ptr[64] = {};

int mtk_noalloc_init(int device) {  // the device must be current
    if (device < 0 || device >= 64) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_noalloc_init: device ordinal %d above 63", device);
    if (g_noalloc_ptr[device]) return MTK_OK;
    float* p = nullptr;
    MTK_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(mtk_noalloc_scratch)));
    if (!p) MTK_FAIL(MTK_ERR_HIP, "mtk_noalloc_init: hipGetSymbolAddress returned a null address");
    g_noalloc_ptr[device] = p;
    return MTK_OK;
}

extern "C" int mtk_mlp_forward_noalloc(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t,
                                       int t_scalar, float* out, int64_t o_rs, int64_t o_cs, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward_noalloc: null context");
    if (!c->have_weights) MTK_FAIL(MTK_ERR_STATE, "mtk_mlp_forward_noalloc: call mtk_mlp_set_weights first");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !x || !out) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward_noalloc: null pointer or negative B");
    if (!t && (t_scalar < 0 || t_scalar >= c->T)) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward_noalloc: t_scalar = %d outside [0, %d)", t_scalar, c->T);
    if (B > kNoallocRows)
        MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_forward_noalloc: B = %lld rows exceed the static scratch's %lld (use mtk_mlp_forward)", (long long)B, (long long)kNoallocRows);
    if (max_pad_of(c) > kNoallocWidth)
        MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_forward_noalloc: a layer %d wide exceeds the static scratch's %d columns (use mtk_mlp_forward)", max_pad_of(c), kNoallocWidth);
    if (c->device < 0 || c->device >= 64 || !g_noalloc_ptr[c->device])
        MTK_FAIL(MTK_ERR_STATE, "mtk_mlp_forward_noalloc: the scratch address of device %d was not resolved at mtk_ctx_create", c->device);
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(c->device));
    float* base = g_noalloc_ptr[c->device];
    return forward_f32_ws(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, base, base + kNoallocRows * kNoallocWidth, kNoallocRows,
                          (hipStream_t)stream);
}

// The denoiser on the reverse chain's STATE instead of on a materialised input matrix (GMD:1406 forms
// cat[z_norm, log_z].float() every step): 1 when this context's weights and `precision` have such a path.
extern "C" int mtk_mlp_state_input_ok(mtk_ctx* c, int precision) {
    if (!c || !c->have_weights || c->train.attached) return 0;
    return (precision == MTK_PREC_BF16 && !c->fused_ok && mtk_layered_state_input_ok(c)) ? 1 : 0;
}

extern "C" int mtk_mlp_forward_state(mtk_ctx* c, int64_t B, const double* z, const int32_t* z_idx, int64_t state_ld, int t_scalar, float* out,
                                     int64_t o_rs, int64_t o_cs, int precision, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward_state: null context");
    if (!c->have_weights) MTK_FAIL(MTK_ERR_STATE, "mtk_mlp_forward_state: call mtk_mlp_set_weights first");
    if (B == 0) return MTK_OK;
    if (B < 0 || !out || (c->Nn > 0 && !z) || (c->C > 0 && !z_idx) || state_ld < B) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward_state: null pointer, negative B or state_ld < B");
    if (t_scalar < 0 || t_scalar >= c->T) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward_state: t = %d outside [0, %d)", t_scalar, c->T);
    if (!mtk_mlp_state_input_ok(c, precision))
        MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_forward_state: no state-input path for this denoiser shape / precision (use mtk_mlp_forward)");
    MTK_HIP(hipSetDevice(c->device));
    return mtk_layered_bf16_forward_state(c, B, z, z_idx, state_ld, t_scalar, out, o_rs, o_cs, (hipStream_t)stream);
}

extern "C" int mtk_mlp_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t,
                               int t_scalar, float* out, int64_t o_rs, int64_t o_cs, int precision, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward: null context");
    if (!c->have_weights) MTK_FAIL(MTK_ERR_STATE, "mtk_mlp_forward: call mtk_mlp_set_weights first");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !x || !out) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward: null pointer or negative B");
    if (!t && (t_scalar < 0 || t_scalar >= c->T)) MTK_FAIL(MTK_ERR_ARG, "mtk_mlp_forward: t_scalar = %d outside [0, %d)", t_scalar, c->T);
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(c->device));
    if (precision == MTK_PREC_FP32) return forward_f32(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, (hipStream_t)stream);
    if (precision == MTK_PREC_BF16) {
        if (c->fused_ok) return mtk_fused_forward(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, (hipStream_t)stream);
        if (c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_mlp_forward: the bf16 images do not track a training state; detach it (set the weights again) first");
        return mtk_layered_bf16_forward(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, (hipStream_t)stream);
    }
    if (precision == MTK_PREC_BF16X3) {
        if (c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_mlp_forward: the bf16 images do not track a training state; detach it (set the weights again) first");
        return mtk_layered_bf16x3_forward(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, (hipStream_t)stream);
    }
    MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_forward: unknown precision %d", precision);
}
