// mtk_tf.hip — Tartan-Federer denoising-loss scoring (TF:40-95).
//
// score[k][r][p] = mean_d (noise[p][d] - eps_hat(x_t, t_k)[d])^2 with
// x_t = sqrt_acp[t_k + add_t] * x[r] + sqrt_1m_acp[t_k + add_t] * noise[p]   (GMD:302)
// Row r*P + p of the reference's repeat_interleave'd batch is (record r, noise draw p).
// MTK_PREC_FP32: noising kernel -> layered fp32 denoiser -> MSE kernel.
// MTK_PREC_BF16: one fused kernel (mtk_mlp_bf16.hip) that never writes x_t or eps_hat to HBM.
#include "mtk_common.h"

int mtk_fused_tf_score(mtk_ctx* c, int64_t B, int P, const float* x, int64_t x_ld, const float* noise, int n_t,
                       const int32_t* t_list_host, int add_t, float* score, hipStream_t s);  // mtk_mlp_bf16.hip

namespace {

// The unfused path scores ALL (timestep, record, noise draw) triples of a call as one long batch of rows with per-row
// timesteps, cut into chunks that do not care where one timestep ends: global row g = (k * B + r) * P + p is also the index
// of its score.  (Round 2 ran one denoiser call per timestep: 100,000 rows are 3.05 rounds of the layered kernels' 256-row
// tiles over 256 CUs, rounded up to 4 - and two passes of six launches each - per timestep; whole chunks waste nothing.)
constexpr int kTfMaxT = 64;  // timesteps one chunk may touch (their list travels in the kernel arguments)
struct TfRowsArgs {
    int64_t g0, rows, BP;
    int P, Nn, k0, add_t, T;
    const float* x; int64_t x_ld;
    const float* noise;
    const SchedRow* sched;
    float* xt;        // [rows][Nn]
    int64_t* tvec;    // [rows]
    int t_list[kTfMaxT];
};

// x_t = sqrt_acp[t + add] x + sqrt_1m_acp[t + add] noise_p (GMD:302), and the row's timestep for the denoiser
__global__ __launch_bounds__(256) void k_tf_rows(TfRowsArgs a) {
    const int64_t n = a.rows * a.Nn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / a.Nn;
        const int d = (int)(i - row * a.Nn);
        const int64_t g = a.g0 + row;
        const int64_t k = g / a.BP, rp = g - k * a.BP;
        const int64_t r = rp / a.P;
        const int p = (int)(rp - r * a.P);
        const int t = a.t_list[(int)(k - a.k0)];
        const SchedRow sr = a.sched[t + a.add_t];
        a.xt[i] = sr.sqrt_acp * a.x[r * a.x_ld + d] + sr.sqrt_1m_acp * a.noise[p * a.Nn + d];
        if (d == 0) a.tvec[row] = t;
    }
}

// score[g] = mean_d (noise_p[d] - eps_hat[g][d])^2, p = g mod P
__global__ __launch_bounds__(256) void k_tf_mse(int64_t g0, int64_t rows, int P, int Nn, const float* eh, const float* noise, float* score) {
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (int64_t)gridDim.x * 256) {
        const int p = (int)((g0 + row) % P);
        float acc = 0.0f;
        for (int d = 0; d < Nn; ++d) {
            const float df = noise[p * Nn + d] - eh[row * Nn + d];
            acc += df * df;
        }
        score[g0 + row] = acc / (float)Nn;
    }
}

int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

constexpr int64_t kTfChunk = 262144;  // a multiple of every pass size of the layered kernels (and of the fp32 path's)

// x_t / eps_hat buffers ([rows][Nn] each) and the per-row timesteps of the unfused path for `rows` rows
int ensure_tf_ws(mtk_ctx* c, int64_t rows, hipStream_t s) {
    const size_t need = (size_t)rows * c->Nn;
    if (need <= c->tf_ws_elems && (size_t)rows <= c->tf_tvec_elems) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    if (need > c->tf_ws_elems) {
        for (int i = 0; i < 2; ++i) {
            if (c->tf_ws[i]) (void)hipFree(c->tf_ws[i]);
            c->tf_ws[i] = nullptr;
            MTK_HIP(hipMalloc(&c->tf_ws[i], sizeof(float) * need));
        }
        c->tf_ws_elems = need;
    }
    if ((size_t)rows > c->tf_tvec_elems) {
        if (c->tf_tvec) (void)hipFree(c->tf_tvec);
        c->tf_tvec = nullptr;
        MTK_HIP(hipMalloc(&c->tf_tvec, sizeof(int64_t) * (size_t)rows));
        c->tf_tvec_elems = (size_t)rows;
    }
    ++c->ws_gen;
    return MTK_OK;
}

}  // namespace

int mtk_reserve_fused_tf(mtk_ctx* c, int64_t records, hipStream_t s);  // mtk_mlp_bf16.hip

// mtk_ctx_reserve: rows = records x noise draws; the compact record copy of the fused kernel is sized for `rows` records
// rows of the largest chunk a call with B * P = BP rows per timestep cuts (the denoiser's workspaces are reserved for it too)
int64_t mtk_tf_chunk_rows(int64_t BP) { return kTfChunk < (kTfMaxT - 2) * BP ? kTfChunk : (kTfMaxT - 2) * BP; }

int mtk_reserve_tf(mtk_ctx* c, int64_t rows, hipStream_t s) {
    if (c->S != 0 || c->Nn == 0) return MTK_OK;  // Tartan-Federer scoring is defined for numeric-only contexts
    int rc = ensure_tf_ws(c, mtk_tf_chunk_rows(rows), s);
    if (rc) return rc;
    return mtk_reserve_fused_tf(c, rows, s);
}

extern "C" int mtk_tf_score(mtk_ctx* c, int64_t B, int P, const float* x, int64_t x_ld, const float* noise, int n_t,
                            const int32_t* t_list_host, int add_t, float* score, int precision, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_tf_score: null context");
    if (!c->have_weights) MTK_FAIL(MTK_ERR_STATE, "mtk_tf_score: call mtk_mlp_set_weights first");
    if (c->S != 0) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_tf_score: the denoiser takes %d one-hot columns; TF scoring feeds the numeric block only (TF:86-91)", c->S);
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || P < 1 || n_t < 0 || !x || !noise || !score || (n_t > 0 && !t_list_host)) MTK_FAIL(MTK_ERR_ARG, "mtk_tf_score: bad argument");
    if (x_ld < c->Nn) MTK_FAIL(MTK_ERR_ARG, "mtk_tf_score: x_ld %lld < Nn %d", (long long)x_ld, c->Nn);
    for (int k = 0; k < n_t; ++k) {
        const int t = t_list_host[k];
        if (t < 0 || t >= c->T || t + add_t < 0 || t + add_t >= c->T)
            MTK_FAIL(MTK_ERR_ARG, "mtk_tf_score: t = %d (+%d) outside [0, %d)", t, add_t, c->T);
    }
    if (B == 0 || n_t == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (precision == MTK_PREC_BF16) {
        // denoiser shapes, wide records or large noise tables the fused kernel does not cover: same arithmetic as three
        // kernels (noising, bf16 MFMA denoiser — fused or layer-by-layer —, MSE) below
        if (c->fused_ok) {
            const int rc = mtk_fused_tf_score(c, B, P, x, x_ld, noise, n_t, t_list_host, add_t, score, s);
            if (rc != MTK_ERR_UNSUPPORTED) return rc;
        }
    } else if (precision != MTK_PREC_FP32 && precision != MTK_PREC_BF16X3) {
        MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_tf_score: unknown precision %d", precision);
    }

    const int Nn = c->Nn;
    const int64_t BP = B * P, total = BP * n_t;
    int64_t chunk = mtk_tf_chunk_rows(BP);  // a chunk touches at most kTfMaxT timesteps
    if (chunk > total) chunk = total;
    {
        const int rc = ensure_tf_ws(c, chunk, s);
        if (rc) return rc;
    }
    for (int64_t g0 = 0; g0 < total; g0 += chunk) {
        const int64_t rows = total - g0 < chunk ? total - g0 : chunk;
        TfRowsArgs a{};
        a.g0 = g0; a.rows = rows; a.BP = BP; a.P = P; a.Nn = Nn; a.add_t = add_t; a.T = c->T;
        a.k0 = (int)(g0 / BP);
        const int k1 = (int)((g0 + rows - 1) / BP);
        if (k1 - a.k0 + 1 > kTfMaxT) MTK_FAIL(MTK_ERR_STATE, "mtk_tf_score: a chunk spans %d timesteps", k1 - a.k0 + 1);
        for (int k = a.k0; k <= k1; ++k) a.t_list[k - a.k0] = t_list_host[k];
        a.x = x; a.x_ld = x_ld; a.noise = noise; a.sched = c->d_rows; a.xt = c->tf_ws[0]; a.tvec = c->tf_tvec;
        hipLaunchKernelGGL(k_tf_rows, dim3(grid_for(rows * Nn)), dim3(256), 0, s, a);
        MTK_HIP(hipGetLastError());
        c->trusted_t = true;  // every timestep was range-checked above: no NaN pass behind the denoiser
        const int rc = mtk_mlp_forward(c, rows, c->tf_ws[0], Nn, 1, c->tf_tvec, 0, c->tf_ws[1], Nn, 1, precision, stream);
        c->trusted_t = false;
        if (rc) return rc;
        hipLaunchKernelGGL(k_tf_mse, dim3(grid_for(rows)), dim3(256), 0, s, g0, rows, P, Nn, c->tf_ws[1], noise, score);
        MTK_HIP(hipGetLastError());
    }
    return MTK_OK;
}
