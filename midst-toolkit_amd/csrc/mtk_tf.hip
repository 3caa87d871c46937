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

__global__ __launch_bounds__(256) void k_tf_noise(int64_t rows, int P, int Nn, const float* x, int64_t x_ld, const float* noise,
                                                  float a, float b, float* xt) {
    const int64_t n = rows * Nn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / Nn;
        const int d = (int)(i - row * Nn);
        const int64_t r = row / P;
        const int p = (int)(row - r * P);
        xt[i] = a * x[r * x_ld + d] + b * noise[p * Nn + d];
    }
}

__global__ __launch_bounds__(256) void k_tf_mse(int64_t rows, int P, int Nn, const float* eh, const float* noise, float* score) {
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < rows; row += (int64_t)gridDim.x * 256) {
        const int p = (int)(row % P);
        float acc = 0.0f;
        for (int d = 0; d < Nn; ++d) {
            const float df = noise[p * Nn + d] - eh[row * Nn + d];
            acc += df * df;
        }
        score[row] = acc / (float)Nn;
    }
}

int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// x_t / eps_hat buffers of the unfused path for `rows` rows (records x noise draws) through the denoiser
int ensure_tf_ws(mtk_ctx* c, int64_t rows, hipStream_t s) {
    const size_t need = (size_t)rows * c->Nn;
    if (need <= c->tf_ws_elems) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 2; ++i) {
        if (c->tf_ws[i]) (void)hipFree(c->tf_ws[i]);
        c->tf_ws[i] = nullptr;
        MTK_HIP(hipMalloc(&c->tf_ws[i], sizeof(float) * need));
    }
    c->tf_ws_elems = need;
    ++c->ws_gen;
    return MTK_OK;
}

}  // namespace

int mtk_reserve_fused_tf(mtk_ctx* c, int64_t records, hipStream_t s);  // mtk_mlp_bf16.hip

// mtk_ctx_reserve: rows = records x noise draws; the compact record copy of the fused kernel is sized for `rows` records
int mtk_reserve_tf(mtk_ctx* c, int64_t rows, hipStream_t s) {
    if (c->S != 0 || c->Nn == 0) return MTK_OK;  // Tartan-Federer scoring is defined for numeric-only contexts
    int rc = ensure_tf_ws(c, rows < (1 << 20) ? rows : (1 << 20), s);
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
    const int64_t rec_chunk = (1 << 20) / P > 0 ? (1 << 20) / P : 1;  // <= 1M rows per pass
    {
        const int rc = ensure_tf_ws(c, (B < rec_chunk ? B : rec_chunk) * P, s);
        if (rc) return rc;
    }
    for (int k = 0; k < n_t; ++k) {
        const int t = t_list_host[k];
        const float a = (float)c->tab(MTK_TAB_SQRT_ACP, t + add_t), b = (float)c->tab(MTK_TAB_SQRT_1M_ACP, t + add_t);
        for (int64_t r0 = 0; r0 < B; r0 += rec_chunk) {
            const int64_t nr = (B - r0 < rec_chunk) ? B - r0 : rec_chunk;
            const int64_t rows = nr * P;
            hipLaunchKernelGGL(k_tf_noise, dim3(grid_for(rows * Nn)), dim3(256), 0, s, rows, P, Nn, x + r0 * x_ld, x_ld, noise, a, b,
                               c->tf_ws[0]);
            MTK_HIP(hipGetLastError());
            int rc = mtk_mlp_forward(c, rows, c->tf_ws[0], Nn, 1, nullptr, t, c->tf_ws[1], Nn, 1, precision, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(k_tf_mse, dim3(grid_for(rows)), dim3(256), 0, s, rows, P, Nn, c->tf_ws[1], noise,
                               score + ((int64_t)k * B + r0) * P);
            MTK_HIP(hipGetLastError());
        }
    }
    return MTK_OK;
}
