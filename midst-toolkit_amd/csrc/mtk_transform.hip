// mtk_transform.hip — the data format right after the reverse chain: the inverse of the
// quantile normalisation the pipeline applies to numeric columns (dataset_transformations.py:57-66 builds a
// sklearn.preprocessing.QuantileTransformer(output_distribution="normal"); synthesizer.py:251-277 calls its
// inverse_transform on every synthesised table).  Restates scikit-learn 1.7.2 QuantileTransformer._transform_col
// (inverse=True): Phi(x) in fp64, linear interpolation of the fitted quantiles over the reference grid exactly as
// numpy's interp evaluates it (slope * (x - xp[j]) + fp[j]), then the two bound overrides with BOUNDS_THRESHOLD.
//
// HBM-bound: 8 B read + 8 B written per element; the reference grid (<= 1000 doubles) sits in LDS, the per-column
// quantile table (n_quantiles x n_columns doubles, <= 160 KB for 20 columns) is served by the L2.
#include <math.h>

#include "mtk_common.h"

namespace {

constexpr int kMaxQ = 4096;  // reference grid entries held in LDS (32 KB)

struct QinvArgs {
    int64_t B;
    int ncol, nq, normal;
    const double* x; int64_t x_ld;
    const double* quant;  // [nq][ncol]
    const double* refs;   // [nq]
    double* out; int64_t o_ld;
};

__global__ __launch_bounds__(256) void k_quantile_inverse(QinvArgs a) {
    __shared__ double refs[kMaxQ];
    for (int i = threadIdx.x; i < a.nq; i += 256) refs[i] = a.refs[i];
    __syncthreads();
    const int64_t total = a.B * a.ncol;
    const double thr = 1e-7;  // BOUNDS_THRESHOLD
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.ncol;
        const int c = (int)(i - r * a.ncol);
        double v = a.x[r * a.x_ld + c];
        if (a.normal) v = 0.5 * erfc(-v * 0.70710678118654752440);  // scipy.stats.norm.cdf
        double res = v;  // NaN stays NaN (isfinite_mask)
        if (!isnan(v)) {
            const double* q = a.quant + c;
            if (v <= refs[0]) {
                res = q[0];
            } else if (v >= refs[a.nq - 1]) {
                res = q[(int64_t)(a.nq - 1) * a.ncol];
            } else {
                int lo = 0, hi = a.nq - 1;  // refs[lo] <= v < refs[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (refs[mid] <= v) lo = mid; else hi = mid;
                }
                const double y0 = q[(int64_t)lo * a.ncol], y1 = q[(int64_t)(lo + 1) * a.ncol];
                const double slope = (y1 - y0) / (refs[lo + 1] - refs[lo]);
                res = slope * (v - refs[lo]) + y0;
                if (isnan(res)) {  // numpy's fallback for infinities in the table
                    res = slope * (v - refs[lo + 1]) + y1;
                    if (isnan(res) && y0 == y1) res = y0;
                }
            }
            if (a.normal) {
                if (v + thr > 1.0) res = q[(int64_t)(a.nq - 1) * a.ncol];
                if (v - thr < 0.0) res = q[0];
            } else {
                if (v == 1.0) res = q[(int64_t)(a.nq - 1) * a.ncol];
                if (v == 0.0) res = q[0];
            }
        }
        a.out[r * a.o_ld + c] = res;
    }
}

// StandardScaler.inverse_transform (x * scale_ + mean_) and MinMaxScaler.inverse_transform ((x - min_) / scale_): the other
// two normalisers of common/dataset_transformations.py:66-69, each in scikit-learn's own operation order (two roundings).
__global__ __launch_bounds__(256) void k_affine_inverse(int64_t B, int ncol, int minmax, const double* x, int64_t x_ld, const double* p0,
                                                        const double* p1, double* out, int64_t o_ld) {
    const int64_t total = B * ncol;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ncol;
        const int c = (int)(i - r * ncol);
        const double v = x[r * x_ld + c];
        out[r * o_ld + c] = minmax ? (v - p0[c]) / p1[c] : v * p0[c] + p1[c];
    }
}

}  // namespace

extern "C" int mtk_affine_inverse(int device, int64_t B, int n_columns, const double* x, int64_t x_ld, int kind, const double* p0,
                                  const double* p1, double* out, int64_t out_ld, mtk_stream stream) {
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || n_columns < 1) MTK_FAIL(MTK_ERR_ARG, "mtk_affine_inverse: bad shape");
    if (kind != 0 && kind != 1) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_affine_inverse: kind = %d (0 = standard, 1 = min-max)", kind);
    if (B == 0) return MTK_OK;
    if (!x || !p0 || !p1 || !out) MTK_FAIL(MTK_ERR_ARG, "mtk_affine_inverse: null pointer");
    if (x_ld < n_columns || out_ld < n_columns) MTK_FAIL(MTK_ERR_ARG, "mtk_affine_inverse: leading dimension below n_columns");
    MTK_HIP(hipSetDevice(device));
    int64_t grid = (B * n_columns + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_affine_inverse, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, B, n_columns, kind, x, x_ld, p0, p1, out, out_ld);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_quantile_inverse(int device, int64_t B, int n_columns, const double* x, int64_t x_ld, const double* quantiles,
                                    const double* references, int n_quantiles, int output_normal, double* out, int64_t out_ld,
                                    mtk_stream stream) {
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || n_columns < 1) MTK_FAIL(MTK_ERR_ARG, "mtk_quantile_inverse: bad shape");
    if (B > 0 && (!x || !quantiles || !references || !out)) MTK_FAIL(MTK_ERR_ARG, "mtk_quantile_inverse: null pointer");
    if (n_quantiles < 2 || n_quantiles > kMaxQ) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_quantile_inverse: n_quantiles = %d outside [2, %d]", n_quantiles, kMaxQ);
    if (x_ld < n_columns || out_ld < n_columns) MTK_FAIL(MTK_ERR_ARG, "mtk_quantile_inverse: leading dimension below n_columns");
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(device));
    QinvArgs a{B, n_columns, n_quantiles, output_normal, x, x_ld, quantiles, references, out, out_ld};
    int64_t grid = (B * n_columns + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_quantile_inverse, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}
