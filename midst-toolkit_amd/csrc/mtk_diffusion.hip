// mtk_diffusion.hip — fused, HBM-bound diffusion kernels for gfx950 (wave64).
//
// One lane per record.  Column-major intermediates ([D][B], [C][B]) make every per-lane loop a
// fully coalesced 256-byte wave access; the reference's row-major inputs are read through strides.
// Schedule rows (32 B per timestep) and the per-column category offsets / log K are staged in LDS
// once per workgroup; workgroups are persistent (grid-stride over 256-record chunks).
//
// Built with -ffp-contract=off: every fp32 expression below is evaluated with the same roundings
// as the reference's eager tensor ops (no fused multiply-add), so categorical draws can match.
#include <math.h>
#include <stdlib.h>

#include "mtk_common.h"

namespace {

constexpr int kBlock = 256;

struct Staged {
    const SchedRow* rows;
    const int32_t* off;
    const float* logK;
    unsigned char* end;  // first free byte of dynamic LDS after the tables (16-byte aligned)
};

extern __shared__ __attribute__((aligned(16))) unsigned char mtk_smem[];

// Copies the schedule rows (when LDS_ROWS) and the categorical descriptors into LDS.
template <bool LDS_ROWS>
__device__ __forceinline__ Staged stage_tables(const DevTables& d) {
    Staged s;
    unsigned char* p = mtk_smem;
    if (LDS_ROWS) {
        float4* dst = reinterpret_cast<float4*>(p);
        const float4* src = reinterpret_cast<const float4*>(d.rows);
        for (int i = threadIdx.x; i < d.T * 2; i += blockDim.x) dst[i] = src[i];
        s.rows = reinterpret_cast<const SchedRow*>(p);
        p += (size_t)d.T * sizeof(SchedRow);
    } else {
        s.rows = d.rows;
    }
    int32_t* so = reinterpret_cast<int32_t*>(p);
    float* sk = reinterpret_cast<float*>(p + (((size_t)(d.C + 1) * 4 + 15) & ~(size_t)15));
    for (int i = threadIdx.x; i <= d.C; i += blockDim.x) so[i] = d.off[i];
    for (int i = threadIdx.x; i < d.C; i += blockDim.x) sk[i] = d.logK[i];
    s.off = so;
    s.logK = sk;
    s.end = mtk_smem + ((((reinterpret_cast<unsigned char*>(sk + (d.C > 0 ? d.C : 1)) - mtk_smem) + 15) >> 4) << 4);
    __syncthreads();
    return s;
}

// DU:155  m + log(e^{a-m} + e^{b-m})
__device__ __forceinline__ float lae(float a, float b) {
    const float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}
// DU:250  log(e^{a-m} - e^{b-m}) + m
__device__ __forceinline__ float lse_sub(float a, float b) {
    const float m = fmaxf(a, b);
    return logf(expf(a - m) - expf(b - m)) + m;
}
// GMD:866  -log(-log(u + 1e-30) + 1e-30)
__device__ __forceinline__ float gumbel(float u) { return -logf(-logf(u + MTK_TINY) + MTK_TINY); }
// same map on the hardware log (v_log_f32): used only for in-kernel Philox noise, where no reference
// random stream exists to be bit-compatible with
__device__ __forceinline__ float gumbel_fast(float u) { return -__logf(-__logf(u + MTK_TINY) + MTK_TINY); }

// Sequential reader of one Philox stream of a record: one 10-round Philox call per four values.
struct PhiloxU {
    uint64_t seed, row;
    uint32_t stream;
    int have;
    uint32_t r0, r1, r2, r3;
    __device__ __forceinline__ PhiloxU(uint64_t s, uint64_t rw, uint32_t st) : seed(s), row(rw), stream(st), have(-1) {}
    __device__ __forceinline__ float uniform(int j) {
        if ((j >> 2) != have) {
            have = j >> 2;
            const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)have, stream, seed);
            r0 = r.x; r1 = r.y; r2 = r.z; r3 = r.w;
        }
        return mtk_u01(mtk_pick4(r0, r1, r2, r3, j));
    }
};
struct PhiloxN {
    uint64_t seed, row;
    uint32_t stream;
    int have;
    float n0, n1, n2, n3;
    __device__ __forceinline__ PhiloxN(uint64_t s, uint64_t rw, uint32_t st) : seed(s), row(rw), stream(st), have(-1) {}
    __device__ __forceinline__ float normal(int j) {
        if ((j >> 2) != have) {
            have = j >> 2;
            const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)have, stream, seed);
            mtk_normal2(r.x, r.y, n0, n1);
            mtk_normal2(r.z, r.w, n2, n3);
        }
        return mtk_pick4f(n0, n1, n2, n3, j);
    }
};

// Whole-row cumulative log-sum-exp of DU:268-289, one block at a time.  torch-CPU runs the recurrence
// r <- max + log1p(exp(min - max)) in fp64 over every column and rounds r to fp32 at the block boundaries.
// The same fp64 value (to ~1e-16, i.e. the same fp32 rounding) is obtained with ONE fp64 exp per column:
// b = sum_j exp(x_j - m) with m the block maximum, L = m + log(b), r <- logaddexp(r, L) once per block.
struct RowScan {
    double run;   // fp64 running log-cum-sum-exp
    float prev;   // its fp32 rounding at the previous block boundary
    __device__ __forceinline__ RowScan() : run(-INFINITY), prev(-INFINITY) {}
    // finishes a block whose shifted sum is b (relative to m); returns the reference's fp32 block LSE
    __device__ __forceinline__ float close_block(double b, float m) {
        const double L = (double)m + log(b);
        const double hi = fmax(L, run), lo = fmin(L, run);
        run = (lo != hi || isfinite(lo)) ? log1p(exp(lo - hi)) + hi : L;
        const float cur = (float)run;
        const float out = lse_sub(cur, prev);
        prev = cur;
        return out;
    }
};

// ------------------------------------------------------------------------------------------------
// q_sample: a4 (GMD:302) + a5 (DU:209) + a6 (GMD:742) + a7 (GMD:852, 873)
// ------------------------------------------------------------------------------------------------
struct QSampleArgs {
    DevTables d;
    int64_t B;
    const float* x0; int64_t x0_ld;
    const int64_t* t;
    const float* eps;
    const float* u;
    float* x_in; int64_t xi_rs, xi_cs;
    int32_t* xt_idx;
    uint64_t seed, row0;
    int64_t* t_out;  // non-null: the timestep is drawn here (mtk_draw_t) and written for the kernels that follow; t is not read
    uint16_t* x16;   // non-null: a bf16 copy of x_in, same strides (the bf16 training iteration's weight-gradient operand)
};
// the bf16 copy's element, rounded as the GEMMs round an fp32 operand
__device__ __forceinline__ void put16(uint16_t* p, int64_t i, float v) {
    if (p) {
        const __bf16 e = (__bf16)v;
        p[i] = *reinterpret_cast<const uint16_t*>(&e);
    }
}

template <bool LDS_ROWS>
__global__ __launch_bounds__(kBlock) void k_q_sample(QSampleArgs a) {
    const Staged s = stage_tables<LDS_ROWS>(a.d);
    const int Nn = a.d.Nn, C = a.d.C, S = a.d.S;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const int64_t tv = a.t_out ? mtk_draw_t(a.seed, a.row0 + row, a.d.T) : a.t[row];
        if (a.t_out) a.t_out[row] = tv;
        const int t = mtk_tclamp(tv, a.d.T);
        const SchedRow sr = s.rows[t];
        const float* xr = a.x0 + row * a.x0_ld;
        float* xo = a.x_in + row * a.xi_rs;
        uint16_t* xh = a.x16 ? a.x16 + row * a.xi_rs : nullptr;
        const bool bad16 = tv < 0 || tv >= a.d.T;  // the fp32 rows are poisoned by the entry point, the bf16 copy here
        const float qn16 = __builtin_nanf("");
        PhiloxN pn(a.seed, a.row0 + row, MTK_STREAM_EPS);
        PhiloxU pu(a.seed, a.row0 + row, MTK_STREAM_U);
        for (int j = 0; j < Nn; ++j) {
            const float e = a.eps ? a.eps[row * Nn + j] : pn.normal(j);
            const float xt = sr.sqrt_acp * xr[j] + sr.sqrt_1m_acp * e;
            xo[j * a.xi_cs] = xt;
            put16(xh, j * a.xi_cs, bad16 ? qn16 : xt);
        }
        for (int k = 0; k < C; ++k) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            const int c0 = (int)(long long)xr[Nn + k];
            const float c = sr.log_1m_acp - s.logK[k];
            const float hot = lae(0.0f + sr.log_acp, c);
            const float cold = lae(MTK_LOG_TINY + sr.log_acp, c);
            float best = -INFINITY;
            int arg = 0;
            for (int i = 0; i < Kk; ++i) {
                const float g = a.u ? gumbel(a.u[row * S + o + i]) : gumbel_fast(pu.uniform(o + i));
                const float v = g + (i == c0 ? hot : cold);
                if (v > best || i == 0) { best = v; arg = i; }
            }
            a.xt_idx[(int64_t)k * a.B + row] = arg;
            for (int i = 0; i < Kk; ++i) {
                const float v = (i == arg) ? 0.0f : MTK_LOG_TINY;
                xo[(int64_t)(Nn + o + i) * a.xi_cs] = v;
                put16(xh, (int64_t)(Nn + o + i) * a.xi_cs, bad16 ? qn16 : v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// loss_rows: a9 (GMD:505) + a10 (GMD:761) + a11 (GMD:781, DU:250-289) + a13 (GMD:709, 887, 907, 984)
// ------------------------------------------------------------------------------------------------
struct LossArgs {
    DevTables d;
    int64_t B;
    const float* mo; int64_t mo_rs, mo_cs;
    const float* x0; int64_t x0_ld;
    const int32_t* xt_idx;
    const float* eps;
    const int64_t* t;
    float* loss_multi;
    float* loss_gauss;
    uint64_t seed, row0;
};

template <bool LDS_ROWS>
__global__ __launch_bounds__(kBlock) void k_loss_rows(LossArgs a) {
    const Staged s = stage_tables<LDS_ROWS>(a.d);
    const int Nn = a.d.Nn, C = a.d.C, T = a.d.T;
    const SchedRow srT = s.rows[T - 1];
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const int t = mtk_tclamp(a.t[row], a.d.T);
        const float* mo = a.mo + row * a.mo_rs;
        if (a.loss_gauss && Nn > 0) {  // mean_d (eps - eps_hat)^2
            PhiloxN pn(a.seed, a.row0 + row, MTK_STREAM_EPS);
            float acc = 0.0f;
            for (int j = 0; j < Nn; ++j) {
                const float e = a.eps ? a.eps[row * Nn + j] : pn.normal(j);
                const float df = e - mo[j * a.mo_cs];
                acc += df * df;
            }
            a.loss_gauss[row] = acc / (float)Nn;
        }
        if (!a.loss_multi || C == 0) continue;

        const SchedRow st = s.rows[t];
        const SchedRow s1 = s.rows[t > 0 ? t - 1 : 0];
        const float* xr = a.x0 + row * a.x0_ld;
        RowScan scan_t, scan_m;  // true and model posterior: cumulative over the WHOLE row
        float kl = 0.0f, nll = 0.0f, prior = 0.0f;
        for (int k = 0; k < C; ++k) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            const float logK = s.logK[k];
            const int c0 = (int)(long long)xr[Nn + k];
            const int ct = a.xt_idx[(int64_t)k * a.B + row];
            const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
            // q(x_t | x_{t-1}) evaluated at the sampled x_t (GMD:722)
            const float cq = st.log_1m_alpha - logK;
            const float q_hot = lae(0.0f + st.log_alpha, cq), q_cold = lae(MTK_LOG_TINY + st.log_alpha, cq);
            // q(x_{t-1} | x_0) for the true x_0 (GMD:796-800)
            const float ce = s1.log_1m_acp - logK;
            const float e_hot = (t == 0) ? 0.0f : lae(0.0f + s1.log_acp, ce);
            const float e_cold = (t == 0) ? MTK_LOG_TINY : lae(MTK_LOG_TINY + s1.log_acp, ce);
            // log-softmax of the block's logits (GMD:778)
            float mx = -INFINITY;
            int amax = 0;
            for (int i = 0; i < Kk; ++i) {
                const float v = mc[i * a.mo_cs];
                if (v > mx || i == 0) { mx = v; amax = i; }
            }
            float sum = 0.0f;
            for (int i = 0; i < Kk; ++i) sum += expf(mc[i * a.mo_cs] - mx);
            const float lsum = logf(sum);
            // block maxima of the two unnormalised posteriors (the shift of the fp64 sums): the model term is
            // increasing in the logit, so its maximum sits at the arg-max logit or at the sampled x_t
            // Parametrization.DIRECT (GMD:828-829): the block's log-softmax IS log p(x_{t-1} | x_t) - no posterior, no whole-row
            // log-sum-exp on the model side
            const bool direct = a.d.direct != 0;
            auto un_model = [&](int i) {
                const float lsm = (mc[i * a.mo_cs] - mx) - lsum;
                if (direct) return lsm;
                const float ev = (t == 0) ? lsm : lae(lsm + s1.log_acp, ce);
                return ev + ((i == ct) ? q_hot : q_cold);
            };
            const float m_m = fmaxf(un_model(amax), un_model(ct));
            float m_t = fmaxf(((ct == c0) ? e_hot : e_cold) + q_hot, e_hot + ((ct == c0) ? q_hot : q_cold));
            if (Kk > 2 || (Kk == 2 && ct == c0)) m_t = fmaxf(m_t, e_cold + q_cold);
            double b_t = 0.0, b_m = 0.0;
            for (int i = 0; i < Kk; ++i) {
                const float un_t = ((i == c0) ? e_hot : e_cold) + ((i == ct) ? q_hot : q_cold);
                b_t += exp((double)un_t - (double)m_t);
                b_m += exp((double)un_model(i) - (double)m_m);
            }
            const float lse_t = scan_t.close_block(b_t, m_t), lse_m = direct ? 0.0f : scan_m.close_block(b_m, m_m);
            // prior term q(x_T | x_0) vs uniform (GMD:887)
            const float cT = srT.log_1m_acp - logK;
            const float p_hot = lae(0.0f + srT.log_acp, cT), p_cold = lae(MTK_LOG_TINY + srT.log_acp, cT);
            for (int i = 0; i < Kk; ++i) {
                const float lt = (((i == c0) ? e_hot : e_cold) + ((i == ct) ? q_hot : q_cold)) - lse_t;
                const float lm = un_model(i) - lse_m;
                kl += expf(lt) * (lt - lm);
                nll += expf((i == c0) ? 0.0f : MTK_LOG_TINY) * lm;
                const float qT = (i == c0) ? p_hot : p_cold;
                prior += expf(qT) * (qT + logK);
            }
        }
        const float mask = (t == 0) ? 1.0f : 0.0f;
        const float lt = mask * (-nll) + (1.0f - mask) * kl;
        const float p_t = 1.0f / (float)T;
        a.loss_multi[row] = (lt / p_t + prior) / (float)C;
    }
}

// ------------------------------------------------------------------------------------------------
// p_sample_step: a16 (GMD:358, 545, 323, 663) + a17 (GMD:810, 837, 852); t is uniform over the batch
// ------------------------------------------------------------------------------------------------
struct StepArgs {
    DevTables d;
    int64_t B;
    int t, first;
    // fp32 coefficients of this t and the fp64 standard deviation exp(0.5 * model_log_variance[t])
    float sqrt_recip, sqrt_recipm1, coef1, coef2, log_alpha, log_1m_alpha, log_acp_prev, log_1m_acp_prev;
    double sigma;
    double variance;    // model variance of this t (fp64), used with grad
    const float* grad;  // optional [B, Nn] row-major: classifier guidance, mean <- float(mean) + variance * grad (GMD:614-615)
    const float* mo; int64_t mo_rs, mo_cs;
    double* z;
    int32_t* z_idx;
    const void* noise; int noise_f64;
    const float* u;
    float* x_in; int64_t xi_rs, xi_cs;
    uint64_t seed, row0;
    const uint64_t* seed_src;  // optional device {seed, row0} overriding the two above (graph replay)
};

__global__ __launch_bounds__(kBlock) void k_p_sample_step(StepArgs a) {
    const Staged s = stage_tables<false>(a.d);
    // seed / first global row: kernel arguments, or - for launches replayed from a HIP graph - a device pair (mtk_ctx_set_seed_source)
    const uint64_t seed_ = a.seed_src ? a.seed_src[0] : a.seed, row0_ = a.seed_src ? a.seed_src[1] : a.row0;
    const int Nn = a.d.Nn, C = a.d.C, S = a.d.S, t = a.t;
    const uint32_t stream_n = MTK_STREAM_STEP + 2u * (uint32_t)t, stream_u = stream_n + 1u;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const float* mo = a.mo + row * a.mo_rs;
        float* xo = a.x_in ? a.x_in + row * a.xi_rs : nullptr;
        PhiloxN pn(seed_, row0_ + row, stream_n);
        PhiloxU pu(seed_, row0_ + row, stream_u);
        for (int j = 0; j < Nn; ++j) {
            const float eh = mo[j * a.mo_cs];
            double* zp = a.z + (int64_t)j * a.B + row;
            const double zj = *zp;
            double nz;
            if (a.noise) {
                nz = a.noise_f64 ? reinterpret_cast<const double*>(a.noise)[row * Nn + j]
                                 : (double)reinterpret_cast<const float*>(a.noise)[row * Nn + j];
            } else {
                nz = (double)pn.normal(j);
            }
            double mean;
            // GaussianParametrization.X0 (GMD:404-405): the output is the predicted x_0 itself
            if (a.first) {  // z is still fp32: torch forms the mean in fp32 (SURVEY finding 0.4)
                const float zf = (float)zj;
                const float x0h = a.d.gx0 ? eh : a.sqrt_recip * zf - a.sqrt_recipm1 * eh;
                mean = (double)(a.coef1 * x0h + a.coef2 * zf);
            } else {  // fp32 coefficient x fp64 state promotes; the eps term is an fp32 product
                // (X0: coefficient and output are both fp32, their product is formed in fp32 before the fp64 state promotes the sum)
                const double x0h = (double)a.sqrt_recip * zj - (double)(a.sqrt_recipm1 * eh);
                mean = (a.d.gx0 ? (double)(a.coef1 * eh) : (double)a.coef1 * x0h) + (double)a.coef2 * zj;
            }
            if (a.grad) mean = (double)(float)mean + a.variance * (double)a.grad[row * Nn + j];
            const double mask = (t != 0) ? 1.0 : 0.0;
            const double zn = mean + (mask * a.sigma) * nz;
            *zp = zn;
            if (xo) xo[j * a.xi_cs] = (float)zn;
        }
        RowScan scan;
        for (int k = 0; k < C; ++k) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            const float logK = s.logK[k];
            const int cz = a.z_idx[(int64_t)k * a.B + row];
            const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
            const float cq = a.log_1m_alpha - logK;
            const float q_hot = lae(0.0f + a.log_alpha, cq), q_cold = lae(MTK_LOG_TINY + a.log_alpha, cq);
            const float ce = a.log_1m_acp_prev - logK;
            float mx = -INFINITY;
            int amax = 0;
            for (int i = 0; i < Kk; ++i) {
                const float v = mc[i * a.mo_cs];
                if (v > mx || i == 0) { mx = v; amax = i; }
            }
            float sum = 0.0f;
            for (int i = 0; i < Kk; ++i) sum += expf(mc[i * a.mo_cs] - mx);
            const float lsum = logf(sum);
            const bool direct = a.d.direct != 0;  // GMD:828-829: the log-softmax is the reverse distribution, no posterior
            auto un = [&](int i) {
                const float lsm = (mc[i * a.mo_cs] - mx) - lsum;
                if (direct) return lsm;
                const float ev = (t == 0) ? lsm : lae(lsm + a.log_acp_prev, ce);
                return ev + ((i == cz) ? q_hot : q_cold);
            };
            const float m = fmaxf(un(amax), un(cz));
            double b = 0.0;
            for (int i = 0; i < Kk; ++i) b += exp((double)un(i) - (double)m);
            const float lse = direct ? 0.0f : scan.close_block(b, m);
            float best = -INFINITY;
            int arg = 0;
            for (int i = 0; i < Kk; ++i) {
                const float logp = un(i) - lse;
                const float g = a.u ? gumbel(a.u[row * S + o + i]) : gumbel_fast(pu.uniform(o + i));
                const float v = g + logp;
                if (v > best || i == 0) { best = v; arg = i; }
            }
            a.z_idx[(int64_t)k * a.B + row] = arg;
            if (xo)
                for (int i = 0; i < Kk; ++i) xo[(int64_t)(Nn + o + i) * a.xi_cs] = (i == arg) ? 0.0f : MTK_LOG_TINY;
        }
    }
}

// One DDIM step: gaussian_ddim_step (GMD:1070-1118, with the score-based guidance of condition_score, GMD:617-661, when the
// caller hands the conditioning function's gradient) and multinomial_ddim_step (GMD:1211-1257), for any eta.  Everything
// stays fp32 in the reference (no fp64 table is involved), so the state — kept in the fp64 buffer of the ancestral chain
// — holds fp32-representable values and all arithmetic is fp32 in the reference's operation order.  With eta = 0 (all
// sample_ddim passes) sigma = 0: the noise term vanishes and coef1 = 0 puts -inf on the "stay" branch of the categorical
// mixture, which leaves logsumexp(log(abp) + log_x0, log(1 - abp) - log K).
struct DdimArgs {
    DevTables d;
    int64_t B;
    int t;
    float sqrt_recip, sqrt_recipm1, c_x0, c_eps;  // c_x0 = sqrt(abp), c_eps = sqrt(1 - abp - sigma^2)
    float sig_mask;                               // (t != 0) * sigma: factor of the Gaussian noise (GMD:1115-1116)
    float sqrt_1m_ab;                             // sqrt(1 - alpha_bar): factor of the guidance gradient (GMD:652)
    float log_c1, log_c2, log_c3;                 // logs of the mixture coefficients (GMD:1243-1245)
    const float* mo; int64_t mo_rs, mo_cs;
    double* z;
    int32_t* z_idx;
    const float* noise;  // [B, Nn] row-major fp32, NULL -> Philox (only read when sig_mask != 0)
    const float* grad;   // [B, Nn] row-major fp32 or NULL
    const float* u;
    float* x_in; int64_t xi_rs, xi_cs;
    uint64_t seed, row0;
    const uint64_t* seed_src;
};

__global__ __launch_bounds__(kBlock) void k_ddim_step(DdimArgs a) {
    const Staged s = stage_tables<false>(a.d);
    // seed / first global row: kernel arguments, or - for launches replayed from a HIP graph - a device pair (mtk_ctx_set_seed_source)
    const uint64_t seed_ = a.seed_src ? a.seed_src[0] : a.seed, row0_ = a.seed_src ? a.seed_src[1] : a.row0;
    const int Nn = a.d.Nn, C = a.d.C, S = a.d.S;
    const uint32_t stream_n = MTK_STREAM_STEP + 2u * (uint32_t)a.t, stream_u = stream_n + 1u;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const float* mo = a.mo + row * a.mo_rs;
        float* xo = a.x_in ? a.x_in + row * a.xi_rs : nullptr;
        PhiloxU pu(seed_, row0_ + row, stream_u);
        PhiloxN pn(seed_, row0_ + row, stream_n);
        for (int j = 0; j < Nn; ++j) {
            double* zp = a.z + (int64_t)j * a.B + row;
            const float zf = (float)*zp, eh = mo[j * a.mo_cs];
            float x0h = a.d.gx0 ? eh : a.sqrt_recip * zf - a.sqrt_recipm1 * eh;  // GMD:399-405: from eps (GMD:545), or the output itself
            if (a.grad) {                                                    // condition_score, GMD:649-655
                float e = (a.sqrt_recip * zf - x0h) / a.sqrt_recipm1;
                e = e - a.sqrt_1m_ab * a.grad[row * Nn + j];
                x0h = a.sqrt_recip * zf - a.sqrt_recipm1 * e;
            }
            const float eps = (a.sqrt_recip * zf - x0h) / a.sqrt_recipm1;    // GMD:580-583
            float zn = x0h * a.c_x0 + a.c_eps * eps;                         // GMD:1114
            if (a.sig_mask != 0.0f) zn = zn + a.sig_mask * (a.noise ? a.noise[row * Nn + j] : pn.normal(j));  // GMD:1116
            *zp = (double)zn;
            if (xo) xo[j * a.xi_cs] = zn;
        }
        for (int k = 0; k < C; ++k) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            const float cc = a.log_c3 - s.logK[k];
            const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
            const int cur = a.z_idx[(int64_t)k * a.B + row];
            float mx = -INFINITY;
            for (int i = 0; i < Kk; ++i) mx = fmaxf(mx, mc[i * a.mo_cs]);
            float sum = 0.0f;
            for (int i = 0; i < Kk; ++i) sum += expf(mc[i * a.mo_cs] - mx);
            const float lsum = logf(sum);
            float best = -INFINITY;
            int arg = 0;
            for (int i = 0; i < Kk; ++i) {
                const float st = a.log_c1 + ((i == cur) ? 0.0f : MTK_LOG_TINY);  // log(coef1) + log one-hot of the state
                const float b = a.log_c2 + ((mc[i * a.mo_cs] - mx) - lsum);      // log(coef2) + log_softmax (GMD:761)
                const float m = fmaxf(fmaxf(st, b), cc);
                // torch.logsumexp over (stay, x0, uniform): exp(-inf - m) = 0 drops the stay branch when eta = 0
                const float logp = (m == -INFINITY) ? m : m + logf((expf(st - m) + expf(b - m)) + expf(cc - m));
                const float g = a.u ? gumbel(a.u[row * S + o + i]) : gumbel_fast(pu.uniform(o + i));
                const float v = g + logp;
                if (v > best || i == 0) { best = v; arg = i; }
            }
            a.z_idx[(int64_t)k * a.B + row] = arg;
            if (xo)
                for (int i = 0; i < Kk; ++i) xo[(int64_t)(Nn + o + i) * a.xi_cs] = (i == arg) ? 0.0f : MTK_LOG_TINY;
        }
    }
}

// gaussian_ddim_reverse_step (GMD:1160-1186): the deterministic DDIM map run towards noise, numeric features only, fp32.
struct DdimRevArgs {
    int64_t B;
    int Nn, gx0;
    float sqrt_recip, sqrt_recipm1, c_x0, c_eps;  // sqrt(alpha_bar_next), sqrt(1 - alpha_bar_next)
    const float* mo; int64_t mo_rs, mo_cs;
    double* z;
    float* x_in; int64_t xi_rs, xi_cs;
};

__global__ __launch_bounds__(kBlock) void k_ddim_reverse_step(DdimRevArgs a) {
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        for (int j = 0; j < a.Nn; ++j) {
            double* zp = a.z + (int64_t)j * a.B + row;
            const float zf = (float)*zp, eh = a.mo[row * a.mo_rs + j * a.mo_cs];
            const float x0h = a.gx0 ? eh : a.sqrt_recip * zf - a.sqrt_recipm1 * eh;  // GMD:1177 -> 399-405
            const float eps = (a.sqrt_recip * zf - x0h) / a.sqrt_recipm1;  // GMD:1179-1182
            const float zn = x0h * a.c_x0 + a.c_eps * eps;                 // GMD:1186
            *zp = (double)zn;
            if (a.x_in) a.x_in[row * a.xi_rs + j * a.xi_cs] = zn;
        }
    }
}

// sample() prologue (GMD:1392-1399)
struct InitArgs {
    DevTables d;
    int64_t B;
    const float* z0;
    const float* u0;
    double* z;
    int32_t* z_idx;
    float* x_in; int64_t xi_rs, xi_cs;
    uint64_t seed, row0;
    const uint64_t* seed_src;
};

__global__ __launch_bounds__(kBlock) void k_sample_init(InitArgs a) {
    const Staged s = stage_tables<false>(a.d);
    // seed / first global row: kernel arguments, or - for launches replayed from a HIP graph - a device pair (mtk_ctx_set_seed_source)
    const uint64_t seed_ = a.seed_src ? a.seed_src[0] : a.seed, row0_ = a.seed_src ? a.seed_src[1] : a.row0;
    const int Nn = a.d.Nn, C = a.d.C, S = a.d.S;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        float* xo = a.x_in ? a.x_in + row * a.xi_rs : nullptr;
        PhiloxN pn(seed_, row0_ + row, MTK_STREAM_Z0);
        PhiloxU pu(seed_, row0_ + row, MTK_STREAM_U0);
        for (int j = 0; j < Nn; ++j) {
            const float v = a.z0 ? a.z0[row * Nn + j] : pn.normal(j);
            a.z[(int64_t)j * a.B + row] = (double)v;
            if (xo) xo[j * a.xi_cs] = v;
        }
        for (int k = 0; k < C; ++k) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            float best = -INFINITY;
            int arg = 0;
            if (a.u0) {
                for (int i = 0; i < Kk; ++i) {
                    const float v = gumbel(a.u0[row * S + o + i]) + 0.0f;
                    if (v > best || i == 0) { best = v; arg = i; }
                }
            } else {  // in-kernel noise: the arg-max of K i.i.d. Gumbels is uniform over the K columns
                arg = (int)(pu.uniform(k) * (float)Kk);
                arg = arg > Kk - 1 ? Kk - 1 : arg;
            }
            a.z_idx[(int64_t)k * a.B + row] = arg;
            if (xo)
                for (int i = 0; i < Kk; ++i) xo[(int64_t)(Nn + o + i) * a.xi_cs] = (i == arg) ? 0.0f : MTK_LOG_TINY;
        }
    }
}

// sample() epilogue (GMD:1419-1423): [B, Nn + C] fp64 row-major
__global__ __launch_bounds__(kBlock) void k_sample_decode(int64_t B, int Nn, int C, const double* z, const int32_t* z_idx,
                                                          double* out) {
    const int W = Nn + C;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B * W; i += (int64_t)gridDim.x * kBlock) {
        const int64_t row = i / W;
        const int c = (int)(i - row * W);
        out[i] = (c < Nn) ? z[(int64_t)c * B + row] : (double)z_idx[(int64_t)(c - Nn) * B + row];
    }
}

// ================================================================================================
// MTK_MATH_FAST variants.  Same formulas on the hardware transcendentals (v_exp_f32 / v_log_f32) and
// without the reference's rounding artefacts:
//   * the block normaliser is the exact per-block log-sum-exp m + log(sum exp(x - m)) in fp32 — the
//     reference's cumulative-scan value differs from it by zero-mean fp32 cancellation noise
//     (tests/test_conditioning.py), so batch means agree and per-record values are closer to exact;
//   * the true posterior takes three distinct values per block (x_0's column, x_t's column, the rest) and
//     the prior two, so both are closed forms per block; only the model posterior is per column;
//   * the reverse step skips the normaliser: argmax(g + logp) = argmax(g + unnormalised logp) per block;
//   * the model posterior is evaluated in probability space: with e_i = exp(logit_i - max), q_pred of the
//     log-softmax is log(A e_i + Bc), A = acp_{t-1} / sum e, Bc = (1 - acp_{t-1}) / K (A = 1 / sum e, Bc = 0 at
//     t = 0), so P_i = (A e_i + Bc) Q_i needs one exp per column (and one log where log P_i itself is summed);
//   * blocks of at most kKCache columns are read from memory once and held in registers across the passes;
//   * with in-kernel noise (no injected uniforms) categorical draws use ONE Philox word per block and the
//     inverse CDF of the same distribution instead of K Gumbel variates (noising: two words — x_0's column
//     with its closed-form probability, else uniform over the rest).
// Used for throughput runs (in-kernel Philox noise has no reference stream to be bit-compatible with).
// ================================================================================================
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
__device__ __forceinline__ float flog(float x) { return __logf(x); }
__device__ __forceinline__ float lae_f(float a, float b) {
    const float m = fmaxf(a, b);
    return m + flog(fexp(a - m) + fexp(b - m));
}
__device__ __forceinline__ void normal2_fast(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 0x1p-24f;
    const float u2 = (float)(b >> 8) * 0x1p-24f;
    // the v_sqrt_f32 instruction itself (1 ulp; its argument is 0 or >= 1.2e-7, no denormals): the header's __fsqrt_rn is OCML's
    // "native" square root, which the compiler lowered to the bare instruction in one kernel and to the correctly rounded
    // expansion in another - kernels that must draw the same noise (the lane-per-record and lane-per-item forms) then differed
    // by an ulp in one value out of seven
    const float r = __builtin_amdgcn_sqrtf(-2.0f * flog(u1));
    n0 = r * __cosf(6.283185307179586f * u2);
    n1 = r * __sinf(6.283185307179586f * u2);
}
struct PhiloxNF {
    uint64_t seed, row;
    uint32_t stream;
    int have;
    float n0, n1, n2, n3;
    __device__ __forceinline__ PhiloxNF(uint64_t s, uint64_t rw, uint32_t st) : seed(s), row(rw), stream(st), have(-1) {}
    __device__ __forceinline__ float normal(int j) {
        if ((j >> 2) != have) {
            have = j >> 2;
            const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)have, stream, seed);
            normal2_fast(r.x, r.y, n0, n1);
            normal2_fast(r.z, r.w, n2, n3);
        }
        return mtk_pick4f(n0, n1, n2, n3, j);
    }
};

// one Philox word per use: word j of the row's stream
struct PhiloxW {
    uint64_t seed, row;
    uint32_t stream;
    int have;
    uint32_t r0, r1, r2, r3;
    __device__ __forceinline__ PhiloxW(uint64_t s, uint64_t rw, uint32_t st) : seed(s), row(rw), stream(st), have(-1) {}
    __device__ __forceinline__ float uniform(int j) {
        if ((j >> 2) != have) {
            have = j >> 2;
            const uint4 r = mtk_philox((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)have, stream, seed);
            r0 = r.x; r1 = r.y; r2 = r.z; r3 = r.w;
        }
        return mtk_u01(mtk_pick4(r0, r1, r2, r3, j));  // by masks: see mtk_pick4
    }
};

// The fast kernels are latency-bound unless loads are batched: a lane issues its loads in groups (kGroup numeric
// columns, or the kKCache logits of the NEXT categorical block while the current one is being reduced) so that
// several memory round trips are in flight per wave instead of one.
constexpr int kGroup = 8;
constexpr int kKCache = 16;       // widest categorical block held in registers
constexpr int kKCacheSmall = 12;  // ... when no block of the table is wider than this: 8-16 fewer registers per lane, which
                                  // is what lets k_loss_rows_fast run five waves per SIMD instead of four (it is bound by
                                  // the bytes it keeps in flight: 3.67 -> 4.13 TB/s at shape M)

// The reference's record matrix is row-major: a lane reading "its" row makes every load touch 64 different cache
// lines (120-byte stride), and 30 such loads per record wave cost more L1 tag cycles than the whole kernel's
// coalesced traffic.  When the rows are at most kStageLd floats wide a wave copies its 64 rows (one contiguous span)
// into a private LDS slab with 16-byte coalesced loads and each lane then reads its row from LDS (lgkmcnt, not vmcnt:
// nothing later in the row waits behind the kernel's stores).
constexpr int kStageLd = 32;
__device__ __forceinline__ const float* stage_rows(float* slab, const float* x0, int64_t ld, int64_t row_base, int64_t B, int lane) {
    const int64_t left = B - row_base;
    const int nrow = left < 64 ? (int)left : 64;
    const float* src = x0 + row_base * ld;
    const int nflt = nrow * (int)ld;
    if (nrow == 64) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(slab);
        for (int i = lane; i < nflt / 4; i += 64) d4[i] = s4[i];
    } else {
        for (int i = lane; i < nflt; i += 64) slab[i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return slab + lane * ld;
}
__device__ __forceinline__ void stage_done() {  // before the slab is overwritten by the next record wave
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// raw logits of one categorical block, held in registers when the block has at most KC columns
template <int KC>
struct BlockRaw {
    float v[KC];
    __device__ __forceinline__ void load(const float* mc, int64_t cs, int Kk) {
#pragma unroll
        for (int i = 0; i < KC; ++i) v[i] = (i < Kk) ? mc[i * cs] : -INFINITY;
    }
};
// e_i = exp(logit_i - max) and their sum, from registers (CACHED) or recomputed from memory in every pass
template <bool CACHED, int KC>
struct BlockExp {
    float e[CACHED ? KC : 1];
    const float* mc;
    int64_t cs;
    int Kk;
    float mx, sum;
    __device__ __forceinline__ BlockExp(const BlockRaw<KC>& raw, const float* mc_, int64_t cs_, int Kk_) : mc(mc_), cs(cs_), Kk(Kk_) {
        mx = -INFINITY;
        sum = 0.0f;
        if (CACHED) {
#pragma unroll
            for (int i = 0; i < KC; ++i) mx = fmaxf(mx, raw.v[i]);
#pragma unroll
            for (int i = 0; i < KC; ++i) {
                e[i] = __expf(raw.v[i] - mx);  // 0 for the padding columns
                sum += e[i];
            }
        } else {
            // streamed block (wider than the register cache): chunks of KC columns whose loads are in flight together,
            // running max and rescaled running sum - one pass over the block instead of a max pass and a sum pass of
            // one dependent load each (blocks of 100 levels: loss_rows 0.93 -> TB/s figures in DESIGN section 3)
            for (int i0 = 0; i0 < Kk; i0 += KC) {
                float v[KC];
#pragma unroll
                for (int i = 0; i < KC; ++i) v[i] = (i0 + i < Kk) ? mc[(int64_t)(i0 + i) * cs] : -INFINITY;
                float cm = mx;
#pragma unroll
                for (int i = 0; i < KC; ++i) cm = fmaxf(cm, v[i]);
                float part = 0.0f;
#pragma unroll
                for (int i = 0; i < KC; ++i) part += __expf(v[i] - cm);  // 0 for the padding columns
                sum = sum * __expf(mx - cm) + part;                      // mx = -inf on the first chunk: 0 * 0
                mx = cm;
            }
        }
    }
    __device__ __forceinline__ float at(int i) const { return CACHED ? 0.0f : __expf(mc[(int64_t)i * cs] - mx); }  // streamed blocks only
    template <class F>
    __device__ __forceinline__ void each(F&& f) const {
        if (CACHED) {
#pragma unroll
            for (int i = 0; i < KC; ++i)
                if (i < Kk) f(i, e[i]);
        } else {
            for (int i0 = 0; i0 < Kk; i0 += KC) {
                float v[KC];
#pragma unroll
                for (int i = 0; i < KC; ++i) v[i] = (i0 + i < Kk) ? mc[(int64_t)(i0 + i) * cs] : -INFINITY;
#pragma unroll
                for (int i = 0; i < KC; ++i)
                    if (i0 + i < Kk) f(i0 + i, __expf(v[i] - mx));
            }
        }
    }
};
struct CachedTag { static constexpr bool value = true; };
struct StreamTag { static constexpr bool value = false; };

template <bool LDS_ROWS, bool STAGE>
__global__ __launch_bounds__(kBlock) void k_q_sample_fast(QSampleArgs a) {
    const Staged s = stage_tables<LDS_ROWS>(a.d);
    const int Nn = a.d.Nn, C = a.d.C, S = a.d.S;
    const int lane = threadIdx.x & 63;
    float* slab = reinterpret_cast<float*>(s.end) + (threadIdx.x >> 6) * 64 * (STAGE ? (int)a.x0_ld : 0);
    // the loop bound is wave-uniform (first row of the wave): all 64 lanes take part in the staging
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row - lane < a.B; row += (int64_t)gridDim.x * kBlock) {
        const float* xr;
        if (STAGE) {
            stage_done();
            xr = stage_rows(slab, a.x0, a.x0_ld, row - lane, a.B, lane);
        } else {
            xr = a.x0 + row * a.x0_ld;
        }
        if (row >= a.B) continue;
        const int64_t tv = a.t_out ? mtk_draw_t(a.seed, a.row0 + row, a.d.T) : a.t[row];
        if (a.t_out) a.t_out[row] = tv;
        const int t = mtk_tclamp(tv, a.d.T);
        // a timestep outside [0, T): the record's outputs are NaN (mtk_tclamp keeps the addressing in bounds); folded into
        // the stores here, so this path needs no k_poison_bad_t pass behind it (a launch and its gap per call)
        const bool bad_t = tv < 0 || tv >= a.d.T;
        const float qnan = __builtin_nanf("");
        float* xo = a.x_in + row * a.xi_rs;
        uint16_t* xh = a.x16 ? a.x16 + row * a.xi_rs : nullptr;
        // first group of inputs in flight before anything is consumed
        float xv[kGroup], ev[kGroup];
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            xv[i] = (i < Nn) ? xr[i] : 0.0f;
            ev[i] = (a.eps && i < Nn) ? a.eps[row * Nn + i] : 0.0f;
        }
        const SchedRow sr = s.rows[t];
        PhiloxNF pn(a.seed, a.row0 + row, MTK_STREAM_EPS);
        PhiloxW pu(a.seed, a.row0 + row, MTK_STREAM_U);
        for (int j0 = 0; j0 < Nn; j0 += kGroup) {
            float xn[kGroup], en[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {  // next group
                const int j = j0 + kGroup + i;
                xn[i] = (j < Nn) ? xr[j] : 0.0f;
                en[i] = (a.eps && j < Nn) ? a.eps[row * Nn + j] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int j = j0 + i;
                if (j < Nn) {
                    const float e = a.eps ? ev[i] : pn.normal(j);
                    const float xt = sr.sqrt_acp * xv[i] + sr.sqrt_1m_acp * e;
                    xo[j * a.xi_cs] = bad_t ? qnan : xt;
                    put16(xh, j * a.xi_cs, bad_t ? qnan : xt);
                }
            }
#pragma unroll
            for (int i = 0; i < kGroup; ++i) { xv[i] = xn[i]; ev[i] = en[i]; }
        }
        if (a.u) {  // injected uniforms: the reference's Gumbel arg-max
            for (int k = 0; k < C; ++k) {
                const int o = s.off[k], Kk = s.off[k + 1] - o;
                const int c0 = (int)(long long)xr[Nn + k];
                const float c = sr.log_1m_acp - s.logK[k];
                const float hot = lae_f(0.0f + sr.log_acp, c);
                const float cold = lae_f(MTK_LOG_TINY + sr.log_acp, c);
                float best = -INFINITY;
                int arg = 0;
                for (int i = 0; i < Kk; ++i) {
                    const float v = gumbel_fast(a.u[row * S + o + i]) + (i == c0 ? hot : cold);
                    if (v > best || i == 0) { best = v; arg = i; }
                }
                a.xt_idx[(int64_t)k * a.B + row] = arg;
                for (int i = 0; i < Kk; ++i) {
                    const float v = bad_t ? qnan : ((i == arg) ? 0.0f : MTK_LOG_TINY);
                    xo[(int64_t)(Nn + o + i) * a.xi_cs] = v;
                    put16(xh, (int64_t)(Nn + o + i) * a.xi_cs, v);
                }
            }
            continue;
        }
        // in-kernel noise.  q(x_t | x_0) = acp onehot(x_0) + (1 - acp) / K: x_0 with probability p0, else uniform over
        // the other columns; two Philox words per block.  x_0's codes are fetched kGroup blocks at a time.
        const float acp = fexp(sr.log_acp);
        for (int k0 = 0; k0 < C; k0 += kGroup) {
            float cv[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; ++i) cv[i] = (k0 + i < C) ? xr[Nn + k0 + i] : 0.0f;
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int k = k0 + i;
                if (k < C) {
                    const int o = s.off[k], Kk = s.off[k + 1] - o;
                    const int c0 = (int)(long long)cv[i];
                    const float pc = fexp(sr.log_1m_acp - s.logK[k]);  // (1 - acp) / K
                    const float ph = acp + pc;
                    const float p0 = ph / (ph + (float)(Kk - 1) * pc);
                    const float u1 = pu.uniform(2 * k), u2 = pu.uniform(2 * k + 1);
                    int arg = c0;
                    if (u1 >= p0 && Kk > 1) {
                        int j = (int)(u2 * (float)(Kk - 1));
                        j = j > Kk - 2 ? Kk - 2 : j;
                        arg = j + (j >= c0 ? 1 : 0);
                    }
                    a.xt_idx[(int64_t)k * a.B + row] = arg;
                    float* xc = xo + (int64_t)(Nn + o) * a.xi_cs;
                    const float hot_v = bad_t ? qnan : 0.0f, cold_v = bad_t ? qnan : MTK_LOG_TINY;
                    for (int c = 0; c < Kk; ++c) {
                        xc[c * a.xi_cs] = (c == arg) ? hot_v : cold_v;
                        put16(xh, (int64_t)(Nn + o + c) * a.xi_cs, (c == arg) ? hot_v : cold_v);
                    }
                }
            }
        }
    }
}

// The same draws and stores with one lane per (record, item), item = a numeric column or a categorical block (blockIdx.y): at
// the trainer's batch sizes a lane-per-record launch is a few dozen waves, each walking its record's Nn + C items as one chain
// (2,048 records, 20 + 10 items: 18 us); spread over the items the chain is one Philox block and a handful of stores.  A numeric
// lane forms the whole four-normal Philox block of its column and keeps one value, so the work grows fourfold there: for
// batches that fill the chip by records alone the lane-per-record kernel stays the better one (mtk_q_sample picks).
// In-kernel noise or injected eps; injected uniforms (the reference's Gumbel arg-max) stay with the kernels above.
constexpr int64_t kItemsMaxRows = 16384;
__global__ __launch_bounds__(kBlock) void k_q_sample_items(QSampleArgs a) {
    const int Nn = a.d.Nn;
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= a.B) return;
    const int item = blockIdx.y;
    const int64_t tv = a.t_out ? mtk_draw_t(a.seed, a.row0 + row, a.d.T) : a.t[row];
    if (a.t_out && item == 0) a.t_out[row] = tv;
    const float* xr = a.x0 + row * a.x0_ld;
    const float xv = xr[item];  // the column's value, or the block's code (x_0 = [numeric | codes])
    const int t = mtk_tclamp(tv, a.d.T);
    const bool bad_t = tv < 0 || tv >= a.d.T;
    const float qnan = __builtin_nanf("");
    const SchedRow sr = a.d.rows[t];
    float* xo = a.x_in + row * a.xi_rs;
    uint16_t* xh = a.x16 ? a.x16 + row * a.xi_rs : nullptr;
    if (item < Nn) {
        PhiloxNF pn(a.seed, a.row0 + row, MTK_STREAM_EPS);
        const float e = a.eps ? a.eps[row * Nn + item] : pn.normal(item);
        const float xt = sr.sqrt_acp * xv + sr.sqrt_1m_acp * e;
        xo[item * a.xi_cs] = bad_t ? qnan : xt;
        put16(xh, item * a.xi_cs, bad_t ? qnan : xt);
        return;
    }
    const int k = item - Nn;
    const int o = a.d.off[k], Kk = a.d.off[k + 1] - o;
    PhiloxW pu(a.seed, a.row0 + row, MTK_STREAM_U);
    const float acp = fexp(sr.log_acp);
    const int c0 = (int)(long long)xv;
    const float pc = fexp(sr.log_1m_acp - a.d.logK[k]);  // (1 - acp) / K
    const float ph = acp + pc;
    const float p0 = ph / (ph + (float)(Kk - 1) * pc);
    const float u1 = pu.uniform(2 * k), u2 = pu.uniform(2 * k + 1);
    int arg = c0;
    if (u1 >= p0 && Kk > 1) {
        int j = (int)(u2 * (float)(Kk - 1));
        j = j > Kk - 2 ? Kk - 2 : j;
        arg = j + (j >= c0 ? 1 : 0);
    }
    a.xt_idx[(int64_t)k * a.B + row] = arg;
    float* xc = xo + (int64_t)(Nn + o) * a.xi_cs;
    const float hot_v = bad_t ? qnan : 0.0f, cold_v = bad_t ? qnan : MTK_LOG_TINY;
    for (int c = 0; c < Kk; ++c) {
        xc[c * a.xi_cs] = (c == arg) ? hot_v : cold_v;
        put16(xh, (int64_t)(Nn + o + c) * a.xi_cs, (c == arg) ? hot_v : cold_v);
    }
}

struct TruePost { float lt_c0, lt_ct, w_c0, w_ct, w_cc, w_sum, self; };

template <bool LDS_ROWS, bool STAGE, int KC, bool STREAM>
__device__ __forceinline__ void loss_rows_fast_body(const LossArgs& a) {
    const Staged s = stage_tables<LDS_ROWS>(a.d);
    const int Nn = a.d.Nn, C = a.d.C, T = a.d.T;
    const SchedRow srT = s.rows[T - 1];
    const int lane = threadIdx.x & 63;
    float* slab = reinterpret_cast<float*>(s.end) + (threadIdx.x >> 6) * 64 * (STAGE ? (int)a.x0_ld : 0);
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row - lane < a.B; row += (int64_t)gridDim.x * kBlock) {
        const float* xr;
        if (STAGE) {  // x_0's category codes come from the wave's LDS slab (stage_rows above)
            stage_done();
            xr = stage_rows(slab, a.x0, a.x0_ld, row - lane, a.B, lane);
        } else {
            xr = a.x0 ? a.x0 + row * a.x0_ld : nullptr;
        }
        if (row >= a.B) continue;
        const int64_t tv = a.t[row];
        const int t = mtk_tclamp(tv, a.d.T);
        const bool bad_t = tv < 0 || tv >= a.d.T;  // the multinomial loss of such a record is NaN (the MSE term does not depend on t)
        const float* mo = a.mo + row * a.mo_rs;
        const bool multi = a.loss_multi && C > 0;
        // Categorical blocks are requested TWO blocks ahead of their use (three rotating register images): a lane's chain
        // of dependent round trips - not the byte rate - bounds this kernel (a wave lives ~35 us per record at five waves
        // per SIMD, a dozen exposed HBM latencies), so the bytes a wave keeps in flight are what counts.  The first two
        // blocks' inputs go out before the Gaussian part is consumed.
        BlockRaw<KC> rb[3];
        int c0s[3] = {0, 0, 0}, cts[3] = {0, 0, 0};
        auto fetch = [&](int k, BlockRaw<KC>& dst, int& c0d, int& ctd) {
            const int o = s.off[k];
            dst.load(mo + (int64_t)(Nn + o) * a.mo_cs, a.mo_cs, s.off[k + 1] - o);
            c0d = (int)(long long)xr[Nn + k];
            ctd = a.xt_idx[(int64_t)k * a.B + row];
        };
        if (multi) {
            fetch(0, rb[0], c0s[0], cts[0]);
            if (C > 1) fetch(1, rb[1], c0s[1], cts[1]);
        }
        if (a.loss_gauss && Nn > 0) {
            PhiloxNF pn(a.seed, a.row0 + row, MTK_STREAM_EPS);
            float acc = 0.0f;
            for (int j0 = 0; j0 < Nn; j0 += kGroup) {
                float mv[kGroup], ev[kGroup];
#pragma unroll
                for (int i = 0; i < kGroup; ++i) {
                    const int j = j0 + i;
                    mv[i] = (j < Nn) ? mo[j * a.mo_cs] : 0.0f;
                    ev[i] = (a.eps && j < Nn) ? a.eps[row * Nn + j] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < kGroup; ++i) {
                    const int j = j0 + i;
                    if (j < Nn) {
                        const float df = (a.eps ? ev[i] : pn.normal(j)) - mv[i];
                        acc += df * df;
                    }
                }
            }
            a.loss_gauss[row] = acc / (float)Nn;
        }
        if (!multi) continue;
        const SchedRow st = s.rows[t];
        const SchedRow s1 = s.rows[t > 0 ? t - 1 : 0];
        // Parametrization.DIRECT (GMD:828-829): the model side is the block's softmax itself - P_i = e_i / sum e, i.e. A = 1 / sum,
        // Bc = 0, Q = 1 in the probability-space form below; the true posterior is unchanged
        const bool direct = a.d.direct != 0;
        const float acp1 = (t == 0 || direct) ? 1.0f : fexp(s1.log_acp);
        float kl = 0.0f, nll = 0.0f, prior = 0.0f;
        int Kc = -1;
        TruePost tpE{}, tpN{};
        float logK = 0.0f, q_hot = 0.0f, q_cold = 0.0f, e_hot = 0.0f, e_cold = 0.0f, Bc = 0.0f, Qh = 0.0f, Qc = 0.0f, prior_k = 0.0f;
        auto block = [&](int k, const BlockRaw<KC>& raw, int c0, int ct) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
            // constants of (t, K): recomputed only when K changes from one block to the next (K is wave-uniform)
            if (Kk != Kc) {
                Kc = Kk;
                logK = s.logK[k];
                const float cq = st.log_1m_alpha - logK;
                q_hot = lae_f(0.0f + st.log_alpha, cq);
                q_cold = lae_f(MTK_LOG_TINY + st.log_alpha, cq);
                const float ce = s1.log_1m_acp - logK;
                e_hot = (t == 0) ? 0.0f : lae_f(0.0f + s1.log_acp, ce);
                e_cold = (t == 0) ? MTK_LOG_TINY : lae_f(MTK_LOG_TINY + s1.log_acp, ce);
                Bc = (t == 0 || direct) ? 0.0f : fexp(ce);
                Qh = direct ? 1.0f : fexp(q_hot);
                Qc = direct ? 1.0f : fexp(q_cold);
                const float cT = srT.log_1m_acp - logK;
                const float p_hot = lae_f(0.0f + srT.log_acp, cT), p_cold = lae_f(MTK_LOG_TINY + srT.log_acp, cT);
                prior_k = fexp(p_hot) * (p_hot + logK) + (float)(Kk - 1) * (fexp(p_cold) * (p_cold + logK));
                // true posterior (GMD:781 on the one-hot x_0): three distinct unnormalised values — x_0's column, x_t's
                // column when it differs, every other column — with a closed-form normaliser; both variants
                auto post = [&](bool same, TruePost& o_) {
                    const float v_c0 = e_hot + (same ? q_hot : q_cold);
                    const float v_ct = e_cold + q_hot;
                    const float v_cc = e_cold + q_cold;
                    const int n_cc = Kk - (same ? 1 : 2);
                    const float m_t = fmaxf(v_c0, fmaxf(v_ct, v_cc));
                    float b_t = fexp(v_c0 - m_t) + (float)n_cc * fexp(v_cc - m_t);
                    if (!same) b_t += fexp(v_ct - m_t);
                    const float lse_t = m_t + flog(b_t);
                    o_.lt_c0 = v_c0 - lse_t;
                    o_.lt_ct = v_ct - lse_t;
                    const float lt_cc = v_cc - lse_t;
                    o_.w_c0 = fexp(o_.lt_c0);
                    o_.w_ct = fexp(o_.lt_ct);
                    o_.w_cc = fexp(lt_cc);
                    o_.w_sum = o_.w_c0 + (float)n_cc * o_.w_cc + (same ? 0.0f : o_.w_ct);
                    o_.self = o_.w_c0 * o_.lt_c0 + (float)n_cc * (o_.w_cc * lt_cc) + (same ? 0.0f : o_.w_ct * o_.lt_ct);
                };
                post(true, tpE);
                post(false, tpN);
            }
            // true posterior of this block: the x_t == x_0 or the x_t != x_0 variant
            const bool eq = (c0 == ct);
            const float lt_c0 = eq ? tpE.lt_c0 : tpN.lt_c0, w_c0 = eq ? tpE.w_c0 : tpN.w_c0;
            const float lt_ct = tpN.lt_ct, w_ct = tpN.w_ct;
            const float w_cc = eq ? tpE.w_cc : tpN.w_cc;
            const float self = eq ? tpE.self : tpN.self, w_sum = eq ? tpE.w_sum : tpN.w_sum;
            (void)lt_c0; (void)lt_ct;
            // model posterior in probability space: P_i = (A e_i + Bc) Q_i, log p_i = log P_i - log sum P.  The kernel is
            // VALU-bound (SQ counters: 10 k vector instructions per wave, the matrix of per-column logs the largest
            // part), so the column loop is branch-free and in log2 units: the bare v_log_f32 per column (HIP's __logf is
            // the twelve-instruction precise expansion), every column summed and the two special ones picked by selects
            // and subtracted afterwards, one multiplication by ln 2 per block, v_rcp_f32 for the normaliser.
            float total = 0.0f, l2_all = 0.0f, l2_c0 = 0.0f, l2_ct = 0.0f;
            auto model = [&](auto tag) {
                const BlockExp<decltype(tag)::value, KC> be(raw, mc, a.mo_cs, Kk);
                const float A = acp1 * __builtin_amdgcn_rcpf(be.sum);
                be.each([&](int i, float e) {
                    const float P = (A * e + Bc) * ((i == ct) ? Qh : Qc);
                    total += P;
                    const float l2 = __builtin_amdgcn_logf(P);
                    l2_all += l2;
                    l2_c0 = (i == c0) ? l2 : l2_c0;
                    l2_ct = (i == ct) ? l2 : l2_ct;
                });
            };
            // STREAM = false: the host picked this instantiation because no block is wider than KC, and the streamed
            // path is not even compiled (its chunk registers cost the small kernel 11 spilled VGPRs and 3x its stores)
            if (!STREAM || Kk <= KC) model(CachedTag{}); else model(StreamTag{});
            // t = 0 with a vanishing e_c0 (a logit gap beyond 87): v_log_f32 flushes the denormal P to -inf; the
            // reference's log-softmax stays finite there, this path saturates at 2^-126.  For t > 0, Bc > 0 keeps P normal.
            l2_c0 = fmaxf(l2_c0, -126.0f);
            constexpr float kLn2 = 0.69314718055994531f;
            const float lp_c0 = kLn2 * l2_c0, lp_ct = kLn2 * l2_ct;
            const float sum_log_cc = kLn2 * (l2_all - l2_c0 - (eq ? 0.0f : l2_ct));
            const float lse_m = kLn2 * __builtin_amdgcn_logf(total);
            float cross = w_c0 * lp_c0 + w_cc * sum_log_cc;              // sum_i w_i log P_i
            if (!eq) cross += w_ct * lp_ct;
            kl += (self - cross) + lse_m * w_sum;                        // self = sum_i w_i log w_i
            nll += lp_c0 - lse_m;                  // log-probability of x_0 (columns weighted 1e-30 dropped)
            prior += prior_k;
        };
        for (int k0 = 0; k0 < C; k0 += 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = k0 + j;
                if (k < C) {
                    if (k + 2 < C) fetch(k + 2, rb[(j + 2) % 3], c0s[(j + 2) % 3], cts[(j + 2) % 3]);
                    block(k, rb[j], c0s[j], cts[j]);
                }
            }
        }
        const float lt = (t == 0) ? -nll : kl;
        a.loss_multi[row] = bad_t ? __builtin_nanf("") : (lt * (float)T + prior) / (float)C;
    }
}

template <bool LDS_ROWS, bool STAGE>
__global__ __launch_bounds__(kBlock) void k_loss_rows_fast(LossArgs a) { loss_rows_fast_body<LDS_ROWS, STAGE, kKCache, true>(a); }
// no block wider than kKCacheSmall: three 12-column images in flight fit four waves per SIMD (128 registers)
template <bool LDS_ROWS, bool STAGE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_loss_rows_fast_small(LossArgs a) {
    loss_rows_fast_body<LDS_ROWS, STAGE, kKCacheSmall, false>(a);
}

// One categorical block of a reverse step with in-kernel noise: the new state column = inverse CDF of
// P_i = (A e_i + Bc) Q_i at the uniform u1 (see the notes above k_p_sample_step_fast).  Shared by the lane-per-record and the
// lane-per-(record, item) kernels, which therefore draw the same columns.
template <int KC>
__device__ __forceinline__ int step_draw_block(const BlockRaw<KC>& raw, const float* mc, int64_t cs, int Kk, int cz, float acp1, float al, float logK,
                                               float log_1m_alpha, float log_1m_acp_prev, int t, float u1, bool direct) {
    // Parametrization.DIRECT (GMD:828-829): the softmax itself is the reverse distribution - P_i = e_i, i.e. Q = 1 and Bc = 0
    const float qc = fexp(log_1m_alpha - logK);
    const float Qh = direct ? 1.0f : al + qc, Qc = direct ? 1.0f : qc;
    const float Bc = (t == 0 || direct) ? 0.0f : fexp(log_1m_acp_prev - logK);
    int arg = Kk - 1;
    auto draw = [&](auto tag) {
        const BlockExp<decltype(tag)::value, KC> be(raw, mc, cs, Kk);
        const float A = acp1 * __builtin_amdgcn_rcpf(be.sum);
        // inverse CDF: the first column whose running sum exceeds u * total = the number of columns whose running sum
        // does not (P >= 0, the running sum never falls), capped at the last column - an add, a compare and a counted
        // carry per column instead of a guarded select
        int below = 0;
        if constexpr (decltype(tag)::value) {
            float Pv[KC], total = 0.0f;  // every P_i once (registers), then the search
            be.each([&](int i, float e) {
                Pv[i] = (A * e + Bc) * ((i == cz) ? Qh : Qc);
                total += Pv[i];
            });
            const float target = u1 * total;
            float cum = 0.0f;
            be.each([&](int i, float) {
                cum += Pv[i];
                below += (cum <= target) ? 1 : 0;
            });
        } else {  // streamed block: sum_i (A e_i + Bc) Q_i in closed form (Q_i = Qc except at the current state), no pass
            const float total = A * (Qc * be.sum + (Qh - Qc) * be.at(cz)) + Bc * (Qc * (float)Kk + (Qh - Qc));
            const float target = u1 * total;
            float cum = 0.0f;
            be.each([&](int i, float e) {
                cum += (A * e + Bc) * ((i == cz) ? Qh : Qc);
                below += (cum <= target) ? 1 : 0;
            });
        }
        arg = below < Kk - 1 ? below : Kk - 1;
    };
    if (Kk <= KC) draw(CachedTag{}); else draw(StreamTag{});
    return arg;
}

template <int KC>
__global__ __launch_bounds__(kBlock) void k_p_sample_step_fast(StepArgs a) {
    const Staged s = stage_tables<false>(a.d);
    // seed / first global row: kernel arguments, or - for launches replayed from a HIP graph - a device pair (mtk_ctx_set_seed_source)
    const uint64_t seed_ = a.seed_src ? a.seed_src[0] : a.seed, row0_ = a.seed_src ? a.seed_src[1] : a.row0;
    const int Nn = a.d.Nn, C = a.d.C, S = a.d.S, t = a.t;
    const uint32_t stream_n = MTK_STREAM_STEP + 2u * (uint32_t)t, stream_u = stream_n + 1u;
    constexpr int G = 4;  // numeric columns per load group (fp64 state: 2 registers each)
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const float* mo = a.mo + row * a.mo_rs;
        float* xo = a.x_in ? a.x_in + row * a.xi_rs : nullptr;
        PhiloxNF pn(seed_, row0_ + row, stream_n);
        PhiloxW pu(seed_, row0_ + row, stream_u);
        // the first categorical block's inputs go out before the numeric part is consumed
        BlockRaw<KC> raw;
        int cz = 0;
        if (C > 0 && !a.u) {
            raw.load(mo + (int64_t)Nn * a.mo_cs, a.mo_cs, s.off[1]);
            cz = a.z_idx[row];
        }
        for (int j0 = 0; j0 < Nn; j0 += G) {  // the numeric state stays fp64 (SURVEY finding 0.4)
            float eh[G];
            double zv[G], nv[G];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int j = j0 + i;
                eh[i] = (j < Nn) ? mo[j * a.mo_cs] : 0.0f;
                zv[i] = (j < Nn) ? a.z[(int64_t)j * a.B + row] : 0.0;
                nv[i] = 0.0;
                if (a.noise && j < Nn)
                    nv[i] = a.noise_f64 ? reinterpret_cast<const double*>(a.noise)[row * Nn + j]
                                        : (double)reinterpret_cast<const float*>(a.noise)[row * Nn + j];
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int j = j0 + i;
                if (j < Nn) {
                    const double zj = zv[i];
                    const double nz = a.noise ? nv[i] : (double)pn.normal(j);
                    double mean;
                    if (a.first) {
                        const float zf = (float)zj;
                        const float x0h = a.d.gx0 ? eh[i] : a.sqrt_recip * zf - a.sqrt_recipm1 * eh[i];
                        mean = (double)(a.coef1 * x0h + a.coef2 * zf);
                    } else {
                        const double x0h = (double)a.sqrt_recip * zj - (double)(a.sqrt_recipm1 * eh[i]);
                        mean = (a.d.gx0 ? (double)(a.coef1 * eh[i]) : (double)a.coef1 * x0h) + (double)a.coef2 * zj;
                    }
                    if (a.grad) mean = (double)(float)mean + a.variance * (double)a.grad[row * Nn + j];
                    const double zn = mean + ((t != 0) ? a.sigma : 0.0) * nz;
                    a.z[(int64_t)j * a.B + row] = zn;
                    if (xo) xo[j * a.xi_cs] = (float)zn;
                }
            }
        }
        if (a.u) {  // injected uniforms: Gumbel arg-max over the unnormalised log posterior
            for (int k = 0; k < C; ++k) {
                const int o = s.off[k], Kk = s.off[k + 1] - o;
                const float logK = s.logK[k];
                const int czk = a.z_idx[(int64_t)k * a.B + row];
                const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
                const float cq = a.log_1m_alpha - logK;
                const float ce = a.log_1m_acp_prev - logK;
                float mx = -INFINITY;
                for (int i = 0; i < Kk; ++i) mx = fmaxf(mx, mc[i * a.mo_cs]);
                const float q_hot = lae_f(0.0f + a.log_alpha, cq), q_cold = lae_f(MTK_LOG_TINY + a.log_alpha, cq);
                float sum = 0.0f;
                for (int i = 0; i < Kk; ++i) sum += fexp(mc[i * a.mo_cs] - mx);
                const float lsum = flog(sum);
                float best = -INFINITY;
                int arg = 0;
                for (int i = 0; i < Kk; ++i) {
                    const float lsm = (mc[i * a.mo_cs] - mx) - lsum;
                    const float ev = (t == 0) ? lsm : lae_f(lsm + a.log_acp_prev, ce);
                    const float v = gumbel_fast(a.u[row * S + o + i]) + (a.d.direct ? lsm : ev + ((i == czk) ? q_hot : q_cold));
                    if (v > best || i == 0) { best = v; arg = i; }
                }
                a.z_idx[(int64_t)k * a.B + row] = arg;
                if (xo)
                    for (int i = 0; i < Kk; ++i) xo[(int64_t)(Nn + o + i) * a.xi_cs] = (i == arg) ? 0.0f : MTK_LOG_TINY;
            }
            continue;
        }
        // in-kernel noise: one Philox word per block, inverse CDF of P_i = (A e_i + Bc) Q_i
        const float acp1 = (t == 0) ? 1.0f : fexp(a.log_acp_prev);
        const float al = fexp(a.log_alpha);
        for (int k = 0; k < C; ++k) {
            const int o = s.off[k], Kk = s.off[k + 1] - o;
            const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
            BlockRaw<KC> nxt;
            int czn = 0;
            if (k + 1 < C) {
                nxt.load(mo + (int64_t)(Nn + o + Kk) * a.mo_cs, a.mo_cs, s.off[k + 2] - (o + Kk));
                czn = a.z_idx[(int64_t)(k + 1) * a.B + row];
            }
            const int arg = step_draw_block<KC>(raw, mc, a.mo_cs, Kk, cz, acp1, al, s.logK[k], a.log_1m_alpha, a.log_1m_acp_prev, t, pu.uniform(k), a.d.direct != 0);
            a.z_idx[(int64_t)k * a.B + row] = arg;
            if (xo) {
                float* xc = xo + (int64_t)(Nn + o) * a.xi_cs;
                for (int i = 0; i < Kk; ++i) xc[i * a.xi_cs] = (i == arg) ? 0.0f : MTK_LOG_TINY;
            }
            raw = nxt;
            cz = czn;
        }
    }
}

// The same step with one lane per (record, item), item = a numeric column or a categorical block (blockIdx.y), for the batch
// sizes the reference's callers sample (1 k - 20 k rows per call, SYNTH:79): a lane-per-record launch of 1,024 rows is four
// workgroups whose lanes each walk 30 items in a chain (22 us); spread over the items the chain is one item (k_q_sample_items
// is the same idea).  In-kernel noise or injected Gaussian noise; injected uniforms stay with the kernel above.  Same
// expressions, same draws: the two kernels' outputs are the same bits (tests/test_gpu_parity.py, MTK_STEP_ROWS).
template <int KC>
__global__ __launch_bounds__(kBlock) void k_p_sample_step_items(StepArgs a) {
    const uint64_t seed_ = a.seed_src ? a.seed_src[0] : a.seed, row0_ = a.seed_src ? a.seed_src[1] : a.row0;
    const int Nn = a.d.Nn, t = a.t;
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (row >= a.B) return;
    const int item = blockIdx.y;
    const uint32_t stream_n = MTK_STREAM_STEP + 2u * (uint32_t)t, stream_u = stream_n + 1u;
    const float* mo = a.mo + row * a.mo_rs;
    float* xo = a.x_in ? a.x_in + row * a.xi_rs : nullptr;
    if (item < Nn) {
        const int j = item;
        const float eh = mo[j * a.mo_cs];
        const double zj = a.z[(int64_t)j * a.B + row];
        double nz;
        if (a.noise) {
            nz = a.noise_f64 ? reinterpret_cast<const double*>(a.noise)[row * Nn + j] : (double)reinterpret_cast<const float*>(a.noise)[row * Nn + j];
        } else {
            PhiloxNF pn(seed_, row0_ + row, stream_n);
            nz = (double)pn.normal(j);
        }
        double mean;
        if (a.first) {
            const float zf = (float)zj;
            const float x0h = a.d.gx0 ? eh : a.sqrt_recip * zf - a.sqrt_recipm1 * eh;
            mean = (double)(a.coef1 * x0h + a.coef2 * zf);
        } else {
            const double x0h = (double)a.sqrt_recip * zj - (double)(a.sqrt_recipm1 * eh);
            mean = (a.d.gx0 ? (double)(a.coef1 * eh) : (double)a.coef1 * x0h) + (double)a.coef2 * zj;
        }
        if (a.grad) mean = (double)(float)mean + a.variance * (double)a.grad[row * Nn + j];
        const double zn = mean + ((t != 0) ? a.sigma : 0.0) * nz;
        a.z[(int64_t)j * a.B + row] = zn;
        if (xo) xo[j * a.xi_cs] = (float)zn;
        return;
    }
    const int k = item - Nn;
    const int o = a.d.off[k], Kk = a.d.off[k + 1] - o;
    const float* mc = mo + (int64_t)(Nn + o) * a.mo_cs;
    BlockRaw<KC> raw;
    raw.load(mc, a.mo_cs, Kk);
    const int cz = a.z_idx[(int64_t)k * a.B + row];
    PhiloxW pu(seed_, row0_ + row, stream_u);
    const float acp1 = (t == 0) ? 1.0f : fexp(a.log_acp_prev);
    const float al = fexp(a.log_alpha);
    const int arg = step_draw_block<KC>(raw, mc, a.mo_cs, Kk, cz, acp1, al, a.d.logK[k], a.log_1m_alpha, a.log_1m_acp_prev, t, pu.uniform(k), a.d.direct != 0);
    a.z_idx[(int64_t)k * a.B + row] = arg;
    if (xo) {
        float* xc = xo + (int64_t)(Nn + o) * a.xi_cs;
        for (int i = 0; i < Kk; ++i) xc[i * a.xi_cs] = (i == arg) ? 0.0f : MTK_LOG_TINY;
    }
}

// ------------------------------------------------------------------------------------------------
// loss_rows + its gradient with respect to the denoiser output (a19: the head of the backward pass).
// Same quantities as k_loss_rows_fast (exact per-block log-sum-exp, closed-form true posterior); FAST selects
// hardware or libm transcendentals.  Per block, with lsm = log_softmax(logits), ev = q_pred(lsm, t-1),
// un = ev + q(x_t | .), p = softmax(un), w = true posterior:
//   t > 0:  d KL / d un_i = p_i sum(w) - w_i        t = 0:  d NLL / d un_i = p_i - [i = x_0]
//   d un_i / d lsm_i = s_i = exp(lsm_i + log acp_{t-1} - ev_i)  (1 at t = 0)
//   d loss / d logit_i = g_i - softmax(logits)_i sum_j g_j,  g_i = (d loss / d un_i) s_i
// `out` and `dout` are column-major [D][ld].
// ------------------------------------------------------------------------------------------------
struct LossGradArgs {
    DevTables d;
    int64_t B;
    const float* out; int64_t ld;
    const float* x0; int64_t x0_ld;
    const int32_t* xt_idx;
    const float* eps;
    const int64_t* t;
    float* loss_multi;
    float* loss_gauss;
    float* dout;
    float gscale, mscale;  // 2 / (Nn B_global),  T / (C B_global)
    uint64_t seed, row0;
    // Gaussian loss = the KL / decoder-NLL term of the variational bound (GaussianLossType.KL, GMD:431-484) instead of the MSE:
    // kl_tab = the [T][4] coefficient table of k_gauss_kl_rows (null: MSE), kscale = w_gauss / (Nn ln 2)
    const double* kl_tab;
    float kscale;
    uint16_t* dout16;  // optional bf16 copy of dout, same layout (the bf16 training iteration's head weight-gradient operand)
    // optional device pair {w_multi, w_gauss} that multiplies the two scales (mtk_train_backward_dev: the weights of the two loss
    // sums are an autograd engine's incoming gradients, device scalars the host never reads); null: the scales stand as they are
    const float* w_dev;
};
// derivative of DU:48-59's tanh approximation of the standard normal CDF
__device__ __forceinline__ double approx_normal_cdf_grad(double v) {
    const double k = 0.79788456080286535588, th = tanh(k * (v + 0.044715 * (v * v * v)));
    return 0.5 * k * (1.0 + 3.0 * 0.044715 * (v * v)) * (1.0 - th * th);
}
__device__ __forceinline__ double approx_normal_cdf_fwd(double v) {
    return 0.5 * (1.0 + tanh(0.79788456080286535588 * (v + 0.044715 * (v * v * v))));
}

template <bool FAST> __device__ __forceinline__ float ex(float x) { return FAST ? __expf(x) : expf(x); }
template <bool FAST> __device__ __forceinline__ float lg(float x) { return FAST ? __logf(x) : logf(x); }
template <bool FAST> __device__ __forceinline__ float lae_t(float a, float b) {
    const float m = fmaxf(a, b);
    return m + lg<FAST>(ex<FAST>(a - m) + ex<FAST>(b - m));
}

// One workgroup = 64 records x kLgWaves waves.  A lane-per-record loop over the C blocks is a chain of ~6 dependent
// memory passes per block (110 us at any batch size the trainer uses); here wave w takes the categorical blocks
// k = w, w + kLgWaves, ... and the numeric columns j = w, w + kLgWaves, ... of the same 64 records, and the per-record
// partial losses meet in LDS, summed in wave order (deterministic).
// Sixteen waves (round 3; eight before): at the trainer's batch sizes the launch is a handful of workgroups whose time is the longest
// wave's chain, so blocks go to the low waves (k = w) and numeric columns to the high ones (column j to wave kLgWaves - 1 - j mod
// kLgWaves): with ten blocks and twenty columns no wave carries more than one block plus one column (2,048 rows: 27 -> 17 us).
constexpr int kLgWaves = 16;

// Column loop of one categorical block.  KC > 0: the block's logits sit in KC registers (blocks of at most KC levels: the six passes
// below then cost one round trip to memory, not six, and the intermediate gradient never leaves the lane), the loop is unrolled KC
// times under a predicate; KC = 0: any width, every pass from memory.  (A macro, not a functor over lambdas: arrays captured by
// reference stayed in scratch memory.)
#define MTK_LG_COLS(i) _Pragma("clang loop unroll_count(KC > 0 ? KC : 1)") for (int i = 0; i < (KC > 0 ? KC : Kk); ++i) if (KC == 0 || i < Kk)
#define MTK_LG_COL(i) (KC > 0 ? v[KC > 0 ? (i) : 0] : mc[(i) * ld])

template <bool FAST, int KC, bool KL>
__global__ __launch_bounds__(64 * kLgWaves) void k_loss_grad(LossGradArgs a) {
    const Staged s = stage_tables<false>(a.d);
    __shared__ float part_g[kLgWaves][64], part_m[kLgWaves][64];
    const int Nn = a.d.Nn, C = a.d.C, T = a.d.T;
    const int64_t ld = a.ld;
    const SchedRow srT = s.rows[T - 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float wdm = a.w_dev ? a.w_dev[0] : 1.0f, wdg = a.w_dev ? a.w_dev[1] : 1.0f;
    const float gscale = a.gscale * wdg, kscale = a.kscale * wdg, mscale = a.mscale * wdm;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < a.B; base += (int64_t)gridDim.x * 64) {
        const int64_t row = base + lane;
        float accg = 0.0f, accm = 0.0f;
        if (row < a.B) {
            const int t = mtk_tclamp(a.t[row], a.d.T);
            const float* mo = a.out + row;
            float* go = a.dout + row;
            uint16_t* gh = a.dout16 ? a.dout16 + row : nullptr;
            PhiloxN pn(a.seed, a.row0 + row, MTK_STREAM_EPS);
            PhiloxNF pf(a.seed, a.row0 + row, MTK_STREAM_EPS);
            const SchedRow st = s.rows[t];
            const SchedRow s1 = s.rows[t > 0 ? t - 1 : 0];
            const float* xr = a.x0 + row * a.x0_ld;
            if (!KL) {
                for (int j = kLgWaves - 1 - w; j < Nn; j += kLgWaves) {
                    const float e = a.eps ? a.eps[row * Nn + j] : (FAST ? pf.normal(j) : pn.normal(j));
                    const float df = mo[j * ld] - e;
                    accg += df * df;
                    go[j * ld] = gscale * df;
                    put16(gh, j * ld, gscale * df);
                }
            } else {
                // _vb_terms_bpd (GMD:431-484) and its derivative with respect to the predicted noise: the model mean is
                // c1 (sr x_t - srm1 eps_hat) + c2 x_t, so d mean / d eps_hat = -c1 srm1 for every feature
                const double* tb = a.kl_tab + (size_t)t * 4;
                const float c1 = (float)tb[0], c2 = (float)tb[1];
                const double lv1 = (double)(float)tb[2], lv2 = tb[3];
                const double e2 = exp(-lv2), half_head = 0.5 * (((-1.0 + lv2) - lv1) + exp(lv1 - lv2)), inv_stdv = exp(-(0.5 * lv2));
                // GaussianParametrization.X0 (GMD:404-405): the output is x_0_hat itself, d mean / d out = c1
                const float slope = a.d.gx0 ? -c1 : c1 * st.sqrt_recipm1_acp;  // = -d mean / d out
                for (int j = kLgWaves - 1 - w; j < Nn; j += kLgWaves) {
                    const float e = a.eps ? a.eps[row * Nn + j] : (FAST ? pf.normal(j) : pn.normal(j));
                    const float xs = xr[j];
                    const float xj = st.sqrt_acp * xs + st.sqrt_1m_acp * e;   // GMD:302, as k_q_sample forms it
                    const float x0_hat = a.d.gx0 ? mo[j * ld] : st.sqrt_recip_acp * xj - st.sqrt_recipm1_acp * mo[j * ld];
                    const float mean = c1 * x0_hat + c2 * xj;
                    double term, grad;
                    if (t != 0) {
                        const float d = (c1 * xs + c2 * xj) - mean;
                        term = half_head + 0.5 * ((double)(d * d) * e2);
                        grad = (double)d * e2 * (double)slope;
                    } else {
                        const float centered = xs - mean;
                        const double up = inv_stdv * (double)(centered + (float)(1.0 / 255.0)), um = inv_stdv * (double)(centered - (float)(1.0 / 255.0));
                        const double cp = approx_normal_cdf_fwd(up), cm = approx_normal_cdf_fwd(um);
                        double pr, dpr;  // probability of the bin and its derivative with respect to `centered`
                        if (xs < -0.999f) { pr = cp; dpr = approx_normal_cdf_grad(up) * inv_stdv; }
                        else if (xs > 0.999f) { pr = 1.0 - cm; dpr = -approx_normal_cdf_grad(um) * inv_stdv; }
                        else { pr = cp - cm; dpr = (approx_normal_cdf_grad(up) - approx_normal_cdf_grad(um)) * inv_stdv; }
                        term = -log(fmax(pr, 1e-12));
                        grad = pr > 1e-12 ? -(dpr / pr) * (double)slope : 0.0;  // clamp(min = 1e-12) passes no gradient below it
                    }
                    accg += (float)term;
                    go[j * ld] = kscale * (float)grad;
                    put16(gh, j * ld, kscale * (float)grad);
                }
            }
            for (int k = w; k < C; k += kLgWaves) {
                const int o = s.off[k], Kk = s.off[k + 1] - o;
                const float logK = s.logK[k];
                const int c0 = (int)(long long)xr[Nn + k];
                const int ct = a.xt_idx[(int64_t)k * a.B + row];
                const float* mc = mo + (int64_t)(Nn + o) * ld;
                float* gc = go + (int64_t)(Nn + o) * ld;
                float v[KC > 0 ? KC : 1], gq[KC > 0 ? KC : 1];
                if (KC > 0) {
#pragma unroll
                    for (int i = 0; i < KC; ++i) v[i] = (i < Kk) ? mc[i * ld] : -INFINITY;
                }
                const float cq = st.log_1m_alpha - logK;
                const float q_hot = lae_t<FAST>(0.0f + st.log_alpha, cq), q_cold = lae_t<FAST>(MTK_LOG_TINY + st.log_alpha, cq);
                const float ce = s1.log_1m_acp - logK;
                const float e_hot = (t == 0) ? 0.0f : lae_t<FAST>(0.0f + s1.log_acp, ce);
                const float e_cold = (t == 0) ? MTK_LOG_TINY : lae_t<FAST>(MTK_LOG_TINY + s1.log_acp, ce);
                const float v_c0 = e_hot + ((c0 == ct) ? q_hot : q_cold);
                const float v_ct = e_cold + q_hot;
                const float v_cc = e_cold + q_cold;
                const int n_cc = Kk - ((c0 == ct) ? 1 : 2);
                const float m_t = fmaxf(v_c0, fmaxf(v_ct, v_cc));
                float b_t = ex<FAST>(v_c0 - m_t) + (float)n_cc * ex<FAST>(v_cc - m_t);
                if (c0 != ct) b_t += ex<FAST>(v_ct - m_t);
                const float lse_t = m_t + lg<FAST>(b_t);
                const float lt_c0 = v_c0 - lse_t, lt_ct = v_ct - lse_t, lt_cc = v_cc - lse_t;
                const float w_c0 = ex<FAST>(lt_c0), w_ct = ex<FAST>(lt_ct), w_cc = ex<FAST>(lt_cc);
                const float w_sum = w_c0 + (float)n_cc * w_cc + ((c0 != ct) ? w_ct : 0.0f);
                float mx = -INFINITY;
                MTK_LG_COLS(i) mx = fmaxf(mx, MTK_LG_COL(i));
                float sum = 0.0f;
                MTK_LG_COLS(i) sum += ex<FAST>(MTK_LG_COL(i) - mx);
                const float lsum = lg<FAST>(sum);
                // Parametrization.DIRECT (GMD:828-829): the model side is the log-softmax itself (un = lsm, d un / d lsm = 1)
                const bool direct = a.d.direct != 0, plain = (t == 0) || direct;
                const float qm_hot = direct ? 0.0f : q_hot, qm_cold = direct ? 0.0f : q_cold;
                const float ev_top = plain ? -lsum : lae_t<FAST>(-lsum + s1.log_acp, ce);
                const float m_m = ev_top + qm_hot;
                float b_m = 0.0f, acc_kl = 0.0f, un_c0 = 0.0f;
                MTK_LG_COLS(i) {
                    const float lsm = (MTK_LG_COL(i) - mx) - lsum;
                    const float ev = plain ? lsm : lae_t<FAST>(lsm + s1.log_acp, ce);
                    const float un = ev + ((i == ct) ? qm_hot : qm_cold);
                    b_m += ex<FAST>(un - m_m);
                    const float lt = (i == c0) ? lt_c0 : ((i == ct) ? lt_ct : lt_cc);
                    const float w = (i == c0) ? w_c0 : ((i == ct) ? w_ct : w_cc);
                    acc_kl += w * (lt - un);
                    if (i == c0) un_c0 = un;
                }
                const float lse_m = m_m + lg<FAST>(b_m);
                const float cT = srT.log_1m_acp - logK;
                const float p_hot = lae_t<FAST>(0.0f + srT.log_acp, cT), p_cold = lae_t<FAST>(MTK_LOG_TINY + srT.log_acp, cT);
                const float prior_k = ex<FAST>(p_hot) * (p_hot + logK) + (float)(Kk - 1) * (ex<FAST>(p_cold) * (p_cold + logK));
                accm += ((t == 0) ? -(un_c0 - lse_m) : (acc_kl + lse_m * w_sum)) * (float)T + prior_k;
                // gradient: first pass keeps g_i and accumulates their sum, second pass projects through the softmax
                float G = 0.0f;
                MTK_LG_COLS(i) {
                    const float lsm = (MTK_LG_COL(i) - mx) - lsum;
                    const float ev = plain ? lsm : lae_t<FAST>(lsm + s1.log_acp, ce);
                    const float un = ev + ((i == ct) ? qm_hot : qm_cold);
                    const float p = ex<FAST>(un - lse_m);
                    const float w = (i == c0) ? w_c0 : ((i == ct) ? w_ct : w_cc);
                    const float gun = (t == 0) ? (p - ((i == c0) ? 1.0f : 0.0f)) : (p * w_sum - w);
                    const float sl = plain ? 1.0f : ex<FAST>((lsm + s1.log_acp) - ev);
                    const float gi = gun * sl;
                    G += gi;
                    if (KC > 0) gq[KC > 0 ? i : 0] = gi;
                    else gc[i * ld] = gi;
                }
                MTK_LG_COLS(i) {
                    const float sm = ex<FAST>((MTK_LG_COL(i) - mx) - lsum);
                    const float gi = KC > 0 ? gq[KC > 0 ? i : 0] : gc[i * ld];
                    const float gv = mscale * (gi - sm * G);
                    gc[i * ld] = gv;
                    put16(gh, (int64_t)(Nn + o + i) * ld, gv);
                }
            }
        }
        part_g[w][lane] = accg;
        part_m[w][lane] = accm;
        __syncthreads();
        if (w == 0 && row < a.B) {
            float g = 0.0f, m = 0.0f;
#pragma unroll
            for (int i = 0; i < kLgWaves; ++i) { g += part_g[i][lane]; m += part_m[i][lane]; }
            if (a.loss_gauss && Nn > 0) a.loss_gauss[row] = KL ? (g / (float)Nn) * 1.44269504088896340736f : g / (float)Nn;
            if (a.loss_multi && C > 0) a.loss_multi[row] = m / (float)C;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// KL Gaussian loss (a9, GaussianLossType.KL): _vb_terms_bpd GMD:431-484 with normal_kl DU:9-45 and
// discretized_gaussian_log_likelihood DU:62-91, EPS parametrisation.  Same promotion as the reference: means and the
// clipped posterior log-variance are fp32 expressions, the model log-variance is fp64, so everything it touches is.
// One lane per record; what depends on t alone (two exponentials, the inverse std-dev) is formed once per record.
// ------------------------------------------------------------------------------------------------
struct KlArgs {
    int64_t B;
    int Nn, T, gx0;
    const SchedRow* rows;
    const double* tab;  // [T][4]: posterior_mean_coef1, posterior_mean_coef2, posterior_log_variance_clipped, model log-variance
    const float* mo; int64_t mo_rs, mo_cs;
    const float* x0; int64_t x0_ld;
    const float* xt; int64_t xt_rs, xt_cs;
    const int64_t* t;
    double* loss;
};

__device__ __forceinline__ double approx_normal_cdf(double v) {  // DU:48-59
    return 0.5 * (1.0 + tanh(0.79788456080286535588 * (v + 0.044715 * (v * v * v))));
}

__global__ __launch_bounds__(kBlock) void k_gauss_kl_rows(KlArgs a) {
    const double inv_ln2 = 1.0 / 0.69314718055994530942;
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const int t = mtk_tclamp(a.t[row], a.T);
        const SchedRow sr = a.rows[t];
        const double* tb = a.tab + (size_t)t * 4;
        const float c1 = (float)tb[0], c2 = (float)tb[1], lv1 = (float)tb[2];
        const double lv2 = tb[3];
        const float* mo = a.mo + row * a.mo_rs;
        const float* xt = a.xt + row * a.xt_rs;
        const float* x0 = a.x0 + row * a.x0_ld;
        double acc = 0.0;
        if (t != 0) {
            const double head = (-1.0 + lv2) - (double)lv1;
            const double e1 = exp((double)lv1 - lv2), e2 = exp(-lv2);
            for (int j = 0; j < a.Nn; ++j) {
                const float xs = x0[j], xj = xt[j * a.xt_cs], ej = mo[j * a.mo_cs];
                const float true_mean = c1 * xs + c2 * xj;
                const float x0_hat = a.gx0 ? ej : sr.sqrt_recip_acp * xj - sr.sqrt_recipm1_acp * ej;
                const float mean = c1 * x0_hat + c2 * xj;
                const float d = true_mean - mean;
                acc += 0.5 * ((head + e1) + (double)(d * d) * e2);
            }
        } else {
            const double inv_stdv = exp(-(0.5 * lv2));
            for (int j = 0; j < a.Nn; ++j) {
                const float xs = x0[j], xj = xt[j * a.xt_cs], ej = mo[j * a.mo_cs];
                const float x0_hat = a.gx0 ? ej : sr.sqrt_recip_acp * xj - sr.sqrt_recipm1_acp * ej;
                const float mean = c1 * x0_hat + c2 * xj;
                const float centered = xs - mean;
                const double cdf_plus = approx_normal_cdf(inv_stdv * (double)(centered + (float)(1.0 / 255.0)));
                const double cdf_min = approx_normal_cdf(inv_stdv * (double)(centered - (float)(1.0 / 255.0)));
                double p;
                if (xs < -0.999f) p = cdf_plus;
                else if (xs > 0.999f) p = 1.0 - cdf_min;
                else p = cdf_plus - cdf_min;
                acc -= log(fmax(p, 1e-12));
            }
        }
        a.loss[row] = acc / (double)a.Nn * inv_ln2;
    }
}

// rows whose timestep lies outside [0, T): every output element of the row becomes NaN (see mtk_tclamp)
struct PoisonArgs {
    const int64_t* t; int64_t B; int T;
    float* p0; int64_t rs0, cs0; int n0;
    float* p1; int64_t rs1, cs1; int n1;
    double* p2;
};
__global__ __launch_bounds__(kBlock) void k_poison_bad_t(PoisonArgs a) {
    for (int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.B; row += (int64_t)gridDim.x * kBlock) {
        const int64_t tv = a.t[row];
        if (tv >= 0 && tv < a.T) continue;
        const float qn = __builtin_nanf("");
        if (a.p0) for (int j = 0; j < a.n0; ++j) a.p0[row * a.rs0 + j * a.cs0] = qn;
        if (a.p1) for (int j = 0; j < a.n1; ++j) a.p1[row * a.rs1 + j * a.cs1] = qn;
        if (a.p2) a.p2[row] = __builtin_nan("");
    }
}

int max_levels(const mtk_ctx* c) {  // widest categorical block of the table
    int m = 0;
    for (int32_t k : c->K) m = k > m ? k : m;
    return m;
}
// cap = 2048: 256 CUs x 8 resident workgroups, grid-stride covers the rest.  The record-staging variant of q_sample (no
// per-workgroup copy of the schedule) takes 16384: one pass per workgroup up to 4 M rows, the dispatcher balances the tail
// (rocprof average at 2^20 rows: 0.150 -> 0.143 ms).  loss_rows and p_sample_step keep 2048: caps of 1024, 4096 and 16384
// measured the same within 1 % at 2^20 rows (1280, an uneven 3.2 rows per lane, 12 % slower); neither occupancy, nor the
// prefetch depth, nor the VALU work moves these kernels off 4.4-4.7 TB/s - 120 column streams 4 MB apart per wave.
int grid_for(int64_t B, int cap = 2048) {
    int64_t g = (B + kBlock - 1) / kBlock;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
size_t lds_bytes(const mtk_ctx* c, bool rows) {
    size_t b = rows ? (size_t)c->T * sizeof(SchedRow) : 0;
    b += (((size_t)(c->C + 1) * 4 + 15) & ~(size_t)15) + (size_t)(c->C > 0 ? c->C : 1) * 4;
    return (b + 15) & ~(size_t)15;
}

}  // namespace

int mtk_poison_bad_t(const int64_t* t, int64_t B, int T, float* p0, int64_t rs0, int64_t cs0, int n0, float* p1, int64_t rs1,
                     int64_t cs1, int n1, double* p2, hipStream_t s) {
    if (!t || B <= 0 || (!p0 && !p1 && !p2)) return MTK_OK;
    PoisonArgs a{t, B, T, p0, rs0, cs0, n0, p1, rs1, cs1, n1, p2};
    hipLaunchKernelGGL(k_poison_bad_t, dim3(grid_for(B)), dim3(kBlock), 0, s, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// per-timestep coefficients of the KL Gaussian loss, built on first use (synchronous copy)
int mtk_ensure_kl_tab(mtk_ctx* ctx) {
    if (ctx->d_kl) return MTK_OK;
    std::vector<double> h((size_t)ctx->T * 4);
    for (int i = 0; i < ctx->T; ++i) {
        h[(size_t)i * 4 + 0] = ctx->tab(MTK_TAB_POST_COEF1, i);
        h[(size_t)i * 4 + 1] = ctx->tab(MTK_TAB_POST_COEF2, i);
        h[(size_t)i * 4 + 2] = ctx->tab(MTK_TAB_POST_LOGVAR_CLIPPED, i);
        h[(size_t)i * 4 + 3] = ctx->tab(MTK_TAB_MODEL_LOGVAR64, i);
    }
    MTK_HIP(hipMalloc(&ctx->d_kl, h.size() * sizeof(double)));
    MTK_HIP(hipMemcpy(ctx->d_kl, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return MTK_OK;
}

// internal (mtk_train.hip): per-record losses and d(mean loss)/d(out); out / dout column-major [D][ld]
int mtk_launch_loss_grad(mtk_ctx* ctx, int64_t B, const float* out, int64_t ld, const float* x0, int64_t x0_ld,
                         const int32_t* xt_idx, const float* eps, const int64_t* t, float* loss_multi, float* loss_gauss,
                         float* dout, double w_multi, double w_gauss, const float* w_dev, uint64_t seed, uint64_t row0, hipStream_t stream,
                         uint16_t* dout16) {
    // d(w_gauss * sum_rows loss_gauss + w_multi * sum_rows loss_multi) / d out  (w_dev: both weights times the device pair)
    LossGradArgs a{ctx->dev(), B, out, ld, x0, x0_ld, xt_idx, eps, t, loss_multi, loss_gauss, dout,
                   ctx->Nn > 0 ? (float)(2.0 * w_gauss / ctx->Nn) : 0.0f,
                   ctx->C > 0 ? (float)((double)ctx->T * w_multi / ctx->C) : 0.0f, seed, row0, nullptr, 0.0f, dout16, w_dev};
    if (ctx->train_gauss_kl && ctx->Nn > 0) {
        const int rc = mtk_ensure_kl_tab(ctx);
        if (rc) return rc;
        a.kl_tab = ctx->d_kl;
        a.kscale = (float)(w_gauss / ((double)ctx->Nn * 0.69314718055994530942));
    }
    int64_t grid = (B + 63) / 64;
    if (grid > 2048) grid = 2048;
    // blocks of at most kKCache levels are held in registers (MTK_LG_STREAM: the from-memory loops, the A/B switch)
    const int ml = max_levels(ctx);
    const int kc = getenv("MTK_LG_STREAM") ? 0 : (ml <= kKCacheSmall ? kKCacheSmall : (ml <= kKCache ? kKCache : 0));
    const bool kl = a.kl_tab != nullptr;
    void (*kern)(LossGradArgs);
    if (ctx->fast_math) {
        if (kl) kern = kc == kKCacheSmall ? k_loss_grad<true, kKCacheSmall, true> : (kc == kKCache ? k_loss_grad<true, kKCache, true> : k_loss_grad<true, 0, true>);
        else kern = kc == kKCacheSmall ? k_loss_grad<true, kKCacheSmall, false> : (kc == kKCache ? k_loss_grad<true, kKCache, false> : k_loss_grad<true, 0, false>);
    } else {
        if (kl) kern = kc == kKCacheSmall ? k_loss_grad<false, kKCacheSmall, true> : (kc == kKCache ? k_loss_grad<false, kKCache, true> : k_loss_grad<false, 0, true>);
        else kern = kc == kKCacheSmall ? k_loss_grad<false, kKCacheSmall, false> : (kc == kKCache ? k_loss_grad<false, kKCache, false> : k_loss_grad<false, 0, false>);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * kLgWaves), lds_bytes(ctx, false), stream, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_q_sample(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                            const float* u, float* x_in, int64_t xi_rs, int64_t xi_cs, int32_t* xt_idx, uint64_t seed,
                            uint64_t row0, mtk_stream stream) {
    if (ctx && B > 0 && !t) MTK_FAIL(MTK_ERR_ARG, "mtk_q_sample: null pointer or negative B");
    return mtk_q_sample_t(ctx, B, x0, x0_ld, t, eps, u, x_in, xi_rs, xi_cs, xt_idx, seed, row0, nullptr, nullptr, (hipStream_t)stream);
}

int mtk_q_sample_t(mtk_ctx* ctx, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps, const float* u, float* x_in,
                   int64_t xi_rs, int64_t xi_cs, int32_t* xt_idx, uint64_t seed, uint64_t row0, int64_t* t_out, uint16_t* x16, hipStream_t stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_q_sample: null context");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !x0 || (!t && !t_out) || !x_in) MTK_FAIL(MTK_ERR_ARG, "mtk_q_sample: null pointer or negative B");
    if (ctx->C > 0 && !xt_idx) MTK_FAIL(MTK_ERR_ARG, "mtk_q_sample: xt_idx is required when C > 0");
    if (x0_ld < ctx->Nn + ctx->C) MTK_FAIL(MTK_ERR_ARG, "mtk_q_sample: x0_ld %lld < Nn + C = %d", (long long)x0_ld, ctx->Nn + ctx->C);
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    QSampleArgs a{ctx->dev(), B, x0, x0_ld, t, eps, u, x_in, xi_rs, xi_cs, xt_idx, seed, row0, t_out, x16};
    const bool lds = ctx->T <= MTK_LDS_T_MAX && !getenv("MTK_NO_LDS_ROWS");
    // fast math: stage the row-major records through LDS when they are narrow and 16-byte aligned (the slab
    // replaces the LDS copy of the schedule rows, which then come from L2: measured equal)
    const bool stage = ctx->fast_math && x0_ld <= kStageLd && (reinterpret_cast<uintptr_t>(x0) & 15) == 0 && !getenv("MTK_NO_STAGE");
    // training-sized batches: one lane per (record, item); MTK_QS_ROWS keeps the lane-per-record kernels (the A/B switch of
    // tests/test_gpu_train.py, which checks that the two give the same bits)
    if (ctx->fast_math && !u && B <= kItemsMaxRows && ctx->Nn + ctx->C > 0 && ctx->Nn + ctx->C <= 65535 && !getenv("MTK_QS_ROWS")) {
        hipLaunchKernelGGL(k_q_sample_items, dim3((unsigned)((B + kBlock - 1) / kBlock), (unsigned)(ctx->Nn + ctx->C)), dim3(kBlock), 0,
                           (hipStream_t)stream, a);
        MTK_HIP(hipGetLastError());
        return MTK_OK;
    }
    if (stage) {
        hipLaunchKernelGGL((k_q_sample_fast<false, true>), dim3(grid_for(B, 16384)), dim3(kBlock),
                           lds_bytes(ctx, false) + (size_t)kBlock * x0_ld * sizeof(float), (hipStream_t)stream, a);
        MTK_HIP(hipGetLastError());
        return MTK_OK;  // the fast kernels poison records with an out-of-range timestep themselves
    }
    auto k = ctx->fast_math ? (lds ? k_q_sample_fast<true, false> : k_q_sample_fast<false, false>) : (lds ? k_q_sample<true> : k_q_sample<false>);
    hipLaunchKernelGGL(k, dim3(grid_for(B)), dim3(kBlock), lds_bytes(ctx, lds), (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    if (ctx->fast_math || t_out) return MTK_OK;  // drawn timesteps are in range
    return mtk_poison_bad_t(t, B, ctx->T, x_in, xi_rs, xi_cs, ctx->D, nullptr, 0, 0, 0, nullptr, (hipStream_t)stream);
}

extern "C" int mtk_loss_rows(mtk_ctx* ctx, int64_t B, const float* model_out, int64_t mo_rs, int64_t mo_cs, const float* x0,
                             int64_t x0_ld, const int32_t* xt_idx, const float* eps, const int64_t* t, float* loss_multi,
                             float* loss_gauss, uint64_t seed, uint64_t row0, mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_loss_rows: null context");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !model_out || !t) MTK_FAIL(MTK_ERR_ARG, "mtk_loss_rows: null pointer or negative B");
    if (loss_multi && ctx->C > 0 && (!x0 || !xt_idx)) MTK_FAIL(MTK_ERR_ARG, "mtk_loss_rows: x0 and xt_idx are required for the multinomial loss");
    if (x0 && x0_ld < ctx->Nn + ctx->C) MTK_FAIL(MTK_ERR_ARG, "mtk_loss_rows: x0_ld too small");
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    LossArgs a{ctx->dev(), B, model_out, mo_rs, mo_cs, x0, x0_ld, xt_idx, eps, t, loss_multi, loss_gauss, seed, row0};
    const bool lds = ctx->T <= MTK_LDS_T_MAX && !getenv("MTK_NO_LDS_ROWS");
    const bool stage = ctx->fast_math && loss_multi && ctx->C > 0 && x0_ld <= kStageLd && (reinterpret_cast<uintptr_t>(x0) & 15) == 0 && !getenv("MTK_NO_STAGE");
    const bool small = max_levels(ctx) <= kKCacheSmall;
    if (stage) {  // as in mtk_q_sample
        auto ks = small ? k_loss_rows_fast_small<false, true> : k_loss_rows_fast<false, true>;
        hipLaunchKernelGGL(ks, dim3(grid_for(B)), dim3(kBlock),
                           lds_bytes(ctx, false) + (size_t)kBlock * x0_ld * sizeof(float), (hipStream_t)stream, a);
        MTK_HIP(hipGetLastError());
        return MTK_OK;  // poisoned in the kernel
    }
    auto kf = small ? (lds ? k_loss_rows_fast_small<true, false> : k_loss_rows_fast_small<false, false>)
                    : (lds ? k_loss_rows_fast<true, false> : k_loss_rows_fast<false, false>);
    auto k = ctx->fast_math ? kf : (lds ? k_loss_rows<true> : k_loss_rows<false>);
    hipLaunchKernelGGL(k, dim3(grid_for(B)), dim3(kBlock), lds_bytes(ctx, lds), (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    if (ctx->fast_math) return MTK_OK;  // poisoned in the kernel
    return mtk_poison_bad_t(t, B, ctx->T, ctx->C > 0 ? loss_multi : nullptr, 1, 0, 1, nullptr, 0, 0, 0, nullptr, (hipStream_t)stream);  // the MSE term does not depend on t
}

extern "C" int mtk_p_sample_step_guided(mtk_ctx* ctx, int64_t B, int t, int first_step, const float* model_out, int64_t mo_rs,
                                        int64_t mo_cs, double* z, int32_t* z_idx, const void* noise, int noise_is_f64,
                                        const float* u, const float* grad, float* x_in_next, int64_t xi_rs, int64_t xi_cs,
                                        uint64_t seed, uint64_t row0, mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_p_sample_step: null context");
    if (t < 0 || t >= ctx->T) MTK_FAIL(MTK_ERR_ARG, "mtk_p_sample_step: t = %d outside [0, %d)", t, ctx->T);
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !model_out || (ctx->Nn > 0 && !z) || (ctx->C > 0 && !z_idx)) MTK_FAIL(MTK_ERR_ARG, "mtk_p_sample_step: null pointer or negative B");
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    const int tp = t > 0 ? t - 1 : 0;
    StepArgs a;
    a.d = ctx->dev();
    a.B = B; a.t = t; a.first = first_step;
    a.sqrt_recip = (float)ctx->tab(MTK_TAB_SQRT_RECIP_ACP, t);
    a.sqrt_recipm1 = (float)ctx->tab(MTK_TAB_SQRT_RECIPM1_ACP, t);
    a.coef1 = (float)ctx->tab(MTK_TAB_POST_COEF1, t);
    a.coef2 = (float)ctx->tab(MTK_TAB_POST_COEF2, t);
    a.log_alpha = (float)ctx->tab(MTK_TAB_LOG_ALPHA, t);
    a.log_1m_alpha = (float)ctx->tab(MTK_TAB_LOG_1M_ALPHA, t);
    a.log_acp_prev = (float)ctx->tab(MTK_TAB_LOG_ACP, tp);
    a.log_1m_acp_prev = (float)ctx->tab(MTK_TAB_LOG_1M_ACP, tp);
    a.sigma = exp(0.5 * ctx->tab(MTK_TAB_MODEL_LOGVAR64, t));
    a.variance = ctx->tab(MTK_TAB_MODEL_VAR64, t);
    a.grad = grad;
    a.mo = model_out; a.mo_rs = mo_rs; a.mo_cs = mo_cs;
    a.z = z; a.z_idx = z_idx; a.noise = noise; a.noise_f64 = noise_is_f64; a.u = u;
    a.x_in = x_in_next; a.xi_rs = xi_rs; a.xi_cs = xi_cs; a.seed = seed; a.row0 = row0; a.seed_src = ctx->seed_src;
    // caller-sized batches: one lane per (record, item); MTK_STEP_ROWS keeps the lane-per-record kernel (A/B switch of the tests)
    if (ctx->fast_math && !u && B <= kItemsMaxRows && ctx->Nn + ctx->C > 0 && ctx->Nn + ctx->C <= 65535 && !getenv("MTK_STEP_ROWS")) {
        auto kitems = max_levels(ctx) <= kKCacheSmall ? k_p_sample_step_items<kKCacheSmall> : k_p_sample_step_items<kKCache>;
        hipLaunchKernelGGL(kitems, dim3((unsigned)((B + kBlock - 1) / kBlock), (unsigned)(ctx->Nn + ctx->C)), dim3(kBlock), 0, (hipStream_t)stream, a);
        MTK_HIP(hipGetLastError());
        return MTK_OK;
    }
    auto kstep = !ctx->fast_math ? k_p_sample_step : (max_levels(ctx) <= kKCacheSmall ? k_p_sample_step_fast<kKCacheSmall> : k_p_sample_step_fast<kKCache>);
    hipLaunchKernelGGL(kstep, dim3(grid_for(B)), dim3(kBlock), lds_bytes(ctx, false),
                       (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_p_sample_step(mtk_ctx* ctx, int64_t B, int t, int first_step, const float* model_out, int64_t mo_rs,
                                 int64_t mo_cs, double* z, int32_t* z_idx, const void* noise, int noise_is_f64,
                                 const float* u, float* x_in_next, int64_t xi_rs, int64_t xi_cs, uint64_t seed,
                                 uint64_t row0, mtk_stream stream) {
    return mtk_p_sample_step_guided(ctx, B, t, first_step, model_out, mo_rs, mo_cs, z, z_idx, noise, noise_is_f64, u, nullptr,
                                    x_in_next, xi_rs, xi_cs, seed, row0, stream);
}

extern "C" int mtk_ddim_step_ex(mtk_ctx* ctx, int64_t B, int t, double eta, const float* model_out, int64_t mo_rs, int64_t mo_cs,
                                double* z, int32_t* z_idx, const float* noise, const float* u, const float* grad, float* x_in_next,
                                int64_t xi_rs, int64_t xi_cs, uint64_t seed, uint64_t row0, mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_step: null context");
    if (t < 0 || t >= ctx->T) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_step: t = %d outside [0, %d)", t, ctx->T);
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !model_out || (ctx->Nn > 0 && !z) || (ctx->C > 0 && !z_idx)) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_step: null pointer or negative B");
    if (!(eta >= 0.0)) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_step: eta = %g must be >= 0", eta);
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    DdimArgs a;
    a.d = ctx->dev();
    a.B = B; a.t = t;
    a.sqrt_recip = (float)ctx->tab(MTK_TAB_SQRT_RECIP_ACP, t);
    a.sqrt_recipm1 = (float)ctx->tab(MTK_TAB_SQRT_RECIPM1_ACP, t);
    // GMD:1109-1111 / 1239-1245 in fp32 torch ops (this file is built without fma contraction, host side included)
    const float ab = (float)ctx->tab(MTK_TAB_ACP, t), abp = (float)ctx->tab(MTK_TAB_ACP_PREV, t);
    const float sigma = ((float)eta * sqrtf((1.0f - abp) / (1.0f - ab))) * sqrtf(1.0f - ab / abp);
    const float coef1 = sigma, coef2 = abp - sigma * ab, coef3 = (1.0f - coef1) - coef2;
    a.c_x0 = sqrtf(abp);
    a.c_eps = sqrtf((1.0f - abp) - sigma * sigma);
    a.sig_mask = (t != 0 ? 1.0f : 0.0f) * sigma;
    a.sqrt_1m_ab = sqrtf(1.0f - ab);
    a.log_c1 = coef1 > 0.0f ? logf(coef1) : -INFINITY;
    a.log_c2 = logf(coef2);
    a.log_c3 = coef3 > 0.0f ? logf(coef3) : (coef3 == 0.0f ? -INFINITY : NAN);
    a.mo = model_out; a.mo_rs = mo_rs; a.mo_cs = mo_cs;
    a.z = z; a.z_idx = z_idx; a.noise = noise; a.grad = grad; a.u = u;
    a.x_in = x_in_next; a.xi_rs = xi_rs; a.xi_cs = xi_cs; a.seed = seed; a.row0 = row0; a.seed_src = ctx->seed_src;
    hipLaunchKernelGGL(k_ddim_step, dim3(grid_for(B)), dim3(kBlock), lds_bytes(ctx, false), (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_ddim_step(mtk_ctx* ctx, int64_t B, int t, const float* model_out, int64_t mo_rs, int64_t mo_cs, double* z,
                             int32_t* z_idx, const float* u, float* x_in_next, int64_t xi_rs, int64_t xi_cs, uint64_t seed,
                             uint64_t row0, mtk_stream stream) {
    return mtk_ddim_step_ex(ctx, B, t, 0.0, model_out, mo_rs, mo_cs, z, z_idx, nullptr, u, nullptr, x_in_next, xi_rs, xi_cs, seed, row0, stream);
}

extern "C" int mtk_ddim_reverse_step(mtk_ctx* ctx, int64_t B, int t, const float* model_out, int64_t mo_rs, int64_t mo_cs, double* z,
                                     float* x_in_next, int64_t xi_rs, int64_t xi_cs, mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_reverse_step: null context");
    if (t < 0 || t >= ctx->T) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_reverse_step: t = %d outside [0, %d)", t, ctx->T);
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !model_out || (ctx->Nn > 0 && !z)) MTK_FAIL(MTK_ERR_ARG, "mtk_ddim_reverse_step: null pointer or negative B");
    if (B == 0 || ctx->Nn == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    DdimRevArgs a;
    a.B = B; a.Nn = ctx->Nn; a.gx0 = ctx->gauss_x0;
    a.sqrt_recip = (float)ctx->tab(MTK_TAB_SQRT_RECIP_ACP, t);
    a.sqrt_recipm1 = (float)ctx->tab(MTK_TAB_SQRT_RECIPM1_ACP, t);
    const float abn = t + 1 < ctx->T ? (float)ctx->tab(MTK_TAB_ACP, t + 1) : 0.0f;  // alphas_cumprod_next, GMD:262
    a.c_x0 = sqrtf(abn);
    a.c_eps = sqrtf(1.0f - abn);
    a.mo = model_out; a.mo_rs = mo_rs; a.mo_cs = mo_cs;
    a.z = z; a.x_in = x_in_next; a.xi_rs = xi_rs; a.xi_cs = xi_cs;
    hipLaunchKernelGGL(k_ddim_reverse_step, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_sample_init(mtk_ctx* ctx, int64_t B, const float* z0, const float* u0, double* z, int32_t* z_idx,
                               float* x_in, int64_t xi_rs, int64_t xi_cs, uint64_t seed, uint64_t row0, mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_sample_init: null context");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || (ctx->Nn > 0 && !z) || (ctx->C > 0 && !z_idx)) MTK_FAIL(MTK_ERR_ARG, "mtk_sample_init: null pointer or negative B");  // x_in may be NULL (mtk_mlp_forward_state reads the state)
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    InitArgs a{ctx->dev(), B, z0, u0, z, z_idx, x_in, xi_rs, xi_cs, seed, row0, ctx->seed_src};
    hipLaunchKernelGGL(k_sample_init, dim3(grid_for(B)), dim3(kBlock), lds_bytes(ctx, false), (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_gauss_kl_rows(mtk_ctx* ctx, int64_t B, const float* model_out, int64_t mo_rs, int64_t mo_cs, const float* x0,
                                 int64_t x0_ld, const float* x_t, int64_t xt_rs, int64_t xt_cs, const int64_t* t, double* loss,
                                 mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_gauss_kl_rows: null context");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || ctx->Nn < 1) MTK_FAIL(MTK_ERR_ARG, "mtk_gauss_kl_rows: negative B or no numeric columns");
    if (B == 0) return MTK_OK;
    if (!model_out || !x0 || !x_t || !t || !loss) MTK_FAIL(MTK_ERR_ARG, "mtk_gauss_kl_rows: null pointer");
    if (x0_ld < ctx->Nn) MTK_FAIL(MTK_ERR_ARG, "mtk_gauss_kl_rows: x0_ld too small");
    MTK_HIP(hipSetDevice(ctx->device));
    {
        const int rck = mtk_ensure_kl_tab(ctx);
        if (rck) return rck;
    }
    KlArgs a{B, ctx->Nn, ctx->T, ctx->gauss_x0, ctx->d_rows, ctx->d_kl, model_out, mo_rs, mo_cs, x0, x0_ld, x_t, xt_rs, xt_cs, t, loss};
    hipLaunchKernelGGL(k_gauss_kl_rows, dim3(grid_for(B)), dim3(kBlock), 0, (hipStream_t)stream, a);
    MTK_HIP(hipGetLastError());
    return mtk_poison_bad_t(t, B, ctx->T, nullptr, 0, 0, 0, nullptr, 0, 0, 0, loss, (hipStream_t)stream);
}

extern "C" int mtk_sample_decode(mtk_ctx* ctx, int64_t B, const double* z, const int32_t* z_idx, double* sample,
                                 mtk_stream stream) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_sample_decode: null context");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !sample) MTK_FAIL(MTK_ERR_ARG, "mtk_sample_decode: null pointer or negative B");
    if (B == 0) return MTK_OK;
    MTK_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_sample_decode, dim3(grid_for(B * (ctx->Nn + ctx->C))), dim3(kBlock), 0, (hipStream_t)stream, B,
                       ctx->Nn, ctx->C, z, z_idx, sample);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_ctx_set_seed_source(mtk_ctx* ctx, const uint64_t* seed_row0_dev) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_set_seed_source: null context");
    ctx->seed_src = seed_row0_dev;
    return MTK_OK;
}

extern "C" int mtk_ctx_set_parametrization(mtk_ctx* ctx, int gaussian_x0, int multinomial_direct) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_set_parametrization: null context");
    ctx->gauss_x0 = gaussian_x0 != 0;
    ctx->direct = multinomial_direct != 0;
    return MTK_OK;
}

extern "C" int mtk_ctx_set_math(mtk_ctx* ctx, int mode) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_ctx_set_math: null context");
    if (mode != MTK_MATH_REFERENCE && mode != MTK_MATH_FAST) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_ctx_set_math: unknown mode %d", mode);
    ctx->fast_math = (mode == MTK_MATH_FAST);
    return MTK_OK;
}

extern "C" int mtk_train_set_gaussian_loss(mtk_ctx* ctx, int kl) {
    if (!ctx) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_gaussian_loss: null context");
    if (kl != 0 && kl != 1) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_train_set_gaussian_loss: %d (0 = MSE, 1 = KL)", kl);
    ctx->train_gauss_kl = kl != 0;
    return MTK_OK;
}
