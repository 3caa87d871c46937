// mtk_train.hip — one training iteration of the denoiser on gfx950.
//
// Reference: models/clavaddpm/trainer.py:70-94 (zero_grad, mixed_loss, backward, AdamW.step),
// :58-68 (learning-rate anneal, applied after the step) and :138-155 (EMA of the denoiser).
//
// Layout.  The five parameter images (weights, gradients, Adam m, Adam v, EMA) have the padded layout the
// forward GEMMs already use ([out_pad][in_pad] + [out_pad] per Linear, zero padding) and sit back to back in ONE
// caller-owned buffer: the host all-reduces the gradient image over RCCL between mtk_train_step and
// mtk_train_apply, and padding entries stay exactly zero under AdamW (zero gradient, zero weight).
//
// Kernels.  One fp32 GEMM kernel on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate), 128x128x32
// workgroup tile, with either operand read row-major or transposed and three epilogues: forward
// (bias [+ time embedding] [+ ReLU]), dgrad (ReLU mask of the saved activation) and wgrad (split-K over the batch,
// hardware fp32 atomic add into the zeroed gradient image).  The loss and its gradient come from
// k_loss_grad (mtk_diffusion.hip); AdamW + EMA is one elementwise pass over the image.
#include <math.h>
#include <stdlib.h>

#include "mtk_common.h"

int mtk_launch_loss_grad(mtk_ctx* ctx, int64_t B, const float* out, int64_t ld, const float* x0, int64_t x0_ld,
                         const int32_t* xt_idx, const float* eps, const int64_t* t, float* loss_multi, float* loss_gauss,
                         float* dout, double w_multi, double w_gauss, const float* w_dev, uint64_t seed, uint64_t row0, hipStream_t stream,
                         uint16_t* dout16 = nullptr);  // mtk_diffusion.hip
static int materialize_grads(mtk_ctx* c, hipStream_t s);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, TN = 128, TK = 32;  // padding granules of the operands (the 64-wide tile variant divides them)
constexpr int LDN = TK + 4;                 // row-major tile in LDS: [rows][36]

struct G2 {
    int M, N, K;            // M and K are tile multiples; N is the number of columns stored (grid.y covers it)
    int kchunk;             // reduction range per blockIdx.z (multiple of 32)
    const float* A; int64_t lda;   // AT ? A[k * lda + m] : A[m * lda + k]
    const float* B; int64_t ldb;   // BT ? B[k * ldb + n] : B[n * ldb + k]
    float* C; int64_t c_rs, c_cs;  // C[m * c_rs + n * c_cs]
    const float* bias;             // [grid.y * tile] or null
    const float* emb; int emb_ld;  // optional [T][emb_ld], gathered by tvec[m]
    const float* lab; const int64_t* yvec; int n_lab, lab_ld;  // with emb, optional: += silu(lab[yvec[m]][n]) (label embedding, MODEL:610-612)
    const int64_t* tvec; int64_t rows_valid; int T;  // per-row timesteps are clamped into [0, T) (mtk_tclamp)
    const float* mask; int64_t m_rs;  // optional row-major activation: result zeroed (times neg_slope) where mask <= 0
    int relu;                      // 1: ReLU / LeakyReLU (neg_slope), 2: SiLU (x / (1 + exp(-x)), the TabSyn denoiser)
    float neg_slope;               // 0: ReLU; > 0: LeakyReLU slope, forward (relu != 0) and backward (mask != null) alike
    const float* addend; int64_t add_rs;  // optional row-major [M][>= N] matrix added before the activation
    float* pre; int64_t pre_rs;    // optional (k_gemm2 only): the value before the activation, row-major (SiLU's derivative needs it)
    int atomic;                    // accumulate with fp32 atomic add (split-K)
    float* rowsum;                 // optional [M]: += sum_k A[m][k] (the bias gradient rides in the wgrad GEMM: the waves of
                                   // column tile 0 sum the A fragments they hold anyway; atomics, like C under split-K)
    // deterministic split-K: slice z of the reduction stores its partial tile (and row sums) to slab + z * slab_stride
    // (rowsum_slab + z * M) with plain stores; k_reduce_splits then adds the slices in order.  Replaces the atomics.
    float* slab; int64_t slab_stride;
    float* rowsum_slab;
    int bf16;                      // operands rounded to bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate): k_gemm2b
    // k_gemm2b only: A / B are bf16 arrays (lda / ldb in elements), C is a bf16 array (plain store: no slab, no atomics), mask is
    int a16, b16, c16, mask16;
    uint16_t* Ct; int64_t ct_ld;   // optional second copy of the result, transposed and bf16: Ct[n * ct_ld + m]
    int tile64;                    // 64 x 64 tiles whatever the grid size (weight gradients split over the batch: see wgrad_chunk)
    // Dropout after the activation (MODEL:162-180 in training mode; drop_thresh = 0: none): element (row, n) of hidden layer
    // drop_layer is kept when its Philox word - counter (drop_row0 + row, layer, n), stream MTK_STREAM_DROP, key drop_seed - is
    // >= drop_thresh = p 2^32, and scaled by drop_scale = 1 / (1 - p); a dropped one is 0.  The saved activation then carries the
    // mask (kept and positive <=> stored value > 0), so the backward only needs the scale: mask_scale (0 reads as 1).
    uint32_t drop_thresh; float drop_scale; uint64_t drop_seed, drop_row0; int drop_layer;
    float mask_scale;
    int x3;                        // split-bf16 product of fp32 operands (k_gemm2x): 64 x 64 tiles, fp32 result; no row sums, no bf16 operands
    int dma;                       // the caller vouches for k_gemm2c's conditions (see there): 1 = B has rows up to the next
                                   // multiple of 64 past N, 2 = of 128
};

__device__ __forceinline__ bool g2_keep(uint64_t seed, uint64_t grow, int layer, int n, uint32_t thresh) {
    return mtk_philox((uint32_t)grow, (uint32_t)(grow >> 32), ((uint32_t)layer << 16) | (uint32_t)n, MTK_STREAM_DROP, seed).x >= thresh;
}
// silu(label_embedding[y[row]][n]) (MODEL:610-612); rows past the batch read label 0
__device__ __forceinline__ float g2_label(const float* lab, int lab_ld, int n_lab, const int64_t* yvec, int64_t row, int64_t rows_valid, int n) {
    const int y = (row < rows_valid) ? mtk_tclamp(yvec[row], n_lab) : 0;
    const float w = lab[(int64_t)y * lab_ld + n];
    return w / (1.0f + expf(-w));
}
// (scalars, not the argument struct by reference: that put a copy of the struct into scratch memory in the 128-tile kernels)
__device__ __forceinline__ float g2_dropout(uint32_t thresh, float scale, uint64_t seed, uint64_t grow, int layer, float v, int n) {
    if (thresh == 0) return v;
    return g2_keep(seed, grow, layer, n, thresh) ? v * scale : 0.0f;
}

// One operand tile (T rows x 32 k) moves global -> registers -> LDS; the registers are loaded one k-block ahead of
// the MFMAs that consume the previous block (software pipelining: at the batch sizes the reference trains with there
// are too few workgroups per CU to hide a global round trip per k-block any other way).
template <bool TR, int T>
struct TileMover {
    static constexpr int NLD = T * TK / 4 / 256;  // float4 per thread: 2 (T = 64) or 4 (T = 128)
    static constexpr int LDT = T + 4;             // transposed tile in LDS: [32][T + 4]
    static_assert(NLD == 2 || NLD == 4, "tile sizes 64 and 128");
    // Named registers, not an array: as `float4 r[NLD]` the prefetch lived in a stack slot (scratch_store right after
    // each global load, behind a vmcnt wait, and scratch_load + vmcnt(0) before each LDS write), which serialised the
    // global round trip the prefetch is there to hide.
    typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 (a union type) kept the struct in memory
    v4f r0, r1, r2, r3;
    template <int Q>
    __device__ __forceinline__ v4f& reg() {
        if constexpr (Q == 0) return r0;
        else if constexpr (Q == 1) return r1;
        else if constexpr (Q == 2) return r2;
        else return r3;
    }
    template <int Q>
    __device__ __forceinline__ void load_q(const float* src, int64_t ld, int64_t row0, int k0, int tid) {
        const int idx = tid + 256 * Q;
        if (!TR) {
            const int rr = idx >> 3, c = (idx & 7) * 4;
            reg<Q>() = *reinterpret_cast<const v4f*>(src + (row0 + rr) * ld + k0 + c);
        } else {
            const int kk = idx / (T / 4), r4 = (idx % (T / 4)) * 4;
            reg<Q>() = *reinterpret_cast<const v4f*>(src + (int64_t)(k0 + kk) * ld + row0 + r4);
        }
    }
    template <int Q>
    __device__ __forceinline__ void store_q(float* lds, int tid) {
        const int idx = tid + 256 * Q;
        if (!TR) {
            const int rr = idx >> 3, c = (idx & 7) * 4;
            *reinterpret_cast<v4f*>(&lds[rr * LDN + c]) = reg<Q>();
        } else {
            const int kk = idx / (T / 4), r4 = (idx % (T / 4)) * 4;
            *reinterpret_cast<v4f*>(&lds[kk * LDT + r4]) = reg<Q>();
        }
    }
    __device__ __forceinline__ void load(const float* src, int64_t ld, int64_t row0, int k0, int tid) {
        load_q<0>(src, ld, row0, k0, tid);
        load_q<1>(src, ld, row0, k0, tid);
        if constexpr (NLD == 4) {
            load_q<2>(src, ld, row0, k0, tid);
            load_q<3>(src, ld, row0, k0, tid);
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) {
        store_q<0>(lds, tid);
        store_q<1>(lds, tid);
        if constexpr (NLD == 4) {
            store_q<2>(lds, tid);
            store_q<3>(lds, tid);
        }
    }
    // lane (li, hh) takes k in [16 hh, 16 hh + 16) of the K block for both operands
    static __device__ __forceinline__ void frag(const float* lds, int row, int hh, float (&f)[16]) {
        if (!TR) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(&lds[row * LDN + 16 * hh + 4 * q]);
                f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) f[s] = lds[(16 * hh + s) * LDT + row];
        }
    }
};

// WT = 2: 128 x 128 workgroup tile (each wave 2 x 2 MFMA tiles); WT = 1: 64 x 64 (each wave one 32 x 32 tile), used when
// the larger tile would leave the chip with fewer than two workgroups per CU.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t to_bf16x8(const float* f) {
    bf16x8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
    return v;
}

// BF (mixed-precision training, mtk_train_set_precision): same tiles, same fp32 operands in memory and LDS, same fp32
// accumulators and epilogue - only the product changes.  A lane holds 16 consecutive k of its row per k-block; as fp32
// they feed sixteen v_mfma_f32_32x32x2f32, rounded to bf16 they are the operands of two v_mfma_f32_32x32x16_bf16 (any
// assignment of k values to lanes is valid as long as both operands use the same one).  Master weights, gradients,
// AdamW and EMA stay fp32.
template <bool AT, bool BT, int WT, bool BF, bool DROP = false>  // DROP: the forward GEMM of a hidden layer under dropout (its own instantiation: the Philox block per element costs the others nothing)
__global__ __launch_bounds__(256) void k_gemm2(G2 g) {
    constexpr int T = 64 * WT;
    constexpr int TILE_FLOATS = T * LDN > TK * (T + 4) ? T * LDN : TK * (T + 4);
    __shared__ __attribute__((aligned(16))) float As[TILE_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[TILE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * T;
    const int n0 = blockIdx.y * T;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;

    f32x16 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const bool sum_rows = (g.rowsum != nullptr || g.rowsum_slab != nullptr) && blockIdx.y == 0 && wc == 0;  // wave-uniform
    float rs[WT];
#pragma unroll
    for (int a = 0; a < WT; ++a) rs[a] = 0.0f;

    TileMover<AT, T> ma;
    TileMover<BT, T> mb;
    ma.load(g.A, g.lda, m0, kbeg, tid);
    mb.load(g.B, g.ldb, n0, kbeg, tid);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        ma.store(As, tid);
        mb.store(Bs, tid);
        __syncthreads();
        if (k0 + TK < kend) {  // next k-block's global loads fly while this one is multiplied
            ma.load(g.A, g.lda, m0, k0 + TK, tid);
            mb.load(g.B, g.ldb, n0, k0 + TK, tid);
        }
        float a[WT][16], b[WT][16];
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) TileMover<AT, T>::frag(As, wr * 32 * WT + mt * 32 + li, hh, a[mt]);
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) TileMover<BT, T>::frag(Bs, wc * 32 * WT + nt * 32 + li, hh, b[nt]);
        if (sum_rows) {
#pragma unroll
            for (int mt = 0; mt < WT; ++mt) {
                float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                for (int s = 0; s < 16; s += 2) { t0 += a[mt][s]; t1 += a[mt][s + 1]; }
                rs[mt] += t0 + t1;
            }
        }
        if constexpr (BF) {
            bf16x8_t ab[WT][2], bb[WT][2];
#pragma unroll
            for (int q = 0; q < WT; ++q) {
                ab[q][0] = to_bf16x8(a[q]); ab[q][1] = to_bf16x8(a[q] + 8);
                bb[q][0] = to_bf16x8(b[q]); bb[q][1] = to_bf16x8(b[q] + 8);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[mt][h], bb[nt][h], acc[mt][nt], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < WT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    if (sum_rows) {  // lanes l and l + 32 hold the two k-halves of row (lane & 31)
#pragma unroll
        for (int mt = 0; mt < WT; ++mt) {
            const float tot = rs[mt] + __shfl_xor(rs[mt], 32);
            if (hh == 0) {
                if (g.rowsum_slab) g.rowsum_slab[(int64_t)blockIdx.z * g.M + m0 + wr * 32 * WT + mt * 32 + li] = tot;
                else unsafeAtomicAdd(&g.rowsum[m0 + wr * 32 * WT + mt * 32 + li], tot);
            }
        }
    }

    // C/D map of the 32x32 MFMA: column (n) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int mt = 0; mt < WT; ++mt)
#pragma unroll
        for (int nt = 0; nt < WT; ++nt) {
            const int n = n0 + wc * 32 * WT + nt * 32 + li;
            if (n >= g.N) continue;
            const float bias = g.bias ? g.bias[n] : 0.0f;
            // the time-embedding values of the lane's sixteen results, requested together (all timesteps, then all the rows they
            // select): two round trips per sub-tile - as sixteen dependent (timestep -> row) pairs inside the loop below they cost a
            // 2,048 x 128 x 128 proj GEMM 8 of its 17 us
            float ev[16];
            if (g.emb) {
                int trs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = (m0 + wr * 32 * WT + mt * 32) + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    trs[r] = (row < g.rows_valid) ? mtk_tclamp(g.tvec[row], g.T) : 0;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) ev[r] = g.emb[(int64_t)trs[r] * g.emb_ld + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * 32 * WT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float v = acc[mt][nt][r] + bias;
                if (g.emb) {
                    v += ev[r];
                    if (g.lab) v += g2_label(g.lab, g.lab_ld, g.n_lab, g.yvec, row, g.rows_valid, n);
                }
                if (g.addend) v += g.addend[row * g.add_rs + n];
                if (g.pre) g.pre[row * g.pre_rs + n] = v;
                if (g.relu == 2) v = v / (1.0f + expf(-v));
                else if (g.relu) v = v > 0.0f ? v : g.neg_slope * v;
                if constexpr (DROP) v = g2_dropout(g.drop_thresh, g.drop_scale, g.drop_seed, g.drop_row0 + (uint64_t)row, g.drop_layer, v, n);
                if (g.mask) v = (g.mask[row * g.m_rs + n] > 0.0f) ? v * (g.mask_scale != 0.0f ? g.mask_scale : 1.0f) : g.neg_slope * v;
                float* dst = (g.slab ? g.slab + (int64_t)blockIdx.z * g.slab_stride : g.C) + row * g.c_rs + (int64_t)n * g.c_cs;
                if (g.atomic && !g.slab) unsafeAtomicAdd(dst, v); else *dst = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 training GEMM (mtk_train_set_precision(MTK_PREC_BF16)), round 3.  k_gemm2<..., BF = true> above keeps fp32 tiles in
// LDS and rounds at the fragment read.  At config 5's per-rank shape (2,048 rows, 4 x 1024) a 1024 x 1024 layer took 23 us
// that way, 7 % of the bf16 peak, and the iteration was those GEMMs; rocprof and two rewrites of the loop (bf16 tiles in LDS,
// 64 k per step; two k-steps of loads in flight) left the time where it was: the loop is bound by the bytes a CU pulls from
// the L2 - 50 GB/s per CU, 12.5 TB/s chip-wide, for fp32 operands that are rounded to bf16 on arrival.  So in this mode the
// operands are STORED as bf16 (saved activations, activation gradients, a bf16 copy of the weight image kept by the
// optimizer kernel): half the bytes through the same pipe, and the same numbers - every one of them was rounded to bf16
// before its product anyway.  Tiles are bf16 in LDS, [row][k] order whatever the source layout (a transposed source is
// transposed by the LDS write), 64 k per step, one ds_read_b128 per fragment.  fp32 accumulation, fp32 gradient image,
// split-K slabs, AdamW and EMA are unchanged.
constexpr int KB2 = 64;          // k per step
constexpr int PITCH2 = KB2 + 8;  // bf16 elements per LDS row: 144 B, 16-byte aligned, rows 36 dwords apart (conflict-free b128 reads)

__device__ __forceinline__ unsigned g2_pack2(float a, float b) {
    const __bf16 e[2] = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<const unsigned*>(e);
}

// LO: the part the bf16 rounding left behind, itself rounded to bf16 (split-bf16: v ~ hi + lo to 2^-17 relative)
template <bool LO>
__device__ __forceinline__ unsigned g2_pack2x(float a, float b) {
    if constexpr (!LO) return g2_pack2(a, b);
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    return g2_pack2(a - (float)ha, b - (float)hb);
}

// SRC16 = false: fp32 source, rounded on the way into LDS.  SRC16 = true: bf16 source, copied.
template <bool TR, int T, bool SRC16>
struct TileMoverB;

template <bool TR, int T>
struct TileMoverB<TR, T, false> {
    static constexpr int NLD = T * KB2 / 4 / 256;  // float4 per thread and k-step: 4 (T = 64) or 8 (T = 128)
    static_assert(NLD == 4 || NLD == 8, "tile sizes 64 and 128");
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f r0, r1, r2, r3, r4, r5, r6, r7;
    template <int Q>
    __device__ __forceinline__ v4f& reg() {
        if constexpr (Q == 0) return r0;
        else if constexpr (Q == 1) return r1;
        else if constexpr (Q == 2) return r2;
        else if constexpr (Q == 3) return r3;
        else if constexpr (Q == 4) return r4;
        else if constexpr (Q == 5) return r5;
        else if constexpr (Q == 6) return r6;
        else return r7;
    }
    // !TR: thread (Q) -> row idx / 16, k = 4 (idx % 16).  TR (source [k][row]): registers 2j and 2j + 1 hold k = 2 kp and 2 kp + 1
    // of the same four rows, so the LDS write packs the two k of each row into one dword.
    template <int Q>
    __device__ __forceinline__ void load_q(const void* src_, int64_t ld, int64_t row0, int k0, int kend, int tid) {
        const float* src = reinterpret_cast<const float*>(src_);
        if (!TR) {
            const int idx = tid + 256 * Q, rr = idx >> 4, c = (idx & 15) * 4;
            const bool ok = k0 + c < kend;  // K is a multiple of 32: a float4 is inside or outside as a whole
            const v4f v = *reinterpret_cast<const v4f*>(src + (row0 + rr) * ld + (ok ? k0 + c : k0));
            reg<Q>() = ok ? v : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        } else {
            const int idx = tid + 256 * (Q >> 1), kp = idx / (T / 4), rq = (idx % (T / 4)) * 4, kk = 2 * kp + (Q & 1);
            const bool ok = k0 + kk < kend;
            const v4f v = *reinterpret_cast<const v4f*>(src + (int64_t)(ok ? k0 + kk : k0) * ld + row0 + rq);
            reg<Q>() = ok ? v : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    template <int Q, bool LO = false>
    __device__ __forceinline__ void store_q(uint16_t* lds, int tid) {
        if (!TR) {
            const int idx = tid + 256 * Q, rr = idx >> 4, c = (idx & 15) * 4;
            const v4f v = reg<Q>();
            *reinterpret_cast<uint2*>(&lds[rr * PITCH2 + c]) = make_uint2(g2_pack2x<LO>(v[0], v[1]), g2_pack2x<LO>(v[2], v[3]));
        } else if constexpr ((Q & 1) == 0) {
            const int idx = tid + 256 * (Q >> 1), kp = idx / (T / 4), rq = (idx % (T / 4)) * 4;
            const v4f lo = reg<Q>(), hi = reg<Q + 1>();
            // the lanes of a write differ in their row group (4 rows = 144 dwords apart: two banks for sixteen lanes); lane i starts
            // its four rows at row i mod 4, which spreads a write over eight banks (2-way: free for ds_write_b32)
            const int rot = (rq >> 2) & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int jj = (j + rot) & 3;
                const float a = mtk_pick4f(lo[0], lo[1], lo[2], lo[3], jj), b = mtk_pick4f(hi[0], hi[1], hi[2], hi[3], jj);
                *reinterpret_cast<unsigned*>(&lds[(rq + jj) * PITCH2 + 2 * kp]) = g2_pack2x<LO>(a, b);
            }
        }
    }
    template <int Q = 0>
    __device__ __forceinline__ void load(const void* src, int64_t ld, int64_t row0, int k0, int kend, int tid) {
        load_q<Q>(src, ld, row0, k0, kend, tid);
        if constexpr (Q + 1 < NLD) load<Q + 1>(src, ld, row0, k0, kend, tid);
    }
    template <int Q = 0>
    __device__ __forceinline__ void store(uint16_t* lds, int tid) {
        store_q<Q>(lds, tid);
        if constexpr (Q + 1 < NLD) store<Q + 1>(lds, tid);
    }
    template <int Q = 0>
    __device__ __forceinline__ void store_lo(uint16_t* lds, int tid) {  // the low parts of the same registers, same layout
        store_q<Q, true>(lds, tid);
        if constexpr (Q + 1 < NLD) store_lo<Q + 1>(lds, tid);
    }
};

template <bool TR, int T>
struct TileMoverB<TR, T, true> {
    static constexpr int NLD = T * KB2 / 8 / 256;  // 16-byte loads per thread and k-step: 2 (T = 64) or 4 (T = 128)
    static_assert(NLD == 2 || NLD == 4, "tile sizes 64 and 128");
    uint4 r0, r1, r2, r3;
    template <int Q>
    __device__ __forceinline__ uint4& reg() {
        if constexpr (Q == 0) return r0;
        else if constexpr (Q == 1) return r1;
        else if constexpr (Q == 2) return r2;
        else return r3;
    }
    // !TR: thread (Q) -> row idx / 8, k = 8 (idx % 8): a 16-byte load is a 16-byte LDS write.  TR (source [k][row]): a load is 8 rows
    // of one k; registers 2j and 2j + 1 hold k = 2 kp and 2 kp + 1 of the same eight rows.
    template <int Q>
    __device__ __forceinline__ void load_q(const void* src_, int64_t ld, int64_t row0, int k0, int kend, int tid) {
        const uint16_t* src = reinterpret_cast<const uint16_t*>(src_);
        if (!TR) {
            const int idx = tid + 256 * Q, rr = idx >> 3, c = (idx & 7) * 8;
            const bool ok = k0 + c < kend;
            const uint4 v = *reinterpret_cast<const uint4*>(src + (row0 + rr) * ld + (ok ? k0 + c : k0));
            reg<Q>() = ok ? v : make_uint4(0u, 0u, 0u, 0u);
        } else {
            const int idx = tid + 256 * (Q >> 1), kp = idx / (T / 8), rq = (idx % (T / 8)) * 8, kk = 2 * kp + (Q & 1);
            const bool ok = k0 + kk < kend;
            const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)(ok ? k0 + kk : k0) * ld + row0 + rq);
            reg<Q>() = ok ? v : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    template <int Q>
    __device__ __forceinline__ void store_q(uint16_t* lds, int tid) {
        if (!TR) {
            const int idx = tid + 256 * Q, rr = idx >> 3, c = (idx & 7) * 8;
            *reinterpret_cast<uint4*>(&lds[rr * PITCH2 + c]) = reg<Q>();
        } else if constexpr ((Q & 1) == 0) {
            const int idx = tid + 256 * (Q >> 1), kp = idx / (T / 8), rq = (idx % (T / 8)) * 8;
            const uint4 lo = reg<Q>(), hi = reg<Q + 1>();
            // lanes of a write sit 8 rows (288 dwords, = 0 mod 32 banks) apart: lane i starts at row i mod 8 of its eight
            const int rot = (rq >> 3) & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int jj = (j + rot) & 7;
                const unsigned wl = mtk_pick4(lo.x, lo.y, lo.z, lo.w, jj >> 1), wh = mtk_pick4(hi.x, hi.y, hi.z, hi.w, jj >> 1);
                const unsigned el = (jj & 1) ? (wl >> 16) : (wl & 0xffffu), eh = (jj & 1) ? (wh >> 16) : (wh & 0xffffu);
                *reinterpret_cast<unsigned*>(&lds[(rq + jj) * PITCH2 + 2 * kp]) = el | (eh << 16);
            }
        }
    }
    template <int Q = 0>
    __device__ __forceinline__ void load(const void* src, int64_t ld, int64_t row0, int k0, int kend, int tid) {
        load_q<Q>(src, ld, row0, k0, kend, tid);
        if constexpr (Q + 1 < NLD) load<Q + 1>(src, ld, row0, k0, kend, tid);
    }
    template <int Q = 0>
    __device__ __forceinline__ void store(uint16_t* lds, int tid) {
        store_q<Q>(lds, tid);
        if constexpr (Q + 1 < NLD) store<Q + 1>(lds, tid);
    }
};

// What follows the k-loop of the bf16 GEMMs (k_gemm2b, k_gemm2c): bias-gradient row sums, then the result tile.  `smem` is the
// kernel's LDS, free by now (the caller's loop ends with a barrier), at least 4 * 2 * 32 * 40 elements.
// Wave layout: (waves / 2) x 2, wave (wr, wc) owns MT x NT sub-tiles of 32 x 32 from row 32 MT wr, column 32 NT wc of the tile.
template <int MT, int NT, bool DROP>
__device__ __forceinline__ void gemm2b_finish(const G2& g, f32x16 (&acc)[MT][NT], float (&rs)[MT], bool sum_rows, uint16_t* smem,
                                              int64_t m0, int n0, int lane, int wid) {
    constexpr int CP = 40;  // pitch (elements) of a wave's 32 x 32 staging tiles: 80-byte rows, 16-byte aligned
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    if (sum_rows) {  // lanes l and l + 32 hold the two k-halves of row (lane & 31)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float tot = rs[mt] + __shfl_xor(rs[mt], 32);
            if (hh == 0) {
                if (g.rowsum_slab) g.rowsum_slab[(int64_t)blockIdx.z * g.M + m0 + wr * 32 * MT + mt * 32 + li] = tot;
                else unsafeAtomicAdd(&g.rowsum[m0 + wr * 32 * MT + mt * 32 + li], tot);
            }
        }
    }
    // bf16 results (activations, activation gradients) leave through LDS: the MFMA layout gives a lane ONE column of the tile, so
    // storing from it is a 2-byte store per element in 64-byte runs, the ReLU mask is read the same way, and the transposed copy
    // goes out in 8-byte pieces a row apart (+3 us per copy, +5 us for the mask on a 2,048 x 1,024 result).  Each wave stages its
    // own 32 x 32 sub-tiles (no workgroup barrier): mask tile in by 16-byte loads, results written over it in place (every
    // element belongs to one lane) and, transposed, beside it; both leave as 16-byte stores.
    if (g.c16 && g.c_cs == 1 && (!g.mask || (g.mask16 && g.neg_slope == 0.0f))) {
        uint16_t* const cs = smem + wid * (2 * 32 * CP);
        uint16_t* const cst = cs + 32 * CP;
        uint16_t* const C16 = reinterpret_cast<uint16_t*>(g.C);
        const uint16_t* const M16 = reinterpret_cast<const uint16_t*>(g.mask);
        const int prow = lane >> 2, pch = (lane & 3) * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nb = n0 + wc * 32 * NT + nt * 32;  // wave-uniform (N is a multiple of 32 for these results)
                if (nb >= g.N) continue;
                const int64_t rb = m0 + wr * 32 * MT + mt * 32;
                if (M16) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        *reinterpret_cast<uint4*>(&cs[(prow + 16 * i) * CP + pch]) =
                            *reinterpret_cast<const uint4*>(M16 + (rb + prow + 16 * i) * g.m_rs + nb + pch);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
                const int n = nb + li;
                const float bias = g.bias ? g.bias[n] : 0.0f;
                unsigned tq[8];
                // the time-embedding values of the lane's sixteen results, requested together (all timesteps, then all the rows they
                // select): two round trips per sub-tile - as sixteen dependent (timestep -> row) pairs inside the loop below they cost a
                // 2,048 x 128 x 128 proj GEMM 8 of its 17 us
                float ev[16];
                if (g.emb) {
                    int trs[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = rb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        trs[r] = (row < g.rows_valid) ? mtk_tclamp(g.tvec[row], g.T) : 0;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) ev[r] = g.emb[(int64_t)trs[r] * g.emb_ld + n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const int64_t row = rb + lr;
                    float v = acc[mt][nt][r] + bias;
                    if (g.emb) {
                        v += ev[r];
                        if (g.lab) v += g2_label(g.lab, g.lab_ld, g.n_lab, g.yvec, row, g.rows_valid, n);
                    }
                    if (g.addend) v += g.addend[row * g.add_rs + n];
                    if (g.relu == 2) v = v / (1.0f + expf(-v));
                    else if (g.relu) v = v > 0.0f ? v : g.neg_slope * v;
                    if constexpr (DROP) v = g2_dropout(g.drop_thresh, g.drop_scale, g.drop_seed, g.drop_row0 + (uint64_t)row, g.drop_layer, v, n);
                    if (M16) {
                        const unsigned mbits = cs[lr * CP + li];
                        v = (mbits != 0 && !(mbits & 0x8000u)) ? v * (g.mask_scale != 0.0f ? g.mask_scale : 1.0f) : 0.0f;  // > 0 for a bf16 that is not a NaN (activations are not)
                    }
                    const __bf16 e = (__bf16)v;
                    const unsigned bits = *reinterpret_cast<const uint16_t*>(&e);
                    cs[lr * CP + li] = (uint16_t)bits;
                    if (r & 1) tq[r >> 1] |= bits << 16; else tq[r >> 1] = bits;
                }
                if (g.Ct) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(&cst[li * CP + 8 * q + 4 * hh]) = make_uint2(tq[2 * q], tq[2 * q + 1]);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int pr = prow + 16 * i;
                    *reinterpret_cast<uint4*>(C16 + (rb + pr) * g.c_rs + nb + pch) = *reinterpret_cast<const uint4*>(&cs[pr * CP + pch]);
                    if (g.Ct) *reinterpret_cast<uint4*>(g.Ct + (int64_t)(nb + pr) * g.ct_ld + rb + pch) = *reinterpret_cast<const uint4*>(&cst[pr * CP + pch]);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        return;
    }
    // epilogue: as k_gemm2 (C/D map of the 32x32 MFMA: column (n) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5));
    // c16: the result is an activation (or an activation gradient) stored as bf16; mask16: the saved activation is
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + wc * 32 * NT + nt * 32 + li;
            if (n >= g.N) continue;
            const float bias = g.bias ? g.bias[n] : 0.0f;
            unsigned tq[8];  // the 16 results as bf16 pairs, for the transposed copy
            // the time-embedding values of the lane's sixteen results, requested together (all timesteps, then all the rows they
            // select): two round trips per sub-tile - as sixteen dependent (timestep -> row) pairs inside the loop below they cost a
            // 2,048 x 128 x 128 proj GEMM 8 of its 17 us
            float ev[16];
            if (g.emb) {
                int trs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = (m0 + wr * 32 * MT + mt * 32) + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    trs[r] = (row < g.rows_valid) ? mtk_tclamp(g.tvec[row], g.T) : 0;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) ev[r] = g.emb[(int64_t)trs[r] * g.emb_ld + n];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wr * 32 * MT + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float v = acc[mt][nt][r] + bias;
                if (g.emb) {
                    v += ev[r];
                    if (g.lab) v += g2_label(g.lab, g.lab_ld, g.n_lab, g.yvec, row, g.rows_valid, n);
                }
                if (g.addend) v += g.addend[row * g.add_rs + n];
                if (g.relu == 2) v = v / (1.0f + expf(-v));
                else if (g.relu) v = v > 0.0f ? v : g.neg_slope * v;
                if constexpr (DROP) v = g2_dropout(g.drop_thresh, g.drop_scale, g.drop_seed, g.drop_row0 + (uint64_t)row, g.drop_layer, v, n);
                if (g.mask) {
                    bool pos;
                    if (g.mask16) {
                        const uint16_t mbits = reinterpret_cast<const uint16_t*>(g.mask)[row * g.m_rs + n];
                        pos = mbits != 0 && !(mbits & 0x8000u);  // > 0 for a bf16 that is not a NaN (activations are not)
                    } else {
                        pos = g.mask[row * g.m_rs + n] > 0.0f;
                    }
                    v = pos ? v * (g.mask_scale != 0.0f ? g.mask_scale : 1.0f) : g.neg_slope * v;
                }
                if (g.c16) {
                    const __bf16 e = (__bf16)v;
                    const unsigned bits = *reinterpret_cast<const uint16_t*>(&e);
                    reinterpret_cast<uint16_t*>(g.C)[row * g.c_rs + (int64_t)n * g.c_cs] = (uint16_t)bits;
                    if (r & 1) tq[r >> 1] |= bits << 16; else tq[r >> 1] = bits;
                } else {
                    float* dst = (g.slab ? g.slab + (int64_t)blockIdx.z * g.slab_stride : g.C) + row * g.c_rs + (int64_t)n * g.c_cs;
                    if (g.atomic && !g.slab) unsafeAtomicAdd(dst, v); else *dst = v;
                }
            }
            if (g.c16 && g.Ct) {  // registers 4 q .. 4 q + 3 are rows 8 q + 4 hh .. + 3: eight contiguous bytes of column n of C^T
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = m0 + wr * 32 * MT + mt * 32 + 8 * q + 4 * hh;
                    *reinterpret_cast<uint2*>(&g.Ct[(int64_t)n * g.ct_ld + row]) = make_uint2(tq[2 * q], tq[2 * q + 1]);
                }
            }
        }
}

template <bool AT, bool BT, int WT, bool A16, bool B16, bool DROP = false>
__global__ __launch_bounds__(256) void k_gemm2b(G2 g) {
    constexpr int T = 64 * WT;
    constexpr int kSmem = 2 * T * PITCH2 > 4 * 2 * 32 * 40 ? 2 * T * PITCH2 : 4 * 2 * 32 * 40;  // the epilogue stages 32 x 32 tiles here
    __shared__ __attribute__((aligned(16))) uint16_t smem[kSmem];
    uint16_t* const As = smem;
    uint16_t* const Bs = smem + T * PITCH2;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * T;
    const int n0 = blockIdx.y * T;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;

    f32x16 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const bool sum_rows = (g.rowsum != nullptr || g.rowsum_slab != nullptr) && blockIdx.y == 0 && wc == 0;  // wave-uniform
    float rs[WT];
#pragma unroll
    for (int a = 0; a < WT; ++a) rs[a] = 0.0f;

    TileMoverB<AT, T, A16> ma;
    TileMoverB<BT, T, B16> mb;
    ma.load(g.A, g.lda, m0, kbeg, kend, tid);
    mb.load(g.B, g.ldb, n0, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += KB2) {
        ma.store(As, tid);
        mb.store(Bs, tid);
        __syncthreads();
        if (k0 + KB2 < kend) {  // next k-step's global loads fly while this one is multiplied
            ma.load(g.A, g.lda, m0, k0 + KB2, kend, tid);
            mb.load(g.B, g.ldb, n0, k0 + KB2, kend, tid);
        }
#pragma unroll
        for (int s = 0; s < KB2 / 16; ++s) {
            bf16x8_t a[WT], b[WT];
#pragma unroll
            for (int mt = 0; mt < WT; ++mt) a[mt] = *reinterpret_cast<const bf16x8_t*>(&As[(wr * 32 * WT + mt * 32 + li) * PITCH2 + 16 * s + 8 * hh]);
#pragma unroll
            for (int nt = 0; nt < WT; ++nt) b[nt] = *reinterpret_cast<const bf16x8_t*>(&Bs[(wc * 32 * WT + nt * 32 + li) * PITCH2 + 16 * s + 8 * hh]);
            if (sum_rows) {
#pragma unroll
                for (int mt = 0; mt < WT; ++mt) {
                    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) { t0 += (float)a[mt][j]; t1 += (float)a[mt][j + 1]; }
                    rs[mt] += t0 + t1;
                }
            }
#pragma unroll
            for (int mt = 0; mt < WT; ++mt)
#pragma unroll
                for (int nt = 0; nt < WT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm2b_finish<WT, WT, DROP>(g, acc, rs, sum_rows, smem, m0, n0, lane, wid);
}

// Split-bf16 product of fp32 operands (the guidance classifier's MTK_PREC_BF16X3 mode): k_gemm2b's 64 x 64 tile with every fp32 value
// carried into LDS as hi = bf16(v) and lo = bf16(v - hi), and a.b ~ ah.bh + al.bh + ah.bl - three bf16 MFMAs per product into the
// same fp32 accumulators (the dropped al.bl term is ~2^-16 of the product), against sixteen fp32 MFMAs at a sixteenth of the rate.
template <bool AT, bool BT>
__global__ __launch_bounds__(256) void k_gemm2x(G2 g) {
    constexpr int T = 64;
    __shared__ __attribute__((aligned(16))) uint16_t smem[4 * T * PITCH2];
    uint16_t* const Ah = smem;
    uint16_t* const Bh = smem + T * PITCH2;
    uint16_t* const Al = smem + 2 * T * PITCH2;
    uint16_t* const Bl = smem + 3 * T * PITCH2;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * T;
    const int n0 = blockIdx.y * T;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    float rs[1] = {0.0f};
    TileMoverB<AT, T, false> ma;
    TileMoverB<BT, T, false> mb;
    ma.load(g.A, g.lda, m0, kbeg, kend, tid);
    mb.load(g.B, g.ldb, n0, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += KB2) {
        ma.store(Ah, tid);
        ma.store_lo(Al, tid);
        mb.store(Bh, tid);
        mb.store_lo(Bl, tid);
        __syncthreads();
        if (k0 + KB2 < kend) {
            ma.load(g.A, g.lda, m0, k0 + KB2, kend, tid);
            mb.load(g.B, g.ldb, n0, k0 + KB2, kend, tid);
        }
#pragma unroll
        for (int s = 0; s < KB2 / 16; ++s) {
            const int oa = (wr * 32 + li) * PITCH2 + 16 * s + 8 * hh, ob = (wc * 32 + li) * PITCH2 + 16 * s + 8 * hh;
            const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(&Ah[oa]), al = *reinterpret_cast<const bf16x8_t*>(&Al[oa]);
            const bf16x8_t bh = *reinterpret_cast<const bf16x8_t*>(&Bh[ob]), bl = *reinterpret_cast<const bf16x8_t*>(&Bl[ob]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[0][0], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm2b_finish<1, 1, false>(g, acc, rs, false, smem, m0, n0, lane, wid);
}

// The GEMM most of a bf16 iteration runs in: both operands bf16, both read along their rows (the transposed copies exist for
// this), 64 x 64 tiles.  k_gemm2b holds ONE k-step in registers on its way to LDS, and with two workgroups on a CU a k-step then
// costs a global round trip (14.7 us for 2,048 x 1,024 x 1,024: 35 GB/s per CU from an L2 that holds both operands).  Here the
// tiles go L2 -> LDS by LDS-DMA into a ring of kRing stages, kRing - 1 k-steps requested ahead of the one being multiplied, no
// registers involved, one barrier per k-step.  An LDS-DMA instruction writes 64 lanes x 16 bytes linearly (8 tile rows of 128
// bytes), so the bank swizzle is applied on the global side, as in mtk_mlp_bf16_layered.hip: slot (r, c) of a tile holds piece
// c ^ ((r >> 1) & 7) of row r, and a fragment read asks for slot kp ^ ((r >> 1) & 7).
// Host guarantees (launch): A, B bf16 with rows of >= K elements, K and kchunk multiples of 64, M a multiple of 64, B with rows up to
// the next multiple of 64 past N.
constexpr int kRing = 4;
extern __shared__ __attribute__((aligned(16))) uint16_t g2c_smem[];

// Wave layout WR x 2, every wave MT x NT sub-tiles of 32 x 32: <2, 1, 1> = 64 x 64 tiles, 4 waves, 64 KiB of LDS (two workgroups
// per CU); <4, 1, 2> = 128 x 128 tiles, 8 waves, 128 KiB.  With 64 x 64 tiles a 2,048 x 1,024 x 1,024 product pulls 134 MB out of
// the L2s and runs at their ~11 TB/s (12 us whatever the latency hiding); 128 x 128 halves the traffic.  Every wave moves two
// 8-row pieces of each operand per k-step in both shapes.
template <int WR, int MT, int NT, bool DROP = false>
__global__ __launch_bounds__(WR * 128) void k_gemm2c(G2 g) {
    constexpr int TM_ = WR * 32 * MT, TN_ = 2 * 32 * NT, NW = WR * 2;
    constexpr int PA = TM_ / 8 / NW, PB = TN_ / 8 / NW;  // 8-row pieces per wave and k-step
    static_assert(PA == 2 && PB == 2, "the swizzle key below is written for two pieces per wave");
    constexpr int kOpA = TM_ * 128, kStage = (TM_ + TN_) * 128;  // bytes: the A tile, one ring stage (A tile | B tile)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * TM_;
    const int n0 = blockIdx.y * TN_;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
    const int nk = (kend - kbeg) / 64;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const bool sum_rows = (g.rowsum != nullptr || g.rowsum_slab != nullptr) && blockIdx.y == 0 && wc == 0;  // wave-uniform
    float rs[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) rs[a] = 0.0f;

    // wave wv moves rows [16 wv, 16 wv + 16) of both tiles
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int lr = lane >> 3;
    unsigned voff_a[2], voff_b[2];  // [piece]: row in tile = 16 wv + 8 j + lr, its key (row >> 1) & 7 = (4 j + (lr >> 1)) & 7
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lc = (lane & 7) ^ ((4 * j + (lr >> 1)) & 7);
        voff_a[j] = (unsigned)(lr * g.lda * 2 + lc * 16);
        voff_b[j] = (unsigned)(lr * g.ldb * 2 + lc * 16);
    }
    const unsigned char* a_base = reinterpret_cast<const unsigned char*>(g.A) + ((m0 + 16 * wv) * g.lda + kbeg) * 2;
    const unsigned char* b_base = reinterpret_cast<const unsigned char*>(g.B) + (((int64_t)n0 + 16 * wv) * g.ldb + kbeg) * 2;
    auto issue = [&](int i) {  // k-step i into stage i % kRing
        const unsigned dst = (unsigned)((i % kRing) * kStage + wv * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* ga = a_base + (int64_t)j * 8 * g.lda * 2 + (int64_t)i * 128;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_a[j]), "s"(ga), "s"(dst + j * 1024u) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* gb = b_base + (int64_t)j * 8 * g.ldb * 2 + (int64_t)i * 128;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_b[j]), "s"(gb), "s"(dst + (unsigned)kOpA + j * 1024u) : "memory");
        }
    };
    for (int i = 0; i < kRing - 1 && i < nk; ++i) issue(i);
    const unsigned char* lds = reinterpret_cast<const unsigned char*>(g2c_smem);
    for (int i = 0; i < nk; ++i) {
        // k-step i has landed when at most the requests made after it are outstanding: min(kRing - 2, nk - 1 - i) steps of 4
        const int after = nk - 1 - i;
        if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's pieces of step i are in LDS, and every wave is done reading step i - 1's stage
        if (i + kRing - 1 < nk) issue(i + kRing - 1);
        const unsigned char* st = lds + (i % kRing) * kStage;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kp = 2 * s + hh;
            bf16x8_t a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int ra = wr * 32 * MT + mt * 32 + li;
                a[mt] = *reinterpret_cast<const bf16x8_t*>(st + ra * 128 + ((kp ^ ((ra >> 1) & 7)) * 16));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int rb = wc * 32 * NT + nt * 32 + li;
                b[nt] = *reinterpret_cast<const bf16x8_t*>(st + kOpA + rb * 128 + ((kp ^ ((rb >> 1) & 7)) * 16));
            }
            if (sum_rows) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) { t0 += (float)a[mt][j]; t1 += (float)a[mt][j + 1]; }
                    rs[mt] += t0 + t1;
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();  // the ring is free: the epilogue stages its tiles in it
    gemm2b_finish<MT, NT, DROP>(g, acc, rs, sum_rows, g2c_smem, m0, n0, lane, wid);
}

template <int WR, int MT, int NT, bool DROP = false>
int launch_c(const G2& g, dim3 grid, hipStream_t s) {
    constexpr int kBytes = kRing * (WR * 32 * MT + 64 * NT) * 128;
    static bool attr_set[64] = {};
    int dev = 0;
    MTK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        MTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm2c<WR, MT, NT, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_gemm2c<WR, MT, NT, DROP>), grid, dim3(WR * 128), kBytes, s, g);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

template <bool AT, bool BT, int WT>
void launch_b(const G2& g, dim3 grid, hipStream_t s) {
    if constexpr (!AT && !BT) {  // hidden-layer forward under dropout: bf16 activations times bf16 weights
        if (g.drop_thresh != 0 && g.a16 && g.b16) {
            hipLaunchKernelGGL((k_gemm2b<AT, BT, WT, true, true, true>), grid, dim3(256), 0, s, g);
            return;
        }
    }
    if (g.a16 && g.b16) hipLaunchKernelGGL((k_gemm2b<AT, BT, WT, true, true>), grid, dim3(256), 0, s, g);
    else if (g.a16) hipLaunchKernelGGL((k_gemm2b<AT, BT, WT, true, false>), grid, dim3(256), 0, s, g);
    else if (g.b16) hipLaunchKernelGGL((k_gemm2b<AT, BT, WT, false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((k_gemm2b<AT, BT, WT, false, false>), grid, dim3(256), 0, s, g);
}

template <bool AT, bool BT>
int launch(const G2& g, hipStream_t s) {
    const int splits = (g.K + g.kchunk - 1) / g.kchunk;
    const int64_t wg128 = (int64_t)(g.M / TM) * ((g.N + TN - 1) / TN) * splits;
    const dim3 g128((unsigned)(g.M / 128), (unsigned)((g.N + 127) / 128), (unsigned)splits), g64((unsigned)(g.M / 64), (unsigned)((g.N + 63) / 64), (unsigned)splits);
    if (g.x3) {
        if (g.a16 || g.b16 || g.c16 || g.rowsum || g.rowsum_slab || g.drop_thresh != 0) MTK_FAIL(MTK_ERR_STATE, "split-bf16 GEMM: fp32 operands and result only");
        hipLaunchKernelGGL((k_gemm2x<AT, BT>), g64, dim3(256), 0, s, g);
        MTK_HIP(hipGetLastError());
        return MTK_OK;
    }
    static const bool old_bf = getenv("MTK_TRAIN_BF16_OLD") != nullptr;  // A/B: the round-2 bf16 instantiation (fp32 tiles in LDS)
    static const bool no_dma = getenv("MTK_TRAIN_NO_DMA") != nullptr;    // A/B: k_gemm2b where k_gemm2c would run
    if (g.bf16 && !old_bf && !no_dma && g.dma && !AT && !BT && g.a16 && g.b16 && g.K % 64 == 0 && (g.kchunk % 64 == 0 || g.kchunk == g.K)) {
        static const int force_wt = getenv("MTK_TRAIN_DMA_WT") ? atoi(getenv("MTK_TRAIN_DMA_WT")) : 0;  // A/B
        // measured (2,048 and 8,192 rows x 1,024 x 1,024): the 128 x 128 shape is no faster in the loop and slower at 128 workgroups
        const bool big = force_wt == 2;
        // a 128-wide B tile reads rows up to the next multiple of 128 past N: the caller's `dma` covers multiples of 64 only
        if (big && (g.N % 128 == 0 || g.dma == 2)) return g.drop_thresh != 0 ? launch_c<4, 1, 2, true>(g, g128, s) : launch_c<4, 1, 2>(g, g128, s);
        return g.drop_thresh != 0 ? launch_c<2, 1, 1, true>(g, g64, s) : launch_c<2, 1, 1>(g, g64, s);
    }
    if (wg128 >= 512 && !g.tile64) {
        if (g.bf16 && !old_bf) launch_b<AT, BT, 2>(g, g128, s);
        else if (g.drop_thresh != 0 && !AT && !BT) {
            if (g.bf16) hipLaunchKernelGGL((k_gemm2<false, false, 2, true, true>), g128, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((k_gemm2<false, false, 2, false, true>), g128, dim3(256), 0, s, g);
        }
        else if (g.bf16) hipLaunchKernelGGL((k_gemm2<AT, BT, 2, true>), g128, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_gemm2<AT, BT, 2, false>), g128, dim3(256), 0, s, g);
    } else {
        if (g.bf16 && !old_bf) launch_b<AT, BT, 1>(g, g64, s);
        else if (g.drop_thresh != 0 && !AT && !BT) {
            if (g.bf16) hipLaunchKernelGGL((k_gemm2<false, false, 1, true, true>), g64, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((k_gemm2<false, false, 1, false, true>), g64, dim3(256), 0, s, g);
        }
        else if (g.bf16) hipLaunchKernelGGL((k_gemm2<AT, BT, 1, true>), g64, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_gemm2<AT, BT, 1, false>), g64, dim3(256), 0, s, g);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// bias gradient of a column-major dY [n][ld]: one workgroup per feature
__global__ __launch_bounds__(256) void k_rowsum_cm(const float* dy, int64_t ld, int64_t rows, float* db) {
    __shared__ float part[256];
    const float* src = dy + (int64_t)blockIdx.x * ld;
    float acc = 0.0f;
    for (int64_t r = threadIdx.x; r < rows; r += 256) acc += src[r];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) db[blockIdx.x] = part[0];
}
// d emb[t[row]][:] += d act0[row][:]   (the time embedding is added to the proj output, MODEL:608-611)
// One wave per row (blockDim = 64 x 4): the row's timestep is a wave-uniform load and the lanes walk its columns — no
// 64-bit division per element (the flat-index version spent 51 us on 4096 x 128 values).
template <bool S16>  // S16: d act0 is stored as bf16 (bf16 training mode)
__device__ __forceinline__ float dact_at(const void* dact, int64_t i) {
    if (S16) return __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(dact)[i] << 16);
    return reinterpret_cast<const float*>(dact)[i];
}
template <bool S16>
__global__ __launch_bounds__(256) void k_scatter_emb(const void* dact, int64_t ld, const int64_t* t, int64_t rows, int n, float* demb, int emb_ld, int T) {
    for (int64_t r = (int64_t)blockIdx.x * 4 + threadIdx.y; r < rows; r += (int64_t)gridDim.x * 4) {
        float* dst = demb + (int64_t)mtk_tclamp(t[r], T) * emb_ld;
        for (int c = threadIdx.x; c < n; c += 64) unsafeAtomicAdd(&dst[c], dact_at<S16>(dact, r * ld + c));
    }
}
// dst[i] = sum over the `splits` slices of a slab region, slices added in index order (run-to-run identical), 4 floats per
// thread; one launch serves every weight / bias gradient of the iteration (blockIdx.y = segment).
constexpr int kMaxSegs = MTK_MAX_SEGS;
typedef MtkSegs ReduceSegs;  // mtk_common.h: n4 = floats / 4 per slice (= slice stride / 4)
__global__ __launch_bounds__(256) void k_reduce_splits(ReduceSegs r) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int sg = blockIdx.y;
    const int64_t n4 = r.n4[sg];
    const float* slab = r.slab[sg];
    float* dst = r.dst[sg];
    const int splits = r.splits[sg];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        v4f acc = *reinterpret_cast<const v4f*>(slab + 4 * i);
        int z = 1;
        for (; z + 8 <= splits; z += 8) {  // eight slices' loads in flight together, added in slice order
            v4f v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const v4f*>(slab + ((int64_t)(z + j) * n4 + i) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (; z < splits; ++z) acc += *reinterpret_cast<const v4f*>(slab + ((int64_t)z * n4 + i) * 4);
        *reinterpret_cast<v4f*>(dst + 4 * i) = acc;
    }
}
// Deterministic scatter-add of d act0 into the [T, dim_t] time-embedding gradient: 64-bit fixed-point integer atomics
// (integer addition is associative, so the result does not depend on the order the rows arrive in), 2^-36 resolution.
constexpr double kFixScale = 68719476736.0;  // 2^36: |sum| < 1.3e8, resolution 1.5e-11
template <bool S16>
__global__ __launch_bounds__(256) void k_scatter_emb_fixed(const void* dact, int64_t ld, const int64_t* t, int64_t rows, int n, long long* acc, int emb_ld, int T) {
    for (int64_t r = (int64_t)blockIdx.x * 4 + threadIdx.y; r < rows; r += (int64_t)gridDim.x * 4) {
        long long* dst = acc + (int64_t)mtk_tclamp(t[r], T) * emb_ld;
        for (int c = threadIdx.x; c < n; c += 64)
            atomicAdd(reinterpret_cast<unsigned long long*>(&dst[c]), (unsigned long long)__double2ll_rn((double)dact_at<S16>(dact, r * ld + c) * kFixScale));
    }
}
// ... and leaves the accumulator zero for the next iteration (it is zeroed once, when allocated)
__global__ void k_fixed_to_float(long long* acc, float* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        dst[i] = (float)((double)acc[i] / kFixScale);
        acc[i] = 0;
    }
}
// d label_embedding[y][c] = (sum over the rows labelled y of d act0[row][c]) silu'(label_embedding[y][c]); accumulator left zero
__global__ void k_label_grad(long long* acc, const float* w, float* dw, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = w[i], sg = 1.0f / (1.0f + expf(-v));
        dw[i] = (float)((double)acc[i] / kFixScale) * (sg * (1.0f + v * (1.0f - sg)));
        acc[i] = 0;
    }
}
// means[0] = mean of lm[0..B), means[1] = mean of lg[0..B) (0 for a null vector): one workgroup, fp64 partial sums per thread over a
// fixed stride, added through LDS in a fixed order (run-to-run identical)
__global__ __launch_bounds__(1024) void k_loss_means(const float* lm, const float* lg, int64_t B, float* means) {
    __shared__ double sh[2][1024];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += 1024) {
        if (lm) a += (double)lm[i];
        if (lg) b += (double)lg[i];
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + w];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        means[0] = (float)(sh[0][0] / (double)B);
        means[1] = (float)(sh[1][0] / (double)B);
    }
}
__global__ void k_silu(const float* z, float* y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = z[i];
        y[i] = v / (1.0f + expf(-v));
    }
}
// dz = dy * silu'(z),  silu'(z) = sig (1 + z (1 - sig))
__global__ void k_silu_bwd(const float* z, float* dy, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = z[i], sg = 1.0f / (1.0f + expf(-v));
        dy[i] *= sg * (1.0f + v * (1.0f - sg));
    }
}
// MODEL:123-144 sinusoid table (same expression as k_sinusoid of mtk_mlp_f32.hip)
__global__ void k_sin_tab(float* dst, int Tpad, int ld, int T, int dim, const float* freqs) {
    const int half = dim / 2;
    const int64_t n = (int64_t)Tpad * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / ld), c = (int)(i % ld);
        float v = 0.0f;
        if (t < T && c < 2 * half) {
            const float arg = (float)t * freqs[c < half ? c : c - half];
            v = c < half ? cosf(arg) : sinf(arg);
        }
        dst[i] = v;
    }
}

// torch.optim.AdamW (single-tensor path) followed by the EMA of TRAINER:138-155, one pass over the image.
struct AdamArgs {
    uint16_t* p16;  // optional bf16 copy of the updated weights (bf16 training mode)
    float *p, *m, *v, *ema;
    const float* g;
    int64_t n;                  // elements this launch updates: two ranges of the image, [beg0, beg0 + len0) then [beg1, beg1 + n - len0)
    int64_t beg0, len0, beg1;
    float decay;        // 1 - lr * weight_decay
    float beta1_c;      // 1 - beta1
    float beta2, beta2_c;
    float step_size;    // lr / (1 - beta1^step)
    float bc2_sqrt;     // sqrt(1 - beta2^step)
    float eps, ema_rate, ema_c;
};
__global__ __launch_bounds__(256) void k_adamw_ema(AdamArgs a) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < a.n; j += (int64_t)gridDim.x * 256) {
        const int64_t i = j < a.len0 ? a.beg0 + j : a.beg1 + (j - a.len0);
        const float g = a.g[i];
        float p = a.p[i] * a.decay;
        float m = a.m[i];
        m = m + a.beta1_c * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
        const float v = a.v[i] * a.beta2 + a.beta2_c * g * g;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        p = p - a.step_size * (m / denom);
        a.p[i] = p;
        if (a.p16) {
            const __bf16 e = (__bf16)p;
            a.p16[i] = *reinterpret_cast<const uint16_t*>(&e);
        }
        a.m[i] = m;
        a.v[i] = v;
        a.ema[i] = a.ema[i] * a.ema_rate + a.ema_c * p;
    }
}

// The same update per segment of the image (blockIdx.y), four elements per thread, with the split-batch partials of the segment's
// gradient added on the way in (slice order, as k_reduce_splits adds them: the same sums) and the sum written to the gradient image.
// splits = 0: the gradient is read from the image.  Same arithmetic per element as k_adamw_ema.
__global__ __launch_bounds__(256) void k_adamw_ema_segs(AdamArgs a, float* gw, MtkSegs r) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int sg = blockIdx.y;
    const int64_t n4 = r.n4[sg];
    const float* slab = r.slab[sg];
    const int splits = r.splits[sg];
    const int64_t base = r.dst[sg] - gw;  // element offset of the segment in an image
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = base + 4 * i;
        v4f g;
        if (splits > 0) {
            g = *reinterpret_cast<const v4f*>(slab + 4 * i);
            int z = 1;
            for (; z + 8 <= splits; z += 8) {
                v4f q[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const v4f*>(slab + ((int64_t)(z + j) * n4 + i) * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) g += q[j];
            }
            for (; z < splits; ++z) g += *reinterpret_cast<const v4f*>(slab + ((int64_t)z * n4 + i) * 4);
            *reinterpret_cast<v4f*>(gw + e) = g;
        } else {
            g = *reinterpret_cast<const v4f*>(gw + e);
        }
        v4f p = *reinterpret_cast<const v4f*>(a.p + e), m = *reinterpret_cast<const v4f*>(a.m + e), v = *reinterpret_cast<const v4f*>(a.v + e);
        v4f em = *reinterpret_cast<const v4f*>(a.ema + e);
        unsigned short h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float pc = p[c] * a.decay;
            float mc = m[c];
            mc = mc + a.beta1_c * (g[c] - mc);
            const float vc = v[c] * a.beta2 + a.beta2_c * g[c] * g[c];
            const float denom = sqrtf(vc) / a.bc2_sqrt + a.eps;
            pc = pc - a.step_size * (mc / denom);
            p[c] = pc; m[c] = mc; v[c] = vc;
            em[c] = em[c] * a.ema_rate + a.ema_c * pc;
            const __bf16 b = (__bf16)pc;
            h[c] = *reinterpret_cast<const unsigned short*>(&b);
        }
        *reinterpret_cast<v4f*>(a.p + e) = p;
        *reinterpret_cast<v4f*>(a.m + e) = m;
        *reinterpret_cast<v4f*>(a.v + e) = v;
        *reinterpret_cast<v4f*>(a.ema + e) = em;
        if (a.p16) *reinterpret_cast<uint2*>(a.p16 + e) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    }
}

// image <-> caller tensors (unpadded [out][in] / [out])
__global__ void k_unpad(const float* src, int ld, float* dst, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[(i / cols) * ld + (i % cols)];
}
__global__ void k_pad(float* dst, int ld, const float* src, int rows, int cols) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[(i / cols) * ld + (i % cols)] = src[i];
}

__global__ void k_f32_to_bf16(const float* src, uint16_t* dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const __bf16 e = (__bf16)src[i];
        dst[i] = *reinterpret_cast<const uint16_t*>(&e);
    }
}

// bf16 W^T of the layers the backward takes input gradients through: 64 x 64 tiles through LDS, one launch for all of them
constexpr int kMaxTr = 40;
struct TrSegs {
    int64_t off[kMaxTr];  // element offset of the layer's weights in an image
    int rows[kMaxTr], cols[kMaxTr];  // out_pad, in_pad (multiples of 32)
    int tile0[kMaxTr + 1];           // first tile of each layer in the flat tile list
    int n;
};
__global__ __launch_bounds__(256) void k_w16_transpose(const float* img, uint16_t* w16t, TrSegs sg) {
    __shared__ uint16_t tile[32][34];
    const int total = sg.tile0[sg.n];
    for (int tix = blockIdx.x; tix < total; tix += gridDim.x) {
        int l = 0;
        while (tix >= sg.tile0[l + 1]) ++l;
        const int tpr = sg.cols[l] / 32, tl = tix - sg.tile0[l], r0 = (tl / tpr) * 32, c0 = (tl % tpr) * 32;
        const float* src = img + sg.off[l];
        uint16_t* dst = w16t + sg.off[l];
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 e = (__bf16)src[(int64_t)(r0 + ty + 8 * j) * sg.cols[l] + c0 + tx];
            tile[ty + 8 * j][tx] = *reinterpret_cast<const uint16_t*>(&e);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(int64_t)(c0 + ty + 8 * j) * sg.rows[l] + r0 + tx] = tile[tx][ty + 8 * j];
        __syncthreads();
    }
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

std::vector<MlpLayer*> layer_list(mtk_ctx* c) {
    std::vector<MlpLayer*> v{&c->proj, &c->temb0, &c->temb2};
    if (c->train.n_label > 0) v.push_back(&c->train.label_l);  // with the time-embedding MLP: same gradient bucket, same point of the backward
    for (auto& l : c->blocks) v.push_back(&l);
    v.push_back(&c->head);
    return v;
}

void free_acts(TrainState& tr) {
    auto fr = [](auto*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(tr.xin); fr(tr.xt_idx); fr(tr.tdraw); fr(tr.out); fr(tr.dout); fr(tr.slab); fr(tr.slab_rs); fr(tr.xin16); fr(tr.dout16);
    tr.t_drawn = false;
    for (auto& p : tr.dact) fr(p);
    tr.dact.clear();
    tr.slab_elems = tr.slab_rs_elems = 0;
    for (auto& p : tr.act) fr(p);
    tr.act.clear();
    tr.cap = 0;
    for (auto& p : tr.act16) fr(p);
    tr.act16.clear();
    for (auto& p : tr.act16t) fr(p);
    tr.act16t.clear();
    for (auto& p : tr.dact16) fr(p);
    for (auto& p : tr.dact16t) fr(p);
    tr.dact16.clear();
    tr.dact16t.clear();
    tr.cap16 = 0;
}

int kchunk_for(int64_t K, int tiles);
int splits_for(int64_t K, int tiles) {
    const int ch = kchunk_for(K, tiles);
    return (int)((K + ch - 1) / ch);
}

int ensure_acts(mtk_ctx* c, int64_t Mpad, hipStream_t s) {
    TrainState& tr = c->train;
    if (Mpad <= tr.cap) return MTK_OK;
    {   // partials left for the optimizer pass live in the slabs that are about to be freed
        const int rc = materialize_grads(c, s);
        if (rc) return rc;
    }
    MTK_HIP(hipStreamSynchronize(s));
    free_acts(tr);
    ++c->ws_gen;
    const int Dp = round_up(c->D, 128);
    int maxpad = c->proj.out_pad;
    for (auto& l : c->blocks) maxpad = l.out_pad > maxpad ? l.out_pad : maxpad;
    MTK_HIP(hipMalloc(&tr.xin, sizeof(float) * (size_t)Dp * Mpad));
    MTK_HIP(hipMalloc(&tr.xt_idx, sizeof(int32_t) * (size_t)(c->C > 0 ? c->C : 1) * Mpad));
    MTK_HIP(hipMalloc(&tr.tdraw, sizeof(int64_t) * (size_t)Mpad));
    MTK_HIP(hipMalloc(&tr.out, sizeof(float) * (size_t)Dp * Mpad));
    MTK_HIP(hipMalloc(&tr.dout, sizeof(float) * (size_t)Dp * Mpad));
    MTK_HIP(hipMemsetAsync(tr.xin, 0, sizeof(float) * (size_t)Dp * Mpad, s));
    MTK_HIP(hipMemsetAsync(tr.dout, 0, sizeof(float) * (size_t)Dp * Mpad, s));
    MTK_HIP(hipMalloc(&tr.xin16, sizeof(uint16_t) * (size_t)Dp * Mpad));
    MTK_HIP(hipMalloc(&tr.dout16, sizeof(uint16_t) * (size_t)Dp * Mpad));
    MTK_HIP(hipMemsetAsync(tr.xin16, 0, sizeof(uint16_t) * (size_t)Dp * Mpad, s));
    MTK_HIP(hipMemsetAsync(tr.dout16, 0, sizeof(uint16_t) * (size_t)Dp * Mpad, s));
    tr.xin_rows = tr.dout_rows = 0;
    tr.act.assign(1 + c->blocks.size(), nullptr);
    tr.dact.assign(1 + c->blocks.size(), nullptr);
    for (size_t i = 0; i <= c->blocks.size(); ++i) {
        const size_t w = i == 0 ? c->proj.out_pad : c->blocks[i - 1].out_pad;
        MTK_HIP(hipMalloc(&tr.act[i], sizeof(float) * (size_t)Mpad * w));
        MTK_HIP(hipMalloc(&tr.dact[i], sizeof(float) * (size_t)Mpad * w));
    }
    // split-K slabs of the weight-gradient GEMMs (deterministic reductions): every layer has its own region (splits x
    // layer), so that ONE reduction kernel at the end of the backward pass adds all of them
    size_t need = 0, need_rs = 0;
    // splits_for is not monotonic in K (the chunk is rounded up to TK, so a slightly SMALLER batch can need one more split):
    // size for the most splits any batch up to the capacity can ask for, ceil(512 / tiles) (kchunk_for's target)
    auto account = [&](const MlpLayer& l, int64_t K) {
        const int tiles = (l.out_pad / TM) * ((l.in_pad + TN - 1) / TN);
        const int sp_k = splits_for(K, tiles), sp_max = (512 + tiles - 1) / tiles;
        const int sp = sp_k > sp_max ? sp_k : sp_max;
        need += (size_t)sp * l.out_pad * l.in_pad;
        need_rs += (size_t)sp * l.out_pad;
    };
    account(c->proj, Mpad);
    account(c->head, Mpad);
    for (auto& l : c->blocks) account(l, Mpad);
    account(c->temb0, tr.Tpad);
    account(c->temb2, tr.Tpad);
    MTK_HIP(hipMalloc(&tr.slab, sizeof(float) * need));
    MTK_HIP(hipMalloc(&tr.slab_rs, sizeof(float) * need_rs));
    tr.slab_elems = need;
    tr.slab_rs_elems = need_rs;
    tr.cap = Mpad;
    return MTK_OK;
}

// bf16 mode: the bf16 activation / activation-gradient buffers and the bf16 weight copy (allocated on first use of the mode)
int refresh_w16t(mtk_ctx* c, hipStream_t s) {
    TrainState& tr = c->train;
    if (!tr.w16t) return MTK_OK;
    TrSegs sg;
    sg.n = 0;
    sg.tile0[0] = 0;
    auto add = [&](const MlpLayer& l) {
        sg.off[sg.n] = l.w32 - tr.state; sg.rows[sg.n] = l.out_pad; sg.cols[sg.n] = l.in_pad;
        sg.tile0[sg.n + 1] = sg.tile0[sg.n] + (l.out_pad / 32) * (l.in_pad / 32);
        ++sg.n;
    };
    if ((int)c->blocks.size() + 1 > kMaxTr) MTK_FAIL(MTK_ERR_UNSUPPORTED, "bf16 training: more than %d layers", kMaxTr - 1);
    for (auto& l : c->blocks) add(l);
    add(c->head);
    const int total = sg.tile0[sg.n];
    hipLaunchKernelGGL(k_w16_transpose, dim3((unsigned)(total < 4096 ? total : 4096)), dim3(256), 0, s, tr.state, tr.w16t, sg);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}
int refresh_w16(mtk_ctx* c, hipStream_t s) {
    TrainState& tr = c->train;
    if (!tr.w16) return MTK_OK;
    hipLaunchKernelGGL(k_f32_to_bf16, dim3(1024), dim3(256), 0, s, tr.state, tr.w16, (int64_t)tr.n_img);
    MTK_HIP(hipGetLastError());
    return refresh_w16t(c, s);
}
int ensure_acts16(mtk_ctx* c, int64_t Mpad, hipStream_t s) {
    TrainState& tr = c->train;
    if (!tr.w16) {
        MTK_HIP(hipStreamSynchronize(s));
        MTK_HIP(hipMalloc(&tr.w16, sizeof(uint16_t) * tr.n_img));
        MTK_HIP(hipMalloc(&tr.w16t, sizeof(uint16_t) * tr.n_img));
        ++c->ws_gen;
        const int rc = refresh_w16(c, s);
        if (rc) return rc;
    }
    if (Mpad <= tr.cap16) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    auto fr = [](uint16_t*& p) { if (p) (void)hipFree(p); p = nullptr; };
    for (auto* v : {&tr.act16, &tr.act16t, &tr.dact16, &tr.dact16t}) {
        for (auto& p : *v) fr(p);
        v->assign(1 + c->blocks.size(), nullptr);
    }
    ++c->ws_gen;
    for (size_t i = 0; i <= c->blocks.size(); ++i) {
        const size_t bytes = sizeof(uint16_t) * (size_t)Mpad * (i == 0 ? c->proj.out_pad : c->blocks[i - 1].out_pad);
        for (auto* v : {&tr.act16, &tr.act16t, &tr.dact16, &tr.dact16t}) MTK_HIP(hipMalloc(&(*v)[i], bytes));
    }
    tr.cap16 = Mpad;
    return MTK_OK;
}

// emb[t] = Linear2(SiLU(Linear0(sinusoid(t)))) from the CURRENT weights, keeping the intermediates
int refresh_emb(mtk_ctx* c, hipStream_t s) {
    TrainState& tr = c->train;
    const int tp = c->emb_ld;
    int rc;
    G2 g{};
    g.M = tr.Tpad; g.N = tp; g.K = tp; g.kchunk = tp;
    g.A = tr.sin_tab; g.lda = tp; g.B = c->temb0.w32; g.ldb = tp; g.C = tr.z0_tab; g.c_rs = tp; g.c_cs = 1; g.bias = c->temb0.b32;
    if ((rc = launch<false, false>(g, s))) return rc;
    hipLaunchKernelGGL(k_silu, dim3(256), dim3(256), 0, s, tr.z0_tab, tr.e1_tab, (int64_t)tr.Tpad * tp);
    g.A = tr.e1_tab; g.B = c->temb2.w32; g.C = c->d_emb; g.bias = c->temb2.b32;
    if ((rc = launch<false, false>(g, s))) return rc;
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

int kchunk_for(int64_t K, int tiles) {
    // enough workgroups for 256 CUs: tiles * splits >= ~512, chunks of at least 256 reduction steps
    int64_t splits = (512 + tiles - 1) / tiles;
    int64_t ch = (K + splits - 1) / splits;
    if (ch < 256) ch = 256;
    ch = (ch + TK - 1) / TK * TK;
    return (int)(ch > K ? K : ch);
}

}  // namespace

int mtk_reserve_train(mtk_ctx* c, int64_t rows, hipStream_t s) {  // mtk_ctx_reserve: activations of a training iteration
    if (!c->train.attached || rows <= 0) return MTK_OK;
    if (rows > (1 << 22)) rows = 1 << 22;
    const int rc = ensure_acts(c, (rows + TM - 1) / TM * TM, s);
    if (rc || !c->train.bf16) return rc;
    return ensure_acts16(c, (rows + TM - 1) / TM * TM, s);
}

void mtk_free_train(mtk_ctx* c) {
    TrainState& tr = c->train;
    auto fr = [](float*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(tr.sin_tab); fr(tr.z0_tab); fr(tr.e1_tab); fr(tr.demb); fr(tr.dtab);
    if (tr.demb_fix) (void)hipFree(tr.demb_fix);
    tr.demb_fix = nullptr;
    if (tr.dlab_fix) (void)hipFree(tr.dlab_fix);
    tr.dlab_fix = nullptr;
    tr.yvec = nullptr;
    tr.loss_means = nullptr;
    tr.segs_pending = false;
    tr.defer_reduce = false;
    tr.n_label = 0;
    tr.label_l = MlpLayer();
    free_acts(tr);
    if (tr.side) {
        (void)hipStreamSynchronize(tr.side);
        (void)hipStreamDestroy(tr.side);
        (void)hipEventDestroy(tr.ev_fork);
        (void)hipEventDestroy(tr.ev_join);
        tr.side = nullptr;
        tr.ev_fork = tr.ev_join = nullptr;
    }
    if (tr.w16) (void)hipFree(tr.w16);
    if (tr.w16t) (void)hipFree(tr.w16t);
    tr.w16 = nullptr;
    tr.w16t = nullptr;
    tr.layers.clear();
    tr.attached = false;
    tr.state = nullptr;
    tr.n_img = 0;
}

extern "C" int64_t mtk_train_image_elems(mtk_ctx* c) {
    if (!c || !c->have_weights) return 0;
    int64_t n = 0;
    for (MlpLayer* l : layer_list(c)) n += (int64_t)l->out_pad * l->in_pad + l->out_pad;
    return n;
}

extern "C" int mtk_train_attach(mtk_ctx* c, float* state, const float* freqs_host, mtk_stream stream) {
    if (!c || !state || !freqs_host) MTK_FAIL(MTK_ERR_ARG, "mtk_train_attach: null argument");
    if (!c->have_weights) MTK_FAIL(MTK_ERR_STATE, "mtk_train_attach: call mtk_mlp_set_weights first");
    if (c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_attach: a training image is already attached (set the weights again to detach)");
    if (c->d_label) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_train_attach: label-conditioned denoisers (mtk_mlp_set_label_embedding) are forward only");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    MTK_HIP(hipStreamSynchronize(s));
    TrainState& tr = c->train;
    const size_t n = (size_t)mtk_train_image_elems(c);  // with the label embedding's layer when mtk_train_set_label_classes asked for one (zero until imported)
    tr.layers.clear();
    size_t off = 0;
    for (MlpLayer* l : layer_list(c)) {
        TrainLayerRef r;
        r.l = l;
        r.w_off = off;
        off += (size_t)l->out_pad * l->in_pad;
        r.b_off = off;
        off += (size_t)l->out_pad;
        tr.layers.push_back(r);
    }
    // image 0 <- current weights; image 4 (EMA) <- the same; gradients, m, v <- 0
    for (auto& r : tr.layers) {
        if (r.l == &tr.label_l) {
            MTK_HIP(hipMemsetAsync(state + r.w_off, 0, sizeof(float) * ((size_t)r.l->out_pad * r.l->in_pad + r.l->out_pad), s));
            continue;
        }
        MTK_HIP(hipMemcpyAsync(state + r.w_off, r.l->w32, sizeof(float) * (size_t)r.l->out_pad * r.l->in_pad, hipMemcpyDeviceToDevice, s));
        MTK_HIP(hipMemcpyAsync(state + r.b_off, r.l->b32, sizeof(float) * (size_t)r.l->out_pad, hipMemcpyDeviceToDevice, s));
    }
    MTK_HIP(hipMemsetAsync(state + n, 0, sizeof(float) * 3 * n, s));
    MTK_HIP(hipMemcpyAsync(state + 4 * n, state, sizeof(float) * n, hipMemcpyDeviceToDevice, s));
    MTK_HIP(hipStreamSynchronize(s));
    for (auto& r : tr.layers) {
        if (r.l->owned) {
            mtk_release(c, r.l->w32);
            mtk_release(c, r.l->b32);
        }
        r.l->w32 = state + r.w_off;
        r.l->b32 = state + r.b_off;
        r.l->owned = false;
    }
    tr.state = state;
    tr.n_img = n;
    // time-embedding intermediates
    const int tp = c->emb_ld;
    tr.Tpad = round_up(c->T, TM);
    const size_t tb = sizeof(float) * (size_t)tr.Tpad * tp;
    MTK_HIP(hipMalloc(&tr.sin_tab, tb));
    MTK_HIP(hipMalloc(&tr.z0_tab, tb));
    MTK_HIP(hipMalloc(&tr.e1_tab, tb));
    MTK_HIP(hipMalloc(&tr.demb, tb));
    MTK_HIP(hipMalloc(&tr.dtab, tb));
    MTK_HIP(hipMalloc(&tr.demb_fix, sizeof(long long) * (size_t)tr.Tpad * tp));
    MTK_HIP(hipMemsetAsync(tr.demb_fix, 0, sizeof(long long) * (size_t)tr.Tpad * tp, s));
    if (tr.n_label > 0) {
        MTK_HIP(hipMalloc(&tr.dlab_fix, sizeof(long long) * (size_t)tr.label_l.out_pad * tp));
        MTK_HIP(hipMemsetAsync(tr.dlab_fix, 0, sizeof(long long) * (size_t)tr.label_l.out_pad * tp, s));
    }
    if (!tr.side) {
        MTK_HIP(hipStreamCreateWithFlags(&tr.side, hipStreamNonBlocking));
        MTK_HIP(hipEventCreateWithFlags(&tr.ev_fork, hipEventDisableTiming));
        MTK_HIP(hipEventCreateWithFlags(&tr.ev_join, hipEventDisableTiming));
    }
    float* d_freq = nullptr;
    MTK_HIP(hipMalloc(&d_freq, sizeof(float) * (c->dim_t / 2)));
    MTK_HIP(hipMemcpyAsync(d_freq, freqs_host, sizeof(float) * (c->dim_t / 2), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sin_tab, dim3(256), dim3(256), 0, s, tr.sin_tab, tr.Tpad, tp, c->T, c->dim_t, d_freq);
    int rc = refresh_emb(c, s);
    MTK_HIP(hipStreamSynchronize(s));
    (void)hipFree(d_freq);
    if (rc) return rc;
    c->fused_ok = false;  // the packed bf16 image no longer tracks the weights
    tr.attached = true;
    return MTK_OK;
}

// One training iteration's device work, in two phases that may be issued by separate calls (the autograd seam of the
// drop-in's mixed_loss): FORWARD = noising, denoiser forward with every activation kept, per-record losses;
// BACKWARD = d[(w_multi sum_rows loss_multi + w_gauss sum_rows loss_gauss)] / d weights into image 1, from the
// activations the last FORWARD on this context left behind (x0 / t / eps / seed / row0 must be that call's).
static int train_phases(mtk_ctx* c, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps, const float* u,
                        double w_multi, double w_gauss, float* loss_multi, float* loss_gauss, uint64_t seed, uint64_t row0,
                        bool do_fwd, bool do_bwd, mtk_stream stream, const float* w_dev = nullptr) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_step: null context");
    if (!c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_step: call mtk_train_attach first");
    if (B <= 0 || !x0) MTK_FAIL(MTK_ERR_ARG, "mtk_train_step: null pointer or B <= 0");
    // t = null: the forward phase draws the timesteps itself (uniform over [0, T), Philox: mtk_draw_t) and keeps them for the
    // backward phase and for mtk_train_timesteps
    if (!t && !do_fwd && !c->train.t_drawn) MTK_FAIL(MTK_ERR_ARG, "mtk_train_backward: t is null and the last forward phase did not draw its timesteps");
    if (B > (1 << 22)) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_train_step: B = %lld above 4194304 rows per call", (long long)B);
    if (x0_ld < c->Nn + c->C) MTK_FAIL(MTK_ERR_ARG, "mtk_train_step: x0_ld too small");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    TrainState& tr = c->train;
    const int64_t Mpad = (B + TM - 1) / TM * TM;
    int rc;
    if (do_bwd && !do_fwd && (Mpad > tr.cap || tr.fwd_rows != B))
        MTK_FAIL(MTK_ERR_STATE, "mtk_train_backward: the last mtk_train_forward on this context was for %lld rows, not %lld", (long long)tr.fwd_rows, (long long)B);
    if ((rc = ensure_acts(c, Mpad, s))) return rc;
    const int64_t ld = tr.cap;  // leading dimension of the column-major buffers
    const int Dp = round_up(c->D, 128);
    const size_t n = tr.n_img;
    float* grads = tr.state + n;

    // bf16 mode with bf16-stored operands (see TrainState::w16); MTK_TRAIN_BF16_OLD keeps the fp32-stored operands
    static const bool old_bf = getenv("MTK_TRAIN_BF16_OLD") != nullptr;
    const bool st16 = tr.bf16 && !old_bf;
    if (st16 && (rc = ensure_acts16(c, Mpad, s))) return rc;
    auto f16 = [](const uint16_t* p) { return reinterpret_cast<const float*>(p); };
    auto w_of = [&](const MlpLayer& l) { return st16 ? f16(tr.w16 + (l.w32 - tr.state)) : (const float*)l.w32; };
    auto act_of = [&](size_t i) { return st16 ? f16(tr.act16[i]) : (const float*)tr.act[i]; };
    auto dact_of = [&](int i) { return st16 ? f16(tr.dact16[i]) : (const float*)tr.dact[i]; };  // d loss / d act[i]
    auto wt_of = [&](const MlpLayer& l) { return f16(tr.w16t + (l.w32 - tr.state)); };  // st16 only: [in_pad][out_pad]
    // Reduction chunk of a weight-gradient GEMM [M x N] over K rows.  A split costs a partial tile written to the slab and read back
    // by k_reduce_splits, and a 128 x 128 tile with 8 splits of a 2,048-row batch runs 4 k-steps per workgroup: with bf16 operands the
    // 64 x 64 tiles fill the chip with a quarter of the splits (1024 x 1024 over 2,048 rows: 23 -> 14 us, k_reduce_splits 20 -> 8 us).
    auto wgrad_chunk = [&](G2& g) {
        const int t128 = (g.M / TM) * ((g.N + TN - 1) / TN), t64 = (g.M / 64) * ((g.N + 63) / 64);
        g.tile64 = st16 ? 1 : 0;
        g.kchunk = kchunk_for(g.K, st16 ? t64 : t128);
        if (st16 && g.kchunk < g.K) g.kchunk = (g.kchunk + 63) / 64 * 64;  // k_gemm2c walks 64 rows of the batch per k-step
    };
    if (do_fwd) {
    tr.fwd_rows = B;
    if (B < tr.xin_rows) {
        MTK_HIP(hipMemsetAsync(tr.xin, 0, sizeof(float) * (size_t)Dp * ld, s));
        MTK_HIP(hipMemsetAsync(tr.xin16, 0, sizeof(uint16_t) * (size_t)Dp * ld, s));
    }
    tr.xin_rows = B;

    // ---- forward, activations kept
    // xt_idx is addressed [C][B] by the kernels (row stride B)
    if ((rc = mtk_q_sample_t(c, B, x0, x0_ld, t, eps, u, tr.xin, 1, ld, tr.xt_idx, seed, row0, t ? nullptr : tr.tdraw, st16 ? tr.xin16 : nullptr, s))) return rc;
    tr.t_drawn = !t;
    if (!t) t = tr.tdraw;
    {
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = (int)Mpad; g.N = c->proj.out_pad; g.K = c->proj.in_pad; g.kchunk = g.K;
        g.A = tr.xin; g.lda = ld; g.B = w_of(c->proj); g.ldb = c->proj.in_pad; g.b16 = st16;
        g.C = const_cast<float*>(act_of(0)); g.c16 = st16; g.c_rs = c->proj.out_pad; g.c_cs = 1; g.bias = c->proj.b32;
        if (st16) { g.Ct = tr.act16t[0]; g.ct_ld = Mpad; }
        g.emb = c->d_emb; g.emb_ld = c->emb_ld; g.tvec = t; g.rows_valid = B; g.T = c->T;
        if (tr.n_label > 0) {
            if (!tr.yvec) MTK_FAIL(MTK_ERR_STATE, "mtk_train_step: the image has a label embedding (mtk_train_set_label_classes): hand over the batch's labels with mtk_train_set_labels first");
            g.lab = tr.label_l.w32; g.lab_ld = tr.label_l.in_pad; g.n_lab = tr.n_label; g.yvec = tr.yvec;
        }
        if ((rc = launch<true, false>(g, s))) return rc;
    }
    for (size_t i = 0; i < c->blocks.size(); ++i) {
        MlpLayer& l = c->blocks[i];
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = (int)Mpad; g.N = l.out_pad; g.K = l.in_pad; g.kchunk = g.K;
        g.A = act_of(i); g.a16 = st16; g.lda = l.in_pad; g.B = w_of(l); g.b16 = st16; g.ldb = l.in_pad;
        g.C = const_cast<float*>(act_of(i + 1)); g.c16 = st16; g.c_rs = l.out_pad; g.c_cs = 1; g.bias = l.b32; g.relu = 1;
        if (st16) { g.Ct = tr.act16t[i + 1]; g.ct_ld = Mpad; g.dma = 2; }  // B = W: out_pad rows, a multiple of 128
        if (tr.drop_p > 0.0f) {
            const double th = (double)tr.drop_p * 4294967296.0;
            g.drop_thresh = th >= 4294967295.0 ? 4294967295u : (uint32_t)th;
            g.drop_scale = 1.0f / (1.0f - tr.drop_p); g.drop_seed = tr.drop_seed; g.drop_row0 = row0; g.drop_layer = (int)i;
        }
        if ((rc = launch<false, false>(g, s))) return rc;
    }
    {
        MlpLayer& l = c->head;
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = (int)Mpad; g.N = l.out; g.K = l.in_pad; g.kchunk = g.K;
        g.A = act_of(c->blocks.size()); g.a16 = st16; g.lda = l.in_pad; g.B = w_of(l); g.b16 = st16; g.ldb = l.in_pad;
        g.C = tr.out; g.c_rs = 1; g.c_cs = ld; g.bias = l.b32;
        g.dma = st16 ? 2 : 0;  // B = W: out_pad rows
        if ((rc = launch<false, false>(g, s))) return rc;
    }
    }
    if (!t) t = tr.tdraw;  // backward phase of a forward that drew them
    auto means = [&](hipStream_t q) {  // batch means of the two per-record loss vectors, for the caller that asked (mtk_train_set_loss_means)
        if (tr.loss_means && (loss_multi || loss_gauss))
            hipLaunchKernelGGL(k_loss_means, dim3(1), dim3(1024), 0, q, (const float*)loss_multi, (const float*)loss_gauss, B, tr.loss_means);
    };
    if (!do_bwd) {  // losses only (the gradient scratch it also fills is rewritten by the backward phase)
        if (B > tr.dout_rows) tr.dout_rows = B;
        if ((rc = mtk_launch_loss_grad(c, B, tr.out, ld, x0, x0_ld, tr.xt_idx, eps, t, loss_multi, loss_gauss, tr.dout, 0.0, 0.0, nullptr, seed, row0, s))) return rc;
        means(s);
        MTK_HIP(hipGetLastError());
        return MTK_OK;
    }
    // deterministic mode stores every element of the gradient image (the split-batch partials are summed into it, the head
    // bias row sums cover its padding) and of d emb; the atomic mode accumulates into both
    if (!tr.deterministic) {
        MTK_HIP(hipMemsetAsync(grads, 0, sizeof(float) * n, s));
        MTK_HIP(hipMemsetAsync(tr.demb, 0, sizeof(float) * (size_t)tr.Tpad * c->emb_ld, s));
    }
    if (B < tr.dout_rows) {
        MTK_HIP(hipMemsetAsync(tr.dout, 0, sizeof(float) * (size_t)Dp * ld, s));
        MTK_HIP(hipMemsetAsync(tr.dout16, 0, sizeof(uint16_t) * (size_t)Dp * ld, s));
    }
    tr.dout_rows = B;
    // ---- loss and d loss / d out (column-major, rows >= B and features >= D stay zero)
    if ((rc = mtk_launch_loss_grad(c, B, tr.out, ld, x0, x0_ld, tr.xt_idx, eps, t, loss_multi, loss_gauss, tr.dout,
                                   w_multi, w_gauss, w_dev, seed, row0, s, st16 ? tr.dout16 : nullptr)))
        return rc;

    // ---- backward
    auto ref_of = [&](MlpLayer* l) -> TrainLayerRef& {
        for (auto& r : tr.layers) if (r.l == l) return r;
        return tr.layers[0];
    };
    // weight-gradient GEMM, split-K over the batch: deterministic (partial tiles to slabs, added in order) or fp32 atomics
    ReduceSegs segs;
    segs.n = 0;
    tr.segs_pending = false;  // partials of an iteration that was never applied: this backward overwrites the slabs
    size_t slab_used = 0, slab_rs_used = 0;
    auto wgrad = [&](G2& g, auto launcher, float* dw, float* db) -> int {
        const int splits = (g.K + g.kchunk - 1) / g.kchunk;
        const size_t elems = (size_t)g.M * g.N;
        if (!tr.deterministic) {
            g.C = dw; g.atomic = 1; g.rowsum = db;
            return launcher(g);
        }
        if (slab_used + (size_t)splits * elems > tr.slab_elems || slab_rs_used + (size_t)splits * g.M > tr.slab_rs_elems || segs.n + 2 > kMaxSegs)
            MTK_FAIL(MTK_ERR_STATE, "mtk_train_step: split-K slab too small (%d x %zu)", splits, elems);
        g.C = dw; g.atomic = 0; g.rowsum = nullptr;
        g.slab = tr.slab + slab_used; g.slab_stride = (int64_t)elems; g.rowsum_slab = db ? tr.slab_rs + slab_rs_used : nullptr;
        segs.dst[segs.n] = dw; segs.slab[segs.n] = g.slab; segs.splits[segs.n] = splits; segs.n4[segs.n] = (int64_t)elems / 4;
        ++segs.n;
        slab_used += (size_t)splits * elems;
        if (db) {
            segs.dst[segs.n] = db; segs.slab[segs.n] = g.rowsum_slab; segs.splits[segs.n] = splits; segs.n4[segs.n] = (int64_t)g.M / 4;
            ++segs.n;
            slab_rs_used += (size_t)splits * g.M;
        }
        return launcher(g);
    };
    // Gradient buckets in the order the backward finishes them: 0 head, 1.. blocks last to first, then proj, then the
    // time-embedding MLP.  With bucket events set (data-parallel training), a bucket's split-batch partials are added
    // right away (instead of by the one launch at the end) and its event is recorded: the caller's all-reduce of that
    // slice of the gradient image runs on another stream under the rest of the backward.
    int seg_mark = 0, bucket = 0;
    const bool bucketed = !tr.bucket_events.empty();
    auto bucket_done = [&]() -> int {
        if (!bucketed) return MTK_OK;
        if (segs.n > seg_mark) {
            ReduceSegs part;
            part.n = segs.n - seg_mark;
            for (int i = 0; i < part.n; ++i) {
                part.dst[i] = segs.dst[seg_mark + i]; part.slab[i] = segs.slab[seg_mark + i];
                part.splits[i] = segs.splits[seg_mark + i]; part.n4[i] = segs.n4[seg_mark + i];
            }
            hipLaunchKernelGGL(k_reduce_splits, dim3(256, (unsigned)part.n), dim3(256), 0, s, part);
            seg_mark = segs.n;
        }
        MTK_HIP(hipEventRecord(tr.bucket_events[bucket++], s));
        return MTK_OK;
    };
    // Order: the input-gradient chain first (head, hidden layers last to first: each step needs the one before), every
    // d loss / d act[i] kept; then the weight gradients, which depend on nothing but those - and, on the side stream under them,
    // the time-embedding MLP's backward (scatter of d act[0] into the table, three small GEMMs), which used to run after the
    // last of them with the chip idle around it.
    const int nb = (int)c->blocks.size();
    const float drop_gain = tr.drop_p > 0.0f ? 1.0f / (1.0f - tr.drop_p) : 1.0f;  // d [relu(x) keep / (1 - p)] / dx where the saved value is > 0
    {   // head: d act[nb] = dout^T W, masked by act[nb] > 0
        MlpLayer& l = c->head;
        G2 d{};
        d.bf16 = tr.bf16;
        d.M = (int)Mpad; d.N = l.in_pad; d.K = l.out_pad; d.kchunk = d.K;
        d.A = tr.dout; d.lda = ld; d.B = w_of(l); d.b16 = st16; d.ldb = l.in_pad;
        d.C = const_cast<float*>(dact_of(nb)); d.c16 = st16; d.c_rs = l.in_pad; d.c_cs = 1;
        if (nb > 0) { d.mask = act_of(nb); d.mask16 = st16; d.m_rs = l.in_pad; d.mask_scale = drop_gain; }
        if (st16) { d.B = wt_of(l); d.ldb = l.out_pad; d.Ct = tr.dact16t[nb]; d.ct_ld = Mpad; }
        if ((rc = st16 ? launch<true, false>(d, s) : launch<true, true>(d, s))) return rc;
    }
    for (int i = nb - 1; i >= 0; --i) {  // d act[i] = d act[i + 1] W_i (d act[i + 1] is already masked by layer i's ReLU)
        MlpLayer& l = c->blocks[i];
        G2 d{};
        d.bf16 = tr.bf16;
        d.M = (int)Mpad; d.N = l.in_pad; d.K = l.out_pad; d.kchunk = d.K;
        d.A = dact_of(i + 1); d.a16 = st16; d.lda = l.out_pad; d.B = w_of(l); d.b16 = st16; d.ldb = l.in_pad;
        d.C = const_cast<float*>(dact_of(i)); d.c16 = st16; d.c_rs = l.in_pad; d.c_cs = 1;
        if (i > 0) { d.mask = act_of(i); d.mask16 = st16; d.m_rs = l.in_pad; d.mask_scale = drop_gain; }  // act[0] (proj + emb) has no ReLU
        if (st16) { d.B = wt_of(l); d.ldb = l.out_pad; d.Ct = tr.dact16t[i]; d.ct_ld = Mpad; d.dma = l.in_pad % 64 == 0; }  // W^T has in_pad rows
        if ((rc = st16 ? launch<false, false>(d, s) : launch<false, true>(d, s))) return rc;
    }
    static const bool no_side = getenv("MTK_TRAIN_NO_SIDE") != nullptr;  // A/B: everything on the caller's stream
    hipStream_t ss = (tr.side && !no_side) ? tr.side : s;
    if (ss != s) {
        MTK_HIP(hipEventRecord(tr.ev_fork, s));
        MTK_HIP(hipStreamWaitEvent(ss, tr.ev_fork, 0));
    }
    {   // side stream: d emb table = scatter of d act[0] by timestep, then the time-embedding MLP over the T-row table
        means(ss);  // off the main stream's chain: a one-workgroup reduction nothing in the iteration waits for
        const float* dy = dact_of(0);  // [Mpad][proj.out_pad]
        const int64_t dy_ld = c->proj.out_pad;
        if (tr.deterministic) {
            const int64_t ne = (int64_t)tr.Tpad * c->emb_ld;
            if (st16) hipLaunchKernelGGL(k_scatter_emb_fixed<true>, dim3(1024), dim3(64, 4), 0, ss, (const void*)dy, dy_ld, t, B, c->dim_t, tr.demb_fix, c->emb_ld, c->T);
            else hipLaunchKernelGGL(k_scatter_emb_fixed<false>, dim3(1024), dim3(64, 4), 0, ss, (const void*)dy, dy_ld, t, B, c->dim_t, tr.demb_fix, c->emb_ld, c->T);
            hipLaunchKernelGGL(k_fixed_to_float, dim3(256), dim3(256), 0, ss, tr.demb_fix, tr.demb, ne);
        } else {
            if (st16) hipLaunchKernelGGL(k_scatter_emb<true>, dim3(1024), dim3(64, 4), 0, ss, (const void*)dy, dy_ld, t, B, c->dim_t, tr.demb, c->emb_ld, c->T);
            else hipLaunchKernelGGL(k_scatter_emb<false>, dim3(1024), dim3(64, 4), 0, ss, (const void*)dy, dy_ld, t, B, c->dim_t, tr.demb, c->emb_ld, c->T);
        }
        const int tp = c->emb_ld;
        if (tr.n_label > 0) {  // d label_embedding: d act[0] scattered by label (fixed point: order-independent in either reduction mode), times silu'
            if (!tr.yvec) MTK_FAIL(MTK_ERR_STATE, "mtk_train_backward: no labels (mtk_train_set_labels)");
            TrainLayerRef& rl = ref_of(&tr.label_l);
            if (st16) hipLaunchKernelGGL(k_scatter_emb_fixed<true>, dim3(1024), dim3(64, 4), 0, ss, (const void*)dy, dy_ld, tr.yvec, B, c->dim_t, tr.dlab_fix, tp, tr.n_label);
            else hipLaunchKernelGGL(k_scatter_emb_fixed<false>, dim3(1024), dim3(64, 4), 0, ss, (const void*)dy, dy_ld, tr.yvec, B, c->dim_t, tr.dlab_fix, tp, tr.n_label);
            hipLaunchKernelGGL(k_label_grad, dim3(64), dim3(256), 0, ss, tr.dlab_fix, tr.label_l.w32, grads + rl.w_off, (int64_t)tr.label_l.out_pad * tp);
        }
        MlpLayer &l2 = c->temb2, &l0 = c->temb0;
        TrainLayerRef &r2 = ref_of(&l2), &r0 = ref_of(&l0);
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = tp; g.N = tp; g.K = tr.Tpad; g.kchunk = kchunk_for(tr.Tpad, (tp / TM) * (tp / TN));
        g.A = tr.demb; g.lda = tp; g.B = tr.e1_tab; g.ldb = tp; g.c_rs = tp; g.c_cs = 1;
        // the bias gradients (column sums of demb / dtab) ride in the GEMMs as the row sums of their transposed A operand
        if ((rc = wgrad(g, [&](const G2& q) { return launch<true, true>(q, ss); }, grads + r2.w_off, grads + r2.b_off))) return rc;
        G2 d{};
        d.bf16 = tr.bf16;
        d.M = tr.Tpad; d.N = tp; d.K = tp; d.kchunk = tp;
        d.A = tr.demb; d.lda = tp; d.B = l2.w32; d.ldb = tp; d.C = tr.dtab; d.c_rs = tp; d.c_cs = 1;
        if ((rc = launch<false, true>(d, ss))) return rc;
        hipLaunchKernelGGL(k_silu_bwd, dim3(256), dim3(256), 0, ss, tr.z0_tab, tr.dtab, (int64_t)tr.Tpad * tp);
        g.A = tr.dtab; g.B = tr.sin_tab;
        g.slab = nullptr; g.rowsum_slab = nullptr;
        if ((rc = wgrad(g, [&](const G2& q) { return launch<true, true>(q, ss); }, grads + r0.w_off, grads + r0.b_off))) return rc;
        if (ss != s) MTK_HIP(hipEventRecord(tr.ev_join, ss));
    }
    // The time-embedding segments were entered first; the buckets below are cut from the segment list in completion order, so
    // the main stream's segments start after them and the time-embedding bucket (the last) takes the leading ones.
    const int temb_segs = segs.n;
    seg_mark = temb_segs;
    {   // head: dW[o][k] = sum_m dout[o][m] act[nb][m][k];  db = row sums of dout
        MlpLayer& l = c->head;
        TrainLayerRef& r = ref_of(&l);
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = l.out_pad; g.N = l.in_pad; g.K = (int)Mpad; wgrad_chunk(g);
        g.A = tr.dout; g.lda = ld; g.B = act_of(nb); g.b16 = st16; g.ldb = l.in_pad;
        // bf16 mode: both operands bf16 along the batch (the loss-gradient kernel's bf16 copy of dout): the LDS-DMA GEMM
        if (st16) { g.A = f16(tr.dout16); g.a16 = 1; g.B = f16(tr.act16t[nb]); g.ldb = Mpad; g.dma = l.in_pad % 64 == 0; }
        g.c_rs = l.in_pad; g.c_cs = 1;
        // db = row sums of dout [out_pad][Mpad] ride along (rows >= out are zero: the padding of the bias gradient is written too)
        if ((rc = wgrad(g, [&](const G2& q) { return st16 ? launch<false, false>(q, s) : launch<false, true>(q, s); }, grads + r.w_off, grads + r.b_off))) return rc;
        if ((rc = bucket_done())) return rc;
    }
    for (int i = nb - 1; i >= 0; --i) {  // dW_i[o][k] = sum_m d act[i + 1][m][o] act[i][m][k];  db[o] = sum_m d act[i + 1][m][o] rides along
        MlpLayer& l = c->blocks[i];
        TrainLayerRef& r = ref_of(&l);
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = l.out_pad; g.N = l.in_pad; g.K = (int)Mpad; wgrad_chunk(g);
        g.A = dact_of(i + 1); g.a16 = st16; g.lda = l.out_pad; g.B = act_of(i); g.b16 = st16; g.ldb = l.in_pad;
        if (st16) { g.A = f16(tr.dact16t[i + 1]); g.lda = Mpad; g.B = f16(tr.act16t[i]); g.ldb = Mpad; g.dma = 2; }  // act16t[i]: a multiple of 128 rows
        g.c_rs = l.in_pad; g.c_cs = 1;
        if ((rc = wgrad(g, [&](const G2& q) { return st16 ? launch<false, false>(q, s) : launch<true, true>(q, s); }, grads + r.w_off, grads + r.b_off))) return rc;
        if ((rc = bucket_done())) return rc;
    }
    {   // proj: dW[o][k] = sum_m d act[0][m][o] xin[k][m];  db
        MlpLayer& l = c->proj;
        TrainLayerRef& r = ref_of(&l);
        G2 g{};
        g.bf16 = tr.bf16;
        g.M = l.out_pad; g.N = l.in_pad; g.K = (int)Mpad; wgrad_chunk(g);
        g.A = dact_of(0); g.a16 = st16; g.lda = l.out_pad; g.B = tr.xin; g.ldb = ld;
        if (st16) { g.A = f16(tr.dact16t[0]); g.lda = Mpad; g.B = f16(tr.xin16); g.b16 = 1; g.dma = 1; }  // xin16: round_up(D, 128) rows
        g.c_rs = l.in_pad; g.c_cs = 1;
        if ((rc = wgrad(g, [&](const G2& q) { return st16 ? launch<false, false>(q, s) : launch<true, false>(q, s); }, grads + r.w_off, grads + r.b_off))) return rc;
        if ((rc = bucket_done())) return rc;
    }
    if (ss != s) MTK_HIP(hipStreamWaitEvent(s, tr.ev_join, 0));
    if (bucketed) {  // the last bucket: the time-embedding MLP's segments, entered first
        seg_mark = 0;
        segs.n = temb_segs;
        if ((rc = bucket_done())) return rc;
    } else if (segs.n > 0) {  // every split-batch partial of the iteration, added in slice order by one launch ...
        if (tr.defer_reduce) {  // ... or by the optimizer pass that follows (mtk_train_set_deferred_reduce)
            tr.pending = segs;
            tr.segs_pending = true;
        } else {
            hipLaunchKernelGGL(k_reduce_splits, dim3(256, (unsigned)segs.n), dim3(256), 0, s, segs);
        }
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// Gradient buckets of the image in the order the backward completes them (see train_phases): element offset and length
// of each inside one image.  Layers sit in the image as [W | b] in the order proj, temb0, temb2, blocks..., head.
extern "C" int mtk_train_bucket_layout(mtk_ctx* c, int max_buckets, int64_t* off, int64_t* len) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_bucket_layout: null context");
    if (!c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_bucket_layout: call mtk_train_attach first");
    TrainState& tr = c->train;
    const int lab = tr.n_label > 0 ? 1 : 0;
    const int nl = (int)tr.layers.size(), nb = nl - 1 - lab;  // temb0, temb2 (and the label embedding) share a bucket
    if (!off || !len) return nb;
    if (max_buckets < nb) MTK_FAIL(MTK_ERR_ARG, "mtk_train_bucket_layout: %d buckets, room for %d", nb, max_buckets);
    auto span = [&](int first, int last, int b) {
        off[b] = (int64_t)tr.layers[first].w_off;
        len[b] = (int64_t)(tr.layers[last].b_off + tr.layers[last].l->out_pad) - off[b];
    };
    int b = 0;
    for (int i = nl - 1; i >= 3 + lab; --i) span(i, i, b++);  // head, blocks last to first
    span(0, 0, b++);                                           // proj
    span(1, 2 + lab, b++);                                     // time-embedding MLP (+ label embedding)
    return nb;
}

extern "C" int mtk_train_set_bucket_events(mtk_ctx* c, int n, void* const* events) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_bucket_events: null context");
    if (!c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_set_bucket_events: call mtk_train_attach first");
    TrainState& tr = c->train;
    if (n == 0 || !events) {
        tr.bucket_events.clear();
        return MTK_OK;
    }
    const int want = (int)tr.layers.size() - 1 - (tr.n_label > 0 ? 1 : 0);
    if (n != want) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_bucket_events: %d events for %d buckets", n, want);
    tr.bucket_events.assign(n, nullptr);
    for (int i = 0; i < n; ++i) {
        if (!events[i]) {
            tr.bucket_events.clear();
            MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_bucket_events: event %d is null", i);
        }
        tr.bucket_events[i] = (hipEvent_t)events[i];
    }
    return MTK_OK;
}

// adds partials that mtk_train_step left for the optimizer pass (deferred reduction) when something else needs the gradient image first
static int materialize_grads(mtk_ctx* c, hipStream_t s) {
    TrainState& tr = c->train;
    if (!tr.segs_pending) return MTK_OK;
    hipLaunchKernelGGL(k_reduce_splits, dim3(256, (unsigned)tr.pending.n), dim3(256), 0, s, tr.pending);
    MTK_HIP(hipGetLastError());
    tr.segs_pending = false;
    return MTK_OK;
}

extern "C" int mtk_train_set_deferred_reduce(mtk_ctx* c, int on, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_deferred_reduce: null context");
    MTK_HIP(hipSetDevice(c->device));
    if (!on) {
        const int rc = materialize_grads(c, (hipStream_t)stream);
        if (rc) return rc;
    }
    c->train.defer_reduce = on != 0;
    return MTK_OK;
}

extern "C" int mtk_train_set_loss_means(mtk_ctx* c, float* means_dev) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_loss_means: null context");
    c->train.loss_means = means_dev;  // null: off
    return MTK_OK;
}

extern "C" int mtk_train_timesteps(mtk_ctx* c, int64_t B, int64_t* t_dev, mtk_stream stream) {
    if (!c || !t_dev) MTK_FAIL(MTK_ERR_ARG, "mtk_train_timesteps: null pointer");
    TrainState& tr = c->train;
    if (!tr.attached || !tr.t_drawn || !tr.tdraw) MTK_FAIL(MTK_ERR_STATE, "mtk_train_timesteps: the last forward phase on this context did not draw its timesteps (t = null)");
    if (B != tr.fwd_rows) MTK_FAIL(MTK_ERR_ARG, "mtk_train_timesteps: %lld rows, the last forward phase had %lld", (long long)B, (long long)tr.fwd_rows);
    MTK_HIP(hipSetDevice(c->device));
    MTK_HIP(hipMemcpyAsync(t_dev, tr.tdraw, sizeof(int64_t) * (size_t)B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MTK_OK;
}

extern "C" int mtk_train_set_precision(mtk_ctx* c, int precision) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_precision: null context");
    if (precision != MTK_PREC_FP32 && precision != MTK_PREC_BF16) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_train_set_precision: %d (fp32 or bf16)", precision);
    c->train.bf16 = precision == MTK_PREC_BF16 ? 1 : 0;
    return MTK_OK;
}

extern "C" int mtk_train_set_label_classes(mtk_ctx* c, int n_labels) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_label_classes: null context");
    if (c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_set_label_classes: call before mtk_train_attach (the image's layout depends on it)");
    if (n_labels < 0 || n_labels > (1 << 20)) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_label_classes: %d labels", n_labels);
    if (!c->have_weights) MTK_FAIL(MTK_ERR_STATE, "mtk_train_set_label_classes: call mtk_mlp_set_weights first (it fixes dim_t, and detaching clears the label layer)");
    TrainState& tr = c->train;
    tr.n_label = n_labels;
    tr.label_l = MlpLayer();
    if (n_labels > 0) {  // rows = labels, columns = dim_t: nn.Embedding's [num_classes, dim_t] weight as a layer's [out][in]
        tr.label_l.in = c->dim_t; tr.label_l.out = n_labels;
        tr.label_l.in_pad = c->emb_ld; tr.label_l.out_pad = round_up(n_labels, TN);
        tr.label_l.owned = false;
    }
    return MTK_OK;
}

extern "C" int mtk_train_set_labels(mtk_ctx* c, const int64_t* y_dev) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_labels: null context");
    c->train.yvec = y_dev;
    return MTK_OK;
}

extern "C" int mtk_train_set_dropout(mtk_ctx* c, double p, uint64_t seed) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_dropout: null context");
    if (!(p >= 0.0 && p < 1.0)) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_dropout: p = %g outside [0, 1)", p);
    c->train.drop_p = (float)p;
    c->train.drop_seed = seed;
    return MTK_OK;
}

__global__ void k_dropout_mask(uint8_t* out, int64_t rows, int width, uint64_t seed, uint64_t row0, int layer, uint32_t thresh) {
    const int64_t n = rows * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = g2_keep(seed, row0 + (uint64_t)(i / width), layer, (int)(i % width), thresh) ? 1 : 0;
}

extern "C" int mtk_train_dropout_mask(mtk_ctx* c, int layer, int64_t B, uint64_t row0, uint8_t* keep, mtk_stream stream) {
    if (!c || !keep) MTK_FAIL(MTK_ERR_ARG, "mtk_train_dropout_mask: null argument");
    if (layer < 0 || layer >= (int)c->blocks.size()) MTK_FAIL(MTK_ERR_ARG, "mtk_train_dropout_mask: hidden layer %d of %zu", layer, c->blocks.size());
    if (B <= 0) MTK_FAIL(MTK_ERR_ARG, "mtk_train_dropout_mask: B <= 0");
    MTK_HIP(hipSetDevice(c->device));
    const double th = (double)c->train.drop_p * 4294967296.0;
    const uint32_t thresh = c->train.drop_p > 0.0f ? (th >= 4294967295.0 ? 4294967295u : (uint32_t)th) : 0u;
    hipLaunchKernelGGL(k_dropout_mask, dim3(1024), dim3(256), 0, (hipStream_t)stream, keep, B, c->blocks[layer].out, c->train.drop_seed, row0, layer, thresh);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_train_set_deterministic(mtk_ctx* c, int on) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_set_deterministic: null context");
    c->train.deterministic = on != 0;
    return MTK_OK;
}

extern "C" int mtk_train_step(mtk_ctx* c, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                              const float* u, double inv_global_batch, float* loss_multi, float* loss_gauss, uint64_t seed,
                              uint64_t row0, mtk_stream stream) {
    return train_phases(c, B, x0, x0_ld, t, eps, u, inv_global_batch, inv_global_batch, loss_multi, loss_gauss, seed, row0, true, true, stream);
}

extern "C" int mtk_train_forward(mtk_ctx* c, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                                 const float* u, float* loss_multi, float* loss_gauss, uint64_t seed, uint64_t row0, mtk_stream stream) {
    return train_phases(c, B, x0, x0_ld, t, eps, u, 0.0, 0.0, loss_multi, loss_gauss, seed, row0, true, false, stream);
}

extern "C" int mtk_train_backward(mtk_ctx* c, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                                  double w_multi, double w_gauss, uint64_t seed, uint64_t row0, mtk_stream stream) {
    return train_phases(c, B, x0, x0_ld, t, eps, nullptr, w_multi, w_gauss, nullptr, nullptr, seed, row0, false, true, stream);
}

extern "C" int mtk_train_backward_dev(mtk_ctx* c, int64_t B, const float* x0, int64_t x0_ld, const int64_t* t, const float* eps,
                                      const float* w_dev, uint64_t seed, uint64_t row0, mtk_stream stream) {
    if (!w_dev) MTK_FAIL(MTK_ERR_ARG, "mtk_train_backward_dev: null weight pair");
    return train_phases(c, B, x0, x0_ld, t, eps, nullptr, 1.0, 1.0, nullptr, nullptr, seed, row0, false, true, stream, w_dev);
}

extern "C" int mtk_train_apply(mtk_ctx* c, double lr, double weight_decay, double beta1, double beta2, double eps,
                               int64_t step, double ema_rate, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_train_apply: null context");
    if (!c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_apply: call mtk_train_attach first");
    if (step < 1) MTK_FAIL(MTK_ERR_ARG, "mtk_train_apply: step is 1-based");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    TrainState& tr = c->train;
    const size_t n = tr.n_img;
    AdamArgs a;
    a.p16 = tr.w16;
    a.p = tr.state; a.g = tr.state + n; a.m = tr.state + 2 * n; a.v = tr.state + 3 * n; a.ema = tr.state + 4 * n;
    a.decay = (float)(1.0 - lr * weight_decay);
    a.beta1_c = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.beta2_c = (float)(1.0 - beta2);
    a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    a.eps = (float)eps;
    a.ema_rate = (float)ema_rate;
    a.ema_c = (float)(1.0 - ema_rate);
    // The image is [proj | temb0 | temb2 | hidden layers | head] (layer_list).  The time-embedding MLP's two small layers are updated
    // first, on the side stream, followed there by the rebuild of the [T, dim_t] table from them (three small launches); the
    // pass over everything else and the bf16 transposes run beside them on the caller's stream.
    static const bool no_side = getenv("MTK_TRAIN_NO_SIDE") != nullptr;
    hipStream_t ss = (tr.side && !no_side) ? tr.side : s;
    const int64_t t_beg = (int64_t)tr.layers[1].w_off, t_end = (int64_t)tr.layers[3].w_off;
    if (ss != s) {
        MTK_HIP(hipEventRecord(tr.ev_fork, s));
        MTK_HIP(hipStreamWaitEvent(ss, tr.ev_fork, 0));
    }
    int rc;
    if (tr.segs_pending) {
        // deferred reduction: the optimizer pass runs segment by segment and adds each segment's split-batch partials as it reads the
        // gradient.  Layers without partials (the label embedding: its gradient is written directly) enter with splits = 0.
        float* gw = tr.state + n;
        MtkSegs side{}, mainl{};
        auto add = [&](float* dst, const float* slab, int splits, int64_t n4) -> int {
            const int64_t off = dst - gw;
            MtkSegs& q = (off >= t_beg && off < t_end) ? side : mainl;
            if (q.n >= MTK_MAX_SEGS) MTK_FAIL(MTK_ERR_STATE, "mtk_train_apply: more than %d gradient segments", MTK_MAX_SEGS);
            q.dst[q.n] = dst; q.slab[q.n] = slab; q.splits[q.n] = splits; q.n4[q.n] = n4;
            ++q.n;
            return MTK_OK;
        };
        for (int i = 0; i < tr.pending.n; ++i)
            if ((rc = add(tr.pending.dst[i], tr.pending.slab[i], tr.pending.splits[i], tr.pending.n4[i]))) return rc;
        for (auto& lr : tr.layers) {
            bool have_w = false, have_b = false;
            for (int i = 0; i < tr.pending.n; ++i) {
                have_w = have_w || tr.pending.dst[i] == gw + lr.w_off;
                have_b = have_b || tr.pending.dst[i] == gw + lr.b_off;
            }
            if (!have_w && (rc = add(gw + lr.w_off, nullptr, 0, (int64_t)lr.l->out_pad * lr.l->in_pad / 4))) return rc;
            if (!have_b && (rc = add(gw + lr.b_off, nullptr, 0, (int64_t)lr.l->out_pad / 4))) return rc;
        }
        tr.segs_pending = false;
        if (side.n > 0) hipLaunchKernelGGL(k_adamw_ema_segs, dim3(16, (unsigned)side.n), dim3(256), 0, ss, a, gw, side);
        if ((rc = refresh_emb(c, ss))) return rc;
        if (ss != s) MTK_HIP(hipEventRecord(tr.ev_join, ss));
        if (mainl.n > 0) hipLaunchKernelGGL(k_adamw_ema_segs, dim3(128, (unsigned)mainl.n), dim3(256), 0, s, a, gw, mainl);
    } else {
        a.beg0 = t_beg; a.len0 = t_end - t_beg; a.beg1 = 0; a.n = a.len0;
        hipLaunchKernelGGL(k_adamw_ema, dim3(64), dim3(256), 0, ss, a);
        if ((rc = refresh_emb(c, ss))) return rc;
        if (ss != s) MTK_HIP(hipEventRecord(tr.ev_join, ss));
        a.beg0 = 0; a.len0 = t_beg; a.beg1 = t_end; a.n = (int64_t)n - (t_end - t_beg);
        hipLaunchKernelGGL(k_adamw_ema, dim3(1024), dim3(256), 0, s, a);
    }
    MTK_HIP(hipGetLastError());
    if ((rc = refresh_w16t(c, s))) return rc;
    if (ss != s) MTK_HIP(hipStreamWaitEvent(s, tr.ev_join, 0));
    return MTK_OK;
}

extern "C" int mtk_train_export(mtk_ctx* c, int image, float* const* W, float* const* b, mtk_stream stream) {
    if (!c || !W || !b) MTK_FAIL(MTK_ERR_ARG, "mtk_train_export: null argument");
    if (!c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_export: no training image attached");
    if (image < 0 || image > 4) MTK_FAIL(MTK_ERR_ARG, "mtk_train_export: image %d outside [0, 4]", image);
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    TrainState& tr = c->train;
    if (image == 1) {
        const int rc = materialize_grads(c, s);
        if (rc) return rc;
    }
    const float* img = tr.state + (size_t)image * tr.n_img;
    for (size_t i = 0; i < tr.layers.size(); ++i) {
        const MlpLayer& l = *tr.layers[i].l;
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_train_export: W[%zu] or b[%zu] is null", i, i);
        hipLaunchKernelGGL(k_unpad, dim3(256), dim3(256), 0, s, img + tr.layers[i].w_off, l.in_pad, W[i], l.out, l.in);
        hipLaunchKernelGGL(k_unpad, dim3(4), dim3(256), 0, s, img + tr.layers[i].b_off, l.out_pad, b[i], 1, l.out);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

extern "C" int mtk_train_import(mtk_ctx* c, int image, const float* const* W, const float* const* b, mtk_stream stream) {
    if (!c || !W || !b) MTK_FAIL(MTK_ERR_ARG, "mtk_train_import: null argument");
    if (!c->train.attached) MTK_FAIL(MTK_ERR_STATE, "mtk_train_import: no training image attached");
    if (image < 0 || image > 4) MTK_FAIL(MTK_ERR_ARG, "mtk_train_import: image %d outside [0, 4]", image);
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    TrainState& tr = c->train;
    float* img = tr.state + (size_t)image * tr.n_img;
    if (image == 1) tr.segs_pending = false;  // imported gradients replace the iteration's: partials left for the optimizer pass are dropped
    for (size_t i = 0; i < tr.layers.size(); ++i) {
        const MlpLayer& l = *tr.layers[i].l;
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_train_import: W[%zu] or b[%zu] is null", i, i);
        hipLaunchKernelGGL(k_pad, dim3(256), dim3(256), 0, s, img + tr.layers[i].w_off, l.in_pad, W[i], l.out, l.in);
        hipLaunchKernelGGL(k_pad, dim3(4), dim3(256), 0, s, img + tr.layers[i].b_off, l.out_pad, b[i], 1, l.out);
    }
    MTK_HIP(hipGetLastError());
    if (image == 0) {
        const int rc = refresh_w16(c, s);
        if (rc) return rc;
        return refresh_emb(c, s);
    }
    return MTK_OK;
}

// =================================================================================================================
// Classifier guidance: the reference's `Classifier` (models/clavaddpm/model.py:45-120) in
// eval mode and the input gradient its conditioning function takes through autograd
// (models/clavaddpm/synthesizer.py:429-451): grad = scale * d log_softmax(classifier(x, t))[y] / d x.
//
// Network as the library sees it: h0 = proj(x) + emb[t] (emb = Linear(SiLU(Linear(sinusoid(t)))), a [T, dim_t] table
// built here), then n_lin Linear layers, each but the last followed by ReLU (slope 0) or LeakyReLU (slope > 0).  The
// reference's eval-mode BatchNorm1d after each activation is an affine map per feature; the host folds it into the
// NEXT Linear (W' = W diag(s), b' = b + W o) before handing the weights over, and Dropout is the identity in eval mode.
// Forward and the hand-written backward (softmax residual -> dgrad chain with activation masks -> proj^T) run on the
// same fp32 MFMA GEMM as the training iteration above; nothing goes through autograd.
// =================================================================================================================
struct mtk_clf {
    int device = 0, T = 0, D_in = 0, n_out = 0, dim_t = 0, n_lin = 0;
    std::vector<float> slopes;      // activation after Linear i, i < n_lin - 1
    MlpLayer proj;
    std::vector<MlpLayer> lin;
    float* emb = nullptr;           // [Tpad][emb_ld]
    int emb_ld = 0;
    // workspaces for `cap` rows (a multiple of 128)
    int64_t cap = 0;
    float* xin = nullptr;           // [cap][proj.in_pad]
    std::vector<float*> act;        // act[0] = h0 [cap][proj.out_pad]; act[i + 1] = output of Linear i [cap][lin[i].out_pad]
    float* d[2] = {nullptr, nullptr};  // gradient ping-pong [cap][widest padded layer]
    int x3 = 0;                     // mtk_clf_set_precision(MTK_PREC_BF16X3): the GEMMs as split-bf16 products (k_gemm2x)
};

namespace {

__global__ void k_clf_pad_rows(float* dst, int ld, const float* src, int64_t src_ld, int64_t rows, int64_t rows_pad, int cols) {
    const int64_t n = rows_pad * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = (r < rows && c < cols) ? src[r * src_ld + c] : 0.0f;
    }
}
// d log_softmax(logits)[y] / d logits = onehot(y) - softmax(logits), times `scale`; padding rows / columns zero.
// One lane per row (n_out is the number of clusters: tens).
// One wave per row, lane j -> column j (j + 64, ...): every lane runs the row's max / sum loops itself, in column order (the loads are
// wave-wide broadcasts), so the values are those of a sequential pass over the row; the stores are one coalesced line per row.
// (One thread per row took 23 us whatever the batch: a chain of strided loads and 64 serial stores per thread.)
__global__ __launch_bounds__(256) void k_clf_dlogits(const float* logits, int ld, int64_t rows, int64_t rows_pad, int n_out, const int64_t* y,
                                                     float scale, float* dl) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows_pad; r += (int64_t)gridDim.x * 4) {
        float* o = dl + r * ld;
        if (r >= rows) {
            for (int j = lane; j < ld; j += 64) o[j] = 0.0f;
            continue;
        }
        const float* z = logits + r * ld;
        float mx = -INFINITY;
        for (int j = 0; j < n_out; ++j) mx = fmaxf(mx, z[j]);
        float sum = 0.0f;
        for (int j = 0; j < n_out; ++j) sum += expf(z[j] - mx);
        const float lse = mx + logf(sum);
        const int64_t yy = y[r];
        for (int j = lane; j < ld; j += 64) o[j] = j < n_out ? scale * ((j == yy ? 1.0f : 0.0f) - expf(z[j] - lse)) : 0.0f;
    }
}
// t != null: rows whose timestep lies outside [0, T) leave as NaN (what mtk_poison_bad_t writes, without its launch)
__global__ void k_clf_copy_out(float* dst, int64_t dst_ld, const float* src, int ld, int64_t rows, int cols, const int64_t* t = nullptr, int T = 0) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        const bool bad = t && (t[r] < 0 || t[r] >= T);
        dst[r * dst_ld + c] = bad ? __builtin_nanf("") : src[r * ld + c];
    }
}

void clf_free_layer(MlpLayer& l) {
    if (l.w32) (void)hipFree(l.w32);
    if (l.b32) (void)hipFree(l.b32);
    l = MlpLayer();
}
int clf_make_layer(MlpLayer& l, int in, int out, int in_pad, const float* W, const float* b, hipStream_t s) {
    l.in = in; l.out = out; l.in_pad = in_pad; l.out_pad = round_up(out, TN);
    MTK_HIP(hipMalloc(&l.w32, sizeof(float) * (size_t)l.out_pad * l.in_pad));
    MTK_HIP(hipMalloc(&l.b32, sizeof(float) * (size_t)l.out_pad));
    MTK_HIP(hipMemsetAsync(l.w32, 0, sizeof(float) * (size_t)l.out_pad * l.in_pad, s));
    MTK_HIP(hipMemsetAsync(l.b32, 0, sizeof(float) * (size_t)l.out_pad, s));
    hipLaunchKernelGGL(k_pad, dim3(256), dim3(256), 0, s, l.w32, l.in_pad, W, out, in);
    hipLaunchKernelGGL(k_pad, dim3(4), dim3(256), 0, s, l.b32, l.out_pad, b, 1, out);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}
void clf_free_ws(mtk_clf* c) {
    auto fr = [](float*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(c->xin); fr(c->d[0]); fr(c->d[1]);
    for (auto& p : c->act) fr(p);
    c->act.clear();
    c->cap = 0;
}
int clf_ensure_ws(mtk_clf* c, int64_t Mpad, hipStream_t s) {
    if (Mpad <= c->cap) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    clf_free_ws(c);
    int maxpad = c->proj.out_pad > c->proj.in_pad ? c->proj.out_pad : c->proj.in_pad;
    for (auto& l : c->lin) maxpad = l.out_pad > maxpad ? l.out_pad : maxpad;
    MTK_HIP(hipMalloc(&c->xin, sizeof(float) * (size_t)Mpad * c->proj.in_pad));
    for (int i = 0; i < 2; ++i) MTK_HIP(hipMalloc(&c->d[i], sizeof(float) * (size_t)Mpad * maxpad));
    c->act.assign(1 + c->lin.size(), nullptr);
    MTK_HIP(hipMalloc(&c->act[0], sizeof(float) * (size_t)Mpad * c->proj.out_pad));
    for (size_t i = 0; i < c->lin.size(); ++i) MTK_HIP(hipMalloc(&c->act[1 + i], sizeof(float) * (size_t)Mpad * c->lin[i].out_pad));
    c->cap = Mpad;
    return MTK_OK;
}

}  // namespace

extern "C" void mtk_clf_destroy(mtk_clf* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    clf_free_layer(c->proj);
    for (auto& l : c->lin) clf_free_layer(l);
    if (c->emb) (void)hipFree(c->emb);
    clf_free_ws(c);
    delete c;
}

extern "C" int mtk_clf_create(int device, int T, int D_in, int dim_t, int n_lin, const int32_t* widths_host, const float* slopes_host,
                              const float* const* W, const float* const* b, const float* freqs_host, mtk_stream stream, mtk_clf** out) {
    if (!out) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_create: out is null");
    *out = nullptr;
    if (T < 1 || D_in < 1 || dim_t < 2 || dim_t > 1024 || n_lin < 1 || n_lin > 32 || !widths_host || !W || !b || !freqs_host || (n_lin > 1 && !slopes_host))
        MTK_FAIL(MTK_ERR_ARG, "mtk_clf_create: bad argument");
    for (int i = 0; i < n_lin + 3; ++i)
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_create: W[%d] or b[%d] is null", i, i);
    for (int i = 0; i < n_lin; ++i)
        if (widths_host[i] < 1 || widths_host[i] > 8192) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_clf_create: layer width %d", widths_host[i]);
    int n_dev = 0;
    MTK_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_create: device %d of %d", device, n_dev);
    MTK_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    mtk_clf* c = new mtk_clf();
    c->device = device; c->T = T; c->D_in = D_in; c->dim_t = dim_t; c->n_lin = n_lin; c->n_out = widths_host[n_lin - 1];
    for (int i = 0; i + 1 < n_lin; ++i) c->slopes.push_back(slopes_host[i]);
    const int tp = round_up(dim_t, TN);
    c->emb_ld = tp;
    int rc = MTK_OK;
    MlpLayer t0, t2;  // the time-embedding MLP, needed only to build the table
    float *d_freq = nullptr, *sin_tab = nullptr, *z0 = nullptr, *e1 = nullptr;
    auto fail = [&](int code) {
        clf_free_layer(t0); clf_free_layer(t2);
        if (d_freq) (void)hipFree(d_freq);
        if (sin_tab) (void)hipFree(sin_tab);
        if (z0) (void)hipFree(z0);
        if (e1) (void)hipFree(e1);
        mtk_clf_destroy(c);
        return code;
    };
    // weight order: 0 proj, 1 timestep_embedding.0, 2 timestep_embedding.2, 3.. the n_lin Linear layers (BatchNorm folded in)
    if ((rc = clf_make_layer(c->proj, D_in, dim_t, round_up(D_in, TN), W[0], b[0], s))) return fail(rc);  // in_pad = 128 k: the transposed-operand tiles of the final dgrad read whole 64/128-wide rows
    if ((rc = clf_make_layer(t0, dim_t, dim_t, tp, W[1], b[1], s))) return fail(rc);
    if ((rc = clf_make_layer(t2, dim_t, dim_t, tp, W[2], b[2], s))) return fail(rc);
    c->lin.resize(n_lin);
    int prev = dim_t, prev_pad = tp;
    for (int i = 0; i < n_lin; ++i) {
        if ((rc = clf_make_layer(c->lin[i], prev, widths_host[i], prev_pad, W[3 + i], b[3 + i], s))) return fail(rc);
        prev = widths_host[i];
        prev_pad = c->lin[i].out_pad;
    }
    const int Tpad = round_up(T, TM);
    const size_t tb = sizeof(float) * (size_t)Tpad * tp;
    hipError_t e = hipMalloc(&d_freq, sizeof(float) * (dim_t / 2));
    if (e == hipSuccess) e = hipMalloc(&sin_tab, tb);
    if (e == hipSuccess) e = hipMalloc(&z0, tb);
    if (e == hipSuccess) e = hipMalloc(&e1, tb);
    if (e == hipSuccess) e = hipMalloc(&c->emb, tb);
    if (e == hipSuccess) e = hipMemcpyAsync(d_freq, freqs_host, sizeof(float) * (dim_t / 2), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        mtk_set_error("mtk_clf_create: %s", hipGetErrorString(e));
        return fail(MTK_ERR_HIP);
    }
    hipLaunchKernelGGL(k_sin_tab, dim3(256), dim3(256), 0, s, sin_tab, Tpad, tp, T, dim_t, d_freq);
    G2 g{};
    g.M = Tpad; g.N = tp; g.K = tp; g.kchunk = tp;
    g.A = sin_tab; g.lda = tp; g.B = t0.w32; g.ldb = tp; g.C = z0; g.c_rs = tp; g.c_cs = 1; g.bias = t0.b32;
    if ((rc = launch<false, false>(g, s))) return fail(rc);
    hipLaunchKernelGGL(k_silu, dim3(256), dim3(256), 0, s, z0, e1, (int64_t)Tpad * tp);
    g.A = e1; g.B = t2.w32; g.C = c->emb; g.bias = t2.b32;
    if ((rc = launch<false, false>(g, s))) return fail(rc);
    if (hipStreamSynchronize(s) != hipSuccess) {
        mtk_set_error("mtk_clf_create: building the time-embedding table failed");
        return fail(MTK_ERR_HIP);
    }
    clf_free_layer(t0); clf_free_layer(t2);
    (void)hipFree(d_freq); (void)hipFree(sin_tab); (void)hipFree(z0); (void)hipFree(e1);
    *out = c;
    return MTK_OK;
}

extern "C" int mtk_clf_set_precision(mtk_clf* c, int precision) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_set_precision: null classifier");
    if (precision != MTK_PREC_FP32 && precision != MTK_PREC_BF16X3)
        MTK_FAIL(MTK_ERR_ARG, "mtk_clf_set_precision: precision %d (MTK_PREC_FP32 or MTK_PREC_BF16X3; plain bf16 operands do not carry the input gradient)", precision);
    c->x3 = precision == MTK_PREC_BF16X3 ? 1 : 0;
    return MTK_OK;
}

// logits (optional, [B, n_out] row-major) and / or grad (optional, [B, D_in] row-major fp32):
// grad = scale * d log_softmax(logits)[y] / d x.  x: [B, x_ld] row-major fp32; t: [B] int64 (clamped into [0, T) for
// addressing; rows with a timestep outside it come back NaN); y: [B] int64 class indices (required with grad).
extern "C" int mtk_clf_guidance(mtk_clf* c, int64_t B, const float* x, int64_t x_ld, const int64_t* t, const int64_t* y, double scale,
                                float* logits, float* grad, mtk_stream stream) {
    if (!c) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_guidance: null classifier");
    if (B == 0) return MTK_OK;  // empty batch: nothing to read (a framework's empty tensor has a null data pointer)
    if (B < 0 || !x || !t || (grad && !y)) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_guidance: null pointer or negative B");
    if (x_ld < c->D_in) MTK_FAIL(MTK_ERR_ARG, "mtk_clf_guidance: x_ld %lld < input width %d", (long long)x_ld, c->D_in);
    if (B == 0 || (!logits && !grad)) return MTK_OK;
    if (B > (1 << 22)) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_clf_guidance: B = %lld above 4194304 rows per call", (long long)B);
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    const int64_t Mpad = (B + TM - 1) / TM * TM;
    int rc;
    if ((rc = clf_ensure_ws(c, Mpad, s))) return rc;
    hipLaunchKernelGGL(k_clf_pad_rows, dim3(1024), dim3(256), 0, s, c->xin, c->proj.in_pad, x, x_ld, B, Mpad, c->D_in);
    {   // h0 = proj(x) + emb[t]   (MODEL:118-119)
        G2 g{};
        g.x3 = c->x3;
        g.M = (int)Mpad; g.N = c->proj.out_pad; g.K = c->proj.in_pad; g.kchunk = g.K;
        g.A = c->xin; g.lda = c->proj.in_pad; g.B = c->proj.w32; g.ldb = c->proj.in_pad;
        g.C = c->act[0]; g.c_rs = c->proj.out_pad; g.c_cs = 1; g.bias = c->proj.b32;
        g.emb = c->emb; g.emb_ld = c->emb_ld; g.tvec = t; g.rows_valid = B; g.T = c->T;
        if ((rc = launch<false, false>(g, s))) return rc;
    }
    for (int i = 0; i < c->n_lin; ++i) {   // MODEL:95-117 in eval mode
        MlpLayer& l = c->lin[i];
        G2 g{};
        g.x3 = c->x3;
        g.M = (int)Mpad; g.N = l.out_pad; g.K = l.in_pad; g.kchunk = g.K;
        g.A = c->act[i]; g.lda = l.in_pad; g.B = l.w32; g.ldb = l.in_pad;
        g.C = c->act[i + 1]; g.c_rs = l.out_pad; g.c_cs = 1; g.bias = l.b32;
        if (i + 1 < c->n_lin) { g.relu = 1; g.neg_slope = c->slopes[i]; }
        if ((rc = launch<false, false>(g, s))) return rc;
    }
    const MlpLayer& last = c->lin.back();
    if (logits) {
        hipLaunchKernelGGL(k_clf_copy_out, dim3(1024), dim3(256), 0, s, logits, (int64_t)c->n_out, c->act.back(), last.out_pad, B, c->n_out, t, c->T);
    }
    if (!grad) {
        MTK_HIP(hipGetLastError());
        return MTK_OK;
    }
    // SYNTH:446-449: selected = log_softmax(logits)[range, y]; grad of selected.sum() w.r.t. the input, times the scale
    hipLaunchKernelGGL(k_clf_dlogits, dim3((unsigned)(Mpad / 4 < 4096 ? Mpad / 4 : 4096)), dim3(256), 0, s, c->act.back(), last.out_pad, B, Mpad, c->n_out, y,
                       (float)scale, c->d[0]);
    int cur = 0;
    for (int i = c->n_lin - 1; i >= 0; --i) {
        MlpLayer& l = c->lin[i];
        G2 d{};
        d.x3 = c->x3;
        d.M = (int)Mpad; d.N = l.in_pad; d.K = l.out_pad; d.kchunk = d.K;
        d.A = c->d[cur]; d.lda = l.out_pad; d.B = l.w32; d.ldb = l.in_pad;
        d.C = c->d[cur ^ 1]; d.c_rs = l.in_pad; d.c_cs = 1;
        if (i > 0) { d.mask = c->act[i]; d.m_rs = l.in_pad; d.neg_slope = c->slopes[i - 1]; }  // act[0] = h0 has no activation
        if ((rc = launch<false, true>(d, s))) return rc;
        cur ^= 1;
    }
    {   // d x = d h0 . W_proj
        G2 d{};
        d.x3 = c->x3;
        d.M = (int)Mpad; d.N = c->proj.in_pad; d.K = c->proj.out_pad; d.kchunk = d.K;
        d.A = c->d[cur]; d.lda = c->proj.out_pad; d.B = c->proj.w32; d.ldb = c->proj.in_pad;
        d.C = c->d[cur ^ 1]; d.c_rs = c->proj.in_pad; d.c_cs = 1;
        if ((rc = launch<false, true>(d, s))) return rc;
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_clf_copy_out, dim3(1024), dim3(256), 0, s, grad, (int64_t)c->D_in, c->d[cur], c->proj.in_pad, B, c->D_in, t, c->T);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}


// =================================================================================================================
// TabSyn's EDM denoiser (models/tabsyn/model/modules.py:23-185): MLPDiffusion = proj(x) +
// time_embed(swap(PositionalEmbedding(noise_labels))) followed by Linear/SiLU x 3 and a Linear head; Preconditioner
// (c_skip, c_out, c_in, c_noise of Karras et al.) around it; EDMLoss on injected draws.  The noise label is a continuous
// log-sigma, so the time embedding is computed per record (no table).  Forward only, on the fp32 MFMA GEMM above; the
// element-wise preconditioning / loss is one fused kernel either side of the network.
// =================================================================================================================
struct mtk_edm {
    int device = 0, d_in = 0, dim_t = 0;
    MlpLayer proj, te0, te2, m[4];
    float* freqs = nullptr;  // [dim_t / 2]
    int64_t cap = 0;
    float *xin = nullptr, *pe = nullptr, *e1 = nullptr, *emb = nullptr, *h = nullptr, *a0 = nullptr, *a1 = nullptr, *a2 = nullptr, *f = nullptr;
    float *sig = nullptr;    // [cap] sigma per record (EDMLoss draws it)
    // backward of EDMLoss (mtk_edm_loss with keep = 1, then mtk_edm_backward): pre-activations of the four SiLUs, d loss / d F per
    // unit upstream gradient, gradient ping-pong, padded weight / bias gradients per layer
    int64_t cap_bwd = 0, kept_rows = 0;
    float *zt = nullptr, *z0 = nullptr, *z1 = nullptr, *z2 = nullptr, *dfac = nullptr, *dF = nullptr, *d0 = nullptr, *d1 = nullptr;
    float *gw[7] = {}, *gb[7] = {};
};

namespace {

// rows of the network input and of the positional embedding (modules.py:46-50, 90-91, 133-141).
// mode 0: x_in = x (raw MLPDiffusion.forward, label = lab[row]);  mode 1: x_in = c_in * x, label = log(sigma) / 4 (Preconditioner);
// mode 2: sigma = exp(rnd * p_std + p_mean), x_in = c_in * (y + noise * sigma) (EDMLoss then Preconditioner).
struct EdmPre {
    int mode, d_in, dim_t, in_pad, tp;
    int64_t B, Mpad;
    const float* x; int64_t x_ld;
    const float* lab; int64_t lab_stride;   // noise labels (mode 0) / sigma (mode 1) / rnd_normal (mode 2); stride 0 broadcasts one value
    const float* noise;                     // mode 2: [B][d_in]
    float p_mean, p_std, sigma_data;
    const float* freqs;
    float *xin, *pe, *sig;
};
__global__ __launch_bounds__(256) void k_edm_pre(EdmPre a) {
    const int half = a.dim_t / 2;
    for (int64_t r = blockIdx.x; r < a.Mpad; r += gridDim.x) {
        float sigma = 1.0f, label = 0.0f, c_in = 1.0f;
        if (r < a.B) {
            const float v = a.lab[r * a.lab_stride];
            if (a.mode == 0) label = v;
            else {
                sigma = a.mode == 2 ? expf(v * a.p_std + a.p_mean) : v;
                c_in = 1.0f / sqrtf(a.sigma_data * a.sigma_data + sigma * sigma);
                label = logf(sigma) / 4.0f;
            }
            if (threadIdx.x == 0 && a.sig) a.sig[r] = sigma;
        }
        for (int j = threadIdx.x; j < a.in_pad; j += 256) {
            float v = 0.0f;
            if (r < a.B && j < a.d_in) {
                v = a.x[r * a.x_ld + j];
                if (a.mode == 2) v = v + a.noise[r * a.d_in + j] * sigma;
                if (a.mode != 0) v = c_in * v;
            }
            a.xin[r * a.in_pad + j] = v;
        }
        for (int j = threadIdx.x; j < a.tp; j += 256) {  // [cos | sin] with the halves swapped = [sin | cos]
            float v = 0.0f;
            if (r < a.B && j < a.dim_t) {
                const float arg = label * a.freqs[j < half ? j : j - half];
                v = j < half ? sinf(arg) : cosf(arg);
            }
            a.pe[r * a.tp + j] = v;
        }
    }
}
// D = c_skip * x + c_out * F (modules.py:135-144); mode 2 also weight * (D - y)^2 (modules.py:178-185), x being y + n again.
struct EdmPost {
    int mode, d_in, f_ld;
    int64_t B;
    const float* x; int64_t x_ld;
    const float* noise;
    const float* sig; int64_t sig_stride;
    float sigma_data;
    const float* f;
    float* out;
    float* dfac;  // optional (mode 2): d loss[r][j] / d F[r][j] = 2 w (D - y) c_out, row-major with f's leading dimension
};
__global__ __launch_bounds__(256) void k_edm_post(EdmPost a) {
    const int64_t n = a.B * a.d_in;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.d_in;
        const int j = (int)(i - r * a.d_in);
        const float fx = a.f[r * a.f_ld + j];
        if (a.mode == 0) { a.out[i] = fx; continue; }
        const float sigma = a.sig[r * a.sig_stride], sd = a.sigma_data;
        const float y = a.x[r * a.x_ld + j];
        const float x = a.mode == 2 ? y + a.noise[i] * sigma : y;
        const float c_skip = (sd * sd) / (sigma * sigma + sd * sd);
        const float c_out = sigma * sd / sqrtf(sigma * sigma + sd * sd);
        const float d = c_skip * x + c_out * fx;
        if (a.mode == 1) { a.out[i] = d; continue; }
        const float w = (sigma * sigma + sd * sd) / ((sigma * sd) * (sigma * sd));
        a.out[i] = w * ((d - y) * (d - y));
        if (a.dfac) a.dfac[r * a.f_ld + j] = 2.0f * w * (d - y) * c_out;
    }
}

void edm_free_ws(mtk_edm* e) {
    auto fr = [](float*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(e->xin); fr(e->pe); fr(e->e1); fr(e->emb); fr(e->h); fr(e->a0); fr(e->a1); fr(e->a2); fr(e->f); fr(e->sig);
    fr(e->zt); fr(e->z0); fr(e->z1); fr(e->z2); fr(e->dfac); fr(e->dF); fr(e->d0); fr(e->d1);
    e->cap = 0;
    e->cap_bwd = 0;
    e->kept_rows = 0;
}
int edm_ensure_bwd(mtk_edm* e, int64_t Mpad, hipStream_t s) {  // after edm_ensure_ws: cap >= Mpad
    if (Mpad <= e->cap_bwd) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    auto fr = [](float*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(e->zt); fr(e->z0); fr(e->z1); fr(e->z2); fr(e->dfac); fr(e->dF); fr(e->d0); fr(e->d1);
    const int64_t rows = e->cap;  // same row capacity as the forward buffers
    auto al = [&](float*& p, int ld) { return hipMalloc(&p, sizeof(float) * (size_t)rows * ld); };
    int wide = e->proj.in_pad;
    for (const MlpLayer* l : {&e->proj, &e->te0, &e->te2, &e->m[0], &e->m[1], &e->m[2], &e->m[3]}) {
        wide = l->in_pad > wide ? l->in_pad : wide;
        wide = l->out_pad > wide ? l->out_pad : wide;
    }
    MTK_HIP(al(e->zt, e->te0.out_pad)); MTK_HIP(al(e->z0, e->m[0].out_pad)); MTK_HIP(al(e->z1, e->m[1].out_pad)); MTK_HIP(al(e->z2, e->m[2].out_pad));
    MTK_HIP(al(e->dfac, e->m[3].out_pad)); MTK_HIP(al(e->dF, e->m[3].out_pad)); MTK_HIP(al(e->d0, wide)); MTK_HIP(al(e->d1, wide));
    MTK_HIP(hipMemsetAsync(e->dfac, 0, sizeof(float) * (size_t)rows * e->m[3].out_pad, s));  // padding of dfac stays zero: k_edm_post writes rows < B, columns < d_in
    const MlpLayer* ls[7] = {&e->proj, &e->te0, &e->te2, &e->m[0], &e->m[1], &e->m[2], &e->m[3]};
    for (int i = 0; i < 7; ++i) {
        if (!e->gw[i]) MTK_HIP(hipMalloc(&e->gw[i], sizeof(float) * (size_t)ls[i]->out_pad * ls[i]->in_pad));
        if (!e->gb[i]) MTK_HIP(hipMalloc(&e->gb[i], sizeof(float) * (size_t)ls[i]->out_pad));
    }
    e->cap_bwd = rows;
    return MTK_OK;
}
int edm_ensure_ws(mtk_edm* e, int64_t Mpad, hipStream_t s) {
    if (Mpad <= e->cap) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    edm_free_ws(e);
    auto al = [&](float*& p, int ld) { return hipMalloc(&p, sizeof(float) * (size_t)Mpad * ld); };
    MTK_HIP(al(e->xin, e->proj.in_pad)); MTK_HIP(al(e->pe, e->te0.in_pad)); MTK_HIP(al(e->e1, e->te0.out_pad));
    MTK_HIP(al(e->emb, e->te2.out_pad)); MTK_HIP(al(e->h, e->proj.out_pad)); MTK_HIP(al(e->a0, e->m[0].out_pad));
    MTK_HIP(al(e->a1, e->m[1].out_pad)); MTK_HIP(al(e->a2, e->m[2].out_pad)); MTK_HIP(al(e->f, e->m[3].out_pad)); MTK_HIP(al(e->sig, 1));
    e->cap = Mpad;
    return MTK_OK;
}
int edm_linear(const MlpLayer& l, int64_t Mpad, const float* in, float* out, int act, const float* addend, int64_t add_rs, hipStream_t s,
               float* pre = nullptr) {
    G2 g{};
    g.M = (int)Mpad; g.N = l.out_pad; g.K = l.in_pad; g.kchunk = g.K;
    g.A = in; g.lda = l.in_pad; g.B = l.w32; g.ldb = l.in_pad; g.C = out; g.c_rs = l.out_pad; g.c_cs = 1; g.bias = l.b32;
    g.relu = act; g.addend = addend; g.add_rs = add_rs;
    g.pre = pre; g.pre_rs = l.out_pad;
    return launch<false, false>(g, s);
}

// mode: see EdmPre.  lab: labels / sigma / rnd_normal with stride lab_stride.  out: [B][d_in] row-major.
int edm_run(mtk_edm* e, int mode, int64_t B, const float* x, int64_t x_ld, const float* lab, int64_t lab_stride, const float* noise, double p_mean,
            double p_std, double sigma_data, float* out, hipStream_t s, bool keep = false) {
    if (B == 0) return MTK_OK;
    if (B > (1 << 22)) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_edm: B = %lld above 4194304 rows per call", (long long)B);
    MTK_HIP(hipSetDevice(e->device));
    const int64_t Mpad = (B + TM - 1) / TM * TM;
    int rc;
    if ((rc = edm_ensure_ws(e, Mpad, s))) return rc;
    if (keep && (rc = edm_ensure_bwd(e, Mpad, s))) return rc;
    e->kept_rows = keep ? B : 0;
    EdmPre pa{mode, e->d_in, e->dim_t, e->proj.in_pad, e->te0.in_pad, B, Mpad, x, x_ld, lab, lab_stride, noise, (float)p_mean, (float)p_std,
              (float)sigma_data, e->freqs, e->xin, e->pe, e->sig};
    hipLaunchKernelGGL(k_edm_pre, dim3((unsigned)(Mpad < 4096 ? Mpad : 4096)), dim3(256), 0, s, pa);
    if ((rc = edm_linear(e->te0, Mpad, e->pe, e->e1, 2, nullptr, 0, s, keep ? e->zt : nullptr))) return rc;   // time_embed.0 + SiLU
    if ((rc = edm_linear(e->te2, Mpad, e->e1, e->emb, 0, nullptr, 0, s))) return rc;                     // time_embed.2
    if ((rc = edm_linear(e->proj, Mpad, e->xin, e->h, 0, e->emb, e->te2.out_pad, s))) return rc;         // proj(x) + emb (modules.py:94)
    if ((rc = edm_linear(e->m[0], Mpad, e->h, e->a0, 2, nullptr, 0, s, keep ? e->z0 : nullptr))) return rc;
    if ((rc = edm_linear(e->m[1], Mpad, e->a0, e->a1, 2, nullptr, 0, s, keep ? e->z1 : nullptr))) return rc;
    if ((rc = edm_linear(e->m[2], Mpad, e->a1, e->a2, 2, nullptr, 0, s, keep ? e->z2 : nullptr))) return rc;
    if ((rc = edm_linear(e->m[3], Mpad, e->a2, e->f, 0, nullptr, 0, s))) return rc;
    EdmPost po{mode, e->d_in, e->m[3].out_pad, B, x, x_ld, noise, e->sig, 1, (float)sigma_data, e->f, out, (keep && mode == 2) ? e->dfac : nullptr};
    hipLaunchKernelGGL(k_edm_post, dim3(1024), dim3(256), 0, s, po);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

}  // namespace

extern "C" void mtk_edm_destroy(mtk_edm* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    clf_free_layer(e->proj); clf_free_layer(e->te0); clf_free_layer(e->te2);
    for (auto& l : e->m) clf_free_layer(l);
    if (e->freqs) (void)hipFree(e->freqs);
    edm_free_ws(e);
    for (int i = 0; i < 7; ++i) {
        if (e->gw[i]) (void)hipFree(e->gw[i]);
        if (e->gb[i]) (void)hipFree(e->gb[i]);
    }
    delete e;
}

extern "C" int mtk_edm_create(int device, int d_in, int dim_t, const float* const* W, const float* const* b, const float* freqs_host,
                              mtk_stream stream, mtk_edm** out) {
    if (!out) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_create: out is null");
    *out = nullptr;
    if (d_in < 1 || d_in > 8192 || dim_t < 2 || dim_t > 4096 || (dim_t & 1) || !W || !b || !freqs_host) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_create: bad argument");
    for (int i = 0; i < 7; ++i)
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_create: W[%d] or b[%d] is null", i, i);
    int n_dev = 0;
    MTK_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_create: device %d of %d", device, n_dev);
    MTK_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    mtk_edm* e = new mtk_edm();
    e->device = device; e->d_in = d_in; e->dim_t = dim_t;
    const int tp = round_up(dim_t, TN), tp2 = round_up(2 * dim_t, TN);
    int rc;
    // weight order: 0 proj, 1 time_embed.0, 2 time_embed.2, 3..6 mlp.0 / mlp.2 / mlp.4 / mlp.6 (modules.py:64-77)
    if ((rc = clf_make_layer(e->proj, d_in, dim_t, round_up(d_in, TN), W[0], b[0], s)) || (rc = clf_make_layer(e->te0, dim_t, dim_t, tp, W[1], b[1], s)) ||
        (rc = clf_make_layer(e->te2, dim_t, dim_t, tp, W[2], b[2], s)) || (rc = clf_make_layer(e->m[0], dim_t, 2 * dim_t, tp, W[3], b[3], s)) ||
        (rc = clf_make_layer(e->m[1], 2 * dim_t, 2 * dim_t, tp2, W[4], b[4], s)) || (rc = clf_make_layer(e->m[2], 2 * dim_t, dim_t, tp2, W[5], b[5], s)) ||
        (rc = clf_make_layer(e->m[3], dim_t, d_in, tp, W[6], b[6], s))) {
        mtk_edm_destroy(e);
        return rc;
    }
    hipError_t he = hipMalloc(&e->freqs, sizeof(float) * (dim_t / 2));
    if (he == hipSuccess) he = hipMemcpyAsync(e->freqs, freqs_host, sizeof(float) * (dim_t / 2), hipMemcpyHostToDevice, s);
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    if (he != hipSuccess) {
        mtk_set_error("mtk_edm_create: %s", hipGetErrorString(he));
        mtk_edm_destroy(e);
        return MTK_ERR_HIP;
    }
    *out = e;
    return MTK_OK;
}

extern "C" int mtk_edm_forward(mtk_edm* e, int64_t B, const float* x, int64_t x_ld, const float* noise_labels, int64_t label_stride, float* out,
                               mtk_stream stream) {
    if (!e || B < 0 || !x || !noise_labels || !out || x_ld < (e ? e->d_in : 0)) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_forward: bad argument");
    return edm_run(e, 0, B, x, x_ld, noise_labels, label_stride, nullptr, 0.0, 0.0, 0.0, out, (hipStream_t)stream);
}
extern "C" int mtk_edm_denoise(mtk_edm* e, int64_t B, const float* x, int64_t x_ld, const float* sigma, int64_t sigma_stride, double sigma_data,
                               float* out, mtk_stream stream) {
    if (!e || B < 0 || !x || !sigma || !out || x_ld < (e ? e->d_in : 0) || !(sigma_data > 0.0)) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_denoise: bad argument");
    return edm_run(e, 1, B, x, x_ld, sigma, sigma_stride, nullptr, 0.0, 0.0, sigma_data, out, (hipStream_t)stream);
}
extern "C" int mtk_edm_loss(mtk_edm* e, int64_t B, const float* y, int64_t y_ld, const float* rnd_normal, const float* noise, double p_mean,
                            double p_std, double sigma_data, float* loss, mtk_stream stream) {
    if (!e || B < 0 || !y || !rnd_normal || !noise || !loss || y_ld < (e ? e->d_in : 0) || !(sigma_data > 0.0)) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_loss: bad argument");
    return edm_run(e, 2, B, y, y_ld, rnd_normal, 1, noise, p_mean, p_std, sigma_data, loss, (hipStream_t)stream);
}

// mtk_edm_loss that keeps what the backward needs on the handle: layer inputs, the pre-activations of the four SiLUs and
// d loss / d F per unit of upstream gradient.
extern "C" int mtk_edm_loss_keep(mtk_edm* e, int64_t B, const float* y, int64_t y_ld, const float* rnd_normal, const float* noise, double p_mean,
                                 double p_std, double sigma_data, float* loss, mtk_stream stream) {
    if (!e || B <= 0 || !y || !rnd_normal || !noise || !loss || y_ld < (e ? e->d_in : 0) || !(sigma_data > 0.0)) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_loss_keep: bad argument");
    return edm_run(e, 2, B, y, y_ld, rnd_normal, 1, noise, p_mean, p_std, sigma_data, loss, (hipStream_t)stream, true);
}

namespace {
// dF[r][j] = gout[r][j] * dfac[r][j] for r < B, j < d_in; the padding of the [Mpad][ld] matrix is zero
__global__ void k_edm_dF(float* dF, const float* dfac, int ld, const float* gout, int64_t g_ld, int64_t B, int d_in, int64_t Mpad) {
    const int64_t n = Mpad * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int j = (int)(i - r * ld);
        dF[i] = (r < B && j < d_in) ? gout[r * g_ld + j] * dfac[i] : 0.0f;
    }
}
// dW[o][k] = sum_m dy[m][o] in[m][k], db[o] = sum_m dy[m][o] (plain stores: one reduction chunk, run-to-run identical)
int edm_wgrad(const MlpLayer& l, int64_t Mpad, const float* dy, const float* in, float* dw, float* db, hipStream_t s) {
    G2 g{};
    g.M = l.out_pad; g.N = l.in_pad; g.K = (int)Mpad; g.kchunk = g.K;
    g.A = dy; g.lda = l.out_pad; g.B = in; g.ldb = l.in_pad; g.C = dw; g.c_rs = l.in_pad; g.c_cs = 1;
    g.rowsum_slab = db;
    return launch<true, true>(g, s);
}
// dx[m][k] = sum_o dy[m][o] W[o][k]
int edm_dgrad(const MlpLayer& l, int64_t Mpad, const float* dy, float* dx, hipStream_t s) {
    G2 g{};
    g.M = (int)Mpad; g.N = l.in_pad; g.K = l.out_pad; g.kchunk = g.K;
    g.A = dy; g.lda = l.out_pad; g.B = l.w32; g.ldb = l.in_pad; g.C = dx; g.c_rs = l.in_pad; g.c_cs = 1;
    return launch<false, true>(g, s);
}
}  // namespace

// EDMLoss backward (modules.py:165-185 under autograd): gout [B][d_in] = d objective / d loss of the last mtk_edm_loss_keep on
// this handle; dW[i] [out][in] / db[i] [out] (unpadded device tensors, order of mtk_edm_create) receive d objective / d parameters.
extern "C" int mtk_edm_backward(mtk_edm* e, int64_t B, const float* gout, int64_t g_ld, float* const* dW, float* const* db, mtk_stream stream) {
    if (!e || !gout || !dW || !db || B <= 0 || g_ld < (e ? e->d_in : 0)) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_backward: bad argument");
    if (e->kept_rows != B) MTK_FAIL(MTK_ERR_STATE, "mtk_edm_backward: the last mtk_edm_loss_keep on this handle was for %lld rows, not %lld", (long long)e->kept_rows, (long long)B);
    for (int i = 0; i < 7; ++i)
        if (!dW[i] || !db[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_backward: dW[%d] or db[%d] is null", i, i);
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(e->device));
    const int64_t Mpad = (B + TM - 1) / TM * TM;
    int rc;
    const int64_t n_el = Mpad;  // rows of every activation matrix
    auto silu_bwd = [&](const float* z, float* d, int ld) {
        hipLaunchKernelGGL(k_silu_bwd, dim3(1024), dim3(256), 0, s, z, d, n_el * ld);
    };
    hipLaunchKernelGGL(k_edm_dF, dim3(1024), dim3(256), 0, s, e->dF, e->dfac, e->m[3].out_pad, gout, g_ld, B, e->d_in, Mpad);
    // mlp.6 .. mlp.0 (activations a2, a1, a0, h are the layers' inputs; z2, z1, z0 the pre-activations behind them)
    if ((rc = edm_wgrad(e->m[3], Mpad, e->dF, e->a2, e->gw[6], e->gb[6], s))) return rc;
    if ((rc = edm_dgrad(e->m[3], Mpad, e->dF, e->d0, s))) return rc;
    silu_bwd(e->z2, e->d0, e->m[2].out_pad);
    if ((rc = edm_wgrad(e->m[2], Mpad, e->d0, e->a1, e->gw[5], e->gb[5], s))) return rc;
    if ((rc = edm_dgrad(e->m[2], Mpad, e->d0, e->d1, s))) return rc;
    silu_bwd(e->z1, e->d1, e->m[1].out_pad);
    if ((rc = edm_wgrad(e->m[1], Mpad, e->d1, e->a0, e->gw[4], e->gb[4], s))) return rc;
    if ((rc = edm_dgrad(e->m[1], Mpad, e->d1, e->d0, s))) return rc;
    silu_bwd(e->z0, e->d0, e->m[0].out_pad);
    if ((rc = edm_wgrad(e->m[0], Mpad, e->d0, e->h, e->gw[3], e->gb[3], s))) return rc;
    if ((rc = edm_dgrad(e->m[0], Mpad, e->d0, e->d1, s))) return rc;                      // d1 = d h = d emb (modules.py:94: h = proj(x) + emb)
    if ((rc = edm_wgrad(e->proj, Mpad, e->d1, e->xin, e->gw[0], e->gb[0], s))) return rc;
    if ((rc = edm_wgrad(e->te2, Mpad, e->d1, e->e1, e->gw[2], e->gb[2], s))) return rc;
    if ((rc = edm_dgrad(e->te2, Mpad, e->d1, e->d0, s))) return rc;
    silu_bwd(e->zt, e->d0, e->te0.out_pad);
    if ((rc = edm_wgrad(e->te0, Mpad, e->d0, e->pe, e->gw[1], e->gb[1], s))) return rc;
    const MlpLayer* ls[7] = {&e->proj, &e->te0, &e->te2, &e->m[0], &e->m[1], &e->m[2], &e->m[3]};
    for (int i = 0; i < 7; ++i) {
        hipLaunchKernelGGL(k_unpad, dim3(256), dim3(256), 0, s, e->gw[i], ls[i]->in_pad, dW[i], ls[i]->out, ls[i]->in);
        hipLaunchKernelGGL(k_unpad, dim3(4), dim3(256), 0, s, e->gb[i], ls[i]->out_pad, db[i], 1, ls[i]->out);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// New weights into an existing handle (same shapes), in place and stream-ordered: what an optimizer step needs.
extern "C" int mtk_edm_set_weights(mtk_edm* e, const float* const* W, const float* const* b, mtk_stream stream) {
    if (!e || !W || !b) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_set_weights: null argument");
    for (int i = 0; i < 7; ++i)
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_edm_set_weights: W[%d] or b[%d] is null", i, i);
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(e->device));
    MlpLayer* ls[7] = {&e->proj, &e->te0, &e->te2, &e->m[0], &e->m[1], &e->m[2], &e->m[3]};
    for (int i = 0; i < 7; ++i) {
        hipLaunchKernelGGL(k_pad, dim3(256), dim3(256), 0, s, ls[i]->w32, ls[i]->in_pad, W[i], ls[i]->out, ls[i]->in);
        hipLaunchKernelGGL(k_pad, dim3(4), dim3(256), 0, s, ls[i]->b32, ls[i]->out_pad, b[i], 1, ls[i]->out);
    }
    e->kept_rows = 0;
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}


// =================================================================================================================
// The guidance classifier under TRAINING (models/clavaddpm/model.py:45-120 in train mode, as train.py:310-430 drives it:
// logits = classifier(x_t, t); cross_entropy(logits, y).backward(); AdamW.step()).  Per hidden stage
//     u = Linear(h);  a = ReLU | LeakyReLU(u);  y = BatchNorm1d(a) with BATCH statistics;  h' = Dropout(y)
// forward and the hand-written backward of exactly that, on the fp32 MFMA GEMM above plus four small column kernels for the
// batch-norm statistics.  The time embedding is evaluated per record (sinusoid -> Linear -> SiLU -> Linear), as the reference does,
// so its gradients are plain weight gradients.  Dropout masks are the Philox keep masks of the denoiser's iteration
// (counter = row, stage, column).  The caller owns parameters, running statistics and the optimizer; this object owns
// padded copies and the activations of one forward pass.
// =================================================================================================================
struct mtk_clft {
    int device = 0, D_in = 0, dim_t = 0, n_hidden = 0, n_out = 0;
    std::vector<float> slopes;             // [n_hidden]
    MlpLayer proj, te0, te2;
    std::vector<MlpLayer> lin;             // n_hidden + 1 (the last one is the output layer)
    std::vector<float*> gamma, beta;       // [n_hidden] padded [out_pad]
    std::vector<float*> mean, istd;        // batch statistics of the last forward, padded
    std::vector<float*> gw, gb, ggam, gbet, s1, s2;
    float* freqs = nullptr;
    int64_t cap = 0, kept_rows = 0;
    float p_drop = 0.0f, bn_eps = 1e-5f;
    uint64_t seed = 0;
    float *xin = nullptr, *pe = nullptr, *zt = nullptr, *e1 = nullptr, *emb = nullptr, *logits = nullptr, *d0 = nullptr, *d1 = nullptr;
    std::vector<float*> h, a;              // h[i]: input of stage i (h[0] = proj + emb); a[i]: activation output of stage i
};

namespace {

__global__ __launch_bounds__(256) void k_clft_pre(float* xin, int in_pad, float* pe, int tp, const float* x, int64_t x_ld, const int64_t* t, int64_t B,
                                                  int64_t Mpad, int D_in, int dim_t, const float* freqs) {
    const int half = dim_t / 2;
    for (int64_t r = blockIdx.x; r < Mpad; r += gridDim.x) {
        for (int j = threadIdx.x; j < in_pad; j += 256) xin[r * in_pad + j] = (r < B && j < D_in) ? x[r * x_ld + j] : 0.0f;
        const float tv = r < B ? (float)t[r] : 0.0f;
        for (int j = threadIdx.x; j < tp; j += 256) {  // MODEL:123-144: [cos | sin]
            float v = 0.0f;
            if (r < B && j < 2 * half) {
                const float arg = tv * freqs[j < half ? j : j - half];
                v = j < half ? cosf(arg) : sinf(arg);
            }
            pe[r * tp + j] = v;
        }
    }
}
// per column of a [B][ld] matrix: mean and 1 / sqrt(biased variance + eps) over the B rows (two passes, rows in a fixed order)
__global__ __launch_bounds__(256) void k_col_stats(const float* a, int ld, int64_t B, float eps, float* mean, float* istd, float* var_out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float sm = 0.0f;
    for (int64_t r = g; r < B; r += 4) sm += a[r * ld + c];
    red[g][threadIdx.x & 63] = sm;
    __syncthreads();
    const float mu = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)B;
    __syncthreads();
    float sq = 0.0f;
    for (int64_t r = g; r < B; r += 4) {
        const float d = a[r * ld + c] - mu;
        sq += d * d;
    }
    red[g][threadIdx.x & 63] = sq;
    __syncthreads();
    if (g == 0) {
        const float var = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)B;
        mean[c] = mu;
        istd[c] = 1.0f / sqrtf(var + eps);
        var_out[c] = var;
    }
}
// h'[r][c] = dropout(gamma xhat + beta), xhat = (a - mean) istd; rows >= B and padding columns (gamma = beta = 0) come out zero
__global__ void k_bn_fwd(const float* a, float* hn, int ld, int64_t B, int64_t Mpad, const float* mean, const float* istd, const float* gamma,
                         const float* beta, uint32_t thresh, float scale, uint64_t seed, int stage) {
    const int64_t n = Mpad * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.0f;
        if (r < B) {
            v = gamma[c] * ((a[i] - mean[c]) * istd[c]) + beta[c];
            if (thresh != 0) v = g2_keep(seed, (uint64_t)r, stage, c, thresh) ? v * scale : 0.0f;
        }
        hn[i] = v;
    }
}
// s1[c] = sum_r dy, s2[c] = sum_r dy xhat, dy = dropout'(d h')
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* dh, const float* a, int ld, int64_t B, const float* mean, const float* istd,
                                                      uint32_t thresh, float scale, uint64_t seed, int stage, float* s1, float* s2) {
    __shared__ float r1[4][64], r2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float a1 = 0.0f, a2 = 0.0f;
    const float mu = mean[c], is = istd[c];
    for (int64_t r = g; r < B; r += 4) {
        float dy = dh[r * ld + c];
        if (thresh != 0) dy = g2_keep(seed, (uint64_t)r, stage, c, thresh) ? dy * scale : 0.0f;
        a1 += dy;
        a2 += dy * ((a[r * ld + c] - mu) * is);
    }
    r1[g][threadIdx.x & 63] = a1;
    r2[g][threadIdx.x & 63] = a2;
    __syncthreads();
    if (g == 0) {
        s1[c] = r1[0][threadIdx.x] + r1[1][threadIdx.x] + r1[2][threadIdx.x] + r1[3][threadIdx.x];
        s2[c] = r2[0][threadIdx.x] + r2[1][threadIdx.x] + r2[2][threadIdx.x] + r2[3][threadIdx.x];
    }
}
// d u = act'(a) gamma istd / B (B dy - s1 - xhat s2)   (batch-norm backward through the batch statistics, then the activation);
// written over d h' in place; rows >= B zero
__global__ void k_bn_bwd_apply(float* dh, const float* a, int ld, int64_t B, int64_t Mpad, const float* mean, const float* istd, const float* gamma,
                               const float* s1, const float* s2, float slope, uint32_t thresh, float scale, uint64_t seed, int stage) {
    const int64_t n = Mpad * ld;
    const float invB = 1.0f / (float)B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.0f;
        if (r < B) {
            float dy = dh[i];
            if (thresh != 0) dy = g2_keep(seed, (uint64_t)r, stage, c, thresh) ? dy * scale : 0.0f;
            const float av = a[i], xh = (av - mean[c]) * istd[c];
            const float da = gamma[c] * istd[c] * invB * ((float)B * dy - s1[c] - xh * s2[c]);
            v = av > 0.0f ? da : slope * da;
        }
        dh[i] = v;
    }
}
__global__ void k_scale_cols(const float* s2, const float* s1, float* dgamma, float* dbeta, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { dgamma[i] = s2[i]; dbeta[i] = s1[i]; }
}

void clft_free_ws(mtk_clft* c) {
    auto fr = [](float*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(c->xin); fr(c->pe); fr(c->zt); fr(c->e1); fr(c->emb); fr(c->logits); fr(c->d0); fr(c->d1);
    for (auto& p : c->h) fr(p);
    for (auto& p : c->a) fr(p);
    c->h.clear(); c->a.clear();
    c->cap = 0; c->kept_rows = 0;
}
int clft_ensure_ws(mtk_clft* c, int64_t Mpad, hipStream_t s) {
    if (Mpad <= c->cap) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    clft_free_ws(c);
    auto al = [&](float*& p, int ld) { return hipMalloc(&p, sizeof(float) * (size_t)Mpad * ld); };
    int wide = c->proj.in_pad > c->proj.out_pad ? c->proj.in_pad : c->proj.out_pad;
    for (auto& l : c->lin) wide = l.out_pad > wide ? l.out_pad : wide;
    MTK_HIP(al(c->xin, c->proj.in_pad)); MTK_HIP(al(c->pe, c->te0.in_pad)); MTK_HIP(al(c->zt, c->te0.out_pad)); MTK_HIP(al(c->e1, c->te0.out_pad));
    MTK_HIP(al(c->emb, c->te2.out_pad)); MTK_HIP(al(c->logits, c->lin.back().out_pad)); MTK_HIP(al(c->d0, wide)); MTK_HIP(al(c->d1, wide));
    c->h.assign(c->n_hidden + 1, nullptr);
    c->a.assign(c->n_hidden, nullptr);
    MTK_HIP(al(c->h[0], c->proj.out_pad));
    for (int i = 0; i < c->n_hidden; ++i) {
        MTK_HIP(al(c->a[i], c->lin[i].out_pad));
        MTK_HIP(al(c->h[i + 1], c->lin[i].out_pad));
    }
    c->cap = Mpad;
    return MTK_OK;
}

}  // namespace

extern "C" void mtk_clft_destroy(mtk_clft* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    clf_free_layer(c->proj); clf_free_layer(c->te0); clf_free_layer(c->te2);
    for (auto& l : c->lin) clf_free_layer(l);
    auto frv = [](std::vector<float*>& v) { for (auto& p : v) if (p) (void)hipFree(p); v.clear(); };
    frv(c->gamma); frv(c->beta); frv(c->mean); frv(c->istd); frv(c->gw); frv(c->gb); frv(c->ggam); frv(c->gbet); frv(c->s1); frv(c->s2);
    if (c->freqs) (void)hipFree(c->freqs);
    clft_free_ws(c);
    delete c;
}

// widths_host: n_hidden hidden widths then n_out; slopes_host: activation slope of every hidden stage (0 = ReLU, 0.01 = LeakyReLU);
// W / b: n_hidden + 4 device tensors in the order proj, timestep_embedding.0, timestep_embedding.2, model's Linear layers;
// gamma / beta: the n_hidden BatchNorm1d affine parameters.
extern "C" int mtk_clft_create(int device, int D_in, int dim_t, int n_hidden, const int32_t* widths_host, const float* slopes_host,
                               const float* freqs_host, mtk_stream stream, mtk_clft** out) {
    if (!out) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_create: out is null");
    *out = nullptr;
    if (D_in < 1 || dim_t < 2 || dim_t > 1024 || (dim_t & 1) || n_hidden < 1 || n_hidden > 30 || !widths_host || !slopes_host || !freqs_host)
        MTK_FAIL(MTK_ERR_ARG, "mtk_clft_create: bad argument");
    for (int i = 0; i <= n_hidden; ++i)
        if (widths_host[i] < 1 || widths_host[i] > 8192) MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_clft_create: layer width %d", widths_host[i]);
    int n_dev = 0;
    MTK_HIP(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_create: device %d of %d", device, n_dev);
    MTK_HIP(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    mtk_clft* c = new mtk_clft();
    c->device = device; c->D_in = D_in; c->dim_t = dim_t; c->n_hidden = n_hidden; c->n_out = widths_host[n_hidden];
    c->slopes.assign(slopes_host, slopes_host + n_hidden);
    const int tp = round_up(dim_t, TN);
    auto shape = [&](MlpLayer& l, int in, int outw, int in_pad) -> int {  // zero-filled padded buffers; values come with mtk_clft_set_params
        l.in = in; l.out = outw; l.in_pad = in_pad; l.out_pad = round_up(outw, TN);
        MTK_HIP(hipMalloc(&l.w32, sizeof(float) * (size_t)l.out_pad * l.in_pad));
        MTK_HIP(hipMalloc(&l.b32, sizeof(float) * (size_t)l.out_pad));
        MTK_HIP(hipMemsetAsync(l.w32, 0, sizeof(float) * (size_t)l.out_pad * l.in_pad, s));
        MTK_HIP(hipMemsetAsync(l.b32, 0, sizeof(float) * (size_t)l.out_pad, s));
        return MTK_OK;
    };
    int rc;
    auto fail = [&](int code) { mtk_clft_destroy(c); return code; };
    if ((rc = shape(c->proj, D_in, dim_t, round_up(D_in, TN))) || (rc = shape(c->te0, dim_t, dim_t, tp)) || (rc = shape(c->te2, dim_t, dim_t, tp))) return fail(rc);
    c->lin.resize(n_hidden + 1);
    int prev = dim_t, prev_pad = tp;
    for (int i = 0; i <= n_hidden; ++i) {
        if ((rc = shape(c->lin[i], prev, widths_host[i], prev_pad))) return fail(rc);
        prev = widths_host[i];
        prev_pad = c->lin[i].out_pad;
    }
    auto alz = [&](std::vector<float*>& v, int i, size_t n) -> int {
        float* p = nullptr;
        MTK_HIP(hipMalloc(&p, sizeof(float) * n));
        MTK_HIP(hipMemsetAsync(p, 0, sizeof(float) * n, s));
        v[i] = p;
        return MTK_OK;
    };
    for (auto* v : {&c->gamma, &c->beta, &c->mean, &c->istd, &c->ggam, &c->gbet, &c->s1, &c->s2}) v->assign(n_hidden, nullptr);
    c->gw.assign(n_hidden + 4, nullptr);
    c->gb.assign(n_hidden + 4, nullptr);
    for (int i = 0; i < n_hidden; ++i)
        for (auto* v : {&c->gamma, &c->beta, &c->mean, &c->istd, &c->ggam, &c->gbet, &c->s1, &c->s2})
            if ((rc = alz(*v, i, (size_t)c->lin[i].out_pad))) return fail(rc);
    std::vector<MlpLayer*> all{&c->proj, &c->te0, &c->te2};
    for (auto& l : c->lin) all.push_back(&l);
    for (size_t i = 0; i < all.size(); ++i)
        if ((rc = alz(c->gw, (int)i, (size_t)all[i]->out_pad * all[i]->in_pad)) || (rc = alz(c->gb, (int)i, (size_t)all[i]->out_pad))) return fail(rc);
    hipError_t e = hipMalloc(&c->freqs, sizeof(float) * (dim_t / 2));
    if (e == hipSuccess) e = hipMemcpyAsync(c->freqs, freqs_host, sizeof(float) * (dim_t / 2), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        mtk_set_error("mtk_clft_create: %s", hipGetErrorString(e));
        return fail(MTK_ERR_HIP);
    }
    *out = c;
    return MTK_OK;
}

extern "C" int mtk_clft_set_params(mtk_clft* c, const float* const* W, const float* const* b, const float* const* gamma, const float* const* beta,
                                   mtk_stream stream) {
    if (!c || !W || !b || !gamma || !beta) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_set_params: null argument");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    std::vector<MlpLayer*> all{&c->proj, &c->te0, &c->te2};
    for (auto& l : c->lin) all.push_back(&l);
    for (size_t i = 0; i < all.size(); ++i) {
        if (!W[i] || !b[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_set_params: W[%zu] or b[%zu] is null", i, i);
        hipLaunchKernelGGL(k_pad, dim3(256), dim3(256), 0, s, all[i]->w32, all[i]->in_pad, W[i], all[i]->out, all[i]->in);
        hipLaunchKernelGGL(k_pad, dim3(4), dim3(256), 0, s, all[i]->b32, all[i]->out_pad, b[i], 1, all[i]->out);
    }
    for (int i = 0; i < c->n_hidden; ++i) {
        if (!gamma[i] || !beta[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_set_params: gamma[%d] or beta[%d] is null", i, i);
        hipLaunchKernelGGL(k_pad, dim3(4), dim3(256), 0, s, c->gamma[i], c->lin[i].out_pad, gamma[i], 1, c->lin[i].out);
        hipLaunchKernelGGL(k_pad, dim3(4), dim3(256), 0, s, c->beta[i], c->lin[i].out_pad, beta[i], 1, c->lin[i].out);
    }
    c->kept_rows = 0;
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// Training-mode forward: logits [B][n_out] row-major; batch_mean[i] / batch_var[i] (biased) [width_i] of every BatchNorm stage, for the
// caller's running statistics.  Keeps what mtk_clft_backward needs.  dropout_p in [0, 1), masks keyed by seed.
extern "C" int mtk_clft_forward(mtk_clft* c, int64_t B, const float* x, int64_t x_ld, const int64_t* t, double dropout_p, uint64_t seed, double bn_eps,
                                float* logits, float* const* batch_mean, float* const* batch_var, mtk_stream stream) {
    if (!c || !x || !t || !logits || !batch_mean || !batch_var) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_forward: null argument");
    if (B < 2) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_forward: batch statistics need at least 2 rows (torch raises on 1 too)");
    if (B > (1 << 22) || x_ld < c->D_in || !(dropout_p >= 0.0 && dropout_p < 1.0) || !(bn_eps > 0.0)) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_forward: bad argument");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    const int64_t Mpad = (B + TM - 1) / TM * TM;
    int rc;
    if ((rc = clft_ensure_ws(c, Mpad, s))) return rc;
    c->p_drop = (float)dropout_p; c->seed = seed; c->bn_eps = (float)bn_eps;
    const double th = dropout_p * 4294967296.0;
    const uint32_t thresh = dropout_p > 0.0 ? (th >= 4294967295.0 ? 4294967295u : (uint32_t)th) : 0u;
    const float scale = 1.0f / (1.0f - (float)dropout_p);
    hipLaunchKernelGGL(k_clft_pre, dim3((unsigned)(Mpad < 4096 ? Mpad : 4096)), dim3(256), 0, s, c->xin, c->proj.in_pad, c->pe, c->te0.in_pad, x, x_ld, t, B,
                       Mpad, c->D_in, c->dim_t, c->freqs);
    if ((rc = edm_linear(c->te0, Mpad, c->pe, c->e1, 2, nullptr, 0, s, c->zt))) return rc;          // timestep_embedding.0 + SiLU
    if ((rc = edm_linear(c->te2, Mpad, c->e1, c->emb, 0, nullptr, 0, s))) return rc;
    if ((rc = edm_linear(c->proj, Mpad, c->xin, c->h[0], 0, c->emb, c->te2.out_pad, s))) return rc;  // MODEL:118
    for (int i = 0; i < c->n_hidden; ++i) {
        const MlpLayer& l = c->lin[i];
        G2 g{};
        g.M = (int)Mpad; g.N = l.out_pad; g.K = l.in_pad; g.kchunk = g.K;
        g.A = c->h[i]; g.lda = l.in_pad; g.B = l.w32; g.ldb = l.in_pad; g.C = c->a[i]; g.c_rs = l.out_pad; g.c_cs = 1; g.bias = l.b32;
        g.relu = 1; g.neg_slope = c->slopes[i];
        if ((rc = launch<false, false>(g, s))) return rc;
        hipLaunchKernelGGL(k_col_stats, dim3(l.out_pad / 64), dim3(256), 0, s, c->a[i], l.out_pad, B, c->bn_eps, c->mean[i], c->istd[i], c->s1[i]);
        if (!batch_mean[i] || !batch_var[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_forward: batch_mean[%d] or batch_var[%d] is null", i, i);
        MTK_HIP(hipMemcpyAsync(batch_mean[i], c->mean[i], sizeof(float) * l.out, hipMemcpyDeviceToDevice, s));
        MTK_HIP(hipMemcpyAsync(batch_var[i], c->s1[i], sizeof(float) * l.out, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_bn_fwd, dim3(1024), dim3(256), 0, s, c->a[i], c->h[i + 1], l.out_pad, B, Mpad, c->mean[i], c->istd[i], c->gamma[i], c->beta[i],
                           thresh, scale, seed, i);
    }
    if ((rc = edm_linear(c->lin.back(), Mpad, c->h[c->n_hidden], c->logits, 0, nullptr, 0, s))) return rc;
    hipLaunchKernelGGL(k_unpad, dim3(256), dim3(256), 0, s, c->logits, c->lin.back().out_pad, logits, (int)B, c->n_out);
    MTK_HIP(hipGetLastError());
    c->kept_rows = B;
    return MTK_OK;
}

// dlogits [B][g_ld]: d objective / d logits of the last mtk_clft_forward.  dW / db: n_hidden + 4 unpadded device tensors (order of
// mtk_clft_set_params); dgamma / dbeta: n_hidden.
extern "C" int mtk_clft_backward(mtk_clft* c, int64_t B, const float* dlogits, int64_t g_ld, float* const* dW, float* const* db, float* const* dgamma,
                                 float* const* dbeta, mtk_stream stream) {
    if (!c || !dlogits || !dW || !db || !dgamma || !dbeta) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_backward: null argument");
    if (c->kept_rows != B || B <= 0) MTK_FAIL(MTK_ERR_STATE, "mtk_clft_backward: the last mtk_clft_forward on this object was for %lld rows, not %lld", (long long)c->kept_rows, (long long)B);
    if (g_ld < c->n_out) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_backward: g_ld too small");
    hipStream_t s = (hipStream_t)stream;
    MTK_HIP(hipSetDevice(c->device));
    const int64_t Mpad = (B + TM - 1) / TM * TM;
    int rc;
    const double th = (double)c->p_drop * 4294967296.0;
    const uint32_t thresh = c->p_drop > 0.0f ? (th >= 4294967295.0 ? 4294967295u : (uint32_t)th) : 0u;
    const float scale = 1.0f / (1.0f - c->p_drop);
    const int nl = c->n_hidden;
    // d logits, padded
    hipLaunchKernelGGL(k_clf_pad_rows, dim3(1024), dim3(256), 0, s, c->d0, c->lin[nl].out_pad, dlogits, g_ld, B, Mpad, c->n_out);
    float *cur = c->d0, *nxt = c->d1;
    if ((rc = edm_wgrad(c->lin[nl], Mpad, cur, c->h[nl], c->gw[3 + nl], c->gb[3 + nl], s))) return rc;
    if ((rc = edm_dgrad(c->lin[nl], Mpad, cur, nxt, s))) return rc;
    std::swap(cur, nxt);  // cur = d h[nl]
    for (int i = nl - 1; i >= 0; --i) {
        const MlpLayer& l = c->lin[i];
        hipLaunchKernelGGL(k_bn_bwd_stats, dim3(l.out_pad / 64), dim3(256), 0, s, cur, c->a[i], l.out_pad, B, c->mean[i], c->istd[i], thresh, scale, c->seed, i,
                           c->s1[i], c->s2[i]);
        hipLaunchKernelGGL(k_scale_cols, dim3(4), dim3(256), 0, s, c->s2[i], c->s1[i], c->ggam[i], c->gbet[i], l.out_pad);
        hipLaunchKernelGGL(k_bn_bwd_apply, dim3(1024), dim3(256), 0, s, cur, c->a[i], l.out_pad, B, Mpad, c->mean[i], c->istd[i], c->gamma[i], c->s1[i], c->s2[i],
                           c->slopes[i], thresh, scale, c->seed, i);
        if ((rc = edm_wgrad(l, Mpad, cur, c->h[i], c->gw[3 + i], c->gb[3 + i], s))) return rc;
        if ((rc = edm_dgrad(l, Mpad, cur, nxt, s))) return rc;
        std::swap(cur, nxt);  // cur = d h[i]
    }
    // h[0] = proj(x) + emb
    if ((rc = edm_wgrad(c->proj, Mpad, cur, c->xin, c->gw[0], c->gb[0], s))) return rc;
    if ((rc = edm_wgrad(c->te2, Mpad, cur, c->e1, c->gw[2], c->gb[2], s))) return rc;
    if ((rc = edm_dgrad(c->te2, Mpad, cur, nxt, s))) return rc;
    hipLaunchKernelGGL(k_silu_bwd, dim3(1024), dim3(256), 0, s, c->zt, nxt, Mpad * (int64_t)c->te0.out_pad);
    if ((rc = edm_wgrad(c->te0, Mpad, nxt, c->pe, c->gw[1], c->gb[1], s))) return rc;
    std::vector<MlpLayer*> all{&c->proj, &c->te0, &c->te2};
    for (auto& l : c->lin) all.push_back(&l);
    for (size_t i = 0; i < all.size(); ++i) {
        if (!dW[i] || !db[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_backward: dW[%zu] or db[%zu] is null", i, i);
        hipLaunchKernelGGL(k_unpad, dim3(256), dim3(256), 0, s, c->gw[i], all[i]->in_pad, dW[i], all[i]->out, all[i]->in);
        hipLaunchKernelGGL(k_unpad, dim3(4), dim3(256), 0, s, c->gb[i], all[i]->out_pad, db[i], 1, all[i]->out);
    }
    for (int i = 0; i < nl; ++i) {
        if (!dgamma[i] || !dbeta[i]) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_backward: dgamma[%d] or dbeta[%d] is null", i, i);
        hipLaunchKernelGGL(k_unpad, dim3(4), dim3(256), 0, s, c->ggam[i], c->lin[i].out_pad, dgamma[i], 1, c->lin[i].out);
        hipLaunchKernelGGL(k_unpad, dim3(4), dim3(256), 0, s, c->gbet[i], c->lin[i].out_pad, dbeta[i], 1, c->lin[i].out);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// The keep mask (1 = kept) Dropout stage `stage` used in the last mtk_clft_forward: uint8 [B][width of the stage], row-major.
extern "C" int mtk_clft_dropout_mask(mtk_clft* c, int stage, int64_t B, uint8_t* keep, mtk_stream stream) {
    if (!c || !keep || stage < 0 || stage >= c->n_hidden || B <= 0) MTK_FAIL(MTK_ERR_ARG, "mtk_clft_dropout_mask: bad argument");
    MTK_HIP(hipSetDevice(c->device));
    const double th = (double)c->p_drop * 4294967296.0;
    const uint32_t thresh = c->p_drop > 0.0f ? (th >= 4294967295.0 ? 4294967295u : (uint32_t)th) : 0u;
    hipLaunchKernelGGL(k_dropout_mask, dim3(1024), dim3(256), 0, (hipStream_t)stream, keep, B, c->lin[stage].out, c->seed, (uint64_t)0, stage, thresh);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}
