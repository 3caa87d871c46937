// mtk_mlp_bf16_layered.hip — layer-by-layer bf16 MFMA denoiser for the shapes the fused kernel does not cover
// (hidden widths other than 512, e.g. the 4x1024 network of BASELINE configs 4/5, unequal widths, D > 128).
//
// Same arithmetic contract as the fused kernel (mtk_mlp_bf16.hip): bf16 operands, fp32 accumulate, fp32 bias /
// time embedding / ReLU; activations round to bf16 between layers — but here they round-trip HBM as bf16
// [rows][K_pad] matrices (2 B per feature per layer each way): 4x1024 moves about 17 KB per record against
// 6.8 MFLOP, i.e. ~3.4 ms of HBM time per 2^20 records beside 2.9 ms of MFMA time.  A 1024-wide activation tile does
// not fit one CU's registers or LDS, so this shape cannot be fused the way 4x512 is.
//
// One GEMM kernel: 256 x 256 x 64 workgroup tile, 8 waves as 2 x 4, each wave 4 x 2 tiles of
// v_mfma_f32_32x32x16_bf16 (128 accumulator registers, two waves per SIMD: one wave's LDS-DMA issue and fragment
// reads ride under the other's MFMAs; with 4 waves of 4 x 4 tiles the same loop ran at 690 instead of 830 TFLOP/s).
// The tile size is set by the memory system: per 16-wide k-step a workgroup pulls 16 KB from L2 for 64 MFMAs
// (32 B/clk/CU, half of what a 128 x 128 tile needs).  Both operands are row-major [rows][K] bf16, so a lane's 8
// k-values are one 16-byte read.  L2 -> LDS by LDS-DMA into a double buffer: the next k-block lands during the
// MFMAs, one barrier per k-block, no staging registers.
//
// Measured (2^20 records, 4x1024, MI355X): 8.6 ms = 830 TFLOP/s.  Ablation builds (-DDIAG_*; wrong results, timing
// only): without the epilogue 6.1 ms, without the in-loop DMA 7.3 ms, without the MFMAs 8.1 ms — the output
// write-back (workgroups run in lock step, so all 256 store their 128 KB tiles at once, with no MFMA behind it) is
// the largest single cost; a persistent tile loop that drains tile i's stores under tile i+1's k-loop is the next step.
#include <math.h>
#include <stdlib.h>

#include "mtk_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BT = 256, BK = 64;
constexpr int NT = 2, WN = NT * 32;  // a wave's tile: 128 records (4 MFMA tiles) x 64 features (2)
constexpr int kTileBytes = BT * BK * 2, kBufBytes = 2 * kTileBytes;  // one operand tile 32 KB; records + weights 64 KB
constexpr size_t kLdsBytes = 2 * kBufBytes;  // double buffer = 128 KB (the 64 KB output staging reuses it after the k-loop)

struct GemmB {
    int64_t M;       // valid rows (fp32 output only)
    int N;           // valid output features (fp32 output only)
    int Kpad;        // multiple of BK: extent of the bf16 operands
    const uint16_t* X; int64_t x_ld;   // bf16 [Mpad][x_ld]
    const uint16_t* W;                 // bf16 [Npad256][Kpad], zero padded
    const float* bias;                 // [Npad256], zero padded
    const float* emb; int emb_ld; const int64_t* tvec; int t_scalar;  // optional time-embedding gather (first layer)
    int relu;
    // output: bf16 row-major [Mpad][y_rs] (OF32 = false) or fp32 with generic strides, bounds-checked (OF32 = true)
    void* Y; int64_t y_rs, y_cs;
};

extern __shared__ __attribute__((aligned(16))) uint16_t gb_smem[];

template <bool OF32>
__global__ __launch_bounds__(512, 1) void k_gemm_bf16(GemmB g) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3, li = lane & 31, hh = lane >> 5;
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, each with its own L2.  The nb_n column tiles
    // of one 256-record row tile get consecutive slots on ONE XCD, so the record tile is fetched from HBM once and
    // the other column tiles hit that XCD's L2 (the weights, <= 2 MiB a layer, stay resident in every L2).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nb_n = (g.N + BT - 1) / BT;
    const int64_t mblk = (int64_t)(slot / nb_n) * 8 + xcd;
    if (mblk * BT >= g.M) return;
    const int64_t m0 = mblk * BT;
    const int n0 = (slot % nb_n) * BT;

    f32x16 acc[4][NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    // Operand tiles (256 rows x 64 bf16 = one 128-byte line per row, 32 KB) go L2 -> LDS by LDS-DMA, no registers
    // involved: one wave-instruction moves 8 full lines (64 lanes x 16 B) into 1 KiB of LDS, 64 of them per k-block,
    // 8 per wave.  The LDS image is linear in the lane, so the bank swizzle is applied on the GLOBAL side: LDS slot
    // (r, cs) holds piece cs ^ ((r >> 1) & 7) of row r.  ds_read_b128 resolves banks over 256-byte rows (two tile rows)
    // in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: each group reads 8 row PAIRS whose pair index
    // is distinct mod 8, so with this key its 16 lanes land on the 16 different 16-byte slots (keyed on r & 7 instead,
    // SQ_LDS_BANK_CONFLICT showed every read 2-way).
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int lr = lane >> 3;
    unsigned voff_x[2], voff_w[2];  // [parity of the 8-row chunk]: the swizzle key (row >> 1) & 7 takes row bit 3 from it
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int lc = (lane & 7) ^ ((lr >> 1) | (par << 2));
        voff_x[par] = (unsigned)(lr * g.x_ld * 2 + lc * 16);
        voff_w[par] = (unsigned)(lr * g.Kpad * 2 + lc * 16);
    }
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(g.X + m0 * g.x_ld);
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(g.W + (int64_t)n0 * g.Kpad);
    auto dma_ops = [&](int kb, int buf, int j0, int j1) {
#ifdef DIAG_NODMA
        if (kb > 1) return;
#endif
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const int o = wv * 8 + j, operand = o >> 5, chunk = o & 31;
            const unsigned char* gp = (operand ? wb + (int64_t)chunk * 8 * g.Kpad * 2 : xb + (int64_t)chunk * 8 * g.x_ld * 2) + (int64_t)kb * BK * 2;
            const unsigned voff = operand ? voff_w[chunk & 1] : voff_x[chunk & 1];
            const unsigned dst = (unsigned)(buf * kBufBytes + operand * kTileBytes + chunk * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(gp), "s"(dst)
                         : "memory");
        }
    };
    const unsigned char* lds = reinterpret_cast<const unsigned char*>(gb_smem);
    struct Frags { bf16x8 a[4], b[NT]; };
    auto read_frags = [&](Frags& f, int buf, int ks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ra = wr * 128 + q * 32 + li, kp = ks * 2 + hh;
            f.a[q] = *reinterpret_cast<const bf16x8*>(lds + buf * kBufBytes + ra * 128 + ((kp ^ ((ra >> 1) & 7)) * 16));
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int rb = wc * WN + q * 32 + li, kp = ks * 2 + hh;
            f.b[q] = *reinterpret_cast<const bf16x8*>(lds + buf * kBufBytes + kTileBytes + rb * 128 + ((kp ^ ((rb >> 1) & 7)) * 16));
        }
    };
    // A operand = weights (rows n), B operand = records: D[n][m], i.e. lane -> record, register -> feature
    auto mfma_row = [&](const Frags& f, int mt) {
#ifdef DIAG_NOMFMA
        asm volatile("" :: "v"(f.a[mt]), "v"(f.b[0]), "v"(f.b[1]));
        return;
#endif
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[nt], f.a[mt], acc[mt][nt], 0, 0, 0);
    };
    auto mfma_all = [&](const Frags& f) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) mfma_row(f, mt);
        __builtin_amdgcn_sched_barrier(0);
    };

    // Schedule of one k-block (four 16-wide k-steps, fragment sets F0/F1 alternating, each read one k-step ahead):
    // the barrier sits BEFORE the last k-step's MFMAs, so the fragment reads that follow it (first k-step of the next
    // block) and the DMA issue (block after next, into the buffer just released) are covered by 16 MFMAs instead of
    // stalling the matrix pipe once per k-block.
    const int nkb = g.Kpad / BK;
    Frags F0, F1;
    dma_ops(0, 0, 0, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nkb > 1) dma_ops(1, 1, 0, 8);
    read_frags(F0, 0, 0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        read_frags(F1, buf, 1);
        mfma_all(F0);
        read_frags(F0, buf, 2);
        mfma_all(F1);
        read_frags(F1, buf, 3);
        mfma_all(F0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // block kb+1 landed (this wave's share); F1 is in registers
        __syncthreads();                                            // ... everyone's share, and buffer `buf` has no readers left
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kb + 1 < nkb, more2 = kb + 2 < nkb;
        if (more) read_frags(F0, buf ^ 1, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(F1, mt);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) dma_ops(kb + 2, buf, 2 * mt, 2 * mt + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // C/D map with weights as the A operand: record (m) = lane & 31, feature n = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
    // a lane holds 4 consecutive features per register quad.  fp32 output (head): stored directly through the generic
    // strides.  bf16 output: the wave's 128 x 64 tile goes through LDS (free after the k-loop) in two 64-row halves so
    // that it leaves as row-contiguous 16-byte pieces (128 B per row per wave) instead of 8-byte granules 2 KB apart.
#ifdef DIAG_NOEPI
    if (g.relu >= 0) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(g.Y)[0] = acc[1][1][3] + acc[2][0][5] + acc[3][1][7]; return; }
#endif
    // Staging image per wave: 64 rows x 128 B, no padding; the 8-byte granule gi of row r sits at granule gi ^ (r & 15),
    // so the lanes' ds_write_b64 (32 rows, same gi) and the ds_read_b128 of the copy-out both touch every bank once
    // (a padded row stride of 144 B made the writes 4-way conflicted, a quarter of the kernel's time).
    unsigned char* stage = reinterpret_cast<unsigned char*>(gb_smem) + wid * 64 * (WN * 2);  // 8 KB per wave
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t row = m0 + wr * 128 + mt * 32 + li;
        const int tr = g.emb ? ((g.tvec && row < g.M) ? (int)g.tvec[row] : (g.tvec ? 0 : g.t_scalar)) : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = nt * 32 + 8 * q + 4 * hh;  // feature inside the wave's 64
                const int n = n0 + wc * WN + nl;
                const float4 bq = *reinterpret_cast<const float4*>(g.bias + n);
                float v[4] = {acc[mt][nt][4 * q] + bq.x, acc[mt][nt][4 * q + 1] + bq.y, acc[mt][nt][4 * q + 2] + bq.z,
                              acc[mt][nt][4 * q + 3] + bq.w};
                if (g.emb && n < g.emb_ld) {
                    const float4 eq = *reinterpret_cast<const float4*>(g.emb + (int64_t)tr * g.emb_ld + n);
                    v[0] += eq.x; v[1] += eq.y; v[2] += eq.z; v[3] += eq.w;
                }
                if (g.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                }
                if (OF32) {
                    float* yf = reinterpret_cast<float*>(g.Y);
                    if (row < g.M) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (n + j < g.N) yf[row * g.y_rs + (int64_t)(n + j) * g.y_cs] = v[j];
                    }
                } else {
                    __bf16 e[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<uint2*>(stage + ((mt & 1) * 32 + li) * (WN * 2) + (((nl >> 2) ^ (li & 15)) * 8)) = *reinterpret_cast<const uint2*>(e);
                }
            }
        }
        if (!OF32 && (mt & 1)) {  // rows [wr * 128 + (mt - 1) * 32, + 64) of the wave are staged: 512 pieces, 8 per lane
            uint16_t* yb = reinterpret_cast<uint16_t*>(g.Y);
#pragma unroll
            for (int i = 0; i < WN / 8; ++i) {
                const int pce = lane + 64 * i, r = pce / (WN / 8), c8 = (pce % (WN / 8)) * 8;
                uint4 val = *reinterpret_cast<const uint4*>(stage + r * (WN * 2) + (((c8 >> 3) ^ ((r & 15) >> 1)) * 16));
                if (r & 1) val = make_uint4(val.z, val.w, val.x, val.y);  // odd key: the pair's two granules are swapped
                *reinterpret_cast<uint4*>(yb + (m0 + wr * 128 + (mt - 1) * 32 + r) * g.y_rs + n0 + wc * WN + c8) = val;
            }
        }
    }
}

__global__ void k_to_bf16_pad(uint16_t* dst, int rows_pad, int cols_pad, const float* src, int src_ld, int rows, int cols) {
    const int64_t n = (int64_t)rows_pad * cols_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
        const __bf16 b = (__bf16)((r < rows && c < cols) ? src[(int64_t)r * src_ld + c] : 0.0f);
        dst[i] = *reinterpret_cast<const uint16_t*>(&b);
    }
}
__global__ void k_pad_f32(float* dst, int n_pad, const float* src, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) dst[i] = i < n ? src[i] : 0.0f;
}
// the denoiser input (fp32, generic strides, B rows x D columns) as a zero-padded bf16 [Mpad][ld] matrix: one thread
// converts 8 consecutive columns of one row (one 16-byte store); consecutive threads take consecutive rows, the
// contiguous direction of the column-major intermediates
__global__ void k_x_to_bf16(uint16_t* dst, int ld, int64_t Mpad, const float* x, int64_t x_rs, int64_t x_cs, int64_t M, int D) {
    const int64_t n = Mpad * (ld / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i / Mpad);
        const int64_t r = i - (int64_t)cg * Mpad;
        __bf16 e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            const bool ok = r < M && c < D;
            const float v = x[(ok ? r : 0) * x_rs + (int64_t)(ok ? c : 0) * x_cs];
            e[j] = (__bf16)(ok ? v : 0.0f);
        }
        *reinterpret_cast<uint4*>(dst + r * ld + cg * 8) = *reinterpret_cast<const uint4*>(e);
    }
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <bool OF32>
int launch(const GemmB& g, int64_t Mpad, int Npad, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf16<OF32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        attr = true;
    }
    hipLaunchKernelGGL((k_gemm_bf16<OF32>), dim3((unsigned)(((Mpad / BT + 7) / 8) * 8 * (Npad / BT))), dim3(512), kLdsBytes, s, g);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

}  // namespace

void mtk_free_layered(mtk_ctx* c) {
    for (void* p : c->wb16) if (p) (void)hipFree(p);
    c->wb16.clear();
    for (float* p : c->bb16) if (p) (void)hipFree(p);
    c->bb16.clear();
    for (int i = 0; i < 2; ++i) {
        if (c->ws16[i]) (void)hipFree(c->ws16[i]);
        c->ws16[i] = nullptr;
    }
    c->ws16_elems = 0;
}

// bf16 weight images [round_up(out, 256)][round_up(in_pad, 64)] and fp32 biases padded alike (called when the weights are
// set and the fused kernel does not cover the shape)
int mtk_pack_layered(mtk_ctx* c, hipStream_t s) {
    mtk_free_layered(c);
    std::vector<const MlpLayer*> ls{&c->proj};
    for (auto& l : c->blocks) ls.push_back(&l);
    ls.push_back(&c->head);
    for (const MlpLayer* l : ls) {
        const int kp = round_up(l->in_pad, BK), np = round_up(l->out_pad, BT);
        void* p = nullptr;
        float* b = nullptr;
        MTK_HIP(hipMalloc(&p, sizeof(uint16_t) * (size_t)np * kp));
        c->wb16.push_back(p);
        MTK_HIP(hipMalloc(&b, sizeof(float) * (size_t)np));
        c->bb16.push_back(b);
        hipLaunchKernelGGL(k_to_bf16_pad, dim3(512), dim3(256), 0, s, reinterpret_cast<uint16_t*>(p), np, kp, l->w32, l->in_pad, l->out_pad, l->in_pad);
        hipLaunchKernelGGL(k_pad_f32, dim3(8), dim3(256), 0, s, b, np, l->b32, l->out_pad);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

int mtk_layered_bf16_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                             float* out, int64_t o_rs, int64_t o_cs, hipStream_t s) {
    if (c->wb16.size() != c->blocks.size() + 2) MTK_FAIL(MTK_ERR_STATE, "layered bf16 denoiser: weights are not packed");
    static const int64_t chunk = [] { const char* e = getenv("MTK_LAYERED_CHUNK"); return e ? (int64_t)atol(e) : (int64_t)(1 << 17); }();
    int maxld = round_up(c->proj.in_pad, BK);  // widest bf16 activation row (input included)
    if (round_up(c->proj.out_pad, BT) > maxld) maxld = round_up(c->proj.out_pad, BT);
    for (auto& l : c->blocks) maxld = round_up(l.out_pad, BT) > maxld ? round_up(l.out_pad, BT) : maxld;
    const int64_t need_rows = (B < chunk ? (B + BT - 1) / BT * BT : chunk);
    const size_t need = (size_t)need_rows * maxld;
    if (need > c->ws16_elems) {
        MTK_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < 2; ++i) {
            if (c->ws16[i]) (void)hipFree(c->ws16[i]);
            c->ws16[i] = nullptr;
            MTK_HIP(hipMalloc(&c->ws16[i], sizeof(uint16_t) * need));
        }
        c->ws16_elems = need;
    }
    uint16_t* ws[2] = {reinterpret_cast<uint16_t*>(c->ws16[0]), reinterpret_cast<uint16_t*>(c->ws16[1])};
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t M = (B - r0 < chunk) ? B - r0 : chunk;
        const int64_t Mpad = (M + BT - 1) / BT * BT;
        int rc;
        // input -> bf16 [Mpad][kp0]
        const int kp0 = round_up(c->proj.in_pad, BK);
        hipLaunchKernelGGL(k_x_to_bf16, dim3(2048), dim3(256), 0, s, ws[0], kp0, Mpad, x + r0 * x_rs, x_rs, x_cs, M, c->D);
        MTK_HIP(hipGetLastError());
        int cur = 0;
        GemmB g{};
        g.M = M; g.N = c->proj.out; g.Kpad = kp0;
        g.X = ws[cur]; g.x_ld = kp0;
        g.W = reinterpret_cast<const uint16_t*>(c->wb16[0]); g.bias = c->bb16[0];
        g.emb = c->d_emb; g.emb_ld = c->emb_ld; g.tvec = t ? t + r0 : nullptr; g.t_scalar = t_scalar;
        int ld = round_up(c->proj.out_pad, BT);
        g.Y = ws[cur ^ 1]; g.y_rs = ld;
        if ((rc = launch<false>(g, Mpad, ld, s))) return rc;
        cur ^= 1;
        for (size_t i = 0; i < c->blocks.size(); ++i) {
            const MlpLayer& l = c->blocks[i];
            GemmB h{};
            h.M = M; h.N = l.out; h.Kpad = round_up(l.in_pad, BK);
            h.X = ws[cur]; h.x_ld = ld;
            h.W = reinterpret_cast<const uint16_t*>(c->wb16[1 + i]); h.bias = c->bb16[1 + i]; h.relu = 1;
            ld = round_up(l.out_pad, BT);
            h.Y = ws[cur ^ 1]; h.y_rs = ld;
            if ((rc = launch<false>(h, Mpad, ld, s))) return rc;
            cur ^= 1;
        }
        GemmB h{};
        h.M = M; h.N = c->head.out; h.Kpad = round_up(c->head.in_pad, BK);
        h.X = ws[cur]; h.x_ld = ld;
        h.W = reinterpret_cast<const uint16_t*>(c->wb16.back()); h.bias = c->bb16.back();
        h.Y = out + r0 * o_rs; h.y_rs = o_rs; h.y_cs = o_cs;
        if ((rc = launch<true>(h, Mpad, round_up(c->head.out_pad, BT), s))) return rc;
    }
    return MTK_OK;
}
