// mtk_mlp_bf16.hip — fused MLPDiffusion forward on CDNA4 bf16 MFMA (v_mfma_f32_32x32x16_bf16).
//
// Design (MI355X-first, no reference counterpart — the reference is eager nn.Linear calls, MODEL:596-615):
//   * One workgroup = 4 waves = one wave per SIMD, 512 registers each.  A wave owns 32 records and keeps
//     their activations IN REGISTERS for the whole network: HBM sees only the D-wide input row and the
//     D-wide output row (or one 4-byte score in Tartan-Federer mode).
//   * The network is evaluated transposed, Y^T[n][record] = W[n][k] . X^T[k][record]: the weight tile is the
//     MFMA A operand, the activations are the B operand.  A 32x32 accumulator (feature on the register
//     index, record on the lane) is then, after bias/ReLU and a pairwise cvt to bf16, directly the B
//     operand of two k-steps of the next layer — no LDS round trip, no cross-lane movement.  The k order
//     inside such a k-step is permuted (feature 16s + 8(j>>2) + 4(lane>>5) + (j&3)); the weights are
//     packed ONCE on the device (mtk_mlp_set_weights) in exactly that order, as a linear stream of 1-KiB
//     MFMA A-fragments in consumption order.
//   * That stream is what every workgroup reads, 32 KiB chunk by chunk, through LDS-DMA
//     (global_load_lds_dwordx4, one fragment per wave-instruction) into a double buffer; the four waves
//     share each fragment (ds_read_b128, lane-linear, conflict-free).  One barrier per chunk (32 MFMAs per
//     wave).  The stream (1.7 MB at 4x512) stays L2-resident; workgroups are persistent over row tiles.
//   * Time embedding is a [T, 128] table (built when the weights are set) gathered by each record's t and
//     folded into the initial accumulator together with the bias.
#include "mtk_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChunkFrags = 32;
constexpr int kFragBytes = 1024;
constexpr int kChunkBytes = kChunkFrags * kFragBytes;
constexpr int kDimT = 128;
constexpr int kMaxTF = 64;  // timesteps per fused Tartan-Federer launch

struct FusedArgs {
    const unsigned char* wfrag;
    const float* bias;
    int n_bias;
    const float* emb;
    int emb_ld;
    int n_hidden;
    int total_chunks;
    int64_t rows;
    int mode;  // 0: denoiser forward, 1: Tartan-Federer score
    int D;
    // mode 0
    const float* x; int64_t x_rs, x_cs;
    const int64_t* tvec; int t_scalar;
    float* out; int64_t o_rs, o_cs;
    // mode 1
    int64_t BP; int P; int64_t x_ld;
    const float* noise;
    float* score;
    const SchedRow* sched;
    int add_t;
    int t_list[kMaxTF];
};

extern __shared__ __attribute__((aligned(16))) unsigned char mtk_smem_f[];

// feature index held by element j of lane-half hh in k-step ks (== the C/D row map of the 32x32 MFMA)
__device__ __forceinline__ constexpr int feat(int ks, int hh, int j) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
}

struct Pipe {
    const unsigned char* wfrag;
    int total_chunks;
    int next;      // next chunk of the stream to fetch
    int buf;       // buffer (0/1) of the chunk being consumed
    int lane, wid;

    __device__ __forceinline__ void issue(int into) {
        const unsigned char* src = wfrag + (size_t)next * kChunkBytes + (size_t)lane * 16;
#pragma unroll
        for (int q = 0; q < kChunkFrags / 4; ++q) {
            const int f = wid + 4 * q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)f * kFragBytes),
                                             (__attribute__((address_space(3))) void*)(mtk_smem_f + into * kChunkBytes + f * kFragBytes),
                                             16, 0, 0);
        }
        next = (next + 1 == total_chunks) ? 0 : next + 1;
    }
    // make the next chunk current: its DMA (issued one chunk ago) is drained by the barrier's vmcnt(0);
    // every wave has finished reading the other buffer, so the chunk after it can be fetched into it
    __device__ __forceinline__ void advance() {
        __syncthreads();
        buf ^= 1;
        issue(buf ^ 1);
    }
    __device__ __forceinline__ bf16x8 frag(int f) const {
        return *reinterpret_cast<const bf16x8*>(mtk_smem_f + buf * kChunkBytes + f * kFragBytes + lane * 16);
    }
};

__device__ __forceinline__ f32x16 load16(const float* p) {  // p + {0..3, 8..11, 16..19, 24..27}
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + 8 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
}

__device__ __forceinline__ void pack_out(const f32x16& v, bf16x8& lo, bf16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = (__bf16)v[j];
        hi[j] = (__bf16)v[8 + j];
    }
}

// One dense layer: NT output tiles of 32 features, KS k-steps of 16 inputs.  NPC tiles share a chunk.
template <int KS, int NT, bool RELU, int NIN, int NOUT>
__device__ __forceinline__ void dense(const bf16x8 (&Xin)[NIN], bf16x8 (&Xout)[NOUT], Pipe& p, const float* sbias, int hh) {
    constexpr int NPC = kChunkFrags / KS;
    static_assert(NPC >= 1 && NT % NPC == 0, "layer must fill whole chunks");
#pragma unroll
    for (int c = 0; c < NT / NPC; ++c) {
        p.advance();
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const int nt = c * NPC + q;
            f32x16 acc = load16(sbias + 32 * nt + 4 * hh);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.frag(q * KS + ks), Xin[ks], acc, 0, 0, 0);
            if (RELU) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
            }
            pack_out(acc, Xout[2 * nt], Xout[2 * nt + 1]);
        }
    }
}

// Output layer: DT tiles of 32 features out of H inputs; the result is stored (mode 0) or reduced (mode 1).
template <int KS, int DT, int NIN>
__device__ __forceinline__ void head(const bf16x8 (&Xin)[NIN], Pipe& p, const float* sbias, int hh, const FusedArgs& a, int64_t row,
                                     bool valid) {
    constexpr int NPC = (kChunkFrags / KS) < 1 ? 1 : (kChunkFrags / KS);
    constexpr int NCH = (DT + NPC - 1) / NPC;
    float sq = 0.0f;
    const int p_idx = (a.mode == 1) ? (int)(row % a.P) : 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        p.advance();
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const int nt = c * NPC + q;
            if (nt >= DT) continue;
            f32x16 acc = load16(sbias + 32 * nt + 4 * hh);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.frag(q * KS + ks), Xin[ks], acc, 0, 0, 0);
            if (a.mode == 0) {
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * nt + 8 * (r >> 2) + 4 * hh + (r & 3);
                        if (f < a.D) a.out[row * a.o_rs + (int64_t)f * a.o_cs] = acc[r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * nt + 8 * (r >> 2) + 4 * hh + (r & 3);
                    if (f < a.D) {
                        const float d = a.noise[p_idx * a.D + f] - acc[r];
                        sq += d * d;
                    }
                }
            }
        }
    }
    if (a.mode == 1) {
        sq += __shfl_xor(sq, 32);
        if (valid && hh == 0) a.score[row] = sq / (float)a.D;
    }
}

template <int H, int DT>
__global__ __launch_bounds__(256, 1) void k_fused(FusedArgs a) {
    constexpr int KS0 = 2 * DT;   // input k-steps (D padded to 32 DT)
    constexpr int KSH = H / 16;   // k-steps of a hidden activation
    constexpr int NTH = H / 32;   // tiles of a hidden layer
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sbias = reinterpret_cast<float*>(mtk_smem_f + 2 * kChunkBytes);
    for (int i = tid; i < a.n_bias; i += 256) sbias[i] = a.bias[i];

    Pipe p{a.wfrag, a.total_chunks, 0, 1, lane, wid};
    p.issue(0);  // chunk 0 -> buffer 0; the first advance() flips buf to 0

    const int64_t n_tiles = (a.rows + 127) / 128;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row = tile * 128 + wid * 32 + li;
        const bool valid = row < a.rows;
        bf16x8 XA[KSH], XB[KSH];
        int t_row = 0;
        // ---- input fragments
        if (a.mode == 0) {
            t_row = a.tvec ? (valid ? (int)a.tvec[row] : 0) : a.t_scalar;
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = feat(ks, hh, j);
                    const float v = (valid && f < a.D) ? a.x[row * a.x_rs + (int64_t)f * a.x_cs] : 0.0f;
                    XA[ks][j] = (__bf16)v;
                }
        } else {
            const int64_t k = valid ? row / a.BP : 0;
            const int64_t rr = row - k * a.BP;
            const int64_t rec = rr / a.P;
            const int pp = (int)(rr - rec * a.P);
            t_row = a.t_list[k];
            const SchedRow sr = a.sched[t_row + a.add_t];
#pragma unroll
            for (int ks = 0; ks < KS0; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = feat(ks, hh, j);
                    const float v = (valid && f < a.D) ? sr.sqrt_acp * a.x[rec * a.x_ld + f] + sr.sqrt_1m_acp * a.noise[pp * a.D + f] : 0.0f;
                    XA[ks][j] = (__bf16)v;
                }
        }
        // ---- proj (D -> 128) + bias + emb[t]   (MODEL:614)
        {
            p.advance();
            const float* er = a.emb + (int64_t)t_row * a.emb_ld;
#pragma unroll
            for (int nt = 0; nt < kDimT / 32; ++nt) {
                f32x16 acc = load16(sbias + 32 * nt + 4 * hh);
                const f32x16 e = load16(er + 32 * nt + 4 * hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += e[r];
#pragma unroll
                for (int ks = 0; ks < KS0; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p.frag(nt * KS0 + ks), XA[ks], acc, 0, 0, 0);
                pack_out(acc, XB[2 * nt], XB[2 * nt + 1]);
            }
        }
        // ---- hidden layers (MODEL:300-301), activations alternate between the two register arrays
        const float* sb = sbias + kDimT;
        dense<kDimT / 16, NTH, true>(XB, XA, p, sb, hh);
        sb += H;
        int rem = a.n_hidden - 1;
        while (rem >= 2) {
            dense<KSH, NTH, true>(XA, XB, p, sb, hh);
            sb += H;
            dense<KSH, NTH, true>(XB, XA, p, sb, hh);
            sb += H;
            rem -= 2;
        }
        if (rem == 1) {
            dense<KSH, NTH, true>(XA, XB, p, sb, hh);
            sb += H;
            head<KSH, DT>(XB, p, sb, hh, a, row, valid);
        } else {
            head<KSH, DT>(XA, p, sb, hh, a, row, valid);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last prefetch must land before the LDS is released
}

__global__ void k_pack_frags(unsigned short* dst, const float* W, int out, int in, int NT, int KS, int64_t frag0) {
    const int64_t n = (int64_t)NT * KS * 512;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t frag = e >> 9;
        const int lane = (int)((e >> 3) & 63), j = (int)(e & 7);
        const int nt = (int)(frag / KS), ks = (int)(frag % KS);
        const int nrow = 32 * nt + (lane & 31);
        const int f = feat(ks, lane >> 5, j);
        const float v = (nrow < out && f < in) ? W[(int64_t)nrow * in + f] : 0.0f;
        const __bf16 b = (__bf16)v;
        dst[(frag0 + frag) * 512 + lane * 8 + j] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

struct Layout {
    int H, DT, n_hidden, total_chunks, n_bias;
};

bool layout_for(const mtk_ctx* c, Layout& L) {
    if (c->dim_t != kDimT || c->n_hidden < 1) return false;
    const int H = c->hidden[0];
    for (int h : c->hidden)
        if (h != H) return false;
    if (H != 512) return false;
    if (c->D < 1 || c->D > 128) return false;
    L.H = H;
    L.DT = (c->D + 31) / 32;
    L.n_hidden = c->n_hidden;
    const int ksh = H / 16, nth = H / 32;
    const int c_l1 = nth * (kDimT / 16) / kChunkFrags;
    const int c_hh = nth * ksh / kChunkFrags;
    const int c_head = (L.DT * ksh + kChunkFrags - 1) / kChunkFrags;
    L.total_chunks = 1 + c_l1 + (c->n_hidden - 1) * c_hh + c_head;
    L.n_bias = kDimT + c->n_hidden * H + 32 * L.DT;
    return true;
}

template <int H>
int launch(const Layout& L, const FusedArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (L.DT) {
        case 1: hipLaunchKernelGGL((k_fused<H, 1>), dim3(grid), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((k_fused<H, 2>), dim3(grid), dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL((k_fused<H, 3>), dim3(grid), dim3(256), lds, s, a); break;
        case 4: hipLaunchKernelGGL((k_fused<H, 4>), dim3(grid), dim3(256), lds, s, a); break;
        default: MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused denoiser: D tiles = %d", L.DT);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

int n_cus(int device) {
    static int cached[64] = {0};
    if (device < 64 && cached[device]) return cached[device];
    int n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
    if (device < 64) cached[device] = n;
    return n;
}

int run(mtk_ctx* c, FusedArgs& a, hipStream_t s) {
    Layout L;
    if (!layout_for(c, L)) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused denoiser: unsupported shape");
    a.wfrag = reinterpret_cast<const unsigned char*>(c->d_wfrag);
    a.bias = c->d_bias_all;
    a.n_bias = L.n_bias;
    a.emb = c->d_emb;
    a.emb_ld = c->emb_ld;
    a.n_hidden = L.n_hidden;
    a.total_chunks = L.total_chunks;
    a.D = c->D;
    const int64_t tiles = (a.rows + 127) / 128;
    const int cus = n_cus(c->device);
    const int grid = (int)(tiles < cus ? tiles : cus);
    const size_t lds = 2 * kChunkBytes + (((size_t)L.n_bias * 4 + 15) & ~(size_t)15);
    return launch<512>(L, a, grid, lds, s);
}

}  // namespace

void mtk_free_fused(mtk_ctx* c) {
    if (c->d_wfrag) (void)hipFree(c->d_wfrag);
    if (c->d_bias_all) (void)hipFree(c->d_bias_all);
    c->d_wfrag = nullptr;
    c->d_bias_all = nullptr;
    c->wfrag_bytes = 0;
    c->fused_ok = false;
}

// Packs the fp32 weights already held by the context (padded images of mtk_mlp_f32.hip) into the bf16
// fragment stream.  Shapes the fused kernel does not cover leave fused_ok = false (MTK_PREC_BF16 then
// reports MTK_ERR_UNSUPPORTED; MTK_PREC_FP32 always works).
int mtk_pack_fused(mtk_ctx* c, hipStream_t s) {
    mtk_free_fused(c);
    Layout L;
    if (!layout_for(c, L)) return MTK_OK;
    static bool attr_set = false;
    if (!attr_set) {
        const int max_lds = 2 * kChunkBytes + 64 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        attr_set = true;
    }
    c->wfrag_bytes = (size_t)L.total_chunks * kChunkBytes;
    MTK_HIP(hipMalloc(&c->d_wfrag, c->wfrag_bytes));
    MTK_HIP(hipMalloc(&c->d_bias_all, sizeof(float) * L.n_bias));
    MTK_HIP(hipMemsetAsync(c->d_wfrag, 0, c->wfrag_bytes, s));
    MTK_HIP(hipMemsetAsync(c->d_bias_all, 0, sizeof(float) * L.n_bias, s));
    unsigned short* dst = reinterpret_cast<unsigned short*>(c->d_wfrag);
    const int H = L.H, ksh = H / 16, nth = H / 32;
    int64_t chunk = 0;
    int boff = 0;
    auto pack = [&](const MlpLayer& l, int NT, int KS, int chunks) {
        hipLaunchKernelGGL(k_pack_frags, dim3(512), dim3(256), 0, s, dst, l.w32, l.out_pad, l.in_pad, NT, KS, chunk * kChunkFrags);
        (void)hipMemcpyAsync(c->d_bias_all + boff, l.b32, sizeof(float) * (l.out < 32 * NT ? l.out : 32 * NT), hipMemcpyDeviceToDevice, s);
        chunk += chunks;
    };
    // note: w32 is the zero-padded [out_pad][in_pad] image, so out_pad/in_pad are its true extents
    pack(c->proj, kDimT / 32, 2 * L.DT, 1);
    boff += kDimT;
    pack(c->blocks[0], nth, kDimT / 16, nth * (kDimT / 16) / kChunkFrags);
    boff += H;
    for (int i = 1; i < c->n_hidden; ++i) {
        pack(c->blocks[i], nth, ksh, nth * ksh / kChunkFrags);
        boff += H;
    }
    pack(c->head, L.DT, ksh, (L.DT * ksh + kChunkFrags - 1) / kChunkFrags);
    MTK_HIP(hipGetLastError());
    MTK_HIP(hipStreamSynchronize(s));
    c->fused_ok = true;
    c->fused_h = H;
    return MTK_OK;
}

int mtk_fused_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                      float* out, int64_t o_rs, int64_t o_cs, hipStream_t s) {
    FusedArgs a{};
    a.rows = B;
    a.mode = 0;
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs;
    a.tvec = t; a.t_scalar = t_scalar;
    a.out = out; a.o_rs = o_rs; a.o_cs = o_cs;
    return run(c, a, s);
}

int mtk_fused_tf_score(mtk_ctx* c, int64_t B, int P, const float* x, int64_t x_ld, const float* noise, int n_t,
                       const int32_t* t_list_host, int add_t, float* score, hipStream_t s) {
    for (int k0 = 0; k0 < n_t; k0 += kMaxTF) {
        const int nk = (n_t - k0 < kMaxTF) ? n_t - k0 : kMaxTF;
        FusedArgs a{};
        a.mode = 1;
        a.BP = B * P;
        a.P = P;
        a.rows = a.BP * nk;
        a.x = x; a.x_ld = x_ld;
        a.noise = noise;
        a.score = score + (int64_t)k0 * a.BP;
        a.sched = c->d_rows;
        a.add_t = add_t;
        for (int k = 0; k < nk; ++k) a.t_list[k] = t_list_host[k0 + k];
        const int rc = run(c, a, s);
        if (rc) return rc;
    }
    return MTK_OK;
}
