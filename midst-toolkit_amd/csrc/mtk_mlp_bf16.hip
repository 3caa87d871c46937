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
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "mtk_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kChunkFrags = 32;
constexpr int kFragBytes = 1024;
constexpr int kChunkBytes = kChunkFrags * kFragBytes;
constexpr int kDimT = 128;
constexpr int kMaxTF = 64;  // timesteps per fused Tartan-Federer launch
constexpr int kBufs = 4;     // LDS chunk buffers
constexpr int kMaxNoise = 2048;  // floats of the [P, Nn] noise table staged in LDS

struct FusedArgs {
    const unsigned char* wfrag;
    const float* bias;
    int n_bias;
    const float* embb;  // [T][emb_ld] time-embedding table with the proj bias folded in
    int emb_ld;
    int n_noise;        // floats of the Tartan-Federer noise table staged in LDS (P * D)
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
    unsigned long long* stamps;  // diagnostic: per-phase cycle sums (MTK_FUSED_STAMPS=1), else null
    int exp;  // diagnostic switches (MTK_FUSED_EXP): 1 = no weight DMA after the prologue, 2 = no chunk barrier
    int t_list[kMaxTF];
};

extern __shared__ __attribute__((aligned(16))) unsigned char mtk_smem_f[];

// feature index held by element j of lane-half hh in k-step ks (== the C/D row map of the 32x32 MFMA)
__device__ __forceinline__ constexpr int feat(int ks, int hh, int j) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Weight-fragment pipeline.  Chunk c of the stream lives in LDS buffer c % kBufs (kBufs = 4).  While chunk c is
// consumed, chunk c+1 has landed, chunk c+2 is in flight and chunk c+3 is being requested, one 1-KiB piece per
// wave every fourth MFMA (an LDS-DMA costs the issuing wave tens of cycles, so the pieces are spread out rather
// than issued as a burst that would drain the matrix pipe).
struct Pipe {
    const unsigned char* src;  // wfrag + lane * 16
    int total_chunks;
    int next;                  // chunk of the stream being requested
    unsigned rd, dma;          // LDS byte offsets: buffer being consumed, buffer being filled
    int lane, wid;
    int exp;

    // piece K (0..7) of the chunk being requested: fragment wid + 4 K; 64 lanes x 16 B, dest = base + lane * 16
    template <int K>
    __device__ __forceinline__ void piece() {
        if (!(exp & 1)) {
            const int f = wid + 4 * K;
            const unsigned dst = __builtin_amdgcn_readfirstlane(dma + (unsigned)f * kFragBytes);
            const unsigned char* gp = src + (size_t)next * kChunkBytes + (size_t)f * kFragBytes;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(gp), "s"(dst)
                : "memory");
        }
        if constexpr (K == 7) {
            next = (next + 1 == total_chunks) ? 0 : next + 1;
            dma = (dma + kChunkBytes == kBufs * kChunkBytes) ? 0u : dma + kChunkBytes;
        }
    }
    __device__ __forceinline__ void fetch_all() {
        static_for<8>([&](auto ic) { piece<decltype(ic)::value>(); });
    }
    // Placed before the last quarter of chunk m.  vmcnt(8): all of this wave's pieces except the 8 youngest
    // (chunk m+2) have landed, so chunk m+1 is complete; the barrier extends that to every wave and also says
    // every wave is done with chunk m-1, whose buffer the following pieces (chunk m+3) overwrite.
    __device__ __forceinline__ void sync() {
        if (!(exp & 2)) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    __device__ __forceinline__ void rotate() { rd = (rd + kChunkBytes == kBufs * kChunkBytes) ? 0u : rd + kChunkBytes; }
    __device__ __forceinline__ bf16x8 ld(int f) const {
        return *reinterpret_cast<const bf16x8*>(mtk_smem_f + rd + f * kFragBytes + lane * 16);
    }
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)

// Consume the current chunk (32 fragments) in four quarters.  `step(ic<I>, w)` issues the MFMA of fragment I.
// Fragment i of the NEXT quarter is read from LDS right before the MFMA of fragment i of the current one, so
// every ds_read_b128 has eight MFMAs (256 cycles) to return; the issue order is pinned (the scheduler otherwise
// sinks the reads next to their uses).  On entry w0 holds fragments 0..7 of the chunk; on exit w0 holds
// fragments 0..7 of the next chunk.
template <int Q, class Step>
__device__ __forceinline__ void quarter(Pipe& p, bf16x8 (&cur)[8], bf16x8 (&nxt)[8], Step&& step) {
    static_for<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        nxt[i] = p.ld(((Q + 1) & 3) * 8 + i);
        step(std::integral_constant<int, Q * 8 + i>{}, cur[i]);
        if constexpr (i == 1) p.template piece<(2 * Q + 2) & 7>();
        if constexpr (i == 5) p.template piece<(2 * Q + 3) & 7>();
        PIN();
    });
}
template <class Step>
__device__ __forceinline__ void consume_chunk(Pipe& p, bf16x8 (&w0)[8], bf16x8 (&w1)[8], Step&& step) {
    quarter<0>(p, w0, w1, step);  // pieces 2, 3
    quarter<1>(p, w1, w0, step);  // pieces 4, 5
    quarter<2>(p, w0, w1, step);  // pieces 6, 7 -> the request moves on to the next chunk
    p.sync();
    PIN();
    p.rotate();                   // the last quarter prefetches fragments 0..7 of the next chunk
    quarter<3>(p, w1, w0, step);  // pieces 0, 1 of the new request
}

__device__ __forceinline__ f32x16 load16(const float* p) {  // p + {0..3, 8..11, 16..19, 24..27}
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + 8 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
}

// max(x, 0) as one integer max on the bit pattern (negative floats are negative integers)
__device__ __forceinline__ float relu_bits(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// bias/ReLU epilogue of one 32-feature tile: the fp32 accumulator becomes two bf16 k-steps of the next layer
template <bool RELU>
__device__ __forceinline__ void pack_out(const f32x16& v, bf16x8& lo, bf16x8& hi) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lo[j] = (__bf16)(RELU ? relu_bits(v[j]) : v[j]);
        hi[j] = (__bf16)(RELU ? relu_bits(v[8 + j]) : v[8 + j]);
    }
}

// One dense layer, k-step-major: fragment g of the layer multiplies k-step g / NT into accumulator g % NT, so
// all NT accumulators (AGPRs) advance together and one activation fragment feeds NT MFMAs.  In the LAST k-step
// each MFMA writes a fresh register tile whose epilogue (ReLU, cvt to bf16, in-place store into the activation
// array) is issued one MFMA later, and the accumulator is reloaded with the next layer's bias.
template <int KS, int NT, int CHUNKS, bool RELU, int NT_NEXT, int NIN, int NOUT, int NACC>
__device__ __forceinline__ void dense(Pipe& p, bf16x8 (&w0)[8], bf16x8 (&w1)[8], const bf16x8 (&Xin)[NIN], bf16x8 (&Xout)[NOUT],
                                      f32x16 (&acc)[NACC], const float* next_bias, int hh) {
    // epilogue of one finished tile: ReLU, cvt to bf16 into the activation array, accumulator <- next bias
    auto finish = [&](auto tc) {
        constexpr int nt = decltype(tc)::value;
        pack_out<RELU>(acc[nt], Xout[2 * nt], Xout[2 * nt + 1]);
        if constexpr (nt < NT_NEXT) acc[nt] = load16(next_bias + 32 * nt + 4 * hh);
    };
    static_for<CHUNKS>([&](auto cc) {
        consume_chunk(p, w0, w1, [&](auto ic, const bf16x8& w) {
            constexpr int g = decltype(cc)::value * kChunkFrags + decltype(ic)::value, ks = g / NT, nt = g % NT;
            if constexpr (ks < KS) acc[nt] = MFMA(w, Xin[ks], acc[nt]);
            if constexpr (ks == KS - 1) {  // last k-step: finish the tile issued two MFMAs ago
                if (p.exp & 4) return;
                if constexpr (nt >= 2) finish(std::integral_constant<int, nt - 2>{});
            }
        });
    });
    if (!(p.exp & 4)) {
        if constexpr (NT >= 2) finish(std::integral_constant<int, NT - 2>{});
        finish(std::integral_constant<int, NT - 1>{});
    }
    static_for<NT_NEXT>([&](auto ic) {
        constexpr int nt = decltype(ic)::value;
        if constexpr (nt >= NT) acc[nt] = load16(next_bias + 32 * nt + 4 * hh);
    });
}

template <int H, int DT, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void k_fused(FusedArgs a) {
    constexpr int KS0 = 2 * DT;   // input k-steps (D padded to 32 DT)
    constexpr int KSH = H / 16;   // k-steps of a hidden activation
    constexpr int NTH = H / 32;   // tiles of a hidden layer
    constexpr int HA = NTH - 4;   // first of the four accumulators the head uses (split-K)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS after the chunk buffers: biases (+H zeros), the Tartan-Federer noise table, per-timestep (a, b)
    float* sbias = reinterpret_cast<float*>(mtk_smem_f + kBufs * kChunkBytes);
    float* snoise = sbias + a.n_bias + H;
    float2* scoef = reinterpret_cast<float2*>(snoise + a.n_noise);
    for (int i = tid; i < a.n_bias + H; i += 256) sbias[i] = i < a.n_bias ? a.bias[i] : 0.0f;
    if (a.mode == 1) {
        for (int i = tid; i < a.n_noise; i += 256) snoise[i] = a.noise[i];
        for (int i = tid; i < kMaxTF; i += 256) {
            const SchedRow sr = a.sched[a.t_list[i] + a.add_t];
            scoef[i] = make_float2(sr.sqrt_acp, sr.sqrt_1m_acp);
        }
    }

    Pipe p{a.wfrag + (size_t)lane * 16, a.total_chunks, 0, 0u, 0u, lane, wid, a.exp};
    {
        const int keep_exp = p.exp;
        p.exp &= ~1;
        p.fetch_all();  // chunk 0 -> buffer 0
        p.fetch_all();  // chunk 1 -> buffer 1
        p.template piece<0>();  // chunk 2 -> buffer 2, the rest of it is requested while chunk 0 is consumed
        p.template piece<1>();
        p.exp = keep_exp;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunks 0, 1 and the LDS tables visible to every wave
    bf16x8 w0[8], w1[8];
    static_for<8>([&](auto ic) { w0[decltype(ic)::value] = p.ld(decltype(ic)::value); });

    f32x16 acc[NTH];
    bf16x8 X[KSH];
    unsigned long long st_prev = 0, st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            PIN();
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            PIN();
            if (slot >= 0) st_sum[slot] += tnow - st_prev;
            st_prev = tnow;
        }
    };

    // ---- per-tile inputs are requested one tile ahead (a single wave per SIMD has nothing else to hide a
    // ---- global-memory round trip behind): the timestep first, then the record and the emb[t] + bias rows
    const int64_t n_tiles = (a.rows + 127) / 128;
    struct Where { int64_t row; bool valid; int64_t xoff; int kk, pp; };
    auto locate = [&](int64_t tile) {
        Where w;
        w.row = tile * 128 + wid * 32 + li;
        w.valid = tile < n_tiles && w.row < a.rows;
        w.kk = 0; w.pp = 0;
        if (a.mode == 0) {
            w.xoff = w.row * a.x_rs;
        } else {
            const int64_t k = w.valid ? w.row / a.BP : 0;
            const int64_t rr = w.row - k * a.BP;
            const int64_t rec = rr / a.P;
            w.kk = (int)k;
            w.pp = (int)(rr - rec * a.P);
            w.xoff = rec * a.x_ld;
        }
        return w;
    };
    auto request_t = [&](const Where& w) -> int {
        if (a.mode == 0) return a.tvec ? (w.valid ? (int)a.tvec[w.row] : 0) : a.t_scalar;
        return a.t_list[w.kk];
    };
    float raw[KS0 * 8];
    auto request_x = [&](const Where& w) {
        const int64_t cs = (a.mode == 0) ? a.x_cs : 1;
        static_for<KS0>([&](auto ic) {
            constexpr int ks = decltype(ic)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int f = feat(ks, hh, j);
                raw[ks * 8 + j] = (w.valid && f < a.D) ? a.x[w.xoff + (int64_t)f * cs] : 0.0f;
            }
        });
    };
    auto request_emb = [&](int t) {  // proj accumulators <- emb[t] + proj bias (MODEL:614), straight from the table
        const float* er = a.embb + (int64_t)t * a.emb_ld;
        static_for<kDimT / 32>([&](auto ic) { acc[decltype(ic)::value] = load16(er + 32 * decltype(ic)::value + 4 * hh); });
    };

    Where cur = locate(blockIdx.x);
    int t_cur = request_t(cur);
    request_x(cur);
    request_emb(t_cur);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        stamp(-1);
        // ---- input fragments: bf16 of x (mode 0) or of x_t = a x + b noise_p (mode 1, GMD:302 / TF:86)
        bf16x8 XI[KS0];
        {
            const float2 ab = (a.mode == 1) ? scoef[cur.kk] : make_float2(1.0f, 0.0f);
            static_for<KS0>([&](auto ic) {
                constexpr int ks = decltype(ic)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = feat(ks, hh, j);
                    float v = raw[ks * 8 + j];
                    if (a.mode == 1) v = (cur.valid && f < a.D) ? ab.x * v + ab.y * snoise[cur.pp * a.D + f] : 0.0f;
                    XI[ks][j] = (__bf16)v;
                }
            });
        }
        const Where nxt = locate(tile + gridDim.x);
        const int t_nxt = request_t(nxt);
        const float* sb = sbias + kDimT;
        stamp(0);  // prologue
        // ---- proj (D -> 128)
        bf16x8 XP[kDimT / 16];  // proj output; not in X: the 128 -> H layer reads its last k-step while X is rewritten
        dense<KS0, kDimT / 32, 1, false, NTH>(p, w0, w1, XI, XP, acc, sb, hh);
        // ---- hidden layers (MODEL:300-301); activations are rewritten in place
        sb += H;
        stamp(1);  // proj
        dense<kDimT / 16, NTH, NTH*(kDimT / 16) / kChunkFrags, true, NTH>(p, w0, w1, XP, X, acc, sb, hh);
        stamp(2);  // 128 -> H
        for (int l = 1; l < a.n_hidden; ++l) {
            sb += H;
            dense<KSH, NTH, NTH * KSH / kChunkFrags, true, NTH>(p, w0, w1, X, X, acc, sb, hh);
        }
        stamp(3);  // H -> H layers
        // ---- head (H -> D), MODEL:303.  Only four accumulators are busy now: the next tile's record and its
        // ---- emb rows are requested here and land (in the free accumulators) while the head runs.
        request_x(nxt);
        request_emb(t_nxt);
        float sq = 0.0f;
        static_for<DT>([&](auto tc) {
            constexpr int nt = decltype(tc)::value;
            acc[HA] = load16(sb + 32 * nt + 4 * hh);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[HA + 1][r] = acc[HA + 2][r] = acc[HA + 3][r] = 0.0f;
            // four independent accumulation chains over k (a single chain would wait out the MFMA latency)
            consume_chunk(p, w0, w1, [&](auto ic, const bf16x8& w) {
                constexpr int ks = decltype(ic)::value;
                acc[HA + (ks & 3)] = MFMA(w, X[ks], acc[HA + (ks & 3)]);
            });
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = (acc[HA][r] + acc[HA + 1][r]) + (acc[HA + 2][r] + acc[HA + 3][r]);
            if (a.mode == 0) {
                if (cur.valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = 32 * nt + 8 * (r >> 2) + 4 * hh + (r & 3);
                        if (f < a.D) a.out[cur.row * a.o_rs + (int64_t)f * a.o_cs] = o[r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * nt + 8 * (r >> 2) + 4 * hh + (r & 3);
                    if (f < a.D) {
                        const float d = snoise[cur.pp * a.D + f] - o[r];
                        sq += d * d;
                    }
                }
            }
        });
        if (a.mode == 1) {  // TF:92-95: mean over the numeric features
            sq += __shfl_xor(sq, 32);
            if (cur.valid && hh == 0) a.score[cur.row] = sq / (float)a.D;
        }
        cur = nxt;
        t_cur = t_nxt;
        stamp(4);  // head + output
        if constexpr (STAMP) st_sum[5] += 1;
    }
    if constexpr (STAMP) {
        if (lane == 0 && wid == 0)
            for (int i = 0; i < 6; ++i) atomicAdd(a.stamps + i, st_sum[i]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last prefetches must land before the LDS is released
}

__global__ void k_add_bias_rows(float* dst, const float* src, const float* bias, int rows, int ld, int n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * ld; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld);
        dst[i] = src[i] + (c < n ? bias[c] : 0.0f);
    }
}

// order 0: k-step-major (fragment = ks * NT + nt), order 1: tile-major (fragment = nt * KS + ks)
__global__ void k_pack_frags(unsigned short* dst, const float* W, int out, int in, int NT, int KS, int order, int64_t frag0) {
    const int64_t n = (int64_t)NT * KS * 512;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t frag = e >> 9;
        const int lane = (int)((e >> 3) & 63), j = (int)(e & 7);
        const int nt = order ? (int)(frag / KS) : (int)(frag % NT), ks = order ? (int)(frag % KS) : (int)(frag / NT);
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
    a.embb = c->d_embb;
    a.n_noise = (a.mode == 1) ? a.P * c->D : 0;
    if (a.n_noise > kMaxNoise) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused Tartan-Federer kernel: P * Nn = %d exceeds %d", a.n_noise, kMaxNoise);
    a.emb_ld = c->emb_ld;
    a.n_hidden = L.n_hidden;
    a.total_chunks = L.total_chunks;
    a.D = c->D;
    static const int exp_env = getenv("MTK_FUSED_EXP") ? atoi(getenv("MTK_FUSED_EXP")) : 0;
    a.exp = exp_env;
    const int64_t tiles = (a.rows + 127) / 128;
    const int cus = n_cus(c->device);
    const int grid = (int)(tiles < cus ? tiles : cus);
    static const bool stamps_env = getenv("MTK_FUSED_STAMPS") != nullptr;
    if (stamps_env && L.DT == 1) {
        static unsigned long long* d_st = nullptr;
        if (!d_st) {
            MTK_HIP(hipMalloc(&d_st, 64));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kBufs * kChunkBytes + 32 * 1024);
        }
        MTK_HIP(hipMemsetAsync(d_st, 0, 64, s));
        a.stamps = d_st;
        const size_t lds2 = kBufs * kChunkBytes + (((size_t)(L.n_bias + L.H + a.n_noise + 2 * kMaxTF) * 4 + 15) & ~(size_t)15);
        hipLaunchKernelGGL((k_fused<512, 1, true>), dim3(grid), dim3(256), lds2, s, a);
        MTK_HIP(hipStreamSynchronize(s));
        unsigned long long h[8];
        MTK_HIP(hipMemcpy(h, d_st, 64, hipMemcpyDeviceToHost));
        const double n = (double)h[5] > 0 ? (double)h[5] : 1.0;
        fprintf(stderr, "[mtk stamps] tiles/wg-sum=%llu cycles/tile: prologue %.0f proj %.0f l1 %.0f hh %.0f head %.0f\n", h[5], h[0] / n, h[1] / n,
                h[2] / n, h[3] / n, h[4] / n);
        return MTK_OK;
    }
    const size_t lds = kBufs * kChunkBytes + (((size_t)(L.n_bias + L.H + a.n_noise + 2 * kMaxTF) * 4 + 15) & ~(size_t)15);
    return launch<512>(L, a, grid, lds, s);
}

}  // namespace

void mtk_free_fused(mtk_ctx* c) {
    if (c->d_wfrag) (void)hipFree(c->d_wfrag);
    if (c->d_bias_all) (void)hipFree(c->d_bias_all);
    if (c->d_embb) (void)hipFree(c->d_embb);
    c->d_embb = nullptr;
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
        const int max_lds = kBufs * kChunkBytes + 32 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fused<512, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        attr_set = true;
    }
    c->wfrag_bytes = (size_t)L.total_chunks * kChunkBytes;
    MTK_HIP(hipMalloc(&c->d_wfrag, c->wfrag_bytes));
    MTK_HIP(hipMalloc(&c->d_bias_all, sizeof(float) * L.n_bias));
    MTK_HIP(hipMalloc(&c->d_embb, sizeof(float) * (size_t)c->T * c->emb_ld));
    hipLaunchKernelGGL(k_add_bias_rows, dim3(256), dim3(256), 0, s, c->d_embb, c->d_emb, c->proj.b32, c->T, c->emb_ld, kDimT);
    MTK_HIP(hipMemsetAsync(c->d_wfrag, 0, c->wfrag_bytes, s));
    MTK_HIP(hipMemsetAsync(c->d_bias_all, 0, sizeof(float) * L.n_bias, s));
    unsigned short* dst = reinterpret_cast<unsigned short*>(c->d_wfrag);
    const int H = L.H, ksh = H / 16, nth = H / 32;
    int64_t chunk = 0;
    int boff = 0;
    auto pack = [&](const MlpLayer& l, int NT, int KS, int order, int chunks) {
        hipLaunchKernelGGL(k_pack_frags, dim3(512), dim3(256), 0, s, dst, l.w32, l.out_pad, l.in_pad, NT, KS, order, chunk * kChunkFrags);
        (void)hipMemcpyAsync(c->d_bias_all + boff, l.b32, sizeof(float) * (l.out < 32 * NT ? l.out : 32 * NT), hipMemcpyDeviceToDevice, s);
        chunk += chunks;
    };
    // note: w32 is the zero-padded [out_pad][in_pad] image, so out_pad/in_pad are its true extents
    pack(c->proj, kDimT / 32, 2 * L.DT, 0, 1);
    boff += kDimT;
    pack(c->blocks[0], nth, kDimT / 16, 0, nth * (kDimT / 16) / kChunkFrags);
    boff += H;
    for (int i = 1; i < c->n_hidden; ++i) {
        pack(c->blocks[i], nth, ksh, 0, nth * ksh / kChunkFrags);
        boff += H;
    }
    pack(c->head, L.DT, ksh, 1, (L.DT * ksh + kChunkFrags - 1) / kChunkFrags);
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
