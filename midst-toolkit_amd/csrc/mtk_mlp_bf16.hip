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
    const float* embb;  // [T][emb_ld] time-embedding table with the proj bias folded in (mode 1: the folded [T][H] table, emb_ld = H)
    int emb_ld;
    int n_noise;        // floats of the Tartan-Federer noise table staged in LDS (P * D)
    int n_hidden;
    int total_chunks;
    int64_t rows;
    int mode;  // 0: denoiser forward (per-row timesteps), 1: Tartan-Federer score, 2: denoiser forward at one timestep (sampling)
    int D;
    // mode 0
    const float* x; int64_t x_rs, x_cs;
    const int64_t* tvec; int t_scalar;
    int T;  // rows of embb: per-row timesteps are clamped into it (mtk_tclamp); the host poisons such rows afterwards
    float* out; int64_t o_rs, o_cs;
    // mode 1
    int64_t BP; int P; int64_t x_ld;
    int nk; int64_t tiles_per_k;  // timesteps of this launch; 128-row tiles per timestep (tiles never straddle one)
    const float* noise;
    float* score;
    const SchedRow* sched;
    int add_t;
    unsigned long long* stamps;  // diagnostic: per-phase cycle sums (MTK_FUSED_STAMPS=1), else null
    int exp;  // diagnostic (MTK_FUSED_EXP with MTK_FUSED_STAMPS): which ablation build of the stamp kernel runs
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
template <int EXP>  // EXP: compile-time ablation switches of the diagnostic (stamp) builds; 0 in the product kernels
struct Pipe {
    const unsigned char* src;  // wfrag (wave-uniform: lives in SGPRs)
    int total_chunks;
    int next;                  // chunk of the stream being requested
    unsigned rd, dma;          // LDS byte offsets: buffer being consumed, buffer being filled
    int lane, wid;
    unsigned rdv;              // rd + lane * 16: the per-lane LDS address of the chunk being consumed (see rotate)
    const unsigned char* gpc;  // this wave's share of the chunk being requested (+ 4096), set by piece 0
    static constexpr int kExp = EXP;  // 1 = no weight DMA after the prologue, 2 = no chunk barrier, 4 = no layer epilogues

    // piece K (0..7) of the chunk being requested: fragment 8 wid + K; 64 lanes x 16 B, dest = base + lane * 16.
    // Piece 0 sets up the request: M0 = LDS address of the wave's 8-KiB share of the buffer being filled (+ 4096), and a
    // scalar pointer to the same share of the chunk in the stream; pieces 1..7 are then ONE instruction each - the 13-bit
    // signed instruction offset of global_load_lds applies to the global AND the LDS address, so (K - 4) KiB walks both.
    // (Before: per piece an s_add for the destination, the M0 write and its wait state, and a 64-bit scalar add for the
    // source - 128 pieces per hidden layer.)  M0 has to keep its value between the pieces of a request: the compiler
    // puts nothing in M0 in this kernel (no LDS-direct, GWS, s_movrel or message instruction; checked in the ISA - the
    // only M0 accesses are these and dma16's, which saves and restores it).
    template <int K>
    __device__ __forceinline__ void piece() {
        if constexpr (!(EXP & 1)) {
            const unsigned voff = (unsigned)lane * 16u;
            if constexpr (K == 0) {
                const unsigned share = (unsigned)wid * (8u * (unsigned)kFragBytes) + 4096u;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(dma + share);
                gpc = src + (size_t)next * kChunkBytes + share;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:-4096" : : "v"(voff), "s"(m0v), "s"(gpc) : "memory");
            } else {
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(gpc), "n"((K - 4) * 1024) : "memory");
            }
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
        if constexpr (!(EXP & 2)) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    // The read base is re-formed behind an opaque copy at every chunk: a hidden layer is 16 chunks = four full turns of the
    // buffer ring, so `rd` is invariant over the layer loop; seeing that, the compiler folded the ring into ~60 distinct
    // per-lane addresses, hoisted them out of the loop, parked them in AGPRs and paid a v_accvgpr_read per ds_read (mode 2,
    // DT = 4: +270 vector instructions per layer, hidden layers 59.2 k instead of 56.7 k cycles per tile).
    __device__ __forceinline__ void rotate() {
        rd = (rd + kChunkBytes == kBufs * kChunkBytes) ? 0u : rd + kChunkBytes;
        rdv = rd + (unsigned)lane * 16u;
        asm volatile("" : "+v"(rdv));
    }
    __device__ __forceinline__ bf16x8 ld(int f) const {
        return *reinterpret_cast<const bf16x8*>(mtk_smem_f + rdv + f * kFragBytes);
    }
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)

// Consume the current chunk (32 fragments) in eight groups of kGrp = 4.  `step(ic<I>, w)` issues the MFMA of
// fragment I.  Fragment i of the NEXT group is read from LDS right before the MFMA of fragment i of the current
// one, so every ds_read_b128 has four MFMAs (128 cycles) to return and only 2 x 4 fragment registers are live;
// the issue order is pinned (the scheduler otherwise sinks the reads next to their uses).  One LDS-DMA piece is
// issued per group.  On entry w0 holds fragments 0..3 of the chunk; on exit w0 holds fragments 0..3 of the next.
constexpr int kGrp = 4;
template <int G, class P, class Step>
__device__ __forceinline__ void group(P& p, bf16x8 (&cur)[kGrp], bf16x8 (&nxt)[kGrp], Step&& step) {
    static_for<kGrp>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        nxt[i] = p.ld(((G + 1) & 7) * kGrp + i);
        step(std::integral_constant<int, G * kGrp + i>{}, cur[i]);
        if constexpr (i == 1) p.template piece<(G + 2) & 7>();
        PIN();
    });
}
template <class P, class Step>
__device__ __forceinline__ void consume_chunk(P& p, bf16x8 (&w0)[kGrp], bf16x8 (&w1)[kGrp], Step&& step) {
    group<0>(p, w0, w1, step);  // piece 2
    group<1>(p, w1, w0, step);
    group<2>(p, w0, w1, step);
    group<3>(p, w1, w0, step);
    group<4>(p, w0, w1, step);
    group<5>(p, w1, w0, step);  // piece 7 -> the request moves on to the next chunk
    p.sync();                   // chunk m+1 complete everywhere; chunk m-1's buffer is free for the next request
    PIN();
    group<6>(p, w0, w1, step);  // piece 0 of the new request
    p.rotate();                 // the last group prefetches fragments 0..3 of the next chunk
    group<7>(p, w1, w0, step);  // piece 1
}

// LDS rows (biases, staged emb row) are read through ONE opaque 32-bit LDS address per layer plus immediate offsets.
// Left to itself the compiler materialises a separate address register per 32-feature tile, hoists all of them
// out of the tile loop and spills them; each reload then waits on vmcnt(0), i.e. on every weight DMA in flight.
typedef __attribute__((address_space(3))) const float lds_f;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4_t lds_f4;
// The row pointer is uniform; the lane's share (16 bytes per lane half) goes through an opaque copy and is added
// AFTER it, so the sum is re-formed at every use (one v_add) instead of being hoisted out of the tile loop as a
// loop-invariant register, which at 512 registers got spilled and reloaded behind a vmcnt(0).
__device__ __forceinline__ unsigned lds_base(const float* p_uniform, int hh) {
    const unsigned off = (unsigned)(uintptr_t)(lds_f*)p_uniform;
    unsigned lane_part = 16u * (unsigned)hh;
    asm volatile("" : "+v"(lane_part));
    return off + lane_part;
}
__device__ __forceinline__ f32x16 load16_lds(unsigned base, int elem) {  // base + 4 (elem + {0..3, 8..11, 16..19, 24..27})
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4_t t = *(lds_f4*)(uintptr_t)(base + 4u * (unsigned)(elem + 8 * q));
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
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
template <int KS, int NT, int CHUNKS, bool RELU, int NT_NEXT, class PIPE, int NIN, int NOUT, int NACC>
__device__ __forceinline__ void dense(PIPE& p, bf16x8 (&w0)[kGrp], bf16x8 (&w1)[kGrp], const bf16x8 (&Xin)[NIN], bf16x8 (&Xout)[NOUT],
                                      f32x16 (&acc)[NACC], const float* next_bias, int hh) {
    const unsigned nb = lds_base(next_bias, hh);
    // epilogue of one finished tile: ReLU, cvt to bf16 into the activation array, accumulator <- next bias
    auto finish = [&](auto tc) {
        constexpr int nt = decltype(tc)::value;
        pack_out<RELU>(acc[nt], Xout[2 * nt], Xout[2 * nt + 1]);
        if constexpr (nt < NT_NEXT) acc[nt] = load16_lds(nb, 32 * nt);
    };
    static_for<CHUNKS>([&](auto cc) {
        consume_chunk(p, w0, w1, [&](auto ic, const bf16x8& w) {
            constexpr int g = decltype(cc)::value * kChunkFrags + decltype(ic)::value, ks = g / NT, nt = g % NT;
            if constexpr (ks < KS) acc[nt] = MFMA(w, Xin[ks], acc[nt]);
            if constexpr (ks == KS - 1) {  // last k-step: finish the tile issued two MFMAs ago
                if constexpr (!(PIPE::kExp & 4) && nt >= 2) finish(std::integral_constant<int, nt - 2>{});
            }
        });
    });
    if constexpr (!(PIPE::kExp & 4)) {
        if constexpr (NT >= 2) finish(std::integral_constant<int, NT - 2>{});
        PIN();
        finish(std::integral_constant<int, NT - 1>{});
        PIN();
    }
    static_for<NT_NEXT>([&](auto ic) {  // one tile at a time: 16 LDS reads, 16 accumulator writes, not 16 x that in flight
        constexpr int nt = decltype(ic)::value;
        if constexpr (nt >= NT) {
            acc[nt] = load16_lds(nb, 32 * nt);
            PIN();
        }
    });
}

// ---- Tile-major layers (128 -> H and H -> H).  Fragment g of the layer multiplies k-step g % KS into tile g / KS: ONE
// accumulator tile is busy at a time (a dependent chain of 32x32x16 MFMAs issues back to back), and two of them alternate.
// While tile nt accumulates, the epilogue of tile nt - 1 (ReLU, cvt to bf16) rides in its MFMA gaps, a pair of values at a
// time, followed by the bias of tile nt + 1 into the accumulator that epilogue just freed.  Nothing of a layer's
// epilogue is exposed: the last tile's is still pending when the layer ends (PENDING) and rides under the first steps
// of whatever comes next (the next layer's tile 0 or the head).
//   Activations ping-pong without a second layer body: a layer reads its input from Xb and writes its output there.
// Tile nt's output replaces k-steps 2nt, 2nt+1, which later tiles still need, so tile nt copies those two k-steps to Xa
// (in its gaps) before tile nt + 1 overwrites them, and a tile reads k-steps < 2 nt from Xa, the others from Xb.
template <int KS>
__device__ __forceinline__ constexpr int epi_first(int ks) {  // first epilogue part (of 8) issued after the MFMA of step ks
    if (KS == 32) {
        constexpr int tab[32] = {-1, -1, 0, 1, 2, -1, 3, 4, 5, -1, 6, 7, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
        return tab[ks];
    }
    if (KS == 8) return (ks >= 2 && ks <= 5) ? 2 * (ks - 2) : -1;  // two parts per step
    return ks * 8 / KS;  // KS = 2, 4, 6 (the folded first layer, D <= 32 KS / 2 inputs): the parts spread over every step
}
template <int KS>
__device__ __forceinline__ constexpr int epi_count(int ks) {
    if (KS == 32) return epi_first<KS>(ks) < 0 ? 0 : 1;
    if (KS == 8) return epi_first<KS>(ks) < 0 ? 0 : 2;
    return (ks + 1) * 8 / KS - ks * 8 / KS;
}
template <int KS>
__device__ __forceinline__ constexpr int bias_first(int ks) {  // first 16-byte piece (of 4) of the next tile's bias
    if (KS == 32) return ks == 15 ? 0 : ks == 16 ? 1 : ks == 18 ? 2 : ks == 19 ? 3 : -1;
    if (KS == 8) return ks == 6 ? 0 : -1;
    return ks == KS - 1 ? 0 : -1;  // after the last parts of the previous tile's epilogue
}
template <int KS>
__device__ __forceinline__ constexpr int bias_count(int ks) { return bias_first<KS>(ks) < 0 ? 0 : (KS == 32 ? 1 : 4); }

// epilogue part q of a finished tile held in `a`: values 2q, 2q+1 -> one packed bf16 pair (one dword) of the tile's two
// k-steps.  The conversion is volatile asm: as plain C++ the compiler parks the ReLU'd fp32 values in spare registers and
// sinks all of a layer's conversions to the layer's end (120 v_cvt_pk in a row, exposed).  ReLU on the packed pair: a
// negative bf16 is a negative 16-bit integer.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int Q>
__device__ __forceinline__ void epi_part(const f32x16& a, u32x4& lo, u32x4& hi) {
    unsigned pk;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(pk) : "v"(a[2 * Q]), "v"(a[2 * Q + 1]));
    if constexpr (Q < 4) lo[Q] = pk;
    else hi[Q - 4] = pk;
}
template <int R>
__device__ __forceinline__ void bias_piece(f32x16& a, unsigned base, int elem) {
    const f32x4_t t = *(lds_f4*)(uintptr_t)(base + 4u * (unsigned)(elem + 8 * R));
    a[4 * R] = t.x; a[4 * R + 1] = t.y; a[4 * R + 2] = t.z; a[4 * R + 3] = t.w;
}

// the pending epilogue of tile PT (held in `a`), as far as step ks of a KS-step tile carries it
template <int KS, int KSTEP, int PT, int NX>
__device__ __forceinline__ void epi_at(const f32x16& a, u32x4 (&Xb)[NX]) {
    constexpr int q0 = epi_first<KS>(KSTEP), n = epi_count<KS>(KSTEP);
    static_for<n>([&](auto qc) { epi_part<q0 + decltype(qc)::value>(a, Xb[2 * PT], Xb[2 * PT + 1]); });
}

template <int KS, int NT, int CHUNKS, bool PENDING, bool COPY, class PIPE, int NIN, int NX>
__device__ __forceinline__ void dense_tm(PIPE& p, bf16x8 (&w0)[kGrp], bf16x8 (&w1)[kGrp], const bf16x8 (&Xin)[NIN], u32x4 (&Xa)[NX],
                                         u32x4 (&Xb)[NX], f32x16 (&acc)[2], const float* bias_this, const float* bias_next, int hh) {
    static_assert(NT % 2 == 0 && NX == 2 * NT && CHUNKS * kChunkFrags == NT * KS, "tile-major layer shape");
    const unsigned nb_this = lds_base(bias_this, hh), nb_next = lds_base(bias_next, hh);
    static_for<CHUNKS>([&](auto cc) {
        consume_chunk(p, w0, w1, [&](auto ic, const bf16x8& w) {
            constexpr int g = decltype(cc)::value * kChunkFrags + decltype(ic)::value, nt = g / KS, ks = g % KS, b = nt & 1;
            if constexpr (!COPY) acc[b] = MFMA(w, Xin[ks], acc[b]);
            else if constexpr (ks < 2 * nt) acc[b] = MFMA(w, __builtin_bit_cast(bf16x8, Xa[ks]), acc[b]);
            else acc[b] = MFMA(w, __builtin_bit_cast(bf16x8, Xb[ks]), acc[b]);
            if constexpr (PIPE::kExp & 4) return;  // diagnostic build only: no epilogue, no bias
            if constexpr (nt > 0 || PENDING) epi_at<KS, ks, (nt > 0 ? nt - 1 : NT - 1)>(acc[b ^ 1], Xb);
            if constexpr (COPY && nt < NT - 1) {
                if constexpr (ks == 12) Xa[2 * nt] = Xb[2 * nt];
                if constexpr (ks == 14) Xa[2 * nt + 1] = Xb[2 * nt + 1];
            }
            constexpr int r0 = bias_first<KS>(ks), rn = bias_count<KS>(ks);
            static_for<rn>([&](auto rc) {
                if constexpr (nt < NT - 1) bias_piece<r0 + decltype(rc)::value>(acc[b ^ 1], nb_this, 32 * (nt + 1));
                else bias_piece<r0 + decltype(rc)::value>(acc[b ^ 1], nb_next, 0);
            });
        });
    });
}

// One 1-KiB LDS-DMA transfer outside the weight pipeline (next tile's records / time-embedding row): 64 lanes x 16 B
// from `gp` (per lane) to LDS byte offset `dst` + lane * 16.  Counted by vmcnt in issue order like the weight pieces.
__device__ __forceinline__ void dma16(const void* gp, unsigned dst) {
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gp), "s"(d)
                 : "memory");
}

// MODE 0: denoiser forward (per-lane global loads of the record and of emb[t]).
// MODE 1: Tartan-Federer score.  Tiles never straddle a timestep, so t is tile-uniform: the tile's records
//         (a contiguous block of x, <= 128 x x_ld floats) and its emb[t] + bias row are brought into LDS by DMA during
//         the previous tile's last hidden layer — no per-tile global load ever lands in a VGPR.
// MODE 2: denoiser forward with ONE timestep for the whole launch (every step of the reverse chain): records in and
//         outputs out as in mode 0, layers as in mode 1 - proj folded into the first hidden layer, whose bias row
//         (row t of the folded [T][H] table) is copied into LDS once per launch.
template <int H, int DT, int MODE, bool STAMP = false, int EXP = 0>
__global__ __launch_bounds__(256, 1) void k_fused(FusedArgs a) {
    constexpr int KS0 = 2 * DT;   // input k-steps (D padded to 32 DT)
    constexpr int KSH = H / 16;   // k-steps of a hidden activation
    constexpr int NTH = H / 32;   // tiles of a hidden layer
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS after the chunk buffers: biases (+H zeros), the Tartan-Federer noise table, per-timestep (a, b),
    // the tile's emb row and the tile's records (MODE 1)
    float* sbias = reinterpret_cast<float*>(mtk_smem_f + kBufs * kChunkBytes);
    float* snoise = sbias + a.n_bias + H;
    float2* scoef = reinterpret_cast<float2*>(snoise + a.n_noise);
    float* semb = reinterpret_cast<float*>(scoef + kMaxTF);
    float* sx = semb + (MODE != 0 ? H : kDimT);  // modes 1, 2 hold the folded [H] row (see the proj fold below)
    for (int i = tid; i < a.n_bias + H; i += 256) sbias[i] = i < a.n_bias ? a.bias[i] : 0.0f;
    if (MODE == 2)
        for (int i = tid; i < H; i += 256) semb[i] = a.embb[(int64_t)a.t_scalar * a.emb_ld + i];
    if (MODE == 1) {
        for (int i = tid; i < a.n_noise; i += 256) snoise[i] = a.noise[i];
        for (int i = tid; i < kMaxTF; i += 256) {
            const SchedRow sr = a.sched[a.t_list[i] + a.add_t];
            scoef[i] = make_float2(sr.sqrt_acp, sr.sqrt_1m_acp);
        }
    }

    Pipe<EXP> p{a.wfrag, a.total_chunks, 0, 0u, 0u, lane, wid, (unsigned)lane * 16u, a.wfrag};
    p.fetch_all();  // chunk 0 -> buffer 0
    p.fetch_all();  // chunk 1 -> buffer 1
    p.template piece<0>();  // chunk 2 -> buffer 2, the rest of it is requested while chunk 0 is consumed
    p.template piece<1>();

    const int64_t n_tiles = (MODE == 1) ? a.tiles_per_k * a.nk : (a.rows + 127) / 128;
    // MODE 1: DMA of tile `tile`'s records and emb row into LDS (all waves take part)
    auto stage = [&](int64_t tile) {
        if (MODE != 1 || tile >= n_tiles) return;
        const unsigned tpk = (unsigned)a.tiles_per_k, P = (unsigned)a.P, BP = (unsigned)a.BP;
        const unsigned k = (unsigned)tile / tpk;
        const unsigned r0 = ((unsigned)tile - k * tpk) * 128u;
        const unsigned r1 = (r0 + 127u < BP - 1u) ? r0 + 127u : BP - 1u;
        const int64_t rec0 = (P == 1) ? r0 : r0 / P;
        const int bytes = (int)((((P == 1) ? r1 : r1 / P) - rec0 + 1) * a.x_ld * 4);  // multiple of 16: x_ld % 4 == 0
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.x + rec0 * a.x_ld);
        const unsigned sx_off = (unsigned)(reinterpret_cast<unsigned char*>(sx) - mtk_smem_f);
        int l16 = lane * 16;  // opaque per call: `src + lane * 16` is otherwise hoisted as a 64-bit per-lane pointer and spilled
        asm volatile("" : "+v"(l16));
        for (int q = wid; q * 1024 < bytes; q += 4)
            if (q * 1024 + l16 < bytes) dma16(src + q * 1024 + l16, sx_off + q * 1024);
        if (wid >= 4 - H / 256) {  // the folded emb row of the tile's timestep: H floats, one 1-KiB piece per wave
            const int t = a.t_list[k], piece = wid - (4 - H / 256);
            dma16(reinterpret_cast<const unsigned char*>(a.embb + (int64_t)t * a.emb_ld) + piece * 1024 + lane * 16,
                  (unsigned)(reinterpret_cast<unsigned char*>(semb) - mtk_smem_f) + piece * 1024);
        }
    };
    stage(blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunks 0, 1, the first tile's inputs and the LDS tables visible to every wave
    bf16x8 w0[kGrp], w1[kGrp];
    static_for<kGrp>([&](auto ic) { w0[decltype(ic)::value] = p.ld(decltype(ic)::value); });

    f32x16 acc[2];               // the two alternating accumulator tiles of the tile-major layers
    f32x16 pacc[kDimT / 32];     // proj accumulators (emb[t] + proj bias, then D -> 128)
    u32x4 Xa[KSH], Xb[KSH];     // activations as packed bf16 pairs, see dense_tm
    if constexpr (EXP & 4) {  // ablation without epilogues: the activations are never written, give them defined (opaque) values
#pragma unroll
        for (int i = 0; i < KSH; ++i) asm volatile("" : "=v"(Xa[i]), "=v"(Xb[i]));
    }
    unsigned long long st_prev = 0, st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int slot) {
        PIN();  // phase boundary: keeps the scheduler from mixing the phases' register peaks (fewer spills)
        if constexpr (STAMP) {
            PIN();
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            PIN();
            if (slot >= 0) st_sum[slot] += tnow - st_prev;
            st_prev = tnow;
        }
    };

    // ---- where a tile's lane sits
    struct Where { int64_t row; bool valid; int64_t xoff; int kk, pp, rl; };
    auto locate = [&](int64_t tile) {
        Where w;
        w.kk = 0; w.pp = 0; w.rl = 0; w.xoff = 0;
        if (MODE != 1) {
            // the lane's row inside the tile goes through an opaque copy: left visible, its 64-bit image, the address of
            // its timestep and its share of the emb row address are loop invariants that get hoisted and spilled
            int lrow = wid * 32 + li;
            asm volatile("" : "+v"(lrow));
            w.row = tile * 128 + lrow;
            w.valid = tile < n_tiles && w.row < a.rows;
            w.xoff = w.row * a.x_rs;
        } else {  // 32-bit arithmetic (the host checks BP * P and the tile count fit): 64-bit divisions cost hundreds of cycles
            const unsigned tpk = (unsigned)a.tiles_per_k, P = (unsigned)a.P, BP = (unsigned)a.BP;
            const unsigned k = (unsigned)tile / tpk;
            const unsigned r0 = ((unsigned)tile - k * tpk) * 128u;
            const unsigned rr = r0 + (unsigned)(wid * 32 + li);
            w.valid = tile < n_tiles && rr < BP;
            const unsigned rec = (P == 1) ? rr : rr / P;
            w.kk = (int)k;
            w.pp = (int)(rr - rec * P);
            w.rl = (int)(rec - ((P == 1) ? r0 : r0 / P));  // record index inside the staged block
            w.row = (int64_t)k * a.BP + rr;
        }
        return w;
    };
    // ---- MODE 0: per-tile inputs are requested one tile ahead (a single wave per SIMD has nothing else to hide a
    // ---- global-memory round trip behind): the timestep first, then the record and the emb[t] + bias rows
    auto request_t = [&](const Where& w) -> int {
        return a.tvec ? (w.valid ? mtk_tclamp(a.tvec[w.row], a.T) : 0) : a.t_scalar;
    };
    float raw[MODE != 1 ? KS0 * 8 : 1];
    auto request_x = [&](const Where& w) {
        if constexpr (MODE != 1) {
            // Branch-free (every load is issued, clamped to a valid address, and masked afterwards) and with the lane
            // half, D and the column stride made opaque per tile: left visible, the 8 KS0 clamped feature indices,
            // their 64-bit offsets and their `f < D` masks are loop invariants that get hoisted out of the tile loop and
            // spilled (VGPRs and SGPRs), and each reload waits on every weight DMA in flight.
            int hq = hh, Dq = a.D;
            int64_t xcs = a.x_cs;
            asm volatile("" : "+v"(hq), "+s"(Dq), "+s"(xcs));
            // Address = [per-lane pointer: the lane's row and its half's first feature, formed once per tile] + [feature * column
            // stride: uniform, formed on the scalar unit]: a 64-bit add per load.  The per-load 64-bit multiply of a per-lane
            // feature index it replaces cost eight vector instructions per load (1 k per tile, in front of the exposed round trip).
            const float* row0 = a.x + (w.valid ? w.xoff : 0);
            const float* pl = row0 + (int64_t)(4 * hq) * xcs;
            static_for<KS0>([&](auto ic) {
                constexpr int ks = decltype(ic)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cf = feat(ks, 0, j);  // the half-independent part of the feature index (compile time)
                    const float* p = pl + (int64_t)cf * xcs;
                    if constexpr ((ks >> 1) == DT - 1) {
                        // the last 32-feature tile can reach past D: such lanes read the row's first feature (masked below)
                        if (cf + 4 * hq >= Dq) p = row0;
                    }
                    raw[ks * 8 + j] = *p;
                }
            });
            asm volatile("" ::: "memory");  // the loads stay above this line, unconditional; the masks below cannot pull them down
            static_for<KS0>([&](auto ic) {
                constexpr int ks = decltype(ic)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = feat(ks, hq, j);
                    raw[ks * 8 + j] = (w.valid && f < Dq) ? raw[ks * 8 + j] : 0.0f;
                }
            });
        }
    };
    auto request_emb = [&](int t) {  // proj accumulators <- emb[t] + proj bias (MODE:614), straight from the table
        int h4 = 4 * hh;
        asm volatile("" : "+v"(h4));
        const float* er = a.embb + (int64_t)t * a.emb_ld + h4;
        static_for<kDimT / 32>([&](auto ic) {
            pacc[decltype(ic)::value] = load16(er + 32 * decltype(ic)::value);
            PIN();
        });
    };

    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (STAMP) {
        clk0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    Where cur = locate(blockIdx.x);
    int t_cur = 0;
    if constexpr (MODE == 0) {
        t_cur = request_t(cur);
    }
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        stamp(-1);
        // ---- input fragments: bf16 of x (MODE 0) or of x_t = a x + b noise_p (MODE 1, GMD:302 / TF:86)
        bf16x8 XI[KS0];
        if constexpr (MODE != 1) {
            // the record and its emb[t] + bias rows are loaded HERE, not a tile ahead: only the timestep (one register) is
            // requested a tile ahead.  One exposed global round trip per 128-row tile (≈7 k of a tile's 85 k cycles at
            // D = 120).  Requesting them before the previous tile's head instead was built and measured: the round trip
            // left the prologue, and the same cycles reappeared in the hidden layers and the head (128 more live registers
            // changed their allocation; the compiler's vmcnt wait for them drains the weight DMA either way) - no gain.
            request_x(cur);
            if constexpr (MODE == 0) request_emb(t_cur);
            static_for<KS0>([&](auto ic) {
                constexpr int ks = decltype(ic)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) XI[ks][j] = (__bf16)raw[ks * 8 + j];
            });
        } else {
            const float2 ab = scoef[cur.kk];
            int hq = hh, Dq = a.D;  // opaque per tile, see request_x
            asm volatile("" : "+v"(hq), "+s"(Dq));
            const float* xr = sx + cur.rl * (int)a.x_ld;
            const float* nr = snoise + cur.pp * Dq;
            static_for<KS0>([&](auto ic) {
                constexpr int ks = decltype(ic)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int f = feat(ks, hq, j);
                    const float v = (cur.valid && f < Dq) ? ab.x * xr[f] + ab.y * nr[f] : 0.0f;
                    XI[ks][j] = (__bf16)v;
                }
            });
        }
        const Where nxt = locate(tile + gridDim.x);
        int t_nxt = 0;
        if constexpr (MODE == 0) t_nxt = request_t(nxt);
        const float* sb = sbias + kDimT;
        stamp(0);  // prologue
        bf16x8 XP[kDimT / 16];  // proj output (forward mode); not in X: the 128 -> H layer reads all of it for every tile
        if constexpr (MODE != 0) {
            // ---- proj folded into the first hidden layer: ReLU((W1 Wp) x + [W1 (emb[t] + b_proj) + b1]).  No nonlinearity sits
            // ---- between the two layers, the tile's timestep is uniform, so the bracket is one staged row of a [T][H] table and
            // ---- the layer is D -> H: 16 tiles of 2 DT MFMAs instead of 8 + 128 (same function; one bf16 rounding fewer)
            acc[0] = load16_lds(lds_base(semb, hh), 0);
            stamp(1);
            if (MODE == 1 && a.n_hidden == 1) stage(tile + gridDim.x);
            dense_tm<KS0, NTH, NTH * KS0 / kChunkFrags, false, false>(p, w0, w1, XI, Xa, Xb, acc, semb, sb + H, hh);
        } else {
        // ---- proj (D -> 128): four tiles of 2 DT k-steps, k-step-major, its (short) epilogue exposed
        dense<KS0, kDimT / 32, 1, false, 0>(p, w0, w1, XI, XP, pacc, sb, hh);
        acc[0] = load16_lds(lds_base(sb, hh), 0);  // bias of the 128 -> H layer's tile 0; every later one rides in the stream
        stamp(1);  // proj
        // ---- 128 -> H, then the H -> H layers (MODEL:300-301), tile-major (dense_tm)
        dense_tm<kDimT / 16, NTH, NTH*(kDimT / 16) / kChunkFrags, false, false>(p, w0, w1, XP, Xa, Xb, acc, sb, sb + H, hh);
        }
        stamp(2);  // 128 -> H
        for (int l = 1; l < a.n_hidden; ++l) {
            sb += H;
            if (l == a.n_hidden - 1) stage(tile + gridDim.x);  // lands (and is fenced by the chunk barriers) before the head ends
            dense_tm<KSH, NTH, NTH * KSH / kChunkFrags, true, true>(p, w0, w1, XP, Xa, Xb, acc, sb, sb + H, hh);
        }
        sb += H;
        stamp(3);  // H -> H layers
        // ---- head (H -> D), MODEL:303: one accumulation chain per 32 output features; the last hidden tile's epilogue
        // ---- (k-steps KSH-2, KSH-1 of the head's input) rides under the first steps of the head's first tile.  The output
        // ---- stores of a head tile stay exposed after its chunk: issued one per MFMA gap of the next tile instead (each an
        // ---- exec-masked store with its own 64-bit address) they took the head from 9.2 k to 14.6 k cycles per tile.
        float sq = 0.0f;
        static_for<DT>([&](auto tc) {
            constexpr int nt = decltype(tc)::value, hb = nt & 1;
            if constexpr (nt > 0) acc[hb] = load16_lds(lds_base(sb, hh), 32 * nt);
            consume_chunk(p, w0, w1, [&](auto ic, const bf16x8& w) {
                constexpr int ks = decltype(ic)::value;
                acc[hb] = MFMA(w, __builtin_bit_cast(bf16x8, Xb[ks]), acc[hb]);
                if constexpr (nt == 0) epi_at<KSH, ks, NTH - 1>(acc[1], Xb);
            });
            const f32x16 o = acc[hb];
            if constexpr (MODE != 1) {
                if (cur.valid) {
                    int64_t ocs = a.o_cs;  // opaque for the same reason as in request_x
                    int hq = hh, Dq = a.D;
                    asm volatile("" : "+s"(ocs), "+v"(hq), "+s"(Dq));
                    float* ol = a.out + cur.row * a.o_rs + (int64_t)(4 * hq) * ocs;  // per lane, once per tile; + uniform feature offsets
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cf = 32 * nt + 8 * (r >> 2) + (r & 3);
                        // only the last tile can reach past D = the number of outputs (DT = ceil(D / 32))
                        if (nt < DT - 1 || cf + 4 * hq < Dq) ol[(int64_t)cf * ocs] = o[r];
                    }
                }
            } else {
                int hq = hh, Dq = a.D;
                asm volatile("" : "+v"(hq), "+s"(Dq));
                const float* nr = snoise + cur.pp * Dq;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = 32 * nt + 8 * (r >> 2) + 4 * hq + (r & 3);
                    if (f < Dq) {
                        const float d = nr[f] - o[r];
                        sq += d * d;
                    }
                }
            }
        });
        if constexpr (MODE == 1) {  // TF:92-95: mean over the numeric features
            sq += __shfl_xor(sq, 32);
            if (cur.valid && hh == 0) a.score[cur.row] = sq / (float)a.D;
        }
        cur = nxt;
        t_cur = t_nxt;
        stamp(4);  // head + output
        if constexpr (STAMP) st_sum[5] += 1;
    }
    (void)t_cur;
    if constexpr (STAMP) {
        st_sum[6] = __builtin_amdgcn_s_memtime() - clk0;      // shader cycles ...
        st_sum[7] = __builtin_amdgcn_s_memrealtime() - rt0;   // ... per 100 MHz tick: the clock held under this load
        if (lane == 0 && wid == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(a.stamps + i, st_sum[i]);
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

// proj folded into the first hidden layer (Tartan-Federer mode): Wf[h][d] = sum_k W1[h][k] Wp[k][d] on the zero-padded fp32 images
__global__ void k_fold_w(float* Wf, int ldf, const float* W1, int ld1, const float* Wp, int ldp, int H, int dim_t, int D) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * ldf; i += gridDim.x * blockDim.x) {
        const int h = i / ldf, d = i - h * ldf;
        float acc = 0.0f;
        if (d < D)
            for (int k = 0; k < dim_t; ++k) acc += W1[(int64_t)h * ld1 + k] * Wp[(int64_t)k * ldp + d];
        Wf[i] = acc;
    }
}
// embf[t][h] = sum_k W1[h][k] (emb[t][k] + b_proj[k]) + b1[h]
__global__ void k_fold_emb(float* embf, const float* W1, int ld1, const float* embb, int emb_ld, const float* b1, int T, int H, int dim_t) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)T * H; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / H), h = (int)(i - (int64_t)t * H);
        float acc = b1[h];
        for (int k = 0; k < dim_t; ++k) acc += W1[(int64_t)h * ld1 + k] * embb[(int64_t)t * emb_ld + k];
        embf[i] = acc;
    }
}

struct Layout {
    int H, DT, n_hidden, total_chunks, n_bias;
    int tf_chunks, tail_chunk0;  // folded stream: its length; first chunk of the unfolded stream it shares (hidden layers, head)
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
    L.tail_chunk0 = 1 + c_l1;
    L.tf_chunks = nth * 2 * L.DT / kChunkFrags + (c->n_hidden - 1) * c_hh + c_head;  // folded D -> H layer: nth tiles x 2 DT k-steps
    L.n_bias = kDimT + c->n_hidden * H + 32 * L.DT;
    return true;
}

constexpr size_t kMaxLds = 160 * 1024;

template <int H, int MODE>
int launch(const Layout& L, const FusedArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (L.DT) {
        case 1: hipLaunchKernelGGL((k_fused<H, 1, MODE>), dim3(grid), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((k_fused<H, 2, MODE>), dim3(grid), dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL((k_fused<H, 3, MODE>), dim3(grid), dim3(256), lds, s, a); break;
        case 4: hipLaunchKernelGGL((k_fused<H, 4, MODE>), dim3(grid), dim3(256), lds, s, a); break;
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

// bytes of LDS a launch needs: chunk buffers, biases (+H zeros), noise table, (a, b) per timestep, emb row, records
size_t lds_need(const Layout& L, const FusedArgs& a) {
    const size_t floats = (size_t)L.n_bias + L.H + a.n_noise + 2 * kMaxTF + (a.mode != 0 ? L.H : kDimT) + (a.mode == 1 ? 128 * (size_t)a.x_ld : 0);
    return kBufs * kChunkBytes + ((floats * 4 + 15) & ~(size_t)15);
}

int run(mtk_ctx* c, FusedArgs& a, hipStream_t s) {
    Layout L;
    if (!layout_for(c, L)) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused denoiser: unsupported shape");
    a.wfrag = reinterpret_cast<const unsigned char*>(a.mode != 0 ? c->d_wfrag_tf : c->d_wfrag);
    a.bias = c->d_bias_all;
    a.n_bias = L.n_bias;
    a.embb = a.mode != 0 ? c->d_embf : c->d_embb;
    a.n_noise = (a.mode == 1) ? a.P * c->D : 0;
    a.emb_ld = a.mode != 0 ? L.H : c->emb_ld;
    a.n_hidden = L.n_hidden;
    a.total_chunks = a.mode != 0 ? L.tf_chunks : L.total_chunks;
    a.D = c->D;
    static const int exp_env = getenv("MTK_FUSED_EXP") ? atoi(getenv("MTK_FUSED_EXP")) : 0;
    a.exp = exp_env;
    const int64_t tiles = (a.mode == 1) ? a.tiles_per_k * a.nk : (a.rows + 127) / 128;
    const int cus = n_cus(c->device);
    const int grid = (int)(tiles < cus ? tiles : cus);
    const size_t lds = lds_need(L, a);
    if (lds > kMaxLds) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused kernel: %zu bytes of LDS needed, %zu available", lds, kMaxLds);
    static const bool stamps_env = getenv("MTK_FUSED_STAMPS") != nullptr;
    if (stamps_env && ((L.DT == 1 && a.mode == 1) || (L.DT == 4 && a.mode != 1))) {
        static unsigned long long* d_st = nullptr;
        using Kern = void (*)(FusedArgs);
        static const Kern kerns[6] = {k_fused<512, 1, 1, true, 0>, k_fused<512, 1, 1, true, 1>, k_fused<512, 1, 1, true, 2>, k_fused<512, 1, 1, true, 4>,
                                      k_fused<512, 4, 0, true, 0>, k_fused<512, 4, 2, true, 0>};  // the last two: forward modes, shape M (mixed_loss; sampling)
        const Kern kern = a.mode == 0 ? kerns[4] : a.mode == 2 ? kerns[5] : kerns[a.exp == 1 ? 1 : a.exp == 2 ? 2 : a.exp == 4 ? 3 : 0];
        if (!d_st) {
            MTK_HIP(hipMalloc(&d_st, 64));
            for (Kern k : kerns) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
        }
        MTK_HIP(hipMemsetAsync(d_st, 0, 64, s));
        a.stamps = d_st;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
        MTK_HIP(hipStreamSynchronize(s));
        unsigned long long h[8];
        MTK_HIP(hipMemcpy(h, d_st, 64, hipMemcpyDeviceToHost));
        const double n = (double)h[5] > 0 ? (double)h[5] : 1.0;
        fprintf(stderr, "[mtk stamps] tiles/wg-sum=%llu cycles/tile: prologue %.0f proj %.0f l1 %.0f hh %.0f head %.0f; clock %.0f MHz\n", h[5], h[0] / n,
                h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[7] ? 100.0 * (double)h[6] / (double)h[7] : 0.0);
        return MTK_OK;
    }
    return a.mode == 1 ? launch<512, 1>(L, a, grid, lds, s) : a.mode == 2 ? launch<512, 2>(L, a, grid, lds, s) : launch<512, 0>(L, a, grid, lds, s);
}

// [rows][cols] strided -> compact [rows][ld] (zero padded): the fused Tartan-Federer kernel stages records by 16-byte DMA
__global__ void k_compact_rows(float* dst, int ld, const float* src, int64_t src_ld, int64_t rows, int cols) {
    const int64_t n = rows * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ld;
        const int cidx = (int)(i - r * ld);
        dst[i] = cidx < cols ? src[r * src_ld + cidx] : 0.0f;
    }
}

}  // namespace

void mtk_free_fused(mtk_ctx* c, bool keep_ws) {
    mtk_release(c, c->d_wfrag);
    mtk_release(c, c->d_bias_all);
    mtk_release(c, c->d_embb);
    mtk_release(c, c->d_wfrag_tf);
    mtk_release(c, c->d_embf);
    c->d_wfrag_tf = nullptr;
    c->d_embf = nullptr;
    c->tf_chunks = 0;
    if (!keep_ws) {  // the compact record copy of the Tartan-Federer path is scratch, not part of the weight image
        if (c->tf_pack) (void)hipFree(c->tf_pack);
        c->tf_pack = nullptr;
        c->tf_pack_elems = 0;
    }
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
    mtk_free_fused(c, true);
    int rc;
    Layout L;
    if (!layout_for(c, L)) return MTK_OK;
    static bool attr_set = false;
    if (!attr_set) {
        const int max_lds = (int)kMaxLds;
        const void* fns[] = {
            reinterpret_cast<const void*>(k_fused<512, 1, 0>), reinterpret_cast<const void*>(k_fused<512, 2, 0>),
            reinterpret_cast<const void*>(k_fused<512, 3, 0>), reinterpret_cast<const void*>(k_fused<512, 4, 0>),
            reinterpret_cast<const void*>(k_fused<512, 1, 1>), reinterpret_cast<const void*>(k_fused<512, 2, 1>),
            reinterpret_cast<const void*>(k_fused<512, 3, 1>), reinterpret_cast<const void*>(k_fused<512, 4, 1>),
            reinterpret_cast<const void*>(k_fused<512, 1, 2>), reinterpret_cast<const void*>(k_fused<512, 2, 2>),
            reinterpret_cast<const void*>(k_fused<512, 3, 2>), reinterpret_cast<const void*>(k_fused<512, 4, 2>)};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        attr_set = true;
    }
    c->wfrag_bytes = (size_t)L.total_chunks * kChunkBytes;
    if ((rc = mtk_alloc(c, &c->d_wfrag, c->wfrag_bytes))) return rc;
    if ((rc = mtk_alloc_t(c, &c->d_bias_all, sizeof(float) * L.n_bias))) return rc;
    if ((rc = mtk_alloc_t(c, &c->d_embb, sizeof(float) * (size_t)c->T * c->emb_ld))) return rc;
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
    pack(c->blocks[0], nth, kDimT / 16, 1, nth * (kDimT / 16) / kChunkFrags);
    boff += H;
    for (int i = 1; i < c->n_hidden; ++i) {
        pack(c->blocks[i], nth, ksh, 1, nth * ksh / kChunkFrags);
        boff += H;
    }
    pack(c->head, L.DT, ksh, 1, (L.DT * ksh + kChunkFrags - 1) / kChunkFrags);
    {   // the Tartan-Federer stream: the folded D -> H layer, then the hidden layers and the head as they stand above
        const int ldf = 32 * L.DT, c_f = nth * 2 * L.DT / kChunkFrags;
        float* Wf = nullptr;
        if ((rc = mtk_alloc_t(c, &Wf, sizeof(float) * (size_t)H * ldf))) return rc;
        if ((rc = mtk_alloc(c, &c->d_wfrag_tf, (size_t)L.tf_chunks * kChunkBytes))) return rc;
        if ((rc = mtk_alloc_t(c, &c->d_embf, sizeof(float) * (size_t)c->T * H))) return rc;
        hipLaunchKernelGGL(k_fold_w, dim3(64), dim3(256), 0, s, Wf, ldf, c->blocks[0].w32, c->blocks[0].in_pad, c->proj.w32, c->proj.in_pad, H, kDimT, c->D);
        hipLaunchKernelGGL(k_fold_emb, dim3(1024), dim3(256), 0, s, c->d_embf, c->blocks[0].w32, c->blocks[0].in_pad, c->d_embb, c->emb_ld,
                           c->blocks[0].b32, c->T, H, kDimT);
        unsigned short* dtf = reinterpret_cast<unsigned short*>(c->d_wfrag_tf);
        hipLaunchKernelGGL(k_pack_frags, dim3(512), dim3(256), 0, s, dtf, Wf, H, ldf, nth, 2 * L.DT, 1, (int64_t)0);
        MTK_HIP(hipMemcpyAsync(reinterpret_cast<unsigned char*>(c->d_wfrag_tf) + (size_t)c_f * kChunkBytes,
                               reinterpret_cast<const unsigned char*>(c->d_wfrag) + (size_t)L.tail_chunk0 * kChunkBytes,
                               (size_t)(L.total_chunks - L.tail_chunk0) * kChunkBytes, hipMemcpyDeviceToDevice, s));
        MTK_HIP(hipGetLastError());
        if (!c->pooling) MTK_HIP(hipStreamSynchronize(s));  // outside mtk_mlp_set_weights the scratch is freed, not pooled
        mtk_release(c, Wf);
        c->tf_chunks = L.tf_chunks;
    }
    MTK_HIP(hipGetLastError());
    c->fused_ok = true;
    c->fused_h = H;
    return MTK_OK;
}

int mtk_fused_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                      float* out, int64_t o_rs, int64_t o_cs, hipStream_t s) {
    FusedArgs a{};
    a.rows = B;
    a.mode = t ? 0 : 2;  // one timestep for the launch: the folded first layer (mode 2)
    a.x = x; a.x_rs = x_rs; a.x_cs = x_cs;
    a.tvec = t; a.t_scalar = t_scalar; a.T = c->T;
    a.out = out; a.o_rs = o_rs; a.o_cs = o_cs;
    const int rc = run(c, a, s);
    return rc ? rc : mtk_poison_bad_t(c->trusted_t ? nullptr : t, B, c->T, out, o_rs, o_cs, c->D, nullptr, 0, 0, 0, nullptr, s);
}

// compact [records][round_up(Nn, 4)] copy of the records (used when the caller's rows are not 16-byte aligned / compact)
int mtk_reserve_fused_tf(mtk_ctx* c, int64_t records, hipStream_t s) {
    if (!c->fused_ok) return MTK_OK;
    const size_t need = (size_t)records * ((c->D + 3) & ~3);
    if (need <= c->tf_pack_elems) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    if (c->tf_pack) (void)hipFree(c->tf_pack);
    c->tf_pack = nullptr;
    c->tf_pack_elems = 0;
    MTK_HIP(hipMalloc(&c->tf_pack, sizeof(float) * need));
    c->tf_pack_elems = need;
    ++c->ws_gen;
    return MTK_OK;
}

// Returns MTK_ERR_UNSUPPORTED (without setting up anything) when the staged-record layout cannot hold this shape:
// the caller then runs noising / denoiser / MSE as three kernels.
int mtk_fused_tf_score(mtk_ctx* c, int64_t B, int P, const float* x, int64_t x_ld, const float* noise, int n_t,
                       const int32_t* t_list_host, int add_t, float* score, hipStream_t s) {
    Layout L;
    if (!layout_for(c, L)) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused denoiser: unsupported shape");
    if ((int64_t)P * c->D > kMaxNoise) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused Tartan-Federer kernel: P * Nn = %lld exceeds %d", (long long)P * c->D, kMaxNoise);
    // records are staged in LDS by 16-byte DMA: compact, 16-byte aligned rows of a multiple of 4 floats
    const int ld = (c->D + 3) & ~3;
    {
        FusedArgs probe{};
        probe.mode = 1;
        probe.x_ld = ld;
        probe.n_noise = P * c->D;
        if (lds_need(L, probe) > kMaxLds) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused Tartan-Federer kernel: records of %d features do not fit the LDS staging area", c->D);
    }
    const float* xs = x;
    int64_t xs_ld = x_ld;
    if (x_ld != ld || (reinterpret_cast<uintptr_t>(x) & 15)) {
        const int rcp = mtk_reserve_fused_tf(c, B, s);
        if (rcp) return rcp;
        hipLaunchKernelGGL(k_compact_rows, dim3(1024), dim3(256), 0, s, c->tf_pack, ld, x, x_ld, B, c->D);
        MTK_HIP(hipGetLastError());
        xs = c->tf_pack;
        xs_ld = ld;
    }
    for (int k0 = 0; k0 < n_t; k0 += kMaxTF) {
        const int nk = (n_t - k0 < kMaxTF) ? n_t - k0 : kMaxTF;
        FusedArgs a{};
        a.mode = 1;
        a.BP = B * P;
        a.P = P;
        a.nk = nk;
        a.tiles_per_k = (a.BP + 127) / 128;
        if (a.BP + 128 >= (int64_t)1 << 31 || a.tiles_per_k * nk >= (int64_t)1 << 31) MTK_FAIL(MTK_ERR_UNSUPPORTED, "fused Tartan-Federer kernel: B * P = %lld rows per timestep exceed 2^31", (long long)a.BP);
        a.rows = a.BP * nk;
        a.x = xs; a.x_ld = xs_ld;
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
