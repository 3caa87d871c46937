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
// k-values are one 16-byte read.  L2 -> LDS by LDS-DMA, no staging registers; THREE record buffers and two weight
// buffers (160 KB): records are the long-latency operand.  The kernel is persistent (one workgroup per CU walks its
// tiles; the k-blocks of all of them form one DMA stream), its epilogue uses neither vector loads (bias through the
// scalar cache into the accumulator init) nor LDS (v_permlane32_swap pairs the two half-pieces of a record's 16 bytes).
//
// Measured (2^20 records, 4x1024, MI355X, same box): 7.6 ms = 945 TFLOP/s (round start 8.75 ms).  In-kernel stamps
// (-DDIAG_STAMPS, tools/build_variant.sh), per 256 x 256 x 1024 tile: k-loop 24.5 us, epilogue 2.5 us, accumulator
// init 0.35 us, first wait of a tile (drains the previous tile's stores) 1.4 us, other waits 0.45 us per k-block
// (1.5 us) against 0.85 us of MFMA time.  History of the epilogue: 5.8 us with sixteen load -> vmcnt(0) round trips
// per 32 rows; 3.2 us staged through LDS with a vmcnt(0) the compiler had put at the join of the `if (emb)` branch.
#include <math.h>
#include <stdlib.h>

#include "mtk_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int BT = 256, BK = 64;
constexpr int kRowPad = 256;  // activation matrices are padded to the record tile
constexpr int NT = 2, WN = NT * 32;  // a wave's tile: 128 records (4 MFMA tiles) x 64 features (2)
// Wave grid WR x (8 / WR), MW 32-record MFMA tiles per wave.  <2, 4> is the 256 x 256 tile of the hidden layers (record
// tile 32 KB + weight tile 32 KB per k-block).  <4, 2> is a 256 x 128 tile for a narrow head (N <= 128: 32 KB + 16 KB):
// no MFMAs spent on 128 padding columns, and still one tile per 256 records, so a 65,536-row pass fills all 256 CUs.
template <int WR, int MW> struct Geo {
    static constexpr int WC = 8 / WR, BM = WR * MW * 32, BN = WC * WN;
    static constexpr int kTileX = BM * BK * 2, kTileW = BN * BK * 2;  // bytes of a record / weight tile of one k-block
    static constexpr int XPW = BM / 64, WPW = BN / 64, OPW = XPW + WPW;  // 8-row, 1 KiB DMA pieces per wave and k-block: records, weights
};
// THREE record buffers and two weight buffers (96 + 64 KB = 160 KB for the hidden layers): records are the long-latency
// operand (HBM / Infinity Cache, ~2 us), weights hit the L2, so the record stream runs one k-block further ahead.
template <int WR, int MW> constexpr size_t lds_bytes() { return 3 * Geo<WR, MW>::kTileX + 2 * Geo<WR, MW>::kTileW; }

struct GemmB {
    int64_t M;       // valid rows (fp32 output only)
    int64_t MT;      // row tiles (Mpad / BM)
    int N;           // valid output features (fp32 output only)
    int nb_n;        // column tiles (Npad / BN)
    int Kpad;        // multiple of BK: extent of the bf16 operands
    const uint16_t* X; int64_t x_ld;   // bf16 [Mpad][x_ld]
    int64_t x_lo;                      // split mode: column offset of the low halves inside a record row (>= Kpad)
    const uint16_t* W; int64_t w_ld;   // bf16 [Npad256][w_ld], zero padded (w_ld = Kpad; split mode: 2 Kpad, hi | lo)
    const float* bias;                 // [Npad256], zero padded
    const float* emb; int emb_ld; const int64_t* tvec; int t_scalar;  // optional time-embedding gather (first layer)
    int T;           // rows of emb: per-row timesteps are clamped into it; the host poisons such rows afterwards (mtk_tclamp)
    int relu;
    // output: bf16 row-major [Mpad][y_rs] (OF32 = false) or fp32 with generic strides, bounds-checked (OF32 = true)
    void* Y; int64_t y_rs, y_cs;
    int64_t y_lo;    // SPLIT, bf16 output: column offset of the low halves inside an output row
    unsigned long long* stamps;  // DIAG_STAMPS builds only
};

extern __shared__ __attribute__((aligned(16))) uint16_t gb_smem[];

// Persistent: one workgroup per CU walks a list of output tiles, and the k-blocks of ALL its tiles form one stream
// through the LDS buffers — the weight cursor runs two blocks ahead of the MFMAs, the record cursor three, across tile
// boundaries.  A tile's epilogue (ReLU, bf16 rounding, staging, stores) therefore runs with the next tile's first blocks
// already in flight, its stores drain under the next tile's MFMAs, and there is no per-tile launch / first-load latency.
// EMB: the first layer's epilogue adds the record's time-embedding row (vector loads).  It is a template parameter, not
// a run-time branch: the compiler put the loads' `s_waitcnt vmcnt(0)` at the join of the branch, where every layer
// then waited, once per 32 rows, for its previous stores and the DMA in flight (~1 us each).
// SPLIT (MTK_PREC_BF16X3): every fp32 value v travels as two bf16 numbers, hi = bf16(v) and lo = bf16(v - hi) (v = hi + lo
// to ~2^-17 relative), stored as the two halves [hi | lo] of a row of 2 Kpad elements, for activations and weights
// alike.  x.w ~ xh.wh + xl.wh + xh.wl (the dropped xl.wl term is ~2^-16 of the product): the same k-loop walks 3 nkb
// k-blocks - phase 0 reads (xh, wh), phase 1 (xl, wh), phase 2 (xh, wl) - into the same fp32 accumulators, so the
// arithmetic is three bf16 MFMAs per product with fp32 accumulation and nothing else changes: same tiles, same DMA
// stream, same barriers.  The epilogue splits its fp32 result again and stores both halves.
template <bool OF32, int WR, int MW, bool EMB, bool SPLIT>
__global__ __launch_bounds__(512, 1) void k_gemm_bf16(GemmB g) {
    typedef Geo<WR, MW> G;
    constexpr int kTileX = G::kTileX, kTileW = G::kTileW, kWBase = 3 * G::kTileX;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / G::WC, wc = wid % G::WC, li = lane & 31, hh = lane >> 5;
    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, each with its own L2.  Row tile mblk belongs to
    // XCD mblk & 7; its nb_n column tiles are consecutive slots of that XCD's list, and the XCD's workgroups take the
    // slots round-robin, so the column tiles of a record tile run at the same time on ONE XCD: the records come from
    // HBM once (the weights, <= 2 MiB a layer, stay resident in every L2).
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3, wstride = gridDim.x >> 3;
    const int64_t slots = ((g.MT - xcd + 7) / 8) * g.nb_n;  // this XCD's tiles
    if (wq >= slots) return;
    const int nkb1 = g.Kpad / BK;               // k-blocks of one operand half
    const int nkb = SPLIT ? 3 * nkb1 : nkb1;    // k-blocks a tile consumes
    // first column (element offset inside a row) that k-block kb of a tile reads from the record / weight matrix
    auto xcol = [&](int kb) -> int64_t {
        if (!SPLIT) return (int64_t)kb * BK;
        return kb < nkb1 ? (int64_t)kb * BK : (kb < 2 * nkb1 ? g.x_lo + (int64_t)(kb - nkb1) * BK : (int64_t)(kb - 2 * nkb1) * BK);
    };
    auto wcol = [&](int kb) -> int64_t { return (int64_t)((SPLIT && kb >= nkb1) ? kb - nkb1 : kb) * BK; };  // weights: [hi | lo], lo at Kpad

    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int lr = lane >> 3;
    unsigned voff_x[2], voff_w[2];  // [parity of the 8-row chunk]: the swizzle key (row >> 1) & 7 takes row bit 3 from it
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int lc = (lane & 7) ^ ((lr >> 1) | (par << 2));
        voff_x[par] = (unsigned)(lr * g.x_ld * 2 + lc * 16);
        voff_w[par] = (unsigned)(lr * g.w_ld * 2 + lc * 16);
    }
    // Operand tiles (256 rows x 64 bf16 = one 128-byte line per row, 32 KB) go L2 -> LDS by LDS-DMA, no registers
    // involved: one wave-instruction moves 8 full lines (64 lanes x 16 B) into 1 KiB of LDS, 4 + 4 of them per wave and
    // k-block.  The LDS image is linear in the lane, so the bank swizzle is applied on the GLOBAL side: LDS slot
    // (r, cs) holds piece cs ^ ((r >> 1) & 7) of row r.  ds_read_b128 resolves banks over 256-byte rows (two tile rows)
    // in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: each group reads 8 row PAIRS whose pair index
    // is distinct mod 8, so with this key its 16 lanes land on the 16 different 16-byte slots (keyed on r & 7 instead,
    // SQ_LDS_BANK_CONFLICT showed every read 2-way).
    // DMA cursors (records, weights): tile and k-block of the next block to request, the buffer it goes to, its source
    struct Cursor { int64_t slot; int kb; int buf; const unsigned char* base; };
    Cursor cx{wq, 0, 0, nullptr}, cw{wq, 0, 0, nullptr};
    auto x_base = [&](int64_t slot) { return reinterpret_cast<const unsigned char*>(g.X + ((slot / g.nb_n) * 8 + xcd) * G::BM * g.x_ld); };
    auto w_base = [&](int64_t slot) { return reinterpret_cast<const unsigned char*>(g.W + (int64_t)((int)(slot % g.nb_n) * G::BN) * g.w_ld); };
    // A wave's n pieces of a block land in n consecutive KiB of LDS.  Piece 0 writes M0 (their middle); every piece carries its
    // KiB as the instruction offset, which global_load_lds applies to the LDS and to the global address alike - so the scalar
    // source pointer is handed over with the same offset taken off.  One M0 write and wait state per operand and block instead
    // of a save / write / wait / restore around every piece.  M0 keeps its value between the pieces: the compiler puts nothing
    // in M0 in this kernel (checked in the ISA: these are its only M0 accesses).
    auto dma = [&](const unsigned char* gp, unsigned voff, unsigned dst0, int j, int n) {
        const int imm = (j - n / 2) * 1024;
        const unsigned char* gq = gp - imm;
        if (j == 0) {
            const unsigned m0v = dst0 + (unsigned)(n / 2) * 1024u;
            if (n == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048" : : "v"(voff), "s"(gq), "s"(m0v) : "memory");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-1024" : : "v"(voff), "s"(gq), "s"(m0v) : "memory");
            return;
        }
        switch (imm) {
            case -1024: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-1024" : : "v"(voff), "s"(gq) : "memory"); break;
            case 0: asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gq) : "memory"); break;
            default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voff), "s"(gq) : "memory"); break;
        }
    };
    static_assert((G::XPW == 4 || G::XPW == 2) && (G::WPW == 4 || G::WPW == 2), "piece offsets above cover 2 or 4 pieces per wave");
    auto x_op = [&](int j) {  // piece j of this wave's XPW of the record block at the cursor
        if (cx.slot >= slots) return;
        const int chunk = wv * G::XPW + j;
        dma(cx.base + (int64_t)chunk * 8 * g.x_ld * 2 + xcol(cx.kb) * 2, voff_x[chunk & 1], (unsigned)(cx.buf * kTileX + wv * G::XPW * 1024), j, G::XPW);
    };
    auto w_op = [&](int j) {
        if (cw.slot >= slots) return;
        const int chunk = wv * G::WPW + j;
        dma(cw.base + (int64_t)chunk * 8 * g.w_ld * 2 + wcol(cw.kb) * 2, voff_w[chunk & 1], (unsigned)(kWBase + cw.buf * kTileW + wv * G::WPW * 1024), j, G::WPW);
    };
    auto x_advance = [&]() {
        if (cx.slot >= slots) return;
        cx.buf = cx.buf == 2 ? 0 : cx.buf + 1;
        if (++cx.kb == nkb) {
            cx.kb = 0;
            cx.slot += wstride;
            if (cx.slot < slots) cx.base = x_base(cx.slot);
        }
    };
    auto w_advance = [&]() {
        if (cw.slot >= slots) return;
        cw.buf ^= 1;
        if (++cw.kb == nkb) {
            cw.kb = 0;
            cw.slot += wstride;
            if (cw.slot < slots) cw.base = w_base(cw.slot);
        }
    };
    const unsigned char* lds = reinterpret_cast<const unsigned char*>(gb_smem);
    struct Frags { bf16x8 a[MW], b[NT]; };
    auto read_frags = [&](Frags& f, int xb, int wb, int ks) {
#pragma unroll
        for (int q = 0; q < MW; ++q) {
            const int ra = (wr * MW + q) * 32 + li, kp = ks * 2 + hh;
            f.a[q] = *reinterpret_cast<const bf16x8*>(lds + xb * kTileX + ra * 128 + ((kp ^ ((ra >> 1) & 7)) * 16));
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int rb = wc * WN + q * 32 + li, kp = ks * 2 + hh;
            f.b[q] = *reinterpret_cast<const bf16x8*>(lds + kWBase + wb * kTileW + rb * 128 + ((kp ^ ((rb >> 1) & 7)) * 16));
        }
    };

    f32x16 acc[MW][NT];
    // A operand = weights (rows n), B operand = records: D[n][m], i.e. lane -> record, register -> feature
    auto mfma_row = [&](const Frags& f, int mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[nt], f.a[mt], acc[mt][nt], 0, 0, 0);
    };
    auto mfma_all = [&](const Frags& f) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) mfma_row(f, mt);
        __builtin_amdgcn_sched_barrier(0);
    };

    // Schedule of one k-block (four 16-wide k-steps, fragment sets F0/F1 alternating, each read one k-step ahead):
    // the barrier sits BEFORE the last k-step's MFMAs, so the fragment reads that follow it (first k-step of the next
    // block) and the DMA issue (into the buffers just released) are covered by 16 MFMAs instead of stalling the matrix
    // pipe once per k-block.  After the barrier of block c a wave requests its pieces of weight block c + 2, THEN of
    // record block c + 3: loads return in order, so at the next block's wait `vmcnt(XPW)` — everything but those last
    // record pieces — means record block c + 2 (requested a block earlier) and weight block c + 2 have landed.
    Frags F0, F1;
    cx.base = x_base(wq);
    cw.base = w_base(wq);
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int j = 0; j < G::XPW; ++j) x_op(j);
        x_advance();
        if (b < 2) {
#pragma unroll
            for (int j = 0; j < G::WPW; ++j) w_op(j);
            w_advance();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int xi = 0, wi = 0;   // buffers of the block being consumed
    bool x_tail = false;  // the previous block's tail requested record pieces after its weight pieces
#ifdef DIAG_STAMPS
    unsigned long long st_k = 0, st_e = 0, st_w0 = 0, st_wn = 0, st_n = 0, st_s = 0, st_i = 0, st_m0 = 0, st_m1 = 0, st_a, st_b;
#define STAMP(x) x = wall_clock64()
#else
#define STAMP(x)
#endif
    for (int64_t slot = wq; slot < slots; slot += wstride) {
        const int64_t m0 = ((slot / g.nb_n) * 8 + xcd) * G::BM;
        const int n0 = (int)(slot % g.nb_n) * G::BN;
        // The accumulators start from the bias.  It comes through the SCALAR cache (the wave's 64 columns are uniform;
        // a lane picks its half of each group of 8 by lane >> 5): vector loads here or in the epilogue would queue behind
        // the DMA in flight and behind the previous tile's stores on vmcnt — the epilogue used to spend 5.8 of a tile's
        // 34 us in sixteen load -> s_waitcnt vmcnt(0) round trips per 32 rows.
        {
            const float* bp = g.bias + n0 + (wv % G::WC) * WN;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x8 b0, b1, b2, b3;
                asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dwordx8 %3, %4, 0x60\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&s"(b0), "=&s"(b1), "=&s"(b2), "=&s"(b3)
                             : "s"(bp + nt * 32)
                             : "memory");
                const f32x8 bq[4] = {b0, b1, b2, b3};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bv = hh ? bq[q][4 + j] : bq[q][j];
#pragma unroll
                        for (int a = 0; a < MW; ++a) acc[a][nt][4 * q + j] = bv;
                    }
            }
        }
#ifdef DIAG_STAMPS
        { const unsigned long long t0 = st_a; STAMP(st_a); if (st_n) st_i += st_a - t0; }
#endif
        read_frags(F0, xi, wi, 0);
        for (int kb = 0; kb < nkb; ++kb) {
            read_frags(F1, xi, wi, 1);
            mfma_all(F0);
            read_frags(F0, xi, wi, 2);
            mfma_all(F1);
            read_frags(F1, xi, wi, 3);
            mfma_all(F0);
            STAMP(st_b);
            // next block landed (this wave's share); F1 is in registers.  The previous tile's stores may be outstanding
            // too and do not return in order with the loads, but the count still holds: with S stores, the weight
            // pieces and XPW younger record pieces in flight, "at most XPW outstanding" needs S + WPW completions, and
            // fewer than WPW loads done (they return in order: weights first) would leave at most S + WPW - 1.
            if (x_tail) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::XPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();  // ... everyone's share, and this block's buffers have no readers left
#ifdef DIAG_STAMPS
            { const unsigned long long now = wall_clock64(); if (kb == 0) st_w0 += now - st_b; else st_wn += now - st_b; }
#endif
            __builtin_amdgcn_sched_barrier(0);
            const int xn = xi == 2 ? 0 : xi + 1;
            if (kb + 1 < nkb) read_frags(F0, xn, wi ^ 1, 0);
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(F1, mt);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = mt * G::OPW / MW; j < (mt + 1) * G::OPW / MW; ++j) {
                    if (j < G::WPW) w_op(j);
                    else x_op(j - G::WPW);
                }
            }
            w_advance();
            x_tail = cx.slot < slots;
            x_advance();
            xi = xn;
            wi ^= 1;
            __builtin_amdgcn_sched_barrier(0);
        }

#ifdef DIAG_STAMPS
        STAMP(st_b); st_k += st_b - st_a; ++st_n;
#endif
        // C/D map with weights as the A operand: record (m) = lane & 31, feature n = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5):
        // a lane holds 4 consecutive features per register quad.  fp32 output (head): stored directly through the generic
        // strides.  bf16 output: the wave's 128 x 64 tile goes through its own 4 KB of LDS in four 32-row pieces so that it
        // leaves as row-contiguous 16-byte pieces (128 B per row per wave) instead of 8-byte granules 2 KB apart.
        // Staging image: 32 rows x 128 B, no padding; the 8-byte granule gi of row r sits at granule gi ^ (r & 15), so
        // the lanes' ds_write_b64 (32 rows, same gi) and the ds_read_b128 of the copy-out both touch every bank once
        // (a padded row stride of 144 B made the writes 4-way conflicted, a quarter of the kernel's time).  LDS
        // operations of one wave execute in order, so the piece needs no barrier.
        // opaque copy of the lane id: everything the epilogue derives from it (staging slots, output addresses) is
        // recomputed per tile instead of being hoisted out of the tile loop and kept alive — at 255 registers that
        // spilled an address pair whose reload put a vmcnt(0) in front of every epilogue
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 31, hh_e = lane_e >> 5;
        int trs[MW];  // first layer: the timesteps of the lane's MW records, requested together (one round trip, not MW)
        if (EMB) {
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) {
                const int64_t row = m0 + (wr * MW + mt) * 32 + li_e;
                trs[mt] = (g.tvec && row < g.M) ? mtk_tclamp(g.tvec[row], g.T) : (g.tvec ? 0 : g.t_scalar);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
            const int64_t row = m0 + (wr * MW + mt) * 32 + li_e;
            float4 eq[NT * 4];  // time-embedding row of the record (first layer only): all eight loads in flight together
            if (EMB) {
                const int tr = trs[mt];
#pragma unroll
                for (int i = 0; i < NT * 4; ++i) {
                    const int n = n0 + wc * WN + (i >> 2) * 32 + 8 * (i & 3) + 4 * hh_e;
                    eq[i] = n < g.emb_ld ? *reinterpret_cast<const float4*>(g.emb + (int64_t)tr * g.emb_ld + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                unsigned pk[4][2];  // [q]: the lane's 4 features of group q as two packed bf16 pairs
                unsigned pl[SPLIT ? 4 : 1][2];  // SPLIT: their low halves
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = nt * 32 + 8 * q + 4 * hh_e;  // feature inside the wave's 64
                    const int n = n0 + wc * WN + nl;
                    float v[4] = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                    if (EMB) {
                        const float4 e4 = eq[nt * 4 + q];
                        v[0] += e4.x; v[1] += e4.y; v[2] += e4.z; v[3] += e4.w;
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
                        const __bf16 e[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        pk[q][0] = reinterpret_cast<const unsigned*>(e)[0];
                        pk[q][1] = reinterpret_cast<const unsigned*>(e)[1];
                        if (SPLIT) {  // low halves: what the bf16 rounding of v left behind, itself rounded to bf16
                            const __bf16 l[4] = {(__bf16)(v[0] - (float)e[0]), (__bf16)(v[1] - (float)e[1]), (__bf16)(v[2] - (float)e[2]),
                                                 (__bf16)(v[3] - (float)e[3])};
                            pl[q][0] = reinterpret_cast<const unsigned*>(l)[0];
                            pl[q][1] = reinterpret_cast<const unsigned*>(l)[1];
                        }
                    }
                }
                if (!OF32) {
                    // A lane holds features 8q + 4 (lane >> 5) + [0, 4) of its record: lanes l and l + 32 hold the two
                    // halves of every 8-feature (16-byte) piece.  v_permlane32_swap trades the upper lanes' half of the
                    // even group for the lower lanes' half of the odd group, so each lane ends up with one whole piece
                    // (lower lanes: the odd group, upper lanes: the even one) and stores 16 bytes; the two lanes of a
                    // record write 32 contiguous bytes, a row's 128 bytes leave in four instructions.  No LDS staging.
                    uint16_t* yb = reinterpret_cast<uint16_t*>(g.Y) + row * g.y_rs + n0 + wc * WN + nt * 32;
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][0], pk[2 * p2][0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][1], pk[2 * p2][1], false, false);
                        *reinterpret_cast<uint4*>(yb + 8 * (2 * p2 + (hh_e ? 0 : 1))) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        if (SPLIT) {  // the low halves go to the second half of the output row (y_lo = column offset of it)
                            const auto l0 = __builtin_amdgcn_permlane32_swap(pl[2 * p2 + 1][0], pl[2 * p2][0], false, false);
                            const auto l1 = __builtin_amdgcn_permlane32_swap(pl[2 * p2 + 1][1], pl[2 * p2][1], false, false);
                            *reinterpret_cast<uint4*>(yb + g.y_lo + 8 * (2 * p2 + (hh_e ? 0 : 1))) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
                        }
                    }
                }
            }
#ifdef DIAG_STAMPS
            if (mt == 0) st_m1 += wall_clock64() - st_b;
#endif
        }
#ifdef DIAG_STAMPS
        { const unsigned long long now = wall_clock64(); st_s += now - st_b; }
#endif
#ifdef DIAG_STAMPS
        STAMP(st_a); st_e += st_a - st_b;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DIAG_STAMPS
    if (tid == 0 && g.stamps) {
        atomicAdd(g.stamps + 0, st_k); atomicAdd(g.stamps + 1, st_e); atomicAdd(g.stamps + 2, st_w0); atomicAdd(g.stamps + 3, st_wn);
        atomicAdd(g.stamps + 4, st_n); atomicAdd(g.stamps + 5, st_n * (unsigned long long)nkb);
        atomicAdd(g.stamps + 6, st_s); atomicAdd(g.stamps + 7, st_i); atomicAdd(g.stamps + 8, st_m0); atomicAdd(g.stamps + 9, st_m1);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// First block of a one-timestep launch whose input is the reverse chain's STATE (mtk_mlp_forward_state): the denoiser
// input of GMD:1406 is cat[float(z), log-one-hot(z_idx)], which the step kernel used to write out as 480 B/row of fp32
// (and k_x_to_bf16 then re-read and re-wrote as bf16).  Here a workgroup builds the bf16 operand tile of its 256 records
// in LDS straight from z [Nn][ld] (fp64, what the step kernel keeps anyway) and z_idx [C][ld]: 200 B/row read, nothing
// written, no conversion launch.  Same rounding as the old route (double -> float -> bf16, both round-to-nearest-even;
// log(1e-30) rounded to bf16 once), same folded weights W1 Wp and bias row embf[t], same MFMA order over k: same output
// bits.  K = kp0 <= 128 (one or two 64-wide k-blocks), so the whole weight tile of a column tile (<= 64 KB) stays in
// LDS while the workgroup walks its record tiles; the launch is bound by its output stores (134 MB per 65,536 rows).
// LDS image of a k-block: as in k_gemm_bf16 (row r, 16-byte piece p at r * 128 + ((p ^ ((r >> 1) & 7)) * 16)).
struct FirstArgs {
    int64_t M, MT;           // valid rows of this pass, row tiles (of 256)
    int nb_n, nkb;           // column tiles (of 256), k-blocks (1 or 2)
    int Nn, C, D;
    const int32_t* off;      // [C + 1] one-hot column offsets (device)
    const double* z; const int32_t* z_idx; int64_t ld;   // state of this pass's first row; column stride ld
    const uint16_t* W; int64_t w_ld;                      // bf16 [Npad][w_ld], w_ld = nkb * 64
    const float* bias;                                    // [Npad]: row t of the folded bias table
    uint16_t* Y; int64_t y_rs;                            // bf16 [Mpad][y_rs]
};

constexpr int kStateMaxNn = 32, kStateMaxC = 16;  // what a thread prefetches into registers (mtk_mlp_forward_state's limits)

__global__ __launch_bounds__(512, 1) void k_first_state(FirstArgs g) {
    constexpr int MW = 4, WC = 4, BM = 256, BN = 256;
    constexpr int kTile = BM * BK * 2;  // bytes of one k-block of the record or of the weight tile (32 KB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid / WC, wc = wid % WC, li = lane & 31, hh = lane >> 5;
    const int xcd = blockIdx.x & 7, wq = blockIdx.x >> 3, wstride = gridDim.x >> 3;
    const int64_t slots = ((g.MT - xcd + 7) / 8) * g.nb_n;
    if (wq >= slots) return;
    unsigned char* lds = reinterpret_cast<unsigned char*>(gb_smem);
    unsigned char* ldsW = lds + 2 * kTile;
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const __bf16 cold_b = (__bf16)MTK_LOG_TINY;
    const uint16_t cold = *reinterpret_cast<const uint16_t*>(&cold_b);
    // thread (r, half) owns the 16-byte pieces p = half, half + 2, ... of record r's operand row
    const int r = tid & 255, half = __builtin_amdgcn_readfirstlane(tid >> 8);  // half is wave-uniform: column bases stay scalar
    // The state of a tile is requested into registers BEFORE the previous tile's output stores are issued: vector memory
    // operations retire in order, so a wait for loads issued behind 128 KB of stores would wait for the stores to drain
    // (the launch is bound by them) - this way the stores drain under the next tile's operand build and MFMAs.
    double zpf[2][8];
    int ipf[kStateMaxC];
    // A tile's state is loaded under the PREVIOUS tile's MFMAs and is complete in registers before that tile's output stores are
    // issued (`landed`): vector memory operations retire in order, and the compiler waits with vmcnt(0) for any load that has
    // stores around it - loads issued behind 128 KB of stores would make the operand build wait for the stores to drain, and the
    // launch is bound by exactly those stores.  This way they drain under the next tile's build and MFMAs.  (Loads by inline asm
    // with a hand-counted wait are not an option: the compiler, unaware that they land late, reuses their registers.)
    // Branch-free: clamped addresses, every load issued (the host hands over valid pointers even for Nn = 0 or C = 0) - with
    // loads under branches the compiler gives up counting altogether.
    auto prefetch = [&](int64_t slot) {
        const int64_t row = ((slot / g.nb_n) * 8 + xcd) * BM + r;
        const int64_t rowc = row < g.M ? row : g.M - 1;
        const int kz = g.Nn > 0 ? g.Nn - 1 : 0, kc = g.C > 0 ? g.C - 1 : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = (half + 2 * q) * 8 + i;
                zpf[q][i] = g.z[(int64_t)(k < kz ? k : kz) * g.ld + rowc];
            }
#pragma unroll
        for (int k = 0; k < kStateMaxC; ++k) ipf[k] = g.z_idx[(int64_t)(k < kc ? k : kc) * g.ld + rowc];
    };
    auto landed = [&]() {
        asm volatile("" : "+v"(zpf[0][0]), "+v"(zpf[0][1]), "+v"(zpf[0][2]), "+v"(zpf[0][3]), "+v"(zpf[0][4]), "+v"(zpf[0][5]), "+v"(zpf[0][6]), "+v"(zpf[0][7]),
                          "+v"(zpf[1][0]), "+v"(zpf[1][1]), "+v"(zpf[1][2]), "+v"(zpf[1][3]), "+v"(zpf[1][4]), "+v"(zpf[1][5]), "+v"(zpf[1][6]), "+v"(zpf[1][7]));
        asm volatile("" : "+v"(ipf[0]), "+v"(ipf[1]), "+v"(ipf[2]), "+v"(ipf[3]), "+v"(ipf[4]), "+v"(ipf[5]), "+v"(ipf[6]), "+v"(ipf[7]),
                          "+v"(ipf[8]), "+v"(ipf[9]), "+v"(ipf[10]), "+v"(ipf[11]), "+v"(ipf[12]), "+v"(ipf[13]), "+v"(ipf[14]), "+v"(ipf[15]));
    };
    static_assert(kStateMaxC == 16, "landed() names sixteen index registers");
    prefetch(wq);
    landed();  // on EVERY path into the tile loop the state registers are asm results: no load is pending on them inside the loop
    // one-hot column offsets: a copy in LDS behind the two tiles (as scalar loads the compiler hoists all of them out of the
    // tile loop and spills them)
    int* offs = reinterpret_cast<int*>(lds + 4 * kTile);
    if (tid <= g.C) offs[tid] = g.off[tid];
    __syncthreads();
    int n_have = -1;
    f32x16 acc[MW][NT];
    for (int64_t slot = wq; slot < slots; slot += wstride) {
        const int64_t m0 = ((slot / g.nb_n) * 8 + xcd) * BM;
        const int ntile = (int)(slot % g.nb_n), n0 = ntile * BN;
        const bool new_w = ntile != n_have;
        if (new_w) {  // weight tile of this column tile: 8-row, 1-KiB pieces by LDS-DMA, 4 per wave and k-block
            n_have = ntile;
            const int lr = lane >> 3;
            for (int kb = 0; kb < g.nkb; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int chunk = wv * 4 + j;  // rows chunk * 8 .. + 8 of the tile
                    const int lc = (lane & 7) ^ ((lr >> 1) | ((chunk & 1) << 2));
                    const unsigned voff = (unsigned)(lr * g.w_ld * 2 + lc * 16);
                    const unsigned char* gp = reinterpret_cast<const unsigned char*>(g.W + ((int64_t)n0 + chunk * 8) * g.w_ld + kb * BK);
                    const unsigned m0v = (unsigned)(2 * kTile + kb * kTile + chunk * 1024);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gp), "s"(m0v) : "memory");
                }
        }
        // ---- the record tile from the prefetched state
        {
            int r_ = r;
            asm volatile("" : "+v"(r_));  // addresses below are formed per tile, not hoisted out of the tile loop and kept alive
            const bool live = m0 + r_ < g.M;
            const int np = g.nkb * 8;  // 16-byte pieces of a row
            unsigned char* rowp = lds + r_ * 128;
            const int key_ = (r_ >> 1) & 7;
            auto put = [&](int p, const uint16_t (&v)[8]) {
                const uint4 w4 = make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16));
                *reinterpret_cast<uint4*>(rowp + (p >> 3) * kTile + (((p & 7) ^ key_) * 16)) = w4;
            };
#pragma unroll
            for (int q = 0; q < 2; ++q) {  // the (partly) numeric pieces: float(z) rounded to bf16
                const int p = half + 2 * q, k0 = p * 8;  // p <= 3: inside the first k-block
                {
                    uint16_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        // two roundings, as on the materialised route (fp64 state -> .float() -> bf16 operand): kept apart by an opaque
                        // copy, or the compiler folds them into one fp64 -> bf16 conversion (a different result on ties: 1 value in 1e5)
                        float zf = (float)zpf[q][i];
                        asm volatile("" : "+v"(zf));
                        const __bf16 b = (__bf16)zf;
                        const uint16_t e = (k0 + i < g.Nn) ? *reinterpret_cast<const uint16_t*>(&b) : ((k0 + i < g.D) ? cold : (uint16_t)0);
                        v[i] = live ? e : (uint16_t)0;
                    }
                    put(p, v);
                }
            }
#pragma unroll 1
            for (int p = half + 4; p < np; p += 2) {  // all-categorical (or padding) pieces
                uint16_t v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (live && p * 8 + i < g.D) ? cold : (uint16_t)0;
                put(p, v);
            }
            asm volatile("" ::: "memory");  // the fills above before the two-byte stores below (different access types)
            // the hot column of every categorical block: log(1) = 0.  The owner of the piece writes it, after its own fill
            // (LDS operations of one wave execute in order).
            // Branch-free: a write that is not this thread's to make lands in a spare word behind the offsets.
            // (indexed off the __shared__ array itself: through a selected pointer the store loses its address space and becomes
            // a flat store with a vmcnt(0) behind it)
            constexpr int spare = (4 * kTile + 96) / 2;
#pragma unroll
            for (int k = 0; k < kStateMaxC; ++k) {
                const int col = g.Nn + offs[k < g.C ? k : 0] + ipf[k];
                const int p = col >> 3;
                const int dst = (r_ * 128 + (p >> 3) * kTile + (((p & 7) ^ key_) * 16) + (col & 7) * 2) / 2;
                const bool mine = k < g.C && live && (p & 1) == half && p < np && ipf[k] >= 0;
                gb_smem[mine ? dst : spare] = 0;
            }
            asm volatile("" ::: "memory");
        }
        // accumulators start from the bias (scalar loads, as in k_gemm_bf16)
        {
            const float* bp = g.bias + n0 + (wv % WC) * WN;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x8 b0, b1, b2, b3;
                asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %4, 0x40\n\ts_load_dwordx8 %3, %4, 0x60\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&s"(b0), "=&s"(b1), "=&s"(b2), "=&s"(b3)
                             : "s"(bp + nt * 32)
                             : "memory");
                const f32x8 bq[4] = {b0, b1, b2, b3};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float bv = hh ? bq[q][4 + j] : bq[q][j];
#pragma unroll
                        for (int a = 0; a < MW; ++a) acc[a][nt][4 * q + j] = bv;
                    }
            }
        }
        if (new_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's weight pieces have landed (first tile: nothing else is in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (slot + wstride < slots) prefetch(slot + wstride);  // lands under the MFMAs below
        {   // rolling fragment prefetch: the record fragment of the NEXT 32-record sub-tile (and, at a k-step's last sub-tile, the
            // next k-step's weight fragments) is read while the two MFMAs of the current one issue - one spare record fragment
            // and one spare pair of weight fragments instead of a second full set (the kernel has no registers for that)
            auto rd_a = [&](int s, int q) -> bf16x8 {
                const int kb = s >> 2, kp = (s & 3) * 2 + hh, ra = (wr * MW + q) * 32 + li;
                return *reinterpret_cast<const bf16x8*>(lds + kb * kTile + ra * 128 + ((kp ^ ((ra >> 1) & 7)) * 16));
            };
            auto rd_b = [&](int s, int q) -> bf16x8 {
                const int kb = s >> 2, kp = (s & 3) * 2 + hh, rb = wc * WN + q * 32 + li;
                return *reinterpret_cast<const bf16x8*>(ldsW + kb * kTile + rb * 128 + ((kp ^ ((rb >> 1) & 7)) * 16));
            };
            const int ns = g.nkb * 4;
            bf16x8 b0 = rd_b(0, 0), b1 = rd_b(0, 1), a = rd_a(0, 0), an, bn0 = b0, bn1 = b1;
            for (int s = 0; s < ns; ++s) {
#pragma unroll
                for (int mt = 0; mt < MW; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (mt + 1 < MW) {
                        an = rd_a(s, mt + 1);
                    } else if (s + 1 < ns) {
                        an = rd_a(s + 1, 0);
                        bn0 = rd_b(s + 1, 0);
                        bn1 = rd_b(s + 1, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a, acc[mt][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a = an;
                }
                b0 = bn0;
                b1 = bn1;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's fragment reads are done: the next tile's build may overwrite the record image
        landed();         // the next tile's state is in registers before this tile's stores are issued (see prefetch)
        // epilogue: ReLU, bf16, v_permlane32_swap pairs the two half-pieces of a record's 16 bytes (see k_gemm_bf16)
#pragma unroll
        for (int mt = 0; mt < MW; ++mt) {
            const int64_t row = m0 + (wr * MW + mt) * 32 + li;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                unsigned pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const __bf16 e[4] = {(__bf16)fmaxf(acc[mt][nt][4 * q], 0.0f), (__bf16)fmaxf(acc[mt][nt][4 * q + 1], 0.0f),
                                         (__bf16)fmaxf(acc[mt][nt][4 * q + 2], 0.0f), (__bf16)fmaxf(acc[mt][nt][4 * q + 3], 0.0f)};
                    pk[q][0] = reinterpret_cast<const unsigned*>(e)[0];
                    pk[q][1] = reinterpret_cast<const unsigned*>(e)[1];
                }
                uint16_t* yb = g.Y + row * g.y_rs + n0 + wc * WN + nt * 32;
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][0], pk[2 * p2][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][1], pk[2 * p2][1], false, false);
                    *reinterpret_cast<uint4*>(yb + 8 * (2 * p2 + (hh ? 0 : 1))) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
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
// split image: [rows_pad][2 cols_pad], hi = bf16(w) in the first half of a row, lo = bf16(w - hi) in the second
__global__ void k_to_bf16_split_pad(uint16_t* dst, int rows_pad, int cols_pad, const float* src, int src_ld, int rows, int cols) {
    const int64_t n = (int64_t)rows_pad * cols_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
        const float v = (r < rows && c < cols) ? src[(int64_t)r * src_ld + c] : 0.0f;
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        dst[(int64_t)r * 2 * cols_pad + c] = *reinterpret_cast<const uint16_t*>(&h);
        dst[(int64_t)r * 2 * cols_pad + cols_pad + c] = *reinterpret_cast<const uint16_t*>(&l);
    }
}
// proj folded into the first block for launches that carry ONE timestep (every step of the reverse chain): no nonlinearity
// sits between them (MODEL:614, 300), so ReLU(W1 (Wp x + b_proj + emb[t]) + b1) = ReLU((W1 Wp) x + embf[t]) - one GEMM
// and one activation round trip fewer per pass.  Both products in fp32 on the zero-padded images, rounded to bf16 once.
__global__ void k_fold_w_l(float* Wf, int ldf, const float* W1, int ld1, const float* Wp, int ldp, int H, int dim_t) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)H * ldf; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i / ldf), d = (int)(i - (int64_t)h * ldf);
        float acc = 0.0f;
        for (int k = 0; k < dim_t; ++k) acc += W1[(int64_t)h * ld1 + k] * Wp[(int64_t)k * ldp + d];
        Wf[i] = acc;
    }
}
__global__ void k_fold_emb_l(float* embf, int ldo, int H, const float* W1, int ld1, const float* emb, int emb_ld, const float* bp, const float* b1,
                             int T, int dim_t) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)T * ldo; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / ldo), h = (int)(i - (int64_t)t * ldo);
        float acc = 0.0f;
        if (h < H) {
            acc = b1[h];
            for (int k = 0; k < dim_t; ++k) acc += W1[(int64_t)h * ld1 + k] * (emb[(int64_t)t * emb_ld + k] + bp[k]);
        }
        embf[i] = acc;
    }
}
__global__ void k_pad_f32(float* dst, int n_pad, const float* src, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) dst[i] = i < n ? src[i] : 0.0f;
}
// The denoiser input (fp32, generic strides, B rows x D columns) as a zero-padded bf16 [Mpad][ld] matrix.  A workgroup
// converts 64 rows x 128 columns through LDS: it reads along whichever direction of the source is contiguous (lanes
// over rows for the column-major intermediates, over columns for the row-major tensors of the drop-in API) and writes
// whole 256-byte rows of the destination (a thread per 8 columns wrote 16-byte pieces 256 B apart: 2 TB/s).
// SPLIT: the destination row is [hi (ld) | lo (ld)], lo = bf16(x - hi).
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_x_to_bf16(uint16_t* dst, int ld, const float* x, int64_t x_rs, int64_t x_cs, int64_t M, int D) {
    __shared__ uint32_t tile[64 * 65];  // [row][column pair], odd row stride: conflict-free both ways
    __shared__ uint32_t tile_lo[SPLIT ? 64 * 65 : 1];
    const int t = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int col0 = blockIdx.y * 128;
    const bool along_cols = x_cs == 1;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int r = along_cols ? (t >> 6) + 4 * i : (t & 63);
        const int cp = along_cols ? (t & 63) : (t >> 6) + 4 * i;
        const int64_t row = row0 + r;
        const int c = col0 + 2 * cp;
        const bool ok0 = row < M && c < D, ok1 = row < M && c + 1 < D;
        const float v0 = x[(ok0 ? row : 0) * x_rs + (int64_t)(ok0 ? c : 0) * x_cs];
        const float v1 = x[(ok1 ? row : 0) * x_rs + (int64_t)(ok1 ? c + 1 : 0) * x_cs];
        const float f0 = ok0 ? v0 : 0.0f, f1 = ok1 ? v1 : 0.0f;
        const __bf16 e[2] = {(__bf16)f0, (__bf16)f1};
        tile[r * 65 + cp] = *reinterpret_cast<const uint32_t*>(e);
        if (SPLIT) {
            const __bf16 l[2] = {(__bf16)(f0 - (float)e[0]), (__bf16)(f1 - (float)e[1])};
            tile_lo[r * 65 + cp] = *reinterpret_cast<const uint32_t*>(l);
        }
    }
    __syncthreads();
    const int64_t dld = SPLIT ? 2 * (int64_t)ld : ld;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pc = t + 256 * i, r = pc >> 4, c8 = pc & 15;
        if (col0 + c8 * 8 < ld) {
            const uint32_t* q = tile + r * 65 + c8 * 4;
            *reinterpret_cast<uint4*>(dst + (row0 + r) * dld + col0 + c8 * 8) = make_uint4(q[0], q[1], q[2], q[3]);
            if (SPLIT) {
                const uint32_t* ql = tile_lo + r * 65 + c8 * 4;
                *reinterpret_cast<uint4*>(dst + (row0 + r) * dld + ld + col0 + c8 * 8) = make_uint4(ql[0], ql[1], ql[2], ql[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small batches (round 3).  A reverse step for 1 k - 8 k rows — what the reference's callers sample per call (SYNTH:79) —
// gives k_gemm_bf16 4 ... 32 row tiles: 16 ... 128 of 256 CUs each walk a whole K = 1024 k-loop (24 us) per layer, 139 us a
// step.  k_gemm_bf16_s is the same product on 64 x 64 tiles: (rows / 64) x (N / 64) workgroups of 4 waves (a wave = 32
// records x 32 features), both operands L2 -> LDS by LDS-DMA into a four-stage ring (three k-blocks requested ahead, one
// barrier per k-block; the ring of k_gemm2c in mtk_train.hip), two workgroups per CU.  Same arithmetic as the big tile, bit
// for bit: weights are the A operand and records the B operand of the same MFMA over the same 16-wide k-steps in the same
// order, accumulators start from the bias, the epilogue is k_gemm_bf16's (time embedding, ReLU, bf16 rounding,
// v_permlane32_swap pairing, 16-byte stores) — tests/test_gpu_parity.py holds the two against each other with torch.equal.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRingS = 4;
// Pass sizes (padded rows) per tile shape, measured on 4 x 1024 reverse chains (us per step).  64 x 64 against 128 x 128 tiles:
// 2,048 rows 66.2 / 74.6, 3,072 rows 94.1 / 75.7, 4,096 rows 96.4 / 77.4.  128 x 128 against 256 x 256: 8,192 rows 97.0 / 128.5,
// 12,288 rows 123.6 / 130.7, 16,384 rows 154.2 / 145.7.  MTK_LAYERED_SMALL_MAX / MTK_LAYERED_MID_MAX move the limits (A/B).
constexpr int64_t kSmallRowsMax = 2048;   // up to here k_gemm_bf16_s (64 x 64 tiles, two workgroups per CU)
// 128 x 128 tiles of the persistent kernel (Geo<4, 1>, two workgroups per CU) while the 256 x 256 tiles of the widest layer number
// at most kMidTilesMax (three quarters of the CUs: 12,288 rows at 1024 wide), or 256 + at most kMidTailMax (a second round that would
// be a quarter full: 18,432 rows 204 / 235 us, 20,000 rows 213 / 235, 24,576 rows 244 / 240); otherwise 256 x 256
constexpr int64_t kMidTilesMax = 192, kMidTailMax = 64;
constexpr int64_t kFirstSmallRowsMax = 8192;  // the state-input first block on 64 x 64 tiles (k_first_state_s) up to here
constexpr int kStageS = 2 * 64 * 128;  // bytes: record tile (64 rows x 128 B) | weight tile
extern __shared__ __attribute__((aligned(16))) unsigned char gs_smem[];

template <bool OF32, bool EMB>
__global__ __launch_bounds__(256) void k_gemm_bf16_s(GemmB g) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int nk = g.Kpad / BK;

    // accumulators start from the bias: the lane's features are 8 q + 4 hh + [0, 4) of the wave's 32
    f32x16 acc;
    {
        const float* bp = g.bias + n0 + wc * 32 + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * q);
            acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
        }
    }
    int tr = 0;
    float4 eq[4];  // first layer with per-record timesteps: the record's embedding row, requested before the k-loop
    const int64_t row = m0 + wr * 32 + li;
    if (EMB) {
        tr = (g.tvec && row < g.M) ? mtk_tclamp(g.tvec[row], g.T) : (g.tvec ? 0 : g.t_scalar);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wc * 32 + 8 * q + 4 * hh;
            eq[q] = n < g.emb_ld ? *reinterpret_cast<const float4*>(g.emb + (int64_t)tr * g.emb_ld + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // the values above are in registers before the first DMA request: the vmcnt arithmetic below counts DMA pieces only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // wave wv moves rows [16 wv, 16 wv + 16) of both tiles: two 8-row pieces per operand and k-block
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int lr = lane >> 3;
    unsigned voff_x[2], voff_w[2];  // [piece]: row in tile = 16 wv + 8 j + lr, its swizzle key (row >> 1) & 7 = (4 j + (lr >> 1)) & 7
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lc = (lane & 7) ^ ((4 * j + (lr >> 1)) & 7);
        voff_x[j] = (unsigned)(lr * g.x_ld * 2 + lc * 16);
        voff_w[j] = (unsigned)(lr * g.w_ld * 2 + lc * 16);
    }
    const unsigned char* x_base = reinterpret_cast<const unsigned char*>(g.X) + (m0 + 16 * wv) * g.x_ld * 2;
    const unsigned char* w_base = reinterpret_cast<const unsigned char*>(g.W) + ((int64_t)n0 + 16 * wv) * g.w_ld * 2;
    auto issue = [&](int i) {  // k-block i into stage i % kRingS
        const unsigned dst = (unsigned)((i % kRingS) * kStageS + wv * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* gx = x_base + (int64_t)j * 8 * g.x_ld * 2 + (int64_t)i * 128;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_x[j]), "s"(gx), "s"(dst + j * 1024u) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned char* gw = w_base + (int64_t)j * 8 * g.w_ld * 2 + (int64_t)i * 128;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_w[j]), "s"(gw), "s"(dst + 8192u + j * 1024u) : "memory");
        }
    };
    for (int i = 0; i < kRingS - 1 && i < nk; ++i) issue(i);
    for (int i = 0; i < nk; ++i) {
        // k-block i has landed when at most the requests made after it are outstanding: min(kRingS - 2, nk - 1 - i) blocks of 4 pieces
        const int after = nk - 1 - i;
        if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's pieces of block i are in LDS, and every wave is done reading block i - 1's stage
        if (i + kRingS - 1 < nk) issue(i + kRingS - 1);
        const unsigned char* st = gs_smem + (i % kRingS) * kStageS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kp = 2 * ks + hh;
            const int rx = wr * 32 + li, rw = wc * 32 + li;
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(st + rx * 128 + ((kp ^ ((rx >> 1) & 7)) * 16));
            const bf16x8 wf = *reinterpret_cast<const bf16x8*>(st + 8192 + rw * 128 + ((kp ^ ((rw >> 1) & 7)) * 16));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
        }
    }
    // epilogue of k_gemm_bf16 for one 32 x 32 sub-tile: record = lane & 31, features 8 q + 4 (lane >> 5) + [0, 4)
    unsigned pk[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + wc * 32 + 8 * q + 4 * hh;
        float v[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        if (EMB) { v[0] += eq[q].x; v[1] += eq[q].y; v[2] += eq[q].z; v[3] += eq[q].w; }
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
            const __bf16 e[4] = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            pk[q][0] = reinterpret_cast<const unsigned*>(e)[0];
            pk[q][1] = reinterpret_cast<const unsigned*>(e)[1];
        }
    }
    if (!OF32) {
        uint16_t* yb = reinterpret_cast<uint16_t*>(g.Y) + row * g.y_rs + n0 + wc * 32;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][0], pk[2 * p2][0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][1], pk[2 * p2][1], false, false);
            *reinterpret_cast<uint4*>(yb + 8 * (2 * p2 + (hh ? 0 : 1))) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
    }
}

// The state-input first block (k_first_state) on 64 x 64 tiles, for the passes that take the smaller tiles above: at 1,024 rows the
// 256 x 256 kernel runs on 16 workgroups and its one tile per workgroup is a chain of round trips (weights, state, build, MFMAs,
// stores: 15 us whatever the rows).  Here a workgroup builds the operand rows of its 64 records (thread = record x quarter of the
// 16-byte pieces), takes its 64 x K weight tile by LDS-DMA and multiplies once.  Same operand bits (fp64 -> fp32 -> bf16 kept as two
// roundings, log(1e-30) as bf16, zeros on the hot columns), same bias start, same k order: the same output bits.
__global__ __launch_bounds__(256) void k_first_state_s(FirstArgs g) {
    constexpr int kOp = 64 * 128;  // bytes of one k-block of the record or of the weight tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    unsigned char* lds = gs_smem;              // [nkb][64][128 B] record image
    unsigned char* ldsW = gs_smem + 2 * kOp;   // [nkb][64][128 B] weight image
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    f32x16 acc;
    {
        const float* bp = g.bias + n0 + wc * 32 + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b4 = *reinterpret_cast<const float4*>(bp + 8 * q);
            acc[4 * q] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
        }
    }
    // weight tile: wave wv moves rows [16 wv, 16 wv + 16), two 8-row pieces per k-block (swizzle as in k_gemm_bf16_s)
    {
        const int lr = lane >> 3;
        for (int kb = 0; kb < g.nkb; ++kb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lc = (lane & 7) ^ ((4 * j + (lr >> 1)) & 7);
                const unsigned voff = (unsigned)(lr * g.w_ld * 2 + lc * 16);
                const unsigned char* gp = reinterpret_cast<const unsigned char*>(g.W + ((int64_t)n0 + 16 * wv + 8 * j) * g.w_ld + kb * BK);
                const unsigned m0v = (unsigned)(2 * kOp + kb * kOp + wv * 2048 + j * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(gp), "s"(m0v) : "memory");
            }
    }
    // record image: thread (r, quarter) owns the pieces p = quarter, quarter + 4, ... of record r's row
    {
        const __bf16 cold_b = (__bf16)MTK_LOG_TINY;
        const uint16_t cold = *reinterpret_cast<const uint16_t*>(&cold_b);
        const int r = tid & 63, quarter = wv;
        const int64_t row = m0 + r;
        const bool live = row < g.M;
        const int64_t rowc = live ? row : g.M - 1;
        const int kz = g.Nn > 0 ? g.Nn - 1 : 0, kc = g.C > 0 ? g.C - 1 : 0;
        const int np = g.nkb * 8;
        unsigned char* rowp = lds + r * 128;
        const int key = (r >> 1) & 7;
        int idx[kStateMaxC];
#pragma unroll
        for (int k = 0; k < kStateMaxC; ++k) idx[k] = g.z_idx[(int64_t)(k < kc ? k : kc) * g.ld + rowc];
        for (int p = quarter; p < np; p += 4) {
            uint16_t v[8];
            if (p * 8 < g.Nn) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = p * 8 + i;
                    float zf = (float)g.z[(int64_t)(k < kz ? k : kz) * g.ld + rowc];
                    asm volatile("" : "+v"(zf));  // two roundings (fp64 -> fp32 -> bf16), not one: see k_first_state
                    const __bf16 b = (__bf16)zf;
                    const uint16_t e = (k < g.Nn) ? *reinterpret_cast<const uint16_t*>(&b) : ((k < g.D) ? cold : (uint16_t)0);
                    v[i] = live ? e : (uint16_t)0;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (live && p * 8 + i < g.D) ? cold : (uint16_t)0;
            }
            *reinterpret_cast<uint4*>(rowp + (p >> 3) * kOp + (((p & 7) ^ key) * 16)) =
                make_uint4(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16), v[6] | ((unsigned)v[7] << 16));
        }
        asm volatile("" ::: "memory");  // the fills before the two-byte pokes of the same thread (LDS operations of a wave execute in order)
#pragma unroll
        for (int k = 0; k < kStateMaxC; ++k) {
            if (k < g.C && live && idx[k] >= 0) {
                const int col = g.Nn + g.off[k] + idx[k];
                const int p = col >> 3;
                if ((p & 3) == quarter && p < np)
                    *reinterpret_cast<uint16_t*>(rowp + (p >> 3) * kOp + (((p & 7) ^ key) * 16) + (col & 7) * 2) = 0;  // log(1) on the hot column
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int ns = g.nkb * 4;
    for (int sidx = 0; sidx < ns; ++sidx) {
        const int kb = sidx >> 2, kp = (sidx & 3) * 2 + hh;
        const int rx = wr * 32 + li, rw = wc * 32 + li;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(lds + kb * kOp + rx * 128 + ((kp ^ ((rx >> 1) & 7)) * 16));
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(ldsW + kb * kOp + rw * 128 + ((kp ^ ((rw >> 1) & 7)) * 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
    }
    // epilogue of k_first_state for one 32 x 32 sub-tile
    const int64_t row = m0 + wr * 32 + li;
    unsigned pk[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const __bf16 e[4] = {(__bf16)fmaxf(acc[4 * q], 0.0f), (__bf16)fmaxf(acc[4 * q + 1], 0.0f), (__bf16)fmaxf(acc[4 * q + 2], 0.0f), (__bf16)fmaxf(acc[4 * q + 3], 0.0f)};
        pk[q][0] = reinterpret_cast<const unsigned*>(e)[0];
        pk[q][1] = reinterpret_cast<const unsigned*>(e)[1];
    }
    uint16_t* yb = g.Y + row * g.y_rs + n0 + wc * 32;
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][0], pk[2 * p2][0], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * p2 + 1][1], pk[2 * p2][1], false, false);
        *reinterpret_cast<uint4*>(yb + 8 * (2 * p2 + (hh ? 0 : 1))) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

int launch_first_state_small(FirstArgs& g, int64_t Mpad, int Npad, hipStream_t s) {
    if (Mpad % 64 || Npad % 64 || g.nkb < 1 || g.nkb > 2 || Mpad / 64 > 65535 * 32) MTK_FAIL(MTK_ERR_STATE, "state-input block (small tiles): %lld x %d, %d k-blocks", (long long)Mpad, Npad, g.nkb);
    hipLaunchKernelGGL(k_first_state_s, dim3((unsigned)(Mpad / 64), (unsigned)(Npad / 64)), dim3(256), 4 * 64 * 128, s, g);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

template <bool OF32, bool EMB>
int launch_small(GemmB& g, int device, int64_t Mpad, int Npad, hipStream_t s) {
    constexpr int kBytes = kRingS * kStageS;  // 64 KiB: two workgroups per CU
    static bool attr[64] = {};
    if (device < 0 || device >= 64 || !attr[device]) {
        MTK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf16_s<OF32, EMB>), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes));
        if (device >= 0 && device < 64) attr[device] = true;
    }
    if (Mpad % 64 || Npad % 64 || g.Kpad % BK || Mpad / 64 > 65535 * 32) MTK_FAIL(MTK_ERR_STATE, "layered bf16 GEMM (small tiles): %lld x %d x %d", (long long)Mpad, Npad, g.Kpad);
    hipLaunchKernelGGL((k_gemm_bf16_s<OF32, EMB>), dim3((unsigned)(Mpad / 64), (unsigned)(Npad / 64)), dim3(256), kBytes, s, g);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

int n_cus_layered(int device) {
    static int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device]) return cached[device];
    int n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device);
    if (device >= 0 && device < 64) cached[device] = n;
    return n;
}

template <bool OF32, int WR, int MW, bool EMB = false, bool SPLIT = false>
int launch(GemmB& g, int device, int64_t Mpad, int Npad, hipStream_t s) {
    typedef Geo<WR, MW> G;
    constexpr size_t lds = lds_bytes<WR, MW>();
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf16<OF32, WR, MW, EMB, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (Mpad % G::BM || Npad % G::BN) MTK_FAIL(MTK_ERR_STATE, "layered bf16 GEMM: %lld x %d is not a multiple of the %d x %d tile", (long long)Mpad, Npad, G::BM, G::BN);
    g.MT = Mpad / G::BM;
    g.nb_n = Npad / G::BN;
    // one workgroup per CU (160 KB of LDS each; two where a workgroup's buffers take half of it); per XCD no more workgroups than
    // its longest tile list
    int64_t per_xcd = n_cus_layered(device) / 8 * (lds <= 80 * 1024 ? 2 : 1);
    if (per_xcd < 1) per_xcd = 1;
    const int64_t most = ((g.MT + 7) / 8) * g.nb_n;
    if (most < per_xcd) per_xcd = most;
#ifdef DIAG_STAMPS
    static unsigned long long* d_st = nullptr;
    if (!d_st) MTK_HIP(hipMalloc(&d_st, 128));
    MTK_HIP(hipMemsetAsync(d_st, 0, 128, s));
    g.stamps = d_st;
#endif
    hipLaunchKernelGGL((k_gemm_bf16<OF32, WR, MW, EMB, SPLIT>), dim3((unsigned)(per_xcd * 8)), dim3(512), lds, s, g);
    MTK_HIP(hipGetLastError());
#ifdef DIAG_STAMPS
    MTK_HIP(hipStreamSynchronize(s));
    unsigned long long h[16];
    MTK_HIP(hipMemcpy(h, d_st, 128, hipMemcpyDeviceToHost));
    const double n = h[4] ? (double)h[4] : 1.0;
    fprintf(stderr, "[layered stamps] K=%d N=%d tiles=%llu per tile (10 ns ticks): k-loop %.0f epilogue %.0f first-wait %.1f other-waits(avg each) %.1f | epilogue to stores issued %.0f (first 32 rows staged %.0f, their stores issued %.0f), accumulator init %.0f\n", g.Kpad,
            g.nb_n * G::BN, h[4], h[0] / n, h[1] / n, h[2] / n, h[5] > h[4] ? (double)h[3] / (double)(h[5] - h[4]) : 0.0, h[6] / n, h[8] / n, h[9] / n, h[7] / n);
#endif
    return MTK_OK;
}

}  // namespace

void mtk_free_layered(mtk_ctx* c, bool keep_ws) {
    for (void* p : c->wb16) mtk_release(c, p);
    c->wb16.clear();
    for (void* p : c->wb16x) mtk_release(c, p);
    c->wb16x.clear();
    for (float* p : c->bb16) mtk_release(c, p);
    c->bb16.clear();
    mtk_release(c, c->wb16_fold);
    mtk_release(c, c->embf_l);
    c->wb16_fold = nullptr;
    c->embf_l = nullptr;
    c->embf_ld = 0;
    if (!keep_ws) {
        for (int i = 0; i < 2; ++i) {
            if (c->ws16[i]) (void)hipFree(c->ws16[i]);
            c->ws16[i] = nullptr;
        }
        c->ws16_elems = 0;
    }
}

// Weight images for the layer-by-layer bf16 GEMMs, built when the weights are set: fp32 biases padded to
// round_up(out, 256); split images [round_up(out, 256)][2 round_up(in_pad, 64)] (hi | lo) for MTK_PREC_BF16X3, always; plain
// bf16 images [round_up(out, 256)][round_up(in_pad, 64)] for MTK_PREC_BF16 when the fused kernel does not cover the shape.
int mtk_pack_layered(mtk_ctx* c, hipStream_t s, bool plain) {
    mtk_free_layered(c, true);
    int rc;
    std::vector<const MlpLayer*> ls{&c->proj};
    for (auto& l : c->blocks) ls.push_back(&l);
    ls.push_back(&c->head);
    for (const MlpLayer* l : ls) {
        const int kp = round_up(l->in_pad, BK), np = round_up(l->out_pad, BT);
        void* p = nullptr;
        float* b = nullptr;
        if (plain) {
            if ((rc = mtk_alloc(c, &p, sizeof(uint16_t) * (size_t)np * kp))) return rc;
            c->wb16.push_back(p);
            hipLaunchKernelGGL(k_to_bf16_pad, dim3(512), dim3(256), 0, s, reinterpret_cast<uint16_t*>(p), np, kp, l->w32, l->in_pad, l->out_pad, l->in_pad);
        }
        void* px = nullptr;
        if ((rc = mtk_alloc(c, &px, sizeof(uint16_t) * (size_t)np * kp * 2))) return rc;
        c->wb16x.push_back(px);
        hipLaunchKernelGGL(k_to_bf16_split_pad, dim3(512), dim3(256), 0, s, reinterpret_cast<uint16_t*>(px), np, kp, l->w32, l->in_pad, l->out_pad, l->in_pad);
        if ((rc = mtk_alloc_t(c, &b, sizeof(float) * (size_t)np))) return rc;
        c->bb16.push_back(b);
        hipLaunchKernelGGL(k_pad_f32, dim3(8), dim3(256), 0, s, b, np, l->b32, l->out_pad);
    }
    if (plain && !c->blocks.empty()) {  // the folded first block (see k_fold_w_l)
        const MlpLayer& l1 = c->blocks[0];
        const int kp0 = round_up(c->proj.in_pad, BK), np1 = round_up(l1.out_pad, BT);
        float* Wf = nullptr;
        if ((rc = mtk_alloc_t(c, &Wf, sizeof(float) * (size_t)l1.out_pad * c->proj.in_pad))) return rc;
        if ((rc = mtk_alloc(c, &c->wb16_fold, sizeof(uint16_t) * (size_t)np1 * kp0))) return rc;
        if ((rc = mtk_alloc_t(c, &c->embf_l, sizeof(float) * (size_t)c->T * np1))) return rc;
        c->embf_ld = np1;
        hipLaunchKernelGGL(k_fold_w_l, dim3(512), dim3(256), 0, s, Wf, c->proj.in_pad, l1.w32, l1.in_pad, c->proj.w32, c->proj.in_pad, l1.out_pad, c->dim_t);
        hipLaunchKernelGGL(k_to_bf16_pad, dim3(512), dim3(256), 0, s, reinterpret_cast<uint16_t*>(c->wb16_fold), np1, kp0, Wf, c->proj.in_pad, l1.out_pad,
                           c->proj.in_pad);
        hipLaunchKernelGGL(k_fold_emb_l, dim3(1024), dim3(256), 0, s, c->embf_l, np1, l1.out, l1.w32, l1.in_pad, c->d_emb, c->emb_ld, c->proj.b32, l1.b32,
                           c->T, c->dim_t);
        MTK_HIP(hipGetLastError());
        if (!c->pooling) MTK_HIP(hipStreamSynchronize(s));
        mtk_release(c, Wf);
    }
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

namespace {

int max_ld_layered(const mtk_ctx* c) {  // widest bf16 activation row (input included)
    int maxld = round_up(c->proj.in_pad, BK);
    if (round_up(c->proj.out_pad, BT) > maxld) maxld = round_up(c->proj.out_pad, BT);
    for (auto& l : c->blocks) maxld = round_up(l.out_pad, BT) > maxld ? round_up(l.out_pad, BT) : maxld;
    return maxld;
}

// Rows per pass through the layer chain: a layer's input and output (2 x rows x row bytes) should fit the 256 MB Infinity
// Cache, so that what one layer writes the next reads from there and not from HBM (4x1024, 2^20 records: 8.7 ms at 131,072
// rows a pass, 8.1 ms at 65,536; below 32,768 the six launches per pass dominate).
int64_t chunk_rows(int maxld, bool split) {
    static const int64_t chunk_env = [] { const char* e = getenv("MTK_LAYERED_CHUNK"); return e ? (int64_t)atol(e) : (int64_t)0; }();
    int64_t chunk = (chunk_env + kRowPad - 1) / kRowPad * kRowPad;
    if (chunk <= 0) {
        chunk = split ? 65536 : 32768;  // split mode, 1024 wide: 65,536-row passes measured 3.5 % ahead of 32,768 (fewer, fuller launches)
        while (chunk < 262144 && 2 * chunk * (int64_t)maxld * (split ? 8 : 4) <= ((int64_t)256 << 20)) chunk *= 2;
    }
    return chunk;
}

// bf16 activation ping-pong for calls of up to B rows (grown on demand, or ahead of time by mtk_ctx_reserve)
int ensure_ws16(mtk_ctx* c, int64_t B, bool split, hipStream_t s) {
    const int maxld = max_ld_layered(c);
    const int64_t chunk = chunk_rows(maxld, split);
    const int64_t need_rows = (B < chunk ? (B + kRowPad - 1) / kRowPad * kRowPad : chunk);
    const size_t need = (size_t)need_rows * maxld * (split ? 2 : 1);
    if (need <= c->ws16_elems) return MTK_OK;
    MTK_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 2; ++i) {
        if (c->ws16[i]) (void)hipFree(c->ws16[i]);
        c->ws16[i] = nullptr;
        MTK_HIP(hipMalloc(&c->ws16[i], sizeof(uint16_t) * need));
    }
    c->ws16_elems = need;
    ++c->ws_gen;
    return MTK_OK;
}

// can the first block take its input from the chain's state (k_first_state)?  Plain bf16 with the folded first block, an
// operand row of at most two k-blocks, and no more numeric / categorical columns than a thread prefetches into registers
bool state_input_ok(const mtk_ctx* c) {
    return !c->wb16.empty() && c->wb16_fold && c->embf_l && !c->blocks.empty() && round_up(c->proj.in_pad, BK) <= 2 * BK && c->Nn <= kStateMaxNn &&
           c->C <= kStateMaxC && c->wb16.size() == c->blocks.size() + 2;
}

int launch_first_state(mtk_ctx* c, FirstArgs& g, int64_t Mpad, int Npad, hipStream_t s) {
    constexpr size_t lds = 4 * 256 * BK * 2 + 128;  // record tile + weight tile, two k-blocks each (128 KB) + the one-hot offsets
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_first_state), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (Mpad % 256 || Npad % 256) MTK_FAIL(MTK_ERR_STATE, "state-input block: %lld x %d is not a multiple of the 256 x 256 tile", (long long)Mpad, Npad);
    g.MT = Mpad / 256;
    g.nb_n = Npad / 256;
    int64_t per_xcd = n_cus_layered(c->device) / 8;
    if (per_xcd < 1) per_xcd = 1;
    const int64_t most = ((g.MT + 7) / 8) * g.nb_n;
    if (most < per_xcd) per_xcd = most;
    hipLaunchKernelGGL(k_first_state, dim3((unsigned)(per_xcd * 8)), dim3(512), lds, s, g);
    MTK_HIP(hipGetLastError());
    return MTK_OK;
}

// zst / zi / st_ld: the reverse chain's state (fp64 [Nn][st_ld], int32 [C][st_ld]) as the denoiser's input instead of x
template <bool SPLIT>
int forward_layered(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar, float* out,
                    int64_t o_rs, int64_t o_cs, hipStream_t s, const double* zst = nullptr, const int32_t* zi = nullptr, int64_t st_ld = 0) {
    const bool from_state = zst || zi;
    if (from_state && (SPLIT || t || !state_input_ok(c) || t_scalar < 0 || t_scalar >= c->T))
        MTK_FAIL(MTK_ERR_UNSUPPORTED, "mtk_mlp_forward_state: this denoiser / precision takes its input as a matrix (mtk_mlp_forward)");
    const std::vector<void*>& wimg = SPLIT ? c->wb16x : c->wb16;
    if (wimg.size() != c->blocks.size() + 2) MTK_FAIL(MTK_ERR_STATE, "layered bf16 denoiser: weights are not packed");
    constexpr int F = SPLIT ? 2 : 1;  // bf16 numbers per activation / weight
    const int maxld = max_ld_layered(c);
    const int64_t chunk = chunk_rows(maxld, SPLIT);
    int rc0;
    if ((rc0 = ensure_ws16(c, B, SPLIT, s))) return rc0;
    uint16_t* ws[2] = {reinterpret_cast<uint16_t*>(c->ws16[0]), reinterpret_cast<uint16_t*>(c->ws16[1])};
    for (int64_t r0 = 0; r0 < B; r0 += chunk) {
        const int64_t M = (B - r0 < chunk) ? B - r0 : chunk;
        const int64_t Mpad = (M + kRowPad - 1) / kRowPad * kRowPad;
        int rc;
        // input -> bf16 [Mpad][F kp0]
        const int kp0 = round_up(c->proj.in_pad, BK);
        if (!from_state) {
            hipLaunchKernelGGL(k_x_to_bf16<SPLIT>, dim3((unsigned)(Mpad / 64), (unsigned)((kp0 + 127) / 128)), dim3(256), 0, s, ws[0], kp0, x + r0 * x_rs, x_rs, x_cs, M, c->D);
            MTK_HIP(hipGetLastError());
        }
        int cur = 0;
        // small passes take the 64 x 64-tile kernel (k_gemm_bf16_s: same bits, all CUs busy from 1 k rows on); MTK_LAYERED_BIG: A/B switch
        static const int64_t small_max = getenv("MTK_LAYERED_SMALL_MAX") ? atol(getenv("MTK_LAYERED_SMALL_MAX")) : kSmallRowsMax;  // A/B
        static const int64_t mid_max = getenv("MTK_LAYERED_MID_MAX") ? atol(getenv("MTK_LAYERED_MID_MAX")) : -1;  // A/B: a plain row limit
        const bool small = !SPLIT && Mpad <= small_max && !getenv("MTK_LAYERED_BIG");
        // passes between the two: 128 x 128 tiles of the persistent kernel (80 KB of LDS: two workgroups per CU), four times the
        // tiles of the 256 x 256 shape for a chip it would leave half empty; MTK_LAYERED_MID=0: A/B switch
        static const bool no_mid = getenv("MTK_LAYERED_MID") && atoi(getenv("MTK_LAYERED_MID")) == 0;
        // ... chosen by how the 256 x 256 tiles of the widest layer would fill the chip's 256 CUs: at most three quarters of one round
        // (4 x 1024: up to 12,288 rows), or a second round that is at most a quarter full (16,385 - 20,480 rows)
        const int64_t tiles_big = (Mpad / BT) * (round_up(maxld, BT) / BT);
        const bool mid_fill = mid_max >= 0 ? Mpad <= mid_max : (tiles_big <= kMidTilesMax || (tiles_big > 256 && tiles_big <= 256 + kMidTailMax));
        const bool mid = !SPLIT && !small && mid_fill && !no_mid && !getenv("MTK_LAYERED_BIG");
        // one timestep for the whole launch (plain bf16): proj is folded into the first block, whose bias is row t of embf_l
        const bool fold = !SPLIT && !t && c->wb16_fold && c->embf_l && t_scalar >= 0 && t_scalar < c->T;
        int ld = kp0;
        if (!fold) {
            GemmB g{};
            g.M = M; g.N = c->proj.out; g.Kpad = kp0;
            g.X = ws[cur]; g.x_ld = (int64_t)F * kp0; g.x_lo = kp0;
            g.W = reinterpret_cast<const uint16_t*>(wimg[0]); g.w_ld = (int64_t)F * kp0; g.bias = c->bb16[0];
            g.emb = c->d_emb; g.emb_ld = c->emb_ld; g.tvec = t ? t + r0 : nullptr; g.t_scalar = t_scalar; g.T = c->T;
            ld = round_up(c->proj.out_pad, BT);
            g.Y = ws[cur ^ 1]; g.y_rs = (int64_t)F * ld; g.y_lo = ld;
            if (small) rc = launch_small<false, true>(g, c->device, Mpad, ld, s);
            else if (mid) rc = launch<false, 4, 1, true, false>(g, c->device, Mpad, ld, s);
            else rc = launch<false, 2, 4, true, SPLIT>(g, c->device, Mpad, ld, s);
            if (rc) return rc;
            cur ^= 1;
        }
        for (size_t i = 0; i < c->blocks.size(); ++i) {
            const MlpLayer& l = c->blocks[i];
            const bool folded = fold && i == 0;
            if (folded && from_state) {  // the folded first block with its operand built in LDS from the state
                FirstArgs f{};
                f.M = M; f.nkb = kp0 / BK; f.Nn = c->Nn; f.C = c->C; f.D = c->D; f.off = c->d_off;
                // the kernel's state loads are unconditional (clamped): with no numeric / no categorical columns they read the other array
                f.z = zst && c->Nn > 0 ? zst + r0 : reinterpret_cast<const double*>(c->embf_l);
                f.z_idx = zi && c->C > 0 ? zi + r0 : reinterpret_cast<const int32_t*>(c->embf_l);
                f.ld = st_ld;
                f.W = reinterpret_cast<const uint16_t*>(c->wb16_fold); f.w_ld = kp0;
                f.bias = c->embf_l + (size_t)t_scalar * c->embf_ld;
                ld = round_up(l.out_pad, BT);
                f.Y = ws[cur ^ 1]; f.y_rs = ld;
                static const bool first_big = getenv("MTK_FIRST_STATE_BIG") != nullptr;  // A/B: the 256 x 256 kernel at every pass size
                // measured, 4 x 1024 chain, us per step with the 64 x 64 / the 256 x 256 first block: 1,024 rows 54.7 / 63.4, 4,096 rows
                // 71.7 / 77.4, 8,192 rows 95.6 / 95.9, 12,288 rows 127.7 / 121.1
                if ((rc = (Mpad <= kFirstSmallRowsMax && !first_big) ? launch_first_state_small(f, Mpad, ld, s) : launch_first_state(c, f, Mpad, ld, s))) return rc;
                cur ^= 1;
                continue;
            }
            GemmB h{};
            h.M = M; h.N = l.out; h.Kpad = folded ? kp0 : round_up(l.in_pad, BK);
            h.X = ws[cur]; h.x_ld = (int64_t)F * ld; h.x_lo = ld;
            h.W = reinterpret_cast<const uint16_t*>(folded ? c->wb16_fold : wimg[1 + i]); h.w_ld = (int64_t)F * h.Kpad;
            h.bias = folded ? c->embf_l + (size_t)t_scalar * c->embf_ld : c->bb16[1 + i]; h.relu = 1;
            ld = round_up(l.out_pad, BT);
            h.Y = ws[cur ^ 1]; h.y_rs = (int64_t)F * ld; h.y_lo = ld;
            if (small) rc = launch_small<false, false>(h, c->device, Mpad, ld, s);
            else if (mid) rc = launch<false, 4, 1, false, false>(h, c->device, Mpad, ld, s);
            else rc = launch<false, 2, 4, false, SPLIT>(h, c->device, Mpad, ld, s);
            if (rc) return rc;
            cur ^= 1;
        }
        GemmB h{};
        h.M = M; h.N = c->head.out; h.Kpad = round_up(c->head.in_pad, BK);
        h.X = ws[cur]; h.x_ld = (int64_t)F * ld; h.x_lo = ld;
        h.W = reinterpret_cast<const uint16_t*>(wimg.back()); h.w_ld = (int64_t)F * h.Kpad; h.bias = c->bb16.back();
        h.Y = out + r0 * o_rs; h.y_rs = o_rs; h.y_cs = o_cs;
        // the head of a mid-sized pass stays on the 64 x 64 tiles (N <= 128: the larger tiles leave most CUs without one; measured)
        if (small || mid) rc = launch_small<true, false>(h, c->device, Mpad, c->head.out <= 128 ? 128 : round_up(c->head.out_pad, BT), s);
        else if (c->head.out <= 128) rc = launch<true, 4, 2, false, SPLIT>(h, c->device, Mpad, 128, s);
        else rc = launch<true, 2, 4, false, SPLIT>(h, c->device, Mpad, round_up(c->head.out_pad, BT), s);
        if (rc) return rc;
    }
    return mtk_poison_bad_t(c->trusted_t ? nullptr : t, B, c->T, out, o_rs, o_cs, c->D, nullptr, 0, 0, 0, nullptr, s);
}

}  // namespace

int mtk_reserve_layered(mtk_ctx* c, int64_t rows, hipStream_t s) {  // mtk_ctx_reserve: the split image is the larger one
    if (c->wb16x.empty() && c->wb16.empty()) return MTK_OK;
    return ensure_ws16(c, rows, !c->wb16x.empty(), s);
}

bool mtk_layered_state_input_ok(const mtk_ctx* c) { return state_input_ok(c); }

int mtk_layered_bf16_forward_state(mtk_ctx* c, int64_t B, const double* z, const int32_t* z_idx, int64_t ld, int t_scalar, float* out, int64_t o_rs,
                                   int64_t o_cs, hipStream_t s) {
    return forward_layered<false>(c, B, nullptr, 0, 0, nullptr, t_scalar, out, o_rs, o_cs, s, z, z_idx, ld);
}

int mtk_layered_bf16_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                             float* out, int64_t o_rs, int64_t o_cs, hipStream_t s) {
    return forward_layered<false>(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, s);
}

// MTK_PREC_BF16X3: split-bf16 operands (hi + lo), three MFMAs per product, fp32 accumulate - the fast path that meets the
// fp32 parity bar (<= 1e-4; measured ~1e-5 of the output scale)
int mtk_layered_bf16x3_forward(mtk_ctx* c, int64_t B, const float* x, int64_t x_rs, int64_t x_cs, const int64_t* t, int t_scalar,
                               float* out, int64_t o_rs, int64_t o_cs, hipStream_t s) {
    return forward_layered<true>(c, B, x, x_rs, x_cs, t, t_scalar, out, o_rs, o_cs, s);
}
