// gemm.hip — out = epilogue(A · Wᵀ) on the CDNA4 matrix cores.
//
// Three kernels, one epilogue:
//   gemm_bf16_tile   128x128x64 block tile, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, XOR-swizzled LDS image,
//                    register-staged prefetch of the next K tile.  Encoder GEMMs (M = B*1500).       [MFMA-bound]
//   gemm_bf16_decode M <= 80 decode GEMMs: weights stream once from HBM straight to VGPRs, the activation slice is
//                    shared through LDS, optional split-K with consumer-side reduction.     [HBM/latency-bound]
//   gemm_f32_tile    fp32 "token-exact" mode on v_mfma_f32_16x16x4_f32 (bitwise a k-ordered fmaf chain).
// The MFMA operands are swapped (W is the A operand, the activation the B operand) so each lane ends up with 4
// CONSECUTIVE output columns of one row: bias/GELU/residual/positional adds and the stores are 8/16-byte vectors.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace wx {

// The epilogue of 4 consecutive columns n..n+3 of row m is split in two so that a kernel can issue ALL of a tile's
// residual / positional loads before the first store: the residual stream is updated in place (out == residual), so
// the compiler must keep every load behind every earlier store, and a fused load-add-store per fragment degenerates
// into one exposed memory round trip per fragment (measured: ~50 us of a 256x256 tile's 85 us).
typedef f32x4 EpiAddend;     // rowadd and residual are never used together (launch_gemm rejects it): one register set

__device__ __forceinline__ EpiAddend epilogue_load4(const GemmEpilogue& e, int m, int n, int M, int N) {
    if (m >= M || n >= N) return f32x4{0.f, 0.f, 0.f, 0.f};
    if (e.rowadd) return *reinterpret_cast<const f32x4*>(e.rowadd + (int64_t)(m % e.rowadd_period) * N + n);
    if (e.residual) return *reinterpret_cast<const f32x4*>(e.residual + (int64_t)m * e.ldr + n);
    return f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void epilogue_finish4(const GemmEpilogue& e, int m, int n, f32x4 v, const EpiAddend& add, int M, int N) {
    if (m >= M || n >= N) return;
    if (e.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + n);
        v += b;
    }
    if (e.colscale_period) {
        const float s = e.colscale[n / e.colscale_period];
        v *= s;
    }
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = e.out_dtype == DT_BF16 ? gelu_erf_fast(v[j]) : gelu_erf(v[j]);
    }
    v += add;
    int64_t off;
    void* base;
    if (e.mode == 0) {
        int64_t row = m;
        if (e.out_rows_per_batch) row = (int64_t)(m / e.out_rows_per_batch) * e.out_batch_rows + e.out_row_offset + (m % e.out_rows_per_batch);
        off = row * e.ldo + n;
        base = e.out;
    } else {
        const int s = n / e.hs_d, rem = n % e.hs_d;
        const int h = rem / HEAD_DIM, c = rem % HEAD_DIM;
        const int b = m / e.hs_T, t = m % e.hs_T;
        off = (((int64_t)b * e.hs_H + h) * e.hs_T + t) * HEAD_DIM + c;
        base = e.out_split[s];
    }
    if (e.out_dtype == DT_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
    } else {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(base) + off) = o;
    }
}

__device__ __forceinline__ void epilogue_store4(const GemmEpilogue& e, int m, int n, f32x4 v, int M, int N) {
    epilogue_finish4(e, m, n, v, epilogue_load4(e, m, n, M, N), M, N);
}

// The same epilogue for a lane that keeps ONE group of 4 columns and walks down rows (the 256x256 kernel's slab read-back: row
// m, m + step, ...).  Everything that depends on the column only -- bias, column scale, split / head / channel of the head-split
// layout -- is worked out once, and the row's (batch, position) pair advances by addition instead of two integer divisions per
// element group: the generic form above costs ~250 VALU instructions per group, 15 us of a 49-us tile.
struct EpiColumn {
    f32x4 bias; float scale; char* base; int64_t col_off; bool live;
    int b, t;                      // (m / period, m % period) of the current row: hs_T (mode 1) or out_rows_per_batch (mode 0 remap)
    int period;
};
__device__ __forceinline__ EpiColumn epi_column(const GemmEpilogue& e, int m, int n, int N) {
    EpiColumn c;
    c.live = n < N;
    const int nn = c.live ? n : 0;
    c.bias = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + nn) : f32x4{0.f, 0.f, 0.f, 0.f};
    c.scale = e.colscale_period ? e.colscale[nn / e.colscale_period] : 1.0f;
    const int es = e.out_dtype == DT_F32 ? 4 : 2;
    if (e.mode == 0) {
        c.base = reinterpret_cast<char*>(e.out); c.col_off = (int64_t)nn * es;
        c.period = e.out_rows_per_batch;
    } else {
        const int s = nn / e.hs_d, rem = nn % e.hs_d;
        c.base = reinterpret_cast<char*>(e.out_split[s]);
        c.col_off = ((int64_t)(rem / HEAD_DIM) * e.hs_T * HEAD_DIM + rem % HEAD_DIM) * es;
        c.period = e.hs_T;
    }
    c.b = c.period ? m / c.period : 0; c.t = c.period ? m % c.period : m;
    return c;
}
__device__ __forceinline__ void epi_advance(EpiColumn& c, int step) {       // step < period
    c.t += step;
    if (c.period && c.t >= c.period) { c.t -= c.period; ++c.b; }
}
__device__ __forceinline__ void epi_finish_row(const GemmEpilogue& e, const EpiColumn& c, bool row_ok, f32x4 v, const EpiAddend& add) {
    if (!row_ok || !c.live) return;
    v += c.bias;
    v *= c.scale;
    if (e.act == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = e.out_dtype == DT_BF16 ? gelu_erf_fast(v[j]) : gelu_erf(v[j]);
    }
    v += add;
    int64_t row;
    if (e.mode == 0) row = c.period ? (int64_t)c.b * e.out_batch_rows + e.out_row_offset + c.t : (int64_t)c.t;
    else row = ((int64_t)c.b * e.hs_H) * e.hs_T + c.t;
    if (e.out_dtype == DT_F32) {
        const int64_t ld = e.mode == 0 ? e.ldo : HEAD_DIM;
        *reinterpret_cast<f32x4*>(c.base + row * ld * 4 + c.col_off) = v;
    } else {
        const int64_t ld = e.mode == 0 ? e.ldo : HEAD_DIM;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x4*>(c.base + row * ld * 2 + c.col_off) = o;
    }
}

__device__ __forceinline__ int64_t a_row_offset(const GemmArgs& g, int m) {
    if (g.a_rows_per_batch > 0) return (int64_t)(m / g.a_rows_per_batch) * g.a_batch_stride + (int64_t)(m % g.a_rows_per_batch) * g.lda;
    return (int64_t)m * g.lda;
}

// Blocks that share an XCD (blockIdx % 8) get a contiguous run of tiles, and inside the run tiles are ordered in bands of
// GM row-tiles with m fastest, so the ~64-128 tiles an XCD runs at once form a squarish GM x (concurrency / GM) patch
// that shares GM activation panels and a dozen weight panels through that XCD's 4 MiB L2 (n-fastest order made every
// patch sweep the whole weight matrix: 5-10 GB of L2 misses per encoder GEMM).  Bijective for any tile count.
template <int GM = 8>
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n, orig = blockIdx.x;
    const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int band = lin / (GM * tiles_n), in_band = lin % (GM * tiles_n);
    const int rows = min(GM, tiles_m - band * GM);          // the last band may be shorter
    tm = band * GM + in_band % rows;
    tn = in_band / rows;
}

// ---------------------------------------------------------------------------------------------------------
// bf16, 128x128x64
// ---------------------------------------------------------------------------------------------------------
constexpr int TB_BM = 128, TB_BN = 128, TB_BK = 64;

__global__ __launch_bounds__(256) void gemm_bf16_tile_kernel(GemmArgs g) {
    // LDS image: rows of 64 bf16 (128 B) = 8 chunks of 16 B; chunk kc of row r lives in slot kc ^ (r & 7).
    __shared__ __attribute__((aligned(16))) char lds[(TB_BM + TB_BN) * 128];
    char* ldsA = lds;
    char* ldsW = lds + TB_BM * 128;
    const int tiles_m = (g.M + TB_BM - 1) / TB_BM, tiles_n = (g.N + TB_BN - 1) / TB_BN;
    int tm, tn;
    tile_coords<8>(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TB_BM, n0 = tn * TB_BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int gq = lane >> 4, r = lane & 15;

    const bf16* A = reinterpret_cast<const bf16*>(g.A);
    const bf16* W = reinterpret_cast<const bf16*>(g.W);
    const bf16* a_src[4];
    const bf16* w_src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = tid + p * 256, row = c >> 3, slot = c & 7, kc = slot ^ (row & 7);
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
        a_src[p] = A + a_row_offset(g, m) + kc * 8;
        w_src[p] = W + (int64_t)n * g.ldw + kc * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[4], rw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { ra[p] = *reinterpret_cast<const bf16x8*>(a_src[p]); rw[p] = *reinterpret_cast<const bf16x8*>(w_src[p]); }
    const int nk = g.K / TB_BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<bf16x8*>(ldsA + (tid + p * 256) * 16) = ra[p];
            *reinterpret_cast<bf16x8*>(ldsW + (tid + p * 256) * 16) = rw[p];
        }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k1 = (kt + 1) * TB_BK;
#pragma unroll
            for (int p = 0; p < 4; ++p) { ra[p] = *reinterpret_cast<const bf16x8*>(a_src[p] + k1); rw[p] = *reinterpret_cast<const bf16x8*>(w_src[p] + k1); }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xa[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + r;
                xa[i] = *reinterpret_cast<const bf16x8*>(ldsA + row * 128 + (((ks * 4 + gq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + r;
                wb[j] = *reinterpret_cast<const bf16x8*>(ldsW + row * 128 + (((ks * 4 + gq) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[j][i], 0, 0, 0);
        }
        __syncthreads();
    }
    EpiAddend add[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) add[j][i] = epilogue_load4(g.ep, m0 + wm * 64 + i * 16 + r, n0 + wn * 64 + j * 16 + gq * 4, g.M, g.N);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            epilogue_finish4(g.ep, m0 + wm * 64 + i * 16 + r, n0 + wn * 64 + j * 16 + gq * 4, acc[j][i], add[j][i], g.M, g.N);
}

// ---------------------------------------------------------------------------------------------------------
// bf16, 256x256x64, 8 waves (2 M x 4 N, 128x64 per wave = 32 accumulator tiles), LDS-DMA staging, ping-pong phases.
//
// * Both operand tiles go HBM/L2 -> LDS with global_load_lds (16 B per lane, no VGPR round trip, no ds_write pass).  A
//   wave-instruction writes 8 rows x 128 B linearly; the XOR swizzle the fragment reads need (chunk c of row r in slot
//   c ^ (r & 7): conflict-free ds_read_b128) is applied to the per-lane SOURCE address instead.
// * A K tile (64 deep) is staged as four 16 KiB units -- A-lo / A-hi = the first / second 64 rows of each M wave-row,
//   B-lo / B-hi = the first / second 32 columns of each N wave-column -- and computed as four phases of one 64x32
//   quadrant (16 MFMAs per wave) each:  P1 A-lo x B-lo, P2 A-lo x B-hi, P3 A-hi x B-hi, P4 A-hi x B-lo.  A phase reads
//   only the fragments that change (8 + 4, 4, 8, 4 ds_read_b128) and issues the LDS-DMA of ONE unit, two K tiles
//   ahead at the earliest point its LDS slot is free (>= 2 phases after the slot's last read):
//       P1: A-hi(t+1)   P2: B-lo(t+1)   P3: A-lo(t+2)   P4: B-hi(t+2)
//   so there are always 2-3 units (32-48 KiB per CU) in flight and the DMA is never drained inside the loop: one counted
//   s_waitcnt vmcnt(4) per K tile (in P4, one barrier before the first read of the tile it retires).
// * The two M wave-rows run half a phase apart (waves 4-7 take one extra barrier up front, waves 0-3 one at the end):
//   each SIMD holds one wave of either group, and while one issues its 16 MFMAs the other does its LDS reads and DMA
//   issue, so the matrix pipe alternates between them instead of waiting for either.
// LDS: 2 K-tile buffers x 4 units x 16 KiB = 128 KiB, one workgroup per CU.
// ---------------------------------------------------------------------------------------------------------
constexpr int TL_BM = 256, TL_BN = 256, TL_BK = 64;
constexpr int TL_UNIT = 128 * 128;                   // one staging unit: 128 rows x 64 bf16
constexpr int TL_BUF = 4 * TL_UNIT;                  // one K tile: A-lo, A-hi, B-lo, B-hi
constexpr int TL_SLAB_ROW = 64 * 4 + 16;              // epilogue slab row: 64 fp32 + 16 B pad
constexpr int TL_SLAB = 64 * TL_SLAB_ROW;            // per wave: 64 rows
constexpr int TL_LDS = (8 * TL_SLAB > 2 * TL_BUF) ? 8 * TL_SLAB : 2 * TL_BUF;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__global__ __launch_bounds__(512) void gemm_bf16_tile256_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tiles_m = (g.M + TL_BM - 1) / TL_BM, tiles_n = (g.N + TL_BN - 1) / TL_BN;
    int tm, tn;
    tile_coords<4>(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * TL_BM, n0 = tn * TL_BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int gq = lane >> 4, r = lane & 15;

    // staging sources: unit U (0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi), piece j (0..1): unit-local rows (wid*2 + j)*8 .. +7
    const bf16* src[4][2];
    {
        const bf16* A = reinterpret_cast<const bf16*>(g.A);
        const bf16* W = reinterpret_cast<const bf16*>(g.W);
        const int chunk = (lane & 7) ^ (lane >> 3);               // source chunk that lands in this lane's slot
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = (wid * 2 + j) * 8 + (lane >> 3);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                int m = m0 + (u / 64) * 128 + hf * 64 + (u % 64); if (m > g.M - 1) m = g.M - 1;
                int n = n0 + (u / 32) * 64 + hf * 32 + (u % 32); if (n > g.N - 1) n = g.N - 1;
                src[hf][j] = A + a_row_offset(g, m) + chunk * 8;
                src[2 + hf][j] = W + (int64_t)n * g.ldw + chunk * 8;
            }
        }
    }
    auto stage = [&](int unit, int kt) {
        char* dst = lds + (kt & 1) * TL_BUF + unit * TL_UNIT + (wid * 2) * 1024;
        const int k0 = kt * TL_BK;
        __builtin_amdgcn_global_load_lds((gbl_void*)(src[unit][0] + k0), (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(src[unit][1] + k0), (lds_void*)(dst + 1024), 16, 0, 0);
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[2][4], wf[2][2];
    auto read_a = [&](const char* buf, int hf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = wm * 64 + i * 16 + r;
                xa[ks][i] = *reinterpret_cast<const bf16x8*>(buf + hf * TL_UNIT + u * 128 + (((ks * 4 + gq) ^ (r & 7)) << 4));
            }
    };
    auto read_b = [&](const char* buf, int hf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int u = wn * 32 + j * 16 + r;
                wf[ks][j] = *reinterpret_cast<const bf16x8*>(buf + (2 + hf) * TL_UNIT + u * 128 + (((ks * 4 + gq) ^ (r & 7)) << 4));
            }
    };
    auto mma = [&](int ha, int hb) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[hb * 2 + j][ha * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], xa[ks][i], acc[hb * 2 + j][ha * 4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    // one K tile = four phases; S12 / S34: stage units of tile t+1 / t+2; WAIT: 4 = steady state, 0 = drain, -1 = none
    auto ktile = [&](int t, auto S12, auto S34, auto WAIT) {
        const char* buf = lds + (t & 1) * TL_BUF;
        read_b(buf, 0); __builtin_amdgcn_sched_barrier(0); read_a(buf, 0);
        if constexpr (decltype(S12)::value) stage(1, t + 1);
        mma(0, 0);
        read_b(buf, 1);
        if constexpr (decltype(S12)::value) stage(2, t + 1);
        mma(0, 1);
        read_a(buf, 1);
        if constexpr (decltype(S34)::value) stage(0, t + 2);
        mma(1, 1);
        read_b(buf, 0);
        if constexpr (decltype(S34)::value) stage(3, t + 2);
        if constexpr (decltype(WAIT)::value == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (decltype(WAIT)::value == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mma(1, 0);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    const int nk = g.K / TL_BK;                                   // launch_gemm guarantees nk >= 2
    stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0); stage(0, 1); stage(3, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              // tile 0 landed; A-lo(1), B-hi(1) still in flight
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                    // waves 4-7 run half a phase behind waves 0-3
    for (int t = 0; t + 2 < nk; ++t) ktile(t, T_{}, T_{}, std::integral_constant<int, 4>{});
    ktile(nk - 2, T_{}, F_{}, std::integral_constant<int, 0>{});
    ktile(nk - 1, F_{}, F_{}, std::integral_constant<int, -1>{});
    if (wm == 0) __builtin_amdgcn_s_barrier();
    // Epilogue through LDS (the operand tiles are dead): an accumulator tile gives a lane 4 columns of 16 different
    // rows, i.e. 32/64-B global segments; transposed through a wave-private LDS slab every global access of the epilogue
    // (residual / positional loads, output stores) is instead 16 lanes x 16 B (8 B for bf16) along one row.  Two passes
    // of 64 rows; slab rows padded to 272 B (conflict-free ds_write_b128 from the accumulator layout).
    char* slab = lds + wid * TL_SLAB;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(slab + (i * 16 + r) * TL_SLAB_ROW + (j * 16 + gq * 4) * 4) = acc[j][p * 4 + i];
        const int mrow = m0 + wm * 128 + p * 64 + (lane >> 4), ncol = n0 + wn * 64 + (lane & 15) * 4;
        EpiColumn col = epi_column(g.ep, mrow, ncol, g.N);
        if (g.ep.residual || g.ep.rowadd) {
            // in-place residual update: every load of the pass is issued before its first store (see epilogue_load4)
            EpiAddend add[16];
#pragma unroll
            for (int it = 0; it < 16; ++it)
                add[it] = epilogue_load4(g.ep, mrow + it * 4, ncol, g.M, g.N);
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4), c = lane & 15;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * TL_SLAB_ROW + c * 16);
                epi_finish_row(g.ep, col, mrow + it * 4 < g.M, v, add[it]);
                epi_advance(col, 4);
            }
        } else if (g.ep.out_dtype == DT_BF16 && n0 + 256 <= g.N) {
            // bf16 output: 8 columns = one 16-B store per lane, 8 rows per wave instruction (two column groups of one EpiColumn)
            const int mrow8 = m0 + wm * 128 + p * 64 + (lane >> 3), ncol8 = n0 + wn * 64 + (lane & 7) * 8;
            EpiColumn c8 = epi_column(g.ep, mrow8, ncol8, g.N);
            const f32x4 bias_hi = g.ep.bias ? *reinterpret_cast<const f32x4*>(g.ep.bias + ncol8 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), c = lane & 7;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + row * TL_SLAB_ROW + c * 32);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + row * TL_SLAB_ROW + c * 32 + 16);
                if (mrow8 + it * 8 < g.M) {
                    v0 = (v0 + c8.bias) * c8.scale; v1 = (v1 + bias_hi) * c8.scale;
                    if (g.ep.act == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v0[j] = gelu_erf_fast(v0[j]); v1[j] = gelu_erf_fast(v1[j]); }
                    }
                    const int64_t orow = g.ep.mode == 0 ? (c8.period ? (int64_t)c8.b * g.ep.out_batch_rows + g.ep.out_row_offset + c8.t : (int64_t)c8.t)
                                                        : ((int64_t)c8.b * g.ep.hs_H) * g.ep.hs_T + c8.t;
                    const int64_t ld = g.ep.mode == 0 ? g.ep.ldo : HEAD_DIM;
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { o[j] = (bf16)v0[j]; o[4 + j] = (bf16)v1[j]; }
                    *reinterpret_cast<bf16x8*>(c8.base + orow * ld * 2 + c8.col_off) = o;
                }
                epi_advance(c8, 8);
            }
        } else {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4), c = lane & 15;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * TL_SLAB_ROW + c * 16);
                epi_finish_row(g.ep, col, mrow + it * 4 < g.M, v, f32x4{0.f, 0.f, 0.f, 0.f});
                epi_advance(col, 4);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 decode GEMM (M <= DG_MAX_ROWS = 80): the weight matrix is read exactly once, straight from HBM to VGPRs as the MFMA A operand.
// A decode step at batch 64 reads 3-13 MB per GEMM: less than one HBM round trip per CU, so the kernel is built around
// latency, not bandwidth:
//   * one workgroup = 4 waves = 64 output columns (16 per wave) x all M rows x one K slice [split*Kb, (split+1)*Kb);
//   * the activation slice is staged through LDS in chunks of CKS*32 k (full 128-B lines, coalesced; rows padded by 32 B so
//     the ds_read_b128 B-fragment reads are conflict-free) and shared by the 4 waves: L2 sees each activation byte once per
//     workgroup instead of once per 16 columns;
//   * all weight loads of a chunk (CKS x 16 B per lane) are issued before the first MFMA and the next chunk's loads are
//     issued before the current chunk's MFMAs, so a wave keeps up to 2*CKS KiB in flight;
//   * no load sits under a branch (the last chunk is peeled instead of predicated);
//   * n_split > 1: K is split across workgroups and raw fp32 partial tiles go to partial[split][M][N]; the CONSUMER kernel
//     (reduce + LayerNorm, or the attention kernel's query load) adds them in a fixed order, so there is no atomics, no
//     fence and no extra launch, and results are deterministic.
// ---------------------------------------------------------------------------------------------------------
constexpr int DG_BN = 64;

// NCH = chunks per K slice (compile time, <= 4): the chunk loop is fully unrolled, EVERY weight load of the slice is in
// flight before the first MFMA (NCH*CKS KiB per wave), and activation chunks are fetched two ahead.
template <int MT, int CKS, int NCH, bool PACKED>
__global__ __launch_bounds__(256) void gemm_bf16_decode_kernel(GemmArgs g) {
    constexpr int KS = PACKED ? 512 : 32;        // elements between consecutive k-steps of a lane's weight stream (GemmArgs::Wpk)
    constexpr int ROWB = CKS * 64 + 32;           // padded LDS row (bytes)
    constexpr int CK = CKS * 32;                  // k per chunk
    constexpr int CPR = CKS * 4;                  // 16-B pieces per activation row per chunk
    constexpr int PIECES = MT * 16 * CPR;
    constexpr int PPT = (PIECES + 255) / 256;     // pieces staged per thread
    constexpr int BUF = MT * 16 * ROWB;
    __shared__ __attribute__((aligned(16))) char lds[(NCH > 1 ? 2 : 1) * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gq = lane >> 4, r = lane & 15;
    const int ntiles = (g.N + DG_BN - 1) / DG_BN;
    const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
    const int n0 = tile * DG_BN + wid * 16;
    const int kbeg = split * (NCH * CK);

    int n = n0 + r; if (n > g.N - 1) n = g.N - 1;
    int u = n0 / 16; if (u > g.N / 16 - 1) u = g.N / 16 - 1;     // (waves past a ragged last tile shadow the last unit, as n does)
    const bf16* wp = PACKED ? reinterpret_cast<const bf16*>(g.Wpk) + ((int64_t)u * (g.K / 32) + kbeg / 32) * 512 + lane * 8
                            : reinterpret_cast<const bf16*>(g.W) + (int64_t)n * g.ldw + kbeg + gq * 8;
    const bf16* xsrc[PPT];
    int xdst[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int p = tid + j * 256; if (p > PIECES - 1) p = PIECES - 1;     // duplicates write identical bytes
        int row = p / CPR; const int c = p % CPR;
        xdst[j] = row * ROWB + c * 16;
        if (row > g.M - 1) row = g.M - 1;
        xsrc[j] = reinterpret_cast<const bf16*>(g.A) + (int64_t)row * g.lda + kbeg + c * 8;
    }
    bf16x8 xs[2][PPT], w[NCH][CKS];
#pragma unroll
    for (int j = 0; j < PPT; ++j) xs[0][j] = *reinterpret_cast<const bf16x8*>(xsrc[j]);
    if constexpr (NCH > 1) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) xs[1][j] = *reinterpret_cast<const bf16x8*>(xsrc[j] + CK);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) w[c][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (c * CKS + ks) * KS));
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PPT; ++j) *reinterpret_cast<bf16x8*>(lds + xdst[j]) = xs[0][j];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 2 < NCH) {   // compile-time condition
#pragma unroll
            for (int j = 0; j < PPT; ++j) xs[c & 1][j] = *reinterpret_cast<const bf16x8*>(xsrc[j] + (c + 2) * CK);
        }
        const char* buf = lds + (c & 1) * BUF;
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) {
            bf16x8 xf[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(buf + (i * 16 + r) * ROWB + ks * 64 + gq * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][ks], xf[i], acc[i], 0, 0, 0);
        }
        if (c + 1 < NCH) {
            char* nb = lds + ((c + 1) & 1) * BUF;
#pragma unroll
            for (int j = 0; j < PPT; ++j) *reinterpret_cast<bf16x8*>(nb + xdst[j]) = xs[(c + 1) & 1][j];
            __syncthreads();
        }
    }
    if (g.n_split == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i) epilogue_store4(g.ep, i * 16 + r, n0 + gq * 4, acc[i], g.M, g.N);
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = i * 16 + r, nn = n0 + gq * 4;
            if (m < g.M && nn < g.N) *reinterpret_cast<f32x4*>(g.partial + ((int64_t)split * g.M + m) * g.N + nn) = acc[i];
        }
    }
}

template <int MT, int CKS>
static bool launch_decode_nch(const GemmArgs& g, int nch, int blocks, hipStream_t st) {
#define WX_DG(N) case N: if (g.Wpk) hipLaunchKernelGGL((gemm_bf16_decode_kernel<MT, CKS, N, true>), dim3(blocks), dim3(256), 0, st, g); \
                         else hipLaunchKernelGGL((gemm_bf16_decode_kernel<MT, CKS, N, false>), dim3(blocks), dim3(256), 0, st, g); return true;
    switch (nch) {
        WX_DG(1) WX_DG(2) WX_DG(3) WX_DG(4)
        default: return false;
    }
#undef WX_DG
}
template <int MT>
static bool launch_decode_mt(const GemmArgs& g, int cks, int nch, int blocks, hipStream_t st) {
    switch (cks) {
        case 10: return launch_decode_nch<MT, 10>(g, nch, blocks, st);
        case 8: return launch_decode_nch<MT, 8>(g, nch, blocks, st);
        case 4: return launch_decode_nch<MT, 4>(g, nch, blocks, st);
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 decode GEMM, wide N, whole K per workgroup (the logits: N = vocabulary, 133 MB of weights at d = 1280).
// The kernel above gives such a shape 812 workgroups of one round each, one resident per CU (its LDS): 3.2 rounds with
// every stream drained and refilled in between.  Here the grid is one workgroup of 8 waves per CU and stays: the 16-column
// units are dealt out evenly, a wave takes one unit per round, and its weight stream is a ring of two K chunks (2 x CKS KiB
// per wave in flight) that runs on across rounds -- the refill of a chunk register set is issued right after its MFMAs, for
// the same unit or the next one.  The activation chunk for step s + 1 is written to the other LDS buffer after the MFMAs of
// step s (it was fetched a step earlier, ahead of the weight refills, so its wait never covers an HBM load).  k order and
// MFMA shape are those of the kernel above: results are bit-identical.  NCH must be even (the ring index is static).
// Epilogue: plain fp32 row-major store only (no loads, no branches between the rounds).
// ---------------------------------------------------------------------------------------------------------
template <int MT, int CKS, int NCH, int NW, bool PACKED>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_decode_wide_kernel(GemmArgs g) {
    static_assert(NCH % 2 == 0, "ring of two chunk buffers");
    constexpr int ROWB = CKS * 64 + 32, CK = CKS * 32, CPR = CKS * 4;
    constexpr int PIECES = MT * 16 * CPR, PPT = (PIECES + NW * 64 - 1) / (NW * 64), BUF = MT * 16 * ROWB;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gq = lane >> 4, r = lane & 15;
    const int nunits = (g.N + 15) / 16;
    const int u_begin = (int)((int64_t)blockIdx.x * nunits / gridDim.x), u_end = (int)((int64_t)(blockIdx.x + 1) * nunits / gridDim.x);
    const int rounds = (u_end - u_begin + NW - 1) / NW;          // >= 1: the launcher keeps gridDim.x <= nunits

    const bf16* xsrc[PPT];
    int xdst[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int p = tid + j * NW * 64; if (p > PIECES - 1) p = PIECES - 1;     // duplicates write identical bytes
        int row = p / CPR; const int c = p % CPR;
        xdst[j] = row * ROWB + c * 16;
        if (row > g.M - 1) row = g.M - 1;
        xsrc[j] = reinterpret_cast<const bf16*>(g.A) + (int64_t)row * g.lda + c * 8;
    }
    auto unit_of = [&](int rd) { int u = u_begin + rd * NW + wid; return u > u_end - 1 ? u_end - 1 : u; };   // idle waves shadow the last unit
    constexpr int KS = PACKED ? 512 : 32;          // elements between consecutive k-steps of a lane's weight stream
    auto wptr = [&](int u) {
        if constexpr (PACKED) return reinterpret_cast<const bf16*>(g.Wpk) + (int64_t)u * (g.K / 32) * 512 + lane * 8;
        int n = u * 16 + r; if (n > g.N - 1) n = g.N - 1;
        return reinterpret_cast<const bf16*>(g.W) + (int64_t)n * g.ldw + gq * 8;
    };
    bf16x8 xs[PPT], w[2][CKS];
    const bf16* wp = wptr(unit_of(0));
#pragma unroll
    for (int j = 0; j < PPT; ++j) xs[j] = *reinterpret_cast<const bf16x8*>(xsrc[j]);
    const bool act0 = u_begin + wid < u_end;          // false only for the spare waves of a single-round workgroup
    if (act0) {
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) w[0][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ks * KS));
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) *reinterpret_cast<bf16x8*>(lds + xdst[j]) = xs[j];
#pragma unroll
    for (int j = 0; j < PPT; ++j) xs[j] = *reinterpret_cast<const bf16x8*>(xsrc[j] + CK);
    __builtin_amdgcn_sched_barrier(0);      // activations stay ahead of the weights in the load queue (waits retire in order)
    if (act0) {
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) w[1][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (CKS + ks) * KS));
    }
    __syncthreads();

    // Only the last round can have idle waves (units are dealt 8 per round): they skip weights and MFMAs but keep staging
    // activations and meeting the barriers.  act = this wave has a unit in this round, act_nxt = in the next one.
    auto round = [&](auto last_tag, int rd, const bf16* wp_cur, const bf16* wp_nxt, bool act, bool act_nxt) {
        constexpr bool LAST = decltype(last_tag)::value;
        f32x4 acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const char* buf = lds + (c & 1) * BUF;      // (the conditions on c below fold once the chunk loop is unrolled)
            if (!LAST || act) {
#pragma unroll
                for (int ks = 0; ks < CKS; ++ks) {
                    bf16x8 xf[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(buf + (i * 16 + r) * ROWB + ks * 64 + gq * 16);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c & 1][ks], xf[i], acc[i], 0, 0, 0);
                }
            }
            if (!(LAST && c == NCH - 1)) {         // activations of the next step: fetched a step ago
                char* nb = lds + ((c + 1) & 1) * BUF;
#pragma unroll
                for (int j = 0; j < PPT; ++j) *reinterpret_cast<bf16x8*>(nb + xdst[j]) = xs[j];
            }
            if (!(LAST && c + 2 >= NCH)) {         // fetch for the step after next: activations first, then the ring refill
#pragma unroll
                for (int j = 0; j < PPT; ++j) xs[j] = *reinterpret_cast<const bf16x8*>(xsrc[j] + ((c + 2) % NCH) * CK);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < NCH) {
                    if (!LAST || act) {
#pragma unroll
                        for (int ks = 0; ks < CKS; ++ks) w[c & 1][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp_cur + ((c + 2) * CKS + ks) * KS));
                    }
                } else if (act_nxt) {
#pragma unroll
                    for (int ks = 0; ks < CKS; ++ks) w[c & 1][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp_nxt + ((c + 2 - NCH) * CKS + ks) * KS));
                }
            }
            if (!(LAST && c == NCH - 1)) __syncthreads();
        }
        if (!LAST || act) {              // plain fp32 row-major store (the launcher admits no other epilogue here)
            const int n = (u_begin + rd * NW + wid) * 16 + gq * 4;
            float* out = reinterpret_cast<float*>(g.ep.out);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = i * 16 + r;
                if (m < g.M && n < g.N) *reinterpret_cast<f32x4*>(out + (int64_t)m * g.ep.ldo + n) = acc[i];
            }
        }
    };
    auto active = [&](int rd) { return u_begin + rd * NW + wid < u_end; };
    for (int rd = 0; rd + 1 < rounds; ++rd) {
        const bf16* wn = wptr(unit_of(rd + 1));
        round(std::false_type{}, rd, wp, wn, true, active(rd + 1));
        wp = wn;
    }
    round(std::true_type{}, rounds - 1, wp, wp, active(rounds - 1), false);
}

// (CKS, NCH) of the wide kernel for a K; 0 = not covered.  (A 16-wave build with chunks of 5 k-steps, one round of units per
// workgroup instead of two, measured slower: 32 vs 28 us at d = 1280 -- twice the barriers per byte.)
static int decode_wide_cks(int K, int* nch) {
    for (int cks : {10, 8, 6})
        for (int n : {4, 2})
            if (K == cks * 32 * n) { *nch = n; return cks; }
    return 0;
}
template <int MT>
static bool launch_decode_wide(const GemmArgs& g, int cks, int nch, int blocks, hipStream_t st) {
#define WX_WIDE(C, N) if (cks == C && nch == N) { \
        if (g.Wpk) hipLaunchKernelGGL((gemm_bf16_decode_wide_kernel<MT, C, N, 8, true>), dim3(blocks), dim3(512), 0, st, g); \
        else hipLaunchKernelGGL((gemm_bf16_decode_wide_kernel<MT, C, N, 8, false>), dim3(blocks), dim3(512), 0, st, g); \
        return true; }
    WX_WIDE(10, 4) WX_WIDE(10, 2) WX_WIDE(8, 4) WX_WIDE(8, 2) WX_WIDE(6, 4) WX_WIDE(6, 2)
#undef WX_WIDE
    return false;
}

// k-steps (of 32) per LDS chunk for a K slice of kb; 0 = unsupported slice length
static int decode_gemm_cks(int kb) {
    if (kb % 320 == 0 && kb / 320 <= 4) return 10;
    if (kb % 256 == 0 && kb / 256 <= 4) return 8;
    if (kb % 128 == 0 && kb / 128 <= 4) return 4;
    return 0;
}

// K-split that fills the chip for a decode GEMM whose consumer can add partial tiles (see above); 1 = no split.
int decode_gemm_split(int N, int K) {
    const int ntiles = (N + DG_BN - 1) / DG_BN;
    int best = 1;
    for (int s = 2; s <= 16; ++s) {
        if (K % s) continue;
        const int kb = K / s;
        if (kb < 128 || !decode_gemm_cks(kb)) continue;
        if (ntiles * s <= 256) best = s;
    }
    return best;
}

// whether the decode GEMMs can split K = `K` over `s` workgroups (the slice must be 1..4 chunks of 128, 256 or 320)
bool decode_gemm_split_ok(int K, int s) { return s >= 1 && K % s == 0 && K / s >= 128 && decode_gemm_cks(K / s) != 0; }

// ---------------------------------------------------------------------------------------------------------
// fp32 on v_mfma_f32_16x16x4_f32
// ---------------------------------------------------------------------------------------------------------
constexpr int TF_BK = 16, TF_LD = TF_BK + 4;   // LDS row stride 20 floats keeps rows 16-B aligned

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_tile_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float ldsA[BM * TF_LD];
    __shared__ __attribute__((aligned(16))) float ldsW[BN * TF_LD];
    constexpr int WM = BM / 2, WN = BN / 2, TI = WM / 16, TJ = WN / 16;   // per-wave tile and MFMA repeats
    constexpr int PA = BM * 4 / 256, PW = BN * 4 / 256;                   // float4 chunks per thread
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    int tm, tn;
    tile_coords<8>(tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int gq = lane >> 4, r = lane & 15;
    const float* A = reinterpret_cast<const float*>(g.A);
    const float* W = reinterpret_cast<const float*>(g.W);
    const float* a_src[PA];
    const float* w_src[PW];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int c = tid + p * 256, row = c >> 2, kc = c & 3;
        int m = m0 + row; if (m > g.M - 1) m = g.M - 1;
        a_src[p] = A + a_row_offset(g, m) + kc * 4;
    }
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int c = tid + p * 256, row = c >> 2, kc = c & 3;
        int n = n0 + row; if (n > g.N - 1) n = g.N - 1;
        w_src[p] = W + (int64_t)n * g.ldw + kc * 4;
    }
    f32x4 acc[TJ][TI];
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[PA], rw[PW];
#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(a_src[p]);
#pragma unroll
    for (int p = 0; p < PW; ++p) rw[p] = *reinterpret_cast<const f32x4*>(w_src[p]);
    const int nk = g.K / TF_BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int p = 0; p < PA; ++p) { const int c = tid + p * 256; *reinterpret_cast<f32x4*>(&ldsA[(c >> 2) * TF_LD + (c & 3) * 4]) = ra[p]; }
#pragma unroll
        for (int p = 0; p < PW; ++p) { const int c = tid + p * 256; *reinterpret_cast<f32x4*>(&ldsW[(c >> 2) * TF_LD + (c & 3) * 4]) = rw[p]; }
        __syncthreads();
        if (kt + 1 < nk) {
            const int k1 = (kt + 1) * TF_BK;
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const f32x4*>(a_src[p] + k1);
#pragma unroll
            for (int p = 0; p < PW; ++p) rw[p] = *reinterpret_cast<const f32x4*>(w_src[p] + k1);
        }
#pragma unroll
        for (int kk = 0; kk < TF_BK / 4; ++kk) {
            float xa[TI], wb[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) xa[i] = ldsA[(wm * WM + i * 16 + r) * TF_LD + kk * 4 + gq];
#pragma unroll
            for (int j = 0; j < TJ; ++j) wb[j] = ldsW[(wn * WN + j * 16 + r) * TF_LD + kk * 4 + gq];
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int i = 0; i < TI; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j], xa[i], acc[j][i], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < TI; ++i)
            epilogue_store4(g.ep, m0 + wm * WM + i * 16 + r, n0 + wn * WN + j * 16 + gq * 4, acc[j][i], g.M, g.N);
}

// ---------------------------------------------------------------------------------------------------------
// bf16 decode GEMM with the K split INSIDE the workgroup (the MLP up projection): 32 output columns per workgroup, waves
// (cu, kh) = 16 columns x one half of K each, the two halves added through LDS in a fixed order and the full epilogue
// (bias, GELU, bf16 store) applied -- the split-K variant above needs a separate reduce + bias + GELU launch for that.
// Same bytes per CU and per wave in flight as the 64-column x half-K workgroups of the split variant (N / 32 workgroups).
// Weights come from the fragment-ordered copy (GemmArgs::Wpk) only.  NCH = chunks per K half (1 or 2).
// ---------------------------------------------------------------------------------------------------------
template <int MT, int CKS, int NCH>
__global__ __launch_bounds__(256) void gemm_bf16_decode_khalf_kernel(GemmArgs g) {
    constexpr int ROWB = CKS * 64 + 32, CK = CKS * 32, CPR = CKS * 4;
    constexpr int HALF_PIECES = MT * 16 * CPR, PIECES = 2 * HALF_PIECES, PPT = (PIECES + 255) / 256, BUF = MT * 16 * ROWB;
    static_assert(2 * BUF >= 2 * MT * 64 * 16, "the reduction slab reuses the staging buffers");
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];          // [K half][row][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gq = lane >> 4, r = lane & 15;
    const int cu = wid & 1, kh = wid >> 1;
    const int n0 = blockIdx.x * 32 + cu * 16;
    const int khalf = NCH * CK;                                          // launcher: g.K == 2 * khalf, g.N % 32 == 0
    const bf16* wp = reinterpret_cast<const bf16*>(g.Wpk) + ((int64_t)(n0 / 16) * (g.K / 32) + kh * (khalf / 32)) * 512 + lane * 8;
    const bf16* xsrc[PPT];
    int xdst[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int p = tid + j * 256; if (p > PIECES - 1) p = PIECES - 1;     // duplicates write identical bytes
        const int h = p / HALF_PIECES, q = p % HALF_PIECES;
        int row = q / CPR; const int c = q % CPR;
        xdst[j] = h * BUF + row * ROWB + c * 16;
        if (row > g.M - 1) row = g.M - 1;
        xsrc[j] = reinterpret_cast<const bf16*>(g.A) + (int64_t)row * g.lda + h * khalf + c * 8;
    }
    bf16x8 xs[PPT], w[NCH][CKS];
#pragma unroll
    for (int j = 0; j < PPT; ++j) xs[j] = *reinterpret_cast<const bf16x8*>(xsrc[j]);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) w[c][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (c * CKS + ks) * 512));
    f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PPT; ++j) *reinterpret_cast<bf16x8*>(lds + xdst[j]) = xs[j];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) xs[j] = *reinterpret_cast<const bf16x8*>(xsrc[j] + (c + 1) * CK);
        }
        const char* buf = lds + kh * BUF;
#pragma unroll
        for (int ks = 0; ks < CKS; ++ks) {
            bf16x8 xf[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(buf + (i * 16 + r) * ROWB + ks * 64 + gq * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][ks], xf[i], acc[i], 0, 0, 0);
        }
        __syncthreads();                                                 // every wave is done with this chunk's activations
        if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) *reinterpret_cast<bf16x8*>(lds + xdst[j]) = xs[j];
            __syncthreads();
        }
    }
    // The two waves of a column unit swap half of their row tiles through LDS and each finishes the epilogue (bias, GELU,
    // store) of its share: even tiles on the lower-half wave, odd tiles on the upper-half wave, always lower + upper.
    f32x4* red = reinterpret_cast<f32x4*>(lds) + cu * (MT * 64);
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if ((i & 1) != kh) red[i * 64 + lane] = acc[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if ((i & 1) == kh) {
            const f32x4 o = red[i * 64 + lane];
            f32x4 v;
            if (kh == 0) { v[0] = acc[i][0] + o[0]; v[1] = acc[i][1] + o[1]; v[2] = acc[i][2] + o[2]; v[3] = acc[i][3] + o[3]; }
            else { v[0] = o[0] + acc[i][0]; v[1] = o[1] + acc[i][1]; v[2] = o[2] + acc[i][2]; v[3] = o[3] + acc[i][3]; }
            epilogue_store4(g.ep, i * 16 + r, n0 + gq * 4, v, g.M, g.N);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// gemm_bf16_decode_kernel<1, CKS, 1, true> behind the dense-K/V decode cross-attention at <= ACOMB_ROWS rows, with the combine of
// that attention's key splits in its prologue: column k of activation row m is element k % 64 of head k / 64, combined from the
// splits' (acc, m, l) states by attn_combine_splits -- the function attn_decode_combine_kernel is made of -- rounded to bf16 where
// that kernel stores it.  A K slice is whole heads (128, 256 or 320 columns), so a wave's 64 elements share one head's states; the
// weight stream is issued first and the 1-5 elements per thread (18 four-byte loads each at six splits, L2 hits) land under it.
// One launch less per layer: at one window a launch is 3.4-4.4 us of a 53-us layer whatever it does.
// ---------------------------------------------------------------------------------------------------------
template <int CKS>
__global__ __launch_bounds__(256) void gemm_bf16_decode_acomb_kernel(GemmArgs g) {
    constexpr int KB = CKS * 32, ROWB = KB * 2 + 32, EL = (ACOMB_ROWS * KB + 255) / 256;
    __shared__ __attribute__((aligned(16))) char xs[ACOMB_ROWS * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gq = lane >> 4, r = lane & 15;
    const int ntiles = (g.N + DG_BN - 1) / DG_BN;
    const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
    const int n0 = tile * DG_BN + wid * 16;
    const int kbeg = split * KB;
    int u = n0 / 16; if (u > g.N / 16 - 1) u = g.N / 16 - 1;
    const bf16* wp = reinterpret_cast<const bf16*>(g.Wpk) + ((int64_t)u * (g.K / 32) + kbeg / 32) * 512 + lane * 8;
    bf16x8 w[CKS];
#pragma unroll
    for (int ks = 0; ks < CKS; ++ks) w[ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + ks * 512));
    float val[EL];
#pragma unroll
    for (int j = 0; j < EL; ++j) {
        int e = tid + j * 256; if (e > g.M * KB - 1) e = g.M * KB - 1;           // duplicates write identical bytes
        const int row = e / KB, k = kbeg + e % KB;
        val[j] = attn_combine_splits(g.ac_partial + ((int64_t)row * g.ac_H + (k >> 6)) * g.ac_nsplit * (HEAD_DIM + 2), g.ac_nsplit, k & 63);
    }
#pragma unroll
    for (int j = 0; j < EL; ++j) {
        int e = tid + j * 256; if (e > g.M * KB - 1) e = g.M * KB - 1;
        *reinterpret_cast<bf16*>(xs + (e / KB) * ROWB + (e % KB) * 2) = (bf16)val[j];
    }
    __syncthreads();
    const char* xr = xs + (r < g.M ? r : g.M - 1) * ROWB + gq * 16;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < CKS; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + ks * 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], xf, acc, 0, 0, 0);
    }
    if (g.n_split == 1) epilogue_store4(g.ep, r, n0 + gq * 4, acc, g.M, g.N);
    else {
        const int nn = n0 + gq * 4;
        if (r < g.M && nn < g.N) *reinterpret_cast<f32x4*>(g.partial + ((int64_t)split * g.M + r) * g.N + nn) = acc;
    }
}

int launch_gemm_acomb(const GemmArgs& g, hipStream_t st) {
    if (g.in_dtype != DT_BF16 || !g.Wpk || g.M < 1 || g.M > ACOMB_ROWS || g.a_rows_per_batch != 0 || g.N % 16 || g.n_split < 1 || g.K % g.n_split) return 1;
    if (!g.ac_partial || g.ac_nsplit < 2 || g.ac_nsplit > ACOMB_MAX_SPLIT || g.ac_H * HEAD_DIM != g.K) return 1;
    if ((reinterpret_cast<uintptr_t>(g.Wpk) & 15) || (reinterpret_cast<uintptr_t>(g.ac_partial) & 3)) return -1;
    if (g.n_split > 1 && (!g.partial || (reinterpret_cast<uintptr_t>(g.partial) & 15))) return -1;
    if (g.n_split == 1 && !g.ep.out) return -1;
    const int kb = g.K / g.n_split, blocks = ((g.N + DG_BN - 1) / DG_BN) * g.n_split;
    switch (kb) {                                                   // one chunk per K slice, whole heads per slice
        case 128: hipLaunchKernelGGL((gemm_bf16_decode_acomb_kernel<4>), dim3(blocks), dim3(256), 0, st, g); break;
        case 256: hipLaunchKernelGGL((gemm_bf16_decode_acomb_kernel<8>), dim3(blocks), dim3(256), 0, st, g); break;
        case 320: hipLaunchKernelGGL((gemm_bf16_decode_acomb_kernel<10>), dim3(blocks), dim3(256), 0, st, g); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// 0 = launched, 1 = shape not covered (the caller takes the split-K path), < 0 = error
int launch_gemm_khalf(const GemmArgs& g, hipStream_t st) {
    if (g.in_dtype != DT_BF16 || !g.Wpk || g.M <= 0 || g.M > DG_MAX_ROWS || g.a_rows_per_batch != 0 || g.N % 32 || g.K % 64) return 1;
    if (g.ep.mode != 0 || g.ep.residual || g.ep.rowadd || !g.ep.out) return 1;
    if ((g.lda * 2) % 16 != 0 || ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.Wpk)) & 15)) return -1;
    const int khalf = g.K / 2, blocks = g.N / 32, mt = (g.M + 15) / 16;
#define WX_KH(M_, C, N) if (mt == M_ && khalf == C * 32 * N) { hipLaunchKernelGGL((gemm_bf16_decode_khalf_kernel<M_, C, N>), dim3(blocks), dim3(256), 0, st, g); return 0; }
#define WX_KH_ALL(M_) WX_KH(M_, 10, 2) WX_KH(M_, 8, 2) WX_KH(M_, 6, 2) WX_KH(M_, 8, 1) WX_KH(M_, 6, 1)
    WX_KH_ALL(1) WX_KH_ALL(2) WX_KH_ALL(3) WX_KH_ALL(4) WX_KH_ALL(5)
#undef WX_KH_ALL
#undef WX_KH
    return 1;
}

// W [N][K] bf16 row-major -> MFMA-fragment order [N/16][K/32][64][8] (GemmArgs::Wpk); N % 16 == 0, K % 32 == 0
__global__ void pack_gemm_weight_kernel(const bf16* __restrict__ W, int64_t ldw, int K, bf16* __restrict__ out, int64_t pieces) {
    const int kb_n = K / 32;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pieces; p += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(p & 63);
        const int64_t blk = p >> 6;
        const int kb = (int)(blk % kb_n);
        const int64_t u = blk / kb_n;
        *reinterpret_cast<bf16x8*>(out + p * 8) = *reinterpret_cast<const bf16x8*>(W + (u * 16 + (lane & 15)) * ldw + kb * 32 + (lane >> 4) * 8);
    }
}
int launch_pack_gemm_weight(const void* W, int64_t ldw, int N, int K, void* out, hipStream_t st) {
    if (!W || !out || N <= 0 || K <= 0 || N % 16 || K % 32 || ldw % 8) return -1;
    if ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(out)) & 15) return -1;
    const int64_t pieces = (int64_t)N * K / 8;
    hipLaunchKernelGGL(pack_gemm_weight_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const bf16*>(W), ldw, K, reinterpret_cast<bf16*>(out), pieces);
    return 0;
}

int launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    if (g.ep.rowadd && g.ep.residual) return -1;
    if (g.N % 4 != 0) return -1;
    const size_t es = dtype_size(g.in_dtype);
    if ((g.lda * es) % 16 != 0 || (g.ldw * es) % 16 != 0 || (g.a_batch_stride * es) % 16 != 0) return -1;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) return -1;
    if (g.ep.mode == 1 && (g.ep.hs_d % HEAD_DIM != 0 || g.ep.hs_T <= 0 || g.N % g.ep.hs_d != 0 || g.N / g.ep.hs_d > 3)) return -1;
    if (g.ep.colscale_period && (g.ep.colscale_period % 4 != 0 || (g.N + g.ep.colscale_period - 1) / g.ep.colscale_period > 3)) return -1;
    if (g.in_dtype == DT_BF16) {
        const int kb = (g.n_split >= 1 && g.K % g.n_split == 0) ? g.K / g.n_split : 0;
        const int cks = kb ? decode_gemm_cks(kb) : 0;
        if (g.n_split > 1 && (!cks || !g.partial || g.M > DG_MAX_ROWS || g.a_rows_per_batch != 0)) return -1;
        static const bool no_wide = getenv("WX_NO_WIDE") != nullptr;
        int wnch = 0;
        const GemmEpilogue& ep = g.ep;
        const bool plain_f32 = ep.out && ep.out_dtype == DT_F32 && !ep.bias && !ep.act && !ep.residual && !ep.rowadd && !ep.colscale_period
                               && ep.mode == 0 && !ep.out_rows_per_batch && ep.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(ep.out) & 15) == 0;
        if (g.Wpk && (g.N % 16 != 0 || (reinterpret_cast<uintptr_t>(g.Wpk) & 15))) return -1;
        const int wcks = (!no_wide && g.n_split <= 1 && g.N >= 16384 && plain_f32) ? decode_wide_cks(g.K, &wnch) : 0;
        if (g.M <= DG_MAX_ROWS && g.a_rows_per_batch == 0 && wcks) {
            const int nunits = (g.N + 15) / 16, blocks = nunits / 8 < 256 ? nunits / 8 : 256;     // one workgroup per CU
            bool ok;
            switch ((g.M + 15) / 16) {
                case 1: ok = launch_decode_wide<1>(g, wcks, wnch, blocks, st); break;
                case 2: ok = launch_decode_wide<2>(g, wcks, wnch, blocks, st); break;
                case 3: ok = launch_decode_wide<3>(g, wcks, wnch, blocks, st); break;
                case 4: ok = launch_decode_wide<4>(g, wcks, wnch, blocks, st); break;
                default: ok = launch_decode_wide<5>(g, wcks, wnch, blocks, st); break;
            }
            if (!ok) return -1;
        } else if (g.M <= DG_MAX_ROWS && g.a_rows_per_batch == 0 && cks) {
            const int blocks = ((g.N + DG_BN - 1) / DG_BN) * g.n_split, nch = kb / (cks * 32);
            bool ok;
            switch ((g.M + 15) / 16) {
                case 1: ok = launch_decode_mt<1>(g, cks, nch, blocks, st); break;
                case 2: ok = launch_decode_mt<2>(g, cks, nch, blocks, st); break;
                case 3: ok = launch_decode_mt<3>(g, cks, nch, blocks, st); break;
                case 4: ok = launch_decode_mt<4>(g, cks, nch, blocks, st); break;
                default: ok = launch_decode_mt<5>(g, cks, nch, blocks, st); break;
            }
            if (!ok) return -1;
        } else {
            if (g.K % TB_BK != 0) return -1;
            static const bool no256 = getenv("WX_NO_TILE256") != nullptr;
            if (!no256 && g.M >= 4096 && g.N % TL_BN == 0 && g.K >= 2 * TL_BK) {
                static bool attr_set[64] = {};                    // the attribute is per device
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
                if (!attr_set[dev]) {
                    if (hipFuncSetAttribute((const void*)gemm_bf16_tile256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS) != hipSuccess) return -2;
                    attr_set[dev] = true;
                }
                const int tiles = ((g.M + TL_BM - 1) / TL_BM) * (g.N / TL_BN);
                hipLaunchKernelGGL(gemm_bf16_tile256_kernel, dim3(tiles), dim3(512), TL_LDS, st, g);
            } else {
                const int tiles = ((g.M + TB_BM - 1) / TB_BM) * ((g.N + TB_BN - 1) / TB_BN);
                hipLaunchKernelGGL(gemm_bf16_tile_kernel, dim3(tiles), dim3(256), 0, st, g);
            }
        }
    } else {
        if (g.K % TF_BK != 0) return -1;
        if (g.M <= 64) {
            const int tiles = ((g.M + 63) / 64) * ((g.N + 63) / 64);
            hipLaunchKernelGGL((gemm_f32_tile_kernel<64, 64>), dim3(tiles), dim3(256), 0, st, g);
        } else {
            const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
            hipLaunchKernelGGL((gemm_f32_tile_kernel<128, 128>), dim3(tiles), dim3(256), 0, st, g);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
