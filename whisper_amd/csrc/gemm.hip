// gemm.hip — out = epilogue(A · Wᵀ) on the CDNA4 matrix cores.
//
// Three kernels, one epilogue:
//   gemm_bf16_tile   128x128x64 block tile, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, XOR-swizzled LDS image,
//                    register-staged prefetch of the next K tile.  Encoder GEMMs (M = B*1500).       [MFMA-bound]
//   gemm_bf16_decode M <= 64 decode GEMMs: weights stream once from HBM straight to VGPRs, the activation slice is
//                    shared through LDS, optional split-K with consumer-side reduction.     [HBM/latency-bound]
//   gemm_f32_tile    fp32 "token-exact" mode on v_mfma_f32_16x16x4_f32 (bitwise a k-ordered fmaf chain).
// The MFMA operands are swapped (W is the A operand, the activation the B operand) so each lane ends up with 4
// CONSECUTIVE output columns of one row: bias/GELU/residual/positional adds and the stores are 8/16-byte vectors.
#include "common.h"

namespace wx {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// 4 consecutive columns n..n+3 of row m
__device__ __forceinline__ void epilogue_store4(const GemmEpilogue& e, int m, int n, f32x4 v, int M, int N) {
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
        for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
    }
    if (e.rowadd) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(e.rowadd + (int64_t)(m % e.rowadd_period) * N + n);
        v += r;
    }
    if (e.residual) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(e.residual + (int64_t)m * e.ldr + n);
        v += r;
    }
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

__device__ __forceinline__ int64_t a_row_offset(const GemmArgs& g, int m) {
    if (g.a_rows_per_batch > 0) return (int64_t)(m / g.a_rows_per_batch) * g.a_batch_stride + (int64_t)(m % g.a_rows_per_batch) * g.lda;
    return (int64_t)m * g.lda;
}

// Blocks that share an XCD (blockIdx % 8) get a contiguous run of tiles, n fastest, so an XCD's L2 sees each
// activation row-panel once and keeps the (small) weight panel resident.  Bijective for any tile count.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n, orig = blockIdx.x;
    const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    tm = lin / tiles_n;
    tn = lin % tiles_n;
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
    tile_coords(tiles_m, tiles_n, tm, tn);
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
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            epilogue_store4(g.ep, m0 + wm * 64 + i * 16 + r, n0 + wn * 64 + j * 16 + gq * 4, acc[j][i], g.M, g.N);
}

// ---------------------------------------------------------------------------------------------------------
// bf16 decode GEMM (M <= 64): the weight matrix is read exactly once, straight from HBM to VGPRs as the MFMA A operand.
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
template <int MT, int CKS, int NCH>
__global__ __launch_bounds__(256) void gemm_bf16_decode_kernel(GemmArgs g) {
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
    const bf16* wp = reinterpret_cast<const bf16*>(g.W) + (int64_t)n * g.ldw + kbeg + gq * 8;
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
        for (int ks = 0; ks < CKS; ++ks) w[c][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + c * CK + ks * 32));
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
    switch (nch) {
        case 1: hipLaunchKernelGGL((gemm_bf16_decode_kernel<MT, CKS, 1>), dim3(blocks), dim3(256), 0, st, g); return true;
        case 2: hipLaunchKernelGGL((gemm_bf16_decode_kernel<MT, CKS, 2>), dim3(blocks), dim3(256), 0, st, g); return true;
        case 3: hipLaunchKernelGGL((gemm_bf16_decode_kernel<MT, CKS, 3>), dim3(blocks), dim3(256), 0, st, g); return true;
        case 4: hipLaunchKernelGGL((gemm_bf16_decode_kernel<MT, CKS, 4>), dim3(blocks), dim3(256), 0, st, g); return true;
        default: return false;
    }
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
    tile_coords(tiles_m, tiles_n, tm, tn);
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

int launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    if (g.N % 4 != 0) return -1;
    const size_t es = dtype_size(g.in_dtype);
    if ((g.lda * es) % 16 != 0 || (g.ldw * es) % 16 != 0 || (g.a_batch_stride * es) % 16 != 0) return -1;
    if ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.W)) & 15) return -1;
    if (g.ep.mode == 1 && (g.ep.hs_d % HEAD_DIM != 0 || g.ep.hs_T <= 0 || g.N % g.ep.hs_d != 0 || g.N / g.ep.hs_d > 3)) return -1;
    if (g.ep.colscale_period && (g.ep.colscale_period % 4 != 0 || (g.N + g.ep.colscale_period - 1) / g.ep.colscale_period > 3)) return -1;
    if (g.in_dtype == DT_BF16) {
        const int kb = (g.n_split >= 1 && g.K % g.n_split == 0) ? g.K / g.n_split : 0;
        const int cks = kb ? decode_gemm_cks(kb) : 0;
        if (g.n_split > 1 && (!cks || !g.partial || g.M > 64 || g.a_rows_per_batch != 0)) return -1;
        if (g.M <= 64 && g.a_rows_per_batch == 0 && cks) {
            const int blocks = ((g.N + DG_BN - 1) / DG_BN) * g.n_split, nch = kb / (cks * 32);
            bool ok;
            switch ((g.M + 15) / 16) {
                case 1: ok = launch_decode_mt<1>(g, cks, nch, blocks, st); break;
                case 2: ok = launch_decode_mt<2>(g, cks, nch, blocks, st); break;
                case 3: ok = launch_decode_mt<3>(g, cks, nch, blocks, st); break;
                default: ok = launch_decode_mt<4>(g, cks, nch, blocks, st); break;
            }
            if (!ok) return -1;
        } else {
            if (g.K % TB_BK != 0) return -1;
            const int tiles = ((g.M + TB_BM - 1) / TB_BM) * ((g.N + TB_BN - 1) / TB_BN);
            hipLaunchKernelGGL(gemm_bf16_tile_kernel, dim3(tiles), dim3(256), 0, st, g);
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
