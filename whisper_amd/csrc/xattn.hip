// xattn.hip — decoder cross-attention on the ENCODER OUTPUT itself ("absorbed" K/V projections), bf16, gfx950.
//
// The per-layer cross-attention K and V of a window are projections of one and the same encoder output E [1500][d]:
//     k_j = Wk e_j (no bias),  v_j = Wv e_j + bv.
// For one query row the scores and the output of head h are therefore
//     s_j  = (q_h s)·(k_j,h s) = (s² Wk_hᵀ q_h)·e_j  = q̃_h · e_j          q̃_h = s² Wk_hᵀ q_h  in R^d,  s = 64^-0.25
//     o_h  = Σ_j p_j v_j,h     = Wv_h (Σ_j p_j e_j) + bv_h = Wv_h c_h + bv_h   (Σ_j p_j = 1)
// i.e. attention with 20 "heads" of width d whose keys AND values are the rows of E.  A decode step then streams
// 1500·d·2 B per window and layer (3.84 MB at d = 1280) instead of the K and V caches (7.68 MB): half the bytes of what
// was the dominant, HBM-bound kernel, no per-layer K/V cache (15.7 GB at 64 windows) and no K/V projection in the
// encoder stage (0.315 TFLOP per window).  The price: 20x the attention FLOPs, which is why this runs on the matrix cores
// (≈15 % MFMA load at 64 rows), and two small latency-bound GEMMs per layer around it:
//
//   xattn_qtilde_kernel   q̃[row][h][:] = s · Wk_hᵀ bf16((bq + Σ split-K tiles of the query projection)·s)     K = 64 per head
//   xattn_estream_kernel  one workgroup per (row, key split): E tiles of 32 keys through LDS (one image serves the
//                         row-wise S = q̃·Eᵀ reads and the transposed P·E reads), online softmax, locally normalised
//                         partial context c̄ [H][d] + (m, l) per key split
//   xattn_cproj_kernel    att[row][h·64 + c] = bv + Wv_h · bf16(Σ_splits w_s c̄_s)                                K = d per head
//
// Every row is processed on its own (a beam / prompt-pass row streams its window's E again), so a row's arithmetic
// does not depend on the batch it is in.  Beam search at width > 1 keeps the K/V-cache kernel (attention.hip), which
// shares one K/V stream among the beams of a window.
#include "common.h"
#include <cstdlib>

namespace wx {

typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
constexpr float XA_LOG2E = 1.4426950408889634f;
#ifndef XA_SKIP
#define XA_SKIP 0                                   // diagnostics (tools/mb_xattn.sh), bit mask of the parts left out of the E-stream kernel: 1 = S product,
#endif                                              // 2 = global loads inside the loop, 4 = C phase, 8 = softmax, 16 = image write / staging -> register fill, 32 = the wait for the staged tile
#ifndef XA_CPD
#define XA_CPD 2
#endif
constexpr int XA_KT = 32;                          // keys per tile (one MFMA k-step of the P·E product)

// ---------------------------------------------------------------------------------------------------------
// E-stream attention.  8 waves, tiles of 32 keys, two barriers per tile.
//   E arrives in MFMA-FRAGMENT ORDER (xattn_pack_e_kernel: [window][16-key tile][k-step][lane][8]: lane (g, r) of block
//   (kt, ks) holds E[16 kt + r][32 ks + 8g .. +8], zero beyond the last key), so a wave's 16-B-per-lane load is 1 KiB
//   contiguous AND already the A operand of the S product:
//   S phase   wave w takes the k-steps [w KSW, (w+1) KSW) of the width for both 16-key halves and all query tiles,
//             straight from the registers its loads landed in (B operand = q~, registers, loaded once); the same
//             registers go to the row-major LDS image for the C phase.  Partial S tiles meet in LDS.
//   softmax   wave qt < NQT sums the partials of query tile qt in a fixed order, runs the online softmax (query on the
//             lane, 8 keys per lane) and publishes P (bf16, already in B-operand order) and the rescale factor.
//   C phase   wave w owns the 16-wide column tiles [w D/128, (w+1) D/128) of c: C^T[i][q] += E[key][i]^T . P^T, A operand =
//             E^T via ds_read_b64_tr_b16 from the image (pitch 2D + 32 bytes: conflict-free), B operand = P from LDS.
//   Two register sets: the loads of tile t + 2 are issued as soon as tile t's fragments are in the image; the S product of
//   tile t + 1 (matrix pipe) shares a phase with the image write (LDS) and the softmax (2 waves) of tile t.
// ---------------------------------------------------------------------------------------------------------
template <int D, int NQT>
__global__ __launch_bounds__(512) void xattn_estream_kernel(XAttnArgs a) {
    constexpr int PITCH = D * 2 + 32;
    constexpr int KS_ALL = D / 32;
    constexpr int NSW = (KS_ALL % 8 == 0) ? 8 : 4;             // waves that take part in the S phase (d = 384: 4 x 3 k-steps)
    constexpr int KSW = KS_ALL / NSW;                          // k-steps per S wave
    constexpr int ITW = D / 128;                               // 16-wide column tiles of c per wave
    __shared__ __attribute__((aligned(16))) char img[XA_KT * PITCH];
    constexpr int SB = NSW * NQT * 2 * 64;
    __shared__ __attribute__((aligned(16))) f32x4 sbuf[2 * SB];                    // 2 x [S wave][query tile][key half][lane]
    __shared__ __attribute__((aligned(16))) bf16x8 pbuf[NQT * 64];
    __shared__ float abuf[NQT * 16], lbuf[NQT * 16];
    const int row = blockIdx.x / a.n_split, split = blockIdx.x % a.n_split;
    const int win = row / a.group;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const bool s_role = NSW == 8 || wid < NSW;                 // wave-uniform
    const int kw = s_role ? wid : 0;
    const int T = a.T, H = a.H;

    bf16x8 qf[NQT][KSW];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int q = qt * 16 + r;
        const bf16* qp = a.qt + ((int64_t)row * H + (q < H ? q : H - 1)) * D + kw * KSW * 32 + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
            if (q >= H) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            qf[qt][ks] = v;
        }
    }
    const int nt = a.nt16 / 2;
    const int t0 = (int)((int64_t)split * nt / a.n_split), t1 = (int)((int64_t)(split + 1) * nt / a.n_split);
    // Buffer loads: a tile index past this split's range turns into an offset past the window's records, which the range check
    // drops (zeros, no traffic).  The loads are therefore UNCONDITIONAL, and the compiler's wait counts stay exact: with the loads
    // under a branch it has to drain every outstanding load at the top of a tile (vmcnt(0)), which serialises the two register sets.
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(a.E) + (int64_t)win * a.nt16 * KS_ALL * 512, 0, a.nt16 * KS_ALL * 1024, 0x00020000);
    const uint32_t voff0 = (uint32_t)(kw * KSW * 1024 + lane * 16);
    char* sdst = img + r * PITCH + (kw * KSW * 32 + g * 8) * 2;
    bf16x8 stA[2][KSW], stB[2][KSW];
    auto load_tile = [&](bf16x8 (&st)[2][KSW], int t) {
        const uint32_t vo = (s_role && t < t1) ? voff0 + (uint32_t)t * (2 * KS_ALL * 1024) : 0x80000000u;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
                st[hh][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ers, vo, (hh * KS_ALL + ks) * 1024, 2));
    };
    f32x4 accC[ITW][NQT];
#pragma unroll
    for (int it = 0; it < ITW; ++it)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) accC[it][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                      // softmax waves only: query tile wid

    load_tile(stA, t0);
    __builtin_amdgcn_sched_barrier(0);                         // issue order = consumption order, or the first wait in the loop has to be vmcnt(0)
    load_tile(stB, t0 + 1);
    __builtin_amdgcn_sched_barrier(0);
    // S product of one tile, from the registers its loads landed in, to the partial-score buffer
    auto s_phase = [&](bf16x8 (&st)[2][KSW], int sb) {
#if !(XA_SKIP & 1)
        if (s_role) {
            // the NQT x 2 accumulation chains advance together: consecutive MFMAs are independent
            f32x4 acc[NQT][2];
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) acc[qt][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
                        acc[qt][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[hh][ks], qf[qt][ks], acc[qt][hh], 0, 0, 0);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) sbuf[sb * SB + ((wid * NQT + qt) * 2 + hh) * 64 + lane] = acc[qt][hh];
        }
#endif
    };
    // phase A of tile t: its fragments -> the image (then its registers take the loads of tile t + 2), wave qt < NQT runs the online
    // softmax of query tile qt on the score tiles of buffer sb, and every wave the S product of tile t + 1 into buffer sb ^ 1
    auto phase_a = [&](bf16x8 (&st)[2][KSW], bf16x8 (&stn)[2][KSW], int t, int sb) {
#if !(XA_SKIP & 16)
        if (s_role) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int ks = 0; ks < KSW; ++ks) *reinterpret_cast<bf16x8*>(sdst + hh * 16 * PITCH + ks * 64) = st[hh][ks];
        }
#endif
#if !(XA_SKIP & 2)
        load_tile(st, t + 2);
#endif
#if !(XA_SKIP & 8)
        if (wid < NQT) {              // lane (g, r) holds keys 4g..4g+3 of both 16-key halves for query r
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
            for (int w = 0; w < NSW; ++w) {                     // fixed order: the K slices in ascending order
                s0 += sbuf[sb * SB + ((w * NQT + wid) * 2 + 0) * 64 + lane];
                s1 += sbuf[sb * SB + ((w * NQT + wid) * 2 + 1) * 64 + lane];
            }
            if ((t + 1) * XA_KT > T) {                          // only the last tile of the window can run past the keys
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (t * XA_KT + 4 * g + j >= T) s0[j] = -INFINITY;
                    if (t * XA_KT + 16 + 4 * g + j >= T) s1[j] = -INFINITY;
                }
            }
            float mt = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * XA_LOG2E);
            const float mn2 = mn * XA_LOG2E;
            float p[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = __builtin_amdgcn_exp2f(fmaf(s0[j], XA_LOG2E, -mn2));
                p[4 + j] = __builtin_amdgcn_exp2f(fmaf(s1[j], XA_LOG2E, -mn2));
            }
            l_run = l_run * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
            m_run = mn;
            bf16x8 pv;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = (bf16)p[j];
            pbuf[wid * 64 + lane] = pv;
            if (g == 0) abuf[wid * 16 + r] = alpha;
        }
#endif
        s_phase(stn, sb ^ 1);       // scores of tile t + 1 from the other register set (its loads have had a whole tile to land)
    };
    // C phase of the tile in the image: lane (g, r) receives E[key = 16hh + 4g + j][i = 16 itile + r]
    auto c_phase = [&]() {
#if !(XA_SKIP & 4)
        bf16x8 pf[NQT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            pf[qt] = pbuf[qt * 64 + lane];
            const float alpha = abuf[qt * 16 + r];
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int it = 0; it < ITW; ++it) accC[it][qt] *= alpha;
            }
        }
        const int tq = r >> 2, tp = r & 3;
        const char* p0 = img + (4 * g + tq) * PITCH + (wid * ITW * 16 + 4 * tp) * 2;
        // the transposed reads run XA_CPD column tiles ahead of the MFMAs that consume them (one LDS latency per tile otherwise)
        constexpr int CPD = ITW < XA_CPD ? ITW : XA_CPD;
        i16x4 lo[ITW], hi[ITW];
#pragma unroll
        for (int it = 0; it < CPD; ++it) {
            lo[it] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + it * 32));
            hi[it] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + it * 32 + 16 * PITCH));
        }
#pragma unroll
        for (int it = 0; it < ITW; ++it) {
            if (it + CPD < ITW) {
                lo[it + CPD] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + (it + CPD) * 32));
                hi[it + CPD] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + (it + CPD) * 32 + 16 * PITCH));
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 ef = __builtin_bit_cast(bf16x8, (i16x8)__builtin_shufflevector(lo[it], hi[it], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) accC[it][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, pf[qt], accC[it][qt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };
    // Two barriers per tile.  Phase A(t): image write of tile t (LDS), the S product of tile t + 1 from the other register set
    // (matrix pipe; its score tiles go to the other half of the double-buffered score buffer) and, on NQT waves, the softmax of
    // tile t on the scores the previous phase A left.  Phase B(t): the C product of tile t (image + P).  Past the last tile the
    // loads return zeros and the S product is idle work that nobody reads.  (The softmax beside the C product instead, with P
    // double-buffered: 121 instead of 106 us at one key split.)
    s_phase(stA, 0);
    __syncthreads();
    int t = t0;
    for (; t + 1 < t1; t += 2) {        // straight-line body: every path through the loop issues the same loads, the wait counts stay exact
        phase_a(stA, stB, t, 0);
        __syncthreads();
        c_phase();
        __syncthreads();
        phase_a(stB, stA, t + 1, 1);
        __syncthreads();
        c_phase();
        __syncthreads();
    }
    if (t < t1) {
        phase_a(stA, stB, t, 0);
        __syncthreads();
        c_phase();
    }
    // ---- partial context, normalised by this split's own sum; lane holds c[q = r][i = 16 itile + 4g .. +3] -----------------
    if (wid < NQT) {
        float l = l_run;
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int q = wid * 16 + r;
        if (g == 0) {
            lbuf[wid * 16 + r] = l;
            if (q < H) {
                float* mp = a.ml + (((int64_t)row * a.n_split + split) * H + q) * 2;
                mp[0] = m_run; mp[1] = l;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const float l = lbuf[qt * 16 + r];
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int q = qt * 16 + r;
        if (q < H) {
            bf16* cp = a.cpart + (((int64_t)row * a.n_split + split) * H + q) * D + wid * ITW * 16 + 4 * g;
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)(accC[it][qt][j] * inv);
                *reinterpret_cast<bf16x4*>(cp + it * 16) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// E-stream attention, round 3: NO LDS image.  The image (82 KB per 32-key tile at d = 1280) cost its store -- LDS stores move at
// ~79 B/clk per CU, a third of the read rate: 0.43 us of a 1.7 us tile -- its transposed read-back and a barrier.  The tile a wave
// needs TRANSPOSED for the context product is exactly the tile it holds in registers for the score product (wave w: k-steps
// [w KSW, (w+1) KSW) of the width = the 16-wide column tiles [2 w KSW, 2 (w+1) KSW) of c), and the matrix pipe transposes for
// free: with E as the A operand (row = key, k = feature) and a 0/1 selection matrix as B,
//     D[key][n] = sum_f E[key][f] * [f == 16 c + n]  =  E[key][16 c + n]          (exact: one non-zero term)
// comes back with the KEY index in the lane's registers (lane (g, r): keys 4g .. 4g+3, feature r) -- the A operand of the context
// product after a bf16 pack.  60 MFMAs per wave and tile instead of 40 (the pipe was 12 % busy), no image store, no transposed
// LDS reads, one barrier per tile, and the context product depends on nothing but the wave's own registers and P.
//   HBM -> LDS by DMA (global_load_lds, no registers: a third register set does not fit next to the accumulators), each wave into
//   its own 1-KiB-per-fragment slice of a staging tile; LDS -> registers by ds_read_b128 when the previous tile's registers die.
//   tile t:   softmax(t) on wave qt < NQT (scores the phase before left) -> P, rescale factor         |  S product of tile t + 1
//             barrier
//             context product of tile t; its registers <- staging (tile t + 2); DMA of tile t + 3 into the staging slice
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) void xa_gbl_void;
typedef __attribute__((address_space(3))) void xa_lds_void;
#define XA_WAIT_VM0() __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define XA_WAIT_LGKM0() __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define XA_BARRIER() __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")      /* LDS traffic only: the DMA stays in flight */

template <int D, int NQT>
__global__ __launch_bounds__(512) void xattn_estream_reg_kernel(XAttnArgs a) {
    constexpr int KS_ALL = D / 32;
    constexpr int NSW = (KS_ALL % 8 == 0) ? 8 : 4;             // waves that hold E (d = 384: 4 x 3 k-steps); the others only meet the barriers
    constexpr int KSW = KS_ALL / NSW;                          // k-steps per wave = half its column tiles
    constexpr int SB = NSW * NQT * 2 * 64;
    __shared__ __attribute__((aligned(16))) char lds[2 * KS_ALL * 1024 + 2 * SB * 16 + 2 * NQT * 64 * 16 + 2 * NQT * 16 * 4 + NQT * 16 * 4];
    char* const stage = lds;                                                       // [S wave][key half][k-step][64 lanes][16 B]
    f32x4* const sbuf = reinterpret_cast<f32x4*>(lds + 2 * KS_ALL * 1024);         // 2 x [S wave][query tile][key half][lane]
    bf16x8* const pbuf = reinterpret_cast<bf16x8*>(lds + 2 * KS_ALL * 1024 + 2 * SB * 16);      // 2 x [query tile][lane]
    float* const abuf = reinterpret_cast<float*>(lds + 2 * KS_ALL * 1024 + 2 * SB * 16 + 2 * NQT * 64 * 16);   // 2 x [query tile][16]
    float* const lbuf = abuf + 2 * NQT * 16;
    const int row = blockIdx.x / a.n_split, split = blockIdx.x % a.n_split;
    const int win = row / a.group;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const bool s_role = NSW == 8 || wid < NSW;                 // wave-uniform
    const int kw = s_role ? wid : 0;
    const int T = a.T, H = a.H;

    bf16x8 qf[NQT][KSW];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        const int q = qt * 16 + r;
        const bf16* qp = a.qt + ((int64_t)row * H + (q < H ? q : H - 1)) * D + kw * KSW * 32 + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
            if (q >= H) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            qf[qt][ks] = v;
        }
    }
    // selection operands: as B (column n = r, k = 8g + j): 1 where 8g + j == 16 c + r
    bf16x8 sel[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) sel[c][j] = (8 * g + j == 16 * c + r) ? (bf16)1.0f : (bf16)0.0f;
    XA_WAIT_VM0();                                             // the ordinary loads above are done before the first DMA is issued

    const int nt = a.nt16 / 2;
    const int t0 = (int)((int64_t)split * nt / a.n_split), t1 = (int)((int64_t)(split + 1) * nt / a.n_split);
    const char* const ebase = reinterpret_cast<const char*>(a.E) + ((int64_t)win * a.nt16 * KS_ALL + kw * KSW) * 1024 + lane * 16;
    char* const myst = stage + kw * 2 * KSW * 1024;            // this wave's slice of the staging tile
    auto dma_tile = [&](int t) {                               // wave-uniform condition; 2 KSW loads of 1 KiB, no registers
        if (s_role && t < t1) {
            const char* src = ebase + (int64_t)t * (2 * KS_ALL * 1024);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int ks = 0; ks < KSW; ++ks)
                    __builtin_amdgcn_global_load_lds((xa_gbl_void*)(src + (hh * KS_ALL + ks) * 1024), (xa_lds_void*)(myst + (hh * KSW + ks) * 1024), 16, 0, 2);
        }
    };
    auto fill_set = [&](bf16x8 (&st)[2][KSW]) {               // staging slice -> registers (past the split's end: stale data nobody uses)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) st[hh][ks] = *reinterpret_cast<const bf16x8*>(myst + (hh * KSW + ks) * 1024 + lane * 16);
    };
    // context accumulators: query tile 0 (heads 0..15) one per column tile; the heads past 16 (at most 4: d = 1280 has 20) PACKED,
    // four column tiles per accumulator -- column tile it = 4u + s lives in columns 4s..4s+3 of accX[u], put there by a P operand
    // that is zero outside those columns (a full second query tile would be 40 registers of which 30 hold padding)
    constexpr int NX = NQT == 2 ? (2 * KSW + 3) / 4 : 0;
    f32x4 accC[2 * KSW], accX[NX > 0 ? NX : 1];
#pragma unroll
    for (int it = 0; it < 2 * KSW; ++it) accC[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NX; ++u) accX[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                      // softmax waves only: query tile wid

    auto s_phase = [&](bf16x8 (&st)[2][KSW], int sb) {
        if (s_role && !(XA_SKIP & 1)) {
            f32x4 acc[NQT][2];
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) acc[qt][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) acc[qt][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[hh][ks], qf[qt][ks], acc[qt][hh], 0, 0, 0);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) sbuf[sb * SB + ((wid * NQT + qt) * 2 + hh) * 64 + lane] = acc[qt][hh];
        }
    };
    auto softmax = [&](int t, int sb) {
        if (wid < NQT && !(XA_SKIP & 8)) {              // lane (g, r) holds keys 4g..4g+3 of both 16-key halves for query r
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
            for (int w = 0; w < NSW; ++w) {                     // fixed order: the K slices in ascending order
                s0 += sbuf[sb * SB + ((w * NQT + wid) * 2 + 0) * 64 + lane];
                s1 += sbuf[sb * SB + ((w * NQT + wid) * 2 + 1) * 64 + lane];
            }
            if ((t + 1) * XA_KT > T) {                          // only the last tile of the window can run past the keys
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (t * XA_KT + 4 * g + j >= T) s0[j] = -INFINITY;
                    if (t * XA_KT + 16 + 4 * g + j >= T) s1[j] = -INFINITY;
                }
            }
            float mt = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * XA_LOG2E);
            const float mn2 = mn * XA_LOG2E;
            float p[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = __builtin_amdgcn_exp2f(fmaf(s0[j], XA_LOG2E, -mn2));
                p[4 + j] = __builtin_amdgcn_exp2f(fmaf(s1[j], XA_LOG2E, -mn2));
            }
            l_run = l_run * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
            m_run = mn;
            bf16x8 pv;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = (bf16)p[j];
            pbuf[(sb * NQT + wid) * 64 + lane] = pv;
            if (g == 0) abuf[(sb * NQT + wid) * 16 + r] = alpha;
        }
    };
    // context product of one tile from the registers it was read into
    auto c_phase = [&](bf16x8 (&st)[2][KSW], int sb) {
        if (s_role && !(XA_SKIP & 4)) {
            const bf16x8 pf = pbuf[(sb * NQT + 0) * 64 + lane];
            {
                const float alpha = abuf[(sb * NQT + 0) * 16 + r];
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                    for (int it = 0; it < 2 * KSW; ++it) accC[it] *= alpha;
                }
            }
            bf16x8 px = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};        // P of head 16 + (r & 3) for this lane's keys
            if (NQT == 2) {
                px = pbuf[(sb * NQT + 1) * 64 + g * 16 + (r & 3)];
                const float alpha = abuf[(sb * NQT + 1) * 16 + (r & 3)];
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                    for (int u = 0; u < NX; ++u) accX[u] *= alpha;
                }
            }
            const bf16x8 none = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            // the transposing MFMAs run one column tile ahead of the context MFMAs that consume them
            f32x4 d0[2], d1[2];
            d0[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[0][0], sel[0], zero, 0, 0, 0);       // E[keys 4g..4g+3][16 c + r], half 0
            d1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[1][0], sel[0], zero, 0, 0, 0);       // ... half 1
#pragma unroll
            for (int it = 0; it < 2 * KSW; ++it) {
                if (it + 1 < 2 * KSW) {
                    d0[(it + 1) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[0][(it + 1) >> 1], sel[(it + 1) & 1], zero, 0, 0, 0);
                    d1[(it + 1) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[1][(it + 1) >> 1], sel[(it + 1) & 1], zero, 0, 0, 0);
                }
                bf16x8 ef;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ef[j] = (bf16)d0[it & 1][j]; ef[4 + j] = (bf16)d1[it & 1][j]; }
                accC[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, pf, accC[it], 0, 0, 0);
                if (NQT == 2) {
                    const bf16x8 pxs = (r >> 2) == (it & 3) ? px : none;
                    accX[it >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, pxs, accX[it >> 2], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // One barrier interval: the softmax of tile t + 1 FIRST, on its NQT waves (LDS + VALU work that runs under the matrix work of
    // the waves they share a SIMD with), the context product of tile t from register set X, X <- staging (tile t + 2), the DMA
    // of tile t + 3, and the S product of tile t + 2 from the registers just filled.  Scores run two tiles ahead of the context
    // product, score tiles and P are double-buffered, so one barrier per tile separates every producer from its consumer.
    auto tile = [&](bf16x8 (&X)[2][KSW], int t) {
        const int sb = (t - t0) & 1;
        if (t + 1 < t1) softmax(t + 1, sb ^ 1);       // (wave-uniform; the running maximum and sum must not see tiles of another split)
        c_phase(X, sb);
        if (!(XA_SKIP & 32)) XA_WAIT_VM0();                // tile t + 2 has landed in this wave's staging slice
        if (!(XA_SKIP & 16)) fill_set(X);
        XA_WAIT_LGKM0();              // ... and is in registers: the slice may be overwritten
        if (!(XA_SKIP & 2)) dma_tile(t + 3);
        s_phase(X, sb);               // scores of tile t + 2 into the buffer softmax(t) read in the previous interval
        XA_BARRIER();
    };
    bf16x8 stA[2][KSW], stB[2][KSW];
    dma_tile(t0);
    XA_WAIT_VM0();
    fill_set(stA);
    XA_WAIT_LGKM0();
    dma_tile(t0 + 1);
    s_phase(stA, 0);
    XA_WAIT_VM0();
    fill_set(stB);
    XA_WAIT_LGKM0();
    dma_tile(t0 + 2);
    s_phase(stB, 1);
    XA_BARRIER();
    softmax(t0, 0);
    XA_BARRIER();
    int t = t0;
    for (; t + 1 < t1; t += 2) {
        tile(stA, t);
        tile(stB, t + 1);
    }
    if (t < t1) tile(stA, t);
    XA_WAIT_VM0();
    // ---- partial context, normalised by this split's own sum; lane holds c[q = r][i = 16 itile + 4g .. +3] -----------------
    if (wid < NQT) {
        float l = l_run;
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int q = wid * 16 + r;
        if (g == 0) {
            lbuf[wid * 16 + r] = l;
            if (q < H) {
                float* mp = a.ml + (((int64_t)row * a.n_split + split) * H + q) * 2;
                mp[0] = m_run; mp[1] = l;
            }
        }
    }
    __syncthreads();
    if (s_role) {
        {
            const float l = lbuf[r];
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            if (r < H) {
                bf16* cp = a.cpart + (((int64_t)row * a.n_split + split) * H + r) * D + kw * KSW * 32 + 4 * g;
#pragma unroll
                for (int it = 0; it < 2 * KSW; ++it) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (bf16)(accC[it][j] * inv);
                    *reinterpret_cast<bf16x4*>(cp + it * 16) = o;
                }
            }
        }
        if (NQT == 2) {              // packed heads: lane (g, r) holds head 16 + (r & 3), column tile 4u + (r >> 2), features 4g .. 4g+3
            const int q = 16 + (r & 3);
            const float l = lbuf[q];
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            if (q < H) {
                bf16* cp = a.cpart + (((int64_t)row * a.n_split + split) * H + q) * D + kw * KSW * 32 + 4 * g;
#pragma unroll
                for (int u = 0; u < NX; ++u) {
                    const int it = 4 * u + (r >> 2);
                    if (it < 2 * KSW) {
                        bf16x4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (bf16)(accX[u][j] * inv);
                        *reinterpret_cast<bf16x4*>(cp + it * 16) = o;
                    }
                }
            }
        }
    }
}

// encoder output [windows][T][D] row-major -> MFMA-fragment order [windows][nt16][D/32][64 lanes][8], zero beyond key T - 1
__global__ __launch_bounds__(256) void xattn_pack_e_kernel(const bf16* __restrict__ E, bf16* __restrict__ Ef, int T, int D, int nt16, int64_t pieces) {
    const int ks_all = D / 32;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pieces; p += (int64_t)gridDim.x * 256) {
        const int lane = (int)(p & 63);
        const int64_t blk = p >> 6;
        const int ks = (int)(blk % ks_all);
        const int64_t wk = blk / ks_all;
        const int kt = (int)(wk % nt16);
        const int64_t win = wk / nt16;
        const int key = kt * 16 + (lane & 15);
        bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (key < T) v = *reinterpret_cast<const bf16x8*>(E + (win * T + key) * D + ks * 32 + (lane >> 4) * 8);
        *reinterpret_cast<bf16x8*>(Ef + p * 8) = v;
    }
}
int launch_xattn_pack_e(const void* E, void* Ef, int windows, int T, int D, int nt16, hipStream_t st) {
    if (!E || !Ef || windows <= 0 || T <= 0 || D % 32 || nt16 * 16 < T || (nt16 & 1)) return -1;
    const int64_t pieces = (int64_t)windows * nt16 * (D / 32) * 64;
    int64_t blocks = (pieces + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(xattn_pack_e_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)E, (bf16*)Ef, T, D, nt16, pieces);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------
// q̃: one workgroup (4 waves) per (head, 256 output columns, 32 rows).  The head's 64 query columns of the rows are assembled
// from the query projection's split-K tiles (+ bias, x s, bf16: the rounding point of a stored query) into LDS; each wave then
// runs 4 column units x MT row tiles x 2 k-steps.  Weights: Wk_pk = the cross-attention key weight, transposed and cut per head
// into MFMA-fragment order [head][d / 16 column tiles][2 k-steps][64 lanes][8] (lane (g, r) of block (h, it, ks) holds
// Wk[64 h + 32 ks + 8 g .. +8][16 it + r]): every wave load is 1 KiB contiguous.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int MS>                                     // MS: split-K tiles fetched per element (>= q_nsplit)
__global__ __launch_bounds__(256) void xattn_qtilde_kernel(XQTildeArgs a) {
    constexpr int ROWB = 64 * 2 + 32;                         // LDS row: 64 bf16 + pad
    __shared__ __attribute__((aligned(16))) char xs[MT * 16 * ROWB];
    const int h = blockIdx.x, n0 = blockIdx.y * 256, m0 = blockIdx.z * 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int d = a.d, M = a.rows, n_it = d / 16;
    // weights first: they depend on nothing
    bf16x8 w[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int it = n0 / 16 + wid * 4 + u; if (it > n_it - 1) it = n_it - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w[u][ks] = *reinterpret_cast<const bf16x8*>(a.Wk_pk + (((int64_t)h * n_it + it) * 2 + ks) * 512 + lane * 8);
    }
    // the head's query slice: MT*16 rows x 64 columns, 4 columns (one f32x4 per partial tile) per item
    // (all tiles of an element in flight at once: a runtime-bounded loop of loads is a chain of round trips; tiles past
    // q_nsplit are clamped to the last one and not added)
#pragma unroll
    for (int it0 = 0; it0 < MT * 16 * 16; it0 += 256) {
        const int it = it0 + tid;
        const int row = it >> 4, c4 = (it & 15) * 4;
        const int rr = m0 + row < M ? m0 + row : M - 1;
        f32x4 pt[MS];
#pragma unroll
        for (int s = 0; s < MS; ++s)
            pt[s] = *reinterpret_cast<const f32x4*>(a.q_partial + ((int64_t)(s < a.q_nsplit ? s : a.q_nsplit - 1) * M + rr) * d + h * 64 + c4);
        f32x4 v = *reinterpret_cast<const f32x4*>(a.q_bias + h * 64 + c4);
#pragma unroll
        for (int s = 0; s < MS; ++s) if (s < a.q_nsplit) v += pt[s];
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16)(v[j] * a.scale);
        *reinterpret_cast<bf16x4*>(xs + row * ROWB + c4 * 2) = o;
    }
    __syncthreads();
    f32x4 acc[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xf[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xs + (i * 16 + r) * ROWB + ks * 64 + g * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][ks], xf[i], acc[u][i], 0, 0, 0);
    }
    // lane holds q̃[row = m0 + 16 i + r][col = n0 + 16 (4 wid + u) + 4g .. +3]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int col = n0 + (wid * 4 + u) * 16 + 4 * g;
        if (col >= d) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = m0 + i * 16 + r;
            if (row >= M) continue;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[u][i][j] * a.scale);
            *reinterpret_cast<bf16x4*>(a.qt + ((int64_t)row * a.H + h) * d + col) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Output side: one workgroup (4 waves) per (head, 8 rows, 32 of the head's 64 output columns).  The rows' context vectors are
// combined from the key splits (w_s = l_s e^(m_s - M) / Σ, fp32) and rounded to bf16 into LDS; waves = 2 column units x 2 K
// halves; the halves meet in LDS.  Wv_pk: the value weight [d][d] in MFMA-fragment order [d / 16][d / 32][64 lanes][8].
// ---------------------------------------------------------------------------------------------------------
template <int D, int MS>                                      // MS: key splits fetched per chunk (>= n_split)
__global__ __launch_bounds__(256) void xattn_cproj_kernel(XCProjArgs a) {
    constexpr int PITCH = D * 2 + 32, KH = D / 2, KSTEPS = KH / 32;
    constexpr int CR = D / 8, CPT = (CR + 31) / 32;           // 16-B chunks per row; per thread (32 threads per row)
    __shared__ __attribute__((aligned(16))) char cs[8 * PITCH];
    __shared__ __attribute__((aligned(16))) f32x4 red[2 * 64];
    const int h = blockIdx.x, m0 = blockIdx.y * 8, half = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int unit = wid & 1, kh = wid >> 1;
    const int M = a.rows, H = a.H, S = a.n_split;
    // every load that depends on nothing is issued before the first wait: weights, split statistics, split contexts
    bf16x8 w[KSTEPS];
    {
        const bf16* wp = a.Wv_pk + (((int64_t)(h * 4 + half * 2 + unit) * (D / 32)) + kh * KSTEPS) * 512 + lane * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
    }
    const int lr = tid >> 5, c0 = tid & 31;
    const int row = m0 + lr < M ? m0 + lr : M - 1;
    f32x2 mls[MS];
    bf16x8 x[CPT][MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        const int sc = s < S ? s : S - 1;
        mls[s] = *reinterpret_cast<const f32x2*>(a.ml + (((int64_t)row * S + sc) * H + h) * 2);
#pragma unroll
        for (int p = 0; p < CPT; ++p) {
            const int c = c0 + 32 * p < CR ? c0 + 32 * p : CR - 1;
            x[p][s] = *reinterpret_cast<const bf16x8*>(a.cpart + (((int64_t)row * S + sc) * H + h) * D + c * 8);
        }
    }
    // combine weights w_s = l_s e^(m_s - M) / sum (every thread of a row computes the same numbers)
    float ws[MS], mx = -INFINITY, tot = 0.f;
#pragma unroll
    for (int s = 0; s < MS; ++s) if (s < S && mls[s][1] > 0.f) mx = fmaxf(mx, mls[s][0]);
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        ws[s] = (s < S && mls[s][1] > 0.f) ? mls[s][1] * __builtin_amdgcn_exp2f((mls[s][0] - mx) * XA_LOG2E) : 0.f;
        tot += ws[s];
    }
    const float inv = 1.0f / tot;
#pragma unroll
    for (int p = 0; p < CPT; ++p) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
        for (int s = 0; s < MS; ++s) {
            const float f = ws[s] * inv;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(f, (float)x[p][s][j], v[j]);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
        if (c0 + 32 * p < CR) *reinterpret_cast<bf16x8*>(cs + lr * PITCH + (c0 + 32 * p) * 16) = o;
    }
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xp = cs + (r & 7) * PITCH + (kh * KH + g * 8) * 2;          // B operand columns 8..15 repeat rows 0..7 and are dropped
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp + ks * 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], xf, acc, 0, 0, 0);
    }
    if (kh == 1) red[unit * 64 + lane] = acc;
    __syncthreads();
    if (kh == 0) {
        const f32x4 o2 = red[unit * 64 + lane];
        const int orow = m0 + r, col = h * 64 + half * 32 + unit * 16 + 4 * g;       // lane holds att[row][col .. col+3]
        if (r < 8 && orow < M) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.bv + col);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)((acc[j] + o2[j]) + b[j]);
            *reinterpret_cast<bf16x4*>(a.out + (int64_t)orow * a.ldo + col) = o;
        }
    }
}

template <int MS>
static void launch_qtilde_mt(const XQTildeArgs& a, dim3 grid, hipStream_t st) {
    if (a.rows <= 16) hipLaunchKernelGGL((xattn_qtilde_kernel<1, MS>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((xattn_qtilde_kernel<2, MS>), grid, dim3(256), 0, st, a);
}
int launch_xattn_qtilde(const XQTildeArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.rows > DG_MAX_ROWS || a.d % 128 || a.H * 64 != a.d || !a.q_partial || !a.q_bias || !a.Wk_pk || !a.qt || a.q_nsplit < 1 || a.q_nsplit > 16) return -1;
    const dim3 grid(a.H, (a.d + 255) / 256, (a.rows + 31) / 32);
    if (a.q_nsplit <= 4) launch_qtilde_mt<4>(a, grid, st);
    else if (a.q_nsplit <= 10) launch_qtilde_mt<10>(a, grid, st);
    else launch_qtilde_mt<16>(a, grid, st);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_xattn_estream(const XAttnArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.group < 1 || a.n_split < 1 || a.n_split > 8 || a.T <= 0 || !a.qt || !a.E || !a.cpart || !a.ml || a.H * 64 != a.d) return -1;
    if (a.nt16 * 16 < a.T || (a.nt16 & 1) || a.nt16 / 2 < a.n_split) return -1;
    const dim3 grid(a.rows * a.n_split), block(512);
    const bool image = getenv("WX_XATTN_IMAGE") != nullptr;       // A/B: round 2's kernel (E tile through an LDS image); read per launch -- a captured graph keeps what it captured
    if (image) {
        switch (a.d) {
            case 384: hipLaunchKernelGGL((xattn_estream_kernel<384, 1>), grid, block, 0, st, a); break;
            case 512: hipLaunchKernelGGL((xattn_estream_kernel<512, 1>), grid, block, 0, st, a); break;
            case 768: hipLaunchKernelGGL((xattn_estream_kernel<768, 1>), grid, block, 0, st, a); break;
            case 1024: hipLaunchKernelGGL((xattn_estream_kernel<1024, 1>), grid, block, 0, st, a); break;
            case 1280: hipLaunchKernelGGL((xattn_estream_kernel<1280, 2>), grid, block, 0, st, a); break;
            case 128: hipLaunchKernelGGL((xattn_estream_kernel<128, 1>), grid, block, 0, st, a); break;
            case 256: hipLaunchKernelGGL((xattn_estream_kernel<256, 1>), grid, block, 0, st, a); break;
            default: return -1;
        }
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    switch (a.d) {
        case 384: hipLaunchKernelGGL((xattn_estream_reg_kernel<384, 1>), grid, block, 0, st, a); break;
        case 512: hipLaunchKernelGGL((xattn_estream_reg_kernel<512, 1>), grid, block, 0, st, a); break;
        case 768: hipLaunchKernelGGL((xattn_estream_reg_kernel<768, 1>), grid, block, 0, st, a); break;
        case 1024: hipLaunchKernelGGL((xattn_estream_reg_kernel<1024, 1>), grid, block, 0, st, a); break;
        case 1280: hipLaunchKernelGGL((xattn_estream_reg_kernel<1280, 2>), grid, block, 0, st, a); break;
        case 128: hipLaunchKernelGGL((xattn_estream_reg_kernel<128, 1>), grid, block, 0, st, a); break;
        case 256: hipLaunchKernelGGL((xattn_estream_reg_kernel<256, 1>), grid, block, 0, st, a); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int MS>
static bool launch_cproj_d(const XCProjArgs& a, dim3 grid, hipStream_t st) {
    switch (a.d) {
        case 128: hipLaunchKernelGGL((xattn_cproj_kernel<128, MS>), grid, dim3(256), 0, st, a); return true;
        case 256: hipLaunchKernelGGL((xattn_cproj_kernel<256, MS>), grid, dim3(256), 0, st, a); return true;
        case 384: hipLaunchKernelGGL((xattn_cproj_kernel<384, MS>), grid, dim3(256), 0, st, a); return true;
        case 512: hipLaunchKernelGGL((xattn_cproj_kernel<512, MS>), grid, dim3(256), 0, st, a); return true;
        case 768: hipLaunchKernelGGL((xattn_cproj_kernel<768, MS>), grid, dim3(256), 0, st, a); return true;
        case 1024: hipLaunchKernelGGL((xattn_cproj_kernel<1024, MS>), grid, dim3(256), 0, st, a); return true;
        case 1280: hipLaunchKernelGGL((xattn_cproj_kernel<1280, MS>), grid, dim3(256), 0, st, a); return true;
        default: return false;
    }
}
int launch_xattn_cproj(const XCProjArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.n_split < 1 || a.n_split > 8 || !a.cpart || !a.ml || !a.Wv_pk || !a.bv || !a.out || a.H * 64 != a.d || a.ldo % 4) return -1;
    const dim3 grid(a.H, (a.rows + 7) / 8, 2);
    const bool ok = a.n_split <= 4 ? launch_cproj_d<4>(a, grid, st) : launch_cproj_d<8>(a, grid, st);
    if (!ok) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
