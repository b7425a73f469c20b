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

namespace wx {

typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
constexpr float XA_LOG2E = 1.4426950408889634f;
#ifndef XA_VARIANT
#define XA_VARIANT 0                                // diagnostics (tools/mb_xattn.sh): 1 = staging only, no S / softmax / C phases
#endif
constexpr int XA_KT = 32;                          // keys per tile (one MFMA k-step of the P·E product)

// ---------------------------------------------------------------------------------------------------------
// E-stream attention.  8 waves.
//   S phase   wave (qtile, kpart): Sᵀ[32 keys][16 queries] partial over its d / KSPLIT slice of the width (both 16-key
//             halves, so no two waves hold the same q̃ fragments), A operand = E rows (ds_read_b128), B operand = q̃
//             (registers, loaded once); the partial tiles meet in LDS.
//   softmax   every wave redundantly (same values, same order): query on the lane, 8 keys per lane and query tile.
//   C phase   wave w owns the 16-wide column tiles [w·D/128, (w+1)·D/128) of c: Cᵀ[i][q] += E[key][i]ᵀ · Pᵀ, A operand =
//             Eᵀ via ds_read_b64_tr_b16 from the same image, B operand = the S accumulators converted in place.
// LDS image: 32 rows of D bf16 at a pitch of 2D + 32 bytes (conflict-free for both kinds of read).
// ---------------------------------------------------------------------------------------------------------
template <int D, int NQT>
__global__ __launch_bounds__(512) void xattn_estream_kernel(XAttnArgs a) {
    constexpr int PITCH = D * 2 + 32;
    constexpr int KSPLIT = ((D / 32) % (8 / NQT) == 0) ? 8 / NQT : 4 / NQT;      // waves past NQT * KSPLIT sit the S phase out
    constexpr int KSLICE = D / KSPLIT, KSTEPS = KSLICE / 32;
    constexpr int CPT = D / 128;                   // 16-B chunks staged per thread and tile (32 rows x D/8 chunks / 512)
    constexpr int ITW = D / 128;                   // 16-wide column tiles of c per wave
    __shared__ __attribute__((aligned(16))) char img[XA_KT * PITCH];
    __shared__ __attribute__((aligned(16))) char qimg[(D / 64 + 1) * PITCH];          // H + 1 rows (H = D / 64)
    __shared__ __attribute__((aligned(16))) f32x4 sbuf[KSPLIT * NQT * 2 * 64];     // [kpart][qtile][key half][lane]
    const int row = blockIdx.x / a.n_split, split = blockIdx.x % a.n_split;
    const int win = row / a.group;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int qtile = wid % NQT, kpart = wid / NQT;
    const bool s_role = kpart < KSPLIT;
    const int T = a.T, H = a.H;

    // q~ of this row: [H][D] into LDS once (+ one zero row that the padding queries H .. 16 NQT - 1 read); the S phase
    // takes its B fragments from there (in registers they cost 40 VGPRs at d = 1280, which the two staging sets need)
    for (int id = tid; id < (H + 1) * (D / 8); id += 512) {
        const int q = id / (D / 8), c = id % (D / 8);
        bf16x8 v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (q < H) v = *reinterpret_cast<const bf16x8*>(a.qt + ((int64_t)row * H + q) * D + c * 8);
        *reinterpret_cast<bf16x8*>(qimg + q * PITCH + c * 16) = v;
    }
    const int qrow = qtile * 16 + r < H ? qtile * 16 + r : H;
    const int nt = (T + XA_KT - 1) / XA_KT;
    const int t0 = (int)((int64_t)split * nt / a.n_split), t1 = (int)((int64_t)(split + 1) * nt / a.n_split);
    // staging map: thread -> row tid >> 4 of the tile, 16-B chunks (tid & 15) + 16 k: one address per thread and tile, the
    // rest are compile-time offsets.  TWO register sets: the loads of tile t + 2 are issued as soon as tile t has gone to
    // LDS, so a CU always has one to two tiles (80-160 KB at d = 1280) in flight instead of an empty queue after every tile.
    const int srow = tid >> 4, scs = tid & 15;
    const bf16* Eb = a.E + (int64_t)win * T * D + scs * 8;
    char* sdst = img + srow * PITCH + scs * 16;
    bf16x8 stA[CPT], stB[CPT];
    auto load_tile = [&](bf16x8 (&st)[CPT], int t) {
        int key = t * XA_KT + srow; if (key > T - 1) key = T - 1;
        const bf16* src = Eb + (int64_t)key * D;
#pragma unroll
        for (int p = 0; p < CPT; ++p) st[p] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src + p * 128));
    };
    f32x4 accC[ITW][NQT];
#pragma unroll
    for (int it = 0; it < ITW; ++it)
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) accC[it][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[NQT], l_run[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    if (t0 < t1) load_tile(stA, t0);
    if (t0 + 1 < t1) load_tile(stB, t0 + 1);
    auto tile_body = [&](bf16x8 (&st)[CPT], int t) {
#pragma unroll
        for (int p = 0; p < CPT; ++p) *reinterpret_cast<bf16x8*>(sdst + p * 256) = st[p];
        __syncthreads();
        if (t + 2 < t1) load_tile(st, t + 2);
#if XA_VARIANT == 1
        __syncthreads();
        return;
#endif
        // ---- S phase -----------------------------------------------------------------------------------------
        if (s_role) {                                               // wave-uniform
            f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
            const char* ap = img + r * PITCH + (kpart * KSLICE + g * 8) * 2;
            const char* qp = qimg + qrow * PITCH + (kpart * KSLICE + g * 8) * 2;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8 e0 = *reinterpret_cast<const bf16x8*>(ap + ks * 64);
                const bf16x8 e1 = *reinterpret_cast<const bf16x8*>(ap + ks * 64 + 16 * PITCH);
                const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qp + ks * 64);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e0, qf, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(e1, qf, acc1, 0, 0, 0);
            }
            sbuf[((kpart * NQT + qtile) * 2 + 0) * 64 + lane] = acc0;
            sbuf[((kpart * NQT + qtile) * 2 + 1) * 64 + lane] = acc1;
        }
        __syncthreads();
        // ---- online softmax: lane (g, r) holds keys 4g..4g+3 of both 16-key halves for query r of each query tile --------
        bf16x8 pf[NQT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
            for (int kp = 0; kp < KSPLIT; ++kp) {               // fixed order: the K slices in ascending order
                s0 += sbuf[((kp * NQT + qt) * 2 + 0) * 64 + lane];
                s1 += sbuf[((kp * NQT + qt) * 2 + 1) * 64 + lane];
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
            const float mn = fmaxf(m_run[qt], mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - mn) * XA_LOG2E);
            const float mn2 = mn * XA_LOG2E;
            float p[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = __builtin_amdgcn_exp2f(fmaf(s0[j], XA_LOG2E, -mn2));
                p[4 + j] = __builtin_amdgcn_exp2f(fmaf(s1[j], XA_LOG2E, -mn2));
            }
            l_run[qt] = l_run[qt] * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
            m_run[qt] = mn;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int it = 0; it < ITW; ++it) accC[it][qt] *= alpha;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[qt][j] = (bf16)p[j];
        }
        // ---- C phase: lane (g, r) receives E[key = 16hh + 4g + j][i = 16 itile + r] ------------------------------------
        {
            const int tq = r >> 2, tp = r & 3;
            const char* p0 = img + (4 * g + tq) * PITCH + (wid * ITW * 16 + 4 * tp) * 2;
#pragma unroll
            for (int it = 0; it < ITW; ++it) {
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + it * 32));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + it * 32 + 16 * PITCH));
                const bf16x8 ef = __builtin_bit_cast(bf16x8, (i16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) accC[it][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, pf[qt], accC[it][qt], 0, 0, 0);
            }
        }
        __syncthreads();            // the image is rewritten at the top of the next tile
    };
    for (int t = t0; t < t1; t += 2) {
        tile_body(stA, t);
        if (t + 1 < t1) tile_body(stB, t + 1);
    }
    // ---- partial context, normalised by this split's own sum; lane holds c[q = r][i = 16 itile + 4g .. +3] -----------------
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
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
            if (wid == 0 && g == 0) {
                float* mp = a.ml + (((int64_t)row * a.n_split + split) * H + q) * 2;
                mp[0] = m_run[qt]; mp[1] = l;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// q̃: one workgroup (4 waves) per (head, 256 output columns).  The head's 64 query columns of every row are assembled from
// the query projection's split-K tiles (+ bias, x s, bf16: the rounding point of a stored query) into LDS; each wave then
// runs 4 column units x MT row tiles x 2 k-steps.  Weights: WkT [d][d] = the cross-attention key weight transposed, so that
// the 64 inputs of a head are contiguous in a row.
// ---------------------------------------------------------------------------------------------------------
template <int MT, int MS>                                     // MS: split-K tiles fetched per element (>= q_nsplit)
__global__ __launch_bounds__(256) void xattn_qtilde_kernel(XQTildeArgs a) {
    constexpr int ROWB = 64 * 2 + 32;                         // LDS row: 64 bf16 + pad
    __shared__ __attribute__((aligned(16))) char xs[MT * 16 * ROWB];
    const int h = blockIdx.x, n0 = blockIdx.y * 256;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int d = a.d, M = a.rows;
    // weights first: they depend on nothing
    bf16x8 w[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int i = n0 + (wid * 4 + u) * 16 + r; if (i > d - 1) i = d - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w[u][ks] = *reinterpret_cast<const bf16x8*>(a.WkT + (int64_t)i * d + h * 64 + ks * 32 + g * 8);
    }
    // the head's query slice: MT*16 rows x 64 columns, 4 columns (one f32x4 per partial tile) per item
    // (all tiles of an element in flight at once: a runtime-bounded loop of loads is a chain of round trips; tiles past
    // q_nsplit are clamped to the last one and not added)
#pragma unroll
    for (int it0 = 0; it0 < MT * 16 * 16; it0 += 256) {
        const int it = it0 + tid;
        const int row = it >> 4, c4 = (it & 15) * 4;
        const int rr = row < M ? row : M - 1;
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
    // lane holds q̃[row = 16 i + r][col = n0 + 16 (4 wid + u) + 4g .. +3]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int col = n0 + (wid * 4 + u) * 16 + 4 * g;
        if (col >= d) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = i * 16 + r;
            if (row >= M) continue;
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[u][i][j] * a.scale);
            *reinterpret_cast<bf16x4*>(a.qt + ((int64_t)row * a.H + h) * d + col) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Output side: one workgroup (8 waves) per (head, 16 rows).  The rows' context vectors are combined from the key splits
// (w_s = l_s e^(m_s - M) / Σ, fp32) and rounded to bf16 into LDS; waves = 4 column units x 2 K halves; the halves meet in LDS.
// ---------------------------------------------------------------------------------------------------------
template <int D, int MS>                                      // MS: key splits fetched per chunk (>= n_split)
__global__ __launch_bounds__(512) void xattn_cproj_kernel(XCProjArgs a) {
    constexpr int PITCH = D * 2 + 32, KH = D / 2, KSTEPS = KH / 32;
    constexpr int CR = D / 8, CPT = (CR + 31) / 32;           // 16-B chunks per row; per thread (32 threads per row)
    __shared__ __attribute__((aligned(16))) char cs[16 * PITCH];
    __shared__ __attribute__((aligned(16))) f32x4 red[4 * 64];
    const int h = blockIdx.x, m0 = blockIdx.y * 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int unit = wid & 3, kh = wid >> 2;
    const int M = a.rows, H = a.H, S = a.n_split;
    // every load that depends on nothing is issued before the first wait: weights, split statistics, split contexts
    bf16x8 w[KSTEPS];
    {
        const bf16* wp = a.Wv + (int64_t)(h * 64 + unit * 16 + r) * D + kh * KH + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
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
    const char* xp = cs + r * PITCH + (kh * KH + g * 8) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xp + ks * 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], xf, acc, 0, 0, 0);
    }
    if (kh == 1) red[unit * 64 + lane] = acc;
    __syncthreads();
    if (kh == 0) {
        const f32x4 o2 = red[unit * 64 + lane];
        const int orow = m0 + r, col = h * 64 + unit * 16 + 4 * g;       // lane holds att[row][col .. col+3]
        if (orow < M) {
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
    switch ((a.rows + 15) / 16) {
        case 1: hipLaunchKernelGGL((xattn_qtilde_kernel<1, MS>), grid, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((xattn_qtilde_kernel<2, MS>), grid, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((xattn_qtilde_kernel<3, MS>), grid, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((xattn_qtilde_kernel<4, MS>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((xattn_qtilde_kernel<5, MS>), grid, dim3(256), 0, st, a); break;
    }
}
int launch_xattn_qtilde(const XQTildeArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.rows > DG_MAX_ROWS || a.d % 128 || a.H * 64 != a.d || !a.q_partial || !a.q_bias || !a.WkT || !a.qt || a.q_nsplit < 1 || a.q_nsplit > 16) return -1;
    const dim3 grid(a.H, (a.d + 255) / 256);
    if (a.q_nsplit <= 4) launch_qtilde_mt<4>(a, grid, st);
    else if (a.q_nsplit <= 10) launch_qtilde_mt<10>(a, grid, st);
    else launch_qtilde_mt<16>(a, grid, st);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_xattn_estream(const XAttnArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.group < 1 || a.n_split < 1 || a.n_split > 8 || a.T <= 0 || !a.qt || !a.E || !a.cpart || !a.ml || a.H * 64 != a.d) return -1;
    if ((a.T + XA_KT - 1) / XA_KT < a.n_split) return -1;
    const dim3 grid(a.rows * a.n_split), block(512);
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

template <int MS>
static bool launch_cproj_d(const XCProjArgs& a, dim3 grid, hipStream_t st) {
    switch (a.d) {
        case 128: hipLaunchKernelGGL((xattn_cproj_kernel<128, MS>), grid, dim3(512), 0, st, a); return true;
        case 256: hipLaunchKernelGGL((xattn_cproj_kernel<256, MS>), grid, dim3(512), 0, st, a); return true;
        case 384: hipLaunchKernelGGL((xattn_cproj_kernel<384, MS>), grid, dim3(512), 0, st, a); return true;
        case 512: hipLaunchKernelGGL((xattn_cproj_kernel<512, MS>), grid, dim3(512), 0, st, a); return true;
        case 768: hipLaunchKernelGGL((xattn_cproj_kernel<768, MS>), grid, dim3(512), 0, st, a); return true;
        case 1024: hipLaunchKernelGGL((xattn_cproj_kernel<1024, MS>), grid, dim3(512), 0, st, a); return true;
        case 1280: hipLaunchKernelGGL((xattn_cproj_kernel<1280, MS>), grid, dim3(512), 0, st, a); return true;
        default: return false;
    }
}
int launch_xattn_cproj(const XCProjArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.n_split < 1 || a.n_split > 8 || !a.cpart || !a.ml || !a.Wv || !a.bv || !a.out || a.H * 64 != a.d || a.ldo % 4) return -1;
    const dim3 grid(a.H, (a.rows + 15) / 16);
    const bool ok = a.n_split <= 4 ? launch_cproj_d<4>(a, grid, st) : launch_cproj_d<8>(a, grid, st);
    if (!ok) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
