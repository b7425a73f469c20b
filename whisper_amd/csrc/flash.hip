// flash.hip — fused multi-head attention for the encoder on bf16 MFMA tiles (SURVEY.md §7 `attn_enc`).
//
// Non-causal softmax(Q Kᵀ) V, head_dim 64, T = 1500, Q/K/V in [B][H][T][64] bf16 (q and k already carry
// head_dim^-0.25).  One workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 rows (two
// 16-row MFMA tiles that share every K/V fragment read).  K/V tiles of 64 keys are register-staged into a
// double-buffered LDS image (one barrier per tile).
//
// Everything keeps the QUERY on the MFMA column (lane & 15):
//   Sᵀ[key][q] = K · Qᵀ      A operand = K rows (ds_read_b128 from an XOR-swizzled image), B operand = Q (registers)
//   softmax                  a lane holds 16 keys of ONE query per tile: row max/sum are in-lane + 2 cross-group shuffles
//   Oᵀ[d][q]  += Vᵀ · Pᵀ     B operand = the Sᵀ accumulator itself, converted to bf16 in place (the accumulator's
//                            key order defines the k-slot order); A operand = Vᵀ via ds_read_b64_tr_b16 from a row-major
//                            V image with 160-B rows (conflict-free for this read)
// so P never touches LDS and the running rescale factor is already on the right lane.
#include "common.h"

namespace wx {

typedef __attribute__((ext_vector_type(4))) short i16x4;

constexpr int FA_KT = 64;                 // keys per tile
constexpr int FA_KROW = 128;              // K image row bytes (64 bf16), chunk c of row r at slot c ^ (r & 7)
constexpr int FA_VROW = 160;              // V image row bytes (128 data + 32 pad)
constexpr int FA_KBYTES = FA_KT * FA_KROW, FA_VBYTES = FA_KT * FA_VROW;
constexpr float FA_LOG2E = 1.4426950408889634f;

// launch_bounds(256, 3): three waves per SIMD (158 VGPRs, no spills).  The softmax is VALU work (~200 instructions per lane
// per 64-key tile) that can only overlap the MFMAs of ANOTHER wave: at the compiler's default allocation (204 registers, two
// waves per SIMD) a launch took 1439 us, at three 1082 us; four needs 128 registers, spills, and takes 3518 us.
__global__ __launch_bounds__(256, 3) void attn_flash_bf16_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2 * (FA_KBYTES + FA_VBYTES)];
    // 1-D grid, XCD-aware: blocks that share an XCD (blockIdx % 8) take ALL query blocks of one (batch, head) before the
    // next, so a head's K/V (384 KB) is fetched into one L2 instead of all eight (measured: 4.2 GB -> of L2 misses per launch).
    const int nqb = (a.n_q + 127) / 128, nbh = a.B * a.H;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int bh = (slot / nqb) * 8 + xcd, qb = slot % nqb;
    if (bh >= nbh) return;
    const int b = bh / a.H, h = bh % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, r = lane & 15;
    const int n_q = a.n_q, n_keys = a.n_keys;
    const int q0 = qb * 128 + wid * 32;

    const bf16* Qp = reinterpret_cast<const bf16*>(a.Q) + (int64_t)b * a.q_sb + (int64_t)h * a.q_sh;
    const bf16* Kp = reinterpret_cast<const bf16*>(a.K) + ((int64_t)b * a.H + h) * a.kv_sbh;
    const bf16* Vp = reinterpret_cast<const bf16*>(a.V) + ((int64_t)b * a.H + h) * a.kv_sbh;

    bf16x8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int iq = q0 + qt * 16 + r; if (iq > n_q - 1) iq = n_q - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)iq * a.q_sq + ks * 32 + g * 8);
    }
    // staging: chunk c = tid + p*256 -> (row = c >> 3, 16-B chunk ch = c & 7) of the 64 x 128 B tile
    int st_row[2], st_ch[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) { const int c = tid + p * 256; st_row[p] = c >> 3; st_ch[p] = c & 7; }
    bf16x8 rk[2], rv[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int key = t * FA_KT + st_row[p]; if (key > n_keys - 1) key = n_keys - 1;
            rk[p] = *reinterpret_cast<const bf16x8*>(Kp + (int64_t)key * HEAD_DIM + st_ch[p] * 8);
            rv[p] = *reinterpret_cast<const bf16x8*>(Vp + (int64_t)key * HEAD_DIM + st_ch[p] * 8);
        }
    };
    auto store_tile = [&](int buf) {
        char* kb = lds + buf * (FA_KBYTES + FA_VBYTES);
        char* vb = kb + FA_KBYTES;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<bf16x8*>(kb + st_row[p] * FA_KROW + ((st_ch[p] ^ (st_row[p] & 7)) << 4)) = rk[p];
            *reinterpret_cast<bf16x8*>(vb + st_row[p] * FA_VROW + (st_ch[p] << 4)) = rv[p];
        }
    };

    f32x4 acc_o[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    const int nt = (n_keys + FA_KT - 1) / FA_KT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);
        const char* kb = lds + (t & 1) * (FA_KBYTES + FA_VBYTES);
        const char* vb = kb + FA_KBYTES;
        // ---- S^T = K Q^T -------------------------------------------------------------------------------
        f32x4 acc_s[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int row = kt * 16 + r;
            bf16x8 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kb + row * FA_KROW + (((ks * 4 + g) ^ (row & 7)) << 4));
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[qt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[qt][1], c, 0, 0, 0);
                acc_s[kt][qt] = c;
            }
        }
        // mask keys past the end (only the last tile can have them)
        if (t == nt - 1 && (n_keys % FA_KT) != 0) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = t * FA_KT + kt * 16 + g * 4 + j;
                    if (key >= n_keys) { acc_s[kt][0][j] = -INFINITY; acc_s[kt][1][j] = -INFINITY; }
                }
        }
        // ---- online softmax, query on the lane -------------------------------------------------------------
        bf16x8 pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mt = acc_s[0][qt][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mt = fmaxf(mt, acc_s[kt][qt][j]);
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m_run[qt], mt);
            const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - mn) * FA_LOG2E);
            const float mn2 = mn * FA_LOG2E;
            float ls = 0.f;
            float p[4][4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p[kt][j] = __builtin_amdgcn_exp2f(fmaf(acc_s[kt][qt][j], FA_LOG2E, -mn2));
                    ls += p[kt][j];
                }
            l_run[qt] = l_run[qt] * alpha + ls;
            m_run[qt] = mn;
            // after the first few tiles the running maximum rarely moves: alpha is exactly 1 on every lane and the 16
            // multiplies (of ~100 VALU instructions per tile and query tile) are skipped -- bit-identical either way
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc_o[dt][qt] *= alpha;
            }
            // k-slot order of the P^T operand = key order of the S^T accumulators: step s takes tiles 2s and 2s+1
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) { pf[qt][s][j] = (bf16)p[2 * s][j]; pf[qt][s][4 + j] = (bf16)p[2 * s + 1][j]; }
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------------------------
        const int tq = r >> 2, tp = r & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // lane (g, r) receives V[key = 32s + 16hh + 4g + j][d = 16dt + r], j = 0..3, for hh = 0, 1
                const char* p0 = vb + (32 * s + 4 * g + tq) * FA_VROW + (16 * dt + 4 * tp) * 2;
                const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0));
                const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p0 + 16 * FA_VROW));
                typedef __attribute__((ext_vector_type(8))) short i16x8;
                const i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) acc_o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][s], acc_o[dt][qt], 0, 0, 0);
            }
        }
        if (more) store_tile((t + 1) & 1);
        __syncthreads();
    }
    // ---- normalise and store: lane holds O[q = r][d = 16dt + 4g + 0..3] ---------------------------------------------
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int iq = q0 + qt * 16 + r;
        if (iq < n_q) {
            bf16* op = reinterpret_cast<bf16*>(a.O) + (int64_t)b * a.o_sb + (int64_t)iq * a.o_sq + (int64_t)h * a.o_sh;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc_o[dt][qt][j] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 16 + g * 4) = o;
            }
        }
    }
}

int launch_attn_flash_bf16(const AttnArgs& a, hipStream_t st) {
    if (a.dt != DT_BF16 || !a.Q || !a.K || !a.V || !a.O || a.n_q <= 0 || a.n_keys <= 0 || a.B <= 0 || a.H <= 0) return -1;
    if ((a.q_sb | a.q_sq | a.q_sh | a.kv_sbh) % 8 != 0 || (a.o_sb | a.o_sq | a.o_sh) % 4 != 0) return -1;
    dim3 grid((unsigned)(((a.n_q + 127) / 128) * ((a.B * a.H + 7) / 8 * 8)));
    hipLaunchKernelGGL(attn_flash_bf16_kernel, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
