// attention.hip — streaming attention kernels for gfx950.
//
// attn_decode  one query per (batch, head): decoder self-attention over the paged KV cache and decoder
//              cross-attention over the cached encoder K/V (SURVEY.md §7 `attn_dec_self`, `attn_dec_cross` — the
//              dominant decode cost: 2*1500*64*sizeof(T) bytes per (b, h) per layer per step, read exactly once).
//              HBM-bound by construction: every wave-instruction is one fully coalesced 1 KiB load (64 lanes x 16 B),
//              16 of them in flight per wave, no LDS staging (the data has no reuse), dot products reduced inside
//              8-lane (bf16) / 16-lane (fp32) groups with DPP, one independent online-softmax stream per lane group
//              so the key loop has no cross-lane max and no barrier.
// attn_rows    same lane mapping, NQ queries per wave sharing each K/V load: fp32 encoder attention ("token-exact"
//              mode) and the reference implementation the MFMA flash kernel is tested against.
#include "common.h"
#include <cstdlib>

namespace wx {

template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int EPL = 4;
    typedef f32x4 Raw;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = r[j];
    }
    static __device__ __forceinline__ Raw pack(const float* f) { return Raw{f[0], f[1], f[2], f[3]}; }
};
template <> struct Chunk<bf16> {
    static constexpr int EPL = 8;
    typedef bf16x8 Raw;
    static __device__ __forceinline__ void unpack(const Raw& r, float* f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)r[j];
    }
    static __device__ __forceinline__ Raw pack(const float* f) {
        Raw r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (bf16)f[j];
        return r;
    }
};

template <int CTRL> __device__ __forceinline__ float dpp_move(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// sum over aligned groups of LPK lanes, result in every lane of the group
template <int LPK> __device__ __forceinline__ float group_sum(float x) {
    x += dpp_move<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x);   // row_half_mirror: joins the two quads of 8 lanes
    if (LPK == 16) x += dpp_move<0x140>(x);   // row_mirror: joins the two halves of 16 lanes
    return x;
}

constexpr float LOG2E = 1.4426950408889634f;

// merge two online-softmax states (m, l, o[]) <- (m2, l2, o2[])
template <int E> __device__ __forceinline__ void merge_state(float& m, float& l, float* o, float m2, float l2, const float* o2) {
    const float mn = fmaxf(m, m2);
    const float a = (m == -INFINITY) ? 0.f : exp2f((m - mn) * LOG2E);
    const float b = (m2 == -INFINITY) ? 0.f : exp2f((m2 - mn) * LOG2E);
    l = l * a + l2 * b;
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = o[e] * a + o2[e] * b;
    m = mn;
}

// ---------------------------------------------------------------------------------------------------------
// decode: one query per (b, h)
// ---------------------------------------------------------------------------------------------------------
// query (and, for self-attention, this step's key/value row) of head h of sequence b for lane `lane`: element `lane` of
// each 64-wide row, either read directly or assembled from the projection GEMM's split-K partial tiles (fixed order,
// all tiles in flight at once: loads past q_nsplit are clamped to the last tile and not added).
template <typename T, int NROWS, int MS>
__device__ __forceinline__ void load_head_rows_ms(const AttnArgs& a, int b, int h, int lane, float (&out)[NROWS]) {
    const int dm = a.H * HEAD_DIM;
#pragma unroll
    for (int which = 0; which < NROWS; ++which) {
        const int col = which * dm + h * HEAD_DIM + lane;
        float pt[MS];
#pragma unroll
        for (int s = 0; s < MS; ++s) pt[s] = a.q_partial[((int64_t)(s < a.q_nsplit ? s : a.q_nsplit - 1) * a.B + b) * a.q_sb + col];
        float v = a.q_bias[col];
#pragma unroll
        for (int s = 0; s < MS; ++s) if (s < a.q_nsplit) v += pt[s];
        if (which < 2) v *= a.q_scale;        // q and k carry head_dim^-0.25; v does not
        out[which] = (float)(T)v;            // same rounding point as a stored q/k/v row
    }
}

template <typename T, int NROWS>
__device__ __forceinline__ void load_head_rows(const AttnArgs& a, int b, int h, int lane, float (&out)[NROWS]) {
    if (a.q_partial) {
        // every load costs issue time even when it is a clamped duplicate: fetch 4, 10 or 16 tiles, whichever covers the
        // split (uniform branch on a kernel argument)
        if (a.q_nsplit <= 4) load_head_rows_ms<T, NROWS, 4>(a, b, h, lane, out);
        else if (a.q_nsplit <= 10) load_head_rows_ms<T, NROWS, 10>(a, b, h, lane, out);
        else load_head_rows_ms<T, NROWS, 16>(a, b, h, lane, out);
    } else {
        out[0] = (float)reinterpret_cast<const T*>(a.Q)[(int64_t)b * a.q_sb + h * a.q_sh + lane];
        if (NROWS > 1) {
            out[1] = (float)reinterpret_cast<const T*>(a.Knew)[(int64_t)b * a.new_sb + h * HEAD_DIM + lane];
            out[2] = (float)reinterpret_cast<const T*>(a.Vnew)[(int64_t)b * a.new_sb + h * HEAD_DIM + lane];
        }
    }
}

// Decoder self-attention over the paged KV cache: NWH waves per (sequence, head) interleave the key iterations, four
// heads per workgroup, so a batch-64 launch is 320 workgroups that are all resident at once.  Round trip 1 fetches
// everything that depends on no other load (position, page-table row, q/k/v partial tiles); round trip 2 is the K/V
// stream: <= 448 keys, 64 keys (8 KiB K + 8 KiB V in bf16) per iteration in flight per wave.
template <typename T, int NWH, int HPB = 4, int U = 8, bool GROUPED = false, int R = 1>
__global__ __launch_bounds__(64 * HPB * NWH) void attn_self_decode_kernel(AttnArgs a) {
    typedef Chunk<T> C;
    constexpr int EPL = C::EPL, LPK = HEAD_DIM / EPL, KPI = 64 / LPK, IT = KPI * U;
    __shared__ float s_row[HPB][3][HEAD_DIM];
    __shared__ int s_pt[HPB][448];                 // page-table row (page size 16: 28 entries; page size 1: n_text_ctx)
    __shared__ float s_m[HPB][NWH], s_l[HPB][NWH];
    __shared__ float s_o[HPB][NWH][HEAD_DIM];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int hs = wid / NWH, kw = wid % NWH;          // head slot in the workgroup, key-split wave of that head
    int bh = blockIdx.x * HPB + hs;
    const bool valid = bh < a.B * a.H;
    if (!valid) bh = a.B * a.H - 1;
    const int b = bh / a.H, h = bh % a.H;
    const int sub = lane % LPK, grp = lane / LPK;
    // prompt pass (q_group > 1): row b is position *d_pos + b % q_group of sequence b / q_group, whose k/v rows are already
    // in the cache (launch_kv_append ran first): nothing is appended or patched here
    constexpr bool single = !GROUPED;
    const int seq = GROUPED ? b / a.q_group : b;
    const int t_cur = *a.d_pos + (GROUPED ? b % a.q_group : 0), n_keys = t_cur + 1;
    const int PS = a.page_size;
    if (kw == 0) {
        for (int i = lane; i < a.pt_ld; i += 64) s_pt[hs][i] = a.page_table[seq * a.pt_ld + i];
        float rows[3];
        load_head_rows<T, 3>(a, b, h, lane, rows);
#pragma unroll
        for (int w = 0; w < 3; ++w) s_row[hs][w][lane] = rows[w];
    }
    __syncthreads();
    float qf[EPL], kf[EPL], vf[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { qf[e] = s_row[hs][0][sub * EPL + e]; kf[e] = s_row[hs][1][sub * EPL + e]; vf[e] = s_row[hs][2][sub * EPL + e]; }
    const typename C::Raw kn_r = C::pack(kf), vn_r = C::pack(vf);
    T* Kp = reinterpret_cast<T*>(a.Kpool);
    T* Vp = reinterpret_cast<T*>(a.Vpool);
    if (single && valid && kw == 0 && lane < LPK) {     // append this step's key/value row to the page of position t_cur
        const int64_t off = (((int64_t)s_pt[hs][t_cur / PS] * a.H + h) * PS + t_cur % PS) * HEAD_DIM + sub * EPL;
        *reinterpret_cast<typename C::Raw*>(Kp + off) = kn_r;
        *reinterpret_cast<typename C::Raw*>(Vp + off) = vn_r;
    }
    float m = -INFINITY, l = 0.f, o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;
    const int total_it = (n_keys + IT - 1) / IT;
    // R iterations' loads are issued before the first of them is consumed (few rows: the launch is a handful of workgroups and
    // every iteration would be one exposed round trip).  Iterations past the end are clamped duplicates whose scores are masked,
    // so the online softmax sees the same keys in the same order for every R.
    for (int it0 = kw; it0 < total_it; it0 += NWH * R) {
        typename C::Raw kr[R][U], vr[R][U];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int key0 = (it0 + rr * NWH) * IT + grp;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int key = key0 + u * KPI;
                if (key > n_keys - 1) key = n_keys - 1;
                // the slot of position t_cur is being written by this launch: it is read (address valid, value ignored)
                // and replaced by the register copy
                const int64_t off = (((int64_t)s_pt[hs][key / PS] * a.H + h) * PS + key % PS) * HEAD_DIM + sub * EPL;
                kr[rr][u] = __builtin_nontemporal_load(reinterpret_cast<const typename C::Raw*>(Kp + off));
                vr[rr][u] = __builtin_nontemporal_load(reinterpret_cast<const typename C::Raw*>(Vp + off));
                if (single && key == t_cur) { kr[rr][u] = kn_r; vr[rr][u] = vn_r; }
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int key0 = (it0 + rr * NWH) * IT + grp;
            float s[U];
            float mt = -INFINITY;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float kk[EPL];
                C::unpack(kr[rr][u], kk);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) d = fmaf(qf[e], kk[e], d);
                d = group_sum<LPK>(d);
                s[u] = (key0 + u * KPI < n_keys) ? d : -INFINITY;
                mt = fmaxf(mt, s[u]);
            }
            if (mt > -INFINITY) {
                const float mn = fmaxf(m, mt);
                const float alpha = (m == -INFINITY) ? 0.f : exp2f((m - mn) * LOG2E);
                l *= alpha;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] *= alpha;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float p = exp2f((s[u] - mn) * LOG2E);   // exp2(-inf) = 0 for masked keys
                    l += p;
                    float vv[EPL];
                    C::unpack(vr[rr][u], vv);
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[e] = fmaf(p, vv[e], o[e]);
                }
                m = mn;
            }
        }
    }
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        float o2[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o2[e] = __shfl_xor(o[e], off);
        merge_state<EPL>(m, l, o, m2, l2, o2);
    }
    if (lane < LPK) {
        if (lane == 0) { s_m[hs][kw] = m; s_l[hs][kw] = l; }
#pragma unroll
        for (int e = 0; e < EPL; ++e) s_o[hs][kw][sub * EPL + e] = o[e];
    }
    __syncthreads();
    if (kw == 0 && valid) {
        const int d = lane;
        float M = s_m[hs][0];
#pragma unroll
        for (int w = 1; w < NWH; ++w) M = fmaxf(M, s_m[hs][w]);
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < NWH; ++w) {
            const float f = (s_m[hs][w] == -INFINITY) ? 0.f : exp2f((s_m[hs][w] - M) * LOG2E);
            L += s_l[hs][w] * f;
            acc += s_o[hs][w][d] * f;
        }
        reinterpret_cast<T*>(a.O)[(int64_t)b * a.o_sb + h * a.o_sh + d] = (T)(acc / L);
    }
}

// Decoder cross-attention over the cached encoder K/V (dense [windows][H][n_keys][64]): one workgroup per (window, head)
// and key split; its four waves interleave the key iterations.  NQ query rows (the beams of the window, kv_group of
// them) share every K/V load: beam search at width 5 streams the cache once per step, not five times.
template <typename T, int NQ>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
    typedef Chunk<T> C;
    constexpr int EPL = C::EPL, LPK = HEAD_DIM / EPL, KPI = 64 / LPK, U = 8, IT = KPI * U;
    __shared__ float s_m[4][NQ], s_l[4][NQ];
    __shared__ float s_o[4][NQ][HEAD_DIM];
    __shared__ float s_q[NQ][HEAD_DIM];
    const int w = blockIdx.x / a.H, h = blockIdx.x % a.H, split = blockIdx.y;     // w: K/V entry (window)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sub = lane % LPK, grp = lane / LPK;
    const int n_keys = a.n_keys;
    const int row0 = w * a.kv_group;
    for (int q = wid; q < NQ; q += 4) {
        int row = row0 + (q < a.kv_group ? q : a.kv_group - 1);      // NQ may round kv_group up: extra slots repeat the last row
        if (row > a.B - 1) row = a.B - 1;
        float rows[1];
        load_head_rows<T, 1>(a, row, h, lane, rows);
        s_q[q][lane] = rows[0];
    }
    __syncthreads();
    float qf[NQ][EPL];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < EPL; ++e) qf[q][e] = s_q[q][sub * EPL + e];
    const T* kbase = reinterpret_cast<const T*>(a.K) + ((int64_t)w * a.H + h) * a.kv_sbh + sub * EPL;
    const T* vbase = reinterpret_cast<const T*>(a.V) + ((int64_t)w * a.H + h) * a.kv_sbh + sub * EPL;
    const int total_it = (n_keys + IT - 1) / IT;
    const int it0 = (int)((int64_t)split * total_it / a.n_split), it1 = (int)((int64_t)(split + 1) * total_it / a.n_split);

    float m[NQ], l[NQ], o[NQ][EPL];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        m[q] = -INFINITY; l[q] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[q][e] = 0.f;
    }
    for (int it = it0 + wid; it < it1; it += 4) {
        typename C::Raw kr[U], vr[U];
        const int key0 = it * IT + grp;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int key = key0 + u * KPI;
            if (key > n_keys - 1) key = n_keys - 1;
            kr[u] = __builtin_nontemporal_load(reinterpret_cast<const typename C::Raw*>(kbase + (int64_t)key * HEAD_DIM));
            vr[u] = __builtin_nontemporal_load(reinterpret_cast<const typename C::Raw*>(vbase + (int64_t)key * HEAD_DIM));
        }
        float s[NQ][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float kf[EPL];
            C::unpack(kr[u], kf);
            const bool valid = key0 + u * KPI < n_keys;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) d = fmaf(qf[q][e], kf[e], d);
                d = group_sum<LPK>(d);
                s[q][u] = valid ? d : -INFINITY;
            }
        }
        float vf[U][EPL];
#pragma unroll
        for (int u = 0; u < U; ++u) C::unpack(vr[u], vf[u]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float mt = s[q][0];
#pragma unroll
            for (int u = 1; u < U; ++u) mt = fmaxf(mt, s[q][u]);
            if (mt > -INFINITY) {
                const float mn = fmaxf(m[q], mt);
                const float alpha = (m[q] == -INFINITY) ? 0.f : exp2f((m[q] - mn) * LOG2E);
                l[q] *= alpha;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[q][e] *= alpha;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float p = exp2f((s[q][u] - mn) * LOG2E);   // exp2(-inf) = 0 for masked keys
                    l[q] += p;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[q][e] = fmaf(p, vf[u][e], o[q][e]);
                }
                m[q] = mn;
            }
        }
    }
    // merge the KPI lane groups of the wave (lanes that hold the same 16-B chunk of the head)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(m[q], off), l2 = __shfl_xor(l[q], off);
            float o2[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) o2[e] = __shfl_xor(o[q][e], off);
            merge_state<EPL>(m[q], l[q], o[q], m2, l2, o2);
        }
        if (lane < LPK) {
            if (lane == 0) { s_m[wid][q] = m[q]; s_l[wid][q] = l[q]; }
#pragma unroll
            for (int e = 0; e < EPL; ++e) s_o[wid][q][sub * EPL + e] = o[q][e];
        }
    }
    __syncthreads();
    for (int q = wid; q < NQ; q += 4) {
        const int row = row0 + q;
        if (q >= a.kv_group || row >= a.B) continue;
        const int d = lane;
        const float M = fmaxf(fmaxf(s_m[0][q], s_m[1][q]), fmaxf(s_m[2][q], s_m[3][q]));
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float f = (s_m[ww][q] == -INFINITY) ? 0.f : exp2f((s_m[ww][q] - M) * LOG2E);
            L += s_l[ww][q] * f;
            acc += s_o[ww][q][d] * f;
        }
        if (a.n_split == 1) {
            reinterpret_cast<T*>(a.O)[(int64_t)row * a.o_sb + h * a.o_sh + d] = (T)(acc / L);
        } else {
            float* pp = a.partial + (((int64_t)row * a.H + h) * a.n_split + split) * (HEAD_DIM + 2);
            pp[d] = acc;
            if (d == 0) { pp[HEAD_DIM] = M; pp[HEAD_DIM + 1] = L; }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_decode_combine_kernel(AttnArgs a) {
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H, d = threadIdx.x;
    const float* pp = a.partial + (int64_t)blockIdx.x * a.n_split * (HEAD_DIM + 2);
    T* op = reinterpret_cast<T*>(a.O) + (int64_t)b * a.o_sb + h * a.o_sh + d;
    *op = (T)attn_combine_splits(pp, a.n_split, d);
}

// ---------------------------------------------------------------------------------------------------------
// rows: NQ queries per wave, all keys, dense non-causal
// ---------------------------------------------------------------------------------------------------------
template <typename T, int NQ, int U>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnArgs a) {
    typedef Chunk<T> C;
    constexpr int EPL = C::EPL, LPK = HEAD_DIM / EPL, KPI = 64 / LPK, IT = KPI * U;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sub = lane % LPK, grp = lane / LPK;
    const int iq0 = (blockIdx.y * 4 + wid) * NQ;
    if (iq0 >= a.n_q) return;
    const int n_keys = a.n_keys;
    float qf[NQ][EPL];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int iq = iq0 + q; if (iq > a.n_q - 1) iq = a.n_q - 1;
        const T* qp = reinterpret_cast<const T*>(a.Q) + (int64_t)b * a.q_sb + (int64_t)iq * a.q_sq + h * a.q_sh + sub * EPL;
        C::unpack(*reinterpret_cast<const typename C::Raw*>(qp), qf[q]);
    }
    const T* kbase = reinterpret_cast<const T*>(a.K) + ((int64_t)b * a.H + h) * a.kv_sbh + sub * EPL;
    const T* vbase = reinterpret_cast<const T*>(a.V) + ((int64_t)b * a.H + h) * a.kv_sbh + sub * EPL;
    float m[NQ], l[NQ], o[NQ][EPL];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        m[q] = -INFINITY; l[q] = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[q][e] = 0.f;
    }
    const int total_it = (n_keys + IT - 1) / IT;
    for (int it = 0; it < total_it; ++it) {
        typename C::Raw kr[U], vr[U];
        const int key0 = it * IT + grp;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int key = key0 + u * KPI;
            if (key > n_keys - 1) key = n_keys - 1;
            kr[u] = *reinterpret_cast<const typename C::Raw*>(kbase + (int64_t)key * HEAD_DIM);
            vr[u] = *reinterpret_cast<const typename C::Raw*>(vbase + (int64_t)key * HEAD_DIM);
        }
        float s[NQ][U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float kf[EPL];
            C::unpack(kr[u], kf);
            const bool valid = key0 + u * KPI < n_keys;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) d = fmaf(qf[q][e], kf[e], d);
                d = group_sum<LPK>(d);
                s[q][u] = valid ? d : -INFINITY;
            }
        }
        float vf[U][EPL];
#pragma unroll
        for (int u = 0; u < U; ++u) C::unpack(vr[u], vf[u]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float mt = s[q][0];
#pragma unroll
            for (int u = 1; u < U; ++u) mt = fmaxf(mt, s[q][u]);
            if (mt > -INFINITY) {
                const float mn = fmaxf(m[q], mt);
                const float alpha = (m[q] == -INFINITY) ? 0.f : exp2f((m[q] - mn) * LOG2E);
                l[q] *= alpha;
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[q][e] *= alpha;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float p = exp2f((s[q][u] - mn) * LOG2E);
                    l[q] += p;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[q][e] = fmaf(p, vf[u][e], o[q][e]);
                }
                m[q] = mn;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(m[q], off), l2 = __shfl_xor(l[q], off);
            float o2[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) o2[e] = __shfl_xor(o[q][e], off);
            merge_state<EPL>(m[q], l[q], o[q], m2, l2, o2);
        }
        const int iq = iq0 + q;
        if (lane < LPK && iq < a.n_q) {
            float r[EPL];
            const float inv = 1.0f / l[q];
#pragma unroll
            for (int e = 0; e < EPL; ++e) r[e] = o[q][e] * inv;
            T* op = reinterpret_cast<T*>(a.O) + (int64_t)b * a.o_sb + (int64_t)iq * a.o_sq + h * a.o_sh + sub * EPL;
            *reinterpret_cast<typename C::Raw*>(op) = C::pack(r);
        }
    }
}

static int check_common(const AttnArgs& a) {
    if (a.B <= 0 || a.H <= 0 || (!a.Q && !a.q_partial) || !a.O) return -1;
    if (a.q_partial && (!a.q_bias || a.q_nsplit < 1 || a.q_nsplit > 16 || a.q_sb % 4 != 0)) return -1;
    const size_t es = dtype_size(a.dt);
    if ((a.q_sb * es) % 16 || (a.q_sq * es) % 16 || (a.q_sh * es) % 16 || (a.o_sb * es) % 16 || (a.o_sq * es) % 16 || (a.o_sh * es) % 16 || (a.kv_sbh * es) % 16 || (a.new_sb * es) % 16) return -1;
    return 0;
}

// Prompt pass: the k/v rows of all a.B rows (q_group consecutive positions of each sequence), assembled from the q|k|v
// projection's split-K tiles exactly as attn_self_decode_kernel assembles them, written to the paged cache.  One wave per
// (row, head), 128-B stores.
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int rh = blockIdx.x * 4 + wid;
    if (rh >= a.B * a.H) return;
    const int r = rh / a.H, h = rh % a.H;
    const int seq = r / a.q_group, t = *a.d_pos + r % a.q_group, PS = a.page_size;
    float rows[3];
    load_head_rows<T, 3>(a, r, h, lane, rows);
    const int page = a.page_table[seq * a.pt_ld + t / PS];
    const int64_t off = (((int64_t)page * a.H + h) * PS + t % PS) * HEAD_DIM + lane;
    reinterpret_cast<T*>(a.Kpool)[off] = (T)rows[1];
    reinterpret_cast<T*>(a.Vpool)[off] = (T)rows[2];
}

int launch_kv_append(const AttnArgs& a, hipStream_t st) {
    if (check_common(a)) return -1;
    if (!a.page_table || !a.d_pos || !a.Kpool || !a.Vpool || a.q_group < 1 || a.B % a.q_group || a.page_size != KV_PAGE) return -1;
    if (!a.q_partial && (!a.Q || !a.Knew || !a.Vnew)) return -1;
    dim3 grid((a.B * a.H + 3) / 4);
    if (a.dt == DT_BF16) hipLaunchKernelGGL(kv_append_kernel<bf16>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(kv_append_kernel<float>, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_attn_decode(const AttnArgs& a, hipStream_t st) {
    if (check_common(a)) return -1;
    if (a.n_q != 1 || a.n_split < 1) return -1;
    const bool paged = a.page_table != nullptr;
    if (paged) {
        if (a.q_group < 1 || a.B % a.q_group || (a.q_group > 1 && a.page_size != KV_PAGE)) return -1;
        if (!a.d_pos || !a.Kpool || !a.Vpool || a.pt_ld > 448 || (a.page_size != 1 && a.page_size != KV_PAGE) || a.n_split != 1 || (!a.q_partial && (!a.Knew || !a.Vnew))) return -1;
        dim3 grid((a.B * a.H + 3) / 4);
        // bf16: 2 waves per head, 2 heads per workgroup (640 workgroups of 256 threads at batch 64), 16 keys per
        // wave-iteration.  Swept over waves/head x heads/workgroup x keys/iteration, with and without a software-pipelined
        // load (tools/microbench, t = 8 / 64 / 224 / 440): 4 heads x 64 keys 9.5 / 13.5 / 22.6 / 35.4 us, this shape
        // 6.1 / 9.4 / 17.4 / 27.7 us; pipelining the loads changes nothing once the iterations are this fine.  The grouped
        // (prompt pass) variant is a separate instantiation so the per-token step carries no index arithmetic for it.
        const dim3 grid2((a.B * a.H + 1) / 2);
        // few rows (<= 8 at 20 heads): four iterations' loads in flight per wave -- same keys, same order, same bits
        const int few_r = getenv("WX_SELF_ATTN_R1") ? 0 : 160;     // (read per eager launch / capture: the test toggles it)
        if (a.dt == DT_BF16 && a.q_group == 1 && a.B * a.H <= few_r) hipLaunchKernelGGL((attn_self_decode_kernel<bf16, 2, 2, 2, false, 4>), grid2, dim3(256), 0, st, a);
        else if (a.dt == DT_BF16 && a.q_group == 1) hipLaunchKernelGGL((attn_self_decode_kernel<bf16, 2, 2, 2, false>), grid2, dim3(256), 0, st, a);
        else if (a.dt == DT_BF16) hipLaunchKernelGGL((attn_self_decode_kernel<bf16, 2, 2, 2, true>), grid2, dim3(256), 0, st, a);
        else if (a.q_group > 1) hipLaunchKernelGGL((attn_self_decode_kernel<float, 2, 4, 8, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((attn_self_decode_kernel<float, 2>), grid, dim3(512), 0, st, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (!a.K || !a.V || a.n_keys <= 0) return -1;
    if (a.n_split > 1 && !a.partial) return -1;
    if (a.kv_group < 1 || a.kv_group > 8) return -1;
    const int n_kv = (a.B + a.kv_group - 1) / a.kv_group;
    dim3 grid(n_kv * a.H, a.n_split), block(256);
    const int nq = a.kv_group == 1 ? 1 : a.kv_group == 2 ? 2 : a.kv_group <= 4 ? 4 : a.kv_group == 5 ? 5 : 8;
#define WX_LAUNCH_CROSS(T)                                                                                      \
    switch (nq) {                                                                                               \
        case 1: hipLaunchKernelGGL((attn_decode_kernel<T, 1>), grid, block, 0, st, a); break;                   \
        case 2: hipLaunchKernelGGL((attn_decode_kernel<T, 2>), grid, block, 0, st, a); break;                   \
        case 4: hipLaunchKernelGGL((attn_decode_kernel<T, 4>), grid, block, 0, st, a); break;                   \
        case 5: hipLaunchKernelGGL((attn_decode_kernel<T, 5>), grid, block, 0, st, a); break;                   \
        default: hipLaunchKernelGGL((attn_decode_kernel<T, 8>), grid, block, 0, st, a); break;                  \
    }                                                                                                           \
    if (a.n_split > 1 && !a.defer_combine) hipLaunchKernelGGL(attn_decode_combine_kernel<T>, dim3(a.B * a.H), dim3(64), 0, st, a);
    if (a.n_split > ACOMB_MAX_SPLIT) return -1;
    if (a.dt == DT_BF16) { WX_LAUNCH_CROSS(bf16) } else { WX_LAUNCH_CROSS(float) }
#undef WX_LAUNCH_CROSS
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_attn_combine(const AttnArgs& a, hipStream_t st) {
    if (check_common(a) || a.n_split < 2 || a.n_split > ACOMB_MAX_SPLIT || !a.partial || !a.O) return -1;
    if (a.dt == DT_BF16) hipLaunchKernelGGL(attn_decode_combine_kernel<bf16>, dim3(a.B * a.H), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(attn_decode_combine_kernel<float>, dim3(a.B * a.H), dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_attn_rows(const AttnArgs& a, hipStream_t st) {
    if (check_common(a)) return -1;
    if (!a.Q || !a.K || !a.V || a.n_keys <= 0 || a.n_q <= 0) return -1;
    constexpr int NQ = 4;
    dim3 grid(a.B * a.H, (a.n_q + 4 * NQ - 1) / (4 * NQ)), block(256);
    if (a.dt == DT_BF16) hipLaunchKernelGGL((attn_rows_kernel<bf16, NQ, 4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_rows_kernel<float, NQ, 4>), grid, block, 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
