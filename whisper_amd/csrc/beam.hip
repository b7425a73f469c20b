// beam.hip — device-side beam search step for gfx950 (replaces upstream BeamSearchDecoder.update + rearrange_kv_cache,
// reached from DecodingTask.run when beam_size is set; SURVEY.md §8 row a11, BASELINE.json configs[3]).
//
// Rows: R = B windows x n beams.  One step after the decoder layers has four tiny launches, all graph-replayable (the
// position is read from device memory):
//   beam_topk    one workgroup per row: logit filters (same rules as the greedy sampler), ONE pass over the logit row
//                keeping an online log-sum-exp and every thread's best n+1 entries, then n+1 block-wide arg-max rounds
//                -> candidates (sum_logprob + log_softmax, token), best first.
//   beam_select  one workgroup per window: the n(n+1) candidates in upstream's visiting order (score descending, ties
//                by insertion order), EOT candidates go to the window's finished list (while it has room), the first n
//                others become the new beams (source row, token, cumulative log-probability).
//   beam_gather  new token rows and KV node-index rows gathered from the source rows into scratch ...
//   beam_commit  ... and copied back; every row gets a fresh KV node for the next position; position += 1.
// The self-attention cache is a pool of per-token nodes addressed through the index rows (page size 1), so following
// the beams never copies K/V.
#include "common.h"

namespace wx {

constexpr int BEAM_MAXK = 9;                 // beam width <= 8 -> up to 9 proposals per beam
constexpr int BT_NT = 1024, BT_NW = BT_NT / 64;

struct Cand { float v; int i; };
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }

template <int K>
struct TopList {
    Cand c[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < K; ++k) c[k] = Cand{-INFINITY, 0x7fffffff};
    }
    __device__ __forceinline__ void insert(Cand x) {
        if (!cand_better(x, c[K - 1])) return;
        c[K - 1] = x;
#pragma unroll
        for (int k = K - 1; k > 0; --k) {
            if (cand_better(c[k], c[k - 1])) { const Cand t = c[k]; c[k] = c[k - 1]; c[k - 1] = t; }
        }
    }
    __device__ __forceinline__ void pop() {
#pragma unroll
        for (int k = 0; k + 1 < K; ++k) c[k] = c[k + 1];
        c[K - 1] = Cand{-INFINITY, 0x7fffffff};
    }
};

struct MS2 { float m, s; };
__device__ __forceinline__ void ms2_add(MS2& a, float v) {
    if (v > a.m) { a.s = a.s * __expf(a.m - v) + 1.f; a.m = v; } else a.s += __expf(v - a.m);
}
__device__ __forceinline__ MS2 ms2_merge(MS2 a, MS2 b) {
    if (b.m == -INFINITY) return a;
    if (a.m == -INFINITY) return b;
    const float m = fmaxf(a.m, b.m);
    return MS2{m, a.s * expf(a.m - m) + b.s * expf(b.m - m)};
}

template <int K>
__global__ __launch_bounds__(BT_NT) void beam_topk_kernel(BeamArgs s) {
    __shared__ Cand sh_c[BT_NW];
    __shared__ MS2 sh_msa[BT_NW], sh_mst[BT_NW];
    __shared__ float sh_f[BT_NW], sh_g[BT_NW];
    __shared__ int sh_flags[8];
    __shared__ float sh_dec[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int t = *s.d_pos;
    const float* lg = s.logits + (int64_t)row * s.ldl;
    const int V = s.n_vocab;

    if (t == s.sot_index && s.no_speech && row % s.n == 0) {        // unfiltered softmax at the SOT position, beam 0
        float mx = -INFINITY;
        for (int i = tid; i < V; i += BT_NT) mx = fmaxf(mx, lg[i]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        if (lane == 0) sh_f[wid] = mx;
        __syncthreads();
        mx = sh_f[0];
        for (int w = 1; w < BT_NW; ++w) mx = fmaxf(mx, sh_f[w]);
        float se = 0.f;
        for (int i = tid; i < V; i += BT_NT) se += expf(lg[i] - mx);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) se += __shfl_xor(se, off);
        if (lane == 0) sh_g[wid] = se;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < BT_NW; ++w) tot += sh_g[w];
            s.no_speech[row / s.n] = expf(lg[s.tok_no_speech] - mx) / tot;
        }
        __syncthreads();
    }
    if (t < s.n_prompt - 1) return;                                 // prompt positions: nothing to propose

    const int32_t* trow = s.tokens + (int64_t)row * s.ld_tok;
    const int tb = s.tok_timestamp_begin;
    const bool first_step = (t == s.n_prompt - 1);
    if (tid == 0) {
        int a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0, c_hi = 0, d_lo = V;
        if (!s.without_timestamps) {
            const int cnt = t - s.n_prompt + 1;
            const bool last_ts = cnt >= 1 && trow[t] >= tb;
            const bool penult_ts = cnt < 2 || trow[t - 1] >= tb;
            if (last_ts) {
                if (penult_ts) { a_lo = tb; a_hi = V; } else { a_lo = 0; a_hi = s.tok_eot; }
            }
            int ts_last = -1;
            for (int j = t; j >= s.n_prompt; --j) if (trow[j] >= tb) { ts_last = trow[j]; break; }
            if (ts_last >= 0) { b_lo = tb; b_hi = (last_ts && !penult_ts) ? ts_last : ts_last + 1; }
            if (first_step) {
                c_hi = tb;
                if (s.max_initial_ts >= 0) d_lo = tb + s.max_initial_ts + 1;
            }
        }
        sh_flags[0] = a_lo; sh_flags[1] = a_hi; sh_flags[2] = b_lo; sh_flags[3] = b_hi; sh_flags[4] = c_hi; sh_flags[5] = d_lo;
    }
    __syncthreads();
    const int a_lo = sh_flags[0], a_hi = sh_flags[1], b_lo = sh_flags[2], b_hi = sh_flags[3], c_hi = sh_flags[4], d_lo = sh_flags[5];
    const bool ts_mode = !s.without_timestamps;
    const bool blank_rule = first_step && s.suppress_blank;
    TopList<K> top_all, top_ts;
    top_all.init(); top_ts.init();
    MS2 ms_all{-INFINITY, 0.f}, ms_ts{-INFINITY, 0.f};
    float max_text = -INFINITY;
    for (int i0 = tid * 4; i0 < V; i0 += BT_NT * 4) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(lg + i0);
        const uint32_t sup = *reinterpret_cast<const uint32_t*>(s.suppress + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j;
            bool m = i >= V || ((sup >> (8 * j)) & 0xff);
            if (blank_rule && (i == s.tok_blank || i == s.tok_eot)) m = true;
            if (ts_mode) {
                if (i == s.tok_no_timestamps) m = true;
                if (i >= a_lo && i < a_hi) m = true;
                if (i >= b_lo && i < b_hi) m = true;
                if (i < c_hi || i >= d_lo) m = true;
            }
            if (m) continue;
            const float v = v4[j];
            top_all.insert(Cand{v, i});
            ms2_add(ms_all, v);
            if (i >= tb) { if (ts_mode) { top_ts.insert(Cand{v, i}); ms2_add(ms_ts, v); } }
            else max_text = fmaxf(max_text, v);
        }
    }
    // block-wide normalisers
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        MS2 y; y.m = __shfl_xor(ms_all.m, off); y.s = __shfl_xor(ms_all.s, off); ms_all = ms2_merge(ms_all, y);
        MS2 z; z.m = __shfl_xor(ms_ts.m, off); z.s = __shfl_xor(ms_ts.s, off); ms_ts = ms2_merge(ms_ts, z);
        max_text = fmaxf(max_text, __shfl_xor(max_text, off));
    }
    if (lane == 0) { sh_msa[wid] = ms_all; sh_mst[wid] = ms_ts; sh_f[wid] = max_text; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < BT_NW; ++w) { ms_all = ms2_merge(ms_all, sh_msa[w]); ms_ts = ms2_merge(ms_ts, sh_mst[w]); max_text = fmaxf(max_text, sh_f[w]); }
        // upstream ApplyTimestampRules: if the timestamp tokens together outweigh every text token, only timestamps remain
        float lse = ms_all.m + logf(ms_all.s);
        float use_ts = 0.f;
        if (ts_mode && ms_ts.m > -INFINITY) {
            const float ts_lp = ms_ts.m + logf(ms_ts.s) - lse, text_lp = max_text - lse;
            if (ts_lp > text_lp) { use_ts = 1.f; lse = ms_ts.m + logf(ms_ts.s); }
        }
        sh_dec[0] = lse; sh_dec[1] = use_ts;
    }
    __syncthreads();
    const float lse = sh_dec[0];
    const bool use_ts = sh_dec[1] != 0.f;
    if (use_ts) top_all = top_ts;
    const float base = s.sum_logprob[row];
    // K rounds of block-wide arg-max over every thread's current best
    for (int k = 0; k < s.n + 1; ++k) {
        Cand c = top_all.c[0];
        Cand best = c;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            Cand y; y.v = __shfl_xor(best.v, off); y.i = __shfl_xor(best.i, off);
            if (cand_better(y, best)) best = y;
        }
        if (lane == 0) sh_c[wid] = best;
        __syncthreads();
        best = sh_c[0];
        for (int w = 1; w < BT_NW; ++w) if (cand_better(sh_c[w], best)) best = sh_c[w];
        if (c.i == best.i && c.v == best.v) top_all.pop();           // token ids are unique: exactly one thread owns the winner
        if (tid == 0) {
            s.cand_score[(int64_t)row * BEAM_MAXK + k] = best.v > -INFINITY ? base + (best.v - lse) : -INFINITY;
            s.cand_tok[(int64_t)row * BEAM_MAXK + k] = best.i;
        }
        __syncthreads();
    }
}

// one workgroup (64 threads) per window
__global__ __launch_bounds__(64) void beam_select_kernel(BeamArgs s) {
    __shared__ int sh_fin_src[BEAM_MAXK * 8], sh_fin_slot[BEAM_MAXK * 8], sh_nfin;
    const int w = blockIdx.x, n = s.n, tid = threadIdx.x;
    const int t = *s.d_pos;
    const int r0 = w * n;
    if (t < s.n_prompt - 1) {                                       // prompt: identity reorder, token already in the row
        if (tid < n) { s.src[r0 + tid] = r0 + tid; s.next_tok[r0 + tid] = s.tokens[(int64_t)(r0 + tid) * s.ld_tok + t + 1]; s.sum_tmp[r0 + tid] = s.sum_logprob[r0 + tid]; }
        return;
    }
    // every lane fetches candidates (one round trip for all of them), one lane then walks them in LDS
    __shared__ float sh_sc[8 * BEAM_MAXK], sh_wsc[8 * BEAM_MAXK];
    __shared__ int sh_tk[8 * BEAM_MAXK], sh_wtk[8 * BEAM_MAXK], sh_wsr[8 * BEAM_MAXK], sh_word[8 * BEAM_MAXK];
    for (int q = tid; q < n * (n + 1); q += 64) {
        const int j = q / (n + 1), k = q % (n + 1);
        sh_sc[j * BEAM_MAXK + k] = s.cand_score[(int64_t)(r0 + j) * BEAM_MAXK + k];
        sh_tk[j * BEAM_MAXK + k] = s.cand_tok[(int64_t)(r0 + j) * BEAM_MAXK + k];
    }
    __syncthreads();
    if (tid == 0) {
        const bool first_step = (t == s.n_prompt - 1);
        // candidates in upstream's insertion order (beam-major, proposal rank); at the first step every beam holds the
        // same prefix and the same logits, so the sequence-keyed dict keeps one beam's proposals only
        // (work arrays in LDS: dynamically indexed private arrays live in scratch memory, a cache round trip per access)
        float* sc = sh_wsc; int *tk = sh_wtk, *sr = sh_wsr, *ord = sh_word;
        int nc = 0;
        const int nb = first_step ? 1 : n;
        for (int j = 0; j < nb; ++j)
            for (int k = 0; k < n + 1; ++k) {
                const float v = sh_sc[j * BEAM_MAXK + k];
                if (v == -INFINITY) continue;
                sc[nc] = v; tk[nc] = sh_tk[j * BEAM_MAXK + k]; sr[nc] = r0 + j; ord[nc] = nc; ++nc;
            }
        for (int i = 1; i < nc; ++i) {                              // stable insertion sort, score descending
            const int o = ord[i]; int j = i - 1;
            while (j >= 0 && sc[ord[j]] < sc[o]) { ord[j + 1] = ord[j]; --j; }
            ord[j + 1] = o;
        }
        int saved = 0, nfin = 0;
        int count = s.fin_count[w];
        const int before = count;
        for (int i = 0; i < nc && saved < n; ++i) {
            const int c = ord[i];
            if (tk[c] == s.tok_eot) {
                if (count < s.max_candidates) {                     // newly finished, best first, while the list has room
                    s.fin_score[(int64_t)w * s.max_candidates + count] = sc[c];
                    sh_fin_src[nfin] = sr[c]; sh_fin_slot[nfin] = count; ++nfin; ++count;
                }
            } else {
                s.src[r0 + saved] = sr[c]; s.next_tok[r0 + saved] = tk[c]; s.sum_tmp[r0 + saved] = sc[c];
                ++saved;
            }
        }
        for (; saved < n; ++saved) {                                // cannot happen with >= n non-EOT proposals; keep rows defined
            s.src[r0 + saved] = r0; s.next_tok[r0 + saved] = s.tok_eot; s.sum_tmp[r0 + saved] = -INFINITY;
        }
        s.fin_count[w] = count;
        if (before < s.max_candidates && count >= s.max_candidates) atomicAdd(s.n_complete, 1);
        sh_nfin = nfin;
    }
    __syncthreads();
    for (int f = 0; f < sh_nfin; ++f) {                             // finished sequence = source prefix + EOT
        const int32_t* from = s.tokens + (int64_t)sh_fin_src[f] * s.ld_tok;
        int32_t* to = s.fin_tokens + ((int64_t)w * s.max_candidates + sh_fin_slot[f]) * s.ld_tok;
        for (int i = tid; i < s.ld_tok; i += 64) to[i] = i <= t ? from[i] : s.tok_eot;
    }
}

__global__ __launch_bounds__(256) void beam_gather_kernel(BeamArgs s) {
    const int r = blockIdx.x, t = *s.d_pos;
    const int from = s.src[r];
    for (int i = threadIdx.x; i <= t; i += 256) {
        s.tokens_tmp[(int64_t)r * s.ld_tok + i] = s.tokens[(int64_t)from * s.ld_tok + i];
        s.node_tmp[(int64_t)r * s.ld_node + i] = s.node[(int64_t)from * s.ld_node + i];
    }
}

__global__ __launch_bounds__(256) void beam_commit_kernel(BeamArgs s) {
    const int r = blockIdx.x, t = *s.d_pos, R = gridDim.x;
    for (int i = threadIdx.x; i <= t; i += 256) {
        s.tokens[(int64_t)r * s.ld_tok + i] = s.tokens_tmp[(int64_t)r * s.ld_tok + i];
        s.node[(int64_t)r * s.ld_node + i] = s.node_tmp[(int64_t)r * s.ld_node + i];
    }
    if (threadIdx.x == 0) {
        if (t + 1 < s.ld_tok) s.tokens[(int64_t)r * s.ld_tok + t + 1] = s.next_tok[r];
        if (t + 1 < s.ld_node) s.node[(int64_t)r * s.ld_node + t + 1] = (t + 1) * R + r;  // a fresh KV node for the next position
        s.sum_logprob[r] = s.sum_tmp[r];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(s.d_counter, 1);
        if (ticket == R - 1) { *s.d_counter = 0; *s.d_pos = t + 1; }
    }
}

int launch_beam_step(const BeamArgs& s, hipStream_t st) {
    if (s.B <= 0 || s.n < 1 || s.n > 8 || s.max_candidates < s.n || !s.logits || !s.tokens || !s.d_pos) return -1;
    if (s.ld_node <= 0 || s.ld_tok < s.ld_node) return -1;
    const int R = s.B * s.n;
    if (s.n + 1 <= 6) hipLaunchKernelGGL(beam_topk_kernel<6>, dim3(R), dim3(BT_NT), 0, st, s);
    else hipLaunchKernelGGL(beam_topk_kernel<BEAM_MAXK>, dim3(R), dim3(BT_NT), 0, st, s);
    hipLaunchKernelGGL(beam_select_kernel, dim3(s.B), dim3(64), 0, st, s);
    hipLaunchKernelGGL(beam_gather_kernel, dim3(R), dim3(256), 0, st, s);
    hipLaunchKernelGGL(beam_commit_kernel, dim3(R), dim3(256), 0, st, s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
