// common.h — shared types and launcher declarations for the gfx950 Whisper engine.
// Everything here targets CDNA4 (wave64, MFMA, 160 KB LDS) only; there is no other back end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wx {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

enum DType : int { DT_F32 = 0, DT_BF16 = 1 };
static inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

// GELU.  fp32 outputs (token-exact engines): the library erf.  bf16 outputs: erf from Abramowitz & Stegun 7.1.26 (|error| <=
// 1.5e-7, far below a bf16 output's rounding step): one exp2, one rcp and six fma instead of the library's two-branch polynomial
// -- the MLP up projection applies it to 5120 values per row.  Every bf16 kernel uses the same form, so a row's result does not
// depend on which GEMM kernel its batch size selects.
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float erfc_z = p * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);        // 1 - erf(z), z >= 0
    return 0.5f * x * (1.0f + copysignf(1.0f - erfc_z, x));
}

constexpr int HEAD_DIM = 64;      // every Whisper size uses 64-wide heads
constexpr int KV_PAGE = 16;       // tokens per page of the decoder self-attention KV cache
constexpr int N_FFT = 400;
constexpr int HOP = 160;
constexpr int N_FREQ = 201;
constexpr int WIN_FRAMES = 3000;  // mel frames per 30 s window
constexpr int RS_MAX_TAPS = 65536; // resampler filter table (wx_resample_pcm16)
constexpr int DG_MAX_ROWS = 80;    // rows one pass of the split-K decode GEMM path takes (5 MFMA row tiles): 64 greedy windows, or
                                   // 15 windows x 5 beams (BASELINE configs[3]: 120 windows of a 1 h file over 8 GPUs); larger greedy
                                   // batches run as several passes (engine.hip decode_step_bf16)

// ---- GEMM -------------------------------------------------------------------------------------------
// out[m][n] = epilogue( sum_k A[m][k] * W[n][k] ), A rows addressed as
//   A + (m / a_rows_per_batch) * a_batch_stride + (m % a_rows_per_batch) * lda      (elements)
// which makes a k=3 convolution over a zero-padded [T+2][C] activation an ordinary GEMM (lda = stride*C, K = 3C).
struct GemmEpilogue {
    const float* bias = nullptr;        // [N]
    int act = 0;                        // 1: exact-erf GELU
    const float* residual = nullptr;    // fp32 [M][ldr], added after activation
    int64_t ldr = 0;
    const float* rowadd = nullptr;      // fp32 [rowadd_period][N] added per (m % period) (encoder positional table)
    int rowadd_period = 0;
    int colscale_period = 0;            // v *= colscale[n / period] (q/k pre-scaling by head_dim^-0.25)
    float colscale[3] = {1.f, 1.f, 1.f};
    int mode = 0;                       // 0: row-major out[m*ldo + n]; 1: head split (below)
    void* out = nullptr;
    int out_dtype = DT_F32;
    int64_t ldo = 0;
    // row-major with per-batch row remap: out row = (m / rows_per_batch) * out_batch_rows + out_row_offset + m % rows_per_batch
    int out_rows_per_batch = 0;         // 0 = no remap
    int64_t out_batch_rows = 0, out_row_offset = 0;
    // head split: column n -> split s = n / hs_d, head h = (n % hs_d) / 64, c = n % 64; row m -> (b, t) = (m / hs_T, m % hs_T)
    // element goes to out_split[s][((b * hs_H + h) * hs_T + t) * 64 + c]
    void* out_split[3] = {nullptr, nullptr, nullptr};
    int hs_T = 0, hs_H = 0, hs_d = 0;
};

struct GemmArgs {
    const void* A = nullptr;
    int64_t lda = 0, a_rows_per_batch = 0, a_batch_stride = 0;
    const void* W = nullptr;
    int64_t ldw = 0;
    // optional bf16 copy of W in MFMA-fragment order, [N/16][K/32][64 lanes][8]: lane (gq, r) of block (u, kb) holds
    // W[u*16 + r][kb*32 + gq*8 .. +8], so a wave's 16-B-per-lane load is 1 KiB contiguous.  Read by the wide decode kernel.
    const void* Wpk = nullptr;
    int M = 0, N = 0, K = 0;
    int in_dtype = DT_F32;              // dtype of A and W
    // bf16, M <= DG_MAX_ROWS only: split K over n_split workgroups; raw fp32 partial tiles go to partial[split][M][N] and the
    // epilogue is NOT applied (the consumer adds the partials, bias, scale, residual itself)
    int n_split = 1;
    float* partial = nullptr;
    GemmEpilogue ep;
    // launch_gemm_acomb only (M <= ACOMB_ROWS): the activation is not read from A; row m, column k of it is the cross-attention
    // output of head k / 64, element k % 64, combined from the key splits' partial states ac_partial[(m*ac_H + h)*ac_nsplit + s][66]
    // exactly as attn_decode_combine_kernel combines them (attn_combine_splits below)
    const float* ac_partial = nullptr; int ac_nsplit = 0, ac_H = 0;
};
constexpr int ACOMB_ROWS = 4, ACOMB_MAX_SPLIT = 6;
// Returns 0 or a negative wx_status. Chooses the kernel from (dtype, M).
int launch_gemm(const GemmArgs& g, hipStream_t st);
// bf16 decode GEMM (M <= DG_MAX_ROWS) with K halved inside the workgroup and the full epilogue applied (no partial tiles);
// needs g.Wpk.  0 = launched, 1 = shape not covered (use the split-K path), < 0 = error.
int launch_gemm_khalf(const GemmArgs& g, hipStream_t st);
// The split-K decode GEMM behind the K/V-form cross-attention at <= ACOMB_ROWS rows, with the combine of the attention's key
// splits moved into its prologue (GemmArgs::ac_*): one launch less per layer where a launch is 3.4-4.4 us of a 53-us layer.
// Same combine arithmetic, same MFMA order.  0 = launched, 1 = shape not covered (launch the combine, then launch_gemm), < 0 = error.
int launch_gemm_acomb(const GemmArgs& g, hipStream_t st);
// bf16 W [N][K] (row stride ldw) -> the fragment order GemmArgs::Wpk describes; N % 16 == 0, K % 32 == 0
int launch_pack_gemm_weight(const void* W, int64_t ldw, int N, int K, void* out, hipStream_t st);
// K-split factor that fills the chip for a bf16 decode GEMM (M <= DG_MAX_ROWS) whose consumer can add partial tiles; 1 = none.
int decode_gemm_split(int N, int K);
bool decode_gemm_split_ok(int K, int s);

#if defined(__HIPCC__)
// Element d of one (row, head) of the dense-K/V decode attention from its key splits' states pp[s][66] = (acc[64], m, l): the one
// definition behind attn_decode_combine_kernel and the GEMM prologue that stands in for it.  All loads in flight at once (loads past
// n_split are clamped to the last split and not used).
__device__ __forceinline__ float attn_combine_splits(const float* __restrict__ pp, int n_split, int d) {
    float ms[ACOMB_MAX_SPLIT], ls[ACOMB_MAX_SPLIT], as[ACOMB_MAX_SPLIT];
#pragma unroll
    for (int s = 0; s < ACOMB_MAX_SPLIT; ++s) {
        const float* ps = pp + (s < n_split ? s : n_split - 1) * (HEAD_DIM + 2);
        ms[s] = ps[HEAD_DIM]; ls[s] = ps[HEAD_DIM + 1]; as[s] = ps[d];
    }
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < ACOMB_MAX_SPLIT; ++s) if (s < n_split) M = fmaxf(M, ms[s]);
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int s = 0; s < ACOMB_MAX_SPLIT; ++s) {
        if (s < n_split) {
            const float f = (ms[s] == -INFINITY) ? 0.f : exp2f((ms[s] - M) * 1.4426950408889634f);
            L = fmaf(ls[s], f, L);
            acc = fmaf(as[s], f, acc);
        }
    }
    return acc / L;
}
#endif

// ---- elementwise / norms ---------------------------------------------------------------------------
int launch_layernorm(const float* x, const float* gamma, const float* beta, void* out, int out_dtype,
                     int rows, int d, hipStream_t st);
// x[r][:] += bias + sum_s partial[s][r][:] (fixed order), written back, then LayerNorm(x) -> out.  Consumer side of a
// split-K decode GEMM whose epilogue is "residual add" (n_split == 0: plain LayerNorm, x untouched).
int launch_reduce_layernorm(float* x, const float* partial, int n_split, const float* bias, const float* gamma,
                            const float* beta, void* out, int out_dtype, int rows, int d, hipStream_t st);
// out[r][:] = bf16(gelu(bias + sum_s partial[s][r][:])): consumer side of a split-K decode GEMM with a GELU epilogue
int launch_reduce_bias_gelu(const float* partial, int n_split, const float* bias, void* out_bf16, int rows, int N, hipStream_t st);
// x[b][:] = tok_emb[tokens[b*ld_tok + *d_pos]][:] + pos_emb[*d_pos][:]   (fp32 out; embedding tables in `dt`)
// group > 1 (prompt pass): row r embeds position *d_pos + r % group of sequence r / group (rows = sequences x group)
// n_tok / n_pos: rows of the two tables; an id or position outside them is clamped (a row of NaN logits ends with EOT in the
// sampler, but nothing that reads a table trusts an id it did not check)
int launch_embed(const void* tok_emb, const void* pos_emb, int dt, const int32_t* tokens, int ld_tok,
                 const int32_t* d_pos, float* x, int B, int d, hipStream_t st, int group, int n_tok, int n_pos);
int launch_convert_f32(const float* in, void* out, int out_dtype, int64_t n, hipStream_t st);
int launch_to_f32(const void* in, int in_dtype, float* out, int64_t n, hipStream_t st);
// mel (w, m, t) -> conv input xpad[w][1 + t][c_pad] (dtype dt); zero rows 0 and T+1, zero channels >= n_mels,
// zero for global frame w*3000 + t >= n_valid_frames.
int launch_mel_to_conv_input(const float* mel, int64_t win_stride, int64_t row_stride, int64_t n_valid_frames,
                             void* xpad, int dt, int B, int n_mels, int c_pad, hipStream_t st);
int launch_zero_pad_rows(void* xpad, int dt, int B, int rows_per_batch, int C, hipStream_t st);

// ---- mel front end ------------------------------------------------------------------------------------
struct MelTables {                       // device tables built at wx_finalize
    const float* window = nullptr;       // [400] periodic Hann
    const float* cs = nullptr;           // [400][2] cos, sin of 2*pi*i/400
    const float* filters = nullptr;      // packed non-zero runs of the [n_mels][201] filterbank, n_pack floats
    const int32_t* frange = nullptr;     // [n_mels][4]: first / last+1 non-zero bin of each filter, offset of its run, 0
    int n_mels = 0, n_pack = 0;
};
int launch_mel_stage1(const MelTables& t, const float* pcm, int64_t pcm_offset, int64_t n_valid, int64_t n_file,
                      int64_t n_total, int64_t frame0, int64_t n_frames, float* out, int64_t ld, float* d_max,
                      hipStream_t st, float* block_max = nullptr);
// block_max != nullptr: every workgroup stores its maximum to block_max[blockIdx] (mel_stage1_blocks(n_frames) floats)
// instead of folding it into *d_max; launch_mel_fold_block_max reduces them to one float
int mel_stage1_blocks(int64_t n_frames);
int launch_mel_fold_block_max(const float* part, int n, float* out, hipStream_t st);
int launch_mel_stage2(float* x, int64_t ld, int n_mels, int64_t n_frames, const float* d_max, hipStream_t st);
int launch_fill_f32(float* p, float v, int64_t n, hipStream_t st);
int launch_pcm16_to_f32(const int16_t* in, float* out, int64_t n, hipStream_t st);   // x / 32768
// channel average + polyphase resampling of interleaved PCM16 to mono float32 (see wx_resample_pcm16); taps on the device
int launch_resample_pcm16(const int16_t* in, int64_t n_frames, int n_channels, int up, int down, const float* d_taps,
                          int half_len, int quantize, float* out, int64_t n_out, hipStream_t st);

// ---- attention ----------------------------------------------------------------------------------------
struct AttnArgs {
    int dt = DT_F32;
    // queries: q(b, iq, h) at Q + b*q_sb + iq*q_sq + h*q_sh
    const void* Q = nullptr; int64_t q_sb = 0, q_sq = 0, q_sh = HEAD_DIM;
    int n_q = 1;                         // queries per (b, h)
    // decode only: the query as split-K partial tiles of its projection GEMM (Q == nullptr):
    // q(b, h)[c] = (q_bias[h*64 + c] + sum_s q_partial[(s*B + b)*q_sb + h*64 + c]) * q_scale
    const float* q_partial = nullptr; const float* q_bias = nullptr; int q_nsplit = 0; float q_scale = 1.f;
    // dense keys/values: K + (b*H + h)*kv_sbh + t*64
    const void* K = nullptr; const void* V = nullptr; int64_t kv_sbh = 0;
    int n_keys = 0;                      // dense key count (ignored when d_pos != nullptr)
    // paged self-attention cache (decoder): page_table[b*pt_ld + t/PS] -> page p; row at ((p*H + h)*PS + t%PS)*64 with
    // PS = page_size: KV_PAGE (16) for greedy decoding; 1 for beam search, where a sequence's cache is the list of
    // token nodes on its path through the beam tree (reordering beams = gathering index rows, no K/V is ever copied)
    const int32_t* page_table = nullptr; int pt_ld = 0; int page_size = KV_PAGE;
    // cross-attention: kv_group consecutive query rows (the beams of one window) share one K/V entry: row b reads the
    // cache of window b / kv_group, and one workgroup streams it once for the whole group
    int kv_group = 1;
    void* Kpool = nullptr; void* Vpool = nullptr;   // written at position *d_pos with the new token's k/v
    const void* Knew = nullptr; const void* Vnew = nullptr; int64_t new_sb = 0;  // k/v of the current token: + b*new_sb + h*64
    const int32_t* d_pos = nullptr;      // device scalar: current position t (keys 0..t are attended)
    // prompt pass (paged self-attention only): q_group consecutive rows are consecutive positions *d_pos + r % q_group of
    // ONE sequence r / q_group (its page-table row, its cache); their k/v rows are appended by launch_kv_append first
    int q_group = 1;
    // output o(b, iq, h) at O + b*o_sb + iq*o_sq + h*o_sh
    void* O = nullptr; int64_t o_sb = 0, o_sq = 0, o_sh = HEAD_DIM;
    int B = 0, H = 0;
    // split-key workspace (used when few (b,h) pairs would leave the chip idle)
    float* partial = nullptr; int n_split = 1;
    bool defer_combine = false;          // dense K/V, n_split > 1: leave the key splits' states in `partial`; the caller's next kernel combines them
};
int launch_attn_decode(const AttnArgs& a, hipStream_t st);     // n_q == 1, waves split the keys
int launch_attn_combine(const AttnArgs& a, hipStream_t st);    // the combine a deferred launch_attn_decode left out
int launch_kv_append(const AttnArgs& a, hipStream_t st);       // prompt pass: write the k/v rows of a.B rows into the paged cache
int launch_attn_rows(const AttnArgs& a, hipStream_t st);       // many queries, non-causal, dense K/V (fp32 encoder path)
int launch_attn_flash_bf16(const AttnArgs& a, hipStream_t st); // MFMA flash attention, dense bf16 K/V, non-causal

// ---- decoder cross-attention on the encoder output (xattn.hip) ---------------------------------------------
struct XQTildeArgs {                     // q~[row][h][:] = s * Wk_h^T bf16((bias + sum of split-K tiles) * s)
    const float* q_partial = nullptr; const float* q_bias = nullptr; int q_nsplit = 0;   // [split][rows][d] tiles of the query projection
    const bf16* Wk_pk = nullptr;         // cross-attention key weight, per head in MFMA-fragment order [H][d / 16][2][64][8] (xattn.hip)
    bf16* qt = nullptr;                  // [rows][H][d]
    float scale = 1.f;                   // head_dim^-0.25
    int rows = 0, H = 0, d = 0;
};
struct XAttnArgs {                       // one workgroup per (row, key split)
    const bf16* qt = nullptr;            // [rows][H][d]
    const bf16* E = nullptr;             // encoder output in MFMA-fragment order [windows][nt16][d/32][64][8] (launch_xattn_pack_e);
                                         // row r attends window r / group
    bf16* cpart = nullptr;               // [rows][n_split][H][d]: context of the split's keys, normalised by the split's own sum
    float* ml = nullptr;                 // [rows][n_split][H][2]: running maximum and sum of the split
    int rows = 0, H = 0, d = 0, T = 0, group = 1, n_split = 1;
    int nt16 = 0;                        // 16-key tiles stored per window: even, >= ceil(T / 16)
};
struct XCProjArgs {                      // out[row][h*64 + c] = bv + Wv_h . bf16(sum_s w_s cpart_s)
    const bf16* cpart = nullptr; const float* ml = nullptr; int n_split = 1;
    const bf16* Wv_pk = nullptr;         // cross-attention value weight [d][d] in MFMA-fragment order [d / 16][d / 32][64][8]
    const float* bv = nullptr;           // [d]
    bf16* out = nullptr; int64_t ldo = 0;
    int rows = 0, H = 0, d = 0;
};
int launch_xattn_pack_e(const void* E, void* Ef, int windows, int T, int D, int nt16, hipStream_t st);
int launch_xattn_qtilde(const XQTildeArgs& a, hipStream_t st);
int launch_xattn_estream(const XAttnArgs& a, hipStream_t st);
int launch_xattn_cproj(const XCProjArgs& a, hipStream_t st);

// ---- sampler ------------------------------------------------------------------------------------------
struct SamplerArgs {
    const float* logits = nullptr; int64_t ldl = 0; int n_vocab = 0;
    int32_t* tokens = nullptr; int ld_tok = 0;     // [B][n_text_ctx + 1]
    int32_t* d_pos = nullptr;                      // incremented by the last block
    int32_t* d_block_counter = nullptr;            // zero-initialised scratch counter
    const uint8_t* suppress = nullptr;             // [n_vocab] 1 = masked
    float* sum_logprob = nullptr; float* no_speech = nullptr;
    int32_t* n_finished = nullptr;                 // device counter of sequences that have emitted EOT
    int B = 0, n_prompt = 0, sot_index = 0;        // prompt length / SOT position shared by all rows ...
    const int32_t* row_n_prompt = nullptr;         // ... or per row (device arrays [B]): sequential long-form decoding
    const int32_t* row_sot = nullptr;              //     carries a different amount of previous text into every window
    int sample_len = 1 << 30;                      // per-row cap on sampled tokens (rows with short prompts stop first)
    int tok_eot = 0, tok_blank = 0, tok_no_speech = 0, tok_no_timestamps = 0, tok_timestamp_begin = 0;
    int without_timestamps = 1, suppress_blank = 1, max_initial_ts = 50;
    int forced = 0;                                // 1: teacher forcing, tokens are never written
    int group = 1;                                 // prompt pass: B rows = sequences x group positions; row r is position
                                                   // *d_pos + r % group of sequence r / group; the position advances by group
    float temperature = 0.f;                       // > 0: sample from softmax(logits / T) (Gumbel-max, counter-based RNG)
    uint32_t seed = 0;
    // when set, {temperature bits, seed, sample_len} are read from device memory instead, so a captured decode-step graph
    // stays valid across calls that only change them (temperature fallback re-decodes the same window)
    const uint32_t* dyn = nullptr;
    // sampling steps only: `parts` workgroups scan a row (1, 2, 4 or 8) and leave records in part_buf [B][parts][16] floats
    // for a small finishing launch; 1 = one workgroup per row does everything
    int parts = 1;
    float* part_buf = nullptr;
};
int launch_sampler(const SamplerArgs& s, hipStream_t st);
// n_tokens[b] = sampled ids before the first EOT; the tail of every row is EOT padded.  row_n_prompt may be null.
int launch_finish_tokens(int32_t* tokens, int ld_tok, int n_prompt, const int32_t* row_n_prompt, int tok_eot, const int32_t* d_pos,
                         int32_t* n_tokens, int B, hipStream_t st);

// ---- beam search ----------------------------------------------------------------------------------------
struct BeamArgs {
    const float* logits = nullptr; int64_t ldl = 0; int n_vocab = 0;
    int32_t* tokens = nullptr; int32_t* tokens_tmp = nullptr; int ld_tok = 0;   // [R][n_text_ctx + 1]
    int32_t* node = nullptr; int32_t* node_tmp = nullptr; int ld_node = 0;      // [R][n_text_ctx] KV node of every position
    float* sum_logprob = nullptr; float* sum_tmp = nullptr;                     // [R]
    float* cand_score = nullptr; int32_t* cand_tok = nullptr;                   // [R][9] proposals, best first
    int32_t* src = nullptr; int32_t* next_tok = nullptr;                        // [R]
    int32_t* fin_tokens = nullptr; float* fin_score = nullptr; int32_t* fin_count = nullptr;   // [B][max_candidates][ctx], [B][mc], [B]
    int32_t* d_pos = nullptr; int32_t* d_counter = nullptr; int32_t* n_complete = nullptr;
    const uint8_t* suppress = nullptr; float* no_speech = nullptr;              // no_speech: [B]
    int B = 0, n = 0, max_candidates = 0, n_prompt = 0, sot_index = 0;
    int tok_eot = 0, tok_blank = 0, tok_no_speech = 0, tok_no_timestamps = 0, tok_timestamp_begin = 0;
    int without_timestamps = 1, suppress_blank = 1, max_initial_ts = 50;
};
int launch_beam_step(const BeamArgs& s, hipStream_t st);   // topk -> select -> gather -> commit (position += 1)

}  // namespace wx
