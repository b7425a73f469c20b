// elementwise.hip — LayerNorm, embedding, layout changes and the greedy sampler (logit filters + argmax +
// log-softmax bookkeeping) for gfx950.  All of these are HBM/L2-bound streaming kernels: 16-byte vector accesses,
// one wave per row where a row reduction is needed.
#include "common.h"
#include <cstdlib>

namespace wx {

// ---------------------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5, affine), fp32 in, fp32 or bf16 out.  One wave per row, two-pass statistics in registers.
// ---------------------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TO* __restrict__ out, int rows, int d) {
    constexpr int MAXV = 8;                       // d <= 64 * 4 * 8 = 2048
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)row * d);
    const int nv = d / 4;
    f32x4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        v[i] = (c < nv) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float t = v[i][j] - mean; sq = fmaf(t, t, sq); }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = 1.0f / sqrtf(sq / (float)d + 1e-5f);
    const f32x4* gr = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* br = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 g = gr[c], bb = br[c];
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * g[j] + bb[j];
            if constexpr (sizeof(TO) == 4) {
                reinterpret_cast<f32x4*>(out + (int64_t)row * d)[c] = y;
            } else {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)y[j];
                reinterpret_cast<bf16x4*>(out + (int64_t)row * d)[c] = o;
            }
        }
    }
}

int launch_layernorm(const float* x, const float* gamma, const float* beta, void* out, int out_dtype, int rows, int d,
                     hipStream_t st) {
    if (d % 4 != 0 || d > 2048 || rows <= 0) return -1;
    const int blocks = (rows + 3) / 4;
    if (out_dtype == DT_F32) hipLaunchKernelGGL(layernorm_kernel<float>, dim3(blocks), dim3(256), 0, st, x, gamma, beta, (float*)out, rows, d);
    else hipLaunchKernelGGL(layernorm_kernel<bf16>, dim3(blocks), dim3(256), 0, st, x, gamma, beta, (bf16*)out, rows, d);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Consumer side of a split-K decode GEMM with a residual epilogue: x += bias + sum_s partial[s] (fixed order, so the
// result is deterministic), then the next sub-layer's LayerNorm of the updated row.  One WORKGROUP per row (a decode
// batch has <= 64 rows, and a CU sustains only ~25 GB/s of loads: the n_split*d*4 bytes of a row are spread over as
// many CUs as there are rows), 16 B per lane, two-pass statistics through LDS.
template <typename TO, int NT, int MAX_SPLIT>
__global__ __launch_bounds__(NT) void reduce_layernorm_kernel(float* __restrict__ x, const float* __restrict__ partial, int n_split,
                                                              const float* __restrict__ bias, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, TO* __restrict__ out, int rows, int d) {
    constexpr int MAXV = (2048 + NT * 4 - 1) / (NT * 4);       // d <= 2048; NT = 384 covers d <= 1536 with one chunk per lane
    constexpr int NW = NT / 64;                                // MAX_SPLIT: partial tiles fetched per chunk (>= n_split)
    __shared__ float red[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, row = blockIdx.x;
    f32x4* xr = reinterpret_cast<f32x4*>(x + (int64_t)row * d);
    const int nv = d / 4;
    f32x4 v[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = tid + i * NT;
        v[i] = (c < nv) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (n_split > 0) {
        const f32x4* br = reinterpret_cast<const f32x4*>(bias);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = tid + i * NT;
            if (c < nv) {
                // all partial tiles in flight at once: a runtime-bounded loop would serialise n_split round trips.
                // Loads past n_split are clamped to the last tile and not added (no load under a branch).
                f32x4 pt[MAX_SPLIT];
#pragma unroll
                for (int s = 0; s < MAX_SPLIT; ++s) {
                    const int sc = s < n_split ? s : n_split - 1;
                    pt[s] = reinterpret_cast<const f32x4*>(partial + ((int64_t)sc * rows + row) * d)[c];
                }
                f32x4 t = pt[0];
#pragma unroll
                for (int s = 1; s < MAX_SPLIT; ++s) if (s < n_split) t += pt[s];
                v[i] += t + br[c];
                xr[c] = v[i];
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w];
    const float mean = tot / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = tid + i * NT;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float t = v[i][j] - mean; sq = fmaf(t, t, sq); }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) red[NW + wid] = sq;
    __syncthreads();
    float tsq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tsq += red[NW + w];
    const float rstd = 1.0f / sqrtf(tsq / (float)d + 1e-5f);
    const f32x4* gr = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* be = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = tid + i * NT;
        if (c < nv) {
            const f32x4 gg = gr[c], bb = be[c];
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
            if constexpr (sizeof(TO) == 4) {
                reinterpret_cast<f32x4*>(out + (int64_t)row * d)[c] = y;
            } else {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)y[j];
                reinterpret_cast<bf16x4*>(out + (int64_t)row * d)[c] = o;
            }
        }
    }
}

// Same reduce + LayerNorm with TWO lanes per 16-B chunk (lanes l and l + 32 of a wave): each fetches half of the partial
// tiles, one shuffle joins them.  Halves the loads per lane again (7 instead of 12 at split 10); 32 chunks per wave.
template <typename TO, int NW, int HS>       // NW waves (32 * NW chunks >= d / 4), HS = tiles per half (>= ceil(n_split / 2))
__global__ __launch_bounds__(NW * 64) void reduce_layernorm2_kernel(float* __restrict__ x, const float* __restrict__ partial, int n_split,
                                                                    const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, TO* __restrict__ out, int rows, int d) {
    __shared__ float red[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, row = blockIdx.x;
    const int half = lane >> 5, c = wid * 32 + (lane & 31), nv = d / 4;
    const bool live = c < nv;
    const int cc = live ? c : nv - 1;
    const int h0 = (n_split + 1) / 2;                      // tiles 0 .. h0-1 on half 0, h0 .. n_split-1 on half 1
    const int s_lo = half ? h0 : 0, s_n = half ? n_split - h0 : h0;
    f32x4 pt[HS];
#pragma unroll
    for (int s = 0; s < HS; ++s) {
        int sc = s_lo + (s < s_n ? s : s_n - 1); if (sc > n_split - 1) sc = n_split - 1; if (sc < 0) sc = 0;
        pt[s] = reinterpret_cast<const f32x4*>(partial + ((int64_t)sc * rows + row) * d)[cc];
    }
    // half 0 also carries the residual row, half 1 the bias
    const f32x4 extra = half ? reinterpret_cast<const f32x4*>(bias)[cc] : reinterpret_cast<const f32x4*>(x + (int64_t)row * d)[cc];
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < HS; ++s) if (s < s_n) t += pt[s];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __shfl_xor(t[j], 32);
    f32x4 e2;
#pragma unroll
    for (int j = 0; j < 4; ++j) e2[j] = __shfl_xor(extra[j], 32);
    // v = x + ((tiles of half 0 + tiles of half 1) + bias), the same expression on both lanes
    const f32x4 t0 = half ? o : t, t1 = half ? t : o, xv = half ? e2 : extra, bv = half ? extra : e2;
    f32x4 v = xv + ((t0 + t1) + bv);
    if (!live) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live && half == 0) reinterpret_cast<f32x4*>(x + (int64_t)row * d)[c] = v;
    float sum = half == 0 ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) red[wid] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w];
    const float mean = tot / (float)d;
    float sq = 0.f;
    if (live && half == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float u = v[j] - mean; sq = fmaf(u, u, sq); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) red[NW + wid] = sq;
    __syncthreads();
    float tsq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tsq += red[NW + w];
    const float rstd = 1.0f / sqrtf(tsq / (float)d + 1e-5f);
    if (live && half == 1) {
        const f32x4 gg = reinterpret_cast<const f32x4*>(gamma)[c], bb = reinterpret_cast<const f32x4*>(beta)[c];
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[j] - mean) * rstd * gg[j] + bb[j];
        if constexpr (sizeof(TO) == 4) {
            reinterpret_cast<f32x4*>(out + (int64_t)row * d)[c] = y;
        } else {
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) ob[j] = (bf16)y[j];
            reinterpret_cast<bf16x4*>(out + (int64_t)row * d)[c] = ob;
        }
    }
}

int launch_reduce_layernorm(float* x, const float* partial, int n_split, const float* bias, const float* gamma, const float* beta,
                            void* out, int out_dtype, int rows, int d, hipStream_t st) {
    if (d % 4 != 0 || d > 2048 || rows <= 0 || n_split < 0 || n_split > 16 || (n_split > 0 && (!partial || !bias))) return -1;
    const int blocks = rows;
    // one 16-B chunk per lane when the row fits 384 lanes (d = 1280): half as many loads per lane as with 256
    const bool wide = d > 1024 && d <= 1536;
#define WX_RLN_LAUNCH(TO_, NT_, MS_) hipLaunchKernelGGL((reduce_layernorm_kernel<TO_, NT_, MS_>), dim3(blocks), dim3(NT_), 0, st, x, partial, n_split, bias, gamma, beta, (TO_*)out, rows, d)
#define WX_RLN_SPLIT(TO_, NT_) do { if (n_split <= 4) WX_RLN_LAUNCH(TO_, NT_, 4); else if (n_split <= 10) WX_RLN_LAUNCH(TO_, NT_, 10); else WX_RLN_LAUNCH(TO_, NT_, 16); } while (0)
    // bf16 decode path: two lanes per chunk (3.64 -> 3.20 us at d = 1280, split 10); WX_RLN1=1 keeps one lane per chunk
    static const bool one_lane = getenv("WX_RLN1") != nullptr;
    if (!one_lane && out_dtype == DT_BF16 && n_split >= 2 && n_split <= 10 && d <= 1280) {
#define WX_RLN2_LAUNCH(NW_) do { \
        if (n_split <= 4) hipLaunchKernelGGL((reduce_layernorm2_kernel<bf16, NW_, 2>), dim3(blocks), dim3(NW_ * 64), 0, st, x, partial, n_split, bias, gamma, beta, (bf16*)out, rows, d); \
        else hipLaunchKernelGGL((reduce_layernorm2_kernel<bf16, NW_, 5>), dim3(blocks), dim3(NW_ * 64), 0, st, x, partial, n_split, bias, gamma, beta, (bf16*)out, rows, d); } while (0)
        const int nv = d / 4;
        if (nv <= 32 * 3) WX_RLN2_LAUNCH(3); else if (nv <= 32 * 4) WX_RLN2_LAUNCH(4); else if (nv <= 32 * 6) WX_RLN2_LAUNCH(6);
        else if (nv <= 32 * 8) WX_RLN2_LAUNCH(8); else WX_RLN2_LAUNCH(10);
#undef WX_RLN2_LAUNCH
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    // as few clamped (wasted) tile loads as the split allows: 4, 10 or 16 per chunk
    if (wide) { if (out_dtype == DT_F32) WX_RLN_SPLIT(float, 384); else WX_RLN_SPLIT(bf16, 384); }
    else { if (out_dtype == DT_F32) WX_RLN_SPLIT(float, 256); else WX_RLN_SPLIT(bf16, 256); }
#undef WX_RLN_SPLIT
#undef WX_RLN_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Consumer side of a split-K decode GEMM with a GELU epilogue (the MLP's first projection):
// out = bf16(gelu(bias + sum_s partial[s])), fixed summation order.  Pure streaming, 16 B per lane, whole chip.
__global__ __launch_bounds__(256) void reduce_bias_gelu_kernel(const float* __restrict__ partial, int n_split, int64_t split_stride,
                                                               const float* __restrict__ bias, bf16* __restrict__ out, int64_t total4, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    f32x4 v;
    if (n_split == 2) {           // the usual case (fc1 at d = 1280): exactly two loads, none clamped
        const f32x4 p0 = reinterpret_cast<const f32x4*>(partial)[i], p1 = reinterpret_cast<const f32x4*>(partial + split_stride)[i];
        v = p0 + p1;
    } else {
        f32x4 pt[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) pt[s] = reinterpret_cast<const f32x4*>(partial + (s < n_split ? s : n_split - 1) * split_stride)[i];
        v = pt[0];
#pragma unroll
        for (int s = 1; s < 4; ++s) if (s < n_split) v += pt[s];
        for (int s = 4; s < n_split; ++s) v += reinterpret_cast<const f32x4*>(partial + s * split_stride)[i];
    }
    v += *reinterpret_cast<const f32x4*>(bias + (i * 4) % N);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16)gelu_erf_fast(v[j]);
    reinterpret_cast<bf16x4*>(out)[i] = o;
}

int launch_reduce_bias_gelu(const float* partial, int n_split, const float* bias, void* out_bf16, int rows, int N, hipStream_t st) {
    if (rows <= 0 || N % 4 != 0 || n_split < 1 || !partial || !bias || !out_bf16) return -1;
    const int64_t total4 = (int64_t)rows * N / 4;
    hipLaunchKernelGGL(reduce_bias_gelu_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, partial, n_split,
                       (int64_t)rows * N, bias, (bf16*)out_bf16, total4, N);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------
// token + positional embedding for the current decode position
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(const T* __restrict__ tok_emb, const T* __restrict__ pos_emb, const int32_t* __restrict__ tokens,
                             int ld_tok, const int32_t* __restrict__ d_pos, float* __restrict__ x, int d, int group, int n_tok, int n_pos) {
    const int r = blockIdx.x, b = r / group;
    int t = *d_pos + r % group; if (t > n_pos - 1) t = n_pos - 1; if (t < 0) t = 0;          // ids index two tables: never outside them
    int token = tokens[b * ld_tok + t]; if ((unsigned)token >= (unsigned)n_tok) token = n_tok - 1;
    const T* te = tok_emb + (int64_t)token * d;
    const T* pe = pos_emb + (int64_t)t * d;
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(int64_t)r * d + i] = (float)te[i] + (float)pe[i];
}

int launch_embed(const void* tok_emb, const void* pos_emb, int dt, const int32_t* tokens, int ld_tok, const int32_t* d_pos,
                 float* x, int B, int d, hipStream_t st, int group, int n_tok, int n_pos) {
    if (group < 1 || n_tok < 1 || n_pos < 1) return -1;
    if (dt == DT_F32) hipLaunchKernelGGL(embed_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)tok_emb, (const float*)pos_emb, tokens, ld_tok, d_pos, x, d, group, n_tok, n_pos);
    else hipLaunchKernelGGL(embed_kernel<bf16>, dim3(B), dim3(256), 0, st, (const bf16*)tok_emb, (const bf16*)pos_emb, tokens, ld_tok, d_pos, x, d, group, n_tok, n_pos);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------
// dtype conversion
// ---------------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void convert_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (TO)(float)in[i];
}
static int grid_for(int64_t n) { int64_t b = (n + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }

int launch_convert_f32(const float* in, void* out, int out_dtype, int64_t n, hipStream_t st) {
    if (out_dtype == DT_F32) hipLaunchKernelGGL((convert_kernel<float, float>), dim3(grid_for(n)), dim3(256), 0, st, in, (float*)out, n);
    else hipLaunchKernelGGL((convert_kernel<float, bf16>), dim3(grid_for(n)), dim3(256), 0, st, in, (bf16*)out, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int launch_to_f32(const void* in, int in_dtype, float* out, int64_t n, hipStream_t st) {
    if (in_dtype == DT_F32) hipLaunchKernelGGL((convert_kernel<float, float>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)in, out, n);
    else hipLaunchKernelGGL((convert_kernel<bf16, float>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)in, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------
// mel [w][m][t] -> conv input xpad[w][1 + t][c_pad] (time-major, channel-last, one zero row either side)
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mel_to_conv_kernel(const float* __restrict__ mel, int64_t win_stride, int64_t row_stride,
                                                          int64_t n_valid_frames, T* __restrict__ xpad, int n_mels, int c_pad) {
    __shared__ float tile[32][33];
    const int w = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        float v = 0.f;
        if (c < n_mels && t < WIN_FRAMES && (int64_t)w * WIN_FRAMES + t < n_valid_frames)
            v = mel[(int64_t)w * win_stride + (int64_t)c * row_stride + t];
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        if (t < WIN_FRAMES && c < c_pad) xpad[((int64_t)w * (WIN_FRAMES + 2) + 1 + t) * c_pad + c] = (T)tile[tx][ty + i * 8];
    }
}

int launch_mel_to_conv_input(const float* mel, int64_t win_stride, int64_t row_stride, int64_t n_valid_frames, void* xpad,
                             int dt, int B, int n_mels, int c_pad, hipStream_t st) {
    dim3 grid((WIN_FRAMES + 31) / 32, (c_pad + 31) / 32, B);
    if (dt == DT_F32) hipLaunchKernelGGL(mel_to_conv_kernel<float>, grid, dim3(256), 0, st, mel, win_stride, row_stride, n_valid_frames, (float*)xpad, n_mels, c_pad);
    else hipLaunchKernelGGL(mel_to_conv_kernel<bf16>, grid, dim3(256), 0, st, mel, win_stride, row_stride, n_valid_frames, (bf16*)xpad, n_mels, c_pad);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// zero the first and last row of every [rows_per_batch][C] slab
template <typename T>
__global__ void zero_pad_rows_kernel(T* __restrict__ x, int rows_per_batch, int C) {
    const int b = blockIdx.x;
    T* first = x + (int64_t)b * rows_per_batch * C;
    T* last = first + (int64_t)(rows_per_batch - 1) * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) { first[i] = (T)0.f; last[i] = (T)0.f; }
}
int launch_zero_pad_rows(void* xpad, int dt, int B, int rows_per_batch, int C, hipStream_t st) {
    if (dt == DT_F32) hipLaunchKernelGGL(zero_pad_rows_kernel<float>, dim3(B), dim3(256), 0, st, (float*)xpad, rows_per_batch, C);
    else hipLaunchKernelGGL(zero_pad_rows_kernel<bf16>, dim3(B), dim3(256), 0, st, (bf16*)xpad, rows_per_batch, C);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------
// Sampler: upstream DecodingTask logit filters + GreedyDecoder.update at temperature 0, one workgroup per sequence.
//   SuppressBlank (first sampled position), SuppressTokens (byte mask), ApplyTimestampRules (optional),
//   argmax (lowest index wins ties), log-softmax of the filtered row for sum_logprob, no-speech probability from
//   the UNfiltered row at the SOT position, EOT stickiness, device-side finished counter, step counter advance.
// ---------------------------------------------------------------------------------------------------------
struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ MaxIdx wave_best(MaxIdx x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        MaxIdx y; y.v = __shfl_xor(x.v, off); y.i = __shfl_xor(x.i, off);
        x = better(x, y);
    }
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
    return x;
}

// online (max, sum exp) pair
struct MS { float m, s; };
// __expf (v_exp_f32, ~2 ulp) instead of expf: the sum feeds one logf per row, and the ~15-instruction expf made the
// sampler ALU-bound (one call per vocabulary entry)
__device__ __forceinline__ void ms_add(MS& a, float v) {
    if (v > a.m) { a.s = a.s * __expf(a.m - v) + 1.f; a.m = v; } else a.s += __expf(v - a.m);
}
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
    if (b.m == -INFINITY) return a;
    if (a.m == -INFINITY) return b;
    const float m = fmaxf(a.m, b.m);
    return MS{m, a.s * expf(a.m - m) + b.s * expf(b.m - m)};
}
__device__ __forceinline__ MS wave_ms(MS x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        MS y; y.m = __shfl_xor(x.m, off); y.s = __shfl_xor(x.s, off);
        x = ms_merge(x, y);
    }
    return x;
}

constexpr int SAMP_NT = 1024, SAMP_NW = SAMP_NT / 64;

// One workgroup of 16 waves per sequence; ONE pass over the 200 KB logit row (16-B loads, suppress mask read 4 bytes at a
// time, online max / sum-exp), because a CU sustains only ~25 GB/s of loads and a second pass would double the time.
// candidate under selection: key = logit (temperature 0) or logit / T + Gumbel noise; raw = the unscaled logit
struct Pick { float key; int i; float raw; };
__device__ __forceinline__ Pick pick_better(Pick a, Pick b) {
    if (b.key > a.key || (b.key == a.key && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ Pick wave_pick(Pick x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Pick y; y.key = __shfl_xor(x.key, off); y.i = __shfl_xor(x.i, off); y.raw = __shfl_xor(x.raw, off);
        x = pick_better(x, y);
    }
    return x;
}
// counter-based uniform in (0, 1): one hash of (seed, row, position, token) per draw, no RNG state to carry
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t row, uint32_t pos, uint32_t idx) {
    uint32_t h = seed ^ (row * 0x9E3779B1u) ^ (pos * 0x85EBCA77u) ^ (idx * 0xC2B2AE3Du);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// last step of a row's sampling: choose between the best candidate overall and the best timestamp, score it, write the id
__device__ __forceinline__ void sampler_commit(const SamplerArgs& s, int b, int t, bool ts_mode, Pick best_all, Pick best_ts,
                                               float max_text, MS ms_all, MS ms_ts, int prev_id, float prev_sum) {
    // log-probabilities are always those of the UNSCALED filtered logits (upstream GreedyDecoder.update)
    const float lse_all = ms_all.m + logf(ms_all.s);
    int chosen = best_all.i;
    float logprob = best_all.raw - lse_all;
    if (ts_mode && ms_ts.m > -INFINITY) {
        const float lse_ts = ms_ts.m + logf(ms_ts.s);
        const float ts_lp = lse_ts - lse_all, text_lp = max_text - lse_all;
        if (ts_lp > text_lp) { chosen = best_ts.i; logprob = best_ts.raw - lse_ts; }   // only timestamps remain
    }
    // a row whose logits are all NaN (NaN weights or audio) has no best candidate: it ends here instead of carrying an id that
    // indexes nothing
    if ((unsigned)chosen >= (unsigned)s.n_vocab) { chosen = s.tok_eot; logprob = 0.f; }
    const bool alive = prev_id != s.tok_eot;
    if (alive) s.sum_logprob[b] = prev_sum + logprob;
    const int next = alive ? chosen : s.tok_eot;
    s.tokens[(int64_t)b * s.ld_tok + t + 1] = next;
    if (alive && next == s.tok_eot) atomicAdd(s.n_finished, 1);
}
// one butterfly step over (Pick, Pick, max, MS, MS): merge as (lower lane's value, upper lane's value) on both sides
__device__ __forceinline__ void sampler_butterfly(int lane, int off, Pick& best_all, Pick& best_ts, float& max_text, MS& ms_all, MS& ms_ts) {
    Pick pa, pt; MS ma, mt;
    pa.key = __shfl_xor(best_all.key, off); pa.i = __shfl_xor(best_all.i, off); pa.raw = __shfl_xor(best_all.raw, off);
    pt.key = __shfl_xor(best_ts.key, off); pt.i = __shfl_xor(best_ts.i, off); pt.raw = __shfl_xor(best_ts.raw, off);
    ma.m = __shfl_xor(ms_all.m, off); ma.s = __shfl_xor(ms_all.s, off);
    mt.m = __shfl_xor(ms_ts.m, off); mt.s = __shfl_xor(ms_ts.s, off);
    const bool low = (lane & off) == 0;
    best_all = pick_better(best_all, pa); best_ts = pick_better(best_ts, pt);
    max_text = fmaxf(max_text, __shfl_xor(max_text, off));
    ms_all = low ? ms_merge(ms_all, ma) : ms_merge(ma, ms_all);
    ms_ts = low ? ms_merge(ms_ts, mt) : ms_merge(mt, ms_ts);
}
constexpr int SAMP_REC = 16;          // floats per (row, part) record of the two-launch form

// s.parts == 1: one workgroup per row scans it and commits the id.  s.parts == P > 1 (sampling steps only): the scan is VALU
// bound (~45 instructions per vocabulary entry), so P workgroups share a row, each leaves a record in s.part_buf, and
// sampler_finish_kernel merges the P records and commits -- one more (tiny) launch, the scan on 4x the CUs.
__global__ __launch_bounds__(SAMP_NT) void sampler_kernel(SamplerArgs s) {
    __shared__ Pick sh_all[SAMP_NW], sh_ts[SAMP_NW];
    __shared__ MS sh_msa[SAMP_NW], sh_mst[SAMP_NW];
    __shared__ float sh_f[SAMP_NW], sh_g[SAMP_NW];
    __shared__ int sh_flags[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int P = s.parts, rb = blockIdx.x / P, part = blockIdx.x % P;
    // The position moves on once every workgroup has READ it, which is all the ticket guards -- so it is drawn here, where
    // its round trip hides behind the row scan, not at the end of the kernel.  One lane reads the position for the whole
    // workgroup, and the ticket's operand depends on the value read, so the atomic cannot be issued ahead of that load.
    // (With P > 1 the finishing kernel still needs the position: it draws the tickets.)
    if (tid == 0) {
        const int p = *s.d_pos;
        sh_flags[7] = p;
        if (P == 1) {
            const int ticket = atomicAdd(s.d_block_counter, p >= 0 ? 1 : 2);      // always 1
            if (ticket == s.B - 1) { *s.d_block_counter = 0; *s.d_pos = p + s.group; }
        }
    }
    __syncthreads();
    const int t0 = sh_flags[7];
    const int b = rb / s.group, t = t0 + rb % s.group;     // group > 1: forced prompt pass, one row per position
    const float* lg = s.logits + (int64_t)rb * s.ldl;
    const int V = s.n_vocab;
    const int n_prompt = s.row_n_prompt ? s.row_n_prompt[b] : s.n_prompt;
    const int sot_index = s.row_sot ? s.row_sot[b] : s.sot_index;

    if (t == sot_index && s.no_speech && part == 0) {
        float mx = -INFINITY;
        for (int i = tid; i < V; i += SAMP_NT) mx = fmaxf(mx, lg[i]);
        mx = wave_max(mx);
        if (lane == 0) sh_f[wid] = mx;
        __syncthreads();
        mx = sh_f[0];
        for (int w = 1; w < SAMP_NW; ++w) mx = fmaxf(mx, sh_f[w]);
        float se = 0.f;
        for (int i = tid; i < V; i += SAMP_NT) se += expf(lg[i] - mx);
        se = wave_sum(se);
        if (lane == 0) sh_g[wid] = se;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < SAMP_NW; ++w) tot += sh_g[w];
            s.no_speech[b] = expf(lg[s.tok_no_speech] - mx) / tot;
        }
        __syncthreads();
    }

    if (t >= n_prompt - 1 && !s.forced && t + 1 < s.ld_tok) {
        int32_t* row = s.tokens + (int64_t)b * s.ld_tok;
        const int tb = s.tok_timestamp_begin;
        const bool first_step = (t == n_prompt - 1);
        const int cnt = t - n_prompt + 1;                            // sampled tokens so far
        const float temperature = s.dyn ? __uint_as_float(s.dyn[0]) : s.temperature;
        const uint32_t seed = s.dyn ? s.dyn[1] : s.seed;
        const int sample_len = s.dyn ? (int)s.dyn[2] : s.sample_len;
        int row_cap = s.ld_tok - n_prompt; if (row_cap > sample_len) row_cap = sample_len;
        if (cnt >= row_cap) {
            // upstream stops this window at sample_len / n_ctx: the row is closed with EOT (not counted, not scored)
            if (tid == 0 && P == 1) {
                const bool alive = row[t] != s.tok_eot;
                row[t + 1] = s.tok_eot;
                if (alive && cnt == row_cap) atomicAdd(s.n_finished, 1);
            }
        } else {
        if (tid == 0) {
            // timestamp-rule ranges (all empty in without_timestamps mode)
            int a_lo = 0, a_hi = 0, b_lo = 0, b_hi = 0, c_hi = 0, d_lo = V;
            if (!s.without_timestamps) {
                const bool last_ts = cnt >= 1 && row[t] >= tb;
                const bool penult_ts = cnt < 2 || row[t - 1] >= tb;
                if (last_ts) {
                    if (penult_ts) { a_lo = tb; a_hi = V; } else { a_lo = 0; a_hi = s.tok_eot; }
                }
                int ts_last = -1;
                for (int j = t; j >= n_prompt; --j) if (row[j] >= tb) { ts_last = row[j]; break; }
                if (ts_last >= 0) { b_lo = tb; b_hi = (last_ts && !penult_ts) ? ts_last : ts_last + 1; }
                if (first_step) {
                    c_hi = tb;                                      // [0, tb) masked
                    if (s.max_initial_ts >= 0) d_lo = tb + s.max_initial_ts + 1;   // [d_lo, V) masked
                }
            }
            sh_flags[0] = a_lo; sh_flags[1] = a_hi; sh_flags[2] = b_lo; sh_flags[3] = b_hi; sh_flags[4] = c_hi; sh_flags[5] = d_lo;
        }
        __syncthreads();
        const int a_lo = sh_flags[0], a_hi = sh_flags[1], b_lo = sh_flags[2], b_hi = sh_flags[3], c_hi = sh_flags[4], d_lo = sh_flags[5];
        const bool ts_mode = !s.without_timestamps;
        const bool blank_rule = first_step && s.suppress_blank;
        // what the last lane standing needs at the very end is fetched now, behind the scan
        const int prev_id = row[t];
        const float prev_sum = s.sum_logprob[b];
        const bool sampling = temperature > 0.f;
        const float inv_t = sampling ? 1.0f / temperature : 1.0f;
        Pick best_all{-INFINITY, 0x7fffffff, -INFINITY}, best_ts{-INFINITY, 0x7fffffff, -INFINITY};
        MS ms_all{-INFINITY, 0.f}, ms_ts{-INFINITY, 0.f};
        float max_text = -INFINITY;
        // suppress[] is padded to a multiple of 4 bytes (wx_finalize), logits rows to 16 floats
        for (int i0 = (part * SAMP_NT + tid) * 4; i0 < V; i0 += P * SAMP_NT * 4) {
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
                float key = v;
                if (sampling) key = v * inv_t - logf(-logf(uniform01(seed, b, t, i)));   // Gumbel-max
                best_all = pick_better(best_all, Pick{key, i, v});
                ms_add(ms_all, v);
                if (i >= tb) { best_ts = pick_better(best_ts, Pick{key, i, v}); if (ts_mode) ms_add(ms_ts, v); }
                else max_text = fmaxf(max_text, v);
            }
        }
        best_all = wave_pick(best_all); best_ts = wave_pick(best_ts); max_text = wave_max(max_text);
        ms_all = wave_ms(ms_all); ms_ts = wave_ms(ms_ts);
        if (lane == 0) { sh_all[wid] = best_all; sh_ts[wid] = best_ts; sh_f[wid] = max_text; sh_msa[wid] = ms_all; sh_mst[wid] = ms_ts; }
        __syncthreads();
        if (wid == 0) {
            // the 16 per-wave results, one per lane, merged by a fixed 4-step butterfly (lanes >= 16 carry neutral elements)
            const bool has = lane < SAMP_NW;
            best_all = has ? sh_all[lane] : Pick{-INFINITY, 0x7fffffff, -INFINITY};
            best_ts = has ? sh_ts[lane] : Pick{-INFINITY, 0x7fffffff, -INFINITY};
            max_text = has ? sh_f[lane] : -INFINITY;
            ms_all = has ? sh_msa[lane] : MS{-INFINITY, 0.f};
            ms_ts = has ? sh_mst[lane] : MS{-INFINITY, 0.f};
#pragma unroll
            for (int off = SAMP_NW / 2; off >= 1; off >>= 1) sampler_butterfly(lane, off, best_all, best_ts, max_text, ms_all, ms_ts);
        }
        if (tid == 0) {
            if (P == 1) {
                sampler_commit(s, b, t, ts_mode, best_all, best_ts, max_text, ms_all, ms_ts, prev_id, prev_sum);
            } else {
                float* rec = s.part_buf + ((int64_t)rb * P + part) * SAMP_REC;
                rec[0] = best_all.key; rec[1] = __int_as_float(best_all.i); rec[2] = best_all.raw;
                rec[3] = best_ts.key; rec[4] = __int_as_float(best_ts.i); rec[5] = best_ts.raw;
                rec[6] = max_text; rec[7] = ms_all.m; rec[8] = ms_all.s; rec[9] = ms_ts.m; rec[10] = ms_ts.s;
            }
        }
        }
    }
}

// second launch of the P > 1 form: one wave per row merges the P records (same butterfly) and commits; draws the tickets
__global__ __launch_bounds__(64) void sampler_finish_kernel(SamplerArgs s) {
    const int lane = threadIdx.x, b = blockIdx.x, P = s.parts;
    const int t = *s.d_pos;                                  // wave-uniform; the ticket below depends on it
    if (lane == 0) {
        const int ticket = atomicAdd(s.d_block_counter, t >= 0 ? 1 : 2);          // always 1
        if (ticket == s.B - 1) { *s.d_block_counter = 0; *s.d_pos = t + 1; }
    }
    const int n_prompt = s.row_n_prompt ? s.row_n_prompt[b] : s.n_prompt;
    if (!(t >= n_prompt - 1 && t + 1 < s.ld_tok)) return;
    int32_t* row = s.tokens + (int64_t)b * s.ld_tok;
    const int cnt = t - n_prompt + 1;
    const int sample_len = s.dyn ? (int)s.dyn[2] : s.sample_len;
    int row_cap = s.ld_tok - n_prompt; if (row_cap > sample_len) row_cap = sample_len;
    const int prev_id = row[t];
    if (cnt >= row_cap) {
        if (lane == 0) {
            row[t + 1] = s.tok_eot;
            if (prev_id != s.tok_eot && cnt == row_cap) atomicAdd(s.n_finished, 1);
        }
        return;
    }
    const float prev_sum = s.sum_logprob[b];
    const bool has = lane < P;
    const float* rec = s.part_buf + ((int64_t)b * P + (has ? lane : 0)) * SAMP_REC;
    Pick best_all = has ? Pick{rec[0], __float_as_int(rec[1]), rec[2]} : Pick{-INFINITY, 0x7fffffff, -INFINITY};
    Pick best_ts = has ? Pick{rec[3], __float_as_int(rec[4]), rec[5]} : Pick{-INFINITY, 0x7fffffff, -INFINITY};
    float max_text = has ? rec[6] : -INFINITY;
    MS ms_all = has ? MS{rec[7], rec[8]} : MS{-INFINITY, 0.f};
    MS ms_ts = has ? MS{rec[9], rec[10]} : MS{-INFINITY, 0.f};
    for (int off = P / 2; off >= 1; off >>= 1) sampler_butterfly(lane, off, best_all, best_ts, max_text, ms_all, ms_ts);
    if (lane == 0) sampler_commit(s, b, t, !s.without_timestamps, best_all, best_ts, max_text, ms_all, ms_ts, prev_id, prev_sum);
}

int launch_sampler(const SamplerArgs& s, hipStream_t st) {
    if (s.B <= 0 || !s.logits || !s.tokens || !s.d_pos || !s.d_block_counter || !s.suppress) return -1;
    if (s.group < 1 || (s.group > 1 && (!s.forced || s.B % s.group))) return -1;
    if (s.parts != 1 && (s.forced || s.group != 1 || !s.part_buf || (s.parts != 2 && s.parts != 4 && s.parts != 8))) return -1;
    hipLaunchKernelGGL(sampler_kernel, dim3(s.B * s.parts), dim3(SAMP_NT), 0, st, s);
    if (s.parts > 1) hipLaunchKernelGGL(sampler_finish_kernel, dim3(s.B), dim3(64), 0, st, s);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// After the loop: count sampled ids before the first EOT and pad the tail of every row with EOT.
__global__ void finish_tokens_kernel(int32_t* tokens, int ld_tok, int n_prompt_all, const int32_t* row_n_prompt, int tok_eot,
                                     const int32_t* d_pos, int32_t* n_tokens) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int n_prompt = row_n_prompt ? row_n_prompt[b] : n_prompt_all;
    int32_t* row = tokens + (int64_t)b * ld_tok;
    const int last = *d_pos;             // index of the last written token
    int n = 0;
    bool ended = false;
    for (int j = n_prompt; j < ld_tok; ++j) {
        if (!ended && j <= last && row[j] != tok_eot) ++n; else { ended = true; row[j] = tok_eot; }
    }
    n_tokens[b] = n;
}
int launch_finish_tokens(int32_t* tokens, int ld_tok, int n_prompt, const int32_t* row_n_prompt, int tok_eot, const int32_t* d_pos,
                         int32_t* n_tokens, int B, hipStream_t st) {
    hipLaunchKernelGGL(finish_tokens_kernel, dim3(B), dim3(64), 0, st, tokens, ld_tok, n_prompt, row_n_prompt, tok_eot, d_pos, n_tokens);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
