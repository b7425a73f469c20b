// mel.hip — log-mel front end for gfx950 (replaces whisper.log_mel_spectrogram).
//
// Stage 1 (one fused kernel): PCM span -> LDS (coalesced, reflect / zero padding applied while loading) -> Hann
// window + even/odd folding -> 201-bin real DFT from an LDS twiddle table -> |X|^2 -> sparse triangular mel
// filterbank -> log10(max(.,1e-10)) -> HBM, plus the slice maximum folded into one device float.
// Stage 2: max(x, gmax - 8), (x + 4) / 4 in place.
// Algorithmic HBM bytes per 30 s window: 480000*4 read + n_mels*3000*4 written (+ the same again for stage 2).
#include "common.h"

namespace wx {

constexpr int MEL_FT = 16;                               // frames per workgroup
constexpr int MEL_SPAN = MEL_FT * HOP + (N_FFT - HOP);   // 2800 samples feed 16 frames

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    // ordered-bits trick; *addr starts at -inf
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void mel_stage1_kernel(
    const float* __restrict__ window, const float* __restrict__ cs_tab, const float* __restrict__ filters,
    const int32_t* __restrict__ frange, int n_mels, const float* __restrict__ pcm, int64_t pcm_offset,
    int64_t n_valid, int64_t n_file, int64_t n_total, int64_t frame0, int64_t n_frames, float* __restrict__ out,
    int64_t ld, float* d_max) {
    __shared__ __attribute__((aligned(16))) float s_cs[N_FFT * 2];
    __shared__ __attribute__((aligned(16))) float s_span[MEL_SPAN];
    __shared__ __attribute__((aligned(16))) float s_e[(N_FFT / 2 + 1) * MEL_FT];   // e[n][f], n = 0..200
    __shared__ __attribute__((aligned(16))) float s_o[(N_FFT / 2) * MEL_FT];       // o[n][f], n = 1..199
    __shared__ __attribute__((aligned(16))) float s_p[N_FREQ * MEL_FT];            // power[k][f]
    __shared__ float s_red[4];

    const int tid = threadIdx.x;
    const int64_t fb = frame0 + (int64_t)blockIdx.x * MEL_FT;     // first file frame of this block
    for (int i = tid; i < N_FFT * 2; i += 256) s_cs[i] = cs_tab[i];
    // PCM span: file samples fb*160 - 200 ... + 2800
    const int64_t s0 = fb * HOP - N_FFT / 2;
    for (int i = tid; i < MEL_SPAN; i += 256) {
        int64_t j = s0 + i;
        if (j < 0) j = -j;                                   // reflect at the start of the file
        if (j >= n_total) j = 2 * (n_total - 1) - j;         // reflect at the end of the padded file
        float v = 0.f;
        if (j >= 0 && j < n_file) {
            int64_t r = j - pcm_offset;
            if (r >= 0 && r < n_valid) v = pcm[r];
        }
        s_span[i] = v;
    }
    __syncthreads();
    // window + fold: e[n] = xw[n] + xw[400-n], o[n] = xw[n] - xw[400-n]
    for (int i = tid; i < (N_FFT / 2 + 1) * MEL_FT; i += 256) {
        const int n = i / MEL_FT, f = i % MEL_FT;
        const float a = s_span[f * HOP + n] * window[n];
        if (n == 0 || n == N_FFT / 2) {
            s_e[i] = a;
        } else {
            const float b = s_span[f * HOP + N_FFT - n] * window[N_FFT - n];
            s_e[i] = a + b;
            s_o[i] = a - b;
        }
    }
    __syncthreads();
    if (tid < N_FREQ) {
        const int k = tid;
        float re[MEL_FT], im[MEL_FT];
        const float sgn = (k & 1) ? -1.f : 1.f;
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) {
            re[f] = s_e[f] + sgn * s_e[(N_FFT / 2) * MEL_FT + f];
            im[f] = 0.f;
        }
        int idx = 0;
        for (int n = 1; n < N_FFT / 2; ++n) {
            idx += k;
            if (idx >= N_FFT) idx -= N_FFT;
            const f32x2 c = *reinterpret_cast<const f32x2*>(&s_cs[idx * 2]);
            const f32x4* ep = reinterpret_cast<const f32x4*>(&s_e[n * MEL_FT]);
            const f32x4* op = reinterpret_cast<const f32x4*>(&s_o[n * MEL_FT]);
#pragma unroll
            for (int q = 0; q < MEL_FT / 4; ++q) {
                const f32x4 ev = ep[q], ov = op[q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    re[q * 4 + j] = fmaf(ev[j], c[0], re[q * 4 + j]);
                    im[q * 4 + j] = fmaf(ov[j], c[1], im[q * 4 + j]);
                }
            }
        }
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) s_p[k * MEL_FT + f] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();
    float lmax = -INFINITY;
    {
        const int m = tid % 128, fg = tid / 128;          // 8 frames per thread
        for (int mm = m; mm < n_mels; mm += 128) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            const int k0 = frange[mm * 2], k1 = frange[mm * 2 + 1];
            for (int k = k0; k < k1; ++k) {
                const float w = filters[mm * N_FREQ + k];
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&s_p[k * MEL_FT + fg * 8]);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(&s_p[k * MEL_FT + fg * 8 + 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(w, p0[j], acc[j]);
                    acc[4 + j] = fmaf(w, p1[j], acc[4 + j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t f = (int64_t)blockIdx.x * MEL_FT + fg * 8 + j;   // frame index relative to frame0
                if (f < n_frames) {
                    const float v = log10f(fmaxf(acc[j], 1e-10f));
                    out[(int64_t)mm * ld + f] = v;
                    lmax = fmaxf(lmax, v);
                }
            }
        }
    }
    // block max -> one atomic
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) s_red[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) {
        const float v = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        if (v > -INFINITY) atomic_max_float(d_max, v);
    }
}

__global__ void mel_stage2_kernel(float* __restrict__ x, int64_t ld, int n_mels, int64_t n_frames,
                                  const float* __restrict__ d_max) {
    const float floor_v = *d_max - 8.0f;
    const int64_t total = (int64_t)n_mels * n_frames;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / n_frames, f = i % n_frames;
        float v = x[m * ld + f];
        v = fmaxf(v, floor_v);
        x[m * ld + f] = (v + 4.0f) / 4.0f;
    }
}

// PCM16 -> float32 in [-1, 1): x / 32768, the scaling upstream load_audio applies after ffmpeg (exact in fp32).
// 8 samples (16 B in, 32 B out) per lane.
__global__ void pcm16_to_f32_kernel(const int16_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t n8 = n / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        typedef __attribute__((ext_vector_type(8))) short i16x8;
        const i16x8 v = *reinterpret_cast<const i16x8*>(in + i * 8);
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = (float)v[j] * (1.0f / 32768.0f); b[j] = (float)v[4 + j] * (1.0f / 32768.0f); }
        *reinterpret_cast<f32x4*>(out + i * 8) = a;
        *reinterpret_cast<f32x4*>(out + i * 8 + 4) = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n8 * 8) out[n8 * 8 + threadIdx.x] = (float)in[n8 * 8 + threadIdx.x] * (1.0f / 32768.0f);
}

int launch_pcm16_to_f32(const int16_t* in, float* out, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return -1;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Channel average + polyphase windowed-sinc resampling of interleaved PCM16 (any rate) to mono float32 at 16 kHz: the
// reference's ffmpeg normalisation step done on the device, so a raw upload never takes a host pass.  One workgroup = 256
// consecutive outputs; the input frames they touch (256*down/up + 2*half_len/up of them) are averaged over channels once
// and staged in LDS as float (coalesced int16 reads), the taps come from a device table (<= 256 KB, L2 resident):
//   y[n] = sum_m x[m] * h[half_len + n*down - m*up],  x = 0 outside [0, n_frames)
// accumulated in ascending m with fmaf.  HBM bytes per output: 2*C*down/up read + 4 written.
constexpr int RS_BLOCK = 256;
constexpr int RS_SPAN_MAX = 6144;      // floats of LDS: 256 outputs * down/up <= 8 -> 2048 frames + 2 * half_len / up

__global__ __launch_bounds__(RS_BLOCK) void resample_pcm16_kernel(const int16_t* __restrict__ in, int64_t n_frames, int C, int up,
                                                                   int down, const float* __restrict__ taps, int half_len,
                                                                   int quantize, float* __restrict__ out, int64_t n_out) {
    __shared__ float s_x[RS_SPAN_MAX];
    const int64_t n0 = (int64_t)blockIdx.x * RS_BLOCK;
    const int64_t n1 = min(n0 + RS_BLOCK, n_out) - 1;                        // last output of this block
    // m range of one output n: ceil((n*down - half_len) / up) .. floor((n*down + half_len) / up)
    auto m_first = [&](int64_t n) { const int64_t a = n * down - half_len; return a >= 0 ? (a + up - 1) / up : -((-a) / up); };
    auto m_last = [&](int64_t n) { const int64_t a = n * down + half_len; return a / up; };
    const int64_t mb = m_first(n0), me = m_last(n1);
    const int span = (int)(me - mb + 1);
    const float inv = 1.0f / (32768.0f * (float)C);
    for (int i = threadIdx.x; i < span; i += RS_BLOCK) {
        const int64_t m = mb + i;
        float v = 0.f;
        if (m >= 0 && m < n_frames) {
            int acc = 0;
            for (int c = 0; c < C; ++c) acc += in[m * C + c];
            v = (float)acc * inv;
        }
        s_x[i] = v;
    }
    __syncthreads();
    const int64_t n = n0 + threadIdx.x;
    if (n >= n_out) return;
    const int64_t lo = m_first(n), hi = m_last(n);
    float y = 0.f;
    int idx = (int)(half_len + n * down - lo * up);                        // tap index of the first input frame
    for (int64_t m = lo; m <= hi; ++m, idx -= up) y = fmaf(s_x[m - mb], taps[idx], y);
    if (quantize) y = fminf(fmaxf(rintf(y * 32768.0f), -32768.0f), 32767.0f) * (1.0f / 32768.0f);
    out[n] = y;
}

int launch_resample_pcm16(const int16_t* in, int64_t n_frames, int n_channels, int up, int down, const float* d_taps,
                          int half_len, int quantize, float* out, int64_t n_out, hipStream_t st) {
    if (n_out <= 0) return 0;
    if (n_frames <= 0 || n_channels < 1 || n_channels > 8 || up < 1 || down < 1 || down > 8 * up || half_len < 0) return -1;
    if (n_out != (n_frames * up + down - 1) / down) return -1;
    // LDS span of one block: frames under 256 outputs plus the filter's reach on either side
    const int64_t span = ((int64_t)(RS_BLOCK - 1) * down + 2 * (int64_t)half_len) / up + 3;
    if (span > RS_SPAN_MAX) return -1;
    const int64_t blocks = (n_out + RS_BLOCK - 1) / RS_BLOCK;
    hipLaunchKernelGGL(resample_pcm16_kernel, dim3((unsigned)blocks), dim3(RS_BLOCK), 0, st, in, n_frames, n_channels, up, down,
                       d_taps, half_len, quantize, out, n_out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

__global__ void fill_f32_kernel(float* p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

int launch_mel_stage1(const MelTables& t, const float* pcm, int64_t pcm_offset, int64_t n_valid, int64_t n_file,
                      int64_t n_total, int64_t frame0, int64_t n_frames, float* out, int64_t ld, float* d_max,
                      hipStream_t st) {
    if (n_frames <= 0) return 0;
    if (t.n_mels > 256 || ld < n_frames) return -1;
    const int64_t blocks = (n_frames + MEL_FT - 1) / MEL_FT;
    hipLaunchKernelGGL(mel_stage1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t.window, t.cs, t.filters, t.frange,
                       t.n_mels, pcm, pcm_offset, n_valid, n_file, n_total, frame0, n_frames, out, ld, d_max);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_mel_stage2(float* x, int64_t ld, int n_mels, int64_t n_frames, const float* d_max, hipStream_t st) {
    if (n_frames <= 0) return 0;
    const int64_t total = (int64_t)n_mels * n_frames;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mel_stage2_kernel, dim3(blocks), dim3(256), 0, st, x, ld, n_mels, n_frames, d_max);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_fill_f32(float* p, float v, int64_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, st, p, v, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace wx
