// mel.hip — log-mel front end for gfx950 (replaces whisper.log_mel_spectrogram).
//
// Stage 1 (one fused kernel): PCM span -> LDS (coalesced, reflect / zero padding applied while loading) -> Hann
// window -> 201-bin real DFT (8 folded 50-point DFTs + one combine, LDS twiddle table) -> |X|^2 -> sparse triangular mel
// filterbank -> log10(max(.,1e-10)) -> HBM, plus the slice maximum folded into one device float.
// Stage 2: max(x, gmax - 8), (x + 4) / 4 in place.
// Algorithmic HBM bytes per 30 s window: 480000*4 read + n_mels*3000*4 written (+ the same again for stage 2).
#include "common.h"

namespace wx {

constexpr int MEL_FT = 8;                                // frames per workgroup
constexpr int MEL_SPAN = MEL_FT * HOP + (N_FFT - HOP);   // samples that feed the frames of one workgroup

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    // ordered-bits trick; *addr starts at -inf
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// 400-point real DFT as 8 interleaved 50-point DFTs plus one combine (decimation in time, one level):
//   X[k] = sum_{r<8} W400^(r k) * Y_r[k mod 50],   Y_r[q] = sum_{j<50} xw[8 j + r] * W50^(q j)
// Y_r is conjugate-symmetric (real input), so only q = 0..25 is computed, each as a FOLDED direct sum (j and 50 - j share
// a cosine and have opposite sines: 24 multiply-adds per part instead of 49).  Per frame that is 8*26*48 + 201*8*4 = 16.4 k
// multiply-adds against 80 k for the folded direct 400-point DFT this kernel started as: the stage is no longer bound by
// fp32 issue.  Stage A maps one thread to one (r, q) pair, stage B one thread to one bin k; both keep the frames of the
// workgroup in registers and read operands that are either shared by many lanes (broadcast) or laid out with padded
// record strides (2*MEL_FT+4 / MEL_FT+4 floats) so that each 16-lane group of a ds_read/write_b128 touches all 64 banks once.
constexpr int DFT_R = 8, DFT_NS = N_FFT / DFT_R, DFT_NQ = DFT_NS / 2 + 1;       // 8 sub-sequences of 50, bins 0..25
constexpr int Y_REC = 2 * MEL_FT + 4;                                           // re[MEL_FT], im[MEL_FT], pad
constexpr int P_REC = MEL_FT + 4;                                               // power[MEL_FT], pad
constexpr int LDS_Y = DFT_R * DFT_NQ * Y_REC;                                   // first holds the PCM span
constexpr int LDS_EO = 2 * DFT_R * DFT_NQ * MEL_FT;                             // later the power spectrum
static_assert(LDS_Y >= MEL_SPAN && LDS_EO >= N_FREQ * P_REC, "LDS aliasing");

constexpr int MEL_PACK_LDS = 512;        // packed filter weights kept in LDS (Slaney banks: 391 / 394 non-zero weights)

template <bool PACK_LDS>
__global__ __launch_bounds__(256, 4) void mel_stage1_kernel(
    const float* __restrict__ window, const float* __restrict__ cs_tab, const float* __restrict__ filters, int n_pack,
    const int32_t* __restrict__ frange, int n_mels, const float* __restrict__ pcm, int64_t pcm_offset,
    int64_t n_valid, int64_t n_file, int64_t n_total, int64_t frame0, int64_t n_frames, float* __restrict__ out,
    int64_t ld, float* d_max, float* block_max) {
    __shared__ __attribute__((aligned(16))) float s_cs[N_FFT * 2];          // cos, sin of 2*pi*i/400
    __shared__ __attribute__((aligned(16))) float s_a[LDS_Y];               // PCM span, then the Y_r[q] records
    __shared__ __attribute__((aligned(16))) float s_b[LDS_EO];              // folded inputs e | o, then power[k]
    __shared__ __attribute__((aligned(16))) float s_f[PACK_LDS ? MEL_PACK_LDS : 4];
    __shared__ __attribute__((aligned(16))) float s_win[N_FFT];
    __shared__ __attribute__((aligned(16))) float s_cs50[DFT_NS * 2];       // cos, sin of 2*pi*i/50: compact, so a wave's 64
                                                                            // twiddle reads spread over 32 banks, not 4
    __shared__ float s_red[4];
    float* s_span = s_a;
    float* s_y = s_a;
    float* s_e = s_b;                                   // e[r][j][f], j = 0..25 (j = 0: x_r[0], j = 25: x_r[25])
    float* s_o = s_b + DFT_R * DFT_NQ * MEL_FT;         // o[r][j][f], j = 1..24
    float* s_p = s_b;                                   // power[k] records of P_REC floats

    const int tid = threadIdx.x;
    const int64_t fb = frame0 + (int64_t)blockIdx.x * MEL_FT;     // first file frame of this block
    // Every global load of the workgroup is issued before the first one is waited for: a load under a branch, or in a
    // runtime loop that stores to LDS, costs one exposed memory round trip per iteration (6 of them were ~2/3 of this kernel).
    // Out-of-range indices are clamped to a valid address and the value replaced afterwards.
    constexpr int SPAN_IT = (MEL_SPAN + 255) / 256, PACK_IT = MEL_PACK_LDS / 256;
    const int64_t s0 = fb * HOP - N_FFT / 2;                      // PCM span: file samples fb*160 - 200 ... + MEL_SPAN
    float sv[SPAN_IT];
    bool sok[SPAN_IT];
#pragma unroll
    for (int it = 0; it < SPAN_IT; ++it) {
        int64_t j = s0 + tid + it * 256;
        if (j < 0) j = -j;                                   // reflect at the start of the file
        if (j >= n_total) j = 2 * (n_total - 1) - j;         // reflect at the end of the padded file
        int64_t r = j - pcm_offset;
        sok[it] = j >= 0 && j < n_file && r >= 0 && r < n_valid;
        r = r < 0 ? 0 : (r > n_valid - 1 ? n_valid - 1 : r);
        sv[it] = n_valid > 0 ? pcm[r] : 0.f;                 // uniform condition (kernel argument), not a divergent branch
    }
    f32x4 tv = f32x4{0.f, 0.f, 0.f, 0.f}, wv = tv;
    if (tid < N_FFT * 2 / 4) tv = reinterpret_cast<const f32x4*>(cs_tab)[tid];
    if (tid < N_FFT / 4) wv = reinterpret_cast<const f32x4*>(window)[tid];
    f32x2 t50 = f32x2{0.f, 0.f};
    if (tid < DFT_NS) t50 = reinterpret_cast<const f32x2*>(cs_tab)[tid * DFT_R];
    float pk[PACK_LDS ? PACK_IT : 1];
    if (PACK_LDS) {
#pragma unroll
        for (int it = 0; it < PACK_IT; ++it) { const int i = tid + it * 256; pk[it] = filters[i < n_pack ? i : n_pack - 1]; }
    }
#pragma unroll
    for (int it = 0; it < SPAN_IT; ++it) { const int i = tid + it * 256; if (i < MEL_SPAN) s_span[i] = sok[it] ? sv[it] : 0.f; }
    if (tid < N_FFT * 2 / 4) reinterpret_cast<f32x4*>(s_cs)[tid] = tv;
    if (tid < N_FFT / 4) reinterpret_cast<f32x4*>(s_win)[tid] = wv;
    if (tid < DFT_NS) reinterpret_cast<f32x2*>(s_cs50)[tid] = t50;
    if (PACK_LDS) {
#pragma unroll
        for (int it = 0; it < PACK_IT; ++it) s_f[tid + it * 256] = pk[it];
    }
    __syncthreads();
    // Hann window + fold of every 50-point sub-sequence: e[j] = x_r[j] + x_r[50-j], o[j] = x_r[j] - x_r[50-j]
    // (j = 0 and j = 25 have no partner: e = x, o = 0; computed branch-free with the partner's weight set to 0)
    for (int i = tid; i < DFT_R * DFT_NQ * MEL_FT; i += 256) {
        const int f = i % MEL_FT, rj = i / MEL_FT, j = rj % DFT_NQ, r = rj / DFT_NQ;
        const int n = DFT_R * j + r;
        const bool single = (j == 0 || j == DFT_NS / 2);
        const int n2 = single ? n : N_FFT - DFT_R * j + r;       // 8 * (50 - j) + r
        const float a = s_span[f * HOP + n] * s_win[n];
        const float b = single ? 0.f : s_span[f * HOP + n2] * s_win[n2];
        s_e[i] = a + b;
        s_o[i] = single ? 0.f : a - b;
    }
    __syncthreads();                                         // the span is dead from here on: s_y overwrites it
    if (tid < DFT_R * DFT_NQ) {                              // stage A: Y_r[q], the frames in registers
        const int r = tid / DFT_NQ, q = tid % DFT_NQ;
        const float* er = s_e + r * DFT_NQ * MEL_FT;
        const float* orr = s_o + r * DFT_NQ * MEL_FT;
        float re[MEL_FT], im[MEL_FT];
        const float sgn = (q & 1) ? -1.f : 1.f;              // W50^(25 q)
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) {
            re[f] = er[f] + sgn * er[(DFT_NS / 2) * MEL_FT + f];
            im[f] = 0.f;
        }
        int idx = 0;                                         // (q * j) mod 50
#pragma unroll 2
        for (int j = 1; j < DFT_NS / 2; ++j) {
            idx += q;
            if (idx >= DFT_NS) idx -= DFT_NS;
            const f32x2 c = *reinterpret_cast<const f32x2*>(&s_cs50[idx * 2]);           // cos, sin of 2*pi*idx/50
            const float ns = -c[1];                          // e^{-i t}: the imaginary part is -sum x sin t
            const f32x4* ep = reinterpret_cast<const f32x4*>(er + j * MEL_FT);
            const f32x4* op = reinterpret_cast<const f32x4*>(orr + j * MEL_FT);
#pragma unroll
            for (int v = 0; v < MEL_FT / 4; ++v) {
                const f32x4 ev = ep[v], ov = op[v];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    re[v * 4 + u] = fmaf(ev[u], c[0], re[v * 4 + u]);
                    im[v * 4 + u] = fmaf(ov[u], ns, im[v * 4 + u]);
                }
            }
        }
        f32x4* yrec = reinterpret_cast<f32x4*>(s_y + tid * Y_REC);
#pragma unroll
        for (int v = 0; v < MEL_FT / 4; ++v) {
            yrec[v] = f32x4{re[v * 4], re[v * 4 + 1], re[v * 4 + 2], re[v * 4 + 3]};
            yrec[MEL_FT / 4 + v] = f32x4{im[v * 4], im[v * 4 + 1], im[v * 4 + 2], im[v * 4 + 3]};
        }
    }
    __syncthreads();                                         // e | o are dead from here on: the power records overwrite them
    if (tid < N_FREQ) {                                      // stage B: X[k] = sum_r W400^(r k) Y_r[k mod 50], then |X|^2
        const int k = tid;
        const int q0 = k % DFT_NS;
        const bool conj = q0 > DFT_NS / 2;                   // Y_r[q0] = conj(Y_r[50 - q0])
        const int q = conj ? DFT_NS - q0 : q0;
        const float isgn = conj ? -1.f : 1.f;
        float xr[MEL_FT], xi[MEL_FT];
#pragma unroll
        for (int f = 0; f < MEL_FT; ++f) { xr[f] = 0.f; xi[f] = 0.f; }
        int idx = 0;                                         // (r * k) mod 400
#pragma unroll 1
        for (int r = 0; r < DFT_R; ++r) {
            const f32x2 w = *reinterpret_cast<const f32x2*>(&s_cs[idx * 2]);             // W = cos - i sin
            const float c = w[0], sn = w[1];
            const f32x4* yrec = reinterpret_cast<const f32x4*>(s_y + (r * DFT_NQ + q) * Y_REC);
#pragma unroll
            for (int v = 0; v < MEL_FT / 4; ++v) {
                const f32x4 yr = yrec[v];
                f32x4 yi = yrec[MEL_FT / 4 + v];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float yiu = yi[u] * isgn;
                    // (yr + i yi)(c - i s) = (yr c + yi s) + i (yi c - yr s)
                    xr[v * 4 + u] = fmaf(yr[u], c, fmaf(yiu, sn, xr[v * 4 + u]));
                    xi[v * 4 + u] = fmaf(yiu, c, fmaf(-yr[u], sn, xi[v * 4 + u]));
                }
            }
            idx += k;
            if (idx >= N_FFT) idx -= N_FFT;
        }
        f32x4* prec = reinterpret_cast<f32x4*>(s_p + k * P_REC);
#pragma unroll
        for (int v = 0; v < MEL_FT / 4; ++v) {
            f32x4 pw;
#pragma unroll
            for (int u = 0; u < 4; ++u) pw[u] = xr[v * 4 + u] * xr[v * 4 + u] + xi[v * 4 + u] * xi[v * 4 + u];
            prec[v] = pw;
        }
    }
    __syncthreads();
    float lmax = -INFINITY;
    {
        constexpr int FPT = MEL_FT / 2;                    // frames per thread: two threads share a mel bin
        const int m = tid % 128, fg = tid / 128;
        for (int mm = m; mm < n_mels; mm += 128) {
            float acc[FPT];
#pragma unroll
            for (int j = 0; j < FPT; ++j) acc[j] = 0.f;
            const i32x4 fr = *reinterpret_cast<const i32x4*>(frange + mm * 4);
            const int k0 = fr[0], k1 = fr[1];
            for (int k = k0; k < k1; ++k) {
                const float w = PACK_LDS ? s_f[fr[2] + k - k0] : filters[fr[2] + k - k0];
#pragma unroll
                for (int v = 0; v < FPT / 4; ++v) {
                    const f32x4 pv = *reinterpret_cast<const f32x4*>(&s_p[k * P_REC + fg * FPT + v * 4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[v * 4 + j] = fmaf(w, pv[j], acc[v * 4 + j]);
                }
            }
#pragma unroll
            for (int j = 0; j < FPT; ++j) {
                const int64_t f = (int64_t)blockIdx.x * MEL_FT + fg * FPT + j;   // frame index relative to frame0
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
        // block_max (wx_log_mel_static): one plain store per workgroup into the caller's partial-maximum array, folded by
        // mel_fold_block_max_kernel afterwards; otherwise one atomic into the running maximum
        if (block_max) block_max[blockIdx.x] = v;
        else if (v > -INFINITY) atomic_max_float(d_max, v);
    }
}

// out[0] = max(part[0 .. n)): the per-workgroup maxima of one mel_stage1 launch (one workgroup; n <= a few 10^4)
__global__ __launch_bounds__(1024) void mel_fold_block_max_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float s_red[16];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, part[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, s_red[w]);
        out[0] = m;
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
                      hipStream_t st, float* block_max) {
    if (n_frames <= 0) return 0;
    if (t.n_mels > 256 || ld < n_frames) return -1;
    const int64_t blocks = (n_frames + MEL_FT - 1) / MEL_FT;
    if (t.n_pack <= MEL_PACK_LDS)
        hipLaunchKernelGGL(mel_stage1_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, t.window, t.cs, t.filters, t.n_pack,
                           t.frange, t.n_mels, pcm, pcm_offset, n_valid, n_file, n_total, frame0, n_frames, out, ld, d_max, block_max);
    else
        hipLaunchKernelGGL(mel_stage1_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, t.window, t.cs, t.filters, t.n_pack,
                           t.frange, t.n_mels, pcm, pcm_offset, n_valid, n_file, n_total, frame0, n_frames, out, ld, d_max, block_max);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int mel_stage1_blocks(int64_t n_frames) { return (int)((n_frames + MEL_FT - 1) / MEL_FT); }

int launch_mel_fold_block_max(const float* part, int n, float* out, hipStream_t st) {
    if (n <= 0 || !part || !out) return -1;
    hipLaunchKernelGGL(mel_fold_block_max_kernel, dim3(1), dim3(1024), 0, st, part, n, out);
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
