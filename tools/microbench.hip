// microbench.hip — stand-alone timing of the decode-step kernels (graph replay, weights rotated over 32 layers so
// nothing is cache resident).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip whisper_amd/csrc/build/{gemm,elementwise,attention,flash,mel}.o -o tools/microbench
#include "../whisper_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace wx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void fill_bf16(bf16* p, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = (bf16)(((int)(h & 0xffff) - 32768) / 32768.0f * scale);
    }
}
__global__ void fill_f32(float* p, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = ((int)(h & 0xffff) - 32768) / 32768.0f * scale;
    }
}

template <typename F>
float time_graph(hipStream_t st, int launches_per_graph, int reps, F body) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    body();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * launches_per_graph);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64;
    const int L = 32, d = 1280;
    hipStream_t st; CK(hipStreamCreate(&st));
    int* dummy; CK(hipMalloc(&dummy, 64));
    printf("M=%d\n", M);
    printf("empty kernel            : %.2f us/launch\n", time_graph(st, 64, 20, [&] { for (int i = 0; i < 64; ++i) hipLaunchKernelGGL(empty_kernel, dim3(80), dim3(256), 0, st, dummy); }));
    struct Shape { int N, K; const char* name; } shapes[] = {{1280, 1280, "out/q proj"}, {3840, 1280, "qkv"}, {5120, 1280, "fc1"}, {1280, 5120, "fc2"}, {51872, 1280, "logits"}};
    for (auto& s : shapes) {
        const int copies = s.N > 10000 ? 4 : L;
        bf16* W; CK(hipMalloc(&W, (size_t)copies * s.N * s.K * 2));
        bf16* X; CK(hipMalloc(&X, (size_t)64 * s.K * 2));
        float* bias; CK(hipMalloc(&bias, (size_t)s.N * 4));
        float* res; CK(hipMalloc(&res, (size_t)64 * s.N * 4));
        bf16* outb; CK(hipMalloc(&outb, (size_t)64 * s.N * 4));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, st, W, (size_t)copies * s.N * s.K, 0.05f);
        hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, st, X, (size_t)64 * s.K, 1.0f);
        hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, bias, (size_t)s.N, 0.1f);
        hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, res, (size_t)64 * s.N, 1.0f);
        CK(hipStreamSynchronize(st));
        float* part; CK(hipMalloc(&part, (size_t)16 * 64 * s.N * 4));
        const int auto_split = decode_gemm_split(s.N, s.K);
        for (int S : {1, 2, 4, 5, 8, 10, 16}) {
            if (s.K % S) continue;
            auto body = [&] {
                for (int l = 0; l < copies; ++l) {
                    GemmArgs g; g.A = X; g.lda = s.K; g.W = W + (size_t)l * s.N * s.K; g.ldw = s.K; g.M = M; g.N = s.N; g.K = s.K; g.in_dtype = DT_BF16;
                    g.ep.bias = bias; g.ep.out = outb; g.ep.out_dtype = DT_BF16; g.ep.ldo = s.N;
                    g.n_split = S; g.partial = part;
                    if (launch_gemm(g, st)) { printf("launch_gemm failed\n"); exit(1); }
                }
            };
            {   // probe: unsupported slice lengths are rejected by launch_gemm
                GemmArgs g; g.A = X; g.lda = s.K; g.W = W; g.ldw = s.K; g.M = M; g.N = s.N; g.K = s.K; g.in_dtype = DT_BF16;
                g.ep.out = outb; g.ep.out_dtype = DT_BF16; g.ep.ldo = s.N; g.n_split = S; g.partial = part;
                if (launch_gemm(g, st)) continue;
            }
            const float us = time_graph(st, copies, 20, body);
            const double bytes = (double)s.N * s.K * 2;
            printf("gemm %-12s N=%5d K=%4d S=%2d%s: %7.2f us/launch  %7.1f GB/s weights\n", s.name, s.N, s.K, S, S == auto_split ? "*" : " ", us, bytes / us * 1e-3);
        }
        CK(hipFree(part));
        CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(bias)); CK(hipFree(res)); CK(hipFree(outb));
    }
    {
        float* x; CK(hipMalloc(&x, 64 * d * 4)); float* gmm; CK(hipMalloc(&gmm, d * 4)); float* bt; CK(hipMalloc(&bt, d * 4)); bf16* o; CK(hipMalloc(&o, 64 * d * 2));
        hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, x, (size_t)64 * d, 1.0f);
        hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, st, gmm, (size_t)d, 1.0f);
        hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, st, bt, (size_t)d, 1.0f);
        CK(hipStreamSynchronize(st));
        printf("layernorm 64x1280       : %.2f us/launch\n", time_graph(st, 64, 20, [&] { for (int i = 0; i < 64; ++i) launch_layernorm(x, gmm, bt, o, DT_BF16, M, d, st); }));
        float* part; CK(hipMalloc(&part, (size_t)16 * 64 * d * 4)); CK(hipMemset(part, 0, (size_t)16 * 64 * d * 4));
        for (int S : {0, 1, 4, 5, 8, 10})
            printf("reduce+layernorm S=%2d   : %.2f us/launch\n", S, time_graph(st, 64, 20, [&] { for (int i = 0; i < 64; ++i) launch_reduce_layernorm(x, part, S, bt, gmm, bt, o, DT_BF16, M, d, st); }));
    }
    {   // decoder self-attention (paged), B = M sequences x 20 heads, at several positions
        const int H = 20, P = 28, B = M;
        int32_t* pt; CK(hipMalloc(&pt, B * P * 4));
        { std::vector<int32_t> h(B * P); for (int i = 0; i < B * P; ++i) h[i] = i; CK(hipMemcpy(pt, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
        const size_t layer = (size_t)B * P * H * 16 * 64;
        bf16 *kp, *vp; CK(hipMalloc(&kp, layer * L * 2)); CK(hipMalloc(&vp, layer * L * 2));
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, st, kp, layer * L, 1.0f);
        hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, st, vp, layer * L, 1.0f);
        float* part; CK(hipMalloc(&part, (size_t)16 * 64 * 3 * d * 4)); CK(hipMemset(part, 0, (size_t)16 * 64 * 3 * d * 4));
        float* bias; CK(hipMalloc(&bias, 3 * d * 4)); CK(hipMemset(bias, 0, 3 * d * 4));
        bf16* qkv; CK(hipMalloc(&qkv, (size_t)64 * 3 * d * 2)); hipLaunchKernelGGL(fill_bf16, dim3(64), dim3(256), 0, st, qkv, (size_t)64 * 3 * d, 1.0f);
        bf16* o; CK(hipMalloc(&o, 64 * d * 2));
        int32_t* dpos; CK(hipMalloc(&dpos, 64));
        for (int t : {8, 64, 224, 440}) for (int S : {0, 4}) {
            CK(hipMemcpy(dpos, &t, 4, hipMemcpyHostToDevice));
            const float us = time_graph(st, L, 20, [&] {
                for (int l = 0; l < L; ++l) {
                    AttnArgs a; a.dt = DT_BF16; a.n_q = 1; a.q_sb = 3 * d;
                    if (S) { a.q_partial = part; a.q_bias = bias; a.q_nsplit = S; a.q_scale = 0.35f; }
                    else { a.Q = qkv; a.Knew = qkv + d; a.Vnew = qkv + 2 * d; a.new_sb = 3 * d; }
                    a.page_table = pt; a.pt_ld = P; a.Kpool = kp + layer * l; a.Vpool = vp + layer * l; a.d_pos = dpos;
                    a.O = o; a.o_sb = d; a.B = B; a.H = H; a.n_split = 1;
                    if (launch_attn_decode(a, st)) { printf("attn launch failed\n"); exit(1); }
                }
            });
            printf("self-attn decode t=%3d partial S=%d: %6.2f us/launch (%.0f GB/s kv)\n", t, S, us, (double)B * H * (t + 1) * 256 / us * 1e-3);
        }
    }
    return 0;
}
