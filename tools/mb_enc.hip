// mb_enc.hip — stand-alone timing of the encoder GEMM shapes of large-v3 at 64 windows (M = 96000), bf16.
#include "../whisper_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace wx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
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
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 96000;
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shape { int N, K; int kind; const char* name; } shapes[] = {   // kind 0: bf16 out + bias, 1: gelu, 2: fp32 in-place residual
        {3840, 1280, 0, "qkv"}, {1280, 1280, 2, "out"}, {5120, 1280, 1, "fc1"}, {1280, 5120, 2, "fc2"}};
    bf16* A; CK(hipMalloc(&A, (size_t)M * 5120 * 2));
    bf16* W; CK(hipMalloc(&W, (size_t)5120 * 1280 * 2));
    float* bias; CK(hipMalloc(&bias, 5120 * 4));
    char* out; CK(hipMalloc(&out, (size_t)M * 5120 * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, st, A, (size_t)M * 5120, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, st, W, (size_t)5120 * 1280, 0.03f);
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, st, bias, (size_t)5120, 0.1f);
    hipLaunchKernelGGL(fill_f32, dim3(4096), dim3(256), 0, st, (float*)out, (size_t)M * 1280, 1.0f);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& s : shapes) {
        GemmArgs g; g.A = A; g.lda = s.K; g.W = W; g.ldw = s.K; g.M = M; g.N = s.N; g.K = s.K; g.in_dtype = DT_BF16;
        g.ep.bias = bias;
        if (s.kind == 2) { g.ep.residual = (float*)out; g.ep.ldr = s.N; g.ep.out = out; g.ep.out_dtype = DT_F32; g.ep.ldo = s.N; }
        else { g.ep.act = s.kind; g.ep.out = out; g.ep.out_dtype = DT_BF16; g.ep.ldo = s.N; }
        if (launch_gemm(g, st)) { printf("launch failed\n"); return 1; }
        CK(hipStreamSynchronize(st));
        const int reps = 10;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_gemm(g, st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        std::vector<uint16_t> h(4096); CK(hipMemcpy(h.data(), out + (size_t)12345 * s.N * (s.kind == 2 ? 4 : 2), 8192, hipMemcpyDeviceToHost));
        unsigned long long cs = 0; for (auto v : h) cs = cs * 1315423911ull + v;
        printf("enc gemm %-4s M=%d N=%4d K=%4d: %8.1f us  %7.1f TFLOP/s  checksum %llx\n", s.name, M, s.N, s.K, us, 2.0 * M * s.N * s.K / us * 1e-6, cs);
    }
    return 0;
}
