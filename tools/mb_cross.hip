// mb_cross.hip — stand-alone timing of the decoder cross-attention launch (B windows x 20 heads x 1500 keys, bf16), K/V
// rotated over 32 layers so nothing is cache resident; the query arrives as split-K partial tiles like in the engine.
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
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int L = 32, d = 1280, H = 20, T = 1500, S = 10;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t layer = (size_t)B * H * T * 64;
    bf16 *K, *V; CK(hipMalloc(&K, layer * L * 2)); CK(hipMalloc(&V, layer * L * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, st, K, layer * L, 1.0f);
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, st, V, layer * L, 1.0f);
    float* part; CK(hipMalloc(&part, (size_t)16 * 64 * d * 4)); CK(hipMemset(part, 0, (size_t)16 * 64 * d * 4));
    float* bias; CK(hipMalloc(&bias, d * 4)); CK(hipMemset(bias, 0, d * 4));
    bf16* o; CK(hipMalloc(&o, (size_t)L * 64 * d * 2));
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int l = 0; l < L; ++l) {
        AttnArgs a; a.dt = DT_BF16; a.n_q = 1; a.q_sb = d; a.q_partial = part; a.q_bias = bias; a.q_nsplit = S; a.q_scale = 0.35f;
        a.K = K + layer * l; a.V = V + layer * l; a.kv_sbh = (int64_t)T * 64; a.n_keys = T; a.kv_group = 1;
        a.O = o + (size_t)l * 64 * d; a.o_sb = d; a.B = B; a.H = H; a.n_split = 1;
        if (launch_attn_decode(a, st)) { printf("launch failed\n"); return 1; }
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * L), bytes = (double)layer * 2 * 2;
    std::vector<uint16_t> ho((size_t)64 * d); CK(hipMemcpy(ho.data(), o, ho.size() * 2, hipMemcpyDeviceToHost));
    unsigned long long cs = 0; for (auto v : ho) cs = cs * 1315423911ull + v;
    printf("cross-attn B=%d variant=%s: %.2f us/launch  %.1f GB/s  checksum %llx\n", B, getenv("WX_XATTN") ? getenv("WX_XATTN") : "0", us, bytes / us * 1e-3, cs);
    return 0;
}
