#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8; s16x4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(threadIdx.x * 0.001f + j); b8[j] = (__bf16)(j * 0.5f); }
    for (int j = 0; j < 4; ++j) { a4[j] = (short)(threadIdx.x + j); b4[j] = (short)(j * 3 + 1); }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 256;
    float* out; long long* cyc; hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 16);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms0, ms1;
        hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms0, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("%d blocks x %d threads: 16x16x32 %.1f ns per MFMA per wave (%lld cycles per MFMA by the shader clock); 16x16x16 %.1f ns\n", blocks, threads, ms0 * 1e6 / (iters * 8.0), h[0] / (iters * 8LL), ms1 * 1e6 / (iters * 8.0));
    }
    return 0;
}
