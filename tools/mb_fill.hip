// mb_fill.hip — L2 -> LDS fill rate of one CU as a function of the bytes kept in flight, in the access shape of the 256x256 encoder GEMM:
// 512 threads, LDS-DMA (global_load_lds, 16 B per lane), one "unit" = 128 rows x 128 B (row stride = K * 2 bytes) = 16 KiB = 2 wave-instructions
// per wave, a ring of RING units in LDS, DEPTH units in flight (a counted s_waitcnt vmcnt + s_barrier per unit, as in the GEMM's K loop).
// The sources are panels shared by the workgroups of an XCD (blockIdx % 8), so the bytes come from L2, not HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
template <int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, int K2, int n_units, int panels, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RING = DEPTH + 1;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // this workgroup's panel: one of `panels` panels of 128 rows, shared with the other workgroups of its XCD group
    const int xcd = blockIdx.x % 8, slot = (blockIdx.x / 8) % panels;
    const char* base = src + ((size_t)(xcd * panels + slot) * 128) * K2;
    const char* s0 = base + (size_t)((wid * 2) * 8 + (lane >> 3)) * K2 + (lane & 7) * 16;
    const char* s1 = s0 + (size_t)8 * K2;
    const int units_per_row = K2 / 128;
    auto issue = [&](int u) {
        char* dst = lds + (u % RING) * 16384 + (wid * 2) * 1024;
        const int k = (u % units_per_row) * 128;
        __builtin_amdgcn_global_load_lds((gbl_void*)(s0 + k), (lds_void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void*)(s1 + k), (lds_void*)(dst + 1024), 16, 0, 0);
    };
    for (int u = 0; u < DEPTH; ++u) issue(u);
    unsigned acc = 0;
    for (int u = 0; u < n_units; ++u) {
        if (u + DEPTH < n_units) issue(u + DEPTH);
        // retire unit u: at most DEPTH units (2 * DEPTH instructions) may stay outstanding
        if (u + DEPTH < n_units) {
            if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if constexpr (DEPTH == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<const unsigned*>(lds + (u % RING) * 16384 + tid * 4);     // one read per unit keeps the loads live
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
template <int DEPTH>
static void run(const char* src, int K2, int n_units, int panels, unsigned* sink, hipStream_t st) {
    const int ldsb = (DEPTH + 1) * 16384;
    CK(hipFuncSetAttribute((const void*)fill<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill<DEPTH>, dim3(256), dim3(512), ldsb, st, src, K2, n_units, panels, sink);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill<DEPTH>, dim3(256), dim3(512), ldsb, st, src, K2, n_units, panels, sink);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 5, bytes = (double)n_units * 16384;
    printf("  %d units (%3d KiB) in flight: %7.1f us per launch, %5.1f GB/s per CU, %5.2f TB/s chip\n", DEPTH, DEPTH * 16, us, bytes / us * 1e-3, bytes * 256 / us * 1e-6);
}
int main(int argc, char** argv) {
    const int K2 = 2560;                       // K = 1280 bf16
    const int n_units = 4096;                  // 64 MiB per workgroup
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    for (int panels : {4, 32}) {               // 4 panels per XCD group = 1.3 MB (L2 resident); 32 = 10.5 MB per group (spills L2: Infinity Cache)
        char* src; const size_t bytes = (size_t)8 * panels * 128 * K2;
        CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
        printf("%d panels of 128 rows x 2560 B per XCD group (%.1f MB per group):\n", panels, panels * 128.0 * K2 / 1e6);
        run<1>(src, K2, n_units, panels, sink, st); run<2>(src, K2, n_units, panels, sink, st); run<3>(src, K2, n_units, panels, sink, st);
        run<4>(src, K2, n_units, panels, sink, st); run<6>(src, K2, n_units, panels, sink, st); run<8>(src, K2, n_units, panels, sink, st);
        CK(hipFree(src));
    }
    return 0;
}
