// mb_cumask.hip — which CUs (XCC, SE, SH, CU) a stream created with hipExtStreamCreateWithCUMask runs on, per mask pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void where(unsigned* out, int spin) {
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv) {
    const int nblk = 4096; setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    unsigned* d; CK(hipMalloc(&d, nblk * 8));
    std::vector<unsigned> h(nblk * 2);
    for (int pat = 0; pat < 6; ++pat) {
        if (only >= 0 && pat != only) continue;
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const char* name = "";
        if (pat == 0) { for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; name = "all 256 bits"; }
        if (pat == 1) { mask[0] = mask[1] = 0xffffffffu; name = "bits 0..63"; }
        if (pat == 2) { mask[2] = mask[3] = 0xffffffffu; name = "bits 64..127"; }
        if (pat == 3) { for (int i = 0; i < 8; ++i) mask[i] = 0x03030303u; name = "bits with (i % 8) < 2"; }
        if (pat == 4) { for (int i = 0; i < 8; ++i) mask[i] = 0x0c0c0c0cu; name = "bits with (i % 8) in {2, 3}"; }
        if (pat == 5) { mask[0] = 0xffu; name = "bits 0..7"; }
        hipStream_t st; fprintf(stderr, "pattern %d: create\n", pat);
        CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        fprintf(stderr, "created\n"); CK(hipMemset(d, 0xff, nblk * 8)); fprintf(stderr, "memset\n");
        hipLaunchKernelGGL(where, dim3(nblk), dim3(64), 0, st, d, 20000);
        CK(hipGetLastError()); fprintf(stderr, "launched\n"); CK(hipStreamSynchronize(st)); fprintf(stderr, "synced\n");
        CK(hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int b = 0; b < nblk; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        int total = 0;
        printf("mask %-28s:", name);
        for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf("  -> %d distinct CUs\n", total);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
