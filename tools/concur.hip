// concur.hip — do two hipGraphs launched on two streams overlap on this GPU?  (experiment for the decode scheduler)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void stream_kernel(const float4* __restrict__ in, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void small_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 1.2345f) p[0] = v;
}
static hipGraphExec_t cap(hipStream_t st, void (*body)(hipStream_t, void*), void* arg) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed)); body(st, arg); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g)); return ge;
}
struct Ctx { float4* big; float* out; size_t n; float* small; };
// a "layer": 5 short latency-bound kernels, then one 245 MB stream; 32 layers
static void bodyL(hipStream_t st, void* a) { Ctx* c = (Ctx*)a; for (int l = 0; l < 32; ++l) { for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(small_kernel, dim3(200), dim3(256), 0, st, c->small, 300); hipLaunchKernelGGL(stream_kernel, dim3(640), dim3(256), 0, st, c->big, c->out, c->n); } }
static void bodyA(hipStream_t st, void* a) { bodyL(st, a); }
static void bodyB(hipStream_t st, void* a) { bodyL(st, a); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    Ctx c; c.n = (size_t)240 << 20 >> 4;   // 240 MiB
    CK(hipMalloc(&c.big, c.n * 16)); CK(hipMemset(c.big, 0, c.n * 16)); CK(hipMalloc(&c.out, 64)); CK(hipMalloc(&c.small, 4096)); CK(hipMemset(c.small, 0, 4096));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipGraphExec_t gA = cap(s1, bodyA, &c), gB = cap(s2, bodyB, &c);
    hipEvent_t evJ, evF; CK(hipEventCreateWithFlags(&evJ, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evF, hipEventDisableTiming)); CK(hipEventRecord(evJ, s1));
    auto run = [&](int mode) {   // 0: A only, 1: B only, 2: both graphs on two streams, 3: both eager on two streams
        CK(hipDeviceSynchronize()); double t0 = now();
        for (int r = 0; r < 5; ++r) {
            if (mode == 0 || mode == 2) CK(hipGraphLaunch(gA, s1));
            if (mode == 1 || mode == 2) CK(hipGraphLaunch(gB, s2));
            if (mode == 3) { bodyA(s1, &c); bodyB(s2, &c); }
            if (mode == 4) {   // the engine's per-step pattern
                CK(hipStreamWaitEvent(s2, evJ, 0)); CK(hipGraphLaunch(gB, s2)); CK(hipEventRecord(evF, s2));
                CK(hipGraphLaunch(gA, s1)); CK(hipStreamWaitEvent(s1, evF, 0));
                hipLaunchKernelGGL(small_kernel, dim3(64), dim3(1024), 0, s1, c.small, 100); CK(hipEventRecord(evJ, s1));
            }
        }
        CK(hipDeviceSynchronize()); return (now() - t0) / 5 * 1e6;
    };
    for (int w = 0; w < 2; ++w) for (int m = 0; m < 5; ++m) printf("mode %d: %.1f us\n", m, run(m));
    return 0;
}
