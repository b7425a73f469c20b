// mb_xattn.hip — stand-alone check + timing of the E-stream cross-attention chain (xattn.hip): q~ -> stream -> projection.
// Checks a few (row, head) results against a double-precision host reference computed from the same bf16 inputs, then
// times each kernel as 32 "layers" in one hipGraph, E alternating between two buffers so that nothing is cache resident.
#include "../whisper_amd/csrc/common.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
using namespace wx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_bf16(bf16* p, size_t n, float scale, unsigned salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ salt; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = (bf16)(((int)(h & 0xffff) - 32768) / 32768.0f * scale);
    }
}
__global__ void fill_f32(float* p, size_t n, float scale, unsigned salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ salt; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = ((int)(h & 0xffff) - 32768) / 32768.0f * scale;
    }
}
static float b2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static float rb(float f) { uint32_t u; memcpy(&u, &f, 4); u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u; float o; memcpy(&o, &u, 4); return o; }
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64;
    const int d = argc > 2 ? atoi(argv[2]) : 1280;
    const int S = argc > 3 ? atoi(argv[3]) : 4;
    const int H = d / 64, T = 1500, L = 32, NSQ = argc > 4 ? atoi(argv[4]) : 4;
    const float scale = powf(64.f, -0.25f);
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t e_elems = (size_t)rows * T * d;
    const int nt16 = ((T + 15) / 16 + 1) / 2 * 2;
    const size_t f_elems = (size_t)rows * nt16 * 16 * d;
    bf16 *E[2], *Ef[2];
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&E[i], e_elems * 2)); CK(hipMalloc(&Ef[i], f_elems * 2));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, st, E[i], e_elems, 1.0f, 17u + i);
        if (launch_xattn_pack_e(E[i], Ef[i], rows, T, d, nt16, st)) { printf("pack failed\n"); return 1; }
    }
    bf16 *WkT, *Wv, *qt, *cpart, *att; float *qpart, *qbias, *bv, *ml;
    CK(hipMalloc(&WkT, (size_t)d * d * 2)); CK(hipMalloc(&Wv, (size_t)d * d * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, WkT, (size_t)d * d, 0.05f, 3u);
    hipLaunchKernelGGL(fill_bf16, dim3(1024), dim3(256), 0, st, Wv, (size_t)d * d, 0.03f, 5u);
    CK(hipMalloc(&qpart, (size_t)NSQ * rows * d * 4)); hipLaunchKernelGGL(fill_f32, dim3(1024), dim3(256), 0, st, qpart, (size_t)NSQ * rows * d, 0.6f, 7u);
    CK(hipMalloc(&qbias, d * 4)); hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, st, qbias, (size_t)d, 0.1f, 9u);
    CK(hipMalloc(&bv, d * 4)); hipLaunchKernelGGL(fill_f32, dim3(8), dim3(256), 0, st, bv, (size_t)d, 0.1f, 11u);
    CK(hipMalloc(&qt, (size_t)rows * H * d * 2)); CK(hipMalloc(&cpart, (size_t)rows * S * H * d * 2)); CK(hipMalloc(&ml, (size_t)rows * S * H * 2 * 4));
    CK(hipMalloc(&att, (size_t)rows * d * 2));
    CK(hipStreamSynchronize(st));
    // the kernels read the two weights in MFMA-fragment order (same packing as wx_finalize's)
    bf16 *Wk_pk, *Wv_pk; CK(hipMalloc(&Wk_pk, (size_t)d * d * 2)); CK(hipMalloc(&Wv_pk, (size_t)d * d * 2));
    {
        std::vector<uint16_t> w0((size_t)d * d), w1((size_t)d * d), k2((size_t)d * d), v2((size_t)d * d);
        CK(hipMemcpy(w0.data(), WkT, w0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(w1.data(), Wv, w1.size() * 2, hipMemcpyDeviceToHost));
        size_t at = 0;
        for (int h = 0; h < H; ++h) for (int it = 0; it < d / 16; ++it) for (int ks = 0; ks < 2; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j)
            k2[at++] = w0[(size_t)(it * 16 + (lane & 15)) * d + h * 64 + ks * 32 + (lane >> 4) * 8 + j];      // WkT[i][c] = Wk[c][i]
        at = 0;
        for (int u = 0; u < d / 16; ++u) for (int kb = 0; kb < d / 32; ++kb) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j)
            v2[at++] = w1[(size_t)(u * 16 + (lane & 15)) * d + kb * 32 + (lane >> 4) * 8 + j];
        CK(hipMemcpy(Wk_pk, k2.data(), k2.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Wv_pk, v2.data(), v2.size() * 2, hipMemcpyHostToDevice));
    }
    XQTildeArgs qa; qa.q_partial = qpart; qa.q_bias = qbias; qa.q_nsplit = NSQ; qa.Wk_pk = Wk_pk; qa.qt = qt; qa.scale = scale; qa.rows = rows; qa.H = H; qa.d = d;
    XAttnArgs xa; xa.nt16 = nt16; xa.qt = qt; xa.E = Ef[0]; xa.cpart = cpart; xa.ml = ml; xa.rows = rows; xa.H = H; xa.d = d; xa.T = T; xa.group = 1; xa.n_split = S;
    XCProjArgs ca; ca.cpart = cpart; ca.ml = ml; ca.n_split = S; ca.Wv_pk = Wv_pk; ca.bv = bv; ca.out = att; ca.ldo = d; ca.rows = rows; ca.H = H; ca.d = d;
    if (launch_xattn_qtilde(qa, st) || launch_xattn_estream(xa, st) || launch_xattn_cproj(ca, st)) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    // ---- host reference for a few rows --------------------------------------------------------------------------
    std::vector<uint16_t> hE((size_t)T * d), hWkT((size_t)d * d), hWv((size_t)d * d), hqt((size_t)rows * H * d), hatt((size_t)rows * d);
    std::vector<float> hqp((size_t)NSQ * rows * d), hqb(d), hbv(d);
    CK(hipMemcpy(hWkT.data(), WkT, hWkT.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hWv.data(), Wv, hWv.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hqt.data(), qt, hqt.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hatt.data(), att, hatt.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hqp.data(), qpart, hqp.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hqb.data(), qbias, d * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hbv.data(), bv, d * 4, hipMemcpyDeviceToHost));
    double worst_q = 0, worst_o = 0, ref_q = 0, ref_o = 0;
    const int probe_rows[3] = {0, rows / 2, rows - 1};
    for (int pr = 0; pr < 3; ++pr) {
        const int row = probe_rows[pr];
        CK(hipMemcpy(hE.data(), E[0] + (size_t)row * T * d, hE.size() * 2, hipMemcpyDeviceToHost));
        for (int h = 0; h < H; h += (H > 4 ? H / 4 : 1)) {
            std::vector<float> qb(64), qtil(d);
            for (int c = 0; c < 64; ++c) { float v = hqb[h * 64 + c]; for (int s = 0; s < NSQ; ++s) v += hqp[((size_t)s * rows + row) * d + h * 64 + c]; qb[c] = rb(v * scale); }
            for (int i = 0; i < d; ++i) {
                double acc = 0; for (int c = 0; c < 64; ++c) acc += (double)qb[c] * b2f(hWkT[(size_t)i * d + h * 64 + c]);
                const double want = acc * scale, got = b2f(hqt[((size_t)row * H + h) * d + i]);
                worst_q = fmax(worst_q, fabs(want - got)); ref_q = fmax(ref_q, fabs(want));
                qtil[i] = (float)got;                       // continue from the device's rounded q~
            }
            std::vector<double> sc(T); double mx = -1e300;
            for (int j = 0; j < T; ++j) { double a = 0; for (int i = 0; i < d; ++i) a += (double)qtil[i] * b2f(hE[(size_t)j * d + i]); sc[j] = a; mx = fmax(mx, a); }
            double l = 0; for (int j = 0; j < T; ++j) { sc[j] = exp(sc[j] - mx); l += sc[j]; }
            std::vector<double> c(d, 0.0);
            for (int j = 0; j < T; ++j) { const double p = sc[j] / l; if (p < 1e-12) continue; for (int i = 0; i < d; ++i) c[i] += p * b2f(hE[(size_t)j * d + i]); }
            for (int cc = 0; cc < 64; ++cc) {
                double o = hbv[h * 64 + cc]; for (int i = 0; i < d; ++i) o += c[i] * b2f(hWv[(size_t)(h * 64 + cc) * d + i]);
                const double got = b2f(hatt[(size_t)row * d + h * 64 + cc]);
                worst_o = fmax(worst_o, fabs(o - got)); ref_o = fmax(ref_o, fabs(o));
            }
        }
    }
    printf("check rows=%d d=%d S=%d: q~ max err %.3e (scale %.3e), out max err %.3e (scale %.3e)\n", rows, d, S, worst_q, ref_q, worst_o, ref_o);
    // ---- timing ---------------------------------------------------------------------------------------------------
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 4; ++which) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int l = 0; l < L; ++l) {
            xa.E = Ef[l & 1];
            int rc = 0;
            if (which == 0 || which == 3) rc |= launch_xattn_qtilde(qa, st);
            if (which == 1 || which == 3) rc |= launch_xattn_estream(xa, st);
            if (which == 2 || which == 3) rc |= launch_xattn_cproj(ca, st);
            if (rc) { printf("launch failed\n"); return 1; }
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        const int reps = 10;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * L);
        const char* names[4] = {"qtilde", "estream", "cproj", "chain"};
        if (which == 1) printf("%-8s %.2f us/launch  %.1f GB/s of E\n", names[which], us, (double)e_elems * 2 / us * 1e-3);
        else printf("%-8s %.2f us/launch\n", names[which], us);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
