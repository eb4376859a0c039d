// One MFMA shape held for N seconds with register-resident operands (no LDS, no global traffic), so that a clock / power sampler
// running beside it (tools/power_trace.py) sees the sustained state:  ./mfma_hold <16|32> <zero 0|1> <seconds>
// hipcc --offload-arch=gfx950 -O3 -o mfma_hold mfma_hold.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k16(const half8 *in, float *out, int iters) {
    half8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    floatx4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (floatx4){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(const half8 *in, float *out, int iters) {
    half8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    floatx16 acc[8];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int shape = argc > 1 ? atoi(argv[1]) : 16, zero = argc > 2 ? atoi(argv[2]) : 0;
    const double seconds = argc > 3 ? atof(argv[3]) : 4.0;
    half8 *in; float *out;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 4096 * 256 * 4);
    _Float16 h[512 * 8];
    for (int i = 0; i < 512 * 8; i++) h[i] = zero ? (_Float16)0.f : (_Float16)((float)(rand() % 2001 - 1000) / 1000.f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000 / (shape == 32 ? 1 : 1);           // ~35-45 ms per launch
    const double flop = shape == 32 ? 8 * 32768.0 : 16 * 16384.0;
    double tot_ms = 0; int n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(k32, dim3(512), dim3(256), 0, 0, in, out, iters);
        else hipLaunchKernelGGL(k16, dim3(512), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (n >= 4) tot_ms += ms;                               // (the first launches ramp the clock)
        n++;
    }
    const double tf = (double)512 * 4 * iters * flop / (tot_ms / (n - 4)) / 1e9;
    printf("{\"shape\": \"%s\", \"operands\": \"%s\", \"launches\": %d, \"ms_per_launch\": %.3f, \"tflops\": %.1f}\n",
           shape == 32 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", zero ? "zeros" : "random", n, tot_ms / (n - 4), tf);
    return 0;
}
