// Sustained MFMA rate of the two fp16 shapes with register-resident operands (no LDS, no global traffic): what the power budget
// allows each shape.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(const half8 *in, float *out, int iters) {
    half8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (floatx4){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the operands CHANGE from MFMA to MFMA like in a real inner loop (A alternates between two fragments, B advances every second
// MFMA through eight fragments): the operand buses toggle, unlike in k16 where every MFMA reads the same registers
template <int NACC>
__global__ __launch_bounds__(256) void k16v(const half8 *in, float *out, int iters) {
    half8 a[2], b[8];
    for (int i = 0; i < 2; i++) a[i] = in[(threadIdx.x + 64 * i) & 511];
    for (int i = 0; i < 8; i++) b[i] = in[(threadIdx.x + 32 * i + 7) & 511];
    floatx4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (floatx4){0, 0, 0, 0};
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 1]), "v"(b[i >> 1]));
    }
    float s = 0; for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(const half8 *in, float *out, int iters) {
    half8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int zero = argc > 1 ? atoi(argv[1]) : 0;           // 1: zero operands (the DVFS give-back case)
    half8 *in; float *out;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 4096 * 256 * 4);
    _Float16 h[512 * 8];
    for (int i = 0; i < 512 * 8; i++) h[i] = zero ? (_Float16)0.f : (_Float16)((float)(rand() % 2001 - 1000) / 1000.f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;    // 200000: ~30-60 ms per launch, the sustained (power-limited) clock
#define RUN(NAME, KERN, NACC, FL) for (int wg = 1; wg <= 2; wg++) { float best = 1e9; for (int rep = 0; rep < 3; rep++) { \
        hipEventRecord(e0); hipLaunchKernelGGL(KERN<NACC>, dim3(256 * wg), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } \
        printf("%s x%d accumulators, %d wave(s)/SIMD, %s: %.0f TFLOP/s\n", NAME, NACC, wg, zero ? "zeros" : "random", (double)256 * wg * 4 * iters * NACC * FL / best / 1e9); }
    RUN("16x16x32", k16, 4, 16384.0) RUN("16x16x32", k16, 8, 16384.0) RUN("16x16x32", k16, 16, 16384.0)
    RUN("16x16x32 changing operands", k16v, 8, 16384.0)
    RUN("32x32x16", k32, 2, 32768.0) RUN("32x32x16", k32, 4, 32768.0) RUN("32x32x16", k32, 8, 32768.0)
    return 0;
}
