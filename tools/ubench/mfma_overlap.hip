// Does a wave's own memory-instruction issue overlap its MFMAs?  One k-step of the one-board tower per iteration: 6 independent
// v_mfma_f32_16x16x32_f16 (96 cycles of matrix pipe), optionally 2 global_load_dwordx4 (weight fragments, L2-resident) and / or 3
// ds_read_b128 (activation fragments), results consumed a ring turn later like in k_tower2.  Run with ONE wave per SIMD (256 threads,
// one workgroup per CU forced by its LDS size) and with TWO (512 threads): if the parts of one wave add up instead of overlapping, the
// one-wave numbers are the sum and the two-wave numbers approach the maximum.
// hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip ; ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); return 1; } } while (0)
// every operand is fetched D = 8 k-steps before the MFMAs that use it (register rings), so no latency is exposed: what is measured is
// ISSUE -- whether a wave can issue its loads while its own MFMAs execute
template <bool MFMA, bool VMEM, bool LDS>
__global__ __launch_bounds__(512) void k(const half8 *w, float *out, unsigned long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < 4096; c += blockDim.x) reinterpret_cast<uint4 *>(smem)[c] = make_uint4(c, c, c, c);
    __syncthreads();
    constexpr int D = 8;
    floatx4 acc[6];
    for (int i = 0; i < 6; i++) acc[i] = (floatx4){0, 0, 0, 0};
    half8 a[D][2], b[D][3];
    const half8 *wp = w + (size_t)(blockIdx.x & 7) * 8192 + (wave & 3) * 128 + lane;
    const char *lp = smem + lane * 16;
#pragma unroll
    for (int j = 0; j < D; j++) {
        a[j][0] = wp[j * 1024]; a[j][1] = wp[j * 1024 + 64];
#pragma unroll
        for (int q = 0; q < 3; q++) b[j][q] = *reinterpret_cast<const half8 *>(lp + ((j * 3 + q) & 31) * 1024);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += D) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            half8 a0 = a[u][0], a1 = a[u][1], b0 = b[u][0], b1 = b[u][1], b2 = b[u][2];
            if (VMEM) { a[u][0] = wp[((it + u + D) & 31) * 1024]; a[u][1] = wp[((it + u + D) & 31) * 1024 + 64]; }
            if (LDS) {
#pragma unroll
                for (int q = 0; q < 3; q++) b[u][q] = *reinterpret_cast<const half8 *>(lp + (((it + u + D) * 3 + q) & 31) * 1024);
            }
            if (MFMA) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[3], 0, 0, 0);
                acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b2, acc[4], 0, 0, 0); acc[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acc[5], 0, 0, 0);
            } else {                                            // (one cheap use of every fetched register, so that the loads are not dead)
                acc[0][0] += (float)a0[0]; acc[1][0] += (float)a1[0]; acc[2][0] += (float)b0[0]; acc[3][0] += (float)b1[0]; acc[4][0] += (float)b2[0];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 6; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool M, bool V, bool L>
static int run(const char *name, const half8 *w, float *out, unsigned long long *cyc, int threads) {
    const int iters = 4096, grid = 256;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k<M, V, L>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<M, V, L>), dim3(grid), dim3(threads), 100 * 1024, 0, w, out, cyc, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long h[256]; CHK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double m = 0; for (int i = 0; i < 256; i++) m += (double)h[i];
    printf("%-34s %d wave(s)/SIMD: %7.1f cycles (s_memtime) per k-step and wave, launch %.3f ms\n", name, threads / 256, m / 256 / iters, best);
    return 0;
}

int main() {
    half8 *w; float *out; unsigned long long *cyc;
    CHK(hipMalloc(&w, 8 * 8192 * 16 + 64 * 1024 * 16)); CHK(hipMalloc(&out, 256 * 512 * 4)); CHK(hipMalloc(&cyc, 256 * 8));
    CHK(hipMemset(w, 0, 8 * 8192 * 16 + 64 * 1024 * 16));
    for (int threads = 256; threads <= 512; threads += 256) {
        run<true, false, false>("6 MFMA", w, out, cyc, threads);
        run<false, true, false>("2 global_load_dwordx4", w, out, cyc, threads);
        run<false, false, true>("3 ds_read_b128", w, out, cyc, threads);
        run<true, true, false>("6 MFMA + 2 global loads", w, out, cyc, threads);
        run<true, false, true>("6 MFMA + 3 ds_reads", w, out, cyc, threads);
        run<true, true, true>("6 MFMA + 2 global + 3 ds_reads", w, out, cyc, threads);
    }
    return 0;
}
