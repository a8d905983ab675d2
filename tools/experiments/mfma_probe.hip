// Issue-rate probe for v_mfma_f32_32x32x16_bf16 on gfx950: how many cycles per MFMA when consecutive MFMAs accumulate into
// 1, 2 or 4 different accumulators, at 1 / 2 / 4 waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2000
template <int NACC>
__global__ __launch_bounds__(1024) void probe(float *out)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 24; ++k)
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int threads, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<NACC>, dim3(256), dim3(threads), 0, 0, out);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<NACC>, dim3(256), dim3(threads), 0, 0, out);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)ITERS * 24 * (threads / 256);
    const double flops = 256.0 * 4 * mfma_per_simd * 32768;
    printf("accumulators %d, waves/SIMD %d: %8.1f us, %6.2f ns per MFMA per SIMD, %7.1f TFLOP/s\n", NACC, threads / 256, ms * 1e3,
           ms * 1e6 / mfma_per_simd, flops / (ms * 1e-3) / 1e12);
}
int main()
{
    float *out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    for (int threads : {256, 512, 1024}) {
        run<1>(threads, out);
        run<2>(threads, out);
        run<4>(threads, out);
    }
    return 0;
}
