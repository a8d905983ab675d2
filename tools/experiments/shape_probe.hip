// Which bf16 MFMA shape sustains more FLOP/s under the chip's power management?  (MI355X_MICROARCH.md, "DVFS give-back" item 7.)
// The conv kernels of this engine run their main loop at 1.34-1.42 GHz on random data (stamps in convgemm16w_kernel, -DWG_DBG_TRACE):
// they are bound by the clock the chip holds, so cycles per FLOP do not decide which shape is faster.
//
// Both kernels do what a compute wave of convgemm16w does per 32-deep chunk: a 64x64 output tile per wave, every operand fragment
// re-read from LDS (hi and lo images of A and B: 16 ds_read_b128), three products per fragment pair (a_lo b_hi + a_hi b_lo + a_hi b_hi):
//   shape 0: v_mfma_f32_32x32x16_bf16, 2x2 blocks, two k-steps  -> 24 MFMAs x 32 cycles per chunk
//   shape 1: v_mfma_f32_16x16x32_bf16, 4x4 blocks, one k-step   -> 48 MFMAs x 16 cycles per chunk
// LDS holds random bf16 data; 8 waves per CU (two per SIMD) as in the real kernel.  Reports wall time, TFLOP/s (issued) and the
// in-kernel clock (s_memtime / s_memrealtime) after ~2 s of back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -o shape_probe shape_probe.hip && ./shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHUNKS 4096
#define IMG (128 * 80)          // one 128-row image, 80-byte rows (padded) -- the 64-byte swizzled form fits inside as well

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(const u32x4 *__restrict__ rnd, float *out, unsigned long long *stamps)
{
    __shared__ __attribute__((aligned(16))) char smem[4 * IMG];      // A hi, A lo, B hi, B lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * IMG / 16; i += 512) reinterpret_cast<u32x4 *>(smem)[i] = rnd[(blockIdx.x * 37 + i) & 65535];
    __syncthreads();
    const int wr = (wave >> 1) & 1, wc = wave & 1;
    auto rd = [&](int off) { return *reinterpret_cast<const bf16x8 *>(smem + off); };
    unsigned long long c0 = 0, w0 = 0;
    if (tid == 0) { c0 = clock64(); w0 = wall_clock64(); }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        const int r = lane & 31, h = lane >> 5;
        const int ao = (wr * 64 + r) * 80 + h * 16, bo = (wc * 64 + r) * 80 + h * 16;
        for (int c = 0; c < CHUNKS; ++c) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = rd(ao + ks * 32 + i * 32 * 80); al[i] = rd(IMG + ao + ks * 32 + i * 32 * 80);
                    bh[i] = rd(2 * IMG + bo + ks * 32 + i * 32 * 80); bl[i] = rd(3 * IMG + bo + ks * 32 + i * 32 * 80);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
        // 64-byte rows, unit position = k-group ^ f(row), f = [0,2,3,1][(row >> 1) & 3]: conflict-free fragment reads for the 16x16x32
        // lane map (row = lane & 15, k-group = lane >> 4) and conflict-free lane-linear staging writes
        const int r16 = lane & 15, kg = lane >> 4;
        const int f = (0x1320 >> (4 * ((r16 >> 1) & 3))) & 3;          // nibbles: q=0 ->0, 1 ->2, 2 ->3, 3 ->1
        const int ao = (wr * 64 + r16) * 64 + ((kg ^ f) << 4), bo = (wc * 64 + r16) * 64 + ((kg ^ f) << 4);
        for (int c = 0; c < CHUNKS; ++c) {
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { ah[i] = rd(ao + i * 1024); al[i] = rd(IMG + ao + i * 1024); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 bh = rd(2 * IMG + bo + j * 1024), bl = rd(3 * IMG + bo + j * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    }
    if (tid == 0) { stamps[2 * blockIdx.x] = clock64() - c0; stamps[2 * blockIdx.x + 1] = wall_clock64() - w0; }
    out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE> void run(const u32x4 *rnd, float *out, unsigned long long *stamps, const char *name)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    int reps = 0;
    for (int round = 0; round < 40; ++round) {                         // ~2 s of back-to-back launches, the last batch is timed
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(probe<SHAPE>, dim3(256), dim3(512), 0, 0, rnd, out, stamps);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        reps = 8;
    }
    std::vector<unsigned long long> st(512);
    (void)hipMemcpy(st.data(), stamps, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int b = 0; b < 256; ++b) ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] / 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double flops = 256.0 * 8 * (double)CHUNKS * 2.0 * 64 * 64 * 32 * 3;        // issued (3 products per element)
    printf("%-22s %8.2f ms per launch  %7.1f TFLOP/s issued   in-kernel clock %.2f GHz (median of 256 workgroups)\n", name, ms / reps,
           flops / (ms / reps * 1e-3) / 1e12, ghz[128]);
}

int main()
{
    u32x4 *rnd;
    float *out;
    unsigned long long *stamps;
    (void)hipMalloc(&rnd, 65536 * 16);
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&stamps, 512 * 8);
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto &v : h) {                                                // random bf16 in (-2, 2): sign, exponent 125..127, random mantissa
        const unsigned m = rand() & 0x7f, e = 125 + rand() % 3, s = rand() & 1;
        v = (unsigned short)((s << 15) | (e << 7) | m);
    }
    (void)hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(rnd, out, stamps, "32x32x16 (2x2 blocks)");
        run<1>(rnd, out, stamps, "16x16x32 (4x4 blocks)");
    }
    return 0;
}
