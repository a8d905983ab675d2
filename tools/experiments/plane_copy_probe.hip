// Probe: HBM throughput of the epilogue's fp32 plane access pattern (out[row][t] with lane = t, one dword per lane, two 128-B
// segments per wave instruction) against a row-contiguous 16-B-per-lane pattern, on planes of the headline shape
// ([24][512 rows][P = 2304] floats).  x' = x + 1 (one read + one write per element).   hipcc --offload-arch=gfx950 -O3 ... -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

// pattern A: tile 128 rows x 128 t per block of 256 threads (4 waves as 2x2 of 64x64); lane -> column (lane & 31), h = lane >> 5
__global__ __launch_bounds__(256) void copy_a(const float *in, float *out, int rows, int P, int H)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int t0 = blockIdx.x * 128, m0 = blockIdx.y * 128, b = blockIdx.z, col = lane & 31, h = lane >> 5;
    float v[2][2][16];
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, t = t0 + wc * 64 + ni * 32 + col;
        v[mi][ni][r] = in[((size_t)b * rows + m) * P + H + t];
    }
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, t = t0 + wc * 64 + ni * 32 + col;
        out[((size_t)b * rows + m) * P + H + t] = v[mi][ni][r] + 1.f;
    }
}
// pattern B: same tile, each lane moves float4 (4 consecutive t); a wave instruction covers 2 rows x 512 B
__global__ __launch_bounds__(256) void copy_b(const float *in, float *out, int rows, int P, int H)
{
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 128, m0 = blockIdx.y * 128, b = blockIdx.z;
    float4 v[16];
    for (int i = 0; i < 16; ++i) {
        const int u = tid + 256 * i, m = m0 + (u >> 5), t = t0 + (u & 31) * 4;
        v[i] = *reinterpret_cast<const float4 *>(in + ((size_t)b * rows + m) * P + H + t);
    }
    for (int i = 0; i < 16; ++i) {
        const int u = tid + 256 * i, m = m0 + (u >> 5), t = t0 + (u & 31) * 4;
        float4 w = v[i]; w.x += 1.f; w.y += 1.f; w.z += 1.f; w.w += 1.f;
        *reinterpret_cast<float4 *>(out + ((size_t)b * rows + m) * P + H + t) = w;
    }
}
int main()
{
    const int B = 24, rows = 512, H = 128, Tt = 2048, P = H + Tt + H;
    const size_t n = (size_t)B * rows * P;
    float *a, *c;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&c, n * 4));
    CHECK(hipMemset(a, 0, n * 4)); CHECK(hipMemset(c, 0, n * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(Tt / 128, rows / 128, B);
    const double bytes = 2.0 * B * rows * Tt * 4;
    for (int pat = 0; pat < 2; ++pat) {
        for (int it = 0; it < 3; ++it) { if (pat == 0) copy_a<<<grid, 256>>>(a, c, rows, P, H); else copy_b<<<grid, 256>>>(a, c, rows, P, H); }
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 20; ++it) { if (pat == 0) copy_a<<<grid, 256>>>(a, c, rows, P, H); else copy_b<<<grid, 256>>>(a, c, rows, P, H); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("pattern %c: %.1f us per launch, %.2f TB/s (read+write)\n", pat ? 'B' : 'A', ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
