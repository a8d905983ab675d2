// Probe: HBM read throughput of 512 persistent workgroups (two per CU) against the number of 16-byte loads each lane keeps in flight,
// i.e. how much memory-level parallelism the conv kernels' loader waves need.  Each workgroup streams its contiguous share of a 1 GB
// buffer: a "chunk" = 256 lanes x 4 x 16 B = 16 KB (the B operand of one K chunk of convgemm16q at NI = 2); DEPTH chunks are requested
// before the oldest is consumed.      hipcc --offload-arch=gfx950 -O3 tools/experiments/mlp_probe.hip -o variants/mlp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void stream(const u32x4 *in, unsigned *out, size_t units_per_wg)
{
    const u32x4 *p = in + (size_t)blockIdx.x * units_per_wg + threadIdx.x;
    const size_t nchunks = units_per_wg / 1024;              // 1024 units of 16 B per chunk
    u32x4 st[DEPTH][4];
    unsigned acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) st[d][j] = __builtin_nontemporal_load(p + (size_t)d * 1024 + j * 256);
    for (size_t c = 0; c + DEPTH < nchunks; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += st[d][j][0] ^ st[d][j][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) st[d][j] = __builtin_nontemporal_load(p + (c + DEPTH + d) * 1024 + j * 256);
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30, units = bytes / 16;
    void *in; unsigned *out;
    CHECK(hipMalloc(&in, bytes)); CHECK(hipMalloc(&out, 4096 * 4));
    CHECK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wgs : {256, 512, 1024}) {
        const size_t upw = units / wgs;
#define RUN(D) { stream<D><<<wgs, 256>>>((const u32x4 *)in, out, upw); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); \
        for (int r = 0; r < 5; ++r) stream<D><<<wgs, 256>>>((const u32x4 *)in, out, upw); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); printf("wgs %4d depth %d (%3d KB in flight per WG): %.2f TB/s\n", wgs, D, D * 16, bytes * 5 / (ms * 1e-3) / 1e12); }
        RUN(1) RUN(2) RUN(3) RUN(4) RUN(6) RUN(8)
    }
    return 0;
}
