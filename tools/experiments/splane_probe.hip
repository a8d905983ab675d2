// Probe: HBM read throughput of the S-plane access pattern of the conv kernels' loader waves against a contiguous stream of the same depth.
// Eight planes [24 items][64 channel groups][P = 2304][16 B] (hi) + the same for lo, as the conditioning-gradient product reads them:
// workgroup (item, 128-step time tile) walks plane 0..7, channel groups 4 at a time: a chunk = 4 rows x 2 KB of hi + 4 x 2 KB of lo, the rows
// 36 KB apart.  DEPTH chunks are requested before the oldest is consumed.   hipcc --offload-arch=gfx950 -O3 ... -o variants/splane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int ITEMS = 24, CG = 64, P = 2304, TILES = 16, PLANES = 8;
constexpr size_t PLANE_UNITS = (size_t)ITEMS * CG * P;      // 16-byte units of one hi (or lo) array

template <int DEPTH, bool CONTIG>
__global__ __launch_bounds__(256) void walk(const u32x4 *in, unsigned *out)
{
    const int item = blockIdx.x / TILES, tile = blockIdx.x % TILES, tid = threadIdx.x;
    const int nchunks = PLANES * (CG / 4);
    // lane -> (row of the chunk 0..7 [4 hi + 4 lo], time step): two loads per lane and chunk
    auto addr = [&](int c, int j) -> const u32x4 * {
        if (CONTIG) return in + ((size_t)blockIdx.x * nchunks + c) * 1024 + j * 256 + tid;
        const int plane = c / (CG / 4), cg0 = (c % (CG / 4)) * 4;
        const int row = (tid >> 7) + 2 * (j & 1), lo = j >> 1;          // j = 0..3: rows {0,1} hi, {2,3} hi, {0,1} lo, {2,3} lo
        return in + (size_t)plane * 2 * PLANE_UNITS + (size_t)lo * PLANE_UNITS + ((size_t)item * CG + cg0 + row) * P + 128 + tile * 128 + (tid & 127);
    };
    u32x4 st[DEPTH][4];
    unsigned acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) st[d][j] = __builtin_nontemporal_load(addr(d, j));
    for (int c = 0; c + DEPTH < nchunks; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += st[d][j][0] ^ st[d][j][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) st[d][j] = __builtin_nontemporal_load(addr(c + DEPTH + d, j));
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main()
{
    const size_t bytes = (size_t)PLANES * 2 * PLANE_UNITS * 16;
    void *in; unsigned *out;
    CHECK(hipMalloc(&in, bytes)); CHECK(hipMalloc(&out, 4096 * 4));
    CHECK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int wgs = ITEMS * TILES;
    const double moved = (double)wgs * PLANES * (CG / 4) * 16384.0;
#define RUN(D, C) { walk<D, C><<<wgs, 256>>>((const u32x4 *)in, out); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); \
    for (int r = 0; r < 5; ++r) walk<D, C><<<wgs, 256>>>((const u32x4 *)in, out); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); printf("%s depth %d: %.2f TB/s (%.0f us per pass of %.0f MB)\n", C ? "contiguous" : "S-plane   ", D, moved * 5 / (ms * 1e-3) / 1e12, ms / 5 * 1e3, moved / 1e6); }
    RUN(1, true) RUN(2, true) RUN(4, true) RUN(1, false) RUN(2, false) RUN(4, false) RUN(8, false)
    return 0;
}
