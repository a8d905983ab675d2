// LDS bank-conflict probe for gfx950: which ds_read_b128 / ds_write_b128 lane->address patterns are conflict free?
//   hipcc --offload-arch=gfx950 -O3 -o lds_probe lds_probe.hip && ./lds_probe
// Each launch runs one pattern: every wave issues ITERS x 8 back-to-back 16-byte LDS accesses at the pattern's addresses; the
// host prints the cycles per access of a saturated CU (8 waves).  Under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE the
// dispatches appear in the order printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ITERS 512
struct Pat { int kind; int a, b, c; const char *name; };   // kind 0: read, 1: write
__device__ __forceinline__ int pat_addr(int id, int a, int b, int c, int l)
{
    switch (id) {
    case 0: return (l & 31) * a + (l >> 5) * b;                       // rows by lane%32, halves offset b       (conv fragment read)
    case 1: return l * a;                                             // consecutive lanes, stride a               (B staging write)
    case 2: return (l >> 2) * a + (l & 3) * 16;                       // 4 lanes per row                           (old A staging write)
    case 3: return (l & 31) * 64 + (((l >> 5) + c) ^ (((l & 31) >> a) & 3)) * 16;   // unpadded rows, XOR swizzle by row >> a
    case 4: return (l & 15) * a + (l >> 4) * b;                       // 16 rows x 4 k-groups                      (16x16 fragment)
    case 5: return (l & 31) * a + (l >> 5) * b + (((l & 31) >> 3) * c);   // extra skew every 8 rows
    }
    return 0;
}
__global__ __launch_bounds__(512) void probe(int kind, int id, int a, int b, int c, unsigned *out, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(smem) + (unsigned)(pat_addr(id, a, b, c, l) & 0x7fff) + w * 4096 * 0;
    u32x4 acc = {0, 0, 0, 0}, v0, v1, v2, v3;
    const long long t0 = clock64();
    if (kind == 0) {
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:10240\n ds_read_b128 %3, %4 offset:15360\n"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
            acc ^= v0 ^ v1 ^ v2 ^ v3;
            asm volatile("ds_read_b128 %0, %4 offset:20480\n ds_read_b128 %1, %4 offset:25600\n ds_read_b128 %2, %4 offset:30720\n ds_read_b128 %3, %4 offset:2560\n"
                         "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
            acc ^= v0 ^ v1 ^ v2 ^ v3;
        }
    } else {
        v0 = u32x4{(unsigned)l, 1, 2, 3};
        for (int i = 0; i < ITERS; ++i) {
            asm volatile("ds_write_b128 %1, %0\n ds_write_b128 %1, %0 offset:5120\n ds_write_b128 %1, %0 offset:10240\n ds_write_b128 %1, %0 offset:15360\n"
                         "ds_write_b128 %1, %0 offset:20480\n ds_write_b128 %1, %0 offset:25600\n ds_write_b128 %1, %0 offset:30720\n ds_write_b128 %1, %0 offset:2560\n"
                         "s_waitcnt lgkmcnt(0)" :: "v"(v0), "v"(addr) : "memory");
        }
        __syncthreads();
        acc = *reinterpret_cast<u32x4 *>(smem + (threadIdx.x & 255) * 16);
    }
    const long long t1 = clock64();
    if (l == 0 && w == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}
int main()
{
    struct P { int kind, id, a, b, c; const char *name; };
    std::vector<P> ps = {
        {0, 1, 16, 0, 0, "read  linear l*16 (reference: conflict free)"},
        {0, 0, 80, 16, 0, "read  (l%32)*80 + (l/32)*16   [conv fragments today]"},
        {0, 0, 64, 16, 0, "read  (l%32)*64 + (l/32)*16   [unpadded]"},
        {0, 0, 96, 16, 0, "read  (l%32)*96 + (l/32)*16"},
        {0, 0, 112, 16, 0, "read  (l%32)*112 + (l/32)*16"},
        {0, 0, 144, 16, 0, "read  (l%32)*144 + (l/32)*16"},
        {0, 0, 80, 32, 0, "read  (l%32)*80 + (l/32)*32"},
        {0, 0, 80, 2560, 0, "read  (l%32)*80 + (l/32)*2560  [halves 32 rows apart]"},
        {0, 0, 80, 2560 + 16, 0, "read  (l%32)*80 + (l/32)*(2560+16)"},
        {0, 0, 72, 16, 0, "read  (l%32)*72 + (l/32)*16    [8-byte aligned rows]"},
        {0, 0, 68, 16, 0, "read  (l%32)*68 + (l/32)*16    [4-byte aligned rows]"},
        {0, 3, 0, 0, 0, "read  unpadded, seg ^ (row & 3)"},
        {0, 3, 1, 0, 0, "read  unpadded, seg ^ ((row>>1) & 3)"},
        {0, 3, 2, 0, 0, "read  unpadded, seg ^ ((row>>2) & 3)"},
        {0, 3, 3, 0, 0, "read  unpadded, seg ^ ((row>>3) & 3)"},
        {0, 4, 80, 16, 0, "read  (l%16)*80 + (l/16)*16   [16-row fragment, 64B of k]"},
        {0, 4, 144, 32, 0, "read  (l%16)*144 + (l/16)*32  [16-row fragment, 128B rows]"},
        {0, 5, 64, 16, 16, "read  unpadded + 16 B skew per 8 rows"},
        {0, 5, 64, 16, 32, "read  unpadded + 32 B skew per 8 rows"},
        {1, 1, 16, 0, 0, "write linear l*16"},
        {1, 1, 80, 0, 0, "write l*80                     [B staging today]"},
        {1, 2, 80, 0, 0, "write (l/4)*80 + (l%4)*16      [old A staging]"},
        {1, 1, 64, 0, 0, "write l*64"},
        {1, 1, 96, 0, 0, "write l*96"},
        {1, 1, 144, 0, 0, "write l*144"},
        {1, 2, 64, 0, 0, "write (l/4)*64 + (l%4)*16 = linear"},
    };
    unsigned *out; long long *cyc;
    const int blocks = 256, threads = 512;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    for (size_t i = 0; i < ps.size(); ++i) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, ps[i].kind, ps[i].id, ps[i].a, ps[i].b, ps[i].c, out, cyc);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        // one block per CU, 8 waves, ITERS * 8 accesses of 1 KB each per wave
        const double kb = 8.0 * ITERS * 8;
        printf("%2zu  %-58s %8.1f us   %6.1f B/ns per CU\n", i, ps[i].name, ms * 1e3, kb * 1024 / (ms * 1e6));
    }
    return 0;
}
