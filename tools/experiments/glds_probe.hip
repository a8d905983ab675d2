// LDS-DMA into LDS addresses above 64 KB through M0 (the form wg_gemm16g.h uses): every wave copies 1 KB to smem[off + wave * 1024] for
// several offsets up to 136 KB and the workgroup checks the bytes.  Prints OK / the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(sbase) : "memory");
}
__global__ __launch_bounds__(512) void k(const unsigned *src, unsigned *bad, int off)
{
    __shared__ __attribute__((aligned(1024))) char smem[139264];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(lds_char *)smem;
    for (int i = threadIdx.x; i < 139264 / 4; i += 512) reinterpret_cast<unsigned *>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    glds16(src + wave * 256, lane * 16, lds0 + off + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int i = threadIdx.x; i < 139264 / 4; i += 512) {
        const unsigned v = reinterpret_cast<unsigned *>(smem)[i];
        const int rel = i - off / 4;
        const unsigned want = (rel >= 0 && rel < 2048) ? src[rel] : 0xdeadbeefu;
        if (v != want) atomicAdd(bad, 1u);
    }
}
int main()
{
    std::vector<unsigned> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = 0x1000000u + i * 7919u;
    unsigned *d, *bad;
    hipMalloc(&d, 8192); hipMalloc(&bad, 4);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    int fails = 0;
    for (int off : {0, 1024, 60 * 1024, 64 * 1024, 65 * 1024, 100 * 1024, 128 * 1024}) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, bad, off);
        unsigned b = 0;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("glds to LDS offset %6d: %s (%u mismatching words)\n", off, b ? "WRONG" : "OK", b);
        fails += b != 0;
    }
    return fails ? 1 : 0;
}
