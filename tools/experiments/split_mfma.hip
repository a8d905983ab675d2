// Numerics probe for split-precision MFMA (fp16x3 / bf16x3) against exact fp32 MFMA and an fp64 host reference.
// One wave computes D[32x32] = A[32xK] * B[Kx32].  Build: hipcc --offload-arch=gfx950 -O3 split_mfma.hip -o split_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short bf16_rn(float x) { unsigned u = __float_as_uint(x); u += 0x7FFF + ((u >> 16) & 1); return u >> 16; }
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// mode 0: fp32 mfma; 1: fp16x3 (a scaled by sa, b by sb); 2: bf16x3 ; 3: fp16x3 with separate correction accumulator
__global__ void probe(const float *A, const float *B, float *D, int K, int mode, float sa, float sb)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0}, acc2 = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], B[(k + h) * 32 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            h8 ah, al, bh, bl; s8 ah_, al_, bh_, bl_;
            for (int j = 0; j < 8; ++j) {
                const int kk = k + 8 * h + j;
                const float a = kk < K ? A[r * K + kk] * sa : 0.f, b = kk < K ? B[kk * 32 + r] * sb : 0.f;
                if (mode == 1 || mode == 3) {
                    ah[j] = (_Float16)a; al[j] = (_Float16)(a - (float)ah[j]);
                    bh[j] = (_Float16)b; bl[j] = (_Float16)(b - (float)bh[j]);
                } else {
                    unsigned short x = bf16_rn(a); ah_[j] = x; al_[j] = bf16_rn(a - bf16_f(x));
                    x = bf16_rn(b); bh_[j] = x; bl_[j] = bf16_rn(b - bf16_f(x));
                }
            }
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else if (mode == 3) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, bh_, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bl_, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah_, bh_, acc, 0, 0, 0);
            }
        }
    }
    const float inv = mode == 0 ? 1.f : 1.f / (sa * sb);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = (acc[i] + acc2[i]) * inv;
    }
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

int main()
{
    const int K = 848;
    const char *names[] = {"fp32 mfma", "fp16x3", "bf16x3", "fp16x3+corr-acc"};
    struct Case { const char *name; double wa, wb, sa, sb; } cases[] = {
        {"weights~U(0.05) acts~N(1)  scale a=1024,b=1", 0.05, 1.0, 1024.0, 1.0},
        {"weights~U(0.05) acts~N(1)  scale a=1,b=1 (lo underflows)", 0.05, 1.0, 1.0, 1.0},
        {"weights~U(0.05) acts~N(10) scale a=1024,b=16", 0.05, 10.0, 1024.0, 16.0},
        {"grads~N(2e-6) acts~N(1)    scale a=2^19,b=1", 2e-6, 1.0, 524288.0, 1.0},
        {"grads~N(2e-6) acts~N(1)    scale a=1,b=1", 2e-6, 1.0, 1.0, 1.0},
    };
    float *dA, *dB, *dD;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dD, 32 * 32 * 4);
    for (auto &c : cases) {
        srand(1);
        std::vector<float> A(32 * K), B(K * 32), D(32 * 32);
        for (auto &x : A) x = (float)(c.wa * (c.wa < 1e-3 ? nrand() : (2 * urand() - 1)));
        for (auto &x : B) x = (float)(c.wb * nrand());
        std::vector<double> ref(32 * 32), mag(32 * 32);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0, m = 0;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; m += fabs((double)A[i * K + k] * B[k * 32 + j]); }
            ref[i * 32 + j] = s; mag[i * 32 + j] = m;
        }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        printf("%s\n", c.name);
        for (int mode = 0; mode < 4; ++mode) {
            probe<<<1, 64>>>(dA, dB, dD, K, mode, (float)c.sa, (float)c.sb);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, rms = 0, worst_rel_max = 0, refmax = 0;
            for (int e = 0; e < 1024; ++e) refmax = fmax(refmax, fabs(ref[e]));
            for (int e = 0; e < 1024; ++e) {
                const double err = fabs(D[e] - ref[e]);
                worst = fmax(worst, err / mag[e]); rms += (err / mag[e]) * (err / mag[e]);
                worst_rel_max = fmax(worst_rel_max, err / refmax);
            }
            printf("   %-16s max err/sum|ab| %.3e   rms %.3e   max err/max|ref| %.3e\n", names[mode], worst, sqrt(rms / 1024), worst_rel_max);
        }
    }
    return 0;
}
