// Micro-benchmark 2: what a narrow matrix-core sweep level (nx = 64, nu = 16, d = 2, <= 16 nodes) spends its time on.
// Variants of bwd_level_mfma: FR 0 fragments from global memory every level | 1 fetched once; PAD: LDS stride of the node vectors
// nx + PAD doubles (bank conflicts of the B-operand reads / the epilogue); NOLDS 1: B operand from a register (no LDS reads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) double *MatPtr;

template <int NX, int NU, int FR, int PAD, int NOLDS, int EPI>
__global__ void __launch_bounds__(1024) k(const double *mats, int cnt, int reps, int nlev, int sw_stride, double *out) {
    extern __shared__ double smem[];
    constexpr int nx = NX, nu = NU, d = 2, KQ = NX / 4, SX = NX + PAD, SU = NU + PAD;
    const int nl = 2 * cnt + 1;
    double *XQ = smem, *ED = XQ + nl * SX;
    for (int i = threadIdx.x; i < nl * (SX + SU); i += blockDim.x) smem[i] = 1e-3 * (i % 17);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    constexpr int mq = (nx + 15) / 16, mt = mq + (nu + 15) / 16;
    const int ntile = (cnt + 15) / 16;
    int nsplit = nwaves / mt; if (nsplit > ntile) nsplit = ntile; if (nsplit < 1) nsplit = 1;
    double af[2][KQ];
    for (int r = 0; r < reps; ++r)
        for (int l = 0; l < nlev; ++l) {
            const MatPtr SW = (MatPtr)mats + (long)l * sw_stride;
            for (int it = wave; it < mt * nsplit; it += nwaves) {
                const int m = it % mt, sp = it / mt;
                const bool isq = m < mq;
                const int row0 = isq ? m * 16 : (m - mq) * 16, ld = isq ? nx : nu;
                if (FR == 0 || (r == 0 && l == 0)) {
#pragma unroll
                    for (int c = 0; c < d; ++c)
#pragma unroll
                        for (int q = 0; q < KQ; ++q) af[c][q] = (isq ? SW + c * nx * nx : SW + 2 * nx * nx + c * nu * nx)[row0 + j + (long)(4 * q + kq) * ld];
                }
                for (int nt = sp; nt < ntile; nt += nsplit) {
                    const int pos = nt * 16 + j;
                    const bool valid = pos < cnt;
                    const int li = valid ? pos : 0;
                    v4d acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        const double *q0 = XQ + (long)(li * d + 1) * SX, *q1 = XQ + (long)(li * d + 2) * SX;
                        const double b0 = NOLDS ? (double)q : (valid ? q0[4 * q + kq] : 0.), b1 = NOLDS ? (double)(q + 1) : (valid ? q1[4 * q + kq] : 0.);
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][q], b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[1][q], b1, acc1, 0, 0, 0);
                    }
                    acc0 += acc1;
                    if (valid && EPI) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int rr = row0 + kq + 4 * reg;
                            if (isq) XQ[(long)li * SX + rr] += acc0[reg];
                            else ED[(long)li * SU + rr] -= acc0[reg];
                        }
                    } else if (valid && acc0[0] == 12345.678) XQ[0] = acc0[1];
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) out[0] = XQ[0];
}

template <int FR, int PAD, int NOLDS, int EPI>
void run(const char *name, const double *d, double *o, int sw_stride) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int cnt : {1, 16, 64}) {
        const size_t lds = (size_t)(2 * cnt + 1) * (64 + 16 + 2 * PAD) * 8 + 1024;
        auto fn = k<64, 16, FR, PAD, NOLDS, EPI>;
        CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const int reps = 200, nlev = 7;
        hipLaunchKernelGGL(fn, dim3(1), dim3(1024), lds, 0, d, cnt, 2, nlev, sw_stride, o);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(1), dim3(1024), lds, 0, d, cnt, reps, nlev, sw_stride, o);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s cnt %2d: %.2f us per level\n", name, cnt, ms * 1e3 / (reps * nlev));
    }
}
int main() {
    const int NX = 64, NU = 16, sw_stride = 2 * NX * NX + 2 * NU * NX + 2 * NX * NX + NU * NX;
    std::vector<double> h((size_t)14 * sw_stride);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * ((i * 7) % 13) - 5e-3;
    double *d, *o; CK(hipMalloc(&d, h.size() * 8)); CK(hipMalloc(&o, 64));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    run<0, 0, 0, 1>("as shipped (2 chains)", d, o, sw_stride);
    run<1, 0, 0, 1>("fragments fetched once", d, o, sw_stride);
    run<1, 0, 1, 1>("fragments once, no LDS operand reads", d, o, sw_stride);
    run<1, 0, 1, 0>("fragments once, no LDS reads, no epilogue", d, o, sw_stride);
    run<0, 4, 0, 1>("padded stride nx+4", d, o, sw_stride);
    run<1, 4, 0, 1>("fragments once, padded stride nx+4", d, o, sw_stride);
    run<0, 1, 0, 1>("padded stride nx+1", d, o, sw_stride);
    return 0;
}
