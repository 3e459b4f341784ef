// Micro-benchmark: what HBM rate does the element-major node pattern reach, as a function of occupancy, loads in flight and
// row stride?  One wave = one tree node of 64 instances; a node reads NR rows of 512 B and writes NW rows, the rows coming
// from ~20 segments (row = off[r] + node * mul[r]).  Layout 0: element-major (row stride = Bp * 8 bytes, tile column offset
// 512 B);  layout 1: tile-major (row stride 512 B, tile stride = rows * 512 B).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o gpurun_out/stream_pattern && gpurun_out/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NR = 96, NW = 80;
struct Tab { int roff[NR], rmul[NR], woff[NW], wmul[NW]; };
__constant__ Tab g_tab;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0)); }
__device__ __forceinline__ void bstore(double x, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, v, s, 0); }

// CH rows in flight per wave; WPB waves per block; nodes per block = 2 * WPB
template <int CH, int WPB>
__global__ void __launch_bounds__(WPB * 64) pattern_kernel(const double *in, double *out, long tile_stride_in, long tile_stride_out,
                                                            unsigned row_stride8, int nodes, int nodes_per_blk) {
    const int tile = blockIdx.y, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (long)tile * tile_stride_in), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (long)tile * tile_stride_out), 0, 0xffffffff, 0x00020000);
    const unsigned voff = lane * 8;
    for (int n = wave; n < nodes_per_blk; n += WPB) {
        const int node = blockIdx.x * nodes_per_blk + n;
        if (node >= nodes) break;
        double acc = 0.;
#pragma unroll
        for (int c = 0; c < NR; c += CH) {
            double t[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) t[k] = bload(ri, voff, (unsigned)(g_tab.roff[c + k] + node * g_tab.rmul[c + k]) * row_stride8);
#pragma unroll
            for (int k = 0; k < CH; ++k) acc += t[k];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) bstore(acc + k, ro, voff, (unsigned)(g_tab.woff[k] + node * g_tab.wmul[k]) * row_stride8);
    }
}

template <int CH, int WPB>
static void run(const char *name, const double *in, double *out, int layout, int bp, int nodes, long rrows, long wrows, int lds_pad, int npw = 2) {
    const int tiles = bp / 64, npb = npw * WPB;
    long ts_in = layout ? rrows * 64 : 64, ts_out = layout ? wrows * 64 : 64;
    unsigned rs8 = layout ? 512u : (unsigned)bp * 8u;
    dim3 grid((nodes + npb - 1) / npb, tiles), blk(WPB * 64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((pattern_kernel<CH, WPB>), grid, blk, lds_pad, 0, in, out, ts_in, ts_out, rs8, nodes, npb);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double bytes = (double)nodes * tiles * (NR + NW) * 512.;
    printf("%-10s layout=%d CH=%2d WPB=%2d npw=%2d lds=%6d : %7.3f ms  %7.1f GB/s\n", name, layout, CH, WPB, npw, lds_pad, ms, bytes / ms * 1e-6);
}

int main() {
    const int bp = 2048, nodes = 2048;
    // segments: length per node; rows of a node inside a segment are consecutive
    int rl[] = {5, 2, 5, 1, 2, 10, 2, 5, 5, 1, 7, 10, 4, 2, 2, 10, 2, 2, 10, 2, 7};   // sum 96
    int wl[] = {5, 1, 7, 10, 4, 2, 2, 10, 2, 2, 10, 5, 2, 10, 5, 2, 1};              // sum 80
    Tab t; long rrows = 0, wrows = 0; int r = 0, w = 0;
    for (int s = 0; s < (int)(sizeof(rl) / sizeof(int)); ++s) { for (int k = 0; k < rl[s]; ++k) { t.roff[r] = (int)rrows + k; t.rmul[r] = rl[s]; ++r; } rrows += (long)rl[s] * nodes; }
    for (int s = 0; s < (int)(sizeof(wl) / sizeof(int)); ++s) { for (int k = 0; k < wl[s]; ++k) { t.woff[w] = (int)wrows + k; t.wmul[w] = wl[s]; ++w; } wrows += (long)wl[s] * nodes; }
    if (r != NR || w != NW) { printf("table mismatch %d %d\n", r, w); return 1; }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tab), &t, sizeof(t)));
    double *in, *out;
    CK(hipMalloc(&in, rrows * bp * 8)); CK(hipMalloc(&out, wrows * bp * 8));
    CK(hipMemset(in, 0, rrows * bp * 8)); CK(hipMemset(out, 0, wrows * bp * 8));
    printf("read %.2f GB write %.2f GB per launch\n", rrows * bp * 8e-9, wrows * bp * 8e-9);
    for (int layout = 0; layout < 2; ++layout) {
        // occupancy through LDS padding per 4-wave block: 80 KB -> 2 waves/SIMD; 40 KB -> 4; 20 KB -> 8
        int pads[] = {80 * 1024 - 512, 40 * 1024 - 256, 20 * 1024 - 128};
        for (int p = 0; p < 3; ++p)
            for (int npw : {2, 8, 32}) {
                run<8, 4>("ch8", in, out, layout, bp, nodes, rrows, wrows, pads[p], npw);
                run<32, 4>("ch32", in, out, layout, bp, nodes, rrows, wrows, pads[p], npw);
            }
        // one wave per block (no co-scheduling constraints), 8 waves per block
        for (int npw : {2, 8, 32}) {
            run<16, 1>("w1 2/simd", in, out, layout, bp, nodes, rrows, wrows, 20 * 1024 - 128, npw);
            run<16, 1>("w1 4/simd", in, out, layout, bp, nodes, rrows, wrows, 10 * 1024 - 64, npw);
            run<16, 8>("w8 2/simd", in, out, layout, bp, nodes, rrows, wrows, 160 * 1024 - 1024, npw);
            run<16, 8>("w8 4/simd", in, out, layout, bp, nodes, rrows, wrows, 80 * 1024 - 512, npw);
        }
    }
    return 0;
}
