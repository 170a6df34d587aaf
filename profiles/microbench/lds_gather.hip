// LDS gather rate on gfx950: cycles per wave-level ds_read of W bytes per lane at random / sequential / broadcast slots,
// with 20 one-wave workgroups per CU (k_extend's composition phase).  hipcc --offload-arch=gfx950 -O3 lds_gather.hip -o lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
template <int MODE, int WIDTH>   // MODE 0 random, 1 lane-sequential, 2 broadcast, 3 random with 16 distinct slots among the wave
__global__ __launch_bounds__(64) void k(const u32* __restrict__ idx, int iters, int nslots, u32* out) {
    extern __shared__ int4 s[];
    for (int i = threadIdx.x; i < nslots; i += 64) s[i] = make_int4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    u32 acc = 0; const int lane = threadIdx.x;
    u32 r = idx[lane];
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int u = 0; u < 6; u++) {
            r = r * 1664525u + 1013904223u;
            u32 a = MODE == 0 ? (r >> 8) % (u32)nslots : MODE == 1 ? (u32)((lane + u * 64 + it) % nslots) : MODE == 2 ? (u32)((it + u) % nslots) : ((r >> 8) % 16u) * 7u % (u32)nslots;
            if (WIDTH == 16) { int4 v = s[a]; acc += v.x ^ v.w; }
            else if (WIDTH == 8) { int2 v = reinterpret_cast<int2*>(s)[a]; acc += v.x ^ v.y; }
            else { int v = reinterpret_cast<int*>(s)[a]; acc += v; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE, int WIDTH> static void run(const char* name, const u32* d_idx, u32* d_out) {
    const int iters = 2000, nslots = 400, blocks = 256 * 20;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, WIDTH>), dim3(blocks), dim3(64), nslots * 16, 0, d_idx, 10, nslots, d_out);
    hipEventRecord(a); hipLaunchKernelGGL((k<MODE, WIDTH>), dim3(blocks), dim3(64), nslots * 16, 0, d_idx, iters, nslots, d_out); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
    const double reads_per_cu = (double)blocks / 256 * iters * 6;
    printf("%-28s %8.3f ms  %6.1f ns per wave-read per CU (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / reads_per_cu, ms * 1e6 / reads_per_cu * 2.4);
}
int main() {
    std::vector<u32> h(64); for (int i = 0; i < 64; i++) h[i] = 12345u * (i + 1) + 7u;
    u32 *d_idx, *d_out; hipMalloc(&d_idx, 256); hipMalloc(&d_out, 4); hipMemcpy(d_idx, h.data(), 256, hipMemcpyHostToDevice);
    run<0, 16>("b128 random (400 slots)", d_idx, d_out); run<1, 16>("b128 lane-sequential", d_idx, d_out); run<2, 16>("b128 broadcast", d_idx, d_out); run<3, 16>("b128 random, 16 distinct", d_idx, d_out);
    run<0, 8>("b64 random", d_idx, d_out); run<1, 8>("b64 lane-sequential", d_idx, d_out);
    run<0, 4>("b32 random", d_idx, d_out); run<1, 4>("b32 lane-sequential", d_idx, d_out);
    return 0;
}
