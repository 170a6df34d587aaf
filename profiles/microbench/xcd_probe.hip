// Random 4-byte probes into a table: every workgroup over the whole table, or every workgroup only into the slice of
// its own XCD (HW_REG_XCC_ID).  Question: how much faster are probes that stay in one XCD's L2?
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_probe xcd_probe.hip ; run: ./xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xF; }
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>     // 0 = whole table, 1 = the XCD's slice
__global__ __launch_bounds__(256) void probe(const uint32_t* __restrict__ tab, uint32_t slice_words, uint32_t n_slices, uint32_t iters, uint32_t* out, uint32_t* xcd_seen) {
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&xcd_seen[x], 1u);
    uint32_t seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const uint32_t total = slice_words * n_slices;
    for (uint32_t it = 0; it < iters; it += 8) {
        uint32_t v[8];
        #pragma unroll
        for (int k = 0; k < 8; k++) {
            seed = mix(seed + k);
            uint32_t idx = MODE ? (x % n_slices) * slice_words + (seed % slice_words) : seed % total;
            v[k] = tab[idx];
        }
        #pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const uint32_t n_slices = 8;
    for (uint32_t slice_mb = 1; slice_mb <= 4; slice_mb *= 2) {
        const uint32_t slice_words = slice_mb * 1024 * 1024 / 4;
        uint32_t *tab, *out, *seen; hipMalloc(&tab, (size_t)slice_words * n_slices * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&seen, 64);
        hipMemset(tab, 1, (size_t)slice_words * n_slices * 4); hipMemset(seen, 0, 64);
        const uint32_t iters = 2048, blocks = 4096;
        for (int mode = 0; mode < 2; mode++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (mode) probe<1><<<blocks, 256>>>(tab, slice_words, n_slices, iters, out, seen); else probe<0><<<blocks, 256>>>(tab, slice_words, n_slices, iters, out, seen);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode) probe<1><<<blocks, 256>>>(tab, slice_words, n_slices, iters, out, seen); else probe<0><<<blocks, 256>>>(tab, slice_words, n_slices, iters, out, seen);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double probes = (double)blocks * 256 * iters;
            printf("table %2u MiB (8 slices of %u MiB)  %-12s %.3f ms  %.1f G probes/s\n", slice_mb * 8, slice_mb, mode ? "own slice" : "whole table", ms, probes / ms / 1e6);
        }
        uint32_t h[16]; hipMemcpy(h, seen, 64, hipMemcpyDeviceToHost);
        if (slice_mb == 1) { printf("workgroups per XCC id:"); for (int i = 0; i < 8; i++) printf(" %u", h[i]); printf("\n"); }
        hipFree(tab); hipFree(out); hipFree(seen);
    }
    return 0;
}
