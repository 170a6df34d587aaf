// VALU issue-rate probe for gfx950: cycles per wave64 instruction for a few integer ops, at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITER 2000
template <int OP>
__global__ void probe(uint32_t* out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    uint32_t k = seed | 1;
    long long t0 = clock64();
    for (int it = 0; it < ITER; it++) {
        #pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (OP == 0) { asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 2) { asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 3) { asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3\n v_bfrev_b32 %4, %4\n v_bfrev_b32 %5, %5\n v_bfrev_b32 %6, %6\n v_bfrev_b32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 4) { asm volatile("v_alignbit_b32 %0, %0, %8, 7\n v_alignbit_b32 %1, %1, %8, 7\n v_alignbit_b32 %2, %2, %8, 7\n v_alignbit_b32 %3, %3, %8, 7\n v_alignbit_b32 %4, %4, %8, 7\n v_alignbit_b32 %5, %5, %8, 7\n v_alignbit_b32 %6, %6, %8, 7\n v_alignbit_b32 %7, %7, %8, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 5) { asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 6) { asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
            if (OP == 7) { asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int OP> void run(const char* name, uint32_t* d) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        int threads = 64 * 4 * wps;                     // waves per CU = 4 SIMDs * wps
        if (threads > 1024) threads = 1024;
        int blocks = 256 * (64 * 4 * wps / threads);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        probe<OP><<<blocks, threads>>>(d, 1); hipDeviceSynchronize();
        hipEventRecord(e0); probe<OP><<<blocks, threads>>>(d, 1); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)ITER * REP * wps;
        printf("%-16s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f clk @2.4GHz)\n", name, wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, (1 << 22) + 64);
    run<0>("v_add_u32", d); run<5>("v_add dependent", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_u32_u24", d); run<7>("v_mad_u32_u24", d);
    run<3>("v_bfrev_b32", d); run<4>("v_alignbit_b32", d); run<6>("v_bcnt_u32_b32", d);
    return 0;
}
