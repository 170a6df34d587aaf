// VALU issue-rate probe #2 for gfx950: ns per wave64 instruction per SIMD at 4 waves/SIMD for the integer ops the
// sieve is built from.  build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip ; run: ./valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2000
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
// each probe: 8 independent instructions per asm block, 8 blocks per iteration
#define PROBE(NAME, T) \
__global__ void NAME(uint32_t* out, uint32_t seed) { \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
    uint32_t k = seed | 1, m = seed * 77; uint64_t q = seed * 1234567ull + threadIdx.x; \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int r = 0; r < 8; r++) asm volatile(R8(T) : OPS, "+v"(q) : "v"(k), "v"(m) : "vcc", "s10", "s11"); } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)q; }
#define T_ADD(i)   "v_add_u32 %" #i ", %" #i ", %9\n"
#define T_XOR(i)   "v_xor_b32 %" #i ", %" #i ", %9\n"
#define T_AND(i)   "v_and_b32 %" #i ", %" #i ", %9\n"
#define T_SHR(i)   "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define T_SHL(i)   "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define T_NOT(i)   "v_not_b32 %" #i ", %" #i "\n"
#define T_MOV(i)   "v_mov_b32 %" #i ", %9\n"
#define T_MAX(i)   "v_max_u32 %" #i ", %" #i ", %9\n"
#define T_SUB(i)   "v_sub_u32 %" #i ", %" #i ", %9\n"
#define T_CND(i)   "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T_CMP(i)   "v_cmp_lt_u32 vcc, %" #i ", %9\n"
#define T_CMP64(i) "v_cmp_lt_u64 vcc, %8, %8\n"
#define T_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %9, %10 bitop3:0x6c\n"
#define T_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define T_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %9\n"
#define T_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %9, %10\n"
#define T_XAD(i)   "v_xad_u32 %" #i ", %" #i ", %9, %10\n"
#define T_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %9, %10\n"
#define T_OR3(i)   "v_or3_b32 %" #i ", %" #i ", %9, %10\n"
#define T_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %9, %10\n"
#define T_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", %9, 1\n"
#define T_PKMIN(i) "v_pk_min_u16 %" #i ", %" #i ", %9\n"
#define T_PKSUB(i) "v_pk_sub_u16 %" #i ", %" #i ", %9\n"
#define T_MUL(i)   "v_mul_lo_u32 %" #i ", %" #i ", %9\n"
#define T_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %9\n"
#define T_BFREV(i) "v_bfrev_b32 %" #i ", %" #i "\n"
#define T_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %9, 24\n"
#define T_SHR64(i) "v_lshrrev_b64 %8, 3, %8\n"
#define T_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %9, %" #i "\n"
#define T_SDWA(i)  "v_and_b32_sdwa %" #i ", %" #i ", %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define T_MIN3(i)  "v_min3_u32 %" #i ", %" #i ", %9, %10\n"
#define T_CMPEQ16(i) "v_cmp_eq_u16 vcc, %" #i ", %9\n"
#define T_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_ADDLIT(i) "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define T_MULLIT(i) "v_mul_lo_u32 %" #i ", %" #i ", 0x9E3779B1\n"
#define T_CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define T_CNDMIX(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n v_add_u32 %" #i ", %" #i ", %9\n v_xor_b32 %" #i ", %" #i ", %9\n v_and_b32 %" #i ", %" #i ", %10\n"
#define T_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %9\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define T_CMPCND64(i) "v_cmp_lt_u32_e64 s[10:11], %" #i ", %9\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define T_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %9, vcc\n"
#define T_CNDK(i) "v_cndmask_b32 %" #i ", 0, %9, vcc\n"
#define T_CND2(i) "v_cndmask_b32 %" #i ", %10, %9, vcc\n"
#define LIST(X) X(p_cnd64, T_CND64) X(p_cndmix4, T_CNDMIX) X(p_cmpcnd2, T_CMPCND) X(p_cmpcnd64_2, T_CMPCND64) X(p_addc, T_ADDC) X(p_cndk, T_CNDK) X(p_cnd2, T_CND2) X(p_add, T_ADD) X(p_xor, T_XOR) X(p_and, T_AND) X(p_shr, T_SHR) X(p_shl, T_SHL) X(p_not, T_NOT) X(p_mov, T_MOV) X(p_max, T_MAX) X(p_sub, T_SUB) \
    X(p_cnd, T_CND) X(p_cmp, T_CMP) X(p_cmp64, T_CMP64) X(p_bitop3, T_BITOP3) X(p_lshlor, T_LSHLOR) X(p_lshladd, T_LSHLADD) X(p_andor, T_ANDOR) X(p_xad, T_XAD) \
    X(p_add3, T_ADD3) X(p_or3, T_OR3) X(p_perm, T_PERM) X(p_bfe, T_BFE) X(p_pkmin, T_PKMIN) X(p_pksub, T_PKSUB) X(p_mul, T_MUL) X(p_mulhi, T_MULHI) X(p_bfrev, T_BFREV) \
    X(p_align, T_ALIGN) X(p_shr64, T_SHR64) X(p_mbcnt, T_MBCNT) X(p_sdwa, T_SDWA) X(p_min3, T_MIN3) X(p_cmpeq16, T_CMPEQ16) X(p_adddpp, T_ADDDPP) X(p_addlit, T_ADDLIT)
#define DEF(NAME, T) PROBE(NAME, T)
LIST(DEF)
typedef void (*kern_t)(uint32_t*, uint32_t);
static void run(const char* name, kern_t kfn, uint32_t* d) {
    const int wps = 4; int threads = 1024, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kfn<<<blocks, threads>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(e0); kfn<<<blocks, threads>>>(d, 1); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)ITER * 64 * wps;
    printf("%-10s %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / instr_per_simd);
}
int main() {
    uint32_t* d; hipMalloc(&d, (1 << 22) + 64);
#define RUN(NAME, T) run(#NAME, NAME, d);
    LIST(RUN)
    return 0;
}
