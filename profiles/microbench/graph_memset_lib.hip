// graph_memset.hip's question asked again UNDER TORCH: the round-3 node sequence (memset, kernel, two copies)
// captured once, replayed while the same process's torch tensors are filled, copied and freed through torch's caching
// allocator and copy paths -- the condition of the original fault that the stand-alone program did not have.  Every buffer is
// the front of a larger allocation with a canary tail, so a stray write shows up as a changed canary and cannot fault.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC graph_memset_lib.hip -o libgraph_memset.so ; python graph_memset_torch.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
__global__ void k_touch(u32* p, u64 n) { for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] += 1u; }
static unsigned char *d_counts, *d_other, *h_pin; static u64 g_n, g_slack = 1 << 20; static hipStream_t g_st; static hipGraph_t g_g; static hipGraphExec_t g_ge;
static std::vector<unsigned char> g_host;
static int g_mode = 7;
// mode: bit 0 = the memset is a graph node (else the kernel k_fill), bit 1 = the device-to-device copy is one, bit 2 = the device-to-host copy is one
__global__ void k_fill(u32* p, u64 n, u32 v) { for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = v; }
__global__ void k_copy(u32* d, const u32* s, u64 n) { for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) d[i] = s[i]; }
extern "C" int gm_setup_mode(u64 n_bytes, void* stream, int mode);
extern "C" int gm_setup(u64 n_bytes, void* stream /* NULL: an own stream */) { return gm_setup_mode(n_bytes, stream, 7); }
extern "C" int gm_setup_mode(u64 n_bytes, void* stream, int mode) {
    g_n = n_bytes; g_mode = mode;
    if (hipMalloc((void**)&d_counts, n_bytes + g_slack) || hipMalloc((void**)&d_other, n_bytes + g_slack) || hipHostMalloc((void**)&h_pin, n_bytes + g_slack, hipHostMallocDefault)) return 2;
    hipMemset(d_counts, 0xAB, n_bytes + g_slack); hipMemset(d_other, 0xCD, n_bytes + g_slack); memset(h_pin, 0xEF, n_bytes + g_slack);
    if (stream) g_st = (hipStream_t)stream; else if (hipStreamCreate(&g_st)) return 2;
    if (hipStreamBeginCapture(g_st, hipStreamCaptureModeThreadLocal)) return 3;
    if (mode & 1) hipMemsetAsync(d_counts, 0, n_bytes, g_st); else hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, g_st, (u32*)d_counts, n_bytes / 4, 0u);
    hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, g_st, (u32*)d_counts, n_bytes / 4);
    if (mode & 2) hipMemcpyAsync(d_other, d_counts, n_bytes, hipMemcpyDeviceToDevice, g_st); else hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, g_st, (u32*)d_other, (const u32*)d_counts, n_bytes / 4);
    if (mode & 4) hipMemcpyAsync(h_pin, d_other, n_bytes, hipMemcpyDeviceToHost, g_st);
    if (hipStreamEndCapture(g_st, &g_g)) return 4;
    if (hipGraphInstantiate(&g_ge, g_g, nullptr, nullptr, 0)) return 5;
    g_host.resize(n_bytes + g_slack);
    return 0;
}
extern "C" int gm_replay() { return hipGraphLaunch(g_ge, g_st) == hipSuccess && hipStreamSynchronize(g_st) == hipSuccess ? 0 : 1; }
// which buffer is wrong and how: per buffer (counts, other, pinned) the number of wrong bytes, the first wrong offset and the four bytes there
extern "C" int gm_detail(u64* out /* 3 x 3 */) {
    for (int b = 0; b < 3; b++) {
        const unsigned char* p = h_pin;
        if (b < 2) { if (hipMemcpy(g_host.data(), b == 0 ? d_counts : d_other, g_n + g_slack, hipMemcpyDeviceToHost)) return 1; p = g_host.data(); }
        u64 n = 0, first = ~0ull;
        for (u64 i = 0; i < g_n; i++) if (p[i] != (i % 4 == 0 ? 1 : 0)) { if (first == ~0ull) first = i; n++; }
        out[3 * b] = n; out[3 * b + 1] = first;
        out[3 * b + 2] = first == ~0ull ? 0 : ((u64)p[first & ~3ull] | ((u64)p[(first & ~3ull) + 1] << 8) | ((u64)p[(first & ~3ull) + 2] << 16) | ((u64)p[(first & ~3ull) + 3] << 24));
    }
    return 0;
}
// bytes wrong inside the ranges / canary bytes overwritten
extern "C" int gm_check(u64* in_bad, u64* out_bad) {
    *in_bad = *out_bad = 0;
    if (hipMemcpy(g_host.data(), d_counts, g_n + g_slack, hipMemcpyDeviceToHost)) return 1;
    for (u64 i = 0; i < g_n; i++) if (g_host[i] != (i % 4 == 0 ? 1 : 0)) (*in_bad)++;
    for (u64 i = g_n; i < g_n + g_slack; i++) if (g_host[i] != 0xAB) (*out_bad)++;
    if (hipMemcpy(g_host.data(), d_other, g_n + g_slack, hipMemcpyDeviceToHost)) return 1;
    for (u64 i = 0; i < g_n; i++) if (g_host[i] != (i % 4 == 0 ? 1 : 0)) (*in_bad)++;
    for (u64 i = g_n; i < g_n + g_slack; i++) if (g_host[i] != 0xCD) (*out_bad)++;
    if (g_mode & 4) for (u64 i = 0; i < g_n; i++) if (h_pin[i] != (i % 4 == 0 ? 1 : 0)) (*in_bad)++;
    for (u64 i = g_n; i < g_n + g_slack; i++) if (h_pin[i] != 0xEF) (*out_bad)++;
    return 0;
}
extern "C" void gm_teardown(int own_stream) {
    hipGraphExecDestroy(g_ge); hipGraphDestroy(g_g); if (own_stream) hipStreamDestroy(g_st);
    hipFree(d_counts); hipFree(d_other); hipHostFree(h_pin);
}
