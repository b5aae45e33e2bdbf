// Write-bandwidth ceiling: 64 MiB of int4 zeros with several launch shapes (plain and nontemporal stores).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ inline void nt_store(int4 *p) { v4i z = {0, 0, 0, 0}; __builtin_nontemporal_store(z, reinterpret_cast<v4i *>(p)); }
template <bool NT>
__global__ void __launch_bounds__(512) k_fill(int4 *p, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        if (NT) nt_store(p + i); else p[i] = make_int4(0, 0, 0, 0);
    }
}
// tile-shaped: block b writes 8 chunks of 8 KB at stride 256 KB (like k_labels_tiles on a 256-wide grid)
template <bool NT>
__global__ void __launch_bounds__(512) k_tiles(int4 *p) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long rt = blockIdx.x % 32, st = blockIdx.x / 32;
    for (int rr = 0; rr < 8; ++rr) {
        const long row = (st * 8 + wv) * 256 + rt * 8 + rr;   // section-major rows of 1 KB
        int4 *dst = p + row * 64 + lane;
        if (NT) nt_store(dst); else *dst = make_int4(0, 0, 0, 0);
    }
}
int main() {
    const long bytes = 64l << 20, n4 = bytes / 16;
    int4 *d; (void)hipMalloc(&d, bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < 20; ++i) launch();
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-28s %7.2f us  %6.2f TB/s\n", name, ms * 50.0, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    for (int g : {256, 512, 1024, 2048, 4096}) {
        char nm[64]; snprintf(nm, 64, "fill plain grid %d", g);
        time(nm, [&] { hipLaunchKernelGGL(k_fill<false>, dim3(g), dim3(512), 0, 0, d, n4); });
        snprintf(nm, 64, "fill nt    grid %d", g);
        time(nm, [&] { hipLaunchKernelGGL(k_fill<true>, dim3(g), dim3(512), 0, 0, d, n4); });
    }
    time("tiles plain 1024x512", [&] { hipLaunchKernelGGL(k_tiles<false>, dim3(1024), dim3(512), 0, 0, d); });
    time("tiles nt    1024x512", [&] { hipLaunchKernelGGL(k_tiles<true>, dim3(1024), dim3(512), 0, 0, d); });
    time("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 0, bytes, 0); });
    return 0;
}
