// Read-bandwidth of the k_tile_label A1 access pattern (256^3 grid, tiles of 256 x 8 x 8, a wave per section, 4-B loads)
// against a linear stream and variants.
#include <hip/hip_runtime.h>
#include <cstdio>
// mode 0: tile pattern, dword loads, 16 in flight, 2 chunks (as k_tile_label)
// mode 1: tile pattern, all 32 loads in flight
// mode 2: tile pattern, dwordx4 loads (8 per wave: 8 rows of 1 KB)
// mode 3: linear: block b reads 64 KB contiguous, dwordx4
template <int MODE>
__global__ void __launch_bounds__(512, 8) k_read(const float *__restrict__ g, float *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long rt = blockIdx.x % 32, st = blockIdx.x / 32;
    float acc = 0.f;
    if (MODE == 0 || MODE == 1) {
        const float *base = g + ((st * 8 + wv) * 256 + rt * 8) * 256 + lane;   // section st*8+wv, rows rt*8.., 1 KB rows
        if (MODE == 0) {
#pragma unroll 1
            for (int chunk = 0; chunk < 2; ++chunk) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = base[(chunk * 4 + j / 4) * 256 + (j % 4) * 64];
#pragma unroll
                for (int j = 0; j < 16; ++j) acc += v[j];
            }
        } else {
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = base[(j / 4) * 256 + (j % 4) * 64];
#pragma unroll
            for (int j = 0; j < 32; ++j) acc += v[j];
        }
    } else if (MODE == 2) {
        const float4 *base = reinterpret_cast<const float4 *>(g + ((st * 8 + wv) * 256 + rt * 8) * 256) + lane;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base[j * 64];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    } else {
        const float4 *base = reinterpret_cast<const float4 *>(g) + (long)blockIdx.x * 4096 + threadIdx.x;
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base[j * 512];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    if (acc == 1.2345e30f) out[0] = acc;
}
int main() {
    const long n = 256l * 256 * 256;
    float *d, *o; (void)hipMalloc(&d, n * 4); (void)hipMalloc(&o, 4); (void)hipMemset(d, 0, n * 4);
    float *scratch; (void)hipMalloc(&scratch, 512l << 20);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    auto time = [&](const char *name, auto launch, bool flush) {
        float total = 0;
        for (int i = 0; i < 12; ++i) {
            if (flush) (void)hipMemsetAsync(scratch, i, 512l << 20, 0);   // push the grid out of L2 / Infinity cache
            (void)hipEventRecord(a, 0);
            launch();
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (i >= 2) total += ms;
        }
        printf("%-44s %7.2f us  %6.2f TB/s\n", name, total * 100.0, n * 4 / (total / 10 * 1e-3) / 1e12);
    };
    for (int flush = 0; flush < 2; ++flush) {
        printf(flush ? "-- cold (512 MiB written between launches)\n" : "-- warm (back to back)\n");
        time("tile pattern, dword x16 x2 chunks", [&] { hipLaunchKernelGGL(k_read<0>, dim3(1024), dim3(512), 0, 0, d, o); }, flush);
        time("tile pattern, dword x32 in flight", [&] { hipLaunchKernelGGL(k_read<1>, dim3(1024), dim3(512), 0, 0, d, o); }, flush);
        time("tile pattern, dwordx4 x8", [&] { hipLaunchKernelGGL(k_read<2>, dim3(1024), dim3(512), 0, 0, d, o); }, flush);
        time("linear 64 KB per block, dwordx4 x8", [&] { hipLaunchKernelGGL(k_read<3>, dim3(1024), dim3(512), 0, 0, d, o); }, flush);
    }
    return 0;
}
