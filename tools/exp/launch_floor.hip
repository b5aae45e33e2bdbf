// What a dependent launch costs by itself: back-to-back launches of (a) an empty kernel, (b) a kernel whose threads make one
// global load and exit, (c) two dependent loads -- for several grid shapes.  hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(const int *p, int *q) {}
__global__ void k_one(const int *p, int *q) { if (p[blockIdx.x * blockDim.x + threadIdx.x] == 12345) q[0] = 1; }
__global__ void k_two(const int *p, int *q) { const int a = p[blockIdx.x * blockDim.x + threadIdx.x]; if (p[a & 1023] == 12345) q[0] = 1; }
int main() {
    int *p, *q;
    hipMalloc(&p, 4 << 20); hipMemset(p, 0, 4 << 20); hipMalloc(&q, 64);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grids[] = {64, 256, 544, 1088, 2176}, blocks[] = {256, 512, 1024};
    for (int kind = 0; kind < 3; ++kind)
        for (int g : grids)
            for (int bl : blocks) {
                if ((long)g * bl > (1 << 20)) continue;
                auto launch = [&]() {
                    if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(g), dim3(bl), 0, s, p, q);
                    else if (kind == 1) hipLaunchKernelGGL(k_one, dim3(g), dim3(bl), 0, s, p, q);
                    else hipLaunchKernelGGL(k_two, dim3(g), dim3(bl), 0, s, p, q);
                };
                for (int i = 0; i < 20; ++i) launch();
                hipStreamSynchronize(s);
                hipEventRecord(a, s);
                for (int i = 0; i < 200; ++i) launch();
                hipEventRecord(b, s);
                hipStreamSynchronize(s);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("%s grid %5d x %4d: %.2f us per dependent launch\n", kind == 0 ? "empty   " : kind == 1 ? "one load" : "two loads", g, bl, 1e3 * ms / 200);
            }
    return 0;
}
