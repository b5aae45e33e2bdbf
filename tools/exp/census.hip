// Census: where do the 1024 x 512-thread workgroups of a k_tile_label-shaped launch land (XCC, SE, CU, SIMD, wave slot)?
// hipcc --offload-arch=gfx950 -O3 tools/exp/census.hip -o /tmp/census && /tmp/census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(512, 8) k(unsigned *out, unsigned long long *t) {
    __shared__ float pad[8192];   // 32 KB like the tile kernel
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + wv) * 2] = hwid; out[(blockIdx.x * 8 + wv) * 2 + 1] = xcc; }
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    // stay resident for a while so that all 1024 blocks are co-resident
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(10);
    if (pad[(threadIdx.x * 7) & 8191] < 0) out[0] = 0;
}
int main() {
    const int nb = 1024;
    unsigned *d; unsigned long long *dt;
    hipMalloc(&d, nb * 8 * 2 * 4); hipMalloc(&dt, nb * 8);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d, dt); hipDeviceSynchronize(); }
    std::vector<unsigned> h(nb * 16); std::vector<unsigned long long> ht(nb);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    int slot_hist[16] = {0};
    for (int b = 0; b < nb; ++b) {
        unsigned minslot = 99, cu = 0;
        for (int w = 0; w < 8; ++w) {
            const unsigned id = h[(b * 8 + w) * 2], xcc = h[(b * 8 + w) * 2 + 1];
            const unsigned wave = id & 15, simd = (id >> 4) & 3, cuid = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            cu = (xcc << 8) | (se << 5) | (sh << 4) | cuid;
            if (wave < minslot) minslot = wave;
            slot_hist[wave]++;
            if (b < 4) printf("block %d wave %d: xcc %u se %u sh %u cu %u simd %u slot %u\n", b, w, xcc, se, sh, cuid, simd, wave);
        }
        by_cu[cu].push_back(b * 100 + minslot);
    }
    printf("distinct CUs %zu\n", by_cu.size());
    int shown = 0;
    for (auto &kv : by_cu) { if (shown++ < 6) { printf("cu %03x:", kv.first); for (int v : kv.second) printf(" blk %d(minslot %d, t %llu)", v / 100, v % 100, ht[v / 100] - ht[0]); printf("\n"); } }
    printf("slot histogram:"); for (int i = 0; i < 16; ++i) printf(" %d", slot_hist[i]); printf("\n");
    std::map<int, int> per; for (auto &kv : by_cu) per[(int)kv.second.size()]++;
    for (auto &kv : per) printf("%d CUs hold %d blocks\n", kv.second, kv.first);
    return 0;
}
