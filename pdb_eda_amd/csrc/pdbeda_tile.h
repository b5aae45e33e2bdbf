// pdbeda_tile.h -- whole-map specific kernels and launch helpers.
#pragma once
#include "pdbeda_kernels.h"

namespace pdbeda {

// Write up to two volume descriptors passed by value (no host staging buffer, no sync).
__global__ void k_set_vols(VolDesc *vols, VolDesc v0, VolDesc v1, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        vols[0] = v0;
        if (n > 1) vols[1] = v1;
    }
}

// Number of blobs whose first key is < key (device side twin of the host accessor).
__device__ inline uint32_t rank_below(const Job &job, int64_t key) {
    if (key <= 0) return 0u;
    if (key >= job.key_words * 64) return job.ctr->n_blobs;
    const int64_t kw = key >> 6;
    return job.chunk_prefix[kw / KEY_CHUNK] + job.key_rank[kw] + (uint32_t)popc64(job.key_bits[kw] & bits_below((int)(key & 63)));
}

// Dense labels of one plane of a whole-map job: wave per word, lane per voxel, coalesced
// 256-B int32 stores; label = blob index inside this plane's list, or -1.
__global__ void __launch_bounds__(256) k_labels_plane(Job job, int vol, int32_t *__restrict__ labels) {
    const int lane = lane_id();
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const VolDesc vd = job.vols[vol];
    const int32_t rank_offset = (int32_t)rank_below(job, vd.key_base);
    const int64_t words = (int64_t)vd.row_words * vd.dim[1] * vd.dim[2];
    for (int64_t lw = wave; lw < words; lw += n_waves) {
        const int64_t w = vd.word_base + lw;
        const uint64_t m = job.mask[w];
        const int wq = (int)(lw % vd.row_words);
        const int64_t row = lw / vd.row_words;
        const int c = wq * 64 + lane;
        int32_t lab = -1;
        if ((m >> lane) & 1ull) {
            const int st = run_start_of(m, lane);
            const uint32_t run = job.run_base[w] + (uint32_t)popc64(run_starts(m) & bits_below(st));
            lab = (int32_t)job.r_rank[job.parent[run]] - rank_offset;
        }
        if (c < vd.dim[0]) labels[row * vd.dim[0] + c] = lab;
    }
}

// Threshold stage of a whole-map job.
inline int tile_or_stream_threshold(hipStream_t st, const float *dens, const Geom *geom_dev, const Geom &g, Job &job, uint64_t *mask_pos,
                                    uint64_t *mask_neg, float cut_pos, float cut_neg, int row_words, int64_t words_per_plane) {
    (void)g;
    (void)job;
    int64_t waves = words_per_plane;
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_threshold, dim3((unsigned)blocks), dim3(256), 0, st, dens, geom_dev, mask_pos, mask_neg, cut_pos, cut_neg, row_words,
                       words_per_plane);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace pdbeda
