// group_kernels_common.h — the curve-independent kernels of the in-step key grouping, shared by the P-256 and
// the Ed25519 pipelines (each translation unit gets its own static copy; no relocatable device code needed).
#pragma once
#include <hip/hip_runtime.h>

#include "p256_group.h"

namespace sbv {

__device__ __forceinline__ u32 group_count(const GroupState& g) {
    const u32 c = g.counters[0];
    return c < g.max_groups ? c : g.max_groups;
}

static __global__ __launch_bounds__(256) void k_group_assign(size_t n, GroupState g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) group_assign_lane(i, g);
}

static __global__ __launch_bounds__(256) void k_pack_bitmap(const uint8_t* __restrict__ acc, size_t n, uint8_t* __restrict__ bitmap) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;           // bitmap byte
    if (b >= ((n + 7) >> 3)) return;
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t i = b * 8 + k;
        if (i < n && acc[i]) v |= 1u << k;
    }
    bitmap[b] = (uint8_t)v;
}

// Compaction shared by both curves: `ung` / `grp` / `rejected` are this lane's class; one atomicAdd per
// workgroup and list (atomicAdds on one word serialise at ~11 ns each on MI355X).
__device__ __forceinline__ void group_split_emit(size_t i, u32 s, bool ung, bool grp, bool rejected, const GroupState& g) {
    __shared__ u32 sh_cnt[3][4];
    __shared__ u32 sh_base[2];
    const unsigned long long mu = __ballot(ung), mg = __ballot(grp), mr = __ballot(rejected);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh_cnt[0][wave] = (u32)__popcll(mu); sh_cnt[1][wave] = (u32)__popcll(mg); sh_cnt[2][wave] = (u32)__popcll(mr); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 tu = sh_cnt[0][0] + sh_cnt[0][1] + sh_cnt[0][2] + sh_cnt[0][3];
        const u32 tg = sh_cnt[1][0] + sh_cnt[1][1] + sh_cnt[1][2] + sh_cnt[1][3];
        const u32 tr = sh_cnt[2][0] + sh_cnt[2][1] + sh_cnt[2][2] + sh_cnt[2][3];
        sh_base[0] = tu ? atomicAdd(&g.counters[2], tu) : 0u;
        sh_base[1] = tg ? atomicAdd(&g.counters[1], tg) : 0u;
        if (tr) atomicAdd(&g.counters[3], tr);
    }
    __syncthreads();
    u32 base_u = sh_base[0], base_g = sh_base[1];
    for (int w = 0; w < wave; ++w) { base_u += sh_cnt[0][w]; base_g += sh_cnt[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ung) g.ung_idx[base_u + (u32)__popcll(mu & below)] = (u32)i;
    if (grp) {
        g.slots[i] = s;
        g.grp_idx[base_g + (u32)__popcll(mg & below)] = (u32)i;
    }
}

}  // namespace sbv
