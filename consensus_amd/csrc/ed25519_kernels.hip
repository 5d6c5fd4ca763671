// ed25519_kernels.hip — gfx950 kernel for the Ed25519 verifier variant (BASELINE.json configs[4]).
//
// One 128-byte tuple (R | S | A | k) per lane, 256 lanes per workgroup.  The workgroup's 32 KiB
// tile of tuples is contiguous in HBM: it is fetched with coalesced 16-byte loads into LDS
// (33-dword row pitch -> conflict-free per-lane ds_read_b32) and each lane then reads its own row.
// [k](-A): 64 signed 4-bit windows from a per-signature projective-Niels table kept in HBM
// (1 KiB per lane); [S]B: 16 signed 16-bit comb windows from a 50 MB affine-Niels table (HBM /
// Infinity Cache resident).  Complete unified addition, no exceptional cases; one field inversion per lane to
// re-encode R for the byte-wise comparison Go performs.  No scalar inversion -> no stage A.
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

#include "ed25519_core.h"
#include "p256_kernels.h"
#include "sha512_dev.h"

namespace sbv {

constexpr int kEdPitch = 33;

struct EdLdsTuple {
    const u32* row;
    __device__ __forceinline__ u32 operator[](int i) const { return row[i]; }
};

#ifndef SBV_ED_LB_WAVES
#define SBV_ED_LB_WAVES 3   // measured: 168 VGPR + 26 spilled at 3 waves/SIMD beats 190 VGPR at 2 by 13 % (profiles/r01)
#endif
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, SBV_ED_LB_WAVES) void k_ed25519_verify(const uint8_t* __restrict__ tuples, size_t n,
                                                                    u32* __restrict__ qtab,
                                                                    const aniels* __restrict__ btab,
                                                                    uint8_t* __restrict__ bitmap) {
    __shared__ u32 lds[SBV_VERIFY_BLOCK * kEdPitch];
    const size_t tile = (size_t)blockIdx.x * SBV_VERIFY_BLOCK;
    const int tid = threadIdx.x;
    const uint4* src = reinterpret_cast<const uint4*>(tuples + tile * 128);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * SBV_VERIFY_BLOCK + tid;          // 16-byte element of the tile (8 per tuple)
        const int t = e >> 3, part = e & 7;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (tile + (size_t)t < n) v = src[e];
        u32* dst = lds + t * kEdPitch + part * 4;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    const size_t i = tile + (size_t)tid;
    bool accept = false;
    if (i < n) accept = ed25519_verify_lane(EdLdsTuple{lds + tid * kEdPitch}, qtab + i * (size_t)(SBV_ED_QTAB_ENTRIES * SBV_ED_PT_WORDS), btab);
    const unsigned long long m = __ballot(accept);
    const int lane = tid & 63;
    const size_t wave_first = i - (size_t)lane;
    if (lane < 8) {
        const size_t byte = (wave_first >> 3) + (size_t)lane;
        if (byte < ((n + 7) >> 3)) bitmap[byte] = (uint8_t)(m >> (8 * lane));
    }
}

hipError_t launch_ed25519_verify(const uint8_t* d_tuples, size_t n, u32* d_qtab, const aniels* d_btab, uint8_t* d_bitmap,
                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    hipLaunchKernelGGL(k_ed25519_verify, dim3(grid), dim3(SBV_VERIFY_BLOCK), 0, stream, d_tuples, n, d_qtab, d_btab, d_bitmap);
    return hipGetLastError();
}

// Message front end (sha512_dev.h): one signature per lane, k = SHA-512(R | A | M) mod L, writes the 128-byte tuples
// the verify kernels take.  sigs = n x 64 bytes, pks = n x 32 bytes, message i = msgs[moff[i] .. moff[i+1]).
__global__ __launch_bounds__(256) void k_ed_msg_frontend(const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ pks,
                                                         const uint8_t* __restrict__ msgs, const u64* __restrict__ moff, size_t n,
                                                         u32* __restrict__ tuples) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 m0 = moff[i], m1 = moff[i + 1];
    u32 rec[32];
    ed_msg_frontend_lane(sigs + i * 64, pks + i * 32, msgs + m0, (size_t)(m1 - m0), rec);
#pragma unroll
    for (int k = 0; k < 32; ++k) tuples[i * 32 + k] = rec[k];
}

hipError_t launch_ed_msg_frontend(const uint8_t* d_sigs, const uint8_t* d_pks, const uint8_t* d_msgs, const u64* d_moff, size_t n,
                                  u32* d_tuples, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ed_msg_frontend, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_sigs, d_pks, d_msgs, d_moff, n, d_tuples);
    return hipGetLastError();
}

void host_build_ed_b16(aniels* out) {
    std::vector<std::thread> th;
    for (int j = 0; j < SBV_ED_B16_WINDOWS; ++j) th.emplace_back([j, out] { build_ed_b16_window(j, out + (size_t)j * SBV_ED_B16_PER_WINDOW); });
    for (auto& t : th) t.join();
}

void host_build_ed_bcomb(int bits, aniels* out) {
    std::vector<std::thread> th;
    for (int j = 0; j < edcomb_windows(bits); ++j) th.emplace_back([bits, j, out] { build_ed_b_window(bits, j, out + ((size_t)j << (bits - 1))); });
    for (auto& t : th) t.join();
}

}  // namespace sbv
