// k256_kernels.hip — the secp256k1 variant of the hot path (k256_core.h): stage A and stage B, one lane per signature.
//
//   k_k256_prep    : range checks, e mod n, s^-1 by division steps, u1, u2 -> the limb-major scratch planes
//   k_k256_verify  : Q on the curve, 8 affine multiples of Q in the lane's strip of HBM, 64 signed 4-bit windows on a Jacobian
//                    accumulator, + u1 * G from the 16-bit comb (35.7 MB, L2 / Infinity Cache resident), one ballot per wavefront
// Doubling-bound like k_p256_verify (256 doublings of 2M + 5S per signature); the grouped / registered-key machinery of the P-256
// path is curve-independent above the field and is the next step for this curve, not part of this file.
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

#include "k256_core.h"
#include "p256_kernels.h"

namespace sbv {

struct KGlobalTuple {
    const u32* p;
    __device__ __forceinline__ u32 operator[](int i) const { return p[i]; }
};

__global__ __launch_bounds__(64) void k_k256_prep(const uint8_t* __restrict__ tuples, size_t n, Scratch s) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    k256_prep_lane(KGlobalTuple{reinterpret_cast<const u32*>(tuples + i * 160)}, i, s);
}

__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_k256_verify(Scratch s, size_t n, u32* __restrict__ qtab, const kapt* __restrict__ gtab,
                                                                   uint8_t* __restrict__ bitmap) {
    const size_t i = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    bool accept = false;
    if (i < n) accept = k256_verify_lane(s, i, qtab + i * (size_t)SBV_QTAB29_WORDS, gtab);
    const unsigned long long m = __ballot(accept);
    const int lane = threadIdx.x & 63;
    const size_t wave_first = i - (size_t)lane;
    if (lane < 8) {
        const size_t byte = (wave_first >> 3) + (size_t)lane;
        if (byte < ((n + 7) >> 3)) bitmap[byte] = (uint8_t)(m >> (8 * lane));
    }
}

static_assert(SBV_K256_QTAB_WORDS <= SBV_QTAB29_WORDS, "the per-lane strip of the P-256 generic kernel is reused");

hipError_t launch_k256_verify(const uint8_t* d_tuples, size_t n, const Scratch& s, u32* d_qtab, const kapt* d_gtab, uint8_t* d_bitmap,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_k256_prep, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_tuples, n, s);
    hipLaunchKernelGGL(k_k256_verify, dim3((unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK)), dim3(SBV_VERIFY_BLOCK), 0, stream, s, n,
                       d_qtab, d_gtab, d_bitmap);
    return hipGetLastError();
}

void host_build_k256_gtable(kapt* out) {
    std::vector<std::thread> th;
    for (int j = 0; j < SBV_K256_G_WINDOWS; ++j)
        th.emplace_back([j, out] { k256_build_g_window(j, out + (size_t)j * SBV_K256_G_PER_WINDOW, SBV_K256_G_PER_WINDOW); });
    for (auto& t : th) t.join();
}

}  // namespace sbv
