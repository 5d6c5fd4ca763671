// p256_kernels.h — launch interface between the C-ABI layer (sbv_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "p256_core.h"

#define SBV_TUPLE_BYTES 160
#define SBV_VERIFY_BLOCK 256

namespace sbv {

int prep_chunk_T(size_t n);
hipError_t launch_p256_prep(const uint8_t* d_tuples, size_t n, const Scratch& s, hipStream_t stream);
hipError_t launch_p256_verify(const Scratch& s, size_t n, u32* d_qtab, const apt* d_gtab, uint8_t* d_bitmap,
                              hipStream_t stream);
void host_build_gtable(apt* out);   // 33 x 128 affine multiples of G (one-time table setup)

}  // namespace sbv
