#pragma once
#include <hip/hip_runtime.h>

#include "ed25519_core.h"
#include "p256_kernels.h"

namespace sbv {
hipError_t launch_ed25519_verify(const uint8_t* d_tuples, size_t n, u32* d_qtab, const aniels* d_btab, uint8_t* d_bitmap,
                                 hipStream_t stream);
// device buffers of the Ed25519 grouped step that the P-256 GroupBuffers do not already provide
struct EdGroupBuffers {
    aniels* ktab = nullptr;       // [max_groups][32 x 128] per-batch combs of -A
    uint8_t* okb = nullptr;       // [cap] S < L && k < L
    size_t cap = 0;
    u32 max_groups = 0;
};
// Grouped step (ed25519_group.h).  `b` supplies the grouping arrays, jbases, tmp, kvalid, acc and gacc (32 words per
// tuple, stride b.gacc_cap); ev_fork of `y` must have been recorded on `stream` before the call.
hipError_t launch_ed25519_verify_grouped(const uint8_t* d_tuples, size_t n, const GroupBuffers& b, const EdGroupBuffers& eb,
                                         u32* d_qtab, const aniels* d_btab, uint8_t* d_bitmap, hipStream_t stream,
                                         const GroupSync& y);
void host_build_ed_btable(aniels* out);   // 32 x 128 affine-Niels multiples of B (one-time setup)
#define SBV_ED_BTAB_ENTRIES (SBV_ED_BTAB_WINDOWS * SBV_ED_BTAB_PER_WINDOW)
}  // namespace sbv
