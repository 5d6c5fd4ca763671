// sbv_common.h — shared definitions for the P-256 verification core.
//
// The arithmetic headers (p256_fe.h, p256_sc.h, p256_pt.h, p256_core.h) are written once and
// compiled by hipcc for gfx950 (the product) and — for the CPU test tier only — by g++ into
// tests/emul (a lane-by-lane emulation of the kernels that lets the container, which has no
// GPU, diff the device algorithm against the oracle before any GPU minute is spent).  The
// product library never contains or calls the host build of these functions for verification;
// the only host use inside libsbv.so is the one-time generation of the fixed-base table.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SBV_HD __host__ __device__ __forceinline__
#define SBV_HD_NOINLINE __host__ __device__ __noinline__
#else
#define SBV_HD inline
#define SBV_HD_NOINLINE
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define SBV_UNROLL _Pragma("unroll")
#define SBV_NOUNROLL _Pragma("unroll 1")
#else
#define SBV_UNROLL
#define SBV_NOUNROLL
#endif

namespace sbv {

typedef uint32_t u32;
typedef uint64_t u64;

// 256-bit value, 8 little-endian 32-bit limbs (limb 0 = least significant).
struct u256 {
    u32 v[8];
};

// add with carry-in/out, sub with borrow-in/out (carry/borrow are 0 or 1).
// Under clang (hipcc) the builtins lower to v_add_co_u32 / v_addc_co_u32 chains on gfx950;
// the 64-bit formulation (used by g++ for tests/emul) would cost 3-4x the instructions there.
#if defined(__clang__)
SBV_HD u32 addc(u32 a, u32 b, u32& carry) {
    unsigned co;
    const u32 r = __builtin_addc(a, b, carry, &co);
    carry = co;
    return r;
}
SBV_HD u32 subb(u32 a, u32 b, u32& borrow) {
    unsigned bo;
    const u32 r = __builtin_subc(a, b, borrow, &bo);
    borrow = bo;
    return r;
}
#else
SBV_HD u32 addc(u32 a, u32 b, u32& carry) {
    u64 t = (u64)a + b + carry;
    carry = (u32)(t >> 32);
    return (u32)t;
}
SBV_HD u32 subb(u32 a, u32 b, u32& borrow) {
    u64 t = (u64)a - b - borrow;
    borrow = (u32)(t >> 63);
    return (u32)t;
}
#endif

SBV_HD u32 add256(u256& r, const u256& a, const u256& b) {
    u32 c = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) r.v[i] = addc(a.v[i], b.v[i], c);
    return c;
}
SBV_HD u32 sub256(u256& r, const u256& a, const u256& b) {
    u32 bw = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) r.v[i] = subb(a.v[i], b.v[i], bw);
    return bw;
}
SBV_HD bool is_zero256(const u256& a) {
    u32 o = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) o |= a.v[i];
    return o == 0;
}
SBV_HD bool eq256(const u256& a, const u256& b) {
    u32 o = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// a < b  (borrow of a - b)
SBV_HD bool lt256(const u256& a, const u256& b) {
    u32 bw = 0;
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) (void)subb(a.v[i], b.v[i], bw);
    return bw != 0;
}
// r = c ? a : b
SBV_HD void select256(u256& r, bool c, const u256& a, const u256& b) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
}
// 32 big-endian bytes -> limbs
SBV_HD void from_be32(u256& r, const uint8_t* b) {
    SBV_UNROLL
    for (int i = 0; i < 8; ++i) {
        const uint8_t* p = b + (7 - i) * 4;
        r.v[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
    }
}
SBV_HD u32 bswap32(u32 x) {
    return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24);
}

}  // namespace sbv
