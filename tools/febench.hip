// febench.hip — A/B of the two field representations on the device: 8 x 32-bit limbs with carry chains
// (p256_fe.h, Jacobian mixed addition of p256_pt.h) against 9 x 29-bit carry-free limbs (p256_fe29.h, XYZZ mixed
// addition of p256_pt29.h).  For each: a dependent chain of field multiplications / squarings per lane, and a
// chain of comb-style mixed additions with table gathers (the shape of the G phase / Q phase kernels), at the
// launch bounds the verify kernels use.  Results of the two forms are cross-checked on the host (same points).
// Prints one JSON object per line (profiles/r02/febench_w*.jsonl).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

#include "../consensus_amd/csrc/p256_core.h"
#include "../consensus_amd/csrc/p256_pt29.h"

using namespace sbv;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef FEB_WAVES
#define FEB_WAVES 3
#endif

// ---- field chains ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, FEB_WAVES) void k_mul_v0(u32* io, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    fe a, b;
    for (int l = 0; l < 8; ++l) { a.v[l] = io[(size_t)l * n + i]; b.v[l] = io[(size_t)(8 + l) * n + i]; }
#pragma unroll 1
    for (int t = 0; t < iters; ++t) {
        fe_mul(a, a, b);
        fe_mul(b, b, a);
    }
    for (int l = 0; l < 8; ++l) { io[(size_t)l * n + i] = a.v[l]; io[(size_t)(8 + l) * n + i] = b.v[l]; }
}
__global__ __launch_bounds__(256, FEB_WAVES) void k_sqr_v0(u32* io, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    fe a;
    for (int l = 0; l < 8; ++l) a.v[l] = io[(size_t)l * n + i];
#pragma unroll 1
    for (int t = 0; t < iters; ++t) { fe_sqr(a, a); fe_sqr(a, a); }
    for (int l = 0; l < 8; ++l) io[(size_t)l * n + i] = a.v[l];
}
__global__ __launch_bounds__(256, FEB_WAVES) void k_mul_29(u32* io, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    fe29 a, b;
    {
        u32 wa[8], wb[8];
        for (int l = 0; l < 8; ++l) { wa[l] = io[(size_t)l * n + i]; wb[l] = io[(size_t)(8 + l) * n + i]; }
        f29_unpack(a, wa); f29_unpack(b, wb);
    }
#pragma unroll 1
    for (int t = 0; t < iters; ++t) {
        f29_mul(a, a, b);
        f29_mul(b, b, a);
    }
    u32 wa[8], wb[8];
    f29_store_canon(wa, a); f29_store_canon(wb, b);
    for (int l = 0; l < 8; ++l) { io[(size_t)l * n + i] = wa[l]; io[(size_t)(8 + l) * n + i] = wb[l]; }
}
__global__ __launch_bounds__(256, FEB_WAVES) void k_sqr_29(u32* io, size_t n, int iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    fe29 a;
    {
        u32 wa[8];
        for (int l = 0; l < 8; ++l) wa[l] = io[(size_t)l * n + i];
        f29_unpack(a, wa);
    }
#pragma unroll 1
    for (int t = 0; t < iters; ++t) { f29_sqr(a, a); f29_sqr(a, a); }
    u32 wa[8];
    f29_store_canon(wa, a);
    for (int l = 0; l < 8; ++l) io[(size_t)l * n + i] = wa[l];
}

// ---- mixed-addition chains with gathers from a comb-shaped table -----------------------------------------------
// table: `windows` rows of 128 affine points (64 B each); lane i takes entry digit(i, j) of row j, j = 0..windows-1,
// sign from the digit's top bit — the access pattern of the key-comb phase.  V0 table entries are 8 x 32 Montgomery
// form (R = 2^256); the fe29 table holds the same points in the R = 2^261 domain.
__device__ __forceinline__ u32 digit_of(u32 i, int j) {
    u32 h = (i * 0x9E3779B1u) ^ ((u32)j * 0x85EBCA77u);
    h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13;
    return h;
}
__global__ __launch_bounds__(256, FEB_WAVES) void k_madd_v0(const apt* __restrict__ tab, u32* out, size_t n, int windows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    jpt R;
    pt_set_inf(R);
    u32 d = digit_of((u32)i, 0);
    apt cur;
    {
        const u32* gp = reinterpret_cast<const u32*>(tab + (d & 127u));
        fe_load16(cur.x, gp); fe_load16(cur.y, gp + 8);
    }
#pragma unroll 1
    for (int j = 0; j < windows; ++j) {
        const int jn = j + 1 < windows ? j + 1 : j;
        const u32 dn = digit_of((u32)i, jn);
        const u32* gp = reinterpret_cast<const u32*>(tab + (size_t)jn * 128 + (dn & 127u));
        apt nxt;
        fe_load16(nxt.x, gp); fe_load16(nxt.y, gp + 8);
        pt_add_mixed(R, cur, (d >> 7) & 1u, false);
        cur = nxt; d = dn;
    }
    for (int l = 0; l < 8; ++l) {
        out[(size_t)l * n + i] = R.X.v[l]; out[(size_t)(8 + l) * n + i] = R.Y.v[l]; out[(size_t)(16 + l) * n + i] = R.Z.v[l];
    }
}
__global__ __launch_bounds__(256, FEB_WAVES) void k_madd_29(const apt* __restrict__ tab, u32* out, size_t n, int windows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    xyzz R;
    pt29_set_inf(R);
    u32 d = digit_of((u32)i, 0);
    apt29 cur;
    apt29_load(cur, reinterpret_cast<const u32*>(tab + (d & 127u)));
#pragma unroll 1
    for (int j = 0; j < windows; ++j) {
        const int jn = j + 1 < windows ? j + 1 : j;
        const u32 dn = digit_of((u32)i, jn);
        apt29 nxt;
        apt29_load(nxt, reinterpret_cast<const u32*>(tab + (size_t)jn * 128 + (dn & 127u)));
        pt29_madd(R, cur, (d >> 7) & 1u);
        cur = nxt; d = dn;
    }
    u32 w[4][8];
    f29_store_canon(w[0], R.X); f29_store_canon(w[1], R.Y); f29_store_canon(w[2], R.ZZ); f29_store_canon(w[3], R.ZZZ);
    for (int c = 0; c < 4; ++c)
        for (int l = 0; l < 8; ++l) out[(size_t)(8 * c + l) * n + i] = w[c][l];
}

static double time_kernel(const std::function<void()>& launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    const size_t n = (size_t)cus * 256 * FEB_WAVES * 4;        // 4 rounds of resident workgroups
    const int iters = 256, windows = 33;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"lanes\": %zu, \"launch_bounds_waves\": %d}\n", prop.gcnArchName, cus, ghz, n, FEB_WAVES);

    // inputs: random field elements < p (as canonical words)
    std::vector<u32> h_io(16 * n);
    uint64_t s = 0x5B7F2026ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (u32)(s >> 16); };
    for (size_t i = 0; i < n; ++i)
        for (int l = 0; l < 8; ++l) {
            h_io[(size_t)l * n + i] = l == 7 ? (rnd() & 0x7FFFFFFFu) : rnd();
            h_io[(size_t)(8 + l) * n + i] = l == 7 ? (rnd() & 0x7FFFFFFFu) : rnd();
        }
    u32 *d_io, *d_out0, *d_out1;
    CHECK(hipMalloc(&d_io, 16 * n * 4)); CHECK(hipMalloc(&d_out0, 32 * n * 4)); CHECK(hipMalloc(&d_out1, 32 * n * 4));

    // comb-shaped table of real curve points: row j = k * 2^(8 j) * G, k = 1..128 (the host generator of the library)
    std::vector<apt> tab0((size_t)SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW), tab1(tab0.size());
    build_gtable(tab0.data());
    for (size_t k = 0; k < tab0.size(); ++k) {        // same points, R = 2^261 domain, canonical words
        fe29 x, y;
        f29_from_fe(x, tab0[k].x); f29_from_fe(y, tab0[k].y);
        f29_store_canon(tab1[k].x.v, x); f29_store_canon(tab1[k].y.v, y);
    }
    apt *d_tab0, *d_tab1;
    CHECK(hipMalloc(&d_tab0, tab0.size() * sizeof(apt))); CHECK(hipMalloc(&d_tab1, tab1.size() * sizeof(apt)));
    CHECK(hipMemcpy(d_tab0, tab0.data(), tab0.size() * sizeof(apt), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tab1, tab1.data(), tab1.size() * sizeof(apt), hipMemcpyHostToDevice));

    const dim3 grid((unsigned)(n / 256)), block(256);
    struct Row { const char* name; double ms; double ops; };
    std::vector<Row> rows;
    auto reset = [&]() { hipMemcpy(d_io, h_io.data(), 16 * n * 4, hipMemcpyHostToDevice); };
    reset();
    rows.push_back({"fe_mul_8x32", time_kernel([&] { hipLaunchKernelGGL(k_mul_v0, grid, block, 0, 0, d_io, n, iters); }, 5), 2.0 * iters});
    reset();
    rows.push_back({"f29_mul_9x29", time_kernel([&] { hipLaunchKernelGGL(k_mul_29, grid, block, 0, 0, d_io, n, iters); }, 5), 2.0 * iters});
    reset();
    rows.push_back({"fe_sqr_8x32", time_kernel([&] { hipLaunchKernelGGL(k_sqr_v0, grid, block, 0, 0, d_io, n, iters); }, 5), 2.0 * iters});
    reset();
    rows.push_back({"f29_sqr_9x29", time_kernel([&] { hipLaunchKernelGGL(k_sqr_29, grid, block, 0, 0, d_io, n, iters); }, 5), 2.0 * iters});
    rows.push_back({"madd_jacobian_8x32", time_kernel([&] { hipLaunchKernelGGL(k_madd_v0, grid, block, 0, 0, d_tab0, d_out0, n, windows); }, 5), (double)windows});
    rows.push_back({"madd_xyzz_9x29", time_kernel([&] { hipLaunchKernelGGL(k_madd_29, grid, block, 0, 0, d_tab1, d_out1, n, windows); }, 5), (double)windows});
    CHECK(hipDeviceSynchronize());
    for (const Row& r : rows) {
        const double wave_ops = (double)n / 64 * r.ops;
        const double cyc = ghz * 1e9 * (cus * 4.0) * (r.ms * 1e-3) / wave_ops;      // SIMD-cycles per wave-level operation
        printf("{\"bench\": \"%s\", \"ms\": %.4f, \"ops_per_lane\": %.0f, \"lane_ops_per_s\": %.4e, \"simd_cycles_per_wave_op\": %.1f}\n",
               r.name, r.ms, r.ops, (double)n * r.ops / (r.ms * 1e-3), cyc);
    }

    // cross-check the two addition chains: same affine x, y for every lane (x = X/Z^2 resp. X/ZZ)
    std::vector<u32> o0(32 * n), o1(32 * n);
    CHECK(hipMemcpy(o0.data(), d_out0, 32 * n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(o1.data(), d_out1, 32 * n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    const size_t check = n < 4096 ? n : 4096;
    for (size_t i = 0; i < check; ++i) {
        fe X, Y, Z, zi, zi2, zi3, x0, y0;
        for (int l = 0; l < 8; ++l) { X.v[l] = o0[(size_t)l * n + i]; Y.v[l] = o0[(size_t)(8 + l) * n + i]; Z.v[l] = o0[(size_t)(16 + l) * n + i]; }
        fe_inv(zi, Z); fe_sqr(zi2, zi); fe_mul(zi3, zi2, zi); fe_mul(x0, X, zi2); fe_mul(y0, Y, zi3);
        fe29 c[4];
        for (int k = 0; k < 4; ++k) { u32 w[8]; for (int l = 0; l < 8; ++l) w[l] = o1[(size_t)(8 * k + l) * n + i]; f29_unpack(c[k], w); }
        fe X1, Y1, ZZ, ZZZ, a, b, x1, y1;
        f29_to_fe(X1, c[0]); f29_to_fe(Y1, c[1]); f29_to_fe(ZZ, c[2]); f29_to_fe(ZZZ, c[3]);
        fe_inv(a, ZZ); fe_inv(b, ZZZ); fe_mul(x1, X1, a); fe_mul(y1, Y1, b);
        if (!fe_eq(x0, x1) || !fe_eq(y0, y1)) ++bad;
    }
    printf("{\"crosscheck_lanes\": %zu, \"mismatches\": %zu}\n", check, bad);
    return bad ? 2 : 0;
}
