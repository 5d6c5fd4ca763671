// microbench.hip — gfx950 integer-pipe instruction rates that price the P-256 kernels.
//
// "measure first with a v_mad_u64_u32 microbenchmark: every later estimate
// hangs on it".  For each instruction: issue rate with 8 independent chains per wave at
// 1/2/4/8 waves per SIMD, and dependent-chain latency.  Prints one JSON object per line.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITER = 2048;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

// ---- independent: 8 accumulators, 8 instructions per unrolled step, x4 steps per iteration
#define DEF_KERNEL_U64(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint32_t a, uint32_t b) {      \
        uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                     \
        for (int i = 0; i < ITER; ++i) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                    \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)           \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(x), "v"(y) : "vcc");                                        \
            }                                                                                  \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;    \
    }
#define DEF_KERNEL_U32(NAME, ASM)                                                              \
    __global__ __launch_bounds__(256) void NAME(uint64_t* out, uint32_t a, uint32_t b) {      \
        uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                                     \
        for (int i = 0; i < ITER; ++i) {                                                       \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                    \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)           \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                             : "v"(x), "v"(y) : "vcc");                                        \
            }                                                                                  \
        }                                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;    \
    }

#define A_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n\t"
#define A_MAD64_S(n) "v_mad_u64_u32 %" #n ", s[20:21], %8, %9, %" #n "\n\t"
#define A_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %" #n "\n\t"
#define A_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %" #n ", %" #n "\n\t"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n\t"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n\t"
#define A_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n\t"
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n\t"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n\t"
#define A_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n\t"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n\t"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n\t"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 31\n\t"

DEF_KERNEL_U64(k_mad64, A_MAD64)
DEF_KERNEL_U64(k_lshladd64, A_LSHLADD64)
DEF_KERNEL_U64(k_fma64, A_FMA64)
DEF_KERNEL_U32(k_mullo, A_MULLO)
DEF_KERNEL_U32(k_mulhi, A_MULHI)
DEF_KERNEL_U32(k_mad24, A_MAD24)
DEF_KERNEL_U32(k_add, A_ADD)
DEF_KERNEL_U32(k_mov, A_MOV)
DEF_KERNEL_U32(k_addco, A_ADDCO)
DEF_KERNEL_U32(k_add3, A_ADD3)
DEF_KERNEL_U32(k_cndmask, A_CNDMASK)
DEF_KERNEL_U32(k_alignbit, A_ALIGNBIT)

// dependent chain of v_mad_u64_u32 on ONE accumulator (latency), 32 per iteration
__global__ __launch_bounds__(256) void k_mad64_dep(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t r = threadIdx.x;
    uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// carry chain as the compiler emits it (v_add_co + 7 v_addc_co, with its own hazard nops)
__global__ __launch_bounds__(256) void k_carry_chain(uint64_t* out, uint32_t a, uint32_t b) {
    uint32_t r[8], s[8];
    for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x + i; s[i] = a * (i + 1) + b; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned c = 0, co;
#pragma unroll
            for (int l = 0; l < 8; ++l) { r[l] = __builtin_addc(r[l], s[l], c, &co); c = co; }
            asm volatile("" : "+v"(r[0]), "+v"(r[7]));
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);

static int run(const char* name, kern_t k, double instr_per_thread, uint64_t* d_out, int cus, double clock_ghz) {
    for (int wps = 1; wps <= 8; wps *= 2) {      // waves per SIMD = blocks per CU (256-thread blocks)
        const int grid = cus * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 12345u, 0x9e3779b9u);   // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d_out, 12345u, 0x9e3779b9u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_instr = (double)grid * 4 * instr_per_thread;          // wave-level instructions
        const double per_s = wave_instr / (best * 1e-3);
        const double cyc_per_instr_simd = (clock_ghz * 1e9) * (cus * 4.0) / per_s;   // SIMD-cycles per wave-instruction
        printf("{\"bench\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s\": %.4e, "
               "\"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd_at_%.1fGHz\": %.3f}\n",
               name, wps, best, per_s, per_s * 64, clock_ghz, cyc_per_instr_simd);
        fflush(stdout);
        CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate / 1e6;
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f}\n", prop.name, prop.gcnArchName, cus, clock_ghz);
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(uint64_t)));
    const double ind = (double)ITER * 4 * 8;
    if (run("v_mad_u64_u32", k_mad64, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_mad_u64_u32_dependent", k_mad64_dep, (double)ITER * 32, d_out, cus, clock_ghz)) return 1;
    if (run("v_mul_lo_u32", k_mullo, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_mul_hi_u32", k_mulhi, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_mad_u32_u24", k_mad24, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_fma_f64", k_fma64, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_lshl_add_u64", k_lshladd64, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_add_u32", k_add, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_add_co_u32", k_addco, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_add3_u32", k_add3, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_cndmask_b32", k_cndmask, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_alignbit_b32", k_alignbit, ind, d_out, cus, clock_ghz)) return 1;
    if (run("v_mov_b32", k_mov, ind, d_out, cus, clock_ghz)) return 1;
    if (run("carry_chain_8limb(addc x8)", k_carry_chain, (double)ITER * 4 * 8, d_out, cus, clock_ghz)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
