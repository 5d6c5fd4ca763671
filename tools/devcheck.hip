// devcheck.hip — on-device self-test of the P-256 arithmetic: every primitive of
// consensus_amd/csrc/p256_*.h is run on the GPU and on the host (same source, compiled twice by
// hipcc) with identical inputs and diffed; then the two kernels are diffed stage by stage
// against the in-process host emulation on a tuple file.  Diagnostic tool (not product, not
// oracle): it pinpoints WHERE device and host disagree.
//   usage: devcheck [tuples.bin]      (tuples.bin = n x 160 B)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../consensus_amd/csrc/ed25519_kernels.h"
#include "../consensus_amd/csrc/p256_kernels.h"
#include "../tests/emul/p256_legacy_m32.h"

using namespace sbv;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

enum { OP_MULW, OP_SQRW, OP_RED, OP_MUL, OP_SQR, OP_ADD, OP_SUB, OP_SCMUL, OP_SCINV, OP_SCINVG, OP_FEINVG, OP_DBL, OP_ADDM, OP_ADDQ, OP_ONCURVE, OP_REDDBG, OP_ED_MUL, OP_ED_SQR, OP_ED_ADD, OP_ED_SUB, OP_ED_FREEZE, OP_ED_INV, OP_ED_DBL, OP_ED_ADDP, OP_ED_DECOMP, OP_COUNT };
static const char* kNames[OP_COUNT] = {"mul_wide", "sqr_wide", "fe_mont_reduce", "fe_mul", "fe_sqr", "fe_add", "fe_sub",
                                       "sc_mul", "sc_inv", "sc_inv_gcd (division steps)", "fe_inv_gcd (division steps)", "pt_dbl", "pt_add_mixed", "pt_add_qent", "pt_on_curve", "reduce_debug_taps", "fe25_mul", "fe25_sqr", "fe25_add", "fe25_sub", "fe25_freeze", "fe25_inv", "ed_dbl", "ed_add_pniels", "ed_decompress"};
constexpr int IN_WORDS = 64, OUT_WORDS = 32;

// fe_mont_reduce with taps: out[0..7] = M, out[8] = k+1, out[9..17] = acc after T_hi+M,
// out[18..26] = after += M>>64, out[27..31] = low limbs after += W
__host__ __device__ inline void reduce_taps(const u32* t, u32* out) {
    u32 m[8];
    {
        u32 c = 0;
        m[0] = t[0]; m[1] = t[1]; m[2] = t[2];
        m[3] = addc(t[3], t[0], c); m[4] = addc(t[4], t[1], c); m[5] = addc(t[5], t[2], c);
        m[6] = addc(t[6], t[3], c); m[7] = addc(t[7], t[4], c);
        c = 0;
        m[6] = addc(m[6], t[0] << 1, c);
        m[7] = addc(m[7], (t[1] << 1) | (t[0] >> 31), c);
        m[7] -= t[0];
    }
    for (int i = 0; i < 8; ++i) out[i] = m[i];
    int64_t V = (int64_t)((u64)t[7] + m[4] + m[1]) - (int64_t)((u64)m[7] + m[0]);
    int32_t k = (int32_t)((V + (int64_t)0x80000000ll) >> 32);
    u32 kp1 = (u32)(k + 1);
    out[8] = kp1;
    u32 acc[9]; u32 c = 0;
    for (int i = 0; i < 8; ++i) acc[i] = addc(t[8 + i], m[i], c);
    acc[8] = c;
    for (int i = 0; i < 9; ++i) out[9 + i] = acc[i];
    c = 0;
    for (int i = 0; i < 6; ++i) acc[i] = addc(acc[i], m[i + 2], c);
    acc[6] = addc(acc[6], 0u, c); acc[7] = addc(acc[7], 0u, c); acc[8] += c;
    for (int i = 0; i < 9; ++i) out[18 + i] = acc[i];
    u32 cw = 0;
    const u32 w0 = addc(m[5], kp1, cw), w1 = addc(m[6], 0u, cw), w2 = addc(m[7], 0u, cw), w3 = cw;
    c = 0;
    acc[0] = addc(acc[0], w0, c); acc[1] = addc(acc[1], w1, c); acc[2] = addc(acc[2], w2, c); acc[3] = addc(acc[3], w3, c);
    acc[4] = addc(acc[4], 0u, c);
    out[27] = acc[0]; out[28] = acc[1]; out[29] = acc[2]; out[30] = acc[3]; out[31] = acc[4];
}

// one test case: 64 input words -> 32 output words
__host__ __device__ inline void run_op(int op, const u32* in, u32* out) {
    for (int i = 0; i < OUT_WORDS; ++i) out[i] = 0;
    fe a, b, c, d, e;
    memcpy(&a, in, 32); memcpy(&b, in + 8, 32); memcpy(&c, in + 16, 32); memcpy(&d, in + 24, 32); memcpy(&e, in + 32, 32);
    switch (op) {
        case OP_MULW: mul_wide(out, a.v, b.v); break;
        case OP_SQRW: sqr_wide(out, a.v); break;
        case OP_RED: { fe r; fe_mont_reduce(r, in); memcpy(out, &r, 32); break; }
        case OP_MUL: { fe r; fe_mul(r, a, b); memcpy(out, &r, 32); break; }
        case OP_SQR: { fe r; fe_sqr(r, a); memcpy(out, &r, 32); break; }
        case OP_ADD: { fe r; fe_add(r, a, b); memcpy(out, &r, 32); break; }
        case OP_SUB: { fe r; fe_sub(r, a, b); memcpy(out, &r, 32); break; }
        case OP_SCMUL: { sc r; sc_mul(r, a, b); memcpy(out, &r, 32); break; }
        case OP_SCINV: { sc r; sc_inv(r, a); memcpy(out, &r, 32); break; }
        case OP_SCINVG: { sc r; sc_inv_gcd(r, a); memcpy(out, &r, 32); break; }
        case OP_FEINVG: { fe r; fe_inv_gcd(r, a); memcpy(out, &r, 32); break; }
        case OP_DBL: { jpt p{a, b, c}, r; pt_dbl(r, p); memcpy(out, &r, 96); break; }
        case OP_ADDM: { jpt p{a, b, c}; apt q{d, e}; pt_add_mixed(p, q, (in[40] & 1) != 0, (in[40] & 2) != 0); memcpy(out, &p, 96); break; }
        case OP_ADDQ: { jpt p{a, b, c}; qent q; memcpy(&q, in + 24, 160); pt_add_qent(p, q, (in[0] & 1) != 0, false); memcpy(out, &p, 96); break; }
        case OP_ONCURVE: out[0] = pt_on_curve(a, b) ? 1u : 0u; break;
        case OP_REDDBG: reduce_taps(in, out); break;
    }
}

// Ed25519 field / point operations on the carry-free field (ed25519_fe.h): inputs are 255-bit words cut into limbs and
// carried to the tight form, outputs are frozen back to canonical words so that host and device compare bit for bit.
__host__ __device__ inline void ed_in(fe25& r, const u32* w) { fe25 t; fe25_from_words(t, w); fe25_carry(r, t); }
__host__ __device__ inline void ed_out(u32* out, const fe25& a) { u256 w; fe25_freeze(w, a); memcpy(out, &w, 32); }
__host__ __device__ inline void run_op_ed(int op, const u32* in, u32* out) {
    for (int i = 0; i < OUT_WORDS; ++i) out[i] = 0;
    fe25 a, b, c, d;
    ed_in(a, in); ed_in(b, in + 8); ed_in(c, in + 16); ed_in(d, in + 24);
    switch (op) {
        case OP_ED_MUL: { fe25 r; fe25_mul(r, a, b); ed_out(out, r); break; }
        case OP_ED_SQR: { fe25 r; fe25_sqr(r, a); ed_out(out, r); break; }
        case OP_ED_ADD: { fe25 r; fe25_add(r, a, b); ed_out(out, r); break; }
        case OP_ED_SUB: { fe25 r; fe25_sub(r, a, b); ed_out(out, r); break; }
        case OP_ED_FREEZE: { ed_out(out, a); break; }
        case OP_ED_INV: { fe25 r; fe25_inv(r, a); ed_out(out, r); break; }
        case OP_ED_DBL: { ept p{a, b, c, d}, r; ed_dbl(r, p); ed_out(out, r.X); ed_out(out + 8, r.Y); ed_out(out + 16, r.Z); ed_out(out + 24, r.T); break; }
        case OP_ED_ADDP: {
            ept p{a, b, c, d};
            pniels q;
            ed_in(q.YpX, in + 32); ed_in(q.YmX, in + 40); ed_in(q.Z, in + 48); ed_in(q.T2d, in + 56);
            ed_add_pniels(p, q, (in[0] & 1) != 0, (in[1] & 3) == 0);
            ed_out(out, p.X); ed_out(out + 8, p.Y); ed_out(out + 16, p.Z); ed_out(out + 24, p.T);
            break;
        }
        case OP_ED_DECOMP: { ept p; const bool ok = ed_decompress(p, in); ed_out(out, p.X); ed_out(out + 8, p.Y); ed_out(out + 16, p.Z); out[31] = ok ? 1u : 0u; break; }
    }
}

__global__ void k_unit(int op, const u32* in, u32* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) run_op(op, in + (size_t)i * IN_WORDS, out + (size_t)i * OUT_WORDS);
}
__global__ void k_unit_ed(int op, const u32* in, u32* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) run_op_ed(op, in + (size_t)i * IN_WORDS, out + (size_t)i * OUT_WORDS);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static u32 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (u32)(rng_state >> 16); }

static void reduce_mod(u32* v, const u32* m) {   // v < 2^256 -> v mod m by repeated subtraction (m has top bits set)
    u256 a, mm, d; memcpy(&a, v, 32); memcpy(&mm, m, 32);
    while (!lt256(a, mm)) { sub256(d, a, mm); a = d; }
    memcpy(v, &a, 32);
}

static int unit_tests() {
    const int n = 4096;
    std::vector<u32> in((size_t)n * IN_WORDS), out_h((size_t)n * OUT_WORDS), out_d((size_t)n * OUT_WORDS);
    u32 *d_in, *d_out;
    CHECK(hipMalloc(&d_in, in.size() * 4)); CHECK(hipMalloc(&d_out, out_d.size() * 4));
    const fe p = fe_p(); const sc nn = sc_n();
    int failures = 0;
    for (int op = 0; op < OP_COUNT; ++op) {
        for (int i = 0; i < n; ++i) {
            u32* w = &in[(size_t)i * IN_WORDS];
            for (int k = 0; k < IN_WORDS; ++k) {
                const u32 r = rnd();
                const int mode = (i >> 4) & 3;          // mix in extreme limbs
                w[k] = mode == 0 ? r : mode == 1 ? ((r & 1) ? 0xFFFFFFFFu : 0u) : mode == 2 ? (r | 0xFFFF0000u) : (r & 0xFFFFu);
            }
            if (op != OP_MULW && op != OP_SQRW && op < OP_ED_MUL) {
                const u32* mod = (op == OP_SCMUL || op == OP_SCINV || op == OP_SCINVG) ? nn.v : p.v;
                for (int f = 0; f < 8; ++f) reduce_mod(w + 8 * f, mod);
                if (op == OP_RED || op == OP_REDDBG) reduce_mod(w + 8, p.v);       // T < p * 2^256
            }
        }
        const int cnt = (op == OP_SCINV || op == OP_ED_INV || op == OP_ED_DECOMP) ? 256 : n;
        for (int i = 0; i < cnt; ++i) {
            if (op >= OP_ED_MUL) run_op_ed(op, &in[(size_t)i * IN_WORDS], &out_h[(size_t)i * OUT_WORDS]);
            else run_op(op, &in[(size_t)i * IN_WORDS], &out_h[(size_t)i * OUT_WORDS]);
        }
        CHECK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(d_out, 0xAB, out_d.size() * 4));
        if (op >= OP_ED_MUL) hipLaunchKernelGGL(k_unit_ed, dim3((cnt + 63) / 64), dim3(64), 0, 0, op, d_in, d_out, cnt);
        else hipLaunchKernelGGL(k_unit, dim3((cnt + 63) / 64), dim3(64), 0, 0, op, d_in, d_out, cnt);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(out_d.data(), d_out, out_d.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0, first = -1;
        for (int i = 0; i < cnt; ++i)
            if (memcmp(&out_h[(size_t)i * OUT_WORDS], &out_d[(size_t)i * OUT_WORDS], OUT_WORDS * 4)) { if (first < 0) first = i; ++bad; }
        printf("unit %-16s cases=%d mismatches=%d%s\n", kNames[op], cnt, bad,
               op == OP_REDDBG ? "   (compiler canary: the pre-fix formulation of the reduction, where a carry is "
                                 "materialised as an addend; hipcc 7.2 miscompiles it for gfx950 - informational)" : "");
        if (bad && op != OP_REDDBG) {
            ++failures;
            const u32* w = &in[(size_t)first * IN_WORDS];
            printf("  first bad case %d\n  in : ", first);
            for (int k = 0; k < 24; ++k) printf("%08x ", w[k]);
            printf("\n  host: "); for (int k = 0; k < (op == OP_REDDBG ? 32 : 16); ++k) printf("%08x ", out_h[(size_t)first * OUT_WORDS + k]);
            printf("\n  dev : "); for (int k = 0; k < (op == OP_REDDBG ? 32 : 16); ++k) printf("%08x ", out_d[(size_t)first * OUT_WORDS + k]);
            printf("\n");
        }
    }
    CHECK(hipFree(d_in)); CHECK(hipFree(d_out));
    return failures;
}

struct HostWords {
    const uint8_t* tuples;
    struct W { const uint8_t* p; u32 operator[](int i) const { u32 v; memcpy(&v, p + 4 * i, 4); return v; } };
    W operator()(int, size_t idx) const { return W{tuples + 160 * idx}; }
};

static int pipeline_test(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { printf("pipeline: cannot open %s, skipped\n", path); return 0; }
    fseek(f, 0, SEEK_END); const size_t bytes = ftell(f); fseek(f, 0, SEEK_SET);
    const size_t n = bytes / 160;
    std::vector<uint8_t> tuples(n * 160);
    if (fread(tuples.data(), 1, n * 160, f) != n * 160) { fclose(f); return 1; }
    fclose(f);
    const size_t cap = (n + 1023) & ~(size_t)1023;
    // host emulation
    std::vector<u32> hs(cap * 8 * 6); std::vector<uint8_t> hok(cap, 0);
    Scratch s{hs.data(), hs.data() + cap * 8, hs.data() + cap * 16, hs.data() + cap * 24, hs.data() + cap * 32, hs.data() + cap * 40, hok.data(), cap};
    const int T = prep_chunk_T(n);
    const size_t per_block = (size_t)64 * T, nblocks = (n + per_block - 1) / per_block;
    HostWords hw{tuples.data()};
    for (size_t b = 0; b < nblocks; ++b) for (int t = 0; t < 64; ++t) prep_chunk<true>(hw, n, s, b * per_block + t, 64, T);
    std::vector<apt> gt(SBV_G16_ENTRIES);
    host_build_g16(gt.data());
    std::vector<uint8_t> hbm((n + 7) / 8, 0);
    std::vector<u32> qt(SBV_QTAB_ENTRIES * 40 + 4);
    u32* qtp = (u32*)(((uintptr_t)qt.data() + 15) & ~(uintptr_t)15);
    for (size_t i = 0; i < n; ++i) if (verify_lane(s, i, qtp, gt.data())) hbm[i >> 3] |= (uint8_t)(1u << (i & 7));
    size_t hacc = 0; for (size_t i = 0; i < n; ++i) hacc += (hbm[i >> 3] >> (i & 7)) & 1;
    // device
    uint8_t *d_t, *d_s, *d_b, *d_rr; u32* d_q; apt* d_g;
    CHECK(hipMalloc(&d_rr, cap / 64 + 64));
    CHECK(hipMalloc(&d_t, cap * 160)); CHECK(hipMalloc(&d_s, cap * 193)); CHECK(hipMalloc(&d_q, cap * (size_t)SBV_QTAB29_WORDS * 4));
    // the device kernels take the G comb in the carry-free field's domain (R = 2^261); the host side above is the 8 x 32 form
    const int gbits = 16;
    std::vector<apt> gc(gcomb_entries(gbits)), gc261(gcomb_entries(gbits));
    host_build_gcomb(gbits, gc.data());
    host_convert_table_r261(gc.data(), gc261.data(), gc.size());
    CHECK(hipMalloc(&d_b, cap / 8)); CHECK(hipMalloc(&d_g, gc261.size() * sizeof(apt)));
    CHECK(hipMemcpy(d_t, tuples.data(), n * 160, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_g, gc261.data(), gc261.size() * sizeof(apt), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_s, 0, cap * 193)); CHECK(hipMemset(d_b, 0, cap / 8));
    u32* base = (u32*)d_s;
    Scratch ds{base, base + cap * 8, base + cap * 16, base + cap * 24, base + cap * 32, base + cap * 40, d_s + cap * 192, cap};
    CHECK(launch_p256_prep(d_t, n, ds, 0));
    CHECK(hipDeviceSynchronize());
    std::vector<u32> dsv(cap * 8 * 6); std::vector<uint8_t> dok(cap);
    CHECK(hipMemcpy(dsv.data(), d_s, cap * 192, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(dok.data(), d_s + cap * 192, cap, hipMemcpyDeviceToHost));
    const char* fld[6] = {"r", "u1", "u2", "qx", "qy", "sm"};
    int fails = 0;
    for (int a = 0; a < 5; ++a) {      // sm is a stage-A temporary on both sides but compare anyway below
        size_t bad = 0, first = (size_t)-1;
        for (size_t i = 0; i < n; ++i) for (int l = 0; l < 8; ++l)
            if (hs[(size_t)a * cap * 8 + l * cap + i] != dsv[(size_t)a * cap * 8 + l * cap + i]) { ++bad; if (first == (size_t)-1) first = i; break; }
        printf("stageA %-3s mismatching tuples=%zu%s\n", fld[a], bad, bad ? "" : "");
        if (bad) {
            ++fails;
            printf("  first bad tuple %zu\n  host: ", first);
            for (int l = 7; l >= 0; --l) printf("%08x", hs[(size_t)a * cap * 8 + l * cap + first]);
            printf("\n  dev : ");
            for (int l = 7; l >= 0; --l) printf("%08x", dsv[(size_t)a * cap * 8 + l * cap + first]);
            printf("\n");
        }
    }
    { size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += hok[i] != dok[i]; printf("stageA ok  mismatching tuples=%zu\n", bad); fails += bad != 0; }
    // stage B on the DEVICE's stage-A output, and on the HOST's stage-A output uploaded
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) { CHECK(hipMemcpy(d_s, hs.data(), cap * 192, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_s + cap * 192, hok.data(), cap, hipMemcpyHostToDevice)); }
        CHECK(hipMemset(d_b, 0, cap / 8));
        CHECK(launch_p256_verify(ds, n, d_q, gcomb_make(d_g, gbits), d_b, d_rr, 0));
        CHECK(hipDeviceSynchronize());
        std::vector<uint8_t> dbm((n + 7) / 8);
        CHECK(hipMemcpy(dbm.data(), d_b, (n + 7) / 8, hipMemcpyDeviceToHost));
        size_t bad = 0, first = (size_t)-1, dacc = 0;
        for (size_t i = 0; i < n; ++i) { const int hb = (hbm[i >> 3] >> (i & 7)) & 1, db = (dbm[i >> 3] >> (i & 7)) & 1; dacc += db; if (hb != db) { ++bad; if (first == (size_t)-1) first = i; } }
        printf("stageB (%s stage-A input): n=%zu host_accept=%zu dev_accept=%zu mismatches=%zu first=%zd\n",
               pass ? "host" : "device", n, hacc, dacc, bad, (ssize_t)first);
        fails += bad != 0;
    }
    return fails;
}

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s (%s)\n", prop.name, prop.gcnArchName);
    int fails = unit_tests();
    if (argc > 1) fails += pipeline_test(argv[1]);
    printf("devcheck: %s\n", fails ? "FAILURES" : "all device results equal host results");
    return fails ? 1 : 0;
}
