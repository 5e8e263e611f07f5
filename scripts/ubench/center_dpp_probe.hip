// Micro-benchmark / semantics check of k_center's replay step (round 4):
//   v_subrev_u32_dpp, v_subrev_co_u32_dpp, v_cndmask_b32, v_fmac_f64_dpp with row_newbcast -- four independent 16-lane
//   rows per wave, entry j of a row broadcast inside the row.
// 1. correctness: the block against a scalar CPU emulation, bit for bit, with and without s_nop between the
//    instructions (are the back-to-back forms free of hazards the assembler does not pad?);
// 2. throughput: SIMD cycles per step at 1 .. 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o center_dpp_probe center_dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define NOP0 ""
#define NOP2 "s_nop 1\n\t"
// FORM 0/1 (round 4, first form): t = dpp(e0) - p; borrow of t - dpp(m - 1); cndmask; fmac.  (Measured on gfx950: a DPP
// *rev* opcode broadcasts its SECOND source -- v_subrev_u32_dpp d, a, b = dpp(b) - a -- hence the plain forms.)
#define CS_SUB(J, T) "v_sub_u32_dpp " T ", %[a0], %[p] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define CS_CMP(J, T) "v_sub_co_u32_dpp " T ", vcc, %[m], " T " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define CS_SEL(OH) "v_cndmask_b32 " OH ", %[k1], v10, vcc\n\t"
#define CS_FMA(J, PAIR) "v_fmac_f64_dpp %[acc], %[val], " PAIR " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define CS_2(N, J0, J1)                                                                                                 \
    CS_SUB(J0, "v8") N CS_SUB(J1, "v9") N CS_CMP(J0, "v8") N CS_SEL("v11") N CS_CMP(J1, "v9") N CS_SEL("v13") N           \
    CS_FMA(J0, "v[10:11]") N CS_FMA(J1, "v[12:13]") N
#define CS_HEAD "v_mov_b32 v10, 0\n\tv_mov_b32 v12, 0\n\ts_nop 1\n\t"
#define CS_16(N) CS_2(N, 0, 1) CS_2(N, 2, 3) CS_2(N, 4, 5) CS_2(N, 6, 7) CS_2(N, 8, 9) CS_2(N, 10, 11) CS_2(N, 12, 13) CS_2(N, 14, 15)
#define STEPS(CODE)                                                                                                    \
    asm volatile(CS_HEAD CODE : [acc] "+v"(acc) : [a0] "v"(a0), [m] "v"(m), [val] "v"(val), [p] "v"(p), [k1] "v"(k1)      \
                 : "v8", "v9", "v10", "v11", "v12", "v13", "vcc")
// FORM 2/3 (what k_center runs): x = dpp(coverage mask) & lane bit; one.hi = x << (30 - lane) (2.0 or 0.0); fmac with
// half the value
#define CM_AND(J, T) "v_and_b32_dpp " T ", %[cm], %[lbit] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define CM_SHL(OH, T) "v_lshlrev_b32 " OH ", %[sh], " T "\n\t"
#define CM_2(N, J0, J1) CM_AND(J0, "v8") N CM_AND(J1, "v9") N CM_SHL("v11", "v8") N CM_SHL("v13", "v9") N CS_FMA(J0, "v[10:11]") N CS_FMA(J1, "v[12:13]") N
#define CM_16(N) CM_2(N, 0, 1) CM_2(N, 2, 3) CM_2(N, 4, 5) CM_2(N, 6, 7) CM_2(N, 8, 9) CM_2(N, 10, 11) CM_2(N, 12, 13) CM_2(N, 14, 15)
#define STEPS_CM(CODE)                                                                                                 \
    asm volatile(CS_HEAD CODE : [acc] "+v"(acc) : [cm] "v"(cm), [val] "v"(valh), [lbit] "v"(lbit), [sh] "v"(sh)          \
                 : "v8", "v9", "v10", "v11", "v12", "v13")

// FORM 0: compare-and-select step; 1: the same, s_nop 1 after every instruction; 2: coverage-mask step; 3: the same + nops
template <int FORM>
__global__ __launch_bounds__(64) void probe(const int* __restrict__ a0s, const int* __restrict__ ms, const double* __restrict__ vals,
                                            int nbatch, int pbase, double* out, int reps) {
    const int lane = threadIdx.x & 63;
    const int p = pbase + lane;
    const int k1 = 0x3ff00000, li = lane & 15, lbit = 1 << li, sh = 30 - li, rs = pbase + (lane & 48);
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        for (int b = 0; b < nbatch; ++b) {
            // entry -> (last covered position, covered positions - 1); an entry that covers nothing: (0x80000000, 0)
            const int a0r = a0s[(size_t)b * 64 + lane], mr = ms[(size_t)b * 64 + lane];
            const int a0 = mr > 0 ? a0r + mr - 1 : (int)0x80000000, m = mr > 0 ? mr - 1 : 0;
            const double val = vals[(size_t)b * 64 + lane], valh = val * 0.5;
            // coverage mask of this lane's row: positions [a0r, a0r + mr) relative to the row start
            const int first = a0r - rs, last = first + mr - 1, b0 = first > 0 ? first : 0, b1 = last < 15 ? last : 15;
            const int cm = (mr > 0 && b1 >= b0) ? (int)((2u << b1) - (1u << b0)) : 0;
            if (FORM == 0) STEPS(CS_16(NOP0));
            else if (FORM == 1) STEPS(CS_16(NOP2));
            else if (FORM == 2) STEPS_CM(CM_16(NOP0));
            else STEPS_CM(CM_16(NOP2));
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;
}

// ---- cost of the single instructions: 64 of one kind per loop trip, `reps` trips, 8 waves per SIMD
#define R16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(64) void opcost(double* out, int reps, int seed) {
    const int lane = threadIdx.x & 63;
    int a = seed + lane, b = seed * 3 + lane, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double val = 1.0 / (double)(3 + lane), one = 1.0, acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0)        // 32-bit DPP subtract, four independent destinations
            asm volatile(R16("v_sub_u32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_sub_u32_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_u32_dpp %2, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_sub_u32_dpp %3, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        else if (KIND == 1)   // the same without DPP
            asm volatile(R16("v_sub_u32 %0, %4, %5\n\tv_sub_u32 %1, %4, %5\n\tv_sub_u32 %2, %4, %5\n\tv_sub_u32 %3, %4, %5\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        else if (KIND == 2)   // v_fmac_f64_dpp, four independent accumulators
            asm volatile(R16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %2, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(val), "v"(one));
        else if (KIND == 3)   // v_fmac_f64 without DPP, four independent accumulators
            asm volatile(R16("v_fmac_f64 %0, %4, %5\n\tv_fmac_f64 %1, %4, %5\n\tv_fmac_f64 %2, %4, %5\n\tv_fmac_f64 %3, %4, %5\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(val), "v"(one));
        else if (KIND == 4)   // v_fmac_f64_dpp, ONE accumulator (dependent chain)
            asm volatile(R16("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_fmac_f64_dpp %0, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(val), "v"(one));
        else if (KIND == 5)   // v_add_f64 with an SGPR operand, one accumulator (round 3's add)
            asm volatile(R16("v_add_f64 %0, %0, %4\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %0, %0, %4\n\tv_add_f64 %0, %0, %4\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(val), "v"(one));
        else if (KIND == 6)   // v_cndmask_b32
            asm volatile(R16("v_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %4, %5, vcc\n\tv_cndmask_b32 %2, %4, %5, vcc\n\tv_cndmask_b32 %3, %4, %5, vcc\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 7)   // v_mov_b64_dpp
            asm volatile(R16("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b64_dpp %2, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(val), "v"(one));
        else if (KIND == 8)   // v_sub_co_u32_dpp (writes vcc)
            asm volatile(R16("v_sub_co_u32_dpp %0, vcc, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_sub_co_u32_dpp %1, vcc, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_sub_co_u32_dpp %2, vcc, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_sub_co_u32_dpp %3, vcc, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");
        else if (KIND == 9)   // v_fma_f64 (VOP3), one accumulator
            asm volatile(R16("v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %0, %4, %5, %0\n\t")
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(val), "v"(one));
        else if (KIND == 10)  // v_mov_b32_dpp
            asm volatile(R16("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc0 + acc1 + acc2 + acc3 + (double)(c0 + c1 + c2 + c3);
}

template <int KIND> static void run_opcost(const char* name, double* d_out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 8; wps *= 8) {
        const int grid = 256 * 4 * wps, reps = 2000;
        hipLaunchKernelGGL(opcost<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 2, 5);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(opcost<KIND>, dim3(grid), dim3(64), 0, 0, d_out, reps, 5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ops_per_simd = (double)wps * reps * 64;
        printf("  %-44s %d waves/SIMD: %.2f cycles per instruction and SIMD (2.4 GHz)\n", name, wps, ms * 1e6 / ops_per_simd * 2.4);
    }
}

int main(int argc, char** argv) {
    const int nbatch = 64;
    std::vector<int> a0(nbatch * 64), m(nbatch * 64);
    std::vector<double> val(nbatch * 64);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const int pbase = 1000;
    for (int i = 0; i < nbatch * 64; ++i) {
        a0[i] = pbase - 40 + (int)(rnd() % 110);
        m[i] = (rnd() % 5 == 0) ? 0 : 20 + (int)(rnd() % 16);
        val[i] = 1.0 / (double)(20 + rnd() % 16);
    }
    // CPU emulation: lane l of row r gets, for batch b and step j, entry (b, 16 r + j)
    std::vector<double> want(64, 0.0);
    for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        const int row = l >> 4, p = pbase + l;
        for (int b = 0; b < nbatch; ++b)
            for (int j = 0; j < 16; ++j) {
                const int e = b * 64 + row * 16 + j;
                if ((unsigned)(p - a0[e]) < (unsigned)m[e]) acc += val[e];
            }
        want[l] = acc;
    }
    int *d_a0, *d_m; double *d_val, *d_out;
    const int grid_max = 256 * 4 * 8 * 4;
    CK(hipMalloc(&d_a0, a0.size() * 4)); CK(hipMalloc(&d_m, m.size() * 4)); CK(hipMalloc(&d_val, val.size() * 8));
    CK(hipMalloc(&d_out, (size_t)grid_max * 64 * 8));
    CK(hipMemcpy(d_a0, a0.data(), a0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_m, m.data(), m.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), val.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> got(64);
    const char* names[4] = {"compare+select", "compare+select+nops", "coverage mask", "coverage mask+nops"};
    for (int form = 0; form < 4; ++form) {
        auto launch = [&](int grid, int reps) {
            switch (form) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 0, 0, d_a0, d_m, d_val, nbatch, pbase, d_out, reps); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, d_a0, d_m, d_val, nbatch, pbase, d_out, reps); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(64), 0, 0, d_a0, d_m, d_val, nbatch, pbase, d_out, reps); break;
            default: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 0, 0, d_a0, d_m, d_val, nbatch, pbase, d_out, reps); break;
            }
        };
        launch(1, 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d_out, 64 * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) if (memcmp(&got[l], &want[l], 8) != 0) { if (bad < 4) printf("  lane %d: got %.17g want %.17g\n", l, got[l], want[l]); ++bad; }
        printf("%-16s correctness: %d of 64 lanes differ\n", names[form], bad);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int grid = 256 * 4 * wps, reps = 200;
            launch(grid, 2);
            CK(hipEventRecord(e0)); launch(grid, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double steps_per_simd = (double)wps * reps * nbatch * 16;
            printf("    %d waves/SIMD: %.3f ms, %.2f ns per step and SIMD = %.1f cycles at 2.4 GHz\n", wps, ms, ms * 1e6 / steps_per_simd, ms * 1e6 / steps_per_simd * 2.4);
        }
    }
    printf("single instructions:\n");
    run_opcost<0>("v_sub_u32_dpp (4 independent)", d_out);
    run_opcost<1>("v_sub_u32 (4 independent)", d_out);
    run_opcost<10>("v_mov_b32_dpp (4 independent)", d_out);
    run_opcost<8>("v_sub_co_u32_dpp (4 independent, vcc)", d_out);
    run_opcost<6>("v_cndmask_b32 (4 independent)", d_out);
    run_opcost<2>("v_fmac_f64_dpp (4 accumulators)", d_out);
    run_opcost<3>("v_fmac_f64 (4 accumulators)", d_out);
    run_opcost<4>("v_fmac_f64_dpp (1 accumulator, chain)", d_out);
    run_opcost<9>("v_fma_f64 (1 accumulator, chain)", d_out);
    run_opcost<5>("v_add_f64 (1 accumulator, chain)", d_out);
    run_opcost<7>("v_mov_b64_dpp (4 independent)", d_out);
    return 0;
}
