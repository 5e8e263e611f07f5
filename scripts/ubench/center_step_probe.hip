// Which form of the center rule's replay step issues fastest on gfx950 (round 6)?  Bare steps, no memory: every wave
// runs `reps` batches of 16 steps on fixed registers, 8 waves per SIMD on every SIMD of the chip; the figure is SIMD
// cycles per step at the nominal 2.4 GHz (v_sub_u32 x 64 as the 2-cycle yardstick of the same run).
//   F0  what k_center2 runs: two steps interleaved (AND AND SHL SHL FMA FMA)
//   F1  four steps interleaved (AND x4, SHL x4, FMA x4)
//   F2  the FMAs alone (v_fmac_f64_dpp, one dependent chain)
//   F3  AND + SHL alone (independent)
//   F4  the FMAs alone without DPP
//   F5  F0 with two accumulators (two independent chains: is the dependent fmac chain the limit?)
//   F6  F0 with a plain AND in place of the DPP one (what does the DPP on the 32-bit instruction cost in the mix?)
//   F7  two coverage masks per v_and_b32_dpp: AND_dpp, SHL, AND, SHL, FMA, FMA -- five instructions per two entries
//       with ONE DPP 32-bit instruction (timing only: the second `one` is not the product's value)
//   F8  eight steps interleaved (AND x8, SHL x8, FMA x8)
// build: hipcc --offload-arch=gfx950 -O3 -o center_step_probe center_step_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define DPPJ(J) " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define AND_D(J, T) "v_and_b32_dpp " T ", %[cm], %[lbit]" DPPJ(J)
#define AND_P(T) "v_and_b32 " T ", %[cm], %[lbit]\n\t"
#define AND_K(T, S) "v_and_b32 " T ", %[hbit], " S "\n\t"
#define SHL(OH, T) "v_lshlrev_b32 " OH ", %[sh], " T "\n\t"
#define FMA_D(J, ACC, PAIR) "v_fmac_f64_dpp " ACC ", %[val], " PAIR DPPJ(J)
#define FMA_P(ACC, PAIR) "v_fmac_f64 " ACC ", %[val], " PAIR "\n\t"
#define HEAD "v_mov_b32 v10, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v24, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v30, 0\n\ts_nop 1\n\t"
#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"

#define F0_2(J0, J1) AND_D(J0, "v8") AND_D(J1, "v9") SHL("v11", "v8") SHL("v13", "v9") FMA_D(J0, "%[a0]", "v[10:11]") FMA_D(J1, "%[a0]", "v[12:13]")
#define F0_16 F0_2(0, 1) F0_2(2, 3) F0_2(4, 5) F0_2(6, 7) F0_2(8, 9) F0_2(10, 11) F0_2(12, 13) F0_2(14, 15)
#define F1_4(J0, J1, J2, J3) AND_D(J0, "v8") AND_D(J1, "v9") AND_D(J2, "v14") AND_D(J3, "v15") SHL("v11", "v8") SHL("v13", "v9") SHL("v17", "v14") SHL("v19", "v15") \
    FMA_D(J0, "%[a0]", "v[10:11]") FMA_D(J1, "%[a0]", "v[12:13]") FMA_D(J2, "%[a0]", "v[16:17]") FMA_D(J3, "%[a0]", "v[18:19]")
#define F1_16 F1_4(0, 1, 2, 3) F1_4(4, 5, 6, 7) F1_4(8, 9, 10, 11) F1_4(12, 13, 14, 15)
#define F2_2(J0, J1) FMA_D(J0, "%[a0]", "v[10:11]") FMA_D(J1, "%[a0]", "v[12:13]")
#define F2_16 F2_2(0, 1) F2_2(2, 3) F2_2(4, 5) F2_2(6, 7) F2_2(8, 9) F2_2(10, 11) F2_2(12, 13) F2_2(14, 15)
#define F3_2(J0, J1) AND_D(J0, "v8") AND_D(J1, "v9") SHL("v11", "v8") SHL("v13", "v9")
#define F3_16 F3_2(0, 1) F3_2(2, 3) F3_2(4, 5) F3_2(6, 7) F3_2(8, 9) F3_2(10, 11) F3_2(12, 13) F3_2(14, 15)
#define F4_2 FMA_P("%[a0]", "v[10:11]") FMA_P("%[a0]", "v[12:13]")
#define F4_16 F4_2 F4_2 F4_2 F4_2 F4_2 F4_2 F4_2 F4_2
#define F5_2(J0, J1) AND_D(J0, "v8") AND_D(J1, "v9") SHL("v11", "v8") SHL("v13", "v9") FMA_D(J0, "%[a0]", "v[10:11]") FMA_D(J1, "%[a1]", "v[12:13]")
#define F5_16 F5_2(0, 1) F5_2(2, 3) F5_2(4, 5) F5_2(6, 7) F5_2(8, 9) F5_2(10, 11) F5_2(12, 13) F5_2(14, 15)
#define F6_2(J0, J1) AND_P("v8") AND_P("v9") SHL("v11", "v8") SHL("v13", "v9") FMA_D(J0, "%[a0]", "v[10:11]") FMA_D(J1, "%[a0]", "v[12:13]")
#define F6_16 F6_2(0, 1) F6_2(2, 3) F6_2(4, 5) F6_2(6, 7) F6_2(8, 9) F6_2(10, 11) F6_2(12, 13) F6_2(14, 15)
#define F7_2(J0, J1) AND_D(J0, "v8") SHL("v11", "v8") AND_K("v9", "v8") SHL("v13", "v9") FMA_D(J0, "%[a0]", "v[10:11]") FMA_D(J1, "%[a0]", "v[12:13]")
#define F7_16 F7_2(0, 1) F7_2(2, 3) F7_2(4, 5) F7_2(6, 7) F7_2(8, 9) F7_2(10, 11) F7_2(12, 13) F7_2(14, 15)
#define F8_8(B) AND_D(B##0, "v8") AND_D(B##1, "v9") AND_D(B##2, "v14") AND_D(B##3, "v15") AND_D(B##4, "v20") AND_D(B##5, "v21") AND_D(B##6, "v26") AND_D(B##7, "v27") \
    SHL("v11", "v8") SHL("v13", "v9") SHL("v17", "v14") SHL("v19", "v15") SHL("v23", "v20") SHL("v25", "v21") SHL("v29", "v26") SHL("v31", "v27") \
    FMA_D(B##0, "%[a0]", "v[10:11]") FMA_D(B##1, "%[a0]", "v[12:13]") FMA_D(B##2, "%[a0]", "v[16:17]") FMA_D(B##3, "%[a0]", "v[18:19]") \
    FMA_D(B##4, "%[a0]", "v[22:23]") FMA_D(B##5, "%[a0]", "v[24:25]") FMA_D(B##6, "%[a0]", "v[28:29]") FMA_D(B##7, "%[a0]", "v[30:31]")
// (row_newbcast takes 0..15: the second half written out)
#define F8_16 F8_8() \
    AND_D(8, "v8") AND_D(9, "v9") AND_D(10, "v14") AND_D(11, "v15") AND_D(12, "v20") AND_D(13, "v21") AND_D(14, "v26") AND_D(15, "v27") \
    SHL("v11", "v8") SHL("v13", "v9") SHL("v17", "v14") SHL("v19", "v15") SHL("v23", "v20") SHL("v25", "v21") SHL("v29", "v26") SHL("v31", "v27") \
    FMA_D(8, "%[a0]", "v[10:11]") FMA_D(9, "%[a0]", "v[12:13]") FMA_D(10, "%[a0]", "v[16:17]") FMA_D(11, "%[a0]", "v[18:19]") \
    FMA_D(12, "%[a0]", "v[22:23]") FMA_D(13, "%[a0]", "v[24:25]") FMA_D(14, "%[a0]", "v[28:29]") FMA_D(15, "%[a0]", "v[30:31]")

template <int FORM>
__global__ __launch_bounds__(64) void probe(double *out, int reps, int seed) {
    const int lane = threadIdx.x & 63, li = lane & 15;
    int cm = (seed * 2654435761u + lane * 40503u) & 0xffff;
    const int lbit = 1 << li, sh = 30 - li, hbit = 0x10000 << li;
    double val = 0.5 / (double)(25 + (lane & 7)), a0 = 0.0, a1 = 0.0;
    for (int r = 0; r < reps; ++r) {
#define RUN(CODE) asm volatile(HEAD CODE : [a0] "+v"(a0), [a1] "+v"(a1) : [cm] "v"(cm), [val] "v"(val), [lbit] "v"(lbit), [sh] "v"(sh), [hbit] "v"(hbit) : CLOB)
        if (FORM == 0) RUN(F0_16);
        else if (FORM == 1) RUN(F1_16);
        else if (FORM == 2) RUN(F2_16);
        else if (FORM == 3) RUN(F3_16);
        else if (FORM == 4) RUN(F4_16);
        else if (FORM == 5) RUN(F5_16);
        else if (FORM == 6) RUN(F6_16);
        else if (FORM == 7) RUN(F7_16);
        else RUN(F8_16);
    }
    out[(size_t)blockIdx.x * 64 + lane] = a0 + a1;
}

__global__ __launch_bounds__(64) void yard(double *out, int reps, int seed) {
    int a = seed + threadIdx.x, b = seed * 3, c0 = 1, c1 = 2, c2 = 3, c3 = 4;
    for (int r = 0; r < reps; ++r)
        asm volatile("v_sub_u32 %0, %4, %5\n\tv_sub_u32 %1, %4, %5\n\tv_sub_u32 %2, %4, %5\n\tv_sub_u32 %3, %4, %5\n\t"
                     "v_sub_u32 %0, %4, %5\n\tv_sub_u32 %1, %4, %5\n\tv_sub_u32 %2, %4, %5\n\tv_sub_u32 %3, %4, %5\n\t"
                     "v_sub_u32 %0, %4, %5\n\tv_sub_u32 %1, %4, %5\n\tv_sub_u32 %2, %4, %5\n\tv_sub_u32 %3, %4, %5\n\t"
                     "v_sub_u32 %0, %4, %5\n\tv_sub_u32 %1, %4, %5\n\tv_sub_u32 %2, %4, %5\n\tv_sub_u32 %3, %4, %5\n\t"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = (double)(c0 + c1 + c2 + c3);
}

template <typename K> static double run(const char *name, K kern, double *d_out, int wps, int per_rep) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 4 * wps, reps = 4000;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out, 8, 5);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out, reps, 5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * reps * per_rep);
    printf("  %-58s %d waves/SIMD: %.2f cycles per %s\n", name, wps, cyc, per_rep == 16 ? "step" : "instruction");
    return cyc;
}

int main() {
    double *d_out; CK(hipMalloc(&d_out, (size_t)256 * 4 * 8 * 64 * 8));
    for (int wps : {8, 4, 1}) {
        run("yardstick: v_sub_u32 (16 per trip)", yard, d_out, wps, 16);
        run("F0 two steps interleaved (the product's)", probe<0>, d_out, wps, 16);
        run("F1 four steps interleaved", probe<1>, d_out, wps, 16);
        run("F8 eight steps interleaved", probe<8>, d_out, wps, 16);
        run("F2 v_fmac_f64_dpp alone (dependent)", probe<2>, d_out, wps, 16);
        run("F4 v_fmac_f64 alone, no DPP (dependent)", probe<4>, d_out, wps, 16);
        run("F3 AND_dpp + SHL alone", probe<3>, d_out, wps, 16);
        run("F5 F0 with two accumulators", probe<5>, d_out, wps, 16);
        run("F6 F0 with a plain AND", probe<6>, d_out, wps, 16);
        run("F7 two masks per AND_dpp (5 instructions / 2 steps)", probe<7>, d_out, wps, 16);
    }
    return 0;
}
