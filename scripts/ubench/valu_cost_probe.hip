// Cost table of single VALU instructions on gfx950 at 8 waves per SIMD (cycles per instruction and SIMD at the
// nominal 2.4 GHz; v_sub_u32 is the 2-cycle yardstick).  Used to choose k_center's replay step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define R16(X) X X X X X X X X X X X X X X X X
#define DPP " row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define OPK(NAME, BODY, ...)                                                                                           \
    __global__ __launch_bounds__(64) void NAME(double* out, int reps, int seed) {                                      \
        const int lane = threadIdx.x & 63;                                                                             \
        int a = seed + lane, b = seed * 3 + lane, c0 = 1, c1 = 2, c2 = 3, c3 = 4;                                       \
        double val = 1.0 / (double)(3 + lane), one = 1.0, acc0 = 0.0, acc1 = 0.0;                                      \
        for (int r = 0; r < reps; ++r)                                                                                 \
            asm volatile(R16(BODY) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(acc0), "+v"(acc1)                     \
                         : "v"(a), "v"(b), "v"(val), "v"(one) : __VA_ARGS__);                                                 \
        out[(size_t)blockIdx.x * 64 + lane] = acc0 + acc1 + (double)(c0 + c1 + c2 + c3);                               \
    }
// %0-%3 ints, %4 %5 doubles, %6 a, %7 b, %8 val, %9 one
OPK(k_sub, "v_sub_u32 %0, %6, %7\n\tv_sub_u32 %1, %6, %7\n\tv_sub_u32 %2, %6, %7\n\tv_sub_u32 %3, %6, %7\n\t", "vcc")
OPK(k_cnd_vcc, "v_cndmask_b32 %0, %6, %7, vcc\n\tv_cndmask_b32 %1, %6, %7, vcc\n\tv_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %6, %7, vcc\n\t", "vcc")
OPK(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %6, %7, s[20:21]\n\tv_cndmask_b32_e64 %1, %6, %7, s[20:21]\n\tv_cndmask_b32_e64 %2, %6, %7, s[20:21]\n\tv_cndmask_b32_e64 %3, %6, %7, s[20:21]\n\t", "s20", "s21")
OPK(k_cmp_cnd, "v_cmp_gt_u32 vcc, %6, %7\n\tv_cndmask_b32 %0, %6, %7, vcc\n\tv_cmp_gt_u32 vcc, %7, %6\n\tv_cndmask_b32 %1, %6, %7, vcc\n\t", "vcc")
OPK(k_cmp, "v_cmp_gt_u32 vcc, %6, %7\n\tv_cmp_gt_u32 vcc, %7, %6\n\tv_cmp_gt_u32 vcc, %6, %7\n\tv_cmp_gt_u32 vcc, %7, %6\n\t", "vcc")
OPK(k_subco, "v_sub_co_u32 %0, vcc, %6, %7\n\tv_sub_co_u32 %1, vcc, %6, %7\n\tv_sub_co_u32 %2, vcc, %6, %7\n\tv_sub_co_u32 %3, vcc, %6, %7\n\t", "vcc")
OPK(k_subco_cnd, "v_sub_co_u32 %0, vcc, %6, %7\n\tv_cndmask_b32 %1, %6, %7, vcc\n\tv_sub_co_u32 %2, vcc, %7, %6\n\tv_cndmask_b32 %3, %6, %7, vcc\n\t", "vcc")
OPK(k_addc, "v_addc_co_u32 %0, vcc, %6, %7, vcc\n\tv_addc_co_u32 %1, vcc, %6, %7, vcc\n\tv_addc_co_u32 %2, vcc, %6, %7, vcc\n\tv_addc_co_u32 %3, vcc, %6, %7, vcc\n\t", "vcc")
OPK(k_sub_sdwa, "v_sub_u32_sdwa %0, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1\n\tv_sub_u32_sdwa %1, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1\n\t"
                "v_sub_u32_sdwa %2, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1\n\tv_sub_u32_sdwa %3, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_1\n\t", "vcc")
OPK(k_bfi, "v_bfi_b32 %0, %6, %7, %1\n\tv_bfi_b32 %1, %6, %7, %2\n\tv_bfi_b32 %2, %6, %7, %3\n\tv_bfi_b32 %3, %6, %7, %0\n\t", "vcc")
OPK(k_ashr, "v_ashrrev_i32 %0, 31, %6\n\tv_ashrrev_i32 %1, 31, %7\n\tv_ashrrev_i32 %2, 31, %6\n\tv_ashrrev_i32 %3, 31, %7\n\t", "vcc")
OPK(k_med3, "v_med3_i32 %0, %6, 0, 1\n\tv_med3_i32 %1, %7, 0, 1\n\tv_med3_i32 %2, %6, 0, 1\n\tv_med3_i32 %3, %7, 0, 1\n\t", "vcc")
OPK(k_and_dpp, "v_and_b32_dpp %0, %6, %7" DPP "v_and_b32_dpp %1, %6, %7" DPP "v_and_b32_dpp %2, %6, %7" DPP "v_and_b32_dpp %3, %6, %7" DPP, "vcc")
OPK(k_mov_dpp_shr, "v_mov_b32_dpp %0, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %6 row_shr:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %6 row_shr:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %6 row_shr:4 row_mask:0xf bank_mask:0xf\n\t", "vcc")
OPK(k_mov_dpp_quad, "v_mov_b32_dpp %0, %6 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %6 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %6 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %6 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t", "vcc")
OPK(k_readlane, "v_readlane_b32 s20, %6, 3\n\tv_readlane_b32 s21, %7, 5\n\tv_readlane_b32 s22, %6, 7\n\tv_readlane_b32 s23, %7, 9\n\t", "s20", "s21", "s22", "s23")
OPK(k_add_sgpr, "v_add_u32 %0, s20, %6\n\tv_add_u32 %1, s21, %7\n\tv_add_u32 %2, s22, %6\n\tv_add_u32 %3, s23, %7\n\t", "s20", "s21", "s22", "s23")
OPK(k_addf64_sgpr, "v_add_f64 %4, %4, s[20:21]\n\tv_add_f64 %5, %5, s[22:23]\n\tv_add_f64 %4, %4, s[20:21]\n\tv_add_f64 %5, %5, s[22:23]\n\t", "s20", "s21", "s22", "s23")
OPK(k_cvt_f64, "v_cvt_f64_u32 %4, %6\n\tv_cvt_f64_u32 %5, %7\n\tv_cvt_f64_u32 %4, %6\n\tv_cvt_f64_u32 %5, %7\n\t", "vcc")
OPK(k_salu, "s_mov_b64 s[20:21], s[22:23]\n\ts_bfe_u32 s24, s25, 0x80008\n\ts_mov_b64 s[22:23], s[20:21]\n\ts_bfe_u32 s25, s24, 0x80008\n\t", "s20", "s21", "s22", "s23", "s24", "s25", "scc")
OPK(k_exec_add, "s_mov_b64 exec, s[20:21]\n\tv_add_f64 %4, %4, %8\n\ts_mov_b64 exec, s[22:23]\n\tv_add_f64 %4, %4, %8\n\t", "s20", "s21", "s22", "s23")
OPK(k_cmpx, "v_cmpx_gt_u32 vcc, %6, %7\n\tv_cmpx_le_u32 vcc, %6, %7\n\tv_cmpx_gt_u32 vcc, %6, %7\n\tv_cmpx_le_u32 vcc, %6, %7\n\t", "vcc")
OPK(k_lshl_add, "v_lshl_add_u32 %0, %6, 3, %7\n\tv_lshl_add_u32 %1, %6, 3, %7\n\tv_lshl_add_u32 %2, %6, 3, %7\n\tv_lshl_add_u32 %3, %6, 3, %7\n\t", "vcc")
OPK(k_pk_sub, "v_pk_sub_u16 %0, %6, %7\n\tv_pk_sub_u16 %1, %6, %7\n\tv_pk_sub_u16 %2, %6, %7\n\tv_pk_sub_u16 %3, %6, %7\n\t", "vcc")

template <typename K> static void run(const char* name, K kern, double* d_out, bool exec_fix = false) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wps = 8, grid = 256 * 4 * wps, reps = 1000;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out, 2, 5);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, d_out, reps, 5); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-34s %.2f cycles per instruction and SIMD\n", name, ms * 1e6 / ((double)wps * reps * 64) * 2.4);
}
int main() {
    double* d_out; CK(hipMalloc(&d_out, (size_t)256 * 4 * 8 * 64 * 8));
    run("v_sub_u32", k_sub, d_out);
    run("v_cndmask_b32 (vcc, never written)", k_cnd_vcc, d_out);
    run("v_cndmask_b32_e64 (sgpr pair)", k_cnd_sgpr, d_out);
    run("v_cmp_gt_u32 -> vcc", k_cmp, d_out);
    run("v_cmp + v_cndmask pairs", k_cmp_cnd, d_out);
    run("v_sub_co_u32", k_subco, d_out);
    run("v_sub_co_u32 + v_cndmask pairs", k_subco_cnd, d_out);
    run("v_addc_co_u32 (vcc in and out)", k_addc, d_out);
    run("v_sub_u32_sdwa", k_sub_sdwa, d_out);
    run("v_bfi_b32", k_bfi, d_out);
    run("v_ashrrev_i32", k_ashr, d_out);
    run("v_med3_i32", k_med3, d_out);
    run("v_lshl_add_u32", k_lshl_add, d_out);
    run("v_pk_sub_u16", k_pk_sub, d_out);
    run("v_and_b32_dpp row_newbcast", k_and_dpp, d_out);
    run("v_mov_b32_dpp row_shr", k_mov_dpp_shr, d_out);
    run("v_mov_b32_dpp quad_perm", k_mov_dpp_quad, d_out);
    run("v_readlane_b32", k_readlane, d_out);
    run("v_add_u32 with an SGPR operand", k_add_sgpr, d_out);
    run("v_add_f64 with an SGPR pair", k_addf64_sgpr, d_out);
    run("v_cvt_f64_u32", k_cvt_f64, d_out);
    run("SALU (s_mov_b64 / s_bfe)", k_salu, d_out);
    return 0;
}
