// What do the DPP forms of k_center's replay step actually compute on gfx950?  Dumps per-lane results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k(int* out, double* dout) {
    const int lane = threadIdx.x;
    int x = 100 + lane, y = 0, p = 1000 + lane, t = 0, u = 0, hi = 0;
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "+v"(y) : "v"(x));
    out[lane] = y;
    // t = p - x[3]
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "=v"(t) : "v"(x), "v"(p));
    out[64 + lane] = t;
    // borrow of t - m[3], m = lane value 900 + 2 lane
    int m = 890 + 2 * lane;
    unsigned long long vcc_out = 0;
    asm volatile("s_nop 4\n\tv_subrev_co_u32_dpp %0, vcc, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4\n\ts_mov_b64 %1, vcc\n\t"
                 : "=v"(u), "=s"(vcc_out) : "v"(m), "v"(t) : "vcc");
    out[128 + lane] = u;
    out[192 + lane] = (int)((vcc_out >> lane) & 1);
    double val = 1.0 / (3 + lane), one = (lane & 1) ? 1.0 : 0.0, acc = 10.0;
    asm volatile("s_nop 4\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "+v"(acc) : "v"(val), "v"(one));
    dout[lane] = acc;
    // non-rev forms: expect dpp(src0) - src1
    int t2 = 0, u2 = 0;
    unsigned long long vcc2 = 0;
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "=v"(t2) : "v"(x), "v"(p));
    out[256 + lane] = t2;
    int tt = 897 + (lane & 15);   // as if t = p - x[3]
    asm volatile("s_nop 4\n\tv_sub_co_u32_dpp %0, vcc, %2, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4\n\ts_mov_b64 %1, vcc\n\t"
                 : "=v"(u2), "=s"(vcc2) : "v"(m), "v"(tt) : "vcc");
    out[320 + lane] = u2;
    out[384 + lane] = (int)((vcc2 >> lane) & 1);
    double mv = 0.0;
    asm volatile("s_nop 4\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "+v"(mv) : "v"(val));
    dout[64 + lane] = mv;
}
int main() {
    int* d; double* dd; CK(hipMalloc(&d, 448 * 4)); CK(hipMalloc(&dd, 128 * 8));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dd);
    int h[448]; double hd[128];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
    const char* names[4] = {"mov_dpp x[3] (expect 103,119,135,151 per row)", "p - x[3] (expect 1000+l - (103+16r))", "t - m[3] (m[3]=896+32r)", "vcc (t < m[3])"};
    for (int k2 = 0; k2 < 4; ++k2) { printf("%s\n", names[k2]); for (int l = 0; l < 64; ++l) printf("%d%s", h[k2 * 64 + l], (l & 15) == 15 ? "\n" : " "); }
    const char* names2[3] = {"v_sub_u32_dpp t, x, p: dpp(x)[3] - p?  (103+16r - 1000 - l)", "v_sub_co_u32_dpp u, vcc, m, tt: dpp(m)[3] - tt? (896+32r - 897 - li)", "its vcc: tt > m[3]?"};
    for (int k2 = 0; k2 < 3; ++k2) { printf("%s\n", names2[k2]); for (int l = 0; l < 64; ++l) printf("%d%s", h[256 + k2 * 64 + l], (l & 15) == 15 ? "\n" : " "); }
    printf("fmac: 10 + val[3]*one  (val[3] = 1/(6+16r): .1667 .0455 .0263 .0185; odd lanes only)\n");
    for (int l = 0; l < 64; ++l) printf("%.4f%s", hd[l], (l & 15) == 15 ? "\n" : " ");
    printf("mov_b64_dpp val[3]\n");
    for (int l = 0; l < 64; ++l) printf("%.4f%s", hd[64 + l], (l & 15) == 15 ? "\n" : " ");
    return 0;
}
