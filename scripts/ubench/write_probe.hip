// write_probe.hip -- what does the WRITE_SIZE counter (and HBM) see for the output pattern of the tile
// kernel?  (scratch experiment; run under `rocprofv3 --pmc WRITE_SIZE` and `--pmc FETCH_SIZE`)
//
// Every workgroup writes `strips_per_wg` strips of `len` 8-byte elements (one 256-thread pass per 256
// elements, lane i -> element i: the epilogue of k_hist_point).  Modes:
//   0  strips back to back, first strip 64-byte aligned, len a multiple of 8      (all lines whole)
//   1  strips back to back, len = 150 (strips start at odd 8-byte offsets; neighbouring strips share lines,
//      and are written by DIFFERENT workgroups)
//   2  as 1, written in descending order inside a strip (step = -1, '-' strand chains)
//   3  strips of 150 separated by gaps of 1 element (each strip's end lines are partial and private)
//   4  as 1, but consecutive strips handled by the SAME workgroup one after the other
// Useful bytes per launch = nstrips * len * 8 in every mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_strips(unsigned long long *dst, long nstrips, int len, int pitch, int mode, int per_wg) {
    for (int k = 0; k < per_wg; ++k) {
        // mode 4: a workgroup owns per_wg consecutive strips; otherwise strips are dealt round-robin
        const long s = mode == 4 ? (long)blockIdx.x * per_wg + k : (long)k * gridDim.x + blockIdx.x;
        if (s >= nstrips) return;
        unsigned long long *p = dst + s * (long)pitch;
        for (int i = threadIdx.x; i < len; i += 256) {
            const int j = mode == 2 ? len - 1 - i : i;
            p[j] = (unsigned long long)(s + j);
        }
    }
}

int main(int argc, char **argv) {
    const long total = argc > 1 ? atol(argv[1]) : (1l << 27);   // elements (1 GiB)
    unsigned long long *d;
    hipMalloc(&d, (size_t)total * 8 + 4096);
    hipMemset(d, 0, (size_t)total * 8 + 4096);
    hipDeviceSynchronize();
    for (int mode = 0; mode <= 4; ++mode) {
        const int len = mode == 0 ? 152 : 150;
        const int pitch = mode == 3 ? len + 1 : len;
        const long nstrips = total / pitch;
        const int per_wg = 16;
        const unsigned grid = (unsigned)((nstrips + per_wg - 1) / per_wg);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_strips, dim3(grid), dim3(256), 0, 0, d, nstrips, len, pitch, mode, per_wg);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("mode %d: %ld strips x %d elements, useful %.1f MB, %.3f ms, %.0f GB/s useful\n", mode, nstrips, len,
               nstrips * (double)len * 8 / 1e6, ms, nstrips * (double)len * 8 / ms / 1e6);
    }
    return 0;
}
