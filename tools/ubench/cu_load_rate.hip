// cu_load_rate.hip -- how fast can ONE CU pull an L2-resident operand panel, by load instruction and by pattern?
// The decode GEMMs (M = 256) move 295 KB per 64 x 64 tile through LDS-DMA and their loop runs at one 16 KiB k-tile per
// 0.33 us = 48 GB/s per CU whatever the ring depth and however many CUs are busy (tools/gemm_phases.py): is that the
// LDS-DMA path, the address pattern (8 rows x 128 B per wave instruction) or the L2?
//   variants: 0 LDS-DMA 16 B/lane, rows 2304 B apart (the GEMM's pattern)       1 LDS-DMA, fully contiguous 1 KiB pieces
//             2 global_load_dwordx4 into registers, GEMM pattern                 3 global_load_dwordx4, contiguous
//             4 LDS-DMA, contiguous 1 KiB pieces with the GEMM's XOR swizzle of the 16-byte chunks inside each 128-B row
//   every wave of a 256- or 512-thread block keeps DEPTH pieces in flight; grid = nblk blocks, every block reads the
//   same `panel_kb` KiB (L2 resident after the first pass) `reps` times.
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_load_rate cu_load_rate.hip ; run on an MI355X.  Tuning tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// piece p (1 KiB) of the panel: GEMM pattern = rows p*8 .. p*8+7 of a [rows][ld] matrix, 128 B of each row at byte
// column kcol; contiguous = bytes [p*1024, +1024)
template <int VAR, int DEPTH, int NW>
__global__ __launch_bounds__(NW * 64) void rate_kernel(const unsigned char* __restrict__ src, int npieces, int ld_bytes,
                                                        int ktiles, int reps, unsigned long long* out, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NW * DEPTH * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        // this wave's pieces: wave, wave + NW, ...; a piece index = (k-tile, row group)
        const int total = npieces * ktiles;
        uint4 regs[DEPTH];
        bool primed = false;
        for (int base = wave; base < total; base += NW * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int i = base + d * NW;
                if (i >= total) continue;
                if (primed) {  // DEPTH pieces in flight; the oldest one sits in slot d
                    wait_vmcnt<DEPTH - 1>();
                    if (VAR >= 2) acc ^= regs[d].x ^ regs[d].w;
                }
                const int kt = i / npieces, p = i - kt * npieces;
                const unsigned char* g;
                if (VAR == 4)  // contiguous 1 KiB, 16-byte chunks XOR-swizzled inside each 128-byte row (as the GEMM's source swizzle)
                    g = src + ((size_t)i * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
                else if (VAR == 5)  // 4 rows x 256 B
                    g = src + ((size_t)((p & 31) * 4 + (lane >> 4)) * ld_bytes + (kt * 2 + (p >> 5)) * 256 + (lane & 15) * 16);
                else if (VAR == 6)  // 2 rows x 512 B
                    g = src + ((size_t)((p & 63) * 2 + (lane >> 5)) * ld_bytes + (size_t)(kt >> 2) * 2048 + ((kt & 3) * 512) % 2048 + (lane & 31) * 16);
                else if (VAR == 7)  // GEMM pattern, swizzled chunks
                    g = src + ((size_t)(p * 8 + (lane >> 3)) * ld_bytes + kt * 128 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
                else if (VAR & 1)
                    g = src + ((size_t)i * 1024 + lane * 16);
                else
                    g = src + ((size_t)(p * 8 + (lane >> 3)) * ld_bytes + kt * 128 + (lane & 7) * 16);
                if (VAR < 2 || VAR >= 4)
                    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
                else
                    regs[d] = *(const uint4*)g;  // the compiler's own counted vmcnt keeps DEPTH loads in flight
            }
            primed = true;
        }
        wait_vmcnt<0>();
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    if (VAR < 2 || VAR >= 4) acc ^= *(volatile unsigned*)(smem + threadIdx.x * 4);
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int VAR, int DEPTH, int NW>
static void run(const char* name, const unsigned char* src, int rows, int ld_bytes, int ktiles, int nblk, unsigned long long* dout,
                unsigned* sink) {
    const int reps = 20, npieces = rows / 8;
    unsigned long long h[1024];
    double best = 1e30, mean = 0;
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL((rate_kernel<VAR, DEPTH, NW>), dim3(nblk), dim3(NW * 64), 0, 0, src, npieces, ld_bytes, ktiles, reps, dout, sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, dout, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double m = 0;
        for (int b = 0; b < nblk; ++b) m += (double)h[b];
        m /= nblk;
        if (m < best) best = m;
        mean = m;
    }
    const double bytes = (double)npieces * ktiles * 1024 * reps;
    printf("  %-44s %3d blocks x %d waves, %2d pieces/wave in flight: %6.1f GB/s per CU (%.2f us per 16 KiB)\n", name, nblk, NW, DEPTH,
           bytes / (best * 10.0), 16384.0 / (bytes / (best * 10.0)) / 1000.0);
    (void)mean;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int rows = 128, ld = 2304, ktiles = 18;  // a 64-row A panel + a 64-row W panel of K = 1152 bf16
    unsigned char* src;
    CK(hipMalloc(&src, (size_t)rows * ld + (1 << 20)));
    CK(hipMemset(src, 1, (size_t)rows * ld + (1 << 20)));
    unsigned long long* dout;
    CK(hipMalloc(&dout, 1024 * sizeof(unsigned long long)));
    unsigned* sink;
    CK(hipMalloc(&sink, 64));
    printf("one CU pulling an L2-resident 288 KiB panel (same panel for every block), wall clock per block:\n");
    for (int nblk : {8, 256}) {
        run<0, 4, 4>("LDS-DMA, 8 rows x 128 B pieces", src, rows, ld, ktiles, nblk, dout, sink);
        run<0, 8, 4>("LDS-DMA, 8 rows x 128 B pieces", src, rows, ld, ktiles, nblk, dout, sink);
        run<0, 8, 8>("LDS-DMA, 8 rows x 128 B pieces", src, rows, ld, ktiles, nblk, dout, sink);
        run<1, 8, 4>("LDS-DMA, contiguous 1 KiB pieces", src, rows, ld, ktiles, nblk, dout, sink);
        run<1, 4, 4>("LDS-DMA, contiguous 1 KiB pieces", src, rows, ld, ktiles, nblk, dout, sink);
        run<4, 4, 4>("LDS-DMA, contiguous + XOR-swizzled chunks", src, rows, ld, ktiles, nblk, dout, sink);
        run<7, 4, 4>("LDS-DMA, 8 rows x 128 B, swizzled chunks", src, rows, ld, ktiles, nblk, dout, sink);
        run<5, 4, 4>("LDS-DMA, 4 rows x 256 B pieces", src, rows, ld, ktiles, nblk, dout, sink);
        run<6, 4, 4>("LDS-DMA, 2 rows x 512 B pieces", src, rows, ld, ktiles, nblk, dout, sink);
    }
    return 0;
}
