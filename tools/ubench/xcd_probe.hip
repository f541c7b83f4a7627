// xcd_probe.hip -- measurements that decide the layout of the decode-step chain kernels:
//   1. census      : which XCD does block b run on (HW_REG_XCC_ID) when the grid is one block per CU?
//   2. xbarrier    : cost of a barrier among the blocks of ONE XCD (agent-scope counter, sc1 polling)
//   3. handoff     : plain stores + vmcnt(0) + arrive  ->  sc1 loads on another CU of the same XCD: any stale word?
//   4. wstream     : every XCD streams the same weight buffer (each of its CUs 1/32 of it) vs every CU a unique
//                    1/256 slice -- what the memory side sustains when the weights are replicated 8x into the L2s
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_probe xcd_probe.hip ; run on an MI355X.  Tuning tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint4 ld16_sc1(const void* p) {
    uint4 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}

struct Group {
    unsigned ticket;   // local index hand-out
    unsigned arrive;   // monotonic barrier counter
    unsigned pad[30];
};

// returns false on timeout
__device__ __forceinline__ bool xcd_barrier(Group* g, unsigned& target, unsigned members, unsigned* err) {
    __shared__ int s_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = true;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&g->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target += members;
        unsigned spins = 0;
        while ((int)(ld_sc1(&g->arrive) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000u) { atomicAdd(err, 1u); ok = false; break; }
        }
        s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    return s_ok != 0;
}

__global__ void census_kernel(unsigned* out) {
    extern __shared__ unsigned char dyn[];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = xcc_id();
        out[blockIdx.x * 2 + 1] = hw_id();
    }
    dyn[threadIdx.x] = 0;
}

__global__ void busy_kernel(float* p, int n) {
    float acc = 0.f;
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) acc += p[i];
    if (acc == 12345.f) p[0] = acc;
}

// mode 0: barrier only; mode 1: hand-off with sc1 reader loads; mode 2: hand-off with plain reader loads
__global__ void xbar_kernel(Group* groups, unsigned* err, unsigned* mism, float* payload, int iters, int mode,
                            unsigned* members_out) {
    extern __shared__ unsigned char dyn[];
    __shared__ unsigned s_g, s_li;
    if (threadIdx.x == 0) {
        const unsigned g = xcc_id();
        s_g = g;
        s_li = __hip_atomic_fetch_add(&groups[g].ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned g = s_g, li = s_li;
    Group* G = &groups[g];
    const unsigned members = 32;  // expected; the host checks the census first
    if (li >= members) { if (threadIdx.x == 0) atomicAdd(err, 1000u); return; }
    unsigned target = 0;
    float* mine = payload + ((size_t)g * 32 + li) * 512;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (mode) {
            for (int i = threadIdx.x; i < 512; i += blockDim.x) mine[i] = (float)(it * 64 + li);
        }
        if (!xcd_barrier(G, target, members, err)) return;
        if (mode) {
            // read every peer's record (32 x 2 KB)
            for (int i = threadIdx.x; i < 32 * 128; i += blockDim.x) {
                const int peer = i >> 7, off = (i & 127) * 4;
                const float* src = payload + ((size_t)g * 32 + peer) * 512 + off;
                float4 v;
                if (mode == 1) {
                    const uint4 u = ld16_sc1(src);
                    v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                } else {
                    v = *(const float4*)src;
                }
                const float want = (float)(it * 64 + peer);
                bad += (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want);
            }
            if (!xcd_barrier(G, target, members, err)) return;
        }
    }
    if (bad) atomicAdd(mism, bad);
    if (threadIdx.x == 0 && li == 0) members_out[g] = ld_sc1(&G->ticket);
    dyn[threadIdx.x] = 0;
}

// weight stream.  replicated = 1: block (g, li) streams slice li of 32 of the buffer; 0: block b streams slice b of 256
__global__ void wstream_kernel(const unsigned char* w, size_t bytes, int passes, int replicated, int nt, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // 8 waves x 8 slots x 1 KiB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned g = blockIdx.x & 7, li = blockIdx.x >> 3;
    const size_t slice = replicated ? bytes / 32 : bytes / 256;
    const unsigned char* base = w + (replicated ? (size_t)li * slice : (size_t)blockIdx.x * slice);
    (void)g;
    const size_t per_wave = slice / nw;
    const unsigned char* p = base + (size_t)wave * per_wave + lane * 16;
    const int n = (int)(per_wave / 1024);
    unsigned char* myring = ring + wave * 8 * 1024;
    for (int ps = 0; ps < passes; ++ps) {
        for (int i = 0; i < n; ++i) {
            if (nt)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(p + (size_t)i * 1024), (lds_void_t*)(myring + (i & 7) * 1024), 16, 0, 2);
            else
                __builtin_amdgcn_global_load_lds((glb_void_t*)(p + (size_t)i * 1024), (lds_void_t*)(myring + (i & 7) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && ring[5] == 123) sink[0] = 1;
}

// register-path variant (16-B global loads, 8 in flight per lane)
__global__ void wstream_reg_kernel(const unsigned char* w, size_t bytes, int passes, int replicated, unsigned* sink) {
    extern __shared__ unsigned char dyn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const unsigned li = blockIdx.x >> 3;
    const size_t slice = replicated ? bytes / 32 : bytes / 256;
    const unsigned char* base = w + (replicated ? (size_t)li * slice : (size_t)blockIdx.x * slice);
    const size_t per_wave = slice / nw;
    const uint4* p = (const uint4*)(base + (size_t)wave * per_wave) + lane;
    const int n = (int)(per_wave / 1024);
    unsigned acc = 0;
    for (int ps = 0; ps < passes; ++ps) {
        for (int i = 0; i + 8 <= n; i += 8) {
            uint4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = p[(size_t)(i + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= r[u].x ^ r[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    dyn[threadIdx.x] = 0;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d\n", prop.name, prop.multiProcessorCount);
    const int NB = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t LDS_BIG = 100 * 1024;  // forces one block per CU

    // ---- 1. census
    unsigned* d_out;
    CK(hipMalloc(&d_out, NB * 8));
    float* d_busy;
    const int nbusy = 64 << 20;
    CK(hipMalloc(&d_busy, (size_t)nbusy * 4));
    CK(hipMemset(d_busy, 0, (size_t)nbusy * 4));
    CK(hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG));
    for (int variant = 0; variant < 3; ++variant) {
        const int threads = variant == 2 ? 512 : 256;
        if (variant == 1) hipLaunchKernelGGL(busy_kernel, dim3(2048), dim3(256), 0, 0, d_busy, nbusy);
        hipLaunchKernelGGL(census_kernel, dim3(NB), dim3(threads), LDS_BIG, 0, d_out);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(NB * 2);
        CK(hipMemcpy(h.data(), d_out, NB * 8, hipMemcpyDeviceToHost));
        int cnt[16] = {0}, mod_ok = 0;
        for (int b = 0; b < NB; ++b) {
            cnt[h[2 * b] & 15]++;
            mod_ok += (h[2 * b] == (unsigned)(b & 7));
        }
        printf("census variant %d (%d thr%s): per-XCD blocks", variant, threads, variant == 1 ? ", behind a busy kernel" : "");
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf(" ; block b on XCD b%%8: %d of %d\n", mod_ok, NB);
    }

    // ---- 2/3. XCD barrier + hand-off
    Group* d_groups;
    unsigned *d_err, *d_mism, *d_members;
    float* d_payload;
    CK(hipMalloc(&d_groups, sizeof(Group) * 8));
    CK(hipMalloc(&d_err, 4));
    CK(hipMalloc(&d_mism, 4));
    CK(hipMalloc(&d_members, 32));
    CK(hipMalloc(&d_payload, 8 * 32 * 512 * 4));
    CK(hipFuncSetAttribute((const void*)xbar_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG));
    for (int mode = 0; mode < 3; ++mode) {
        for (int threads = 256; threads <= 512; threads += 256) {
            const int iters = 2000;
            CK(hipMemset(d_groups, 0, sizeof(Group) * 8));
            CK(hipMemset(d_err, 0, 4));
            CK(hipMemset(d_mism, 0, 4));
            CK(hipMemset(d_payload, 0, 8 * 32 * 512 * 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(xbar_kernel, dim3(NB), dim3(threads), LDS_BIG, 0, d_groups, d_err, d_mism, d_payload, iters,
                               mode, d_members);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            unsigned err, mism, mem[8];
            CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&mism, d_mism, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(mem, d_members, 32, hipMemcpyDeviceToHost));
            const float us = time_ms(e0, e1) * 1000.f / iters;
            printf("xbar mode %d (%s) %d thr: %.2f us per iteration (%s), err %u, mismatched words %u, members %u %u %u %u %u %u %u %u\n",
                   mode, mode == 0 ? "barrier only" : (mode == 1 ? "2KB/blk handoff, sc1 reads" : "2KB/blk handoff, plain reads"),
                   threads, us, mode ? "write + barrier + read 64 KB + barrier" : "one barrier", err, mism, mem[0], mem[1],
                   mem[2], mem[3], mem[4], mem[5], mem[6], mem[7]);
        }
    }

    // ---- 4. weight stream
    const size_t WB = (size_t)128 << 20;
    unsigned char* d_w;
    unsigned* d_sink;
    CK(hipMalloc(&d_w, WB));
    CK(hipMemset(d_w, 1, WB));
    CK(hipMalloc(&d_sink, 4));
    CK(hipFuncSetAttribute((const void*)wstream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG));
    CK(hipFuncSetAttribute((const void*)wstream_reg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BIG));
    for (size_t bytes : {WB, WB / 4, WB / 16}) {
        for (int replicated = 1; replicated >= 0; --replicated) {
            for (int kind = 0; kind < 3; ++kind) {  // 0 lds-dma, 1 lds-dma nt, 2 register loads
                const int passes = replicated ? 10 : 40;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (kind < 2)
                        hipLaunchKernelGGL(wstream_kernel, dim3(NB), dim3(512), LDS_BIG, 0, d_w, bytes, passes, replicated, kind, d_sink);
                    else
                        hipLaunchKernelGGL(wstream_reg_kernel, dim3(NB), dim3(512), LDS_BIG, 0, d_w, bytes, passes, replicated, d_sink);
                    CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                }
                const double us = time_ms(e0, e1) * 1000.0 / passes;
                const double ingest = (replicated ? 8.0 : 1.0) * (double)bytes;
                printf("wstream %4zu MB %s %-10s: %.1f us per pass, L2-side ingest %.2f TB/s, per-CU %.1f GB/s\n", bytes >> 20,
                       replicated ? "replicated-8x" : "unique       ", kind == 0 ? "lds-dma" : (kind == 1 ? "lds-dma-nt" : "registers"),
                       us, ingest / us / 1e6, ingest / NB / us / 1e3);
            }
        }
    }
    printf("done\n");
    return 0;
}
