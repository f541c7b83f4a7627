/* vq_argmin.c -- CPU ORACLE (test infrastructure only) for the VQ codebook search.
 *
 * Plain-C restatement of reference code/models/lib/quantizer.py:38-45
 *     d = sum(z^2) + sum(E^2) - 2 z.E^T ;  idx = argmin_j d   (first index on ties)
 * with the SAME float32 evaluation order as the HIP kernel (csrc/vq.hip): dot_j and ||e_j||^2 are
 * k-ascending fmaf chains, ||z||^2 is the 64-lane butterfly sum of (z[l]^2 + z[l+64]^2), and
 * d_j = (zz + ee_j) - 2*dot_j.  With identical order the DISTANCES (not only the indices) of the GPU
 * kernel can be compared bit for bit.  Only tests/ may load this library.
 *
 * build: oracle/build_oracle.py  ->  oracle/_build/libvqoracle.so
 */
#include <math.h>
#include <stdint.h>

static float wave_butterfly_sum(const float* lane_vals /*64*/) {
    float v[64];
    for (int i = 0; i < 64; ++i) v[i] = lane_vals[i];
    for (int o = 32; o > 0; o >>= 1) {
        float w[64];
        for (int i = 0; i < 64; ++i) w[i] = v[i] + v[i ^ o];
        for (int i = 0; i < 64; ++i) v[i] = w[i];
    }
    return v[0];
}

/* z [N,128], E [512,128] -> idx [N], best_d [N] (optional), margin [N] (optional) */
void vq_argmin_oracle(const float* z, int N, const float* E, int32_t* idx, float* best_d, float* margin) {
    float ee[512];
    for (int j = 0; j < 512; ++j) {
        float s = 0.f;
        for (int k = 0; k < 128; ++k) s = fmaf(E[j * 128 + k], E[j * 128 + k], s);
        ee[j] = s;
    }
    for (int n = 0; n < N; ++n) {
        const float* zn = z + (long)n * 128;
        float lanes[64];
        for (int l = 0; l < 64; ++l) {
            /* the kernel evaluates zr0*zr0 + zr1*zr1, which hipcc contracts to fma(zr1, zr1, zr0*zr0) */
            lanes[l] = fmaf(zn[l + 64], zn[l + 64], zn[l] * zn[l]);
        }
        const float zz = wave_butterfly_sum(lanes);
        float bd = 3.0e38f, bd2 = 3.0e38f;
        int bi = 0x7fffffff;
        for (int j = 0; j < 512; ++j) {
            float dot = 0.f;
            for (int k = 0; k < 128; ++k) dot = fmaf(zn[k], E[j * 128 + k], dot);
            const float d = fmaf(-2.0f, dot, zz + ee[j]);
            if (d < bd || (d == bd && j < bi)) {
                bd2 = bd;
                bd = d;
                bi = j;
            } else if (d < bd2) {
                bd2 = d;
            }
        }
        idx[n] = bi;
        if (best_d) best_d[n] = bd;
        if (margin) margin[n] = bd2 - bd;
    }
}
