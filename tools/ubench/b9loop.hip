// b9loop -- does the bf16 x 9 frontend's inner step overlap with itself?  (round 4; follow-up of pipes3.hip / occ.hip)
// The narrow bf16 x 9 frontend (csrc/kernel_front_b9.hip) runs two waves per SIMD, each a sequence of steps
//     [split 8 fp32 values into 3 x 4 registers of bf16 pairs: ~36 plain VALU] [3 x (2 ds_read_b128 of A fragments + 6 MFMA 16x16x32 bf16)]
// = ~150 VALU cycles + 288 matrix-pipe cycles per wave.  If the VALU of one wave ran beside the MFMAs of the other, two waves would take
// ~576 cycles per step pair (matrix bound); the kernel behaves as if they took the sum.  This program runs exactly that step in a loop,
// with 1 and 2 waves per SIMD and with parts removed, every wave timing itself with s_memtime:
//   mode 0 full step   1 no split   2 no fragment reads   3 MFMAs only   4 split only   5 full step, [MFMA][split] order (the split of the
//   NEXT step behind the MFMAs of this one: independent instructions, same counts)
//   build + run ON the GPU box:  hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops tools/ubench/b9loop.hip -o /tmp/b9loop && /tmp/b9loop
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));
using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using bf2 = __bf16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_b(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split3(float x0, float x1, unsigned &p0, unsigned &p1, unsigned &p2) {
#pragma clang fp contract(off)
    f32x2 r{x0, x1};
    unsigned out[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bf2 h = __builtin_convertvector(r, bf2);
        const unsigned bits = __builtin_bit_cast(unsigned, h);
        out[k] = bits;
        const f32x2 back{__uint_as_float(bits << 16), __uint_as_float(bits & 0xffff0000u)};
        r = r - back;
    }
    p0 = out[0]; p1 = out[1]; p2 = out[2];
}
#ifdef B9LOOP_TRUNC
// pieces by TRUNCATION: p0 = the top 16 bits of x, p1 = the top 16 bits of x - p0, p2 = x - p0 - p1 (<= 8 significant bits: its top 16 bits
// hold all of it); x = p0 + p1 + p2 exactly as with rounding, the chains are two levels shallower and need no conversion instruction
__device__ __forceinline__ void split3(float x0, float x1, unsigned &p0, unsigned &p1, unsigned &p2);
__device__ __forceinline__ void split3t(float x0, float x1, unsigned &p0, unsigned &p1, unsigned &p2) {
#pragma clang fp contract(off)
    const unsigned a0 = __float_as_uint(x0) & 0xffff0000u, b0 = __float_as_uint(x1) & 0xffff0000u;
    const float ra = x0 - __uint_as_float(a0), rb = x1 - __uint_as_float(b0);
    const unsigned a1 = __float_as_uint(ra) & 0xffff0000u, b1 = __float_as_uint(rb) & 0xffff0000u;
    const float sa = ra - __uint_as_float(a1), sb = rb - __uint_as_float(b1);
    p0 = __builtin_amdgcn_perm(b0, a0, 0x07060302u);          // (hi half of b0) : (hi half of a0)
    p1 = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
#define split3 split3t
#endif
__device__ __forceinline__ void split_step(u32x4 (&bp)[3], const float (&v)[8]) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned p0, p1, p2;
        split3(v[2 * d], v[2 * d + 1], p0, p1, p2);
        bp[0][d] = p0; bp[1][d] = p1; bp[2][d] = p2;
    }
}

struct Rec { unsigned long long cycles; unsigned hwid; };

template <int MODE>
__global__ void __launch_bounds__(256, 2) loop_kernel(int n, Rec *rec, float *sink) {
    __shared__ __attribute__((aligned(16))) unsigned lds[24 * 1024 / 4];      // one 24 KiB unit of "A fragments"
    for (int i = threadIdx.x; i < 24 * 1024 / 4; i += 256) lds[i] = 0x3f803f80u + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool odd = __builtin_amdgcn_s_getreg((31 << 11) | 4) & 1;
    f32x4 acc[2] = {};
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 1.0f + 1e-3f * (lane + e);
    u32x4 bp[3], bq[3];
    split_step(bp, v);
    const u32x4 *frag = reinterpret_cast<const u32x4 *>(lds) + lane;
    u32x4 c0 = frag[0], c1 = frag[64];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (MODE == 10 || MODE == 11) {                   // the full step with issue priorities: high while multiplying (10) / while splitting (11)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
            __builtin_amdgcn_s_setprio(MODE == 11 ? 3 : 0);
            split_step(bp, v);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(MODE == 10 ? 3 : 0);
#pragma unroll
            for (int pa = 0; pa < 3; ++pa) {
                const u32x4 n0 = frag[(2 * (pa + 1)) * 64], n1 = frag[(2 * (pa + 1) + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    acc[0] = mfma_b(c0, bp[pb], acc[0]);
                    acc[1] = mfma_b(c1, bp[pb], acc[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0; c1 = n1;
            }
            continue;
        }
        if (MODE == 0 || MODE == 2) {                     // the operand values change every step, as in the kernel
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
            split_step(bp, v);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 6) {                                  // split as usual, but the MFMAs read the same registers every time
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
            split_step(bq, v);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                acc[0] = mfma_b(c0, bp[0], acc[0]);
                acc[1] = mfma_b(c1, bp[0], acc[1]);
            }
            acc[0][0] += __uint_as_float(bq[0][0] ^ bq[1][1] ^ bq[2][2]) * 1e-30f;
            continue;
        }
        if (MODE == 7) {                                  // pure roles: even wave slots multiply, odd wave slots split
            if (odd) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
                split_step(bq, v);
                acc[0][0] += __uint_as_float(bq[0][0] ^ bq[1][1] ^ bq[2][2]) * 1e-30f;
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    acc[0] = mfma_b(c0, bp[k % 3], acc[0]);
                    acc[1] = mfma_b(c1, bp[k % 3], acc[1]);
                }
            }
            continue;
        }
        if (MODE == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
            split_step(bp, v);
            acc[0][0] += __uint_as_float(bp[0][0] ^ bp[1][1] ^ bp[2][2]);
            continue;
        }
        if (MODE == 8 || MODE == 9) {
            // ONE scheduling region: this step's 18 MFMAs (operands in bp) and the NEXT step's split (into bq), interleaved by
            // sched_group_barrier: [1 MFMA][K VALU] x 18 (K = 3: mode 8, K = 2 + the rest at the end: mode 9)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
            split_step(bq, v);
#pragma unroll
            for (int pa = 0; pa < 3; ++pa)
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    acc[0] = mfma_b(c0, bp[pb], acc[0]);
                    acc[1] = mfma_b(c1, bp[pb], acc[1]);
                }
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, MODE == 8 ? 3 : 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) bp[k] = bq[k];
            continue;
        }
#pragma unroll
        for (int pa = 0; pa < 3; ++pa) {
            u32x4 n0 = c0, n1 = c1;
            if (MODE == 0 || MODE == 1 || MODE == 5) {
                n0 = frag[(2 * (pa + 1)) * 64];                 // immediate offsets, as in the kernel
                n1 = frag[(2 * (pa + 1) + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                acc[0] = mfma_b(c0, bp[pb], acc[0]);
                acc[1] = mfma_b(c1, bp[pb], acc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = n0; c1 = n1;
        }
        if (MODE == 5) {                                   // the NEXT step's pieces, behind this step's MFMAs
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], 0.999f, 1e-4f);
            split_step(bq, v);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 3; ++k) bp[k] = bq[k];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
        const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        rec[wv].cycles = t1 - t0;
        rec[wv].hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    const float s = acc[0][0] + acc[0][3] + acc[1][1] + v[0];
    if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void run(const char *what, int wgs, int n, Rec *d_rec, float *d_sink) {
    std::vector<Rec> h(wgs * 4);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, n, d_rec, d_sink);       // warm-up
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(wgs), dim3(256), 0, 0, n, d_rec, d_sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_rec, h.size() * sizeof(Rec), hipMemcpyDeviceToHost);
    std::vector<double> c, ce, co;
    for (auto &r : h) {
        c.push_back((double)r.cycles / n);
        ((r.hwid & 1) ? co : ce).push_back((double)r.cycles / n);
    }
    std::sort(c.begin(), c.end());
    std::sort(ce.begin(), ce.end());
    std::sort(co.begin(), co.end());
    printf("  %-52s cycles per step and wave: median %7.1f  p10 %7.1f  p90 %7.1f", what, c[c.size() / 2], c[c.size() / 10], c[c.size() * 9 / 10]);
    if (MODE == 7 && !ce.empty() && !co.empty()) printf("   (even slots %7.1f, odd slots %7.1f)", ce[ce.size() / 2], co[co.size() / 2]);
    printf("\n");
}

int main() {
    Rec *d_rec;
    float *d_sink;
    hipMalloc(&d_rec, 2048 * 4 * sizeof(Rec));
    hipMalloc(&d_sink, 64);
    const int n = 20000;
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int wgs = 256 * per_cu;
        printf("%d workgroup(s) of 4 waves per CU = %d wave(s) per SIMD (18 MFMAs = 288 matrix-pipe cycles per step and wave)\n", per_cu, per_cu);
        run<3>("MFMAs only", wgs, n, d_rec, d_sink);
        run<4>("split only", wgs, n, d_rec, d_sink);
        run<1>("fragment reads + MFMAs", wgs, n, d_rec, d_sink);
        run<2>("split + MFMAs", wgs, n, d_rec, d_sink);
        run<0>("full step: split, fragment reads, MFMAs", wgs, n, d_rec, d_sink);
        run<5>("full step, next step's split behind the MFMAs", wgs, n, d_rec, d_sink);
        run<6>("split + MFMAs on constant operand registers", wgs, n, d_rec, d_sink);
        run<7>("roles: even slots MFMAs only, odd slots split only", wgs, n, d_rec, d_sink);
        run<8>("next split interleaved: [1 MFMA][3 VALU] x 18", wgs, n, d_rec, d_sink);
        run<9>("next split interleaved: [1 MFMA][2 VALU] x 18", wgs, n, d_rec, d_sink);
        run<10>("full step, s_setprio 3 while multiplying", wgs, n, d_rec, d_sink);
        run<11>("full step, s_setprio 3 while splitting", wgs, n, d_rec, d_sink);
    }
    return 0;
}
