// kernel_front.hip -- the time-parallel part of the Silero-VAD hot path as ONE fused kernel:
//
//   PCM (fp32 or int16, HBM) -> framing + right reflect pad -> Hann window -> 4 x real FFT
//   magnitude -> 4 x ReLU(Conv1d k=3) -> W_ih * feat + (b_ih + b_hh)  => gx (LSTM input gates)
//
// (reference: JIT!/vad/model/vad_annotator.py:58-67 framing, JIT!/vad/utils/pytorch_stft.py:17-34
//  STFT, JIT!/vad/utils/model_utils.py:19-25 encoder, the W_ih half of aten::lstm_cell
//  JIT!/torch/nn/modules/rnn.py:69.)
//
// Design (gfx950 / CDNA4, fp32 everywhere -- SURVEY.md section 0.4):
//   * one WAVE owns 16 chunks (16 streams x one time step) end to end; lane (g, j) = (l>>4, l&15),
//     column j of every MFMA is chunk j.  A workgroup is 4 such waves; two workgroups share a CU
//     so that one's VALU (FFT) phases overlap the other's MFMA phases.
//   * STFT is a 4-lane cooperative real FFT, not the reference's dense 258x256 DFT-basis conv:
//     the 4 lanes of a chunk each load a 2Q-sample slice of the frame, do a radix-4 butterfly
//     ACROSS lanes (two wave shuffles), then a Q-point complex FFT entirely in registers, then the
//     real-FFT split.  Lane group g ends up holding bins 4k'+P[g] -- any fixed bin->(register,
//     group) placement is acceptable because ...
//   * ... every matrix product runs on v_mfma_f32_16x16x4_f32 with activations kept in registers
//     in "chain layout" (layout.hpp): the D fragment of one layer IS the B operand of the next,
//     the k-order permutation being folded into the packed weights on the host.  Activations
//     never touch LDS or HBM between the FFT and the gx store.
//   * weights (0.65 MB packed, L2 resident) stream global -> LDS with global_load_lds_dwordx4
//     (no VGPR round trip) through a 3-slot ring of 16 KiB units shared by the 4 waves, walked by a
//     static schedule, and reach the MFMA A operand by ds_read_b128 one step ahead of their use --
//     across unit and segment boundaries, so that neither the LDS latency nor the DMA wait nor the
//     workgroup barrier (placed in the middle of the previous unit) is exposed at a boundary.
//   * zero-padding taps of the convs are skipped (enc0 10/12, enc1 5/6, enc2 2/3, enc3 1/3 taps).
#include <hip/hip_runtime.h>

#include "fft_wave.hpp"

namespace vad {
namespace {

// ---- build-time knobs (tools/variants.py builds A/B variants; the defaults are the product) -------
#ifndef VAD_SLOT_BLOCKS
#define VAD_SLOT_BLOCKS 16       // 1-KiB blocks per ring slot (one unit = up to this many blocks); 8 / 16 / 24 measured:
                                 // 5.46 / 5.36 / 5.49 ms per C2 launch (profiles/r02a_fp32_ablations.md)
#endif
#ifndef VAD_NYQ_VALU
#define VAD_NYQ_VALU 1           // 1: the Nyquist bin (enc0's 129th / 65th input channel, alone in a 9th / 5th k-group of the
#endif                           // image) is applied as an exact fp32 rank-1 VALU update from Tab::w_nyq instead of an MFMA
                                 // step whose other 3 lane groups multiply zeros: -80 MFMAs, -10 ring units per tile
#ifndef VAD_STAGGER
#define VAD_STAGGER 0            // x 8128 cycles of start delay for odd wave slots (first workgroups)
#endif

#ifndef VAD_RING_SLOTS
#define VAD_RING_SLOTS 3         // 3 (product): "seamless" pipeline -- a static schedule of the whole weight stream (make_sched), the
#endif                           //    workgroup barrier for unit u+1 in the MIDDLE of unit u, fragment reads carried across unit and
                                 //    segment boundaries (gemm_seg).  Needs uniform 16-block units (VAD_SLOT_BLOCKS 16, VAD_NYQ_VALU 1).
                                 // 2: round-1 form: a unit is requested when the previous one starts being consumed, awaited with
                                 //    vmcnt(0) + barrier at the unit boundary (5.27 vs 5.17 ms per C2 launch)
constexpr int kSlotBlocks = VAD_SLOT_BLOCKS;
constexpr int kRingSlotFloats = kSlotBlocks * 256;   // one unit: whole k-groups of one segment
constexpr int kRingSlots = VAD_RING_SLOTS;
static_assert(kRingSlots == 2 || (kRingSlots == 3 && kSlotBlocks == 16 && VAD_NYQ_VALU),
              "the 3-slot ring assumes that every unit is exactly 16 blocks");

// Program order of the weight stream as ring units (float offsets into the packed image): the kernel walks its
// segments in exactly this order; gemm_seg() cross-checks every unit it consumes against it (3-slot ring only).
struct Sched {
    int n;
    int off[128];
};
constexpr Sched make_sched(int Q) {
    using namespace vadl;
    constexpr int order[] = {E0T1, E0T2, E1T1,                    // enc0 frame 0            -> enc1 out 0 tap 1
                             E0T0, E0T1, E0T2, E1T2, E1T0,        // enc0 frame 1            -> out 0 tap 2, out 1 tap 0
                             E0T0, E0T1, E0T2, E1T1,              // enc0 frame 2            -> out 1 tap 1
                             E0T0, E0T1, E1T2,                    // enc0 frame 3            -> out 1 tap 2
                             E2T1, E2T2, E3T1, IH0, IH1, IH2, IH3};
    Sched sc{};
    int n = 0;
    for (int seg : order) {
        const int M = seg_mblocks(seg);
        const int KS = seg <= E0T2 ? Q : seg_ksteps(seg, Q);      // enc0: the Nyquist k-group is not streamed
        const int KG = (KS + 3) / 4, UG = kSlotBlocks / M;
        for (int u = 0; u * UG < KG; ++u) sc.off[n++] = (int)(seg_offset(seg, Q) + (long)u * UG * M * 256);
    }
    sc.n = n;
    return sc;
}
__device__ const Sched kSched32 = make_sched(32);
__device__ const Sched kSched16 = make_sched(16);

// ---- weight ring ----------------------------------------------------------------------------------
struct Ring {
    float *slots;            // LDS, kRingSlots x kRingSlotFloats
    const float *wfront;     // global
    int unit;                // units consumed so far (wave-uniform)
    const Sched *sched;      // 3-slot ring: the static unit schedule
    int bad;                 // 3-slot ring: a consumed unit was not the scheduled one (programming error)
    f32x4 c0, c1;            // 3-slot ring: A fragments of the next step, carried across unit and segment boundaries
#if VAD_TRACE
    long long wait = 0, span = 0, last = 0;   // cycles at unit barriers / between them (bring-up trace)
#endif
};

// k-groups per ring unit for a segment of M row blocks, and the size of a segment's first unit
constexpr int unit_kgroups(int M) { return kSlotBlocks / M; }
constexpr int first_unit_blocks(int M, int KS) {
    const int kg = (KS + 3) / 4, ug = unit_kgroups(M);
    return M * (kg < ug ? kg : ug);
}

template <int BLOCKS>
__device__ __forceinline__ void ring_issue(const Ring &r, long goff, int slot, const Lane &ln) {
    // BLOCKS blocks of 1 KiB.  LDS destination = M0 (wave-uniform base) + instruction offset + lane*16, the
    // layout global_load_lds requires.
    //
    // The LDS-DMA is issued from inline asm ON PURPOSE: while the compiler knows of a pending
    // global_load_lds it degrades every `s_waitcnt lgkmcnt(N)` to lgkmcnt(0), which serialises the
    // ds_read -> MFMA pipeline of gemm_seg (one exposed LDS latency per 8 MFMAs).  The ring's own
    // protocol orders the DMA instead: ring_wait() = s_waitcnt vmcnt(0) before the unit's barrier.
    static_assert(BLOCKS % 4 == 0 && BLOCKS <= kSlotBlocks, "unit must be whole 4-block groups");
    if (VAD_ABLATE & 8) return;
    // Wave w copies the CONTIGUOUS blocks [w n, (w+1) n), n = BLOCKS / 4: one M0 value and one source base
    // per group of up to 4 blocks, the instruction's immediate offset (which advances the global and the
    // LDS address alike, 13 bits) selects the block within the group.
    constexpr int PW = BLOCKS / 4;
    const float *gbase = r.wfront + goff + (long)ln.wave * PW * 256;      // wave-uniform
    const unsigned voff = ln.lane * 16;                                    // bytes
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) float *)(r.slots + slot * kRingSlotFloats))
                           + (unsigned)ln.wave * PW * 1024u;
#pragma unroll
    for (int grp = 0; grp < PW; grp += 4) {
        const float *src = gbase + (long)grp * 256;
        const unsigned dst = lbase + (unsigned)grp * 1024u;
        const int n = PW - grp < 4 ? PW - grp : 4;
        unsigned keep_m0;                      // M0 is restored: the compiler may keep its own value there
        if (n == 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep_m0) : "v"(voff), "s"(src), "s"(dst) : "memory");
        else if (n == 3)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_m0) : "v"(voff), "s"(src), "s"(dst) : "memory");
        else if (n == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep_m0) : "v"(voff), "s"(src), "s"(dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_m0) : "v"(voff), "s"(src), "s"(dst) : "memory");
    }
}
// All of this wave's ring DMA has landed in LDS (and everything else it had in flight on vmcnt).
__device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// One segment = KG k-groups of M row blocks, streamed as units of up to unit_kgroups(M) k-groups.
// acc[m] += A_seg[m][:, k] * B[k][:]  for all k-steps.  bfun(s) must return the B-operand register of
// k-step s (compile-time s).  NEXT_BLOCKS / next_off describe the first unit of the segment that
// follows in program order (prefetch; 0 = none).
//
// 3-slot ring ("seamless" pipeline, VAD_RING_SLOTS == 3): every unit is 16 blocks = 8 steps.  The workgroup barrier that
// makes unit u+1 visible sits in the MIDDLE of unit u (its DMA was requested in the middle of unit u-1, a full unit
// earlier), and the double-buffered fragment reads run across unit and segment boundaries: the last step of a unit
// prefetches the first fragments of the next one (CARRY_OUT / CARRY_IN; dropped only around the FFT of frame 3 and at
// the end).  No LDS latency and no DMA wait is exposed at a unit boundary any more; the slot that the barrier frees is
// the one of unit u-1, which every wave has left.
template <int M, int KS, int NEXT_BLOCKS, bool CARRY_IN, bool CARRY_OUT, class BF>
__device__ __forceinline__ void gemm_seg(f32x4 (&acc)[M], BF bfun, Ring &ring, long seg_off,
                                         long next_off, const Lane &ln) {
    constexpr int KG = (KS + 3) / 4, UG = unit_kgroups(M), NU = (KG + UG - 1) / UG;
    static_assert(UG >= 1, "slot smaller than one k-group");
    if constexpr (kRingSlots == 3) {
        static_assert(UG * (M / 2) == 8 && KG % UG == 0, "uniform units of 8 steps");
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int unit = ring.unit, n_units = ring.sched->n;
            if (ring.sched->off[unit] != (int)(seg_off + (long)u * UG * M * 256)) ring.bad = 1;
            const f32x4 *A = reinterpret_cast<const f32x4 *>(ring.slots + (unit % 3) * kRingSlotFloats) + ln.lane;
            const f32x4 *An = reinterpret_cast<const f32x4 *>(ring.slots + ((unit + 1) % 3) * kRingSlotFloats) + ln.lane;
            if (!CARRY_IN && u == 0) {
                ring.c0 = A[0];
                ring.c1 = A[64];
            }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if (st == 4 && unit + 1 < n_units) {
                    // this wave's share of unit + 1 has landed; then everyone's, and everyone has left unit - 1.  A bare
                    // s_barrier (no lgkmcnt(0) fence): the fragment reads in flight belong to the current slot
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (!(VAD_ABLATE & 1)) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (unit + 2 < n_units) ring_issue<kSlotBlocks>(ring, ring.sched->off[unit + 2], (unit + 2) % 3, ln);
                }
                f32x4 n0 = ring.c0, n1 = ring.c1;
                if (st + 1 < 8) {
                    n0 = A[(2 * (st + 1)) * 64];
                    n1 = A[(2 * (st + 1) + 1) * 64];
                } else if (u + 1 < NU || CARRY_OUT) {
                    n0 = An[0];
                    n1 = An[64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int kg = u * UG + st / (M / 2), mp = 2 * (st % (M / 2));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float bv = bfun(kg * 4 + ks);
                    acc[mp + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.c0[ks], bv, acc[mp + 0], 0, 0, 0);
                    acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring.c1[ks], bv, acc[mp + 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                ring.c0 = n0;
                ring.c1 = n1;
            }
            ring.unit++;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#if VAD_TRACE
        const long long ta = __builtin_readcyclecounter();
        if (ring.last) ring.span += ta - ring.last;
#endif
        int slot;
        {
        ring_wait();
        if (!(VAD_ABLATE & 1)) __syncthreads();   // unit `ring.unit` landed for every wave; other slot free
#if VAD_TRACE
        ring.last = __builtin_readcyclecounter();
        ring.wait += ring.last - ta;
#endif
        slot = ring.unit & 1;
        constexpr int TAIL = (KG % UG == 0) ? UG : KG % UG;     // k-groups in the segment's last unit
        if (u + 1 < NU) {
            const long off = seg_off + (long)(u + 1) * UG * M * 256;
            if (u + 2 < NU) ring_issue<M * UG>(ring, off, slot ^ 1, ln);
            else ring_issue<M * TAIL>(ring, off, slot ^ 1, ln);
        } else if (NEXT_BLOCKS > 0) {
            ring_issue<(NEXT_BLOCKS > 0 ? NEXT_BLOCKS : 4)>(ring, next_off, slot ^ 1, ln);
        }
        }
        // The unit is consumed as "steps" of two row blocks x 4 k-steps (8 MFMAs).  The A fragments of
        // step i+1 are read from LDS BEFORE the MFMAs of step i are issued (explicit double buffer;
        // the sched_barrier keeps the compiler from sinking the reads back down): a lone wave then
        // keeps the matrix pipe busy without exposing the LDS latency once per 8 MFMAs.
        const f32x4 *A = reinterpret_cast<const f32x4 *>(ring.slots + slot * kRingSlotFloats) + ln.lane;
        const int nk = (KG - u * UG) < UG ? (KG - u * UG) : UG;        // k-groups in this unit
        const int nsteps = nk * (M / 2);
        f32x4 c0 = A[0], c1 = A[64];
#pragma unroll
        for (int st = 0; st < UG * (M / 2); ++st) {
            if (st < nsteps) {
                f32x4 n0 = c0, n1 = c1;
                if (st + 1 < nsteps) {
                    n0 = A[(2 * (st + 1)) * 64];
                    n1 = A[(2 * (st + 1) + 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int kg = u * UG + st / (M / 2), mp = 2 * (st % (M / 2));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (kg * 4 + ks < KS) {
                        const float bv = bfun(kg * 4 + ks);
                        acc[mp + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[ks], bv, acc[mp + 0], 0, 0, 0);
                        acc[mp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[ks], bv, acc[mp + 1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                c0 = n0;
                c1 = n1;
            }
        }
        ring.unit++;
    }
}

template <int M>
__device__ __forceinline__ void init_bias(f32x4 (&acc)[M], const float *bias_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < M; ++m)
        acc[m] = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * m + 4 * ln.g);
}
template <int M>
__device__ __forceinline__ void relu(f32x4 (&acc)[M]) {
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][r] = fmaxf(acc[m][r], 0.f);
}


// Nyquist bin of one frame applied to enc0's 8 output blocks: Y[row] += w_nyq[tap][row] * |Y_nyq| (exact fp32 fma; the
// MFMA it replaces is the same fma chain, so only the position of this term in the sum changes)
__device__ __forceinline__ void nyq_update(f32x4 (&Y)[8], float xn, const float *wn_lds, const Lane &ln) {
#pragma unroll
    for (int m = 0; m < 8; m += 2) {                 // two blocks at a time: the kernel has no registers to spare for
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wn_lds + 16 * m + 4 * ln.g);         // all 8 reads up front
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(wn_lds + 16 * (m + 1) + 4 * ln.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Y[m][r] = fmaf(w0[r], xn, Y[m][r]);
            Y[m + 1][r] = fmaf(w1[r], xn, Y[m + 1][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- the kernel ------------------------------------------------------------------------------------
#if VAD_TRACE
#define TRACE(i)                                                                          \
    do {                                                                                  \
        if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define TRACE(i) do {} while (0)
#endif
#ifndef VAD_WG_PER_CU_8K
#define VAD_WG_PER_CU_8K 2       // (the 2-slot 8 kHz instantiation fits three workgroups per CU at 159 VGPRs; measured 224.3 vs
#endif                           //  224.2 M chunks/s -- occupancy is not what limits the kernel)
template <int Q, typename PcmT, int DEC>
__global__ void __launch_bounds__(256, Q == 16 ? VAD_WG_PER_CU_8K : 2) front_kernel(const FrontArgs a) {
    using namespace vadl;
    constexpr Tab tb = make_tab(8 * Q, Q);
    constexpr int TABF = (tb.total + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float lds[TABF + kRingSlots * kRingSlotFloats];
    float *tab = lds;

    Lane ln;
    ln.lane = threadIdx.x & 63;
    ln.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    ln.g = ln.lane >> 4;
    ln.j = ln.lane & 15;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    long wt = (long)blockIdx.x * 4 + ln.wave;
    ln.tile_valid = wt < total;
    if (!ln.tile_valid) wt = total - 1;
    ln.tl = wt % a.nt;
    ln.st = wt / a.nt;
    ln.t = a.t0 + ln.tl;
    const long bb = ln.st * 16 + ln.j;
    ln.b = (int)(bb < a.B ? bb : a.B - 1);
    ln.from_tail = a.tail != nullptr && ln.t == a.T - 1;
    ln.sgnA = ln.g < 2 ? 1.f : -1.f;
    ln.sgnB = (ln.g & 1) ? -1.f : 1.f;

    TRACE(0);
#if VAD_TRACE
    if (a.trace && threadIdx.x == 0) {
        a.trace[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        a.trace[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif
    if (VAD_STAGGER > 0 && blockIdx.x < 512) {
        // The two workgroups that share a CU start together and run the same program, so their VALU
        // (FFT) and MFMA phases coincide for the whole launch (later workgroups inherit the phase of
        // the one they replace).  Delay the one in the odd wave slot once, by about half a tile, so
        // that one group's FFT overlaps the other's MFMAs.  HW_ID[3:0] = wave slot within the SIMD.
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u;
        if (slot)
            for (int i = 0; i < VAD_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
    Ring ring{lds + TABF, a.wfront, 0, Q == 32 ? &kSched32 : &kSched16, 0, f32x4{}, f32x4{}};
    constexpr long o_e0t0 = seg_offset(E0T0, Q), o_e0t1 = seg_offset(E0T1, Q), o_e0t2 = seg_offset(E0T2, Q);
    constexpr long o_e1t0 = seg_offset(E1T0, Q), o_e1t1 = seg_offset(E1T1, Q), o_e1t2 = seg_offset(E1T2, Q);
    constexpr long o_e2t1 = seg_offset(E2T1, Q), o_e2t2 = seg_offset(E2T2, Q), o_e3t1 = seg_offset(E3T1, Q);

    constexpr int FB_E0 = first_unit_blocks(8, VAD_NYQ_VALU ? Q : Q + 1), FB_E1 = first_unit_blocks(4, 32),
                  FB_E2 = first_unit_blocks(4, 16), FB_E3 = first_unit_blocks(8, 16),
                  FB_IH = first_unit_blocks(8, 32);
    if (kRingSlots == 3) {                       // prime: units 0 and 1
        ring_issue<kSlotBlocks>(ring, ring.sched->off[0], 0, ln);
        ring_issue<kSlotBlocks>(ring, ring.sched->off[1], 1, ln);
    } else {
        ring_issue<FB_E0>(ring, o_e0t1, 0, ln);  // first unit of the program: head of E0T1
    }
    {   // tables -> LDS: all loads of a thread are issued before the first is stored
        static_assert(tb.total % 4 == 0, "tables are copied as 16-byte vectors");
        constexpr int NV = tb.total / 4, PER = (NV + 255) / 256;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.tables);
        f32x4 v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            v[k] = src[i < NV ? i : NV - 1];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < NV) reinterpret_cast<f32x4 *>(tab)[i] = v[k];
        }
    }
    if (kRingSlots == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // units 0 and 1 (and the tables) have landed
    __syncthreads();

    float X0[Q + 1], X1[Q + 1], X2[Q + 1], X3[Q + 1];
    TRACE(1);
    // non-finite input: fft_wave.hpp poison_acc (all four frames, raw).  This form has no register to spare: the running value
    // lives in the thread's own LDS word between the passes (same thread reads and writes it: no barrier)
    __shared__ float pz_lds[256];
    auto pz_add = [&](const float (&X)[Q + 1], bool first) {
        float p = first ? 0.f : pz_lds[threadIdx.x];
        poison_acc<Q>(p, p, X);
        pz_lds[threadIdx.x] = fmaf(X[Q], 0.f, p);            // (the Nyquist magnitude: lane group 0; B[k][j] of that lane is enough)
    };
    fft_pass<Q, 0, PcmT, DEC>(X0, a, tab, ln);
    pz_add(X0, true);
    TRACE(2);
    fft_pass<Q, 1, PcmT, DEC>(X1, a, tab, ln);
    pz_add(X1, false);
    TRACE(3);
    fft_pass<Q, 2, PcmT, DEC>(X2, a, tab, ln);
    pz_add(X2, false);
    TRACE(4);

    auto bX0 = [&](int s) { return X0[s]; };
    auto bX1 = [&](int s) { return X1[s]; };
    auto bX2 = [&](int s) { return X2[s]; };
    auto bX3 = [&](int s) { return X3[s]; };

    f32x4 Y[8], Z0[4], Z1[4];
    auto bY = [&](int s) { return Y[s >> 2][s & 3]; };
    constexpr int KS0 = VAD_NYQ_VALU ? Q : Q + 1;
    const float *wn = tab + tb.w_nyq;                  // [tap][row]
    // |Y_nyq| of chunk j lives in lane group 0 (X[Q]); every lane of the chunk needs it
    auto nyq = [&](const float (&X)[Q + 1]) { return __shfl(X[Q], ln.j); };

    // enc0 frame 0 (taps 1,2; tap 0 is the left zero pad)  -> enc1 out 0 tap 1
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, FB_E0, false, true>(Y, bX0, ring, o_e0t1, o_e0t2, ln);
    gemm_seg<8, KS0, FB_E1, true, true>(Y, bX1, ring, o_e0t2, o_e1t1, ln);
    if (VAD_NYQ_VALU) {
        nyq_update(Y, nyq(X0), wn + 128, ln);
        nyq_update(Y, nyq(X1), wn + 256, ln);
    }
    relu<8>(Y);
    init_bias<4>(Z0, tab + tb.b_e1, ln);
    gemm_seg<4, 32, FB_E0, true, true>(Z0, bY, ring, o_e1t1, o_e0t0, ln);
    // enc0 frame 1 -> enc1 out 0 tap 2, out 1 tap 0
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, FB_E0, true, true>(Y, bX0, ring, o_e0t0, o_e0t1, ln);
    gemm_seg<8, KS0, FB_E0, true, true>(Y, bX1, ring, o_e0t1, o_e0t2, ln);
    gemm_seg<8, KS0, FB_E1, true, true>(Y, bX2, ring, o_e0t2, o_e1t2, ln);
    if (VAD_NYQ_VALU) {
        nyq_update(Y, nyq(X0), wn, ln);
        nyq_update(Y, nyq(X1), wn + 128, ln);
        nyq_update(Y, nyq(X2), wn + 256, ln);
    }
    relu<8>(Y);
    gemm_seg<4, 32, FB_E1, true, true>(Z0, bY, ring, o_e1t2, o_e1t0, ln);
    init_bias<4>(Z1, tab + tb.b_e1, ln);
    gemm_seg<4, 32, FB_E0, true, false>(Z1, bY, ring, o_e1t0, o_e0t0, ln);

    TRACE(5);
#if VAD_TRACE
    ring.span += __builtin_readcyclecounter() - ring.last;
    ring.last = 0;
#endif
    fft_pass<Q, 3, PcmT, DEC>(X3, a, tab, ln);
    pz_add(X3, false);
    TRACE(6);

    // enc0 frame 2 -> enc1 out 1 tap 1
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, FB_E0, false, true>(Y, bX1, ring, o_e0t0, o_e0t1, ln);
    gemm_seg<8, KS0, FB_E0, true, true>(Y, bX2, ring, o_e0t1, o_e0t2, ln);
    gemm_seg<8, KS0, FB_E1, true, true>(Y, bX3, ring, o_e0t2, o_e1t1, ln);
    if (VAD_NYQ_VALU) {
        nyq_update(Y, nyq(X1), wn, ln);
        nyq_update(Y, nyq(X2), wn + 128, ln);
        nyq_update(Y, nyq(X3), wn + 256, ln);
    }
    relu<8>(Y);
    gemm_seg<4, 32, FB_E0, true, true>(Z1, bY, ring, o_e1t1, o_e0t0, ln);
    // enc0 frame 3 (taps 0,1; tap 2 is the right zero pad) -> enc1 out 1 tap 2
    init_bias<8>(Y, tab + tb.b_e0, ln);
    gemm_seg<8, KS0, FB_E0, true, true>(Y, bX2, ring, o_e0t0, o_e0t1, ln);
    gemm_seg<8, KS0, FB_E1, true, true>(Y, bX3, ring, o_e0t1, o_e1t2, ln);
    if (VAD_NYQ_VALU) {
        nyq_update(Y, nyq(X2), wn, ln);
        nyq_update(Y, nyq(X3), wn + 128, ln);
    }
    relu<8>(Y);
    gemm_seg<4, 32, FB_E2, true, true>(Z1, bY, ring, o_e1t2, o_e2t1, ln);
    relu<4>(Z0);
    relu<4>(Z1);

    // enc2 (T 2 -> 1, stride 2: taps 1,2 see frames 0,1), enc3 (T = 1: centre tap only)
    f32x4 Vv[4];
    auto bZ0 = [&](int s) { return Z0[s >> 2][s & 3]; };
    auto bZ1 = [&](int s) { return Z1[s >> 2][s & 3]; };
    auto bV = [&](int s) { return Vv[s >> 2][s & 3]; };
    TRACE(7);
    init_bias<4>(Vv, tab + tb.b_e2, ln);
    gemm_seg<4, 16, FB_E2, true, true>(Vv, bZ0, ring, o_e2t1, o_e2t2, ln);
    gemm_seg<4, 16, FB_E3, true, true>(Vv, bZ1, ring, o_e2t2, o_e3t1, ln);
    relu<4>(Vv);
    f32x4 Fe[8];
    auto bF = [&](int s) { return Fe[s >> 2][s & 3]; };
    init_bias<8>(Fe, tab + tb.b_e3, ln);
    gemm_seg<8, 16, FB_IH, true, true>(Fe, bV, ring, o_e3t1, seg_offset(IH0, Q), ln);
    relu<8>(Fe);
    poison_into(Fe[0], pz_lds[threadIdx.x]);
    TRACE(8);

    // LSTM input-gate pre-activations, one gate (8 row blocks) at a time, stored in D-fragment order
    float *gxt = a.gx + ((size_t)(ln.st * a.nt + ln.tl) * 32) * 256 + ln.lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 G[8];
        init_bias<8>(G, tab + tb.b_g + 128 * q, ln);
        if (q < 3) gemm_seg<8, 32, FB_IH, true, true>(G, bF, ring, seg_offset(IH0 + q, Q), seg_offset(IH0 + q + 1, Q), ln);
        else gemm_seg<8, 32, 0, true, false>(G, bF, ring, seg_offset(IH3, Q), 0, ln);
        if (ln.tile_valid) {
            const float nanv = __builtin_nanf("");
#pragma unroll
            for (int m = 0; m < 8; ++m)
                *reinterpret_cast<f32x4 *>(gxt + (size_t)(8 * q + m) * 256) =
                    (kRingSlots == 3 && ring.bad) ? f32x4{nanv, nanv, nanv, nanv} : G[m];
        }
    }
    TRACE(9);
#if VAD_TRACE
    if (a.trace && threadIdx.x == 0) {
        ring.span += __builtin_readcyclecounter() - ring.last;
        a.trace[(size_t)blockIdx.x * 16 + 12] = ring.wait;
        a.trace[(size_t)blockIdx.x * 16 + 13] = ring.span;
        a.trace[(size_t)blockIdx.x * 16 + 14] = ring.unit;
    }
#endif
}

}  // namespace

template <typename PcmT>
hipError_t launch_front(int sr, const FrontArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.nt <= 0) return hipSuccess;
    const long nst = (a.B + 15) / 16, total = nst * a.nt;
    const unsigned grid = (unsigned)((total + 3) / 4);
    // a.dec == 2: 32 kHz input, decimation folded into the loads (16 kHz net only).  (A stride-3 instantiation for
    // 48 kHz does not fit the 256-VGPR budget of two workgroups per CU -- it spills ~230 registers -- so 48 kHz and the
    // other multiples go through the engine's decimation pass.)
    if (sr == 16000 && a.dec == 2) hipLaunchKernelGGL((front_kernel<32, PcmT, 2>), dim3(grid), dim3(256), 0, s, a);
    else if (a.dec > 1) return hipErrorInvalidValue;
    else if (sr == 16000) hipLaunchKernelGGL((front_kernel<32, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((front_kernel<16, PcmT, 1>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
template hipError_t launch_front<float>(int, const FrontArgs &, hipStream_t);
template hipError_t launch_front<int16_t>(int, const FrontArgs &, hipStream_t);

}  // namespace vad
