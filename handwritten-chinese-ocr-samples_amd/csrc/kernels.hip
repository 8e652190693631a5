// gfx950 (MI355X / CDNA4) kernels for the hctr CNN+CTC inference path.
//
// Reference semantics (file:line relative to the reference root) are cited per kernel; the
// tiling, layouts and fusion are this engine's own. Wavefront = 64 lanes throughout.
#include "kernels.h"

#include <cstdlib>

#ifndef RING
#define RING 3         // A-fragment ring slots (prefetch distance RING-1 groups)
#endif
#ifndef BHALF_AT
#define BHALF_AT 1     // MFMA group before which the second half of the B fragments is read
#endif
#ifndef DMA_SPREAD
#define DMA_SPREAD 0   // A/B build switch: 0 = a step's 4 weight-DMA pieces right after the first fragment reads; 1 = one piece
                       // after each of MFMA groups 0..3; 2 = after groups 0, 2, 4, 6 (issue cost in the MFMAs' shadow);
                       // 3 = all four BEFORE the first fragment reads
#endif
#ifndef NOPRIO
#define NOPRIO 0       // A/B build switch (tools/ab_build.sh): 1 drops the s_setprio around MFMA groups
#endif
#ifndef RPRE
#define RPRE 0         // A/B build switch: 1 compiles the residual prefetch of conv2's last K step in (then HCTR_RPRE=0/1 selects it
                       // at run time). Measured neutral (132.5-133.6 ms either way; the epilogue's residual phase stays 3.3 us
                       // because the second half's loads still start there), so it is compiled out by default.
#endif
#ifndef RPRE_GROUPS
#define RPRE_GROUPS 4  // MFMA groups of the last K step after which two residual vectors each are fetched: 4 = the 8 vectors of
                       // the first 64-cout block (the most that stays in registers: 5, 6 and 8 groups spill 144 B per lane);
                       // the second block's 8 vectors are loaded at the start of the epilogue
#endif
#ifndef GEN_ROLL
#define GEN_ROLL 1     // A/B: generic kernel, 8-accumulator-tile waves (256x256 tile: the head GEMM): 1 = the halo kernels' rolling
                       // fragment pipeline (A pairs read two MFMA groups ahead, second-half B fragments during group 1, the next
                       // step's DMA burst after the first reads) instead of "all reads of a half step, then its 32 MFMAs"
#endif
#ifndef GEN_ASM_DMA
#define GEN_ASM_DMA 1  // A/B: generic conv_mfma kernel (head GEMM, conv0_2 A/B paths): 1 = its LDS-DMA issued from inline asm like the
                       // halo kernels (hidden from hipcc's waitcnt pass, which otherwise drains lgkmcnt(0) at every wait)
#endif
#ifndef GEN_PRIO
#define GEN_PRIO 0     // A/B: generic conv_mfma kernel (head GEMM): 1 = static priority 1 over the whole K loop instead of flips
#endif
#ifndef STEM_PRIO
#define STEM_PRIO 0    // A/B: fused stem kernel: 1 = static priority 1 over the whole conv0_2 (MFMA) phase instead of flips
#endif
#ifndef PRIO_MODE
#define PRIO_MODE 1    // A/B build switch, halo4 kernel: 0 = priority 1 around every MFMA group, 0 elsewhere; 1 = the whole K
                       // loop at priority 1 (its LDS reads / DMA issue / barrier beat the partner workgroup's epilogue and
                       // prologue VALU work, which stays at 0), no per-group flips; 2 = loop at 1, MFMA groups at 2;
                       // 3 = as 0 inside the loop, but prologue and epilogue at priority 2 (a slot's overhead phases are
                       // short and fully exposed: let them win the issue port against the partner's loop);
                       // 4 = as 1, with the prologue at priority 1 too (only the epilogue at 0)
#endif

namespace hctr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

// One 1-KiB LDS-DMA piece (global_load_lds_dwordx4: lane i -> LDS lds_dst + 16*i) issued from inline
// asm. hipcc's waitcnt pass drains lgkmcnt(0) at every wait while it can see an LDS-DMA in flight
// (measured: perfect counted waits once the DMA is hidden), so kernels that software-pipeline their
// ds_reads issue the DMA this way and retire it themselves with `s_waitcnt vmcnt(0)` before the
// barrier. M0 carries the LDS byte address and is restored (cdna guide section 5.7).
__device__ __forceinline__ void glds16_asm(const char* gsrc, char* lds_dst) {
    const uint32_t lds = (uint32_t)(uintptr_t)((lptr_t)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds)
                 : "memory");
}

// 4-byte-per-lane LDS-DMA with per-lane source addresses (lane i -> LDS lds_dst + 4*i): used to TOUCH cache lines (the
// data is never read) without a register destination, so no VGPR is clobbered when the load returns late
__device__ __forceinline__ void glds4_asm(const char* gsrc, char* lds_dst) {
    const uint32_t lds = (uint32_t)(uintptr_t)((lptr_t)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds)
                 : "memory");
}

// lane permutation inside a row of 16 lanes on the VALU data path (v_mov_b32_dpp)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// np.argmax order (utils/ctc_codec.py:75): a NaN counts as the maximum, the FIRST maximum wins
__device__ __forceinline__ bool np_gt(float a, float b) { return a > b || (a != a && b == b); }
__device__ __forceinline__ bool np_eq(float a, float b) { return a == b || (a != a && b != b); }

// Same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (no 64-bit VALU address).
__device__ __forceinline__ void glds16_asm_s(const char* sbase, uint32_t voff, char* lds_dst) {
    const uint32_t lds = (uint32_t)(uintptr_t)((lptr_t)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds)
                 : "memory");
}

// Four pieces of one wave in ONE asm statement: same per-lane offset, four wave-uniform bases, four LDS addresses; M0 is
// saved and restored once instead of four times (six scalar moves fewer per K step).
__device__ __forceinline__ void glds16x4_asm_s(const char* s0, const char* s1, const char* s2, const char* s3, uint32_t voff,
                                               char* d0, char* d1, char* d2, char* d3) {
    const uint32_t l0 = (uint32_t)(uintptr_t)((lptr_t)d0), l1 = (uint32_t)(uintptr_t)((lptr_t)d1);
    const uint32_t l2 = (uint32_t)(uintptr_t)((lptr_t)d2), l3 = (uint32_t)(uintptr_t)((lptr_t)d3);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(l0), "s"(l1), "s"(l2), "s"(l3)
                 : "memory");
}

// Four pieces with ONE base, four per-lane offsets and LDS addresses LSTRIDE apart (halo rows): M0 is stepped inside the
// statement, so it takes two scalar inputs instead of five
template <int LSTRIDE>
__device__ __forceinline__ void glds16x4v_asm_s(const char* sbase, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, char* d0) {
    const uint32_t l0 = (uint32_t)(uintptr_t)((lptr_t)d0);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(l0), "i"(LSTRIDE)
                 : "memory", "scc");
}

// -------------------------------------------------------------------------------------------
// Shared epilogue of the MFMA conv kernels: + folded-BN bias, optional SE partial sums, ReLU,
// (2,1) max-pool, zeroing of columns >= W, fp16 NHWC store (or fp32 rows in linear mode).
// lane (q, c): for cout block cb (64 couts) the lane owns couts (jj>>1)*32 + q*8 + (jj&1)*4 + i (jj = j & 3),
// i.e. two runs of 8 consecutive couts, of pixel column c and pixel repeat n.
// -------------------------------------------------------------------------------------------
// cout offset of accumulator tile j inside a wave's cout range (see conv_epilogue)
__device__ __forceinline__ constexpr int acc_cout_offset(int j) { return (j >> 2) * 64 + ((j >> 1) & 1) * 32 + (j & 1) * 4; }

// SE scales below this are treated as this value on both sides of the downsample fusion (the accumulators hold
// residual / max(s, floor) and are multiplied by max(s, floor)): keeps the quotient finite for a saturated sigmoid
constexpr float kSeScaleFloor = 1e-12f;

template <int WN, int WM, int JT, bool LINEAR, bool SPLIT, bool PRIVATE_RED = false, bool BIAS_DONE = false,
          bool RESID_IN_ACC = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[JT][4], char* smem, int tid, int lane,
                                              int wn, int wm, int n0, int mt, int img, int th, int tw, int hbase,
                                              int w0, unsigned long long* st = nullptr,
                                              const f16x8 (*pre_lo)[4] = nullptr, const f16x8 (*pre_hi)[4] = nullptr,
                                              bool use_pre = false) {
    // diagnostic instance only: st = this workgroup's stamp slots 8.. (after bias, residual, rounding, SE sums)
    auto estamp = [&](int i) {
        if (st != nullptr) {
            __builtin_amdgcn_sched_barrier(0);
            if (tid == 0) st[i] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int WC = JT * 16;
    constexpr int BN = WN * WC, BM = WM * 64;
    const int q = lane >> 4;
    const int c = lane & 15;
    const int cw0 = n0 + wn * WC + q * 8;       // + co(j) + i
    // cout offset of accumulator tile j (cb = j>>2 the 64-cout block, jj = j&3): a lane owns two runs of 8
    // consecutive couts per block, q*8.. (jj 0,1) and 32+q*8.. (jj 2,3), so the four lanes of a pixel write
    // 64 contiguous bytes per store instruction (half a cache line without holes).
    auto co = [](int j) { return acc_cout_offset(j); };

    if (!BIAS_DONE) {                   // (the halo4 kernel starts its accumulators at the bias instead)
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const f32x4 b4 = *(const f32x4*)(a.bias + cw0 + co(j));
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][n][i] += b4[i];
        }
    }

    estamp(8);
    if (LINEAR && a.emit_cnt != nullptr) {
        // fused beam front end, PASS 2 (see ConvArgs): list every (class, logit >= vmin[row]) and sum the row's
        // float64 exp terms expf(v - gmax) per class part - the terms row_topk_kernel sums over a stored row.
        static_assert(!LINEAR || WN == kLinearWN, "partials per n-tile");
        const int64_t part = (int64_t)((n0 / BN) * WN + wn) * a.M;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int64_t m = (int64_t)mt * BM + wm * 64 + n * 16 + c;
            const bool mv = m < a.M;
            const float vmin = mv ? a.row_thr[2 * m] : INFINITY;
            const float gmax = mv ? a.row_thr[2 * m + 1] : 0.f;
            double es = 0.0;
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cls = cw0 + co(j) + i;
                    const float v = acc[j][n][i];
                    if (cls < a.Cout) {
                        es += (double)expf(v - gmax);
                        if (v >= vmin) {
                            const int pos = atomicAdd(a.emit_cnt + m, 1);
                            if (pos < a.emit_cap) {
                                int32_t* e = a.emit_list + ((int64_t)m * a.emit_cap + pos) * 2;
                                e[0] = cls;
                                e[1] = __builtin_bit_cast(int32_t, v);
                            }
                        }
                    }
                }
            es += __shfl_xor(es, 16);
            es += __shfl_xor(es, 32);
            if (q == 0 && mv) a.esum[part + m] = es;
        }
        return;
    }
    if (LINEAR && a.amax_idx != nullptr) {
        // fused greedy argmax: classes rise with (j, i) for a lane and with q, wn, nt beyond it, so a strict
        // '>' keeps the first maximum inside a lane and the (value, class) merge keeps it across lanes.
        // Guarded precision (a.amax_val2 set): the part's runner-up value and largest |logit| ride along.
        static_assert(!LINEAR || WN == kLinearWN, "partials per n-tile");
        const int64_t part = (int64_t)((n0 / BN) * WN + wn) * a.M;
        const bool guard = a.amax_val2 != nullptr;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float bv = -INFINITY, sv = -INFINITY, av = 0.f;
            int bi = 0x7fffffff;
            if (!guard) {
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int cls = cw0 + co(j) + i;
                        const float v = acc[j][n][i];
                        if (cls < a.Cout && np_gt(v, bv)) { bv = v; bi = cls; }
                    }
            } else {
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int cls = cw0 + co(j) + i;
                        const float v = acc[j][n][i];
                        if (cls < a.Cout) {
                            av = fmaxf(av, fabsf(v));
                            if (np_gt(v, bv)) { sv = bv; bv = v; bi = cls; }
                            else if (np_gt(v, sv)) sv = v;
                        }
                    }
            }
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                const bool wins = np_gt(ov, bv) || (np_eq(ov, bv) && oi < bi);
                if (guard) {                       // runner-up of the union: the loser's best or the winner's second
                    const float osv = __shfl_xor(sv, off);
                    const float lose = wins ? bv : ov, keep2 = wins ? osv : sv;
                    sv = np_gt(lose, keep2) ? lose : keep2;
                    av = fmaxf(av, __shfl_xor(av, off));
                }
                if (wins) { bv = ov; bi = oi; }
            }
            const int64_t m = (int64_t)mt * BM + wm * 64 + n * 16 + c;
            if (q == 0 && m < a.M) {
                a.amax_val[part + m] = bv;
                a.amax_idx[part + m] = bi;
                if (guard) {
                    a.amax_val2[part + m] = sv;
                    a.amax_abs[part + m] = av;
                }
            }
            if (a.psum != nullptr) {
                // beam front end, PASS 1: sum of expf(v - part max) per (part, row), and the logit of class 0
                float ps = 0.f;
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (cw0 + co(j) + i < a.Cout) ps += expf(acc[j][n][i] - bv);
                ps += __shfl_xor(ps, 16);
                ps += __shfl_xor(ps, 32);
                if (q == 0 && m < a.M) {
                    a.psum[part + m] = ps;
                    if (cw0 == 0) a.blank_logit[m] = acc[0][n][0];          // cw0 == 0: n0 == 0, wn == 0 (and q == 0)
                }
            }
        }
        return;
    }
    if (LINEAR) {
        float* out = (float*)a.y;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int64_t m = (int64_t)mt * BM + wm * 64 + n * 16 + c;
            if (m < a.M) {
#pragma unroll
                for (int j = 0; j < JT; ++j)
                    *(f32x4*)(out + m * a.ldo + cw0 + co(j)) = acc[j][n];
            }
        }
        return;
    }

    const int w = w0 + c;
    const bool wvalid = w < a.W;
    half_t* out = (half_t*)a.y;
    const int64_t pix0 = a.out_off + img * a.out_sb + (int64_t)w * a.out_sw + cw0;

    // fused squeeze-excite + residual (BasicBlock :54-58): acc = acc * scale[img][cout] + residual
    if (SPLIT && !RESID_IN_ACC && a.se_scale != nullptr) {
        // f16x3: the residual is hi + lo (planes Cout apart); loaded per (cb, n), accuracy mode only
        const float* sc = a.se_scale + (int64_t)img * a.Cout + cw0;
        if (w < a.out_wlimit) {
#pragma unroll
            for (int cb = 0; cb < JT / 4; ++cb)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const half_t* r = a.resid + pix0 + (int64_t)(hbase + n) * a.out_sh + cb * 64;
                    const f16x8 h0 = *(const f16x8*)r, h1 = *(const f16x8*)(r + 32);
                    const f16x8 l0 = *(const f16x8*)(r + a.Cout), l1 = *(const f16x8*)(r + a.Cout + 32);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float rv = e < 8 ? (float)h0[e] + (float)l0[e] : (float)h1[e - 8] + (float)l1[e - 8];
                        const float sv = sc[cb * 64 + (e >> 3) * 32 + (e & 7)];
                        acc[cb * 4 + (e >> 2)][n][e & 3] = fmaf(acc[cb * 4 + (e >> 2)][n][e & 3], sv, rv);
                    }
                }
        }
    } else if (RESID_IN_ACC) {
        // fused downsample (halo4 DSFUSE instance): the residual already sits in the accumulators as r / s
        // (compile-time branch: as a third runtime branch it made hipcc spill in every instance of this epilogue)
        const float* sc = a.se_scale + (int64_t)img * a.Cout + cw0;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const f32x4 s4 = *(const f32x4*)(sc + co(j));
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][n][i] *= fmaxf(s4[i], kSeScaleFloor);
        }
    } else if (a.se_scale != nullptr) {
        const float* sc = a.se_scale + (int64_t)img * a.Cout + cw0;
        if (w < a.out_wlimit) {
            // all residual loads first (one latency exposure), 2 x 8 couts = 2 x 16 B per (cb, n) - unless the caller has
            // fetched them already (halo4 kernel: during its last K step, pre_lo / pre_hi)
            f16x8 rlo[JT / 4][4], rhi[JT / 4][4];
#pragma unroll
            for (int cb = 0; cb < JT / 4; ++cb)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (pre_lo != nullptr && use_pre && cb * 4 + n < RPRE_GROUPS) {       // (pre_lo: compile-time per call site; use_pre: run time -
                        rlo[cb][n] = pre_lo[cb][n];           //  a run-time SELECTED pointer would force the arrays into scratch)
                        rhi[cb][n] = pre_hi[cb][n];
                    } else {
                        const half_t* r = a.resid + pix0 + (int64_t)(hbase + n) * a.out_sh + cb * 64;
                        rlo[cb][n] = *(const f16x8*)r;
                        rhi[cb][n] = *(const f16x8*)(r + 32);
                    }
                }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const f32x4 s4 = *(const f32x4*)(sc + co(j));
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = (j & 3) * 4 + i;
                        const float r = e < 8 ? (float)rlo[j >> 2][n][e] : (float)rhi[j >> 2][n][e - 8];
                        acc[j][n][i] = fmaf(acc[j][n][i], s4[i], r);
                    }
            }
        }
    }

    estamp(9);
    if (!SPLIT) {
        // One rounding to fp16 (v_cvt_pk_f16_f32, RNE), then ReLU and the column mask on PACKED halves:
        // rounding is monotonic and keeps the sign, so relu(round(v)) == round(relu(v)). P holds exactly the
        // stored values; about 2 VALU ops per value instead of 7.
        const _Float16 lo1 = a.relu ? (_Float16)0.f : (_Float16)(-INFINITY);
        const f16x2 lo2 = {lo1, lo1};
        const uint32_t keep = wvalid ? 0xffffffffu : 0u;
        uint32_t P[JT][4][2];
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f32x4 v = acc[j][n];
                if (a.pool) {                                    // (2,1) max-pool -> even n; odd n unused
                    const f32x4 u = acc[j][n | 1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[j][n & ~1][i], u[i]);
                }
                f16x2 h0 = __builtin_convertvector((f32x2){v[0], v[1]}, f16x2);
                f16x2 h1 = __builtin_convertvector((f32x2){v[2], v[3]}, f16x2);
                h0 = __builtin_elementwise_max(h0, lo2);
                h1 = __builtin_elementwise_max(h1, lo2);
                P[j][n][0] = __builtin_bit_cast(uint32_t, h0) & keep;
                P[j][n][1] = __builtin_bit_cast(uint32_t, h1) & keep;
            }
        estamp(10);
        if (a.se_part != nullptr) {
            // per-(image, channel) sums of the stored values over this block's pixels, fixed reduction
            // order (deterministic: lane tree -> LDS -> one partial row per block; no float atomics).
            if (!PRIVATE_RED) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a trailing LDS-DMA (if any) has landed
                __syncthreads();                   // main-loop LDS no longer needed
            }
            float* red = (float*)smem;             // [WM][BN]
            // Sum over the wave's 64 pixels per cout: first the lane's four rows, then a REDUCE-SCATTER butterfly over
            // the 16 pixel columns of a lane row (DPP quad_perm xor 1, xor 2, row_half_mirror, row_mirror). At every
            // level a lane keeps half of its values and hands the other half to its partner, so the 4 levels cost
            // 16 + 8 + 4 + 2 exchanges instead of 4 x 32, and every lane ends with JT/4 finished sums (one store each,
            // no masked writes). The additions are exactly the full butterfly's (pair sums, quad sums, half-row sums, row
            // sums; an fp add commutes), so the results are bit-identical to summing every value in every lane.
            // Which half a lane keeps must agree between mirror partners: bits e0 = b0^b2, e1 = b1^b2, e2 = b2^b3, e3 = b3
            // of the column c = b3 b2 b1 b0 (c^1 flips only e0, c^2 only e1, c^7 only e2, c^15 only e3).
            constexpr int NV = JT * 4;
            float vals[NV];
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        t[n] = (float)__builtin_bit_cast(f16x2, P[j][n][i >> 1])[i & 1];
                    vals[j * 4 + i] = (t[0] + t[1]) + (t[2] + t[3]);
                }
            const bool e0 = ((c ^ (c >> 2)) & 1) != 0, e1 = (((c >> 1) ^ (c >> 2)) & 1) != 0;
            const bool e2 = (((c >> 2) ^ (c >> 3)) & 1) != 0, e3 = ((c >> 3) & 1) != 0;
            float r1[NV / 2], r2[NV / 4], r3[NV / 8], r4[NV / 16];
#pragma unroll
            for (int u = 0; u < NV / 2; ++u)
                r1[u] = (e0 ? vals[2 * u + 1] : vals[2 * u]) + dpp_f32<0xB1>(e0 ? vals[2 * u] : vals[2 * u + 1]);
#pragma unroll
            for (int u = 0; u < NV / 4; ++u)
                r2[u] = (e1 ? r1[2 * u + 1] : r1[2 * u]) + dpp_f32<0x4E>(e1 ? r1[2 * u] : r1[2 * u + 1]);
#pragma unroll
            for (int u = 0; u < NV / 8; ++u)
                r3[u] = (e2 ? r2[2 * u + 1] : r2[2 * u]) + dpp_f32<0x141>(e2 ? r2[2 * u] : r2[2 * u + 1]);
#pragma unroll
            for (int u = 0; u < NV / 16; ++u)
                r4[u] = (e3 ? r3[2 * u + 1] : r3[2 * u]) + dpp_f32<0x140>(e3 ? r3[2 * u] : r3[2 * u + 1]);
            const int vlow = (e3 ? 8 : 0) + (e2 ? 4 : 0) + (e1 ? 2 : 0) + (e0 ? 1 : 0);
#pragma unroll
            for (int u = 0; u < NV / 16; ++u) {
                const int v = 16 * u + vlow, j = v >> 2, i = v & 3;      // this lane's finished sum: value j*4 + i
                red[wm * BN + wn * WC + q * 8 + ((j >> 2) * 64 + ((j >> 1) & 1) * 32 + (j & 1) * 4) + i] = r4[u];
            }
            __syncthreads();
            if (tid < BN) {
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < WM; ++m) sum += red[m * BN + tid];
                const int64_t tile = (int64_t)img * (a.tilesH * a.tilesW) + th * a.tilesW + tw;
                if (n0 + tid < a.Cout) a.se_part[tile * a.Cout + n0 + tid] = sum;
            }
        }
        estamp(11);
        if (a.dbg & 256) {           // dbg 256: timing experiment without the output stores
            uint32_t t = 0;
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n) t += P[j][n][0] ^ P[j][n][1];
            if (t == 0x12345678u) out[0] = (half_t)1.f;
        } else if (w < a.out_wlimit) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (a.pool && (n & 1)) continue;
                const int ho = a.pool ? ((hbase + n) >> 1) : (hbase + n);
                half_t* o = out + pix0 + (int64_t)ho * a.out_sh;
#pragma unroll
                for (int cb = 0; cb < JT / 4; ++cb) {
                    const int j0 = cb * 4;
                    *(u32x4*)(o + cb * 64) = (u32x4){P[j0][n][0], P[j0][n][1], P[j0 + 1][n][0], P[j0 + 1][n][1]};
                    *(u32x4*)(o + cb * 64 + 32) =
                        (u32x4){P[j0 + 2][n][0], P[j0 + 2][n][1], P[j0 + 3][n][0], P[j0 + 3][n][1]};
                }
            }
        }
        return;
    }
    // f16x3 (SPLIT) tail. ReLU, column mask, fp16 rounding (in place: acc holds exactly what the planes add up to)
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[j][n][i];
                if (a.pool && (n & 1) == 0) v = fmaxf(v, acc[j][n + 1][i]);   // (2,1) max-pool -> even n
                if (a.relu) v = fmaxf(v, 0.f);
                if (!wvalid) v = 0.f;
                if (SPLIT) {                        // stored as hi + lo: keep what the two planes add up to
                    const half_t hi = (half_t)v;
                    v = a.drop_lo ? (float)hi : (float)hi + (float)(half_t)(v - (float)hi);      // (drop_lo: lo plane = 0 below)
                } else {
                    v = (float)(half_t)v;
                }
                acc[j][n][i] = v;
            }

    estamp(10);
    if (a.se_part != nullptr) {
        // per-(image, channel) sums of the stored values over this block's pixels, fixed reduction
        // order (deterministic: lane tree -> LDS -> one partial row per block; no float atomics).
        if (!PRIVATE_RED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a trailing LDS-DMA (if any) has landed
            __syncthreads();                   // main-loop LDS no longer needed
        }
        float* red = (float*)smem;             // [WM][BN]
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = (acc[j][0][i] + acc[j][1][i]) + (acc[j][2][i] + acc[j][3][i]);
                // butterfly over the 16 pixel columns of a lane row on the VALU (DPP), not through ds_bpermute:
                // pairs (c, c^1), (c, c^2), then the mirrored lane of the other quad / other half, which holds
                // the same partial sum the xor-4 / xor-8 partner would - bitwise the xor butterfly's result
                s += dpp_f32<0xB1>(s);            // quad_perm [1,0,3,2]
                s += dpp_f32<0x4E>(s);            // quad_perm [2,3,0,1]
                s += dpp_f32<0x141>(s);           // row_half_mirror
                s += dpp_f32<0x140>(s);           // row_mirror
                if (c == 0) red[wm * BN + wn * WC + q * 8 + co(j) + i] = s;
            }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) s += red[m * BN + tid];
            const int64_t tile = (int64_t)img * (a.tilesH * a.tilesW) + th * a.tilesW + tw;
            if (n0 + tid < a.Cout) a.se_part[tile * a.Cout + n0 + tid] = s;
        }
    }

    estamp(11);
    if (a.dbg & 256) {           // dbg 256: timing experiment without the output stores
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) t += acc[j][n][0] + acc[j][n][1] + acc[j][n][2] + acc[j][n][3];
        if (t == 123.456f) out[0] = (half_t)t;
    } else if (w < a.out_wlimit) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (a.pool && (n & 1)) continue;
            const int ho = a.pool ? ((hbase + n) >> 1) : (hbase + n);
            half_t* o = out + pix0 + (int64_t)ho * a.out_sh;
#pragma unroll
            for (int cb = 0; cb < JT / 4; ++cb) {
                f16x8 lo, hi;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const half_t hv = (half_t)acc[cb * 4 + (e >> 2)][n][e & 3];
                    if (e < 8) lo[e] = hv; else hi[e - 8] = hv;
                }
                *(f16x8*)(o + cb * 64) = lo;
                *(f16x8*)(o + cb * 64 + 32) = hi;
                if (SPLIT) {                        // planes: [hi | lo | hi]
                    f16x8 l0, l1;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = acc[cb * 4 + (e >> 2)][n][e & 3];
                        const half_t lv = (half_t)(v - (float)(half_t)v);
                        if (e < 8) l0[e] = lv; else l1[e - 8] = lv;
                    }
                    *(f16x8*)(o + a.Cout + cb * 64) = l0;
                    *(f16x8*)(o + a.Cout + cb * 64 + 32) = l1;
                    *(f16x8*)(o + 2 * a.Cout + cb * 64) = lo;
                    *(f16x8*)(o + 2 * a.Cout + cb * 64 + 32) = hi;
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// conv_mfma: 3x3 (pad 1) / 1x1 convolution + folded BatchNorm (+ReLU, +(2,1) max-pool,
// +squeeze-excite partial sums) as an implicit GEMM on v_mfma_f32_16x16x32_f16.
//
// Reference ops fused here: nn.Conv2d(k,1,pad) -> BatchNorm2d(eval) -> ReLU -> max_pool2d((2,1))
// (models/handwritten_ctr_model.py:116-150, 47-53) and the spatial sum that SELayer's
// AdaptiveAvgPool2d needs (:27); in linear mode: self.linear (:175).
//
// GEMM view: D[cout][pixel] = sum_{tap,cin} Wt[tap][cout][cin] * X[pixel + tap][cin]
//   MFMA A operand = weights (rows = couts), B operand = pixels (cols), so every lane ends up with
//   two runs of 8 consecutive couts of one pixel; the 4 lanes of a pixel store 64 contiguous bytes.
// Block = WN x WM waves; each wave owns JT*16 couts x 64 pixels (JT x 4 MFMA tiles): 64 couts
//   (64 fp32 acc regs) for the 64x256 and 128x128 block tiles, 128 couts for the 256x256 tile.
//   conv mode: a wave's 64 pixels are a 4-row x 16-column patch; MFMA column c = image column,
//   pixel repeat n = image row, so the (2,1) max-pool pairs repeats (0,1),(2,3) inside a lane.
// K loop: taps x (Cin/64) steps; both operand tiles ([rows][64 cin] fp16 = 128-byte rows) are
//   staged with global_load_lds_dwordx4 into a double-buffered LDS image. The zero padding of the
//   convolution is stored in memory (1-pixel zero border), so a tap is only an address offset.
// LDS image: linear per wave-instruction (8 rows x 128 B), 16-byte chunk index XOR (row & 7)
//   applied on the global SOURCE address and again on the ds_read_b128 address (conflict-free for
//   the 16x16x32 operand pattern; cdna guide rule 21).
// -------------------------------------------------------------------------------------------
template <int WN, int WM, int JT, int TAPS, bool LINEAR, bool PIPE, bool SPLIT>
__global__ __launch_bounds__(WN* WM * 64) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int NT = WN * WM * 64;
    constexpr int WC = JT * 16;          // couts per wave (64 or 128)
    constexpr int BN = WN * WC, BM = WM * 64;
    constexpr int NIW = BN * 8 / NT;     // 16-byte chunks of the weight tile per thread
    constexpr int NIX = BM * 8 / NT;     // ... of the pixel tile
    constexpr int TILE_BYTES = (BN + BM) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv / WM, wm = wv % WM;

    // ---- block -> (m_tile, n_tile), XCD-aware: each XCD walks a contiguous run of the
    //      (m-major, n-minor) order so the n-tiles of one pixel tile share that XCD's L2.
    const int total = a.mtiles * a.ntiles;
    int lin;
    {
        const int id = blockIdx.x, xcd = id & 7, s = id >> 3;
        const int q = total >> 3, r = total & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
    }
    const int nt = lin % a.ntiles;
    const int mt = lin / a.ntiles;
    const int n0 = nt * BN;

    int img = 0, th = 0, tw = 0;
    if (!LINEAR) {
        tw = mt % a.tilesW;
        const int t2 = mt / a.tilesW;
        th = t2 % a.tilesH;
        img = t2 / a.tilesH;
    }
    const int cin = a.Cin;

    // ---- per-thread staging sources: uniform base (SGPRs) + 32-bit per-lane byte offset ----
    const char* xbase;
    if (LINEAR) xbase = (const char*)(a.x + (int64_t)mt * BM * cin);
    else xbase = (const char*)(a.x + img * a.in_sb);
    const char* wbase = (const char*)(a.w + (int64_t)n0 * cin);
    uint32_t xoff[NIX], woff[NIW];
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
        const int g = (wv * NIX + i) * 64 + lane;
        const int row = g >> 3, cp = (g & 7) ^ (row & 7);
        if (LINEAR) {
            int64_t rem = a.M - (int64_t)mt * BM;            // rows left in this tile (>= 1)
            const int r = (int64_t)row < rem ? row : (int)rem - 1;
            xoff[i] = (uint32_t)r * (uint32_t)cin * 2u + cp * 16;
        } else {
            const int h = th * (4 * WM) + (row >> 6) * 4 + ((row >> 4) & 3);
            const int w = tw * kTileW + (row & 15);
            xoff[i] = ((uint32_t)(h + 1) * (uint32_t)a.in_sh + (uint32_t)(w + 1) * (uint32_t)cin) * 2u + cp * 16;
        }
    }
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int g = (wv * NIW + i) * 64 + lane;
        const int row = g >> 3, cp = (g & 7) ^ (row & 7);
        woff[i] = (uint32_t)row * (uint32_t)cin * 2u + cp * 16;
    }

    const int kc_steps = cin / kBK;
    const int nk = TAPS * kc_steps;

    // uniform source offsets of K step k. Order: 64-channel chunk outermost, the 9 taps innermost, so
    // a block re-reads the same 18x18-pixel x 128-byte halo (41 KB) nine times in a row out of L1/L2.
    // (Tap-major order swept all Cin per tap: reuse distance 256 KB x 32 blocks per XCD >> 4 MB L2,
    // measured 17.4 GB FETCH_SIZE per 512->512@16 launch against 2.4 GB of input.)
    auto step_offsets = [&](int k, int64_t& wo, int64_t& xo) {
        int tap = 0, kc = k;
        if (TAPS > 1) { kc = k / TAPS; tap = k - kc * TAPS; }
        xo = (int64_t)kc * (kBK * 2);
        if (TAPS > 1) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            xo += ((int64_t)dy * a.in_sh + (int64_t)dx * cin) * 2;
        }
        wo = ((int64_t)tap * a.CoutPad * cin + (int64_t)kc * kBK) * 2;
    };
    // one 1-KiB LDS-DMA piece (idx < NIW: weight tile, else pixel tile) into buffer buf
    auto stage_piece = [&](const char* wsrc, const char* xsrc, int buf, int idx) {
        if (idx < NIW) {
            char* wdst = smem + buf * TILE_BYTES + (wv * NIW) * 1024;
            if (GEN_ASM_DMA) glds16_asm_s(wsrc, woff[idx], wdst + idx * 1024);
            else __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + woff[idx]), (lptr_t)(wdst + idx * 1024), 16, 0, 0);
        } else {
            char* xdst = smem + buf * TILE_BYTES + BN * 128 + (wv * NIX) * 1024;
            if (GEN_ASM_DMA) glds16_asm_s(xsrc, xoff[idx - NIW], xdst + (idx - NIW) * 1024);
            else __builtin_amdgcn_global_load_lds((gptr_t)(xsrc + xoff[idx - NIW]), (lptr_t)(xdst + (idx - NIW) * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int k, int buf) {
        int64_t wo, xo;
        step_offsets(k, wo, xo);
#pragma unroll
        for (int i = 0; i < NIW + NIX; ++i) stage_piece(wbase + wo, xbase + xo, buf, i);
    };

    f32x4 acc[JT][4];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment offsets inside a tile: row (lane&15), logical chunk ks*4 + (lane>>4)
    const int q = lane >> 4;
    const int frow = lane & 15;
    const int foff0 = frow * 128 + (((0 + q) ^ (lane & 7)) << 4);
    const int foff1 = frow * 128 + (((4 + q) ^ (lane & 7)) << 4);

    stage(0, 0);
    if (GEN_PRIO) __builtin_amdgcn_s_setprio(1);
    if (!PIPE) {
        for (int k = 0; k < nk; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char* wt = smem + (k & 1) * TILE_BYTES + (wn * WC) * 128;
            const char* xt = smem + (k & 1) * TILE_BYTES + BN * 128 + (wm * 64) * 128;
            if (GEN_ROLL && JT == 8) {
                f16x8 ar[3][2], bq[2][4];
                auto read_a = [&](int g, f16x8 (&dst)[2]) {
                    const char* base = wt + ((g >> 2) ? foff1 : foff0);
                    dst[0] = *(const f16x8*)(base + (2 * (g & 3)) * 2048);
                    dst[1] = *(const f16x8*)(base + (2 * (g & 3) + 1) * 2048);
                };
                auto read_b = [&](int ks, f16x8 (&dst)[4]) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) dst[n] = *(const f16x8*)(xt + n * 2048 + (ks ? foff1 : foff0));
                };
                __builtin_amdgcn_sched_barrier(0);
                read_b(0, bq[0]);
                read_a(0, ar[0]);
                __builtin_amdgcn_sched_barrier(0);
                read_a(1, ar[1]);
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < nk) stage(k + 1, (k + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g + 2 < 8) read_a(g + 2, ar[(g + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 1) { read_b(1, bq[1]); __builtin_amdgcn_sched_barrier(0); }
                    if (!GEN_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[2 * (g & 3) + jj][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[g % 3][jj], bq[g >> 2][n],
                                                                                             acc[2 * (g & 3) + jj][n], 0, 0, 0);
                    if (!GEN_PRIO) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            if (k + 1 < nk) stage(k + 1, (k + 1) & 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int fo = ks ? foff1 : foff0;
                f16x8 af[JT], bf[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[n] = *(const f16x8*)(xt + n * 2048 + fo);
#pragma unroll
                for (int j = 0; j < JT; ++j) af[j] = *(const f16x8*)(wt + j * 2048 + fo);
                if (!GEN_PRIO) __builtin_amdgcn_s_setprio(1);     // keeps the MFMA cluster together (measured +4 %)
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], bf[n], acc[j][n], 0, 0, 0);
                if (!GEN_PRIO) __builtin_amdgcn_s_setprio(0);
            }
        }
    } else {
        // Finely interleaved K step: 2*JT groups of {prefetch the next A fragment, one LDS-DMA piece of
        // the next K step every other group, 4 MFMAs}. The two waves that share a SIMD then mix memory
        // issue and matrix work instead of bursting both at the barrier. Measured on MI355X (round 1,
        // config 2): 8.25 ms vs 7.75-8.05 ms per dominant launch for the burst form, so it is OFF by
        // default (HCTR_PIPE=1 enables it for A/B runs).
        constexpr int NG = 2 * JT;
        constexpr int NP = NIW + NIX;
        for (int k = 0; k < nk; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // next step's DMA is unconditional (branch-free loop): on the last step it re-stages that
            // step into the idle buffer, which nobody reads (drained before the epilogue reuses LDS).
            const int kn = k + 1 < nk ? k + 1 : k;
            int64_t wo_n, xo_n;
            step_offsets(kn, wo_n, xo_n);
            const char* wsrc_n = wbase + wo_n;
            const char* xsrc_n = xbase + xo_n;
            const int buf_n = (k + 1) & 1;
            const char* wt = smem + (k & 1) * TILE_BYTES + (wn * WC) * 128;
            const char* xt = smem + (k & 1) * TILE_BYTES + BN * 128 + (wm * 64) * 128;
            f16x8 bfr[2][4], afr[2];
#pragma unroll
            for (int n = 0; n < 4; ++n) bfr[0][n] = *(const f16x8*)(xt + n * 2048 + foff0);
            afr[0] = *(const f16x8*)(wt + foff0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int ks = g / JT, j = g % JT;
                if (g + 1 < NG) {
                    const int ks1 = (g + 1) / JT, j1 = (g + 1) % JT;
                    afr[(g + 1) & 1] = *(const f16x8*)(wt + j1 * 2048 + (ks1 ? foff1 : foff0));
                }
                if (g == JT / 2) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bfr[1][n] = *(const f16x8*)(xt + n * 2048 + foff1);
                }
                if ((g * NP) / NG != ((g + 1) * NP) / NG)      // NP pieces spread over NG groups
                    stage_piece(wsrc_n, xsrc_n, buf_n, (g * NP) / NG);
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[g & 1], bfr[ks][n], acc[j][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (GEN_PRIO) __builtin_amdgcn_s_setprio(0);
    conv_epilogue<WN, WM, JT, LINEAR, SPLIT>(a, acc, smem, tid, lane, wn, wm, n0, mt, img, th, tw,
                                             th * (4 * WM) + wm * 4, tw * kTileW);
}

// HCTR_DBG (timing experiments, results INVALID) applies to every conv launch - unless HCTR_DBG_LAYER names ONE layer, in
// which case the engine sets ConvArgs::dbg for that layer only (the others then run on real data at the real clock)
static int env_dbg() {
    if (getenv("HCTR_DBG_LAYER")) return 0;
    const char* e = getenv("HCTR_DBG");
    return e ? atoi(e) : 0;
}

// hipFuncSetAttribute is per device: remember which devices already raised a kernel's LDS limit
// (a process may own contexts on several GPUs).
static hipError_t raise_lds_limit(const void* fn, int bytes, bool (&done)[64]) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

// -------------------------------------------------------------------------------------------
// conv3x3_halo: the 3x3 layers on the 256-cout x (16x16-pixel) tile with the pixel operand staged
// ONCE per 64-channel chunk. The generic kernel re-stages a shifted 16x16 window for each of the
// 9 taps (8 LDS-DMA pieces per wave per K step); on MI355X the burst of DMA issue after every
// barrier is what idles the matrix pipe (SQ counters: MFMA busy 56 %, no LDS conflicts, fabric
// traffic irrelevant). Here a block keeps an 18x18-pixel halo x 64 channels (rows of 128 B, row
// stride 20 pixels so the swizzle key stays cheap) in LDS and reads the tap-shifted B fragments from
// it, so only the 32 KB weight tile streams per K step: 4 + 6/9 pieces per wave per step instead of 8.
// LDS: 2 x 32 KB weights (double-buffered per K step) + 2 x 45 KB halo (double-buffered per chunk).
// -------------------------------------------------------------------------------------------
constexpr int kHaloCols = 20;                            // row stride in pixels (18 used)
constexpr int kHaloRows = 18;
constexpr int kHaloBytes = kHaloRows * kHaloCols * 128;  // 46080
constexpr int kHaloPieces = kHaloRows * kHaloCols / 8;   // 45 one-KiB DMA pieces
constexpr int kHaloLds = 2 * 32768 + 2 * kHaloBytes;     // 157696

__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const ConvArgs a) {
    constexpr int WN = 2, WM = 4, JT = 8, BN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv / WM, wm = wv % WM;

    const int total = a.mtiles * a.ntiles;
    int lin;
    {
        const int id = blockIdx.x, xcd = id & 7, s = id >> 3;
        const int q = total >> 3, r = total & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
    }
    const int nt = lin % a.ntiles;
    const int mt = lin / a.ntiles;
    const int n0 = nt * BN;
    const int tw = mt % a.tilesW;
    const int t2 = mt / a.tilesW;
    const int th = t2 % a.tilesH;
    const int img = t2 / a.tilesH;
    const int cin = a.Cin;
    const int nkc = cin / kBK;

    // ---- DMA sources ------------------------------------------------------------------------
    // halo origin = padded pixel (th*16, tw*16) = output pixel (th*16 - 1, tw*16 - 1)
    const char* xbase = (const char*)(a.x + img * a.in_sb + (int64_t)(th * 16) * a.in_sh + (int64_t)(tw * 16) * cin);
    const char* wbase = (const char*)(a.w + (int64_t)n0 * cin);
    // Asymmetric staging: only waves 0..3 issue LDS-DMA (8 weight pieces per K step + 2 halo pieces on
    // the first six steps of a chunk, each); their SIMD partners 4..7 go straight to their fragment
    // reads and MFMAs after the barrier, so the matrix pipe runs the partner's work while the loader
    // wave pays the DMA issue cost, then the loader's own.
    const bool loader = wv < 4;
    uint32_t woff[8], hoff[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int g = ((wv & 3) * 8 + i) * 64 + lane;
        const int row = g >> 3, cp = (g & 7) ^ (row & 7);
        woff[i] = (uint32_t)row * (uint32_t)cin * 2u + cp * 16;
    }
#pragma unroll
    for (int r = 0; r < 12; ++r) {                          // piece ((r>>1)*4 + wv)*2 + (r&1), step r>>1
        const int piece = ((r >> 1) * 4 + (wv & 3)) * 2 + (r & 1);
        const int g = piece * 64 + lane;
        const int row = g >> 3, cp = (g & 7) ^ (row & 7);
        int hy = row / kHaloCols, hx = row - hy * kHaloCols;
        if (hx > 17) hx = 17;                               // pad columns: any valid address
        if (hy > 17) hy = 17;                               // pieces >= 45 are never issued
        hoff[r] = ((uint32_t)hy * (uint32_t)a.in_sh + (uint32_t)hx * (uint32_t)cin) * 2u + cp * 16;
    }
    const int q = lane >> 4, c = lane & 15;
    const int aoff0 = c * 128 + (((0 + q) ^ (lane & 7)) << 4);
    const int aoff1 = c * 128 + (((4 + q) ^ (lane & 7)) << 4);

    auto stage_weights = [&](int kc, int tap, int buf) {
        const char* src = wbase + ((int64_t)tap * a.CoutPad * cin + (int64_t)kc * kBK) * 2;
        char* dst = smem + buf * 32768 + ((wv & 3) * 8) * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            glds16_asm(src + woff[i], dst + i * 1024);
    };
    auto stage_halo_piece = [&](int kc, int buf, int piece, uint32_t off) {
        if (piece < kHaloPieces) {                          // wave-uniform
            const char* src = xbase + (int64_t)kc * (kBK * 2);
            char* dst = smem + 65536 + buf * kHaloBytes + piece * 1024;
            glds16_asm(src + off, dst);
        }
    };

    f32x4 acc[JT][4];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (loader) {
#pragma unroll
        for (int r = 0; r < 12; ++r) stage_halo_piece(0, 0, ((r >> 1) * 4 + wv) * 2 + (r & 1), hoff[r]);
        stage_weights(0, 0, 0);
    }

    for (int kc = 0; kc < nkc; ++kc) {
        const bool next_chunk = kc + 1 < nkc;
        const int hbuf = 65536 + (kc & 1) * kHaloBytes + (wm * 4) * (kHaloCols * 128);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int k = kc * 9 + tap;
            // Retire the previous step's DMA. Steps 0..4 of a chunk issue (after the 8 weight pieces) two
            // pieces of the NEXT chunk's halo from every loader wave; they are cold (HBM) and not needed
            // for several steps, so they may stay in flight across this barrier: vmcnt(2) retires
            // everything older (vmcnt counts in issue order). All other steps drain completely.
            if (tap >= 1 && tap <= 5 && next_chunk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // B fragments: halo pixel row index hr = (wm*4 + n + 1 + dy)*20 + (c + 1 + dx); swizzle key
            // hr & 7 = ((c+1+dx) & 7) ^ 4*((n+1+dy) & 1)  (20 = 4 mod 8, +4 mod 8 = ^4). With the chunk
            // index ks*4 + q, the four (ks, n parity) cases need only two per-lane offsets, v0 and v0^64.
            const int tdy = tap / 3, dx = tap - tdy * 3 - 1;        // tdy = dy + 1
            const int u = c + 1 + dx;
            const int v0 = u * 128 + (((q ^ (u & 7) ^ ((tdy & 1) << 2)) & 7) << 4);
            const char* hb = smem + hbuf + tdy * (kHaloCols * 128);
            const char* be = hb + v0;           // ks=0 & n even, ks=1 & n odd
            const char* bo = hb + (v0 ^ 64);    // ks=0 & n odd,  ks=1 & n even
            const char* wt = smem + (k & 1) * 32768 + (wn * 128) * 128;
            // Rolling fragment pipeline: 8 groups of 8 MFMAs (2 A fragments x 4 B fragments). A pairs
            // are read two groups (16 MFMAs = 256 cycles) ahead into a 3-slot ring, the next ks's B
            // fragments during groups 1-2, so only the first 8 reads after the barrier are exposed.
            f16x8 ar[3][2], bq[2][4];
            auto read_a = [&](int g, f16x8 (&dst)[2]) {
                const int ks = g >> 2, jp = g & 3;
                const char* base = wt + (ks ? aoff1 : aoff0);
                dst[0] = *(const f16x8*)(base + (2 * jp) * 2048);
                dst[1] = *(const f16x8*)(base + (2 * jp + 1) * 2048);
            };
            auto read_b = [&](int ks, f16x8 (&dst)[4]) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    dst[n] = *(const f16x8*)((((n & 1) ^ ks) ? bo : be) + n * (kHaloCols * 128));
            };
            __builtin_amdgcn_sched_barrier(0);
            read_b(0, bq[0]);
            read_a(0, ar[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_a(1, ar[1]);
            __builtin_amdgcn_sched_barrier(0);
            // The DMA of the next step is issued AFTER the first fragment reads so its issue cost overlaps
            // their LDS latency (the asm DMA is invisible to the compiler's waitcnt bookkeeping).
            if (loader) {
                // next K step's weights (clamped on the very last step: re-stages into the idle buffer)
                int kc1 = kc, tap1 = tap + 1;
                if (tap1 == 9) { tap1 = 0; kc1 = next_chunk ? kc + 1 : kc; }
                if (a.dbg == 1) { kc1 = 0; tap1 = 0; }
                if (a.dbg != 2) stage_weights(kc1, tap1, (k + 1) & 1);
                // next chunk's halo: two pieces per loader wave on steps 0..5 (branch-free offset select)
                if (next_chunk && tap < 6 && a.dbg != 2) {
                    uint32_t h0 = hoff[0], h1 = hoff[1];
                    h0 = tap == 1 ? hoff[2] : h0;  h1 = tap == 1 ? hoff[3] : h1;
                    h0 = tap == 2 ? hoff[4] : h0;  h1 = tap == 2 ? hoff[5] : h1;
                    h0 = tap == 3 ? hoff[6] : h0;  h1 = tap == 3 ? hoff[7] : h1;
                    h0 = tap == 4 ? hoff[8] : h0;  h1 = tap == 4 ? hoff[9] : h1;
                    h0 = tap == 5 ? hoff[10] : h0; h1 = tap == 5 ? hoff[11] : h1;
                    const int p0 = (tap * 4 + wv) * 2;
                    const int kcs = a.dbg == 1 ? 0 : kc + 1;
                    stage_halo_piece(kcs, (kc + 1) & 1, p0, h0);
                    stage_halo_piece(kcs, (kc + 1) & 1, p0 + 1, h1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int ks = g >> 2, jp = g & 3;
                if (g + 2 < 8) read_a(g + 2, ar[(g + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 1) { read_b(1, bq[1]); __builtin_amdgcn_sched_barrier(0); }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[2 * jp + jj][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[g % 3][jj], bq[ks][n],
                                                                                    acc[2 * jp + jj][n], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last (redundant) DMA has landed
    conv_epilogue<WN, WM, JT, false, false>(a, acc, smem, tid, lane, wn, wm, n0, mt, img, th, tw, th * 16 + wm * 4,
                                            tw * 16);
}

// -------------------------------------------------------------------------------------------
// conv3x3_halo4: the same halo-reuse scheme as conv3x3_halo_kernel, but as a 4-wave workgroup
// (128 couts x 16x16 pixels, each wave 128 couts x 64 pixels) with 2 x 16 KB weight buffers and ONE
// 45 KB halo buffer = 77 KB of LDS, so TWO workgroups share a CU. The two waves on a SIMD then
// belong to different workgroups with independent barriers and drift out of phase, which hides
// the per-step barrier / DMA-latency bubble that the 8-wave lockstep structure exposes (SQ counters,
// DESIGN.md). The halo of the next chunk can only be fetched after the last tap of the current one
// (single buffer); the partner workgroup covers that stall.
// -------------------------------------------------------------------------------------------
constexpr int kHalo4Lds = 2 * 16384 + kHaloBytes;          // 78848

// GEOM 0: 16 rows x 16 columns (halo 18 x 18, row stride 20); GEOM 1: 8 rows x 32 columns for the
// H = 8 stage (halo 10 x 34, row stride 36). Both strides are 4 (mod 8) and both halos are 360 rows.
// STAMP: diagnostic instance (hctr_debug_stamps). Thread 0 of every workgroup writes wall-clock stamps
// (s_memrealtime, 100 MHz) of its phases plus its hardware placement to a buffer no other code reads:
//   [0] entry  [1] prologue DMA issued  [2] first operands landed (extra wait + barrier, stamp build only)
//   [3] K loop done  [4] epilogue done (stores issued)  [5] stores drained  [6] HW_ID  [7] XCC_ID
//   [8..11] inside the epilogue: bias added, residual applied, values rounded, SE sums written
// DSFUSE: a block's 1x1 downsample branch (models/handwritten_ctr_model.py:101-108, used at :49-50,57) runs as a
// pre-phase of conv2's K loop instead of as its own launch: acc = Wd * x over the block input's channels (centre
// tap of x's halo), then acc <- (acc + bd) / s + b2, then the 3x3 taps of conv2 accumulate on top and the epilogue
// multiplies by s: s * (W2*t + b2) + (Wd*x + bd). The residual is neither written nor read back (4.2 GB per launch).
#ifndef LEAN
#define LEAN 1         // K loop with its nine taps unrolled (tap / row / parity arithmetic and the step's branches fold into
                       // constants), ONE per-lane weight offset (the piece stride moves into the scalar base) and one M0
                       // save / restore per four pieces. -DLEAN=0: the rolled loop of rounds 1-3 (A/B, bit-identical).
#endif
#ifndef BPRE
#define BPRE 1         // (with LEAN) next tap's first pixel fragments are read during the current step, see mma_step
#endif
#ifndef BPRE_AT
#define BPRE_AT 4      // ... behind the MFMAs of this group (4..7)
#endif
#ifndef EARLY_HALO
#define EARLY_HALO 1   // (with LEAN, plain f16 instances) next chunk's halo reload issued inside the last tap, see the tap loop
#endif
#ifndef SPLIT_FAST
#define SPLIT_FAST 1   // fragment prefetch + early halo request in the f16x3 instances too (their K loop is the same code)
#endif
#ifndef EARLY_LGKM
#define EARLY_LGKM 0
#endif
#ifndef EARLY_AT
#define EARLY_AT 1     // ... behind this MFMA group of tap 8
#endif
#ifndef DS_BPRE
#define DS_BPRE 1      // fragment prefetch + early halo request in the main loop of the fused-downsample instances too
#endif
#ifndef RESPRE_BPRE
#define RESPRE_BPRE 0  // BPRE in the residual-in-prologue instances too (spills eight halo offsets as of this writing)
#endif
#ifndef RTOUCH
#define RTOUCH 0       // residual pre-touch experiment (HCTR_RTOUCH=1 needs -DRTOUCH=1; measured neutral, see below)
#endif
// RESPRE: conv2 of an identity block (BasicBlock :54-58, out = relu(o * s + x)). The SE scale s is known before conv2 runs
// (se_premean), so the residual x is fetched in the PROLOGUE - its HBM round trip overlaps the first operands' DMA wait -
// and the accumulators start at x / max(s, floor) + bias; the epilogue multiplies by max(s, floor), exactly what the
// fused downsample does with its 1x1 branch. The epilogue's residual phase (16 loads per lane issued behind the partner
// workgroup's DMA stream: 4.5 of its 7.6 us, profiles/r03_lean_workgroup_phases_*) disappears - and reappears in the
// prologue: measured slower as a whole (see launch_conv_halo4_t), kept as an opt-in experiment.
template <int GEOM, bool SPLIT, bool PERSIST, bool STAMP = false, bool DSFUSE = false, bool RESPRE = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo4_kernel(const ConvArgs a) {
    static_assert(!DSFUSE || !PERSIST, "downsample fusion: non-persistent instances only");
    static_assert(!RESPRE || (!DSFUSE && !PERSIST && !SPLIT), "residual-in-prologue: plain f16 instances only");
    constexpr int WN = 1, WM = 4, JT = 8, BN = 128;
    constexpr int TR = GEOM ? 8 : 16, TC = GEOM ? 32 : 16;       // tile rows / columns
    constexpr int S = TC + 4;                                      // halo row stride in pixels
    static_assert((TR + 2) * S * 128 == kHaloBytes, "halo footprint");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    auto stamp = [&](int i) {
        if (STAMP && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    if (PRIO_MODE == 3) __builtin_amdgcn_s_setprio(2);
    if (PRIO_MODE == 4) __builtin_amdgcn_s_setprio(1);
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv;
    const int cin = a.Cin;
    const int nkc = (a.dbg & 64) ? 0 : cin / kBK;       // dbg 64: timing experiment without the K loop
    const int nk = 9 * nkc;
    // (not with RTOUCH builds: the pre-touch experiment lands its dummy loads in the same LDS scratch as the offset table;
    //  not in persistent instances: their epilogue overwrites the table after every tile)
    constexpr bool kEarly = LEAN && EARLY_HALO && !RTOUCH && !(DSFUSE && !DS_BPRE) && !(SPLIT && !SPLIT_FAST) && !PERSIST && !STAMP && !RESPRE;

    // ---- tiles of this workgroup. Every XCD owns a contiguous run of the (pixel-tile major, cout-tile
    //      minor) order. Non-persistent: one tile per workgroup. Persistent: the workgroups of an XCD
    //      stride through its run, and a workgroup prefetches its NEXT tile's first halo and weights
    //      while the current tile's epilogue runs (the cold per-tile prologue is ~10-30 % of a launch).
    // every kernel argument the prologue needs is requested here, in ONE batch of scalar loads and one wait (the compiler
    // otherwise fetches them in three dependent rounds, each a scalar-cache round trip on the critical path to the first DMA)
    {
        const int k0 = a.mtiles, k1 = a.ntiles, k2 = a.tilesW, k3 = a.tilesH, k4 = a.Cin, k5 = a.in_sh, k6 = a.nt_shift,
                  k7 = a.th_shift, k8 = a.CoutPad;
        const uint64_t k9 = a.tw_magic;
        const int64_t k10 = a.in_sb;
        const half_t *k11 = a.x, *k12 = a.w;
        const float* k13 = a.bias;
        asm volatile("" ::"s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5), "s"(k6), "s"(k7), "s"(k8), "s"(k9), "s"(k10),
                     "s"(k11), "s"(k12), "s"(k13));
    }
    const int total = a.mtiles * a.ntiles;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tq = total >> 3, tr = total & 7;
    const int xbase_lin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int xcnt = tq + (xcd < tr ? 1 : 0);
    const int nloc = PERSIST ? (int)(gridDim.x >> 3) : 1;
    struct Tile { int n0, mt, img, th, tw; const char* xb; const char* wb; };
    auto tile_at = [&](int idx) {
        Tile t;
        const int lin = xbase_lin + idx;
        int nt, t2, th, img;
        if (a.nt_shift >= 0) { nt = lin & (a.ntiles - 1); t.mt = lin >> a.nt_shift; }
        else { nt = lin % a.ntiles; t.mt = lin / a.ntiles; }
        t.n0 = nt * BN;
        t2 = (int)(((uint64_t)(uint32_t)t.mt * a.tw_magic) >> 40);          // = mt / tilesW (launch_conv checks the range)
        const int tw = t.mt - t2 * a.tilesW;
        if (a.th_shift >= 0) { th = t2 & (a.tilesH - 1); img = t2 >> a.th_shift; }
        else { th = t2 % a.tilesH; img = t2 / a.tilesH; }
        t.img = img; t.th = th; t.tw = tw;
        t.xb = (const char*)(a.x + img * a.in_sb + (int64_t)(th * TR) * a.in_sh + (int64_t)(tw * TC) * cin);
        t.wb = (const char*)(a.w + (int64_t)t.n0 * cin);
        return t;
    };
    int tidx = local;
    if (PERSIST && tidx >= xcnt) return;            // (persistent grids may exceed a short XCD run; a plain grid is exact)
    // Persistent variant, DYNAMIC tile queue (a.tile_counter set): the first two tiles of a workgroup are the static
    // ones (local, local + nloc); every later tile index is drawn from the XCD's atomic counter one tile ahead, so the
    // draw's latency hides behind a whole K loop and the hardware dispatcher's balancing is kept.
    const bool dynamic = PERSIST && a.tile_counter != nullptr;
    int tnext = local + nloc;
    volatile int* lds_next = (volatile int*)(smem + kHalo4Lds + 4 * 128 * 4);       // two words behind the epilogue scratch

    const int wrow = GEOM ? (wm >> 1) * 4 : wm * 4;               // this wave's 4 x 16 patch inside the tile
    const int wcol = GEOM ? (wm & 1) * 16 : 0;
    uint32_t woff[4], hoff[12];
    // BPRE: with bq[0] living across the step boundary the loop sits at the 256-register limit, and a spilled halo offset is
    // reloaded in front of its DMA (a reload drains vmcnt: the K loop must not spill). The pieces' (hy, hx) pairs are
    // therefore also kept packed in six registers and the offsets rebuilt from them in tap 8's last four MFMA groups,
    // where bq[0] is free. hipcc is free to hoist that arithmetic (it is loop-invariant) and does for some instances; what
    // is checked is the outcome - tools/kernel_resources.sh must show scratch 0 for the four <GEOM, false, false, false, *>
    // instances after any change here (without this formulation GEOM 1 spills 8 bytes per lane, measured).
    uint32_t hpk[6], hcp16 = 0;
    const int q = lane >> 4, c = lane & 15;
    const int aoff0 = c * 128 + (((0 + q) ^ (lane & 7)) << 4);
    const int aoff1 = c * 128 + (((4 + q) ^ (lane & 7)) << 4);

    auto stage_w = [&](const char* wb, int rowcin, const uint32_t (&wo)[4], int kc, int tap, int buf) {
        const char* src = wb + ((int64_t)tap * a.CoutPad * rowcin + (int64_t)kc * kBK) * 2;
        char* dst = smem + buf * 16384 + (wv * 4) * 1024;
        if (LEAN) {                                          // piece i = 8 cout rows further on: same lane offset, scalar stride
            const int64_t ps = (int64_t)rowcin * 16;
            glds16x4_asm_s(src, src + ps, src + 2 * ps, src + 3 * ps, wo[0], dst, dst + 1024, dst + 2048, dst + 3072);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_asm_s(src, wo[i], dst + i * 1024);
    };
    auto stage_weights = [&](const char* wb, int kc, int tap, int buf) { stage_w(wb, cin, woff, kc, tap, buf); };
    auto stage_weights_piece = [&](const char* wb, int kc, int tap, int buf, int i) {
        const char* src = wb + ((int64_t)tap * a.CoutPad * cin + (int64_t)kc * kBK) * 2;
        glds16_asm_s(src, woff[i], smem + buf * 16384 + (wv * 4) * 1024 + i * 1024);
    };
    auto stage_halo = [&](const char* xb, int kc) {
        const char* src = xb + (int64_t)kc * (kBK * 2);
#pragma unroll
        for (int r = 0; r < 12; ++r)
            if (wv + 4 * r < kHaloPieces) glds16_asm_s(src, hoff[r], smem + 32768 + (wv + 4 * r) * 1024);
    };
    // per-lane DMA offsets of a source tensor with `rowcin` channels and `in_sh` elements per image row
    // (recomputed where needed - from an opaque lane id, so they are not kept alive across the epilogue)
    auto lane_offsets = [&](int rowcin, int in_sh, uint32_t (&wo)[4], bool with_halo) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = (wv * 4 + i) * 64 + ln;
            const int row = g >> 3, cp = (g & 7) ^ (row & 7);
            wo[i] = (uint32_t)row * (uint32_t)rowcin * 2u + cp * 16;
        }
        if (!with_halo) return;
#pragma unroll
        for (int r = 0; r < 12; ++r) {                          // piece wv + 4r
            const int g = (wv + 4 * r) * 64 + ln;
            const int row = g >> 3, cp = (g & 7) ^ (row & 7);
            int hy = row / S, hx = row - hy * S;
            if (hx > TC + 1) hx = TC + 1;                       // pad columns: any valid address
            if (hy > TR + 1) hy = TR + 1;                       // pieces >= 45 are never issued
            hoff[r] = ((uint32_t)hy * (uint32_t)in_sh + (uint32_t)hx * (uint32_t)rowcin) * 2u + cp * 16;
            hcp16 = cp * 16;                                    // (row & 7 does not depend on r: pieces are 32 rows apart)
            const uint32_t pk = (uint32_t)hy << 8 | (uint32_t)hx;
            if (r & 1) hpk[r >> 1] |= pk << 16;
            else hpk[r >> 1] = pk;
        }
    };
    // offsets of pieces r0 .. r0+2 back from the packed coordinates
    auto unpack_hoff = [&](int r0, uint32_t insh2, uint32_t cin2) {
#pragma unroll
        for (int r = r0; r < r0 + 3; ++r) {
            const uint32_t pk = hpk[r >> 1] >> ((r & 1) * 16);
            hoff[r] = ((pk >> 8) & 0xff) * insh2 + (pk & 0xff) * cin2 + hcp16;
        }
    };
    // EARLY_HALO: the twelve piece offsets are not kept in registers through the K loop; their lane-row part
    // (hy * in_sh + hx * cin, the same for the eight lanes of an LDS row) goes into a table in the epilogue's scratch
    // (unused until the loop ends), 4 waves x 12 pieces x 8 rows x 4 B = 1.5 KB; the lane's own 16-byte chunk term is added
    // back when a piece is issued. Called right after the lane_offsets() of the tensor the K loop reads.
    auto write_halo_table = [&]() {
        uint32_t* htab = (uint32_t*)(smem + kHalo4Lds);
        if ((lane & 7) == 0) {
            const uint32_t cp0 = (uint32_t)((lane >> 3) & 7) << 4;           // chunk term of lane & 7 == 0
#pragma unroll
            for (int r = 0; r < 12; ++r) htab[(wv * 12 + r) * 8 + (lane >> 3)] = hoff[r] - cp0;
        }
    };
    const int hbuf = 32768 + wrow * (S * 128);
    // fragment addresses of a tap inside the halo image
    // LEAN: all 18 (tap, even / odd) fragment addresses are six per-lane values (dx = -1, 0, 1; swizzle parity) plus
    // constants - spelled out so that the unrolled loop keeps six address registers, not one or two per tap
    int bx[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int u = wcol + c + d;
        const int v0 = u * 128 + (((q ^ (u & 7)) & 7) << 4);
        bx[d][0] = hbuf + v0;
        bx[d][1] = hbuf + (v0 ^ 64);
    }
    auto b_ptrs = [&](int tap, const char*& be, const char*& bo) {
        if (LEAN) {
            const int tdy = tap / 3, d = tap - tdy * 3;
            be = smem + bx[d][tdy & 1] + tdy * (S * 128);
            bo = smem + bx[d][(tdy & 1) ^ 1] + tdy * (S * 128);
            return;
        }
        const int tdy = tap / 3, dx = tap - tdy * 3 - 1;
        const int u = wcol + c + 1 + dx;
        const int v0 = u * 128 + (((q ^ (u & 7) ^ ((tdy & 1) << 2)) & 7) << 4);
        const char* hb = smem + hbuf + tdy * (S * 128);
        be = hb + v0;
        bo = hb + (v0 ^ 64);
    };

    Tile cur = tile_at(tidx);
    int kbase = 0;                                  // weight-buffer parity continues across tiles / phases
    bool first = true;

    int tile_no = 0;
    for (;;) {
        const bool has_next = PERSIST && (tnext < xcnt);
        const char* nxb = cur.xb;
        const char* nwb = cur.wb;
        if (has_next) {
            const Tile t = tile_at(tnext);
            nxb = t.xb;
            nwb = t.wb;
        }
        int drawn = 0;
        if (dynamic && tid == 0)                        // index of the tile after next (used one tile from now)
            drawn = 2 * nloc + __hip_atomic_fetch_add(a.tile_counter + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x4 acc[JT][4];
        // One K step: 64 MFMAs of this wave on the weight tile at `wt` and the pixel fragments at be / bo.
        // Rolling fragment pipeline: 8 groups of 8 MFMAs (2 A fragments x 4 B fragments); A pairs are read two
        // groups ahead into a 3-slot ring, the second half's B fragments during group 1; `stage_next` issues the
        // next step's weight DMA after the first reads so its issue cost overlaps their LDS latency.
        // BPRE (LEAN only): the pixel fragments of the NEXT tap's first half are read during this step's groups 4-7 into
        // bq[0] (dead after group 3) - the halo does not change inside a chunk - so only weight fragments are read between
        // the barrier and the first MFMA. `have_b0`: bq[0] was filled that way by the previous step; `nbe`: != nullptr ->
        // prefetch from (nbe, nbo).
        f16x8 bq[2][4];
        auto mma_step = [&](const char* wt, const char* be, const char* bo, auto&& stage_next, auto&& stage_piece,
                            auto&& after_group, bool have_b0 = false, const char* nbe = nullptr, const char* nbo = nullptr,
                            int bhalf_at = BHALF_AT) {
            f16x8 ar[RING][2];
            auto read_a = [&](int g, f16x8 (&dst)[2]) {
                const int ks = g >> 2, jp = g & 3;
                const char* base = wt + (ks ? aoff1 : aoff0);
                dst[0] = *(const f16x8*)(base + (2 * jp) * 2048);
                dst[1] = *(const f16x8*)(base + (2 * jp + 1) * 2048);
            };
            auto read_b = [&](int ks, f16x8 (&dst)[4]) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    dst[n] = *(const f16x8*)((((n & 1) ^ ks) ? bo : be) + n * (S * 128));
            };
            __builtin_amdgcn_sched_barrier(0);
            if (DMA_SPREAD == 3) { stage_next(); __builtin_amdgcn_sched_barrier(0); }      // (A/B: DMA before the first reads)
            if (!have_b0) read_b(0, bq[0]);
            read_a(0, ar[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_a(1, ar[1]);
            if (RING > 3) read_a(2, ar[2]);
            __builtin_amdgcn_sched_barrier(0);
            if (DMA_SPREAD != 3) stage_next();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int ks = g >> 2, jp = g & 3;
                if (g + RING - 1 < 8) read_a(g + RING - 1, ar[(g + RING - 1) % RING]);
                __builtin_amdgcn_sched_barrier(0);
                if (g == bhalf_at) { read_b(1, bq[1]); __builtin_amdgcn_sched_barrier(0); }
                if (!(NOPRIO) && (PRIO_MODE == 0 || PRIO_MODE == 3)) __builtin_amdgcn_s_setprio(1);
                if (PRIO_MODE == 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[2 * jp + jj][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            ar[g % RING][jj], bq[ks][n], acc[2 * jp + jj][n], 0, 0, 0);
                if (!(NOPRIO) && (PRIO_MODE == 0 || PRIO_MODE == 3)) __builtin_amdgcn_s_setprio(0);
                if (PRIO_MODE == 2) __builtin_amdgcn_s_setprio(1);
                if (DMA_SPREAD == 1 && g < 4) stage_piece(g);
                if (DMA_SPREAD == 2 && (g & 1) == 0) stage_piece(g >> 1);
                if (g == BPRE_AT && nbe != nullptr) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bq[0][n] = *(const f16x8*)(((n & 1) ? nbo : nbe) + n * (S * 128));
                }
                after_group(g);
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        if (DSFUSE) {
            // ---- pre-phase: acc = Wd * x (1x1: the centre tap of x's halo), ds_cin / 64 steps ----
            const int dcin = a.ds_cin, nds = dcin / kBK;
            const int tw0 = cur.tw, th0 = cur.th, img0 = cur.img;
            const char* dxb = (const char*)(a.ds_x + img0 * a.ds_in_sb + (int64_t)(th0 * TR) * a.ds_in_sh +
                                            (int64_t)(tw0 * TC) * dcin);
            const char* dwb = (const char*)(a.ds_w + (int64_t)cur.n0 * dcin);
            uint32_t woff_x[4];
            lane_offsets(dcin, a.ds_in_sh, woff_x, true);
            lane_offsets(cin, a.in_sh, woff, false);
            stage_halo(dxb, 0);
            stage_w(dwb, dcin, woff_x, 0, 0, 0);
            if (PRIO_MODE == 1 || PRIO_MODE == 2 || PRIO_MODE == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll 1
            for (int kc = 0; kc < nds; ++kc) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const char *be, *bo;
                b_ptrs(4, be, bo);
                mma_step(smem + (kc & 1) * 16384, be, bo, [&] {
                    if (kc + 1 < nds) stage_w(dwb, dcin, woff_x, kc + 1, 0, (kc + 1) & 1);
                    else stage_w(cur.wb, cin, woff, 0, 0, (kc + 1) & 1);            // conv2's first step
                }, [&](int) {}, [&](int) {});
                __builtin_amdgcn_s_barrier();                 // every wave has consumed this chunk's fragments
                asm volatile("" ::: "memory");
                if (kc + 1 < nds) {
                    stage_halo(dxb, kc + 1);
                } else {
                    lane_offsets(cin, a.in_sh, woff, true);   // from here on the halo holds conv2's input t
                    stage_halo(cur.xb, 0);
                    if (kEarly) write_halo_table();
                }
            }
            kbase = nds;
            // residual term and conv2's bias: acc <- (Wd*x + bd) / s + b2   (the epilogue multiplies by s)
            const float* sc = a.se_scale + (int64_t)img0 * a.Cout + cur.n0 + q * 8;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int o = acc_cout_offset(j);
                const f32x4 bd = *(const f32x4*)(a.ds_bias + cur.n0 + q * 8 + o);
                const f32x4 b2 = *(const f32x4*)(a.bias + cur.n0 + q * 8 + o);
                const f32x4 s4 = *(const f32x4*)(sc + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float inv = __builtin_amdgcn_rcpf(fmaxf(s4[i], kSeScaleFloor));
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[j][n][i] = fmaf(acc[j][n][i] + bd[i], inv, b2[i]);
                }
            }
        } else {
            if (PERSIST || first) lane_offsets(cin, a.in_sh, woff, true);
            if (first) {
                stage_halo(cur.xb, 0);
                stage_weights(cur.wb, 0, 0, 0);
                if (kEarly) write_halo_table();
                first = false;
                if (STAMP) {
                    stamp(1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    stamp(2);
                }
            }
            if (RESPRE) {
                // accumulators start at residual / max(s, floor) + bias (element order as in conv_epilogue's residual branch)
                const int w = cur.tw * TC + wcol + c;
                const bool wok = w < a.out_wlimit;
                const half_t* rb = a.resid + a.out_off + cur.img * a.out_sb + (int64_t)w * a.out_sw + cur.n0 + q * 8 +
                                   (int64_t)(cur.th * TR + wrow) * a.out_sh;
                const float* sc = a.se_scale + (int64_t)cur.img * a.Cout + cur.n0 + q * 8;
                f16x8 rlo[JT / 4][4], rhi[JT / 4][4];
#pragma unroll
                for (int cb = 0; cb < JT / 4; ++cb)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const half_t* r = rb + (int64_t)n * a.out_sh + cb * 64;
                        rlo[cb][n] = wok ? *(const f16x8*)r : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                        rhi[cb][n] = wok ? *(const f16x8*)(r + 32) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    }
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const f32x4 b4 = *(const f32x4*)(a.bias + cur.n0 + q * 8 + acc_cout_offset(j));
                    const f32x4 s4 = *(const f32x4*)(sc + acc_cout_offset(j));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float inv = __builtin_amdgcn_rcpf(fmaxf(s4[i], kSeScaleFloor));
                        const int e = (j & 3) * 4 + i;
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const float r = e < 8 ? (float)rlo[j >> 2][n][e] : (float)rhi[j >> 2][n][e - 8];
                            acc[j][n][i] = fmaf(r, inv, b4[i]);
                        }
                    }
                }
            } else {
                // accumulators start at the folded-BN bias (its load hides behind the first operands' DMA)
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const f32x4 b4 = *(const f32x4*)(a.bias + cur.n0 + q * 8 + acc_cout_offset(j));
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[j][n] = b4;
                }
            }
        }

        // Retire every scalar (kernarg) load before the loop. A load still pending at the loop header keeps
        // hipcc's counter model "dirty" on every iteration (scalar loads return out of order), and it then
        // drains lgkmcnt(0) at the first MFMA group of each K step instead of the counted wait.
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0), vmcnt/expcnt untouched
        // Residual pre-touch (A/B experiment, HCTR_RTOUCH=1, OFF by default): three steps before the end every thread
        // touches the two 128-byte lines of one pixel of conv2's residual tile with 4-byte LDS-DMA loads into the
        // epilogue's scratch (no register destination; issued after that step's weight pieces, so the next step's counted
        // wait leaves them in flight). Measured (profiles/r03_overhead_experiments.txt): the epilogue's residual phase
        // drops 3.6 -> 2.7 us, the K loop grows by as much, the step time is unchanged - the phase is the latency of 16
        // dependent-free loads through a busy memory pipeline, not an HBM miss.
        const bool rtouch = RTOUCH && !SPLIT && !DSFUSE && a.rtouch && a.resid != nullptr && nk >= 6;
        const int ktouch = nk - 3;
        // residual prefetch during the last K step (f16, identity blocks; a.rpre: A/B switch HCTR_RPRE)
        const bool rpre = RPRE && !SPLIT && !DSFUSE && !PERSIST && a.rpre && a.resid != nullptr && a.se_scale != nullptr && nk >= 1;

        if (PRIO_MODE == 1 || PRIO_MODE == 2 || PRIO_MODE == 4) __builtin_amdgcn_s_setprio(1);
        if (PRIO_MODE == 3) __builtin_amdgcn_s_setprio(0);
        for (int kc = 0; kc < nkc; ++kc) {
            const bool next_chunk = kc + 1 < nkc;
            // with the residual prefetch the very last K step (tap 8 of the last chunk) runs after this loop, where the
            // loop's per-lane DMA offsets are dead and their registers can hold the residual
            const int ntap = (rpre && !next_chunk) ? 8 : 9;
#if LEAN
#pragma unroll
#else
#pragma unroll 1
#endif
            for (int tap = 0; tap < ntap; ++tap) {
                const int k = kc * 9 + tap;
                if (rtouch && k == ktouch + 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(a.dbg & 1024)) __builtin_amdgcn_s_barrier();       // dbg 1024: timing experiment without the per-step barrier
                asm volatile("" ::: "memory");
                const char *be, *bo;
                b_ptrs(tap, be, bo);
                const bool more = k + 1 < nk;
                const bool wrap = tap == 8;
                const int tap1 = more ? (wrap ? 0 : tap + 1) : 0;
                const int kc1 = more ? (wrap ? kc + 1 : kc) : 0;
                const char* wsrc = more ? cur.wb : nwb;
                const bool bpre = LEAN && BPRE && !(DSFUSE && !DS_BPRE) && !(SPLIT && !SPLIT_FAST) && !PERSIST && !STAMP && !(RESPRE && !RESPRE_BPRE);      // (the others would spill)
                const char *nbe = nullptr, *nbo = nullptr;
                if (bpre && tap < 8) b_ptrs(tap + 1, nbe, nbo);
                const bool early = kEarly;
                const bool reload = kEarly ? next_chunk : ((next_chunk || has_next) && !(a.dbg & 32));
                mma_step(smem + ((kbase + k) & 1) * 16384, be, bo, [&] {
                    // the next K step's weights into the other buffer; on a tile's last step that is the
                    // next tile's first step (persistent variant only).
                    // nothing to stage on the very last step: no DMA is then in flight when the epilogue
                    // starts, so the workgroup can retire without waiting for its output stores
                    if ((DMA_SPREAD == 0 || DMA_SPREAD == 3) && (more || has_next) && !(a.dbg & 512))    // dbg 512: no weight DMA
                        stage_weights(wsrc, kc1, tap1, (kbase + k + 1) & 1);
                    if (rtouch && k == ktouch) {
                        const int prow = GEOM ? (tid >> 5) : (tid >> 4), pcol = GEOM ? (tid & 31) : (tid & 15);
                        const char* rp = (const char*)(a.resid + a.out_off + cur.img * a.out_sb +
                                                       (int64_t)(cur.th * TR + prow) * a.out_sh +
                                                       (int64_t)(cur.tw * TC + pcol) * a.out_sw + cur.n0);
                        char* sc = smem + kHalo4Lds + wv * 512;
                        glds4_asm(rp, sc);
                        glds4_asm(rp + 128, sc + 256);
                    }
                }, [&](int i) {
                    if (more || has_next) stage_weights_piece(wsrc, kc1, tap1, (kbase + k + 1) & 1, i);
                }, [&](int g) {
                    if (bpre && !early && tap == 8 && g >= 4) unpack_hoff(3 * (g - 4), (uint32_t)a.in_sh * 2u, (uint32_t)cin * 2u);
                    if (early && tap == 8 && g == EARLY_AT && reload) {
                        // EARLY_HALO: the last tap reads its second-half pixel fragments at group 0, so after group EARLY_AT
                        // every fragment of the chunk is in registers: one extra barrier, and the next chunk's halo streams
                        // in during the remaining MFMA groups instead of after them
                        // (in-order LDS returns: with EARLY_LGKM = 2 only the two weight-fragment reads issued after the
                        // pixel fragments may still be in flight)
                        if (EARLY_LGKM == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        // pieces wv + 4r: r <= 10 exist for every wave (45 pieces), r = 11 for wave 0 only
                        static_assert(kHaloPieces == 45, "halo piece count");
                        const char* hsrc = (next_chunk ? cur.xb + (int64_t)(kc + 1) * (kBK * 2) : nxb);
                        char* d = smem + 32768 + wv * 1024;
                        int ln = lane;
                        asm volatile("" : "+v"(ln));                       // (nothing of this is hoisted out of the loop)
                        const uint32_t* hrow = (const uint32_t*)(smem + kHalo4Lds) + wv * 96 + (ln >> 3);
                        const uint32_t cpl = (uint32_t)((ln ^ (ln >> 3)) & 7) << 4;
                        uint32_t ho[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) ho[r] = hrow[r * 8] + cpl;
                        glds16x4v_asm_s<4096>(hsrc, ho[0], ho[1], ho[2], ho[3], d);
#pragma unroll
                        for (int r = 0; r < 4; ++r) ho[r] = hrow[(4 + r) * 8] + cpl;
                        glds16x4v_asm_s<4096>(hsrc, ho[0], ho[1], ho[2], ho[3], d + 16384);
#pragma unroll
                        for (int r = 0; r < 4; ++r) ho[r] = hrow[(8 + r) * 8] + cpl;
#pragma unroll
                        for (int r = 8; r < 11; ++r) glds16_asm_s(hsrc, ho[r - 8], d + r * 4096);
                        if (wv == 0) glds16_asm_s(hsrc, ho[3], d + 11 * 4096);
                    }
                }, bpre && tap > 0, nbe, nbo, (early && tap == 8) ? 0 : BHALF_AT);
            }
            if ((next_chunk || has_next) && !(a.dbg & 32) && !kEarly) {     // dbg 32: timing experiment without the reload
                // single halo buffer: every wave has consumed its last B fragments of this chunk (they fed
                // the MFMAs above), so after this barrier the buffer may be overwritten - with the next
                // chunk, or with the next tile's first chunk (whose latency then hides behind the epilogue).
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (next_chunk) stage_halo(cur.xb, kc + 1);
                else stage_halo(nxb, 0);
            }
        }
        f16x8 rlo[2][4], rhi[2][4];
        if (rpre) {
            // LAST K step of a conv2 with an identity residual (non-persistent: nothing is staged any more): the 16
            // residual vectors of the epilogue are fetched here, two after each MFMA group, into the registers the
            // step's operand fragments and the loop's DMA offsets leave behind - their latency (2.7-3.6 us of a 6.6 us
            // epilogue when issued there) hides under the step's MFMAs
            const half_t* rbase = a.resid + a.out_off + cur.img * a.out_sb + (int64_t)(cur.tw * TC + wcol + c) * a.out_sw +
                                  cur.n0 + q * 8 + (int64_t)(cur.th * TR + wrow) * a.out_sh;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char *be, *bo;
            b_ptrs(8, be, bo);
            mma_step(smem + ((kbase + nk - 1) & 1) * 16384, be, bo, [&] {}, [&](int) {}, [&](int g) {
                if (g < RPRE_GROUPS) {
                    const half_t* r = rbase + (int64_t)(g & 3) * a.out_sh + (g >> 2) * 64;
                    rlo[g >> 2][g & 3] = *(const f16x8*)r;
                    rhi[g >> 2][g & 3] = *(const f16x8*)(r + 32);
                }
            });
        }
        if (PRIO_MODE == 1 || PRIO_MODE == 2 || PRIO_MODE == 4) __builtin_amdgcn_s_setprio(0);
        if (PRIO_MODE == 3) __builtin_amdgcn_s_setprio(2);
        // the epilogue's scratch (SE partial sums) lives after the DMA buffers, so DMA may stay in flight
        stamp(3);
        if (a.dbg & 128) {       // dbg 128: timing experiment without the epilogue (keeps the MFMAs alive)
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n) t += acc[j][n][0] + acc[j][n][1] + acc[j][n][2] + acc[j][n][3];
            if (t == 123.456f) ((half_t*)a.y)[0] = (half_t)t;
        } else
        {
            const int tw = cur.tw, th = cur.th, img = cur.img;
            conv_epilogue<WN, WM, JT, false, SPLIT, true, true, DSFUSE || RESPRE>(a, acc, smem + kHalo4Lds, tid, lane, 0, wm, cur.n0, cur.mt,
                                                          img, th, tw, th * TR + wrow, tw * TC + wcol,
                                                          STAMP ? a.stamps + (size_t)blockIdx.x * 16 : nullptr, rlo, rhi, rpre);
        }
        if (!has_next) break;
        kbase += nk;
        tidx = tnext;
        if (dynamic) {                                  // broadcast the drawn index (word alternates per tile: no WAR)
            if (tid == 0) lds_next[tile_no & 1] = drawn;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            tnext = __builtin_amdgcn_readfirstlane(lds_next[tile_no & 1]);
        } else {
            tnext = tidx + nloc;
        }
        ++tile_no;
        cur = tile_at(tidx);
    }
    if (a.dbg & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (K loop skipped: the prologue DMA is still in flight)
    if (STAMP) {
        stamp(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);
        if (tid == 0) {
            a.stamps[(size_t)blockIdx.x * 16 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
            a.stamps[(size_t)blockIdx.x * 16 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        }
    }
}

// -------------------------------------------------------------------------------------------
// conv3x3_halo4h: the halo4 kernel with the halo held as TWO 32-channel halves (64-byte LDS rows) so that it is
// effectively double-buffered without more LDS. halo4 keeps one 64-channel halo and must stop at every chunk boundary:
// barrier, 12 DMA pieces per wave, wait for them (5.5 % of its K loop, profiles/r03_overhead_experiments.txt). Here a
// chunk's 18 (tap, half) units run as half A's nine taps, then half B's; two units (64 MFMAs per wave) per step:
//   step 0..3: (A0,A1) (A2,A3) (A4,A5) (A6,A7)   step 4: (A8,B0)   step 5..8: (B1,B2) (B3,B4) (B5,B6) (B7,B8)
// Half A is last read in step 4, half B in step 8, so the NEXT chunk's half A streams in during steps 5-7 and a chunk's
// own half B during its steps 0-2 (two pieces per wave and step, issued behind the step's weight pieces, so the next
// step's counted wait leaves them in flight): no extra barrier, no stall, the same number of DMA pieces.
// LDS rows are 64 B (32 channels): a 16x16x32 operand fragment is 16 rows x 64 B = 1 KiB; the 16-byte slot of chunk q in
// row r is q ^ 2*((r >> 2) & 1), conflict-free for ds_read_b128 at every row alignment (checked exhaustively against the
// instruction's 4 x 16 lane groups). Weight step buffer = 2 units x [128 couts][32 cin] = 16 KB, double-buffered.
// f16 only, non-persistent, no fused downsample (those launches stay on halo4).
// -------------------------------------------------------------------------------------------
template <int GEOM>
__global__ __launch_bounds__(256, 2) void conv3x3_halo4h_kernel(const ConvArgs a) {
    constexpr int WN = 1, WM = 4, JT = 8, BN = 128;
    constexpr int TR = GEOM ? 8 : 16, TC = GEOM ? 32 : 16;
    constexpr int S = TC + 4;
    constexpr int kHalf = (TR + 2) * S * 64;                  // bytes of one 32-channel halo half (23040)
    constexpr int kHalfPieces = (kHalf + 1023) / 1024;        // 23 one-KiB pieces, the last one half full
    static_assert(2 * kHalf == kHaloBytes && kHalfPieces == 23, "halo footprint");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv;
    const int cin = a.Cin;
    const int nkc = cin / kBK;
    {
        const int k0 = a.mtiles, k1 = a.ntiles, k2 = a.tilesW, k3 = a.tilesH, k4 = a.Cin, k5 = a.in_sh, k6 = a.nt_shift,
                  k7 = a.th_shift, k8 = a.CoutPad;
        const uint64_t k9 = a.tw_magic;
        const int64_t k10 = a.in_sb;
        const half_t *k11 = a.x, *k12 = a.w;
        const float* k13 = a.bias;
        asm volatile("" ::"s"(k0), "s"(k1), "s"(k2), "s"(k3), "s"(k4), "s"(k5), "s"(k6), "s"(k7), "s"(k8), "s"(k9), "s"(k10),
                     "s"(k11), "s"(k12), "s"(k13));
    }
    const int total = a.mtiles * a.ntiles;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int tq = total >> 3, tr = total & 7;
    const int lin = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + local;
    int nt, mt, t2, th, img;
    if (a.nt_shift >= 0) { nt = lin & (a.ntiles - 1); mt = lin >> a.nt_shift; }
    else { nt = lin % a.ntiles; mt = lin / a.ntiles; }
    const int n0 = nt * BN;
    t2 = (int)(((uint64_t)(uint32_t)mt * a.tw_magic) >> 40);
    const int tw = mt - t2 * a.tilesW;
    if (a.th_shift >= 0) { th = t2 & (a.tilesH - 1); img = t2 >> a.th_shift; }
    else { th = t2 % a.tilesH; img = t2 / a.tilesH; }
    const char* xb = (const char*)(a.x + img * a.in_sb + (int64_t)(th * TR) * a.in_sh + (int64_t)(tw * TC) * cin);
    const char* wb = (const char*)(a.w + (int64_t)n0 * cin);

    const int wrow = GEOM ? (wm >> 1) * 4 : wm * 4;
    const int wcol = GEOM ? (wm & 1) * 16 : 0;
    const int q = lane >> 4, c = lane & 15;
    // ---- per-lane DMA offsets: a piece is 16 LDS rows x 64 B; lane l fills slot l & 3 of row l >> 2 ----
    // Weights: piece i of a wave is 16 cout rows further on (the swizzle key (row >> 2) & 1 does not change), so ONE lane
    // offset serves all four and the piece stride goes into the scalar base. Halo: a piece's pixel -> (hy, hx) needs a
    // division, six of them held in registers spill (the K loop is fully unrolled, 128 accumulators + 56 fragment
    // registers live) and a spill reload drains vmcnt; they are recomputed at the issue point instead (a dozen VALU
    // instructions per piece in the shadow of the MFMAs) from an opaque copy of the lane id so the compiler cannot
    // hoist them back out of the loop.
    uint32_t woff0;
    {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int prow = ln >> 2, slot = ln & 3;
        const int row = ((wv * 4) & 7) * 16 + prow;             // cout row inside the unit's [128][32] tile
        woff0 = (uint32_t)row * (uint32_t)cin * 2u + (uint32_t)((slot ^ (((row >> 2) & 1) << 1)) << 4);
    }
    const int wpiece = 16 * cin * 2;                            // bytes between a wave's weight pieces
    // weights of step s of chunk kc (units 2s, 2s+1) into buffer buf: waves 0,1 stage the first unit, 2,3 the second
    auto stage_w = [&](int kc, int s, int buf) {
        const int u = 2 * s + (wv >> 1);
        const int tap = u < 9 ? u : u - 9, half = u < 9 ? 0 : 1;
        const char* src = wb + ((int64_t)tap * a.CoutPad * cin + (int64_t)kc * kBK + half * 32) * 2;
        char* dst = smem + buf * 16384 + (wv * 4) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_asm_s(src + i * wpiece, woff0, dst + i * 1024);
    };
    // pieces r0, r0+1 (of this wave's six; piece wv + 4r, the 24th slot repeats piece 22) of half `half` of chunk kc
    auto stage_h2 = [&](int kc, int half, int r0) {
        const char* src = xb + ((int64_t)kc * kBK + half * 32) * 2;
        char* dst = smem + 32768 + half * kHalf;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int prow = ln >> 2, slot = ln & 3;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = r0 + rr;
            int hp = wv + 4 * r;
            if (hp > kHalfPieces - 1) hp = kHalfPieces - 1;
            const int hr = hp * 16 + prow;                      // halo pixel index hy * S + hx
            int hy = (int)(((uint32_t)hr * (65536u / S + 1u)) >> 16), hx = hr - hy * S;      // hr < 512: exact
            if (hx > TC + 1) hx = TC + 1;                       // pad columns: any valid address
            if (hy > TR + 1) hy = TR + 1;                       // rows past the halo (second half of the last piece): masked
            const uint32_t off = ((uint32_t)hy * (uint32_t)a.in_sh + (uint32_t)hx * (uint32_t)cin) * 2u +
                                 (uint32_t)((slot ^ (((hr >> 2) & 1) << 1)) << 4);
            if (hp == kHalfPieces - 1) {                        // the last piece covers only 8 rows: upper lanes stay out
                if (lane < 32) glds16_asm_s(src, off, dst + hp * 1024);
            } else {
                glds16_asm_s(src, off, dst + hp * 1024);
            }
        }
    };
    const int aoff = c * 64 + ((q ^ (((c >> 2) & 1) << 1)) << 4);

    // prologue: half A of chunk 0 (all six pieces of every wave) and the weights of step 0
    stage_h2(0, 0, 0);
    stage_h2(0, 0, 2);
    stage_h2(0, 0, 4);
    stage_w(0, 0, 0);
    f32x4 acc[JT][4];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const f32x4 b4 = *(const f32x4*)(a.bias + n0 + q * 8 + acc_cout_offset(j));
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[j][n] = b4;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_setprio(1);
    int gstep = 0;
    bool halo_prev = false;
    for (int kc = 0; kc < nkc; ++kc) {
        const bool next_chunk = kc + 1 < nkc;
#pragma unroll
        for (int s = 0; s < 9; ++s, ++gstep) {               // fully unrolled: taps, halves and piece indices are constants
            if (halo_prev) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // the two halo pieces of the last step may fly on
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // fragment addresses of the step's two units
            const char* wt = smem + (gstep & 1) * 16384 + aoff;
            const char *be[2], *bo[2];
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int u = 2 * s + uu;
                const int tap = u < 9 ? u : u - 9, half = u < 9 ? 0 : 1;
                const int tdy = tap / 3, dx = tap - tdy * 3 - 1;
                const int hr0 = (wrow + tdy) * S + wcol + c + 1 + dx;           // halo pixel of pixel-repeat n = 0
                const char* hb = smem + 32768 + half * kHalf + hr0 * 64;
                const int slot = q ^ (((hr0 >> 2) & 1) << 1);
                be[uu] = hb + (slot << 4);                                      // even n (S / 4 is odd: the key flips with n)
                bo[uu] = hb + ((slot ^ 2) << 4);
            }
            const bool more = !(s == 8 && !next_chunk);
            const bool halo_now = (s < 3) || (next_chunk && s >= 5 && s < 8);
            f16x8 ar[3][2], bq[2][4];
            auto read_a = [&](int g, f16x8 (&dst)[2]) {
                const char* base = wt + (g >> 2) * 8192 + (2 * (g & 3)) * 1024;
                dst[0] = *(const f16x8*)base;
                dst[1] = *(const f16x8*)(base + 1024);
            };
            auto read_b = [&](int uu, f16x8 (&dst)[4]) {
#pragma unroll
                for (int n = 0; n < 4; ++n) dst[n] = *(const f16x8*)(((n & 1) ? bo[uu] : be[uu]) + n * (S * 64));
            };
            __builtin_amdgcn_sched_barrier(0);
            read_b(0, bq[0]);
            read_a(0, ar[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_a(1, ar[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                const int s1 = s == 8 ? 0 : s + 1, kc1 = s == 8 ? kc + 1 : kc;
                stage_w(kc1, s1, (gstep + 1) & 1);
            }
            if (halo_now) {
                if (s < 3) stage_h2(kc, 1, 2 * s);                   // this chunk's half B (first read in step 4)
                else stage_h2(kc + 1, 0, 2 * (s - 5));               // the next chunk's half A (half A was last read in step 4)
            }
            halo_prev = halo_now;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                if (g + 2 < 8) read_a(g + 2, ar[(g + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 1) { read_b(1, bq[1]); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[2 * (g & 3) + jj][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[g % 3][jj], bq[g >> 2][n],
                                                                                         acc[2 * (g & 3) + jj][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __builtin_amdgcn_s_setprio(0);
    conv_epilogue<WN, WM, JT, false, false, true, true, false>(a, acc, smem + kHalo4Lds, tid, lane, 0, wm, n0, mt, img, th, tw,
                                                               th * TR + wrow, tw * TC + wcol);
}

#ifndef LDS_PAD
#define LDS_PAD 0      // A/B build switch: extra LDS bytes per workgroup (e.g. 40000 forces ONE workgroup per CU)
#endif
constexpr int kHalo4LdsTotal = kHalo4Lds + 4 * 128 * 4 + 16 + LDS_PAD;   // + [WM][BN] floats of epilogue scratch + 2 queue words

template <int GEOM, bool SPLIT, bool PERSIST>
static hipError_t launch_conv_halo4_tp(const ConvArgs& a0, hipStream_t s) {
    static bool done[64] = {};
    hipError_t e0 = raise_lds_limit((const void*)conv3x3_halo4_kernel<GEOM, SPLIT, PERSIST>, kHalo4LdsTotal, done);
    if (e0 != hipSuccess) return e0;
    static const int dbg = env_dbg();
    ConvArgs a = a0;
    a.dbg |= dbg;
    int grid = a.mtiles * a.ntiles;
    if (PERSIST) {
        static int cus[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && cus[dev] == 0) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus[dev] = prop.multiProcessorCount;
            if (cus[dev] <= 0) cus[dev] = 256;
        }
        const int resident = 2 * ((dev >= 0 && dev < 64) ? cus[dev] : 256);     // two workgroups per CU
        if (grid > resident) grid = resident;
        grid = (grid + 7) / 8 * 8;                         // whole XCD rounds (extra workgroups exit at once)
    }
    hipLaunchKernelGGL((conv3x3_halo4_kernel<GEOM, SPLIT, PERSIST>), dim3(grid), dim3(256), kHalo4LdsTotal, s, a);
    return hipGetLastError();
}
template <int GEOM, bool SPLIT>
static hipError_t launch_conv_halo4_t(const ConvArgs& a, hipStream_t s) {
    if (a.ds_x != nullptr) {
        if (GEOM != 0) return hipErrorInvalidValue;               // (the engine only fuses on the 16x16 geometry)
        static bool done_ds[64] = {};
        hipError_t e0 = raise_lds_limit((const void*)conv3x3_halo4_kernel<0, SPLIT, false, false, true>, kHalo4LdsTotal, done_ds);
        if (e0 != hipSuccess) return e0;
        static const int dbg = env_dbg();
        ConvArgs b = a;
        b.dbg |= dbg;
        hipLaunchKernelGGL((conv3x3_halo4_kernel<0, SPLIT, false, false, true>), dim3(a.mtiles * a.ntiles), dim3(256),
                           kHalo4LdsTotal, s, b);
        return hipGetLastError();
    }
    if (a.stamps != nullptr && !SPLIT) {
        static bool done[64] = {};
        hipError_t e0 = raise_lds_limit((const void*)conv3x3_halo4_kernel<GEOM, false, false, true>, kHalo4LdsTotal, done);
        if (e0 != hipSuccess) return e0;
        hipLaunchKernelGGL((conv3x3_halo4_kernel<GEOM, false, false, true>), dim3(a.mtiles * a.ntiles), dim3(256),
                           kHalo4LdsTotal, s, a);
        return hipGetLastError();
    }
    // persistent tiles with a STATIC stride measured 6-8 % slower on every layer (r01); HCTR_PERSIST=2 draws tiles
    // from an atomic queue instead (a.tile_counter, zeroed by the engine once per forward)
    static const int persist = [] { const char* e = getenv("HCTR_PERSIST"); return e ? atoi(e) : 0; }();
    if constexpr (!SPLIT) {
        // conv2 of an identity block with the residual fetched in the prologue (RESPRE instance): A/B switch HCTR_RESPRE=1,
        // OFF by default - measured 126.7 vs 125.4 ms per step (three interleaved rounds): the fetch costs the same ~4 us
        // in the prologue as in the epilogue, and the instance has no room for the next-tap fragment prefetch
        static const int respre = [] { const char* e = getenv("HCTR_RESPRE"); return e ? atoi(e) : 0; }();
        if (respre && persist == 0 && a.resid != nullptr && a.se_scale != nullptr) {
            static bool done_rp[64] = {};
            hipError_t e0 = raise_lds_limit((const void*)conv3x3_halo4_kernel<GEOM, false, false, false, false, true>,
                                            kHalo4LdsTotal, done_rp);
            if (e0 != hipSuccess) return e0;
            static const int dbg = env_dbg();
            ConvArgs b = a;
            b.dbg |= dbg;
            hipLaunchKernelGGL((conv3x3_halo4_kernel<GEOM, false, false, false, false, true>), dim3(a.mtiles * a.ntiles),
                               dim3(256), kHalo4LdsTotal, s, b);
            return hipGetLastError();
        }
    }
    if (persist == 0 || (persist == 2 && a.tile_counter == nullptr)) return launch_conv_halo4_tp<GEOM, SPLIT, false>(a, s);
    ConvArgs b = a;
    if (persist != 2) b.tile_counter = nullptr;
    return launch_conv_halo4_tp<GEOM, SPLIT, true>(b, s);
}
template <int GEOM>
static hipError_t launch_conv_halo4(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;                 // division-free tile decomposition for the kernel's prologue
    auto shift_of = [](int v) { int sft = 0; while ((1 << sft) < v) ++sft; return (1 << sft) == v ? sft : -1; };
    a.nt_shift = shift_of(a.ntiles);
    a.th_shift = shift_of(a.tilesH);
    if (a.tilesW < 1 || (uint64_t)a.mtiles * (uint64_t)a.tilesW >= ((uint64_t)1 << 40)) return hipErrorInvalidValue;
    a.tw_magic = (((uint64_t)1 << 40) + (uint64_t)a.tilesW - 1) / (uint64_t)a.tilesW;
    // HCTR_HALFHALO=1: the half-buffered halo variant for the plain f16 launches (A/B)
    static const bool halfhalo = [] { const char* e = getenv("HCTR_HALFHALO"); return e ? atoi(e) != 0 : false; }();
    static const int persist = [] { const char* e = getenv("HCTR_PERSIST"); return e ? atoi(e) : 0; }();
    if (halfhalo && !a.split && a.ds_x == nullptr && a.stamps == nullptr && persist == 0 && a.Cin % kBK == 0) {
        static bool done[64] = {};
        hipError_t e0 = raise_lds_limit((const void*)conv3x3_halo4h_kernel<GEOM>, kHalo4LdsTotal, done);
        if (e0 != hipSuccess) return e0;
        static const int dbg = env_dbg();
        a.dbg |= dbg;
        hipLaunchKernelGGL((conv3x3_halo4h_kernel<GEOM>), dim3(a.mtiles * a.ntiles), dim3(256), kHalo4LdsTotal, s, a);
        return hipGetLastError();
    }
    return a.split ? launch_conv_halo4_t<GEOM, true>(a, s) : launch_conv_halo4_t<GEOM, false>(a, s);
}

static hipError_t launch_conv_halo(const ConvArgs& a, hipStream_t s) {
    static bool done[64] = {};
    hipError_t e0 = raise_lds_limit((const void*)conv3x3_halo_kernel, kHaloLds, done);
    if (e0 != hipSuccess) return e0;
    static const int dbg = env_dbg();
    ConvArgs b = a;
    b.dbg |= dbg;
    hipLaunchKernelGGL(conv3x3_halo_kernel, dim3(a.mtiles * a.ntiles), dim3(512), kHaloLds, s, b);
    return hipGetLastError();
}

size_t conv_lds_bytes(ConvTile tile) {
    switch (tile) {
        case TILE_64x256: return 2 * (64 + 256) * 128;
        case TILE_256x256: return 2 * (256 + 256) * 128;
        default: return 2 * (128 + 128) * 128;
    }
}

template <int WN, int WM, int JT, int TAPS, bool LINEAR, bool PIPE, bool SPLIT>
static hipError_t launch_conv_ts(const ConvArgs& a, size_t lds, hipStream_t s) {
    auto kern = conv_mfma_kernel<WN, WM, JT, TAPS, LINEAR, PIPE, SPLIT>;
    static bool done[64] = {};
    hipError_t e0 = raise_lds_limit((const void*)kern, (int)lds, done);
    if (e0 != hipSuccess) return e0;
    const int grid = a.mtiles * a.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WN * WM * 64), lds, s, a);
    return hipGetLastError();
}
template <int WN, int WM, int JT, int TAPS, bool LINEAR, bool PIPE>
static hipError_t launch_conv_t(const ConvArgs& a, size_t lds, hipStream_t s) {
    // the split (f16x3) epilogue only exists for the fp16-output configurations the engine uses in that mode
    if constexpr (!LINEAR && !PIPE) {
        if (a.split) return launch_conv_ts<WN, WM, JT, TAPS, LINEAR, PIPE, true>(a, lds, s);
    }
    return launch_conv_ts<WN, WM, JT, TAPS, LINEAR, PIPE, false>(a, lds, s);
}

hipError_t launch_conv(const ConvArgs& a, ConvTile tile, int taps, bool linear_f32, hipStream_t s) {
    const size_t lds = conv_lds_bytes(tile);
    // HCTR_HALO (read by the engine too): 0 = generic kernels, 1 = 8-wave halo kernel on 256x256 tiles,
    // 2 (default) = TILE_HALO4 chosen by the engine wherever a 3x3 layer allows it
    static const int halo = [] { const char* e = getenv("HCTR_HALO"); return e ? atoi(e) : 2; }();
    static const bool pipe = [] { const char* e = getenv("HCTR_PIPE"); return e ? atoi(e) != 0 : false; }();
    if (linear_f32) {
        if (tile == TILE_256x256)
            return pipe ? launch_conv_t<2, 4, 8, 1, true, true>(a, lds, s) : launch_conv_t<2, 4, 8, 1, true, false>(a, lds, s);
        return launch_conv_t<2, 2, 4, 1, true, false>(a, lds, s);
    }
    switch (tile) {
        case TILE_HALO4:
            return taps == 9 ? launch_conv_halo4<0>(a, s) : hipErrorInvalidValue;
        case TILE_HALO4_8x32:
            return taps == 9 ? launch_conv_halo4<1>(a, s) : hipErrorInvalidValue;
        case TILE_64x256:
            return taps == 9 ? launch_conv_t<1, 4, 4, 9, false, false>(a, lds, s)
                             : launch_conv_t<1, 4, 4, 1, false, false>(a, lds, s);
        case TILE_256x256:
            if (taps == 9 && halo) return launch_conv_halo(a, s);
            if (taps == 9)
                return pipe ? launch_conv_t<2, 4, 8, 9, false, true>(a, lds, s)
                            : launch_conv_t<2, 4, 8, 9, false, false>(a, lds, s);
            return launch_conv_t<2, 4, 8, 1, false, false>(a, lds, s);
        default:
            return taps == 9 ? launch_conv_t<2, 2, 4, 9, false, false>(a, lds, s)
                             : launch_conv_t<2, 2, 4, 1, false, false>(a, lds, s);
    }
}

// -------------------------------------------------------------------------------------------
// stem: NormalizePAD + conv0_1 (1 -> 64, 3x3, pad 1) + folded bn0_1 + ReLU, fp32 math, fp16 out.
// utils/dataset.py:83-93 (x/255, (x-0.5)/0.5, replicate the last valid column to the batch width)
// and models/handwritten_ctr_model.py:116-118. Memory-bound: 1 B in, 128 B out per pixel.
// One thread = one pixel x 8 output channels (one 16-byte store).
// -------------------------------------------------------------------------------------------
constexpr int kStemRows = 16;       // rows per thread (sliding 3x3 window in registers)

__global__ __launch_bounds__(256) void stem_kernel(const void* __restrict__ img, int img_f32,
                                                   const int32_t* __restrict__ widths,
                                                   const float* __restrict__ w9,
                                                   const float* __restrict__ bias,
                                                   half_t* __restrict__ y, int B, int W, int Wa, int split) {
    // thread = (8-channel group, image column); it walks kStemRows rows keeping the 3x3 window and its
    // 72 weights in registers: 3 new pixels and one 16-byte store per output pixel.
    const int cg = threadIdx.x & 7;
    const int w = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int h0 = blockIdx.y * kStemRows;
    const int b = blockIdx.z;
    if (w >= W) return;
    float wt[8][9], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bs[e] = bias[cg * 8 + e];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[e][t] = w9[(cg * 8 + e) * 9 + t];
    }
    const int wlim = widths ? widths[b] : W;
    auto pixel = [&](int hh, int ww) -> float {
        if (hh < 0 || hh >= 128 || ww < 0 || ww >= W) return 0.f;       // conv zero padding (normalised space)
        const int wsrc = ww < wlim ? ww : wlim - 1;                       // NormalizePAD replicate pad
        const int64_t off = ((int64_t)b * 128 + hh) * W + wsrc;
        if (img_f32) return ((const float*)img)[off];
        float x = (float)((const uint8_t*)img)[off] / 255.0f;
        return (x - 0.5f) / 0.5f;
    };
    float win[3][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int d = 0; d < 3; ++d) win[r + 1][d] = pixel(h0 - 1 + r, w - 1 + d);
    const int cs = split ? 192 : 64;                 // channels per pixel ([hi | lo | hi] planes when split)
    for (int r = 0; r < kStemRows; ++r) {
        const int h = h0 + r;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            win[0][d] = win[1][d];
            win[1][d] = win[2][d];
            win[2][d] = pixel(h + 1, w - 1 + d);
        }
        f16x8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float sv = bs[e];
#pragma unroll
            for (int t = 0; t < 9; ++t) sv = fmaf(wt[e][t], win[t / 3][t % 3], sv);
            sv = fmaxf(sv, 0.f);
            o[e] = (half_t)sv;
            ol[e] = split == 2 ? (half_t)0.f : (half_t)(sv - (float)o[e]);
        }
        half_t* dst = y + (((int64_t)b * 130 + h + 1) * Wa + (w + 1)) * cs + cg * 8;
        *(f16x8*)dst = o;
        if (split) {
            *(f16x8*)(dst + 64) = ol;
            *(f16x8*)(dst + 128) = o;
        }
    }
}

// border rows / columns of a padded NHWC activation (one block per (row, image); 16-byte stores)
__global__ __launch_bounds__(256) void zero_borders_kernel(half_t* __restrict__ p, int H, int W, int Wa, int C) {
    const int row = blockIdx.x, b = blockIdx.y;
    u32x4* r = (u32x4*)(p + ((int64_t)b * (H + 2) + row) * (int64_t)Wa * C);
    const int vpp = C >> 3;                                 // 16-byte vectors per pixel
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (row == 0 || row == H + 1) {
        for (int i = threadIdx.x; i < Wa * vpp; i += 256) r[i] = z;
        return;
    }
    for (int i = threadIdx.x; i < vpp; i += 256) r[i] = z;                               // column 0
    const int first = (W + 1) * vpp, n = (Wa - W - 1) * vpp;                              // columns W+1 .. Wa-1
    for (int i = threadIdx.x; i < n; i += 256) r[first + i] = z;
}

hipError_t launch_zero_borders(half_t* p, int B, int H, int W, int Wa, int C, hipStream_t s) {
    if (B == 0) return hipSuccess;
    if (C % 8 != 0 || Wa < W + 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(zero_borders_kernel, dim3(H + 2, B), dim3(256), 0, s, p, H, W, Wa, C);
    return hipGetLastError();
}

hipError_t launch_stem(const void* img, int img_f32, const int32_t* widths_dev, const float* w9,
                       const float* bias, half_t* y, int B, int W, int Wa, int split, hipStream_t s) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(stem_kernel, dim3((W + 31) / 32, 128 / kStemRows, B), dim3(256), 0, s, img, img_f32, widths_dev,
                       w9, bias, y, B, W, Wa, split);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// stem_conv0_2: the two stem convolutions in one kernel (models/handwritten_ctr_model.py:116-123 with NormalizePAD,
// utils/dataset.py:83-93, in front). conv0_1 (1 -> 64 channels) is HBM-bound when it runs alone: 128 B in, 16 kB out
// per pixel column, which conv0_2 reads straight back. Here a workgroup computes conv0_1 + bn0_1 + ReLU for the
// 18 x 18-pixel halo of its 16 x 16 output tile from a 20 x 20 patch of the image (fp32 FMAs in the order of
// stem_kernel: the halo holds bit for bit what that kernel stores) and writes it into the swizzled LDS halo image of
// the halo kernels; conv0_2 + bn0_2 + ReLU + (2,1) max-pool then run as 9 taps of MFMAs from it. Weights of conv0_2
// (9 taps x 64 couts x 64 cin) stream in 2-tap pieces of 16 KB through two buffers.
// 4 waves, each 64 couts x (4 rows x 16 columns); LDS 2 x 16 KB + 45 KB + 1.6 KB patch: two workgroups per CU, so
// one's conv0_1 arithmetic (VALU) overlaps the other's MFMAs.
// -------------------------------------------------------------------------------------------
constexpr int kS2Halo = 32768;                       // LDS offsets
constexpr int kS2Patch = 32768 + kHaloBytes;
constexpr int kS2Lds = kS2Patch + 20 * 20 * 4;       // 80448

__global__ __launch_bounds__(256, 2) void stem_conv0_2_kernel(const ConvArgs a) {
    constexpr int JT = 4, S = kHaloCols;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = a.mtiles;
    int lin;
    {
        const int id = blockIdx.x, xcd = id & 7, sl = id >> 3;
        const int qd = total >> 3, r = total & 7;
        lin = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + sl;
    }
    const int tw = lin % a.tilesW;
    const int t2 = lin / a.tilesW;
    const int th = t2 % a.tilesH;
    const int img = t2 / a.tilesH;
    const int W = a.W;

    // ---- conv0_2 weights of K step st (taps 2st, 2st+1; the last step has one tap): 16 one-KiB pieces, 4 per wave ----
    auto stage_w = [&](int st, int buf) {
        const char* src = (const char*)a.w + (size_t)st * (2 * 64 * 64 * 2);
        char* dst = smem + buf * 16384 + (wv * 4) * 1024;
        const int npieces = st == 4 ? 8 : 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wv * 4 + i;
            if (piece < npieces) {                                  // wave-uniform
                const int g = piece * 64 + lane;
                const int row = g >> 3, cp = (g & 7) ^ (row & 7);
                glds16_asm(src + (uint32_t)row * 128u + cp * 16, dst + i * 1024);
            }
        }
    };
    stage_w(0, 0);
    stage_w(1, 1);                                   // both buffers are free here: two steps' weights fly during the conv0_1 phase

    // ---- 20 x 20 patch of the normalised image (zero outside the image: conv0_1's padding; columns >= the line's
    //      width replicate its last column: NormalizePAD) ----
    float* patch = (float*)(smem + kS2Patch);
    {
        const int wlim = a.img_widths ? a.img_widths[img] : W;
        for (int e = tid; e < 400; e += 256) {
            const int py = e / 20, px = e - py * 20;
            const int hh = th * 16 - 2 + py, ww = tw * 16 - 2 + px;
            float v = 0.f;
            if (hh >= 0 && hh < 128 && ww >= 0 && ww < W) {
                const int wsrc = ww < wlim ? ww : wlim - 1;
                const int64_t off = ((int64_t)img * 128 + hh) * W + wsrc;
                if (a.img_f32) {
                    v = ((const float*)a.img)[off];
                } else {
                    const float x = (float)((const uint8_t*)a.img)[off] / 255.0f;
                    v = (x - 0.5f) / 0.5f;
                }
            }
            patch[e] = v;
        }
    }
    // conv0_1 weights of this thread's 8-channel group
    const int cg = tid & 7;
    float wt[8][9], bs[8];
    {
        const f32x4* wp = (const f32x4*)(a.stem_w + cg * 72);
        const f32x4* bp = (const f32x4*)(a.stem_b + cg * 8);
#pragma unroll
        for (int v = 0; v < 18; ++v) {
            const f32x4 w4 = wp[v];
#pragma unroll
            for (int i = 0; i < 4; ++i) wt[(v * 4 + i) / 9][(v * 4 + i) % 9] = w4[i];
        }
        const f32x4 b0 = bp[0], b1 = bp[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { bs[i] = b0[i]; bs[4 + i] = b1[i]; }
    }
    __syncthreads();
    // ---- conv0_1 + bn0_1 + ReLU into the halo: unit = (halo pixel, 8 channels) -> one 16-byte LDS store ----
    for (int u = tid >> 3; u < kHaloRows * 18; u += 32) {
        const int hy = u / 18, hx = u - hy * 18;
        const int r = th * 16 - 1 + hy, col = tw * 16 - 1 + hx;       // image position of this conv0_1 output
        f16x8 o;
        if (r >= 0 && r < 128 && col >= 0 && col < W) {
            float win[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) win[t] = patch[(hy + t / 3) * 20 + hx + t % 3];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sv = bs[e];
#pragma unroll
                for (int t = 0; t < 9; ++t) sv = fmaf(wt[e][t], win[t], sv);
                o[e] = (half_t)fmaxf(sv, 0.f);
            }
        } else {                                                       // conv0_2's zero padding / columns >= W
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)0.f;
        }
        const int hr = hy * S + hx;
        *(f16x8*)(smem + kS2Halo + hr * 128 + ((cg ^ (hr & 7)) << 4)) = o;
    }

    // ---- conv0_2: 9 taps x 2 k-slices; a wave owns 64 couts x (rows wv*4.. +3) x 16 columns ----
    const int q = lane >> 4, c = lane & 15;
    const int aoff0 = c * 128 + (((0 + q) ^ (lane & 7)) << 4);
    const int aoff1 = c * 128 + (((4 + q) ^ (lane & 7)) << 4);
    const int wrow = wv * 4;
    f32x4 acc[JT][4];
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (STEM_PRIO) __builtin_amdgcn_s_setprio(1);
    for (int st = 0; st < 5; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's weight pieces of step st have landed
        __syncthreads();                                               // ... everyone's, and (st == 0) the halo is written
        if (st >= 1 && st + 1 < 5) stage_w(st + 1, (st + 1) & 1);      // (step 1's weights were issued up front)
        const int ntap = st == 4 ? 1 : 2;
        for (int tt = 0; tt < ntap; ++tt) {
            const int tap = st * 2 + tt;
            const int tdy = tap / 3, dx = tap - tdy * 3 - 1;
            const int u = c + 1 + dx;
            const int v0 = u * 128 + (((q ^ (u & 7) ^ ((tdy & 1) << 2)) & 7) << 4);
            const char* hb = smem + kS2Halo + (wrow + tdy) * (S * 128);
            const char* be = hb + v0;
            const char* bo = hb + (v0 ^ 64);
            const char* wtile = smem + (st & 1) * 16384 + tt * 8192;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 af[JT], bf[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[n] = *(const f16x8*)((((n & 1) ^ ks) ? bo : be) + n * (S * 128));
#pragma unroll
                for (int j = 0; j < JT; ++j) af[j] = *(const f16x8*)(wtile + j * 2048 + (ks ? aoff1 : aoff0));
                if (!STEM_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < JT; ++j)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], bf[n], acc[j][n], 0, 0, 0);
                if (!STEM_PRIO) __builtin_amdgcn_s_setprio(0);
            }
        }
    }
    if (STEM_PRIO) __builtin_amdgcn_s_setprio(0);
    conv_epilogue<1, 4, JT, false, false>(a, acc, smem, tid, lane, 0, wv, 0, lin, img, th, tw, th * 16 + wrow, tw * 16);
}

hipError_t launch_stem_conv0_2(const ConvArgs& a, hipStream_t s) {
    if (a.mtiles == 0) return hipSuccess;
    static bool done[64] = {};
    hipError_t e0 = raise_lds_limit((const void*)stem_conv0_2_kernel, kS2Lds, done);
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(stem_conv0_2_kernel, dim3(a.mtiles), dim3(256), kS2Lds, s, a);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// Squeeze-excite mean WITHOUT materialising conv2's output first.
// SELayer needs mean_{h,w}(bn2(conv2(t))) (models/handwritten_ctr_model.py:27,51-53). The convolution is
// linear, so with S_tap[ci] = sum over output positions of t[h+dy][w+dx][ci] (zero outside the image)
//     mean[c] = bias[c] + (1/HW) * sum_tap sum_ci W[tap][c][ci] * S_tap[ci],
// and S_tap follows from nine statistics of t per (image, channel): the total T, the sums of row 0,
// row H-1, column 0, column W-1 and the four corner values:
//     S(dy,dx) = T - [dy=+1] R0 - [dy=-1] RL - [dx=+1] C0 - [dx=-1] CL + corner(dy,dx)   (if dy,dx != 0)
// T comes from conv1's epilogue (per-tile sums of the stored fp16 values), the rest from se_border
// below and four direct reads. The scale is then known BEFORE conv2 runs, so conv2's epilogue applies
// relu(acc * scale + residual) itself and the separate read-o/read-r/write pass disappears.
// -------------------------------------------------------------------------------------------
constexpr int kBorderSeg = 8;       // each border line is summed by 8 blocks (partials reduced in se_premean)

__global__ __launch_bounds__(256) void se_border_kernel(const half_t* __restrict__ t, int H, int W, int Wa, int C,
                                                        int split, const float* __restrict__ tsum_part, int tiles,
                                                        float* __restrict__ out) {
    __shared__ float red[256 * 8];
    const int b = blockIdx.x, job = blockIdx.y;        // 0: row 0, 1: row H-1, 2: col 0, 3: col W-1, 4: whole-image total
    const int seg = blockIdx.z;
    if (job == 4) {
        // T = sum over the image of t, from conv1's per-tile sums [b][tile][C]: this block adds the tiles of its
        // segment in order (se_premean adds the 8 segment sums in order: fixed association, bit-reproducible)
        const int per = (tiles + kBorderSeg - 1) / kBorderSeg;
        const int t0 = seg * per, t1 = t0 + per < tiles ? t0 + per : tiles;
        for (int ci = threadIdx.x; ci < C; ci += 256) {
            const float* p = tsum_part + (int64_t)b * tiles * C + ci;
            float s0 = 0.f, s1 = 0.f;
            int i = t0;
            for (; i + 1 < t1; i += 2) { s0 += p[(int64_t)i * C]; s1 += p[(int64_t)(i + 1) * C]; }
            if (i < t1) s0 += p[(int64_t)i * C];
            out[(((int64_t)b * 5 + 4) * kBorderSeg + seg) * C + ci] = s0 + s1;
        }
        return;
    }
    const int cv = C >> 3;                             // 16-byte vectors per pixel
    const int v = threadIdx.x % cv, lanes = 256 / cv, p0 = threadIdx.x / cv;
    const int cs = split ? 3 * C : C;                  // channels per pixel in memory
    const half_t* img = t + (int64_t)b * (H + 2) * Wa * cs;
    const int count = job < 2 ? W : H;
    const int per = (count + kBorderSeg - 1) / kBorderSeg;
    const int pbeg = seg * per, pend = pbeg + per < count ? pbeg + per : count;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (threadIdx.x < lanes * cv) {
        auto addr = [&](int p) {
            const int h = job == 0 ? 0 : (job == 1 ? H - 1 : p);
            const int w = job == 2 ? 0 : (job == 3 ? W - 1 : p);
            return img + ((int64_t)(h + 1) * Wa + (w + 1)) * cs + v * 8;
        };
        // four pixels' loads in flight at a time (the kernel is a chain of L2 round trips otherwise); the additions
        // keep their order, so the sums are bit-identical to the one-at-a-time loop
        int p = pbeg + p0;
        for (; p + 3 * lanes < pend; p += 4 * lanes) {
            f16x8 x[4], xl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const half_t* px = addr(p + u * lanes);
                x[u] = *(const f16x8*)px;
                if (split) xl[u] = *(const f16x8*)(px + C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)x[u][e];
                if (split) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += (float)xl[u][e];
                }
            }
        }
        for (; p < pend; p += lanes) {
            const half_t* px = addr(p);
            const f16x8 x = *(const f16x8*)px;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)x[e];
            if (split) {
                const f16x8 xl = *(const f16x8*)(px + C);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)xl[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < cv) {                            // fixed-order reduction over the position lanes
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = 0; l < lanes; ++l)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += red[(l * cv + threadIdx.x) * 8 + e];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            out[(((int64_t)b * 5 + job) * kBorderSeg + seg) * C + threadIdx.x * 8 + e] = s[e];
    }
}

hipError_t launch_se_border(const half_t* t, int B, int H, int W, int Wa, int C, int split, const float* tsum_part,
                            int tiles, float* out, hipStream_t s) {
    hipLaunchKernelGGL(se_border_kernel, dim3(B, 5, kBorderSeg), dim3(256), 0, s, t, H, W, Wa, C, split, tsum_part, tiles, out);
    return hipGetLastError();
}

// one block per (image, 64 output channels); 4 k-slices x 64 couts per block. The block that finishes an image LAST
// (agent-scope counter per image) also runs the SELayer FCs on the completed means (models/handwritten_ctr_model.py:
// 19-29) and writes the channel scales: one launch instead of two, and no second pass over the per-tile sums.
__global__ __launch_bounds__(256) void se_premean_kernel(const float* __restrict__ border,
                                                         const half_t* __restrict__ t,
                                                         const half_t* __restrict__ w,
                                                         const float* __restrict__ bias, int H, int W, int Wa,
                                                         int C, int CoutPad, int split,
                                                         float* __restrict__ mean, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ scale,
                                                         int32_t* __restrict__ counter) {
    __shared__ float S[9 * 512];
    __shared__ float part[256];
    __shared__ int last_flag;
    const int b = blockIdx.x, cg = blockIdx.y;
    const int cs = split ? 3 * C : C;                  // channels per pixel / weight row length
    const half_t* img = t + (int64_t)b * (H + 2) * Wa * cs;
    for (int ci = threadIdx.x; ci < C; ci += 256) {
        float bsum[5];
#pragma unroll
        for (int jb = 0; jb < 5; ++jb) {               // fixed-order sum of the segment partials
            const float* bd = border + (((int64_t)b * 5 + jb) * kBorderSeg) * C + ci;
            float sv = 0.f;
#pragma unroll
            for (int sg = 0; sg < kBorderSeg; ++sg) sv += bd[(int64_t)sg * C];
            bsum[jb] = sv;
        }
        const float R0 = bsum[0], RL = bsum[1], C0 = bsum[2], CL = bsum[3], T = bsum[4];
        auto px = [&](int hp, int wp) {
            const half_t* q = img + ((int64_t)hp * Wa + wp) * cs + ci;
            return split ? (float)q[0] + (float)q[C] : (float)q[0];
        };
        const float t00 = px(1, 1), t0L = px(1, W), tL0 = px(H, 1), tLL = px(H, W);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            float sv = T;
            if (dy == 1) sv -= R0;
            if (dy == -1) sv -= RL;
            if (dx == 1) sv -= C0;
            if (dx == -1) sv -= CL;
            if (dy == 1 && dx == 1) sv += t00;
            if (dy == 1 && dx == -1) sv += t0L;
            if (dy == -1 && dx == 1) sv += tL0;
            if (dy == -1 && dx == -1) sv += tLL;
            S[tap * C + ci] = sv;
        }
    }
    __syncthreads();
    // cout c = cg*64 + cl is stored row blk*64 + s with perm64(s) = cl (engine.cpp perm64)
    const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int jj = (cl >> 5) * 2 + ((cl >> 2) & 1), qq = (cl >> 3) & 3, ii = cl & 3;
    const int srow = jj * 16 + qq * 4 + ii;
    float acc = 0.f;
    const int kper = 9 * C / 4;                        // this slice's share of the 9*C reduction
    // eight 16-byte weight loads in flight per round (one dependent L2 round trip per 8 values otherwise); the FMAs
    // keep their order, so the mean is bit-identical to the one-vector-at-a-time loop
    int kk0 = slice * kper;
    const int kend = (slice + 1) * kper;
    for (; kk0 + 64 <= kend; kk0 += 64) {
        f16x8 wv8[8], wl8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = kk0 + u * 8;
            const int tap = kk / C, ci = kk - tap * C;
            const half_t* wr = w + ((int64_t)tap * CoutPad + cg * 64 + srow) * cs + ci;
            wv8[u] = *(const f16x8*)wr;
            if (split) wl8[u] = *(const f16x8*)(wr + 2 * C);       // weight rows are [w_hi | w_hi | w_lo]
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = kk0 + u * 8;
            const int tap = kk / C, ci = kk - tap * C;
            if (split) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf((float)wv8[u][e] + (float)wl8[u][e], S[tap * C + ci + e], acc);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf((float)wv8[u][e], S[tap * C + ci + e], acc);
            }
        }
    }
    for (int kk = kk0; kk < kend; kk += 8) {           // tail (C = 128: 288 = 4 x 64 + 32 values per slice)
        const int tap = kk / C, ci = kk - tap * C;
        const half_t* wr = w + ((int64_t)tap * CoutPad + cg * 64 + srow) * cs + ci;
        const f16x8 wv = *(const f16x8*)wr;
        if (split) {
            const f16x8 wl = *(const f16x8*)(wr + 2 * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf((float)wv[e] + (float)wl[e], S[tap * C + ci + e], acc);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf((float)wv[e], S[tap * C + ci + e], acc);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const float tot = (part[threadIdx.x] + part[64 + threadIdx.x]) + (part[128 + threadIdx.x] + part[192 + threadIdx.x]);
        const int c = cg * 64 + threadIdx.x;
        mean[(int64_t)b * C + c] = bias[c] + tot / ((float)H * (float)W);
    }
    // ---- hand-off: the means of this block are published (every storing wave drains, barrier, agent-scope release),
    //      then the image's counter is bumped; the block that draws the last ticket acquires and runs the FCs ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(counter + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == (int)gridDim.y - 1;
        if (last) {
            __hip_atomic_store(counter + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    float* mean_s = S;                                 // (S is free again)
    float* hid = S + 512;
    for (int c = threadIdx.x; c < C; c += 256)
        mean_s[c] = __hip_atomic_load(mean + (int64_t)b * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int R = C / 16;
    {
        const int r = threadIdx.x >> 3, l = threadIdx.x & 7;     // hidden: R <= 32 outputs, 8 lanes each
        float sv = 0.f;
        if (r < R)
            for (int c = l; c < C; c += 8) sv = fmaf(w1[r * C + c], mean_s[c], sv);
        sv += __shfl_xor(sv, 1);
        sv += __shfl_xor(sv, 2);
        sv += __shfl_xor(sv, 4);
        if (r < R && l == 0) hid[r] = fmaxf(sv, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float sv = 0.f;
        for (int r = 0; r < R; ++r) sv = fmaf(w2[c * R + r], hid[r], sv);
        scale[(int64_t)b * C + c] = 1.f / (1.f + expf(-sv));
    }
}

hipError_t launch_se_premean(const float* border, const half_t* t, const half_t* w, const float* bias, int B, int H,
                             int W, int Wa, int C, int CoutPad, int split, float* mean, const float* w1,
                             const float* w2, float* scale, int32_t* counter, hipStream_t s) {
    if (C > 512 || C % 64 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(se_premean_kernel, dim3(B, C / 64), dim3(256), 0, s, border, t, w, bias, H, W, Wa, C, CoutPad,
                       split, mean, w1, w2, scale, counter);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// se_fc: SELayer's FC(c -> c/16) ReLU FC(c/16 -> c) sigmoid on the pooled means
// (models/handwritten_ctr_model.py:19-29). One block per image; partial sums are reduced in a
// fixed order so the result is bit-reproducible.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ part, int tiles,
                                                    const float* __restrict__ w1,
                                                    const float* __restrict__ w2,
                                                    float* __restrict__ scale, int C, float inv_hw) {
    __shared__ float mean[512];
    __shared__ float hid[32];
    const int b = blockIdx.x;
    const int R = C / 16;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float* p = part + (int64_t)b * tiles * C + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int t = 0;
        for (; t + 3 < tiles; t += 4) {
            s0 += p[(int64_t)t * C];
            s1 += p[(int64_t)(t + 1) * C];
            s2 += p[(int64_t)(t + 2) * C];
            s3 += p[(int64_t)(t + 3) * C];
        }
        for (; t < tiles; ++t) s0 += p[(int64_t)t * C];
        mean[c] = ((s0 + s1) + (s2 + s3)) * inv_hw;
    }
    __syncthreads();
    // hidden: R <= 32 outputs, 8 lanes each
    {
        const int r = threadIdx.x >> 3, l = threadIdx.x & 7;
        float s = 0.f;
        if (r < R)
            for (int c = l; c < C; c += 8) s = fmaf(w1[r * C + c], mean[c], s);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (r < R && l == 0) hid[r] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s = fmaf(w2[c * R + r], hid[r], s);
        scale[(int64_t)b * C + c] = 1.f / (1.f + expf(-s));
    }
}

hipError_t launch_se_fc(const float* se_part, int tiles_per_img, const float* w1, const float* w2,
                        float* scale, int B, int C, float inv_hw, hipStream_t s) {
    hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(256), 0, s, se_part, tiles_per_img, w1, w2, scale, C, inv_hw);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// se_apply: out = relu(o * scale[b][c] + residual), in place on o
// (models/handwritten_ctr_model.py:30 and :57-58). Runs over the whole padded buffer: border
// elements are 0 in both operands and stay 0. HBM-bound, 16-byte vectors.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_apply_kernel(half_t* __restrict__ o, const half_t* __restrict__ r,
                                                       const float* __restrict__ scale,
                                                       int64_t img_vecs, int64_t total_vecs, int C) {
    const int cv = C >> 3;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total_vecs; v += (int64_t)gridDim.x * 256) {
        const int b = (int)(v / img_vecs);
        const int c0 = (int)(v % cv) << 3;
        const f16x8 ov = *(const f16x8*)(o + v * 8);
        const f16x8 rv = *(const f16x8*)(r + v * 8);
        const f32x4 s0 = *(const f32x4*)(scale + (int64_t)b * C + c0);
        const f32x4 s1 = *(const f32x4*)(scale + (int64_t)b * C + c0 + 4);
        f16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sc = e < 4 ? s0[e] : s1[e - 4];
            y[e] = (half_t)fmaxf(fmaf((float)ov[e], sc, (float)rv[e]), 0.f);
        }
        *(f16x8*)(o + v * 8) = y;
    }
}

hipError_t launch_se_apply(half_t* o, const half_t* r, const float* scale, int64_t img_elems,
                           int B, int C, hipStream_t s) {
    const int64_t img_vecs = img_elems / 8, total = img_vecs * B;
    int64_t grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(se_apply_kernel, dim3((unsigned)grid), dim3(256), 0, s, o, r, scale, img_vecs, total, C);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// argmax_rows: np.argmax(preds, 2) (utils/ctc_codec.py:75): first maximum wins.
// One wave per row; lanes read 16-byte vectors; tie-break on the lower index.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void argmax_merge_np(float& v, int& i, float ov, int oi) {
    if (np_gt(ov, v) || (np_eq(ov, v) && oi < i)) { v = ov; i = oi; }
}

__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t ld, int64_t M,
                                                          int C, int32_t* __restrict__ idx, int tB, int tW) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* p = x + row * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    const int c4 = C & ~3;
    for (int c = lane * 4; c < c4; c += 256) {
        const f32x4 v = *(const f32x4*)(p + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (np_gt(v[e], bv)) { bv = v[e]; bi = c + e; }
    }
    for (int c = c4 + lane; c < C; c += 64) {
        const float v = p[c];
        if (np_gt(v, bv)) { bv = v; bi = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        argmax_merge_np(bv, bi, ov, oi);
    }
    if (lane == 0) {
        // tB > 0: input rows are r = t*tB + b (WBC order), output is [b][t]
        const int64_t o = tB > 0 ? (row % tB) * tW + row / tB : row;
        idx[o] = (bi == 0x7fffffff) ? 0 : bi;
    }
}

// second stage of the fused head argmax: thread per row over the P = ntiles * WN partials (class ranges rise with p).
// Guarded precision: with the runner-up / |logit| partials it also forms the row's top-1/top-2 margin over ALL classes
// (the runner-up of the row is the second largest of the union of every part's best two) and its largest |logit|.
__global__ __launch_bounds__(256) void argmax_partials_kernel(const float* __restrict__ val, const int32_t* __restrict__ cls,
                                                              int P, int64_t M, int32_t* __restrict__ idx,
                                                              const float* __restrict__ val2, const float* __restrict__ absp,
                                                              float* __restrict__ margin, float* __restrict__ rowabs) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float bv = -INFINITY, sv = -INFINITY, av = 0.f;
    int bi = 0x7fffffff;
    for (int p = 0; p < P; ++p) {
        const float v = val[(int64_t)p * M + m];
        const int i = cls[(int64_t)p * M + m];
        const bool wins = np_gt(v, bv) || (np_eq(v, bv) && i < bi);
        if (val2 != nullptr) {
            const float v2 = val2[(int64_t)p * M + m];
            const float lose = wins ? bv : v, keep2 = wins ? v2 : sv;
            sv = np_gt(lose, keep2) ? lose : keep2;
            av = fmaxf(av, absp[(int64_t)p * M + m]);
        }
        if (wins) { bv = v; bi = i; }
    }
    if (idx != nullptr) idx[m] = (bi == 0x7fffffff) ? 0 : bi;
    if (val2 != nullptr) {
        margin[m] = bv - sv;
        rowabs[m] = av;
    }
}

hipError_t launch_argmax_partials(const float* val, const int32_t* cls, int P, int64_t M, int32_t* idx, hipStream_t s,
                                  const float* val2, const float* absp, float* margin, float* rowabs) {
    if (M <= 0) return hipSuccess;
    if (val2 != nullptr && (absp == nullptr || margin == nullptr || rowabs == nullptr)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(argmax_partials_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, val, cls, P, M, idx, val2,
                       absp, margin, rowabs);
    return hipGetLastError();
}

// Guarded precision on stored logits (hctr_forward_logits): one wave per row, top-1/top-2 margin (exact ties give 0,
// a NaN gives NaN) and largest |logit|. Same figures as the fused partials produce.
__global__ __launch_bounds__(256) void row_guard_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int C,
                                                        float* __restrict__ margin, float* __restrict__ rowabs) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* p = x + row * ld;
    float bv = -INFINITY, sv = -INFINITY, av = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = p[c];
        av = fmaxf(av, fabsf(v));
        if (np_gt(v, bv)) { sv = bv; bv = v; }
        else if (np_gt(v, sv)) sv = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off), osv = __shfl_xor(sv, off);
        const bool wins = np_gt(ov, bv);
        const float lose = wins ? bv : ov, keep2 = wins ? osv : sv;
        sv = np_gt(lose, keep2) ? lose : keep2;
        if (wins) bv = ov;
        av = fmaxf(av, __shfl_xor(av, off));
    }
    if (lane == 0) {
        margin[row] = bv - sv;
        rowabs[row] = av;
    }
}

hipError_t launch_row_guard(const float* logits, int64_t ld, int64_t M, int C, float* margin, float* rowabs, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(row_guard_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, logits, ld, M, C, margin, rowabs);
    return hipGetLastError();
}

// one block per line: the smallest top-2 margin over the line's columns (pad columns included: the reference decodes
// them too, utils/ctc_codec.py:75 over test.py:170-186's padded batch) and the line's largest |logit|
__global__ __launch_bounds__(256) void line_guard_kernel(const float* __restrict__ margin, const float* __restrict__ rowabs,
                                                         int W, float* __restrict__ out) {
    __shared__ float smin[4], smax[4];
    __shared__ int sbad[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float mn = INFINITY, mx = 0.f;
    int bad = 0;
    for (int t = threadIdx.x; t < W; t += 256) {
        const float mg = margin[(int64_t)b * W + t];
        bad |= (mg != mg) ? 1 : 0;
        mn = fminf(mn, mg);
        mx = fmaxf(mx, rowabs[(int64_t)b * W + t]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
        bad |= __shfl_xor(bad, off);
    }
    if (lane == 0) { smin[wv] = mn; smax[wv] = mx; sbad[wv] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        bad = sbad[0] | sbad[1] | sbad[2] | sbad[3];
        out[2 * b] = bad ? NAN : mn;
        out[2 * b + 1] = mx;
    }
}

hipError_t launch_line_guard(const float* margin, const float* rowabs, int B, int W, float* out, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(line_guard_kernel, dim3(B), dim3(256), 0, s, margin, rowabs, W, out);
    return hipGetLastError();
}

hipError_t launch_argmax_rows(const float* logits, int64_t ld, int64_t M, int C, int32_t* idx, int tB, int tW,
                              hipStream_t s) {
    const int64_t grid = (M + 3) / 4;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)grid), dim3(256), 0, s, logits, ld, M, C, idx, tB, tW);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// ctc_collapse: greedy CTC collapse (utils/ctc_codec.py:89-93): keep column t iff
// idx[t] != 0 (blank) && idx[t] != C-1 (unknown) && !(t > 0 && idx[t-1] == idx[t]) - the
// previous column is compared RAW. One wave per line; ballot + popcount compaction keeps order.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int32_t* __restrict__ idx, int W, int C,
                                                          int32_t* __restrict__ labels,
                                                          int32_t* __restrict__ lengths) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t* p = idx + (int64_t)b * W;
    int32_t* o = labels + (int64_t)b * W;
    int base = 0;
    for (int t0 = 0; t0 < W; t0 += 64) {
        const int t = t0 + lane;
        int cur = 0, prev = -1;
        if (t < W) {
            cur = p[t];
            if (t > 0) prev = p[t - 1];
        }
        const bool keep = (t < W) && cur != 0 && cur != C - 1 && cur != prev;
        const unsigned long long m = __ballot(keep);
        if (keep) o[base + __popcll(m & ((1ull << lane) - 1ull))] = cur;
        base += __popcll(m);
    }
    if (lane == 0) lengths[b] = base;
}

hipError_t launch_ctc_collapse(const int32_t* idx, int B, int W, int C, int32_t* labels, int32_t* lengths,
                               hipStream_t s) {
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(64), 0, s, idx, W, C, labels, lengths);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// layout helpers for the API-parity paths (the fused fast path never materialises WBC logits)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_to_wbc_kernel(const float* __restrict__ rows, int64_t ld, int B,
                                                          int W, int C, float* __restrict__ out, int Bfull,
                                                          int b0) {
    const int64_t r = blockIdx.x;                  // sub-batch row index t*B + b
    const int t = (int)(r / B), b = (int)(r % B);
    const float* src = rows + ((int64_t)b * W + t) * ld;
    float* dst = out + ((int64_t)t * Bfull + b0 + b) * C;
    for (int c = threadIdx.x; c < C; c += 256) dst[c] = src[c];
}

__global__ __launch_bounds__(256) void wbc_to_rows_kernel(const float* __restrict__ wbc, int B, int W, int C,
                                                          float* __restrict__ rows, int64_t ld) {
    const int64_t r = blockIdx.x;                  // input row index t*B + b
    const int t = (int)(r / B), b = (int)(r % B);
    const float* src = wbc + r * C;
    float* dst = rows + ((int64_t)b * W + t) * ld;
    for (int c = threadIdx.x; c < C; c += 256) dst[c] = src[c];
}

hipError_t launch_logits_to_wbc(const float* logits, int64_t ld, int B, int W, int C, float* out, int Bfull,
                                int b0, hipStream_t s) {
    if ((int64_t)B * W == 0) return hipSuccess;
    hipLaunchKernelGGL(rows_to_wbc_kernel, dim3((unsigned)((int64_t)B * W)), dim3(256), 0, s, logits, ld, B, W, C, out,
                       Bfull, b0);
    return hipGetLastError();
}

hipError_t launch_wbc_to_rows(const float* wbc, int B, int W, int C, float* rows, int64_t ld, hipStream_t s) {
    if ((int64_t)B * W == 0) return hipSuccess;
    hipLaunchKernelGGL(wbc_to_rows_kernel, dim3((unsigned)((int64_t)B * W)), dim3(256), 0, s, wbc, B, W, C, rows, ld);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// row_topk: scipy.special.log_softmax (utils/ctc_codec.py:65) + the top-`search_depth`
// candidates by descending log-prob (:127,186) + the count of candidates above ln(0.001) (:128,144).
// One 256-thread block per (b, t) row; the row is held in LDS; k selection passes.
// log-softmax is evaluated as the reference does in float32: tmp = x - max; tmp - log(sum(exp(tmp)));
// the sum is accumulated in float64 (order-independent to fp32 precision).
// Ties between equal log-probs: the lower class index comes first (the reference's order for
// ties is numpy's unstable argsort, i.e. unspecified).
// -------------------------------------------------------------------------------------------
constexpr int kMaxRowLds = 12288;   // floats (48 KiB): supports C <= 12288

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ x, int64_t ld, int B, int W,
                                                       int C, int k, double thresh,
                                                       int32_t* __restrict__ topk_idx,
                                                       float* __restrict__ topk_logp,
                                                       float* __restrict__ blank_logp,
                                                       float* __restrict__ stats,
                                                       int32_t* __restrict__ cand_count) {
    __shared__ float row[kMaxRowLds];
    __shared__ float rv[4];
    __shared__ int ri[4];
    __shared__ double rd[4];
    const int64_t r = blockIdx.x;                  // t*B + b
    const int t = (int)(r / B), b = (int)(r % B);
    const float* p = x + ((int64_t)b * W + t) * ld;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    float mx = -INFINITY;
    for (int c = tid; c < C; c += 256) {
        const float v = p[c];
        row[c] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) rv[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(rv[0], rv[1]), fmaxf(rv[2], rv[3]));
    double s = 0.0;
    for (int c = tid; c < C; c += 256) s += (double)expf(row[c] - mx);
    s = block_sum_d(s, rd);
    const float logs = logf((float)s);
    // in-place: row[c] = log-prob (float32, as the reference computes it)
    int cnt = 0;
    for (int c = tid; c < C; c += 256) {
        const float lp = (row[c] - mx) - logs;
        row[c] = lp;
        cnt += ((double)lp > thresh) ? 1 : 0;
    }
    if (cand_count) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
        __syncthreads();
        if (lane == 0) ri[wv] = cnt;
        __syncthreads();
        if (tid == 0) cand_count[r] = ri[0] + ri[1] + ri[2] + ri[3];
    }
    __syncthreads();
    if (tid == 0) {
        blank_logp[r] = row[0];
        if (stats) { stats[2 * r] = mx; stats[2 * r + 1] = logs; }
    }
    float pv = INFINITY;
    int pi = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = tid; c < C; c += 256) {
            const float v = row[c];
            const bool after = (v < pv) || (v == pv && c > pi);
            if (after && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_merge(bv, bi, ov, oi);
        }
        __syncthreads();
        if (lane == 0) { rv[wv] = bv; ri[wv] = bi; }
        __syncthreads();
        bv = rv[0]; bi = ri[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) argmax_merge(bv, bi, rv[q], ri[q]);
        pv = bv; pi = bi;
        if (tid == 0) {
            topk_idx[r * k + j] = (bi == 0x7fffffff) ? 0 : bi;
            topk_logp[r * k + j] = bv;
        }
    }
}

hipError_t launch_row_topk(const float* logits, int64_t ld, int B, int W, int C, int k, double thresh,
                           int32_t* topk_idx, float* topk_logp, float* blank_logp, float* stats,
                           int32_t* cand_count, hipStream_t s) {
    if ((int64_t)B * W == 0) return hipSuccess;
    if (C > kMaxRowLds) return hipErrorInvalidValue;
    hipLaunchKernelGGL(row_topk_kernel, dim3((unsigned)((int64_t)B * W)), dim3(256), 0, s, logits, ld, B, W, C, k,
                       thresh, topk_idx, topk_logp, blank_logp, stats, cand_count);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void row_candidates_kernel(const float* __restrict__ x, int64_t ld, int B, int W,
                                                            int C, double thresh, const float* __restrict__ stats,
                                                            const int64_t* __restrict__ cand_off,
                                                            int32_t* __restrict__ cand_idx,
                                                            float* __restrict__ cand_logp) {
    // ascending class order, like np.where (utils/ctc_codec.py:144). stats[2r], stats[2r+1] are the
    // row max and log-sum from row_topk, so the float32 log-prob is recomputed bit-identically.
    const int64_t r = blockIdx.x;
    const int t = (int)(r / B), b = (int)(r % B);
    const float* p = x + ((int64_t)b * W + t) * ld;
    const float mx = stats[2 * r], logs = stats[2 * r + 1];
    const int lane = threadIdx.x;
    int64_t base = cand_off[r];
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float lp = 0.f;
        bool keep = false;
        if (c < C) {
            lp = (p[c] - mx) - logs;
            keep = (double)lp > thresh;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int64_t o = base + __popcll(m & ((1ull << lane) - 1ull));
            cand_idx[o] = c;
            cand_logp[o] = lp;
        }
        base += __popcll(m);
    }
}

// log_softmax_rows: scipy.special.log_softmax(preds, axis=2) (utils/ctc_codec.py:65) for the whole
// tensor, float32 in / float32 out, same arithmetic as row_topk. Rows are independent (any layout).
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int C,
                                                               float* __restrict__ y) {
    __shared__ float rv[4];
    __shared__ double rd[4];
    const int64_t r = blockIdx.x;
    const float* p = x + r * C;
    float* o = y + r * C;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mx = -INFINITY;
    for (int c = tid; c < C; c += 256) mx = fmaxf(mx, p[c]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) rv[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(rv[0], rv[1]), fmaxf(rv[2], rv[3]));
    double s = 0.0;
    for (int c = tid; c < C; c += 256) s += (double)expf(p[c] - mx);
    s = block_sum_d(s, rd);
    const float logs = logf((float)s);
    for (int c = tid; c < C; c += 256) o[c] = (p[c] - mx) - logs;
}

hipError_t launch_log_softmax_rows(const float* x, int64_t rows, int C, float* y, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, C, y);
    return hipGetLastError();
}

hipError_t launch_row_candidates(const float* logits, int64_t ld, int B, int W, int C, double thresh,
                                 const float* stats, const int64_t* cand_off, int32_t* cand_idx,
                                 float* cand_logp, hipStream_t s) {
    if ((int64_t)B * W == 0) return hipSuccess;
    hipLaunchKernelGGL(row_candidates_kernel, dim3((unsigned)((int64_t)B * W)), dim3(64), 0, s, logits, ld, B, W, C,
                       thresh, stats, cand_off, cand_idx, cand_logp);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// Fused beam front end (see ConvArgs in kernels.h): the two small kernels between / after the two head passes.
// Same arithmetic as row_topk_kernel on a stored row: float32 log-prob = (v - max) - logf((float)sum) with the sum
// of expf(v - max) accumulated in float64; top-k by (log-prob desc, class asc); candidates are (double)lp > thresh.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_thresholds_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                              int P, int64_t M, int k, double cand_thresh,
                                                              int want_candidates, float* __restrict__ row_thr,
                                                              int32_t* __restrict__ emit_cnt) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    // k largest part maxima (multiset), kept sorted descending in registers / scratch (k <= kBeamMaxK)
    float top[kBeamMaxK];
#pragma unroll
    for (int i = 0; i < kBeamMaxK; ++i) top[i] = -INFINITY;
    float gmax = -INFINITY;
    for (int p = 0; p < P; ++p) {
        float v = pmax[(int64_t)p * M + m];
        gmax = fmaxf(gmax, v);
#pragma unroll
        for (int i = 0; i < kBeamMaxK; ++i) {              // insertion: v sinks to its place, the smallest drops out
            if (i < k) {
                const float t = top[i];
                const bool sw = v > t;
                top[i] = sw ? v : t;
                v = sw ? t : v;
            }
        }
    }
    float tau = INFINITY;
#pragma unroll
    for (int i = 0; i < kBeamMaxK; ++i)
        if (i == k - 1) tau = top[i];
    // every top-k logit of the row is >= the k-th largest part maximum; a hair below it, so that a class whose
    // log-prob ROUNDS to the k-th one's (and would win the tie on its lower index) is listed as well
    float vmin = tau - 1e-5f * fmaxf(1.f, fabsf(tau));
    if (want_candidates) {
        // value bound of the candidate lists: log-prob > cand_thresh <=> v > max + log(sum) + thresh; the sum
        // estimated from the per-part sums (relative error ~1e-6) and the bound lowered by 1e-3 so that the exact
        // test in beam_select_kernel sees a superset
        double sum = 0.0;
        for (int p = 0; p < P; ++p)
            sum += (double)psum[(int64_t)p * M + m] * exp((double)pmax[(int64_t)p * M + m] - (double)gmax);
        const float vb = (float)((double)gmax + log(sum) + cand_thresh - 1e-3);
        vmin = fminf(vmin, vb);
    }
    row_thr[2 * m] = vmin;
    row_thr[2 * m + 1] = gmax;
    emit_cnt[m] = 0;
}

hipError_t launch_beam_thresholds(const float* pmax, const float* psum, int P, int64_t M, int k, double cand_thresh,
                                  int want_candidates, float* row_thr, int32_t* emit_cnt, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (k < 1 || k > kBeamMaxK || k > P) return hipErrorInvalidValue;
    hipLaunchKernelGGL(beam_thresholds_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, pmax, psum, P, M, k,
                       cand_thresh, want_candidates, row_thr, emit_cnt);
    return hipGetLastError();
}

// one wave per row m = b*W + t; outputs at r = t*B + b
__global__ __launch_bounds__(256) void beam_select_kernel(const float* __restrict__ row_thr, const int32_t* __restrict__ emit_cnt,
                                                          const int32_t* __restrict__ emit_list, int cap,
                                                          const double* __restrict__ esum, int P,
                                                          const float* __restrict__ blank_logit, int B, int W, int k,
                                                          double cand_thresh, int32_t* __restrict__ topk_idx,
                                                          float* __restrict__ topk_logp, float* __restrict__ blank_logp,
                                                          float* __restrict__ stats, int32_t* __restrict__ cand_count,
                                                          int32_t* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int64_t M = (int64_t)B * W;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int b = (int)(m / W), t = (int)(m % W);
    const int64_t r = (int64_t)t * B + b;
    const int cnt = emit_cnt[m];
    if (cnt > cap && lane == 0) atomicExch(overflow, 1);
    const int n = cnt < cap ? cnt : cap;
    const float gmax = row_thr[2 * m + 1];
    double sum = 0.0;                                       // fixed order over the parts
    for (int p = 0; p < P; ++p) sum += esum[(int64_t)p * M + m];
    const float logs = logf((float)sum);
    const int32_t* e = emit_list + (int64_t)m * cap * 2;
    int c_above = 0;
    for (int i = lane; i < n; i += 64) {
        const float lp = (__builtin_bit_cast(float, e[2 * i + 1]) - gmax) - logs;
        c_above += ((double)lp > cand_thresh) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c_above += __shfl_xor(c_above, off);
    if (lane == 0) {
        blank_logp[r] = (blank_logit[m] - gmax) - logs;
        if (stats) { stats[2 * r] = gmax; stats[2 * r + 1] = logs; }
        if (cand_count) cand_count[r] = c_above;
    }
    // k selection rounds over the listed classes, (log-prob desc, class asc) like row_topk_kernel
    float pv = INFINITY;
    int pi = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < n; i += 64) {
            const int cls = e[2 * i];
            const float lp = (__builtin_bit_cast(float, e[2 * i + 1]) - gmax) - logs;
            const bool after = (lp < pv) || (lp == pv && cls > pi);
            if (after && (lp > bv || (lp == bv && cls < bi))) { bv = lp; bi = cls; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            argmax_merge(bv, bi, ov, oi);
        }
        pv = bv; pi = bi;
        if (lane == 0) {
            topk_idx[r * k + j] = (bi == 0x7fffffff) ? 0 : bi;
            topk_logp[r * k + j] = bv;
        }
    }
}

hipError_t launch_beam_select(const float* row_thr, const int32_t* emit_cnt, const int32_t* emit_list, int cap,
                              const double* esum, int P, const float* blank_logit, int B, int W, int k,
                              double cand_thresh, int32_t* topk_idx, float* topk_logp, float* blank_logp,
                              float* stats, int32_t* cand_count, int32_t* overflow, hipStream_t s) {
    const int64_t M = (int64_t)B * W;
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(beam_select_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, row_thr, emit_cnt, emit_list, cap,
                       esum, P, blank_logit, B, W, k, cand_thresh, topk_idx, topk_logp, blank_logp, stats, cand_count,
                       overflow);
    return hipGetLastError();
}

// one wave per row: the listed classes whose log-prob exceeds the threshold, in ascending class order (np.where,
// utils/ctc_codec.py:144), written to the row's CSR slot
__global__ __launch_bounds__(256) void beam_candidates_kernel(const int32_t* __restrict__ emit_cnt,
                                                              const int32_t* __restrict__ emit_list, int cap,
                                                              const float* __restrict__ stats, int B, int W,
                                                              double cand_thresh, const int64_t* __restrict__ cand_off,
                                                              int32_t* __restrict__ cand_idx, float* __restrict__ cand_logp) {
    const int lane = threadIdx.x & 63;
    const int64_t M = (int64_t)B * W;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int b = (int)(m / W), t = (int)(m % W);
    const int64_t r = (int64_t)t * B + b;
    const int cnt = emit_cnt[m];
    const int n = cnt < cap ? cnt : cap;
    const float gmax = stats[2 * r], logs = stats[2 * r + 1];
    const int32_t* e = emit_list + (int64_t)m * cap * 2;
    const int64_t base = cand_off[r];
    for (int i = lane; i < n; i += 64) {
        const int cls = e[2 * i];
        const float lp = (__builtin_bit_cast(float, e[2 * i + 1]) - gmax) - logs;
        if (!((double)lp > cand_thresh)) continue;
        int rank = 0;                                       // candidates with a smaller class come first
        for (int u = 0; u < n; ++u) {
            const float lu = (__builtin_bit_cast(float, e[2 * u + 1]) - gmax) - logs;
            rank += ((double)lu > cand_thresh && e[2 * u] < cls) ? 1 : 0;
        }
        cand_idx[base + rank] = cls;
        cand_logp[base + rank] = lp;
    }
}

hipError_t launch_beam_candidates(const int32_t* emit_cnt, const int32_t* emit_list, int cap, const float* stats,
                                  int B, int W, double cand_thresh, const int64_t* cand_off, int32_t* cand_idx,
                                  float* cand_logp, hipStream_t s) {
    const int64_t M = (int64_t)B * W;
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(beam_candidates_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, emit_cnt, emit_list, cap, stats,
                       B, W, cand_thresh, cand_off, cand_idx, cand_logp);
    return hipGetLastError();
}

}  // namespace hctr
