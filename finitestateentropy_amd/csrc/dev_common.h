// dev_common.h -- shared device-side helpers for the gfx950 kernels of libfsehip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#define FSEHIP_INTERNAL                  // (the library itself: the measurement aids are declared too)
#include "../../include/fsehip.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t s16;

#define DEV __device__ __forceinline__
#define WAVE 64

// (size_t)-code, lib/error_private.h:77
#define FERR(name) ((size_t)0 - (size_t)FSEHIP_error_##name)
DEV bool is_err(size_t c) { return c > FERR(maxCode); }

DEV u32 hibit32(u32 v) { return 31u - (u32)__clz((int)v); }   // lib/bitstream.h:139 (v != 0)
DEV u32 ld16(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8); }
DEV u32 ld32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
DEV u64 ld64(const u8* p) { return (u64)ld32(p) | ((u64)ld32(p + 4) << 32); }

// A strided batch of byte blocks: block b = base + b*stride, size sizes[b] (or `uniform` when sizes == nullptr).
// Packed form (FSEHIP_compact_batch): `offsets` (nBlocks + 1 entries) instead -- block b = base + offsets[b], offsets[b+1] - offsets[b] bytes.
struct BlockView {
    const u8* base;
    size_t stride;
    const size_t* sizes;
    size_t uniform;
    const u64* offsets;
};
DEV size_t view_size(const BlockView& v, size_t b) { return v.offsets ? (size_t)(v.offsets[b + 1] - v.offsets[b]) : v.sizes ? v.sizes[b] : v.uniform; }
DEV const u8* view_ptr(const BlockView& v, size_t b) { return v.offsets ? v.base + v.offsets[b] : v.base + b * v.stride; }

// ---- cross-lane scans and reductions on the VALU's DPP path (round 6) ----------------------------------------------------------------------
// __shfl_* compile to ds_bpermute_b32: an LDS-pipe instruction with an LDS round trip (~100 cycles in a lone wave, and a place in the pipe that the
// kernels' real LDS traffic queues behind).  A scan over a group of W lanes (W = 8 .. 64, groups aligned to W) is instead a chain of DPP moves folded
// into the ALU instruction: row_shr:1/2/4/8 inside the rows of 16 lanes, then row_bcast:15 (the last lane of a row into the next row, rows 1 and 3)
// and row_bcast:31 (lane 31 into rows 2 and 3) -- six VALU instructions for 64 lanes instead of six bpermutes; the group's total sits in its last
// lane (one v_readlane for W = 64, one shuffle for smaller groups).  Lanes a step does not reach keep their value: bound_ctrl feeds them the
// operation's identity (0: add, unsigned max), or -- min -- the instruction leaves them alone (old = all ones).  FSEHIP_NO_DPP_SCANS: the shuffles
// again (A/B aid).
#ifndef FSEHIP_NO_DPP_SCANS
#define FSEHIP_DPP_SCANS 1
#else
#define FSEHIP_DPP_SCANS 0
#endif
template <int CTRL, int ROWMASK, bool BC> DEV u32 dpp_mov(u32 old, u32 v) { return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xF, BC); }
struct ScanAdd { static constexpr u32 ident = 0u; static constexpr bool bc = true; DEV static u32 op(u32 a, u32 b) { return a + b; } };
struct ScanMax { static constexpr u32 ident = 0u; static constexpr bool bc = true; DEV static u32 op(u32 a, u32 b) { return a > b ? a : b; } };
struct ScanMin { static constexpr u32 ident = 0xFFFFFFFFu; static constexpr bool bc = false; DEV static u32 op(u32 a, u32 b) { return a < b ? a : b; } };
// inclusive scan over the lanes of my group of W (sub = lane & (W - 1))
template <int W, class Op> DEV u32 group_scan_incl(u32 v, u32 sub)
{
    static_assert(W == 8 || W == 16 || W == 32 || W == 64, "group widths");
#define FSEHIP_SCAN_STEP(N, CTRL) { const u32 t = dpp_mov<CTRL, 0xF, Op::bc>(Op::ident, v); v = (W >= 16 || sub >= N) ? Op::op(v, t) : v; }
    FSEHIP_SCAN_STEP(1u, 0x111) FSEHIP_SCAN_STEP(2u, 0x112) FSEHIP_SCAN_STEP(4u, 0x114)
    if (W >= 16) FSEHIP_SCAN_STEP(8u, 0x118)
#undef FSEHIP_SCAN_STEP
    if (W >= 32) { const u32 t = dpp_mov<0x142, 0xA, Op::bc>(Op::ident, v); v = Op::op(v, t); }      // row_bcast:15 into rows 1 and 3
    if (W >= 64) { const u32 t = dpp_mov<0x143, 0xC, Op::bc>(Op::ident, v); v = Op::op(v, t); }      // row_bcast:31 into rows 2 and 3
    (void)sub;
    return v;
}
// the last lane's inclusive value = the group's total, in every lane of the group
template <int W> DEV u32 group_last(u32 incl, u32 lane)
{
    if (W == 64) return (u32)__builtin_amdgcn_readlane((int)incl, 63);
    return (u32)__shfl((int)incl, (int)(lane | (u32)(W - 1)), WAVE);
}
template <int W, class Op> DEV u32 group_reduce(u32 v, u32 lane) { return group_last<W>(group_scan_incl<W, Op>(v, lane & (u32)(W - 1)), lane); }
// 64-bit sums: two 32-bit halves per step
template <int W> DEV u64 group_scan_incl_add64(u64 v, u32 sub)
{
#define FSEHIP_SCAN_STEP64(N, CTRL, RM) { const u64 t = (u64)dpp_mov<CTRL, RM, true>(0u, (u32)v) | ((u64)dpp_mov<CTRL, RM, true>(0u, (u32)(v >> 32)) << 32); v = (W >= 16 || sub >= N) ? v + t : v; }
    FSEHIP_SCAN_STEP64(1u, 0x111, 0xF) FSEHIP_SCAN_STEP64(2u, 0x112, 0xF) FSEHIP_SCAN_STEP64(4u, 0x114, 0xF)
    if (W >= 16) FSEHIP_SCAN_STEP64(8u, 0x118, 0xF)
    if (W >= 32) FSEHIP_SCAN_STEP64(0u, 0x142, 0xA)
    if (W >= 64) FSEHIP_SCAN_STEP64(0u, 0x143, 0xC)
#undef FSEHIP_SCAN_STEP64
    (void)sub;
    return v;
}
template <int W> DEV u64 group_last64(u64 incl, u32 lane)
{
    if (W == 64) return (u64)(u32)__builtin_amdgcn_readlane((int)(u32)incl, 63) | ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(incl >> 32), 63) << 32);
    return (u64)__shfl((unsigned long long)incl, (int)(lane | (u32)(W - 1)), WAVE);
}

// the value of lane (lane ^ D), D a power of two, without the LDS pipe: quad_perm for 1 and 2, two row shifts by 4 under bank masks, row_ror:8, and gfx950's
// v_permlane16_swap / v_permlane32_swap (rows / wave halves exchanged between two registers) with one select for 16 and 32
template <int D> DEV u32 lane_xor(u32 v, u32 lane)
{
    static_assert(D == 1 || D == 2 || D == 4 || D == 8 || D == 16 || D == 32, "power-of-two lane distances");
#if FSEHIP_DPP_SCANS
    if (D == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);             // quad_perm [1,0,3,2]
    if (D == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);             // quad_perm [2,3,0,1]
    if (D == 4) {
        const u32 up = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xA, true);            // row_shr:4 into banks 1 and 3 (lanes 4-7, 12-15 of a row)
        return (u32)__builtin_amdgcn_update_dpp((int)up, (int)v, 0x104, 0xF, 0x5, true);              // row_shl:4 into banks 0 and 2; the other banks keep `up`
    }
    if (D == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);            // row_ror:8
    if (D == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16u) ? r[0] : r[1]; }
    { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32u) ? r[0] : r[1]; }
#else
    (void)lane;
    return (u32)__shfl_xor((int)v, D, WAVE);
#endif
}

DEV u32 lane_xor_any(u32 v, u32 d, u32 lane)           // d: a constant once the caller's loops are unrolled, so the switch folds
{
    switch (d) {
    case 1: return lane_xor<1>(v, lane);
    case 2: return lane_xor<2>(v, lane);
    case 4: return lane_xor<4>(v, lane);
    case 8: return lane_xor<8>(v, lane);
    case 16: return lane_xor<16>(v, lane);
    default: return lane_xor<32>(v, lane);
    }
}

// 64-lane reductions
DEV u32 wave_max_u32(u32 v)
{
#if FSEHIP_DPP_SCANS
    return group_reduce<64, ScanMax>(v, 0);
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { u32 o = (u32)__shfl_xor((int)v, off, WAVE); v = o > v ? o : v; }
    return v;
#endif
}
DEV int wave_max_i32(int v)
{
#if FSEHIP_DPP_SCANS
    return (int)(group_reduce<64, ScanMax>((u32)v ^ 0x80000000u, 0) ^ 0x80000000u);      // (order-preserving map to unsigned)
#else
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off, WAVE); v = o > v ? o : v; }
    return v;
#endif
}
