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

// 64-lane reductions (wave64; DPP/bpermute via __shfl_xor)
DEV u32 wave_max_u32(u32 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { u32 o = (u32)__shfl_xor((int)v, off, WAVE); v = o > v ? o : v; }
    return v;
}
DEV int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(v, off, WAVE); v = o > v ? o : v; }
    return v;
}
