// 32x32x16-MFMA tile helpers shared by the attention and the folded variable-aggregation kernels (gfx950).
//
// LDS images are [row][D] bf16 with one XOR swizzle that is conflict-free for both ds_read_b128 row reads and
// ds_read_b64_tr_b16 transposed reads (16-byte chunk c of row r is stored at chunk c ^ swz(r)); tiles arrive by
// LDS-DMA with the swizzle applied on the per-lane SOURCE address (the destination is lane-linear).
#pragma once
#include "common.h"

namespace {

template <int D> struct Cfg {
  static constexpr int CPR = D / 8;            // 16-byte chunks per row
  static constexpr int RB = D * 2;             // row bytes
  static constexpr int RPI = 64 / CPR;         // rows per 1-KiB LDS-DMA instruction
  static constexpr int TILE = 64 * RB;         // bytes of a 64-row tile
  static constexpr int NDS = D / 16;           // k-steps over the head dim
  static constexpr int NDB = D / 32;           // 32-wide blocks of the head dim
};

// XOR applied to the 16-byte chunk index of a row (only its low 4 bits, i.e. within one 256-byte bank line)
template <int D>
__device__ __forceinline__ int swz(int row) {
  if (D >= 128) return ((row & 3) << 2) | ((row >> 2) & 3);     // rows are 1 (d=128) or 2 (d=256) bank lines
  else return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);        // d=64: two rows share a bank line
}

// stage a [64 rows][D] tile; rows are tokens tok0..tok0+63 of one head: src(row) = base + row*stride
// (rows >= nvalid are read from row nvalid-1: ragged sequence lengths never touch memory outside the tensor)
// ASM: issue the pieces as inline-asm LDS-DMA (below); chosen per kernel by measurement
template <int D, bool RAGGED, int NW = 4, bool ASM = false>     // NW: waves of the workgroup sharing the staging (4 or 8)
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ base, size_t stride, char* tile, int wave,
                                        int lane, int nvalid) {
  using C = Cfg<D>;
  constexpr int NI = C::TILE / 1024;  // instructions per tile (8, 16 or 32)
  static_assert(NI % NW == 0, "tile pieces must divide over the waves");
#pragma unroll
  for (int t = 0; t < NI / NW; ++t) {
    const int i = wave * (NI / NW) + t;
    const int row = i * C::RPI + lane / C::CPR;
    const int cp = lane % C::CPR;
    const int c = cp ^ swz<D>(row);
    const int rsrc = (!RAGGED || row < nvalid) ? row : nvalid - 1;
    // 32-bit per-lane byte offset on a wave-uniform base (saddr form): half the address registers of a 64-bit pointer
    const uint32_t off = ((uint32_t)rsrc * (uint32_t)stride + (uint32_t)(c * 8)) * 2u;
    // ASM = inline-asm LDS-DMA (common.h: glds16_asm).  With the builtin, hipcc cannot tell the ring slot being filled from
    // the one being read (run-time stage index, lane-dependent read addresses) and puts s_waitcnt vmcnt(0) in front of the next
    // LDS read: in the dQ kernel that is the FIRST read of the loop body, right behind the prefetch it has just issued -- the
    // next tile's fetch and this tile's compute do not overlap within a wave.  With the asm form the kernels' own s_waitcnt
    // vmcnt(0) + s_barrier at the end of every tile order the DMA against its readers.  Which form is faster is decided by
    // what hipcc makes of the rest of the loop (round 3, per kernel, profiles/r03_attn_dma_asm_ab.txt: dQ with dropout -7.5 %,
    // the no-dropout forward -1...-10 % with the asm form; the dropout forward +6 %, the fused dK+dV passes +2...4 % SLOWER).
    if constexpr (ASM) glds16_asm(base, off, (uint32_t)(uintptr_t)LDS_PTR(char, tile) + i * 1024);
    else glds16(reinterpret_cast<const char*>(base) + off, tile + i * 1024);
  }
}

// row read: 8 consecutive head-dim elements [ds*16 + 8h .. +7] of row `row`
template <int D>
__device__ __forceinline__ bf16x8 row_frag(const char* tile, int row, int ds, int h) {
  const int c = ds * 2 + h;
  return *reinterpret_cast<const bf16x8*>(tile + row * Cfg<D>::RB + ((c ^ swz<D>(row)) << 4));
}

// transposed read: lane (column = db*32 + (lane&31), h = lane>>5) gets rows rbase + {4h..4h+3, 8+4h..8+4h+3}
template <int D>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int rbase, int db, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = lane >> 5;
  const int row = rbase + 4 * h + q;
  const int c = db * 4 + 2 * (g & 1) + (p >> 1);
  const char* a0 = tile + row * Cfg<D>::RB + ((c ^ swz<D>(row)) << 4) + 8 * (p & 1);
  const int row1 = row + 8;
  const char* a1 = tile + row1 * Cfg<D>::RB + ((c ^ swz<D>(row1)) << 4) + 8 * (p & 1);
  const bf16x4 lo = lds_tr4(a0), hi = lds_tr4(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int V> struct IC { static constexpr int value = V; };   // compile-time tag for generic lambdas

}  // namespace
