// gram_kbits.inl -- FMT 2 of gram_packed.hip: the operand stays in HBM at ONE BIT per genotype and becomes MX-FP4 only
// in registers, on the way into the matrix cores.  Included twice by gram_packed.hip (kernels inside its anonymous
// namespace, launchers inside namespace pcoa); not a translation unit of its own.
//
// Why (DESIGN_HISTORY.md 4.1): the FP4 operand costs 1.28 GB written by the pre-pass and 4.3 GB re-read by
// the contraction per 10^6 variants at N = 2504, and that L2-hit stream is what slows the fp32 pre-pass beside it.  A
// genotype indicator is one bit; as bits the operand is 0.33 GB written and ~1.1 GB re-read.
//
// Operand layout K1[V/128][Npad][4 words] ("k-bits"): the 16 bytes at K1[blk][i] are sample i's indicators for the 128
// variants of block blk, bit b of word j = variant 128 blk + 32 j + b.  A stage of the contraction is one block: 2 panels x
// 256 samples x 16 B = 8 KiB (the FP4 stage is 32 KiB), one global_load_lds_dwordx4 per wave, and the LDS image IS the
// global image.  A lane's operand of one MFMA (32 FP4 values = 4 dwords) comes from ONE word:
//     d0 = (w << 1) & M,  d1 = w & M,  d2 = (w >> 1) & M,  d3 = (w >> 2) & M,   M = 0x22222222
// i.e. bit b of the word lands in nibble b / 4 of dword b % 4 as 0x2 = E2M1 1.0 -- 7 VALU operations per fragment.  Which
// variant sits in which of the MFMA's 64 k-slots does not matter: X^T X takes A and B from the same words through the
// same function, so every product pairs a variant with itself (the remark on k-order in gram_packed.hip's header).
// The wave layout, the accumulators, the ping-pong phases and the epilogue are those of gram_packed_kernel<1, ...>;
// a wave reads its 6 operand rows of a stage with 6 ds_read_b64 (words of both k-steps at once, 3 KiB instead of the
// 12 KiB of ds_read_b128 fragments) and expands them while its partner on the SIMD issues MFMAs.
#ifdef PCOA_KBITS_KERNELS

// ---------------------------------------------------------------------------------------------- pre-passes -> K1
// fp32 / uint8 tile -> k-bits; verifies that every value is exactly 0 or 1 (flag bit 3), like pack_fp4_kernel.
// One thread: 128 variants x 4 samples, in 8 batches of 16 rows (all 16 loads of a batch issued before the arithmetic).
// CS (small calls): a wave takes ONE of the four 32-variant words of its samples instead of all four -- 4x the waves, each a
// quarter as long; a 4,096-variant call is 320 waves of 8 dependent load batches otherwise, on a chip with 5,000 wave slots
// (256 x 4,096-variant calls: 106 M variants/s with the k-bits operand against 155 with the FP4 one, whose pre-pass had
// 32-variant units).
template <typename T, int VEC, bool NT = false, bool CS = false>
__global__ __launch_bounds__(256) void pack_kbits_kernel(const T* __restrict__ x, int64_t ld, int64_t nv, int n, int npad,
                                                         int64_t nblk, uint32_t* __restrict__ p,
                                                         int32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = npad >> 8;  // waves per block of 128 variants
  int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int c_only = CS ? (int)(wid & 3) : -1;  // wave-uniform
  if (CS) wid >>= 2;
  const int64_t blk = wid / gw;
  const int g = (int)(wid - blk * gw) * 64 + lane;
  if (blk >= nblk) return;
  const int i0 = g * 4;
  bool bad = false;
  uint32_t badw = 0;
  uint32_t w[4][4];  // [sample][word]
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) w[s][q] = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (CS && c != c_only) continue;  // (a scalar branch around the unrolled body: the register indices stay static)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (VEC == 4) {
        // ld % 4 == 0: a group of 4 columns is wholly inside the row or wholly padding (see pack_fp4_kernel)
        typedef typename std::conditional<sizeof(T) == 4, f32x4_t, uint32_t>::type Raw;
        Raw raw[16];
        const int64_t col = (i0 < ld) ? i0 : 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int64_t row = blk * 128 + c * 32 + h * 16 + t;
          const T* src = x + (row < nv ? row : nv - 1) * ld + col;
          if constexpr (NT) raw[t] = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(src));
          else raw[t] = *reinterpret_cast<const Raw*>(src);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int64_t row = blk * 128 + c * 32 + h * 16 + t;
          const uint32_t valid = (uint32_t)(row < nv) & (uint32_t)(i0 < ld);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            uint32_t is1, is0;
            if constexpr (sizeof(T) == 4) {
              is1 = (uint32_t)(raw[t][s] == 1.0f);
              is0 = (uint32_t)(raw[t][s] == 0.0f);
            } else {
              const uint32_t b = (raw[t] >> (8 * s)) & 0xffu;
              is1 = (uint32_t)(b == 1u);
              is0 = (uint32_t)(b == 0u);
            }
            const uint32_t live = valid & (uint32_t)(i0 + s < n);  // columns [n, ld) may hold anything
            badw |= live & ((is1 | is0) ^ 1u);
            w[s][c] |= (live & is1) << (h * 16 + t);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int64_t row = blk * 128 + c * 32 + h * 16 + t;
          if (row < nv) {
            const T* src = x + row * ld + i0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              if (i0 + s < n) {  // (n <= ld) padding columns [n, ld) may hold anything: ignored
                const T v = src[s];
                const bool is1 = (v == (T)1);
                bad |= !(is1 || v == (T)0);
                w[s][c] |= (is1 ? 1u : 0u) << (h * 16 + t);
              }
            }
          }
        }
      }
    }
  }
  if constexpr (CS) {
    uint32_t* dst = p + ((size_t)blk * npad + i0) * 4 + c_only;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      dst[4 * s] = c_only == 0 ? w[s][0] : c_only == 1 ? w[s][1] : c_only == 2 ? w[s][2] : w[s][3];
  } else {
    uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)blk * npad + i0) * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s) dst[s] = make_uint4(w[s][0], w[s][1], w[s][2], w[s][3]);
  }
  if (bad || badw) atomicOr(flag, 8);
}

// ---- persistent k-bits pre-pass fed by an LDS-DMA ring (fp32 tiles with ld % 4 == 0 and a 16-byte aligned base) ------------
// Same output as pack_kbits_kernel<float, 4>, built to SHARE a CU with the contraction (the fp32 pipeline of pcoa_capi.hip):
// 256 threads = one wave per SIMD, 40 VGPRs (a contraction held to 224 VGPRs per wave leaves 64 of a SIMD's 512), 4 * R KiB
// of LDS, a fixed grid of one or two workgroups per CU that live for the whole launch.  With the FP4 operand sharing a CU
// was negative-sum (r02: the contraction pulled 53 GB/s per CU of L2 hits through the same in-order vector-memory path);
// the k-bits contraction pulls 13, and beside it this kernel keeps ~90 % of the rate it has alone (profiles/r03s .. r03u).  A wave owns units of (block of 128 variants, 256 samples) = 128 rows x 1 KiB; rows travel
// HBM -> LDS by global_load_lds_dwordx4 into the wave's private ring of R one-row slots; per row the wave reads its 16 B
// back (ds_read_b128), converts 4 values (bit 23 of the fp32 pattern is set for 1.0f and clear for 0.0f; fma(f, f, -f) is
// +0 exactly for f in {0, -0, 1}) and re-issues the slot R rows ahead.
// (Tried, r03w: 12 instead of 16 conversion operations per row -- v_perm_b32 byte gathers into a permuted bit order, packed
// fma, v_or3 -- changed nothing beside the contraction, 2.07 vs 2.06 ms per step, and cost 16 registers: not kept.)
template <int R, int AUX, int C>
__device__ __forceinline__ void ring_rows_kbits(const char* xb, int64_t ldb, int nv, int blk, uint32_t voff, int blkn,
                                                uint32_t voffn, uint8_t* myring, int lane, f32x4_t& a, uint32_t (&w)[4][4],
                                                uint32_t (&bad4)[4]) {
  auto issue = [&](int bq, uint32_t vo, int r, int slot) {
    int row = bq * 128 + r;
    row = row < nv ? row : nv - 1;
    // the row address is wave-uniform: pinned into SGPRs so that the per-lane address is one 64-bit add (left alone the
    // compiler hoists xb + vo into a per-lane base and pays a v_mad_u64_u32 + 2 more VALU operations per row)
    const char* rowp = xb + (int64_t)row * ldb;
    asm volatile("" : "+s"(rowp));
    __builtin_amdgcn_global_load_lds((gptr_t)(rowp + vo), (lptr_t)(myring + slot * 1024), 16, 0, AUX);
  };
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    const int r = C * 32 + t;  // row of the unit
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): row r is in `a`, its slot is free
    __builtin_amdgcn_sched_barrier(0);
    if (r + R < 128) issue(blk, voff, r + R, t % R);
    else issue(blkn, voffn, r + R - 128, t % R);
    wait_vmcnt<R - 1>();  // row r + 1 has landed
    __builtin_amdgcn_sched_barrier(0);
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(myring + ((t + 1) % R) * 1024 + lane * 16);
    __builtin_amdgcn_sched_barrier(0);
    if (blk * 128 + r < nv) {  // wave-uniform: rows beyond the tile re-read its last row and are skipped
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const float f = a[s2];
        bad4[s2] |= __float_as_uint(__builtin_fmaf(f, f, -f));
        w[s2][C] |= ((__float_as_uint(f) >> 23) & 1u) << t;
      }
      // pin the conversion here (see ring_unit of gram_packed.hip)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) asm volatile("" : "+v"(w[s2][C]), "+v"(bad4[s2]));
    }
    a = b;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int R, int AUX, int PRIO = 0>
__global__ __launch_bounds__(256, 8) void pack_kbits_ring_kernel(const float* __restrict__ x, int64_t ld, int nv, int n, int npad,
                                                                 int n_units, uint32_t* __restrict__ p,
                                                                 int32_t* __restrict__ flag) {
  static_assert(R == 8 || R == 16 || R == 32, "the slot of row t must be a compile-time constant of the 32-row unrolled body");
  extern __shared__ __attribute__((aligned(16))) uint8_t ring_dyn_kbits[];  // 4 * R KiB, passed at launch (see pack_fp4_ring_kernel)
  if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);  // a memory-bound wave issues rarely: let it go first when it can
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = npad >> 8;  // 256-sample groups per block
  const int stride = (int)gridDim.x * 4;
  const int stride_b = stride / gw, stride_g = stride - stride_b * gw;
  int u = (int)blockIdx.x * 4 + wave;
  if (u >= n_units) return;  // no workgroup barrier anywhere below
  int blk = u / gw, G = u - blk * gw;
  uint8_t* const myring = ring_dyn_kbits + wave * (R * 1024);
  const char* const xb = reinterpret_cast<const char*>(x);
  const int64_t ldb = ld * 4;
  uint32_t bad = 0;
  auto lane_off = [&](int Gq) -> uint32_t {
    const int col = Gq * 256 + lane * 4;
    return col < ld ? (uint32_t)col * 4u : 0u;
  };
  uint32_t voff = lane_off(G);
#pragma unroll
  for (int t = 0; t < R; ++t) {
    int row = blk * 128 + t;
    row = row < nv ? row : nv - 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(xb + (int64_t)row * ldb + voff), (lptr_t)(myring + t * 1024), 16, 0, AUX);
  }
  wait_vmcnt<R - 1>();
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(myring + lane * 16);  // row 0 of the first unit
  for (;;) {
    int blkn = blk + stride_b, Gn = G + stride_g;
    if (Gn >= gw) { Gn -= gw; blkn += 1; }
    const bool last = u + stride >= n_units;
    if (last) { blkn = blk; Gn = G; }  // a harmless re-read keeps the queue depth (and vmcnt) uniform
    const uint32_t voffn = lane_off(Gn);
    const int col = G * 256 + lane * 4;
    uint32_t w[4][4], bad4[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      bad4[s2] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[s2][q] = 0;
    }
    ring_rows_kbits<R, AUX, 0>(xb, ldb, nv, blk, voff, blkn, voffn, myring, lane, a, w, bad4);
    ring_rows_kbits<R, AUX, 1>(xb, ldb, nv, blk, voff, blkn, voffn, myring, lane, a, w, bad4);
    ring_rows_kbits<R, AUX, 2>(xb, ldb, nv, blk, voff, blkn, voffn, myring, lane, a, w, bad4);
    ring_rows_kbits<R, AUX, 3>(xb, ldb, nv, blk, voff, blkn, voffn, myring, lane, a, w, bad4);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const uint32_t cm = (col + s2 < n) ? 0xffffffffu : 0u;  // samples >= n may hold anything
      bad |= bad4[s2] & cm;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[s2][q] &= cm;
    }
    uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)blk * npad + col) * 4);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) dst[s2] = make_uint4(w[s2][0], w[s2][1], w[s2][2], w[s2][3]);
    if (last) break;
    u += stride;
    blk = blkn;
    G = Gn;
    voff = voffn;
  }
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (bad) atomicOr(flag, 8);
}

// ---- the same persistent ring pre-pass for uint8 tiles (ld % 8 == 0, 8-byte aligned base) ----------------------------------
// One wave per SIMD (<= 112 VGPRs beside the contraction), units of 128 variants x 1,024 samples: a row of a unit is the
// 1 KiB of ONE global_load_lds_dwordx4, a lane owns 16 consecutive samples.  Per row 8 VALU operations (the 0/1 bytes of row
// t of an 8-row group are OR-ed in at bit t: one byte of 8 row-bits per sample, as pack_u8x8_kbits_kernel; bytes > 1 raise
// flag bit 3), per 8 rows 32 more to move the 16 bytes into their samples' words.  A lane whose 16 bytes would cross the end
// of a row (ld % 16 == 8) reads the LAST 16 bytes of the row instead and uses the upper half -- no byte beyond the tile is
// ever read.
// ds_read_b128 as an asm statement: a plain load from the ring makes the compiler's wait-count pass put an
// s_waitcnt vmcnt(0) in front of it -- inside a loop it cannot tell the slot being read from the slots the LDS-DMA in
// flight writes -- which drains the ring at every row (measured: 1.59 instead of 0.5 ms per 10^6 variants).  The waits are
// explicit here anyway (vmcnt(R - 1) before the read, lgkmcnt(0) before the value is used).
__device__ __forceinline__ uint4 lds_read_b128_asm(uint32_t lds_addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
template <int R, int AUX>
__global__ __launch_bounds__(256, 4) void pack_u8_kbits_ring_kernel(const uint8_t* __restrict__ x, int64_t ld, int nv, int n,
                                                                    int npad, int n_units, uint32_t* __restrict__ p,
                                                                    int32_t* __restrict__ flag) {
  static_assert(R == 8, "the slot of row t must be a compile-time constant of the 8-row unrolled body");
  extern __shared__ __attribute__((aligned(16))) uint8_t ring_dyn_u8[];  // 4 * R KiB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = (npad + 1023) >> 10;  // 1,024-sample groups per block
  const int stride = (int)gridDim.x * 4;
  const int stride_b = stride / gw, stride_g = stride - stride_b * gw;
  int u = (int)blockIdx.x * 4 + wave;
  if (u >= n_units) return;  // no workgroup barrier anywhere below
  int blk = u / gw, G = u - blk * gw;
  uint8_t* const myring = ring_dyn_u8 + wave * (R * 1024);
  const char* const xb = reinterpret_cast<const char*>(x);
  const int ldi = (int)ld;
  // byte offset of the lane's 16 bytes inside a row, and whether they were moved back by 8 to stay inside it
  auto lane_off = [&](int Gq, bool& shifted) -> uint32_t {
    const int col = Gq * 1024 + lane * 16;
    shifted = col < ldi && col + 16 > ldi;
    return col + 16 <= ldi ? (uint32_t)col : (col < ldi ? (uint32_t)(ldi - 16) : 0u);
  };
  auto issue = [&](int bq, uint32_t vo, int r, int slot) {
    int row = bq * 128 + r;
    row = row < nv ? row : nv - 1;
    const char* rowp = xb + (int64_t)row * ld;
    asm volatile("" : "+s"(rowp));  // wave-uniform: one 64-bit add per lane and row (see ring_rows_kbits)
    __builtin_amdgcn_global_load_lds((gptr_t)(rowp + vo), (lptr_t)(myring + slot * 1024), 16, 0, AUX);
  };
  bool shifted = false, shiftedn = false;
  uint32_t voff = lane_off(G, shifted);
  uint32_t bad = 0;
#pragma unroll
  for (int t = 0; t < R; ++t) issue(blk, voff, t, t);
  wait_vmcnt<R - 1>();
  const uint32_t lds_lane = (uint32_t)(uintptr_t)(lptr_t)(myring + lane * 16);  // the lane's 16 bytes of slot 0 (LDS offset)
  uint4 a = lds_read_b128_asm(lds_lane);  // row 0 of the first unit
  for (;;) {
    int blkn = blk + stride_b, Gn = G + stride_g;
    if (Gn >= gw) { Gn -= gw; blkn += 1; }
    const bool last = u + stride >= n_units;
    if (last) { blkn = blk; Gn = G; }  // a harmless re-read keeps the queue depth (and vmcnt) uniform
    const uint32_t voffn = lane_off(Gn, shiftedn);
    const int col = G * 1024 + lane * 16;
    uint32_t w[16][4], badd[4] = {0, 0, 0, 0};
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
      for (int q = 0; q < 4; ++q) w[s2][q] = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll 1
      for (int g = 0; g < 4; ++g) {  // a real loop: 8 rows of code per word instead of 32
        uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int r = c * 32 + g * 8 + t;  // row of the unit
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): row r is in `a`, its slot is free
          __builtin_amdgcn_sched_barrier(0);
          if (r + R < 128) issue(blk, voff, r + R, t % R);
          else issue(blkn, voffn, r + R - 128, t % R);
          wait_vmcnt<R - 1>();  // row r + 1 has landed
          __builtin_amdgcn_sched_barrier(0);
          const uint4 b = lds_read_b128_asm(lds_lane + ((t + 1) % R) * 1024);
          __builtin_amdgcn_sched_barrier(0);
          if (blk * 128 + r < nv) {  // wave-uniform: rows beyond the tile re-read its last row and are skipped
            const uint32_t av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              badd[d] |= av[d] & 0xfefefefeu;
              acc[d] |= av[d] << t;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) asm volatile("" : "+v"(acc[d]), "+v"(badd[d]));
          }
          a = b;
          __builtin_amdgcn_sched_barrier(0);
        }
        // the 16 bytes of 8 row-bits go to bits 8 g .. 8 g + 7 of word c of their samples (a shifted lane's samples sit in
        // the upper 8 bytes)
        if (shifted) { acc[0] = acc[2]; acc[1] = acc[3]; acc[2] = 0; acc[3] = 0; }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) w[4 * d + bb][c] |= ((acc[d] >> (8 * bb)) & 0xffu) << (8 * g);
      }
    }
    if (shifted) { badd[0] = badd[2]; badd[1] = badd[3]; badd[2] = 0; badd[3] = 0; }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const bool livecol = col + s2 < n;  // samples >= n may hold anything
      const uint32_t cm = livecol ? 0xffffffffu : 0u;
      bad |= (badd[s2 >> 2] >> (8 * (s2 & 3))) & 0xffu & cm;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[s2][q] &= cm;
    }
    if (col < npad) {
      uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)blk * npad + col) * 4);
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) dst[s2] = make_uint4(w[s2][0], w[s2][1], w[s2][2], w[s2][3]);
    }
    if (last) break;
    u += stride;
    blk = blkn;
    G = Gn;
    voff = voffn;
    shifted = shiftedn;
  }
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (bad) atomicOr(flag, 8);
}

// uint8 tile, 8-byte loads (ld % 8 == 0, 8-byte aligned base): one thread = 128 variants x 8 samples.  Per batch of 8 rows
// the 0/1 bytes of row t are OR-ed in at bit t, which leaves one byte of 8 row-bits per sample: byte q of word c.
__global__ __launch_bounds__(256) void pack_u8x8_kbits_kernel(const uint8_t* __restrict__ x, int64_t ld, int64_t nv, int n,
                                                              int npad, int64_t nblk, uint32_t* __restrict__ p,
                                                              int32_t* __restrict__ flag) {
  const int groups = npad >> 3;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t blk = gid / groups;
  const int g = (int)(gid - blk * groups);
  if (blk >= nblk) return;
  const int i0 = g * 8;
  uint32_t m[2];  // byte masks of the columns that exist (< n); columns in [n, ld) may hold anything
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    uint32_t mm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (i0 + 4 * d + b < n) mm |= 0xffu << (8 * b);
    m[d] = mm;
  }
  const bool in_row = i0 < ld;
  uint32_t o[8][4];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) o[s][c] = 0;
  uint32_t bad = 0;
  // a real loop over the four k-blocks (32 rows in flight at a time): unrolled, all 128 row loads are hoisted to the top
  // and the kernel needs 256 VGPRs + AGPR spills; the word index is then selected statically
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    uint32_t oc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) oc[s] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t a0 = 0, a1 = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int64_t row = blk * 128 + c * 32 + q * 8 + t;
        uint2 u = make_uint2(0u, 0u);
        if (row < nv && in_row) u = *reinterpret_cast<const uint2*>(x + row * ld + i0);
        u.x &= m[0];
        u.y &= m[1];
        bad |= (u.x | u.y) & 0xfefefefeu;
        a0 |= (u.x & 0x01010101u) << t;
        a1 |= (u.y & 0x01010101u) << t;
      }
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        oc[sidx] |= ((a0 >> (8 * sidx)) & 0xffu) << (8 * q);
        oc[4 + sidx] |= ((a1 >> (8 * sidx)) & 0xffu) << (8 * q);
      }
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) o[s][cc] = (c == cc) ? oc[s] : o[s][cc];
  }
  uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)blk * npad + i0) * 4);
#pragma unroll
  for (int s = 0; s < 8; ++s) dst[s] = make_uint4(o[s][0], o[s][1], o[s][2], o[s][3]);
  if (bad) atomicOr(flag, 8);
}

// Carrier bitsets (pcoa_accumulate_bits: row v = variant v, bit i & 31 of word i >> 5 = sample i) -> k-bits: 32 x 32 bit
// transposes.  One wave = one block of 128 variants x 256 samples; lane (t, j) loads the 16 bytes of row t that hold
// samples 128 j .. 128 j + 127 of the wave's range, as expand_bits_fp4_kernel does (32 contiguous bytes per row and wave).
// Each of a lane's four dwords belongs to a 32 x 32 bit matrix held by the 32 lanes of its half (lane = variant, bit =
// sample); five butterfly steps (exchange with lane ^ 16, 8, 4, 2, 1: ds_bpermute, v_alignbit, v_bfi) transpose it in
// place, after which lane i holds sample i's word of the k-block.  4 k-blocks x 4 dwords = 16 transposes per lane, then
// four coalesced 16-byte stores.
template <int VEC>
__global__ __launch_bounds__(256) void transpose_bits_kbits_kernel(const uint32_t* __restrict__ bits, int64_t ld_words,
                                                                   int64_t nv, int n, int npad, int64_t nblk,
                                                                   uint32_t* __restrict__ p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gpk = npad >> 8;  // 256-sample groups per block
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t blk = wid / gpk;
  const int G = (int)(wid - blk * gpk);
  if (blk >= nblk) return;
  const int t = lane & 31, j = lane >> 5;
  const int64_t w0 = 8 * (int64_t)G + 4 * j;  // first of this lane's four dwords of a row
  // per-lane constants of the five butterfly steps: which half of the pair this lane is, the bits it keeps, and the
  // rotation that brings the partner's bits into place
  uint32_t keep[5], rot[5];
#pragma unroll
  for (int st = 0; st < 5; ++st) {
    const int jj = 16 >> st;
    const uint32_t m = jj == 16 ? 0x0000ffffu : jj == 8 ? 0x00ff00ffu : jj == 4 ? 0x0f0f0f0fu : jj == 2 ? 0x33333333u : 0x55555555u;
    const bool upper = (t & jj) == 0;
    keep[st] = upper ? m : ~m;
    rot[st] = upper ? (uint32_t)(32 - jj) : (uint32_t)jj;
  }
  uint32_t out[4][4];  // [dword q][k-block c]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t row = blk * 128 + c * 32 + t;
    uint32_t cw[4] = {0u, 0u, 0u, 0u};
    if (row < nv) {
      const uint32_t* r = bits + row * ld_words;
      if (VEC == 4 && w0 + 3 < ld_words) {
        const uint4 u = *reinterpret_cast<const uint4*>(r + w0);
        cw[0] = u.x; cw[1] = u.y; cw[2] = u.z; cw[3] = u.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (w0 + q < ld_words) cw[q] = r[w0 + q];
      }
    }
    // bits of samples >= N (row padding, the tail of the last word) are ignored
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t first = 32 * (w0 + q);
      if (first >= n) cw[q] = 0u;
      else if (first + 32 > n) cw[q] &= (1u << (n - (int)first)) - 1u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t v = cw[q];
#pragma unroll
      for (int st = 0; st < 5; ++st) {
        const int jj = 16 >> st;
        const uint32_t y = (uint32_t)__shfl_xor((int)v, jj, 64);
        const uint32_t ysh = __builtin_amdgcn_alignbit(y, y, rot[st]);  // rotate right
        v = (v & keep[st]) | (ysh & ~keep[st]);
      }
      out[q][c] = v;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)  // sample 256 G + 128 j + 32 q + t
    *reinterpret_cast<uint4*>(p + ((size_t)blk * npad + (size_t)256 * G + 128 * j + 32 * q + t) * 4) =
        make_uint4(out[q][0], out[q][1], out[q][2], out[q][3]);
}

// CSR carrier lists -> k-bits (a list with a repeated callset raises flag bit 5, see below): one wave per variant row, bit (row % 32) of word
// (row % 128) / 32 in sample c's 16-byte slot of block row / 128, through a 32-bit atomic OR.  Zero-filled beforehand.
__global__ __launch_bounds__(256) void densify_csr_kbits_kernel(const int32_t* __restrict__ idx,
                                                                const int64_t* __restrict__ offs, int64_t nv,
                                                                int64_t offs_base, uint32_t* __restrict__ p, int npad,
                                                                int32_t n, int32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int64_t b = offs[row] - offs_base, e = offs[row + 1] - offs_base;
  const int64_t blk = row >> 7;
  const int r = (int)(row & 127);
  for (int64_t q = b + lane; q < e; q += 64) {
    const int32_t c = idx[q];
    if (c < 0 || c >= n) {
      atomicOr(flag, 1);
      flag[2] = c;  // one of the offending indices, for the error message (every flag buffer has >= 4 words)
      continue;
    }
    // (a carrier list that names a callset twice finds its bit already set: flag bit 5 -- the host then redoes the chunk on
    // the int8 kernel, which counts the repeat with multiplicity as the reference's double loop does, VariantsPca.scala:187)
    const uint32_t bit = 1u << (r & 31);
    if (atomicOr(p + ((size_t)blk * npad + c) * 4 + (r >> 5), bit) & bit) atomicOr(flag, 32);
  }
}

// The same through the LDS (r05): one workgroup per block of 128 variants.  A block's carrier entries are ONE contiguous run
// of idx[] (rows are consecutive), so the workgroup streams it with 16-byte loads -- a lane takes 4 consecutive entries
// and finds their rows by a search over the block's 129 offsets in LDS -- and sets bit (row % 32) of word row / 32 of the
// callset's 16-byte slot with a DS atomic OR, whose return value is the repeat check exactly as in the global form.  The
// LDS image IS the global image of the block: it leaves with coalesced 16-byte stores, every word of the block written,
// so the operand needs no zero fill.  Bound by the 4 bytes per carrier it reads, not by the L2's atomic units
// (the global form: 7.4-8.6 ms per 10^6 variants of configs[1]; this one: see profiles/r06*).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void densify_csr_kbits_lds_kernel(const int32_t* __restrict__ idx,
                                                                        const int64_t* __restrict__ offs, int64_t nv,
                                                                        int64_t offs_base, uint32_t* __restrict__ p,
                                                                        int npad, int32_t n, int32_t* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) uint32_t dens_lds[];  // [npad][4] bit words, then 129 relative offsets
  uint32_t* bits = dens_lds;
  int32_t* rel = reinterpret_cast<int32_t*>(dens_lds + (size_t)npad * 4);
  const int tid = threadIdx.x;
  const int64_t blk = blockIdx.x;
  const int64_t r0 = blk * 128;
  const int nrows = r0 < nv ? (int)(nv - r0 < 128 ? nv - r0 : 128) : 0;  // blocks beyond the rows are written as zeros
  for (int i = tid; i < npad; i += THREADS) reinterpret_cast<uint4*>(bits)[i] = make_uint4(0u, 0u, 0u, 0u);
  const int64_t e0 = nrows > 0 ? offs[r0] : 0;
  if (tid <= 128) {
    const int64_t d = nrows > 0 ? offs[r0 + (tid < nrows ? tid : nrows)] - e0 : 0;
    // (a block of sane lists has <= 128 * N entries -- a list names a callset at most a few times, and the loop below advances
    // an int by THREADS * 4 per trip: the cap keeps `q` far from overflow; offsets beyond it, e.g. garbage behind a device
    // pointer, are reported as an offsets error (flag[2] = -2) instead of being followed)
    rel[tid] = (d < 0 || d > (int64_t)0x3fffffff) ? -1 : (int32_t)d;
  }
  __syncthreads();
  int total = rel[128];
  bool bad = total < 0;
  if (tid <= 127 && !bad) bad = rel[tid] < 0 || rel[tid] > rel[tid + 1];
  if (__syncthreads_or(bad)) {
    if (tid == 0) { atomicOr(flag, 1); flag[2] = -2; }   // -2: the row offsets themselves are bad (csr_validate names them)
    total = 0;
  }
  const int32_t* src = idx + (e0 - offs_base);
  const int mis = (int)((reinterpret_cast<uintptr_t>(src) >> 2) & 3);  // src - mis is 16-byte aligned
  for (int q = tid * 4 - mis; q < total; q += THREADS * 4) {
    int32_t cs[4];
    if (q >= 0 && q + 3 < total) {
      const int4 v = *reinterpret_cast<const int4*>(src + q);
      cs[0] = v.x; cs[1] = v.y; cs[2] = v.z; cs[3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) cs[i] = (q + i >= 0 && q + i < total) ? src[q + i] : 0;
    }
    const int qs = q < 0 ? 0 : q;
    int row = 0;  // the largest r with rel[r] <= qs: the (non-empty) row entry qs belongs to
#pragma unroll
    for (int s = 64; s >= 1; s >>= 1)
      if (rel[row + s] <= qs) row += s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qi = q + i;
      if (qi < 0 || qi >= total) continue;
      while (rel[row + 1] <= qi) ++row;  // rel[128] = total > qi: row stays <= 127
      const int32_t c = cs[i];
      if (c < 0 || c >= n) {
        atomicOr(flag, 1);
        flag[2] = c;
        continue;
      }
      const uint32_t bit = 1u << (row & 31);
      if (atomicOr(&bits[(size_t)c * 4 + (row >> 5)], bit) & bit) atomicOr(flag, 32);
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(p + (size_t)blk * npad * 4);
  for (int i = tid; i < npad; i += THREADS) dst[i] = reinterpret_cast<const uint4*>(bits)[i];
}

// ---------------------------------------------------------------------------------------------- contraction
struct StageBits {
  uint32_t pi[256][4];  // panel I: [sample][word]; word 2 * hi + k2 is what lane half `hi` feeds to k-step k2
  uint32_t pj[TJ][4];   // panel J
};

// 8 DMA instructions of 1 KiB per stage, one per wave (waves 0-3: the quarters of panel I, 4-7: of panel J); a diagonal
// tile brings in its single panel with waves 0-3 only.
template <bool DIAG>
__device__ __forceinline__ void issue_stage_bits(StageBits* st, const int8_t* __restrict__ p, int npad, int64_t blk,
                                                 int col_i, int col_j, int wave, int lane) {
  if (DIAG && wave >= 4) return;  // wave-uniform
  const bool is_i = wave < 4;
  const int q = wave & 3;
  const int c0 = (is_i ? col_i : col_j) + q * 64;
  const int8_t* src = p + ((size_t)blk * npad + c0 + lane) * 16;
  uint32_t* dst = is_i ? &st->pi[q * 64][0] : &st->pj[q * 64][0];
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

// the two words (k-steps 0 and 1) of each of the wave's 4 A rows and 2 B rows: 6 ds_read_b64, conflict-free (a half-wave
// reads 32 consecutive 16-byte slots at the same 8-byte offset, the other half the other 8 bytes)
template <bool DIAG>
__device__ __forceinline__ void read_words(const StageBits* st, int wm, int wn, int lane, uint2 (&raw)[6]) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
    raw[mi] = *reinterpret_cast<const uint2*>(&st->pi[wm * 128 + mi * 32 + l31][2 * hi]);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    if constexpr (DIAG) raw[4 + ni] = *reinterpret_cast<const uint2*>(&st->pi[wn * 64 + ni * 32 + l31][2 * hi]);
    else raw[4 + ni] = *reinterpret_cast<const uint2*>(&st->pj[wn * 64 + ni * 32 + l31][2 * hi]);
  }
}

// ENC 0: both operands as E2M1 1.0 (nibble 0x2): d_q = ((w << 1) >> q) & 0x22222222 -- 7 VALU operations per word.
// ENC 1: conjugate weights.  E2M1 also has 0.5 (nibble 0x1) and 2.0 (nibble 0x4), and 0.5 x 2.0 = 1.0 x 1.0 = 1, both
// exact; so bit class q = b % 4 may sit at different heights in the A and the B nibble as long as the heights add up:
//   class   A (rows: 4 fragments per k-step)        B (columns: 2 fragments)
//     0     w & 0x1111..        0.5                 (w << 2) & 0x4444..   2.0
//     1     w & 0x2222..        1.0                 w & 0x2222..          1.0
//     2     w & 0x4444..        2.0                 (w >> 2) & 0x1111..   0.5
//     3     (w >> 1) & 0x4444.. 2.0                 (w >> 3) & 0x1111..   0.5
// 5 operations for an A word, 7 for a B word: 34 instead of 42 per k-step.  (The sign bit of a nibble is the one height
// that cannot be used, which is why class 3 always pays a shift.)
template <int ENC, bool IS_B>
__device__ __forceinline__ i32x4 expand_word_fp4(uint32_t w) {
  i32x4 r;
  if constexpr (ENC == 0) {
    constexpr uint32_t M = 0x22222222u;  // E2M1 1.0 in every nibble
    r[0] = (int)((w << 1) & M);
    r[1] = (int)(w & M);
    r[2] = (int)((w >> 1) & M);
    r[3] = (int)((w >> 2) & M);
  } else if constexpr (!IS_B) {
    r[0] = (int)(w & 0x11111111u);
    r[1] = (int)(w & 0x22222222u);
    r[2] = (int)(w & 0x44444444u);
    r[3] = (int)((w >> 1) & 0x44444444u);
  } else {
    r[0] = (int)((w << 2) & 0x44444444u);
    r[1] = (int)(w & 0x22222222u);
    r[2] = (int)((w >> 2) & 0x11111111u);
    r[3] = (int)((w >> 3) & 0x11111111u);
  }
  return r;
}

// ENC 2 = ENC 1 with the expansion split over the two phases: the read phase only expands the fragments of k-step 0;
// those of k-step 1 are expanded one per MFMA gap while the wave issues its first six MFMAs (which only need k-step 0).
// The read phase (LDS latency + expansion) is then shorter than the partner's MFMA phase instead of longer.
template <int ENC, int K2>
__device__ __forceinline__ void expand_half(const uint2 (&raw)[6], FragsI8<2> (&f)[2]) {
  constexpr int E = ENC == 0 ? 0 : 1;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) f[K2].a[mi] = expand_word_fp4<E, false>(K2 == 0 ? raw[mi].x : raw[mi].y);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) f[K2].b[ni] = expand_word_fp4<E, true>(K2 == 0 ? raw[4 + ni].x : raw[4 + ni].y);
  // The expansion is pure arithmetic: left alone, the optimiser sinks it below the phase barrier to just in front of the
  // MFMAs that consume it -- into the phase where the wave should do nothing but feed the matrix pipe.  An empty asm
  // that "modifies" every fragment pins the arithmetic here, in the wave's read phase.
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) asm volatile("" : "+v"(f[K2].a[mi]));
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) asm volatile("" : "+v"(f[K2].b[ni]));
}

template <int ENC>
__device__ __forceinline__ void expand_frags(uint2 (&raw)[6], FragsI8<2> (&f)[2]) {
  expand_half<ENC, 0>(raw, f);
  if constexpr (ENC != 2) expand_half<ENC, 1>(raw, f);
}

// fragment IDX (in the order the k-step-1 MFMAs need them: a0, b0, b1, a1, a2, a3) of k-step 1, pinned at this point of
// the instruction stream from both sides (its input is "redefined" here, its output "used" here)
template <int IDX>
__device__ __forceinline__ void expand_late(uint2 (&raw)[6], FragsI8<2> (&f)[2]) {
  constexpr int R = IDX == 0 ? 0 : IDX == 1 ? 4 : IDX == 2 ? 5 : IDX - 2;  // row of `raw`
  asm volatile("" : "+v"(raw[R].y));
  if constexpr (R < 4) {
    f[1].a[R] = expand_word_fp4<1, false>(raw[R].y);
    asm volatile("" : "+v"(f[1].a[R]));
  } else {
    f[1].b[R - 4] = expand_word_fp4<1, true>(raw[R].y);
    asm volatile("" : "+v"(f[1].b[R - 4]));
  }
}

// MFMA number T of a stage (order (k-step, mi, ni) as mfma_range), optionally with wait states in front of it INSIDE the
// asm statement, where nothing can be scheduled between them and the instruction.
template <int T, bool PAD>
__device__ __forceinline__ void mfma_one(const FragsI8<2> (&f)[2], f32x16 (&acc)[4][2]) {
  constexpr int k2 = T / 8, mi = (T % 8) / 2, ni = T % 2;
  if constexpr (PAD)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4"
                 : "+v"(acc[mi][ni])
                 : "v"(f[k2].a[mi]), "v"(f[k2].b[ni]));
  else
    asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc[mi][ni]) : "v"(f[k2].a[mi]), "v"(f[k2].b[ni]));
}

// MFMAs LO .. HI-1 of a stage; ENC 2: with the six late expansions behind MFMAs 0 .. 5.
// Two hazards of mixing VALU work into an asm MFMA run, both measured (profiles/r03g_kbits_hazards.txt), neither padded by
// the compiler because it cannot see an MFMA inside an asm statement:
//  * an MFMA that issues in the cycle after a VALU instruction comes back with the first two registers of its
//    accumulator wrong (rows 0, 1, 4, 5 of its 32 x 32 tile): MFMA 6 directly behind the last late expansion (~7,000 of
//    6.3 M entries of S at configs[1] size, group-1 waves only), and, once that one was padded, MFMA 0 / 8 behind
//    compiler-placed moves at small shapes.  One wait state is enough (measured); every MFMA of this schedule carries two,
//    inside its own asm statement where nothing can be scheduled between them and the instruction (free behind another
//    MFMA: the pipe is busy for 32 cycles anyway);
//  * an MFMA reads its A / B registers for several cycles after it has issued: a late expansion must never be allocated
//    to the registers of a k-step-0 fragment that died an instruction ago, so all of k-step 0 is kept alive until its last
//    MFMA has issued.
template <int ENC, int LO, int HI, int T = LO>
__device__ __forceinline__ void mfma_run(uint2 (&raw)[6], FragsI8<2> (&f)[2], f32x16 (&acc)[4][2]) {
  if constexpr (ENC != 2) {
    mfma_range<1, 2, 4, LO, HI>(f, acc);
  } else if constexpr (T < HI) {
    mfma_one<T, true>(f, acc);
    if constexpr (T < 6) expand_late<T>(raw, f);
    if constexpr (T == 7)
      asm volatile("" ::"v"(f[0].a[0]), "v"(f[0].a[1]), "v"(f[0].a[2]), "v"(f[0].a[3]), "v"(f[0].b[0]), "v"(f[0].b[1]));
    mfma_run<ENC, LO, HI, T + 1>(raw, f, acc);
  }
}

// One stage of the ping-pong schedule (pp_stage of gram_packed.hip with the operand expanded in registers).  The phases,
// the barriers and the vmcnt book-keeping are the same; PER_WAVE = 1 DMA instruction per wave and stage.
template <int NST, int BUF, int GRP, bool IDLE, int LEFT, bool DIAG, int ENC>
__device__ __forceinline__ void ppb_stage(StageBits* lds, const int8_t* __restrict__ p, int npad, int64_t blk_begin, int s,
                                          int ns, int col_i, int col_j, int wave, int lane, int wm, int wn,
                                          f32x16 (&acc)[4][2], FragsI8<2> (&f)[2], uint2 (&raw)[6]) {
  constexpr int PER_WAVE = 1;
  constexpr int D = NST - 1;
  constexpr int TOT = 16;  // MFMAs per wave and stage
  const bool more = s + D < ns;
  if constexpr (GRP == 0) {
    // ---- phase 2s: read + expand stage s, issue the DMA of stage s+D
    if constexpr (!IDLE) read_words<DIAG>(&lds[BUF], wm, wn, lane, raw);
    __builtin_amdgcn_sched_barrier(0);
    if (more) issue_stage_bits<DIAG>(&lds[(BUF + D) % NST], p, npad, blk_begin + s + D, col_i, col_j, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!IDLE) expand_frags<ENC>(raw, f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
    // ---- phase 2s+1: the MFMAs of stage s
    if constexpr (!IDLE) {
      asm volatile("s_nop 1");  // VALU-written operands -> MFMA (the barrier covers it; this makes it unconditional)
      __builtin_amdgcn_s_setprio(1);
      mfma_run<ENC, 0, TOT - LEFT>(raw, f, acc);
      __builtin_amdgcn_s_setprio(0);
    }
  } else {
    // ---- phase 2s: issue the DMA of stage s+D, then the MFMAs of stage s-1
    if (more) issue_stage_bits<DIAG>(&lds[(BUF + D) % NST], p, npad, blk_begin + s + D, col_i, col_j, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!IDLE) {
      if (s > 0) {
        asm volatile("s_nop 1");
        __builtin_amdgcn_s_setprio(1);
        mfma_run<ENC, 0, TOT - LEFT>(raw, f, acc);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    raw_barrier();
    // ---- phase 2s+1: (the leftover MFMAs of stage s-1, then) read + expand stage s
    if constexpr (!IDLE) {
      if constexpr (LEFT > 0) {
        if (s > 0) {
          __builtin_amdgcn_s_setprio(2);
          mfma_run<ENC, TOT - LEFT, TOT>(raw, f, acc);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      read_words<DIAG>(&lds[BUF], wm, wn, lane, raw);
      expand_frags<ENC>(raw, f);
    }
  }
  // end of phase 2s+1: stage s+1 must have landed (own share), only the DMA of stage s+2.. may stay in flight
  if (s + 1 < ns) {
    const int rem = ns - 2 - s;
    const int keep = rem < D - 1 ? rem : D - 1;
    if (keep >= 5) wait_vmcnt<(D >= 6 ? 5 * PER_WAVE : 0)>();
    else if (keep == 4) wait_vmcnt<(D >= 5 ? 4 * PER_WAVE : 0)>();
    else if (keep == 3) wait_vmcnt<(D >= 4 ? 3 * PER_WAVE : 0)>();
    else if (keep == 2) wait_vmcnt<(D >= 3 ? 2 * PER_WAVE : 0)>();
    else if (keep == 1) wait_vmcnt<(D >= 2 ? PER_WAVE : 0)>();
    else wait_vmcnt<0>();
  }
  if constexpr (GRP == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
  if constexpr (GRP == 0 && !IDLE && LEFT > 0) {  // group 0's leftover MFMAs of stage s, into phase 2(s+1)
    __builtin_amdgcn_s_setprio(2);
    mfma_run<ENC, TOT - LEFT, TOT>(raw, f, acc);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NST, int GRP, bool IDLE, int LEFT, bool DIAG, int ENC, int... Is>
__device__ __forceinline__ void ppb_round(StageBits* lds, const int8_t* __restrict__ p, int npad, int64_t blk_begin, int s,
                                          int ns, int count, int col_i, int col_j, int wave, int lane, int wm, int wn,
                                          f32x16 (&acc)[4][2], FragsI8<2> (&f)[2], uint2 (&raw)[6],
                                          std::integer_sequence<int, Is...>) {
  ((Is < count ? ppb_stage<NST, Is, GRP, IDLE, LEFT, DIAG, ENC>(lds, p, npad, blk_begin, s + Is, ns, col_i, col_j, wave, lane, wm,
                                                           wn, acc, f, raw)
               : (void)0),
   ...);
}

template <int NST, int GRP, bool IDLE, int LEFT, bool DIAG, int ENC>
__device__ __forceinline__ void ppb_loop(StageBits* lds, const int8_t* __restrict__ p, int npad, int64_t blk_begin, int ns,
                                         int col_i, int col_j, int wave, int lane, int wm, int wn,
                                         f32x16 (&acc)[4][2]) {
  FragsI8<2> f[2];
  uint2 raw[6];
  // prologue: stages 0 .. NST-2 go in flight; stage 0 must have landed before group 0 reads it in phase 0
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < ns) issue_stage_bits<DIAG>(&lds[i], p, npad, blk_begin + i, col_i, col_j, wave, lane);
  if (ns > 1 && NST > 2) wait_vmcnt<(NST > 2 ? 1 : 0)>();
  else wait_vmcnt<0>();
  raw_barrier();
  int s = 0;
  for (; s + NST - 1 < ns; s += NST)
    ppb_round<NST, GRP, IDLE, LEFT, DIAG, ENC>(lds, p, npad, blk_begin, s, ns, NST, col_i, col_j, wave, lane, wm, wn, acc, f,
                                          raw, std::make_integer_sequence<int, NST>{});
  if (s < ns)
    ppb_round<NST, GRP, IDLE, LEFT, DIAG, ENC>(lds, p, npad, blk_begin, s, ns, ns - s, col_i, col_j, wave, lane, wm, wn, acc, f,
                                          raw, std::make_integer_sequence<int, NST - 1>{});
  if constexpr (GRP == 1 && !IDLE) {  // phase 2*ns: group 1's MFMAs of the last stage, nobody to wait for
    asm volatile("s_nop 1");
    mfma_run<ENC, 0, 16>(raw, f, acc);
  }
}

// Work decomposition: xcd_map 0 / 1 / 2 as gram_packed_kernel (legacy split-K, split-K with one k-slice per XCD,
// lock-step); xcd_map 4 = even split: `nwork` = ntri * nstages (tile, stage) units in tile-major order are cut into
// gridDim.x equal runs, a workgroup walks its run and pays one epilogue per tile it touches (at most
// ceil(run / nstages) + 1).  Every CU gets the same number of MFMAs whatever ntri is (55 tiles x split-K 4 leaves 36 of
// 256 CUs idle in the lock-step launch).
// The body is a device function wrapped by the kernel below it, which is held to 224 VGPRs per wave (the compiler takes all
// 256 a 2-waves-per-SIMD kernel may have when left alone, and needs 196): two contraction waves then leave a SIMD 64 of its
// 512 registers -- room for the waves of the persistent ring pre-pass (pack_kbits_ring_kernel), which shares the CU with
// this kernel in the fp32 pipeline (DESIGN_HISTORY.md 4.1, profiles/r03s .. r03u_coreside.txt).
#define PCOA_KBITS_CONTRACTION __device__ __forceinline__ void gram_kbits_body
template <int NST, int LEFT, int ENC>
PCOA_KBITS_CONTRACTION(const int8_t* __restrict__ p, int npad, int64_t nstages, int n,
                                                            int ntile, int ntri, int splitk, int64_t stages_per,
                                                            int32_t* __restrict__ s32, int xcd_map,
                                                            const int32_t* __restrict__ skip, GramStrip strip) {
  __shared__ __attribute__((aligned(16))) StageBits lds[NST];
  if (skip != nullptr && *skip != 0) return;  // auto mode's device-side predicate (gram_packed_kernel)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int b = blockIdx.x;

  // the run of (tile, stage) units of this workgroup: [u, u_end) in tile-major order
  int64_t u, u_end;
  if (xcd_map == 4) {
    const int64_t nwork = (int64_t)ntri * nstages;
    const int64_t nwg = gridDim.x;
    // consecutive runs go to consecutive workgroups of ONE XCD (block b runs on XCD b % 8), so that the workgroups
    // which share a tile's operand panels at about the same k also share an L2
    const int64_t slot = (nwg % kNumXcd == 0) ? (int64_t)(b & 7) * (nwg / kNumXcd) + (b >> 3) : (int64_t)b;
    u = nwork * slot / nwg;
    u_end = nwork * (slot + 1) / nwg;
  } else {
    int tile, ks;
    if (xcd_map == 2) {
      const int g = kNumXcd / splitk;
      const int per = (ntri + g - 1) / g;
      const int xcd = b & 7, slot = b >> 3;
      tile = (xcd % g) * per + slot;
      ks = xcd / g;
      if (slot >= per || tile >= ntri) return;
    } else if (xcd_map) {
      const int q = b >> 3;
      ks = (b & 7) + kNumXcd * (q / ntri);
      tile = q % ntri;
    } else {
      tile = b % ntri;
      ks = b / ntri;
    }
    const int64_t st_begin = (int64_t)ks * stages_per;
    const int64_t st_end = (st_begin + stages_per < nstages) ? (st_begin + stages_per) : nstages;
    if (st_begin >= st_end) return;
    u = (int64_t)tile * nstages + st_begin;
    u_end = (int64_t)tile * nstages + st_end;
  }

  while (u < u_end) {  // workgroup-uniform
    const int tile = (int)(u / nstages);
    const int64_t st_begin = u - (int64_t)tile * nstages;
    const int64_t left = u_end - u;
    const int ns = (int)((nstages - st_begin < left) ? (nstages - st_begin) : left);
    u += ns;

    int row_blk, col_blk;
    if (strip.cols > 0) {  // strip owner: all (row block, column block of the strip) tiles, banded (gram_packed_kernel)
      const int ctiles = ntri / ntile;
      const int per_band = BAND * ctiles;
      const int band = tile / per_band;
      const int r0 = band * BAND;
      const int h = (ntile - r0 < BAND) ? (ntile - r0) : BAND;
      const int rem = tile - band * per_band;
      row_blk = r0 + rem % h;
      col_blk = strip.cb0 + rem / h;
    } else {
      tile_coords<2>(tile, ntile, row_blk, col_blk);
    }
    const int col_i = row_blk * 256, col_j = col_blk * TJ;
    const bool idle = strip.cols == 0 && (col_i + wm * 128) > (col_j + wn * 64 + 63);

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;

    if (row_blk == col_blk && strip.cols == 0) {  // diagonal tile: one panel (workgroup-uniform branch)
      if (wm == 0) ppb_loop<NST, 0, false, LEFT, true, ENC>(lds, p, npad, st_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
      else if (idle) ppb_loop<NST, 1, true, LEFT, true, ENC>(lds, p, npad, st_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
      else ppb_loop<NST, 1, false, LEFT, true, ENC>(lds, p, npad, st_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
    } else if (wm == 0) {
      ppb_loop<NST, 0, false, LEFT, false, ENC>(lds, p, npad, st_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
    } else {
      ppb_loop<NST, 1, false, LEFT, false, ENC>(lds, p, npad, st_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMA -> VALU read of D
    if (!idle) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int j = col_j + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = col_i + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int v = (int)acc[mi][ni][r];  // exact integers below 2^24
            if (strip.cols > 0) {
              if (i < n && j >= strip.col0 && j < strip.col0 + strip.cols && v != 0)
                atomicAdd(&s32[(int64_t)i * strip.cols + (j - strip.col0)], v);
            } else if (j >= i && j < n && v != 0) {
              atomicAdd(&s32[(int64_t)i * n + j], v);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the next tile of this run reuses the LDS ring: every wave must be out of this tile's last stage first (the loops
    // end with a barrier after the last reads; the epilogue touches no LDS)
  }
}
#undef PCOA_KBITS_CONTRACTION
// amdgpu_num_vgpr counts halves of the unified register file on gfx90a+: 112 -> at most 224 registers per wave
template <int NST, int LEFT, int ENC>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(112))) void gram_kbits_kernel(
    const int8_t* __restrict__ p, int npad, int64_t nstages, int n, int ntile, int ntri, int splitk, int64_t stages_per,
    int32_t* __restrict__ s32, int xcd_map, const int32_t* __restrict__ skip, GramStrip strip) {
  gram_kbits_body<NST, LEFT, ENC>(p, npad, nstages, n, ntile, ntri, splitk, stages_per, s32, xcd_map, skip, strip);
}
#ifdef PCOA_EXPERIMENTS
template <int NST, int LEFT, int ENC>
__global__ __launch_bounds__(512, 2) void gram_kbits_uncapped_kernel(
    const int8_t* __restrict__ p, int npad, int64_t nstages, int n, int ntile, int ntri, int splitk, int64_t stages_per,
    int32_t* __restrict__ s32, int xcd_map, const int32_t* __restrict__ skip, GramStrip strip) {
  gram_kbits_body<NST, LEFT, ENC>(p, npad, nstages, n, ntile, ntri, splitk, stages_per, s32, xcd_map, skip, strip);
}
#endif

#endif  // PCOA_KBITS_KERNELS

#ifdef PCOA_KBITS_LAUNCHERS

#ifdef PCOA_EXPERIMENTS
int g_kbits_variant = 0;  // harness knob: which instantiation launch_gram_kbits uses
#endif

// k-bits pre-passes.  nblk_out: blocks of 128 variants to write (the tail beyond nv is zero-filled); p = first block.
hipError_t launch_pack_kbits(const void* x, int is_u8, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                             hipStream_t stream, int64_t nblk_out) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = nblk_out > 0 ? nblk_out : (nv + 127) / 128;
  const int64_t blocks = (nblk * (npad >> 8) + 3) / 4;  // one wave per (block, 256 samples)
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  uint32_t* pw = reinterpret_cast<uint32_t*>(p);
  const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
  const dim3 grid((unsigned)blocks), block(256);
  if (is_u8) {
    const uint8_t* xs = static_cast<const uint8_t*>(x);
    if (((ld & 7) == 0) && ((addr & 7) == 0)) {
      const int64_t blocks8 = (nblk * (npad >> 3) + 255) / 256;
      if (blocks8 > 0x7fffffffLL) return hipErrorInvalidValue;
      hipLaunchKernelGGL(pack_u8x8_kbits_kernel, dim3((unsigned)blocks8), block, 0, stream, xs, ld, nv, n, npad, nblk, pw,
                         flag);
      return hipGetLastError();
    }
    const bool vec = ((ld & 3) == 0) && ((addr & 3) == 0);
    if (vec) hipLaunchKernelGGL((pack_kbits_kernel<uint8_t, 4>), grid, block, 0, stream, xs, ld, nv, n, npad, nblk, pw, flag);
    else hipLaunchKernelGGL((pack_kbits_kernel<uint8_t, 1>), grid, block, 0, stream, xs, ld, nv, n, npad, nblk, pw, flag);
  } else {
    const bool vec = ((ld & 3) == 0) && ((addr & 15) == 0);
    const float* xs = static_cast<const float*>(x);
    // small calls (fewer waves than half the chip's wave slots): one 32-variant word per wave
    const bool small = nblk * (npad >> 8) < 2560;
    if (vec && small)
      hipLaunchKernelGGL((pack_kbits_kernel<float, 4, true, true>), dim3((unsigned)(nblk * (npad >> 8))), block, 0, stream, xs, ld, nv,
                         n, npad, nblk, pw, flag);
    else if (vec) hipLaunchKernelGGL((pack_kbits_kernel<float, 4, true>), grid, block, 0, stream, xs, ld, nv, n, npad, nblk, pw, flag);
    else hipLaunchKernelGGL((pack_kbits_kernel<float, 1>), grid, block, 0, stream, xs, ld, nv, n, npad, nblk, pw, flag);
  }
  return hipGetLastError();
}

// Persistent LDS-DMA-ring pre-pass (fp32 tile, pack_fp4_ring_ok(x, ld)): at most `wgs` workgroups of 4 waves.
// ring = R + 100 * nontemporal + 1000 * (waves at s_setprio 3); the product uses 8 (R = 8 rows in flight per wave, default
// cache policy: beside the contraction, with two workgroups per CU, 2.19 vs 2.25 ms per step for nontemporal loads --
// which win when the kernel has the chip to itself or one workgroup per CU, profiles/r03zl), the rest are harness knobs.
hipError_t launch_pack_kbits_ring(const float* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                                  hipStream_t stream, int64_t nblk_out, int wgs, int ring) {
  if (nv <= 0) return hipSuccess;
  if (!pack_fp4_ring_ok(x, ld) || nv > 0x3fffffffLL) return hipErrorInvalidValue;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = nblk_out > 0 ? nblk_out : (nv + 127) / 128;
  const int64_t units = nblk * (npad >> 8);
  if (units > 0x3fffffffLL) return hipErrorInvalidValue;  // 32-bit unit / row indices in the kernel
  uint32_t* pw = reinterpret_cast<uint32_t*>(p);
  const int64_t most = (units + 3) / 4;
  const dim3 grid((unsigned)(wgs > 0 && wgs < most ? wgs : most)), block(256);
#define PCOA_RINGK(R_, AUX_, PRIO_)                                                                                        \
  hipLaunchKernelGGL((pack_kbits_ring_kernel<R_, AUX_, PRIO_>), grid, block, 4 * R_ * 1024, stream, x, ld, (int)nv, n, npad, \
                     (int)units, pw, flag)
  ring %= 10000;  // (+ 10000 selected the natural bit order while a permuted form existed, r03w)
#ifdef PCOA_EXPERIMENTS
  if (ring >= 5000) {  // 5000 + aux: R = 8, the cache-policy bits of the LDS-DMA given directly (sc0 = 1, nt = 2, sc1 = 16)
    switch (ring - 5000) {
      case 1: PCOA_RINGK(8, 1, 0); break;
      case 3: PCOA_RINGK(8, 3, 0); break;
      case 16: PCOA_RINGK(8, 16, 0); break;
      case 17: PCOA_RINGK(8, 17, 0); break;
      case 18: PCOA_RINGK(8, 18, 0); break;
      default: PCOA_RINGK(8, 0, 0); break;
    }
    return hipGetLastError();
  }
  const int R = ring % 100, nt = (ring / 100) % 10, prio = ring / 1000;
#define PCOA_RINGK2(R_)                                                       \
  do {                                                                        \
    if (nt && prio) PCOA_RINGK(R_, 2, 3);                                     \
    else if (nt) PCOA_RINGK(R_, 2, 0);                                        \
    else if (prio) PCOA_RINGK(R_, 0, 3);                                      \
    else PCOA_RINGK(R_, 0, 0);                                                \
  } while (0)
  if (R == 8) PCOA_RINGK2(8);
  else if (R == 32) PCOA_RINGK2(32);
  else PCOA_RINGK2(16);
#undef PCOA_RINGK2
#else
  if ((ring / 100) % 10) PCOA_RINGK(8, 2, 0);
  else if (ring / 1000) PCOA_RINGK(8, 0, 3);
  else PCOA_RINGK(8, 0, 0);
#endif
#undef PCOA_RINGK
  return hipGetLastError();
}

// The uint8 form (pack_u8_kbits_ring_kernel): ld % 8 == 0 and an 8-byte aligned base, ld >= 16.
bool pack_u8_ring_ok(const void* x, int64_t ld) {
  return ((ld & 7) == 0) && ld >= 16 && ((reinterpret_cast<uintptr_t>(x) & 7) == 0);
}
hipError_t launch_pack_kbits_ring_u8(const uint8_t* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                                     hipStream_t stream, int64_t nblk_out, int wgs, int ring) {
  if (nv <= 0) return hipSuccess;
  if (!pack_u8_ring_ok(x, ld) || nv > 0x3fffffffLL || ld > 0x3fffffffLL) return hipErrorInvalidValue;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = nblk_out > 0 ? nblk_out : (nv + 127) / 128;
  const int64_t units = nblk * ((npad + 1023) >> 10);
  if (units > 0x3fffffffLL) return hipErrorInvalidValue;
  uint32_t* pw = reinterpret_cast<uint32_t*>(p);
  const int64_t most = (units + 3) / 4;
  const dim3 grid((unsigned)(wgs > 0 && wgs < most ? wgs : most)), block(256);
  // 8 rows in flight per wave; ring / 100 odd = nontemporal loads
  if ((ring / 100) % 10)
    hipLaunchKernelGGL((pack_u8_kbits_ring_kernel<8, 2>), grid, block, 32 << 10, stream, x, ld, (int)nv, n, npad, (int)units, pw, flag);
  else
    hipLaunchKernelGGL((pack_u8_kbits_ring_kernel<8, 0>), grid, block, 32 << 10, stream, x, ld, (int)nv, n, npad, (int)units, pw, flag);
  return hipGetLastError();
}

hipError_t launch_transpose_bits_kbits(const uint32_t* bits, int64_t ld_words, int64_t nv, int32_t n, int8_t* p,
                                       hipStream_t stream, int64_t nblk_out) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = nblk_out > 0 ? nblk_out : (nv + 127) / 128;
  const int64_t blocks = (nblk * (npad >> 8) + 3) / 4;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  uint32_t* pw = reinterpret_cast<uint32_t*>(p);
  const bool vec = ((ld_words & 3) == 0) && ((reinterpret_cast<uintptr_t>(bits) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(transpose_bits_kbits_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, bits, ld_words, nv, n,
                       npad, nblk, pw);
  else
    hipLaunchKernelGGL(transpose_bits_kbits_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, bits, ld_words, nv, n,
                       npad, nblk, pw);
  return hipGetLastError();
}

hipError_t launch_densify_csr_kbits(const int32_t* idx_dev, const int64_t* offs_dev, int64_t nv, int64_t offs_base,
                                    int8_t* p, int32_t n, int32_t* flag, hipStream_t stream, int64_t nblk_out) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  // the LDS form while a block of 128 variants x npad samples (16 B per sample) + its offsets fit 128 KiB of LDS
  // (npad <= 8,160); PCOA_CSR_GLOBAL_ATOMICS=1 gives the r04 form back (one wave per row, global atomic OR)
  static const bool force_global = [] {
    const char* v = std::getenv("PCOA_CSR_GLOBAL_ATOMICS");
    return v && std::atoi(v) != 0;
  }();
  const size_t lds = (size_t)npad * 16 + 132 * sizeof(int32_t);
  // what the device lets ONE workgroup have (sharedMemPerBlockOptin; 160 KiB on gfx950): asked once per device
  static thread_local int lds_dev = -1;
  static thread_local size_t lds_optin = 0;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev != lds_dev) {
    int v = 0;
    lds_optin = (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && v > 0) ? (size_t)v : (size_t)64 * 1024;
    lds_dev = dev;
    (void)hipGetLastError();
  }
  bool use_lds = !force_global && lds <= 128 * 1024 && lds <= lds_optin && nblk_out > 0 && nblk_out <= 0x7fffffffLL;
  constexpr int T = 512;
  if (use_lds && lds > 64 * 1024) {  // opt in to more than 64 KiB of dynamic LDS (per device: cheap enough to repeat)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(densify_csr_kbits_lds_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) {   // the global-atomic form below still works
      (void)hipGetLastError();
      use_lds = false;
    }
  }
  if (use_lds) {
    hipLaunchKernelGGL(densify_csr_kbits_lds_kernel<T>, dim3((unsigned)nblk_out), dim3(T), lds, stream, idx_dev, offs_dev, nv,
                       offs_base, reinterpret_cast<uint32_t*>(p), npad, n, flag);
    return hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(p, 0, (size_t)nblk_out * (size_t)npad * 16, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(densify_csr_kbits_kernel, dim3((unsigned)((nv + 3) / 4)), dim3(256), 0, stream, idx_dev, offs_dev, nv,
                     offs_base, reinterpret_cast<uint32_t*>(p), npad, n, flag);
  return hipGetLastError();
}

// Contraction of a k-bits operand.  mode: 0 = legacy split-K launch, 2 = lock-step (splitk k-streams on 8 / splitk XCDs,
// fails with hipErrorInvalidValue where the shape does not fit `num_cu`), 4 = even split over `num_cu` workgroups.
hipError_t launch_gram_kbits(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu, hipStream_t stream, int mode,
                             const int32_t* skip, GramStrip strip) {
  if (mode == 5) mode = 4;   // (the XCD k-segment split exists in the one-wave-per-SIMD kernel only)
  if (nv <= 0) return hipSuccess;
  const int cus = num_cu > 0 ? num_cu : 256;
  const int npad = (int)gram_packed_npad(n);
  const int ntile = npad / TJ;
  int64_t ntri64 = (int64_t)ntile * (ntile + 1) / 2;
  if (strip.cols > 0) {
    strip.cb0 = strip.col0 / TJ;
    const int cb1 = (strip.col0 + strip.cols + TJ - 1) / TJ;
    ntri64 = (int64_t)ntile * (cb1 - strip.cb0);
  }
  if (ntri64 > (1 << 28)) return hipErrorInvalidValue;
  const int ntri = (int)ntri64;
  const int64_t nstages = gram_kb_pad(nv, 2) / 4;  // blocks of 128 variants (the operand is padded to whole blocks)
  int64_t splitk = 1, stages_per = nstages, nblocks = 0;
  int xcd_map = 0;
  if (mode == 2) {
    if (strip.cols > 0) return hipErrorInvalidValue;
    splitk = gram_lockstep_splitk(n, cus);
    if (splitk == 0) return hipErrorInvalidValue;
    const int g = kNumXcd / (int)splitk;
    const int per = (ntri + g - 1) / g;
    stages_per = (nstages + splitk - 1) / splitk;
    nblocks = (int64_t)per * kNumXcd;
    xcd_map = 2;
  } else if (mode == 4) {
    // one workgroup per CU, but never runs shorter than 8 stages (1,024 variants): an epilogue costs about as much
    const int64_t nwork = (int64_t)ntri * nstages;
    nblocks = std::max<int64_t>(1, std::min<int64_t>(cus, nwork / 8));
    if (nblocks >= kNumXcd) nblocks = nblocks / kNumXcd * kNumXcd;  // the kernel's XCD-aware run order needs a multiple of 8
    xcd_map = 4;
  } else {
    const int64_t target = (int64_t)cus * 7;
    splitk = (target + ntri - 1) / ntri;
    if (debug_knobs().gram_splitk > 0) splitk = debug_knobs().gram_splitk;
    const int64_t max_by_work = nstages * 4 / 64;
    if (splitk > max_by_work) splitk = max_by_work;
    if (splitk < 1) splitk = 1;
    if (splitk >= kNumXcd) {
      splitk = (splitk / kNumXcd) * kNumXcd;
      xcd_map = 1;
    }
    stages_per = (nstages + splitk - 1) / splitk;
    nblocks = (int64_t)ntri * splitk;
  }
  if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)nblocks), block(512);
#define PCOA_LAUNCH_KBITS(NST_, LEFT_, ENC_)                                                                              \
  hipLaunchKernelGGL((gram_kbits_kernel<NST_, LEFT_, ENC_>), grid, block, 0, stream, p, npad, nstages, n, ntile, ntri,      \
                     (int)splitk, stages_per, s32, xcd_map, skip, strip)
#ifdef PCOA_EXPERIMENTS
#define PCOA_LAUNCH_KBITS_UNCAPPED(NST_, LEFT_, ENC_)                                                                     \
  hipLaunchKernelGGL((gram_kbits_uncapped_kernel<NST_, LEFT_, ENC_>), grid, block, 0, stream, p, npad, nstages, n, ntile,   \
                     ntri, (int)splitk, stages_per, s32, xcd_map, skip, strip)
  switch (g_kbits_variant) {  // harness knob (tools/exp_bits.hip)
    case 1: PCOA_LAUNCH_KBITS(4, 2, 0); break;   // plain encoding (7 + 7 operations per word pair), whole expansion in the read phase
    case 2: PCOA_LAUNCH_KBITS(3, 2, 1); break;   // conjugate weights, whole expansion in the read phase
    case 3: PCOA_LAUNCH_KBITS(4, 2, 2); break;   // shipped schedule on a 4-stage ring
    case 4: PCOA_LAUNCH_KBITS(3, 0, 2); break;   // no MFMAs behind the phase barrier
    case 5: PCOA_LAUNCH_KBITS(3, 2, 2); break;          // = default (kept for the r03s .. r03u harness numbering)
    case 6: PCOA_LAUNCH_KBITS(4, 2, 2); break;          // 4-stage ring (3 stages in flight)
    case 7: PCOA_LAUNCH_KBITS(6, 2, 2); break;          // 6-stage ring (5 stages = 40 KiB in flight per workgroup)
    case 8: PCOA_LAUNCH_KBITS_UNCAPPED(6, 2, 2); break; // 6-stage ring, no register cap
    case 9: PCOA_LAUNCH_KBITS_UNCAPPED(3, 2, 2); break; // shipped schedule, no register cap
    default: PCOA_LAUNCH_KBITS(3, 2, 2); break;
  }
#else
  PCOA_LAUNCH_KBITS(3, 2, 2);  // conjugate weights, expansion split over the phases, 3-stage ring, two MFMAs behind the barrier
#endif
#undef PCOA_LAUNCH_KBITS
  return hipGetLastError();
}

#endif  // PCOA_KBITS_LAUNCHERS
