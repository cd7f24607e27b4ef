// gram_kbits_w4.inl -- the k-bits contraction with ONE wave per SIMD and a hand-placed software pipeline (round 4).
// Included twice by gram_packed.hip, like gram_kbits.inl (kernels inside its anonymous namespace, launchers inside
// namespace pcoa); not a translation unit of its own.  Reference work it replaces: VariantsPca.scala:186-188.
//
// Why: gram_kbits_kernel<3,2,2> issues 4.25 expansion VALU per MFMA (a 128 x 64 wave tile needs 6
// fragments for 8 MFMAs) and hands the matrix pipe from one wave to its partner on the SIMD every 16 MFMAs; alone on the
// chip it reaches 0.63 of the FP4 peak.  Here:
//   * 256 threads = 4 waves as 2 x 2, one per SIMD; a wave owns a 128 x 128 block = 4 x 4 MFMA tiles, its 256 accumulator
//     registers live in the AGPR half of the unified file.  8 fragments feed 16 MFMAs: 48 expansion VALU per k-step = 3.0
//     per MFMA (conjugate-weight encoding of gram_kbits.inl: 5 operations per A word, 7 per B word);
//   * no phases: the wave's MFMA stream never stops.  Behind MFMA t of a k-step the wave issues its share of the NEXT
//     k-step's expansion (2-4 VALU), which executes in the 32-cycle shadow of the MFMA (MI355X_MICROARCH.md: one wave per
//     SIMD hides <= 5 single-issue instructions per 8-pass MFMA).  Fragments are double-buffered by k-step, the raw words
//     by stage; nothing an MFMA reads is written within 16 MFMAs of it;
//   * per stage (128 variants = 2 k-steps = 32 MFMAs per wave): 8 ds_read_b64, 2 global_load_lds_dwordx4, ONE s_barrier,
//     placed directly behind an MFMA so that the barrier's round trip hides behind it.
// LDS ring: NST stages of 8 KiB (StageBits of gram_kbits.inl: the LDS image is the global image).  Stage s+1 must have
// landed at the barrier of stage s; the DMA of stage s+NST goes into the slot of stage s behind the same barrier (every
// wave has the words of stage s in registers by then).  The vmcnt book-keeping is constant: a wave issues its PER DMA
// instructions every stage, clamped to the operand's last block beyond the run (harmless re-reads), so that
// s_waitcnt vmcnt(PER * (NST - 2)) always means "my share of stage s+1 has landed".
#ifdef PCOA_KBITS_W4_KERNELS

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct FragsW4 {
  i32x4 a[4];
  i32x4 b[4];
};

// MFMA T of a k-step: (mi, ni) = (T / 4, T % 4).  NOP wait states inside the statement (gram_kbits.inl: an MFMA issued in
// the cycle after a VALU instruction was seen to return rows 0, 1, 4, 5 of its tile wrong; nothing can be scheduled between
// the pad and the instruction here).
template <int T, int NOP>
__device__ __forceinline__ void w4_mfma(const FragsW4& f, f32x16 (&acc)[4][4]) {
  constexpr int mi = T / 4, ni = T % 4;
  if constexpr (NOP == 2)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(acc[mi][ni]) : "v"(f.a[mi]), "v"(f.b[ni]));
  else if constexpr (NOP == 1)
    asm volatile("s_nop 0\n\tv_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(acc[mi][ni]) : "v"(f.a[mi]), "v"(f.b[ni]));
  else
    asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+a"(acc[mi][ni]) : "v"(f.a[mi]), "v"(f.b[ni]));
}

// The expansion work placed behind MFMA T: fragments (A_g, B_g), g = T / 4, of the next k-step, 12 operations over four
// gaps as 3 + 3 + 4 + 2.  raw[0..3] = the words of the wave's 4 A rows, raw[4..7] = of its 4 B rows; WORD = k-step.
// The empty asm statements pin the arithmetic between the two MFMA statements around it (asm volatile statements keep
// their order; the arithmetic cannot rise above the statement that "redefines" its input nor sink below the one that
// "redefines" its output).
template <int T, int WORD>
__device__ __forceinline__ void w4_gap(u32x2 (&raw)[8], FragsW4& nf) {
  constexpr int g = T / 4, j = T % 4;
  if constexpr (j == 0) {
    asm volatile("" : "+v"(raw[g]));
    const uint32_t w = raw[g][WORD];
    nf.a[g][0] = (int)(w & 0x11111111u);
    nf.a[g][1] = (int)(w & 0x22222222u);
    nf.a[g][2] = (int)(w & 0x44444444u);
    asm volatile("" : "+v"(nf.a[g]));
  } else if constexpr (j == 1) {
    asm volatile("" : "+v"(raw[g]), "+v"(raw[4 + g]));
    const uint32_t wa = raw[g][WORD], wb = raw[4 + g][WORD];
    nf.a[g][3] = (int)((wa >> 1) & 0x44444444u);
    nf.b[g][1] = (int)(wb & 0x22222222u);
    asm volatile("" : "+v"(nf.a[g]), "+v"(nf.b[g]));
  } else if constexpr (j == 2) {
    asm volatile("" : "+v"(raw[4 + g]));
    const uint32_t wb = raw[4 + g][WORD];
    nf.b[g][0] = (int)((wb << 2) & 0x44444444u);
    nf.b[g][2] = (int)((wb >> 2) & 0x11111111u);
    asm volatile("" : "+v"(nf.b[g]));
  } else {
    asm volatile("" : "+v"(raw[4 + g]));
    const uint32_t wb = raw[4 + g][WORD];
    nf.b[g][3] = (int)((wb >> 3) & 0x11111111u);
    asm volatile("" : "+v"(nf.b[g]));
  }
}

// all eight fragments of one k-step at once (prologue of a run: nothing to hide behind)
template <int WORD>
__device__ __forceinline__ void w4_expand_all(u32x2 (&raw)[8], FragsW4& nf) {
  w4_gap<0, WORD>(raw, nf);  w4_gap<1, WORD>(raw, nf);  w4_gap<2, WORD>(raw, nf);  w4_gap<3, WORD>(raw, nf);
  w4_gap<4, WORD>(raw, nf);  w4_gap<5, WORD>(raw, nf);  w4_gap<6, WORD>(raw, nf);  w4_gap<7, WORD>(raw, nf);
  w4_gap<8, WORD>(raw, nf);  w4_gap<9, WORD>(raw, nf);  w4_gap<10, WORD>(raw, nf); w4_gap<11, WORD>(raw, nf);
  w4_gap<12, WORD>(raw, nf); w4_gap<13, WORD>(raw, nf); w4_gap<14, WORD>(raw, nf); w4_gap<15, WORD>(raw, nf);
}

// The words of slot SLOT: rows FIRST .. FIRST + COUNT - 1 of `raw` (0-3: A rows, 4-7: B rows), one ds_read_b64 each.  As asm
// statements: the compiler's wait-count pass cannot tell a ring slot being read from the slots the LDS-DMA in flight writes
// and would put s_waitcnt vmcnt(0) in front of a plain load (gram_kbits.inl, pack_u8_kbits_ring_kernel).  The wait is the
// caller's (w4_wait_words).
template <int SLOT, int R>
__device__ __forceinline__ void w4_read_one(uint32_t addr_a, uint32_t addr_b, u32x2 (&raw)[8]) {
  if constexpr (R < 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(raw[R]) : "v"(addr_a), "n"(SLOT * 8192 + (R & 3) * 512));
  else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(raw[R]) : "v"(addr_b), "n"(SLOT * 8192 + (R & 3) * 512));
}
template <int SLOT, int FIRST, int COUNT>
__device__ __forceinline__ void w4_read(uint32_t addr_a, uint32_t addr_b, u32x2 (&raw)[8]) {
  w4_read_one<SLOT, FIRST>(addr_a, addr_b, raw);
  if constexpr (COUNT > 1) w4_read<SLOT, FIRST + 1, COUNT - 1>(addr_a, addr_b, raw);
}
__device__ __forceinline__ void w4_wait_words(u32x2 (&raw)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]));
}

// DMA of one stage into slot SLOT: wave w brings in quarter w of panel I and (off the diagonal) quarter w of panel J.
// `base` = the block's first byte (wave-uniform, kept in SGPRs), off_i / off_j = the lane's byte offset inside the block.
template <int SLOT, bool DIAG>
__device__ __forceinline__ void w4_issue(StageBits* lds, const int8_t* base, uint32_t off_i, uint32_t off_j, int wave) {
  asm volatile("" : "+s"(base));
  __builtin_amdgcn_global_load_lds((gptr_t)(base + off_i), (lptr_t)&lds[SLOT].pi[wave * 64][0], 16, 0, 0);
  if constexpr (!DIAG)
    __builtin_amdgcn_global_load_lds((gptr_t)(base + off_j), (lptr_t)&lds[SLOT].pj[wave * 64][0], 16, 0, 0);
}

struct W4Run {
  const int8_t* next;  // block the next DMA reads (wave-uniform)
  int rem;             // blocks between `next` and the operand's last block: the clamp (may run negative)
  int64_t pitch;       // bytes per block = npad * 16
  uint32_t off_i, off_j, addr_a, addr_b;
};
__device__ __forceinline__ void w4_advance(W4Run& run) {  // scalar unit only
  const int64_t step = run.rem > 0 ? run.pitch : 0;
  run.rem -= 1;
  run.next += step;
}

// One stage.  Entry: f[0] = fragments of (stage s, k-step 0); raw[PAR] = words of stage s; this wave's DMA is issued through
// stage s + NST - 1.  Exit: the same for stage s + 1 with PAR flipped.
template <int NST, int SLOT, int PAR, bool DIAG, bool IDLE, int NOP>
__device__ __forceinline__ void w4_stage(StageBits* lds, W4Run& run, int wave, f32x16 (&acc)[4][4], FragsW4 (&f)[2],
                                         u32x2 (&raw)[2][8]) {
  constexpr int PER = DIAG ? 1 : 2;
  constexpr int NSLOT = (SLOT + 1) % NST;
  // ---- k-step 0 of stage s; behind its MFMAs: barrier, the words of stage s+1, the DMA of stage s+NST, fragments of k-step 1
  if constexpr (!IDLE) w4_mfma<0, NOP>(f[0], acc);
  wait_vmcnt<PER * (NST - 2)>();  // my share of stage s+1 has landed
  raw_barrier();                  // everybody's has; everybody holds the words of stage s in registers
  if constexpr (!IDLE) {
    w4_gap<0, 1>(raw[PAR], f[1]);
    w4_mfma<1, NOP>(f[0], acc);
    w4_read<NSLOT, 0, 2>(run.addr_a, run.addr_b, raw[PAR ^ 1]);
    w4_gap<1, 1>(raw[PAR], f[1]);
    w4_mfma<2, NOP>(f[0], acc);
    w4_gap<2, 1>(raw[PAR], f[1]);
    w4_mfma<3, NOP>(f[0], acc);
    w4_read<NSLOT, 2, 2>(run.addr_a, run.addr_b, raw[PAR ^ 1]);
    w4_gap<3, 1>(raw[PAR], f[1]);
    w4_mfma<4, NOP>(f[0], acc);
    w4_gap<4, 1>(raw[PAR], f[1]);
    w4_mfma<5, NOP>(f[0], acc);
    w4_read<NSLOT, 4, 2>(run.addr_a, run.addr_b, raw[PAR ^ 1]);
    w4_gap<5, 1>(raw[PAR], f[1]);
    w4_mfma<6, NOP>(f[0], acc);
    w4_gap<6, 1>(raw[PAR], f[1]);
    w4_mfma<7, NOP>(f[0], acc);
    w4_read<NSLOT, 6, 2>(run.addr_a, run.addr_b, raw[PAR ^ 1]);
    w4_gap<7, 1>(raw[PAR], f[1]);
    w4_mfma<8, NOP>(f[0], acc);
  }
  __builtin_amdgcn_sched_barrier(0);
  w4_issue<SLOT, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
  w4_advance(run);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!IDLE) {
    w4_gap<8, 1>(raw[PAR], f[1]);
    w4_mfma<9, NOP>(f[0], acc);
    w4_gap<9, 1>(raw[PAR], f[1]);
    w4_mfma<10, NOP>(f[0], acc);
    w4_gap<10, 1>(raw[PAR], f[1]);
    w4_mfma<11, NOP>(f[0], acc);
    w4_gap<11, 1>(raw[PAR], f[1]);
    w4_mfma<12, NOP>(f[0], acc);
    w4_gap<12, 1>(raw[PAR], f[1]);
    w4_mfma<13, NOP>(f[0], acc);
    w4_gap<13, 1>(raw[PAR], f[1]);
    w4_mfma<14, NOP>(f[0], acc);
    w4_gap<14, 1>(raw[PAR], f[1]);
    w4_mfma<15, NOP>(f[0], acc);
    w4_gap<15, 1>(raw[PAR], f[1]);
    // ---- k-step 1 of stage s; behind its MFMAs: fragments of (stage s+1, k-step 0)
    w4_mfma<0, NOP>(f[1], acc);
    w4_wait_words(raw[PAR ^ 1]);
    w4_gap<0, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<1, NOP>(f[1], acc);
    w4_gap<1, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<2, NOP>(f[1], acc);
    w4_gap<2, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<3, NOP>(f[1], acc);
    w4_gap<3, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<4, NOP>(f[1], acc);
    w4_gap<4, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<5, NOP>(f[1], acc);
    w4_gap<5, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<6, NOP>(f[1], acc);
    w4_gap<6, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<7, NOP>(f[1], acc);
    w4_gap<7, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<8, NOP>(f[1], acc);
    w4_gap<8, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<9, NOP>(f[1], acc);
    w4_gap<9, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<10, NOP>(f[1], acc);
    w4_gap<10, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<11, NOP>(f[1], acc);
    w4_gap<11, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<12, NOP>(f[1], acc);
    w4_gap<12, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<13, NOP>(f[1], acc);
    w4_gap<13, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<14, NOP>(f[1], acc);
    w4_gap<14, 0>(raw[PAR ^ 1], f[0]);
    w4_mfma<15, NOP>(f[1], acc);
    w4_gap<15, 0>(raw[PAR ^ 1], f[0]);
  }
}

template <int NST, bool DIAG, bool IDLE, int NOP, int... Is>
__device__ __forceinline__ void w4_round(StageBits* lds, W4Run& run, int count, int wave, f32x16 (&acc)[4][4],
                                         FragsW4 (&f)[2], u32x2 (&raw)[2][8], std::integer_sequence<int, Is...>) {
  ((Is < count ? w4_stage<NST, Is % NST, Is & 1, DIAG, IDLE, NOP>(lds, run, wave, acc, f, raw) : (void)0), ...);
}

// One run of `ns` stages of one tile, starting at block `first` (pointer to its first byte).
template <int NST, bool DIAG, bool IDLE, int NOP>
__device__ __forceinline__ void w4_loop(StageBits* lds, W4Run& run, const int8_t* first, int ns, int wave,
                                        f32x16 (&acc)[4][4]) {
  static_assert(NST % 2 == 0, "the raw-word parity of a slot must be a compile-time constant");
  constexpr int PER = DIAG ? 1 : 2;
  FragsW4 f[2];
  u32x2 raw[2][8];
  run.next = first;
  // prologue: stages 0 .. NST-1 go in flight (every slot is free: the previous run ended with vmcnt(0) + barrier)
  w4_issue<0, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
  w4_advance(run);
  w4_issue<1, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
  w4_advance(run);
  if constexpr (NST > 2) {
    w4_issue<2 % NST, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
    w4_advance(run);
    w4_issue<3 % NST, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
    w4_advance(run);
  }
  if constexpr (NST > 4) {
    w4_issue<4 % NST, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
    w4_advance(run);
    w4_issue<5 % NST, DIAG>(lds, run.next, run.off_i, run.off_j, wave);
    w4_advance(run);
  }
  static_assert(NST == 2 || NST == 4 || NST == 6, "prologue written for 2, 4 or 6 stages");
  wait_vmcnt<PER * (NST - 1)>();  // stage 0
  raw_barrier();
  if constexpr (!IDLE) {
    w4_read<0, 0, 8>(run.addr_a, run.addr_b, raw[0]);
    w4_wait_words(raw[0]);
    w4_expand_all<0>(raw[0], f[0]);
    asm volatile("s_nop 1");
  }
  int s = 0;
  for (; s + NST <= ns; s += NST)
    w4_round<NST, DIAG, IDLE, NOP>(lds, run, NST, wave, acc, f, raw, std::make_integer_sequence<int, NST>{});
  if (s < ns) w4_round<NST, DIAG, IDLE, NOP>(lds, run, ns - s, wave, acc, f, raw, std::make_integer_sequence<int, NST - 1>{});
  // drain: the clamped DMAs still in flight write slots the next run's prologue re-uses, and the last stage's speculative
  // ds_reads (words of a stage beyond the run, never used) must have returned before their registers are re-used
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
}

// Work decomposition and epilogue as gram_kbits_body (xcd_map 0 / 1 / 2 / 4); 256 x 256 workgroup tiles, a wave's block is
// rows [128 wm, +128) x columns [128 wn, +128) of it.
template <int NST, int NOP>
__global__ __launch_bounds__(256, 1) void gram_kbits_w4_kernel(const int8_t* __restrict__ p, int npad, int64_t nstages, int n,
                                                               int ntile, int ntri, int splitk, int64_t stages_per,
                                                               int32_t* __restrict__ s32, int xcd_map,
                                                               const int32_t* __restrict__ skip, GramStrip strip) {
  __shared__ __attribute__((aligned(16))) StageBits lds[NST];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.x;

  int64_t u, u_end;
  if (xcd_map == 4) {
    const int64_t nwork = (int64_t)ntri * nstages;
    const int64_t nwg = gridDim.x;
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

  const int l31 = lane & 31, hi = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)&lds[0];
  W4Run run;
  run.pitch = (int64_t)npad * 16;
  run.addr_a = lds0 + (uint32_t)((wm * 128 + l31) * 16 + hi * 8);

  while (u < u_end) {  // workgroup-uniform
    const int tile = (int)(u / nstages);
    const int64_t st_begin = u - (int64_t)tile * nstages;
    const int64_t left = u_end - u;
    const int ns = (int)((nstages - st_begin < left) ? (nstages - st_begin) : left);
    u += ns;

    int row_blk, col_blk;
    if (strip.cols > 0) {
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
    const bool diag = row_blk == col_blk && strip.cols == 0;
    const bool idle = diag && wm > wn;  // the block below the diagonal of a diagonal tile

    f32x16 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;

    run.off_i = (uint32_t)(col_i + wave * 64 + lane) * 16u;
    run.off_j = (uint32_t)(col_j + wave * 64 + lane) * 16u;
    // a diagonal tile brings in panel I only and reads its B rows from it
    run.addr_b = lds0 + (uint32_t)((diag ? 0 : 4096) + (wn * 128 + l31) * 16 + hi * 8);
    const int8_t* first = p + st_begin * run.pitch;
    run.rem = (int)(nstages - 1 - st_begin);
    if (diag) {
      if (idle) w4_loop<NST, true, true, NOP>(lds, run, first, ns, wave, acc);
      else w4_loop<NST, true, false, NOP>(lds, run, first, ns, wave, acc);
    } else {
      w4_loop<NST, false, false, NOP>(lds, run, first, ns, wave, acc);
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMA -> read of D
    if (!idle) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int j = col_j + wn * 128 + ni * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = col_i + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
  }
}

#endif  // PCOA_KBITS_W4_KERNELS

#ifdef PCOA_KBITS_W4_LAUNCHERS

#ifdef PCOA_EXPERIMENTS
int g_w4_variant = 0;  // harness knob
#endif

// Same contract as launch_gram_kbits (modes 0 / 2 / 4); 256-thread workgroups, one per CU.
hipError_t launch_gram_kbits_w4(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu, hipStream_t stream, int mode,
                                const int32_t* skip, GramStrip strip) {
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
  const int64_t nstages = gram_kb_pad(nv, 2) / 4;
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
    const int64_t nwork = (int64_t)ntri * nstages;
    nblocks = std::max<int64_t>(1, std::min<int64_t>(cus, nwork / 8));
    if (nblocks >= kNumXcd) nblocks = nblocks / kNumXcd * kNumXcd;
    xcd_map = 4;
  } else {
    const int64_t target = (int64_t)cus * 4;
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
  const dim3 grid((unsigned)nblocks), block(256);
#define PCOA_LAUNCH_W4(NST_, NOP_)                                                                                      \
  hipLaunchKernelGGL((gram_kbits_w4_kernel<NST_, NOP_>), grid, block, 0, stream, p, npad, nstages, n, ntile, ntri,        \
                     (int)splitk, stages_per, s32, xcd_map, skip, strip)
#ifdef PCOA_EXPERIMENTS
  switch (g_w4_variant) {
    case 1: PCOA_LAUNCH_W4(4, 1); break;
    case 2: PCOA_LAUNCH_W4(4, 0); break;
    case 3: PCOA_LAUNCH_W4(6, 2); break;
    case 4: PCOA_LAUNCH_W4(2, 2); break;
    default: PCOA_LAUNCH_W4(4, 2); break;
  }
#else
  PCOA_LAUNCH_W4(4, 2);
#endif
#undef PCOA_LAUNCH_W4
  return hipGetLastError();
}

#endif  // PCOA_KBITS_W4_LAUNCHERS
