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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct FragsW4 {
  i32x4 a[4];
  i32x4 b[4];
};

// ROLE of a wave in a run (wave-uniform, chosen per tile):
//   0 FULL  all 16 MFMA tiles of its 128 x 128 block (off-diagonal workgroup tiles, strip owners);
//   1 DIAG  a block ON the diagonal: the 10 MFMA tiles with mi <= ni (the other 6 mirror them);
//   2 / 3 HALF  the one block of a diagonal workgroup tile that lies above the diagonal is shared by the two waves that are
//         not on it: column tiles ni < 2 / ni >= 2 of it, 8 MFMAs each (the wave below the diagonal used to idle);
//   4 IDLE  (r04a form, kept for the harness) its share of the DMA and of the barriers, nothing else.
// A diagonal workgroup tile then costs 10 MFMA slots per k-step instead of 16 (10 of 55 tiles at N = 2504).
template <int ROLE, int T>
constexpr bool w4_has_mfma() {
  constexpr int mi = T / 4, ni = T % 4;
  return ROLE == 0 ? true : ROLE == 1 ? (mi <= ni) : ROLE == 2 ? (ni < 2) : ROLE == 3 ? (ni >= 2) : false;
}
template <int ROLE, int G>
constexpr bool w4_needs_b() {  // B fragment G (column tile G) is read by some MFMA of the role
  return ROLE == 2 ? (G < 2) : ROLE == 3 ? (G >= 2) : true;
}

// MFMA T of a k-step: (mi, ni) = (T / 4, T % 4).  NOP wait states inside the statement, where nothing can be scheduled
// between the pad and the instruction.  (gram_kbits.inl pads every MFMA: its late expansions write registers an MFMA issued
// a cycle later may still read.  Here nothing an MFMA reads is written within 16 MFMAs of it and the accumulators are AGPRs no
// VALU instruction touches: NOP = 0 is bit-exact on every shape and launch of tools/exp_w4, profiles/r04a.)
template <int T, int NOP>
__device__ __forceinline__ void w4_mfma(const FragsW4& f) {
  constexpr int mi = T / 4, ni = T % 4, lo = 16 * T, up = 16 * T + 15;
  if constexpr (NOP == 2)
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x64_f8f6f4 a[%c2:%c3], %0, %1, a[%c2:%c3] cbsz:4 blgp:4" ::"v"(f.a[mi]), "v"(f.b[ni]), "n"(lo), "n"(up));
  else if constexpr (NOP == 1)
    asm volatile("s_nop 0\n\tv_mfma_f32_32x32x64_f8f6f4 a[%c2:%c3], %0, %1, a[%c2:%c3] cbsz:4 blgp:4" ::"v"(f.a[mi]), "v"(f.b[ni]), "n"(lo), "n"(up));
  else
    asm volatile("v_mfma_f32_32x32x64_f8f6f4 a[%c2:%c3], %0, %1, a[%c2:%c3] cbsz:4 blgp:4" ::"v"(f.a[mi]), "v"(f.b[ni]), "n"(lo), "n"(up));
}

// The 256 accumulator registers are a[0:255] BY NAME: tuple (mi, ni) = a[16 (4 mi + ni) : +15].  They are not C++ variables:
// with all 256 AGPRs live the register allocator has no temporary to shuffle tuples with, and whenever it decided to move
// one (through VGPRs, through scratch) it did so with v_accvgpr_read / write right behind an asm MFMA whose wait states it
// cannot know -- wrong sums on some launches (profiles/r04j).  The compiler never allocates an AGPR in this kernel (it
// needs < 160 of the 256 architectural VGPRs); the clobber list of w4_zero_acc makes the kernel descriptor reserve them.
#define W4_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void w4_zero_acc() {
  i32x4 z = {0, 0, 0, 0};
  // sixteen MFMAs on a zero fragment with the constant 0 as C: 0.2 us, no temporary tuple
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[0:15], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[16:31], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[32:47], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[48:63], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[64:79], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[80:95], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[96:111], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[112:127], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[128:143], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[144:159], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[160:175], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[176:191], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[192:207], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[208:223], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[224:239], %0, %0, 0 cbsz:4 blgp:4\n\t"
      "v_mfma_f32_32x32x64_f8f6f4 a[240:255], %0, %0, 0 cbsz:4 blgp:4"
      :
      : "v"(z)
      : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", W4_A16(1), W4_A16(2), W4_A16(3), W4_A16(4), W4_A16(5), W4_A16(6),
        W4_A16(7), W4_A16(8), W4_A16(9), W4_A16(10), W4_A16(11), W4_A16(12), W4_A16(13), W4_A16(14), W4_A16(15), W4_A16(16),
        W4_A16(17), W4_A16(18), W4_A16(19), W4_A16(20), W4_A16(21), W4_A16(22), W4_A16(23), W4_A16(24), "a250", "a251", "a252",
        "a253", "a254", "a255");
}
#undef W4_A16
// accumulator register K of the file -> a VGPR (the caller has put the MFMA -> read wait states behind the last MFMA)
template <int K>
__device__ __forceinline__ int w4_acc_to_int() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(K));
  return (int)x;  // exact integers below 2^24
}

// The raw words of a stage: raw[q] = one ds_read2_b64 = {row 2q word 0, row 2q word 1, row 2q+1 word 0, row 2q+1 word 1} of the
// lane's half (rows 0-3: the wave's A rows, 4-7: its B rows; word = k-step).
template <int ROW, int WORD>
__device__ __forceinline__ uint32_t w4_word(const u32x4 (&raw)[4]) {
  return raw[ROW >> 1][(ROW & 1) * 2 + WORD];
}

// The expansion work placed behind MFMA T: fragments (A_g, B_g), g = T / 4, of the next k-step, 12 operations over four
// gaps as 3 + 3 + 4 + 2 (conjugate weights of gram_kbits.inl: A class 0..3 -> E2M1 0.5, 1, 2, 2; B -> 2, 1, 0.5, 0.5).
// w4_pin_in<T> "redefines" the words gap T reads and w4_pin_out<T> the fragments it writes: asm volatile statements keep
// their order, so the arithmetic of gap T can neither rise above its pin_in nor sink below its pin_out.
template <int T>
__device__ __forceinline__ void w4_pin_in(u32x4 (&raw)[4]) {
  constexpr int g = T / 4, j = T % 4;
  if constexpr (j == 0) asm volatile("" : "+v"(raw[g >> 1]));
  else if constexpr (j == 1) asm volatile("" : "+v"(raw[g >> 1]), "+v"(raw[2 + (g >> 1)]));
  else asm volatile("" : "+v"(raw[2 + (g >> 1)]));
}
template <int T, int ROLE = 0>
__device__ __forceinline__ void w4_pin_out(FragsW4& nf) {
  constexpr int g = T / 4, j = T % 4;
  constexpr bool nb = w4_needs_b<ROLE, g>();
  if constexpr (j == 0) asm volatile("" : "+v"(nf.a[g]));
  else if constexpr (j == 1 && nb) asm volatile("" : "+v"(nf.a[g]), "+v"(nf.b[g]));
  else if constexpr (j == 1) asm volatile("" : "+v"(nf.a[g]));
  else if constexpr (nb) asm volatile("" : "+v"(nf.b[g]));
}
template <int T, int WORD, int ROLE = 0>
__device__ __forceinline__ void w4_ops(const u32x4 (&raw)[4], FragsW4& nf) {
  constexpr int g = T / 4, j = T % 4;
  constexpr bool nb = w4_needs_b<ROLE, g>();
  const uint32_t wa = w4_word<g, WORD>(raw), wb = w4_word<4 + g, WORD>(raw);
  if constexpr (j == 0) {
    nf.a[g][0] = (int)(wa & 0x11111111u);
    nf.a[g][1] = (int)(wa & 0x22222222u);
    nf.a[g][2] = (int)(wa & 0x44444444u);
  } else if constexpr (j == 1) {
    nf.a[g][3] = (int)((wa >> 1) & 0x44444444u);
    if constexpr (nb) nf.b[g][1] = (int)(wb & 0x22222222u);
  } else if constexpr (j == 2) {
    if constexpr (nb) {
      nf.b[g][0] = (int)((wb << 2) & 0x44444444u);
      nf.b[g][2] = (int)((wb >> 2) & 0x11111111u);
    }
  } else {
    if constexpr (nb) nf.b[g][3] = (int)((wb >> 3) & 0x11111111u);
  }
}
template <int T, int WORD, int ROLE = 0>
__device__ __forceinline__ void w4_gap(u32x4 (&raw)[4], FragsW4& nf) {
  w4_pin_in<T>(raw);
  w4_ops<T, WORD, ROLE>(raw, nf);
  w4_pin_out<T, ROLE>(nf);
}

// One statement that reads and "redefines" all eight fragments of a buffer: it keeps the buffer's live range unbroken across
// the k-step in which nothing reads it, so that the register allocator never moves it -- left free (OPT & 8 without this),
// it puts a fragment of the NEXT k-step into the registers of a fragment that died one MFMA ago, and an MFMA keeps reading its
// A / B registers for some cycles after it has issued (gram_kbits.inl, profiles/r03g_kbits_hazards.txt).
__device__ __forceinline__ void w4_tie(FragsW4& f) {
  asm volatile("" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]));
}

// all eight fragments of one k-step at once (prologue of a run: nothing to hide behind)
template <int WORD, int ROLE = 0>
__device__ __forceinline__ void w4_expand_all(u32x4 (&raw)[4], FragsW4& nf) {
  w4_gap<0, WORD, ROLE>(raw, nf);  w4_gap<1, WORD, ROLE>(raw, nf);  w4_gap<2, WORD, ROLE>(raw, nf);  w4_gap<3, WORD, ROLE>(raw, nf);
  w4_gap<4, WORD, ROLE>(raw, nf);  w4_gap<5, WORD, ROLE>(raw, nf);  w4_gap<6, WORD, ROLE>(raw, nf);  w4_gap<7, WORD, ROLE>(raw, nf);
  w4_gap<8, WORD, ROLE>(raw, nf);  w4_gap<9, WORD, ROLE>(raw, nf);  w4_gap<10, WORD, ROLE>(raw, nf); w4_gap<11, WORD, ROLE>(raw, nf);
  w4_gap<12, WORD, ROLE>(raw, nf); w4_gap<13, WORD, ROLE>(raw, nf); w4_gap<14, WORD, ROLE>(raw, nf); w4_gap<15, WORD, ROLE>(raw, nf);
}

// raw[Q] of a slot: rows 2Q and 2Q+1 (512 bytes apart) by one ds_read2_b64.  As asm statements: the compiler's wait-count
// pass cannot tell a ring slot being read from the slots the LDS-DMA in flight writes and would put s_waitcnt vmcnt(0) in
// front of a plain load (gram_kbits.inl, pack_u8_kbits_ring_kernel).  The wait is the caller's (w4_wait_words).
// addr = the lane's address of row 0 (A) / row 4 (B) in that slot.
template <int Q>
__device__ __forceinline__ void w4_read(uint32_t addr_a, uint32_t addr_b, u32x4 (&raw)[4]) {
  if constexpr (Q < 2) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(raw[Q]) : "v"(addr_a), "n"(Q * 128), "n"(Q * 128 + 64));
  else asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(raw[Q]) : "v"(addr_b), "n"((Q - 2) * 128), "n"((Q - 2) * 128 + 64));
}
__device__ __forceinline__ void w4_wait_words(u32x4 (&raw)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]));
}

template <int NST>
struct W4Run {
  const int8_t* next;  // block the next DMA reads (wave-uniform)
  int rem;             // blocks between `next` and the operand's last block: the clamp (may run negative)
  int64_t pitch;       // bytes per block = npad * 16
  uint32_t off_i, off_j;            // the lane's byte offsets inside a block (its 16 bytes of panel I / J)
  uint32_t addr_a[NST], addr_b[NST];  // the lane's LDS read addresses per slot
  uint32_t dst_i[NST], dst_j[NST];    // the wave's LDS-DMA destinations per slot (wave-uniform)
};
template <int NST>
__device__ __forceinline__ void w4_advance(W4Run<NST>& run) {  // scalar unit only
  const int64_t step = run.rem > 0 ? run.pitch : 0;
  run.rem -= 1;
  run.next += step;
}

// One 1-KiB piece of a stage into slot SLOT: J = 0 quarter `wave` of panel I, J = 1 of panel J.  OPT & 1: as one asm
// statement in the SGPR-base + 32-bit-VGPR-offset form (the builtin adds base and offset per lane with a 64-bit VALU add
// right in front of the load, and re-materialises M0 around it).
template <int NST, int SLOT, int J, int OPT>
__device__ __forceinline__ void w4_issue_one(StageBits* lds, const W4Run<NST>& run, int wave) {
  const uint32_t off = J ? run.off_j : run.off_i;
  if constexpr (OPT & 1) {
    const uint32_t dst = J ? run.dst_j[SLOT] : run.dst_i[SLOT];
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(run.next), "s"(dst) : "memory");
  } else {
    const int8_t* base = run.next;
    asm volatile("" : "+s"(base));
    if constexpr (J) __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)&lds[SLOT].pj[wave * 64][0], 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)&lds[SLOT].pi[wave * 64][0], 16, 0, 0);
  }
}
template <int NST, int SLOT, bool DIAG, int OPT>
__device__ __forceinline__ void w4_issue(StageBits* lds, W4Run<NST>& run, int wave) {
  w4_issue_one<NST, SLOT, 0, OPT>(lds, run, wave);
  if constexpr (!DIAG) w4_issue_one<NST, SLOT, 1, OPT>(lds, run, wave);
  w4_advance(run);
}

// gap T of a k-step.  OPT & 2: pin_in sits in FRONT of the MFMA statement (hipcc pads one wait state between an asm
// statement and a VALU instruction that reads its outputs; with the MFMA in between the pad is not needed -- the arithmetic
// may then also be scheduled in front of that MFMA, i.e. a gap earlier, which is as good).
#define W4_STEP(T_, K_, W_, P_)                                                                        \
  do {                                                                                                 \
    if constexpr (!(DBG & 1) && (OPT & 2) && !(OPT & 88)) w4_pin_in<T_>(raw[P_]);                      \
    if constexpr (!(DBG & 16) && w4_has_mfma<ROLE, T_>()) w4_mfma<T_, NOP>(f[K_]);                     \
    if constexpr (!(DBG & 1)) {                                                                        \
      if constexpr (OPT & 88) __builtin_amdgcn_sched_barrier(0);                                       \
      if constexpr ((OPT & 16) && (T_) < 15) w4_pin_in<((T_) < 15 ? (T_) + 1 : 15)>(raw[P_]);          \
      if constexpr (!(OPT & 2) && !(OPT & 88)) w4_pin_in<T_>(raw[P_]);                                 \
      w4_ops<T_, W_, ROLE>(raw[P_], f[(K_) ^ 1]);                                                      \
      if constexpr (OPT & 8) __builtin_amdgcn_sched_barrier(0);                                        \
      else w4_pin_out<T_, ROLE>(f[(K_) ^ 1]);                                                          \
    }                                                                                                  \
  } while (0)
#define W4_READ(Q_) do { if constexpr (!(DBG & 4)) w4_read<Q_>(run.addr_a[NSLOT], run.addr_b[NSLOT], raw[PAR ^ 1]); } while (0)
// One stage.  Entry: f[0] = fragments of (stage s, k-step 0); raw[PAR] = words of stage s; this wave's DMA is issued through
// stage s + NST - 1.  Exit: the same for stage s + 1 with PAR flipped.
template <int NST, int SLOT, int PAR, bool DIAG, int ROLE, int NOP, int OPT, int DBG>
__device__ __forceinline__ void w4_stage(StageBits* lds, W4Run<NST>& run, int wave, FragsW4 (&f)[2],
                                         u32x4 (&raw)[2][4]) {
  constexpr int PER = DIAG ? 1 : 2;
  constexpr int NSLOT = (SLOT + 1) % NST;
  if constexpr (ROLE == 4) {
    if constexpr (!(DBG & 8)) wait_vmcnt<PER * (NST - 2)>();
    if constexpr (!(DBG & 2)) raw_barrier();
    if constexpr (!(DBG & 8)) w4_issue<NST, SLOT, DIAG, OPT>(lds, run, wave);
    return;
  }
  // ---- k-step 0 of stage s: MFMAs on f[0]; behind them the barrier, the words of stage s+1, DMA of stage s+NST, and the
  //      fragments of k-step 1 (word 1 of raw[PAR] -> f[1])
  if constexpr (!(DBG & 1) && (OPT & 2) && !(OPT & 88)) w4_pin_in<0>(raw[PAR]);
  if constexpr (!(DBG & 1) && (OPT & 32)) w4_tie(f[1]);
  if constexpr (!(DBG & 16) && w4_has_mfma<ROLE, 0>()) w4_mfma<0, NOP>(f[0]);
  if constexpr (!(DBG & 8)) wait_vmcnt<PER * (NST - 2)>();  // my share of stage s+1 has landed
  if constexpr (!(DBG & 2)) raw_barrier();  // everybody's has; everybody holds the words of stage s in registers
  if constexpr (!(DBG & 1)) {
    if constexpr (OPT & 16) w4_pin_in<1>(raw[PAR]);
    if constexpr (!(OPT & 2) && !(OPT & 88)) w4_pin_in<0>(raw[PAR]);
    w4_ops<0, 1, ROLE>(raw[PAR], f[1]);
    if constexpr (OPT & 8) __builtin_amdgcn_sched_barrier(0);
    else w4_pin_out<0, ROLE>(f[1]);
  }
  W4_STEP(1, 0, 1, PAR);
  W4_READ(0);
  W4_STEP(2, 0, 1, PAR);
  W4_STEP(3, 0, 1, PAR);
  W4_READ(1);
  W4_STEP(4, 0, 1, PAR);
  W4_STEP(5, 0, 1, PAR);
  W4_READ(2);
  W4_STEP(6, 0, 1, PAR);
  W4_STEP(7, 0, 1, PAR);
  W4_READ(3);
  W4_STEP(8, 0, 1, PAR);
  if constexpr (!(DBG & 8)) {
    __builtin_amdgcn_sched_barrier(0);
    w4_issue_one<NST, SLOT, 0, OPT>(lds, run, wave);
    if constexpr (!DIAG && !(OPT & 4)) w4_issue_one<NST, SLOT, 1, OPT>(lds, run, wave);
    if constexpr (DIAG || !(OPT & 4)) w4_advance(run);
    __builtin_amdgcn_sched_barrier(0);
  }
  W4_STEP(9, 0, 1, PAR);
  W4_STEP(10, 0, 1, PAR);
  W4_STEP(11, 0, 1, PAR);
  W4_STEP(12, 0, 1, PAR);
  W4_STEP(13, 0, 1, PAR);
  W4_STEP(14, 0, 1, PAR);
  W4_STEP(15, 0, 1, PAR);
  // ---- k-step 1 of stage s: MFMAs on f[1]; behind them the fragments of (stage s+1, k-step 0) (word 0 of raw[PAR^1] -> f[0])
  if constexpr (!(DBG & 4)) w4_wait_words(raw[PAR ^ 1]);
  if constexpr (!(DBG & 1) && (OPT & 32)) w4_tie(f[0]);
  W4_STEP(0, 1, 0, PAR ^ 1);
  W4_STEP(1, 1, 0, PAR ^ 1);
  W4_STEP(2, 1, 0, PAR ^ 1);
  W4_STEP(3, 1, 0, PAR ^ 1);
  W4_STEP(4, 1, 0, PAR ^ 1);
  W4_STEP(5, 1, 0, PAR ^ 1);
  W4_STEP(6, 1, 0, PAR ^ 1);
  W4_STEP(7, 1, 0, PAR ^ 1);
  W4_STEP(8, 1, 0, PAR ^ 1);
  if constexpr (!(DBG & 8) && !DIAG && (OPT & 4)) {
    __builtin_amdgcn_sched_barrier(0);
    w4_issue_one<NST, SLOT, 1, OPT>(lds, run, wave);
    w4_advance(run);
    __builtin_amdgcn_sched_barrier(0);
  }
  W4_STEP(9, 1, 0, PAR ^ 1);
  W4_STEP(10, 1, 0, PAR ^ 1);
  W4_STEP(11, 1, 0, PAR ^ 1);
  W4_STEP(12, 1, 0, PAR ^ 1);
  W4_STEP(13, 1, 0, PAR ^ 1);
  W4_STEP(14, 1, 0, PAR ^ 1);
  W4_STEP(15, 1, 0, PAR ^ 1);
}
#undef W4_STEP
#undef W4_READ

template <int NST, bool DIAG, int ROLE, int NOP, int OPT, int DBG, int... Is>
__device__ __forceinline__ void w4_round(StageBits* lds, W4Run<NST>& run, int count, int wave,
                                         FragsW4 (&f)[2], u32x4 (&raw)[2][4], std::integer_sequence<int, Is...>) {
  ((Is < count ? w4_stage<NST, Is % NST, Is & 1, DIAG, ROLE, NOP, OPT, DBG>(lds, run, wave, f, raw) : (void)0), ...);
}

template <int NST, bool DIAG, int OPT, int I = 0>
__device__ __forceinline__ void w4_prologue_issue(StageBits* lds, W4Run<NST>& run, int wave) {
  if constexpr (I < NST) {
    w4_issue<NST, I, DIAG, OPT>(lds, run, wave);
    w4_prologue_issue<NST, DIAG, OPT, I + 1>(lds, run, wave);
  }
}

// One run of `ns` stages of one tile, starting at block `first` (pointer to its first byte).
template <int NST, bool DIAG, int ROLE, int NOP, int OPT, int DBG>
__device__ __forceinline__ void w4_loop(StageBits* lds, W4Run<NST>& run, const int8_t* first, int ns, int wave) {
  static_assert(NST % 2 == 0, "the raw-word parity of a slot must be a compile-time constant");
  constexpr int PER = DIAG ? 1 : 2;
  FragsW4 f[2];
  u32x4 raw[2][4];
  {  // workgroup-uniform, but it comes out of a 64-bit division done on the vector unit: the asm DMA wants it in SGPRs
    const uint64_t a = (uint64_t)(uintptr_t)first;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    run.next = reinterpret_cast<const int8_t*>(((uint64_t)hi << 32) | lo);
  }
  // prologue: stages 0 .. NST-1 go in flight (every slot is free: the previous run ended with vmcnt(0) + barrier)
  w4_prologue_issue<NST, DIAG, OPT>(lds, run, wave);
  wait_vmcnt<PER * (NST - 1)>();  // stage 0
  raw_barrier();
  if constexpr (ROLE != 4) {
    w4_read<0>(run.addr_a[0], run.addr_b[0], raw[0]);
    w4_read<1>(run.addr_a[0], run.addr_b[0], raw[0]);
    w4_read<2>(run.addr_a[0], run.addr_b[0], raw[0]);
    w4_read<3>(run.addr_a[0], run.addr_b[0], raw[0]);
    w4_wait_words(raw[0]);
    w4_expand_all<0, ROLE>(raw[0], f[0]);
    asm volatile("s_nop 1");
  }
  int s = 0;
  for (; s + NST <= ns; s += NST)
    w4_round<NST, DIAG, ROLE, NOP, OPT, DBG>(lds, run, NST, wave, f, raw, std::make_integer_sequence<int, NST>{});
  if (s < ns)
    w4_round<NST, DIAG, ROLE, NOP, OPT, DBG>(lds, run, ns - s, wave, f, raw, std::make_integer_sequence<int, NST - 1>{});
  // drain: the clamped DMAs still in flight write slots the next run's prologue re-uses, and the last stage's speculative
  // ds_reads (words of a stage beyond the run, never used) must have returned before their registers are re-used
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
}

// Epilogue of one wave: its 128 x 128 block is added into S32 with integer atomics (exact integers below 2^24 in the fp32
// accumulators).  Element (mi, ni, r) of a lane is row i0 + 32 mi + (r & 3) + 8 (r >> 2) + 4 hi, column j0 + 32 ni + l31.  The row
// base of each mi is wave-uniform (SGPR pair), a lane keeps ONE 32-bit byte offset that walks down the rows, the four column
// tiles are immediate offsets: v_accvgpr_read + v_cvt + v_add + global_atomic_add per element.  (What bounds the epilogue is
// not the instruction count but the atomics themselves, ~27 us per 256 x 256 tile whatever their width: 64-bit atomics on
// column pairs changed nothing, profiles/r04h.)  MASKED: tiles on the diagonal, at the edge of the matrix or of a strip test
// every element (symmetric job: j >= i and j < n; strip: i < n and jorg <= j < jend) and skip zeros.
template <bool MASKED, int MI, int NI, int R>
__device__ __forceinline__ void w4_store_elem(const int32_t* rowbase, uint32_t& voff, uint32_t ld4, int i0, int jj, int jorg, int n,
                                              bool sym, int jend, int hi) {
  const int v = w4_acc_to_int<16 * (4 * MI + NI) + R>();
  bool ok = true;
  if constexpr (MASKED) {
    const int ii = i0 + MI * 32 + (R & 3) + 8 * (R >> 2) + 4 * hi;
    ok = ii < n && jj < jend && jj >= (sym ? ii : jorg) && v != 0;
  }
  if (ok) asm volatile("global_atomic_add %0, %1, %2 offset:%3" ::"v"(voff), "v"(v), "s"(rowbase), "n"(NI * 128) : "memory");
  voff += ((R & 3) == 3) ? 5u * ld4 : ld4;
}
template <bool MASKED, int MI, int NI, int... Rs>
__device__ __forceinline__ void w4_store_tuple(const int32_t* rowbase, uint32_t voff, uint32_t ld4, int i0, int jj, int jorg, int n,
                                               bool sym, int jend, int hi, std::integer_sequence<int, Rs...>) {
  (w4_store_elem<MASKED, MI, NI, Rs>(rowbase, voff, ld4, i0, jj, jorg, n, sym, jend, hi), ...);
}
template <bool MASKED, int MI, int ROLE>
__device__ __forceinline__ void w4_store_rows(int32_t* s32, int64_t ld, uint32_t voff0, int i0, int j0, int jorg, int n, bool sym,
                                              int jend, int l31, int hi) {
  const uint64_t a = (uint64_t)(uintptr_t)(s32 + (int64_t)(i0 + MI * 32) * ld + (j0 - jorg));
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), up = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const int32_t* rowbase = reinterpret_cast<const int32_t*>(((uint64_t)up << 32) | lo);
  const uint32_t ld4 = (uint32_t)ld * 4u;
  constexpr std::make_integer_sequence<int, 16> rs{};
  // only the tiles the role computed (the accumulators of the others hold zeros)
  if constexpr (w4_has_mfma<ROLE, 4 * MI + 0>()) w4_store_tuple<MASKED, MI, 0>(rowbase, voff0, ld4, i0, j0 + l31, jorg, n, sym, jend, hi, rs);
  if constexpr (w4_has_mfma<ROLE, 4 * MI + 1>()) w4_store_tuple<MASKED, MI, 1>(rowbase, voff0, ld4, i0, j0 + 32 + l31, jorg, n, sym, jend, hi, rs);
  if constexpr (w4_has_mfma<ROLE, 4 * MI + 2>()) w4_store_tuple<MASKED, MI, 2>(rowbase, voff0, ld4, i0, j0 + 64 + l31, jorg, n, sym, jend, hi, rs);
  if constexpr (w4_has_mfma<ROLE, 4 * MI + 3>()) w4_store_tuple<MASKED, MI, 3>(rowbase, voff0, ld4, i0, j0 + 96 + l31, jorg, n, sym, jend, hi, rs);
}
template <bool MASKED, int ROLE = 0>
__device__ __forceinline__ void w4_store(int32_t* s32, int64_t ld, int i0, int j0, int jorg, int n, bool sym, int jend, int l31,
                                         int hi) {
  const uint32_t voff0 = ((uint32_t)(4 * hi) * (uint32_t)ld + (uint32_t)l31) * 4u;
  w4_store_rows<MASKED, 0, ROLE>(s32, ld, voff0, i0, j0, jorg, n, sym, jend, l31, hi);
  w4_store_rows<MASKED, 1, ROLE>(s32, ld, voff0, i0, j0, jorg, n, sym, jend, l31, hi);
  w4_store_rows<MASKED, 2, ROLE>(s32, ld, voff0, i0, j0, jorg, n, sym, jend, l31, hi);
  w4_store_rows<MASKED, 3, ROLE>(s32, ld, voff0, i0, j0, jorg, n, sym, jend, l31, hi);
}

// Work decomposition and epilogue as gram_kbits_body (xcd_map 0 / 1 / 2 / 4); 256 x 256 workgroup tiles, a wave's block is
// rows [128 wm, +128) x columns [128 wn, +128) of it.
#ifdef PCOA_EXPERIMENTS
__device__ unsigned long long g_w4_clk[4];  // harness: shader-clock and 100-MHz ticks of block 0's life
#endif
template <int NST, int NOP, int OPT, int DBG = 0>
__global__ __launch_bounds__(256, 1) void gram_kbits_w4_kernel(const int8_t* __restrict__ p, int npad, int64_t nstages, int n,
                                                               int ntile, int ntri, int splitk, int64_t stages_per,
                                                               int32_t* __restrict__ s32, int xcd_map,
                                                               const int32_t* __restrict__ skip, GramStrip strip, int wdiag,
                                                               int wepi) {
  __shared__ __attribute__((aligned(16))) StageBits lds[NST];
  if (skip != nullptr && *skip != 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int b = blockIdx.x;
#ifdef PCOA_EXPERIMENTS
  const unsigned long long clk0 = clock64(), rt0 = wall_clock64();
#endif

  int64_t u, u_end;
  // xcd_map 5 (r06): every XCD takes its own EIGHTH of the k-range -- segment [seg_off, seg_off + seg_len) -- for ALL tiles and
  // splits that by cost over its 32 workgroups: the workgroups of one L2 then stream the same 1 / 8 of the operand (tile after
  // tile, at most 1 / 8 of the k-range apart) instead of 32 unrelated (tile, k) positions of the whole operand
  int64_t seg_len = nstages, seg_off = 0;
  if (xcd_map == 4 || xcd_map == 5) {
    int64_t nwg = gridDim.x;
    int64_t slot = (nwg % kNumXcd == 0) ? (int64_t)(b & 7) * (nwg / kNumXcd) + (b >> 3) : (int64_t)b;
    if (xcd_map == 5) {   // (the launcher makes gridDim.x a multiple of 8 and nstages >= 8)
      const int xcd = b & 7;
      seg_off = nstages * xcd / kNumXcd;
      seg_len = nstages * (xcd + 1) / kNumXcd - seg_off;
      nwg = gridDim.x / kNumXcd;
      slot = b >> 3;
    }
    const int64_t nstages = seg_len;   // the split below works on the segment (shadows the kernel argument on purpose)
    const int64_t nwork = (int64_t)ntri * nstages;
    if (wdiag > 0 && wdiag < 16 && strip.cols == 0 && ntile <= BAND) {
      // a stage of a diagonal tile costs wdiag / 16 of a stage of any other (wave roles, w4_has_mfma): equal shares of the COST.
      // Tiles are in row-major order here (tile_coords, one band) and the first tile of a row is the diagonal one.
      // wepi: what the run a workgroup starts when it crosses into the next tile costs on top (accumulator flush + refill of the
      // ring, in the same units: 16 = a stage of a full tile) -- a stretch of that length in front of every tile
      auto unit_of = [&](int64_t x) -> int64_t {  // cost position -> tile * nstages + stage
        int64_t t0 = 0;
        for (int r = 0; r < ntile; ++r) {
          const int64_t dspan = wepi + nstages * wdiag, fspan = wepi + nstages * 16, rspan = dspan + fspan * (ntile - r - 1);
          if (x < rspan) {
            if (x < dspan) return t0 * nstages + (x > wepi ? (x - wepi) / wdiag : 0);
            const int64_t y = x - dspan, z = y % fspan;
            return (t0 + 1 + y / fspan) * nstages + (z > wepi ? (z - wepi) / 16 : 0);
          }
          x -= rspan;
          t0 += ntile - r;
        }
        return nwork;
      };
      const int64_t cost = nstages * ((int64_t)ntile * wdiag + (int64_t)(ntri - ntile) * 16) + (int64_t)ntri * wepi;
      u = unit_of(cost * slot / nwg);
      u_end = (slot + 1 == nwg) ? nwork : unit_of(cost * (slot + 1) / nwg);
    } else {
      u = nwork * slot / nwg;
      u_end = nwork * (slot + 1) / nwg;
    }
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
  W4Run<NST> run;
  run.pitch = (DBG & 32) ? 0 : (int64_t)npad * 16;  // DBG & 32: every DMA re-reads the run's first block (always L2-warm)
#pragma unroll
  for (int q = 0; q < NST; ++q) {
    run.addr_a[q] = lds0 + (uint32_t)(q * 8192 + (wm * 128 + l31) * 16 + hi * 8);
    run.addr_b[q] = lds0 + (uint32_t)(q * 8192 + 4096 + (wn * 128 + l31) * 16 + hi * 8);
    run.dst_i[q] = lds0 + (uint32_t)(q * 8192 + wave * 1024);
    run.dst_j[q] = lds0 + (uint32_t)(q * 8192 + 4096 + wave * 1024);
  }

  while (u < u_end) {  // workgroup-uniform
    const int tile = (int)(u / seg_len);
    const int64_t st_rel = u - (int64_t)tile * seg_len;
    const int64_t left = u_end - u;
    const int ns = (int)((seg_len - st_rel < left) ? (seg_len - st_rel) : left);
    const int64_t st_begin = seg_off + st_rel;
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
    // A diagonal tile runs the same code: panel J is panel I brought in a second time (10 of 55 tiles at N = 2504; the extra
    // 4 KiB per stage come out of the L1).  In the r04a form (wdiag = 0, role 4) the wave whose block lies below the diagonal
    // keeps its share of the DMA and of the barriers and issues nothing else; the default gives every wave a part of the tile's
    // upper triangle instead (the kernel is power-bound: MFMAs nobody needs cost clock).
    const bool diag_tile = row_blk == col_blk && strip.cols == 0;
    // wave roles on a diagonal tile (w4_has_mfma): the two blocks on the diagonal compute their upper MFMA tiles, the block
    // above it is shared by the other two waves -- the one below the diagonal takes the place of (wm, wn) = (0, 1) too
    int role = 0, rm = wm, rn = wn;
    if (diag_tile && wdiag > 0) {
      role = (wm == wn) ? 1 : (wm < wn) ? 2 : 3;
      if (wm != wn) { rm = 0; rn = 1; }
    } else if (diag_tile && wm > wn) {
      role = 4;
    }
    role = __builtin_amdgcn_readfirstlane(role);
    if (wdiag > 0) {
#pragma unroll
      for (int q = 0; q < NST; ++q) {
        run.addr_a[q] = lds0 + (uint32_t)(q * 8192 + (rm * 128 + l31) * 16 + hi * 8);
        run.addr_b[q] = lds0 + (uint32_t)(q * 8192 + 4096 + (rn * 128 + l31) * 16 + hi * 8);
      }
    }

    run.off_i = (uint32_t)(col_i + wave * 64 + lane) * 16u;
    run.off_j = (uint32_t)(col_j + wave * 64 + lane) * 16u;
    const int8_t* first = p + ((DBG & 64) ? 0 : st_begin) * run.pitch;
    run.rem = __builtin_amdgcn_readfirstlane((int)(nstages - 1 - ((DBG & 64) ? 0 : st_begin)));

    if (role != 4) w4_zero_acc();
    if (role == 0) w4_loop<NST, false, 0, NOP, OPT, DBG>(lds, run, first, ns, wave);
    else if (role == 1) w4_loop<NST, false, 1, NOP, OPT, DBG>(lds, run, first, ns, wave);
    else if (role == 2) w4_loop<NST, false, 2, NOP, OPT, DBG>(lds, run, first, ns, wave);
    else if (role == 3) w4_loop<NST, false, 3, NOP, OPT, DBG>(lds, run, first, ns, wave);
    else w4_loop<NST, false, 4, NOP, OPT, DBG>(lds, run, first, ns, wave);

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMA -> read of D
    if (role != 4) {
      const int i0 = col_i + rm * 128, j0 = col_j + rn * 128;
      const bool sym = strip.cols == 0;
      const int64_t ld = sym ? n : strip.cols;
      const int jorg = sym ? 0 : strip.col0, jend = sym ? n : strip.col0 + strip.cols;
      // inner: the whole tile inside the matrix and (symmetric job) above the diagonal / (strip) inside the strip's columns
      const bool inner = sym ? (col_j + 256 <= n && col_i + 256 <= col_j) : (col_i + 256 <= n && col_j >= jorg && col_j + 256 <= jend);
      if (role == 1) w4_store<true, 1>(s32, ld, i0, j0, jorg, n, sym, jend, l31, hi);
      else if (role == 2) w4_store<true, 2>(s32, ld, i0, j0, jorg, n, sym, jend, l31, hi);
      else if (role == 3) w4_store<true, 3>(s32, ld, i0, j0, jorg, n, sym, jend, l31, hi);
      else if (inner) w4_store<false>(s32, ld, i0, j0, jorg, n, sym, jend, l31, hi);
      else w4_store<true>(s32, ld, i0, j0, jorg, n, sym, jend, l31, hi);
    }
  }
#ifdef PCOA_EXPERIMENTS
  if (threadIdx.x == 0) {
    const unsigned long long dc = clock64() - clk0, dr = wall_clock64() - rt0;
    if (b == 8) { g_w4_clk[0] = dc; g_w4_clk[1] = dr; }
    atomicMax(&g_w4_clk[2], dc);  // the slowest workgroup
    atomicMax(&g_w4_clk[3], dr);
  }
#endif
}

#endif  // PCOA_KBITS_W4_KERNELS

#ifdef PCOA_KBITS_W4_LAUNCHERS

#ifdef PCOA_EXPERIMENTS
int g_w4_variant = 0;  // harness knob
#endif

// Same contract as launch_gram_kbits (modes 0 / 2 / 4); 256-thread workgroups, one per CU.
// wdiag: 0 = the r04a form of diagonal tiles (the wave below the diagonal idles); 1..15 = wave roles on diagonal tiles, and in
// the even split a stage of a diagonal tile counts wdiag / 16 of another tile's; < 0 = the default (kW4DiagCost)
constexpr int kW4DiagCost = 11;
// what the even split charges a workgroup for the run it starts when it crosses into the next tile (16 = one stage of a full
// tile): 1440 = 90 stages ~ 45 us of accumulator flush (65,536 atomics) and ring refill.  Sweep at 14 % ones, two rounds
// (profiles/r05n): 0 -> 0.922 / 0.931 ms per 2^20 variants, 480 -> 0.899 / 0.902, 960 -> 0.901 / 0.887, 1440 -> 0.889 / 0.878,
// 1920 -> 0.899 / 0.896.
int g_w4_epilogue_cost = 1440;
hipError_t launch_gram_kbits_w4(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu, hipStream_t stream, int mode,
                                const int32_t* skip, GramStrip strip, int wdiag) {
  if (wdiag < 0 || wdiag > 16) wdiag = kW4DiagCost;
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
  } else if (mode == 4 || mode == 5) {
    const int64_t nwork = (int64_t)ntri * nstages;
    nblocks = std::max<int64_t>(1, std::min<int64_t>(cus, nwork / 8));
    if (nblocks >= kNumXcd) nblocks = nblocks / kNumXcd * kNumXcd;
    xcd_map = 4;
    // one eighth of the k-range per XCD: needs whole XCDs of workgroups and a k-range worth cutting
    if (mode == 5 && nblocks >= kNumXcd && nstages >= 64 * kNumXcd && strip.cols == 0) xcd_map = 5;
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
#define PCOA_LAUNCH_W4(NST_, NOP_, OPT_, DBG_)                                                                            \
  hipLaunchKernelGGL((gram_kbits_w4_kernel<NST_, NOP_, OPT_, DBG_>), grid, block, 0, stream, p, npad, nstages, n, ntile,   \
                     ntri, (int)splitk, stages_per, s32, xcd_map, skip, strip, wdiag, g_w4_epilogue_cost)
#ifdef PCOA_EXPERIMENTS
  switch (g_w4_variant) {
    case 1: PCOA_LAUNCH_W4(4, 0, 0, 0); break;   // builtin DMA, pins behind the MFMA
    case 2: PCOA_LAUNCH_W4(4, 0, 1, 0); break;   // asm DMA
    case 3: PCOA_LAUNCH_W4(4, 0, 2, 0); break;   // pins in front of the MFMA
    case 4: PCOA_LAUNCH_W4(4, 0, 3, 0); break;
    case 5: PCOA_LAUNCH_W4(4, 0, 7, 0); break;   // + one DMA piece per k-step
    case 6: PCOA_LAUNCH_W4(6, 0, 7, 0); break;
    case 7: PCOA_LAUNCH_W4(4, 2, 7, 0); break;   // s_nop 1 in front of every MFMA
    case 8: PCOA_LAUNCH_W4(4, 0, 13, 0); break;  // sched_barrier instead of pins
    case 9: PCOA_LAUNCH_W4(4, 0, 21, 0); break;  // sched_barrier behind the MFMA, input pins one gap ahead, output pins
    case 10: PCOA_LAUNCH_W4(4, 0, 45, 0); break; // sched_barrier instead of pins + one tie per k-step
    case 11: PCOA_LAUNCH_W4(4, 0, 69, 0); break; // sched_barrier behind the MFMA, output pins only
    // timing-only builds (wrong S): what is left when a part of the stage is taken out
    case 101: PCOA_LAUNCH_W4(4, 0, 69, 1); break;   // no expansion VALU
    case 102: PCOA_LAUNCH_W4(4, 0, 69, 2); break;   // no barrier
    case 104: PCOA_LAUNCH_W4(4, 0, 69, 4); break;   // no ds_read
    case 108: PCOA_LAUNCH_W4(4, 0, 69, 8); break;   // no DMA
    case 112: PCOA_LAUNCH_W4(4, 0, 69, 12); break;  // no ds_read, no DMA
    case 115: PCOA_LAUNCH_W4(4, 0, 69, 15); break;  // bare MFMAs
    case 132: PCOA_LAUNCH_W4(4, 0, 69, 32); break;  // everything, but the operand stream is one block read over and over
    case 133: PCOA_LAUNCH_W4(4, 0, 69, 33); break;  // the same without expansion VALU
    case 164: PCOA_LAUNCH_W4(4, 0, 69, 64); break;  // everything, every run from block 0: the most L2 sharing there can be
    case 139: PCOA_LAUNCH_W4(4, 0, 69, 39); break;  // the same without expansion, barrier, ds_read: MFMAs + L2-warm DMA
    case 12: PCOA_LAUNCH_W4(6, 0, 69, 0); break;
    case 13: PCOA_LAUNCH_W4(4, 0, 65, 0); break; // both DMA pieces in one gap
    default: PCOA_LAUNCH_W4(4, 0, 69, 0); break;
  }
#else
  PCOA_LAUNCH_W4(4, 0, 69, 0);
#endif
#undef PCOA_LAUNCH_W4
  return hipGetLastError();
}

#endif  // PCOA_KBITS_W4_LAUNCHERS
