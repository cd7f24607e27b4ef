/*
 * NativePcoa.scala -- JNI view of include/pcoa.h (libpcoa_hip.so) for the Scala host.
 *
 * SURVEY.md 8(b) caller (3), 8(f) rank 4.  One @native per entry of jni/pcoa_jni.cpp, which forwards each of them
 * to exactly one pcoa_* call.  Buffers are direct, little-endian java.nio.ByteBuffers (zero copy); a ctx or an RCCL
 * communicator is a Long.  Status codes are the pcoa_status values of include/pcoa.h (0 = ok); `check` turns them into
 * the exception the reference would have thrown at that point (INTEGRATION.md, "Error mapping").
 *
 * Source only in this repository: the build image has no JDK / scalac.  tests/jni_replay.cpp drives the same natives,
 * in the order VariantsPcaNative uses them, through a stub JNIEnv.
 */
package com.google.cloud.genomics.spark.examples

import java.nio.{ByteBuffer, ByteOrder}

object NativePcoa {
  System.loadLibrary("pcoa_jni") // jni/libpcoa_jni.so, which links libpcoa_hip.so

  // pcoa_status (include/pcoa.h)
  val Ok = 0
  val ErrInvalidArg = -1
  val ErrNoDevice = -2
  val ErrIndexRange = -5
  val ErrNotConverged = -7

  // flags for create (include/pcoa.h); 0 = auto: MX-FP4 matrix cores for binary tiles, int8 for multiplicities
  val FlagDefault = 0
  val FlagEigHouseholder = 0x20

  /** flags of accumulateCallsEx (pcoa.h: PCOA_CALLS_*) */
  val CallsDevicePtr = 1
  val CallsPinned = 2
  val CallsAsync = 4
  val BedHostAsync = 2 // PCOA_BED_HOST_ASYNC: page-locked .bed rows are only queued (untouched until the second later call returned, or sync)

  @native def create(nSamples: Int, device: Int, flags: Int): Long // pcoa_create; throws IllegalStateException
  @native def destroy(ctx: Long): Unit // pcoa_destroy
  @native def lastError(ctx: Long): String // pcoa_last_error
  @native def reset(ctx: Long): Int // pcoa_reset
  @native def accumulateCalls(ctx: Long, sampleIdx: ByteBuffer, rowOffsets: ByteBuffer, nVariants: Long): Int // pcoa_accumulate_calls
  @native def accumulateCallsEx(ctx: Long, sampleIdx: ByteBuffer, rowOffsets: ByteBuffer, nVariants: Long, flags: Int): Int // pcoa_accumulate_calls_ex
  @native def sync(ctx: Long): Int // pcoa_sync
  @native def gramReduceFrom(dst: Long, src: Long): Int // pcoa_gram_reduce_from (two engines of this JVM)
  @native def accumulateBits(ctx: Long, bits: ByteBuffer, nVariants: Long, ldWords: Long): Int // pcoa_accumulate_bits
  @native def allocPinned(bytes: Long): ByteBuffer // pcoa_host_alloc_pinned as a direct buffer (null on failure); set order(LITTLE_ENDIAN)
  @native def freePinned(buf: ByteBuffer): Int // pcoa_host_free_pinned; the buffer is dead afterwards
  @native def accumulatePlinkBed(ctx: Long, rows: ByteBuffer, nVariants: Long, rowBytes: Long, refIsA1: Int, mode: Int): Int // pcoa_accumulate_plink_bed (mode: 0 host, 1 device address, BedHostAsync)
  @native def gramFinalize(ctx: Long): Int // pcoa_gram_finalize
  @native def commUniqueId(): Array[Byte] // pcoa_comm_unique_id (128 bytes; null on failure)
  @native def commInit(ctx: Long, id: Array[Byte], rank: Int, nRanks: Int): Long // pcoa_comm_init (0 on failure)
  @native def commDestroy(comm: Long): Int // pcoa_comm_destroy
  @native def commCount(comm: Long): Int // pcoa_comm_count: ranks of the communicator (negative: a pcoa_status)
  @native def gramAllreduce(ctx: Long, comm: Long): Int // pcoa_gram_allreduce_rccl
  @native def gramRead(ctx: Long, outNxN: ByteBuffer): Int // pcoa_gram_read_i64
  @native def gramLoad(ctx: Long, inNxN: ByteBuffer): Int // pcoa_gram_load_i64
  @native def compute(ctx: Long, numPc: Int, components: ByteBuffer, eigenvalues: ByteBuffer, nonZeroRows: Array[Int]): Int // pcoa_compute
  @native def timings(ctx: Long, out3Doubles: ByteBuffer): Int // pcoa_get_timings

  /** A direct little-endian buffer: what every native above expects. */
  def direct(bytes: Long): ByteBuffer = {
    require(bytes >= 0 && bytes <= Int.MaxValue, s"direct buffer of $bytes bytes")
    ByteBuffer.allocateDirect(bytes.toInt).order(ByteOrder.LITTLE_ENDIAN)
  }

  /** Status -> the exception of the reference at the same point. */
  def check(ctx: Long, rc: Int): Unit = rc match {
    case Ok => ()
    case ErrIndexRange => // mapping(call.callsetId) / Breeze bounds check (VariantsPca.scala:59, :188)
      throw new NoSuchElementException(lastError(ctx))
    case ErrInvalidArg => // e.g. MLlib's require(k > 0 && k <= n)
      throw new IllegalArgumentException(lastError(ctx))
    case _ =>
      throw new IllegalStateException(s"pcoa status $rc: ${lastError(ctx)}")
  }
}
