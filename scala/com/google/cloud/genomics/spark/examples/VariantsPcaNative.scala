/*
 * VariantsPcaNative.scala -- the two stages of VariantsPcaDriver that run on the GPUs.
 *
 * Replaces the BODIES of
 *   VariantsPcaDriver.getSimilarityMatrix   (VariantsPca.scala:182-191: per-partition DenseMatrix[Int] + reduceByKey)
 *   VariantsPcaDriver.computePca            (VariantsPca.scala:198-231: row sums, centring, MLlib RowMatrix PCA)
 * and leaves everything else of the driver (flags, getData, filterDataset, getCallsRdd, emitResult, main) as it is.
 * The driver's members `sc` and `common` are private, so the stages live here as functions of what they need and the
 * two methods of the class become forwards (the complete patch is in INTEGRATION.md, section 1):
 *
 *   def getSimilarityMatrix(callsets: RDD[Seq[Int]]) =
 *     VariantsPcaNative.getSimilarityMatrix(sc, callsets, common.indexes.size, conf.numGpus())
 *   def computePca(handles: Array[Long]) =
 *     VariantsPcaNative.computePca(handles, common.indexes, conf.numPc())
 *
 * Deployment: one JVM per node, `--spark-master local[k]` with k >= numGpus -- a pcoa handle is a pointer of this
 * process, and the RCCL communicator needs all numGpus tasks alive at the same time (ncclCommInitRank is collective).
 */
package com.google.cloud.genomics.spark.examples

import org.apache.spark.SparkContext
import org.apache.spark.rdd.RDD

object VariantsPcaNative {

  /** Records per batch (the engine buffers them; one matrix-core launch per <= 2^20 variants). */
  val BatchRecords = 65536

  /** ... bounded so that a batch's worst-case carrier lists (every callset in every record) stay below 1 GiB of direct buffer. */
  def recordsPerBatch(size: Int): Int = math.max(1, math.min(BatchRecords.toLong, (1L << 28) / math.max(size, 1)).toInt)

  /** One batch slot: page-locked direct buffers (NativePcoa.allocPinned) that grow on demand and are re-used. */
  final class BatchSlot {
    private var idx: java.nio.ByteBuffer = null
    private var offs: java.nio.ByteBuffer = null
    private var bits: java.nio.ByteBuffer = null
    private def fit(old: java.nio.ByteBuffer, bytes: Long): java.nio.ByteBuffer = {
      if (old != null && old.capacity >= bytes) { old.clear(); old }
      else {
        if (old != null) NativePcoa.freePinned(old)
        val want = math.min(math.max(bytes + bytes / 4, 1L << 16), Int.MaxValue.toLong)
        require(want >= bytes, s"a batch buffer of $bytes bytes exceeds a direct ByteBuffer")
        val b = NativePcoa.allocPinned(want)
        require(b != null, s"pcoa_host_alloc_pinned($want) failed")
        b.order(java.nio.ByteOrder.LITTLE_ENDIAN)
      }
    }
    def idxFor(bytes: Long): java.nio.ByteBuffer = { idx = fit(idx, bytes); idx }
    def offsFor(bytes: Long): java.nio.ByteBuffer = { offs = fit(offs, bytes); offs }
    /** zero-filled: the caller ORs carrier bits in */
    def bitsFor(bytes: Long): java.nio.ByteBuffer = {
      bits = fit(bits, bytes)
      var i = 0
      val n8 = (bytes / 8).toInt
      while (i < n8) { bits.putLong(8 * i, 0L); i += 1 }
      i = 8 * n8
      while (i < bytes) { bits.put(i, 0.toByte); i += 1 }
      bits
    }
    def free(): Unit = {
      Seq(idx, offs, bits).filter(_ != null).foreach(NativePcoa.freePinned)
      idx = null; offs = null; bits = null
    }
  }

  /**
   * getSimilarityMatrix.  The RDD[Seq[Int]] of getCallsRdd (VariantsPca.scala:153-168) is REpartitioned to exactly one
   * partition per GPU (coalesce cannot raise a partition count: with fewer partitions than GPUs fewer than nGpus tasks
   * would enter the collective commInit and the job would hang); each task streams its records to its GPU -- sparse
   * batches as CSR carrier lists (pcoa_accumulate_calls_ex, page-locked + asynchronous), dense ones as carrier bitsets
   * (pcoa_accumulate_bits) --, then the partial N x N matrices are summed over xGMI (== reduceByKey(_ + _), :190).  All nGpus tasks must
   * run at the same time (they meet in an RCCL collective): the executor needs >= nGpus task slots, which is checked
   * up front.  Returns one engine handle per GPU; every one of them holds the full S in HBM.  A task that fails releases
   * its engine before the exception leaves it.
   */
  def getSimilarityMatrix(sc: SparkContext, callsets: RDD[Seq[Int]], size: Int, nGpus: Int): Array[Long] = {
    val uid = sc.broadcast(NativePcoa.commUniqueId()) // rank 0's RCCL id, shipped by Spark
    require(uid.value != null, "pcoa_comm_unique_id failed")
    val slots = concurrentTaskSlots(sc)
    require(nGpus == 1 || slots >= nGpus,
      s"$nGpus GPU tasks meet in one RCCL collective but only $slots tasks can run at the same time: they would wait for each other forever")
    // exactly one partition per GPU: coalesce (no shuffle) when there are enough partitions, a shuffle only when there are too few
    val perGpu = if (callsets.getNumPartitions >= nGpus) callsets.coalesce(nGpus) else callsets.repartition(nGpus)
    perGpu.mapPartitionsWithIndex { (rank, callsInPartition) =>
      val ctx = NativePcoa.create(size, rank, NativePcoa.FlagDefault)
      // Two batch slots of page-locked buffers, re-used for the whole partition (r06): batch k+1 is built in one slot while
      // batch k -- handed over with CallsPinned | CallsAsync: the DMA engine reads the slot in place, the call only queues --
      // travels and is scattered from the other.  A slot is written again only after a synchronising call (pcoa.h: the arrays
      // of an asynchronous call stay untouched until then), i.e. one NativePcoa.sync per two batches.  (r05: two fresh
      // allocateDirect buffers per batch -- zero-filled by the JVM, freed by the GC -- and the synchronous accumulateCalls.)
      val batchSlots = Array(new BatchSlot, new BatchSlot)
      val words = (size + 31) / 32
      var k = 0
      try {
      callsInPartition.grouped(recordsPerBatch(size)).foreach { batch =>
        val slot = batchSlots(k & 1)
        if (k >= 2 && (k & 1) == 0) NativePcoa.check(ctx, NativePcoa.sync(ctx)) // both slots are ours again
        k += 1
        val nnz = batch.iterator.map(_.size.toLong).sum
        // dense batch (mean carrier list longer than N / 32 entries): one bit per (variant, callset) is fewer bytes than the
        // list -- 316 B against 1,356 B per variant for a 1000-Genomes-like cohort -- pcoa_accumulate_bits.  A list that names a
        // callset twice counts with multiplicity in the reference's double loop (:187), which a bitset cannot express: such a
        // batch goes over as lists (the engine then takes its int8 path).
        var asBits = nnz > batch.size.toLong * words
        if (asBits) {
          val bits = slot.bitsFor(4L * words * batch.size)
          var r = 0
          batch.foreach { calls =>
            val base = 4 * words * r
            calls.foreach { c =>
              // an index outside [0, size) is the NoSuchElementException of mapping(call.callsetId) (:59): a bitset cannot carry it
              if (c < 0 || c >= size) throw new NoSuchElementException(s"callset index $c outside [0, $size)")
              val at = base + 4 * (c >>> 5)
              val old = bits.getInt(at)
              if ((old & (1 << (c & 31))) != 0) asBits = false
              bits.putInt(at, old | (1 << (c & 31)))
            }
            r += 1
          }
          if (asBits) NativePcoa.check(ctx, NativePcoa.accumulateBits(ctx, bits, batch.size, words)) // host rows: consumed on return
        }
        if (!asBits) {
          val offs = slot.offsFor(8L * (batch.size + 1))
          val idx = slot.idxFor(4L * math.max(nnz, 1L))
          var o = 0L
          offs.putLong(0L)
          batch.foreach { calls =>
            calls.foreach(c => idx.putInt(c))
            o += calls.size
            offs.putLong(o)
          }
          // an index outside [0, size) is the NoSuchElementException of mapping(call.callsetId) (:59): reported by this call
          // or by the next synchronising one, S unchanged by the offending batches
          NativePcoa.check(ctx, NativePcoa.accumulateCallsEx(ctx, idx, offs, batch.size, NativePcoa.CallsPinned | NativePcoa.CallsAsync))
        }
      }
      NativePcoa.check(ctx, NativePcoa.gramFinalize(ctx))
      if (nGpus > 1) {
        val comm = NativePcoa.commInit(ctx, uid.value, rank, nGpus)
        if (comm == 0L) throw new IllegalStateException(NativePcoa.lastError(ctx))
        if (rank == 0) println(s"RCCL communicator over ${NativePcoa.commCount(comm)} of $nGpus ranks.")
        try NativePcoa.check(ctx, NativePcoa.gramAllreduce(ctx, comm))
        finally NativePcoa.commDestroy(comm)
      }
      } catch {
        case e: Throwable =>
          NativePcoa.destroy(ctx) // the handle never reaches the driver: release the GPU memory here (it also drains the queue)
          batchSlots.foreach(_.free())
          throw e
      }
      batchSlots.foreach(_.free()) // gramFinalize / the all-reduce synchronised: nothing reads the slots any more
      Iterator((rank, ctx))
    }.collect().sortBy(_._1).map(_._2)
  }

  /**
   * Tasks that can run AT THE SAME TIME: cores of the executors registered now (local[k]: k) over spark.task.cpus.
   * sc.defaultParallelism is not that number: spark.default.parallelism overrides it, and on a cluster it counts cores that
   * may belong to executors still starting (ADVICE r03).  (On Spark >= 2.4 a barrier stage -- callsets.barrier().mapPartitions
   * -- gives the gang scheduling this collective needs without the check; the reference builds against 1.6.1.)
   */
  def concurrentTaskSlots(sc: SparkContext): Int = {
    val cpusPerTask = math.max(1, sc.getConf.getInt("spark.task.cpus", 1))
    val localK = "local\\[(\\d+)(?:,\\s*\\d+)?\\]".r
    val cores = sc.master match {
      case "local" => 1
      case m if m.startsWith("local[*") => Runtime.getRuntime.availableProcessors
      case localK(k) => k.toInt
      case _ =>
        val executors = math.max(1, sc.getExecutorMemoryStatus.size - 1) // the driver's block manager is one entry
        executors * math.max(1, sc.getConf.getInt("spark.executor.cores", 1))
    }
    cores / cpusPerTask
  }

  /**
   * computePca on rank 0's engine: row sums, double centring and the numPc principal components on the GPU.
   * Prints the reference's "Non zero rows" line (:208) and returns its (callsetId, PC1, PC2) tuples (:228-230);
   * like the reference it reads components 0 and 1 unconditionally.
   */
  def computePca(handles: Array[Long], indexes: Map[String, Int], numPc: Int): Seq[(String, Double, Double)] = {
    val n = indexes.size
    val ctx = handles(0)
    val comps = NativePcoa.direct(8L * n * numPc)
    val lam = NativePcoa.direct(8L * numPc)
    val nz = new Array[Int](1)
    NativePcoa.check(ctx, NativePcoa.compute(ctx, numPc, comps, lam, nz)) // numPc outside (0, n]: IllegalArgumentException
    println(s"Non zero rows in matrix: ${nz(0)} / $n.")
    val array = comps.asDoubleBuffer() // N x numPc column-major == pca.toArray (:227)
    val reverse = indexes.map(_.swap)
    for (i <- 0 until n) yield (reverse(i), array.get(i), array.get(i + n))
  }

  /** reportIoStats' line for the native stages, and release of the engines (VariantsPcaDriver.stop, :283-285). */
  def reportAndStop(handles: Array[Long]): Unit = {
    if (handles.nonEmpty) {
      val t = NativePcoa.direct(24)
      if (NativePcoa.timings(handles(0), t) == NativePcoa.Ok) {
        val d = t.asDoubleBuffer()
        println(f"Variants accumulated: ${d.get(0).toLong}%d; Gram kernels ${1e3 * d.get(1)}%.3f ms; PCoA ${1e3 * d.get(2)}%.3f ms")
      }
    }
    handles.foreach(NativePcoa.destroy)
  }
}
