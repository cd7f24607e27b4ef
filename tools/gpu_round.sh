#!/bin/bash
# One gpurun call = smoke + GPU parity tests + bench + rocprofv3 kernel trace.  Every step has its own
# timeout and log under gpurun_out/ so that one failure does not hide the others.
# usage: tools/gpu_round.sh <tag> [steps...]   steps default: smoke tests bench prof
set +e
TAG=${1:-r01}; shift
STEPS=${@:-smoke tests bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== $(date) on $(hostname); steps: $STEPS" | tee $OUT/summary.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/summary.txt
nproc >> $OUT/summary.txt
for s in $STEPS; do
  case $s in
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
      echo "smoke exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/smoke.log | tee -a $OUT/summary.txt ;;
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -rA --durations=40 > $OUT/tests.log 2>&1
      echo "tests exit $?" | tee -a $OUT/summary.txt; grep -E "passed|failed|error" $OUT/tests.log | tail -3 | tee -a $OUT/summary.txt
      grep -E "^(FAILED|ERROR)|PCoA wall" $OUT/tests.log | head -40 | tee -a $OUT/summary.txt ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
      echo "bench exit $?" | tee -a $OUT/summary.txt; cat $OUT/bench.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench.err ;;
    pmcf32)
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $OLDPWD/$OUT/pmc_f32_$c -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 2 --warmup 1 --gram-kernel f32 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/pmc_f32_$c.err )
        echo "pmc f32 $c exit $?" | tee -a $OUT/summary.txt
      done
      python tools/pmc_summary.py $OUT | tee -a $OUT/summary.txt
      cp $OUT/pmc_summary.json $OUT/pmc_summary_f32.json; find $OUT -name "*counter_collection*" -size +4M -delete ;;
    pmcx)
      # extra PMC passes: PMC_SETS="A,B,C D,E" -> one rocprofv3 run per space-separated set
      i=0
      for set in $PMC_SETS; do
        i=$((i+1))
        ( cd /tmp && timeout 600 rocprofv3 --pmc $(echo $set | tr ',' ' ') --output-format csv -d $OLDPWD/$OUT/pmc_x$i -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/pmc_x$i.err )
        echo "pmcx set $i ($set) exit $?" | tee -a $OUT/summary.txt
      done
      python tools/pmc_summary.py $OUT | tee -a $OUT/summary.txt
      cp $OUT/pmc_summary.json $OUT/pmc_summary_full.json; find $OUT -name "*counter_collection*" -size +4M -delete ;;
    counters)
      rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|^[A-Za-z_0-9]+\s" | head -400 > $OUT/counters.txt; rocprofv3 -L > $OUT/counters_full.txt 2>&1; wc -l $OUT/counters_full.txt | tee -a $OUT/summary.txt ;;
    benchab)
      # within-run A/B: AB_VAR=<env var> AB_VALUES="a b" (interleaved, two rounds)
      for t in $AB_VALUES $AB_VALUES; do
        env $AB_VAR=$t timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --pcoa-reps 1 > $OUT/bench_ab_$t.json 2>> $OUT/bench_ab.err
        python -c "import json,sys; d=json.load(open('$OUT/bench_ab_$t.json')); print('$AB_VAR=$t: value %.1f M/s, ms/step %.3f, gram %.3f ms, pack %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['gram_ms_per_step'], d['pack_ms_per_step']))" | tee -a $OUT/summary.txt
      done ;;
    abenv)
      # AB_SETS="A=1 B=2;C=3;" -> one bench run per ';'-separated environment (empty = defaults), two rounds
      for round in 1 2; do
        IFS=';' read -ra SETS <<< "$AB_SETS"
        for set in "${SETS[@]}"; do
          env $set timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --pcoa-reps 1 > $OUT/bench_abenv.json 2>> $OUT/bench_ab.err
          python -c "import json,sys; d=json.load(open('$OUT/bench_abenv.json')); print('[%s] value %.1f M/s, ms/step %.3f, gram %.3f ms, pack %.3f ms' % ('$set', d['value']/1e6, d['ms_per_step'], d['gram_ms_per_step'], d['pack_ms_per_step']))" | tee -a $OUT/summary.txt
        done
      done ;;
    testsab)
      for t in $AB_VALUES; do
        env $AB_VAR=$t timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x -k "gram or config2 or full_config2 or multi_launch or synthetic" > $OUT/tests_$t.log 2>&1
        echo "tests $AB_VAR=$t exit $?: $(tail -1 $OUT/tests_$t.log)" | tee -a $OUT/summary.txt
      done ;;
    benchf32)
      timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --gram-kernel f32 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
      echo "bench f32 exit $?" | tee -a $OUT/summary.txt; cat $OUT/bench_f32.json | tee -a $OUT/summary.txt; tail -5 $OUT/bench_f32.err ;;
    prof)
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err )
      echo "prof exit $?" | tee -a $OUT/summary.txt
      find $OUT/prof -name "*kernel_stats*" | head -3 | while read f; do echo "--- $f"; head -25 "$f" | cut -c1-220; done | tee -a $OUT/summary.txt
      # keep the merge-back small: drop the raw per-dispatch trace if it is large
      find $OUT/prof -name "*kernel_trace*" -size +8M -delete ;;
    pmclive)
      # live HBM traffic of the k-bits kernels of THIS tree -> $OUT/gram_pmc_live.json (copy to profiles/): the pipeline's
      # kernels and the serial order's, FETCH_SIZE and WRITE_SIZE in their own passes (they do not fit one)
      for tag in pipe serial; do
        for c in FETCH_SIZE WRITE_SIZE; do
          env=""; [ $tag = serial ] && env="PCOA_PIPELINE=0"
          ( cd /tmp && env $env timeout 600 rocprofv3 --pmc $c --output-format csv -d $OLDPWD/$OUT/pmclive_${tag}_$c -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pcoa-reps 1 > $OLDPWD/$OUT/pmclive_${tag}_$c.json 2> $OLDPWD/$OUT/pmclive_${tag}_$c.err )
          echo "pmclive $tag $c exit $?" | tee -a $OUT/summary.txt
        done
      done
      python tools/pmc_live.py $OUT | tee -a $OUT/summary.txt
      find $OUT -name "*counter_collection*" -size +4M -delete ;;
    pmc)
      ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_fetch -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/pmc_fetch.err )
      echo "pmc fetch exit $?" | tee -a $OUT/summary.txt
      ( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$OUT/pmc_write -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/pmc_write.err )
      echo "pmc write exit $?" | tee -a $OUT/summary.txt
      ( cd /tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OLDPWD/$OUT/pmc_sq -o pmc -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pcoa-reps 1 > /dev/null 2> $OLDPWD/$OUT/pmc_sq.err )
      echo "pmc sq exit $?" | tee -a $OUT/summary.txt
      python tools/pmc_summary.py $OUT | tee -a $OUT/summary.txt
      cp $OUT/pmc_summary.json $OUT/pmc_summary_full.json; find $OUT -name "*counter_collection*" -size +4M -delete ;;
  esac
done
echo "== done $(date)" | tee -a $OUT/summary.txt
