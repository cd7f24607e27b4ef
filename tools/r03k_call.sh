mkdir -p gpurun_out/r03k
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x --durations=15 > gpurun_out/r03k/tests.log 2>&1; echo "tests exit $?" >> gpurun_out/r03k/tests.log )
tail -30 gpurun_out/r03k/tests.log
