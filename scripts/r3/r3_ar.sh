#!/bin/bash
# end-of-round check as the driver runs it: smoke(), then the GPU tier twice (stops at the first failing step)
OUT=gpurun_out/r3ar; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1 || { tail -20 $OUT/smoke.txt; exit 1; }
tail -1 $OUT/smoke.txt
for i in 1 2; do
  timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 180 --timeout-method thread > $OUT/pytest_$i.txt 2>&1 || { tail -30 $OUT/pytest_$i.txt; exit 1; }
  echo "run $i: $(tail -1 $OUT/pytest_$i.txt)"
done
