#!/bin/bash
# Soak: the knob sweep of test_alternative_code_paths_agree N times, each under its own timeout (a hang is reported, not waited for).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=${1:-12}
for i in $(seq 1 $N); do
  s=$(date +%s)
  timeout 120 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -k "test_alternative_code_paths_agree" -o faulthandler_timeout=90 > /tmp/soak_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - s )) s: $(tail -1 /tmp/soak_$i.log)"
  if [ $rc -ne 0 ]; then tail -60 /tmp/soak_$i.log; fi
done
