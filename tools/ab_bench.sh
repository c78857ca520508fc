#!/bin/bash
# Interleaved A/B of bench.py on ONE GPU box in ONE gpurun call (box-to-box spread of one binary is +-3 %: nothing below 5 % can be claimed
# across calls).  Legs = private library builds (TAGS: tools/native/libddpo_hip_<tag>.so, from tools/native/build_rev_lib.sh or `make -C
# tools/native exp`; the in-tree library is always the leg `new`) x environment settings (ENVS, ';'-separated, e.g. "DDPO_ATTN_PLANES=0;
# DDPO_ATTN_PLANES=1").  The library file is swapped in the box's scratch copy only.  A sampling leg costs ~10 s of GPU budget.
#   TAGS="prev" ENVS="DDPO_X=0;DDPO_X=1" ROUNDS=2 MODE=sample bash tools/ab_bench.sh
#   PYTEST="tests/test_gpu_planes.py -k gemm" [PYTEST_TAG=<tag>] runs those tests first (on the private build <tag> if given)
mkdir -p gpurun_out; export TMPDIR=/tmp
LOG=gpurun_out/${LOG:-ab_bench.log}
L=ddpo_amd/libddpo_hip.so
cp $L /tmp/new.so
# PYTEST runs on the in-tree library, or — PYTEST_TAG=<tag> — on that private build (how an experiment build is put through the parity tests)
if [ -n "$PYTEST" ]; then
  [ -n "$PYTEST_TAG" ] && cp tools/native/libddpo_hip_$PYTEST_TAG.so $L
  echo "pytest on lib=${PYTEST_TAG:-new}: $PYTEST" | tee -a $LOG
  eval "timeout 1200 python -m pytest $PYTEST -m gpu -q -x -p no:cacheprovider" 2>&1 | tail -3 | tee -a $LOG
  cp /tmp/new.so $L
fi
case "${MODE:-sample}" in
  train) B="python bench.py --mode train --steps 6 --warmup 1 --no-cpu-baseline --no-roofline";;
  *)     B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-extra --no-roofline --no-alt-datapath-extra";;
esac
IFS=';' read -ra ENVLIST <<< "${ENVS:-}"
[ ${#ENVLIST[@]} -gt 0 ] || ENVLIST=("")
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in $TAGS new; do
    if [ $v = new ]; then cp /tmp/new.so $L; else cp tools/native/libddpo_hip_$v.so $L; fi
    for e in "${ENVLIST[@]}"; do
      line=$(env $e timeout 500 $B 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms')" 2>&1 | tail -1)
      echo "${MODE:-sample} lib=$v env=[$e] (round $round): $line" | tee -a $LOG
    done
  done
done
cp /tmp/new.so $L
