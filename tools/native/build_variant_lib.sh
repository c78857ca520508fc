#!/bin/bash
# Private build of the whole library with extra compiler flags, for tools/ab_bench.sh (how a candidate change is put through the parity tests and
# an interleaved A/B before it becomes the code; round 5's rule: csrc/ carries no experiment macros between rounds — a candidate lands or is deleted):
#   bash tools/native/build_variant_lib.sh <tag> <flags...>     ->   tools/native/libddpo_hip_<tag>.so
set -e
TAG=$1; shift || { echo "usage: build_variant_lib.sh <tag> <flags...>"; exit 64; }
HERE=$(cd "$(dirname "$0")" && pwd); CS=$(cd "$HERE/../../ddpo_amd/csrc" && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
OBJS=""
for f in $CS/*.hip; do
  o=$HERE/$(basename ${f%.hip})_$TAG.o
  hipcc $FLAGS "$@" -c $f -o $o &
  OBJS="$OBJS $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $HERE/libddpo_hip_$TAG.so
echo "built $HERE/libddpo_hip_$TAG.so ($*)"
