#!/bin/bash
# Private build of the library from the kernel sources of an earlier git revision, for interleaved A/B runs on one GPU box (tools/ab_bench.sh):
#   bash tools/native/build_rev_lib.sh <git-rev> <tag> [file.hip ...]   ->   tools/native/libddpo_hip_<tag>.so
# Only the listed csrc files (default: every .hip / .h that differs from the working tree) are taken from <git-rev>; the rest are the current
# objects, so the ABI version and the struct layouts stay those lib.py expects.  Built files are git-ignored and travel with the gpurun snapshot.
set -e
REV=$1; TAG=$2; shift 2 || { echo "usage: build_rev_lib.sh <git-rev> <tag> [file.hip ...]"; exit 64; }
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); CS=$ROOT/ddpo_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
FILES="$@"
[ -n "$FILES" ] || FILES=$(cd $ROOT && git diff --name-only $REV -- ddpo_amd/csrc | grep '\.hip$' | xargs -r -n1 basename)
[ -n "$FILES" ] || { echo "no kernel source differs from $REV"; exit 1; }
TMP=$(mktemp -d $ROOT/ddpo_amd/_rev_XXXXXX)      # a sibling of csrc/: common.h reaches ../../include/ddpo_hip.h
trap "rm -rf $TMP" EXIT
for f in $(cd $ROOT && git ls-tree --name-only $REV ddpo_amd/csrc/ | xargs -n1 basename | grep -E '\.(h|hip)$'); do git -C $ROOT show $REV:ddpo_amd/csrc/$f > $TMP/$f; done
OBJS=""
for f in $FILES; do
  hipcc $FLAGS -c $TMP/$f -o $HERE/${f%.hip}_$TAG.o
  OBJS="$OBJS $HERE/${f%.hip}_$TAG.o"
done
KEEP=$(ls $CS/*.o | grep -v -E "/($(echo $FILES | sed 's/\.hip//g; s/ /|/g'))\.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $KEEP -o $HERE/libddpo_hip_$TAG.so
echo "built $HERE/libddpo_hip_$TAG.so ($FILES from $REV)"
