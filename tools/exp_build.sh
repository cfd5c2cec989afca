#!/bin/bash
# Tuning helper: build variant libraries of ONE kernel file with extra -D flags.
# Usage: tools/exp_build.sh FILE.hip "name:-DFLAG=1 -DOTHER=2" ...   ->  glow-tts-train_amd/lib/exp/lib_<name>.so
# then: GLOWTTS_HIP_LIB=.../lib/exp/lib_<name>.so python tools/microbench_*.py        (never shipped: lib/exp is git-ignored)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/glow-tts-train_amd/csrc"
make -s >/dev/null
f=$1; shift
base=${f%.hip}
mkdir -p build/exp ../lib/exp
rm -f ../lib/exp/*.so
OTHERS=$(ls build/*.o | grep -v "build/$base.o")
# compiler, architecture and flags are the production ones (the Makefile's: incl. -fno-slp-vectorize, DESIGN.md lesson 12)
HIPCC=$(make -s print-HIPCC); ARCH=$(make -s print-ARCH); CXXFLAGS=$(make -s print-CXXFLAGS)
for v in "$@"; do
  n=${v%%:*}; fl=${v#*:}
  ( $HIPCC $CXXFLAGS $fl -c $f -o build/exp/${base}_$n.o \
    && $HIPCC -shared -fPIC --offload-arch=$ARCH -o ../lib/exp/lib_$n.so $OTHERS build/exp/${base}_$n.o ) &
done
wait
ls ../lib/exp
