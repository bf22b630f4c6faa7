#!/bin/bash
# An experiment build of the library from a patched COPY of csrc/ (the tree itself is never patched):
#   tools/experiments/build_variant.sh nodiv -DFTK_CLEAVE_NODIV   ->  finaletoolkit_amd/libftk_cv_nodiv.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
W=$(mktemp -d /tmp/ftk_variant_XXXXXX)
mkdir -p $W/repo/finaletoolkit_amd $W/repo/include
cp -r $R/finaletoolkit_amd/csrc $W/repo/finaletoolkit_amd/csrc
cp $R/include/*.h $W/repo/include/
rm -rf $W/repo/finaletoolkit_amd/csrc/build
(cd $W/repo && patch -p1 -s < $R/tools/experiments/kernel_experiment_switches.patch)
make -C $W/repo/finaletoolkit_amd/csrc -j8 EXTRA_CXXFLAGS="$*" OUT=$R/finaletoolkit_amd/libftk_cv_$name.so > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
echo built $R/finaletoolkit_amd/libftk_cv_$name.so with "$*"
