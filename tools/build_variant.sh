#!/bin/bash
# Build marl_dmfb_amd/lib/libdmfb_vec_<name>.so (droplet counts 4 and 10 only) from the sources of a git revision or of
# the working tree (rev = WORK), for same-box A/B timing with tools/ab_observe.py.
set -eo pipefail
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=/tmp/variant_$NAME; rm -rf $TMP; mkdir -p $TMP/marl_dmfb_amd/csrc $TMP/include
if [ "$REV" = WORK ]; then
  cp $ROOT/marl_dmfb_amd/csrc/dmfb_* $TMP/marl_dmfb_amd/csrc/; cp $ROOT/include/dmfb_vec.h $TMP/include/
else
  for f in marl_dmfb_amd/csrc/dmfb_kernels.h marl_dmfb_amd/csrc/dmfb_vec.hip marl_dmfb_amd/csrc/dmfb_vec_n.hip include/dmfb_vec.h; do git -C $ROOT show $REV:$f > $TMP/$f; done
fi
cd $TMP/marl_dmfb_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-bitwise-instead-of-logical -DDMFB_STAMPS_ONLY_N $EXTRA_FLAGS"
NS="4 10"
grep -q DMFB_STAMPS_ONLY_N dmfb_vec.hip || NS="1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16"  # older sources dispatch to all counts
hipcc $FLAGS -c -o vec.o dmfb_vec.hip &
J=1; OBJS=""
for n in $NS; do
  hipcc $FLAGS -DDMFB_TU_N=$n -c -o n$n.o dmfb_vec_n.hip & OBJS="$OBJS n$n.o"; J=$((J+1))
  if [ $J -ge 8 ]; then wait; J=0; fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/marl_dmfb_amd/lib/libdmfb_vec_$NAME.so vec.o $OBJS
echo built libdmfb_vec_$NAME.so from $REV
