#!/bin/bash
# Same-box A/B of the conv front-end library: builds marl_dmfb_amd/lib/libcrnn_ops_<tag>.so from the working tree with EXTRA_FLAGS
# (e.g. EXTRA_FLAGS=-DCRNN_NO_ROWLANE tools/ab_conv.sh old); run with MARL_DMFB_VARIANT_CRNN_OPS=_<tag> python tools/bench_conv.py
set -eo pipefail
TAG=$1; ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT/marl_dmfb_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $EXTRA_FLAGS"
hipcc $FLAGS -c -o /tmp/crnn_ops_$TAG.o crnn_ops.hip &
hipcc $FLAGS -c -o /tmp/gru_ops_$TAG.o gru_ops.hip &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcrnn_ops_$TAG.so /tmp/crnn_ops_$TAG.o /tmp/gru_ops_$TAG.o
echo built libcrnn_ops_$TAG.so
