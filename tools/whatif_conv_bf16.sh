#!/bin/bash
# What-if builds of the bf16 LDS-DMA convolution kernel (csrc/conv_bf16.hip, LEC_BF_DBG): where do a layer's microseconds go?
#   dbg1: no operand traffic after the ring's first fill (LDS reads + MFMAs + barriers + epilogue alone)
#   dbg2: the operand traffic alone (DMA requests, waits, barriers, epilogue; no LDS reads, no MFMAs)
# Both compute WRONG results by design and are never the product: they are built into variants/ (git-ignored) and selected with LEC_LIB_PATH.
#   tools/whatif_conv_bf16.sh build          (here, no GPU needed)
#   tools/whatif_conv_bf16.sh run [args]     (on the GPU box: bench_conv_bf16.py --no-lib per build -> gpurun_out/whatif/)
set -e
cd "$(dirname "$0")/.."
C=learning_embeddings_amd/csrc
case "$1" in
build)
  mkdir -p variants
  for d in 1 2; do
    /opt/rocm/bin/hipcc -DLEC_BF_DBG=$d -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function \
      -c $C/conv_bf16.hip -o variants/conv_bf16_dbg$d.o
    objs=$(ls $C/*.o | grep -v conv_bf16.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs variants/conv_bf16_dbg$d.o -ldl -o variants/liblecone_dbg$d.so
  done ;;
run)
  shift
  mkdir -p gpurun_out/whatif
  timeout 300 python tools/bench_conv_bf16.py --no-lib "$@" > gpurun_out/whatif/product.txt 2>&1
  for d in 1 2; do
    LEC_LIB_PATH=$PWD/variants/liblecone_dbg$d.so timeout 300 python tools/bench_conv_bf16.py --no-lib "$@" > gpurun_out/whatif/dbg$d.txt 2>&1
  done ;;
*) echo "usage: $0 build | run [bench_conv_bf16.py args]"; exit 2 ;;
esac
