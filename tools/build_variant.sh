#!/bin/bash
# tools/build_variant.sh <name> [-DNF_... flags]: a tuning build of the library with extra macro definitions -> build/variants/<name>.so
# (A/B experiments on one GPU box; `python tools/bench_ibrnet_kernels.py 20 build/variants/<name>.so`).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants/$name
for f in nerfool_amd/csrc/*.hip; do
  o=build/variants/$name/$(basename $f).o
  if [ "$(basename $f)" = ${NF_VARIANT_SRC:-nf_cnn.hip} ] || [ ! -f $o ]; then
    extra=""; [ "$(basename $f)" = nf_wino_bf.hip ] && extra="-fno-slp-vectorize"       # as __graft_entry__.HIPCC_EXTRA
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fPIC -std=c++17 -Iinclude -Inerfool_amd/csrc $extra "$@" -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/variants/$name.so build/variants/$name/*.o
echo built build/variants/$name.so
