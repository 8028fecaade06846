#!/bin/bash
# A diagnostic build of libskyemb beside the product library: the named sources recompiled with extra defines, everything else
# linked from the product build's objects.   usage: tools/build_variant.sh NAME "-DPF_CLOCK" topk_prefilter.hip [more.hip ...]
# -> sky_embeddings_amd/libskyemb_NAME.so (load with SKYEMB_LIB=...); *.so is git-ignored and travels with gpurun.
set -e
cd "$(dirname "$0")/../sky_embeddings_amd/csrc"
name=$1; defs=$2; shift 2
make -s -j8
objs=""
skip=""
for src in "$@"; do
  o="/tmp/variant_${name}_${src%.hip}.o"
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result $defs -c "$src" -o "$o"
  objs="$objs $o"; skip="$skip ${src%.hip}.o"
  if [ -f "${src%.hip}_f16.o" ]; then
    o2="/tmp/variant_${name}_${src%.hip}_f16.o"
    /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -DSKY_F16 $defs -c "$src" -o "$o2"
    objs="$objs $o2"; skip="$skip ${src%.hip}_f16.o"
  fi
done
rest=""
for o in *.o; do case "$o" in *_m.o) continue;; esac; case " $skip " in *" $o "*) ;; *) rest="$rest $o";; esac; done   # (*_m.o: the measurement build's objects)
/opt/rocm/bin/hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o "../libskyemb_${name}.so" $objs $rest
echo "../libskyemb_${name}.so"
