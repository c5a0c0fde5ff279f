#!/bin/bash
# builds rag-arc_amd/lib/librarc_var_<name>.so with extra -D flags for ONE source of csrc: tools/build_variant_any.sh <source stem> <name> <flags...>
stem=$1; name=$2; shift; shift
cd $(dirname $0)/../rag-arc_amd/csrc
objs=""
for o in rarc_api scan_f16 scan_q8 quant finalize prep fuse encoder encoder_f32 decoder; do
  if [ $o = $stem ]; then objs="$objs /tmp/var_${stem}_$name.o"; else objs="$objs ../build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c $stem.hip -o /tmp/var_${stem}_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../lib/librarc_var_$name.so
