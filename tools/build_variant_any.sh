#!/bin/bash
# builds rag-arc_amd/lib/librarc_var_<name>.so with extra -D flags for ONE source of csrc:
#   tools/build_variant_any.sh <source stem> <name> <flags...>
# A wrong-result switch (SCAN_HALFREAD, RARC_Q8_ABLATIONS, G256_SKIP_A1) needs -DRARC_EXPERIMENT among the flags
# (rarc_common.h refuses it otherwise); rarc_api is then rebuilt with it too, so that the library's rarc_version() announces
# a measurement build (load it with RARC_ALLOW_EXPERIMENT=1).
stem=$1; name=$2; shift; shift; src=${VARIANT_SRC:-$stem.hip}
cd $(dirname $0)/../rag-arc_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
api=../build/rarc_api.o
case " $* " in *RARC_EXPERIMENT*) /opt/rocm/bin/hipcc $FLAGS -DRARC_EXPERIMENT -c rarc_api.hip -o /tmp/var_api_$name.o && api=/tmp/var_api_$name.o;; esac
objs=""
# the object list is the Makefile's SRCS: a source added there is linked here too (pairs was missing)
for o in $(sed -n 's/^SRCS *= *//p' Makefile | sed 's/\.hip//g'); do
  if [ $o = $stem ]; then objs="$objs /tmp/var_${stem}_$name.o"; elif [ $o = rarc_api ]; then objs="$objs $api"; else objs="$objs ../build/$o.o"; fi
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o /tmp/var_${stem}_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../lib/librarc_var_$name.so
