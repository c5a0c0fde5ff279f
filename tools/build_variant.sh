#!/bin/bash
# builds rag-arc_amd/lib/librarc_var_<name>.so with extra -D flags for decoder.hip: tools/build_variant.sh <name> <flags...>
name=$1; shift
cd $(dirname $0)/../rag-arc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c decoder.hip -o /tmp/dec_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../build/rarc_api.o ../build/scan_f16.o ../build/scan_q8.o ../build/quant.o ../build/finalize.o ../build/prep.o ../build/fuse.o ../build/encoder.o ../build/encoder_f32.o /tmp/dec_$name.o -o ../lib/librarc_var_$name.so
