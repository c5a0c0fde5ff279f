#!/bin/bash
# builds rag-arc_amd/lib/librarc_var_<name>.so with extra -D flags for scan_f16.hip: tools/build_variant_scan.sh <name> <flags...>
name=$1; shift
cd $(dirname $0)/../rag-arc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c scan_f16.hip -o /tmp/scan_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../build/rarc_api.o /tmp/scan_$name.o ../build/scan_q8.o ../build/quant.o ../build/finalize.o ../build/prep.o ../build/fuse.o ../build/encoder.o ../build/encoder_f32.o ../build/decoder.o -o ../lib/librarc_var_$name.so
