#!/bin/bash
# builds rag-arc_amd/lib/librarc_var_<name>.so with extra -D flags for decoder.hip: tools/build_variant.sh <name> <flags...>
name=$1; shift
exec $(dirname $0)/build_variant_any.sh decoder $name "$@"
