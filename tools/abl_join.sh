#!/bin/bash
# tools/abl_join.sh n...  -> build/abl/libfemhip_abl<n>.so with -DFEM_JOIN_ABL=<n>
for n in "$@"; do bash tools/build_variant.sh abl$n -DFEM_JOIN_ABL=$n & done; wait
