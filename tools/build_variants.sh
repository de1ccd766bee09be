#!/bin/bash
# tuning aid: builds engine variants skirt9_amd/lib/libpmc_<name>.so (make variant) in parallel
#   tools/build_variants.sh "pert_valu_48 -DPMC_PERTURB_VALU=48" "trim8 -DPMC_PROP_TRIM=8" ...
cd "$(dirname "$0")/.."
n=0
for v in "$@"; do
  set -- $v; name=$1; shift
  make -s variant NAME=$name DEFS="$*" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" &
  n=$((n+1)); if [ $((n % 6)) = 0 ]; then wait; fi
done
wait
ls -la skirt9_amd/lib/ | awk '{print $5, $9}'
