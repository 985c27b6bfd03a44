#!/bin/bash
# The attribution builds of k_vote (PPF_ABL_* in ppf_match_kernels.h) into build_var/a_*.so: each leaves one class of the
# kernel's work out (or doubles it); tools/vote_variants.sh times and counts them, tools/vote_classes_summary.py prices the
# classes.  Extra flags are added to every build.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in "counted0:-DPPF_ABL_COUNTED=0" "owncell0:-DPPF_ABL_OWNCELL=0" "dsmall0:-DPPF_ABL_DIRECT_SMALL=0" \
         "dbig0:-DPPF_ABL_DIRECT_BIG=0" "dbig2:-DPPF_ABL_DIRECT_BIG=2" "aggonly:-DPPF_ABL_DIRECT_SMALL=0 -DPPF_ABL_DIRECT_BIG=0" \
         "directonly:-DPPF_ABL_COUNTED=0 -DPPF_ABL_OWNCELL=0" \
         "none:-DPPF_ABL_COUNTED=0 -DPPF_ABL_OWNCELL=0 -DPPF_ABL_DIRECT_SMALL=0 -DPPF_ABL_DIRECT_BIG=0"; do
  "$ROOT/tools/build_variant.sh" "a_${v%%:*}" ${v#*:} "$@" 2>&1 | grep -E "error" || true
done
ls "$ROOT/build_var"
