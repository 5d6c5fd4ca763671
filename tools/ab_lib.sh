#!/bin/bash
# A/B of two builds of libsbv.so with one tool, a fresh process per run, alternating: ab_lib.sh <variant.so> <reps> <tool.py> [ENV=VAL ...]
# (the default library first, then the variant through SBV_LIB; what each run prints last is kept).  AB_ARGS="..." = arguments of the tool.
V=$1; R=$2; T=$3; shift 3
for rep in $(seq 1 "$R"); do
  for lib in default "$V"; do
    if [ "$lib" = default ]; then out=$(env "$@" timeout 300 python "$T" ${AB_ARGS:-} 2>/dev/null | tail -1); else out=$(env "$@" SBV_LIB="$lib" timeout 300 python "$T" ${AB_ARGS:-} 2>/dev/null | tail -1); fi
    echo "{\"lib\": \"$lib\", \"rep\": $rep, \"env\": \"$*\", \"result\": $out}"
  done
done
