#!/bin/bash
# A/B on one box, interleaved rounds: ab_run.sh OUT ROUNDS "COMMAND" TAG1 TAG2 ...
# (TAG "cur" = the regular library, any other = waverange_amd/ab/ab_TAG.so from tools/ab_build.py)
out=$1; rounds=$2; cmd=$3; shift 3
for round in $(seq 1 $rounds); do
  for tag in "$@"; do
    if [ "$tag" = cur ]; then unset WAVERANGE_AMD_LIB; else export WAVERANGE_AMD_LIB=$PWD/waverange_amd/ab/ab_$tag.so; fi
    echo "== $tag round $round" >> $out
    $cmd >> $out 2>&1 || exit 1
  done
done
