#!/bin/bash
# usage: ab_run.sh out.txt n reps what tag1 tag2 ...   (tag "cur" = the regular library); interleaved rounds
out=$1; n=$2; reps=$3; what=$4; shift 4
for round in 1 2 3; do
  for tag in "$@"; do
    if [ "$tag" = cur ]; then unset WAVERANGE_AMD_LIB; else export WAVERANGE_AMD_LIB=$PWD/waverange_amd/ab/ab_$tag.so; fi
    echo "== $tag round $round" >> $out
    python tools/prof_transform.py $n $reps $what 2>&1 | tail -n $((reps>4?8:2*reps)) >> $out || exit 1
  done
done
