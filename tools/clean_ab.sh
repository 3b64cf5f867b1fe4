#!/bin/bash
# clean_server(s) on the warm 10 M x 1 024 table (one node / 10 % of the nodes) with the whole-kilobyte write-back of k_clean from
# more than 0 / 16 (the product) / 32 / 64 (= never: only the changed 16-byte vectors) evicting lanes per wave.  Lab build.
# Usage: tools/clean_ab.sh <tag>
TAG=${1:-round6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
( echo "{"
  for ff in 0 16 32 64 16; do
    echo "\"full_from_$ff$( [ -f /tmp/.clean_ab_second ] && echo '#2' )\": $(RIO_GP_CLEAN_FULL_FROM=$ff timeout 120 python tools/clean_probe.py | tail -1),"
    [ $ff = 64 ] && touch /tmp/.clean_ab_second
  done
  rm -f /tmp/.clean_ab_second
  echo "\"what\": \"tools/clean_probe.py per setting of RIO_GP_CLEAN_FULL_FROM (lab build), wall clock of the synchronous call\"}" ) > $OUT/${TAG}_clean_ab.json
cat $OUT/${TAG}_clean_ab.json
