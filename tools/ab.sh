#!/bin/bash
# Same-box A/B of two builds of libalsmi.so (run on the GPU box from the repo root):
#   tools/ab.sh <libA.so> <libB.so> [rounds] [bench args]
# Alternates the two libraries `rounds` times so that box-to-box and drift effects cancel.
set -e
a=$1; b=$2; rounds=${3:-3}; shift; shift; shift || true
for i in $(seq 1 $rounds); do
    ALSMI_LIB=$(realpath $a) tools/bench_sweep.sh "A$i=$*"
    ALSMI_LIB=$(realpath $b) tools/bench_sweep.sh "B$i=$*"
done
