#!/bin/bash
# Same-box comparison of several builds of libalsmi.so (run on the GPU box from the repo root):
#   tools/abn.sh <rounds> "<bench args>" <lib1.so> <lib2.so> ...
# Alternates the libraries `rounds` times so that box-to-box and drift effects cancel.
set -e
rounds=$1; args=$2; shift; shift
for i in $(seq 1 $rounds); do
    for lib in "$@"; do
        ALSMI_LIB=$(realpath $lib) tools/bench_sweep.sh "$(basename $lib .so)$i=$args"
    done
done
