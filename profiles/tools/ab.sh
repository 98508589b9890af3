#!/bin/bash
# A/B of library builds over the front-end flavours: profiles/tools/ab.sh "flavours" lib1.so lib2.so ...
# (kbench prints each library's best/mean time and the largest difference of its output from the first library's)
flav=$1; shift
for c in $flav; do
  echo "== $c"
  timeout -k 10 200 profiles/tools/build/kbench -c $c -n 100 "$@" || exit 1
done
