#!/bin/bash
# A/B of full library builds over every front-end flavour: profiles/tools/ab_all.sh old.so new.so [...]
# (kbench prints each library's time and the largest difference of its output from the first library's)
b=profiles/tools/build
for c in htsat htsat_i16 yamnet kaldi crnn dbmel htsat1h htsat44 kaldi44; do
  timeout -k 10 300 $b/kbench -c $c -n 100 "$@" || exit 1
done
