#!/bin/bash
# run-to-run spread of the GenEO block iteration count on the elasticity pencil (device Cholesky: atomics make T differ in the last bits)
# usage: [PROBLEM=dg|elasticity] bash tools/geneo_variability.sh <runs> [env assignments ...]
N=$1; shift
for i in $(seq 1 $N); do
  env "$@" python bench_convdiff.py --problem ${PROBLEM:-elasticity} --cpu-iters 0 --no-solve --steps 5 2>&1 >/dev/null | grep -o "GenEO [0-9]* block iterations\|did not converge in [0-9]* block" | tr '\n' ' '
done
echo
