#!/bin/bash
# Same-box A/B of two builds of the library: $1 = the other libbayeslm_hip.so, rest = command.  Runs the command with the
# in-tree library, then with the other one swapped in, twice each (boxes differ by 1-3 % between gpurun calls).
set -e
OTHER=$1; shift
LIB=bayeslms_amd/libbayeslm_hip.so
cp $LIB /tmp/lib_new.so
trap 'cp /tmp/lib_new.so $LIB' EXIT  # whatever happens, the in-tree library is the one that was there
for rep in 1 2; do
  cp /tmp/lib_new.so $LIB; echo "== new (rep $rep)"; "$@"
  cp $OTHER $LIB; echo "== old (rep $rep)"; "$@"
done
cp /tmp/lib_new.so $LIB
