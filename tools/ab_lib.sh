#!/bin/bash
# Same-box A/B of two builds of the library: $1 = the other libbayeslm_hip.so, rest = command.  Runs the command with the
# in-tree library, then with the other one selected through BLM_LIB (bayeslms_amd/_lib.py), twice each (boxes differ by
# 1-3 % between gpurun calls).  The tracked in-tree file is never overwritten.
set -e
OTHER="$(readlink -f "$1")"; shift
[ -f "$OTHER" ] || { echo "ab_lib.sh: no such library: $OTHER" >&2; exit 2; }
for rep in 1 2; do
  echo "== new (rep $rep)"; env -u BLM_LIB "$@"
  echo "== old (rep $rep)"; BLM_LIB="$OTHER" "$@"
done
