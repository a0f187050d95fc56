#!/bin/bash
# usage: ab.sh <tool.py> A B [A B ...]: runs the tool with each variant library in turn on one box
T=$1; shift
for v in "$@"; do
  cp tools/_variants/$v.so ffwm_amd/lib/libffwm_hip.so
  echo "=== variant $v"
  python $T 2>&1 | grep -v amdgpu.ids
done
