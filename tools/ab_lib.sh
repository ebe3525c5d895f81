#!/bin/bash
# A/B two builds of the library on the SAME box: tools/ab_lib.sh base.so new.so [rounds]
# (copies each over vit_som_amd/libvitsom_hip.so in turn and times the step; restores `new` last)
set -e
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    cp "$L" vit_som_amd/libvitsom_hip.so
    echo -n "$(basename $L): "
    python tools/ab_step.py side_stream=1 15 2 | tail -1
  done
done
cp "$B" vit_som_amd/libvitsom_hip.so
