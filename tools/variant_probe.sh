#!/bin/bash
# tools/variant_probe.sh [probe args...] -- runs tools/small_frame_probe.py once per experimental build of the
# library found in build/variants/librc_*.so (each copied over the tree's library in turn; the tree's own
# library is put back at the end).  For A/B experiments on one GPU box.
ORIG=cudaraytracing_amd/librc_hip.so
cp $ORIG /tmp/librc_orig.so
for lib in build/variants/librc_*.so; do
  echo "=== $lib"
  cp $lib $ORIG
  timeout -k 10 200 python tools/small_frame_probe.py "$@" || break
done
cp /tmp/librc_orig.so $ORIG
